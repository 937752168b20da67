// wire.h — the reference's engine <-> runner wire format in C++, header only.
//   framing   src/runner/mod.rs:246-295 — u32-LE payload length, payload, then the RECEIVER writes one ack byte 0x01
//   payload   `MessageType` (src/runner/mod.rs:169-244): serde_json for `Init` (the runner reads it with use_json = true,
//             src/runner/runner.rs:73), bincode 1.x default options afterwards (little endian, usize / u64 / isize = 8 bytes,
//             u32 / enum variant index = 4 bytes, bool = 1 byte, Option = 1 tag byte + value, Vec / String = u64 length +
//             elements, f32 = 4 bytes, structs / tuples = fields in declaration order, HashMap = u64 length + pairs)
//   structs   Sequence / DecodeSequence (src/core/sequence.rs:32-62), SamplingParams (src/utils/config.rs:505-537)
// Same layouts as vllm_rs_amd/wire.py, whose known-answer bytes (tests/test_wire.py) pin both.  Only the variants the
// forward-pass runner exchanges are coded; the others decode to an error carrying the variant name.
#pragma once
#include <stdint.h>
#include <string.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

namespace vra_wire {

// ---------------------------------------------------------------- JSON (serde_json subset: what Init and config.json hold)
struct Json {
  enum Type { Null, Bool, Num, Str, Arr, Obj } t = Null;
  bool b = false;
  double n = 0;
  std::string s;
  std::vector<Json> a;
  std::vector<std::pair<std::string, Json>> o;
  const Json* get(const char* k) const {
    if (t != Obj) return nullptr;
    for (auto& kv : o)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  bool has(const char* k) const {
    const Json* j = get(k);
    return j && j->t != Null;
  }
  double num(const char* k, double d) const {
    const Json* j = get(k);
    return j && j->t == Num ? j->n : d;
  }
  long long i64(const char* k, long long d) const {
    const Json* j = get(k);
    return j && j->t == Num ? (long long)(j->n < 0 ? j->n - 0.5 : j->n + 0.5) : d;
  }
  bool boolean(const char* k, bool d) const {
    const Json* j = get(k);
    return j && j->t == Bool ? j->b : d;
  }
  std::string str(const char* k, const std::string& d) const {
    const Json* j = get(k);
    return j && j->t == Str ? j->s : d;
  }
};
struct JsonParser {
  const char *p, *e;
  std::string err;
  explicit JsonParser(const std::string& s) : p(s.data()), e(s.data() + s.size()) {}
  void ws() {
    while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++;
  }
  bool fail(const char* m) {
    if (err.empty()) err = m;
    return false;
  }
  static void utf8(std::string& s, unsigned c) {
    if (c < 0x80) s += (char)c;
    else if (c < 0x800) s += (char)(0xC0 | (c >> 6)), s += (char)(0x80 | (c & 0x3F));
    else if (c < 0x10000) s += (char)(0xE0 | (c >> 12)), s += (char)(0x80 | ((c >> 6) & 0x3F)), s += (char)(0x80 | (c & 0x3F));
    else s += (char)(0xF0 | (c >> 18)), s += (char)(0x80 | ((c >> 12) & 0x3F)), s += (char)(0x80 | ((c >> 6) & 0x3F)), s += (char)(0x80 | (c & 0x3F));
  }
  bool hex4(unsigned* v) {
    if (e - p < 4) return fail("truncated \\u escape");
    unsigned x = 0;
    for (int i = 0; i < 4; i++) {
      const char c = *p++;
      x <<= 4;
      if (c >= '0' && c <= '9') x |= c - '0';
      else if (c >= 'a' && c <= 'f') x |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') x |= c - 'A' + 10;
      else return fail("bad \\u escape");
    }
    *v = x;
    return true;
  }
  bool string(std::string* out) {
    if (p >= e || *p != '"') return fail("expected string");
    p++;
    while (p < e && *p != '"') {
      if (*p == '\\') {
        if (++p >= e) return fail("truncated escape");
        const char c = *p++;
        switch (c) {
          case '"': *out += '"'; break;
          case '\\': *out += '\\'; break;
          case '/': *out += '/'; break;
          case 'b': *out += '\b'; break;
          case 'f': *out += '\f'; break;
          case 'n': *out += '\n'; break;
          case 'r': *out += '\r'; break;
          case 't': *out += '\t'; break;
          case 'u': {
            unsigned c1;
            if (!hex4(&c1)) return false;
            if (c1 >= 0xD800 && c1 < 0xDC00 && e - p >= 6 && p[0] == '\\' && p[1] == 'u') {
              p += 2;
              unsigned c2;
              if (!hex4(&c2)) return false;
              c1 = 0x10000 + ((c1 - 0xD800) << 10) + (c2 - 0xDC00);
            }
            utf8(*out, c1);
            break;
          }
          default: return fail("bad escape");
        }
      } else {
        *out += *p++;
      }
    }
    if (p >= e) return fail("unterminated string");
    p++;
    return true;
  }
  bool value(Json* j, int depth = 0) {
    if (depth > 64) return fail("nesting too deep");
    ws();
    if (p >= e) return fail("unexpected end");
    if (*p == '{') {
      j->t = Json::Obj;
      p++;
      ws();
      if (p < e && *p == '}') return p++, true;
      for (;;) {
        ws();
        std::string k;
        if (!string(&k)) return false;
        ws();
        if (p >= e || *p != ':') return fail("expected ':'");
        p++;
        Json v;
        if (!value(&v, depth + 1)) return false;
        j->o.emplace_back(std::move(k), std::move(v));
        ws();
        if (p < e && *p == ',') {
          p++;
          continue;
        }
        if (p < e && *p == '}') return p++, true;
        return fail("expected ',' or '}'");
      }
    }
    if (*p == '[') {
      j->t = Json::Arr;
      p++;
      ws();
      if (p < e && *p == ']') return p++, true;
      for (;;) {
        Json v;
        if (!value(&v, depth + 1)) return false;
        j->a.push_back(std::move(v));
        ws();
        if (p < e && *p == ',') {
          p++;
          continue;
        }
        if (p < e && *p == ']') return p++, true;
        return fail("expected ',' or ']'");
      }
    }
    if (*p == '"') {
      j->t = Json::Str;
      return string(&j->s);
    }
    if (e - p >= 4 && !memcmp(p, "true", 4)) return j->t = Json::Bool, j->b = true, p += 4, true;
    if (e - p >= 5 && !memcmp(p, "false", 5)) return j->t = Json::Bool, j->b = false, p += 5, true;
    if (e - p >= 4 && !memcmp(p, "null", 4)) return j->t = Json::Null, p += 4, true;
    const char* q = p;
    while (q < e && (*q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E' || (*q >= '0' && *q <= '9'))) q++;
    if (q == p) return fail("unexpected character");
    j->t = Json::Num;
    j->n = strtod(std::string(p, q).c_str(), nullptr);
    p = q;
    return true;
  }
};
inline bool parse_json(const std::string& text, Json* out, std::string* err) {
  JsonParser ps(text);
  if (!ps.value(out)) {
    if (err) *err = ps.err;
    return false;
  }
  ps.ws();
  if (ps.p != ps.e) {
    if (err) *err = "trailing characters after the JSON value";
    return false;
  }
  return true;
}
inline bool base64_decode(const std::string& in, std::vector<uint8_t>* out) {  // padding optional (mod.rs:31-57 strips it)
  unsigned acc = 0;
  int bits = 0;
  for (char c : in) {
    int v;
    if (c >= 'A' && c <= 'Z') v = c - 'A';
    else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
    else if (c >= '0' && c <= '9') v = c - '0' + 52;
    else if (c == '+' || c == '-') v = 62;
    else if (c == '/' || c == '_') v = 63;
    else if (c == '=') break;
    else return false;
    acc = (acc << 6) | (unsigned)v;
    bits += 6;
    if (bits >= 8) {
      bits -= 8;
      out->push_back((uint8_t)(acc >> bits));
    }
  }
  return true;
}

// ---------------------------------------------------------------- bincode primitives
struct Wr {
  std::vector<uint8_t> b;
  void raw(const void* p, size_t n) { b.insert(b.end(), (const uint8_t*)p, (const uint8_t*)p + n); }
  void u8(uint8_t v) { b.push_back(v); }
  void boolean(bool v) { u8(v ? 1 : 0); }
  void u32(uint32_t v) { raw(&v, 4); }
  void u64(uint64_t v) { raw(&v, 8); }
  void i64(int64_t v) { raw(&v, 8); }
  void f32(float v) { raw(&v, 4); }
  void str(const std::string& s) {
    u64(s.size());
    raw(s.data(), s.size());
  }
  void vec_u32(const std::vector<uint32_t>& v) {
    u64(v.size());
    if (!v.empty()) raw(v.data(), v.size() * 4);
  }
};
struct Rd {
  const uint8_t* p;
  size_t n, i = 0;
  std::string err;
  Rd(const uint8_t* d, size_t len) : p(d), n(len) {}
  bool ok() const { return err.empty(); }
  bool take(void* out, size_t k) {
    if (!ok()) return false;
    if (i + k > n) {
      err = "truncated bincode payload";
      memset(out, 0, k);
      return false;
    }
    memcpy(out, p + i, k);
    i += k;
    return true;
  }
  uint8_t u8() {
    uint8_t v = 0;
    take(&v, 1);
    return v;
  }
  bool boolean() {
    const uint8_t v = u8();
    if (v > 1 && ok()) err = "invalid bool byte";
    return v == 1;
  }
  uint32_t u32() {
    uint32_t v = 0;
    take(&v, 4);
    return v;
  }
  uint64_t u64() {
    uint64_t v = 0;
    take(&v, 8);
    return v;
  }
  int64_t i64() {
    int64_t v = 0;
    take(&v, 8);
    return v;
  }
  float f32() {
    float v = 0;
    take(&v, 4);
    return v;
  }
  std::string str() {
    const uint64_t k = u64();
    if (!ok() || k > n - i) {  // (i <= n always; `i + k > n` wraps for a length near 2^64)
      if (ok()) err = "truncated string";
      return std::string();
    }
    std::string s((const char*)p + i, (size_t)k);
    i += k;
    return s;
  }
  std::vector<uint32_t> vec_u32() {
    const uint64_t k = u64();
    std::vector<uint32_t> v;
    if (!ok() || k > (n - i) / 4) {
      if (ok()) err = "truncated Vec<u32>";
      return v;
    }
    v.resize(k);
    if (k) memcpy(v.data(), p + i, k * 4);
    i += k * 4;
    return v;
  }
  bool tag() {  // Option tag
    const uint8_t t = u8();
    if (t > 1 && ok()) err = "invalid Option tag";
    return t == 1;
  }
};
template <class T>
struct Opt {
  bool some = false;
  T v{};
};

// ---------------------------------------------------------------- structs
struct SamplingParams {  // config.rs:505-537 (non-python layout), fields in declaration order
  Opt<float> temperature;
  Opt<uint64_t> max_tokens;
  bool ignore_eos = false;
  Opt<int64_t> top_k;
  Opt<float> top_p;
  Opt<std::string> session_id;
  Opt<float> frequency_penalty, presence_penalty;
  Opt<std::vector<std::string>> stop_sequences;
  Opt<bool> thinking, mcp_mode;
  Opt<std::string> grammar, grammar_json;
  Opt<uint32_t> reasoning_effort;  // enum variant index
};
inline void put(Wr& w, const SamplingParams& s) {
  auto of = [&](const Opt<float>& o) {
    w.u8(o.some);
    if (o.some) w.f32(o.v);
  };
  auto os = [&](const Opt<std::string>& o) {
    w.u8(o.some);
    if (o.some) w.str(o.v);
  };
  auto ob = [&](const Opt<bool>& o) {
    w.u8(o.some);
    if (o.some) w.boolean(o.v);
  };
  of(s.temperature);
  w.u8(s.max_tokens.some);
  if (s.max_tokens.some) w.u64(s.max_tokens.v);
  w.boolean(s.ignore_eos);
  w.u8(s.top_k.some);
  if (s.top_k.some) w.i64(s.top_k.v);
  of(s.top_p);
  os(s.session_id);
  of(s.frequency_penalty);
  of(s.presence_penalty);
  w.u8(s.stop_sequences.some);
  if (s.stop_sequences.some) {
    w.u64(s.stop_sequences.v.size());
    for (auto& x : s.stop_sequences.v) w.str(x);
  }
  ob(s.thinking);
  ob(s.mcp_mode);
  os(s.grammar);
  os(s.grammar_json);
  w.u8(s.reasoning_effort.some);
  if (s.reasoning_effort.some) w.u32(s.reasoning_effort.v);
}
inline void get(Rd& r, SamplingParams& s) {
  auto of = [&](Opt<float>& o) {
    if ((o.some = r.tag())) o.v = r.f32();
  };
  auto os = [&](Opt<std::string>& o) {
    if ((o.some = r.tag())) o.v = r.str();
  };
  auto ob = [&](Opt<bool>& o) {
    if ((o.some = r.tag())) o.v = r.boolean();
  };
  of(s.temperature);
  if ((s.max_tokens.some = r.tag())) s.max_tokens.v = r.u64();
  s.ignore_eos = r.boolean();
  if ((s.top_k.some = r.tag())) s.top_k.v = r.i64();
  of(s.top_p);
  os(s.session_id);
  of(s.frequency_penalty);
  of(s.presence_penalty);
  if ((s.stop_sequences.some = r.tag())) {
    const uint64_t k = r.u64();
    for (uint64_t i = 0; i < k && r.ok(); i++) s.stop_sequences.v.push_back(r.str());
  }
  ob(s.thinking);
  ob(s.mcp_mode);
  os(s.grammar);
  os(s.grammar_json);
  if ((s.reasoning_effort.some = r.tag())) s.reasoning_effort.v = r.u32();
}
struct Sequence {  // sequence.rs:32-51
  uint64_t id = 0, created_time = 0;
  Opt<uint64_t> swapped_time;
  uint32_t status = 0;
  std::vector<uint32_t> token_ids, output_ids, block_table;
  uint64_t num_cached_tokens = 0;
  Opt<uint64_t> mamba_prefix_hash;
  uint32_t last_token = 0;
  uint64_t block_size = 64;
  SamplingParams sampling_params;
  Opt<uint32_t> pd_first_token;
  bool is_tool_call_end = false, hit_stop_sequence = false;
  Opt<std::string> stop_sequence;
};
inline void put(Wr& w, const Sequence& s) {
  w.u64(s.id), w.u64(s.created_time);
  w.u8(s.swapped_time.some);
  if (s.swapped_time.some) w.u64(s.swapped_time.v);
  w.u32(s.status);
  w.vec_u32(s.token_ids), w.vec_u32(s.output_ids), w.vec_u32(s.block_table);
  w.u64(s.num_cached_tokens);
  w.u8(s.mamba_prefix_hash.some);
  if (s.mamba_prefix_hash.some) w.u64(s.mamba_prefix_hash.v);
  w.u32(s.last_token);
  w.u64(s.block_size);
  put(w, s.sampling_params);
  w.u8(s.pd_first_token.some);
  if (s.pd_first_token.some) w.u32(s.pd_first_token.v);
  w.u8(0);  // images: None (multimodal is outside this path)
  w.boolean(s.is_tool_call_end), w.boolean(s.hit_stop_sequence);
  w.u8(s.stop_sequence.some);
  if (s.stop_sequence.some) w.str(s.stop_sequence.v);
}
inline void get(Rd& r, Sequence& s) {
  s.id = r.u64(), s.created_time = r.u64();
  if ((s.swapped_time.some = r.tag())) s.swapped_time.v = r.u64();
  s.status = r.u32();
  s.token_ids = r.vec_u32(), s.output_ids = r.vec_u32(), s.block_table = r.vec_u32();
  s.num_cached_tokens = r.u64();
  if ((s.mamba_prefix_hash.some = r.tag())) s.mamba_prefix_hash.v = r.u64();
  s.last_token = r.u32();
  s.block_size = r.u64();
  get(r, s.sampling_params);
  if ((s.pd_first_token.some = r.tag())) s.pd_first_token.v = r.u32();
  if (r.u8() != 0 && r.ok()) r.err = "Sequence.images (multimodal) is outside this path";
  s.is_tool_call_end = r.boolean(), s.hit_stop_sequence = r.boolean();
  if ((s.stop_sequence.some = r.tag())) s.stop_sequence.v = r.str();
}
struct DecodeSequence {  // sequence.rs:53-62
  uint64_t id = 0;
  uint32_t last_token = 0;
  uint64_t len = 0, last_block_tokens = 0;
  uint32_t block_table_last = 0;
  std::vector<uint32_t> block_tables;
  SamplingParams sampling_params;
};
inline void put(Wr& w, const DecodeSequence& s) {
  w.u64(s.id), w.u32(s.last_token), w.u64(s.len), w.u64(s.last_block_tokens), w.u32(s.block_table_last);
  w.vec_u32(s.block_tables);
  put(w, s.sampling_params);
}
inline void get(Rd& r, DecodeSequence& s) {
  s.id = r.u64(), s.last_token = r.u32(), s.len = r.u64(), s.last_block_tokens = r.u64(), s.block_table_last = r.u32();
  s.block_tables = r.vec_u32();
  get(r, s.sampling_params);
}

// ---------------------------------------------------------------- MessageType (declaration order = bincode variant index)
static const char* const kVariants[] = {
    "Init", "InitAck", "LoadingProgress", "RunPrefill", "RunDecode", "RunResponse", "RunEmbed", "RunResponseEmbed", "FinishDecode",
    "CaptureMambaPrefixState", "CaptureMambaPrefixStateResponse", "HasMambaPrefixState", "HasMambaPrefixStateResponse", "Error",
    "Heartbeat", "TransferPrefill", "TransferPrefillResponse", "ReceivePrefill", "ReceivePrefillResponse", "CheckPrefillStatus",
    "CheckPrefillStatusResponse", "KVCacheSwap", "KVCacheSwapResponse", "KvCacheSend", "KvCacheSendResponse", "KvCacheReceive",
    "KvCacheReceiveResponse", "KvCacheRelease", "KvCacheReleaseResponse", "CheckKvCacheRelease", "CheckKvCacheReleaseResponse",
    "ClearBlocks", "ClearBlocksResponse", "UsableMemoryLeft", "Shutdown"};
static const int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
inline int variant_index(const char* name) {
  for (int i = 0; i < kNumVariants; i++)
    if (!strcmp(kVariants[i], name)) return i;
  return -1;
}
struct Message {
  std::string name;
  bool flag = false;                       // InitAck / *Response payloads; second field of RunPrefill / RunDecode / KVCacheSwap
  uint64_t a = 0, b = 0;                   // LoadingProgress (a, b); FinishDecode a
  std::vector<Sequence> seqs;              // RunPrefill
  std::vector<DecodeSequence> dseqs;       // RunDecode
  std::vector<uint32_t> ids;               // RunResponse / ClearBlocks
  std::string text;                        // Error
  std::vector<std::pair<uint64_t, uint64_t>> map;  // KVCacheSwap (in wire order)
};
inline bool encode(const Message& m, std::vector<uint8_t>* out, std::string* err) {
  const int vi = variant_index(m.name.c_str());
  if (vi < 0) return *err = "unknown MessageType variant " + m.name, false;
  Wr w;
  w.u32((uint32_t)vi);
  const std::string& n = m.name;
  if (n == "InitAck" || n == "KVCacheSwapResponse" || n == "ClearBlocksResponse") w.boolean(m.flag);
  else if (n == "LoadingProgress") w.u64(m.a), w.u64(m.b);
  else if (n == "RunPrefill") {
    w.u64(m.seqs.size());
    for (auto& s : m.seqs) put(w, s);
    w.boolean(m.flag);
  } else if (n == "RunDecode") {
    w.u64(m.dseqs.size());
    for (auto& s : m.dseqs) put(w, s);
    w.boolean(m.flag);
  } else if (n == "RunResponse" || n == "ClearBlocks") w.vec_u32(m.ids);
  else if (n == "FinishDecode") w.u64(m.a);
  else if (n == "Error") w.str(m.text);
  else if (n == "Heartbeat" || n == "Shutdown") {
  } else if (n == "KVCacheSwap") {
    w.u64(m.map.size());
    for (auto& kv : m.map) w.u64(kv.first), w.u64(kv.second);
    w.boolean(m.flag);
  } else return *err = "MessageType::" + n + " is not part of the forward-pass runner protocol", false;
  out->swap(w.b);
  return true;
}
inline bool decode(const uint8_t* p, size_t len, Message* m, std::string* err) {
  Rd r(p, len);
  const uint32_t vi = r.u32();
  if (!r.ok()) return *err = r.err, false;
  if (vi >= (uint32_t)kNumVariants) return *err = "variant index out of range", false;
  const std::string n = m->name = kVariants[vi];
  if (n == "InitAck" || n == "KVCacheSwapResponse" || n == "ClearBlocksResponse") m->flag = r.boolean();
  else if (n == "LoadingProgress") m->a = r.u64(), m->b = r.u64();
  else if (n == "RunPrefill") {
    const uint64_t k = r.u64();
    for (uint64_t i = 0; i < k && r.ok(); i++) {
      m->seqs.emplace_back();
      get(r, m->seqs.back());
    }
    m->flag = r.boolean();
  } else if (n == "RunDecode") {
    const uint64_t k = r.u64();
    for (uint64_t i = 0; i < k && r.ok(); i++) {
      m->dseqs.emplace_back();
      get(r, m->dseqs.back());
    }
    m->flag = r.boolean();
  } else if (n == "RunResponse" || n == "ClearBlocks") m->ids = r.vec_u32();
  else if (n == "FinishDecode") m->a = r.u64();
  else if (n == "Error") m->text = r.str();
  else if (n == "Heartbeat" || n == "Shutdown") {
  } else if (n == "KVCacheSwap") {
    const uint64_t k = r.u64();
    for (uint64_t i = 0; i < k && r.ok(); i++) {
      const uint64_t a = r.u64(), b = r.u64();
      m->map.emplace_back(a, b);
    }
    m->flag = r.boolean();
  } else return *err = "MessageType::" + n + " is not part of the forward-pass runner protocol", false;
  if (!r.ok()) return *err = r.err, false;
  if (r.i != r.n) return *err = "trailing bytes after the message", false;
  return true;
}

}  // namespace vra_wire
