// runner_main.cpp — `vra_runner`: the reference's `runner` process (src/runner/runner.rs) on top of libvllm_rs_amd.so.
//
//   vra_runner --sock <name> [--uuid <id>]
//
// is what an unmodified vllm.rs engine spawns per GPU (src/core/engine.rs:187-330): it connects to the engine's
// GenericNamespaced local socket (a Linux abstract-namespace Unix socket, "\0<name>"), writes "ready\n", takes `Init` as
// JSON (model config, rank, device, NCCL id, checkpoint paths), loads its shard of the checkpoint (safetensors, by HF tensor
// name: the library slices for tensor parallelism and repacks int4 itself), answers `InitAck`, takes the engine's
// `UsableMemoryLeft(EngineConfig)` (JSON: the negotiated KV plan), sizes its cache from it, answers `InitAck` again, then
// serves `RunPrefill` / `RunDecode` (bincode) with `RunResponse` token ids until `Shutdown`; a heartbeat thread answers the
// engine's command channel (src/utils/heartbeat.rs).  As in the reference the ENGINE owns scheduler and
// block manager; the runner builds the step's InputMetadata from the sequences it is handed (ModelRunner::prepare_prefill /
// prepare_decode, src/core/runner.rs:978-1388) and runs forward + sampling (runner.rs:1390-1570).
// The Python twin (vllm_rs_amd/runner_ipc.py) exists for the tests' engine side; wire.h / wire.py share known-answer bytes.
//
//   vra_runner --wire-echo     bincode MessageType on stdin -> decoded and re-encoded on stdout   (codec parity tests, no GPU)
//   vra_runner --init-echo     JSON Init on stdin -> one summary line on stdout
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/vllm_rs_amd.h"
#include "wire.h"

using namespace vra_wire;

static void die(const std::string& m) {
  fprintf(stderr, "vra_runner: %s\n", m.c_str());
  exit(1);
}

// ---------------------------------------------------------------- framing (mod.rs:246-295)
static void write_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    const ssize_t k = write(fd, c, n);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) die("socket write failed");
    c += k, n -= (size_t)k;
  }
}
static bool read_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) {
    const ssize_t k = read(fd, c, n);
    if (k < 0 && errno == EINTR) continue;
    if (k <= 0) return false;
    c += k, n -= (size_t)k;
  }
  return true;
}
static const uint32_t kMaxFrame = 1u << 30;  // a frame is a step's sequences (a 1M-token batch is ~4 MB): a larger length prefix is a corrupt stream
static std::vector<uint8_t> recv_frame(int fd) {  // receive_local: length, payload, then acknowledge with 0x01
  uint32_t n = 0;
  if (!read_all(fd, &n, 4)) die("peer closed the stream");
  if (n > kMaxFrame) die("frame length " + std::to_string(n) + " exceeds the limit: corrupt stream");
  std::vector<uint8_t> b(n);
  if (n && !read_all(fd, b.data(), n)) die("peer closed the stream inside a frame");
  const uint8_t ack = 1;
  write_all(fd, &ack, 1);
  return b;
}
static void send_frame(int fd, const std::vector<uint8_t>& b) {  // send_local: length, payload, wait for the 1-byte ack
  const uint32_t n = (uint32_t)b.size();
  write_all(fd, &n, 4);
  if (n) write_all(fd, b.data(), n);
  uint8_t ack = 0;
  if (!read_all(fd, &ack, 1) || ack != 1) die("unexpected acknowledgment byte");
}
static void send_msg(int fd, const Message& m) {
  std::vector<uint8_t> b;
  std::string err;
  if (!encode(m, &b, &err)) die(err);
  send_frame(fd, b);
}

// ---------------------------------------------------------------- Init -> configs (config.rs:218-255,285-328; wire.py model_cfg_from_init)
struct Init {
  int rank = 0, dev = 0, world = 1;
  vra_model_config mc{};
  vra_engine_config ec{};
  std::vector<uint8_t> nccl_id;
  std::string config_file;
  std::vector<std::string> files;
  uint64_t seed = 1234;
};
// EngineConfig (config.rs:285-328) as it travels in Init.econfig and again, after the engine's allocation plan, in
// MessageType::UsableMemoryLeft: the fields this path uses
static void econfig_from_json(const Json& e, vra_engine_config* ecp) {
  vra_engine_config& ec = *ecp;
  ec.block_size = (int)e.i64("block_size", 64), ec.max_num_seqs = (int)e.i64("max_num_seqs", 32);
  ec.max_model_len = (int)e.i64("max_model_len", 0), ec.num_gpu_blocks = (int)e.i64("num_blocks", 0);
  ec.kv_fraction = 0.f, ec.prefill_chunk = 8192, ec.enable_prefix_cache = 0, ec.prefix_cache_fraction = 0.65f;
  // decode graphs as the reference's runner captures them (graph.rs:370-377: batches {1..15, 16, 32, ...}), replayed by
  // vra_engine_forward_tokens; VRA_RUNNER_GRAPH=0 keeps eager launches (tensor-parallel ranks stay eager: main() clears it)
  const char* ge = getenv("VRA_RUNNER_GRAPH");
  ec.use_graph = ge && ge[0] == '0' ? 0 : 1;
  ec.seed = (uint64_t)e.i64("seed", 1234);
  ec.fp8_kvcache = e.boolean("fp8_kvcache", false);
  ec.cpu_mem_fold = (float)e.num("cpu_mem_fold", 0.2);  // kvcache_allocator.rs:317: unwrap_or(0.2) — the engine plans its CPU block ids with it
}
static bool init_from_json(const std::string& text, Init* out, std::string* err) {
  Json root;
  if (!parse_json(text, &root, err)) return false;
  const Json* req = root.get("Init");
  if (!req || req->t != Json::Obj) return *err = "expected MessageType::Init as JSON", false;
  out->rank = (int)req->i64("rank", 0), out->dev = (int)req->i64("dev_id", 0), out->world = (int)req->i64("num_shards", 1);
  const Json* c = req->get("config");
  if (!c || c->t != Json::Obj) return *err = "Init.config missing", false;
  vra_model_config& mc = out->mc;
  std::string arch = "LlamaForCausalLM";
  if (const Json* a = c->get("architectures"))
    if (a->t == Json::Arr && !a->a.empty() && a->a[0].t == Json::Str) arch = a->a[0].s;
  const bool qwen = arch.rfind("Qwen2", 0) == 0;
  mc.arch = qwen ? 1 : (arch.rfind("Qwen3", 0) == 0 ? 2 : 0);  // (Qwen3: q_norm / k_norm arrive as tensors, their shape sets the mode)
  mc.hidden_size = (int)c->i64("hidden_size", 0), mc.intermediate_size = (int)c->i64("intermediate_size", 0);
  mc.num_layers = (int)c->i64("num_hidden_layers", 0), mc.num_heads = (int)c->i64("num_attention_heads", 0);
  mc.num_kv_heads = (int)c->i64("num_key_value_heads", mc.num_heads);
  mc.head_dim = c->has("head_dim") ? (int)c->i64("head_dim", 0) : (mc.num_heads ? mc.hidden_size / mc.num_heads : 0);
  mc.vocab_size = (int)c->i64("vocab_size", 0), mc.max_position_embeddings = (int)c->i64("max_position_embeddings", 4096);
  // sliding-window attention (llama.rs:46,284): a window shorter than the model's positions goes into the forward; one that covers them
  // all is full causal attention
  mc.sliding_window = 0;
  if (c->has("sliding_window") && c->i64("sliding_window", 0) > 0 && c->i64("sliding_window", 0) < mc.max_position_embeddings &&
      c->boolean("use_sliding_window", true))
    mc.sliding_window = (int)c->i64("sliding_window", 0);
  mc.rms_norm_eps = (float)c->num("rms_norm_eps", 1e-5), mc.rope_theta = c->has("rope_theta") ? c->num("rope_theta", 10000.0) : 10000.0;
  mc.rope_scaling_type = 0, mc.rope_factor = 1.0, mc.rope_low_freq_factor = 1.0, mc.rope_high_freq_factor = 4.0;
  mc.rope_original_max_position = mc.max_position_embeddings;
  if (const Json* rs = c->get("rope_scaling"))
    if (rs->t == Json::Obj) {
      const std::string ty = rs->has("rope_type") ? rs->str("rope_type", "") : rs->str("type", "");
      mc.rope_scaling_type = ty == "linear" ? 1 : (ty == "llama3" ? 2 : (ty == "dynamic" ? 3 : (ty == "yarn" ? 4 : 0)));
      if (!ty.empty() && ty != "default" && !mc.rope_scaling_type) return *err = "Unknown rope_type: " + ty, false;  // rotary_emb.rs:417-419
      mc.rope_dynamic_alpha = rs->has("alpha") ? 1 : 0;
      mc.rope_factor = rs->has("alpha") ? rs->num("alpha", 1.0) : rs->num("factor", 1.0), mc.rope_low_freq_factor = rs->num("low_freq_factor", 1.0);
      mc.rope_high_freq_factor = rs->num("high_freq_factor", 4.0);
      // rotary_emb.rs:150-164: the key, else max_position_embeddings / factor, else max_position_embeddings
      mc.rope_original_max_position = rs->has("original_max_position_embeddings") ? (int)rs->i64("original_max_position_embeddings", 0)
                                      : (rs->has("factor") ? (int)((double)mc.max_position_embeddings / rs->num("factor", 1.0)) : mc.max_position_embeddings);
      mc.rope_original_max_position_f = rs->has("original_max_position_embeddings") ? rs->num("original_max_position_embeddings", 0.0)
                                        : (rs->has("factor") ? (double)mc.max_position_embeddings / rs->num("factor", 1.0) : (double)mc.max_position_embeddings);
      mc.rope_yarn_explicit = (rs->has("beta_fast") ? 1 : 0) | (rs->has("beta_slow") ? 2 : 0) | (rs->has("attn_factor") ? 4 : 0) | (rs->has("extrapolation_factor") ? 8 : 0);
      mc.rope_yarn_beta_fast = rs->num("beta_fast", 32.0), mc.rope_yarn_beta_slow = rs->num("beta_slow", 1.0);
      mc.rope_yarn_attn_factor = rs->num("attn_factor", 1.0), mc.rope_yarn_extrapolation_factor = rs->num("extrapolation_factor", 1.0);
    }
  mc.attention_bias = c->boolean("attention_bias", false) || c->boolean("qkv_bias", false) || qwen;
  mc.quant_method = 0, mc.bits = 4, mc.group_size = 128;
  if (const Json* q = c->get("quantization_config"))
    if (q->t == Json::Obj) {
      std::string m = q->str("quant_method", "");
      std::transform(m.begin(), m.end(), m.begin(), ::tolower);
      mc.quant_method = m == "gptq" ? 1 : (m == "awq" ? 2 : 0);
      if (!m.empty() && !mc.quant_method) return *err = "quant_method " + m + " is not on this path (gptq, awq)", false;
      if (q->i64("bits", 4) != 4) return *err = "only 4-bit GPTQ/AWQ checkpoints are supported (wna16.rs:154-160)", false;
      if (q->boolean("desc_act", false)) return *err = "desc_act=true checkpoints are rejected, as in the reference (utils/mod.rs:1316-1318)", false;
      mc.group_size = (int)q->i64("group_size", 128);
    }
  mc.dtype = req->str("dtype", "BF16") == "F16" ? VRA_F16 : VRA_BF16;
  mc.tie_word_embeddings = c->boolean("tie_word_embeddings", false);
  if (mc.hidden_size <= 0 || mc.num_layers <= 0 || mc.num_heads <= 0 || mc.vocab_size <= 0) return *err = "Init.config lacks the model dimensions", false;
  vra_engine_config& ec = out->ec;
  const Json* e = req->get("econfig");
  static const Json none;
  if (!e || e->t != Json::Obj) e = &none;
  econfig_from_json(*e, &ec);
  ec.tp_rank = out->rank, ec.tp_world_size = out->world, ec.device = out->dev;
  out->seed = ec.seed;
  if (const Json* id = req->get("nccl_id"))
    if (id->t == Json::Str) {
      if (!base64_decode(id->s, &out->nccl_id) || out->nccl_id.size() != 128) return *err = "nccl_id: expected 128 bytes", false;
    }
  if (const Json* mp = req->get("model_pathes"))
    if (mp->t == Json::Obj) {
      out->config_file = mp->str("config_filename", "");
      if (const Json* fs = mp->get("filenames"))
        if (fs->t == Json::Arr)
          for (auto& f : fs->a)
            if (f.t == Json::Str) out->files.push_back(f.s);
    }
  return true;
}

// ---------------------------------------------------------------- safetensors -> vra_engine_load_tensor (checkpoint.py)
static uint16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static float f16_to_f32(uint16_t h) {  // IEEE binary16 -> binary32, exact
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = sign;
    else {  // subnormal: normalise
      int sh = 0;
      uint32_t mm = m;
      while (!(mm & 0x400u)) mm <<= 1, sh++;
      u = sign | ((uint32_t)(113 - sh) << 23) | ((mm & 0x3ffu) << 13);
    }
  } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
  else u = sign | ((e + 112) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static uint16_t f32_to_f16(float f) {  // round to nearest even, overflow to infinity, NaN quieted
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
  if (a >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);  // >= 65536 (65520 rounds to inf below)
  if (a < 0x33000000u) return sign;                         // < 2^-25: rounds to zero
  const int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;
  int shift = e < -14 ? 13 + (-14 - e) : 13;  // subnormal halves lose more mantissa bits
  const uint32_t half = 1u << (shift - 1), mask = (1u << shift) - 1;
  uint32_t q = m >> shift;
  const uint32_t rem = m & mask;
  if (rem > half || (rem == half && (q & 1u))) q++;
  uint32_t out;
  if (e < -14) out = q;  // subnormal (q may carry into the smallest normal: the bit pattern is right either way)
  else out = ((uint32_t)(e + 15) << 10) + (q - 0x400u);
  if (out >= 0x7c00u) out = 0x7c00u;
  return (uint16_t)(sign | out);
}
static bool ends_with(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && !s.compare(s.size() - n, n, suf);
}
static void load_safetensors(void* eng, const std::string& path, const vra_model_config& mc) {
  const int fd = open(path.c_str(), O_RDONLY);
  if (fd < 0) die("cannot open " + path);
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size < 8) die("cannot stat " + path);
  const uint8_t* base = (const uint8_t*)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  if (base == MAP_FAILED) die("mmap failed for " + path);
  uint64_t hlen;
  memcpy(&hlen, base, 8);
  if (8 + hlen > (uint64_t)st.st_size) die(path + ": bad safetensors header length");
  Json hdr;
  std::string err;
  if (!parse_json(std::string((const char*)base + 8, (size_t)hlen), &hdr, &err) || hdr.t != Json::Obj) die(path + ": header: " + err);
  const bool want_bf16 = mc.dtype == VRA_BF16;
  for (auto& kv : hdr.o) {
    const std::string& name = kv.first;
    if (name == "__metadata__" || ends_with(name, "rotary_emb.inv_freq")) continue;
    const Json& meta = kv.second;
    const std::string dt = meta.str("dtype", "");
    const Json *sh = meta.get("shape"), *off = meta.get("data_offsets");
    if (!sh || !off || off->a.size() != 2) die(path + ": tensor " + name + " lacks shape / data_offsets");
    std::vector<int64_t> shape;
    int64_t numel = 1;
    for (auto& d : sh->a) shape.push_back((int64_t)d.n), numel *= (int64_t)d.n;
    const uint64_t b0 = (uint64_t)off->a[0].n, b1 = (uint64_t)off->a[1].n;
    if (8 + hlen + b1 > (uint64_t)st.st_size || b1 < b0) die(path + ": tensor " + name + " is out of the file");
    const uint8_t* data = base + 8 + hlen + b0;
    int elem = 0;
    std::vector<uint16_t> conv;
    if (dt == "I32" || dt == "U32") {
      elem = 4;
      if (ends_with(name, ".g_idx")) {  // desc_act = false only: must be the trivial i / group_size map
        const int32_t* g = (const int32_t*)data;
        const int64_t gs = mc.group_size > 0 ? mc.group_size : numel;
        for (int64_t i = 0; i < numel; i++)
          if (g[i] != (int32_t)(i / gs)) die(name + ": non-trivial g_idx (act-order) is not supported");
        continue;
      }
    } else if (dt == "BF16" || dt == "F16" || dt == "F32") {
      elem = 2;
      if (!((dt == "BF16" && want_bf16) || (dt == "F16" && !want_bf16))) {  // scales / bias are f16 on disk even for bf16 models
        conv.resize((size_t)numel);
        for (int64_t i = 0; i < numel; i++) {
          float f;
          if (dt == "F32") memcpy(&f, data + 4 * i, 4);
          else {
            uint16_t h;
            memcpy(&h, data + 2 * i, 2);
            f = dt == "F16" ? f16_to_f32(h) : bf16_to_f32(h);
          }
          conv[(size_t)i] = want_bf16 ? f32_to_bf16(f) : f32_to_f16(f);
        }
        data = (const uint8_t*)conv.data();
      }
    } else {
      die(path + ": tensor " + name + " has unsupported dtype " + dt);
    }
    if (vra_engine_load_tensor(eng, name.c_str(), data, shape.data(), (int)shape.size(), elem) != 0)
      die("load_tensor(" + name + "): " + vra_engine_last_error(eng));
  }
  munmap((void*)base, (size_t)st.st_size);
  close(fd);
}
static std::string dir_of(const std::string& p) {
  const size_t k = p.find_last_of('/');
  return k == std::string::npos ? "." : p.substr(0, k);
}

// ---------------------------------------------------------------- the step: metadata, forward, sampling
struct Strategy {  // LogitsProcessor::get_strategy + the runner's defaults (runner.rs:1436-1497); greedy = no sampling
  bool greedy = true;
  int k = 0;
  float p = -1.f, t = 1.f;
};
static Strategy strategy_of(const SamplingParams& sp) {
  Strategy s;
  if (sp.temperature.some && sp.temperature.v == 0.0f) return s;
  const bool has_user = sp.temperature.some || (sp.top_k.some && sp.top_k.v > 0) || (sp.top_p.some && sp.top_p.v > 0.f && sp.top_p.v < 1.f);
  s.greedy = false;
  if (!has_user) return s.k = 32, s.p = 0.95f, s.t = 0.7f, s;  // no generation config: the reference's default (Appendix A4)
  if (!sp.temperature.some || sp.temperature.v < 1e-7f) return s.greedy = true, s;
  s.k = sp.top_k.some && sp.top_k.v > 0 ? (int)sp.top_k.v : 0;
  s.p = sp.top_p.some ? sp.top_p.v : -1.f;
  s.t = sp.temperature.v;
  return s;
}
struct Runner {
  void* eng = nullptr;
  int vocab = 0, block_size = 64;
  uint64_t seed = 1234, calls = 0;
  bool have_strategy = false;
  Strategy cached;  // of the first sequence of the last prefill (runner.rs:1411,1499-1511: Appendix A3)

  // forward + sampling in one engine call (round 6): only the token ids come back — the decode step replays the engine's captured
  // hipGraph, greedy tokens are the first maximal index as candle's argmax (logits_processor.rs:67-70), a stochastic strategy runs on
  // the device logits.  (Rounds 3-5 copied [B, vocab] f32 logits to the host, took the argmax there and sent the logits back to the
  // device for the stochastic strategies.)
  std::vector<uint32_t> forward_tokens(const std::vector<uint32_t>& ids, const std::vector<int64_t>& pos, const std::vector<int64_t>& slots, bool prefill,
                                       const std::vector<uint32_t>& bt, size_t mb, const std::vector<uint32_t>& ctx, const std::vector<uint32_t>* cu, int B,
                                       const Strategy& st, const char* what) {
    std::vector<uint32_t> out((size_t)B);
    ++calls;
    if (vra_engine_forward_tokens(eng, ids.data(), pos.data(), slots.data(), (int)ids.size(), prefill ? 1 : 0, bt.data(), (int)mb, ctx.data(),
                                  cu ? cu->data() : nullptr, B, st.greedy ? 0 : 1, st.k, st.p, st.t, (seed << 20) + calls, out.data()) != 0) {
      // a runner error is answered with an empty RunResponse, as the reference does (runner.rs:246-292): the engine fails the
      // step, the other ranks and later steps live on
      fprintf(stderr, "vra_runner: forward (%s) failed: %s\n", what, vra_engine_last_error(eng));
      return {};
    }
    return out;
  }
  // ModelRunner::prepare_prefill (runner.rs:978-1241) on wire Sequences
  std::vector<uint32_t> run_prefill(const std::vector<Sequence>& seqs) {
    const int B = (int)seqs.size();
    if (!B) return {};
    std::vector<uint32_t> ids, ctx, cu{0};
    std::vector<int64_t> pos, slots;
    size_t mb = 1;
    for (auto& s : seqs) mb = std::max(mb, s.block_table.size());
    std::vector<uint32_t> bt((size_t)B * mb, 0u);
    for (int b = 0; b < B; b++) {
      const Sequence& s = seqs[(size_t)b];
      const int64_t cached = (int64_t)s.num_cached_tokens, len = (int64_t)s.token_ids.size();
      const int64_t n = std::min<int64_t>(8192, len - cached);  // scheduler.rs:203 / runner.rs:984
      for (int64_t p = cached; p < cached + n; p++) {
        ids.push_back(s.token_ids[(size_t)p]);
        pos.push_back(p);
        if ((size_t)(p / block_size) >= s.block_table.size()) {
          fprintf(stderr, "vra_runner: RunPrefill: block table shorter than the chunk\n");
          return {};
        }
        slots.push_back((int64_t)s.block_table[(size_t)(p / block_size)] * block_size + p % block_size);
      }
      cu.push_back((uint32_t)ids.size());
      ctx.push_back((uint32_t)(cached + n));
      std::copy(s.block_table.begin(), s.block_table.end(), bt.begin() + (size_t)b * mb);
    }
    cached = strategy_of(seqs[0].sampling_params), have_strategy = true;
    return forward_tokens(ids, pos, slots, true, bt, mb, ctx, &cu, B, cached, "prefill");
  }
  // ModelRunner::prepare_decode (runner.rs:1243-1388): slot = block_table_last * BS + last_block_tokens - 1 (:1259-1262)
  std::vector<uint32_t> run_decode(const std::vector<DecodeSequence>& seqs) {
    const int B = (int)seqs.size();
    if (!B) return {};
    size_t mb = 1;
    for (auto& s : seqs) mb = std::max(mb, s.block_tables.size());
    std::vector<uint32_t> bt((size_t)B * mb, 0u), ids, ctx;
    std::vector<int64_t> pos, slots;
    for (int b = 0; b < B; b++) {
      const DecodeSequence& s = seqs[(size_t)b];
      std::copy(s.block_tables.begin(), s.block_tables.end(), bt.begin() + (size_t)b * mb);
      ids.push_back(s.last_token);
      pos.push_back((int64_t)s.len - 1);
      slots.push_back((int64_t)s.block_table_last * block_size + (int64_t)s.last_block_tokens - 1);
      ctx.push_back((uint32_t)s.len);
    }
    Strategy st = cached;
    if (!have_strategy) st.greedy = false, st.k = 32, st.p = 0.95f, st.t = 0.7f;
    return forward_tokens(ids, pos, slots, false, bt, mb, ctx, nullptr, B, st, "decode");
  }
};

// ---------------------------------------------------------------- heartbeat (src/utils/heartbeat.rs:8-78, src/utils/command.rs:91-172)
// The daemon side: connect to the engine's command channel "command_{uuid}@vllm-rs-runner-heartbeat.sock" (retry once a second,
// up to 120 times), announce `ready`, then acknowledge one bincode frame (MessageType::Heartbeat) per second; when the engine
// goes away (EOF / broken pipe) the runner exits, as the reference's does.
static void* heartbeat_main(void* arg) {
  const std::string name = "command_" + *static_cast<std::string*>(arg) + "@vllm-rs-runner-heartbeat.sock";
  delete static_cast<std::string*>(arg);
  int fd = -1;
  for (int attempt = 0; attempt < 120 && fd < 0; attempt++) {
    fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) return nullptr;
    sockaddr_un addr{};
    addr.sun_family = AF_UNIX;
    if (name.size() + 1 > sizeof(addr.sun_path)) return nullptr;
    memcpy(addr.sun_path + 1, name.data(), name.size());
    if (connect(fd, (sockaddr*)&addr, (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + name.size())) != 0) {
      close(fd);
      fd = -1;
      sleep(1);
    }
  }
  if (fd < 0) {
    fprintf(stderr, "vra_runner: no heartbeat channel %s (continuing without)\n", name.c_str());
    return nullptr;
  }
  if (write(fd, "ready\n", 6) != 6) return nullptr;
  for (;;) {
    uint32_t n = 0;
    if (!read_all(fd, &n, 4)) break;
    std::vector<uint8_t> b(n > kMaxFrame ? 0 : n);
    if (n > kMaxFrame || (n && !read_all(fd, b.data(), n))) break;
    const uint8_t ack = 1;
    if (write(fd, &ack, 1) != 1) break;
  }
  fprintf(stderr, "vra_runner: parent process disconnected, exiting\n");
  _exit(0);
}
static void start_heartbeat(const std::string& uuid) {
  pthread_t th;
  if (pthread_create(&th, nullptr, heartbeat_main, new std::string(uuid)) == 0) pthread_detach(th);
}

static std::string read_stdin() {
  std::string s;
  char buf[65536];
  size_t k;
  while ((k = fread(buf, 1, sizeof buf, stdin)) > 0) s.append(buf, k);
  return s;
}

int main(int argc, char** argv) {
  std::string sock_name, uuid;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    if (a == "--wire-echo") {
      const std::string in = read_stdin();
      Message m;
      std::string err;
      if (!decode((const uint8_t*)in.data(), in.size(), &m, &err)) die(err);
      std::vector<uint8_t> out;
      if (!encode(m, &out, &err)) die(err);
      fwrite(out.data(), 1, out.size(), stdout);
      return 0;
    }
    if (a == "--cvt" && i + 1 < argc) {  // conversion helpers of the checkpoint loader, for the CPU tests: f32 <-> f16 / bf16 bit patterns
      const std::string mode = argv[i + 1], in = read_stdin();
      std::string out;
      if (mode == "f32-f16" || mode == "f32-bf16") {
        for (size_t k = 0; k + 4 <= in.size(); k += 4) {
          float f;
          memcpy(&f, in.data() + k, 4);
          const uint16_t h = mode == "f32-f16" ? f32_to_f16(f) : f32_to_bf16(f);
          out.append((const char*)&h, 2);
        }
      } else if (mode == "f16-f32" || mode == "bf16-f32") {
        for (size_t k = 0; k + 2 <= in.size(); k += 2) {
          uint16_t h;
          memcpy(&h, in.data() + k, 2);
          const float f = mode == "f16-f32" ? f16_to_f32(h) : bf16_to_f32(h);
          out.append((const char*)&f, 4);
        }
      } else die("--cvt f32-f16 | f32-bf16 | f16-f32 | bf16-f32");
      fwrite(out.data(), 1, out.size(), stdout);
      return 0;
    }
    if (a == "--init-echo") {
      Init in;
      std::string err;
      if (!init_from_json(read_stdin(), &in, &err)) die(err);
      printf("rank %d dev %d world %d arch %d H %d I %d L %d Hq %d Hkv %d D %d V %d maxpos %d eps %.9g theta %.17g rope %d %.17g %.17g %.17g %d bias %d quant %d g %d dtype %d "
             "tie %d bs %d seqs %d len %d blocks %d seed %llu fp8 %d nccl %zu cfg %s files %zu\n",
             in.rank, in.dev, in.world, in.mc.arch, in.mc.hidden_size, in.mc.intermediate_size, in.mc.num_layers, in.mc.num_heads, in.mc.num_kv_heads,
             in.mc.head_dim, in.mc.vocab_size, in.mc.max_position_embeddings, (double)in.mc.rms_norm_eps, in.mc.rope_theta, in.mc.rope_scaling_type,
             in.mc.rope_factor, in.mc.rope_low_freq_factor, in.mc.rope_high_freq_factor, in.mc.rope_original_max_position, in.mc.attention_bias,
             in.mc.quant_method, in.mc.group_size, in.mc.dtype, in.mc.tie_word_embeddings, in.ec.block_size, in.ec.max_num_seqs, in.ec.max_model_len,
             in.ec.num_gpu_blocks, (unsigned long long)in.seed, in.ec.fp8_kvcache, in.nccl_id.size(), in.config_file.c_str(), in.files.size());
      return 0;
    }
    if (a == "--sock" && i + 1 < argc) sock_name = argv[++i];
    else if (a == "--uuid" && i + 1 < argc) uuid = argv[++i];
  }
  if (sock_name.empty()) die("Socket name missing (--sock <name>)");

  // ---- connect (runner.rs:41-57): GenericNamespaced = abstract-namespace Unix socket; retry until the engine listens
  int fd = -1;
  for (int attempt = 0; attempt < 600; attempt++) {
    fd = socket(AF_UNIX, SOCK_STREAM, 0);
    if (fd < 0) die("socket() failed");
    sockaddr_un addr{};
    addr.sun_family = AF_UNIX;
    if (sock_name.size() + 1 > sizeof(addr.sun_path)) die("socket name too long");
    memcpy(addr.sun_path + 1, sock_name.data(), sock_name.size());
    const socklen_t len = (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + sock_name.size());
    if (connect(fd, (sockaddr*)&addr, len) == 0) break;
    close(fd);
    fd = -1;
    usleep(100000);
  }
  if (fd < 0) die("cannot connect to the engine's socket " + sock_name);
  write_all(fd, "ready\n", 6);
  if (!uuid.empty()) start_heartbeat(uuid);  // heartbeat_worker(None, true, ..) (runner.rs:71)

  // ---- Init (JSON) -> engine
  const std::vector<uint8_t> first = recv_frame(fd);
  Init in;
  std::string err;
  if (!init_from_json(std::string((const char*)first.data(), first.size()), &in, &err)) die("Init: " + err);
  if (vra_set_device(in.dev) != 0) die(std::string("vra_set_device: ") + vra_last_error());
  void* comm = nullptr;
  if (in.world > 1) {  // Comm::from_rank (runner.rs:80-89)
    if (in.nccl_id.size() != 128) die("Init: tensor parallel without an NCCL id");
    comm = vra_comm_create(in.nccl_id.data(), in.rank, in.world, in.dev);
    if (!comm) die(std::string("vra_comm_create: ") + vra_last_error());
  }
  if (in.world > 1) in.ec.use_graph = 0;  // (the one-shot all-reduce's epoch words change per step: tensor-parallel ranks launch eagerly)
  void* eng = vra_engine_create(&in.mc, &in.ec);
  if (!eng) die("vra_engine_create failed");
  if (comm && vra_engine_set_comm(eng, comm) != 0) die(std::string("vra_engine_set_comm: ") + vra_engine_last_error(eng));
  std::vector<std::string> files = in.files;
  bool have_ckpt = !files.empty();
  for (auto& f : files)
    if (access(f.c_str(), R_OK) != 0) have_ckpt = false;
  if (!have_ckpt && !in.config_file.empty() && access(in.config_file.c_str(), R_OK) == 0) {
    // the file list may be relative or absent: the checkpoint directory is where config.json lives (utils/mod.rs:90-111)
    const std::string dir = dir_of(in.config_file);
    const std::string idx = dir + "/model.safetensors.index.json";
    files.clear();
    if (access(idx.c_str(), R_OK) == 0) {
      FILE* f = fopen(idx.c_str(), "rb");
      std::string text;
      char buf[65536];
      size_t k;
      while (f && (k = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, k);
      if (f) fclose(f);
      Json j;
      if (!parse_json(text, &j, &err)) die(idx + ": " + err);
      if (const Json* wm = j.get("weight_map"))
        for (auto& kv : wm->o)
          if (kv.second.t == Json::Str && std::find(files.begin(), files.end(), dir + "/" + kv.second.s) == files.end()) files.push_back(dir + "/" + kv.second.s);
      std::sort(files.begin(), files.end());
    } else if (access((dir + "/model.safetensors").c_str(), R_OK) == 0) {
      files.push_back(dir + "/model.safetensors");
    }
    have_ckpt = !files.empty();
  }
  if (have_ckpt) {
    for (auto& f : files) load_safetensors(eng, f, in.mc);
    if (vra_engine_finalize_model(eng) != 0) die(std::string("finalize (weights): ") + vra_engine_last_error(eng));
  } else if (vra_engine_init_synthetic(eng) != 0) {  // no checkpoint on this box: synthetic weights of the configured shape (bench mode)
    die(std::string("init_synthetic: ") + vra_engine_last_error(eng));
  }
  // ---- the reference's handshake (src/core/engine.rs:340-378 on the engine side, src/core/runner.rs:443-455 and
  // src/runner/runner.rs:214-236 on this side): the model is loaded -> InitAck #1 -> the engine plans the KV cache from the
  // memory that is left and answers with MessageType::UsableMemoryLeft(EngineConfig) as JSON -> the cache is sized from THAT
  // configuration (Init.econfig.num_blocks is only the placeholder 128, config.rs:458) -> InitAck #2 -> bincode loop.
  {
    Message ack;
    ack.name = "InitAck", ack.flag = true;
    send_msg(fd, ack);
  }
  {
    const std::vector<uint8_t> second = recv_frame(fd);
    Json j;
    std::string jerr;
    const Json* uml = nullptr;
    if (parse_json(std::string((const char*)second.data(), second.size()), &j, &jerr)) uml = j.get("UsableMemoryLeft");
    if (uml && uml->t == Json::Obj) {
      vra_engine_config neg = in.ec;
      econfig_from_json(*uml, &neg);
      if (neg.block_size != in.ec.block_size) die("UsableMemoryLeft: block_size differs from Init.econfig");
      // the reference replaces the whole EngineConfig (*econfig = ecfg, runner.rs:225); what vra_engine_update_config does not carry
      // over was fixed when the model was built from Init.econfig (the cache element type, the sampler seed): it must not differ
      if (neg.fp8_kvcache != in.ec.fp8_kvcache) die("UsableMemoryLeft: fp8_kvcache differs from Init.econfig (the model was built with Init's)");
      if (neg.seed != in.ec.seed) die("UsableMemoryLeft: seed differs from Init.econfig");
      if (vra_engine_update_config(eng, &neg) != 0) die(std::string("update_config: ") + vra_engine_last_error(eng));
      in.ec.num_gpu_blocks = neg.num_gpu_blocks, in.ec.max_num_seqs = neg.max_num_seqs, in.ec.max_model_len = neg.max_model_len;
    } else {
      // the frame has been consumed: answering it with the second InitAck would leave the engine waiting on a step that never gets
      // its reply — a peer that does not follow engine.rs:355-378 is a protocol error, reported and fatal
      die("expected UsableMemoryLeft(EngineConfig) as JSON after the first InitAck (engine.rs:355-378)" + (jerr.empty() ? std::string() : ": " + jerr));
    }
  }
  if (in.ec.block_size <= 0) die("econfig.block_size must be positive");
  if (vra_engine_finalize_weights(eng) != 0) die(std::string("finalize: ") + vra_engine_last_error(eng));
  Runner r;
  r.eng = eng, r.vocab = in.mc.vocab_size, r.block_size = in.ec.block_size, r.seed = in.seed;
  {
    Message ack;
    ack.name = "InitAck", ack.flag = true;
    send_msg(fd, ack);
  }

  // ---- message loop (runner.rs:246-430)
  for (;;) {
    const std::vector<uint8_t> b = recv_frame(fd);
    Message m;
    if (!decode(b.data(), b.size(), &m, &err)) {
      // the reference only logs what it cannot decode or does not serve; a reply here would be read by the engine as the
      // 1-byte acknowledgment of its NEXT frame
      fprintf(stderr, "vra_runner: undecodable frame (%s): ignored\n", err.c_str());
      err.clear();
      continue;
    }
    if (m.name == "Shutdown") break;
    Message out;
    if (m.name == "RunPrefill") out.name = "RunResponse", out.ids = r.run_prefill(m.seqs);
    else if (m.name == "RunDecode") out.name = "RunResponse", out.ids = r.run_decode(m.dseqs);
    else if (m.name == "FinishDecode" || m.name == "LoadingProgress" || m.name == "Heartbeat") continue;  // bookkeeping only, no reply (runner.rs:294-315)
    else if (m.name == "ClearBlocks") out.name = "ClearBlocksResponse", out.flag = true;
    else if (m.name == "KVCacheSwap") {  // runner.rs:297-312 -> ModelRunner::swap_kvcache
      std::vector<int64_t> pairs;
      for (auto& kv : m.map) pairs.push_back((int64_t)kv.first), pairs.push_back((int64_t)kv.second);
      out.name = "KVCacheSwapResponse";
      out.flag = vra_engine_swap_blocks(eng, pairs.data(), (int)m.map.size(), m.flag ? 1 : 0) == 0;
      if (!out.flag) fprintf(stderr, "vra_runner: KvCache swap failed: %s\n", vra_engine_last_error(eng));
    } else {
      fprintf(stderr, "vra_runner: message %s is not served on this path: ignored\n", m.name.c_str());
      continue;
    }
    send_msg(fd, out);
  }
  vra_engine_destroy(eng);
  if (comm) vra_comm_destroy(comm);
  close(fd);
  return 0;
}
