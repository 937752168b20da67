// model.cpp — see model.h.
#include "model.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../csrc/attn_launch.h"
#include "../csrc/gemm_dense_launch.h"
#include "../csrc/gemm_launch.h"
#include "../csrc/scratch.h"
#include "host_utils.h"

// VRA_X_FRAG=0 / vra_debug_set_x_frag(0): steps of 5..32 rows read h row-major in kernel W (no fragment-order copy)
static int g_x_frag = [] {
  const char* e = getenv("VRA_X_FRAG");
  return e ? atoi(e) : 1;
}();
extern "C" void vra_debug_set_x_frag(int on) { g_x_frag = on; }
namespace vra {

Model::Model(const vra_model_config& mc, const vra_engine_config& ec) : mc_(mc), ec_(ec) {
  rank_ = ec.tp_world_size > 1 ? ec.tp_rank : 0;
  world_ = ec.tp_world_size > 1 ? ec.tp_world_size : 1;
  hq_ = mc.num_heads / world_;
  hkv_ = mc.num_kv_heads >= world_ ? mc.num_kv_heads / world_ : 1;  // kv_head_shard (distributed.rs:498-538)
  inter_ = mc.intermediate_size / world_;
  dt_ = mc.dtype;
  es_ = 2;
  qk_norm_mode_ = mc.qk_norm == 1 || mc.qk_norm == 2 ? mc.qk_norm : 0;
  layers_.resize(mc.num_layers);
  // tensor-parallel preconditions (kv_head_shard bails, distributed.rs:513-536; shard() needs even splits; row-parallel
  // layers shard K in whole scale groups and whole 128-row weight tiles)
  if (world_ > 1) {
    const int g = mc.quant_method != 0 ? (mc.group_size > 0 ? mc.group_size : 0) : 0;
    if (mc.num_heads % world_) tp_error_ = "tensor parallel: num_heads must be divisible by world_size";
    else if (mc.num_kv_heads >= world_ ? mc.num_kv_heads % world_ != 0 : world_ % mc.num_kv_heads != 0)
      tp_error_ = "tensor parallel: kv heads must divide (or be divided by) world_size";
    else if (mc.intermediate_size % world_) tp_error_ = "tensor parallel: intermediate_size must be divisible by world_size";
    else if (rank_ < 0 || rank_ >= world_) tp_error_ = "tensor parallel: rank out of range";
    else if (mc.quant_method != 0 && ((hq_ * mc.head_dim) % 128 || inter_ % 128 || (g > 0 && ((hq_ * mc.head_dim) % g || inter_ % g))))
      tp_error_ = "tensor parallel: the K shard of o_proj / down_proj must be whole 128-row tiles and whole scale groups";
  }
}
Model::~Model() {
  for (void* p : allocs_) (void)hipFree(p);
}
void* Model::dalloc(size_t bytes) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
    error = "hipMalloc failed for " + std::to_string(bytes) + " bytes";
    return nullptr;
  }
  allocs_.push_back(p);
  // debugging aid: VRA_POISON_ALLOC=1 fills every buffer with 0xFF bytes (NaN as bf16 / f16 / f32) before its first use — a kernel
  // that consumes memory nobody wrote then shows up as NaN logits instead of depending on what the allocator handed out
  static const char* poison = getenv("VRA_POISON_ALLOC");
  if (poison && poison[0] == '1') (void)hipMemset(p, 0xFF, bytes ? bytes : 16);
  return p;
}

// ---------------------------------------------------------------------------------------------
// weights
// ---------------------------------------------------------------------------------------------
bool Model::qlinear_synth(QLinear& l, int K, int N, bool bias, uint64_t seed) {
  // BASELINE.md / SURVEY §8d synthetic recipe: qweight uniform u32, GPTQ zeros 8 (stored 0x77777777),
  // AWQ zeros concentrated on 8 (vra_fill_awq_zeros; uniform nibbles until round 3), scales ~ U(0.002, 0.02), dense weights ~ N(0, 0.02).
  l.K = K;
  l.N = N;
  l.quant = mc_.quant_method != 0;
  l.awq = mc_.quant_method == 2;
  if (l.quant) {
    const int g = mc_.group_size > 0 ? mc_.group_size : K;
    const size_t words = (size_t)(K / 8) * N;
    if (!l.w && !(l.w = dalloc(words * 4))) return false;  // (q/k/v: carved out of the layer's fused block)
    vra_fill_hash_u32((uint32_t*)l.w, (int64_t)words, seed, 0);  // tiled layout directly (codes are iid)
    const size_t ns = (size_t)(K / g) * N;
    if (!(l.scales = dalloc(ns * es_))) return false;
    vra_fill_uniform(l.scales, (int64_t)ns, seed + 1, 0.002f, 0.02f, dt_, 0);
    weight_bytes_ += words * 4 + ns * es_;
    if (l.awq) {
      if (!(l.qzeros = (uint32_t*)dalloc(ns / 8 * 4))) return false;
      vra_fill_awq_zeros(l.qzeros, (int64_t)(ns / 8), seed + 2, 0);
      weight_bytes_ += ns / 8 * 4;
    }
  } else {
    if (!(l.w = dalloc((size_t)K * N * es_))) return false;
    vra_fill_normal(l.w, (int64_t)K * N, seed, 0.f, 0.02f, dt_, 0);
    weight_bytes_ += (size_t)K * N * es_;
  }
  if (bias) {
    if (!(l.bias = dalloc((size_t)N * es_))) return false;
    vra_fill_normal(l.bias, N, seed + 3, 0.f, 0.02f, dt_, 0);
  }
  return true;
}

static void* upload(Model* m, std::vector<void*>& allocs, const void* host, size_t bytes, std::string& err) {
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) {
    err = "hipMalloc failed";
    return nullptr;
  }
  allocs.push_back(p);
  if (hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) {
    err = "hipMemcpy failed";
    return nullptr;
  }
  (void)m;
  return p;
}

bool Model::init_synthetic(uint64_t seed) {
  const int H = mc_.hidden_size, D = mc_.head_dim, V = mc_.vocab_size;
  // TP: every rank draws its own shard (seed offset by rank) — good for throughput runs; parity runs
  // load explicit tensors through load_tensor(), which shards exactly like the reference.
  const uint64_t base = seed + (uint64_t)rank_ * 1000003ull;
  if (!(embed_ = dalloc((size_t)V * H * es_))) return false;
  vra_fill_normal(embed_, (int64_t)V * H, seed + 7, 0.f, 0.02f, dt_, 0);
  if (!(final_norm_ = dalloc((size_t)H * es_))) return false;
  vra_fill_normal(final_norm_, H, seed + 8, 1.f, 0.02f, dt_, 0);
  if (mc_.tie_word_embeddings) {
    lm_head_.K = H;
    lm_head_.N = V;
    lm_head_.w = embed_;
  } else {
    lm_head_.K = H;
    lm_head_.N = V;
    if (!(lm_head_.w = dalloc((size_t)V * H * es_))) return false;
    vra_fill_normal(lm_head_.w, (int64_t)V * H, seed + 9, 0.f, 0.02f, dt_, 0);
  }
  weight_bytes_ += (size_t)V * H * es_;
  for (int l = 0; l < mc_.num_layers; l++) {
    LayerWeights& L = layers_[l];
    const uint64_t s = base + 1234 + (uint64_t)l * 64;
    if (!(L.attn_norm = dalloc((size_t)H * es_)) || !(L.ffn_norm = dalloc((size_t)H * es_))) return false;
    vra_fill_normal(L.attn_norm, H, seed + 100 + l * 2, 1.f, 0.02f, dt_, 0);
    vra_fill_normal(L.ffn_norm, H, seed + 101 + l * 2, 1.f, 0.02f, dt_, 0);
    if (qk_norm_mode_) {
      const size_t nq = qk_norm_mode_ == 2 ? (size_t)hq_ * D : (size_t)D, nk = qk_norm_mode_ == 2 ? (size_t)hkv_ * D : (size_t)D;
      if (!(L.q_norm = dalloc(nq * es_)) || !(L.k_norm = dalloc(nk * es_))) return false;
      vra_fill_normal(L.q_norm, (int64_t)nq, seed + 5000 + l * 2, 1.f, 0.05f, dt_, 0);
      vra_fill_normal(L.k_norm, (int64_t)nk, seed + 5001 + l * 2, 1.f, 0.05f, dt_, 0);
    }
    const bool qb = mc_.attention_bias != 0;
    if (mc_.quant_method != 0) {  // q | k | v tiles contiguous
      const size_t wq = (size_t)(H / 8) * hq_ * D * 4, wk = (size_t)(H / 8) * hkv_ * D * 4;
      if (!(L.qkv_w = dalloc(wq + 2 * wk))) return false;
      L.q.w = L.qkv_w;
      L.k.w = (char*)L.qkv_w + wq;
      L.v.w = (char*)L.qkv_w + wq + wk;
    }
    if (!qlinear_synth(L.q, H, hq_ * D, qb, s + 0) || !qlinear_synth(L.k, H, hkv_ * D, qb, s + 4) ||
        !qlinear_synth(L.v, H, hkv_ * D, qb, s + 8) || !qlinear_synth(L.o, hq_ * D, H, false, s + 12) ||
        !qlinear_synth(L.gate, H, inter_, false, s + 16) || !qlinear_synth(L.up, H, inter_, false, s + 20) ||
        !qlinear_synth(L.down, inter_, H, false, s + 24))
      return false;
  }
  if (hipDeviceSynchronize() != hipSuccess) {
    error = "synthetic init: device error";
    return false;
  }
  return finalize_weights();
}

// 2-D host slice: rows [r0,r1) x cols [c0,c1) of a row-major [rows, cols] array
static std::vector<uint8_t> slice2d(const void* src, int64_t cols, size_t es, int64_t r0, int64_t r1, int64_t c0, int64_t c1) {
  std::vector<uint8_t> out((size_t)(r1 - r0) * (c1 - c0) * es);
  for (int64_t r = r0; r < r1; r++)
    memcpy(out.data() + (size_t)(r - r0) * (c1 - c0) * es, (const uint8_t*)src + ((size_t)r * cols + c0) * es, (size_t)(c1 - c0) * es);
  return out;
}

bool Model::load_tensor(const std::string& name, const void* host, const int64_t* shape, int ndim, int elem_bytes) {
  if (!tp_error_.empty()) {
    error = tp_error_;
    return false;
  }
  const int H = mc_.hidden_size;
  auto up1 = [&](void*& dst, size_t n_elems) {
    dst = upload(this, allocs_, host, n_elems * (size_t)elem_bytes, error);
    return dst != nullptr;
  };
  if (name == "model.embed_tokens.weight") {
    if (!up1(embed_, (size_t)shape[0] * shape[1])) return false;
    if (mc_.tie_word_embeddings) {
      lm_head_.K = H;
      lm_head_.N = mc_.vocab_size;
      lm_head_.w = embed_;
    }
    return true;
  }
  if (name == "model.norm.weight") return up1(final_norm_, (size_t)shape[0]);
  if (name == "lm_head.weight") {
    lm_head_.K = H;
    lm_head_.N = mc_.vocab_size;
    return up1(lm_head_.w, (size_t)shape[0] * shape[1]);
  }
  // model.layers.{i}.<sub>
  int li = -1;
  char sub[160] = {0};
  if (sscanf(name.c_str(), "model.layers.%d.%150s", &li, sub) != 2 || li < 0 || li >= mc_.num_layers) {
    error = "unknown tensor name: " + name;
    return false;
  }
  LayerWeights& L = layers_[li];
  std::string s(sub);
  if (s == "input_layernorm.weight") return up1(L.attn_norm, (size_t)shape[0]);
  if (s == "post_attention_layernorm.weight") return up1(L.ffn_norm, (size_t)shape[0]);
  if (s == "self_attn.q_norm.weight" || s == "self_attn.k_norm.weight") {
    // per head ([head_dim], replicated) or over the full row ([heads * head_dim], sharded with the heads: attention.rs:567-590;
    // kv heads replicate when num_kv_heads < world, distributed.rs:526-537)
    const bool is_q = s[10] == 'q';
    void*& dst = is_q ? L.q_norm : L.k_norm;
    const int64_t n = shape[0], D = mc_.head_dim, heads = is_q ? mc_.num_heads : mc_.num_kv_heads;
    if (n == D) {
      if (qk_norm_loaded_ && qk_norm_mode_ == 2) return error = "q_norm / k_norm: per-head and full-row weights mixed", false;
      qk_norm_mode_ = 1, qk_norm_loaded_ = true;
      return up1(dst, (size_t)n);
    }
    if (n != heads * D) return error = "q_norm / k_norm weight of " + std::to_string(n) + " entries: neither head_dim nor heads * head_dim", false;
    if (qk_norm_loaded_ && qk_norm_mode_ == 1) return error = "q_norm / k_norm: per-head and full-row weights mixed", false;
    qk_norm_mode_ = 2, qk_norm_loaded_ = true;
    int nshard = world_, ishard = rank_;
    if (!is_q && mc_.num_kv_heads < world_) {
      nshard = mc_.num_kv_heads;
      ishard = rank_ / (world_ / mc_.num_kv_heads);
    }
    const int64_t per = n / nshard;
    dst = upload(this, allocs_, (const uint8_t*)host + (size_t)(per * ishard) * elem_bytes, (size_t)per * elem_bytes, error);
    return dst != nullptr;
  }
  struct Target {
    const char* prefix;
    QLinear* l;
    bool row_parallel;  // shard K (o_proj, down_proj) vs shard N
    bool is_kv;
  } targets[] = {{"self_attn.q_proj.", &L.q, false, false},    {"self_attn.k_proj.", &L.k, false, true},
                 {"self_attn.v_proj.", &L.v, false, true},     {"self_attn.o_proj.", &L.o, true, false},
                 {"mlp.gate_proj.", &L.gate, false, false},    {"mlp.up_proj.", &L.up, false, false},
                 {"mlp.down_proj.", &L.down, true, false}};
  for (auto& t : targets) {
    const size_t pl = strlen(t.prefix);
    if (s.compare(0, pl, t.prefix) != 0) continue;
    const std::string leaf = s.substr(pl);
    QLinear& l = *t.l;
    // shard index/count: kv projections replicate heads when num_kv_heads < world (distributed.rs:526-537)
    int nshard = world_, ishard = rank_;
    if (t.is_kv && mc_.num_kv_heads < world_) {
      nshard = mc_.num_kv_heads;
      ishard = rank_ / (world_ / mc_.num_kv_heads);
    }
    const int64_t R = shape[0], Ccols = ndim > 1 ? shape[1] : 1;
    auto shard_upload = [&](bool shard_rows, bool shard_cols, void*& dst, int* out_r, int* out_c) {
      int64_t r0 = 0, r1 = R, c0 = 0, c1 = Ccols;
      if ((shard_rows && R % nshard) || (shard_cols && Ccols % nshard)) {
        error = "tensor parallel: " + name + " does not split evenly over " + std::to_string(nshard) + " shards";
        return false;
      }
      if (shard_rows) {
        r0 = R / nshard * ishard;
        r1 = r0 + R / nshard;
      }
      if (shard_cols) {
        c0 = Ccols / nshard * ishard;
        c1 = c0 + Ccols / nshard;
      }
      auto buf = slice2d(host, Ccols, (size_t)elem_bytes, r0, r1, c0, c1);
      dst = upload(this, allocs_, buf.data(), buf.size(), error);
      if (out_r) *out_r = (int)(r1 - r0);
      if (out_c) *out_c = (int)(c1 - c0);
      return dst != nullptr;
    };
    const bool sharded = world_ > 1;
    if (leaf == "weight") {  // dense [N, K]: column-parallel shards dim 0, row-parallel dim 1
      l.quant = false;
      int r, c;
      if (!shard_upload(sharded && !t.row_parallel, sharded && t.row_parallel, l.w, &r, &c)) return false;
      l.N = r;
      l.K = c;
      return true;
    }
    if (leaf == "qweight") {  // gptq [K/8, N] / awq [K, N/8]: packed tensors are stored [in, out] (wna16.rs:35-40)
      l.quant = true;
      l.awq = mc_.quant_method == 2;
      if (!shard_upload(sharded && t.row_parallel, sharded && !t.row_parallel, l.raw_qweight, &l.raw_rows, &l.raw_cols)) return false;
      if (l.awq) {
        l.K = l.raw_rows;
        l.N = l.raw_cols * 8;
      } else {
        l.K = l.raw_rows * 8;
        l.N = l.raw_cols;
      }
      return true;
    }
    if (leaf == "scales") {
      int r, c;
      return shard_upload(sharded && t.row_parallel, sharded && !t.row_parallel, l.scales, &r, &c);
    }
    if (leaf == "qzeros") {
      int r, c;
      void* p = nullptr;
      if (!shard_upload(sharded && t.row_parallel, sharded && !t.row_parallel, p, &r, &c)) return false;
      l.qzeros = (uint32_t*)p;
      return true;
    }
    if (leaf == "g_idx") return true;  // desc_act=false only: trivial map, ignored (Appendix A7)
    if (leaf == "bias") {
      int r, c;
      // bias [N]: sharded with the output dim for column-parallel layers
      int64_t n0 = 0, n1 = R;
      if (sharded && !t.row_parallel) {
        n0 = R / nshard * ishard;
        n1 = n0 + R / nshard;
      }
      (void)r;
      (void)c;
      l.bias = upload(this, allocs_, (const uint8_t*)host + (size_t)n0 * elem_bytes, (size_t)(n1 - n0) * elem_bytes, error);
      return l.bias != nullptr;
    }
    error = "unknown tensor leaf: " + name;
    return false;
  }
  error = "unknown tensor name: " + name;
  return false;
}

bool Model::finalize_weights() {
  if (finalized_) return true;
  if (!tp_error_.empty()) {  // constructor-time validation
    error = tp_error_;
    return false;
  }
  // one-time repack of checkpoint-format qweights into the CDNA4 tile layout (MarlinRepack,
  // src/utils/gptq.rs:266-360, invoked from wna16.rs:220-224 at load time)
  auto fin = [&](QLinear& l, const char* what) {
    if (l.raw_qweight) {
      const size_t words = (size_t)l.raw_rows * l.raw_cols;
      if (!l.w && !(l.w = dalloc(words * 4))) return false;
      if (l.awq) awq_repack(l.raw_qweight, l.w, l.raw_rows, l.raw_cols, 4, 0);
      else gptq_repack(l.raw_qweight, l.w, l.raw_rows, l.raw_cols, 0);
      const char* e = vra_last_error();
      if (e && e[0]) {
        error = std::string("repack ") + what + ": " + e;
        return false;
      }
      weight_bytes_ += words * 4;
      if (hipDeviceSynchronize() != hipSuccess) {
        error = std::string("repack ") + what + ": device error";
        return false;
      }
      for (size_t i = 0; i < allocs_.size(); i++)  // the checkpoint-format copy is not needed any more
        if (allocs_[i] == l.raw_qweight) {
          (void)hipFree(allocs_[i]);
          allocs_.erase(allocs_.begin() + (long)i);
          break;
        }
      l.raw_qweight = nullptr;
    }
    if (!l.w) {
      error = std::string("missing weight: ") + what;
      return false;
    }
    if (l.quant && !l.scales) {
      error = std::string("missing scales: ") + what;
      return false;
    }
    if (l.quant && l.awq && !l.qzeros) {
      error = std::string("missing qzeros (awq): ") + what;
      return false;
    }
    return true;
  };
  for (auto& L : layers_) {
    if (!L.attn_norm || !L.ffn_norm) {
      error = "missing layer norm weights";
      return false;
    }
    if ((L.q_norm == nullptr) != (L.k_norm == nullptr)) {
      error = "q_norm without k_norm (or the reverse)";
      return false;
    }
    if (L.q.raw_qweight && L.k.raw_qweight && L.v.raw_qweight && !L.qkv_w) {  // q | k | v tiles contiguous
      const size_t wq = (size_t)L.q.raw_rows * L.q.raw_cols * 4, wk = (size_t)L.k.raw_rows * L.k.raw_cols * 4,
                   wv = (size_t)L.v.raw_rows * L.v.raw_cols * 4;
      if (!(L.qkv_w = dalloc(wq + wk + wv))) return false;
      L.q.w = L.qkv_w;
      L.k.w = (char*)L.qkv_w + wq;
      L.v.w = (char*)L.qkv_w + wq + wk;
    }
    if (!fin(L.q, "q_proj") || !fin(L.k, "k_proj") || !fin(L.v, "v_proj") || !fin(L.o, "o_proj") || !fin(L.gate, "gate_proj") ||
        !fin(L.up, "up_proj") || !fin(L.down, "down_proj"))
      return false;
  }
  if (!embed_ || !final_norm_ || !lm_head_.w) {
    error = "missing embed/norm/lm_head";
    return false;
  }
  // rotary tables in the model dtype (llama.rs:179-189; rotary_emb.rs:32-73,208-278)
  const int half = mc_.head_dim / 2;
  rope_rows_ = vra_rope_table_rows(&mc_);  // yarn / dynamic: longer than max_position_embeddings
  std::vector<float> c((size_t)rope_rows_ * half), s((size_t)rope_rows_ * half);
  vra_rope_tables_f32(&mc_, rope_rows_, c.data(), s.data());
  std::vector<uint16_t> cb(c.size()), sb(s.size());
  for (size_t i = 0; i < c.size(); i++) {
    cb[i] = dt_ == VRA_BF16 ? host_f32_to_bf16(c[i]) : host_f32_to_f16(c[i]);
    sb[i] = dt_ == VRA_BF16 ? host_f32_to_bf16(s[i]) : host_f32_to_f16(s[i]);
  }
  cos_ = upload(this, allocs_, cb.data(), cb.size() * 2, error);
  sin_ = upload(this, allocs_, sb.data(), sb.size() * 2, error);
  if (!cos_ || !sin_) return false;
  if (!vra_scratch_init()) {
    error = "scratch allocation failed";
    return false;
  }
  if (!build_decode_streams()) return false;
  // the lm_head also in tile-major form for the decode GEMV kernels (one contiguous KiB per wave load instead of 16 half lines:
  // gemv.cuh GemvArgs::dense_tiled); the row-major tensor stays for the embedding gather of tied models and the prefill GEMM
  static const char* lt_env = getenv("VRA_LM_HEAD_TILED");
  if (!(lt_env && lt_env[0] == '0') && !lm_head_.quant && lm_head_.N % 16 == 0 && lm_head_.K % 128 == 0) {
    if (!lm_head_tiled_) {  // (+1 copy of the lm_head in HBM: counted)
      if (!(lm_head_tiled_ = dalloc((size_t)lm_head_.N * lm_head_.K * es_))) return false;
      weight_bytes_ += (size_t)lm_head_.N * lm_head_.K * es_;
    }
    vra_dense_tile_weights(lm_head_.w, lm_head_tiled_, lm_head_.N, lm_head_.K, 0);
    if (const char* e = vra_last_error(); e && e[0]) {
      error = std::string("lm_head tiling: ") + e;
      vra_clear_error();
      return false;
    }
  }
  finalized_ = hipDeviceSynchronize() == hipSuccess;
  return finalized_;
}

bool Model::init_kv_cache(int num_blocks) {
  // per layer: K [NB, Hkv, BS, D], V [NB, Hkv, D, BS] (kvcache_allocator.rs:737-932 shapes, re-laid)
  num_blocks_ = num_blocks;
  const size_t per = (size_t)num_blocks * hkv_ * ec_.block_size * mc_.head_dim * (ec_.fp8_kvcache ? 1 : es_);
  kc_.resize(mc_.num_layers);
  vc_.resize(mc_.num_layers);
  for (int l = 0; l < mc_.num_layers; l++) {
    if (!(kc_[l] = dalloc(per)) || !(vc_[l] = dalloc(per))) return false;
    (void)hipMemset(kc_[l], 0, per);
    (void)hipMemset(vc_[l], 0, per);
  }
  if (hipDeviceSynchronize() != hipSuccess) return false;
  return true;
}

bool Model::init_buffers(int max_tokens, int max_seqs) {
  max_tokens_ = max_tokens;
  max_seqs_ = max_seqs;
  const size_t T = max_tokens, H = mc_.hidden_size, D = mc_.head_dim;
  if (!(h_ = dalloc(T * H * es_)) || !(xn_ = dalloc(T * H * es_)) || !(q_ = dalloc(T * hq_ * D * es_)) ||
      !(k_ = dalloc(T * hkv_ * D * es_)) || !(v_ = dalloc(T * hkv_ * D * es_)) || !(attn_ = dalloc(T * hq_ * D * es_)) ||
      !(act_ = dalloc(T * inter_ * es_)) || !(tmp_ = dalloc(T * H * es_)) || !(last_ = dalloc((size_t)max_seqs * H * es_)) ||
      !(logits_ = (float*)dalloc((size_t)max_seqs * mc_.vocab_size * 4)))
    return false;
  const size_t am_bytes = ((size_t)GEMV_AM_COUNTER + 8) * 8;
  if (!(argmax_ws_ = (unsigned long long*)dalloc(am_bytes)) || hipMemset(argmax_ws_, 0, am_bytes) != hipSuccess) return false;
  if (!(bench_tokens_ = (uint32_t*)dalloc((size_t)max_seqs * 4))) return false;  // launch_gemm(4, ..): the lm_head microbenchmark's tokens
  if (mc_.quant_method == 0) {
    if (!(gate_ = dalloc(T * inter_ * es_)) || !(up_ = dalloc(T * inter_ * es_))) return false;
  }
  const size_t ws = vra_paged_attention_decode_workspace_bytes(max_seqs, hq_, mc_.head_dim, vra_rope_table_rows(&mc_));
  if (!(attn_ws_ = dalloc(ws))) return false;
  if (H % 128 == 0) {
    const size_t fb = (size_t)(H / 128) * 2 * 4096;
    if (!(hfrag_ = dalloc(fb)) || hipMemset(hfrag_, 0, fb) != hipSuccess) return false;
    const size_t sqb = (size_t)GW_PRE_PARTS * 32 * sizeof(float);
    // ready-made operands x̃ = round(h·γ) travel through memory in the model dtype WITHOUT rstd: bf16 only (f16 would leave its
    // exponent range where round(h·rstd·γ) does not) — an f16 model keeps the in-kernel norm of kernel W at 5..32 rows
    if (dt_ == VRA_BF16) {
      if (!(pre_o_ = dalloc(fb)) || hipMemset(pre_o_, 0, fb) != hipSuccess || !(pre_d_ = dalloc(fb)) || hipMemset(pre_d_, 0, fb) != hipSuccess) return false;
      if (!(sq_o_ = (float*)dalloc(sqb)) || hipMemset(sq_o_, 0, sqb) != hipSuccess || !(sq_d_ = (float*)dalloc(sqb)) || hipMemset(sq_d_, 0, sqb) != hipSuccess)
        return false;
      if (!(pre_e_ = dalloc(fb)) || hipMemset(pre_e_, 0, fb) != hipSuccess || !(sq_e_ = (float*)dalloc(sqb)) || hipMemset(sq_e_, 0, sqb) != hipSuccess) return false;
    }
  }
  if (inter_ % 128 == 0) {
    const size_t fb = (size_t)(inter_ / 128) * 2 * 4096;
    if (!(actfrag_ = dalloc(fb)) || hipMemset(actfrag_, 0, fb) != hipSuccess) return false;
  }
  if ((hq_ * mc_.head_dim) % 128 == 0) {
    const size_t fb = (size_t)(hq_ * mc_.head_dim / 128) * 2 * 4096;
    if (!(afrag_ = dalloc(fb)) || hipMemset(afrag_, 0, fb) != hipSuccess) return false;
  }
  // the scratch tensor of the dense prefill path (the dequantised weights of ONE GEMM; process-wide, csrc/gemm_dense.hip) is sized for
  // this model's largest GEMM now, before the KV cache is planned from what is left of the device memory
  if (vra_dense_prefill_min_rows() > 0 && max_tokens >= vra_dense_prefill_min_rows()) {
    size_t need = 0;
    for (size_t l = 0; l < layers_.size(); l++)
      for (int k = 0; k < 4; k++)
        if (dense_shape_ok((int)l, k)) need = std::max(need, dense_bytes((int)l, k));
    if (need) (void)vra_dense_scratch(need, 0);  // (unavailable: the GEMMs keep to the int4 kernels)
  }
  // resident dequantised weights for long prefills (model.h wd_res_): opt-in, only for engines whose steps can reach the row rule
  {
    const char* res = getenv("VRA_DENSE_PREFILL_RESIDENT");
    if (res && atoi(res) && vra_dense_prefill_min_rows() > 0 && max_tokens >= vra_dense_prefill_min_rows()) {
      wd_res_.assign(layers_.size() * 4, nullptr);
      for (size_t l = 0; l < layers_.size(); l++)
        for (int k = 0; k < 4; k++) {
          if (!dense_shape_ok((int)l, k)) continue;
          void* p = dalloc(dense_bytes((int)l, k));
          if (!p) return false;
          dense_fill((int)l, k, p, 0);
          wd_res_[l * 4 + k] = p;
        }
      if (vra_last_error()[0]) {
        error = std::string("resident dequantised weights: ") + vra_last_error();
        vra_clear_error();
        return false;
      }
    }
  }
  // (hipMemset of device memory may return before the fill has run: nothing of this may still be pending when the engine's
  // non-blocking stream starts its first forward)
  if (hipDeviceSynchronize() != hipSuccess) return false;
  return true;
}

// ---------------------------------------------------------------------------------------------
// resident dequantised weights of the dense prefill path (model.h wd_res_; csrc/gemm_dense.cuh)
// ---------------------------------------------------------------------------------------------
bool Model::dense_shape_ok(int layer, int kind) const {
  if (layer < 0 || layer >= (int)layers_.size()) return false;
  const LayerWeights& L = layers_[layer];
  auto q4 = [&](const QLinear& l) { return l.quant && l.w && l.K % 128 == 0 && l.N % 16 == 0; };
  switch (kind) {
    case 0: return q4(L.q) && q4(L.k) && q4(L.v) && L.k.K == L.q.K && L.v.K == L.q.K && L.q.N % 64 == 0 && L.k.N % 64 == 0 && L.v.N % 64 == 0 &&
                   L.k.awq == L.q.awq && L.v.awq == L.q.awq;
    case 1: return q4(L.o);
    case 2: return q4(L.gate) && q4(L.up) && L.up.K == L.gate.K && L.up.N == L.gate.N && !L.gate.bias && !L.up.bias;
    case 3: return q4(L.down);
  }
  return false;
}
size_t Model::dense_bytes(int layer, int kind) const {
  const LayerWeights& L = layers_[layer];
  switch (kind) {
    case 0: return (size_t)L.q.K * (L.q.N + L.k.N + L.v.N) * es_;
    case 1: return (size_t)L.o.K * L.o.N * es_;
    case 2: return (size_t)L.gate.K * L.gate.N * 2 * es_;
    case 3: return (size_t)L.down.K * L.down.N * es_;
  }
  return 0;
}
void Model::dense_fill(int layer, int kind, void* wd, int64_t stream) {
  const LayerWeights& L = layers_[layer];
  auto one = [&](const QLinear& l, int vfrag0, int vstride) {
    vra_launch_dequant_frag(l.w, l.scales, l.qzeros, wd, l.K, l.N, mc_.group_size, l.awq && l.qzeros != nullptr, VRA_SCALES_ROWMAJOR, dt_, vfrag0, vstride, stream);
  };
  switch (kind) {
    case 0: one(L.q, 0, 1), one(L.k, L.q.N / 16, 1), one(L.v, (L.q.N + L.k.N) / 16, 1); break;
    case 1: one(L.o, 0, 1); break;
    case 2: one(L.gate, 0, 2), one(L.up, 1, 2); break;
    case 3: one(L.down, 0, 1); break;
  }
}
const void* Model::dense_resident(int layer, int kind, int M) const {
  if (wd_res_.empty() || layer < 0 || (size_t)layer * 4 + kind >= wd_res_.size() || kind < 0) return nullptr;
  const int mr = vra_dense_prefill_min_rows();
  return mr > 0 && M >= mr ? wd_res_[(size_t)layer * 4 + kind] : nullptr;
}
int Model::dense_kind_of(const QLinear& l) const {
  if (cur_layer_ < 0 || cur_layer_ >= (int)layers_.size() || !l.w) return -1;
  const LayerWeights& L = layers_[cur_layer_];
  return l.w == L.o.w ? 1 : (l.w == L.down.w ? 3 : -1);
}

// ---------------------------------------------------------------------------------------------
// linear layers
// ---------------------------------------------------------------------------------------------
static bool take_err(std::string& error, const char* where) {
  const char* e = vra_last_error();
  if (e && e[0]) {
    error = std::string(where) + ": " + e;
    vra_clear_error();
    return true;
  }
  return false;
}

bool Model::linear(const QLinear& l0, const void* x, void* out, int M, const void* residual, int64_t stream, bool with_bias, void* out_frag,
                   bool* wrote_frag) {
  QLinear l = l0;
  if (!with_bias) l.bias = nullptr;
  if (wrote_frag) *wrote_frag = false;
  // down_proj of a 5..32-row step on kernel C, leaving its outputs in fragment order as well (the dispatch order of vra_wna16_gemm:
  // kernels E / W and A first — they do not take this shape —, then C)
  if (l.quant && out_frag && M >= 5 && M <= 32 && l.N % 16 == 0 && !vra_gemv_s_fits(1, M, l.K, mc_.group_size, l.N / 16, false) &&
      !vra_gemv_w_fits(1, M, l.K, mc_.group_size, l.N / 16, residual != nullptr, l.bias != nullptr) && !vra_gemv_fits(true, 1, M, l.K, mc_.group_size) &&
      vra_gemm_q4_fits(1, M, l.K, mc_.group_size)) {
    GemmCArgs c = {};
    c.nseg = 1;
    c.seg[0] = GemvSeg{l.w, l.scales, l.qzeros, l.bias, out, l.N, l.N, 0};
    c.x = x, c.x_ld = l.K, c.residual = residual, c.res_ld = l.N;
    c.M = M, c.K = l.K, c.group_size = mc_.group_size, c.n_blocks = l.N / 16;
    c.out_frag = out_frag;
    vra_launch_gemm_q4(c, l.awq && l.qzeros != nullptr, dt_, stream);
    if (wrote_frag) *wrote_frag = true;
    return !take_err(error, "linear (kernel C)");
  }
  if (l.quant) {
    if (const void* wd = dense_resident(cur_layer_, dense_kind_of(l0), M)) {  // long prefill, resident dequantised weights: no pass
      GemmXArgs a = {};
      a.x = x, a.x_ld = l.K, a.wd = wd, a.residual = residual, a.res_ld = l.N;
      a.seg[0] = GemmXSeg{out, l.bias, l.N, 0};
      a.nseg = 1;
      a.M = M, a.NV = l.N, a.K = l.K;
      vra_launch_gemm_dense(a, false, dt_, vra_gemm_dense_tile(M, l.N, l.K), stream);
      return !take_err(error, "linear (gemm_dense, resident weights)");
    }
    vra_wna16_gemm(x, l.w, l.scales, l.qzeros, l.bias, residual, out, M, l.K, l.N, mc_.group_size, l.awq ? 1 : 0,
                   VRA_SCALES_ROWMAJOR, dt_, stream);
  } else if (!residual) {
    vra_dense_gemm(x, l.w, l.bias, out, M, l.K, l.N, dt_, dt_, stream);
  } else {
    // dense + residual: kernel A handles it directly for small M, otherwise GEMM then add
    if (vra_gemv_fits(false, 1, M, l.K, -1)) {
      GemvArgs a = {};
      a.nseg = 1;
      a.seg[0] = GemvSeg{l.w, nullptr, nullptr, l.bias, out, l.N, l.N, 0};
      a.x = x;
      a.x_ld = l.K;
      a.residual = residual;
      a.res_ld = l.N;
      a.M = M;
      a.K = l.K;
      a.group_size = -1;
      vra_launch_gemv(a, false, dt_, stream);
    } else {
      vra_dense_gemm(x, l.w, l.bias, tmp_, M, l.K, l.N, dt_, dt_, stream);
      vra_add(tmp_, residual, out, (int64_t)M * l.N, dt_, stream);
    }
  }
  return !take_err(error, "linear");
}

bool Model::linear_fused_norm(const QLinear* ls, int nl, void* const* outs, const void* x, const void* norm_w, int M, int64_t stream) {
  // RMSNorm (NormX::forward, others.rs:11-29) folded into the GEMV prologue when the whole x fits LDS;
  // q/k/v (Separate projections, attention.rs:224-228,660-669) go out in ONE launch.
  const int K = ls[0].K;
  bool fusable = nl <= GEMV_MAX_SEG && vra_gemv_fits(ls[0].quant, 1, M, K, ls[0].quant ? mc_.group_size : -1);
  for (int i = 1; i < nl; i++) fusable = fusable && ls[i].quant == ls[0].quant && ls[i].K == K;
  if (fusable) {
    GemvArgs a = {};
    a.nseg = nl;
    int blk = 0;
    for (int i = 0; i < nl; i++) {
      a.seg[i] = GemvSeg{ls[i].w, ls[i].scales, ls[i].qzeros, ls[i].bias, outs[i], ls[i].N, ls[i].N, blk};
      blk += (ls[i].N + 15) / 16;
    }
    a.x = x;
    a.x_ld = K;
    a.norm_w = norm_w;
    a.eps = mc_.rms_norm_eps;
    a.M = M;
    a.K = K;
    a.group_size = ls[0].quant ? mc_.group_size : -1;
    a.is_awq = ls[0].awq ? 1 : 0;
    a.scales_layout = VRA_SCALES_ROWMAJOR;
    vra_launch_gemv(a, ls[0].quant, dt_, stream);
    return !take_err(error, "fused norm gemv");
  }
  bool same = ls[0].quant && nl <= GEMV_MAX_SEG;
  for (int i = 1; i < nl; i++) same = same && ls[i].quant && ls[i].K == K && ls[i].awq == ls[0].awq;
  // long prefills (gemm_dense.cuh): q/k/v dequantised once into ONE fragment-order tensor (Marlin's 16-bit weights, gptq.rs:116-178),
  // one launch of the 256-row dense GEMM over the concatenated columns
  if (same && nl <= GX_MAX_SEG) {
    int cols = 0;
    bool seg_ok = true;
    for (int i = 0; i < nl; i++) seg_ok = seg_ok && ls[i].N % 64 == 0, cols += ls[i].N;
    const bool fits = seg_ok && vra_dense_prefill_fits(M, K, cols, mc_.group_size);
    const void* res = fits && nl == 3 && ls[0].w == layers_[cur_layer_].q.w ? dense_resident(cur_layer_, 0, M) : nullptr;
    const void* wd = res ? res : (fits ? vra_dense_scratch((size_t)K * cols * es_, stream) : nullptr);
    if (wd) {
      vra_rms_norm(x, norm_w, xn_, M, K, mc_.rms_norm_eps, dt_, stream);
      GemmXArgs a = {};
      a.x = xn_, a.x_ld = K, a.wd = wd;
      a.nseg = nl;
      int c0 = 0;
      const void *tw[3] = {}, *ts[3] = {}, *tz[3] = {};
      int tn[3] = {}, tf[3] = {}, tst[3] = {1, 1, 1};
      bool zeros = ls[0].awq;
      for (int i = 0; i < nl; i++) {
        tw[i] = ls[i].w, ts[i] = ls[i].scales, tz[i] = ls[i].qzeros, tn[i] = ls[i].N, tf[i] = c0 / 16;
        zeros = zeros && ls[i].qzeros != nullptr;
        a.seg[i] = GemmXSeg{outs[i], ls[i].bias, ls[i].N, c0};
        c0 += ls[i].N;
      }
      if (!res)  // q | k | v dequantised by ONE launch
        vra_launch_dequant_frag_batch(nl, tw, ts, tz, tn, tf, tst, const_cast<void*>(wd), K, mc_.group_size, zeros, VRA_SCALES_ROWMAJOR, dt_, stream);
      a.M = M, a.NV = cols, a.K = K;
      vra_launch_gemm_dense(a, false, dt_, vra_gemm_dense_tile(M, cols, K), stream);
      return !take_err(error, "norm + gemm_dense (segments)");
    }
  }
  // prefill: q/k/v in ONE launch of kernel D when the problem fills the chip — decided before the norm launch, which then also
  // leaves kernel D's row-sum table (one launch less per layer)
  GemmDArgs d = {};
  int mb_d = 0;
  if (same && nl <= 3 && M >= 64) {
    d.w0 = ls[0].w, d.sc0 = ls[0].scales, d.qz0 = ls[0].qzeros, d.bias0 = ls[0].bias;
    d.out = outs[0], d.out_ld = ls[0].N, d.N = ls[0].N;
    d.nseg = nl;
    int cols = ls[0].N;
    for (int i = 1; i < nl; i++) {
      d.xseg[i - 1] = GemvSeg{ls[i].w, ls[i].scales, ls[i].qzeros, ls[i].bias, outs[i], ls[i].N, ls[i].N, cols / 16};
      cols += ls[i].N;
    }
    d.x = xn_, d.x_ld = K, d.M = M, d.K = K, d.group_size = mc_.group_size;
    mb_d = vra_gemm_q4_big_fits(false, M, cols, K, mc_.group_size, &d);
  }
  float* xsum_tbl = mb_d && K % 128 == 0 ? vra_gemm_q4_big_xsum_table(M, K) : nullptr;
  if (xsum_tbl) {
    vra_rms_norm_xsum(x, norm_w, xn_, xsum_tbl, M, K, mc_.rms_norm_eps, dt_, stream);
    d.xsum = xsum_tbl;
  } else {
    vra_rms_norm(x, norm_w, xn_, M, K, mc_.rms_norm_eps, dt_, stream);
  }
  if (same && vra_gemm_q4_fits(1, M, K, mc_.group_size)) {  // decode batches 5..32: q/k/v in ONE launch of kernel C
    GemmCArgs c = {};
    c.nseg = nl;
    int blk = 0;
    for (int i = 0; i < nl; i++) {
      c.seg[i] = GemvSeg{ls[i].w, ls[i].scales, ls[i].qzeros, ls[i].bias, outs[i], ls[i].N, ls[i].N, blk};
      blk += (ls[i].N + 15) / 16;
    }
    c.x = xn_;
    c.x_ld = K;
    c.M = M;
    c.K = K;
    c.group_size = mc_.group_size;
    c.n_blocks = blk;
    vra_launch_gemm_q4(c, ls[0].awq, dt_, stream);
    return !take_err(error, "norm + gemm_q4");
  }
  if (mb_d) {
    vra_launch_gemm_q4_big(d, false, ls[0].awq, mb_d, dt_, stream);
    return !take_err(error, "norm + gemm_q4_big (segments)");
  }
  if (same && nl <= 3 && M > 8) {  // q/k/v in ONE launch of kernel B (k and v alone are 8 workgroups wide)
    GemmBArgs b = {};
    b.w0 = ls[0].w, b.sc0 = ls[0].scales, b.qz0 = ls[0].qzeros, b.bias0 = ls[0].bias;
    b.out = outs[0], b.out_ld = ls[0].N, b.N = ls[0].N;
    int blk = (ls[0].N + 15) / 16;
    b.nseg = nl;
    for (int i = 1; i < nl; i++) {
      b.xseg[i - 1] = GemvSeg{ls[i].w, ls[i].scales, ls[i].qzeros, ls[i].bias, outs[i], ls[i].N, ls[i].N, blk};
      blk += (ls[i].N + 15) / 16;
    }
    b.x = xn_;
    b.x_ld = K;
    b.M = M;
    b.K = K;
    b.group_size = mc_.group_size;
    b.is_awq = ls[0].awq ? 1 : 0;
    b.scales_layout = VRA_SCALES_ROWMAJOR;
    vra_launch_skinny(b, true, false, dt_, stream);
    return !take_err(error, "norm + gemm_skinny (segments)");
  }
  for (int i = 0; i < nl; i++)
    if (!linear(ls[i], xn_, outs[i], M, nullptr, stream)) return false;
  return !take_err(error, "norm + linear");
}

bool Model::gate_up(const LayerWeights& L, const void* x, const void* norm_w, void* act, int M, int64_t stream) {
  // MLP::forward (mlp.rs:451-469): down(act(gate(x)) * up(x)); gate/up Separate for GPTQ/AWQ (mlp.rs:142-146)
  const int K = L.gate.K, N = L.gate.N;
  if (L.gate.quant) {
    if (vra_gemv_fits(true, 2, M, K, mc_.group_size)) {
      GemvArgs a = {};
      a.nseg = 2;
      a.seg[0] = GemvSeg{L.gate.w, L.gate.scales, L.gate.qzeros, nullptr, act, N, N, 0};
      a.seg[1] = GemvSeg{L.up.w, L.up.scales, L.up.qzeros, nullptr, act, N, N, 0};
      a.silu_dual = 1;
      a.x = x;
      a.x_ld = K;
      a.norm_w = norm_w;
      a.eps = mc_.rms_norm_eps;
      a.M = M;
      a.K = K;
      a.group_size = mc_.group_size;
      a.is_awq = L.gate.awq ? 1 : 0;
      vra_launch_gemv(a, true, dt_, stream);
    } else {
      // long prefills: vra_wna16_gate_up_silu takes the dequant pass + dense GEMM (gemm_dense.cuh); else
      // prefill through kernel D: the norm launch also leaves the GEMM's row-sum table (one launch less per layer)
      const bool dense = vra_dense_prefill_fits(M, K, 2 * N, mc_.group_size);
      if (const void* wd = dense && &L == &layers_[cur_layer_] ? dense_resident(cur_layer_, 2, M) : nullptr) {
        vra_rms_norm(x, norm_w, xn_, M, K, mc_.rms_norm_eps, dt_, stream);
        GemmXArgs a = {};
        a.x = xn_, a.x_ld = K, a.wd = wd;
        a.seg[0] = GemmXSeg{act, nullptr, N, 0};
        a.nseg = 1;
        a.M = M, a.NV = 2 * N, a.K = K;
        vra_launch_gemm_dense(a, true, dt_, vra_gemm_dense_tile(M, 2 * N, K), stream);
        return !take_err(error, "gate_up (gemm_dense, resident weights)");
      }
      const int mb = !dense && M >= 64 && K % 128 == 0 ? vra_gemm_q4_big_fits(true, M, N, K, mc_.group_size, nullptr) : 0;
      float* tbl = mb ? vra_gemm_q4_big_xsum_table(M, K) : nullptr;
      if (tbl) {
        vra_rms_norm_xsum(x, norm_w, xn_, tbl, M, K, mc_.rms_norm_eps, dt_, stream);
        GemmDArgs d = {};
        d.w0 = L.gate.w, d.w1 = L.up.w, d.sc0 = L.gate.scales, d.sc1 = L.up.scales, d.qz0 = L.gate.qzeros, d.qz1 = L.up.qzeros;
        d.x = xn_, d.x_ld = K, d.out = act, d.out_ld = N;
        d.M = M, d.N = N, d.K = K, d.group_size = mc_.group_size;
        d.xsum = tbl;
        vra_launch_gemm_q4_big(d, true, L.gate.awq && L.gate.qzeros != nullptr, mb, dt_, stream);
      } else {
        vra_rms_norm(x, norm_w, xn_, M, K, mc_.rms_norm_eps, dt_, stream);
        vra_wna16_gate_up_silu(xn_, L.gate.w, L.gate.scales, L.gate.qzeros, L.up.w, L.up.scales, L.up.qzeros, act, M, K, N,
                               mc_.group_size, L.gate.awq ? 1 : 0, VRA_SCALES_ROWMAJOR, dt_, stream);
      }
    }
    return !take_err(error, "gate_up");
  }
  if (!L.up.quant && L.gate.K == L.up.K && L.gate.N == L.up.N && vra_gemv_fits(false, 2, M, K, -1)) {
    // unquantised decode: norm + gate + up + SiLU*mul in ONE launch (kernel A, dense pair), as the int4 path does
    GemvArgs a = {};
    a.nseg = 2;
    a.seg[0] = GemvSeg{L.gate.w, nullptr, nullptr, L.gate.bias, act, N, N, 0};
    a.seg[1] = GemvSeg{L.up.w, nullptr, nullptr, L.up.bias, act, N, N, 0};
    a.silu_dual = 1;
    a.x = x, a.x_ld = K, a.norm_w = norm_w, a.eps = mc_.rms_norm_eps;
    a.M = M, a.K = K, a.group_size = -1;
    vra_launch_gemv(a, false, dt_, stream);
    return !take_err(error, "gate_up dense pair");
  }
  const QLinear ls[2] = {L.gate, L.up};
  void* outs[2] = {gate_, up_};
  if (!linear_fused_norm(ls, 2, outs, x, norm_w, M, stream)) return false;
  vra_silu_mul(gate_, up_, act, (int64_t)M * N, dt_, stream);
  return !take_err(error, "gate_up dense");
}

// ---------------------------------------------------------------------------------------------
// decode streams of kernel E (gemv_q4s.cuh)
// ---------------------------------------------------------------------------------------------
bool Model::build_decode_streams() {
  if (mc_.quant_method == 0) return true;
  auto one = [&](QLinear& l) {
    if (!l.quant || l.s_um) return true;
    const int g = mc_.group_size > 0 && mc_.group_size < l.K ? mc_.group_size : l.K;
    const int G = l.K / g;
    if (!(l.s_um = dalloc((size_t)G * l.N * es_))) return false;
    vra_scales_to_unit_major(l.scales, l.s_um, G, l.N, 0, 0);
    if (l.awq) {
      if (!(l.z_um = (uint32_t*)dalloc((size_t)G * (l.N / 8) * 4))) return false;
      vra_zeros_to_unit_major(l.qzeros, l.z_um, G, l.N, 0, 0);
    }
    return true;
  };
  for (auto& L : layers_) {
    if (!one(L.o) || !one(L.gate) || !one(L.up) || !one(L.down)) return false;
    if (L.qkv_w && !L.qkv_s_um && L.q.K == L.k.K && L.q.K == L.v.K) {
      const int K = L.q.K, g = mc_.group_size > 0 && mc_.group_size < K ? mc_.group_size : K, G = K / g;
      const int Nt = L.q.N + L.k.N + L.v.N;
      if (!(L.qkv_s_um = dalloc((size_t)G * Nt * es_))) return false;
      vra_scales_to_unit_major(L.q.scales, L.qkv_s_um, G, L.q.N, 0, 0);
      vra_scales_to_unit_major(L.k.scales, L.qkv_s_um, G, L.k.N, L.q.N / 16, 0);
      vra_scales_to_unit_major(L.v.scales, L.qkv_s_um, G, L.v.N, (L.q.N + L.k.N) / 16, 0);
      if (L.q.awq) {
        if (!(L.qkv_z_um = (uint32_t*)dalloc((size_t)G * (Nt / 8) * 4))) return false;
        vra_zeros_to_unit_major(L.q.qzeros, L.qkv_z_um, G, L.q.N, 0, 0);
        vra_zeros_to_unit_major(L.k.qzeros, L.qkv_z_um, G, L.k.N, L.q.N / 16, 0);
        vra_zeros_to_unit_major(L.v.qzeros, L.qkv_z_um, G, L.v.N, (L.q.N + L.k.N) / 16, 0);
      }
    }
  }
  return true;
}

// (K, n_units, ns, fused norm) of decode GEMV `which`
static void gemv_s_shape(const LayerWeights& L, int which, int* K, int* units, int* ns, bool* norm) {
  switch (which) {
    case 0: *K = L.q.K, *units = (L.q.N + L.k.N + L.v.N) / 16, *ns = 1, *norm = true; break;
    case 1: *K = L.o.K, *units = L.o.N / 16, *ns = 1, *norm = false; break;
    case 2: *K = L.gate.K, *units = L.gate.N / 16, *ns = 2, *norm = true; break;
    default: *K = L.down.K, *units = L.down.N / 16, *ns = 1, *norm = false; break;
  }
}
// "the decode GEMV family (kernels E / W) takes launch `which` of an M-row step", from shapes: shared by the model and by the parity
// tests' mirror of the norm order (vra_debug_norm_deferred_mask); `bias`: the launch's projection(s) carry a bias
static bool family_takes(int which, int M, int H, int inter, int hq, int hkv, int D, int group_size, bool bias) {
  const int nq = hq * D, nkv = hkv * D;
  if ((nq | nkv | H | inter) % 16) return false;
  int K, units, ns;
  bool norm;
  switch (which) {
    case 0: K = H, units = (nq + 2 * nkv) / 16, ns = 1, norm = true; break;
    case 1: K = nq, units = H / 16, ns = 1, norm = false; break;
    case 2: K = H, units = inter / 16, ns = 2, norm = true; break;
    default: K = inter, units = H / 16, ns = 1, norm = false; break;
  }
  return vra_gemv_s_fits(ns, M, K, group_size, units, norm) || vra_gemv_w_fits(ns, M, K, group_size, units, which == 1 || which == 3, bias, norm || which == 0);
}
bool Model::gemv_s_ok(int which, int M) const {
  const LayerWeights& L = layers_[0];
  if (!L.q.quant || !L.qkv_s_um || !L.o.s_um || !L.gate.s_um || !L.up.s_um || !L.down.s_um) return false;
  const bool bias = which == 0 ? (L.q.bias || L.k.bias || L.v.bias) : (which == 2 ? (L.gate.bias || L.up.bias) : (which == 1 ? L.o.bias != nullptr : L.down.bias != nullptr));
  return family_takes(which, M, mc_.hidden_size, inter_, hq_, hkv_, mc_.head_dim, mc_.group_size, bias);
}
// which fused-norm launches of an M-row step of layer `layer` apply the RMSNorm factor in their epilogue (model.h); takes(which) =
// "the family takes launch `which` of this step"
template <class F>
static int norm_deferred_rule(int M, int layer, int H, int world, bool x_frag_on, F takes) {
  if (M < 1) return 0;
  if (M <= 4) return (takes(0) ? 1 : 0) | (takes(2) ? 2 : 0);  // kernel E
  // kernel W, 5..32 rows: the launches whose producer (a kernel-W o_proj / down_proj of the same step; the embedding launch for layer 0) left ready-made operands
  if (!x_frag_on || M > 32 || world != 1 || H % 128) return 0;
  int m = 0;
  if (takes(1) && takes(2)) m |= 2;
  if ((layer == 0 || takes(3)) && takes(0)) m |= 1;  // (layer 0's producer is the embedding launch)
  return m;
}
int Model::norm_deferred_mask(int M, int layer) const {
  return norm_deferred_rule(M, layer, mc_.hidden_size, world_, g_x_frag && hfrag_ && pre_o_ && pre_d_ && pre_e_, [&](int which) { return gemv_s_ok(which, M); });
}
// the same from shapes alone (Llama / Qwen2 / Qwen3 checkpoints: q/k/v biases only), for the oracle (oracle/model.py deferred_norm_mask)
extern "C" int32_t vra_debug_norm_deferred_mask(int32_t hidden, int32_t inter_local, int32_t heads_local, int32_t kv_heads_local, int32_t head_dim,
                                                int32_t group_size, int32_t quant, int32_t qkv_bias, int32_t world, int32_t rows, int32_t layer,
                                                int32_t dtype) {
  if (!quant) return 0;
  // (f16 models have no ready-made operands: 5..32-row steps keep the reference's order, Model::init_buffers)
  return norm_deferred_rule(rows, layer, hidden, world, g_x_frag != 0 && dtype == VRA_BF16, [&](int which) {
    return family_takes(which, rows, hidden, inter_local, heads_local, kv_heads_local, head_dim, group_size, which == 0 && qkv_bias != 0);
  });
}
// the argument block of decode GEMV `which` of layer l (shared by the single launch and the two-phase launch)
void Model::gemv_s_args(int l, int which, int M, void* out, const void* residual, GemvSArgs* ap, int* nsp) {
  const LayerWeights& L = layers_[l];
  const int H = mc_.hidden_size;
  GemvSArgs& a = *ap;
  a = GemvSArgs{};
  int K, units, ns;
  bool norm;
  gemv_s_shape(L, which, &K, &units, &ns, &norm);
  *nsp = ns;
  const int g = mc_.group_size > 0 && mc_.group_size < K ? mc_.group_size : K, G = K / g;
  a.s_grp_stride = 16, a.s_unit_stride = 16 * G;  // unit-major copies
  a.z_grp_stride = 2, a.z_unit_stride = 2 * G;
  a.M = M, a.K = K, a.n_units = units;
  a.eps = mc_.rms_norm_eps;
  a.nseg = 1;
  switch (which) {
    case 0:
      a.w[0] = L.qkv_w, a.scales[0] = L.qkv_s_um, a.zeros[0] = L.qkv_z_um;
      a.x = h_, a.x_ld = H, a.norm_w = L.attn_norm;
      a.nseg = 3;
      a.seg[0] = GemvSSeg{q_, L.q.bias, L.q.N, 0};
      a.seg[1] = GemvSSeg{k_, L.k.bias, L.k.N, L.q.N / 16};
      a.seg[2] = GemvSSeg{v_, L.v.bias, L.v.N, (L.q.N + L.k.N) / 16};
      break;
    case 1:
      a.w[0] = L.o.w, a.scales[0] = L.o.s_um, a.zeros[0] = L.o.z_um;
      a.x = attn_, a.x_ld = L.o.K;
      a.seg[0] = GemvSSeg{out, world_ > 1 ? nullptr : L.o.bias, L.o.N, 0};
      a.residual = residual, a.res_ld = L.o.N;
      break;
    case 2:
      a.w[0] = L.gate.w, a.scales[0] = L.gate.s_um, a.zeros[0] = L.gate.z_um;
      a.w[1] = L.up.w, a.scales[1] = L.up.s_um, a.zeros[1] = L.up.z_um;
      a.x = h_, a.x_ld = H, a.norm_w = L.ffn_norm;
      a.nseg = 2;  // pair: seg[0] carries the output and the gate bias, seg[1].bias the up bias
      a.seg[0] = GemvSSeg{act_, L.gate.bias, L.gate.N, 0};
      a.seg[1] = GemvSSeg{act_, L.up.bias, L.gate.N, 0x7fffffff};
      break;
    default:
      a.w[0] = L.down.w, a.scales[0] = L.down.s_um, a.zeros[0] = L.down.z_um;
      a.x = act_, a.x_ld = L.down.K;
      a.seg[0] = GemvSSeg{out, world_ > 1 ? nullptr : L.down.bias, L.down.N, 0};
      a.residual = residual, a.res_ld = L.down.N;
      break;
  }
}
bool Model::gemv_s(int l, int which, int M, void* out, const void* residual, int64_t stream, const void* x_frag, void* out_frag, const PreOps* pre) {
  if (!gemv_s_ok(which, M)) return false;
  const LayerWeights& L = layers_[l];
  GemvSArgs a;
  int ns;
  gemv_s_args(l, which, M, out, residual, &a, &ns);
  if (M > 4 && M <= 32) a.x_frag = x_frag, a.out_frag = out_frag;  // kernel W only (1..4 rows: kernel E reads and writes row-major)
  if (pre && M > 4 && M <= 32) {
    if (pre->consume) a.x_frag = pre->frag, a.x_sq = pre->sq;  // the fragments hold x̃ = round(h * g): rstd in the epilogue
    else a.pre_norm_w = pre->next_norm_w, a.pre_frag = pre->frag, a.pre_sq = pre->sq;
  }
  if (M > 4) vra_launch_gemv_w(a, ns, mc_.group_size, L.q.awq, dt_, stream);  // kernel W: 5..32 rows, K <= 4096
  else vra_launch_gemv_s(a, ns, mc_.group_size, L.q.awq, dt_, stream);
  return !take_err(error, "gemv_s");
}


// ---------------------------------------------------------------------------------------------
// stage snapshots of layer 0 (parity instrumentation, model.h)
// ---------------------------------------------------------------------------------------------
bool Model::snap(int idx, const void* src, size_t bytes, int64_t stream) {
  if (!snap_on_) return true;
  if (snap_cap_[idx] < bytes) {
    if (!(snap_[idx] = dalloc(bytes))) return false;
    snap_cap_[idx] = bytes;
  }
  snap_bytes_[idx] = bytes;
  return hipMemcpyAsync(snap_[idx], src, bytes, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)) == hipSuccess;
}
int64_t Model::read_tp_snapshot(int idx, void* host, int64_t max_bytes, int64_t stream) {
  if (idx < 0 || idx >= 9 || !snap_[idx] || !host || (int64_t)snap_bytes_[idx] > max_bytes) return -1;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(host, snap_[idx], snap_bytes_[idx], hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
  return (int64_t)snap_bytes_[idx];
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
bool Model::forward(const InputMetadata& md, int64_t stream, uint32_t* tokens) {
  const int T = md.n_tokens, B = md.n_seqs, H = mc_.hidden_size, D = mc_.head_dim;
  if (world_ > 1 && !comm_) {
    error = "forward: tensor parallel world_size " + std::to_string(world_) + " without a communicator (vra_engine_set_comm)";
    return false;
  }
  if (T <= 0 || T > max_tokens_ || B <= 0 || B > max_seqs_) {
    error = "forward: batch out of range (tokens " + std::to_string(T) + ", seqs " + std::to_string(B) + ")";
    return false;
  }
  const float scale = 1.0f / sqrtf((float)D);
  const int kv_dt = ec_.fp8_kvcache ? VRA_FP8_E4M3 : dt_;
  error.clear();
  // embed_forward (llama.rs:260-267)
  // (+ steps of 5..32 rows: h also in kernel W's fragment order, GemvSArgs::x_frag)
  const bool use_frag = g_x_frag && hfrag_ && T > 4 && T <= 32 && world_ == 1;
  // (+ steps of 5..32 rows whose q/k/v launch of layer 0 runs on kernel W: its ready-made operands, like the ones every later
  // layer's gets from the down_proj in front of it — one norm order for all layers of a step shape)
  const bool make_e = use_frag && pre_e_ && sq_e_ && mc_.num_layers > 0 && gemv_s_ok(0, T);
  vra_embedding_bump(md.input_ids, embed_, h_, T, H, mc_.vocab_size, dt_, nullptr, use_frag ? hfrag_ : nullptr, make_e ? layers_[0].attn_norm : nullptr,
                     pre_e_, sq_e_, stream);
  hfrag_ok_ = use_frag;
  pre_o_ok_ = pre_d_ok_ = false;
  for (int l = 0; l < mc_.num_layers; l++) {
    const LayerWeights& L = layers_[l];
    cur_layer_ = l;
    // ---- attention block (llama.rs:115-126): norm -> q,k,v -> rope -> cache + attention -> o_proj (+ residual)
    const QLinear qkv[3] = {L.q, L.k, L.v};
    void* outs[3] = {q_, k_, v_};
    bool attn_frag = false;  // the attention output of this layer also exists in fragment order (afrag_)
    PreOps take_d;  // x̃ of this layer's attention norm, left by the previous layer's down_proj (5..32 rows, kernel W)
    take_d.consume = true, take_d.frag = l == 0 ? pre_e_ : pre_d_, take_d.sq = l == 0 ? sq_e_ : sq_d_;  // (layer 0: left by the embedding launch)
    if (!gemv_s(l, 0, T, nullptr, nullptr, stream, hfrag_ok_ ? hfrag_ : nullptr, nullptr, (l == 0 ? make_e : pre_d_ok_) ? &take_d : nullptr)) {
      if (!error.empty()) return false;
      if (!linear_fused_norm(qkv, 3, outs, h_, L.attn_norm, T, stream)) return false;
    }
    // q_norm / k_norm (attention.rs:713-735), in place, before the rotary embedding
    if (L.q_norm && L.k_norm) {
      vra_qk_rms_norm(q_, k_, L.q_norm, L.k_norm, T, hq_, hkv_, D, qk_norm_mode_ == 2, mc_.rms_norm_eps, dt_, stream);
      if (take_err(error, "qk_norm")) return false;
    }
    if (l == snap_layer_ && snap_on_ &&
        !(snap(0, q_, (size_t)T * hq_ * D * es_, stream) && snap(1, k_, (size_t)T * hkv_ * D * es_, stream) && snap(2, v_, (size_t)T * hkv_ * D * es_, stream)))
      return false;
    const int sw = mc_.sliding_window > 0 ? mc_.sliding_window : 0;  // llama.rs:46,284 (Mistral-type checkpoints)
    if (md.is_prefill) {
      // RoPE + KV write in one launch (two in the reference: rotary_emb.rs:88-103, attention.rs:808-820)
      vra_rope_cache_prefill(q_, k_, v_, kc_[l], vc_[l], cos_, sin_, md.positions, md.slot_mapping, T, hq_, hkv_, D, ec_.block_size, dt_,
                             kv_dt, stream);
      vra_paged_attention_prefill_sw(attn_, q_, nullptr, nullptr, kc_[l], vc_[l], md.block_tables, md.context_lens, md.cu_seqlens_q,
                                     nullptr, B, T, md.max_seqlen_q, hq_, hkv_, D, ec_.block_size, md.max_blocks, scale, 0.f, sw, dt_, kv_dt, stream);
    } else if (sw > 0) {
      // sliding window: the three calls of the reference (rotary embedding, reshape_and_cache, paged attention with the window;
      // attention.rs:745-820) — the fused decode launch is the full-causal fast path
      vra_fused_rope(q_, k_, cos_, sin_, md.positions, T, hq_, hkv_, D, D, 0, dt_, dt_, stream);
      vra_reshape_and_cache(k_, v_, kc_[l], vc_[l], md.slot_mapping, T, hkv_, D, ec_.block_size, dt_, kv_dt, stream);
      vra_paged_attention_decode_sw(attn_, q_, kc_[l], vc_[l], md.block_tables, md.context_lens, B, hq_, hkv_, D, ec_.block_size, md.max_blocks,
                                    md.max_context_len, scale, 0.f, sw, attn_ws_, dt_, kv_dt, stream);
    } else {
      // decode: RoPE + KV write + paged attention in ONE launch (three in the reference, attention.rs:745-820)
      attn_frag = use_frag && afrag_ && B == T;
      vra_rope_cache_attention_decode_frag(attn_, q_, k_, v_, kc_[l], vc_[l], cos_, sin_, md.positions, md.slot_mapping, md.block_tables,
                                           md.context_lens, B, hq_, hkv_, D, ec_.block_size, md.max_blocks, md.max_context_len, scale,
                                           attn_ws_, dt_, kv_dt, attn_frag ? afrag_ : nullptr, stream);
    }
    if (take_err(error, "attention")) return false;
    if (l == snap_layer_ && !snap(3, attn_, (size_t)T * hq_ * D * es_, stream)) return false;
    if (world_ > 1) {
      // TensorParallelRowLinear::forward (distributed.rs:438-455): partial GEMM -> all_reduce -> + bias; then the layer's
      // residual add (llama.rs:126) — the last two fused behind the reduction
      if (!gemv_s(l, 1, T, tmp_, nullptr, stream) && (!error.empty() || !linear(L.o, attn_, tmp_, T, nullptr, stream, false))) return false;
      if (l == snap_layer_ && !snap(4, tmp_, (size_t)T * H * es_, stream)) return false;
      vra_all_reduce_fused(comm_, tmp_, h_, L.o.bias, h_, T, H, dt_, stream);
      if (take_err(error, "all_reduce(o_proj)")) return false;
      if (l == snap_layer_ && !snap(5, h_, (size_t)T * H * es_, stream)) return false;
    } else {
      // o_proj writes h: on kernel W (5..32 rows) also its fragment-order copy; any other kernel leaves the copy stale
      if (!error.empty()) return false;
      PreOps make_o;  // ... which also leaves x̃ = round(h * ffn_norm) + the partial sums of squares for the gate/up launch
      make_o.next_norm_w = L.ffn_norm, make_o.frag = pre_o_, make_o.sq = sq_o_;
      const bool w_o = gemv_s(l, 1, T, h_, h_, stream, attn_frag ? afrag_ : nullptr, use_frag ? hfrag_ : nullptr, use_frag && pre_o_ ? &make_o : nullptr);
      if (!error.empty()) return false;
      hfrag_ok_ = use_frag && w_o;
      pre_o_ok_ = use_frag && w_o && pre_o_ != nullptr;
      if (!w_o && !linear(L.o, attn_, h_, T, h_, stream)) return false;
    }
    // ---- MLP block (llama.rs:127-130)
    PreOps take_o;
    take_o.consume = true, take_o.frag = pre_o_, take_o.sq = sq_o_;
    const bool w_gu = gemv_s(l, 2, T, nullptr, nullptr, stream, hfrag_ok_ ? hfrag_ : nullptr, use_frag ? actfrag_ : nullptr, pre_o_ok_ ? &take_o : nullptr);
    pre_o_ok_ = false;
    if (!w_gu && (!error.empty() || !gate_up(L, h_, L.ffn_norm, act_, T, stream))) return false;
    if (l == snap_layer_ && !snap(6, act_, (size_t)T * inter_ * es_, stream)) return false;
    if (world_ > 1) {
      if (!gemv_s(l, 3, T, tmp_, nullptr, stream) && (!error.empty() || !linear(L.down, act_, tmp_, T, nullptr, stream, false))) return false;
      if (l == snap_layer_ && !snap(7, tmp_, (size_t)T * H * es_, stream)) return false;
      vra_all_reduce_fused(comm_, tmp_, h_, L.down.bias, h_, T, H, dt_, stream);
      if (take_err(error, "all_reduce(down_proj)")) return false;
      if (l == snap_layer_ && !snap(8, h_, (size_t)T * H * es_, stream)) return false;
    } else {
      // down_proj writes h: kernel E at 1..4 rows (no copy), kernel C with the fragment-order copy at 5..32, anything else: stale
      if (!error.empty()) return false;
      PreOps make_d;  // x̃ for the NEXT layer's q/k/v (the last layer's consumer is the final norm of the lm_head launch: nothing)
      const bool has_next = l + 1 < mc_.num_layers;
      if (has_next) make_d.next_norm_w = layers_[l + 1].attn_norm, make_d.frag = pre_d_, make_d.sq = sq_d_;
      const bool e_d = gemv_s(l, 3, T, h_, h_, stream, use_frag && w_gu && actfrag_ ? actfrag_ : nullptr, use_frag ? hfrag_ : nullptr,
                              use_frag && has_next && pre_d_ ? &make_d : nullptr);
      if (!error.empty()) return false;
      bool wrote = e_d && T > 4;  // (kernel W, K-sliced: writes the fragment copy; kernel E at 1..4 rows does not)
      pre_d_ok_ = use_frag && has_next && e_d && T > 4 && pre_d_ != nullptr;
      if (!e_d && !linear(L.down, act_, h_, T, h_, stream, true, use_frag ? hfrag_ : nullptr, &wrote)) return false;
      hfrag_ok_ = use_frag && wrote;
    }
  }
  // last token of every sequence (llama.rs:306-310), final norm, lm_head -> f32 (llama.rs:311-320)
  const void* xin = h_;
  int rows = T;
  if (md.is_prefill) {
    vra_index_select_rows(h_, md.last_token_rows, last_, B, H, dt_, stream);
    xin = last_;
    rows = B;
  }
  return lm_head(xin, rows, tokens, stream);
}

// final norm + lm_head -> f32 logits (llama.rs:311-320) and, when `tokens` is given, the greedy tokens (logits_processor.rs:67-70)
bool Model::lm_head(const void* xin, int rows, uint32_t* tokens, int64_t stream) {
  const int H = mc_.hidden_size;
  // 1..3 rows: kernel A (x in LDS); 4..32 rows: the dense W kernel where it fits (measured 0.6 % of the step faster than kernel A at 4..7
  // rows, equal at 1..2), else kernel A / kernel B behind a norm launch
  if (!(rows >= 4 && vra_gemv_dw_fits(rows, H, lm_head_.N)) && vra_gemv_fits(false, 1, rows, H, -1)) {
    GemvArgs a = {};
    a.nseg = 1;
    a.seg[0] = GemvSeg{lm_head_tiled_ ? lm_head_tiled_ : lm_head_.w, nullptr, nullptr, nullptr, logits_, lm_head_.N, lm_head_.N, 0};
    a.dense_tiled = lm_head_tiled_ ? 1 : 0;
    a.x = xin;
    a.x_ld = H;
    a.norm_w = final_norm_;
    a.eps = mc_.rms_norm_eps;
    a.M = rows;
    a.K = H;
    a.group_size = -1;
    a.out_f32 = 1;
    if (tokens && rows <= 8) {  // the greedy tokens out of the same launch (the last workgroup to arrive reduces the candidates)
      a.am_out = tokens, a.am_ws = argmax_ws_;
      tokens = nullptr;
    }
    vra_launch_gemv(a, false, dt_, stream);
  } else if (vra_gemv_dw_fits(rows, H, lm_head_.N)) {  // 9..32 rows: final norm + lm_head in one launch (gemv_dw.cuh)
    GemvDWArgs a = {};
    a.x = xin, a.x_ld = H, a.norm_w = final_norm_, a.eps = mc_.rms_norm_eps;
    a.w = lm_head_tiled_ ? lm_head_tiled_ : lm_head_.w, a.dense_tiled = lm_head_tiled_ ? 1 : 0, a.out = logits_, a.out_ld = lm_head_.N, a.out_f32 = 1;
    a.M = rows, a.K = H, a.n_units = lm_head_.N / 16;
    static const char* dwam_env = getenv("VRA_DW_ARGMAX");  // tuning aid: 0 = separate argmax launch behind the dense W kernel
    if (tokens && !(dwam_env && atoi(dwam_env) == 0)) {  // the greedy tokens out of the same launch (as at <= 8 rows above)
      a.am_out = tokens, a.am_ws = argmax_ws_;
      tokens = nullptr;
    }
    vra_launch_gemv_dw(a, dt_, stream);
  } else {
    vra_rms_norm(xin, final_norm_, xn_, rows, H, mc_.rms_norm_eps, dt_, stream);
    vra_dense_gemm(xn_, lm_head_.w, nullptr, logits_, rows, H, lm_head_.N, dt_, VRA_F32, stream);
  }
  if (tokens) vra_argmax_f32(logits_, tokens, rows, lm_head_.N, stream);
  return !take_err(error, "lm_head");
}

// ---------------------------------------------------------------------------------------------
// roofline microbench hooks
// ---------------------------------------------------------------------------------------------
bool Model::launch_gemm(int which, int layer, int M, int64_t stream) {
  if (layer < 0 || layer >= mc_.num_layers || M < 1 || M > max_tokens_) return false;
  if (which == 4) return M <= max_seqs_ && lm_head(h_, M, bench_tokens_, stream);  // final norm + lm_head (+ greedy tokens), the decode form
  const LayerWeights& L = layers_[layer];
  cur_layer_ = layer;
  // (5..32 rows: x in fragment order where the forward pass reads it that way — the timing does not depend on the values)
  const bool xf = g_x_frag && M > 4 && M <= 32 && world_ == 1;
  const void* x_frag = !xf ? nullptr : (which == 1 ? afrag_ : (which == 3 ? actfrag_ : hfrag_));
  // ... and the launch in the form the forward pass runs it: fragment-order copies of the outputs, o_proj / down_proj as PRODUCERS of the
  // next fused-norm launch's ready-made operands, norm + q/k/v / norm + gate/up as their CONSUMERS where the step's rule says so
  void* out_frag = !xf ? nullptr : (which == 2 ? actfrag_ : (which == 1 || which == 3 ? hfrag_ : nullptr));
  PreOps pre;
  const PreOps* pp = nullptr;
  if (xf && pre_o_ && pre_d_ && pre_e_) {
    const int dm = norm_deferred_mask(M, layer);
    if (which == 0 && (dm & 1)) pre.consume = true, pre.frag = layer == 0 ? pre_e_ : pre_d_, pre.sq = layer == 0 ? sq_e_ : sq_d_, pp = &pre;
    else if (which == 2 && (dm & 2)) pre.consume = true, pre.frag = pre_o_, pre.sq = sq_o_, pp = &pre;
    else if (which == 1) pre.next_norm_w = L.ffn_norm, pre.frag = pre_o_, pre.sq = sq_o_, pp = &pre;
    else if (which == 3 && layer + 1 < mc_.num_layers) pre.next_norm_w = layers_[layer + 1].attn_norm, pre.frag = pre_d_, pre.sq = sq_d_, pp = &pre;
  }
  if (gemv_s(layer, which, M, which == 1 || which == 3 ? tmp_ : nullptr, which == 1 || which == 3 ? h_ : nullptr, stream, x_frag, out_frag, pp)) return true;
  if (!error.empty()) return false;
  switch (which) {
    case 0: {
      const QLinear qkv[3] = {L.q, L.k, L.v};
      void* outs[3] = {q_, k_, v_};
      return linear_fused_norm(qkv, 3, outs, h_, L.attn_norm, M, stream);
    }
    case 1: return linear(L.o, attn_, tmp_, M, h_, stream);
    case 2: return gate_up(L, h_, L.ffn_norm, act_, M, stream);
    case 3: return linear(L.down, act_, tmp_, M, h_, stream);
    default: return false;
  }
}
int64_t Model::gemm_algorithmic_bytes(int which, int M) const {
  // K*N/2 (packed) + (K/g)*N*2 (scales) + (K/g)*N/2 (zeros) + M*K*2 (act) + M*N*2 (out)
  auto one = [&](const QLinear& l) -> int64_t {
    const int64_t K = l.K, N = l.N;
    if (!l.quant) return K * N * 2 + (int64_t)M * K * 2 + (int64_t)M * N * 2;
    const int64_t g = mc_.group_size > 0 ? mc_.group_size : K;
    return K * N / 2 + (K / g) * N * 2 + (K / g) * N / 2 + (int64_t)M * K * 2 + (int64_t)M * N * 2;
  };
  const LayerWeights& L = layers_[0];
  switch (which) {
    case 0: return one(L.q) + one(L.k) + one(L.v);
    case 1: return one(L.o);
    case 2: return one(L.gate) + one(L.up);
    case 3: return one(L.down);
    case 4: return one(lm_head_) + (int64_t)M * lm_head_.N * 2;  // f32 logits: M*N*4 instead of M*N*2
    default: return 0;
  }
}

}  // namespace vra
