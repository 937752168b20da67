// model.h — native restatement of the reference's model contract for this path:
//   LLaMaForCausalLM (src/models/llama.rs:107-131,269-321) and Qwen2 via Qwen3ForCausalLM
//   (src/models/qwen3.rs:75-97,308-371; qkv bias per attention.rs:411-415),
//   Attention::forward_ext (attention.rs:648-839), MLP::forward (mlp.rs:451-469),
//   WNA16 (wna16.rs:25-307), TensorParallel{Column,Row}Linear + kv_head_shard (distributed.rs).
// forward(input_ids, positions, kv_caches, input_metadata) -> f32 logits [n_seqs, vocab].
#pragma once
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/vllm_rs_amd.h"

struct GemvSArgs;  // csrc/gemv_q4s.cuh

namespace vra {

struct QLinear {
  int K = 0, N = 0;  // local (per-rank) shape
  bool quant = false, awq = false;
  void* w = nullptr;             // quant: tiled words; dense: [N, K] 16-bit
  void* scales = nullptr;        // [K/g, N] model dtype, row-major
  uint32_t* qzeros = nullptr;    // awq raw [K/g, N/8]
  void* bias = nullptr;          // [N] or null
  // decode streams of kernel E (gemv_q4s.cuh): unit-major copies made at load — [N/16][K/g][16] scales, [N/16][K/g][2] zeros
  void* s_um = nullptr;
  uint32_t* z_um = nullptr;
  // staging of checkpoint-format pieces until finalize()
  void* raw_qweight = nullptr;
  int raw_rows = 0, raw_cols = 0;
};

struct LayerWeights {
  void* attn_norm = nullptr;
  void* ffn_norm = nullptr;
  // q_norm / k_norm of Qwen3-style checkpoints (attention.rs:538-601): [head_dim] (per head) or this rank's shard of
  // [heads * head_dim] (full row); null: none
  void* q_norm = nullptr;
  void* k_norm = nullptr;
  QLinear q, k, v, o, gate, up, down;
  // q | k | v tiles in ONE allocation (q.w, k.w, v.w point into it): the fused q/k/v launch of kernel E walks one
  // contiguous run of units; their unit-major scales / zeros are fused the same way
  void* qkv_w = nullptr;
  void* qkv_s_um = nullptr;
  uint32_t* qkv_z_um = nullptr;
};

// The data contract between runner and kernels (InputMetadata, runner.rs:1222-1238,1369-1385);
// all pointers are device buffers owned by the runner.
struct InputMetadata {
  bool is_prefill = false;
  int n_tokens = 0;
  int n_seqs = 0;
  const uint32_t* input_ids = nullptr;      // [T]
  const int64_t* positions = nullptr;       // [T]
  const int64_t* slot_mapping = nullptr;    // [T]
  const uint32_t* block_tables = nullptr;   // [B, max_blocks]
  const uint32_t* context_lens = nullptr;   // [B]
  const uint32_t* cu_seqlens_q = nullptr;   // [B+1] (prefill)
  const uint32_t* last_token_rows = nullptr;  // [B] = seqlens[i]-1 (llama.rs:306-310)
  int max_blocks = 0;
  int max_seqlen_q = 0;
  int max_context_len = 0;
};

class Model {
 public:
  Model(const vra_model_config& mc, const vra_engine_config& ec);
  ~Model();
  std::string error;

  // weights
  bool init_synthetic(uint64_t seed);
  bool load_tensor(const std::string& name, const void* host, const int64_t* shape, int ndim, int elem_bytes);
  bool finalize_weights();   // repack staged checkpoint tensors, check completeness
  // kv cache
  bool init_kv_cache(int num_blocks);
  int num_blocks() const { return num_blocks_; }
  // one block of one layer's K (or V) cache: contiguous, [Hkv, BS, D] (or [Hkv, D, BS]) elements
  size_t kv_block_bytes() const { return (size_t)hkv_ * ec_.block_size * mc_.head_dim * (ec_.fp8_kvcache ? 1 : es_); }
  void* k_cache(int l) const { return kc_[l]; }
  void* v_cache(int l) const { return vc_[l]; }
  // activations are sized for max_tokens rows
  bool init_buffers(int max_tokens, int max_seqs);
  void set_comm(void* comm) { comm_ = comm; }
  bool has_comm() const { return comm_ != nullptr; }
  int world() const { return world_; }
  // forward → logits (device, f32 [n_seqs, vocab]) ; returns false on argument error
  // tokens != null: the greedy token of every logits row is written there as well (out of the lm_head launch itself where the
  // kernel offers it, else by an argmax launch)
  bool forward(const InputMetadata& md, int64_t stream, uint32_t* tokens = nullptr);
  float* logits() const { return logits_; }
  const vra_model_config& config() const { return mc_; }
  int local_heads() const { return hq_; }
  int local_kv_heads() const { return hkv_; }
  size_t weight_bytes() const { return weight_bytes_; }
  // microbenchmark hooks (bench.py roofline leg): launch one decode-shaped GEMM of layer `layer`
  // which: 0 qkv(fused norm) 1 o_proj 2 gate_up 3 down 4 lm_head ; M rows
  bool launch_gemm(int which, int layer, int M, int64_t stream);
  bool lm_head(const void* xin, int rows, uint32_t* tokens, int64_t stream);
  int64_t gemm_algorithmic_bytes(int which, int M) const;
  // which fused-norm launches of a step of M rows apply the RMSNorm factor in their EPILOGUE (kernel E at 1..4 rows, gemv_q4s.cuh;
  // the oracle restates that order): bit 0 = norm + q/k/v, bit 1 = norm + gate/up.  0 for every other step.
  // (steps of 5..32 rows, round 5: the launches that take ready-made operands from their producer — gate/up behind a kernel-W
  // o_proj, the q/k/v of layers >= 1 behind a kernel-W down_proj: per LAYER)
  int norm_deferred_mask(int M, int layer = 1) const;
  // Parity instrumentation of the tensor-parallel forward (tests/test_gpu_tp.py, tools/tp8_stress.py): with snapshots on, layer 0
  // of every forward leaves copies of its stages — 0 q, 1 k, 2 v (GEMM outputs, before RoPE), 3 attention output (o_proj's x),
  // 4 o_proj partial (what this rank hands to the all-reduce), 5 h after all-reduce + residual, 6 SiLU(gate)*up (down_proj's x),
  // 7 down_proj partial, 8 h after the second all-reduce — so that a deviating logit can be traced to the first stage and rank
  // that deviates from the oracle.  Off (the default): no buffers, no copies.
  void set_tp_snapshots(int on) { snap_on_ = on != 0; snap_layer_ = on > 0 ? on - 1 : 0; }  // on = 1 + the layer to copy
  int64_t read_tp_snapshot(int idx, void* host, int64_t max_bytes, int64_t stream);  // bytes copied, -1 on error

 private:
  void* dalloc(size_t bytes);
  bool qlinear_synth(QLinear& l, int K, int N, bool bias, uint64_t seed);
  bool linear(const QLinear& l, const void* x, void* out, int M, const void* residual, int64_t stream, bool with_bias = true, void* out_frag = nullptr,
              bool* wrote_frag = nullptr);
  bool linear_fused_norm(const QLinear* ls, int nl, void* const* outs, const void* x, const void* norm_w, int M, int64_t stream);
  bool gate_up(const LayerWeights& L, const void* x, const void* norm_w, void* act, int M, int64_t stream);
  bool build_decode_streams();
  // kernel E launch of one decode GEMV of layer `l` (which: 0 norm+q/k/v, 1 o_proj, 2 norm+gate/up+SiLU*mul, 3 down);
  // false = shape not covered (the caller takes the general path).  `out`/`residual` as for linear().
  // pre (kernel W, 5..32 rows; csrc/gemv_q4s.cuh GemvSArgs::pre_*): a producer launch also leaves the NEXT fused-norm launch's
  // operands (x̃ = round(out * next_norm_w) in fragment order + partial sums of squares); a consumer launch takes them
  struct PreOps {
    const void* next_norm_w = nullptr;  // producer
    void* frag = nullptr;               // producer: where x̃ goes / consumer: where it comes from
    float* sq = nullptr;                // the partial-sum table
    bool consume = false;
  };
  bool gemv_s(int l, int which, int M, void* out, const void* residual, int64_t stream, const void* x_frag = nullptr, void* out_frag = nullptr,
              const PreOps* pre = nullptr);
  bool gemv_s_ok(int which, int M) const;
  void gemv_s_args(int l, int which, int M, void* out, const void* residual, ::GemvSArgs* a, int* ns);
  vra_model_config mc_;
  vra_engine_config ec_;
  bool finalized_ = false;
  std::string tp_error_;  // tensor-parallel preconditions that failed at construction
  int rank_, world_;
  int hq_, hkv_, inter_;  // local sizes
  int dt_;
  size_t es_;
  void* comm_ = nullptr;
  std::vector<void*> allocs_;
  size_t weight_bytes_ = 0;
  // weights
  void* embed_ = nullptr;
  void* final_norm_ = nullptr;
  QLinear lm_head_;
  void* lm_head_tiled_ = nullptr;  // tile-major copy of the lm_head (csrc/gemv.cuh GemvArgs::dense_tiled), made at finalize
  std::vector<LayerWeights> layers_;
  void* cos_ = nullptr;
  void* sin_ = nullptr;
  int rope_rows_ = 0;
  // kv cache
  std::vector<void*> kc_, vc_;
  int num_blocks_ = 0;
  // activations
  int max_tokens_ = 0, max_seqs_ = 0;
  void *h_ = nullptr, *xn_ = nullptr, *q_ = nullptr, *k_ = nullptr, *v_ = nullptr, *attn_ = nullptr, *act_ = nullptr,
       *tmp_ = nullptr, *gate_ = nullptr, *up_ = nullptr, *last_ = nullptr, *attn_ws_ = nullptr;
  float* logits_ = nullptr;
  uint32_t* bench_tokens_ = nullptr;
  unsigned long long* argmax_ws_ = nullptr;  // kernel A's candidate keys + arrival counter (gemv.cuh)
  // the hidden state of a step of up to 32 rows in kernel W's fragment order (csrc/gemv_q4s.cuh GemvSArgs::x_frag), written beside
  // h_ by the launches that produce it (embedding, o_proj on kernel W, down_proj on kernel C) and read by the norm + q/k/v and
  // norm + gate/up launches of kernel W; hfrag_ok_: the copy matches h_ (false after any other writer of h_)
  void* hfrag_ = nullptr;
  bool hfrag_ok_ = false;
  void* actfrag_ = nullptr;  // SiLU(gate) * up of a 5..32-row step in fragment order (down_proj's x on the K-sliced kernel W)
  int qk_norm_mode_ = 0;  // 0 none, 1 per head, 2 full row (set by the config for synthetic weights, by the tensor shape when loading)
  bool qk_norm_loaded_ = false;  // a q_norm / k_norm tensor has set the mode: only then can a later tensor "mix" with it (the
                                 // config's value is a default for synthetic weights, a checkpoint's own shapes decide)
  bool snap_on_ = false;
  int snap_layer_ = 0;
  void* snap_[9] = {};
  size_t snap_bytes_[9] = {};
  size_t snap_cap_[9] = {};
  bool snap(int idx, const void* src, size_t bytes, int64_t stream);
  // ready-made operands of the fused-norm launches of a 5..32-row step (PreOps): x̃ for gate/up (written by o_proj) and for the
  // next layer's q/k/v (written by down_proj), each with its table of partial sums of squares; *_ok_: written by this step's
  // producer launch and not overtaken by another writer of h
  void *pre_o_ = nullptr, *pre_d_ = nullptr, *pre_e_ = nullptr;  // producers: o_proj, down_proj, the embedding launch (layer 0)
  float *sq_o_ = nullptr, *sq_d_ = nullptr, *sq_e_ = nullptr;
  bool pre_o_ok_ = false, pre_d_ok_ = false;
  void* afrag_ = nullptr;  // the decode attention's output of a 5..32-sequence step in fragment order (o_proj's x on kernel W)
  // Long prefills (csrc/gemm_dense.cuh) run every int4 GEMM as dequant pass + dense GEMM; by default the pass writes into a scratch tensor
  // in front of each GEMM (int4 stays the only resident format).  Opt-in (VRA_DENSE_PREFILL_RESIDENT=1 at engine creation): the
  // dequantised 16-bit fragments of every layer are made ONCE in init_buffers and kept — 2 bytes per weight next to the int4 tensors' 0.5
  // (Llama-3-8B: 14 GB of the 288) — and the prefill GEMMs skip the pass (3.4 ms per forward of the 8B shape).  kind: 0 q/k/v (one tensor of
  // concatenated columns), 1 o_proj, 2 gate/up (interleaved fragments), 3 down_proj; [layer * 4 + kind], null = not resident.
  std::vector<void*> wd_res_;
  int cur_layer_ = 0;  // the layer whose GEMMs forward() / launch_gemm() is issuing
  bool dense_shape_ok(int layer, int kind) const;
  size_t dense_bytes(int layer, int kind) const;
  void dense_fill(int layer, int kind, void* wd, int64_t stream);
  const void* dense_resident(int layer, int kind, int M) const;  // the resident tensor when this GEMM of M rows takes the dense path
  int dense_kind_of(const QLinear& l) const;                      // 1 / 3 for the current layer's o_proj / down_proj, else -1
};

}  // namespace vra
