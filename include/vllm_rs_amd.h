/*
 * vllm_rs_amd.h — C ABI of the MI355X (gfx950) drop-in for the quantized-forward hot path of
 * guoqingbao/vllm.rs.  Every symbol is `extern "C"`, takes plain pointers and sizes, and is
 * exported by libvllm_rs_amd.so (built from vllm_rs_amd/csrc + vllm_rs_amd/host).
 *
 * Conventions (identical to the reference's FFI):
 *   - all data pointers are DEVICE pointers unless the parameter name starts with `h_`;
 *   - the caller owns every buffer, including outputs; kernels never allocate or retain pointers;
 *   - `stream` is a hipStream_t passed as int64_t (the reference passes `*dev.cu_stream() as i64`,
 *     src/utils/gptq.rs:130); 0 = the null stream;
 *   - functions return void and are asynchronous w.r.t. the host; they are graph-capturable (no
 *     sync, no allocation, no stream creation).  Argument errors are recorded in a thread-safe side
 *     channel readable with vra_last_error() (the reference has no error channel at all,
 *     src/utils/gptq.rs:76-79,197,238 validate in Rust before the call).
 *
 * Section A are the seven symbols vllm.rs imports from `attention_rs::kernels::ffi`
 * (src/utils/gptq.rs:3-6) with the exact names and argument order it uses.
 * Section B are C entry points for the ops vllm.rs reaches through attention_rs Rust wrappers and
 * candle ops; INTEGRATION.md shows the Rust `extern "C"` block for them.
 * Section C is the native host runtime (C++ restatement of src/core + src/models for this path).
 */
#ifndef VLLM_RS_AMD_H
#define VLLM_RS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types for `dtype` parameters */
#define VRA_BF16 0
#define VRA_F16 1
#define VRA_F32 2
/* KV-cache storage format of the `kv_dtype` parameters: the activation dtype (16-bit cache) or one OCP FP8 E4M3 byte
 * per element with scale 1.0 — the reference's `fp8_kvcache` option (kvcache_allocator.rs:188-193,776) */
#define VRA_FP8_E4M3 3

/* scale-tensor layouts accepted by vra_wna16_gemm */
#define VRA_SCALES_ROWMAJOR 0 /* [K/g, N] as stored in the checkpoint                           */
#define VRA_SCALES_MARLIN 1   /* after wna16.rs:193-218 marlin_permute_scales (what Rust passes) */

/* ------------------------------------------------------------------------------------------ */
/* A. Symbols imported by src/utils/gptq.rs:3-6                                               */
/* ------------------------------------------------------------------------------------------ */

/* replaces attention_rs::kernels::ffi::marlin_4bit_bf16 — call site src/utils/gptq.rs:165-178.
 * out[m,n] = in[m,k] · dequant(qweight)[k,n]; GPTQ symmetric int4 (zero point 8, `qzeros` and
 * `g_idx` ignored: wna16.rs:154-160 only routes sym/desc_act=false checkpoints here).
 * `qweight` is the tensor produced by gptq_repack (host shape [k/16, n*2] u32), `scales` is
 * [k/g, n] in VRA_SCALES_MARLIN order, `workspace` ([n] u32 zeros, wna16.rs:238-242) is left
 * untouched. group_size ∈ {32·j, -1}. */
void marlin_4bit_bf16(const void* in, const int32_t* qweight, const void* scales, const void* qzeros,
                      const void* g_idx, void* out, int32_t m, int32_t k, int32_t n,
                      const void* workspace, int32_t group_size, int64_t stream);
/* replaces ffi::marlin_4bit_f16 — src/utils/gptq.rs:133-146 */
void marlin_4bit_f16(const void* in, const int32_t* qweight, const void* scales, const void* qzeros,
                     const void* g_idx, void* out, int32_t m, int32_t k, int32_t n,
                     const void* workspace, int32_t group_size, int64_t stream);
/* replaces ffi::marlin_awq_4bit_bf16 — src/utils/gptq.rs:151-164. AWQ asymmetric: `qzeros` is the
 * RAW checkpoint tensor [k/g, n/8] u32 (AWQ nibble order, no +1 offset), qweight from awq_repack. */
void marlin_awq_4bit_bf16(const void* in, const int32_t* qweight, const void* scales,
                          const void* qzeros, const void* g_idx, void* out, int32_t m, int32_t k,
                          int32_t n, const void* workspace, int32_t group_size, int64_t stream);
/* replaces ffi::marlin_awq_4bit_f16 — src/utils/gptq.rs:118-131 */
void marlin_awq_4bit_f16(const void* in, const int32_t* qweight, const void* scales,
                         const void* qzeros, const void* g_idx, void* out, int32_t m, int32_t k,
                         int32_t n, const void* workspace, int32_t group_size, int64_t stream);
/* replaces ffi::gemm_half_q_half_alt — src/utils/gptq.rs:182-195 (NOTE n before k). Plain GPTQ:
 * qweight [k/8, n] u32 (checkpoint layout), qzeros [k/g, n/8] u32 (stored z-1), scales [k/g, n] f16
 * row-major, g_idx [k] i32, f16 activations only.  bits = 4, or 8 (wna16.rs:154-176 sends every
 * non-Marlin checkpoint here): qweight [k/4, n], qzeros [k/g, n/4], four values per word, zero = stored + 1 as well.
 * The reference's signature carries no group size.  With g_idx the group of row k is g_idx[k].  g_idx == NULL: the group size is read
 * off the extent of the `scales` allocation, so `scales` must then be the BASE of its own [k/g, n] device allocation — a view into a
 * larger buffer (candle's layout.start_offset(), gptq.rs:67) is rejected through vra_last_error() instead of being mis-sized. */
void gemm_half_q_half_alt(const void* in, const uint32_t* qweight, const uint32_t* qzeros,
                          const void* scales, const int32_t* g_idx, void* out, int32_t m, int32_t n,
                          int32_t k, int32_t bits, int64_t stream);
/* replaces ffi::gptq_repack — src/utils/gptq.rs:325-331. in: GPTQ qweight [rows=k/8, cols=n] u32;
 * out: same element count in the CDNA4 tile layout, host shape [k/16, n*2]. */
void gptq_repack(const void* in, void* out, int32_t rows, int32_t cols, int64_t stream);
/* replaces ffi::awq_repack — src/utils/gptq.rs:316-323. in: AWQ qweight [rows=k, cols=n/8] u32. */
void awq_repack(const void* in, void* out, int32_t rows, int32_t cols, int32_t bits, int64_t stream);

/* ------------------------------------------------------------------------------------------ */
/* B. Ops reached through attention_rs wrappers / candle ops                                   */
/* ------------------------------------------------------------------------------------------ */

/* side channel: last argument error recorded by any entry point on this thread ("" if none) */
const char* vra_last_error(void);
void vra_clear_error(void);
/* Device-side error word: 1 if a split-K exchange of the GEMM kernels gave up waiting for a slice (a lost workgroup;
 * the wait is bounded so that the device never hangs, the results of that launch are invalid).  Reads and clears the
 * word; synchronises the device.  The native engine checks it on EVERY step (the word rides along with the token download). */
int32_t vra_take_device_error(void);
/* library/ABI version and the gfx target it was built for */
const char* vra_version(void);

/* General WNA16 GEMM behind Section A (is_awq: qzeros RAW awq tensor or NULL→8; GPTQ path ignores
 * qzeros).  `bias` [n] (nullable, same dtype as activations) is fused (Appendix A9: reference adds
 * it as a separate op, wna16.rs:296-300).  `residual` [m,n] nullable: out = round(round(acc+bias)
 * + residual) — the reference's separate `+` (llama.rs:126,130). */
void vra_wna16_gemm(const void* in, const void* qweight_tiled, const void* scales, const void* qzeros,
                    const void* bias, const void* residual, void* out, int32_t m, int32_t k,
                    int32_t n, int32_t group_size, int32_t is_awq, int32_t scales_layout,
                    int32_t dtype, int64_t stream);
/* gate/up pair with fused SiLU·mul (mlp.rs:451-469): out[m,n] = silu(x·Wg)·(x·Wu), rounding after
 * each reference op (GEMM out, silu, mul). Both weights tiled, n = intermediate size. */
void vra_wna16_gate_up_silu(const void* in, const void* qw_gate, const void* sc_gate,
                            const void* qz_gate, const void* qw_up, const void* sc_up,
                            const void* qz_up, void* out, int32_t m, int32_t k, int32_t n,
                            int32_t group_size, int32_t is_awq, int32_t scales_layout,
                            int32_t dtype, int64_t stream);
/* NormX::forward + QLinear::forward in one call (others.rs:11-29 in front of wna16.rs:263-306):
 * out = rmsnorm(in, norm_weight, eps)·W (+ bias).  Decode batches of 1..4 rows run fused (the
 * normalised activations never reach HBM); other shapes write rmsnorm(in) to `xn_workspace`
 * [m, k] (may be NULL only for the fused shapes) and run the general GEMM.
 * Rounding: at 5+ rows the rounding points are those of the two separate calls.  At 1..4 rows (since round 5) the fused
 * launch applies the normalisation factor in its EPILOGUE: it stages round(in * norm_weight), forms the f32 dot products and
 * multiplies them by rstd before the single output rounding — the reference rounds round(in * rstd) * norm_weight ahead of the
 * GEMM (others.rs:11-29).  Same error size, another rounding pattern: per call <= 4 ulps of the output row's scale apart
 * (tests/test_gpu_gemv_s.py), full depth in bench.py's parity_full_depth_reference_order. */
void vra_rms_norm_wna16_gemm(const void* in, const void* norm_weight, float eps,
                             const void* qweight_tiled, const void* scales, const void* qzeros,
                             const void* bias, void* out, void* xn_workspace, int32_t m, int32_t k,
                             int32_t n, int32_t group_size, int32_t is_awq, int32_t scales_layout,
                             int32_t dtype, int64_t stream);
/* NormX + MLP gate/up + SiLU·mul (llama.rs:127-129, mlp.rs:451-469) in one call. */
void vra_rms_norm_wna16_gate_up_silu(const void* in, const void* norm_weight, float eps,
                                     const void* qw_gate, const void* sc_gate, const void* qz_gate,
                                     const void* qw_up, const void* sc_up, const void* qz_up,
                                     void* out, void* xn_workspace, int32_t m, int32_t k, int32_t n,
                                     int32_t group_size, int32_t is_awq, int32_t scales_layout,
                                     int32_t dtype, int64_t stream);
/* debug/parity: expand a tiled qweight back to nibble indices idx[k,n] u8 (bit-exact check) */
void vra_wna16_unpack_indices(const void* qweight_tiled, uint8_t* idx, int32_t k, int32_t n,
                              int64_t stream);
/* debug/parity: dequantize tiled weights to dense w[k,n] (dtype) = round((q - z)·s) */
void vra_wna16_dequant(const void* qweight_tiled, const void* scales, const void* qzeros, void* w,
                       int32_t k, int32_t n, int32_t group_size, int32_t is_awq,
                       int32_t scales_layout, int32_t dtype, int64_t stream);

/* Prefill of long prompts (round 6): the Marlin GEMM of a prefill chunk (src/utils/gptq.rs:116-178) as TWO launches — a streaming
 * pass that dequantises a tiled int4 tensor ONCE into 16-bit MFMA fragments, w = round_dt((q - z) * s) (the weight Marlin feeds
 * its MMAs), and a 256-row dense GEMM over them.  `wd` holds 1 KiB fragments (16 columns x 32 k in MFMA lane order): fragment
 * (n-frag f, k-chunk c) at byte ((f * k/32) + c) * 1024; the tensor's n-block b lands at n-frag vfrag0 + b * vstride (q/k/v
 * concatenated: stride 1 from the segment's first fragment; gate / up interleaved: vfrag0 0 / 1, stride 2). */
void vra_wna16_dequant_frag(const void* qweight_tiled, const void* scales, const void* qzeros, void* wd, int32_t k, int32_t n,
                            int32_t group_size, int32_t is_awq, int32_t scales_layout, int32_t dtype, int32_t vfrag0,
                            int32_t vstride, int64_t stream);
/* out[m, n] = x[m, k] . wd (+ bias, + residual; roundings as vra_wna16_gemm).  nv = columns of wd; dual != 0: wd interleaves gate and up
 * fragments, out[m, nv/2] = silu(x.Wg) * (x.Wu) (mlp.rs:451-469; no bias).  tile_n: 0 (chosen from the shape), 128 or 256. */
void vra_dense_frag_gemm(const void* x, const void* wd, const void* bias, const void* residual, void* out, int32_t m, int32_t k,
                         int32_t nv, int32_t dual, int32_t dtype, int32_t tile_n, int64_t stream);

/* candle_nn::RmsNorm::forward as used by NormX (src/models/layers/others.rs:11-29):
 * out[t,:] = x[t,:] * rsqrt(mean(x²)+eps) * w, f32 math, one rounding. */
void vra_rms_norm(const void* x, const void* weight, void* out, int32_t tokens, int32_t hidden,
                  float eps, int32_t dtype, int64_t stream);
/* q_norm / k_norm of Attention::forward_ext (src/models/layers/attention.rs:538-601,713-735; Qwen3-style checkpoints), IN PLACE on
 * q [tokens, q_heads, head_dim] and k [tokens, kv_heads, head_dim], before the rotary embedding.  full_dim == 0: one RMSNorm per
 * (token, head) over head_dim channels, weights [head_dim] (attention.rs:724-731); full_dim != 0: one RMSNorm per token over the
 * whole row, weights [q_heads * head_dim] / [kv_heads * head_dim] (`full_dim_qk_norm`, attention.rs:714-722; under tensor
 * parallelism the caller passes its shard of the weight, attention.rs:567-590). */
void vra_qk_rms_norm(void* q, void* k, const void* q_weight, const void* k_weight, int32_t tokens, int32_t q_heads, int32_t kv_heads,
                     int32_t head_dim, int32_t full_dim, float eps, int32_t dtype, int64_t stream);
/* residual add + norm: h = round(x + residual) written to `h_out`, out = rmsnorm(h).
 * Equivalent to llama.rs:126-128 `(attn_output + residual)` followed by the next norm. */
void vra_add_rms_norm(const void* x, const void* residual, const void* weight, void* h_out,
                      void* out, int32_t tokens, int32_t hidden, float eps, int32_t dtype,
                      int64_t stream);
/* elementwise a+b (candle `+`, llama.rs:126,130) */
void vra_add(const void* a, const void* b, void* out, int64_t numel, int32_t dtype, int64_t stream);
/* candle_nn::Embedding::forward (llama.rs:261): out[t,:] = table[ids[t],:]; ids u32 */
void vra_embedding(const uint32_t* ids, const void* table, void* out, int32_t tokens,
                   int32_t hidden, int32_t vocab, int32_t dtype, int64_t stream);
/* Tensor::index_select(dim0) (llama.rs:306-310): out[i,:] = x[idx[i],:] */
void vra_index_select_rows(const void* x, const uint32_t* idx, void* out, int32_t n_idx,
                           int32_t hidden, int32_t dtype, int64_t stream);
/* Activation::Silu + `*` (mlp.rs:468): out = round(round(silu(gate)) * up) */
void vra_silu_mul(const void* gate, const void* up, void* out, int64_t numel, int32_t dtype,
                  int64_t stream);
/* attention_rs::fused_rope::FusedRope::apply_inplace (rotary_emb.rs:103): in-place rotary on
 * q [T,Hq,D] and k [T,Hkv,D]; cos/sin [max_pos, rot_dim/2] (same dtype as q, or f32 when
 * table_dtype=VRA_F32), rows gathered by positions[T] (i64). is_interleaved = `is_rope_i`.
 * rot_dim <= D covers apply_inplace_partial (rotary_emb.rs:88-100). */
void vra_fused_rope(void* q, void* k, const void* cos, const void* sin, const int64_t* positions,
                    int32_t tokens, int32_t q_heads, int32_t kv_heads, int32_t head_dim,
                    int32_t rot_dim, int32_t is_interleaved, int32_t dtype, int32_t table_dtype,
                    int64_t stream);
/* KV-cache geometry: each cache is [num_blocks, kv_heads, block_size, head_dim] (same element
 * count as the reference's flash layout, kvcache_allocator.rs:851-863; blocks stay contiguous so
 * swap_blocks semantics are unchanged).
 * "reshape_and_cache" half of PagedAttention::forward (attention.rs:808-820): scatter k,v [T,Hkv,D]
 * to slot_mapping[T] (i64, slot = block*BS + offset; negative slot = skip, Appendix A6). */
void vra_reshape_and_cache(const void* k, const void* v, void* k_cache, void* v_cache,
                           const int64_t* slot_mapping, int32_t tokens, int32_t kv_heads,
                           int32_t head_dim, int32_t block_size, int32_t dtype, int32_t kv_dtype,
                           int64_t stream);
/* vra_fused_rope (NeoX pairs over the full head, tables in `dtype`) + vra_reshape_and_cache in ONE launch, for prefill:
 * q is rotated in place, the rotated k goes straight to its K-cache row (k itself is left as it was: prefill attention reads
 * the cache), v to its V-cache column.  Bit-identical to the two calls it replaces (attention.rs:745-820 issues them
 * separately). */
void vra_rope_cache_prefill(void* q, const void* k, const void* v, void* k_cache, void* v_cache,
                            const void* cos, const void* sin, const int64_t* positions,
                            const int64_t* slot_mapping, int32_t tokens, int32_t q_heads, int32_t kv_heads,
                            int32_t head_dim, int32_t block_size, int32_t dtype, int32_t kv_dtype,
                            int64_t stream);
/* decode half of PagedAttention::forward: one query token per sequence.
 * q/out [B,Hq,D]; block_tables [B,max_blocks] u32 (right-padded with 0, Appendix A5);
 * context_lens [B] u32 (includes the token just written). `workspace` must hold
 * vra_paged_attention_decode_workspace_bytes(). softcap = 0 disables. */
size_t vra_paged_attention_decode_workspace_bytes(int32_t max_batch, int32_t q_heads,
                                                  int32_t head_dim, int32_t max_context_len);
void vra_paged_attention_decode(void* out, const void* q, const void* k_cache, const void* v_cache,
                                const uint32_t* block_tables, const uint32_t* context_lens,
                                int32_t batch, int32_t q_heads, int32_t kv_heads, int32_t head_dim,
                                int32_t block_size, int32_t max_blocks_per_seq,
                                int32_t max_context_len, float scale, float softcap,
                                void* workspace, int32_t dtype, int32_t kv_dtype, int64_t stream);
/* the same with PagedAttention::new's `sliding_window` (attention.rs:607-616; Mistral-type LlamaForCausalLM checkpoints,
 * utils/mod.rs:1842-1856): > 0: a query at position p attends the keys p-W+1 .. p (the additive form is vra_causal_mask's
 * j <= i && i - j < W); 0 = off = the function above.  KV tiles in front of the window are not read. */
void vra_paged_attention_decode_sw(void* out, const void* q, const void* k_cache, const void* v_cache,
                                   const uint32_t* block_tables, const uint32_t* context_lens,
                                   int32_t batch, int32_t q_heads, int32_t kv_heads, int32_t head_dim,
                                   int32_t block_size, int32_t max_blocks_per_seq,
                                   int32_t max_context_len, float scale, float softcap, int32_t sliding_window,
                                   void* workspace, int32_t dtype, int32_t kv_dtype, int64_t stream);
/* prefill half: causal varlen attention. q [T,Hq,D] with cu_seqlens_q [B+1] u32.
 * If block_tables != NULL keys/values are read from the paged cache (context_lens[b] tokens,
 * query i of sequence b sits at position context_lens[b]-len_q(b)+i) — this covers chunked prefill
 * and prefix-cache hits (runner.rs:1068-1082).  Otherwise they are read from k,v [Tk,Hkv,D] with
 * cu_seqlens_k. */
void vra_paged_attention_prefill(void* out, const void* q, const void* k, const void* v,
                                 const void* k_cache, const void* v_cache,
                                 const uint32_t* block_tables, const uint32_t* context_lens,
                                 const uint32_t* cu_seqlens_q, const uint32_t* cu_seqlens_k,
                                 int32_t batch, int32_t total_q, int32_t max_seqlen_q,
                                 int32_t q_heads, int32_t kv_heads, int32_t head_dim,
                                 int32_t block_size, int32_t max_blocks_per_seq, float scale,
                                 float softcap, int32_t dtype, int32_t kv_dtype, int64_t stream);
/* ... with a sliding window (see vra_paged_attention_decode_sw) */
void vra_paged_attention_prefill_sw(void* out, const void* q, const void* k, const void* v,
                                    const void* k_cache, const void* v_cache,
                                    const uint32_t* block_tables, const uint32_t* context_lens,
                                    const uint32_t* cu_seqlens_q, const uint32_t* cu_seqlens_k,
                                    int32_t batch, int32_t total_q, int32_t max_seqlen_q,
                                    int32_t q_heads, int32_t kv_heads, int32_t head_dim,
                                    int32_t block_size, int32_t max_blocks_per_seq, float scale,
                                    float softcap, int32_t sliding_window, int32_t dtype, int32_t kv_dtype, int64_t stream);
/* Fused decode step of one layer's attention front half, ONE launch for what the reference issues as
 * FusedRope::apply_inplace + reshape_and_cache + PagedAttention::forward (attention.rs:745-820): rotary on q
 * and k (NeoX pairing, tables [n_pos, head_dim/2] in the model dtype), scatter of the rotated k and of v
 * into the cache at slot_mapping (negative slot = padded lane: nothing written), paged decode attention
 * over context_lens tokens (which include the new one).  `out` and the caches are bit-identical to the
 * three separate calls; q and k are READ ONLY here (the rotated copies never reach HBM). */
void vra_rope_cache_attention_decode(void* out, const void* q, const void* k, const void* v, void* k_cache,
                                     void* v_cache, const void* cos, const void* sin,
                                     const int64_t* positions, const int64_t* slot_mapping,
                                     const uint32_t* block_tables, const uint32_t* context_lens,
                                     int32_t batch, int32_t q_heads, int32_t kv_heads,
                                     int32_t head_dim, int32_t block_size,
                                     int32_t max_blocks_per_seq, int32_t max_context_len,
                                     float scale, void* workspace, int32_t dtype, int32_t kv_dtype,
                                     int64_t stream);
/* attention_rs::mask::causal_mask (src/models/layers/mask.rs:24-27): additive [L,L] mask,
 * 0 on/below the diagonal (and within sliding_window if >0), -inf above. */
void vra_causal_mask(void* mask, int32_t len, int32_t sliding_window, int32_t dtype, int64_t stream);
/* attention_rs::cache::swap_blocks (runner.rs:1641-1645): copy whole blocks src[i]→dst[j];
 * `h_pairs` is a HOST array of 2*n_pairs int64 (src,dst). kind: 0 D2D, 1 D2H, 2 H2D. */
void vra_swap_blocks(const void* src, void* dst, const int64_t* h_pairs, int32_t n_pairs,
                     int64_t block_bytes, int32_t kind, int64_t stream);
/* dense Linear::forward for lm_head (linear.rs:75-123): out[m,n] = x[m,k]·W[n,k]ᵀ (+bias),
 * W row-major [n,k]; out_dtype VRA_F32 reproduces `.to_dtype(F32)` of the rounded result
 * (llama.rs:317-319): values are rounded to `dtype` first, then widened. */
void vra_dense_gemm(const void* x, const void* w, const void* bias, void* out, int32_t m, int32_t k,
                    int32_t n, int32_t dtype, int32_t out_dtype, int64_t stream);
/* logits.argmax(-1) (logits_processor.rs:67-70): first maximal index per row. */
void vra_argmax_f32(const float* logits, uint32_t* out, int32_t rows, int32_t cols, int64_t stream);
/* lm_head + greedy sampling as one launch (llama.rs:311-320 followed by logits_processor.rs:67-70): f32 logits as
 * vra_dense_gemm(..., VRA_F32) AND tokens[m] = first maximal index of row m.  Up to 8 rows (kernel A) and at 4..32 rows
 * with k <= 4096 (the dense W kernel) the tokens come out of the GEMV launch itself (per-workgroup candidates, the last
 * workgroup to arrive reduces them); anything else is the two launches.
 * `workspace`: vra_dense_gemm_argmax_workspace_bytes() bytes, zeroed ONCE by the caller, one per concurrent stream. */
int64_t vra_dense_gemm_argmax_workspace_bytes(void);
void vra_dense_gemm_argmax(const void* x, const void* w, const void* bias, float* logits, uint32_t* tokens, void* workspace,
                           int32_t m, int32_t k, int32_t n, int32_t dtype, int64_t stream);
/* Stochastic sampling on the device — LogitsProcessor::sample_with_strategy for Sampling::{All, TopK, TopP,
 * TopKThenTopP} (logits_processor.rs:199-271; the reference's own device path is `sampler.sample_cuda(logits,
 * k, p, t, seed)` with k <= 256, and k = 256 standing in for "top-p only"): probabilities = softmax(logits /
 * temperature) over the whole row; the top_k most probable tokens in descending order (ties: lower id first);
 * top-p keeps a candidate while the mass accumulated before it is < top_p; one draw u in [0, kept mass) picks
 * the first candidate whose running sum exceeds u.  top_k <= 0: unset; top_p outside (0,1): unset; both unset:
 * the whole distribution.  The uniform of row r is the counter hash of (seed, r).  `dbg_idx`/`dbg_prob`
 * [rows, 256] (nullable) receive the ordered candidates and their probabilities (0 = dropped by top-p). */
void vra_sample(const float* logits, uint32_t* out, int32_t rows, int32_t vocab, int32_t top_k,
                float top_p, float temperature, uint64_t seed, uint32_t* dbg_idx, float* dbg_prob,
                int64_t stream);
/* LogitsProcessor::apply_batch_repeat_penalty (logits_processor.rs:288-345), in place on f32 logits [rows, vocab]:
 * logit -= count*frequency_penalty + (count > 0)*presence_penalty with counts over context[row, :context_lens[row]]
 * (u32 token ids, row stride max_context <= 1024); rows with <= 1 context token or penalties in {0, 1} are untouched. */
void vra_apply_penalties(float* logits, const uint32_t* context, const int32_t* context_lens, int32_t rows,
                         int32_t max_context, int32_t vocab, const float* frequency_penalties,
                         const float* presence_penalties, int64_t stream);
/* Tensor::to_dtype between bf16/f16/f32 */
void vra_cast(const void* in, void* out, int64_t numel, int32_t in_dtype, int32_t out_dtype,
              int64_t stream);
/* deterministic synthetic tensors (BASELINE §8d): u32 hash fill / uniform / normal fills */
void vra_fill_hash_u32(uint32_t* out, int64_t numel, uint64_t seed, int64_t stream);
void vra_fill_uniform(void* out, int64_t numel, uint64_t seed, float lo, float hi, int32_t dtype,
                      int64_t stream);
void vra_fill_normal(void* out, int64_t numel, uint64_t seed, float mean, float std, int32_t dtype,
                     int64_t stream);
void vra_fill_const_u32(uint32_t* out, int64_t numel, uint32_t value, int64_t stream);
/* packed AWQ zero points of the synthetic checkpoints: nibbles drawn from {6:1, 7:3, 8:8, 9:3, 10:1}/16 (concentrated on 8,
 * as in real AWQ checkpoints), from the same counter hash as vra_fill_hash_u32 */
void vra_fill_awq_zeros(uint32_t* out, int64_t numel, uint64_t seed, int64_t stream);

/* AllReduce CustomOp1 (src/models/layers/distributed.rs:325-396): sum over TP ranks, bf16/f16.
 * The communicator is created from the 128-byte unique id the engine ships in MessageType::Init
 * (src/runner/mod.rs:25-27; Comm::from_rank at src/runner/runner.rs:80-89).  world_size 1 creates a
 * real one-rank RCCL communicator.  Communicators are caller-owned (vra_engine_set_comm does not
 * take ownership). */
int32_t vra_comm_unique_id(uint8_t h_id_out[128]);
void* vra_comm_create(const uint8_t h_id[128], int32_t rank, int32_t world_size, int32_t device);
void vra_comm_destroy(void* comm);
int32_t vra_comm_rank(const void* comm);
int32_t vra_comm_world_size(const void* comm);
void vra_all_reduce(void* comm, const void* src, void* dst, int64_t numel, int32_t dtype,
                    int64_t stream);
/* One-shot transport for decode-sized messages (also the only transport between
 * ranks that are processes sharing one GPU): (1) vra_comm_ipc_begin allocates this rank's exchange
 * region and exports its 64-byte hipIpcMemHandle — `comm` is an RCCL communicator from
 * vra_comm_create (hybrid) or NULL (one-shot only; a new communicator is returned); (2) the
 * launcher gathers the world_size handles exactly as it ships the unique id; (3)
 * vra_comm_ipc_connect maps the peers.  Messages of up to `oneshot_max_bytes` (0 = 1 MiB) take the
 * one-shot path when RCCL is also present; without RCCL every message does (in 8 MiB launches).
 * All ranks sum in rank order in f32: results are bit-identical on every rank. */
void* vra_comm_ipc_begin(void* comm, int32_t rank, int32_t world_size, int32_t device,
                         uint8_t h_handle_out[64]);
int32_t vra_comm_ipc_connect(void* comm, const uint8_t* h_all_handles /* [world_size][64] */,
                             int64_t oneshot_max_bytes);
/* TensorParallelRowLinear::forward + the decoder layer's residual add in ONE call
 * (distributed.rs:438-455, llama.rs:126,130): dst = all_reduce_sum(partial); dst = round(dst +
 * bias) (bias [cols] or NULL); dst = dst + residual (NULL, or [rows, cols]; may alias dst).
 * `partial` is clobbered when the RCCL transport runs with an epilogue. bf16/f16. */
void vra_all_reduce_fused(void* comm, void* partial, void* dst, const void* bias,
                          const void* residual, int64_t rows, int32_t cols, int32_t dtype,
                          int64_t stream);
/* 1 if a one-shot exchange gave up waiting for a peer (bounded wait: the device never hangs; the
 * results of that launch are invalid).  Reads and clears; synchronises.  vra_comm_error_word is the
 * device word behind it (NULL without the one-shot transport) for callers that poll it themselves. */
int32_t vra_comm_take_error(void* comm);
/* what the first timed-out wait of the one-shot exchange saw (valid after vra_comm_take_error / the error word reported a
 * timeout): h_out[0] slice index, [1] the peer rank waited for, [2] the epoch expected, [3] the flag value last read. */
int32_t vra_comm_error_detail(void* comm, uint32_t h_out[4]);
uint32_t* vra_comm_error_word(void* comm);

/* device plumbing for hosts that have no HIP binding of their own (tests, bench, the runtime) */
int32_t vra_device_count(void);
int32_t vra_set_device(int32_t device);
void* vra_malloc(size_t bytes);
void vra_free(void* p);
void* vra_malloc_host(size_t bytes);
void vra_free_host(void* p);
int32_t vra_memcpy_h2d(void* dst, const void* h_src, size_t bytes, int64_t stream);
int32_t vra_memcpy_d2h(void* h_dst, const void* src, size_t bytes, int64_t stream);
int32_t vra_memcpy_d2d(void* dst, const void* src, size_t bytes, int64_t stream);
int32_t vra_memset(void* dst, int32_t value, size_t bytes, int64_t stream);
int32_t vra_stream_sync(int64_t stream);
int32_t vra_device_sync(void);
int64_t vra_stream_create(void);
void vra_stream_destroy(int64_t stream);
int32_t vra_mem_info(size_t* h_free, size_t* h_total);
void* vra_event_create(void);
void vra_event_destroy(void* ev);
int32_t vra_event_record(void* ev, int64_t stream);
float vra_event_elapsed_ms(void* start, void* stop); /* syncs on `stop` */

/* ------------------------------------------------------------------------------------------ */
/* C. Native host runtime (C++): model, KV allocator, block manager, scheduler, runner, engine */
/* ------------------------------------------------------------------------------------------ */

/* Model/engine configuration — the fields of `Config` (src/utils/config.rs:218-255),
 * `QuantConfig` (:735-757) and `EngineConfig` (:285-328) the hot path consumes. */
typedef struct vra_model_config {
  int32_t arch;              /* 0 = LlamaForCausalLM/Mistral (llama.rs), 1 = Qwen2ForCausalLM, 2 = Qwen3ForCausalLM (both qwen3.rs) */
  int32_t hidden_size, intermediate_size, num_layers;
  int32_t num_heads, num_kv_heads, head_dim, vocab_size;
  int32_t max_position_embeddings;
  float rms_norm_eps;
  double rope_theta;
  int32_t rope_scaling_type; /* 0 none/default, 1 linear, 2 llama3, 3 dynamic (NTK), 4 yarn  (rotary_emb.rs:143-415) */
  double rope_factor, rope_low_freq_factor, rope_high_freq_factor;
  int32_t rope_original_max_position;
  int32_t attention_bias;    /* qkv bias (qwen2 default true, attention.rs:411-415) */
  int32_t quant_method;      /* 0 none (dense bf16), 1 gptq, 2 awq */
  int32_t bits, group_size;
  int32_t dtype;             /* VRA_BF16 / VRA_F16 */
  int32_t tie_word_embeddings;
  /* rope_scaling of the dynamic / yarn types (appended: older callers that zero-initialise the struct keep their meaning).
   * dynamic: rope_dynamic_alpha != 0 => `alpha` form (theta' = (theta*alpha)^(d/(d-2)), alpha in rope_factor), else the `factor`
   * form (rotary_emb.rs:281-333).  yarn (rotary_emb.rs:335-415,435-541): 0 in a field = the reference's default
   * (beta_fast 32, beta_slow 1, attn_factor 1, extrapolation_factor 1). */
  int32_t rope_dynamic_alpha;
  double rope_yarn_beta_fast, rope_yarn_beta_slow, rope_yarn_attn_factor, rope_yarn_extrapolation_factor;
  /* appended in round 4: the reference keeps original_max_position_embeddings (or max_position_embeddings / factor)
   * as f64 for the llama3 wavelengths and the dynamic table (rotary_emb.rs:150-164) — > 0: used instead of the truncated
   * rope_original_max_position; and it honours an EXPLICIT 0 in a yarn field: bit i of rope_yarn_explicit (0 beta_fast, 1 beta_slow,
   * 2 attn_factor, 3 extrapolation_factor) = "the field was given, take it as it is" (0 in the field then means 0, not the default) */
  double rope_original_max_position_f;
  int32_t rope_yarn_explicit;
  /* appended in round 5 (fills the struct's tail padding: the size is unchanged).  q_norm / k_norm of the attention block
   * (attention.rs:538-601): 0 none, 1 per head (weights [head_dim]: Qwen3), 2 over the full q / k row (weights [heads * head_dim]).
   * Loaded checkpoints set it from the shape of `self_attn.q_norm.weight`; synthetic weights follow this field. */
  int32_t qk_norm;
  /* appended in round 6: sliding-window attention (llama.rs:46,284 passes config.sliding_window into attention and mask; Mistral-type
   * LlamaForCausalLM checkpoints, utils/mod.rs:1842-1856).  > 0: a query at position p attends the keys p-W+1 .. p — the decode step
   * runs RoPE + KV write + vra_paged_attention_decode_sw, prefill vra_paged_attention_prefill_sw (KV tiles in front of the window are
   * not read).  0 = full causal attention.  Every cached token stays in its block (the block manager frees nothing early). */
  int32_t sliding_window;
} vra_model_config;

typedef struct vra_engine_config {
  int32_t block_size;             /* 64 (config.rs:466) */
  int32_t max_num_seqs;           /* 0 = auto */
  int32_t max_model_len;          /* 0 = auto */
  int32_t num_gpu_blocks;         /* 0 = derive from free memory × kv_fraction */
  float kv_fraction;              /* 0 = reference default (kvcache_allocator.rs:196-202) */
  int32_t prefill_chunk;          /* 8192 (scheduler.rs:203) */
  int32_t enable_prefix_cache;
  float prefix_cache_fraction;    /* 0.65 (scheduler.rs:55,95) */
  int32_t use_graph;              /* hipGraph decode capture for bs ∈ {1..15,16,32} (graph.rs:370-377) */
  int32_t tp_rank, tp_world_size; /* tensor parallel */
  int32_t device;
  uint64_t seed;                  /* synthetic-weight seed */
  int32_t fp8_kvcache;            /* EngineConfig.fp8_kvcache (config.rs:316): KV cache in FP8 E4M3, 1 byte per element */
  float cpu_mem_fold;             /* CPU swap space as a fraction of the GPU blocks (kvcache_allocator.rs:317,673: the
                                     reference defaults to 0.2); 0 = no swap space, preempted sequences wait or are dropped */
  int32_t swap_cooling_ms;        /* scheduler.rs:49 SWAP_COOLING_PERIOD; 0 = 5000, negative = none */
  int32_t min_tokens_left_for_swap; /* scheduler.rs:50; 0 = 1000, negative = 0 */
} vra_engine_config;

/* Pure-host helpers (no GPU needed; also exported by libvra_host.so for CPU tests) */
/* per_block_bytes (kvcache_allocator.rs:447-468) and the block-count plan (:616-707) */
int64_t vra_kv_per_block_bytes(const vra_model_config* mc, const vra_engine_config* ec);
int64_t vra_kv_plan_num_blocks(const vra_model_config* mc, const vra_engine_config* ec,
                               int64_t free_bytes);
/* rotary tables (rotary_emb.rs:32-73,126-278): f32 cos/sin [n_pos, rot_dim/2] */
void vra_rope_tables_f32(const vra_model_config* mc, int32_t n_pos, float* h_cos, float* h_sin);
/* rows of the table the reference builds for this config (rotary_emb.rs:44-46,296-312,519-526): max_position_embeddings, or
 * for yarn (u32)(max_position_embeddings * factor), for dynamic-by-factor (u32)(original_max * factor) — the engine sizes its
 * tables, max_model_len clamp and position checks with it */
int32_t vra_rope_table_rows(const vra_model_config* mc);
/* marlin_permute_scales (wna16.rs:180-218) on a host array of 16-bit elements [k/g, n] */
void vra_marlin_permute_scales_u16(const uint16_t* h_in, uint16_t* h_out, int32_t rows, int32_t n,
                                   int32_t grouped);

/* Block manager + prefix cache (src/core/block_manager.rs, prefix_cache.rs, sequence.rs) */
void* vra_bm_create(int32_t num_blocks, int32_t block_size, int32_t enable_prefix_cache,
                    float prefix_cache_fraction);
void vra_bm_destroy(void* bm);
int32_t vra_bm_num_free_blocks(const void* bm);
/* creates a sequence, returns its id */
int64_t vra_bm_seq_create(void* bm, const uint32_t* h_tokens, int32_t n_tokens);
void vra_bm_seq_free(void* bm, int64_t seq);
int32_t vra_bm_can_allocate(const void* bm, int64_t seq);
int32_t vra_bm_allocate(void* bm, int64_t seq);          /* returns num_cached_tokens or -1 */
int32_t vra_bm_can_append(const void* bm, int64_t seq);
int32_t vra_bm_may_append(void* bm, int64_t seq);        /* 0 ok, -1 no free block */
void vra_bm_append_token(void* bm, int64_t seq, uint32_t token);
void vra_bm_deallocate(void* bm, int64_t seq);           /* caches full blocks when prefix cache on */
int32_t vra_bm_seq_len(const void* bm, int64_t seq);
int32_t vra_bm_seq_num_cached_tokens(const void* bm, int64_t seq);
int32_t vra_bm_seq_block_table(const void* bm, int64_t seq, uint32_t* h_out, int32_t cap);
int32_t vra_bm_prefix_cached_blocks(const void* bm);
int32_t vra_bm_evict_prefix(void* bm, int32_t n_blocks);

/* PrefixCache on its own (src/core/prefix_cache.rs:72-293): hash-chained full blocks, leaf-LRU eviction.
 * insert_prefix returns the number of blocks inserted and the block ids evicted to stay within
 * max_cached_blocks; match_prefix returns the number of leading full blocks found and their ids. */
void* vra_pc_create(int32_t block_size, int32_t max_cached_blocks);
void vra_pc_destroy(void* pc);
int32_t vra_pc_insert_prefix(void* pc, const uint32_t* h_tokens, int32_t n_tokens, const int32_t* h_blocks,
                             int32_t n_blocks, int32_t* h_evicted, int32_t cap, int32_t* h_n_evicted);
int32_t vra_pc_match_prefix(void* pc, const uint32_t* h_tokens, int32_t n_tokens, int32_t* h_blocks,
                            int32_t cap);
int32_t vra_pc_cached_blocks(const void* pc);
int32_t vra_pc_evict_blocks(void* pc, int32_t n, int32_t* h_evicted, int32_t cap);

/* Engine = scheduler + runner + model (src/core/{engine,scheduler,runner}.rs) */
void* vra_engine_create(const vra_model_config* mc, const vra_engine_config* ec);
void vra_engine_destroy(void* eng);
/* weights: synthetic (BASELINE §8d recipe, seeded) or explicit host tensors by HF name */
int32_t vra_engine_init_synthetic(void* eng);
int32_t vra_engine_load_tensor(void* eng, const char* name, const void* h_data, const int64_t* shape,
                               int32_t ndim, int32_t elem_bytes);
int32_t vra_engine_finalize_weights(void* eng); /* repack + scale layout + KV cache + graphs */
/* The reference's runner sizes its KV cache only AFTER the engine process has answered the first InitAck with
 * MessageType::UsableMemoryLeft(EngineConfig) (src/core/runner.rs:443-455, src/core/engine.rs:355-378): weights first
 * (vra_engine_finalize_model: repack + decode layouts, nothing else is allocated), then the negotiated configuration
 * (vra_engine_update_config: num_gpu_blocks, max_num_seqs, max_model_len, cpu_mem_fold, kv_fraction — allowed
 * until buffers exist), then vra_engine_finalize_weights for activations, KV cache and graphs. */
/* parity instrumentation: the f32 logits [n_seqs, vocab] the last step (hipGraph replay or eager) left on the device */
int32_t vra_engine_copy_logits(void* eng, float* h_out, int32_t n_seqs);
/* parity instrumentation of the tensor-parallel forward: with snapshots on (`on` = 1 + the layer; 0 = off), that layer of every forward keeps copies of its stages
 * (0 q, 1 k, 2 v before RoPE, 3 attention output, 4 o_proj partial of this rank, 5 h after the first all-reduce + residual,
 * 6 SiLU(gate)*up, 7 down_proj partial, 8 h after the second all-reduce); read returns the bytes copied or -1. */
/* parity instrumentation: which fused RMSNorm launches of a step of `rows` rows of decoder layer `layer` apply the normalisation factor in
 * their epilogue (rstd commutes with the product: the 1..4-row decode kernel always; at 5..32 rows the launches whose producer — a
 * kernel-W o_proj / down_proj of the same step, the embedding launch for layer 0's q/k/v — left them ready-made operands x~ = round(h * g)); bit 0 = norm + q/k/v, bit 1 = norm +
 * gate/up.  vra_debug_norm_deferred_mask is the same rule from shapes alone (per-rank sizes; q/k/v biases only) and
 * vra_debug_gemv_s_fits the 1..4-row predicate behind it — the oracle restates the order the engine runs (oracle/model.py ENGINE_RULE). */
int32_t vra_engine_norm_deferred(void* eng, int32_t rows, int32_t layer);
int32_t vra_debug_norm_deferred_mask(int32_t hidden, int32_t inter_local, int32_t heads_local, int32_t kv_heads_local, int32_t head_dim,
                                     int32_t group_size, int32_t quant, int32_t qkv_bias, int32_t world, int32_t rows, int32_t layer,
                                     int32_t dtype); /* f16 models defer at 1..4 rows only (no ready-made operands: their range is bf16's) */
int32_t vra_debug_gemv_s_fits(int32_t ns, int32_t m, int32_t k, int32_t group_size, int32_t n_units, int32_t norm);
/* parity instrumentation: rows from which the int4 GEMMs of a prefill step run as dequant pass + dense GEMM on Marlin-rounded weights
 * (vra_wna16_dequant_frag / vra_dense_frag_gemm; default 768, VRA_DENSE_PREFILL_MIN_ROWS; 0 = never).  The oracle restates the
 * weight rounding the engine runs (oracle/model.py dense_prefill_rows); tests lower it to reach the path with small models. */
int32_t vra_debug_dense_prefill_min_rows(void);
void vra_debug_set_dense_prefill_min_rows(int32_t rows);
void vra_engine_debug_tp_snapshots(void* eng, int32_t on); /* on = 1 + the layer whose stages are kept; 0 = off */
int64_t vra_engine_debug_read_tp_snapshot(void* eng, int32_t idx, void* h_out, int64_t max_bytes);
int32_t vra_engine_finalize_model(void* eng);
int32_t vra_engine_update_config(void* eng, const vra_engine_config* cfg);
/* ModelRunner::swap_kvcache (runner.rs:1626-1670; MessageType::KVCacheSwap): copy whole blocks between the GPU cache and the
 * engine's pinned swap space (vra_engine_config.cpu_mem_fold > 0), all layers, K and V.  h_pairs = 2*n_pairs int64
 * (source block, destination block): GPU -> CPU ids when swap_in == 0, CPU -> GPU ids when 1.  0 = done (synchronous). */
int32_t vra_engine_swap_blocks(void* eng, const int64_t* h_pairs, int32_t n_pairs, int32_t swap_in);
/* CPU swap (block_manager.rs:870-1010): out4 = {cpu blocks, free cpu blocks, blocks swapped out so far, blocks swapped in} */
void vra_engine_swap_stats(const void* engine, int64_t* out4);
int32_t vra_engine_num_gpu_blocks(const void* eng);
/* KVCacheAllocator plan of this rank before the cache exists (kvcache_allocator.rs:564-707): the
 * block count finalize would derive from free memory x kv_fraction.  Under tensor parallelism the
 * launcher calls it on every rank, takes the minimum (the reference's engine process does this via
 * MessageType::UsableMemoryLeft, runner/mod.rs:277) and sets it with vra_engine_set_num_gpu_blocks
 * before vra_engine_finalize_weights — finalize refuses tp_world_size > 1 without an explicit count. */
int64_t vra_engine_plan_kv_blocks(void* eng);
int32_t vra_engine_set_num_gpu_blocks(void* eng, int32_t num_gpu_blocks);
/* request API: token ids in, token ids out (tokenizer-free) */
int64_t vra_engine_add_request(void* eng, const uint32_t* h_prompt, int32_t n_prompt,
                               int32_t max_tokens, int32_t ignore_eos, const uint32_t* h_eos,
                               int32_t n_eos);
/* SamplingParams as ModelRunner::sample reads them (config.rs:476-520, runner.rs:1405-1497).  Unset ("None") values:
 * temperature < 0, top_k <= 0, top_p < 0, has_*_penalty = 0.  temperature == 0 is greedy.  With nothing set the
 * reference's default applies: top-k 32, top-p 0.95, temperature 0.7 (Appendix A4).  As in the reference the strategy
 * and penalties of a batch are those of its FIRST sequence at prefill, cached for the decode steps (Appendix A3);
 * penalties act on decode steps once a sequence has sampled more than 128 tokens, over the last 128 (runner.rs:1519-1534). */
typedef struct vra_sampling_params {
  float temperature;
  int32_t top_k;
  float top_p;
  int32_t has_frequency_penalty, has_presence_penalty;
  float frequency_penalty, presence_penalty;
} vra_sampling_params;
/* vra_engine_add_request with sampling params (NULL = greedy, as vra_engine_add_request) */
int64_t vra_engine_add_request_ex(void* eng, const uint32_t* h_prompt, int32_t n_prompt,
                                  int32_t max_tokens, int32_t ignore_eos, const uint32_t* h_eos,
                                  int32_t n_eos, const vra_sampling_params* sampling);
/* one engine step (engine.rs:1693-1757): schedule → forward → postprocess.
 * returns number of sequences run (0 = idle), *h_is_prefill set. */
int32_t vra_engine_step(void* eng, int32_t* h_is_prefill);
/* Host-only engine (vra_engine_config.device = -1, explicit num_gpu_blocks; no GPU touched): the two
 * halves of vra_engine_step around the forward pass, for CPU parity tests of the scheduler
 * (scheduler.rs:200-380,500-629), the block manager and the metadata arithmetic of
 * ModelRunner::prepare_prefill / prepare_decode (runner.rs:978-1388).  dry_schedule returns the number
 * of sequences in the step and HOST pointers to the staged InputMetadata (valid until the next call);
 * dry_commit feeds the tokens a model would have sampled (one per sequence, in step order). */
typedef struct vra_step_meta {
  int32_t n_tokens, n_seqs, max_blocks, max_seqlen_q, max_context_len;
  const uint32_t* input_ids;    /* [n_tokens] */
  const int64_t* positions;     /* [n_tokens] */
  const int64_t* slot_mapping;  /* [n_tokens] */
  const uint32_t* block_tables; /* [n_seqs, max_blocks] right-padded with 0 */
  const uint32_t* context_lens; /* [n_seqs] */
  const uint32_t* cu_seqlens_q; /* [n_seqs + 1] (prefill) */
  int64_t request_ids[64];      /* request id of each sequence of the step (first 64) */
} vra_step_meta;
int32_t vra_engine_dry_schedule(void* eng, int32_t* h_is_prefill, vra_step_meta* out);
int32_t vra_engine_dry_commit(void* eng, const uint32_t* h_tokens, int32_t n);
int32_t vra_engine_has_unfinished(const void* eng);
int32_t vra_engine_request_finished(const void* eng, int64_t req);
int32_t vra_engine_request_output(const void* eng, int64_t req, uint32_t* h_out, int32_t cap);
/* timestamps in ms (engine.rs:1004-1012): created, first-token (decode_start), finished */
int32_t vra_engine_request_times(const void* eng, int64_t req, double h_times[3]);
void vra_engine_release_request(void* eng, int64_t req);
/* low-level: one model forward with caller-built metadata, for parity tests of a1/a14.
 * Mirrors `forward(input_ids, positions, kv_caches, input_metadata)` (llama.rs:323-339): all
 * arrays are HOST arrays; logits_out is a HOST [n_seqs, vocab] f32 buffer. */
int32_t vra_engine_forward_raw(void* eng, const uint32_t* h_ids, const int64_t* h_positions,
                               const int64_t* h_slot_mapping, int32_t n_tokens, int32_t is_prefill,
                               const uint32_t* h_block_tables, int32_t max_blocks,
                               const uint32_t* h_context_lens, const uint32_t* h_cu_seqlens_q,
                               int32_t n_seqs, float* h_logits_out);
/* The same forward, answered with the SAMPLED TOKEN IDS instead of the logits — what the reference's runner process sends back
 * (RunResponse, runner.rs:246-292).  Decode steps replay the hipGraph of their batch / context bucket when the engine was built with
 * use_graph (the reference replays its captured decode graphs the same way, graph.rs:370-377); greedy tokens are the first maximal
 * index (logits_processor.rs:67-70), `stochastic` != 0 runs LogitsProcessor::sample_with_strategy on the device logits
 * (logits_processor.rs:199-271: top_k <= 256, 0 = off; top_p < 0 = off) with `seed`.  Nothing of vocabulary size crosses PCIe. */
int32_t vra_engine_forward_tokens(void* eng, const uint32_t* h_ids, const int64_t* h_positions,
                                  const int64_t* h_slot_mapping, int32_t n_tokens, int32_t is_prefill,
                                  const uint32_t* h_block_tables, int32_t max_blocks,
                                  const uint32_t* h_context_lens, const uint32_t* h_cu_seqlens_q,
                                  int32_t n_seqs, int32_t stochastic, int32_t top_k, float top_p,
                                  float temperature, uint64_t seed, uint32_t* h_tokens_out);
/* decode-step microbenchmark hook used by bench.py: runs `steps` decode steps of the current
 * running batch back to back (graph replay when enabled) and returns elapsed ms measured with HIP
 * events on the engine stream; tokens are sampled and appended exactly as in vra_engine_step. */
double vra_engine_timed_decode(void* eng, int32_t steps);
/* measurement aid: the decode graph of the most recent step replayed `steps` times back to back with that step's metadata left in
 * place (no upload / download / host work in between; engine state untouched); returns ms per replay, -1 without a graph. */
double vra_engine_bench_replay(void* eng, int32_t steps);
/* tensor parallel: attach a communicator (vra_comm_create and/or vra_comm_ipc_begin/connect) before finalize;
 * the communicator stays caller-owned and must outlive the engine */
int32_t vra_engine_set_comm(void* eng, void* comm);
/* roofline leg of bench.py: average duration (ms) of ONE launch of a decode-shaped GEMM kernel
 * family at m rows, rotating over all layers' weights so nothing is cache resident, measured with
 * HIP events on the engine stream. which: 0 = norm+q/k/v, 1 = o_proj(+residual),
 * 2 = norm+gate/up+SiLU·mul, 3 = down(+residual), 4 = final norm + lm_head (+ greedy tokens; m <= max_num_seqs, one
 * tensor: nothing to rotate over — 1 GB does not stay cache resident either).  vra_engine_gemm_bytes = SURVEY §8(d)
 * algorithmic bytes of that launch. */
double vra_engine_bench_gemm(void* eng, int32_t which, int32_t m, int32_t iters);
int64_t vra_engine_gemm_bytes(const void* eng, int32_t which, int32_t m);
int64_t vra_engine_weight_bytes(const void* eng);
int64_t vra_engine_stream(const void* eng);
const char* vra_engine_last_error(const void* eng);

#ifdef __cplusplus
}
#endif
#endif /* VLLM_RS_AMD_H */
