// gemm_launch.h — launchers shared between the C entry points and the native runtime.
#pragma once
#include "gemm_skinny.cuh"
#include "gemv.cuh"
#include "gemm_q4.cuh"
#include "gemm_q4_big.cuh"
#include "gemv_q4s.cuh"
#include "gemv_q4w.cuh"
#include "gemv_dw.cuh"
bool vra_gemv_fits(bool int4, int nbw, int M, int K, int group_size);
void vra_launch_gemv(const GemvArgs& a, bool int4, int dtype, int64_t stream);
void vra_launch_skinny(GemmBArgs a, bool int4, bool dual, int dtype, int64_t stream);
// kernel C (gemm_q4.cuh): int4, 5..32 rows, scale groups >= 128.  `a.ks` / `a.kc` are chosen by the launcher.
bool vra_gemm_q4_fits(int nbw, int M, int K, int group_size);
void vra_launch_gemm_q4(GemmCArgs a, bool awq, int dtype, int64_t stream);
// kernel D (gemm_q4_big.cuh): int4, many rows (prefill), scale groups >= 128.  `cols` = output columns over all segments
// (DUAL: of one tensor).  Returns the m-tiles per wave to launch with (2 or 4), or 0 when the shape belongs to kernel B.
int vra_gemm_q4_big_fits(bool dual, int M, int cols, int K, int group_size, const GemmDArgs* segs);
void vra_launch_gemm_q4_big(const GemmDArgs& a, bool dual, bool awq, int mb, int dtype, int64_t stream);
// kernel E (gemv_q4s.cuh): int4, 1..4 rows, scale groups >= 128 (or channel-wise), n_units = 16-column blocks (pairs count once).
// ns = 1 | 2 (gate/up pair with SiLU*mul).  The launcher fills KT / TPW / the unit distribution / gsh.
bool vra_gemv_s_fits(int ns, int M, int K, int group_size, int n_units, bool norm);
void vra_gemv_s_plan(int n_units, int* grid, int* q, int* r);
void vra_launch_gemv_s(GemvSArgs a, int ns, int group_size, bool awq, int dtype, int64_t stream);
void vra_scales_to_unit_major(const void* scales, void* out, int G, int N, int unit0, int64_t stream);
void vra_zeros_to_unit_major(const uint32_t* zeros, uint32_t* out, int G, int N, int unit0, int64_t stream);
// kernel W (gemv_q4w.cuh): int4, 5..32 rows, K <= 4096, same argument block and unit distribution as kernel E
// (norm_or_segments: the launch carries a fused RMSNorm or more than one output segment — not available in the K-sliced form, K > 4096)
bool vra_gemv_w_fits(int ns, int M, int K, int group_size, int n_units, bool has_res, bool has_bias = false, bool norm_or_segments = false);
void vra_launch_gemv_w(GemvSArgs a, int ns, int group_size, bool awq, int dtype, int64_t stream);
// kernel W, dense (gemv_dw.cuh): 16-bit [N, K] weights, 4..32 rows, K <= 4096 (the lm_head of decode batches); a.norm_w != null
// fuses the RMSNorm.  The launcher fills KT and the unit distribution.
bool vra_gemv_dw_fits(int M, int K, int N);
void vra_launch_gemv_dw(GemvDWArgs a, int dtype, int64_t stream);
// the row-sum table of kernel D (scratch; null when unavailable) and a norm launch that fills it: a GEMM launched with
// GemmDArgs::xsum set skips its own row-sum pass
float* vra_gemm_q4_big_xsum_table(int M, int K);
void vra_rms_norm_xsum(const void* x, const void* weight, void* out, float* xsum, int tokens, int hidden, float eps, int dtype, int64_t stream);

// ---- internal helpers of ops.hip / wna16_gemm.hip used by the native runtime
// vra_embedding + one increment of *bump (null: none; the forward's epoch word of the experiments build) in the same launch
// (+ rows 0..31 in kernel W's fragment order into `frag`, GemvSArgs::x_frag, when frag != null; + with pre_norm_w and <= 32 rows: the
// ready-made operands of layer 0's attention norm, x̃ = round(row * g) into pre_frag and the rows' sums of squares into pre_sq[0..31])
void vra_embedding_bump(const uint32_t* ids, const void* table, void* out, int32_t tokens, int32_t hidden, int32_t vocab, int32_t dtype,
                        uint32_t* bump, void* frag, const void* pre_norm_w, void* pre_frag, float* pre_sq, int64_t stream);
// dense [n, k] 16-bit row-major -> the tile-major copy the dense GEMV kernels stream one contiguous KiB per wave load from
void vra_dense_tile_weights(const void* w_rowmajor, void* out_tiled, int32_t n, int32_t k, int64_t stream);
