// gemm_dense_launch.h — launchers of kernel X (gemm_dense.cuh) shared between the C entry points and the native runtime.
#pragma once
#include "gemm_dense.cuh"
// tile width (256 | 128) | split-K slices << 16 for an M x nv (virtual columns) x K problem
int vra_gemm_dense_tile(int M, int nv, int K);
// fills MT / NT / the split-K exchange; bn_sk from vra_gemm_dense_tile
void vra_launch_gemm_dense(GemmXArgs a, bool dual, int dtype, int bn_sk, int64_t stream);
// int4 tile layout -> 16-bit fragments w = round_dt((q - z) * s); the tensor's n-block nb lands at virtual n-frag vfrag0 + nb * vstride
void vra_launch_dequant_frag(const void* tiled, const void* scales, const void* qzeros, void* wd, int K, int N, int group_size, bool awq,
                             int layout, int dtype, int vfrag0, int vstride, int64_t stream);
// the same for up to three tensors of one GEMM (q | k | v, gate | up: same K, group size, layout, format) in ONE launch
void vra_launch_dequant_frag_batch(int n, const void* const* tiled, const void* const* scales, const void* const* qzeros, const int* N, const int* vfrag0,
                                   const int* vstride, void* wd, int K, int group_size, bool awq, int layout, int dtype, int64_t stream);
// the dispatch rule: rows from which a prefill GEMM of M rows x K x nv virtual columns runs as dequant pass + kernel X
int vra_dense_prefill_min_rows();
bool vra_dense_prefill_fits(int M, int K, int nv, int group_size);
// process-wide (per device) region for the dequantised weights of one GEMM; null when unavailable (allocation failed, stream capturing)
void* vra_dense_scratch(size_t bytes, int64_t stream);
