// gemm_launch.h — launchers shared between the C entry points and the native runtime.
#pragma once
#include "gemm_skinny.cuh"
#include "gemv.cuh"
bool vra_gemv_fits(bool int4, int nbw, int M, int K, int group_size);
void vra_launch_gemv(const GemvArgs& a, bool int4, int dtype, int64_t stream);
void vra_launch_skinny(GemmBArgs a, bool int4, bool dual, int dtype, int64_t stream);
