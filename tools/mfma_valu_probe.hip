// How do the int4 -> bf16 conversion (VALU) and the MFMAs of a dequant-GEMM inner loop share a gfx950 SIMD?
// One "tile-step" = 4 packed words -> 4 x (8 bf16) B fragments (7 VALU per word: v_and_or / v_lshrrev, the magic-number form of
// wna16.cuh) feeding NM MFMAs each (NM = m-tiles sharing the fragment).  No memory traffic: the words are perturbed per iteration.
// Variants: order of VALU and MFMA inside the step, waves per SIMD (grid is one workgroup per CU, 4..16 waves).
//   V0: all 28 VALU, then all MFMAs (asm volatile, `s_nop 1` in front of each: the form the product kernels use)
//   V1: per word: 7 VALU, then its NM MFMAs (what hipcc emits for kernel W today)
//   V2: software pipelined: the VALU of word j+1 written BETWEEN the MFMAs of word j (one MFMA, ~2 VALU, one MFMA, ...)
//   V3: MFMAs only (no conversion: the B fragment is the raw word)      V4: VALU only
// Output: cycles per tile-step per WAVE and the implied SIMD occupancy split.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mfma(f32x4& acc, s16x8 a, s16x8 b) { asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma_nonop(f32x4& acc, s16x8 a, s16x8 b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); }
__device__ __forceinline__ u32x4 deq(unsigned w) {  // 7 VALU: (w >> 4i) & 0x000f000f | 0x43004300
  u32x4 r;
  r[0] = (w & 0x000f000fu) | 0x43004300u;
  r[1] = ((w >> 4) & 0x000f000fu) | 0x43004300u;
  r[2] = ((w >> 8) & 0x000f000fu) | 0x43004300u;
  r[3] = ((w >> 12) & 0x000f000fu) | 0x43004300u;
  return r;
}
template <int V, int NM>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  u32x4 w = {(unsigned)lane * 2654435761u, (unsigned)lane * 40503u + 7u, (unsigned)threadIdx.x, 12345u + blockIdx.x};
  u32x4 xa[NM];
  for (int m = 0; m < NM; m++) xa[m] = u32x4{0x3f803f80u + m, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  f32x4 acc[NM];
  for (int m = 0; m < NM; m++) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    asm volatile("" : "+v"(w));
    if (V == 0) {
      u32x4 b[4];
      for (int j = 0; j < 4; j++) b[j] = deq(w[j]);
      asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      for (int j = 0; j < 4; j++)
        for (int m = 0; m < NM; m++) mfma(acc[m], __builtin_bit_cast(s16x8, xa[m]), __builtin_bit_cast(s16x8, b[j]));
    } else if (V == 1) {
      for (int j = 0; j < 4; j++) {
        u32x4 b = deq(w[j]);
        asm volatile("" : "+v"(b));
        for (int m = 0; m < NM; m++) mfma(acc[m], __builtin_bit_cast(s16x8, xa[m]), __builtin_bit_cast(s16x8, b));
      }
    } else if (V == 2) {
      u32x4 b = deq(w[0]);
      asm volatile("" : "+v"(b));
      for (int j = 0; j < 4; j++) {
        u32x4 nb;
        const unsigned wn = w[(j + 1) & 3];
        // the conversion of the next word, written between this word's MFMAs
        mfma(acc[0], __builtin_bit_cast(s16x8, xa[0]), __builtin_bit_cast(s16x8, b));
        nb[0] = (wn & 0x000f000fu) | 0x43004300u;
        nb[1] = ((wn >> 4) & 0x000f000fu) | 0x43004300u;
        asm volatile("" : "+v"(nb[0]), "+v"(nb[1]));
        for (int m = 1; m < NM; m++) {
          mfma(acc[m], __builtin_bit_cast(s16x8, xa[m]), __builtin_bit_cast(s16x8, b));
          if (m == 1) {
            nb[2] = ((wn >> 8) & 0x000f000fu) | 0x43004300u;
            nb[3] = ((wn >> 12) & 0x000f000fu) | 0x43004300u;
            asm volatile("" : "+v"(nb[2]), "+v"(nb[3]));
          }
        }
        if (NM == 1) {
          nb[2] = ((wn >> 8) & 0x000f000fu) | 0x43004300u;
          nb[3] = ((wn >> 12) & 0x000f000fu) | 0x43004300u;
          asm volatile("" : "+v"(nb[2]), "+v"(nb[3]));
        }
        b = nb;
      }
    } else if (V == 3) {
      for (int j = 0; j < 4; j++)
        for (int m = 0; m < NM; m++) mfma_nonop(acc[m], __builtin_bit_cast(s16x8, xa[m]), __builtin_bit_cast(s16x8, w));
    } else {
      u32x4 b[4];
      for (int j = 0; j < 4; j++) b[j] = deq(w[j]);
      asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]));
      w[0] ^= b[0][0] ^ b[1][1] ^ b[2][2] ^ b[3][3];
    }
    w[1] += 0x9e3779b9u;
  }
  asm volatile("s_nop 7\n\ts_nop 7");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int m = 0; m < NM; m++) s += acc[m][0] + acc[m][1] + acc[m][2] + acc[m][3];
  if (s == 123.456f || w[0] == 0x1234567u) out[0] = 1;
  if (lane == 0 && blockIdx.x == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int V, int NM>
void run(const char* name, int waves, unsigned* out, unsigned long long* cyc) {
  const int iters = 2000;
  hipMemset(cyc, 0, 16 * 8);
  k<V, NM><<<256, waves * 64>>>(out, iters, cyc);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipEventRecord(e0);
  k<V, NM><<<256, waves * 64>>>(out, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[16];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  unsigned long long mx = 0;
  for (int i = 0; i < waves; i++) mx = h[i] > mx ? h[i] : mx;
  const double per_step_wave = (double)mx / iters;                // shader cycles per tile-step as the slowest wave saw it
  const double per_step_simd = per_step_wave / (waves / 4.0);     // ... per tile-step of the SIMD (waves/4 waves share it)
  printf("%-44s NM %d waves/SIMD %d : %7.1f cycles per tile-step and wave, %6.1f per tile-step of the SIMD (MFMA pipe alone: %3d, VALU alone: %3d); kernel %.1f us\n", name, NM,
         waves / 4, per_step_wave, per_step_simd, 4 * NM * 16, 28 * 4, ms * 1e3);
}
int main() {
  unsigned* out;
  unsigned long long* cyc;
  hipMalloc(&out, 64);
  hipMalloc(&cyc, 16 * 8);
  for (int waves : {4, 8, 16}) {
    run<3, 2>("V3 MFMAs only", waves, out, cyc);
    run<4, 2>("V4 conversion only", waves, out, cyc);
    run<0, 2>("V0 conversion block, then MFMA block", waves, out, cyc);
    run<1, 2>("V1 per word: conversion, then its MFMAs", waves, out, cyc);
    run<2, 2>("V2 next word's conversion between MFMAs", waves, out, cyc);
    run<3, 1>("V3 MFMAs only", waves, out, cyc);
    run<1, 1>("V1 per word: conversion, then its MFMAs", waves, out, cyc);
    run<2, 1>("V2 next word's conversion between MFMAs", waves, out, cyc);
    run<3, 4>("V3 MFMAs only", waves, out, cyc);
    run<1, 4>("V1 per word: conversion, then its MFMAs", waves, out, cyc);
    run<2, 4>("V2 next word's conversion between MFMAs", waves, out, cyc);
  }
  return 0;
}
