// wna16.cuh — int4 (GPTQ/AWQ) dequantisation primitives for the CDNA4 tile layout.
#pragma once
#include "common.cuh"

// AutoAWQ nibble order: column n0+j of a packed word sits at nibble AWQ_REV[j]
__device__ __constant__ const int kAwqRev[8] = {0, 4, 1, 5, 2, 6, 3, 7};
__device__ __forceinline__ int awq_rev(int j) { return ((j & 1) << 2) | (j >> 1); }  // {0,4,1,5,2,6,3,7}

// One tiled word (8 codes of one column, 8 consecutive k) -> MFMA A fragment (8 x 16-bit),
// w_e = round_dt(fma(q_e, s, c)), c = -z*s  ==  round_dt((q_e - z) * s) exactly (the product is
// exactly representable in f32, so the fma performs no rounding of its own).
// Nibble position p holds element e(p) = p<4 ? 2p : 2(p-4)+1, i.e. (w >> 4i) & 0x000F000F carries
// elements (2i, 2i+1) in its (low, high) halves.
template <class DT>
__device__ __forceinline__ s16x8 dequant_word(uint32_t w, float s, float c) {
  uint32_t lo = w & 0x0F0F0F0Fu;         // bytes: p0 p2 p4 p6 -> e0 e4 e1 e5
  uint32_t hi = (w >> 4) & 0x0F0F0F0Fu;  // bytes: p1 p3 p5 p7 -> e2 e6 e3 e7
  // keep the masked words opaque so each element is ONE v_cvt_f32_ubyteN (byte select + convert)
  asm volatile("" : "+v"(lo), "+v"(hi));
  float e0 = fmaf((float)(lo & 0xffu), s, c);
  float e4 = fmaf((float)((lo >> 8) & 0xffu), s, c);
  float e1 = fmaf((float)((lo >> 16) & 0xffu), s, c);
  float e5 = fmaf((float)(lo >> 24), s, c);
  float e2 = fmaf((float)(hi & 0xffu), s, c);
  float e6 = fmaf((float)((hi >> 8) & 0xffu), s, c);
  float e3 = fmaf((float)((hi >> 16) & 0xffu), s, c);
  float e7 = fmaf((float)(hi >> 24), s, c);
  u32x4 r;
  r[0] = DT::pack2(e0, e1);
  r[1] = DT::pack2(e2, e3);
  r[2] = DT::pack2(e4, e5);
  r[3] = DT::pack2(e6, e7);
  return __builtin_bit_cast(s16x8, r);
}
