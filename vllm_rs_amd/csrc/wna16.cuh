// wna16.cuh — int4 (GPTQ/AWQ) primitives for the CDNA4 tile layout.
//
// Arithmetic contract of the fused GEMMs:  out = round_dt( Σ_g s_g · Σ_{k∈g} x_k·(q_k − z_g) ),
// i.e. the mathematically exact W4A16 product with ONE rounding at the output.  The 4-bit codes enter
// the MFMA as the exact 16-bit floats (C + q) (C = 128 for bf16, 64 for f16: "magic number" OR, one
// VALU op per two weights), accumulate in f32 per scale group, and the group result is fixed up with
//     acc += s_g · (acc_g − (C + z_g) · Σ_{k∈g} x_k)
// — 2 VALU ops per output instead of ~2.4 VALU ops per WEIGHT for a per-weight scale multiply.
// (Marlin rounds every dequantised weight to 16 bits before the MMA; that variant is kept as
// `dequant_word` for the explicit dequantisation entry point.  Per GEMM the two differ by 4-8 ulps of
// the output row's scale at the Llama-3-8B shapes (tests/test_gpu_tolerance.py reports both against
// float64); at full depth the engine stays within 1 ulp of the logit scale of either, tokens equal
// (bench.py parity_full_depth_marlin_rounded).)
#pragma once
#include "common.cuh"

__device__ __forceinline__ int awq_rev(int j) { return ((j & 1) << 2) | (j >> 1); }  // {0,4,1,5,2,6,3,7}

// The code q is OR-ed into mantissa bits of the constant C so that C + q is exact: bf16 takes the nibble
// at bits 0..3 of each half (C = 128: 7 mantissa bits, ulp 1), f16 at bits 4..7 (C = 64: ulp 1/16).
// With the tile layout's nibble order that is ONE v_and_or_b32 for the first register of a word and a
// shift + v_and_or_b32 for the other three: 7 VALU ops per 8 weights.  C + q is < 18x larger than
// |q - z|, so the later subtraction loses ~4 of f32's 24 bits — far below one 16-bit output ulp.
template <class DT>
struct Magic;
template <>
struct Magic<BF16> {
  static constexpr uint32_t bits = 0x43004300u;  // 128.0
  static constexpr uint32_t mask = 0x000F000Fu;  // q
  static constexpr int sh = 0;
  static constexpr float bias = 128.0f;
};
template <>
struct Magic<F16> {
  static constexpr uint32_t bits = 0x54005400u;  // 64.0
  static constexpr uint32_t mask = 0x00F000F0u;  // q << 4
  static constexpr int sh = 4;
  static constexpr float bias = 64.0f;
};

// One tiled word (8 codes of one column, 8 consecutive k) -> MFMA A fragment holding (C + q_e).
// Nibble position p holds element e(p) = p<4 ? 2p : 2(p-4)+1, so the nibbles at bits 4i and 16+4i are
// elements (2i, 2i+1) = (low, high) half of fragment register i.
template <class DT>
__device__ __forceinline__ s16x8 magic_word(uint32_t w) {
  constexpr int SH = Magic<DT>::sh;
  // gfx9 VOP3 takes no literals: with mask and constant in registers the AND+OR fuses into v_and_or_b32
  uint32_t mk = Magic<DT>::mask, bt = Magic<DT>::bits;
  asm("" : "+v"(mk));
  asm("" : "+v"(bt));
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t t = (4 * i >= SH) ? (w >> (4 * i >= SH ? 4 * i - SH : 0)) : (w << (4 * i >= SH ? 0 : SH - 4 * i));
    r[i] = (t & mk) | bt;
  }
  return __builtin_bit_cast(s16x8, r);
}

// Marlin-style explicit dequantisation: w_e = round_dt((q_e − z)·s) (one rounding; the fma is exact).
template <class DT>
__device__ __forceinline__ s16x8 dequant_word(uint32_t w, float s, float c) {
  uint32_t lo = w & 0x0F0F0F0Fu;         // bytes: p0 p2 p4 p6 -> e0 e4 e1 e5
  uint32_t hi = (w >> 4) & 0x0F0F0F0Fu;  // bytes: p1 p3 p5 p7 -> e2 e6 e3 e7
  asm volatile("" : "+v"(lo), "+v"(hi));  // keep the masked words opaque: one v_cvt_f32_ubyteN per element
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

// Scales / zero points in the MFMA *output* layout: lane (oct = lane>>4) owns output columns
// n4 .. n4+3 (n4 = nblock*16 + oct*4).  Returns s[r] and zc[r] = C + z[r].
template <class DT>
__device__ __forceinline__ void load_scale4_raw(const void* scales, const uint32_t* qzeros, int grp, int n4, int N, int layout,
                                                bool grouped, bool is_awq, u32x2& sraw, uint32_t& zraw) {
  const uint16_t* sp = static_cast<const uint16_t*>(scales);
  if (layout == VRA_SCALES_ROWMAJOR) {
    sraw = *reinterpret_cast<const u32x2*>(sp + (size_t)grp * N + n4);
  } else {
    uint32_t a = sp[vra_scale_index(grp, n4, N, layout, grouped)], b = sp[vra_scale_index(grp, n4 + 1, N, layout, grouped)];
    uint32_t c = sp[vra_scale_index(grp, n4 + 2, N, layout, grouped)], d = sp[vra_scale_index(grp, n4 + 3, N, layout, grouped)];
    sraw = u32x2{a | (b << 16), c | (d << 16)};
  }
  zraw = (is_awq && qzeros) ? qzeros[(size_t)grp * (N >> 3) + (n4 >> 3)] : 0x88888888u;
}
template <class DT>
__device__ __forceinline__ void unpack_scale4(const u32x2& sraw, uint32_t zraw, int n4, float* s, float* zc) {
  s[0] = DT::to_f32((uint16_t)(sraw[0] & 0xffffu));
  s[1] = DT::to_f32((uint16_t)(sraw[0] >> 16));
  s[2] = DT::to_f32((uint16_t)(sraw[1] & 0xffffu));
  s[3] = DT::to_f32((uint16_t)(sraw[1] >> 16));
  // AWQ word: column j of the 8 sits at nibble awq_rev(j); n4 % 8 is 0 or 4
  const int base = n4 & 4;
#pragma unroll
  for (int r = 0; r < 4; r++) zc[r] = Magic<DT>::bias + (float)((zraw >> (4 * awq_rev(base + r))) & 0xFu);
}
