// kvcache.cuh — element access to the paged KV cache in either of its two storage formats:
//   16-bit (the model dtype), or FP8 E4M3 (OCP, 1 byte per element, scale 1.0) — the reference's `fp8_kvcache` option
//   (EngineConfig.fp8_kvcache -> KVCacheAllocator dtype_size 1, cache dtype U8: kvcache_allocator.rs:188-193,776;
//   PagedAttention::new(.., fp8_kvcache), attention.rs:607-616).
// E4M3 values are exactly representable in bf16 and f16, so reads widen exactly and feed the same MFMA path; writes round
// to nearest even and saturate at +-448 (gfx950 v_cvt_pk_fp8_f32 is the OCP format; the clamp is explicit).
#pragma once
#include "common.cuh"

#define VRA_FP8_E4M3_ID 3

template <bool KV8>
struct KVT {
  typedef uint16_t elem;
};
template <>
struct KVT<true> {
  typedef uint8_t elem;
};

// saturating at +-448; a NaN stays a NaN (E4M3 has a NaN encoding: a cache that turned NaN keys into -448 would hide an
// upstream numerical failure that the 16-bit cache shows)
__device__ __forceinline__ float vra_e4m3_clamp(float x) { return x != x ? x : fminf(fmaxf(x, -448.f), 448.f); }
__device__ __forceinline__ uint32_t vra_f32x4_to_e4m3(float a, float b, float c, float d) {
  a = vra_e4m3_clamp(a), b = vra_e4m3_clamp(b), c = vra_e4m3_clamp(c), d = vra_e4m3_clamp(d);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}
__device__ __forceinline__ void vra_e4m3x4_to_f32(uint32_t w, float* f) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
  f[0] = lo[0], f[1] = lo[1], f[2] = hi[0], f[3] = hi[1];
}
// 8 model-dtype values <-> 8 cache bytes
template <class DT>
__device__ __forceinline__ u32x2 vra_pack_e4m3x8(const u32x4& v) {
  float f[8];
  unpack8<DT>(v, f);
  return u32x2{vra_f32x4_to_e4m3(f[0], f[1], f[2], f[3]), vra_f32x4_to_e4m3(f[4], f[5], f[6], f[7])};
}
template <class DT>
__device__ __forceinline__ u32x4 vra_unpack_e4m3x8(const u32x2& w) {
  float f[8];
  vra_e4m3x4_to_f32(w[0], f);
  vra_e4m3x4_to_f32(w[1], f + 4);
  return pack8<DT>(f);
}
// 8 consecutive cache elements (a K row octet) as a 16-bit x 8 MFMA fragment
template <class DT, bool KV8>
__device__ __forceinline__ u32x4 kv_load8(const typename KVT<KV8>::elem* p) {
  if constexpr (KV8) return vra_unpack_e4m3x8<DT>(*reinterpret_cast<const u32x2*>(p));
  else return *reinterpret_cast<const u32x4*>(p);
}
// 4 consecutive cache elements (4 tokens of one V channel) as 4 x 16 bit
template <class DT, bool KV8>
__device__ __forceinline__ u32x2 kv_load4(const typename KVT<KV8>::elem* p) {
  if constexpr (KV8) {
    float f[4];
    vra_e4m3x4_to_f32(*reinterpret_cast<const uint32_t*>(p), f);
    return u32x2{DT::pack2(f[0], f[1]), DT::pack2(f[2], f[3])};
  } else {
    return *reinterpret_cast<const u32x2*>(p);
  }
}
template <class DT, bool KV8>
__device__ __forceinline__ void kv_store8(typename KVT<KV8>::elem* p, const u32x4& v) {
  if constexpr (KV8) *reinterpret_cast<u32x2*>(p) = vra_pack_e4m3x8<DT>(v);
  else *reinterpret_cast<u32x4*>(p) = v;
}
// the value a cache read will return for a model-dtype value written to it
template <class DT, bool KV8>
__device__ __forceinline__ u32x4 kv_roundtrip8(const u32x4& v) {
  if constexpr (KV8) return vra_unpack_e4m3x8<DT>(vra_pack_e4m3x8<DT>(v));
  else return v;
}
