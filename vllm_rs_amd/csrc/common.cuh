// common.cuh — device/host helpers shared by every kernel translation unit (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vllm_rs_amd.h"

#define VRA_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// error side channel (vra_last_error)
void vra_set_error(const char* fmt, ...);
#define VRA_CHECK_ARG(cond, ...)  \
  do {                            \
    if (!(cond)) {                \
      vra_set_error(__VA_ARGS__); \
      return;                     \
    }                             \
  } while (0)

static inline hipStream_t as_stream(int64_t s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------- 16-bit storage types
struct BF16 {
  static constexpr int id = VRA_BF16;
  typedef bf16x8_t vec8;
  static __device__ __forceinline__ float to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
  static __device__ __forceinline__ uint16_t from_f32(float f) {  // RNE, NaN quieted (== oracle f32_to_bf16)
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
  // two floats -> packed pair (lo = a)
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {a, b};
    bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ f32x4 mfma(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
struct F16 {
  static constexpr int id = VRA_F16;
  typedef f16x8_t vec8;
  static __device__ __forceinline__ float to_f32(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
  static __device__ __forceinline__ uint16_t from_f32(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    f16x2_t r = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, r);
  }
  static __device__ __forceinline__ f32x4 mfma(s16x8 a, s16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};
template <class DT>
__device__ __forceinline__ float rnd_dt(float v) { return DT::to_f32(DT::from_f32(v)); }

template <class DT>
__device__ __forceinline__ void unpack8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    f[2 * i] = DT::to_f32((uint16_t)(v[i] & 0xffffu));
    f[2 * i + 1] = DT::to_f32((uint16_t)(v[i] >> 16));
  }
}
template <class DT>
__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; i++) v[i] = DT::pack2(f[2 * i], f[2 * i + 1]);
  return v;
}

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------- counter-based RNG (== oracle hash32)
__host__ __device__ __forceinline__ uint32_t vra_hash32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
__host__ __device__ __forceinline__ float vra_hash_unit(uint64_t seed, uint64_t idx) {
  return (float)(vra_hash32(seed, idx) >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------- int4 tile layout
// word[((nb*KT + kt)*64 + lane)*4 + j]; lane = oct*16 + nn; rows k = kt*128 + j*32 + oct*8 + e;
// nibble position p holds e = (p<4) ? 2p : 2(p-4)+1.
__host__ __device__ __forceinline__ int vra_tile_e_of_p(int p) { return p < 4 ? 2 * p : 2 * (p - 4) + 1; }

// scale-tensor addressing: element (grp, n) of a [G, N] tensor in either layout
__device__ __forceinline__ int64_t vra_scale_index(int grp, int n, int N, int layout, int grouped) {
  if (layout == VRA_SCALES_ROWMAJOR) return (int64_t)grp * N + n;
  // inverse of wna16.rs:180-218: out[c*64 + i*8 + j] = in[c*64 + i + 8j] (grouped)
  int64_t flat = (int64_t)grp * N + n;
  if (grouped) {
    int r = (int)(flat & 63);
    return (flat & ~63ll) + (r & 7) * 8 + (r >> 3);
  }
  // channel-wise: out[c*32 + 8i + j] = in[c*32 + 2i + base[j]], base = {0,1,8,9,16,17,24,25}
  int r = (int)(flat & 31);
  int i = (r & 7) >> 1, b = (r & 1) + ((r >> 3) << 1);  // r = 2i + (b&1) + 8*(b>>1)
  return (flat & ~31ll) + 8 * i + b;
}
