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

// ---------------------------------------------------------------- MFMA (gfx950 v_mfma_f32_16x16x32_*)
// Two hazards of the double-rate 16x16x32 instruction were found on MI355X with hipcc / ROCm 7.2
// (both show up as result registers 2,3 of the 4-register destination being wrong):
//  (1) builtin form: when SrcA/SrcB die at the instruction the register allocator may place the
//      destination ON TOP of them (legal for the 4-pass 16x16x16 form).  The hardware still reads the
//      upper half of A/B after it has started writing D.  It happens when C is the literal 0 or when
//      control flow makes the compiler move accumulators between registers (D != C).
//  (2) inline-asm form with a tied accumulator: the compiler does not know the statement is an MFMA and
//      may copy the accumulator one wait state later, reading it half written.
// The builtin cannot be told to keep D off A/B (the allocator does it even in straight-line code), so the
// instruction is issued through inline asm with the accumulator TIED ("+v": D = C, never A or B), every
// MFMA region is kept branch free (no accumulator copies at control-flow merges), every chain starts from
// vra_zero_acc(), VRA_MFMA_DRAIN() (12 wait states + scheduling barrier) stands between the last MFMA of
// a chain and the first non-MFMA use of its accumulator, `s_nop 1` in front of each MFMA covers VALU-written
// operands — and `make` runs tools/check_mfma_overlap.py over the generated ISA: the build FAILS if any
// v_mfma has D overlapping A/B or if anything but an accumulating MFMA touches D within 12 wait states.
#define VRA_MFMA_DRAIN()                \
  do {                                  \
    asm volatile("s_nop 7\n\ts_nop 4"); \
    __builtin_amdgcn_sched_barrier(0);  \
  } while (0)
// xor-16 / xor-32 lane exchanges as VALU ops (gfx950 v_permlane16_swap / v_permlane32_swap) instead of __shfl_xor, which
// lowers to ds_bpermute_b32 — an LDS round trip (~100+ cycles) per step.  permlane32_swap(a, b) exchanges a[32..63] with
// b[0..31]; with a = b = x the two results hold x's lower half twice and x's upper half twice, so combining them lane by
// lane is the xor-32 combination; permlane16_swap does the same with the odd and even rows of 16 lanes (xor 16).
__device__ __forceinline__ float vra_xor32_max(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float vra_xor16_max(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float vra_xor32_sum(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float vra_xor16_sum(float x) {
  const unsigned u = __float_as_uint(x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// s_setprio takes an immediate: a wave-uniform priority 0..3 through scalar branches (no memory operation inside, so the
// s_waitcnt bookkeeping of the surrounding loop is unaffected)
__device__ __forceinline__ void vra_setprio_dyn(int p) {
  if (p == 0) __builtin_amdgcn_s_setprio(0);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
}
__device__ __forceinline__ f32x4 vra_zero_acc() {
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  asm volatile("" : "+v"(z));
  return z;
}

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
  static __device__ __forceinline__ void mfma(f32x4& acc, s16x8 a, s16x8 b) {  // acc += A·B
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ void mfma0(f32x4& acc, s16x8 a, s16x8 b) {  // acc = A·B (early clobber: D never on A/B)
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
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
  static __device__ __forceinline__ void mfma(f32x4& acc, s16x8 a, s16x8 b) {  // acc += A·B
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ void mfma0(f32x4& acc, s16x8 a, s16x8 b) {  // acc = A·B (early clobber: D never on A/B)
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
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

// Σ of the 8 16-bit floats of one octet in f32, fixed pairwise order.  (v_dot2_f32_bf16 against (1,1) would be 4 ops
// instead of ~15, but on gfx950 / ROCm 7.2 `__builtin_amdgcn_fdot2_f32_bf16` returned sums that are off by up to
// several units for |x| < 2 — measured with exp/dot2.hip — so the conversion + add form stays.)
template <class DT>
__device__ __forceinline__ float octet_sum(const u32x4& v);

// ---------------------------------------------------------------- reductions
// Sum over aligned groups of 4 / 16 consecutive lanes with DPP modifiers (1 VALU op per step, no LDS traffic):
// quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror.  `__shfl_xor` lowers to ds_bpermute_b32 on
// gfx950 — an LDS-pipeline round trip per step, which made the x staging of kernel C 4x slower than its loads.
template <int CTRL>
__device__ __forceinline__ float vra_dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v) {  // every lane of an aligned quad gets the quad's sum
  v += vra_dpp_f<0xB1>(v);
  v += vra_dpp_f<0x4E>(v);
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {  // every lane of an aligned group of 16 gets the group's sum
  v = quad_sum(v);
  v += vra_dpp_f<0x141>(v);  // row_half_mirror: lane i <-> 7-i (the other quad of the half row)
  v += vra_dpp_f<0x140>(v);  // row_mirror: lane i <-> 15-i (the other half row)
  return v;
}
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

template <class DT>
__device__ __forceinline__ float octet_sum(const u32x4& v) {
  float f[8];
  unpack8<DT>(v, f);
  return ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
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
