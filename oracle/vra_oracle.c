/*
 * vra_oracle.c — CPU ORACLE for the quantized-forward hot path of guoqingbao/vllm.rs.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in vllm_rs_amd/ (the product) may import, link or call this
 * file; only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, as the
 * checker.  PARITY PINNING: the reference's own tests hold no golden vector for this path
 *  and its arithmetic lives in un-vendored dependencies
 * (attention-rs 0.5.5 @0bf5727, candle 0.8.3 @68b6d74, Cargo.toml:14-15,48), so this oracle is
 * pinned against (a) hand-built known-answer vectors of the public GPTQ/AWQ bit layouts and
 * (b) HuggingFace `transformers` LlamaForCausalLM / Qwen2ForCausalLM logits generated in the dev
 * container (tests/golden/make_golden.py), which is the check docs/add_model.md:75 prescribes.
 * Against the reference's *own* tests parity is therefore "unpinned".
 *
 * Each function cites the reference file:line whose behaviour it restates.  Arithmetic contract:
 * storage dtype bf16 (or f16), every reference op rounds its result once to the storage dtype,
 * accumulations are carried in double (the GPU accumulates in f32; the difference is far below
 * one storage ulp and is what the test tolerances cover).
 *
 * Build: make -C oracle   (gcc -O3 -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DT_BF16 0
#define DT_F16 1
#define DT_F32 2

/* ---------------------------------------------------------------- scalar conversions --------- */
static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f32_to_bf16(float f) { /* round-to-nearest-even, NaN preserved */
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float f16_to_f32(uint16_t h) {
  uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ffu, u;
  if (e == 0) {
    if (m == 0) u = s;
    else {
      int sh = 0;
      while (!(m & 0x400u)) { m <<= 1; sh++; }
      m &= 0x3ffu;
      u = s | ((uint32_t)(127 - 15 - sh + 1) << 23) | (m << 13);
    }
  } else if (e == 31) u = s | 0x7f800000u | (m << 13);
  else u = s | ((e + 112) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f32_to_f16(float f) { /* RNE */
  uint32_t u;
  memcpy(&u, &f, 4);
  uint32_t s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(s | 0x7e00u);
  if (a >= 0x47800000u) return (uint16_t)(s | 0x7c00u); /* >= 65536 → inf (65520 rounds to inf too) */
  if (a < 0x38800000u) {                                 /* subnormal or zero in f16 */
    if (a < 0x33000000u) return (uint16_t)s;             /* < 2^-25 → 0 */
    int e = (int)(a >> 23);
    uint32_t m = (a & 0x7fffffu) | 0x800000u;
    int shift = 126 - e; /* 14..24 */
    uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1))) r++;
    return (uint16_t)(s | r);
  }
  uint32_t r = a - 0x38000000u; /* rebias */
  uint32_t rem = r & 0x1fffu;
  r >>= 13;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
  if (r >= 0x7c00u) return (uint16_t)(s | 0x7c00u);
  return (uint16_t)(s | r);
}
/* OCP FP8 E4M3 ("e4m3fn": 4 exponent bits, bias 7, 3 mantissa bits, no infinities, 0x7f/0xff = NaN, max 448) — the storage
 * format of the reference's fp8 KV cache (kvcache_allocator.rs:188-193,776: dtype_size 1, cache dtype U8), scale 1.0.
 * Round to nearest even, saturating at +-448. */
#define DT_FP8 3
static inline uint8_t f32_to_e4m3(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  uint8_t s = (uint8_t)((u >> 24) & 0x80u);
  uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint8_t)(s | 0x7f);
  float af = fabsf(f);
  if (af >= 448.0f) return (uint8_t)(s | 0x7e);
  int e32 = (int)(a >> 23) - 127;
  if (e32 >= -6) { /* normal in e4m3 */
    uint32_t m = a & 0x7fffffu, r = m >> 20, rem = m & 0xfffffu;
    if (rem > 0x80000u || (rem == 0x80000u && (r & 1))) r++;
    int e = e32 + 7;
    if (r == 8) {
      r = 0;
      e++;
    }
    if (e > 15 || (e == 15 && r == 7)) return (uint8_t)(s | 0x7e);
    return (uint8_t)(s | (e << 3) | r);
  }
  /* subnormal: multiples of 2^-9; k = RNE(af * 512) in 0..8 (8 = the smallest normal, 0x08) */
  float sc = af * 512.0f;
  int k = (int)sc;
  float frac = sc - (float)k;
  if (frac > 0.5f || (frac == 0.5f && (k & 1))) k++;
  return (uint8_t)(s | k);
}
static inline float e4m3_to_f32(uint8_t b) {
  int e = (b >> 3) & 15, m = b & 7;
  float v;
  if ((b & 0x7f) == 0x7f) return NAN;
  if (e == 0) v = (float)m * (1.0f / 512.0f);
  else v = (1.0f + (float)m / 8.0f) * ldexpf(1.0f, e - 7);
  return (b & 0x80) ? -v : v;
}
void orc_f32_to_e4m3(const float* in, uint8_t* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) out[i] = f32_to_e4m3(in[i]);
}
void orc_e4m3_to_f32(const uint8_t* in, float* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) out[i] = e4m3_to_f32(in[i]);
}
static inline float ld(const void* p, int64_t i, int dt) {
  if (dt == DT_FP8) return e4m3_to_f32(((const uint8_t*)p)[i]);
  if (dt == DT_BF16) return bf16_to_f32(((const uint16_t*)p)[i]);
  if (dt == DT_F16) return f16_to_f32(((const uint16_t*)p)[i]);
  return ((const float*)p)[i];
}
static inline float rnd(float v, int dt) { /* round to storage dtype, return as f32 */
  if (dt == DT_BF16) return bf16_to_f32(f32_to_bf16(v));
  if (dt == DT_F16) return f16_to_f32(f32_to_f16(v));
  return v;
}
static inline void st(void* p, int64_t i, float v, int dt) {
  if (dt == DT_FP8) ((uint8_t*)p)[i] = f32_to_e4m3(v);
  else if (dt == DT_BF16) ((uint16_t*)p)[i] = f32_to_bf16(v);
  else if (dt == DT_F16) ((uint16_t*)p)[i] = f32_to_f16(v);
  else ((float*)p)[i] = v;
}
void orc_f32_to_bf16(const float* in, uint16_t* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) out[i] = f32_to_bf16(in[i]);
}
void orc_bf16_to_f32(const uint16_t* in, float* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) out[i] = bf16_to_f32(in[i]);
}
void orc_f32_to_f16(const float* in, uint16_t* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) out[i] = f32_to_f16(in[i]);
}
void orc_f16_to_f32(const uint16_t* in, float* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) out[i] = f16_to_f32(in[i]);
}

/* ---------------------------------------------------------------- synthetic data ------------- */
/* Counter-based generator shared (by restatement) with vllm_rs_amd/csrc/common.h vra_hash32:
 * splitmix64 finaliser of seed*GOLDEN + idx; BASELINE.md / SURVEY §8d synthetic recipe. */
static inline uint32_t hash32(uint64_t seed, uint64_t idx) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + idx * 0xD1B54A32D192ED03ull + 0x8CB92BA72F3D8DD7ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
static inline float hash_unit(uint64_t seed, uint64_t idx) { /* [0,1) with 24 bits */
  return (float)(hash32(seed, idx) >> 8) * (1.0f / 16777216.0f);
}
void orc_fill_hash_u32(uint32_t* out, int64_t n, uint64_t seed) {
#pragma omp parallel for
  for (int64_t i = 0; i < n; i++) out[i] = hash32(seed, (uint64_t)i);
}
/* AWQ zero points of the synthetic checkpoints (BASELINE.md recipe, revised in round 4): every nibble of a hash word goes
 * through a table concentrated on 8 (6:1 7:3 8:8 9:3 10:1 of 16), like the zero points of real AWQ checkpoints. */
void orc_fill_awq_zeros(uint32_t* out, int64_t n, uint64_t seed) {
  const uint64_t lut = 0xA999888888887776ull;
#pragma omp parallel for
  for (int64_t i = 0; i < n; i++) {
    const uint32_t h = hash32(seed, (uint64_t)i);
    uint32_t w = 0;
    for (int p = 0; p < 8; p++) w |= (uint32_t)((lut >> (4 * ((h >> (4 * p)) & 0xFu))) & 0xFull) << (4 * p);
    out[i] = w;
  }
}
void orc_fill_uniform(void* out, int64_t n, uint64_t seed, float lo, float hi, int dt) {
#pragma omp parallel for
  for (int64_t i = 0; i < n; i++) st(out, i, lo + (hi - lo) * hash_unit(seed, (uint64_t)i), dt);
}
/* approx normal: Irwin–Hall sum of 4 uniforms, variance-normalised (deterministic, cheap) */
void orc_fill_normal(void* out, int64_t n, uint64_t seed, float mean, float std, int dt) {
#pragma omp parallel for
  for (int64_t i = 0; i < n; i++) {
    float s = hash_unit(seed, 4ull * i) + hash_unit(seed, 4ull * i + 1) +
              hash_unit(seed, 4ull * i + 2) + hash_unit(seed, 4ull * i + 3);
    st(out, i, mean + std * (s - 2.0f) * 1.7320508f, dt);
  }
}

/* ---------------------------------------------------------------- int4 formats --------------- */
/* GPTQ (AutoGPTQ v1) qweight [K/8, N] u32 — tensor shape per wna16.rs:56-63; nibble i of word
 * (k/8, n) at bits 4*(k%8) is the public on-disk convention (SURVEY §8c "what pins the layout"). */
void orc_gptq_unpack(const uint32_t* qw, int K, int N, uint8_t* idx /*[K,N]*/) {
#pragma omp parallel for
  for (int k = 0; k < K; k++)
    for (int n = 0; n < N; n++) idx[(int64_t)k * N + n] = (qw[(int64_t)(k / 8) * N + n] >> (4 * (k % 8))) & 0xF;
}
void orc_gptq_pack(const uint8_t* idx, int K, int N, uint32_t* qw) {
  memset(qw, 0, (size_t)K / 8 * N * 4);
  for (int k = 0; k < K; k++)
    for (int n = 0; n < N; n++) qw[(int64_t)(k / 8) * N + n] |= (uint32_t)(idx[(int64_t)k * N + n] & 0xF) << (4 * (k % 8));
}
/* GPTQ qzeros [G, N/8] u32 (wna16.rs:127-132): packed along N, nibble (n%8); stored value is z-1 */
void orc_gptq_unpack_zeros(const uint32_t* qz, int G, int N, uint8_t* z /*[G,N], = stored+1*/) {
  for (int g = 0; g < G; g++)
    for (int n = 0; n < N; n++) z[(int64_t)g * N + n] = (uint8_t)(((qz[(int64_t)g * (N / 8) + n / 8] >> (4 * (n % 8))) & 0xF) + 1);
}
/* AWQ (AutoAWQ GEMM) qweight [K, N/8] u32 (wna16.rs:64-69): packed along N with nibble order
 * [0,2,4,6,1,3,5,7]: nibble i holds column n0 + order[i]  ⇒  column n0+j sits at nibble rev[j]. */
static const int AWQ_REV[8] = {0, 4, 1, 5, 2, 6, 3, 7};
void orc_awq_unpack(const uint32_t* qw, int K, int N, uint8_t* idx /*[K,N]*/) {
#pragma omp parallel for
  for (int k = 0; k < K; k++)
    for (int n = 0; n < N; n++) idx[(int64_t)k * N + n] = (qw[(int64_t)k * (N / 8) + n / 8] >> (4 * AWQ_REV[n % 8])) & 0xF;
}
void orc_awq_pack(const uint8_t* idx, int K, int N, uint32_t* qw) {
  memset(qw, 0, (size_t)K * (N / 8) * 4);
  for (int k = 0; k < K; k++)
    for (int n = 0; n < N; n++) qw[(int64_t)k * (N / 8) + n / 8] |= (uint32_t)(idx[(int64_t)k * N + n] & 0xF) << (4 * AWQ_REV[n % 8]);
}
/* AWQ qzeros [G, N/8] u32: same packing as AWQ qweight rows, no offset */
void orc_awq_unpack_zeros(const uint32_t* qz, int G, int N, uint8_t* z) { orc_awq_unpack(qz, G, N, z); }

/* CDNA4 tile layout produced by gptq_repack / awq_repack (include/vllm_rs_amd.h §A):
 * word[((nb*KT + kt)*64 + lane)*4 + j], lane = oct*16 + nn, holds column n = nb*16+nn and the 8
 * rows k = kt*128 + j*32 + oct*8 + e; nibble position p holds e = (p<4) ? 2p : 2(p-4)+1. */
static inline int tile_e_of_p(int p) { return p < 4 ? 2 * p : 2 * (p - 4) + 1; }
void orc_tile_from_indices(const uint8_t* idx /*[K,N]*/, int K, int N, uint32_t* out) {
  int KT = K / 128, NB = N / 16;
#pragma omp parallel for
  for (int nb = 0; nb < NB; nb++)
    for (int kt = 0; kt < KT; kt++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 4; j++) {
          int nn = lane & 15, oct = lane >> 4, n = nb * 16 + nn;
          uint32_t w = 0;
          for (int p = 0; p < 8; p++) {
            int k = kt * 128 + j * 32 + oct * 8 + tile_e_of_p(p);
            w |= (uint32_t)(idx[(int64_t)k * N + n] & 0xF) << (4 * p);
          }
          out[(((int64_t)nb * KT + kt) * 64 + lane) * 4 + j] = w;
        }
}
void orc_tile_to_indices(const uint32_t* tiled, int K, int N, uint8_t* idx) {
  int KT = K / 128, NB = N / 16;
#pragma omp parallel for
  for (int nb = 0; nb < NB; nb++)
    for (int kt = 0; kt < KT; kt++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 4; j++) {
          int nn = lane & 15, oct = lane >> 4, n = nb * 16 + nn;
          uint32_t w = tiled[(((int64_t)nb * KT + kt) * 64 + lane) * 4 + j];
          for (int p = 0; p < 8; p++) {
            int k = kt * 128 + j * 32 + oct * 8 + tile_e_of_p(p);
            idx[(int64_t)k * N + n] = (w >> (4 * p)) & 0xF;
          }
        }
}
/* gptq_repack restatement (call site src/utils/gptq.rs:325-331): rows=K/8, cols=N */
void orc_gptq_repack(const uint32_t* qw, int rows, int cols, uint32_t* out) {
  int K = rows * 8, N = cols;
  uint8_t* idx = (uint8_t*)malloc((size_t)K * N);
  orc_gptq_unpack(qw, K, N, idx);
  orc_tile_from_indices(idx, K, N, out);
  free(idx);
}
/* awq_repack restatement (src/utils/gptq.rs:316-323): rows=K, cols=N/8 */
void orc_awq_repack(const uint32_t* qw, int rows, int cols, uint32_t* out) {
  int K = rows, N = cols * 8;
  uint8_t* idx = (uint8_t*)malloc((size_t)K * N);
  orc_awq_unpack(qw, K, N, idx);
  orc_tile_from_indices(idx, K, N, out);
  free(idx);
}

/* marlin_permute_scales (src/models/layers/wna16.rs:180-218): reshape [-1,64] and take columns
 * perm[i*8+j] = i + 8j (grouped), or [-1,32] with perm[4i'..] = 2i + {0,1,8,9,16,17,24,25}
 * (channel-wise). 16-bit elements. */
void orc_marlin_permute_scales(const uint16_t* in, uint16_t* out, int rows, int n, int grouped) {
  int64_t total = (int64_t)rows * n;
  if (grouped) {
    int perm[64];
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 8; j++) perm[i * 8 + j] = i + 8 * j;
    for (int64_t c = 0; c < total / 64; c++)
      for (int t = 0; t < 64; t++) out[c * 64 + t] = in[c * 64 + perm[t]];
  } else {
    static const int base[8] = {0, 1, 8, 9, 16, 17, 24, 25};
    int perm[32];
    for (int i = 0; i < 4; i++)
      for (int j = 0; j < 8; j++) perm[i * 8 + j] = 2 * i + base[j];
    for (int64_t c = 0; c < total / 32; c++)
      for (int t = 0; t < 32; t++) out[c * 32 + t] = in[c * 32 + perm[t]];
  }
}

/* Dequantisation as the published Marlin algorithm does it (IST-DASLab/marlin, vLLM gptq_marlin
 * `dequant` + `scale`): the 4-bit code minus zero point is exact in the 16-bit type, then ONE
 * multiply by the scale rounded to the 16-bit type:  w = round_dt((q - z) * s).
 * idx [K,N] u8, zeros [G,N] u8 or NULL (=> 8, GPTQ symmetric uint4b8), scales [G,N] dt row-major.
 * group_size -1 => one group. Output w [K,N] dt. */
void orc_dequant(const uint8_t* idx, const uint8_t* zeros, const void* scales, int K, int N,
                 int group_size, int dt, void* w) {
  int g = group_size > 0 ? group_size : K;
#pragma omp parallel for
  for (int k = 0; k < K; k++)
    for (int n = 0; n < N; n++) {
      int grp = k / g;
      float s = ld(scales, (int64_t)grp * N + n, dt);
      int z = zeros ? zeros[(int64_t)grp * N + n] : 8;
      st(w, (int64_t)k * N + n, (float)((int)idx[(int64_t)k * N + n] - z) * s, dt);
    }
}
/* GPTQMatMul (src/utils/gptq.rs:26-204 → marlin_* / gemm_half_q_half_alt) + bias
 * (wna16.rs:296-300) + optional residual add (llama.rs:126,130):
 * out = rnd(rnd(rnd(Σ_k x·w) + bias) + residual); w [K,N] dt already dequantised. */
void orc_gemm_wdense(const void* x, const void* w, const void* bias, const void* residual, int M,
                     int K, int N, int dt, void* out) {
#pragma omp parallel for
  for (int n = 0; n < N; n++) {
    for (int m = 0; m < M; m++) {
      double acc = 0.0;
      for (int k = 0; k < K; k++) acc += (double)ld(x, (int64_t)m * K + k, dt) * (double)ld(w, (int64_t)k * N + n, dt);
      float v = rnd((float)acc, dt);
      if (bias) v = rnd(v + ld(bias, n, dt), dt);
      if (residual) v = rnd(v + ld(residual, (int64_t)m * N + n, dt), dt);
      st(out, (int64_t)m * N + n, v, dt);
    }
  }
}
/* The fused W4A16 GEMM of the build: the mathematically exact product with the true
 * GPTQ/AWQ weights W = s·(q − z) and ONE rounding at the output,
 *     out = rnd( Σ_k x_k · s_g(k) · (q_k − z_g(k)) )   [+ bias, + residual, each rounded as the reference's ops].
 * (The Marlin kernels behind src/utils/gptq.rs:116-178 additionally round every dequantised weight
 * to 16 bits before the MMA — orc_dequant/orc_gemm_wdense above, and `wround` here, restate that
 * variant.  MEASURED distance between the two at the Llama-3-8B shapes: 4-8 ulps of the output row's
 * scale per GEMM (tests/test_gpu_tolerance.py reports both against float64), NOT "< 1 ulp" as this
 * comment claimed until round 5; the reference's kernel sources are not in the tree to arbitrate, so
 * the full-depth reports carry both (tests/full_depth.py: `reference_order`, `marlin_rounded`).)
 * wround != 0: every dequantised weight rounded to dt first, w = rnd((q - z) * s)  (Marlin's arithmetic).
 * x [M,K] dt; idx [K,N]; scales [G,N]; zeros [G,N] or NULL. */
static void wna16_gemm_rs(const void* x, const uint8_t* idx, const uint8_t* zeros, const void* scales,
                          const void* bias, const void* residual, const float* row_scale, int M, int K, int N, int group_size,
                          int dt, void* out, int wround) {
  /* Blocked for the caches (32 columns x 64 rows of accumulators per thread, k outermost inside a block, weights
   * dequantised once per (k, column) and row block); every output is still the SEQUENTIAL sum over k = 0..K-1 in double,
   * i.e. bit-identical to the plain triple loop. */
  enum { NB = 32, MB = 64 };
  int g = group_size > 0 ? group_size : K;
  float* xf = (float*)malloc((size_t)M * K * sizeof(float));
  for (int64_t i = 0; i < (int64_t)M * K; i++) xf[i] = ld(x, i, dt);
  int nblk = (N + NB - 1) / NB, mblk = (M + MB - 1) / MB;
#pragma omp parallel
  {
    double acc[MB][NB];
    double wv[NB];
#pragma omp for collapse(2) schedule(dynamic)
    for (int nb = 0; nb < nblk; nb++)
      for (int mb = 0; mb < mblk; mb++) {
        int n0 = nb * NB, nc = N - n0 < NB ? N - n0 : NB;
        int m0 = mb * MB, mc = M - m0 < MB ? M - m0 : MB;
        for (int m = 0; m < mc; m++)
          for (int c = 0; c < nc; c++) acc[m][c] = 0.0;
        for (int k = 0; k < K; k++) {
          int grp = k / g;
          for (int c = 0; c < nc; c++) {
            int n = n0 + c;
            int z = zeros ? zeros[(int64_t)grp * N + n] : 8;
            int qz = (int)idx[(int64_t)k * N + n] - z;
            float sc = ld(scales, (int64_t)grp * N + n, dt);
            wv[c] = wround ? (double)rnd((float)qz * sc, dt) : (double)qz * (double)sc;  /* (q - z) * s is exact in f32: 5 x 11 bits */
          }
          for (int m = 0; m < mc; m++) {
            double xv = (double)xf[(int64_t)(m0 + m) * K + k];
            for (int c = 0; c < nc; c++) acc[m][c] += xv * wv[c];
          }
        }
        for (int m = 0; m < mc; m++)
          for (int c = 0; c < nc; c++) {
            int n = n0 + c;
            float v = (float)acc[m][c];
            if (row_scale) v = v * row_scale[m0 + m];  /* the deferred RMSNorm factor, on the f32 dot product (see orc_rms_norm_deferred) */
            v = rnd(v, dt);
            if (bias) v = rnd(v + ld(bias, n, dt), dt);
            if (residual) v = rnd(v + ld(residual, (int64_t)(m0 + m) * N + n, dt), dt);
            st(out, (int64_t)(m0 + m) * N + n, v, dt);
          }
      }
  }
  free(xf);
}
void orc_wna16_gemm(const void* x, const uint8_t* idx, const uint8_t* zeros, const void* scales,
                    const void* bias, const void* residual, int M, int K, int N, int group_size,
                    int dt, void* out) {
  wna16_gemm_rs(x, idx, zeros, scales, bias, residual, NULL, M, K, N, group_size, dt, out, 0);
}
/* Marlin's arithmetic (weights rounded to 16 bits before the product) without materialising the dense tensor:
 * bit-identical to orc_dequant + orc_gemm_wdense (tests/test_oracle.py). */
void orc_wna16_gemm_marlin(const void* x, const uint8_t* idx, const uint8_t* zeros, const void* scales,
                           const void* bias, const void* residual, int M, int K, int N, int group_size,
                           int dt, void* out) {
  wna16_gemm_rs(x, idx, zeros, scales, bias, residual, NULL, M, K, N, group_size, dt, out, 1);
}
/* The same product with a per-row f32 factor applied to the f32 sum BEFORE the output rounding:
 *     out = rnd( row_scale[m] · Σ_k x_k · s_g(k) · (q_k − z_g(k)) )   [+ bias, + residual as above]
 * — the second, stated order of RMSNorm ∘ GEMV (orc_rms_norm_deferred below). */
void orc_wna16_gemm_row_scale(const void* x, const uint8_t* idx, const uint8_t* zeros, const void* scales,
                              const void* bias, const void* residual, const float* row_scale, int M, int K, int N,
                              int group_size, int dt, void* out) {
  wna16_gemm_rs(x, idx, zeros, scales, bias, residual, row_scale, M, K, N, group_size, dt, out, 0);
}
/* Fast path of the same arithmetic straight from GPTQ-packed words (timed CPU baseline, bs small):
 * qw [K/8,N] u32 (gptq layout), symmetric zero 8, scales [G,N] dt. f32 accumulation. */
void orc_gptq_gemv_fast(const void* x, const uint32_t* qw, const void* scales, int M, int K, int N,
                        int group_size, int dt, void* out) {
  int g = group_size > 0 ? group_size : K;
  float* xf = (float*)malloc((size_t)M * K * sizeof(float));
  for (int64_t i = 0; i < (int64_t)M * K; i++) xf[i] = ld(x, i, dt);
#pragma omp parallel
  {
    float* acc = (float*)malloc((size_t)M * 64 * sizeof(float));
#pragma omp for schedule(static)
    for (int nb = 0; nb < N / 64; nb++) {
      memset(acc, 0, (size_t)M * 64 * sizeof(float));
      for (int kw = 0; kw < K / 8; kw++) {
        int grp = (kw * 8) / g;
        for (int c = 0; c < 64; c++) {
          int n = nb * 64 + c;
          uint32_t w = qw[(int64_t)kw * N + n];
          float s = ld(scales, (int64_t)grp * N + n, dt);
          for (int e = 0; e < 8; e++) {
            float wv = (float)((int)((w >> (4 * e)) & 0xF) - 8) * s;
            for (int m = 0; m < M; m++) acc[m * 64 + c] += xf[(int64_t)m * K + kw * 8 + e] * wv;
          }
        }
      }
      for (int m = 0; m < M; m++)
        for (int c = 0; c < 64; c++) st(out, (int64_t)m * N + nb * 64 + c, acc[m * 64 + c], dt);
    }
    free(acc);
  }
  free(xf);
}

/* dense Linear::forward (src/models/layers/linear.rs:75-123): out = x·Wᵀ (+bias), W [N,K].
 * out_dt F32 reproduces `.to_dtype(F32)` applied to the dt-rounded result (llama.rs:317-319). */
void orc_dense_gemm(const void* x, const void* w, const void* bias, int M, int K, int N, int dt,
                    int out_dt, void* out) {
#pragma omp parallel for
  for (int n = 0; n < N; n++)
    for (int m = 0; m < M; m++) {
      double acc = 0.0;
      for (int k = 0; k < K; k++) acc += (double)ld(x, (int64_t)m * K + k, dt) * (double)ld(w, (int64_t)n * K + k, dt);
      float v = rnd((float)acc, dt);
      if (bias) v = rnd(v + ld(bias, n, dt), dt);
      st(out, (int64_t)m * N + n, v, out_dt);
    }
}

/* ---------------------------------------------------------------- norms / elementwise -------- */
/* NormX::forward → candle_nn::RmsNorm (src/models/layers/others.rs:11-29): f32 math, one rounding */
void orc_rms_norm(const void* x, const void* w, int T, int H, float eps, int dt, void* out) {
#pragma omp parallel for
  for (int t = 0; t < T; t++) {
    double ss = 0.0;
    for (int i = 0; i < H; i++) {
      double v = ld(x, (int64_t)t * H + i, dt);
      ss += v * v;
    }
    float r = 1.0f / sqrtf((float)(ss / H) + eps);
    for (int i = 0; i < H; i++) st(out, (int64_t)t * H + i, ld(x, (int64_t)t * H + i, dt) * r * ld(w, i, dt), dt);
  }
}
/* RMSNorm with the normalisation factor DEFERRED to the consumer GEMV (round 5; the order of the engine's 1..4-row decode launches,
 * vllm_rs_amd/csrc/gemv_q4s.cuh): rstd = 1/sqrt(mean(x²) + eps) is one scalar per row and commutes with the matrix product,
 *     Σ_k (x_k · rstd · g_k) · w_kn  =  rstd · Σ_k (x_k · g_k) · w_kn,
 * so the kernel stages  out = rnd(x · g)  (ONE rounding per element, as NormX::forward has — of x·g instead of x·rstd·g) without
 * waiting for the row's sum of squares, and multiplies the f32 dot products by rstd (returned here, f32) before they are rounded.
 * A stated second variant next to orc_rms_norm (others.rs:11-29 order), like orc_gemm_wdense next to orc_wna16_gemm: the same
 * error size against the unrounded truth (tests/test_gpu_tolerance.py), another rounding pattern. */
void orc_rms_norm_deferred(const void* x, const void* w, int T, int H, float eps, int dt, void* out, float* rstd) {
#pragma omp parallel for
  for (int t = 0; t < T; t++) {
    double ss = 0.0;
    for (int i = 0; i < H; i++) {
      double v = ld(x, (int64_t)t * H + i, dt);
      ss += v * v;
    }
    rstd[t] = 1.0f / sqrtf((float)(ss / H) + eps);
    for (int i = 0; i < H; i++) st(out, (int64_t)t * H + i, ld(x, (int64_t)t * H + i, dt) * ld(w, i, dt), dt);
  }
}
/* candle `+` (llama.rs:126,130) */
void orc_add(const void* a, const void* b, int64_t n, int dt, void* out) {
  for (int64_t i = 0; i < n; i++) st(out, i, ld(a, i, dt) + ld(b, i, dt), dt);
}
/* Activation::Silu then `*` (mlp.rs:468): two ops, two roundings */
void orc_silu_mul(const void* gate, const void* up, int64_t n, int dt, void* out) {
  for (int64_t i = 0; i < n; i++) {
    float g = ld(gate, i, dt);
    float s = rnd(g / (1.0f + expf(-g)), dt);
    st(out, i, s * ld(up, i, dt), dt);
  }
}
/* candle_nn::Embedding (llama.rs:261) */
void orc_embedding(const uint32_t* ids, const void* table, int T, int H, int dt, void* out) {
  size_t es = dt == DT_F32 ? 4 : 2;
  for (int t = 0; t < T; t++) memcpy((char*)out + (size_t)t * H * es, (const char*)table + (size_t)ids[t] * H * es, (size_t)H * es);
}

/* ---------------------------------------------------------------- rotary --------------------- */
/* RotaryEmbedding::new / ScalingRotaryEmbedding::new (src/models/layers/rotary_emb.rs:32-73,
 * 143-278): inv_freq = 1f32 / (theta^(i/d) computed in f64, cast to f32); llama3 smoothing in
 * f32; freqs = pos(f32) * inv_freq (f32); tables are cos/sin of that, returned in f32 (the caller
 * rounds to the model dtype as the reference does, llama.rs:179-189).
 * scaling_type: 0 default, 1 linear (inv_freq / factor), 2 llama3; 3 dynamic and 4 yarn through orc_rope_tables_ext. */
void orc_rope_tables(int rot_dim, double theta, int scaling_type, double factor, double low_freq_factor,
                     double high_freq_factor, double original_max_pos, int n_pos, float* cosv,
                     float* sinv) {
  int half = rot_dim / 2;
  float* inv = (float*)malloc(sizeof(float) * half);
  for (int i = 0; i < half; i++) inv[i] = 1.0f / (float)pow(theta, (double)(2 * i) / (double)rot_dim);
  if (scaling_type == 1) {
    /* `(inv_freq / factor)?` on an f32 candle Tensor is affine(mul = 1/factor): x * (f32)(1/factor) */
    for (int i = 0; i < half; i++) inv[i] = inv[i] * (float)(1.0 / factor);
  } else if (scaling_type == 2) {
    float low_wl = (float)(original_max_pos / low_freq_factor);
    float high_wl = (float)(original_max_pos / high_freq_factor);
    for (int i = 0; i < half; i++) {
      float freq = inv[i];
      float wavelen = 2.0f * 3.14159265358979323846f / freq;
      if (wavelen < high_wl) {
      } else if (wavelen > low_wl) inv[i] = freq / (float)factor;
      else {
        float smooth = ((float)original_max_pos / wavelen - (float)low_freq_factor) / (float)(high_freq_factor - low_freq_factor);
        inv[i] = (1.0f - smooth) * freq / (float)factor + smooth * freq;
      }
    }
  }
  for (int p = 0; p < n_pos; p++)
    for (int i = 0; i < half; i++) {
      float ang = (float)p * inv[i];
      cosv[(int64_t)p * half + i] = cosf(ang);
      sinv[(int64_t)p * half + i] = sinf(ang);
    }
  free(inv);
}
/* "dynamic" (rotary_emb.rs:281-333) and "yarn" (rotary_emb.rs:335-415 + YarnRotaryEmbedding, :435-541).
 * dynamic: the DEFAULT table of a rescaled base, base' = (base * s)^(d/(d-2)) with s = alpha, or for the `factor` form
 *   s = factor * max_len / orig - (factor - 1), max_len = (u32)(orig * factor); all in f64, then the f32 inverse frequencies.
 * yarn: f32 throughout.  freq_extra_i = 1/base^(2i/d), freq_inter_i = 1/(factor * base^(2i/d));
 *   low = max(floor(cd(beta_fast)), 0), high = min(ceil(cd(beta_slow)), d-1), cd(r) = d*ln(orig/(2*pi*r)) / (2*ln(base));
 *   ramp_i = clamp((i - low) * (f32)(1/(high-low)), 0, 1) over i < d/2 (high += 0.001 when equal);
 *   mask = (1 - ramp) * extrapolation_factor;  inv_freq = inter*(1-mask) + extra*mask;
 *   cos/sin of pos*inv_freq times mscale = (factor <= 1 ? 1 : 0.1*ln(factor)+1) * attn_factor. */
void orc_rope_tables_ext(int rot_dim, double theta, int scaling_type, double factor, int dynamic_alpha, double original_max_pos,
                         double beta_fast, double beta_slow, double attn_factor, double extrapolation_factor, int n_pos, float* cosv,
                         float* sinv) {
  int half = rot_dim / 2;
  float* inv = (float*)malloc(sizeof(float) * half);
  float mscale = 1.0f;
  if (scaling_type == 3) {
    double s = factor;
    if (!dynamic_alpha) {
      uint32_t max_len = (uint32_t)(original_max_pos * factor);
      s = (factor * (double)max_len / original_max_pos) - (factor - 1.0);
    }
    double base = pow(theta * s, (double)rot_dim / (double)(rot_dim - 2));
    for (int i = 0; i < half; i++) inv[i] = 1.0f / (float)pow(base, (double)(2 * i) / (double)rot_dim);
  } else {
    float base = (float)theta, fac = (float)factor;
    float two_pi = 2.0f * 3.14159265358979323846f;
    float cd_fast = ((float)rot_dim * logf((float)(size_t)original_max_pos / ((float)beta_fast * two_pi))) / (2.0f * logf(base));
    float cd_slow = ((float)rot_dim * logf((float)(size_t)original_max_pos / ((float)beta_slow * two_pi))) / (2.0f * logf(base));
    float low = floorf(cd_fast), high = ceilf(cd_slow);
    if (low < 0.0f) low = 0.0f;
    if (high > (float)rot_dim - 1.0f) high = (float)rot_dim - 1.0f;
    if (low == high) high += 0.001f;
    float rmul = (float)(1.0 / ((double)high - (double)low));
    for (int i = 0; i < half; i++) {
      float pw = powf(base, (float)(2 * i) / (float)rot_dim);
      float extra = 1.0f / pw, inter = 1.0f / (fac * pw);
      float ramp = ((float)i - low) * rmul;
      ramp = ramp < 0.0f ? 0.0f : (ramp > 1.0f ? 1.0f : ramp);
      float mask = (1.0f - ramp) * (float)extrapolation_factor;
      inv[i] = inter * (1.0f - mask) + extra * mask;
    }
    mscale = (fac <= 1.0f ? 1.0f : 0.1f * logf(fac) + 1.0f) * (float)attn_factor;
  }
  for (int p = 0; p < n_pos; p++)
    for (int i = 0; i < half; i++) {
      float ang = (float)p * inv[i];
      cosv[(int64_t)p * half + i] = scaling_type == 4 ? cosf(ang) * mscale : cosf(ang);
      sinv[(int64_t)p * half + i] = scaling_type == 4 ? sinf(ang) * mscale : sinf(ang);
    }
  free(inv);
}
/* FusedRope::apply_inplace (rotary_emb.rs:103; kernel body NOT IN TREE).  Definition used here:
 * rows gathered by positions; NeoX (is_interleaved=0): pairs (i, i+rot/2); interleaved: (2i,2i+1);
 * x' = x1*c - x2*s, y' = x2*c + x1*s in f32 from dt inputs/table (table_dt), ONE rounding.
 * x is [T, heads, D], in place. */
void orc_rope(void* x, int T, int heads, int D, int rot_dim, const void* cosv, const void* sinv,
              const int64_t* positions, int is_interleaved, int dt, int table_dt) {
  int half = rot_dim / 2;
  for (int t = 0; t < T; t++)
    for (int h = 0; h < heads; h++) {
      int64_t base = ((int64_t)t * heads + h) * D;
      for (int i = 0; i < half; i++) {
        int i1 = is_interleaved ? 2 * i : i, i2 = is_interleaved ? 2 * i + 1 : i + half;
        float c = ld(cosv, positions[t] * half + i, table_dt), s = ld(sinv, positions[t] * half + i, table_dt);
        float x1 = ld(x, base + i1, dt), x2 = ld(x, base + i2, dt);
        st(x, base + i1, x1 * c - x2 * s, dt);
        st(x, base + i2, x2 * c + x1 * s, dt);
      }
    }
}

/* ---------------------------------------------------------------- paged KV + attention ------- */
/* Cache geometry used by the build (same element count as kvcache_allocator.rs:851-863):
 *   K [num_blocks, kv_heads, block_size, head_dim]   (token rows contiguous)
 *   V [num_blocks, kv_heads, head_dim, block_size]   (token-minor, as the reference's non-flash V
 *                                                     cache, kvcache_allocator.rs:170-173,844) */
static inline int64_t cache_off(int64_t slot, int h, int Hkv, int BS, int D) {
  int64_t blk = slot / BS, off = slot % BS;
  return ((blk * Hkv + h) * BS + off) * D;
}
static inline int64_t vcache_off(int64_t slot, int h, int d, int Hkv, int BS, int D) {
  int64_t blk = slot / BS, off = slot % BS;
  return ((blk * Hkv + h) * D + d) * BS + off;
}
/* reshape_and_cache half of PagedAttention::forward (attention.rs:808-820), slots from
 * ModelRunner::prepare_* (runner.rs:1020-1038,1259-1262). Negative slot = skip (Appendix A6). */
void orc_reshape_and_cache_kv(const void* k, const void* v, void* kc, void* vc, const int64_t* slots,
                              int T, int Hkv, int D, int BS, int dt, int kv_dt) {
  for (int t = 0; t < T; t++) {
    if (slots[t] < 0) continue;
    for (int h = 0; h < Hkv; h++)
      for (int d = 0; d < D; d++) {
        int64_t src = ((int64_t)t * Hkv + h) * D + d;
        st(kc, cache_off(slots[t], h, Hkv, BS, D) + d, ld(k, src, dt), kv_dt);  /* 16-bit -> 16-bit is the identity */
        st(vc, vcache_off(slots[t], h, d, Hkv, BS, D), ld(v, src, dt), kv_dt);
      }
  }
}
void orc_reshape_and_cache(const void* k, const void* v, void* kc, void* vc, const int64_t* slots,
                           int T, int Hkv, int D, int BS, int dt) {
  orc_reshape_and_cache_kv(k, v, kc, vc, slots, T, Hkv, D, BS, dt, dt);
}
/* Attention over the paged cache, covering both halves of PagedAttention::forward:
 *   decode  (runner.rs:1243-1388): cu_q == NULL, one query per sequence at position ctx-1;
 *   prefill (runner.rs:978-1241):  query i of sequence b at position ctx_b - len_q_b + i, causal.
 * scores = (q·k)*scale in f32-equivalent (double acc), optional softcap tanh, softmax, P·V, one
 * rounding of the output.  out/q [Tq,Hq,D].  block_tables [B,max_blocks] zero padded (A5). */
void orc_paged_attention_kv(void* out, const void* q, const void* kc, const void* vc,
                            const uint32_t* block_tables, const uint32_t* context_lens,
                            const uint32_t* cu_q, int B, int Hq, int Hkv, int D, int BS, int max_blocks,
                            float scale, float softcap, int dt, int kv_dt);
void orc_paged_attention(void* out, const void* q, const void* kc, const void* vc,
                         const uint32_t* block_tables, const uint32_t* context_lens,
                         const uint32_t* cu_q, int B, int Hq, int Hkv, int D, int BS, int max_blocks,
                         float scale, float softcap, int dt) {
  orc_paged_attention_kv(out, q, kc, vc, block_tables, context_lens, cu_q, B, Hq, Hkv, D, BS, max_blocks, scale, softcap, dt, dt);
}
void orc_paged_attention_kv_sw(void* out, const void* q, const void* kc, const void* vc,
                               const uint32_t* block_tables, const uint32_t* context_lens,
                               const uint32_t* cu_q, int B, int Hq, int Hkv, int D, int BS, int max_blocks,
                               float scale, float softcap, int sliding_window, int dt, int kv_dt);
void orc_paged_attention_kv(void* out, const void* q, const void* kc, const void* vc,
                            const uint32_t* block_tables, const uint32_t* context_lens,
                            const uint32_t* cu_q, int B, int Hq, int Hkv, int D, int BS, int max_blocks,
                            float scale, float softcap, int dt, int kv_dt) {
  orc_paged_attention_kv_sw(out, q, kc, vc, block_tables, context_lens, cu_q, B, Hq, Hkv, D, BS, max_blocks, scale, softcap, 0, dt, kv_dt);
}
/* sliding_window > 0 (PagedAttention::new, attention.rs:607-616): the query at position pos attends the keys
 * max(0, pos - W + 1) .. pos (attention_rs::mask::causal_mask: j <= i && i - j < W, orc_causal_mask above); 0 = off */
void orc_paged_attention_kv_sw(void* out, const void* q, const void* kc, const void* vc,
                               const uint32_t* block_tables, const uint32_t* context_lens,
                               const uint32_t* cu_q, int B, int Hq, int Hkv, int D, int BS, int max_blocks,
                               float scale, float softcap, int sliding_window, int dt, int kv_dt) {
  /* Same arithmetic as the plain loops (scores, softmax and P.V in double, keys in ascending order); the keys and values
   * of one (sequence, kv head) are widened to double once and the (query head, query row) pairs run in parallel, so that a
   * 32k-token context (BASELINE config 5) is a matter of seconds on the host cores. */
  int G = Hq / Hkv;
  for (int b = 0; b < B; b++) {
    int ctx = (int)context_lens[b];
    int q0 = cu_q ? (int)cu_q[b] : b, lq = cu_q ? (int)(cu_q[b + 1] - cu_q[b]) : 1;
    if (ctx <= 0 || lq <= 0) continue;
    for (int hk = 0; hk < Hkv; hk++) {
      double* Kd = (double*)malloc(sizeof(double) * (size_t)ctx * D);
      double* Vd = (double*)malloc(sizeof(double) * (size_t)ctx * D);
#pragma omp parallel for schedule(static)
      for (int j = 0; j < ctx; j++) {
        int64_t slot = (int64_t)block_tables[(int64_t)b * max_blocks + j / BS] * BS + j % BS;
        int64_t o = cache_off(slot, hk, Hkv, BS, D);
        for (int d = 0; d < D; d++) {
          Kd[(size_t)j * D + d] = (double)ld(kc, o + d, kv_dt);
          Vd[(size_t)j * D + d] = (double)ld(vc, vcache_off(slot, hk, d, Hkv, BS, D), kv_dt);
        }
      }
#pragma omp parallel
      {
        double* sc = (double*)malloc(sizeof(double) * (size_t)ctx);
        double* acc = (double*)malloc(sizeof(double) * D);
        double* qd = (double*)malloc(sizeof(double) * D);
#pragma omp for collapse(2) schedule(dynamic, 4)
        for (int gi = 0; gi < G; gi++)
          for (int i = 0; i < lq; i++) {
            int h = hk * G + gi;
            int pos = ctx - lq + i; /* attends keys j0..pos (j0 = 0 without a sliding window) */
            int j0 = sliding_window > 0 && pos - sliding_window + 1 > 0 ? pos - sliding_window + 1 : 0;
            int64_t qb = ((int64_t)(q0 + i) * Hq + h) * D;
            for (int d = 0; d < D; d++) qd[d] = (double)ld(q, qb + d, dt);
            double mx = -1e300;
            for (int j = j0; j <= pos; j++) {
              const double* kr = Kd + (size_t)j * D;
              double sv = 0.0;
              for (int d = 0; d < D; d++) sv += qd[d] * kr[d];
              sv *= scale;
              if (softcap > 0.0f) sv = softcap * tanh(sv / softcap);
              sc[j] = sv;
              if (sv > mx) mx = sv;
            }
            double den = 0.0;
            for (int j = j0; j <= pos; j++) {
              sc[j] = exp(sc[j] - mx);
              den += sc[j];
            }
            for (int d = 0; d < D; d++) acc[d] = 0.0;
            for (int j = j0; j <= pos; j++) {
              const double* vr = Vd + (size_t)j * D;
              const double pj = sc[j];
              for (int d = 0; d < D; d++) acc[d] += pj * vr[d];
            }
            for (int d = 0; d < D; d++) st(out, qb + d, (float)(acc[d] / den), dt);
          }
        free(sc);
        free(acc);
        free(qd);
      }
      free(Kd);
      free(Vd);
    }
  }
}
/* Prefill without cache indirection: k,v [Tk,Hkv,D] with cu_k (no prefix: cu_k == cu_q). */
void orc_varlen_attention(void* out, const void* q, const void* k, const void* v, const uint32_t* cu_q,
                          const uint32_t* cu_k, int B, int Hq, int Hkv, int D, float scale,
                          float softcap, int dt) {
  int G = Hq / Hkv;
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (int b = 0; b < B; b++)
    for (int h = 0; h < Hq; h++) {
      int lq = (int)(cu_q[b + 1] - cu_q[b]), lk = (int)(cu_k[b + 1] - cu_k[b]);
      int hk = h / G;
      double* sc = (double*)malloc(sizeof(double) * (lk > 0 ? lk : 1));
      double* acc = (double*)malloc(sizeof(double) * D);
      for (int i = 0; i < lq; i++) {
        int pos = lk - lq + i;
        int64_t qb = ((int64_t)(cu_q[b] + i) * Hq + h) * D;
        double mx = -1e300;
        for (int j = 0; j <= pos; j++) {
          int64_t o = ((int64_t)(cu_k[b] + j) * Hkv + hk) * D;
          double s = 0.0;
          for (int d = 0; d < D; d++) s += (double)ld(q, qb + d, dt) * (double)ld(k, o + d, dt);
          s *= scale;
          if (softcap > 0.0f) s = softcap * tanh(s / softcap);
          sc[j] = s;
          if (s > mx) mx = s;
        }
        double den = 0.0;
        for (int j = 0; j <= pos; j++) {
          sc[j] = exp(sc[j] - mx);
          den += sc[j];
        }
        for (int d = 0; d < D; d++) acc[d] = 0.0;
        for (int j = 0; j <= pos; j++) {
          int64_t o = ((int64_t)(cu_k[b] + j) * Hkv + hk) * D;
          for (int d = 0; d < D; d++) acc[d] += sc[j] * (double)ld(v, o + d, dt);
        }
        for (int d = 0; d < D; d++) st(out, qb + d, (float)(acc[d] / den), dt);
      }
      free(sc);
      free(acc);
    }
}
/* attention_rs::mask::causal_mask (src/models/layers/mask.rs:18-54): additive [L,L] */
void orc_causal_mask(void* mask, int L, int sliding_window, int dt) {
  for (int i = 0; i < L; i++)
    for (int j = 0; j < L; j++) {
      int ok = j <= i && (sliding_window <= 0 || i - j < sliding_window);
      st(mask, (int64_t)i * L + j, ok ? 0.0f : -INFINITY, dt);
    }
}
/* logits.argmax(-1) (src/utils/logits_processor.rs:67-70): first maximal index */
void orc_argmax_f32(const float* logits, int rows, int cols, uint32_t* out) {
  for (int r = 0; r < rows; r++) {
    const float* p = logits + (int64_t)r * cols;
    int best = 0;
    for (int c = 1; c < cols; c++)
      if (p[c] > p[best]) best = c;
    out[r] = (uint32_t)best;
  }
}
