// ops.hip — RMSNorm, residual add, embedding, SiLU·mul, fused rotary, KV scatter, mask, argmax,
// casts, block swap and the synthetic fills.  All HBM-bound streaming kernels: 16 B per lane,
// f32 math, one rounding per reference op (include/vllm_rs_amd.h §B).
#include "common.cuh"
#include "kvcache.cuh"

// ---------------------------------------------------------------- RMSNorm (+ residual add)
// one workgroup (256 threads) per token row; row cached in registers between the two passes.
// XS: also write Σ of the ROUNDED outputs per 128-column k-tile to xsum[row][H/128] — the row-sum table kernel D (gemm_q4_big.cuh)
// needs, in exactly the order `xsum_rows_kernel` uses (octet sums, then the 16-lane DPP tree), so a GEMM fed with this table is
// bit-identical to one that ran the separate pass; H % 128 == 0
template <class DT, bool ADD, bool XS = false>
__global__ __launch_bounds__(256) void rms_norm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ res,
                                                       const uint16_t* __restrict__ w, uint16_t* __restrict__ h_out,
                                                       uint16_t* __restrict__ out, int H, float eps, float* __restrict__ xsum = nullptr) {
  __shared__ float red[4];
  const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int octs = H >> 3;
  const u32x4* xr = reinterpret_cast<const u32x4*>(x + (size_t)t * H);
  const u32x4* rr = ADD ? reinterpret_cast<const u32x4*>(res + (size_t)t * H) : nullptr;
  constexpr int MAXV = 8;  // up to H = 256*8*8 = 16384 kept in registers
  u32x4 keep[MAXV];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; i++) {
    int o = tid + i * 256;
    if (o < octs) {
      u32x4 v = xr[o];
      float f[8];
      unpack8<DT>(v, f);
      if (ADD) {
        float r[8];
        unpack8<DT>(rr[o], r);
#pragma unroll
        for (int e = 0; e < 8; e++) f[e] = rnd_dt<DT>(f[e] + r[e]);
        v = pack8<DT>(f);
        reinterpret_cast<u32x4*>(h_out + (size_t)t * H)[o] = v;
      }
#pragma unroll
      for (int e = 0; e < 8; e++) ss += f[e] * f[e];
      keep[i] = v;
    }
  }
  ss = wave_sum(ss);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
  const u32x4* wr = reinterpret_cast<const u32x4*>(w);
#pragma unroll
  for (int i = 0; i < MAXV; i++) {
    int o = tid + i * 256;
    if (o < octs) {
      float f[8], g[8];
      unpack8<DT>(keep[i], f);
      unpack8<DT>(wr[o], g);
#pragma unroll
      for (int e = 0; e < 8; e++) f[e] = f[e] * rstd * g[e];
      const u32x4 pv = pack8<DT>(f);
      reinterpret_cast<u32x4*>(out + (size_t)t * H)[o] = pv;
      if (XS) {  // (octs % 16 == 0: the 16 lanes of a k-tile are all in or all out of this branch)
        const float s16 = row16_sum(octet_sum<DT>(pv));
        if ((o & 15) == 0) xsum[(size_t)t * (H >> 7) + (o >> 4)] = s16;
      }
    }
  }
}
static bool norm_args_ok(const char* who, int tokens, int hidden, int dtype) {
  if (tokens < 0 || hidden < 8 || hidden % 8 || hidden > 16384) {
    vra_set_error("%s: hidden must be a multiple of 8 in [8,16384] (tokens=%d hidden=%d)", who, tokens, hidden);
    return false;
  }
  if (dtype != VRA_BF16 && dtype != VRA_F16) {
    vra_set_error("%s: dtype must be bf16/f16", who);
    return false;
  }
  return true;
}
extern "C" void vra_rms_norm(const void* x, const void* weight, void* out, int32_t tokens, int32_t hidden, float eps,
                             int32_t dtype, int64_t stream) {
  if (!norm_args_ok("vra_rms_norm", tokens, hidden, dtype) || tokens == 0) return;
  if (dtype == VRA_BF16)
    rms_norm_kernel<BF16, false><<<tokens, 256, 0, as_stream(stream)>>>((const uint16_t*)x, nullptr, (const uint16_t*)weight, nullptr, (uint16_t*)out, hidden, eps);
  else
    rms_norm_kernel<F16, false><<<tokens, 256, 0, as_stream(stream)>>>((const uint16_t*)x, nullptr, (const uint16_t*)weight, nullptr, (uint16_t*)out, hidden, eps);
}
// internal (gemm_launch.h): RMSNorm + the row-sum table of kernel D in one launch
void vra_rms_norm_xsum(const void* x, const void* weight, void* out, float* xsum, int tokens, int hidden, float eps, int dtype, int64_t stream) {
  if (!norm_args_ok("vra_rms_norm_xsum", tokens, hidden, dtype) || tokens == 0) return;
  if (hidden % 128) {
    vra_set_error("vra_rms_norm_xsum: hidden %d is not a multiple of 128", hidden);
    return;
  }
  if (dtype == VRA_BF16)
    rms_norm_kernel<BF16, false, true><<<tokens, 256, 0, as_stream(stream)>>>((const uint16_t*)x, nullptr, (const uint16_t*)weight, nullptr, (uint16_t*)out, hidden, eps, xsum);
  else
    rms_norm_kernel<F16, false, true><<<tokens, 256, 0, as_stream(stream)>>>((const uint16_t*)x, nullptr, (const uint16_t*)weight, nullptr, (uint16_t*)out, hidden, eps, xsum);
}
extern "C" void vra_add_rms_norm(const void* x, const void* residual, const void* weight, void* h_out, void* out,
                                 int32_t tokens, int32_t hidden, float eps, int32_t dtype, int64_t stream) {
  if (!norm_args_ok("vra_add_rms_norm", tokens, hidden, dtype) || tokens == 0) return;
  if (dtype == VRA_BF16)
    rms_norm_kernel<BF16, true><<<tokens, 256, 0, as_stream(stream)>>>((const uint16_t*)x, (const uint16_t*)residual, (const uint16_t*)weight, (uint16_t*)h_out, (uint16_t*)out, hidden, eps);
  else
    rms_norm_kernel<F16, true><<<tokens, 256, 0, as_stream(stream)>>>((const uint16_t*)x, (const uint16_t*)residual, (const uint16_t*)weight, (uint16_t*)h_out, (uint16_t*)out, hidden, eps);
}

// ---------------------------------------------------------------- q/k-norm (Attention::forward_ext, attention.rs:713-735)
// Qwen3-style checkpoints carry `q_norm` / `k_norm`: an RMSNorm over every head's head_dim channels of q and k (weight [head_dim],
// `q.flatten(0, 1)` -> NormX::forward, attention.rs:724-731) — or, when the weight has num_heads * head_dim entries, over the whole
// q / k row (`full_dim_qk_norm`, attention.rs:714-722, the weight sharded with the heads: attention.rs:567-590) — applied BEFORE the
// rotary embedding, in place.  Same arithmetic as rms_norm_kernel: f32 sum of squares, rstd = 1 / sqrt(mean + eps), x * rstd * g,
// one rounding.  Per-head form: 16 lanes per (token, head) row (head_dim <= 128: 8 channels per lane), sums by DPP, q and k rows
// in ONE launch; the full-dim form is two rows of rms_norm_kernel per token.
template <class DT>
__global__ __launch_bounds__(256) void qk_head_norm_kernel(uint16_t* __restrict__ q, uint16_t* __restrict__ k, const uint16_t* __restrict__ qw,
                                                           const uint16_t* __restrict__ kw, int64_t q_rows, int64_t rows, int D, float eps) {
  const int tid = threadIdx.x, l16 = tid & 15;
  const int64_t row = (int64_t)blockIdx.x * 16 + (tid >> 4);
  const bool act = row < rows && l16 * 8 < D;
  const bool is_q = row < q_rows;
  uint16_t* xp = (is_q ? q + row * D : k + (row - q_rows) * D) + l16 * 8;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (act) v = *reinterpret_cast<const u32x4*>(xp);
  float f[8], g[8];
  unpack8<DT>(v, f);
  float ss = 0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) ss += f[e] * f[e];
  ss = row16_sum(ss);
  const float rstd = 1.0f / sqrtf(ss / (float)D + eps);
  if (act) {
    unpack8<DT>(*reinterpret_cast<const u32x4*>((is_q ? qw : kw) + l16 * 8), g);
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = f[e] * rstd * g[e];
    *reinterpret_cast<u32x4*>(xp) = pack8<DT>(f);
  }
}
extern "C" void vra_qk_rms_norm(void* q, void* k, const void* q_weight, const void* k_weight, int32_t tokens, int32_t q_heads, int32_t kv_heads,
                                int32_t head_dim, int32_t full_dim, float eps, int32_t dtype, int64_t stream) {
  VRA_CHECK_ARG(q && k && q_weight && k_weight, "vra_qk_rms_norm: null pointer");
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_qk_rms_norm: dtype must be bf16/f16");
  VRA_CHECK_ARG(tokens >= 0 && q_heads > 0 && kv_heads > 0, "vra_qk_rms_norm: bad shape");
  if (tokens == 0) return;
  if (full_dim) {  // weight [heads * head_dim]: one RMSNorm over the token's whole q (k) row
    vra_rms_norm(q, q_weight, q, tokens, q_heads * head_dim, eps, dtype, stream);
    vra_rms_norm(k, k_weight, k, tokens, kv_heads * head_dim, eps, dtype, stream);
    return;
  }
  VRA_CHECK_ARG(head_dim % 8 == 0 && head_dim >= 8 && head_dim <= 128, "vra_qk_rms_norm: per-head form needs head_dim %% 8 == 0, <= 128 (got %d)", head_dim);
  const int64_t q_rows = (int64_t)tokens * q_heads, rows = q_rows + (int64_t)tokens * kv_heads;
  const unsigned grid = (unsigned)((rows + 15) / 16);
  if (dtype == VRA_BF16)
    qk_head_norm_kernel<BF16><<<grid, 256, 0, as_stream(stream)>>>((uint16_t*)q, (uint16_t*)k, (const uint16_t*)q_weight, (const uint16_t*)k_weight, q_rows, rows, head_dim, eps);
  else
    qk_head_norm_kernel<F16><<<grid, 256, 0, as_stream(stream)>>>((uint16_t*)q, (uint16_t*)k, (const uint16_t*)q_weight, (const uint16_t*)k_weight, q_rows, rows, head_dim, eps);
}

// ---------------------------------------------------------------- elementwise
template <class DT, int OP>  // OP 0: add, 1: silu(a)*b
__global__ __launch_bounds__(256) void ew_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                 uint16_t* __restrict__ out, int64_t numel) {
  const int64_t nv = numel >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
    float fa[8], fb[8];
    unpack8<DT>(reinterpret_cast<const u32x4*>(a)[i], fa);
    unpack8<DT>(reinterpret_cast<const u32x4*>(b)[i], fb);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if (OP == 0) fa[e] = fa[e] + fb[e];
      else fa[e] = rnd_dt<DT>(fa[e] / (1.0f + expf(-fa[e]))) * fb[e];
    }
    reinterpret_cast<u32x4*>(out)[i] = pack8<DT>(fa);
  }
  // tail (numel % 8)
  for (int64_t i = (nv << 3) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
    float x = DT::to_f32(a[i]), y = DT::to_f32(b[i]);
    out[i] = DT::from_f32(OP == 0 ? x + y : rnd_dt<DT>(x / (1.0f + expf(-x))) * y);
  }
}
static inline int ew_grid(int64_t numel) {
  int64_t g = ((numel >> 3) + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
extern "C" void vra_add(const void* a, const void* b, void* out, int64_t numel, int32_t dtype, int64_t stream) {
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_add: dtype must be bf16/f16");
  if (numel <= 0) return;
  if (dtype == VRA_BF16) ew_kernel<BF16, 0><<<ew_grid(numel), 256, 0, as_stream(stream)>>>((const uint16_t*)a, (const uint16_t*)b, (uint16_t*)out, numel);
  else ew_kernel<F16, 0><<<ew_grid(numel), 256, 0, as_stream(stream)>>>((const uint16_t*)a, (const uint16_t*)b, (uint16_t*)out, numel);
}
extern "C" void vra_silu_mul(const void* gate, const void* up, void* out, int64_t numel, int32_t dtype, int64_t stream) {
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_silu_mul: dtype must be bf16/f16");
  if (numel <= 0) return;
  if (dtype == VRA_BF16) ew_kernel<BF16, 1><<<ew_grid(numel), 256, 0, as_stream(stream)>>>((const uint16_t*)gate, (const uint16_t*)up, (uint16_t*)out, numel);
  else ew_kernel<F16, 1><<<ew_grid(numel), 256, 0, as_stream(stream)>>>((const uint16_t*)gate, (const uint16_t*)up, (uint16_t*)out, numel);
}

// ---------------------------------------------------------------- row gathers
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint32_t* __restrict__ idx, const unsigned char* __restrict__ table,
                                                          unsigned char* __restrict__ out, int rows, size_t row_bytes,
                                                          uint32_t n_table_rows, uint32_t* __restrict__ bump = nullptr,
                                                          u32x4* __restrict__ frag = nullptr) {
  const int r = blockIdx.x;
  if (bump && r == 0 && threadIdx.x == 0) *bump += 1u;  // the forward's epoch word (qkv_attn.h): one writer, read by LATER launches
  uint32_t src = idx[r];
  if (src >= n_table_rows) src = n_table_rows - 1;  // clamp (the reference would fault)
  const u32x4* s = reinterpret_cast<const u32x4*>(table + (size_t)src * row_bytes);
  u32x4* d = reinterpret_cast<u32x4*>(out + (size_t)r * row_bytes);
  for (size_t i = threadIdx.x; i < row_bytes / 16; i += blockDim.x) {
    const u32x4 v = s[i];
    d[i] = v;
    // rows 0..31 also in kernel W's fragment order (GemvSArgs::x_frag; 16-bit rows, hidden % 128 == 0: 16-byte chunk i = columns
    // i*8 .. +7 = k-tile i >> 4, k-step (i >> 2) & 3, octet i & 3)
    if (frag && r < 32) frag[(((i >> 4) * 2 + (size_t)(r >> 4)) * 4 + ((i >> 2) & 3)) * 64 + (i & 3) * 16 + (r & 15)] = v;
  }
}
// the embedding launch of a 5..32-row step as a PRODUCER of ready-made operands (gemv_q4s.cuh GemvSArgs::pre_*): next to the rows
// and their fragment-order copy, x̃ = round(row * g) for layer 0's attention norm in fragment order and the rows' sums of squares
// (slot 0 of the partial-sum table; the other slots of that table are never written and stay zero) — layer 0's q/k/v launch then
// normalises in the same (deferred) order as every later layer's and as kernel E's
template <class DT>
__global__ __launch_bounds__(256) void embed_rows_pre_kernel(const uint32_t* __restrict__ idx, const uint16_t* __restrict__ table,
                                                              uint16_t* __restrict__ out, int hidden, uint32_t n_table_rows,
                                                              uint32_t* __restrict__ bump, u32x4* __restrict__ frag,
                                                              const uint16_t* __restrict__ norm_w, u32x4* __restrict__ pre_frag,
                                                              float* __restrict__ pre_sq) {
  __shared__ float red[4];
  const int r = blockIdx.x;  // < 32
  if (bump && r == 0 && threadIdx.x == 0) *bump += 1u;
  uint32_t src = idx[r];
  if (src >= n_table_rows) src = n_table_rows - 1;
  const u32x4* s = reinterpret_cast<const u32x4*>(table + (size_t)src * hidden);
  u32x4* d = reinterpret_cast<u32x4*>(out + (size_t)r * hidden);
  float ss = 0.f;
  for (int i = threadIdx.x; i < hidden / 8; i += 256) {
    const u32x4 v = s[i];
    d[i] = v;
    const size_t fi = (((size_t)(i >> 4) * 2 + (size_t)(r >> 4)) * 4 + ((i >> 2) & 3)) * 64 + (i & 3) * 16 + (r & 15);
    frag[fi] = v;
    float f[8], g[8];
    unpack8<DT>(v, f);
    unpack8<DT>(reinterpret_cast<const u32x4*>(norm_w)[i], g);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      ss = fmaf(f[e], f[e], ss);
      f[e] *= g[e];
    }
    pre_frag[fi] = pack8<DT>(f);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) pre_sq[r] = (red[0] + red[1]) + (red[2] + red[3]);
}
extern "C" void vra_embedding(const uint32_t* ids, const void* table, void* out, int32_t tokens, int32_t hidden,
                              int32_t vocab, int32_t dtype, int64_t stream) {
  size_t es = dtype == VRA_F32 ? 4 : 2;
  VRA_CHECK_ARG((hidden * es) % 16 == 0, "vra_embedding: row bytes must be a multiple of 16");
  if (tokens <= 0) return;
  gather_rows_kernel<<<tokens, 256, 0, as_stream(stream)>>>(ids, (const unsigned char*)table, (unsigned char*)out, tokens, hidden * es, (uint32_t)vocab);
}
// dense [N, K] 16-bit row-major -> tile-major (gemv.cuh GemvArgs::dense_tiled): one 16-byte word per thread
__global__ void dense_tile_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int N, int K) {
  const int KT = K >> 7;
  const int64_t total = (int64_t)(N >> 4) * KT * 256;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), l = (int)((i >> 6) & 3);
    const int64_t t = i >> 8;
    const int kt = (int)(t % KT), nb = (int)(t / KT);
    const int nn = lane & 15, oct = lane >> 4;
    out[i] = in[((int64_t)(nb * 16 + nn) * K + kt * 128 + l * 32 + oct * 8) >> 3];
  }
}
void vra_dense_tile_weights(const void* w_rowmajor, void* out_tiled, int32_t n, int32_t k, int64_t stream) {
  VRA_CHECK_ARG(n % 16 == 0 && k % 128 == 0, "vra_dense_tile_weights: need n %% 16 == 0 and k %% 128 == 0");
  const int64_t total = (int64_t)n * k / 8;
  int grid = (int)((total + 255) / 256);
  if (grid > 16384) grid = 16384;
  dense_tile_kernel<<<grid, 256, 0, as_stream(stream)>>>((const u32x4*)w_rowmajor, (u32x4*)out_tiled, n, k);
}
void vra_embedding_bump(const uint32_t* ids, const void* table, void* out, int32_t tokens, int32_t hidden, int32_t vocab, int32_t dtype,
                        uint32_t* bump, void* frag, const void* pre_norm_w, void* pre_frag, float* pre_sq, int64_t stream) {
  size_t es = dtype == VRA_F32 ? 4 : 2;
  VRA_CHECK_ARG((hidden * es) % 16 == 0, "vra_embedding: row bytes must be a multiple of 16");
  if (tokens <= 0) return;
  if (es != 2 || hidden % 128) frag = nullptr;
  if (frag && pre_norm_w && pre_frag && pre_sq && tokens <= 32) {
    if (dtype == VRA_BF16)
      embed_rows_pre_kernel<BF16><<<tokens, 256, 0, as_stream(stream)>>>(ids, (const uint16_t*)table, (uint16_t*)out, hidden, (uint32_t)vocab, bump, (u32x4*)frag,
                                                                         (const uint16_t*)pre_norm_w, (u32x4*)pre_frag, pre_sq);
    else
      embed_rows_pre_kernel<F16><<<tokens, 256, 0, as_stream(stream)>>>(ids, (const uint16_t*)table, (uint16_t*)out, hidden, (uint32_t)vocab, bump, (u32x4*)frag,
                                                                        (const uint16_t*)pre_norm_w, (u32x4*)pre_frag, pre_sq);
    return;
  }
  gather_rows_kernel<<<tokens, 256, 0, as_stream(stream)>>>(ids, (const unsigned char*)table, (unsigned char*)out, tokens, hidden * es, (uint32_t)vocab, bump,
                                                            (u32x4*)frag);
}
extern "C" void vra_index_select_rows(const void* x, const uint32_t* idx, void* out, int32_t n_idx, int32_t hidden,
                                      int32_t dtype, int64_t stream) {
  size_t es = dtype == VRA_F32 ? 4 : 2;
  VRA_CHECK_ARG((hidden * es) % 16 == 0, "vra_index_select_rows: row bytes must be a multiple of 16");
  if (n_idx <= 0) return;
  gather_rows_kernel<<<n_idx, 256, 0, as_stream(stream)>>>(idx, (const unsigned char*)x, (unsigned char*)out, n_idx, hidden * es, 0xffffffffu);
}

// ---------------------------------------------------------------- fused rotary (in place)
// one thread per (token, head, pair-octet): NeoX pairs (i, i+rot/2) are handled 8 pairs at a time
// (two 16 B loads), interleaved pairs (2i, 2i+1) 4 pairs per 16 B.
template <class DT, class TT>
__global__ __launch_bounds__(256) void rope_kernel(uint16_t* __restrict__ q, uint16_t* __restrict__ k,
                                                   const void* __restrict__ cosv, const void* __restrict__ sinv,
                                                   const int64_t* __restrict__ positions, int T, int Hq, int Hkv, int D,
                                                   int rot, int interleaved) {
  const int half = rot >> 1;
  const int per_head = interleaved ? rot >> 3 : half >> 3;  // work items (16 B groups) per head
  const int64_t total = (int64_t)T * (Hq + Hkv) * per_head;
  for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < total; w += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(w % per_head);
    int64_t th = w / per_head;
    int h = (int)(th % (Hq + Hkv));
    int t = (int)(th / (Hq + Hkv));
    uint16_t* base = h < Hq ? q + ((size_t)t * Hq + h) * D : k + ((size_t)t * Hkv + (h - Hq)) * D;
    const int64_t pos = positions[t];
    if (!interleaved) {
      u32x4 a = *reinterpret_cast<u32x4*>(base + c * 8), b = *reinterpret_cast<u32x4*>(base + half + c * 8);
      float x1[8], x2[8], cs[8], sn[8];
      unpack8<DT>(a, x1);
      unpack8<DT>(b, x2);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        cs[e] = TT::load(cosv, pos * half + c * 8 + e);
        sn[e] = TT::load(sinv, pos * half + c * 8 + e);
      }
      float y1[8], y2[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        y1[e] = x1[e] * cs[e] - x2[e] * sn[e];
        y2[e] = x2[e] * cs[e] + x1[e] * sn[e];
      }
      *reinterpret_cast<u32x4*>(base + c * 8) = pack8<DT>(y1);
      *reinterpret_cast<u32x4*>(base + half + c * 8) = pack8<DT>(y2);
    } else {
      u32x4 a = *reinterpret_cast<u32x4*>(base + c * 8);
      float x[8], y[8];
      unpack8<DT>(a, x);
#pragma unroll
      for (int p = 0; p < 4; p++) {
        float cs = TT::load(cosv, pos * half + c * 4 + p), sn = TT::load(sinv, pos * half + c * 4 + p);
        y[2 * p] = x[2 * p] * cs - x[2 * p + 1] * sn;
        y[2 * p + 1] = x[2 * p + 1] * cs + x[2 * p] * sn;
      }
      *reinterpret_cast<u32x4*>(base + c * 8) = pack8<DT>(y);
    }
  }
}
template <class DT>
struct TblSame {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return DT::to_f32(static_cast<const uint16_t*>(p)[i]); }
};
struct TblF32 {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return static_cast<const float*>(p)[i]; }
};
extern "C" void vra_fused_rope(void* q, void* k, const void* cos, const void* sin, const int64_t* positions, int32_t tokens,
                               int32_t q_heads, int32_t kv_heads, int32_t head_dim, int32_t rot_dim, int32_t is_interleaved,
                               int32_t dtype, int32_t table_dtype, int64_t stream) {
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_fused_rope: dtype must be bf16/f16");
  VRA_CHECK_ARG(table_dtype == dtype || table_dtype == VRA_F32, "vra_fused_rope: table dtype must equal dtype or be f32");
  VRA_CHECK_ARG(rot_dim <= head_dim && rot_dim % 16 == 0 && head_dim % 8 == 0, "vra_fused_rope: rot_dim must be a multiple of 16 <= head_dim");
  if (tokens <= 0) return;
  int per_head = is_interleaved ? rot_dim / 8 : rot_dim / 16;
  int64_t total = (int64_t)tokens * (q_heads + kv_heads) * per_head;
  int grid = (int)((total + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipStream_t st = as_stream(stream);
  uint16_t *qp = (uint16_t*)q, *kp = (uint16_t*)k;
#define VRA_ROPE(DT, TT) rope_kernel<DT, TT><<<grid, 256, 0, st>>>(qp, kp, cos, sin, positions, tokens, q_heads, kv_heads, head_dim, rot_dim, is_interleaved)
  if (dtype == VRA_BF16) {
    if (table_dtype == VRA_F32) VRA_ROPE(BF16, TblF32);
    else VRA_ROPE(BF16, TblSame<BF16>);
  } else {
    if (table_dtype == VRA_F32) VRA_ROPE(F16, TblF32);
    else VRA_ROPE(F16, TblSame<F16>);
  }
#undef VRA_ROPE
}

// ---------------------------------------------------------------- KV scatter ("reshape_and_cache")
// K cache [NB, Hkv, BS, D] (token rows contiguous), V cache [NB, Hkv, D, BS] (token-minor, so the
// attention kernel reads 4/8 consecutive tokens of one channel with one load — the same choice the
// reference makes for its non-flash V cache, kvcache_allocator.rs:170-173,844).
// FP8 (E4M3) cache: same geometry with one byte per element (kvcache.cuh)
template <class DT>
__global__ __launch_bounds__(256) void reshape_and_cache_fp8_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                                                    uint8_t* __restrict__ kc, uint8_t* __restrict__ vc,
                                                                    const int64_t* __restrict__ slots, int Hkv, int D, int BS) {
  const int t = blockIdx.x;
  const int64_t slot = slots[t];
  if (slot < 0) return;
  const int64_t blk = slot / BS;
  const int off = (int)(slot % BS);
  const int n = Hkv * D;
  for (int i = threadIdx.x * 4; i < n; i += blockDim.x * 4) {  // D % 4 == 0: four channels of one head per thread
    const int h = i / D, d = i - h * D;
    float fk[4], fv[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      fk[e] = DT::to_f32(k[(size_t)t * n + i + e]);
      fv[e] = DT::to_f32(v[(size_t)t * n + i + e]);
    }
    *reinterpret_cast<uint32_t*>(kc + ((blk * Hkv + h) * BS + off) * D + d) = vra_f32x4_to_e4m3(fk[0], fk[1], fk[2], fk[3]);
    const uint32_t q = vra_f32x4_to_e4m3(fv[0], fv[1], fv[2], fv[3]);
#pragma unroll
    for (int e = 0; e < 4; e++) vc[((blk * Hkv + h) * D + d + e) * BS + off] = (uint8_t)(q >> (8 * e));
  }
}
__global__ __launch_bounds__(256) void reshape_and_cache_kernel(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                                                uint16_t* __restrict__ kc, uint16_t* __restrict__ vc,
                                                                const int64_t* __restrict__ slots, int Hkv, int D, int BS) {
  const int t = blockIdx.x;
  const int64_t slot = slots[t];
  if (slot < 0) return;
  const int64_t blk = slot / BS;
  const int off = (int)(slot % BS);
  const int n = Hkv * D;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int h = i / D, d = i - h * D;
    kc[((blk * Hkv + h) * BS + off) * D + d] = k[(size_t)t * n + i];
    vc[((blk * Hkv + h) * D + d) * BS + off] = v[(size_t)t * n + i];
  }
}
extern "C" void vra_reshape_and_cache(const void* k, const void* v, void* k_cache, void* v_cache, const int64_t* slot_mapping,
                                      int32_t tokens, int32_t kv_heads, int32_t head_dim, int32_t block_size, int32_t dtype,
                                      int32_t kv_dtype, int64_t stream) {
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_reshape_and_cache: dtype must be bf16/f16");
  VRA_CHECK_ARG(kv_dtype == dtype || kv_dtype == VRA_FP8_E4M3, "vra_reshape_and_cache: kv_dtype must be the activation dtype or VRA_FP8_E4M3");
  if (tokens <= 0) return;
  if (kv_dtype == VRA_FP8_E4M3) {
    VRA_CHECK_ARG(head_dim % 4 == 0, "vra_reshape_and_cache: head_dim %% 4 != 0");
    if (dtype == VRA_BF16)
      reshape_and_cache_fp8_kernel<BF16><<<tokens, 256, 0, as_stream(stream)>>>((const uint16_t*)k, (const uint16_t*)v, (uint8_t*)k_cache, (uint8_t*)v_cache, slot_mapping, kv_heads, head_dim, block_size);
    else
      reshape_and_cache_fp8_kernel<F16><<<tokens, 256, 0, as_stream(stream)>>>((const uint16_t*)k, (const uint16_t*)v, (uint8_t*)k_cache, (uint8_t*)v_cache, slot_mapping, kv_heads, head_dim, block_size);
    return;
  }
  reshape_and_cache_kernel<<<tokens, 256, 0, as_stream(stream)>>>((const uint16_t*)k, (const uint16_t*)v, (uint16_t*)k_cache, (uint16_t*)v_cache, slot_mapping, kv_heads, head_dim, block_size);
}

// ---------------------------------------------------------------- RoPE + KV scatter in one launch (prefill)
// One workgroup per token: q is rotated in place, k is rotated straight into its K-cache row (the rotated k is not written back:
// prefill attention reads K and V from the cache), v goes to its token-minor V-cache column.  NeoX pairs, full rotary width,
// tables in the model dtype — the arithmetic of rope_kernel and the bytes of reshape_and_cache_kernel, i.e. bit-identical to the
// two launches (tests/test_gpu_kernels.py); everything else takes the two launches.
// TG = tokens per workgroup.  8 consecutive tokens whose slots are consecutive inside one block (the normal case of a prefill:
// positions and slots run together) write the token-minor V cache with ONE 16-byte (FP8: 8-byte) store per channel instead of
// 8 two-byte stores 128 bytes apart — the scattered stores were 3/4 of this kernel's time at 4096 tokens.  Groups that are not
// (chunk edges, block edges that are not multiples of 8, padded lanes) take the per-token stores.  Same bytes either way.
template <class DT, bool KV8, int TG>
__global__ __launch_bounds__(256) void rope_cache_kernel(uint16_t* __restrict__ q, const uint16_t* __restrict__ k, const uint16_t* __restrict__ v,
                                                         typename KVT<KV8>::elem* __restrict__ kc, typename KVT<KV8>::elem* __restrict__ vc,
                                                         const uint16_t* __restrict__ cosv, const uint16_t* __restrict__ sinv,
                                                         const int64_t* __restrict__ positions, const int64_t* __restrict__ slots, int T, int Hq,
                                                         int Hkv, int D, int BS) {
  typedef typename KVT<KV8>::elem kv_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char rc_smem[];  // TG == 8: the group's V rows [8][Hkv*D] in cache format
  const int tid = threadIdx.x;
  const int t0 = blockIdx.x * TG, nt = min(TG, T - t0);
  const int half = D >> 1, per_head = half >> 3;  // 16-byte pair groups per head
  const int n = Hkv * D;
  auto rotate = [&](const uint16_t* base, int c, const uint16_t* cs_row, const uint16_t* sn_row, u32x4& r1, u32x4& r2) {
    float x1[8], x2[8], cs[8], sn[8], y1[8], y2[8];
    unpack8<DT>(*reinterpret_cast<const u32x4*>(base + c * 8), x1);
    unpack8<DT>(*reinterpret_cast<const u32x4*>(base + half + c * 8), x2);
    unpack8<DT>(*reinterpret_cast<const u32x4*>(cs_row + c * 8), cs);
    unpack8<DT>(*reinterpret_cast<const u32x4*>(sn_row + c * 8), sn);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      y1[e] = x1[e] * cs[e] - x2[e] * sn[e];
      y2[e] = x2[e] * cs[e] + x1[e] * sn[e];
    }
    r1 = pack8<DT>(y1), r2 = pack8<DT>(y2);
  };
  // the group's slots: vector V stores need slot0 % 8 == 0 inside one block and slots slot0 .. slot0 + 7
  bool vec = TG == 8 && nt == 8;
  const int64_t slot0 = slots[t0];
  if (TG == 8) {
#pragma unroll
    for (int i = 0; i < 8; i++) vec = vec && i < nt && slots[t0 + min(i, nt - 1)] == slot0 + i;
    vec = vec && slot0 >= 0 && (slot0 % BS) % 8 == 0 && (slot0 % BS) + 8 <= BS;
  }
  for (int ti = 0; ti < nt; ti++) {
    const int t = t0 + ti;
    const int64_t pos = positions[t], slot = slots[t];
    const uint16_t* cs_row = cosv + pos * half;
    const uint16_t* sn_row = sinv + pos * half;
    for (int w = tid; w < Hq * per_head; w += blockDim.x) {  // q: in place
      const int h = w / per_head, c = w - h * per_head;
      uint16_t* base = q + ((size_t)t * Hq + h) * D;
      u32x4 r1, r2;
      rotate(base, c, cs_row, sn_row, r1, r2);
      *reinterpret_cast<u32x4*>(base + c * 8) = r1;
      *reinterpret_cast<u32x4*>(base + half + c * 8) = r2;
    }
    if (slot < 0) continue;  // padded lane: nothing is cached
    const int64_t blk = slot / BS;
    const int off = (int)(slot % BS);
    for (int w = tid; w < Hkv * per_head; w += blockDim.x) {  // k: rotated into the cache row
      const int h = w / per_head, c = w - h * per_head;
      u32x4 r1, r2;
      rotate(k + ((size_t)t * Hkv + h) * D, c, cs_row, sn_row, r1, r2);
      kv_t* row = kc + ((blk * Hkv + h) * BS + off) * D;
      kv_store8<DT, KV8>(row + c * 8, r1);
      kv_store8<DT, KV8>(row + half + c * 8, r2);
    }
    for (int w = tid; w < Hkv * (D >> 3); w += blockDim.x) {  // v: 8 channels per thread
      const int h = w / (D >> 3), c = w - h * (D >> 3);
      const u32x4 vv = *reinterpret_cast<const u32x4*>(v + ((size_t)t * Hkv + h) * D + c * 8);
      if (TG == 8 && vec) {  // staged in cache format; stored token-minor below
        kv_store8<DT, KV8>(reinterpret_cast<kv_t*>(rc_smem) + (size_t)ti * n + h * D + c * 8, vv);
        continue;
      }
      kv_t* col = vc + ((blk * Hkv + h) * D + c * 8) * BS + off;
      if constexpr (KV8) {
        const u32x2 q8 = vra_pack_e4m3x8<DT>(vv);
#pragma unroll
        for (int e = 0; e < 8; e++) col[(size_t)e * BS] = (uint8_t)(q8[e >> 2] >> (8 * (e & 3)));
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          col[(size_t)(2 * e) * BS] = (uint16_t)(vv[e] & 0xffffu);
          col[(size_t)(2 * e + 1) * BS] = (uint16_t)(vv[e] >> 16);
        }
      }
    }
  }
  if (TG == 8 && vec) {
    __syncthreads();
    const int64_t blk = slot0 / BS;
    const int off = (int)(slot0 % BS);
    const kv_t* sv = reinterpret_cast<const kv_t*>(rc_smem);
    for (int i = tid; i < n; i += blockDim.x) {  // channel i = (h, d): its 8 tokens are 8 consecutive cache elements
      const int h = i / D, d = i - h * D;
      kv_t tok[8];
#pragma unroll
      for (int e = 0; e < 8; e++) tok[e] = sv[(size_t)e * n + i];
      kv_t* dst = vc + ((blk * Hkv + h) * D + d) * BS + off;
      if constexpr (KV8) {
        u32x2 o;
        o[0] = (uint32_t)tok[0] | ((uint32_t)tok[1] << 8) | ((uint32_t)tok[2] << 16) | ((uint32_t)tok[3] << 24);
        o[1] = (uint32_t)tok[4] | ((uint32_t)tok[5] << 8) | ((uint32_t)tok[6] << 16) | ((uint32_t)tok[7] << 24);
        *reinterpret_cast<u32x2*>(dst) = o;
      } else {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (uint32_t)tok[2 * e] | ((uint32_t)tok[2 * e + 1] << 16);
        *reinterpret_cast<u32x4*>(dst) = o;
      }
    }
  }
}
extern "C" void vra_rope_cache_prefill(void* q, const void* k, const void* v, void* k_cache, void* v_cache, const void* cos, const void* sin,
                                       const int64_t* positions, const int64_t* slot_mapping, int32_t tokens, int32_t q_heads, int32_t kv_heads,
                                       int32_t head_dim, int32_t block_size, int32_t dtype, int32_t kv_dtype, int64_t stream) {
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_rope_cache_prefill: dtype must be bf16/f16");
  VRA_CHECK_ARG(kv_dtype == dtype || kv_dtype == VRA_FP8_E4M3, "vra_rope_cache_prefill: kv_dtype must be the activation dtype or VRA_FP8_E4M3");
  VRA_CHECK_ARG(q && k && v && k_cache && v_cache && cos && sin && positions && slot_mapping, "vra_rope_cache_prefill: null pointer");
  VRA_CHECK_ARG(head_dim % 16 == 0 && head_dim <= 256, "vra_rope_cache_prefill: head_dim must be a multiple of 16 (<= 256)");
  if (tokens <= 0) return;
  hipStream_t st = as_stream(stream);
  const bool kv8 = kv_dtype == VRA_FP8_E4M3;
  // groups of 8 tokens (vector V stores) when there are enough tokens to fill the chip that way and the staging fits in LDS
  const size_t lds8 = (size_t)8 * kv_heads * head_dim * (kv8 ? 1 : 2);
  const bool g8 = tokens >= 1024 && lds8 <= 64 * 1024 && block_size % 8 == 0;
#define VRA_RC(DT, K8)                                                                                                                    \
  do {                                                                                                                                    \
    if (g8)                                                                                                                               \
      rope_cache_kernel<DT, K8, 8><<<(tokens + 7) / 8, 256, lds8, st>>>((uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,             \
                                                                        (typename KVT<K8>::elem*)k_cache, (typename KVT<K8>::elem*)v_cache, \
                                                                        (const uint16_t*)cos, (const uint16_t*)sin, positions, slot_mapping, \
                                                                        tokens, q_heads, kv_heads, head_dim, block_size);                  \
    else                                                                                                                                  \
      rope_cache_kernel<DT, K8, 1><<<tokens, 256, 0, st>>>((uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v,                           \
                                                           (typename KVT<K8>::elem*)k_cache, (typename KVT<K8>::elem*)v_cache,              \
                                                           (const uint16_t*)cos, (const uint16_t*)sin, positions, slot_mapping, tokens,     \
                                                           q_heads, kv_heads, head_dim, block_size);                                       \
  } while (0)
  if (dtype == VRA_BF16) {
    if (kv8) VRA_RC(BF16, true);
    else VRA_RC(BF16, false);
  } else {
    if (kv8) VRA_RC(F16, true);
    else VRA_RC(F16, false);
  }
#undef VRA_RC
}

// ---------------------------------------------------------------- causal mask
template <class DT>
__global__ void causal_mask_kernel(uint16_t* mask, int L, int sw) {
  const int64_t total = (int64_t)L * L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(i / L), c = (int)(i % L);
    bool ok = c <= r && (sw <= 0 || r - c < sw);
    mask[i] = DT::from_f32(ok ? 0.f : -INFINITY);
  }
}
__global__ void causal_mask_kernel_f32(float* mask, int L, int sw) {
  const int64_t total = (int64_t)L * L;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(i / L), c = (int)(i % L);
    bool ok = c <= r && (sw <= 0 || r - c < sw);
    mask[i] = ok ? 0.f : -INFINITY;
  }
}
extern "C" void vra_causal_mask(void* mask, int32_t len, int32_t sliding_window, int32_t dtype, int64_t stream) {
  if (len <= 0) return;
  int grid = (int)(((int64_t)len * len + 255) / 256);
  if (grid > 4096) grid = 4096;
  if (dtype == VRA_BF16) causal_mask_kernel<BF16><<<grid, 256, 0, as_stream(stream)>>>((uint16_t*)mask, len, sliding_window);
  else if (dtype == VRA_F16) causal_mask_kernel<F16><<<grid, 256, 0, as_stream(stream)>>>((uint16_t*)mask, len, sliding_window);
  else causal_mask_kernel_f32<<<grid, 256, 0, as_stream(stream)>>>((float*)mask, len, sliding_window);
}

// ---------------------------------------------------------------- argmax (first maximal index)
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, uint32_t* __restrict__ out, int cols) {
  __shared__ float bv[16];
  __shared__ uint32_t bi[16];
  const float* p = logits + (size_t)blockIdx.x * cols;
  float best = -INFINITY;
  uint32_t besti = 0xffffffffu;
  // 16 B loads, 4 independent loads in flight per thread (the row is read once: latency bound otherwise)
  const int nv = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? cols >> 2 : 0;
  const f32x4* p4 = reinterpret_cast<const f32x4*>(p);
  for (int c0 = threadIdx.x; c0 < nv; c0 += blockDim.x * 4) {
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int c = c0 + u * blockDim.x;
      v[u] = c < nv ? p4[c] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int c = c0 + u * blockDim.x;
      if (c < nv) {
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (v[u][e] > best || besti == 0xffffffffu) {  // ascending index order per thread: strict > keeps the first
            best = v[u][e];
            besti = (uint32_t)(c * 4 + e);
          }
      }
    }
  }
  for (int c = nv * 4 + threadIdx.x; c < cols; c += blockDim.x) {
    float v = p[c];
    if (v > best || besti == 0xffffffffu) {
      best = v;
      besti = c;
    }
  }
  // wave reduce: larger value wins, ties -> smaller index
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float ov = __shfl_xor(best, o, 64);
    uint32_t oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) {
      best = ov;
      besti = oi;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    bv[wave] = best;
    bi[wave] = besti;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); w++)
      if (bv[w] > best || (bv[w] == best && bi[w] < besti)) {
        best = bv[w];
        besti = bi[w];
      }
    out[blockIdx.x] = besti == 0xffffffffu ? 0u : besti;
  }
}
extern "C" void vra_argmax_f32(const float* logits, uint32_t* out, int32_t rows, int32_t cols, int64_t stream) {
  if (rows <= 0 || cols <= 0) return;
  argmax_kernel<<<rows, 1024, 0, as_stream(stream)>>>(logits, out, cols);
}

// ---------------------------------------------------------------- casts
template <class SRC, class DST>
__global__ void cast_kernel(const void* in, void* out, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) DST::store(out, i, SRC::load(in, i));
}
template <class DT>
struct Io16 {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return DT::to_f32(static_cast<const uint16_t*>(p)[i]); }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { static_cast<uint16_t*>(p)[i] = DT::from_f32(v); }
};
struct Io32 {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return static_cast<const float*>(p)[i]; }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { static_cast<float*>(p)[i] = v; }
};
extern "C" void vra_cast(const void* in, void* out, int64_t numel, int32_t in_dtype, int32_t out_dtype, int64_t stream) {
  if (numel <= 0) return;
  int grid = (int)((numel + 255) / 256);
  if (grid > 4096) grid = 4096;
  hipStream_t st = as_stream(stream);
#define VRA_CAST(S, D) cast_kernel<S, D><<<grid, 256, 0, st>>>(in, out, numel)
  int key = in_dtype * 3 + out_dtype;
  switch (key) {
    case 0: VRA_CAST(Io16<BF16>, Io16<BF16>); break;
    case 1: VRA_CAST(Io16<BF16>, Io16<F16>); break;
    case 2: VRA_CAST(Io16<BF16>, Io32); break;
    case 3: VRA_CAST(Io16<F16>, Io16<BF16>); break;
    case 4: VRA_CAST(Io16<F16>, Io16<F16>); break;
    case 5: VRA_CAST(Io16<F16>, Io32); break;
    case 6: VRA_CAST(Io32, Io16<BF16>); break;
    case 7: VRA_CAST(Io32, Io16<F16>); break;
    case 8: VRA_CAST(Io32, Io32); break;
    default: vra_set_error("vra_cast: bad dtypes %d -> %d", in_dtype, out_dtype);
  }
#undef VRA_CAST
}

// ---------------------------------------------------------------- block swap
extern "C" void vra_swap_blocks(const void* src, void* dst, const int64_t* h_pairs, int32_t n_pairs, int64_t block_bytes,
                                int32_t kind, int64_t stream) {
  hipMemcpyKind mk = kind == 0 ? hipMemcpyDeviceToDevice : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice);
  for (int i = 0; i < n_pairs; i++) {
    const char* s = static_cast<const char*>(src) + h_pairs[2 * i] * block_bytes;
    char* d = static_cast<char*>(dst) + h_pairs[2 * i + 1] * block_bytes;
    if (hipMemcpyAsync(d, s, (size_t)block_bytes, mk, as_stream(stream)) != hipSuccess) {
      vra_set_error("vra_swap_blocks: memcpy failed for pair %d", i);
      return;
    }
  }
}

// ---------------------------------------------------------------- synthetic fills (== oracle)
__global__ void fill_hash_kernel(uint32_t* out, int64_t n, uint64_t seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = vra_hash32(seed, (uint64_t)i);
}
// AWQ zero points of the synthetic checkpoints: every nibble of a hash word is mapped through a 16-entry table that is
// concentrated on 8 (6:1 7:3 8:8 9:3 10:1 of 16) — what real AWQ checkpoints look like (asymmetric min/max quantisation of
// near-symmetric weight groups).  Uniform zero points (the recipe of rounds 1-3) put a common-mode term of up to 7.5 scales
// on every group, which made the full-depth synthetic network a noise amplifier.  == oracle orc_fill_awq_zeros
#define VRA_AWQ_ZERO_LUT 0xA999888888887776ull
__global__ void fill_awq_zeros_kernel(uint32_t* out, int64_t n, uint64_t seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = vra_hash32(seed, (uint64_t)i);
    uint32_t w = 0;
#pragma unroll
    for (int p = 0; p < 8; p++) w |= (uint32_t)((VRA_AWQ_ZERO_LUT >> (4 * ((h >> (4 * p)) & 0xFu))) & 0xFull) << (4 * p);
    out[i] = w;
  }
}
__global__ void fill_const_kernel(uint32_t* out, int64_t n, uint32_t v) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = v;
}
template <class IO, int NORMAL>
__global__ void fill_float_kernel(void* out, int64_t n, uint64_t seed, float a, float b) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v;
    if (NORMAL) {
      float s = vra_hash_unit(seed, 4ull * i) + vra_hash_unit(seed, 4ull * i + 1) + vra_hash_unit(seed, 4ull * i + 2) + vra_hash_unit(seed, 4ull * i + 3);
      v = a + b * (s - 2.0f) * 1.7320508f;
    } else {
      v = a + (b - a) * vra_hash_unit(seed, (uint64_t)i);
    }
    IO::store(out, i, v);
  }
}
static inline int fill_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}
extern "C" void vra_fill_hash_u32(uint32_t* out, int64_t numel, uint64_t seed, int64_t stream) {
  if (numel > 0) fill_hash_kernel<<<fill_grid(numel), 256, 0, as_stream(stream)>>>(out, numel, seed);
}
extern "C" void vra_fill_awq_zeros(uint32_t* out, int64_t numel, uint64_t seed, int64_t stream) {
  if (numel > 0) fill_awq_zeros_kernel<<<fill_grid(numel), 256, 0, as_stream(stream)>>>(out, numel, seed);
}
extern "C" void vra_fill_const_u32(uint32_t* out, int64_t numel, uint32_t value, int64_t stream) {
  if (numel > 0) fill_const_kernel<<<fill_grid(numel), 256, 0, as_stream(stream)>>>(out, numel, value);
}
extern "C" void vra_fill_uniform(void* out, int64_t numel, uint64_t seed, float lo, float hi, int32_t dtype, int64_t stream) {
  if (numel <= 0) return;
  if (dtype == VRA_BF16) fill_float_kernel<Io16<BF16>, 0><<<fill_grid(numel), 256, 0, as_stream(stream)>>>(out, numel, seed, lo, hi);
  else if (dtype == VRA_F16) fill_float_kernel<Io16<F16>, 0><<<fill_grid(numel), 256, 0, as_stream(stream)>>>(out, numel, seed, lo, hi);
  else fill_float_kernel<Io32, 0><<<fill_grid(numel), 256, 0, as_stream(stream)>>>(out, numel, seed, lo, hi);
}
extern "C" void vra_fill_normal(void* out, int64_t numel, uint64_t seed, float mean, float std, int32_t dtype, int64_t stream) {
  if (numel <= 0) return;
  if (dtype == VRA_BF16) fill_float_kernel<Io16<BF16>, 1><<<fill_grid(numel), 256, 0, as_stream(stream)>>>(out, numel, seed, mean, std);
  else if (dtype == VRA_F16) fill_float_kernel<Io16<F16>, 1><<<fill_grid(numel), 256, 0, as_stream(stream)>>>(out, numel, seed, mean, std);
  else fill_float_kernel<Io32, 1><<<fill_grid(numel), 256, 0, as_stream(stream)>>>(out, numel, seed, mean, std);
}
