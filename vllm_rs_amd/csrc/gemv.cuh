// gemv.cuh — "kernel A": skinny GEMM (M <= 8 rows) streaming each weight byte exactly once.
//
// Roofline: HBM.  Algorithmic bytes per call: K*N/2 (packed int4) + (K/g)*N*2 (scales)
// [+ (K/g)*N/2 zeros] + M*K*2 + M*N*2   (dense variant: N*K*2 + ...).
//
// Shape of the launch:
//   workgroup = 512 threads = 8 waves; it owns NBW adjacent-in-role 16-column n-blocks and the
//   FULL K range.  Wave w takes k-tiles w, w+8, ... (128 rows each, one 1 KiB coalesced
//   `global_load_dwordx4` per tile: 64 lanes x 16 B) so the eight waves of a workgroup walk one
//   contiguous K/128 KiB run of the tiled weight tensor.  Partial sums meet in LDS — no inter-
//   workgroup traffic, no atomics, no second launch.
//   x (optionally RMS-normalised on the fly: the reference's separate NormX launch,
//   others.rs:11-29) and the scales/zeros of the owned columns are staged in LDS once.
//   The multiply is `v_mfma_f32_16x16x32` with A = dequantised weights (16 columns x 32 k),
//   B = xT (32 k x 16 rows, rows >= M are zero): MFMA does the cross-lane k reduction for free
//   and the VALU only dequantises.
#pragma once
#include "wna16.cuh"

#define GEMV_THREADS 512
#define GEMV_WAVES 8
#define GEMV_MAX_SEG 3

struct GemvSeg {
  const void* w;           // int4: tiled words; dense: row-major [n, K] 16-bit
  const void* scales;      // int4: [K/g, n]
  const uint32_t* qzeros;  // awq: raw [K/g, n/8]; else null (zero point 8)
  const void* bias;        // [n] or null
  void* out;               // [M, out_ld]
  int n;                   // columns of this segment
  int out_ld;
  int blk_start;           // first flattened n-block of this segment
};
struct GemvArgs {
  GemvSeg seg[GEMV_MAX_SEG];
  int nseg;
  const void* x;  // [M, K] (ld = x_ld)
  int x_ld;
  const void* norm_w;  // non-null: x <- rmsnorm(x) * norm_w while staging
  float eps;
  const void* residual;  // [M, res_ld] added after bias (single-segment launches)
  int res_ld;
  int M, K;
  int group_size;  // -1 => K
  int is_awq, scales_layout;
  int silu_dual;  // 1: seg0 = gate, seg1 = up, out = silu(gate)*up into seg0.out
  int out_f32;    // 1: store (float)round_dt(v)
};

template <class DT>
__device__ __forceinline__ void gemv_stage_x(const GemvArgs& a, uint32_t* xs, float* red8) {
  // LDS image: xs[(oct * M + m) * 4 .. +4] (u32) = x[m][oct*8 .. +8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int octs = a.K >> 3;
  for (int m = 0; m < a.M; m++) {
    const u32x4* xr = reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.x) + (size_t)m * a.x_ld);
    float rstd = 1.0f;
    if (a.norm_w) {
      float ss = 0.f;
      for (int o = tid; o < octs; o += GEMV_THREADS) {
        float f[8];
        unpack8<DT>(xr[o], f);
#pragma unroll
        for (int i = 0; i < 8; i++) ss += f[i] * f[i];
      }
      ss = wave_sum(ss);
      __syncthreads();  // red8 reuse across rows
      if (lane == 0) red8[wave] = ss;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < GEMV_WAVES; w++) tot += red8[w];
      rstd = 1.0f / sqrtf(tot / (float)a.K + a.eps);
    }
    const u32x4* nw = reinterpret_cast<const u32x4*>(a.norm_w);
    for (int o = tid; o < octs; o += GEMV_THREADS) {
      u32x4 v = xr[o];
      if (a.norm_w) {
        float f[8], g[8];
        unpack8<DT>(v, f);
        unpack8<DT>(nw[o], g);
#pragma unroll
        for (int i = 0; i < 8; i++) f[i] = f[i] * rstd * g[i];
        v = pack8<DT>(f);
      }
      *reinterpret_cast<u32x4*>(xs + ((size_t)o * a.M + m) * 4) = v;
    }
  }
}

// INT4 = true: tiled int4 weights; false: dense row-major 16-bit weights.
template <class DT, bool INT4, int NBW>
__global__ __launch_bounds__(GEMV_THREADS) void gemv_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nn = lane & 15, oct = lane >> 4;
  const int K = a.K, M = a.M, KT = K >> 7;
  const int g = a.group_size > 0 ? a.group_size : K;
  const int G = K / g;

  // ---- which n-blocks does this workgroup own?
  int segi[NBW], nb[NBW];
  if (a.silu_dual) {  // NBW == 2: block b of gate and block b of up
    segi[0] = 0;
    nb[0] = blockIdx.x;
    if (NBW > 1) {
      segi[NBW - 1] = 1;
      nb[NBW - 1] = blockIdx.x;
    }
  } else {
#pragma unroll
    for (int b = 0; b < NBW; b++) {
      int fb = blockIdx.x * NBW + b;
      int s = 0;
      if (a.nseg > 1 && fb >= a.seg[1].blk_start) s = 1;
      if (a.nseg > 2 && fb >= a.seg[2].blk_start) s = 2;
      segi[b] = s;
      nb[b] = fb - a.seg[s].blk_start;
    }
  }

  // ---- LDS carve-up
  uint32_t* xs = reinterpret_cast<uint32_t*>(smem);                           // M*K*2 bytes
  size_t off = (((size_t)M * K * 2 + 15) & ~(size_t)15) + 16;  // + one 16 B all-zero slot
  f32x2* sc = reinterpret_cast<f32x2*>(smem + off);                           // NBW*G*16 pairs
  off += INT4 ? (size_t)NBW * G * 16 * sizeof(f32x2) : 0;
  f32x4* red = reinterpret_cast<f32x4*>(smem + off);                          // 8*NBW*64 f32x4
  off += (size_t)GEMV_WAVES * NBW * 64 * sizeof(f32x4);
  float* red8 = reinterpret_cast<float*>(smem + off);                         // 8 floats

  // ---- stage scales / zero points of the owned columns as (s, -z*s) in f32
  if (INT4) {
    for (int idx = tid; idx < NBW * G * 16; idx += GEMV_THREADS) {
      int b = idx / (G * 16), grp = (idx >> 4) % G, c = idx & 15;
      const GemvSeg& sg = a.seg[segi[b]];
      int n = nb[b] * 16 + c;
      float s = 0.f, z = 8.f;
      if (n < sg.n) {
        s = DT::to_f32(static_cast<const uint16_t*>(sg.scales)[vra_scale_index(grp, n, sg.n, a.scales_layout, a.group_size > 0 && a.group_size < K)]);
        if (a.is_awq && sg.qzeros) z = (float)((sg.qzeros[(size_t)grp * (sg.n >> 3) + (n >> 3)] >> (4 * awq_rev(n & 7))) & 0xFu);
      }
      f32x2 p = {s, -z * s};
      sc[idx] = p;
    }
  }
  const uint32_t zero_slot = (uint32_t)((((size_t)M * K * 2 + 15) & ~(size_t)15) >> 2);
  if (tid < 4) xs[zero_slot + tid] = 0u;
  gemv_stage_x<DT>(a, xs, red8);
  __syncthreads();

  // ---- main loop: wave `wave` owns k-tiles wave, wave+8, ...
  f32x4 acc[NBW];
#pragma unroll
  for (int b = 0; b < NBW; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nt = (KT - wave + GEMV_WAVES - 1) / GEMV_WAVES;  // tiles of this wave (may be <= 0)
  // B fragment addressing: lanes whose batch row does not exist read the zero slot (stride 0)
  const bool row_ok = nn < M;
  const uint32_t xbase = row_ok ? (uint32_t)(oct * M + nn) * 4u : zero_slot;
  const uint32_t xstride = row_ok ? (uint32_t)M * 4u : 0u;  // u32 per octet step

  if (INT4) {
    const u32x4* wp[NBW];
#pragma unroll
    for (int b = 0; b < NBW; b++)
      wp[b] = reinterpret_cast<const u32x4*>(a.seg[segi[b]].w) + ((size_t)nb[b] * KT) * 64 + lane;
    constexpr int UK = 4 / NBW;  // k-tiles per batch (4 loads in flight per batch, 2 batches deep)
    u32x4 cur[UK][NBW], nxt[UK][NBW];
#pragma unroll
    for (int u = 0; u < UK; u++)
#pragma unroll
      for (int b = 0; b < NBW; b++)
        if (u < nt) cur[u][b] = __builtin_nontemporal_load(wp[b] + (size_t)(wave + GEMV_WAVES * u) * 64);
    for (int t0 = 0; t0 < nt; t0 += UK) {
#pragma unroll
      for (int u = 0; u < UK; u++)
#pragma unroll
        for (int b = 0; b < NBW; b++)
          if (t0 + UK + u < nt) nxt[u][b] = __builtin_nontemporal_load(wp[b] + (size_t)(wave + GEMV_WAVES * (t0 + UK + u)) * 64);
#pragma unroll
      for (int u = 0; u < UK; u++) {
        if (t0 + u < nt) {
          const int kt = wave + GEMV_WAVES * (t0 + u);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const u32x4 xb = *reinterpret_cast<const u32x4*>(xs + xbase + (uint32_t)(kt * 16 + j * 4) * xstride);
            const s16x8 bfrag = __builtin_bit_cast(s16x8, xb);
            const int grp = (kt * 128 + j * 32) / g;
#pragma unroll
            for (int b = 0; b < NBW; b++) {
              f32x2 p = sc[(b * G + grp) * 16 + nn];
              s16x8 afrag = dequant_word<DT>(cur[u][b][j], p[0], p[1]);
              acc[b] = DT::mfma(afrag, bfrag, acc[b]);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UK; u++)
#pragma unroll
        for (int b = 0; b < NBW; b++) cur[u][b] = nxt[u][b];
    }
  } else {
    // dense: lane (nn, oct) streams row n = nb*16+nn; per k-tile 4 x 16 B at k = kt*128 + j*32 + oct*8
    const u32x4* wp[NBW];
    bool col_ok[NBW];
#pragma unroll
    for (int b = 0; b < NBW; b++) {
      int n = nb[b] * 16 + nn;
      col_ok[b] = n < a.seg[segi[b]].n;
      wp[b] = reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.seg[segi[b]].w) + (size_t)(col_ok[b] ? n : 0) * K) + oct;
    }
    for (int t = 0; t < nt; t++) {
      const int kt = wave + GEMV_WAVES * t;
      u32x4 q[NBW][4];
#pragma unroll
      for (int b = 0; b < NBW; b++)
#pragma unroll
        for (int j = 0; j < 4; j++) q[b][j] = __builtin_nontemporal_load(wp[b] + kt * 16 + j * 4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const u32x4 xb = *reinterpret_cast<const u32x4*>(xs + xbase + (uint32_t)(kt * 16 + j * 4) * xstride);
        const s16x8 bfrag = __builtin_bit_cast(s16x8, xb);
#pragma unroll
        for (int b = 0; b < NBW; b++) {
          u32x4 qa = col_ok[b] ? q[b][j] : u32x4{0u, 0u, 0u, 0u};
          acc[b] = DT::mfma(__builtin_bit_cast(s16x8, qa), bfrag, acc[b]);
        }
      }
    }
  }

  // ---- cross-wave reduction in LDS, then the fused epilogue
#pragma unroll
  for (int b = 0; b < NBW; b++) red[(wave * NBW + b) * 64 + lane] = acc[b];
  __syncthreads();
  const int nout = a.silu_dual ? 16 * M : NBW * 16 * M;
  for (int idx = tid; idx < nout; idx += GEMV_THREADS) {
    const int nl = idx & 15, m = (idx >> 4) % M, b = idx / (16 * M);
    const int rl = (nl >> 2) * 16 + m, rr = nl & 3;  // D layout: row = (lane>>4)*4 + reg, col = lane&15
    float v = 0.f, v2 = 0.f;
#pragma unroll
    for (int w = 0; w < GEMV_WAVES; w++) {
      v += red[(w * NBW + b) * 64 + rl][rr];
      if (NBW > 1) v2 += red[(w * NBW + (NBW - 1)) * 64 + rl][rr];
    }
    const GemvSeg& sg = a.seg[segi[b]];
    const int n = nb[b] * 16 + nl;
    if (n >= sg.n) continue;
    v = rnd_dt<DT>(v);
    if (sg.bias) v = rnd_dt<DT>(v + DT::to_f32(static_cast<const uint16_t*>(sg.bias)[n]));
    if (a.silu_dual) {
      const GemvSeg& su = a.seg[1];
      v2 = rnd_dt<DT>(v2);
      if (su.bias) v2 = rnd_dt<DT>(v2 + DT::to_f32(static_cast<const uint16_t*>(su.bias)[n]));
      float sl = rnd_dt<DT>(v / (1.0f + expf(-v)));
      v = sl * v2;
    }
    if (a.residual) v = rnd_dt<DT>(v) + DT::to_f32(static_cast<const uint16_t*>(a.residual)[(size_t)m * a.res_ld + n]);
    if (a.out_f32) static_cast<float*>(sg.out)[(size_t)m * sg.out_ld + n] = rnd_dt<DT>(v);
    else static_cast<uint16_t*>(sg.out)[(size_t)m * sg.out_ld + n] = DT::from_f32(v);
  }
}

static inline size_t gemv_lds_bytes(bool int4, int nbw, int M, int K, int group_size) {
  int g = group_size > 0 ? group_size : K;
  size_t b = (((size_t)M * K * 2 + 15) & ~(size_t)15) + 16;
  if (int4) b += (size_t)nbw * (K / g) * 16 * 8;
  b += (size_t)GEMV_WAVES * nbw * 64 * 16 + 64;
  return b;
}
