// gemv_dw.cuh — "kernel W, dense": 16-bit [N, K] weights (the lm_head) for decode batches of 4..32 rows (engine; vra_dense_gemm: what kernel A does not take) and K <= 4096.
//
// Before: RMSNorm launch + kernel B (gemm_skinny.cuh), which stages x through LDS in 256-k chunks behind a workgroup barrier
// per chunk and keeps a whole chunk of weights in registers: 222..235 µs for the Llama-3 lm_head (1.05 GB) at 8..32 rows
// (4.5..4.7 TB/s, plus 5 µs for the norm) against 187 µs for the single-row kernel.  This kernel: 183..188 µs (5.6..5.75 TB/s).
// It is kernel W's decomposition (gemv_q4w.cuh) without the int4 machinery:
//   * one workgroup per CU, a contiguous run of units (16 output columns = 16 rows of W) per workgroup, 8 waves;
//   * wave w owns the k-tiles w, w+8, w+16, w+24: its x fragments (32 rows x 512 columns) live in REGISTERS for the whole
//     launch (loaded once, with W's half-line swizzle), optionally RMS-normalised in place — Σx² per row from the matrix cores
//     (the diagonal of X·Xᵀ), one barrier;
//   * the weight stream is a 2-slot (two m-tiles) or 4-slot (one m-tile) ring of tile-steps (4 KiB per wave and slot), branch-free,
//     no store inside the loop; a unit's 8 partial tiles meet in a double-buffered LDS slab behind one barrier per unit and
//     the finished outputs are parked in LDS until the stream has ended.
// Roofline: HBM (every weight byte once).  Arithmetic contract: exact 16-bit products, f32 accumulation, one rounding.
#pragma once
#include "gemv_q4w.cuh"

// slots of the weight ring: 2 or 4 (a divisor of the four tile-steps of a unit, so that the slot of a step is a compile-time index).
// One m-tile (4..16 rows): 4 slots, 172 -> 165..166 us for the Llama-3 lm_head; two m-tiles (17..32 rows): the x fragments take 128
// VGPRs, 4 slots spill 28 registers and lose (186 -> 208 us), 2 stay (profiles/r05_ab_dense_w_ring.txt)
#ifndef GDW_RING_MT1
#define GDW_RING_MT1 4
#endif
#ifndef GDW_RING_MT2
#define GDW_RING_MT2 2
#endif
#define GDW_MAX_UNITS 40  // units per workgroup at most (parked outputs: 40 x 32 rows x 16 columns x 4 B = 80 KiB)

struct GemvDWArgs {
  const void* x;       // [M, x_ld]
  int x_ld;
  const void* norm_w;  // fused RMSNorm weights [K] (NORM variants), else null
  float eps;
  const void* w;       // [N, K] row-major, 16-bit
  const void* bias;    // [N] or null
  void* out;           // [M, out_ld], model dtype or f32
  int out_ld, out_f32;
  int M, K, KT;
  int n_units, units_q, units_r;  // workgroup b owns units_q (+1 if b < units_r) units starting at b*units_q + min(b, units_r)
  int dense_tiled;  // w is the tile-major copy of vra_dense_tile_weights (gemv.cuh GemvArgs::dense_tiled)
  // fused greedy argmax (f32 output, grid * M <= GEMV_AM_COUNTER): the hand-off of gemv.cuh — per-workgroup candidate keys
  // [M][grid] in am_ws, the arrival counter at am_ws[GEMV_AM_COUNTER], the last workgroup to arrive writes am_out[m] (first
  // maximal index of the logits this launch wrote, as vra_argmax_f32) and re-arms the counter
  uint32_t* am_out;
  unsigned long long* am_ws;
};

static inline size_t gemv_dw_lds_bytes(int mt, int max_units) {
  return (size_t)2 * GW_WAVES * mt * 1024 + (size_t)GW_WAVES * 32 * 4 + (size_t)max_units * mt * 256 * 4 + 64;
}

template <class DT, int MT, bool NORM>
__global__ __launch_bounds__(GW_THREADS) void gemv_dw_kernel(const GemvDWArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 15, oct = lane >> 4;
  const int M = a.M, KT = a.KT, K = a.K;
  const int wg = (int)blockIdx.x;
  const int u0 = wg * a.units_q + min(wg, a.units_r);
  const int nu = a.units_q + (wg < a.units_r ? 1 : 0);
  f32x4* red = reinterpret_cast<f32x4*>(smem);  // [2][wave][MT][64 lanes]
  size_t off = (size_t)2 * GW_WAVES * MT * 1024;
  float* part = reinterpret_cast<float*>(smem + off);  // [wave][32]
  off += (size_t)GW_WAVES * 32 * 4;
  float* outs = reinterpret_cast<float*>(smem + off);  // [unit][MT*16 rows][16 columns] f32
  constexpr int OPU = MT * 256;

  // ---- the weight stream: global step g = unit * 4 + tile index; slot g % GDW_RING.  Lane (oct, nn) holds the B fragment of column
  // unit*16 + nn: W[n][kt*128 + j*32 + oct*8 .. +8]
  constexpr int GDW_RING = MT == 1 ? GDW_RING_MT1 : GDW_RING_MT2;
  static_assert(GDW_RING == 2 || GDW_RING == 4, "GDW_RING");
  u32x4 wb[GDW_RING][4];
  auto issue = [&](int g, u32x4 (&w)[4]) {
    const int ui = min(g >> 2, nu - 1), ti = g & 3;  // steps past the end re-read the last unit (never consumed)
    const int kt = min(wave + GW_WAVES * ti, KT - 1);
    const u32x4* p = a.dense_tiled ? reinterpret_cast<const u32x4*>(a.w) + ((size_t)(u0 + ui) * KT + kt) * 256 + lane
                                   : reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.w) + (size_t)((u0 + ui) * 16 + nn) * K + kt * 128 + oct * 8);
    const int jstep = a.dense_tiled ? 64 : 4;
#pragma unroll
    for (int j = 0; j < 4; j++) w[j] = __builtin_nontemporal_load(p + j * jstep);
  };
#pragma unroll
  for (int sl = 0; sl < GDW_RING; sl++) issue(sl, wb[sl]);
  __builtin_amdgcn_sched_barrier(0);

  // ---- x fragments (see gemv_q4w.cuh for the odd-row swizzle)
  u32x4 xf[GW_TPW][4][MT];
#pragma unroll
  for (int ti = 0; ti < GW_TPW; ti++) {
    const int kt = min(wave + GW_WAVES * ti, KT - 1);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const uint16_t* xr = static_cast<const uint16_t*>(a.x) + (size_t)min(mt * 16 + nn, M - 1) * a.x_ld + kt * 128 + oct * 8;
#pragma unroll
      for (int j = 0; j < 4; j++) xf[ti][j][mt] = *reinterpret_cast<const u32x4*>(xr + (j ^ (nn & 1)) * 32);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (nn & 1) {
#pragma unroll
    for (int ti = 0; ti < GW_TPW; ti++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2)
#pragma unroll
          for (int c = 0; c < 4; c++) asm volatile("v_swap_b32 %0, %1" : "+v"(xf[ti][jp][mt][c]), "+v"(xf[ti][jp + 1][mt][c]));
  }
  __builtin_amdgcn_sched_barrier(0);
  if (NORM) {  // Σx² per row = the diagonal of X·Xᵀ (a chain per tile: tiles this wave does not have are masked on the VALU)
    f32x4 g2[GW_TPW][MT];
#pragma unroll
    for (int ti = 0; ti < GW_TPW; ti++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          const s16x8 f = __builtin_bit_cast(s16x8, xf[ti][j][mt]);
          if (j == 0) DT::mfma0(g2[ti][mt], f, f);
          else DT::mfma(g2[ti][mt], f, f);
        }
    VRA_MFMA_DRAIN();
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      float v = 0.f;
#pragma unroll
      for (int ti = 0; ti < GW_TPW; ti++) {
        const float tmask = wave + GW_WAVES * ti < KT ? 1.0f : 0.0f;
        const f32x4 g = g2[ti][mt];
        const float d01 = (nn & 1) ? g[1] : g[0], d23 = (nn & 1) ? g[3] : g[2];
        v = fmaf((nn & 2) ? d23 : d01, tmask, v);
      }
      if (oct == (nn >> 2)) part[wave * 32 + mt * 16 + nn] = v;
    }
    __syncthreads();
    float rstd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < GW_WAVES; w++) tot += part[w * 32 + mt * 16 + nn];
      rstd[mt] = 1.0f / sqrtf(tot / (float)K + a.eps);
    }
#pragma unroll
    for (int ti = 0; ti < GW_TPW; ti++) {
      const int kt = min(wave + GW_WAVES * ti, KT - 1);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float g[8];
        unpack8<DT>(*reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.norm_w) + kt * 128 + j * 32 + oct * 8), g);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          float f[8];
          unpack8<DT>(xf[ti][j][mt], f);
#pragma unroll
          for (int e = 0; e < 8; e++) f[e] = f[e] * rstd[mt] * g[e];
          xf[ti][j][mt] = pack8<DT>(f);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- main loop: one unit = four tile-steps
  for (int ui = 0; ui < nu; ui++) {
    f32x4 acc[MT];
#pragma unroll
    for (int ti = 0; ti < GW_TPW; ti++) {
      const uint32_t vmask = wave + GW_WAVES * ti < KT ? 0xffffffffu : 0u;  // a tile this wave does not have contributes zeros
#pragma unroll
      for (int j = 0; j < 4; j++) {
        u32x4 wv = wb[ti % GDW_RING][j];
#pragma unroll
        for (int c = 0; c < 4; c++) wv[c] &= vmask;
        const s16x8 bf = __builtin_bit_cast(s16x8, wv);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          if (ti == 0 && j == 0) DT::mfma0(acc[mt], __builtin_bit_cast(s16x8, xf[ti][j][mt]), bf);
          else DT::mfma(acc[mt], __builtin_bit_cast(s16x8, xf[ti][j][mt]), bf);
        }
      }
      issue(ui * 4 + ti + GDW_RING, wb[ti % GDW_RING]);
    }
    VRA_MFMA_DRAIN();
    // ---- the unit's partial tiles meet in LDS (double-buffered by unit parity: one barrier per unit)
    f32x4* rbuf = red + (size_t)(ui & 1) * GW_WAVES * MT * 64;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) rbuf[(wave * MT + mt) * 64 + lane] = acc[mt];
    __syncthreads();
    if (tid < OPU) {
      const int row = tid >> 4, col = tid & 15, mt = row >> 4;
      const int dl = (((row & 15) >> 2) << 4) + col, r = row & 3;  // D layout: lane = (row/4)*16 + column, register = row % 4
      const float* rf = reinterpret_cast<const float*>(rbuf);
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < GW_WAVES; w++) v += rf[(((w * MT + mt) * 64) + dl) * 4 + r];
      outs[ui * OPU + tid] = v;
    }
  }
  __syncthreads();
  // ---- everything is stored after the stream has ended (bias and the roundings of the other dense kernels: the sum is rounded
  // to the model dtype, the bias added, rounded again; an f32 output carries the same value widened)
  // (OPU divides GW_THREADS: a thread holds the same row / column of every unit it stores — its argmax candidate is a register)
  unsigned long long am_best = 0ull;
  for (int idx = tid; idx < nu * OPU; idx += GW_THREADS) {
    const int ui = idx / OPU, rem = idx - ui * OPU, row = rem >> 4, col = rem & 15;
    if (row >= M) continue;
    const int n = (u0 + ui) * 16 + col;
    float v = rnd_dt<DT>(outs[idx]);
    if (a.bias) v = rnd_dt<DT>(v + DT::to_f32(static_cast<const uint16_t*>(a.bias)[n]));
    if (a.am_out) am_best = gemv_am_max(am_best, gemv_am_key(v, (uint32_t)n));
    if (a.out_f32) static_cast<float*>(a.out)[(size_t)row * a.out_ld + n] = v;
    else static_cast<uint16_t*>(a.out)[(size_t)row * a.out_ld + n] = DT::from_f32(v);
  }
  if (a.am_out) {  // uniform.  The ordering argument is gemv.cuh's (agent-scope candidate stores acknowledged at vmcnt(0), a
                   // returning counter RMW); `part` (the norm prologue's table) is free by now
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(part);  // [32 groups of 16 threads]
    uint32_t* flag = reinterpret_cast<uint32_t*>(cand + 32);
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) am_best = gemv_am_max(am_best, gemv_am_shfl_xor(am_best, d));
    if ((tid & 15) == 0) cand[tid >> 4] = am_best;
    __syncthreads();
    if (tid < M) {  // MT = 2: group g holds row g; MT = 1: groups g and g + 16 hold row g (threads t and t + 256 store alternate units)
      const unsigned long long b = MT == 2 ? cand[tid] : gemv_am_max(cand[tid], cand[tid + 16]);
      __hip_atomic_store(a.am_ws + (size_t)tid * gridDim.x + blockIdx.x, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const uint32_t old = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(a.am_ws + GEMV_AM_COUNTER), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = old == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (*flag) {  // the last workgroup to arrive: wave w reduces rows w, w + 8, ..
      for (int row = wave; row < M; row += GW_WAVES) {
        unsigned long long b = 0ull;
        for (int c = lane; c < (int)gridDim.x; c += 64)
          b = gemv_am_max(b, __hip_atomic_load(a.am_ws + (size_t)row * gridDim.x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) b = gemv_am_max(b, gemv_am_shfl_xor(b, d));
        if (lane == 0) a.am_out[row] = 0xffffffffu - (uint32_t)b;
      }
      if (tid == 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(a.am_ws + GEMV_AM_COUNTER), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
