// gemm_q4_big.cuh — "kernel D": int4 GEMM for many activation rows (prefill), register blocked.
//
// Roofline: MFMA.  Kernels B/C give every wave ONE n-block (16 output columns), so every MFMA needs a fresh x fragment
// from LDS (`ds_read_b128`, 1 KiB per wave): at 64+ rows kernel B is LDS-bandwidth bound at ~22 % of the MFMA peak
// (measured: 560 TFLOP/s at M = 4096), and its 64 x 128 workgroup tile re-reads x and the weights from L2 at 87 FLOP/B.
// Here a wave owns GD_NB = 4 n-blocks x MB m-tiles:
//   * every x fragment read from LDS feeds 4 MFMAs (LDS traffic per MFMA / 4);
//   * workgroup = 4 waves side by side in n: tile 256 columns x 16*MB rows, TWO workgroups co-resident per CU (<= 256 VGPRs,
//     52 KB of LDS each at MB = 4).  Round 1 ran 8 waves as 4 (n) x 2 (m) under one barrier per k-tile: the phase timers
//     (tools/gemm_big_ts.py) showed the two waves of a SIMD serialised — both start their VALU phase (dequant) together after
//     the barrier, one wins the VALU, and while the loser runs its MFMA phase the winner already idles at the barrier
//     (1840 of 5300 cycles per k-tile).  Two independent 4-wave workgroups drift apart instead and one's MFMA phase covers
//     the other's VALU phase: M = 4096 layer 2299 -> 2036 us, M = 2048 1206 -> 1052 us.  Each workgroup fetches its own copy
//     of the weights from L2 (117 FLOP per L2 byte at MB = 4 instead of 175): far below the L2 rate;
//   * the 4 x 4 weight fragments of a k-tile are dequantised ONCE per k-tile ((C + q) "magic" words, wna16.cuh: 7 VALU
//     ops per 8 weights, amortised over MB m-tiles) and kept in registers; the exact per-group fix-up
//     acc += s·(acc_g − (C + z)·Σx)  runs per m-tile (16 group-accumulator registers instead of 64);
//   * Σx of every (row, k-tile) comes from a table computed once per GEMM by `xsum_rows_kernel` (every one of the
//     16..112 workgroups of an m-tile would otherwise redo the same DPP reductions while staging);
//   * every global access is a buffer instruction with a wave-uniform (SGPR) offset and ONE shared VGPR offset per
//     stream — 64-bit per-lane addresses for 4 weight, 4 scale and MB x streams cost ~30 VGPRs the tile needs.
// Operand roles as in kernel B: A = weights (16 columns x 32 k), B = x (16 rows x 32 k), D[column][row].
// x chunks (16*MB rows x 128 k) are staged through LDS, triple buffered, one workgroup barrier per k-tile
// (16*MB MFMAs per wave between barriers).  Scale groups of >= 128 (or per channel) only; row-major scales.
// DUAL: a wave owns 2 gate + 2 up n-blocks of the same columns and the epilogue is silu(gate)·up (mlp.rs:451-469).
//
// Round 6 — the zero-point term leaves the k loop (ZH: GPTQ symmetric + bf16, the headline configuration).  With z = 8 for every
// group the fix-up  acc += s_g·(acc_g − (C + 8)·Σx_g)  splits into  acc += s_g·acc_g  in the loop and ONE correction at the end,
//     acc −= (C + 8) · Σ_g s_g[n] · Σx_g[m],
// which is itself a small GEMM over the k-tiles — S [16 columns x k-tiles] times Σx [k-tiles x 16 rows] — and runs on the matrix
// core: the scales are 16-bit floats already, the f32 row sums go in as three bf16 pieces (hi + mid + lo = 24 bits), 3 MFMAs per
// 32 k-tiles and (n-block, m-tile) against the 128 of the main loop.  Per k-tile and wave that removes half of the fix-up (32 of 64
// v_pk_fma_f32 at MB = 4) and the MB row-sum loads.  The kernel is bound by the SUM of its VALU and MFMA cycles (tools/mfma_valu_probe.hip),
// so the removed VALU work is time.  Same exact arithmetic contract, another f32 summation order (<= 1 output ulp against the oracle as before).
// AWQ (per-group zero points: the correction's A operand (C + z_g)·s_g is not a 16-bit float) and f16 (row sums beyond its range)
// keep the in-loop form.
#pragma once
#include "gemv.cuh"  // GemvSeg
#include "wna16.cuh"

#ifndef GD_WM
#define GD_WM 1  // waves along m per workgroup (workgroup = 4 (n) x GD_WM (m) waves)
#endif
#define GD_THREADS (256 * GD_WM)
#define GD_NB 4
#ifndef GD_PK_FIXUP
#define GD_PK_FIXUP 1
#endif
#ifndef GD_XDEPTH
#define GD_XDEPTH 2
#endif
#ifndef GD_ZHOIST
#define GD_ZHOIST 1
#endif

struct GemmDArgs {
  const void* w0;  // int4 tiled
  const void* w1;  // DUAL: up tensor
  const void* sc0;
  const void* sc1;
  const uint32_t* qz0;
  const uint32_t* qz1;
  const void* bias0;
  const void* bias1;
  const void* x;  // [M, x_ld]
  int x_ld;
  const void* residual;
  int res_ld;
  void* out;  // [M, out_ld]
  int out_ld;
  int M, N, K;
  int group_size, out_f32;
  // more tensors with the same x in ONE launch (q/k/v): segment 0 is w0/sc0/qz0/bias0/out/N above; every segment's
  // column count is a multiple of 64 (a wave's 4 n-blocks never straddle tensors); not with DUAL / residual
  int nseg;
  GemvSeg xseg[2];
  const float* xsum;  // [M][K/128] row sums of x per k-tile (xsum_rows_kernel), set by the launcher
  unsigned long long* ts;  // -DVRA_GEMV_TS builds: per-workgroup phase cycle sums (tools/gemm_big_ts.py)
  // split-K (gridDim.z slices; short prefills where the output tiles alone do not fill the chip): the exchange of kernels
  // B/C — write-through fp32 slabs, one flag line per slice, the last slice reduces in fixed order
  int splitk;
  float* slabs;
  uint32_t* counters;
  uint32_t* err;
};
#ifdef VRA_GEMV_TS
#define GD_STAMP(v)                         \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    v = (long long)__builtin_readcyclecounter(); \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)
#else
#define GD_STAMP(v) \
  do {              \
  } while (0)
#endif

static inline size_t gemm_q4_big_lds_bytes(int mb) { return (size_t)3 * (16 * GD_WM * mb) * (16 + 1) * 16; }

// Σ over each (row, k-tile) of x, from the 16-bit values the MFMA sees: 16 lanes per (row, k-tile), one octet each.
template <class DT>
__global__ __launch_bounds__(256) void xsum_rows_kernel(const void* x, int x_ld, int M, int KT, float* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)M * KT * 16;
  const int64_t ic = i < total ? i : total - 1;  // every lane takes part in the DPP sums
  const int o = (int)(ic & 15);
  const int64_t rt = ic >> 4;
  const int m = (int)(rt / KT), kt = (int)(rt - (int64_t)m * KT);
  const u32x4 v = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(x) + (size_t)m * x_ld + (size_t)kt * 128 + o * 8);
  const float s = row16_sum(octet_sum<DT>(v));
  if (o == 0 && i < total) out[rt] = s;
}

template <class DT, bool DUAL, bool AWQ, int MB>
__global__ __launch_bounds__(GD_THREADS, 2) void gemm_q4_big_kernel(const GemmDArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NB = GD_NB;
  constexpr bool ZH = GD_ZHOIST && !AWQ && std::is_same<DT, BF16>::value;  // zero-point term hoisted out of the k loop (header)
  constexpr int ROWS = 16 * GD_WM * MB;  // rows of x per workgroup
  constexpr int RPP = GD_THREADS / 16;   // rows one pass of the workgroup's threads stages
  constexpr int RS = (16 + 1) * 4;    // LDS row stride in u32: 16 octets + one of padding (see gemm_skinny.cuh)
  constexpr int XS_U32 = ROWS * RS;   // one x buffer
  constexpr uint32_t RSRC3 = 0x00020000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 15, oct = lane >> 4;
  const int wn = wave & 3, wm = wave >> 2;
  const int K = a.K, M = a.M, KT = K >> 7;
  const bool grouped = a.group_size > 0 && a.group_size < K;
  const int gsh = grouped ? 31 - __builtin_clz(a.group_size) : 31;
  const int m0 = (int)blockIdx.y * ROWS;

  uint32_t* xs = reinterpret_cast<uint32_t*>(smem);  // [3][ROWS][RS]

  // ---- this wave's tensor(s) and n-blocks (wave-uniform)
  const void* wt[2] = {a.w0, a.w1};
  const void* sct[2] = {a.sc0, a.sc1};
  const uint32_t* qzt[2] = {a.qz0, a.qz1};
  const void* biast[2] = {a.bias0, a.bias1};
  void* outp = a.out;
  int N = a.N, out_ld = a.out_ld;
  int nb0;  // first n-block of the wave within its tensor
  if (DUAL) {
    nb0 = (int)blockIdx.x * 8 + wn * 2;
  } else {
    nb0 = (int)blockIdx.x * 16 + wn * 4;
    if (a.nseg > 1) {
      const int s = (a.nseg > 2 && nb0 >= a.xseg[1].blk_start) ? 1 : (nb0 >= a.xseg[0].blk_start ? 0 : -1);
      if (s >= 0) {
        const GemvSeg& sg = a.xseg[s];
        wt[0] = sg.w, sct[0] = sg.scales, qzt[0] = sg.qzeros, biast[0] = sg.bias, outp = sg.out, N = sg.n, out_ld = sg.out_ld;
        nb0 -= sg.blk_start;
      }
    }
  }
  // n-block b of the wave: tensor tb(b), block nbv(b); a missing block streams block 0 (its results are never stored)
  auto tb = [&](int b) { return DUAL ? (b >> 1) : 0; };
  auto nbv = [&](int b) { return nb0 + (DUAL ? (b & 1) : b); };
  bool ok[NB];
  int nbc[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    ok[b] = nbv(b) * 16 < N;
    nbc[b] = ok[b] ? nbv(b) : 0;
  }
  // ---- buffer resources (SGPRs) and the per-stream VGPR offsets
  constexpr int NT = DUAL ? 2 : 1;
  __amdgpu_buffer_rsrc_t rw[NT], rsc[NT], rqz[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    rw[t] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wt[t]), 0, 0x7FFFFFF0, RSRC3);
    rsc[t] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sct[t]), 0, 0x7FFFFFF0, RSRC3);
    rqz[t] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(AWQ ? qzt[t] : reinterpret_cast<const uint32_t*>(sct[t])), 0, 0x7FFFFFF0, RSRC3);
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, 0x7FFFFFF0, RSRC3);
  const __amdgpu_buffer_rsrc_t rsum = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xsum), 0, 0x7FFFFFF0, RSRC3);
  const uint32_t vo_w = (uint32_t)lane * 16u;                                        // weights: lane's 16 bytes of a tile
  const uint32_t vo_s = (uint32_t)oct * 8u;                                          // scales: the lane's 4 output columns
  const uint32_t vo_z = (uint32_t)(oct >> 1) * 4u;                                   // awq zero word of those columns
  const int Nt = DUAL ? a.N : N;
  constexpr int XPT = ROWS * 16 / GD_THREADS;  // octets per thread per k-tile (= MB)
  // x: rows m0 + r*32 + tid/16, octet tid%16; Σx: the lane's row of every m-tile.  Rows >= M alias row M-1 (never stored).
  uint32_t vo_x[XPT], vo_sum[MB];
#pragma unroll
  for (int r = 0; r < XPT; r++) vo_x[r] = ((uint32_t)min(m0 + r * RPP + (tid >> 4), M - 1) * (uint32_t)a.x_ld + (uint32_t)(tid & 15) * 8u) * 2u;
#pragma unroll
  for (int mt = 0; mt < MB; mt++) vo_sum[mt] = (uint32_t)min(m0 + wm * (MB * 16) + mt * 16 + nn, M - 1) * (uint32_t)KT * 4u;
  // x tiles go global -> registers at the START of an iteration and registers -> LDS at its END, two k-tiles ahead of
  // their use (three LDS buffers): a whole iteration of MFMAs hides the load, and the buffer written is the one read in
  // the PREVIOUS iteration (every wave has passed that iteration's barrier)
  auto x_load = [&](int kt, u32x4 (&xr)[XPT]) {
#pragma unroll
    for (int r = 0; r < XPT; r++) xr[r] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo_x[r], (uint32_t)(kt * 256), 0);
  };
  auto x_store = [&](int buf, const u32x4 (&xr)[XPT]) {
    uint32_t* dst = xs + (size_t)buf * XS_U32 + (size_t)(tid >> 4) * RS + (tid & 15) * 4;
#pragma unroll
    for (int r = 0; r < XPT; r++) *reinterpret_cast<u32x4*>(dst + (size_t)(r * RPP) * RS) = xr[r];
  };
  // weight stream: one 16-byte word per (n-block, k-tile) and lane; scales / zero points in the MFMA OUTPUT layout
  auto w_load = [&](int kt, u32x4 (&wq)[NB], u32x2 (&sc)[NB], uint32_t (&zw)[NB]) {
    const int grp = grouped ? (kt * 128) >> gsh : 0;
#pragma unroll
    for (int b = 0; b < NB; b++) {
      wq[b] = __builtin_amdgcn_raw_buffer_load_b128(rw[tb(b)], vo_w, (uint32_t)((nbc[b] * KT + kt) * 1024), 2);  // nt
      sc[b] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsc[tb(b)], vo_s, (uint32_t)((grp * Nt + nbc[b] * 16) * 2), 0));
      zw[b] = AWQ ? __builtin_amdgcn_raw_buffer_load_b32(rqz[tb(b)], vo_z, (uint32_t)((grp * (Nt >> 3) + nbc[b] * 2) * 4), 0) : 0x88888888u;
    }
  };
  auto sum_load = [&](int kt, float (&sx)[MB]) {
    if constexpr (ZH) return;  // (the row sums are read once, behind the loop)
#pragma unroll
    for (int mt = 0; mt < MB; mt++)
      sx[mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsum, vo_sum[mt], (uint32_t)(kt * 4), 0));
  };

  f32x2 acc[NB][MB][2];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int t = 0; t < MB; t++) acc[b][t][0] = acc[b][t][1] = f32x2{0.f, 0.f};
  // this workgroup's k-tiles: all of them, or slice blockIdx.z of a.splitk
  const int SK = a.splitk > 1 ? a.splitk : 1, zi = (int)blockIdx.z;
  const int kt0 = (int)((long)KT * zi / SK), kt1 = (int)((long)KT * (zi + 1) / SK);

  u32x4 wq[NB];
  u32x2 scw[NB];
  uint32_t zw[NB];
  float sxn[MB] = {};
  {
    u32x4 x0[XPT], x1[XPT];
    x_load(kt0, x0);
    x_load(min(kt0 + 1, kt1 - 1), x1);
    w_load(kt0, wq, scw, zw);
    sum_load(kt0, sxn);
    x_store(0, x0);
    x_store(1, x1);
  }
  __syncthreads();

  long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, sB = 0, sC = 0, sD = 0, sE = 0;
  (void)tA, (void)tB, (void)tC, (void)tD, (void)tE, (void)sB, (void)sC, (void)sD, (void)sE;
  for (int kt = kt0; kt < kt1; kt++) {
    const int buf = (kt - kt0) % 3;
    GD_STAMP(tA);
    // ---- the k-tile's 4 x 4 weight fragments (C + q), its scales and row sums; then the loads of the next k-tile
    s16x8 af[NB][4];
    f32x2 sc2[NB][2], nzc2[NB][2];
    float sxc[MB];
#pragma unroll
    for (int b = 0; b < NB; b++) {
#pragma unroll
      for (int j = 0; j < 4; j++) af[b][j] = magic_word<DT>(wq[b][j]);
      float s4[4], z4[4];
      unpack_scale4<DT>(scw[b], zw[b], nbc[b] * 16 + oct * 4, s4, z4);
      // (a missing n-block streams block 0 with block 0's scales: its accumulators are never stored — no masking here)
      sc2[b][0] = f32x2{s4[0], s4[1]}, sc2[b][1] = f32x2{s4[2], s4[3]};
      nzc2[b][0] = f32x2{-z4[0], -z4[1]}, nzc2[b][1] = f32x2{-z4[2], -z4[3]};
    }
#pragma unroll
    for (int mt = 0; mt < MB; mt++) sxc[mt] = sxn[mt];
    const int ktn = min(kt + 1, kt1 - 1);  // the last iteration re-loads its own tile (never consumed)
    u32x4 xr[XPT];
    x_load(min(kt + 2, kt1 - 1), xr);
    w_load(ktn, wq, scw, zw);
    sum_load(ktn, sxn);
    GD_STAMP(tB);
    const uint32_t* xb = xs + (size_t)buf * XS_U32 + (size_t)(wm * (MB * 16) + nn) * RS + oct * 4;
    // 4*MB steps (m-tile, j): the x fragment of step s+1 is read from LDS BEFORE the MFMAs of step s are issued (the
    // chain ds_read -> wait -> 4 MFMAs per step left the matrix pipe idle for the LDS latency: 3.4 us per k-tile)
    auto frag = [&](int s) { return *reinterpret_cast<const u32x4*>(xb + (size_t)((s >> 2) * 16) * RS + (s & 3) * 16); };
    constexpr int XD = GD_XDEPTH;  // x fragments in flight ahead of the MFMAs that consume them
    u32x4 xv[XD + 1];
#pragma unroll
    for (int s = 0; s < XD; s++) xv[s] = frag(s);
    f32x4 ag[NB];
#pragma unroll
    for (int s = 0; s < 4 * MB; s++) {
      const int mt = s >> 2, j = s & 3;
      if (s + XD < 4 * MB) xv[(s + XD) % (XD + 1)] = frag(s + XD);
      __builtin_amdgcn_sched_barrier(0);
      const s16x8 bfrag = __builtin_bit_cast(s16x8, xv[s % (XD + 1)]);
#pragma unroll
      for (int b = 0; b < NB; b++) {
        if (j == 0) DT::mfma0(ag[b], af[b][j], bfrag);
        else DT::mfma(ag[b], af[b][j], bfrag);
      }
      if (j == 3) {
        VRA_MFMA_DRAIN();
#if GD_PK_FIXUP
        if constexpr (ZH) {
#pragma unroll
          for (int b = 0; b < NB; b++) {
            const f32x2 g0 = {ag[b][0], ag[b][1]}, g1 = {ag[b][2], ag[b][3]};
            acc[b][mt][0] = __builtin_elementwise_fma(sc2[b][0], g0, acc[b][mt][0]);
            acc[b][mt][1] = __builtin_elementwise_fma(sc2[b][1], g1, acc[b][mt][1]);
          }
        } else {
        const f32x2 sx2 = {sxc[mt], sxc[mt]};
#pragma unroll
        for (int b = 0; b < NB; b++) {
          const f32x2 g0 = {ag[b][0], ag[b][1]}, g1 = {ag[b][2], ag[b][3]};
          acc[b][mt][0] = __builtin_elementwise_fma(sc2[b][0], __builtin_elementwise_fma(nzc2[b][0], sx2, g0), acc[b][mt][0]);
          acc[b][mt][1] = __builtin_elementwise_fma(sc2[b][1], __builtin_elementwise_fma(nzc2[b][1], sx2, g1), acc[b][mt][1]);
        }
        }
#else
        // scalar v_fma_f32 (2 cycles each), not v_pk_fma_f32: packed f32 VALU beside MFMAs costs more than its two halves
        // (MI355X_MICROARCH.md, per-instruction constants).  The asm keeps the SLP vectoriser from re-packing the pairs.
#pragma unroll
        for (int b = 0; b < NB; b++) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            float t = ZH ? ag[b][r] : __builtin_fmaf(nzc2[b][r >> 1][r & 1], sxc[mt], ag[b][r]);
            asm volatile("" : "+v"(t));
            float u = __builtin_fmaf(sc2[b][r >> 1][r & 1], t, acc[b][mt][r >> 1][r & 1]);
            asm volatile("" : "+v"(u));
            acc[b][mt][r >> 1][r & 1] = u;
          }
        }
#endif
      }
    }
    GD_STAMP(tC);
    x_store((kt - kt0 + 2) % 3, xr);
    GD_STAMP(tD);
    __syncthreads();
    GD_STAMP(tE);
    sB += tB - tA, sC += tC - tB, sD += tD - tC, sE += tE - tD;
  }
#ifdef VRA_GEMV_TS
  if (a.ts && lane == 0 && (int)(blockIdx.y * gridDim.x + blockIdx.x) < 4096) {
    unsigned long long* t = a.ts + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + wave) * 4;
    t[0] = (unsigned long long)sB, t[1] = (unsigned long long)sC, t[2] = (unsigned long long)sD, t[3] = (unsigned long long)sE;
  }
#endif

  // ---- ZH: the zero-point term of this workgroup's k-tiles [kt0, kt1), 32 k-tiles per MFMA step (header).
  //   A (scales): lane (column nn, octet oct) holds s[grp(t)][column] for the 8 tiles t = t0 + oct*8 .. +7 (tiles >= kt1: zero);
  //   B (row sums): lane (row nn of the m-tile, octet oct) holds the 8 sums of those tiles as hi / mid / lo bf16 pieces.
  if constexpr (ZH) {
    const uint32_t vo_sn = (uint32_t)nn * 2u;  // the lane's column within the n-block (16-bit scales)
    for (int t0 = kt0; t0 < kt1; t0 += 32) {
      const int tb0 = t0 + oct * 8;  // this lane's first tile
      s16x8 sa[NB];
#pragma unroll
      for (int b = 0; b < NB; b++) {
        uint32_t sv[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const int t = min(tb0 + e, kt1 - 1);
          const int grp = grouped ? (t * 128) >> gsh : 0;
          sv[e] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rsc[tb(b)], vo_sn, (uint32_t)((grp * Nt + nbc[b] * 16) * 2), 0);
        }
        u32x4 w;
#pragma unroll
        for (int i = 0; i < 4; i++) w[i] = (tb0 + 2 * i < kt1 ? sv[2 * i] : 0u) | ((tb0 + 2 * i + 1 < kt1 ? sv[2 * i + 1] : 0u) << 16);
        sa[b] = __builtin_bit_cast(s16x8, w);
      }
#pragma unroll
      for (int mt = 0; mt < MB; mt++) {
        // the 8 sums of this lane's tiles (index clamped to the slice: tiles >= kt1 meet zero scales in the A operand)
        float sxv[8];
#pragma unroll
        for (int e = 0; e < 8; e++)
          sxv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsum, vo_sum[mt], (uint32_t)(min(tb0 + e, kt1 - 1) * 4), 0));
        u32x4 ph, pm, pl;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          uint32_t hw[2], mw[2], lw[2];
#pragma unroll
          for (int c = 0; c < 2; c++) {
            const float v = sxv[2 * i + c];
            const uint32_t hb = __float_as_uint(v) & 0xffff0000u;  // top 8 significant bits (truncated: v - hi is exact)
            const float r1 = v - __uint_as_float(hb);
            const uint32_t mb_ = __float_as_uint(r1) & 0xffff0000u;
            const float r2 = r1 - __uint_as_float(mb_);
            hw[c] = hb >> 16, mw[c] = mb_ >> 16, lw[c] = (uint32_t)BF16::from_f32(r2);
          }
          ph[i] = hw[0] | (hw[1] << 16), pm[i] = mw[0] | (mw[1] << 16), pl[i] = lw[0] | (lw[1] << 16);
        }
        f32x4 zt[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) {
          BF16::mfma0(zt[b], sa[b], __builtin_bit_cast(s16x8, pl));  // smallest pieces first
          BF16::mfma(zt[b], sa[b], __builtin_bit_cast(s16x8, pm));
          BF16::mfma(zt[b], sa[b], __builtin_bit_cast(s16x8, ph));
        }
        VRA_MFMA_DRAIN();
        constexpr float NZC = -(Magic<BF16>::bias + 8.0f);
#pragma unroll
        for (int b = 0; b < NB; b++) {
          acc[b][mt][0] = __builtin_elementwise_fma(f32x2{NZC, NZC}, f32x2{zt[b][0], zt[b][1]}, acc[b][mt][0]);
          acc[b][mt][1] = __builtin_elementwise_fma(f32x2{NZC, NZC}, f32x2{zt[b][2], zt[b][3]}, acc[b][mt][1]);
        }
      }
    }
  }

  // ---- split-K: the slices of a tile meet through memory (see gemm_skinny.cuh): write-through 16-byte stores, one flag
  // line per slice, the last slice ("owner") polls, adds the slabs to its own partial in slice order and resets the flags
  if (SK > 1) {
    const int tile = (int)(blockIdx.y * gridDim.x + blockIdx.x), ntiles = (int)(gridDim.x * gridDim.y);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, 0x7FFFFFF0, RSRC3);
    auto slab_off = [&](int z, int b, int mt) {  // bytes: [slice][tile][n-block][m-tile][thread] x 16 B
      return (uint32_t)((((z * ntiles + tile) * NB + b) * MB + mt) * GD_THREADS + tid) * 16u;
    };
    uint32_t* fl = a.counters + (size_t)tile * SK * 16;
    if (zi != SK - 1) {
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int mt = 0; mt < MB; mt++) {
          const u32x4 v = {__float_as_uint(acc[b][mt][0][0]), __float_as_uint(acc[b][mt][0][1]), __float_as_uint(acc[b][mt][1][0]),
                           __float_as_uint(acc[b][mt][1][1])};
          __builtin_amdgcn_raw_buffer_store_b128(v, srs, slab_off(zi, b, mt), 0, 16);
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-through stores: acknowledged by memory
      __syncthreads();
      if (tid == 0) __hip_atomic_store(fl + zi * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    if (tid < SK - 1) {
      const uint64_t t0 = __builtin_readcyclecounter();
      while (__hip_atomic_load(fl + tid * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_readcyclecounter() - t0 > (1ull << 31)) {  // never hang the device on a lost slice
          __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __hip_atomic_store(fl + tid * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int z = 0; z < SK - 1; z++) {  // one slice per round trip: NB*MB 16-byte loads in flight per lane; fixed order
      u32x4 p[NB][MB];
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int mt = 0; mt < MB; mt++) p[b][mt] = __builtin_amdgcn_raw_buffer_load_b128(srs, slab_off(z, b, mt), 0, 16);
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int mt = 0; mt < MB; mt++) {
          acc[b][mt][0] += f32x2{__uint_as_float(p[b][mt][0]), __uint_as_float(p[b][mt][1])};
          acc[b][mt][1] += f32x2{__uint_as_float(p[b][mt][2]), __uint_as_float(p[b][mt][3])};
        }
    }
  }

  // ---- epilogue: D[column (lane>>4)*4 + r][row lane&15] of tile (b, mt)
  constexpr int NBO = DUAL ? 2 : NB;  // output n-blocks of the wave
#pragma unroll
  for (int b = 0; b < NBO; b++) {
    if (!ok[b]) continue;
    const int n = nbv(b) * 16 + oct * 4;
    float bs[4] = {0.f, 0.f, 0.f, 0.f}, bs2[4] = {0.f, 0.f, 0.f, 0.f};
    if (biast[0]) {
      const u32x2 bw = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(biast[0]) + n);
      bs[0] = DT::to_f32((uint16_t)(bw[0] & 0xffffu)), bs[1] = DT::to_f32((uint16_t)(bw[0] >> 16));
      bs[2] = DT::to_f32((uint16_t)(bw[1] & 0xffffu)), bs[3] = DT::to_f32((uint16_t)(bw[1] >> 16));
    }
    if (DUAL && biast[1]) {
      const u32x2 bw = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(biast[1]) + n);
      bs2[0] = DT::to_f32((uint16_t)(bw[0] & 0xffffu)), bs2[1] = DT::to_f32((uint16_t)(bw[0] >> 16));
      bs2[2] = DT::to_f32((uint16_t)(bw[1] & 0xffffu)), bs2[3] = DT::to_f32((uint16_t)(bw[1] >> 16));
    }
#pragma unroll
    for (int mt = 0; mt < MB; mt++) {
      const int m = m0 + wm * (MB * 16) + mt * 16 + nn;
      if (m >= M) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float t = rnd_dt<DT>(acc[b][mt][r >> 1][r & 1]);
        if (biast[0]) t = rnd_dt<DT>(t + bs[r]);
        if (DUAL) {
          float u = rnd_dt<DT>(acc[b + 2][mt][r >> 1][r & 1]);
          if (biast[1]) u = rnd_dt<DT>(u + bs2[r]);
          const float sl = rnd_dt<DT>(t / (1.0f + expf(-t)));
          t = sl * u;
        }
        v[r] = t;
      }
      if (a.residual) {
        const u32x2 rw2 = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(a.residual) + (size_t)m * a.res_ld + n);
        v[0] = rnd_dt<DT>(v[0]) + DT::to_f32((uint16_t)(rw2[0] & 0xffffu));
        v[1] = rnd_dt<DT>(v[1]) + DT::to_f32((uint16_t)(rw2[0] >> 16));
        v[2] = rnd_dt<DT>(v[2]) + DT::to_f32((uint16_t)(rw2[1] & 0xffffu));
        v[3] = rnd_dt<DT>(v[3]) + DT::to_f32((uint16_t)(rw2[1] >> 16));
      }
      if (a.out_f32) {
        const f32x4 o = {rnd_dt<DT>(v[0]), rnd_dt<DT>(v[1]), rnd_dt<DT>(v[2]), rnd_dt<DT>(v[3])};
        *reinterpret_cast<f32x4*>(static_cast<float*>(outp) + (size_t)m * out_ld + n) = o;
      } else {
        const u32x2 o = {DT::pack2(v[0], v[1]), DT::pack2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(static_cast<uint16_t*>(outp) + (size_t)m * out_ld + n) = o;
      }
    }
  }
}
