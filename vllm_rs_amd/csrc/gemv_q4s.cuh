// gemv_q4s.cuh — "kernel E": int4 GEMV for decode batches of 1..4 rows (the bs = 1 headline), built around what the
// round-1 timelines of kernel A showed: at these sizes a launch is dominated by its fixed costs (three workgroup barriers
// and two passes over LDS before the first MFMA, a barrier per work item, an epilogue wave that meets the compute waves
// per item), not by streaming.
//
// Roofline: HBM.  Algorithmic bytes per call: K*N/2 (packed int4) + (K/g)*N*2 (scales) [+ (K/g)*N/2 AWQ zeros]
// + M*K*2 + M*N*2.
//
// Shape of the launch:
//   * ONE workgroup of 16 waves per CU (grid <= #CUs), no persistence loop.  The unit of work is one 16-column n-block
//     over the full K ("unit"; NS = 2: the same n-block of the gate and of the up tensor).  Workgroup b owns a contiguous
//     range of units — a contiguous run of the packed weights, (units x K/128) KiB — and every wave w takes the k-tiles
//     w, w+16, ... of each unit: 1 KiB coalesced per tile-step, 2 KiB in flight per wave (the depth that measured best on
//     this memory system), branch-free issue so that hipcc's s_waitcnt stays at vmcnt(ring-1).
//   * x never crosses waves: a wave needs only the x slices of ITS k-tiles.  It loads them itself (one 16-byte load per lane
//     covers 4 rows x 128 columns), parks them in a wave-private LDS region (row-major, one octet of padding per row: the
//     MFMA A-fragment reads of 4 rows are 4 banks apart) together with the per-tile sums Σx of the zero-point fix-up.  With
//     a fused RMSNorm there is NO cross-wave step either (round 5): a wave stages round(x · g) and leaves the partial Σx² of its
//     slices for the end-of-stream reduction; rstd is applied to the f32 dot products in the epilogue (it commutes with the GEMV).
//   * partial tiles of all units meet in LDS ONCE, behind the single barrier at the end of the stream; the first
//     units*64 threads then add the 16 wave partials in fixed order and run the fused epilogue (bias, SiLU·mul, residual);
//     bias / residual were requested before the weight stream started.
//   * scales / AWQ zeros are addressed as  grp*grp_stride + unit*unit_stride + column: row-major checkpoint tensors
//     ([K/g, N]: strides N, 16), the unit-major copies the native runtime makes at load ([N/16][K/g][16]: strides 16,
//     16*K/g — the scales of a workgroup's stream are then one contiguous run, like its weights) and the Marlin-permuted
//     scales the reference passes (wna16.rs:180-218: a permutation inside 64 columns, i.e. a per-lane index) all work.
#pragma once
#include <type_traits>

#include "gemv.cuh"

#define GS_WAVES 16
#define GS_THREADS (GS_WAVES * 64)
#define GS_MAX_UNITS 8   // units (pairs) per workgroup at most
#define GS_MAX_TPW 10    // k-tiles per wave and unit at most (K <= 20480: Qwen2-7B's down projection, K = 18944, is 148 tiles)
#define GS_NORM_TPW 4    // ... with a fused RMSNorm (the norm weights of a wave's tiles stay in registers)
#define GS_MAX_SEG 3
#ifndef GS_RING_KIB
#define GS_RING_KIB 2
#endif
#define GS_TILE_LDS 1104  // bytes of a wave's LDS per k-tile: 4 row regions x (256 + 16) + 16 (Σx of the 4 rows).  When that does not
                          // fit (K > 16384) the launcher keeps only M row regions: xrows * 272 + 16 (GemvSArgs::xrows)

struct GemvSSeg {  // output segment (q / k / v of one launch): epilogue only
  void* out;
  const void* bias;  // [columns of the segment] or null
  int out_ld;
  int unit_start;  // first unit of the segment
};
struct GemvSArgs {
  const void* w[2];          // tiled weights of stream 0 (and 1: the up tensor of a gate/up pair), unit-major
  const void* scales[2];     // 16-bit
  const uint32_t* zeros[2];  // AWQ: packed zero points (8 columns per word), else null
  int s_grp_stride, s_unit_stride;  // scale element index = grp*s_grp_stride + unit*s_unit_stride + column (both even)
  int z_grp_stride, z_unit_stride;  // zero word index   = grp*z_grp_stride + unit*z_unit_stride + column/8
  int marlin;                       // scales in the reference's marlin_permute_scales order (grouped form), N % 64 == 0
  const void* x;                    // [M, K] (ld = x_ld)
  int x_ld;
  const void* norm_w;  // non-null: x <- rmsnorm(x) * norm_w
  float eps;
  const void* residual;  // [M, res_ld] added after bias
  int res_ld;
  GemvSSeg seg[GS_MAX_SEG];
  int nseg;
  int M, K, KT, TPW, gsh;  // KT = K/128, TPW = k-tiles per wave and unit at most (LDS stride of a wave's slices), k >> gsh = scale group
  // skew (round 5): 0 = wave w owns the k-tiles w, w + 16, ... of every unit (equal shares); d > 0 = CONTIGUOUS shares, the four
  // waves that start first own d tiles more and the four that start last d fewer than the middle eight.  A 16-wave workgroup's
  // waves are launched ~0.1 us apart and the four waves of a SIMD are served oldest first: with equal shares the last group
  // finishes 1.6 us (q/k/v) .. 3.8 us (gate/up) after the first (profiles/r02_timeline_kernel_e.txt), and the final barrier
  // waits for it with most of the CU's loads no longer in flight
  int skew;
  int n_units, units_q, units_r;  // workgroup b owns units_q (+1 if b < units_r) units starting at b*units_q + min(b, units_r)
  int dbg;  // VRA_EXP=1: prologue only (timeline tool)
  unsigned long long* ts;
  // kernel W, launches of up to 32 rows: x in FRAGMENT order (u32x4 word ((kt*2 + mt)*4 + j)*64 + lane = row mt*16 + nn, columns
  // kt*128 + j*32 + oct*8 .. +7 — what the producing launch left beside the row-major tensor): one contiguous KiB per wave load;
  // and the same for the outputs of a single-segment launch (the next consumer's x).  tools/xload_probe.hip: 256 KB per CU in 1.0
  // instead of 2.5 us
  const void* x_frag;
  void* out_frag;
  // kernel W, K > 4096 (down_proj of 5..32 rows): K SLICES across workgroups.  Workgroup (slice z, unit group g) = blockIdx.x
  // z * kz_groups + g holds the x fragments of k-tiles z*ktz .. (the last slice may be shorter) and streams those tiles of its
  // group's units; the f32 partial tiles of slices 0..kz-2 travel through `slabs` ([z][unit][rows x 16] f32, write-through
  // stores), every slice raises its flag (`counters`, 16 words apart, zero on entry and on exit), and the LAST slice of a group
  // — the only workgroups that wait — sums them in slice order and runs the epilogue (the exchange of kernel C, gemm_q4.cuh)
  int kz, ktz, kz_groups;
  float* slabs;
  uint32_t* counters;
  uint32_t* err;
  // kernel W, steps of 5..32 rows (round 5): READY-MADE OPERANDS for the next fused-norm launch.  Every workgroup of a norm + q/k/v
  // or norm + gate/up launch used to normalise the whole [32, K] x for itself — X.X^T MFMAs, a barrier, ~900 VALU instructions per
  // wave, two waves per SIMD: 3.5 of the 12 us of the q/k/v launch (profiles/r05_timeline_kernel_w.txt; with that work skipped a
  // bs-32 step drops from 2.69 to 2.30 ms, profiles/r05_probe_kernel_w_without_norm.txt).  rstd commutes with the GEMM, so the
  // PRODUCER of the hidden state (o_proj / down_proj epilogue: it holds the rows in LDS anyway) writes them a second time as
  // x̃ = round(h * g_next) in fragment order plus its workgroup's partial sums of squares per row, and the consumer loads finished
  // MFMA operands, adds the partial sums in a fixed order and multiplies its f32 dot products by rstd in the epilogue — the
  // order kernel E uses (gemv_q4s.cuh header), restated by the oracle's deferred variant.
  const void* pre_norm_w;  // producer: the NEXT launch's RMSNorm weights, [columns of this launch]; null: nothing of this
  void* pre_frag;          //   x̃ in fragment order (the layout of out_frag)
  float* pre_sq;           //   [GW_PRE_PARTS][32 rows] partial sums of squares, slot = this workgroup (slots never written stay 0)
  const float* x_sq;       // consumer (NORM launch whose x_frag holds x̃): the producers' partial sums, [GW_PRE_PARTS][32]
};
#define GW_PRE_PARTS 256  // slots of the partial-sum table: one per producing workgroup (grid <= CUs), unused ones zero

static inline size_t gemv_q4s_lds_bytes(int ns, int tpw, int max_units, int xrows = 4) {
  return (size_t)GS_WAVES * tpw * (xrows * 272 + 16) + 256 + (size_t)max_units * ns * GS_WAVES * 256;
}

#define GS_MIN_WAVES_PER_SIMD 4
#ifndef GS_XRING
#define GS_XRING 1
#endif
#ifndef GS_RING_PAIR
#define GS_RING_PAIR 2
#endif
// GS_RING_PAIR: ring depth (tile-steps of 2 KiB) of the gate/up pair stream (measured: 2 beats 1 by 2 % of the decode step)
// Variants that were built, parity-green, measured and REMOVED in round 3 (their numbers: DESIGN.md §3.1): two 16-wave
// workgroups per CU (GS_OCC2), rotating wave priorities, the ring issued before the x loads, waiting for x before the first
// HBM load, a tail prefetch of the next launch's first tiles, and the two-phase launches joined by a grid barrier
// (gemv_q4s2_kernel: o_proj -> gate/up and down -> next q/k/v in one launch each — break-even).  What the whole-layer form of
// that idea became is csrc/decode_step.hip.

// XR = row regions per tile in LDS (compile time: the addressing of the hot loop stays constant-folded): 4, or M = 1..3 for
// K > 16384 where four do not fit
template <class DT, int NS, bool AWQ, int XR = 4>
__device__ __forceinline__ void gemv_q4s_body(const GemvSArgs& a, unsigned char* smem) {
  constexpr int D = NS == 2 ? GS_RING_PAIR : GS_RING_KIB;  // ring depth in tile-steps (1 KiB per stream and step)
  // every kernel argument the way to the first load needs, requested in ONE batch of scalar loads
  asm volatile("" ::"s"(a.x), "s"(a.x_ld), "s"(a.norm_w), "s"(a.K), "s"(a.M), "s"(a.KT), "s"(a.TPW), "s"(a.gsh), "s"(a.units_q), "s"(a.units_r),
                 "s"(a.w[0]), "s"(a.scales[0]), "s"(a.s_grp_stride), "s"(a.s_unit_stride), "s"(a.marlin), "s"(a.residual), "s"(a.res_ld),
                 "s"(a.nseg), "s"(a.seg[0].out), "s"(a.seg[0].bias), "s"(a.seg[0].out_ld), "s"(a.seg[1].unit_start), "s"(a.seg[2].unit_start));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 15, oct = lane >> 4;
  const int M = a.M, KT = a.KT, TPW = a.TPW;
  const int wg = (int)blockIdx.x;
  const int u0 = wg * a.units_q + min(wg, a.units_r);
  const int nu = a.units_q + (wg < a.units_r ? 1 : 0);
  // this wave's share of a unit's k-tiles: CW tiles, the i-th of them k-tile  skew ? O0 + i : wave + 16*i
  const int kbase = KT >> 4, krem = KT & 15, sk = a.skew;
  const int wgrp = wave < 4 ? 1 : (wave >= 12 ? -1 : 0);
  const int CW = kbase + (wave < krem ? 1 : 0) + sk * wgrp;
  //   (first tile of a contiguous share: the shares of the waves in front of this one)
  const int O0 = wave * kbase + min(wave, krem) + sk * (wave < 4 ? wave : (wave >= 12 ? 4 - (wave - 12) : 4));
  const int S = (a.dbg & 1) ? 0 : nu * CW;  // tile-steps of this wave
  GEMV_STAMP(0);

  // ---- LDS carve-up
  constexpr int TLS = XR * 272 + 16;  // bytes per tile
  unsigned char* xw = smem + (size_t)wave * TPW * TLS;  // this wave's x slices
  float* part = reinterpret_cast<float*>(smem + (size_t)GS_WAVES * TPW * TLS);  // [16 waves][4 rows] partial Σx² (fused norm)
  f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)GS_WAVES * TPW * TLS + 256);  // [unit][NS][wave][16 lanes]

  // ---- epilogue operands of the threads that will finish the outputs (requested before the weight stream starts):
  // thread -> (unit ui, row m, column nl) for tid < nu*64
  // (one unit per wave: everything about the unit is wave-uniform and comes from SGPRs — selects, no indexed kernarg reads)
  const int e_ui = wave, e_m = oct, e_nl = nn;
  const bool e_act = e_ui < nu && e_m < M;
  const int e_unit = u0 + min(e_ui, nu - 1);
  const bool e_s1 = a.nseg > 1 && e_unit >= a.seg[1].unit_start, e_s2 = a.nseg > 2 && e_unit >= a.seg[2].unit_start;
  void* const e_out = e_s2 ? a.seg[2].out : (e_s1 ? a.seg[1].out : a.seg[0].out);
  const void* const e_biasp = e_s2 ? a.seg[2].bias : (e_s1 ? a.seg[1].bias : a.seg[0].bias);
  const int e_ld = e_s2 ? a.seg[2].out_ld : (e_s1 ? a.seg[1].out_ld : a.seg[0].out_ld);
  const int e_col = (e_unit - (e_s2 ? a.seg[2].unit_start : (e_s1 ? a.seg[1].unit_start : a.seg[0].unit_start))) * 16 + e_nl;
  const void* const e_bias2p = NS == 2 ? a.seg[1].bias : nullptr;  // pair: seg[1].bias is the up tensor's bias
  // (the 32-bit word holding the value: a 16-bit load gets its zero extension — i.e. a wait — right behind the issue)
  uint32_t e_bias_w = 0u, e_bias2_w = 0u, e_res_w = 0u;
  const size_t e_res_idx = (size_t)e_m * a.res_ld + e_col;
  if (e_act) {
    if (e_biasp) e_bias_w = static_cast<const uint32_t*>(e_biasp)[e_col >> 1];
    if (NS == 2 && e_bias2p) e_bias2_w = static_cast<const uint32_t*>(e_bias2p)[e_col >> 1];
    if (a.residual) e_res_w = static_cast<const uint32_t*>(a.residual)[e_res_idx >> 1];
  }

  // ---- the weight stream of this wave: step s = (unit ui, tile ti); everything on the issue path is branch free
  u32x4 wb[D][NS];
  uint32_t sb[D][NS];
  uint32_t zb[D][AWQ ? NS : 1];
  const int gsh = a.gsh;
  const int mperm = ((nn & 7) << 3) + (nn >> 3);  // marlin: column r = (unit&3)*16 + nn sits at (r&7)*8 + (r>>3) of its 64
  // every ring load is (buffer resource of the tensor, ONE per-lane offset fixed for the launch, a scalar offset for the unit / tile /
  // scale group): no per-load 64-bit VALU address arithmetic.  The stream phase of this kernel is bound by VALU issue, not by HBM
  // (tools/stream_shape_probe.hip: the same loads without any compute run at 7.8 TB/s; isa_mix.py: 48 VALU instructions per KiB tile
  // before this change, 29 of them the int4 -> bf16 conversion itself)
  const uint32_t vo_w = (uint32_t)lane * 16u;
  const uint32_t vo_s = (uint32_t)((a.marlin ? mperm : nn) >> 1) * 4u;  // the 32-bit word that holds the lane's 16-bit scale
  const uint32_t vo_z = (uint32_t)(nn >> 3) * 4u;
  __amdgpu_buffer_rsrc_t rw[NS], rs[NS], rz[AWQ ? NS : 1];
#pragma unroll
  for (int b = 0; b < NS; b++) {
    rw[b] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w[b]), 0, 0x7FFFFFF0, 0x00020000);
    rs[b] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.scales[b]), 0, 0x7FFFFFF0, 0x00020000);
    if (AWQ) rz[AWQ ? b : 0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.zeros[b]), 0, 0x7FFFFFF0, 0x00020000);
  }
  auto issue = [&](int ui, int ti, u32x4 (&w)[NS], uint32_t (&sc)[NS], uint32_t (&zp)[AWQ ? NS : 1]) {
    const int unit = u0 + ui;
    const int kt = min(sk ? O0 + ti : wave + 16 * ti, KT - 1);
    const int grp = (kt * 128) >> gsh;
    const int ucol = a.marlin ? ((unit >> 2) << 6) + ((unit & 3) << 1) : unit * a.s_unit_stride;  // (even)
    const uint32_t so_w = (uint32_t)(unit * KT + kt) * 1024u;
    const uint32_t so_s = (uint32_t)(grp * a.s_grp_stride + ucol) * 2u;
    const uint32_t so_z = (uint32_t)(grp * a.z_grp_stride + unit * a.z_unit_stride) * 4u;
#pragma unroll
    for (int b = 0; b < NS; b++) {
      w[b] = __builtin_amdgcn_raw_buffer_load_b128(rw[b], vo_w, so_w, 2);  // nt
      sc[b] = __builtin_amdgcn_raw_buffer_load_b32(rs[b], vo_s, so_s, 0);  // the word holding the scale; its half is a per-lane constant
      if (AWQ) zp[AWQ ? b : 0] = __builtin_amdgcn_raw_buffer_load_b32(rz[AWQ ? b : 0], vo_z, so_z, 0);
    }
  };
  int iu = 0, it = 0;  // issue cursor, clamped to the last step
  auto advance_issue = [&]() {
    const bool last = iu == nu - 1 && it >= CW - 1;
    const bool wrap = it >= CW - 1;
    it = last ? it : (wrap ? 0 : it + 1);
    iu = last ? iu : (wrap ? iu + 1 : iu);
  };

  // ---- prologue: this wave's x slices.  Staging lane = (row group oct, octet nn): region min(oct, XR-1) of a tile holds row
  // min(oct, M-1), so rows >= M alias the last row (their outputs are never stored; with XR = M regions the lanes of the
  // missing regions rewrite the last one with the same bytes).
  const int srg = (XR == 4 ? oct : min(oct, XR - 1)) * 272;
  const bool norm = a.norm_w != nullptr;
  float unstage[4] = {1.f, 1.f, 1.f, 1.f};  // F16 + fused norm: 2^-k of rows 0..3 of this wave's slices (wave-uniform), see below
  const uint16_t* xrow = static_cast<const uint16_t*>(a.x) + (size_t)min(oct, M - 1) * a.x_ld + nn * 8;
  auto fill_ring = [&]() {
#pragma unroll
    for (int r = 0; r < D; r++) {
      issue(iu, it, wb[r], sb[r], zb[r]);
      advance_issue();
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // (round 6) single-stream launches put the ring right behind the x requests; the gate/up pair (4 KiB per wave in its ring) still waits
  // for x first — same-box A/B, profiles/r06_ab_kernel_e_ring.txt: norm + q/k/v 7.5 -> 7.3 us, o_proj 5.7 -> 5.3, gate/up 13.9 -> 14.3
  constexpr bool XRING = GS_XRING && NS == 1;
  if (!norm) {
   if constexpr (XRING) {
    // (round 6) ALL of the wave's x slices are requested first, the ring right behind them with no wait in between, and the staging
    // arithmetic runs while both are in flight: tools/prologue_probe.hip, first tile of the median wave 1.45 -> 1.25 us after launch
    // (the slices of a wave with more than 4 k-tiles take a second batch of registers: K = 14336 has 7..8 per wave)
    u32x4 xa[4], xb[GS_MAX_TPW - 4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int kt = min(sk ? O0 + min(i, TPW - 1) : wave + 16 * min(i, TPW - 1), KT - 1);
      xa[i] = *reinterpret_cast<const u32x4*>(xrow + (size_t)kt * 128);
    }
    const bool more = TPW > 4;  // (wave-uniform)
    if (more) {
#pragma unroll
      for (int i = 0; i < GS_MAX_TPW - 4; i++) {
        const int kt = min(sk ? O0 + min(4 + i, TPW - 1) : wave + 16 * min(4 + i, TPW - 1), KT - 1);
        xb[i] = *reinterpret_cast<const u32x4*>(xrow + (size_t)kt * 128);
      }
    }
    fill_ring();
    auto stage = [&](int ti, u32x4 v) {
      const bool valid = ti < CW;
      if (!valid) v = u32x4{0u, 0u, 0u, 0u};  // a tile beyond this wave's share: zero slice (never multiplied: S = nu * CW steps)
      unsigned char* tp = xw + (size_t)ti * TLS;
      *reinterpret_cast<u32x4*>(tp + srg + nn * 16) = v;
      const float s8 = row16_sum(octet_sum<DT>(v));
      if (nn == 0) reinterpret_cast<float*>(tp + XR * 272)[oct] = s8;
    };
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i < TPW) stage(i, xa[i]);
    if (more) {
#pragma unroll
      for (int i = 0; i < GS_MAX_TPW - 4; i++)
        if (4 + i < TPW) stage(4 + i, xb[i]);
    }
    GEMV_STAMP(16);
    GEMV_STAMP(1);
   } else {
    for (int t0 = 0; t0 < TPW; t0 += 4) {
      u32x4 xv[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int ti = min(t0 + i, TPW - 1);
        const int kt = min(sk ? O0 + ti : wave + 16 * ti, KT - 1);
        xv[i] = *reinterpret_cast<const u32x4*>(xrow + (size_t)kt * 128);
      }
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int ti = t0 + i;
        if (ti < TPW) {
          const bool valid = ti < CW;
          if (!valid) xv[i] = u32x4{0u, 0u, 0u, 0u};  // a tile beyond this wave's share: zero slice (never multiplied: S = nu * CW steps)
          unsigned char* tp = xw + (size_t)ti * TLS;
          *reinterpret_cast<u32x4*>(tp + srg + nn * 16) = xv[i];
          const float s8 = row16_sum(octet_sum<DT>(xv[i]));
          if (nn == 0) reinterpret_cast<float*>(tp + XR * 272)[oct] = s8;
        }
      }
    }
    GEMV_STAMP(16);
    // the ring is filled right behind the x loads (in order per wave: x first); the staging above overlaps nothing of it
    fill_ring();
    GEMV_STAMP(1);
   }
  } else {
    // Fused RMSNorm (round 5): the normalisation factor is applied in the EPILOGUE.  rstd = 1 / sqrt(mean(x²) + eps) is one scalar
    // per row and commutes with the GEMV:  Σ_k (x_k · rstd · g_k) · w_kn  =  rstd · Σ_k (x_k · g_k) · w_kn.  Rounds 1-4 normalised x
    // BEFORE the stream, which put a workgroup barrier (the 16 x 4 table of partial Σx²) and a second pass over the wave's slices
    // between the arrival of x and the first MFMA — profiles/r02_timeline_kernel_e.txt: x in LDS at 0.99 us, barrier released at
    // 2.31, slices normalised at 2.96.  Now a wave stages x̃ = round(x · g) the moment its slices land, keeps the partial Σx² of
    // its slices for the end-of-stream reduction that exists anyway (`part`, read behind the final barrier), and the epilogue
    // multiplies the f32 dot products by rstd before anything is rounded.  One rounding per element of x̃ as before (of x·g instead
    // of x·rstd·g): the same error size, another rounding pattern — the oracle restates this order as its stated second variant
    // (oracle/vra_oracle.c orc_rms_norm_deferred / the `row_scale` of orc_wna16_gemm), tests compare against that variant.
    u32x4 xv[GS_NORM_TPW], nr[GS_NORM_TPW];
    const uint16_t* nwp = static_cast<const uint16_t*>(a.norm_w) + nn * 8;
#pragma unroll
    for (int i = 0; i < GS_NORM_TPW; i++) {  // (norm => TPW <= GS_NORM_TPW)
      const int ti = min(i, TPW - 1);
      const int kt = min(sk ? O0 + ti : wave + 16 * ti, KT - 1);
      xv[i] = *reinterpret_cast<const u32x4*>(xrow + (size_t)kt * 128);
      nr[i] = *reinterpret_cast<const u32x4*>(nwp + (size_t)kt * 128);
    }
    // the ring goes out the moment x and g have landed, BEFORE the staging arithmetic (0.3 us of VALU per wave): -0.4 % of the bs-1
    // step against issuing it behind the staging (profiles/r05_ab_kernel_e_prologue.txt); issuing it before x is requested loses
    if constexpr (!XRING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fill_ring();
    // F16: x·g without rstd leaves the range the reference's round(x·rstd·g) lives in — |x|, g ~ 1e-2 fall into f16
    // subnormals, an outlier channel with g > 1 passes 65504.  x never crosses waves, so every wave stages ITS slices of a row times
    // a power of two of its own, 2^k with the largest |x·g| of the slices in [2^10, 2^11), and takes 2^-k back out of its partial tiles
    // when they are parked (both exact in f32): the same bits as round(x·g) wherever that is a normal f16, full precision where it is
    // not.  bf16 has the exponent range of f32: nothing to do.
    float stage_sc = 1.f;
    if constexpr (std::is_same<DT, F16>::value) {
      float mx = 0.f;
#pragma unroll
      for (int ti = 0; ti < GS_NORM_TPW; ti++) {
        if (ti < TPW && ti < CW) {
          float f[8], g[8];
          unpack8<DT>(xv[ti], f);
          unpack8<DT>(nr[ti], g);
#pragma unroll
          for (int e = 0; e < 8; e++) mx = fmaxf(mx, fabsf(f[e] * g[e]));
        }
      }
      mx = fmaxf(mx, vra_dpp_f<0xB1>(mx));
      mx = fmaxf(mx, vra_dpp_f<0x4E>(mx));
      mx = fmaxf(mx, vra_dpp_f<0x141>(mx));
      mx = fmaxf(mx, vra_dpp_f<0x140>(mx));  // the row group's (16 lanes = one row's slices) maximum
      const int ex = (int)((__float_as_uint(mx) >> 23) & 0xFFu) - 127;  // floor(log2(mx)); -127 for 0 / f32 subnormals
      const int k = mx > 0.f ? min(max(10 - ex, -60), 60) : 0;
      stage_sc = __uint_as_float((uint32_t)(127 + k) << 23);
      unstage[0] = __uint_as_float((uint32_t)(127 - __builtin_amdgcn_readlane(k, 0)) << 23);
      unstage[1] = __uint_as_float((uint32_t)(127 - __builtin_amdgcn_readlane(k, 16)) << 23);
      unstage[2] = __uint_as_float((uint32_t)(127 - __builtin_amdgcn_readlane(k, 32)) << 23);
      unstage[3] = __uint_as_float((uint32_t)(127 - __builtin_amdgcn_readlane(k, 48)) << 23);
    }
    float ss = 0.f;
#pragma unroll
    for (int ti = 0; ti < GS_NORM_TPW; ti++) {
      if (ti < TPW) {
        unsigned char* tp = xw + (size_t)ti * TLS;
        float f[8], g[8];
        unpack8<DT>(xv[ti], f);
        unpack8<DT>(nr[ti], g);
        const bool valid = ti < CW;  // a tile beyond this wave's share: zero slice, and nothing of it in the sum of squares
#pragma unroll
        for (int e = 0; e < 8; e++) {
          ss += valid ? f[e] * f[e] : 0.f;
          f[e] = valid ? (std::is_same<DT, F16>::value ? (f[e] * g[e]) * stage_sc : f[e] * g[e]) : 0.f;
        }
        const u32x4 v = pack8<DT>(f);
        *reinterpret_cast<u32x4*>(tp + srg + nn * 16) = v;
        const float s8 = row16_sum(octet_sum<DT>(v));  // over the ROUNDED values the MFMA will see
        if (nn == 0) reinterpret_cast<float*>(tp + XR * 272)[oct] = s8;
      }
    }
    const float rsum = row16_sum(ss);  // Σx² of this wave's slices of row min(oct, M-1): met in the epilogue, fixed order
    if (nn == 0) part[wave * 4 + oct] = rsum;
    GEMV_STAMP(16);
    GEMV_STAMP(1);
  }
  GEMV_STAMP(2);

  // ---- main loop
  // A fragment: lane (oct, nn) reads row min(nn, 3) (region r holds row min(r, M-1)), columns j*32 + oct*8 ..
  const unsigned char* xfrag = xw + min(nn, XR - 1) * 272 + oct * 16;
  const int zsh = 4 * awq_rev(nn & 7);
  const bool shalf = a.marlin ? (nn >> 3) & 1 : nn & 1;  // which half of the loaded word is this lane's scale
  constexpr float CB = Magic<DT>::bias;
  f32x4 acc[NS];
#pragma unroll
  for (int b = 0; b < NS; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  int cu = 0, ct = 0;  // consume cursor
  // (The four waves of a SIMD are served oldest first: the oldest wave of every SIMD finishes its (equal) share ~4 us before
  // the youngest — tools/gemv_s_ts.py.  The stream phase of the gate/up launch already moves 272 KB per CU in 9.6 us = 6.3 TB/s
  // chip-wide: it is bandwidth-bound, and levelling the waves with priorities did not move the launch time.)
  const int S_pad = (S + D - 1) / D * D;  // the only loop exit is the back edge (see gemv_q4.cuh)
  for (int s0 = 0; s0 < S_pad; s0 += D) {
#pragma unroll
    for (int r = 0; r < D; r++) {
      if (s0 + r < S) {
        const unsigned char* xp = xfrag + (size_t)ct * TLS;
        f32x4 ag[NS];
        // (all LDS reads of the step go out first: with one read in front of each MFMA pair hipcc waited lgkmcnt(0) four times per step)
        u32x4 xv[4];
#pragma unroll
        for (int j = 0; j < 4; j++) xv[j] = *reinterpret_cast<const u32x4*>(xp + j * 64);
        const f32x4 sx = *reinterpret_cast<const f32x4*>(xw + (size_t)ct * TLS + XR * 272);  // Σx of rows 0..3 over this tile
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const s16x8 xf = __builtin_bit_cast(s16x8, xv[j]);
#pragma unroll
          for (int b = 0; b < NS; b++) {
            if (j == 0) DT::mfma0(ag[b], xf, magic_word<DT>(wb[r][b][j]));  // C = 0: no accumulator to clear
            else DT::mfma(ag[b], xf, magic_word<DT>(wb[r][b][j]));
          }
        }
        VRA_MFMA_DRAIN();
#pragma unroll
        for (int b = 0; b < NS; b++) {
          float s = DT::to_f32((uint16_t)(shalf ? sb[r][b] >> 16 : sb[r][b]));
          // (a wave without this k-tile holds a ZERO x slice — ag and Σx are exactly 0 and the clamped tile's scale is finite: nothing to mask)
          const float zc = AWQ ? CB + (float)((zb[r][AWQ ? b : 0] >> zsh) & 0xFu) : CB + 8.f;
#pragma unroll
          for (int e = 0; e < 4; e++) acc[b][e] = fmaf(s, fmaf(-zc, sx[e], ag[b][e]), acc[b][e]);
        }
        GEMV_STAMP(3 + 2 * (s0 + r < 5 ? s0 + r : 5));
        if (++ct == CW) {  // end of a unit: park the partial tile (rows 0..3 live in lanes 0..15)
          ct = 0;
          if (oct == 0) {
#pragma unroll
            for (int b = 0; b < NS; b++) {
              if constexpr (std::is_same<DT, F16>::value) {
#pragma unroll
                for (int e = 0; e < 4; e++) acc[b][e] *= unstage[e];  // (1.0 without a fused norm)
              }
              red[((cu * NS + b) * GS_WAVES + wave) * 16 + nn] = acc[b];
            }
          }
#pragma unroll
          for (int b = 0; b < NS; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
          ++cu;
        }
      }
      issue(iu, it, wb[r], sb[r], zb[r]);  // refill this ring slot with the step D ahead (unconditionally)
      advance_issue();
    }
  }
  if (CW == 0 && oct == 0) {  // a wave without any k-tile (K < 2048): its partial tiles are zeros
    for (int u = 0; u < nu; u++) {
#pragma unroll
      for (int b = 0; b < NS; b++) red[((u * NS + b) * GS_WAVES + wave) * 16 + nn] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  GEMV_STAMP(15);
#ifdef VRA_GEMV_TS
  if (a.ts && lane == 0 && blockIdx.x < 1024) a.ts[(size_t)2048 * 32 + (size_t)blockIdx.x * 16 + wave] = wall_clock64();  // every wave's loop end
#endif

  // ---- all partial tiles of the workgroup meet once; units*64 threads finish the outputs
  __syncthreads();
  GEMV_STAMP(17);
  if (e_act) {
    const float* rf = reinterpret_cast<const float*>(red);
    float v = 0.f, v2 = 0.f;
#pragma unroll
    for (int w = 0; w < GS_WAVES; w++) {
      v += rf[(((e_ui * NS + 0) * GS_WAVES + w) * 16 + e_nl) * 4 + e_m];
      if (NS == 2) v2 += rf[(((e_ui * NS + 1) * GS_WAVES + w) * 16 + e_nl) * 4 + e_m];
    }
    const float e_bias = DT::to_f32((uint16_t)((e_col & 1) ? e_bias_w >> 16 : e_bias_w));
    const float e_bias2 = DT::to_f32((uint16_t)((e_col & 1) ? e_bias2_w >> 16 : e_bias2_w));
    const float e_res = DT::to_f32((uint16_t)((e_res_idx & 1) ? e_res_w >> 16 : e_res_w));
    if (norm) {  // the deferred normalisation: rstd of row e_m from the 16 waves' partial sums (fixed order), on the f32 dot products
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < GS_WAVES; w++) tot += part[w * 4 + e_m];
      const float rstd = 1.0f / sqrtf(tot / (float)a.K + a.eps);
      v *= rstd;
      if (NS == 2) v2 *= rstd;
    }
    v = rnd_dt<DT>(v);
    if (e_biasp) v = rnd_dt<DT>(v + e_bias);
    if (NS == 2) {
      v2 = rnd_dt<DT>(v2);
      if (e_bias2p) v2 = rnd_dt<DT>(v2 + e_bias2);
      const float sl = rnd_dt<DT>(v / (1.0f + expf(-v)));
      v = sl * v2;
    }
    if (a.residual) v = rnd_dt<DT>(v) + e_res;
    uint16_t* const op = static_cast<uint16_t*>(e_out) + (size_t)e_m * e_ld + e_col;
    *op = DT::from_f32(v);
  }
  GEMV_STAMP(14);
  GEMV_STAMP_FLUSH();
}

template <class DT, int NS, bool AWQ>
__global__ __launch_bounds__(GS_THREADS, GS_MIN_WAVES_PER_SIMD) void gemv_q4s_kernel(const GemvSArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gemv_q4s_body<DT, NS, AWQ, 4>(a, smem);
}
// K > 16384 (Qwen2-7B down): XR = M row regions per tile
template <class DT, bool AWQ, int XR>
__global__ __launch_bounds__(GS_THREADS, GS_MIN_WAVES_PER_SIMD) void gemv_q4s_rows_kernel(const GemvSArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  gemv_q4s_body<DT, 1, AWQ, XR>(a, smem);
}
