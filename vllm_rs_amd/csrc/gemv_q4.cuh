// gemv_q4.cuh — "kernel A" for int4 weights: skinny GEMM (M <= 8 rows) that streams every packed
// weight byte exactly once.  (The dense 16-bit variant used for lm_head lives in gemv.cuh.)
//
// Roofline: HBM.  Algorithmic bytes per call: K*N/2 (packed int4) + (K/g)*N*2 (scales)
// [+ (K/g)*N/2 AWQ zeros] + M*K*2 + M*N*2.
//
// Shape of the launch:
//   PERSISTENT workgroups of NW = 8 or 15 compute waves + ONE epilogue wave.  A work item is one
//   16-column n-block (NBW = 2: the same block of the gate AND the up tensor) over the FULL K range; the
//   NW compute waves split its k-tiles
//   (wave w takes tiles w, w+NW, ...: 1 KiB coalesced per tile, the workgroup walks one contiguous
//   K/128 KiB run) and meet in LDS — no inter-workgroup traffic, no atomics, no second launch.
//   Every wave keeps a register ring of D tile-steps (8 KiB) in flight; the ring is filled BEFORE the
//   prologue computes anything, but AFTER the prologue's own loads have been issued: vector memory
//   returns in order per wave, so x / norm-weight loads queued behind 8 KiB of weights would wait for
//   HBM instead of L2.
//   x (optionally RMS-normalised on the fly: the reference's separate NormX launch, others.rs:11-29)
//   is staged in LDS once per workgroup together with its per-group sums Σx (zero-point fix-up, see
//   wna16.cuh) and reused for all work items.
//   MFMA operands: A = x (16 batch rows x 32 k, rows >= M alias row M-1 and are never stored),
//   B = (C + q) "magic" 16-bit floats (32 k x 16 columns).  D[m][n]: lane holds column n = lane&15 and
//   rows m = (lane>>4)*4 + r — so a lane needs ONE scale / zero point per tile, and for M <= 4 only
//   lanes 0..15 carry results into the cross-wave reduction.
#pragma once
#include <type_traits>

#include "gemv.cuh"

#define GQ_MAX_WAVES 16
#define GQ_ROWS 16  // max rows of x per workgroup = one MFMA tile (LDS holds rows * K * 2 bytes)
#define GQ_XR 2  // x octets per compute thread the fast prologue keeps in registers (5 measured: bs 8 -5 %, bs 1 +12 % time)
#define GQ_MAX_CHUNKS 16  // x octets per compute thread at most (loop-staged prologue)
#ifndef GQ_RING_KIB
#define GQ_RING_KIB 2  // weights in flight per wave (deeper rings measured SLOWER: see DESIGN.md §4.1)
#endif

template <class DT>
__device__ __forceinline__ float gq_scale_f32(uint32_t raw16) { return DT::to_f32((uint16_t)raw16); }
#ifdef VRA_GEMV_TS
#define GEMV_STAMP_E(i)                                                                \
  do {                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                 \
    if (a.ts && lane == 0) a.ts[(size_t)blockIdx.x * 32 + (i)] = wall_clock64();        \
    __builtin_amdgcn_sched_barrier(0);                                                 \
  } while (0)
#else
#define GEMV_STAMP_E(i) do {} while (0)
#endif

// Element `nl` (0..15, per lane) of 16 consecutive 16-bit values at a WAVE-UNIFORM address, fetched with
// scalar loads (constant address space => s_load_dwordx8, lgkmcnt): the epilogue's bias / residual reads
// must stay out of the vector-memory queue, or their s_waitcnt vmcnt(0) would drain the weight ring.
template <class DT>
__device__ __forceinline__ float gq_row16_elem(const uint16_t* base_uniform, int nl) {
  typedef const __attribute__((address_space(4))) uint32_t* cptr32;
  cptr32 p = (cptr32)(uintptr_t)base_uniform;
  uint32_t v = p[0];
#pragma unroll
  for (int i = 1; i < 8; i++) {
    const uint32_t d = p[i];
    v = (nl >> 1) == i ? d : v;
  }
  return DT::to_f32((uint16_t)((nl & 1) ? (v >> 16) : (v & 0xffffu)));
}

// LDS: xs | xsum [NF][16] | red [1 or 2][NW][NBW][RL] f32x4 (RL = 32 lanes for <= 8 rows, 64 above) | part + rstd
static inline size_t gemv_q4_lds_bytes(int nbw, int nw, int M, int K, int group_size, bool single_red = false) {
  const int spt = (group_size > 0 && group_size < 128) ? 4 : 1;
  size_t b = (size_t)M * (K / 8 + 1) * 16;  // x image: row-major, one octet of padding per row
  b += (size_t)(K / 128) * spt * 16 * 4;
  b += (size_t)(single_red ? 1 : 2) * nw * nbw * (M > 8 ? 64 : 32) * 16;
  b += (GQ_MAX_CHUNKS * GQ_MAX_WAVES + 16) * 4;
  return b;
}

// NBW = tensors per work item (2 = gate/up pair).  SPT = fix-up steps per k-tile (1: g >= 128, 4: g = 32/64).
// AWQ = per-group zero points (otherwise the zero point is the constant 8 and no zero words are loaded).
template <class DT, int NBW, int SPT, bool AWQ>
__global__ __launch_bounds__(1024) void gemv_q4_kernel(const GemvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int DK = (SPT == 4 && AWQ) ? GQ_RING_KIB / 2 : GQ_RING_KIB;  // fine AWQ groups carry 9 registers per tile
  constexpr int D = DK / NBW < 1 ? 1 : DK / NBW;  // ring depth in tile-steps
  // every kernel argument the way to the first load needs, requested in ONE batch of scalar loads (left to itself hipcc
  // loads them where they are first used: ~7 dependent scalar-cache round trips before the first global load)
  asm volatile("" ::"s"(a.x), "s"(a.x_ld), "s"(a.norm_w), "s"(a.K), "s"(a.M), "s"(a.group_size), "s"(a.m_groups), "s"(a.steps_per_item),
               "s"(a.items_q), "s"(a.items_r), "s"(a.x_chunks), "s"(a.octs_shift), "s"(a.nseg), "s"(a.seg[0].w), "s"(a.seg[0].scales),
               "s"(a.seg[0].n));
  const int tid = threadIdx.x, lane = tid & 63;
  const int NW = (int)(blockDim.x >> 6) - 1, nthr = NW << 6;  // NW compute waves + ONE epilogue wave (the last)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches, no exec masking
  const bool is_epi = wave == NW;
  const int nn = lane & 15, oct = lane >> 4;
  // row groups (more than 8 rows, where kernel C does not apply): MG workgroup families, each running the M <= 8 algorithm on its own
  // rows_per_group rows of x (8, or fewer when K is long: the rows must fit LDS).  The MG workgroups that stream the same n-blocks get consecutive-by-8 ids, i.e. the
  // same XCD and the same dispatch moment, so every packed tile is fetched from HBM once and from the
  // XCD's L2 by the other MG-1.
  const int MG = a.m_groups > 1 ? a.m_groups : 1;
  const int wg = (int)blockIdx.x;
  const int slot = MG > 1 ? (wg / (8 * MG)) * 8 + (wg & 7) : wg;  // persistent slot within its family
  const int mgrp = MG > 1 ? (wg >> 3) % MG : 0;
  const int nslots = MG > 1 ? (int)gridDim.x / MG : (int)gridDim.x;
  const int m_base = mgrp * a.rows_per_group;
  const int K = a.K, M = min(a.M - m_base, MG > 1 ? a.rows_per_group : a.M), KT = K >> 7;
  const bool grouped = a.group_size > 0 && a.group_size < K;
  const int gsh = grouped ? 31 - __builtin_clz(a.group_size) : 31;  // k >> gsh = scale group (power-of-two groups)
  const int NF = KT * SPT;
  GEMV_STAMP(0);

  // ---- LDS carve-up
  // x image: xs[(m*(octs+1) + o)*4 .. +4] = x[m][o*8 .. +8] — row-major with one octet of padding per row: the staging
  // stores of a wave (consecutive octets of a row) are consecutive in LDS, and the 16 rows of an MFMA fragment read are 4
  // banks apart.  (The former [octet][row] image made every staging store a 32-way bank conflict at 8 rows.)
  uint32_t* xs = reinterpret_cast<uint32_t*>(smem);
  const int XRS = ((K >> 3) + 1) * 4;  // row stride in u32
  size_t off = (size_t)M * XRS * 4;
  float* xsum = reinterpret_cast<float*>(smem + off);  // [NF][16]: Σx of row m over fix-up step f
  off += (size_t)NF * 16 * 4;
  const int RL = M > 8 ? 64 : 32;  // lanes of a partial tile that carry rows < M
  const bool single_red = a.single_red != 0;  // one reduction buffer + a second barrier per item (LDS-tight shapes)
  f32x4* red = reinterpret_cast<f32x4*>(smem + off);  // [1 or 2][NW][NBW][RL]
  off += (size_t)(single_red ? 1 : 2) * NW * NBW * RL * sizeof(f32x4);
  float* part = reinterpret_cast<float*>(smem + off);  // [GQ_MAX_CHUNKS][GQ_MAX_WAVES] partial Σx², then rstd[8]
  float* rstd_s = part + GQ_MAX_CHUNKS * GQ_MAX_WAVES;  // [16]

  // ---- this workgroup's stream of tile-steps
  const int T = a.steps_per_item;  // = ceil(KT / NW)
  const int my_items = a.items_q + (slot < a.items_r ? 1 : 0);  // = ceil((n_items - slot) / nslots)
  const int S = (a.dbg & 1) ? 0 : my_items * T;

  u32x4 wb[D][NBW];
  uint32_t sb[D][NBW][SPT];
  uint32_t zb[D][NBW][AWQ ? SPT : 1];

  // Everything on the issue path is BRANCH FREE (selects and clamps only): hipcc's s_waitcnt insertion
  // takes the minimum outstanding-load count over all control-flow paths, so a single conditional
  // load anywhere in the ring turns every `vmcnt(N)` of the consume loop into `vmcnt(0)` — a full
  // HBM round trip per tile-step.  Steps past the end of the stream re-issue the last valid step
  // (L2 hits, never consumed).
  const int nseg = a.nseg;
  const int blk1 = nseg > 1 ? a.seg[1].blk_start : 0x7fffffff, blk2 = nseg > 2 ? a.seg[2].blk_start : 0x7fffffff;
  auto issue = [&](int item, int i, u32x4 (&w)[NBW], uint32_t (&sc)[NBW][SPT], uint32_t (&zp)[NBW][AWQ ? SPT : 1]) {
    const int kt = min(wave + NW * i, KT - 1);  // waves without a tile in this step re-read the last one; their scale is zeroed
    const int fb = slot + item * nslots;
#pragma unroll
    for (int b = 0; b < NBW; b++) {
      const void* wp;
      const void* scp;
      const uint32_t* qzp;
      int n, nb;
      if (NBW == 2) {  // gate/up pair: tensor b of the same n-block
        wp = a.seg[b].w, scp = a.seg[b].scales, qzp = a.seg[b].qzeros, n = a.seg[b].n, nb = fb;
      } else {
        const bool s1 = fb >= blk1, s2 = fb >= blk2;
        wp = s2 ? a.seg[2].w : (s1 ? a.seg[1].w : a.seg[0].w);
        scp = s2 ? a.seg[2].scales : (s1 ? a.seg[1].scales : a.seg[0].scales);
        qzp = s2 ? a.seg[2].qzeros : (s1 ? a.seg[1].qzeros : a.seg[0].qzeros);
        n = s2 ? a.seg[2].n : (s1 ? a.seg[1].n : a.seg[0].n);
        nb = fb - (s2 ? blk2 : (s1 ? blk1 : 0));
      }
      w[b] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp) + ((size_t)nb * KT + kt) * 64 + lane);
      const uint16_t* sp = static_cast<const uint16_t*>(scp);
#pragma unroll
      for (int q = 0; q < SPT; q++) {
        const int grp = (kt * 128 + q * 32) >> gsh;
        // the 32-bit word holding the scale (its half is a per-lane constant, `shalf`): a 16-bit load gets
        // an immediate v_and (zero extension) from hipcc, i.e. a wait for the load right after its issue
        const int64_t si = (int64_t)grp * n + nb * 16 + nn;  // row-major [K/g, n] (permuted layouts are converted by the caller)
        sc[b][q] = reinterpret_cast<const uint32_t*>(sp)[si >> 1];
        if (AWQ) zp[b][AWQ ? q : 0] = qzp[(size_t)grp * (n >> 3) + nb * 2 + (nn >> 3)];
      }
    }
  };
  int iitem = 0, ii = 0;  // issue cursor (clamped to the last step of the stream)
  auto advance_issue = [&]() {
    const bool last = iitem == my_items - 1 && ii == T - 1;
    const bool wrap = ii == T - 1;
    ii = last ? ii : (wrap ? 0 : ii + 1);
    iitem = last ? iitem : (wrap ? iitem + 1 : iitem);
  };
  auto fill_ring = [&]() {
#pragma unroll
    for (int r = 0; r < D; r++) {
      issue(iitem, ii, wb[r], sb[r], zb[r]);
      advance_issue();
    }
  };

  // ---- prologue: stage x (and Σx) in LDS  [compute waves; the epilogue wave only keeps the barriers]
  // The prologue's loads (L2 hits) are complete BEFORE the first HBM load of the ring is queued: measured on
  // MI355X, an L2-hit load queued behind streaming HBM loads of other waves of the CU returns microseconds late.
  // Two shapes: <= GQ_XR octets per thread (decode batches 1..4 at K = 4096: the headline) keep x in registers
  // and overlap the normalisation with the ring's first HBM round trip; larger x is staged by loops (two
  // passes over L2 when the RMSNorm is fused) and the ring is filled afterwards.
  const int octs = K >> 3, OC = M * octs;
  const int opg = 16 / SPT;  // octets per fix-up step
  const int nch = MG > 1 ? (OC + nthr - 1) / nthr : a.x_chunks;  // x chunks (one octet per compute thread each)
  const bool fast = nch <= GQ_XR;
  const bool norm = a.norm_w != nullptr;
  const uint16_t* np = static_cast<const uint16_t*>(norm ? a.norm_w : a.x);
  auto row_of = [&](int c) {  // octs % 64 == 0: a wave never straddles rows
    if (M == 1) return 0;
    const int o0 = c * nthr + (wave << 6);
    return min(a.octs_shift >= 0 ? o0 >> a.octs_shift : o0 / octs, M - 1);
  };
  auto stage_octet = [&](int c, u32x4 v, u32x4 nv, float rs) {  // normalise (optional), write the LDS image and Σx
    const int m = row_of(c), o = c * nthr + tid, oo = o - m * octs;
    if (norm) {
      float f[8], g[8];
      unpack8<DT>(v, f);
      unpack8<DT>(nv, g);
#pragma unroll
      for (int i = 0; i < 8; i++) f[i] = f[i] * rs * g[i];
      v = pack8<DT>(f);
    }
    if (o < OC) *reinterpret_cast<u32x4*>(xs + (size_t)m * XRS + oo * 4) = v;
    float s8 = octet_sum<DT>(v);  // over the ROUNDED values the MFMA will see
    s8 = SPT == 4 ? quad_sum(s8) : row16_sum(s8);  // opg = 4 or 16 consecutive octets (lanes)
    if (o < OC && (oo & (opg - 1)) == 0) xsum[(oo / opg) * 16 + m] = s8;
  };
  auto octet_ss = [&](u32x4 v) {
    float f[8];
    unpack8<DT>(v, f);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) ss += f[i] * f[i];
    return wave_sum(ss);
  };
  auto load_x = [&](int c) {
    const int m = row_of(c), oo = min(c * nthr + tid - m * octs, octs - 1);
    return reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.x) + (size_t)(m_base + m) * a.x_ld)[oo];
  };
  auto load_nw = [&](int c) {
    const int m = row_of(c), oo = min(c * nthr + tid - m * octs, octs - 1);
    return reinterpret_cast<const u32x4*>(np)[oo];
  };
  // rstd of every row from the per-(chunk, wave) partial sums in `part` (fixed summation order)
  auto reduce_rows = [&]() {
    if (wave == 0) {
      for (int m = 0; m < M; m++) {
        float acc = 0.f;
        for (int p0 = 0; p0 < nch * NW; p0 += 64) {
          const int pp = p0 + lane, c = pp / NW, w = pp - c * NW;
          const int o0 = c * nthr + (w << 6);
          const bool mine = c < nch && o0 < OC && o0 / octs == m;
          acc += wave_sum(mine ? part[c * GQ_MAX_WAVES + w] : 0.f);
        }
        if (lane == 0) rstd_s[m] = 1.0f / sqrtf(acc / (float)K + a.eps);
      }
    }
  };
  u32x4 xr[GQ_XR], nr[GQ_XR];
  if (fast) {
    if (!is_epi) {
#pragma unroll
      for (int c = 0; c < GQ_XR; c++) {  // straight line, clamped: no conditional load before the ring (see `issue`)
        xr[c] = load_x(c);
        nr[c] = load_nw(c);
      }
      __builtin_amdgcn_sched_barrier(0);
      GEMV_STAMP(16);
      if (norm) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); without a norm the staging is short: waiting costs ~4 % (measured)
      fill_ring();
      __builtin_amdgcn_sched_barrier(0);
      GEMV_STAMP(1);
      if (norm) {
#pragma unroll
        for (int c = 0; c < GQ_XR; c++) {
          if (c < nch) {
            const float ss = octet_ss(xr[c]);
            if (lane == 0) part[c * GQ_MAX_WAVES + wave] = ss;
          }
        }
      }
      GEMV_STAMP(17);
    }
    float rs1 = 1.0f;
    if (norm) {
      __syncthreads();
      GEMV_STAMP(18);
      if (M > 1) {
        reduce_rows();
        __syncthreads();
      } else if (!is_epi) {  // single row: every thread adds the partials itself (fixed order), no second barrier
        float tot = 0.f;
        for (int c = 0; c < nch; c++)
          for (int w = 0; w < NW; w++)
            if (c * nthr + (w << 6) < OC) tot += part[c * GQ_MAX_WAVES + w];
        rs1 = 1.0f / sqrtf(tot / (float)K + a.eps);
      }
    }
    GEMV_STAMP(19);
    if (!is_epi) {
#pragma unroll
      for (int c = 0; c < GQ_XR; c++) {
        if (c < nch) {
          // (wave-uniform; kept in an SGPR: as a VGPR the packed multiply reads it as a register PAIR whose
          // second half may be a ring register with a load in flight — a false vmcnt dependency)
          const float rs = norm ? __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(M == 1 ? rs1 : rstd_s[row_of(c)]))) : 1.0f;
          stage_octet(c, xr[c], nr[c], rs);
        }
      }
    }
  } else {
    // large x (decode batches > 2 at K = 4096, or K = 14336): batches of 4 octets per thread through L2 (4 loads
    // in flight per iteration), raw values parked in their final LDS slots; with a fused RMSNorm a second pass
    // rescales them in place (norm weights again 4 loads per iteration).  The ring is filled afterwards.
    if (!is_epi) {
      for (int c0 = 0; c0 < nch; c0 += 4) {
        u32x4 xv[4];
#pragma unroll
        for (int i = 0; i < 4; i++) xv[i] = load_x(min(c0 + i, nch - 1));
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int c = c0 + i;
          if (c < nch) {
            if (norm) {
              const float ss = octet_ss(xv[i]);
              if (lane == 0) part[c * GQ_MAX_WAVES + wave] = ss;
              const int m = row_of(c), o = c * nthr + tid, oo = o - m * octs;
              if (o < OC) *reinterpret_cast<u32x4*>(xs + (size_t)m * XRS + oo * 4) = xv[i];  // raw, rescaled in pass 2
            } else {
              stage_octet(c, xv[i], xv[i], 1.0f);
            }
          }
        }
      }
    }
    if (norm) {
      __syncthreads();
      reduce_rows();
      __syncthreads();
      if (!is_epi) {
        for (int c0 = 0; c0 < nch; c0 += 4) {
          u32x4 nv[4];
#pragma unroll
          for (int i = 0; i < 4; i++) nv[i] = load_nw(min(c0 + i, nch - 1));
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const int c = c0 + i;
            if (c < nch) {
              const int m = row_of(c), o = c * nthr + tid, oo = min(o - m * octs, octs - 1);
              const u32x4 raw = *reinterpret_cast<const u32x4*>(xs + (size_t)m * XRS + oo * 4);
              const float rs = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(rstd_s[m])));
              stage_octet(c, raw, nv[i], rs);
            }
          }
        }
      }
    }
    if (!is_epi) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
      fill_ring();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  GEMV_STAMP(20);
  __syncthreads();
  GEMV_STAMP(2);

  const int n_it = (a.dbg & 1) ? 0 : my_items;  // work items of this workgroup
  if (is_epi) {
    // ================= epilogue wave: reduce the NW partial tiles of every work item, fused epilogue, store.
    // It owns ALL global stores (gfx9 counts loads and stores in one vmcnt and they retire out of order
    // with each other: one possibly-pending store in a compute wave would force its ring waits to
    // vmcnt(0)) and prefetches bias / residual BEFORE it waits for the item's partials.
    int parity = 0;
    const int nout = 16 * M;
    for (int it = 0; it < n_it; it++) {
      const int fb = slot + it * nslots;
      const int segi = NBW == 2 ? 0 : (fb >= blk2 ? 2 : (fb >= blk1 ? 1 : 0));
      const int nb = fb - (NBW == 2 ? 0 : a.seg[segi].blk_start);
      const GemvSeg& sg = a.seg[segi];
      float bv[4] = {0.f, 0.f, 0.f, 0.f}, bu[4] = {0.f, 0.f, 0.f, 0.f}, rv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 4; h++) {
        const int idx = lane + 64 * h;
        if (idx < nout) {
          const int n = nb * 16 + (idx & 15), m = idx >> 4;
          if (sg.bias) bv[h] = DT::to_f32(static_cast<const uint16_t*>(sg.bias)[n]);
          if (NBW == 2 && a.seg[1].bias) bu[h] = DT::to_f32(static_cast<const uint16_t*>(a.seg[1].bias)[n]);
          if (a.residual) rv[h] = DT::to_f32(static_cast<const uint16_t*>(a.residual)[(size_t)(m_base + m) * a.res_ld + n]);
        }
      }
      __syncthreads();  // the item's partial tiles are in red[parity]
      GEMV_STAMP_E(21 + 2 * (it < 3 ? it : 3));
      const float* rf = reinterpret_cast<const float*>(red + (size_t)(single_red ? 0 : parity) * NW * NBW * RL);
      float vs[4], v2s[4];
#pragma unroll
      for (int h = 0; h < 4; h++) {
        const int idx = lane + 64 * h;
        vs[h] = v2s[h] = 0.f;
        if (idx < nout) {
          const int nl = idx & 15, m = idx >> 4;
          const int slot = ((m >> 2) * 16 + nl) * 4 + (m & 3);  // D layout: column = lane&15, row = (lane>>4)*4 + reg
          for (int w = 0; w < NW; w++) {
            vs[h] += rf[(w * NBW) * RL * 4 + slot];
            if (NBW > 1) v2s[h] += rf[(w * NBW + (NBW - 1)) * RL * 4 + slot];
          }
        }
      }
      if (single_red && it + 1 < n_it) __syncthreads();  // the buffer may be overwritten by the next item
#pragma unroll
      for (int h = 0; h < 4; h++) {
        const int idx = lane + 64 * h;
        if (idx < nout) {
          const int nl = idx & 15, m = idx >> 4;
          const int n = nb * 16 + nl;
          float v = rnd_dt<DT>(vs[h]);
          if (sg.bias) v = rnd_dt<DT>(v + bv[h]);
          if (NBW == 2) {
            float v2 = rnd_dt<DT>(v2s[h]);
            if (a.seg[1].bias) v2 = rnd_dt<DT>(v2 + bu[h]);
            const float sl = rnd_dt<DT>(v / (1.0f + expf(-v)));
            v = sl * v2;
          }
          if (a.residual) v = rnd_dt<DT>(v) + rv[h];
          if (a.out_f32) static_cast<float*>(sg.out)[(size_t)(m_base + m) * sg.out_ld + n] = rnd_dt<DT>(v);
          else static_cast<uint16_t*>(sg.out)[(size_t)(m_base + m) * sg.out_ld + n] = DT::from_f32(v);
        }
      }
      GEMV_STAMP_E(22 + 2 * (it < 3 ? it : 3));
      parity ^= 1;  // red[parity] is rewritten only after the NEXT barrier, which this wave must reach first
    }
    return;
  }

  // ================= compute waves
  // A-fragment addressing: lanes whose batch row does not exist alias row M-1 (their D rows are never stored)
  const uint32_t xbase = (uint32_t)min(nn, M - 1) * (uint32_t)XRS + (uint32_t)oct * 4u;
  constexpr uint32_t xstep = 16u;  // u32 per 4 octets (one j step)
  const int zsh = 4 * awq_rev(nn & 7);
  // which half of the loaded word is this lane's scale
  const bool shalf = nn & 1;
  constexpr float CB = Magic<DT>::bias;

  f32x4 acc[NBW];
#pragma unroll
  for (int b = 0; b < NBW; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
  int ci = 0, parity = 0, items_done = 0;  // consume cursor
  // The stream is padded to a multiple of D: the only loop exit is the back edge.  (An exit between two
  // unrolled bodies is routed by the CFG structurizer through a shared "Flow" block from which the loop
  // header is syntactically reachable with a half-issued refill — a false path that again degrades the
  // ring waits to vmcnt(0).)  Padding steps skip the arithmetic; their clamped re-loads hit L2.
  const int S_pad = (S + D - 1) / D * D;
  for (int s0 = 0; s0 < S_pad; s0 += D) {
#pragma unroll
    for (int r = 0; r < D; r++) {
      if (s0 + r < S) {
        // ---- consume tile-step ci of the current item: one branch-free MFMA region per fix-up step (common.cuh)
        const int ktr = wave + NW * ci;
        const bool valid = ktr < KT;
        const int kt = min(ktr, KT - 1);
        const uint32_t* xp = xs + xbase + (uint32_t)(kt * 4) * xstep;  // tile kt = octets kt*16 ..
        const f32x4* sxp = reinterpret_cast<const f32x4*>(xsum + (size_t)kt * SPT * 16) + oct;
        f32x4 ag[NBW];
#pragma unroll
        for (int b = 0; b < NBW; b++) ag[b] = vra_zero_acc();
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const s16x8 xf = __builtin_bit_cast(s16x8, *reinterpret_cast<const u32x4*>(xp + j * xstep));
#pragma unroll
          for (int b = 0; b < NBW; b++) DT::mfma(ag[b], xf, magic_word<DT>(wb[r][b][j]));
          if (SPT == 4 || j == 3) {  // end of a fix-up step: acc += s * (acc_g - (C+z) * Σx_g)
            VRA_MFMA_DRAIN();
            const int q = SPT == 4 ? j : 0;
            const f32x4 sx = sxp[q * 4];
#pragma unroll
            for (int b = 0; b < NBW; b++) {
              float s = gq_scale_f32<DT>(shalf ? sb[r][b][q] >> 16 : sb[r][b][q]);
              s = valid ? s : 0.f;
              const float zc = AWQ ? CB + (float)((zb[r][b][AWQ ? q : 0] >> zsh) & 0xFu) : CB + 8.f;
#pragma unroll
              for (int e = 0; e < 4; e++) acc[b][e] = fmaf(s, fmaf(-zc, sx[e], ag[b][e]), acc[b][e]);
              if (SPT == 4 && j < 3) ag[b] = vra_zero_acc();
            }
          }
        }
        GEMV_STAMP(3 + 2 * (s0 + r < 5 ? s0 + r : 5));
        // ---- end of a work item: hand the partial tile to the epilogue wave
        if (++ci == T) {
          ci = 0;
          if (single_red && items_done > 0) __syncthreads();  // the epilogue wave has read the previous item
          ++items_done;
          f32x4* rbuf = red + (size_t)(single_red ? 0 : parity) * NW * NBW * RL;
          if (oct * 4 < M) {
#pragma unroll
            for (int b = 0; b < NBW; b++) rbuf[(wave * NBW + b) * RL + lane] = acc[b];
          }
#pragma unroll
          for (int b = 0; b < NBW; b++) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
          __syncthreads();
          parity ^= 1;
        }
        GEMV_STAMP(4 + 2 * (s0 + r < 5 ? s0 + r : 5));
      }
      // ---- refill this ring slot with the step D ahead (unconditionally: see the note at `issue`)
      issue(iitem, ii, wb[r], sb[r], zb[r]);
      advance_issue();
    }
  }
  GEMV_STAMP(15);
  GEMV_STAMP_FLUSH();
}
