// gemv_q4w.cuh — "kernel W": int4 GEMM for decode batches of 5..32 rows (and, in row blocks of 32, short prefills of 33..256 rows) and K <= 4096 (q/k/v, o_proj, gate/up of the 7-8B
// shapes), on the unit distribution of kernel E (gemv_q4s.cuh): one workgroup per CU, a contiguous run of units (16-column
// n-blocks, or gate/up pairs) per workgroup.  EIGHT waves here (two per SIMD: 256 VGPRs each), every wave streaming the
// k-tiles w, w+8, w+16, w+24 of each unit through a 4-deep register ring (4 KiB per wave, 32 KiB per CU in flight as in E).
// What changes with the rows:
//   * x lives in REGISTERS: a wave only ever needs the 32 rows x 512 columns of its own four k-tiles, as MFMA A fragments
//     (lane (oct, nn): row mt*16 + nn, 8 columns) — 4 tiles x 4 k-steps x MT m-tiles x 4 VGPRs (128 at 32 rows; with 16 waves
//     of 128 VGPRs the fragments of two tiles already pushed hipcc into spilling them right behind their loads).  Nothing is
//     staged through LDS, no workgroup ever holds the 256 KiB of a 32-row x (kernel C stages it in K-chunks with a barrier per
//     chunk), and the fused RMSNorm costs ONE barrier (the 8 x 32 table of partial Σx²), like kernel E.
//   * every dequantised weight fragment feeds MT MFMAs (one per 16-row m-tile): the int4 -> bf16 work is done once for all rows.
//   * the partial tiles of a unit (8 waves x NS x MT KiB) meet in a double-buffered LDS slab behind one barrier per unit; the
//     first 16*MT*16 threads sum them in fixed order, run the epilogue from LDS-resident bias / residual and park the results in
//     LDS — the main loop contains no global access besides the weight ring (a load or a store there would turn hipcc's ring
//     waits into vmcnt(0), gemv_q4.cuh) — and everything is stored after the stream has ended.
// Roofline: HBM (weights once; x is re-read by every workgroup from L2: M*K*2 bytes x grid).  Same arithmetic contract as the
// other int4 kernels (wna16.cuh): exact product, one rounding.
#pragma once
#include "gemv_q4s.cuh"

#define GW_MAX_UNITS 8         // units per workgroup of a launch with a bias or a residual (their values ride in registers: EPI_IT)
#define GW_MAX_UNITS_PLAIN 32  // ... of a launch without either (gate/up pairs in 16-row blocks: 28 per workgroup at 128 rows)
#define GW_WAVES 8
#define GW_THREADS (GW_WAVES * 64)
#define GW_TPW 4  // k-tiles per wave and unit: w + 8*ti (K <= 4096)

static inline size_t gemv_q4w_lds_bytes(int ns, int mt, int max_units, bool has_res, bool kz = false) {
  size_t b = (size_t)2 * GW_WAVES * ns * mt * 1024;           // double-buffered partial tiles
  if (kz) b += (size_t)max_units * mt * 16 * 16 * 4;          // K-sliced launches: the workgroup's f32 partial tiles
  b += (size_t)GW_WAVES * 32 * 4;                              // Σx² partials [wave][32 rows]
  b += (size_t)GW_WAVES * GW_TPW * 32 * 4;                     // Σx per (wave, tile, row)
  b += (size_t)max_units * mt * 16 * 16 * 2;                   // finished outputs (16-bit), stored after the stream
  if (has_res) b += (size_t)max_units * mt * 16 * 16 * 2;      // residual tiles
  b += (size_t)max_units * ns * 16 * 2 + 64;                   // bias values
  return b;
}

// NS = streams (2: gate/up pair with SiLU*mul), MT = 16-row m-tiles (1: up to 16 rows, 2: up to 32), NORM = fused RMSNorm
// (compile time: a run-time branch around code that rewrites 128 fragment registers ends in spills at its merge point)
// PSEQ ("pair, sequential", NS = 1, MT = 2, launches without bias / residual): a gate/up pair of 17+ rows.  The pair kernel proper
// (NS = 2) keeps one m-tile — with two it spills — and is bound by instruction issue (~2.1 us per pair and workgroup: every
// dequantised fragment feeds ONE MFMA).  Here the units of a workgroup alternate gate block nb, up block nb, gate nb+1, ...: each
// is an ordinary single-stream unit over two m-tiles (every fragment feeds two MFMAs), the gate result waits in the LDS output
// tile of its unit and the up unit's epilogue applies SiLU(gate) * up — the arithmetic of the pair kernel, rounding for rounding.
// units_q / units_r count PAIRS.
// XF: x comes from a.x_frag (fragment order, launches of up to 32 rows: GemvSArgs::x_frag); compile time, as NORM
// KZ: K > 4096 as K slices across workgroups (GemvSArgs::kz; single stream, no fused norm — a slice does not see whole rows)
template <class DT, int NS, int MT, bool AWQ, bool NORM, bool PSEQ = false, bool XF = false, bool KZ = false>
__global__ __launch_bounds__(GW_THREADS) void gemv_q4w_kernel(const GemvSArgs a) {
  static_assert(!PSEQ || (NS == 1 && MT == 2), "PSEQ is the single-stream two-m-tile kernel over alternating gate / up units");
  static_assert(!KZ || (NS == 1 && !NORM && !PSEQ), "K slices: single stream, no fused RMSNorm");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int D = GW_TPW;  // ring slot = the wave's tile index within a unit (w + 8*ti)
  asm volatile("" ::"s"(a.x), "s"(a.x_ld), "s"(a.norm_w), "s"(a.K), "s"(a.M), "s"(a.KT), "s"(a.gsh), "s"(a.units_q), "s"(a.units_r), "s"(a.w[0]),
               "s"(a.scales[0]), "s"(a.s_grp_stride), "s"(a.s_unit_stride), "s"(a.marlin), "s"(a.residual), "s"(a.res_ld), "s"(a.nseg),
               "s"(a.seg[0].out), "s"(a.seg[0].bias), "s"(a.seg[0].out_ld), "s"(a.seg[1].unit_start), "s"(a.seg[2].unit_start));
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 15, oct = lane >> 4;
  // rows: up to 32 per launch as a decode GEMM; 33..256 rows (short prefills, round 4) as blockIdx.y ROW BLOCKS of 16*MT rows, every
  // block an independent workgroup over its own rows and a contiguous run of units — the weights of a unit are then read once
  // per row block (L2 / MALL hits), the x of a workgroup stays at 16*MT rows, and nothing meets across workgroups
  const int row0 = (int)blockIdx.y * (MT * 16);
  const int M = min(a.M - row0, MT * 16), KT = a.KT;
  const int zi = KZ ? (int)blockIdx.x / a.kz_groups : 0;                    // K slice
  const int wg = KZ ? (int)blockIdx.x - zi * a.kz_groups : (int)blockIdx.x;  // unit group
  const int kt0 = KZ ? zi * a.ktz : 0, KTL = KZ ? min(a.ktz, KT - kt0) : KT;  // this workgroup's k-tiles: kt0 .. kt0 + KTL - 1
  const bool owner = !KZ || zi == a.kz - 1;                                  // runs the epilogue and stores
  const int u0 = (wg * a.units_q + min(wg, a.units_r)) * (PSEQ ? 2 : 1);
  const int nu = (a.units_q + (wg < a.units_r ? 1 : 0)) * (PSEQ ? 2 : 1);
  const int max_u = (a.units_q + (a.units_r ? 1 : 0)) * (PSEQ ? 2 : 1);
  const bool has_res = a.residual != nullptr;
  // K slices in ROW BLOCKS (33..128 rows, down_proj of short prefills): the owner takes bias and residual straight from memory in its
  // exchange loop instead of carrying them in registers through the stream (EPI_IT bounds the units of a workgroup to GW_MAX_UNITS;
  // a row-blocked launch gives a workgroup up to GW_MAX_UNITS_PLAIN units); slabs and flags get a row-block index
  const int rbi = (int)blockIdx.y;
  const bool late_epi = KZ && gridDim.y > 1;
  GEMV_STAMP(0);

  // ---- LDS carve-up
  f32x4* red = reinterpret_cast<f32x4*>(smem);  // [2][wave][NS][MT][64 lanes]
  size_t off = (size_t)2 * GW_WAVES * NS * MT * 1024;
  float* part = reinterpret_cast<float*>(smem + off);  // [wave][32]
  off += (size_t)GW_WAVES * 32 * 4;
  float* xsum = reinterpret_cast<float*>(smem + off) + (size_t)wave * GW_TPW * 32;  // this wave's [4 tiles][32 rows]
  off += (size_t)GW_WAVES * GW_TPW * 32 * 4;
  float* outf = reinterpret_cast<float*>(smem + off);  // KZ: [unit][MT*16 rows][16 cols] f32 partial tiles of this slice
  if (KZ) off += (size_t)max_u * MT * 256 * 4;
  uint16_t* outs = reinterpret_cast<uint16_t*>(smem + off);  // [unit][MT*16 rows][16 cols]
  off += (size_t)max_u * MT * 256 * 2;
  uint16_t* ress = reinterpret_cast<uint16_t*>(smem + off);
  if (has_res) off += (size_t)max_u * MT * 256 * 2;
  uint16_t* biass = reinterpret_cast<uint16_t*>(smem + off);  // [unit][NS][16]

  // ---- epilogue operands of all units of this workgroup -> LDS (nothing but the weight ring touches memory in the main loop)
  // thread t: (unit t >> 9 for MT = 2, row, column) covers 16*MT rows x 16 columns per unit
  constexpr int OPU = MT * 256;  // outputs per unit
  auto seg_of = [&](int unit, void*& out, const void*& bias, int& ld, int& col0) {
    if (PSEQ) {  // (plain launches only: no bias) unit = 2*block + {0: gate, 1: up}; only the up unit's tile is stored
      out = a.seg[0].out, bias = nullptr, ld = a.seg[0].out_ld, col0 = (unit >> 1) * 16;
      return;
    }
    // (`unit` is wave-uniform at every call — the callers pass it through readfirstlane: with a per-lane unit hipcc selected the
    // ADDRESS of the segment field inside the kernel-argument block and fetched it with a VECTOR load, each followed by a vmcnt(0)
    // that also waited for the weight ring: ~0.5 us per unit of a launch with a residual, tools/gemv_w_ts.py)
    const bool s1 = a.nseg > 1 && NS == 1 && unit >= a.seg[1].unit_start, s2 = a.nseg > 2 && NS == 1 && unit >= a.seg[2].unit_start;
    out = s2 ? a.seg[2].out : (s1 ? a.seg[1].out : a.seg[0].out);
    bias = s2 ? a.seg[2].bias : (s1 ? a.seg[1].bias : a.seg[0].bias);
    ld = s2 ? a.seg[2].out_ld : (s1 ? a.seg[1].out_ld : a.seg[0].out_ld);
    col0 = (unit - (s2 ? a.seg[2].unit_start : (s1 ? a.seg[1].unit_start : a.seg[0].unit_start))) * 16;
  };
  // (requested BEFORE the x fragments and written to LDS behind their issue: the round trip of the residual then runs under
  // the x loads instead of in front of them; skipped entirely when the launch has neither bias nor residual)
  const bool any_bias = a.seg[0].bias || (a.nseg > 1 && a.seg[1].bias) || (a.nseg > 2 && a.seg[2].bias);
  constexpr int EPI_IT = GW_MAX_UNITS * OPU / GW_THREADS;
  uint16_t e_res[EPI_IT], e_b0[EPI_IT], e_b1[EPI_IT];
  auto request_epilogue_operands = [&]() {
#pragma unroll
    for (int it = 0; it < EPI_IT; it++) {
      const int idx = tid + it * GW_THREADS;
      e_res[it] = e_b0[it] = e_b1[it] = 0;
      if (idx < nu * OPU) {
        const int ui = __builtin_amdgcn_readfirstlane(idx / OPU), rem = idx - ui * OPU, row = rem >> 4, col = rem & 15;  // (OPU is a multiple of 64)
        void* o_;
        const void* b_;
        int ld_, c0;
        seg_of(u0 + ui, o_, b_, ld_, c0);
        if (has_res && row < M) e_res[it] = static_cast<const uint16_t*>(a.residual)[(size_t)(row0 + row) * a.res_ld + c0 + col];
        if (row == 0) {
          if (b_) e_b0[it] = static_cast<const uint16_t*>(b_)[c0 + col];
          if (NS == 2 && a.seg[1].bias) e_b1[it] = static_cast<const uint16_t*>(a.seg[1].bias)[c0 + col];
        }
      }
    }
  };
  auto stage_epilogue_operands = [&]() {
#pragma unroll
    for (int it = 0; it < EPI_IT; it++) {
      const int idx = tid + it * GW_THREADS;
      if (idx < nu * OPU) {
        const int ui = idx / OPU, rem = idx - ui * OPU, row = rem >> 4, col = rem & 15;
        if (has_res) ress[idx] = e_res[it];
        if (row == 0) {
          biass[(ui * NS + 0) * 16 + col] = e_b0[it];
          if (NS == 2) biass[(ui * NS + 1) * 16 + col] = e_b1[it];
        }
      }
    }
  };

  __builtin_amdgcn_sched_barrier(0);
  // ---- the weight stream: step = (unit ui, tile ti); branch-free issue.  (Its first four tiles are requested BEFORE the x
  // fragments: the HBM round trip then runs under the x loads, which keep the CU's texture path busy for ~4 µs at 32 rows.)
  u32x4 wb[D][NS];
  uint32_t sb[D][NS];
  uint32_t zb[D][AWQ ? NS : 1];
  const int gsh = a.gsh;
  const int mperm = ((nn & 7) << 3) + (nn >> 3);
  // every ring load is (buffer resource of the tensor, ONE per-lane offset fixed for the launch, a scalar offset for the unit / tile /
  // scale group): no per-load 64-bit VALU address arithmetic — the loop is bound by VALU issue (tools/gemv_w_ts.py, isa_mix.py)
  const uint32_t vo_w = (uint32_t)lane * 16u;
  const uint32_t vo_s = (uint32_t)((a.marlin ? mperm : nn) >> 1) * 4u;  // the 32-bit word that holds the lane's 16-bit scale
  const uint32_t vo_z = (uint32_t)(nn >> 3) * 4u;
  auto issue = [&](int ui, int ti, u32x4 (&w)[NS], uint32_t (&sc)[NS], uint32_t (&zp)[AWQ ? NS : 1]) {
    const int unit_l = u0 + min(ui, nu - 1);
    const int unit = PSEQ ? unit_l >> 1 : unit_l;   // the 16-column block inside its tensor
    const bool up = PSEQ && (unit_l & 1);           // (wave-uniform: scalar selects of the kernel arguments, no indexed access)
    const int kt = kt0 + min(wave + GW_WAVES * ti, KTL - 1);
    const int grp = (kt * 128) >> gsh;
    const int ucol = a.marlin ? ((unit >> 2) << 6) + ((unit & 3) << 1) : unit * a.s_unit_stride;  // (even)
    const uint32_t so_w = (uint32_t)(unit * KT + kt) * 1024u;
    const uint32_t so_s = (uint32_t)(grp * a.s_grp_stride + ucol) * 2u;
    const uint32_t so_z = (uint32_t)(grp * a.z_grp_stride + unit * a.z_unit_stride) * 4u;
#pragma unroll
    for (int b = 0; b < NS; b++) {
      const void* wp = PSEQ ? (up ? a.w[1] : a.w[0]) : a.w[b];
      const void* sp = PSEQ ? (up ? a.scales[1] : a.scales[0]) : a.scales[b];
      const uint32_t* zq = PSEQ ? (up ? a.zeros[1] : a.zeros[0]) : a.zeros[b];
      w[b] = __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wp), 0, 0x7FFFFFF0, 0x00020000), vo_w, so_w, 2);  // nt
      sc[b] = __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sp), 0, 0x7FFFFFF0, 0x00020000), vo_s, so_s, 0);
      if (AWQ) zp[AWQ ? b : 0] = __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(zq), 0, 0x7FFFFFF0, 0x00020000), vo_z, so_z, 0);
    }
  };
#pragma unroll
  for (int ti = 0; ti < GW_TPW; ti++) issue(0, ti, wb[ti], sb[ti], zb[ti]);
  if ((has_res || any_bias) && owner && !late_epi) request_epilogue_operands();
  GEMV_STAMP(1);

  __builtin_amdgcn_sched_barrier(0);  // (phase boundaries are scheduling barriers: hipcc otherwise interleaves the phases for
                                      // ILP and the prologue, not the main loop, sets the register peak)
  // ---- x fragments of this wave's two k-tiles: lane (oct, nn) holds row mt*16 + nn, columns kt*128 + j*32 + oct*8 ..
  constexpr bool norm = NORM;
  u32x4 xf[GW_TPW][4][MT];
#pragma unroll
  for (int ti = 0; ti < GW_TPW; ti++) {
    // (a wave without this k-tile re-reads the last one: its scale is zeroed in the main loop and its Σx² share below — a
    // select on the loaded fragments would double their registers)
    const int kt = kt0 + min(wave + GW_WAVES * ti, KTL - 1);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const uint16_t* xr = static_cast<const uint16_t*>(a.x) + (size_t)(row0 + min(mt * 16 + nn, M - 1)) * a.x_ld + kt * 128 + oct * 8;
      // (odd rows fetch the two 64-byte halves of a 128-byte line in the opposite order and swap the registers afterwards:
      // with all 16 rows of a wave-load at the SAME offset inside their lines — the row stride is a multiple of 128 bytes —
      // the L1 served them at half rate; any row stride that is an odd multiple of 16..64 bytes measured 2.3..2.8 µs faster
      // per launch at 32 rows, and this is that effect without touching the layout)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if constexpr (XF) xf[ti][j][mt] = static_cast<const u32x4*>(a.x_frag)[(size_t)(((kt * 2 + mt) * 4 + j) * 64) + lane];
        else xf[ti][j][mt] = *reinterpret_cast<const u32x4*>(xr + (j ^ (nn & 1)) * 32);
      }
    }
  }

  __builtin_amdgcn_sched_barrier(0);
  GEMV_STAMP(2);
  if ((has_res || any_bias) && owner && !late_epi) stage_epilogue_operands();
  if (!XF && (nn & 1)) {
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
  // ---- fused RMSNorm.  Σx² and Σx come from the MATRIX cores: with A = B = a wave's x fragment the MFMA returns X·Xᵀ, whose
  // diagonal is the rows' Σx² (bf16 x bf16 products are exact in f32); with B = ones it returns the rows' Σx in exactly the
  // D layout the fix-up reads (lane (oct, ·): rows oct*4 .. +3).  The round-1 VALU form (convert + fma per element, convert +
  // add + two ds_bpermute per tile) was ~1250 VALU instructions per wave at 32 rows — x 2 waves per SIMD x 4 cycles: the
  // launch was VALU-bound BEFORE its first weight MFMA (tools/gemv_w_ts.py: 10 of 18.6 µs of the q/k/v launch).
  const bool pre = NORM && a.x_sq != nullptr;  // the fragments already hold x̃ = round(h * g): no normalisation here
  if (norm && pre) {
    // the producers' partial sums of squares -> `part` (what the epilogue reads), fixed order: wave w adds the slots
    // [32w, 32w + 32) — lane (r4 = lane & 7, pl = lane >> 3) rows 4*r4 .. +3 of the slots 32w + pl + 8j, j ascending — then the
    // eight pl of a wave by xor-shuffles (8, 16, 32)
    const int r4 = lane & 7, pl = lane >> 3;
    f32x4 sq[GW_PRE_PARTS / 64];
#pragma unroll
    for (int j = 0; j < GW_PRE_PARTS / 64; j++)
      sq[j] = *reinterpret_cast<const f32x4*>(a.x_sq + (size_t)(wave * (GW_PRE_PARTS / 8) + pl + 8 * j) * 32 + r4 * 4);
    f32x4 t = sq[0];
#pragma unroll
    for (int j = 1; j < GW_PRE_PARTS / 64; j++) t += sq[j];
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; e++) t[e] += __shfl_xor(t[e], o, 64);
    }
    if (pl == 0) *reinterpret_cast<f32x4*>(part + wave * 32 + r4 * 4) = t;
  }
  if (norm && !pre) {
    f32x4 g2[GW_TPW][MT];
#pragma unroll
    for (int ti = 0; ti < GW_TPW; ti++)  // (a chain per tile: a tile this wave does not have is masked on the VALU side)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) g2[ti][mt] = vra_zero_acc();
#pragma unroll
    for (int ti = 0; ti < GW_TPW; ti++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) DT::mfma(g2[ti][mt], __builtin_bit_cast(s16x8, xf[ti][j][mt]), __builtin_bit_cast(s16x8, xf[ti][j][mt]));
    VRA_MFMA_DRAIN();
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {  // the diagonal: lane (oct, nn) holds rows oct*4 + e of column nn -> row nn sits in lane (nn >> 2, nn), e = nn & 3
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
    GEMV_STAMP(3);
    __syncthreads();
    GEMV_STAMP(4);
    float rstd[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < GW_WAVES; w++) tot += part[w * 32 + mt * 16 + nn];
      rstd[mt] = 1.0f / sqrtf(tot / (float)a.K + a.eps);
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
  GEMV_STAMP(5);
  {  // Σx of rows (mt, oct*4 .. +3) over tile ti, over the ROUNDED values the weight MFMAs will see -> this wave's LDS table
    const uint32_t one2 = (uint32_t)DT::from_f32(1.0f) * 0x10001u;
    u32x4 ones = u32x4{one2, one2, one2, one2};
    asm volatile("" : "+v"(ones));
    f32x4 sx[GW_TPW][MT];
#pragma unroll
    for (int ti = 0; ti < GW_TPW; ti++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) sx[ti][mt] = vra_zero_acc();
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int ti = 0; ti < GW_TPW; ti++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) DT::mfma(sx[ti][mt], __builtin_bit_cast(s16x8, xf[ti][j][mt]), __builtin_bit_cast(s16x8, ones));
    VRA_MFMA_DRAIN();
    if (nn == 0) {
#pragma unroll
      for (int ti = 0; ti < GW_TPW; ti++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) *reinterpret_cast<f32x4*>(xsum + ti * 32 + mt * 16 + oct * 4) = sx[ti][mt];
    }
  }
  GEMV_STAMP(6);
  // (no workgroup barrier here: the Σx table is private to the wave — DS operations of one wave execute in order — and the
  // staged epilogue operands are first read behind the barrier of the first unit)

  // ---- main loop: one unit = four tile-steps (ring slots 0..3)
  const int zsh = 4 * awq_rev(nn & 7);
  const bool shalf = a.marlin ? (nn >> 3) & 1 : nn & 1;
  constexpr float CB = Magic<DT>::bias;
  f32x4 acc[NS][MT];
  for (int ui = 0; ui < nu; ui++) {
#pragma unroll
    for (int b = 0; b < NS; b++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) acc[b][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ti = 0; ti < GW_TPW; ti++) {
      const bool valid = wave + GW_WAVES * ti < KTL;
#pragma unroll
      for (int b = 0; b < NS; b++) {  // stream by stream: only the MT accumulators of one stream's tile are live at a time
        f32x4 ag[MT];  // (every chain STARTS with the C = 0 form of the MFMA: no accumulator is zeroed on the VALU)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const s16x8 bf = magic_word<DT>(wb[ti][b][j]);  // dequantised ONCE for all m-tiles
#pragma unroll
          for (int mt = 0; mt < MT; mt++) {
            if (j == 0) DT::mfma0(ag[mt], __builtin_bit_cast(s16x8, xf[ti][j][mt]), bf);
            else DT::mfma(ag[mt], __builtin_bit_cast(s16x8, xf[ti][j][mt]), bf);
          }
        }
        VRA_MFMA_DRAIN();
        float s = DT::to_f32((uint16_t)(shalf ? sb[ti][b] >> 16 : sb[ti][b]));
        s = valid ? s : 0.f;
        const float zc = AWQ ? CB + (float)((zb[ti][AWQ ? b : 0] >> zsh) & 0xFu) : CB + 8.f;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          const f32x4 sx = *reinterpret_cast<const f32x4*>(xsum + ti * 32 + mt * 16 + oct * 4);  // rows mt*16 + oct*4 .. +3
#pragma unroll
          for (int e = 0; e < 4; e++) acc[b][mt][e] = fmaf(s, fmaf(-zc, sx[e], ag[mt][e]), acc[b][mt][e]);
        }
      }
      issue(ui + 1, ti, wb[ti], sb[ti], zb[ti]);  // the same tile of the next unit (clamped: re-reads the last unit, never consumed)
    }
    GEMV_STAMP(8 + 3 * min(ui, 1));
    // ---- the unit's partial tiles meet in LDS (double-buffered by unit parity: one barrier per unit)
    f32x4* rbuf = red + (size_t)(ui & 1) * GW_WAVES * NS * MT * 64;
#pragma unroll
    for (int b = 0; b < NS; b++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) rbuf[((wave * NS + b) * MT + mt) * 64 + lane] = acc[b][mt];
    __syncthreads();
    GEMV_STAMP(9 + 3 * min(ui, 1));
    if (tid < OPU) {
      const int row = tid >> 4, col = tid & 15, mt = row >> 4;
      const int dl = (((row & 15) >> 2) << 4) + col, r = row & 3;  // D layout: lane = (row/4)*16 + column, register = row % 4
      const float* rf = reinterpret_cast<const float*>(rbuf);
      float v = 0.f, v2 = 0.f;
#pragma unroll
      for (int w = 0; w < GW_WAVES; w++) {
        v += rf[((((w * NS + 0) * MT + mt) * 64) + dl) * 4 + r];
        if (NS == 2) v2 += rf[((((w * NS + 1) * MT + mt) * 64) + dl) * 4 + r];
      }
      if constexpr (KZ) {  // the slice's partial tile waits in LDS; exchange and epilogue run after the stream
        outf[ui * OPU + tid] = v;
      } else {
        if (NORM && pre) {  // the deferred normalisation factor of this row, on the f32 dot products
          float tot = 0.f;
#pragma unroll
          for (int w = 0; w < GW_WAVES; w++) tot += part[w * 32 + row];
          const float rstd = 1.0f / sqrtf(tot / (float)a.K + a.eps);
          v *= rstd;
          if (NS == 2) v2 *= rstd;
        }
        const float bias = DT::to_f32(biass[(ui * NS + 0) * 16 + col]);
        v = rnd_dt<DT>(v);
        if (any_bias) v = rnd_dt<DT>(v + bias);  // (a segment without a bias was staged as +0: adding it changes nothing but the sign of -0)
        if (NS == 2) {
          v2 = rnd_dt<DT>(v2);
          if (a.seg[1].bias) v2 = rnd_dt<DT>(v2 + DT::to_f32(biass[(ui * NS + 1) * 16 + col]));
          const float sl = rnd_dt<DT>(v / (1.0f + expf(-v)));
          v = sl * v2;
        }
        if (has_res) v = rnd_dt<DT>(v) + DT::to_f32(ress[ui * OPU + tid]);
        if (PSEQ && (ui & 1)) {  // the up unit: its gate value was parked by THIS thread one unit ago (u0 is even)
          const float gt = DT::to_f32(outs[(ui - 1) * OPU + tid]);
          v = rnd_dt<DT>(gt / (1.0f + expf(-gt))) * v;
        }
        outs[ui * OPU + tid] = DT::from_f32(v);
      }
    }
    GEMV_STAMP(10 + 3 * min(ui, 1));
  }
  __syncthreads();
  GEMV_STAMP(14);
  if constexpr (KZ) {
    // ---- the K slices of a unit group meet through memory (kernel C's exchange, gemm_q4.cuh: the XCDs' L2s are not coherent,
    // so partials leave as write-through stores, every slice raises its own flag line once they are acknowledged, and the last
    // slice polls the flags, sums the slabs with agent-scope loads in slice order — deterministic — and resets the flags).
    // Only owners wait and there are kz_groups < CUs of them: some non-owner is always resident and never waits.
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, 0x7FFFFFF0, 0x00020000);
    uint32_t* fl = a.counters + (size_t)(rbi * a.kz_groups + wg) * a.kz * 16;
    const int zbase = rbi * a.kz;  // slab index of this row block's slice 0
    const int n4 = nu * OPU / 4;
    if (!owner) {
      for (int i4 = tid; i4 < n4; i4 += GW_THREADS) {
        const int idx = i4 * 4, ui = idx / OPU, rem = idx - ui * OPU;
        const f32x4 pv = *reinterpret_cast<const f32x4*>(outf + idx);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pv), srs, (uint32_t)((((zbase + zi) * a.n_units + u0 + ui) * OPU + rem) * 4), 0, 16);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-through stores: acknowledged by memory
      __syncthreads();
      if (tid == 0) __hip_atomic_store(fl + zi * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      GEMV_STAMP(15);
      return;
    }
    if (tid < a.kz - 1) {
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
    GEMV_STAMP(7);
    constexpr int ZMAX = 7;  // other slices (kz <= 8): all their loads in flight at once
    for (int i4 = tid; i4 < n4; i4 += GW_THREADS) {
      const int idx = i4 * 4, ui = idx / OPU, rem = idx - ui * OPU, col = rem & 15;
      u32x4 pz[ZMAX];
#pragma unroll
      for (int z = 0; z < ZMAX; z++)
        pz[z] = __builtin_amdgcn_raw_buffer_load_b128(srs, (uint32_t)((((zbase + min(z, a.kz - 2)) * a.n_units + u0 + ui) * OPU + rem) * 4), 0, 16);
      // (row-blocked launches: bias and residual of the 4 outputs straight from memory — single segment: column = unit * 16 + col)
      u32x2 lres = {0u, 0u}, lbias = {0u, 0u};
      if (late_epi) {
        const int row = rem >> 4, n = (u0 + ui) * 16 + col;
        if (has_res && row < M) lres = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(a.residual) + (size_t)(row0 + row) * a.res_ld + n);
        if (any_bias) lbias = *reinterpret_cast<const u32x2*>(static_cast<const uint16_t*>(a.seg[0].bias) + n);
      }
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int z = 0; z < ZMAX; z++)
        if (z < a.kz - 1) v += __builtin_bit_cast(f32x4, pz[z]);
      v += *reinterpret_cast<const f32x4*>(outf + idx);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float t = rnd_dt<DT>(v[e]);
        const uint16_t bv = late_epi ? (uint16_t)(lbias[e >> 1] >> (16 * (e & 1))) : biass[ui * 16 + col + e];
        const uint16_t rv = late_epi ? (uint16_t)(lres[e >> 1] >> (16 * (e & 1))) : (has_res ? ress[idx + e] : (uint16_t)0);
        if (any_bias) t = rnd_dt<DT>(t + DT::to_f32(bv));
        if (has_res) t = rnd_dt<DT>(t) + DT::to_f32(rv);
        outs[idx + e] = DT::from_f32(t);
      }
    }
    __syncthreads();
  }
  // ---- everything is stored after the stream has ended
  for (int idx = tid; idx < nu * OPU; idx += GW_THREADS) {
    const int ui = __builtin_amdgcn_readfirstlane(idx / OPU), rem = idx - ui * OPU, row = rem >> 4, col = rem & 15;
    if (row >= M || (PSEQ && !(ui & 1))) continue;
    void* o_;
    const void* b_;
    int ld_, c0;
    seg_of(u0 + ui, o_, b_, ld_, c0);
    static_cast<uint16_t*>(o_)[(size_t)(row0 + row) * ld_ + c0 + col] = outs[idx];
  }
  if (a.out_frag && gridDim.y == 1) {  // the same outputs in fragment order: 8 columns = one 16-byte word
    for (int i8 = tid; i8 < nu * OPU / 8; i8 += GW_THREADS) {
      const int idx = i8 * 8, ui = idx / OPU, rem = idx - ui * OPU, row = rem >> 4, col = rem & 15;
      if (row >= M || (PSEQ && !(ui & 1))) continue;
      const int n = (PSEQ ? (u0 + ui) >> 1 : u0 + ui) * 16 + col;  // (single-segment launch: the launch-wide column is the output column)
      static_cast<u32x4*>(a.out_frag)[(size_t)((((n >> 7) * 2 + (row >> 4)) * 4 + ((n >> 5) & 3)) * 64) + ((n >> 3) & 3) * 16 + (row & 15)] =
          *reinterpret_cast<const u32x4*>(outs + idx);
    }
  }
  if (a.pre_norm_w && gridDim.y == 1 && !PSEQ) {
    // ---- ready-made operands for the next fused-norm launch (GemvSArgs::pre_*): x̃ = round(out * g_next) in fragment order ...
    for (int i8 = tid; i8 < nu * OPU / 8; i8 += GW_THREADS) {
      const int idx = i8 * 8, ui = idx / OPU, rem = idx - ui * OPU, row = rem >> 4, col = rem & 15;
      if (row >= M) continue;
      const int n = (u0 + ui) * 16 + col;
      float f[8], g[8];
      unpack8<DT>(*reinterpret_cast<const u32x4*>(outs + idx), f);
      unpack8<DT>(*reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.pre_norm_w) + n), g);
#pragma unroll
      for (int e = 0; e < 8; e++) f[e] *= g[e];
      static_cast<u32x4*>(a.pre_frag)[(size_t)((((n >> 7) * 2 + (row >> 4)) * 4 + ((n >> 5) & 3)) * 64) + ((n >> 3) & 3) * 16 + (row & 15)] = pack8<DT>(f);
    }
    // ... and this workgroup's share of the rows' sums of squares (of the ROUNDED outputs, what the reference's norm would read):
    // thread r adds the columns of row r in order; rows beyond M: 0
    if (tid < 32) {
      float s2 = 0.f;
      if (tid < M) {
        for (int ui = 0; ui < nu; ui++) {
#pragma unroll
          for (int c8 = 0; c8 < 2; c8++) {
            float f[8];
            unpack8<DT>(*reinterpret_cast<const u32x4*>(outs + ui * OPU + tid * 16 + c8 * 8), f);
#pragma unroll
            for (int e = 0; e < 8; e++) s2 = fmaf(f[e], f[e], s2);
          }
        }
      }
      a.pre_sq[(size_t)wg * 32 + tid] = s2;
    }
  }
  GEMV_STAMP(15);
  GEMV_STAMP_FLUSH();
}
