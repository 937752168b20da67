// (experiments/: NOT in the product library — measured no faster than kernel D, profiles/r06_kernel_d_probes.txt)
// gemm_q4_big2.cuh — "kernel D2": kernel D's tile (a wave = 4 n-blocks x 4 m-tiles, 4 waves side by side in n, two workgroups per CU)
// with its instruction stream SOFTWARE-PIPELINED inside the wave (round 6).
//
// Roofline: MFMA.  Why: kernel D runs a k-tile in phases — 112 VALU of int4 -> bf16 conversion for all 16 weight fragments, then 64
// MFMAs with the scale fix-up between the m-tiles — and its phase timers (tools/gemm_big_ts.py, profiles/r06_kernel_d_phases.txt) read
// 1203 | 2102 | 198 | 322 cycles per k-tile and wave for conversion + issue | MFMA steps | x store | barrier = 3826, two waves per SIMD:
// the SIMD spends 2 x 64 x 17 = 2176 cycles of that in the matrix pipe (57 %) and the rest in VALU phases that run beside NOTHING — the
// two co-resident waves do not overlap one's VALU with the other's MFMAs (tools/mfma_valu_probe.hip, round 4), and removing half of the
// fix-up (round 6: the zero-point term hoisted) or un-packing its v_pk_fma_f32 moved the kernel by < 1 %.  What does hide under an MFMA is
// independent work of the SAME wave issued right behind it: a 16x16x32 MFMA holds the pipe ~17 cycles, about four issue slots
// (MI355X_MICROARCH.md, per-instruction constants: "single-issue instructions hidden per MFMA gap").
//
// So every k-tile is 4 steps j (the 32-wide k slices of the tile), 16 MFMAs each (m-tile x n-block), and the work that used to be
// phases rides in the gaps as `fillers`, in program order pinned by sched_barrier:
//   * the conversion of word j+1 of the tile's four weight words (word 0 of the NEXT tile in step 3): 7 VALU per n-block = one or two
//     per gap, into the other half of a two-deep fragment buffer (32 registers instead of kernel D's 64);
//   * the scale fix-up  acc += s * acc_g  of the tile's m-tiles 0, 1 in the second half of step 3 and of m-tiles 2, 3 in the first half
//     of the next tile's step 0 — each behind the last MFMA that wrote its group accumulator, in front of the first one that clears it
//     (the group accumulators of all 16 (n-block, m-tile) pairs are live: 64 registers; the order inside a tile is j-major);
//   * the next tile's scales (8 conversions, step 1), the x fragment reads from LDS (one per m-tile, two ahead), the global loads of
//     the next weight tile and of the x tile two ahead (tile start), the x tile's store into LDS (tile end, one barrier per tile).
// Arithmetic: exactly kernel D's ZH form (gemm_q4_big.cuh: GPTQ symmetric + bf16, the zero-point term as one MFMA correction behind the
// loop); same tiles, same k order, same f32 operations per output — results are bit-identical to kernel D's, which the parity tests
// check (tests/test_gpu_kernels.py::test_gemm_q4_big2_matches_kernel_d).  Everything else (AWQ, f16, 2 m-tiles, split-K) stays on kernel D.
#pragma once
#include "gemm_q4_big.cuh"

#define GD2_THREADS 256

// MFMAs without the conservative `s_nop 1` of DT::mfma: every operand of these is written at least one step (16 MFMAs) earlier
__device__ __forceinline__ void gd2_mfma(f32x4& acc, s16x8 a, s16x8 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void gd2_mfma0(f32x4& acc, s16x8 a, s16x8 b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
}

template <bool DUAL>
__global__ __launch_bounds__(GD2_THREADS, 2) void gemm_q4_big2_kernel(const GemmDArgs a) {
  typedef BF16 DT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NB = 4, MB = 4;
  constexpr int ROWS = 16 * MB;
  constexpr int RPP = GD2_THREADS / 16;
  constexpr int RS = (16 + 1) * 4;
  constexpr int XS_U32 = ROWS * RS;
  constexpr int XPT = ROWS * 16 / GD2_THREADS;  // 4
  constexpr uint32_t RSRC3 = 0x00020000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nn = lane & 15, oct = lane >> 4;
  const int wn = wave;
  const int K = a.K, M = a.M, KT = K >> 7;
  const bool grouped = a.group_size > 0 && a.group_size < K;
  const int gsh = grouped ? 31 - __builtin_clz(a.group_size) : 31;
  const int m0 = (int)blockIdx.y * ROWS;
  uint32_t* xs = reinterpret_cast<uint32_t*>(smem);  // [3][ROWS][RS]

  // ---- this wave's tensor(s) and n-blocks (wave-uniform), as kernel D
  const void* wt[2] = {a.w0, a.w1};
  const void* sct[2] = {a.sc0, a.sc1};
  const void* biast[2] = {a.bias0, a.bias1};
  void* outp = a.out;
  int N = a.N, out_ld = a.out_ld;
  int nb0;
  if (DUAL) {
    nb0 = (int)blockIdx.x * 8 + wn * 2;
  } else {
    nb0 = (int)blockIdx.x * 16 + wn * 4;
    if (a.nseg > 1) {
      const int s = (a.nseg > 2 && nb0 >= a.xseg[1].blk_start) ? 1 : (nb0 >= a.xseg[0].blk_start ? 0 : -1);
      if (s >= 0) {
        const GemvSeg& sg = a.xseg[s];
        wt[0] = sg.w, sct[0] = sg.scales, biast[0] = sg.bias, outp = sg.out, N = sg.n, out_ld = sg.out_ld;
        nb0 -= sg.blk_start;
      }
    }
  }
  auto tb = [&](int b) { return DUAL ? (b >> 1) : 0; };
  auto nbv = [&](int b) { return nb0 + (DUAL ? (b & 1) : b); };
  bool ok[NB];
  int nbc[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    ok[b] = nbv(b) * 16 < N;
    nbc[b] = ok[b] ? nbv(b) : 0;
  }
  constexpr int NT = DUAL ? 2 : 1;
  __amdgpu_buffer_rsrc_t rw[NT], rsc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) {
    rw[t] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(wt[t]), 0, 0x7FFFFFF0, RSRC3);
    rsc[t] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(sct[t]), 0, 0x7FFFFFF0, RSRC3);
  }
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, 0x7FFFFFF0, RSRC3);
  const __amdgpu_buffer_rsrc_t rsum = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xsum), 0, 0x7FFFFFF0, RSRC3);
  const uint32_t vo_w = (uint32_t)lane * 16u;
  const uint32_t vo_s = (uint32_t)oct * 8u;
  const int Nt = DUAL ? a.N : N;
  uint32_t vo_x[XPT];
#pragma unroll
  for (int r = 0; r < XPT; r++) vo_x[r] = ((uint32_t)min(m0 + r * RPP + (tid >> 4), M - 1) * (uint32_t)a.x_ld + (uint32_t)(tid & 15) * 8u) * 2u;

  auto x_load = [&](int kt, u32x4 (&xr)[XPT]) {
#pragma unroll
    for (int r = 0; r < XPT; r++) xr[r] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo_x[r], (uint32_t)(kt * 256), 0);
  };
  auto x_store = [&](int buf, const u32x4 (&xr)[XPT]) {
    uint32_t* dst = xs + (size_t)buf * XS_U32 + (size_t)(tid >> 4) * RS + (tid & 15) * 4;
#pragma unroll
    for (int r = 0; r < XPT; r++) *reinterpret_cast<u32x4*>(dst + (size_t)(r * RPP) * RS) = xr[r];
  };
  auto w_load = [&](int kt, u32x4 (&wq)[NB]) {
#pragma unroll
    for (int b = 0; b < NB; b++) wq[b] = __builtin_amdgcn_raw_buffer_load_b128(rw[tb(b)], vo_w, (uint32_t)((nbc[b] * KT + kt) * 1024), 2);  // nt
  };
  auto s_load = [&](int kt, u32x2 (&sc)[NB]) {
    const int grp = grouped ? (kt * 128) >> gsh : 0;
#pragma unroll
    for (int b = 0; b < NB; b++)
      sc[b] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsc[tb(b)], vo_s, (uint32_t)((grp * Nt + nbc[b] * 16) * 2), 0));
  };

  f32x4 acc[NB][MB], ag[NB][MB];
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int t = 0; t < MB; t++) acc[b][t] = f32x4{0.f, 0.f, 0.f, 0.f}, ag[b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kt0 = 0, kt1 = KT;

  // conversion of ONE word of a tile's weights in four parts (1 + 2 + 2 + 2 VALU): part p fills register p of the fragment
  uint32_t mk = Magic<DT>::mask, bt = Magic<DT>::bits;
  asm("" : "+v"(mk));
  asm("" : "+v"(bt));
  auto deq_part = [&](uint32_t w, int part, u32x4& dst) { dst[part] = ((part ? w >> (4 * part) : w) & mk) | bt; };

  u32x4 wq[2][NB];   // the weight words of the tile in flight and of the next one
  u32x2 scw[NB];     // raw scales of the NEXT tile
  float sc[NB][4];   // scales of the tile whose fix-up is pending
  u32x4 af[2][NB];   // fragment buffer: word j (even steps -> [0], odd -> [1])
  {
    u32x4 x0[XPT], x1[XPT];
    x_load(kt0, x0);
    x_load(min(kt0 + 1, kt1 - 1), x1);
    w_load(kt0, wq[0]);
    s_load(kt0, scw);
    x_store(0, x0);
    x_store(1, x1);
#pragma unroll
    for (int b = 0; b < NB; b++) {
#pragma unroll
      for (int p = 0; p < 4; p++) deq_part(wq[0][b][0], p, af[0][b]);
      sc[b][0] = DT::to_f32((uint16_t)(scw[b][0] & 0xffffu)), sc[b][1] = DT::to_f32((uint16_t)(scw[b][0] >> 16));
      sc[b][2] = DT::to_f32((uint16_t)(scw[b][1] & 0xffffu)), sc[b][3] = DT::to_f32((uint16_t)(scw[b][1] >> 16));
    }
  }
  __syncthreads();

  // the scale fix-up of one (n-block, m-tile): acc += s * acc_g (scalar v_fma_f32: packed f32 VALU beside MFMAs costs more than its halves)
  auto fixup = [&](int b, int mt) {
#ifdef GD2_PROBE_NO_FIX
    if (kt1 > 0) return;
#endif
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float u = __builtin_fmaf(sc[b][r], ag[b][mt][r], acc[b][mt][r]);
      asm volatile("" : "+v"(u));
      acc[b][mt][r] = u;
    }
  };

  // ---- one k-tile; P = parity of the tile (which half of wq holds it), compile time
  auto tile = [&](int kt, auto parity) {
    constexpr int P = decltype(parity)::value;
    const int buf = (kt - kt0) % 3;
    const int ktn = min(kt + 1, kt1 - 1);
    // the x tile two ahead travels in two halves (rows 0..31 during steps 0-1, rows 32..63 during steps 2-3: 8 staging registers
    // instead of 16 — its LDS buffer is nobody's until the next barrier); the next weight tile is requested at step 1, when half of
    // this tile's words are dead (the register peak, not the latency, decides: 256 VGPRs = two workgroups per CU)
    const int kx = min(kt + 2, kt1 - 1);
    const int xbuf = (kt - kt0 + 2) % 3;
    u32x4 xh[2];
    auto xh_load = [&](int h) {
#pragma unroll
      for (int r = 0; r < 2; r++) xh[r] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo_x[2 * h + r], (uint32_t)(kx * 256), 0);
    };
    auto xh_store = [&](int h) {
      uint32_t* dst = xs + (size_t)xbuf * XS_U32 + (size_t)(tid >> 4) * RS + (tid & 15) * 4;
#pragma unroll
      for (int r = 0; r < 2; r++) *reinterpret_cast<u32x4*>(dst + (size_t)((2 * h + r) * RPP) * RS) = xh[r];
    };
    xh_load(0);
    const uint32_t* xb = xs + (size_t)buf * XS_U32 + (size_t)nn * RS + oct * 4;
    auto frag = [&](int j, int mt) { return *reinterpret_cast<const u32x4*>(xb + (size_t)(mt * 16) * RS + j * 16); };
    u32x4 xv[3];
    xv[0] = frag(0, 0), xv[1] = frag(0, 1);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int cur = j & 1, nxt = cur ^ 1;
#pragma unroll
      for (int mt = 0; mt < MB; mt++) {
        // the x fragment two m-tiles ahead (of this step, or of the next one)
        {
          const int s = j * 4 + mt + 2;
          if (s < 16) xv[s % 3] = frag(s >> 2, s & 3);
        }
        const s16x8 bfrag = __builtin_bit_cast(s16x8, xv[(j * 4 + mt) % 3]);
#pragma unroll
        for (int b = 0; b < NB; b++) {
          const int slot = mt * 4 + b;
          // ---- fillers of this gap
          {  // conversion: word j+1 of this tile, or word 0 of the next one
            const int fb = slot >> 2, part = slot & 3;
            const uint32_t w = j < 3 ? wq[P][fb][j + 1] : wq[P ^ 1][fb][0];
#ifndef GD2_PROBE_NO_DEQ  // (probe builds, wrong results: what the kernel costs without its conversion / fix-up work)
            deq_part(w, part, af[nxt][fb]);
#else
            if (kt < 0) deq_part(w, part, af[nxt][fb]);
#endif
          }
          if (j == 0 && slot < 8) fixup(slot & 3, 2 + (slot >> 2));   // previous tile's m-tiles 2, 3, before slots 8..15 clear them (first tile: + s * 0)
          if (j == 3 && slot >= 8) fixup(slot & 3, (slot - 8) >> 2);            // this tile's m-tiles 0, 1: written in slots 0..7 of this step
          if (j == 1 && slot < 8) {                                            // the scales of THIS tile replace the previous tile's (its fix-ups are done)
            const int sb = slot >> 1, h = slot & 1;
            sc[sb][2 * h] = DT::to_f32((uint16_t)(scw[sb][h] & 0xffffu));
            sc[sb][2 * h + 1] = DT::to_f32((uint16_t)(scw[sb][h] >> 16));
          }
          if (j == 1 && slot == 0) w_load(ktn, wq[P ^ 1]);                     // (the next tile's words: first used in step 3)
          if (j == 2 && slot == 0) {
            s_load(ktn, scw);  // (raw scales of the next tile: consumed in its step 1)
            xh_store(0);
            xh_load(1);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (j == 0) gd2_mfma0(ag[b][mt], __builtin_bit_cast(s16x8, af[cur][b]), bfrag);
          else gd2_mfma(ag[b][mt], __builtin_bit_cast(s16x8, af[cur][b]), bfrag);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    xh_store(1);
    __syncthreads();
  };
  // the first tile's scales are already in `sc` (prologue): its step 1 must not re-read scw before the s_load of step 2 —
  // scw still holds tile kt0's raw scales there, so the conversion is idempotent for the first tile
  int kt = kt0;
  for (; kt + 1 < kt1; kt += 2) {
    tile(kt, std::integral_constant<int, 0>{});
    tile(kt + 1, std::integral_constant<int, 1>{});
  }
  if (kt < kt1) tile(kt, std::integral_constant<int, 0>{});
  // (the last tile's step-3 conversions feed nothing; kept alive so that every instance of the tile has the same fillers between an
  // MFMA and the fix-up that reads its result — hipcc deleted them from the tail instance and tools/check_mfma_overlap.py, which counts
  // wait states per instruction like LLVM's hazard recogniser, flagged the fix-up 9 states behind its MFMA)
#pragma unroll
  for (int b = 0; b < NB; b++) asm volatile("" ::"v"(af[0][b]), "v"(af[1][b]));
  // m-tiles 2, 3 of the last tile
  VRA_MFMA_DRAIN();
#pragma unroll
  for (int b = 0; b < NB; b++) {
    fixup(b, 2);
    fixup(b, 3);
  }

  // ---- the zero-point term (gemm_q4_big.cuh ZH): acc -= (C + 8) * Σ_t s[grp(t)][n] * Σx_t[m], 32 k-tiles per MFMA step
  {
    const uint32_t vo_sn = (uint32_t)nn * 2u;
    uint32_t vo_sum[MB];
#pragma unroll
    for (int mt = 0; mt < MB; mt++) vo_sum[mt] = (uint32_t)min(m0 + mt * 16 + nn, M - 1) * (uint32_t)KT * 4u;
    for (int t0 = kt0; t0 < kt1; t0 += 32) {
      const int tb0 = t0 + oct * 8;
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
            const uint32_t hb = __float_as_uint(v) & 0xffff0000u;
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
          BF16::mfma0(zt[b], sa[b], __builtin_bit_cast(s16x8, pl));
          BF16::mfma(zt[b], sa[b], __builtin_bit_cast(s16x8, pm));
          BF16::mfma(zt[b], sa[b], __builtin_bit_cast(s16x8, ph));
        }
        VRA_MFMA_DRAIN();
        constexpr float NZC = -(Magic<BF16>::bias + 8.0f);
#pragma unroll
        for (int b = 0; b < NB; b++) {
          // (the same two v_pk_fma_f32 per tile as kernel D: bit-identical results)
          const f32x2 lo = __builtin_elementwise_fma(f32x2{NZC, NZC}, f32x2{zt[b][0], zt[b][1]}, f32x2{acc[b][mt][0], acc[b][mt][1]});
          const f32x2 hi = __builtin_elementwise_fma(f32x2{NZC, NZC}, f32x2{zt[b][2], zt[b][3]}, f32x2{acc[b][mt][2], acc[b][mt][3]});
          acc[b][mt] = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      }
    }
  }

  // ---- epilogue: D[column (lane>>4)*4 + r][row lane&15] of tile (b, mt), as kernel D
  constexpr int NBO = DUAL ? 2 : NB;
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
      const int m = m0 + mt * 16 + nn;
      if (m >= M) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float t = rnd_dt<DT>(acc[b][mt][r]);
        if (biast[0]) t = rnd_dt<DT>(t + bs[r]);
        if (DUAL) {
          float u = rnd_dt<DT>(acc[b + 2][mt][r]);
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
