// gemm_q4.cuh — "kernel C": int4 GEMM for 9..32 activation rows (decode batches) that streams every packed weight
// byte ONCE and dequantises it ONCE for all rows.
//
// Roofline: HBM.  Algorithmic bytes per call as for kernel A: K*N/2 + (K/g)*N*2 [+ AWQ zeros] + M*K*2 + M*N*2.
//
// Why a third kernel: kernel A (gemv_q4.cuh) keeps all of x in LDS (M*K*2 bytes: 256 KiB at M = 32, K = 4096) and can
// only take more rows by re-running the dequantisation per 8-row family (VALU bound); kernel B (gemm_skinny.cuh) splits K
// across workgroups (fp32 slabs + arrival counters) and synchronises its waves every 256 k.
//
// Shape:
//   workgroup = 8 COMPUTE waves + 4 PRODUCER waves, persistent over work items.
//   A work item = CG = 8/KS consecutive n-blocks (16 columns each; NBW = 2: the same blocks of gate AND up) over the full
//   K.  Compute wave w = (cg, ks) owns n-block cg of the item and the k-tiles kt ≡ ks (mod KS): a flat stream of T =
//   KT/KS tile-steps with a 2-deep register ring, branch-free, exact vmcnt (see gemv_q4.cuh for why that matters).
//   MT m-tiles (16 rows each) share every dequantised B fragment: 4*MT MFMAs per tile and tensor.
//   x reaches the MFMAs through LDS in K-chunks of KC = 512 or 1024 (double buffered, XOR-swizzled like kernel B), staged
//   by the producer waves — which have no weight loads, so their waits never touch the compute waves' ring — together
//   with the per-tile row sums Σx of the zero-point fix-up.  One barrier per chunk.
//   At the end of an item the KS partial tiles of every n-block meet in LDS and the producer waves run the fused epilogue
//   (bias, SiLU·mul, residual) and own all global stores.
#pragma once
#include "gemv.cuh"

#define GC_CW 8   // compute waves
#define GC_PW 4   // producer waves
#define GC_THREADS ((GC_CW + GC_PW) * 64)

struct GemmCArgs {
  GemvSeg seg[GEMV_MAX_SEG];
  int nseg;
  const void* x;  // [M, x_ld], already normalised (the RMSNorm is a separate launch above 8 rows)
  int x_ld;
  const void* residual;
  int res_ld;
  int M, K;
  int group_size;  // -1 => K ; >= 128 only (fine groups go to kernel B)
  int silu_dual, out_f32;
  int n_blocks;  // n-blocks (or gate/up pairs) in total
  int ks;        // k-split inside the workgroup: 1, 2, 4, 8
  int kc;        // k per staged chunk: 512 or 1024 (K % kc == 0, (kc/128) % ks == 0)
};

static inline size_t gemm_q4_lds_bytes(int nbw, int mt, int kc) {
  const int rows = 16 * mt;
  return (size_t)2 * rows * kc * 2 + (size_t)2 * (kc / 128) * rows * 4 + (size_t)GC_CW * nbw * mt * 64 * 16 + 64;
}

template <class DT, int NBW, int MT, bool AWQ>
__global__ __launch_bounds__(GC_THREADS) void gemm_q4_kernel(const GemmCArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROWS = 16 * MT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool is_prod = wave >= GC_CW;
  const int nn = lane & 15, oct = lane >> 4;
  const int K = a.K, M = a.M, KT = K >> 7;
  const int KC = a.kc, TPC = KC >> 7, NC = K / KC, OPC = KC >> 3;  // tiles, chunks, octets per chunk
  const int KS = a.ks, CGN = GC_CW / KS;
  const int SC = TPC / KS;        // steps of one compute wave per chunk
  const int T = KT / KS;          // steps per item
  const bool grouped = a.group_size > 0 && a.group_size < K;
  const int gsh = grouped ? 31 - __builtin_clz(a.group_size) : 31;
  const int m0 = (int)blockIdx.y * ROWS;
  const int n_items = (a.n_blocks + CGN - 1) / CGN;

  // ---- LDS
  const int XS_U32 = OPC * ROWS * 4;  // one x buffer in u32
  uint32_t* xs = reinterpret_cast<uint32_t*>(smem);
  float* xsum = reinterpret_cast<float*>(smem + (size_t)2 * XS_U32 * 4);  // [2][TPC][ROWS]
  f32x4* red = reinterpret_cast<f32x4*>(smem + (size_t)2 * XS_U32 * 4 + (size_t)2 * TPC * ROWS * 4);  // [GC_CW][NBW][MT][64]

  const int nseg = a.nseg;
  const int blk1 = nseg > 1 ? a.seg[1].blk_start : 0x7fffffff, blk2 = nseg > 2 ? a.seg[2].blk_start : 0x7fffffff;

  if (is_prod) {
    // ============================== producer waves: x chunks -> LDS, then the epilogue of every item
    const int pt = tid - GC_CW * 64;                   // 0..255
    constexpr int PTHREADS = GC_PW * 64;
    const int per = (ROWS * OPC) / PTHREADS;            // octets per thread and chunk (4 .. 16)
    const int osh = OPC == 128 ? 7 : 6;                 // OPC is 64 or 128
    auto stage = [&](int c, int buf) {
      uint32_t* dst = xs + (size_t)buf * XS_U32;
      float* sdst = xsum + (size_t)buf * TPC * ROWS;
      for (int r0 = 0; r0 < per; r0 += 8) {  // 8 loads in flight per lane (per is 4, 8 or 16)
        u32x4 v[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
          if (r0 + r < per) {
            const int i = pt + (r0 + r) * PTHREADS;
            const int row = i >> osh, o = i & (OPC - 1);
            const int m = min(m0 + row, M - 1);           // rows >= M alias row M-1 (never stored)
            v[r] = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.x) + (size_t)m * a.x_ld + (size_t)c * KC + o * 8);
          }
        }
#pragma unroll
        for (int r = 0; r < 8; r++) {
          if (r0 + r < per) {
            const int i = pt + (r0 + r) * PTHREADS;
            const int row = i >> osh, o = i & (OPC - 1);
            *reinterpret_cast<u32x4*>(dst + ((size_t)o * ROWS + (row ^ (o & 7))) * 4) = v[r];
            float f[8];
            unpack8<DT>(v[r], f);
            float s8 = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) s8 += __shfl_xor(s8, d, 64);   // 16 consecutive octets = one k-tile of one row
            if ((o & 15) == 0) sdst[(o >> 4) * ROWS + row] = s8;
          }
        }
      }
    };
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
      stage(0, 0);
      __syncthreads();
      for (int c = 1; c < NC; c++) {
        stage(c, c & 1);
        __syncthreads();
      }
      __syncthreads();  // the item's partial tiles are in `red`
      // ---- fused epilogue: CGN n-blocks x ROWS rows x 16 columns
      const int nout = CGN * ROWS * 16;
      for (int idx = pt; idx < nout; idx += PTHREADS) {
        const int cg = idx / (ROWS * 16), rem = idx - cg * ROWS * 16;
        const int mrow = rem >> 4, nl = rem & 15;
        const int fb = it * CGN + cg;
        const int m = m0 + mrow;
        if (fb >= a.n_blocks || m >= M) continue;
        const int segi = NBW == 2 ? 0 : (fb >= blk2 ? 2 : (fb >= blk1 ? 1 : 0));
        const int nb = fb - (NBW == 2 ? 0 : a.seg[segi].blk_start);
        const GemvSeg& sg = a.seg[segi];
        const int mt = mrow >> 4, mm = mrow & 15;
        const int lslot = ((mm >> 2) * 16 + nl) * 4 + (mm & 3);  // D layout: column = lane&15, row = (lane>>4)*4 + reg
        const float* rf = reinterpret_cast<const float*>(red);
        float v = 0.f, v2 = 0.f;
        for (int q = 0; q < KS; q++) {
          const int w = cg * KS + q;
          v += rf[(size_t)((w * NBW + 0) * MT + mt) * 256 + lslot];
          if (NBW > 1) v2 += rf[(size_t)((w * NBW + (NBW - 1)) * MT + mt) * 256 + lslot];
        }
        const int n = nb * 16 + nl;
        v = rnd_dt<DT>(v);
        if (sg.bias) v = rnd_dt<DT>(v + DT::to_f32(static_cast<const uint16_t*>(sg.bias)[n]));
        if (NBW == 2) {
          v2 = rnd_dt<DT>(v2);
          if (a.seg[1].bias) v2 = rnd_dt<DT>(v2 + DT::to_f32(static_cast<const uint16_t*>(a.seg[1].bias)[n]));
          const float sl = rnd_dt<DT>(v / (1.0f + expf(-v)));
          v = sl * v2;
        }
        if (a.residual) v = rnd_dt<DT>(v) + DT::to_f32(static_cast<const uint16_t*>(a.residual)[(size_t)m * a.res_ld + n]);
        if (a.out_f32) static_cast<float*>(sg.out)[(size_t)m * sg.out_ld + n] = rnd_dt<DT>(v);
        else static_cast<uint16_t*>(sg.out)[(size_t)m * sg.out_ld + n] = DT::from_f32(v);
      }
    }
    return;
  }

  // ============================== compute waves
  const int cg = wave / KS, ksi = wave - cg * KS;
  const int zsh = 4 * awq_rev(nn & 7);
  const bool shalf = nn & 1;
  constexpr float CB = Magic<DT>::bias;

  for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int fb = it * CGN + cg;
    const bool have = fb < a.n_blocks;
    const int fbc = have ? fb : a.n_blocks - 1;  // a wave without an n-block streams the last one with zeroed scales
    // wave-uniform tensor pointers of this item
    const u32x4* wp[NBW];
    const uint16_t* sp[NBW];
    const uint32_t* qp[NBW];
    int ncols[NBW], nbv;
    if (NBW == 2) {
      nbv = fbc;
#pragma unroll
      for (int b = 0; b < NBW; b++) {
        wp[b] = reinterpret_cast<const u32x4*>(a.seg[b].w) + (size_t)nbv * KT * 64 + lane;
        sp[b] = static_cast<const uint16_t*>(a.seg[b].scales);
        qp[b] = a.seg[b].qzeros;
        ncols[b] = a.seg[b].n;
      }
    } else {
      const bool s1 = fbc >= blk1, s2 = fbc >= blk2;
      const GemvSeg& sg = s2 ? a.seg[2] : (s1 ? a.seg[1] : a.seg[0]);
      nbv = fbc - (s2 ? blk2 : (s1 ? blk1 : 0));
      wp[0] = reinterpret_cast<const u32x4*>(sg.w) + (size_t)nbv * KT * 64 + lane;
      sp[0] = static_cast<const uint16_t*>(sg.scales);
      qp[0] = sg.qzeros;
      ncols[0] = sg.n;
    }
    u32x4 wb[2][NBW];
    uint32_t sb[2][NBW], zb[2][NBW];
    auto issue = [&](int i, u32x4 (&w)[NBW], uint32_t (&sc)[NBW], uint32_t (&zp)[NBW]) {
      const int kt = ksi + KS * min(i, T - 1);  // steps past the end re-read the last tile (never consumed)
#pragma unroll
      for (int b = 0; b < NBW; b++) {
        w[b] = __builtin_nontemporal_load(wp[b] + (size_t)kt * 64);
        const int grp = (kt * 128) >> gsh;
        const int64_t si = (int64_t)grp * ncols[b] + nbv * 16 + nn;
        sc[b] = reinterpret_cast<const uint32_t*>(sp[b])[si >> 1];
        if (AWQ) zp[b] = qp[b][(size_t)grp * (ncols[b] >> 3) + nbv * 2 + (nn >> 3)];
        else zp[b] = 0;
      }
    };
    issue(0, wb[0], sb[0], zb[0]);
    issue(1, wb[1], sb[1], zb[1]);

    f32x4 acc[NBW][MT];
#pragma unroll
    for (int b = 0; b < NBW; b++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) acc[b][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int T_pad = (T + 1) & ~1;
    for (int i0 = 0; i0 < T_pad; i0 += 2) {
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const int i = i0 + r;
        if (i < T && (i % SC) == 0) __syncthreads();  // chunk i/SC is staged (and chunk i/SC - 2's buffer is free)
        if (i < T) {
          const int kt = ksi + KS * i;
          const int c = kt / TPC, tl = kt - c * TPC;
          const uint32_t* xb = xs + (size_t)(c & 1) * XS_U32;
          const float* sxb = xsum + ((size_t)(c & 1) * TPC + tl) * ROWS;
          f32x4 ag[NBW][MT];
#pragma unroll
          for (int b = 0; b < NBW; b++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) ag[b][mt] = vra_zero_acc();
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int o = tl * 16 + j * 4 + oct;
            s16x8 bfrag[NBW];
#pragma unroll
            for (int b = 0; b < NBW; b++) bfrag[b] = magic_word<DT>(wb[r][b][j]);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
              const u32x4 xv = *reinterpret_cast<const u32x4*>(xb + ((size_t)o * ROWS + ((mt * 16 + nn) ^ (o & 7))) * 4);
              const s16x8 afrag = __builtin_bit_cast(s16x8, xv);
#pragma unroll
              for (int b = 0; b < NBW; b++) DT::mfma(ag[b][mt], afrag, bfrag[b]);
            }
          }
          VRA_MFMA_DRAIN();
#pragma unroll
          for (int b = 0; b < NBW; b++) {
            float s = DT::to_f32((uint16_t)(shalf ? sb[r][b] >> 16 : sb[r][b]));
            s = have ? s : 0.f;
            const float zc = AWQ ? CB + (float)((zb[r][b] >> zsh) & 0xFu) : CB + 8.f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
              const f32x4 sx = *reinterpret_cast<const f32x4*>(sxb + mt * 16 + oct * 4);
#pragma unroll
              for (int e = 0; e < 4; e++) acc[b][mt][e] = fmaf(s, fmaf(-zc, sx[e], ag[b][mt][e]), acc[b][mt][e]);
            }
          }
        }
        issue(i + 2, wb[r], sb[r], zb[r]);  // unconditional refill (clamped)
      }
    }
    // ---- hand the partial tiles to the producer waves
#pragma unroll
    for (int b = 0; b < NBW; b++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) red[(size_t)((wave * NBW + b) * MT + mt) * 64 + lane] = acc[b][mt];
    __syncthreads();
  }
}
