// gemm_skinny.cuh — "kernel B": skinny/medium-M GEMM (any M, processed in 16*MT-row chunks).
//
// Used wherever kernels A / C / D do not apply: 33..255 rows (short prefills), fine scale groups, narrow GEMMs below 768
// rows, the dense lm_head above 8 rows.  Roofline: HBM for M <= ~128; above that it is LDS-bandwidth bound (one x fragment
// read per MFMA), which is what kernel D (gemm_q4_big.cuh) removes for large M.
// Workgroup = 512 threads = 8 waves; wave w owns n-block (blockIdx.x*8 + w) = 16 output columns
// (DUAL: the same block of the gate AND the up tensor) and walks ALL k-tiles of its K slice, so
// each weight byte is read once per 16*MT rows.  The x chunk (16*MT rows x 256 k) is staged through
// LDS (double buffered, row-major with one octet of padding per row so both the 16 B stores and the MFMA
// B-fragment `ds_read_b128` are conflict-free) and shared by the 8 waves.  grid.y = M chunks, grid.z = K slices (split-K):
// slice partials go to fp32 slabs (write-through) and the last-dispatched slice of a tile reduces them in fixed
// slice order (deterministic) and applies the epilogue.
#pragma once
#include "wna16.cuh"
#include "gemv.cuh"  // GemvSeg

#define GB_THREADS 512
#define GB_WAVES 8
#define GB_KC 256  // k per staged chunk (2 k-tiles)

struct GemmBArgs {
  const void* w0;  // int4 tiled / dense [N,K]
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
  int group_size, is_awq, scales_layout, out_f32;
  int splitk;         // grid.z
  float* slabs;       // [splitk][tile][tensor][m-tile][thread][4] fp32 partials (splitk > 1)
  uint32_t* counters;  // arrival flags, 16 words apart, one per (tile, slice): zero on entry, zero on exit
  uint32_t* err;       // scratch error word: set when a slice wait timed out (vra_scratch_error)
  // more tensors with the same x in ONE launch (q/k/v): segment 0 is w0/sc0/qz0/bias0/out/N above, segments 1..nseg-1
  // follow; `blk_start` = first n-block of the segment in the flattened n-block space (not with DUAL / residual)
  int nseg;
  GemvSeg xseg[2];
};

// SPT = scale groups per 128-row k-tile held in registers (1: group_size >= 128 or -1; 4: 32/64).
template <class DT, bool INT4, bool DUAL, int MT, int SPT>
__global__ __launch_bounds__(GB_THREADS) void gemm_skinny_kernel(const GemmBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROWS = MT * 16;
  constexpr int NW = DUAL ? 2 : 1;
  // x buffer: row-major, one octet of padding per row — the staging stores of a wave are consecutive in LDS and the
  // MFMA fragment reads (16 rows x 4 octets) are conflict-free per quarter wave (the former [octet][row] XOR-swizzled
  // layout made every staging ds_write_b128 a multi-way bank conflict: measured on kernel C, -25 % kernel time)
  constexpr int RS = (GB_KC / 8 + 1) * 4;  // row stride, in u32
  constexpr int XS_U32 = ROWS * RS;        // one x buffer, in u32
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches, no exec masking
  const int nn = lane & 15, oct = lane >> 4;
  const int K = a.K, M = a.M;
  const int KT = K >> 7;
  const int g = a.group_size > 0 ? a.group_size : K;
  const bool grouped = a.group_size > 0 && a.group_size < K;
  // this wave's tensor (wave-uniform) and n-block within it
  const void* w0p = a.w0;
  const void* sc0p = a.sc0;
  const uint32_t* qz0p = a.qz0;
  const void* bias0p = a.bias0;
  void* outp = a.out;
  int N = a.N, out_ld = a.out_ld;
  int nb = blockIdx.x * GB_WAVES + wave;
  if (!DUAL && a.nseg > 1) {
    const int s = (a.nseg > 2 && nb >= a.xseg[1].blk_start) ? 1 : (nb >= a.xseg[0].blk_start ? 0 : -1);
    if (s >= 0) {
      const GemvSeg& sg = a.xseg[s];
      w0p = sg.w, sc0p = sg.scales, qz0p = sg.qzeros, bias0p = sg.bias, outp = sg.out, N = sg.n, out_ld = sg.out_ld;
      nb -= sg.blk_start;
    }
  }
  const bool nb_ok = nb * 16 < N;
  const int m0 = blockIdx.y * ROWS;
  // K slice of this workgroup, in chunks of GB_KC (slices are chunk aligned)
  const int nchunk_total = (K + GB_KC - 1) / GB_KC;
  const int cps = (nchunk_total + a.splitk - 1) / a.splitk;
  const int c_begin = blockIdx.z * cps, c_end = min(nchunk_total, c_begin + cps);

  constexpr int TPC = GB_KC / 128;   // tiles per chunk
  constexpr int FPC = TPC * SPT;     // zero-point fix-up steps per chunk
  constexpr int OPG = 16 / SPT;      // octets per fix-up step
  uint32_t* xs = reinterpret_cast<uint32_t*>(smem);                      // 2 buffers
  float* xsum = reinterpret_cast<float*>(smem + 2 * XS_U32 * 4);          // [2][ROWS][FPC]: Σx per row and fix-up step

  // ---- x chunk staging, split in two so that the global loads are issued BEFORE the weight loads of the same
  // iteration and consumed after the MFMAs: everything on the load path is straight-line code with clamped
  // addresses (a conditional load anywhere makes hipcc wait with vmcnt(0): see gemv_q4.cuh).
  constexpr int XPT = ROWS * (GB_KC / 8) / GB_THREADS;  // octets per thread per chunk (= MT)
  auto x_load = [&](int c, u32x4 (&xr)[XPT]) {
#pragma unroll
    for (int r = 0; r < XPT; r++) {
      const int i = tid + r * GB_THREADS;
      const int row = i >> 5, o = i & 31;
      const int m = min(m0 + row, M - 1), k = min(c * GB_KC + o * 8, K - 8);  // rows >= M alias row M-1 (never stored)
      xr[r] = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.x) + (size_t)m * a.x_ld + k);
    }
  };
  auto x_store = [&](int buf, const u32x4 (&xr)[XPT]) {
    uint32_t* dst = xs + buf * XS_U32;
#pragma unroll
    for (int r = 0; r < XPT; r++) {
      const int i = tid + r * GB_THREADS;
      const int row = i >> 5, o = i & 31;
      *reinterpret_cast<u32x4*>(dst + (size_t)row * RS + o * 4) = xr[r];
      if (INT4) {  // every lane takes part in the shuffles
        float s8 = octet_sum<DT>(xr[r]);
        s8 = OPG == 4 ? quad_sum(s8) : row16_sum(s8);
        if ((o % OPG) == 0) xsum[((size_t)buf * ROWS + row) * FPC + o / OPG] = s8;
      }
    }
  };

  f32x4 acc[NW][MT];
#pragma unroll
  for (int w = 0; w < NW; w++)
#pragma unroll
    for (int t = 0; t < MT; t++) acc[w][t] = vra_zero_acc();

  // weight stream of this wave: tiles kt = 2c, 2c+1 for c in [c_begin, c_end); a wave without an n-block streams
  // block 0 and contributes nothing (its scales / weights are zeroed at the point of use)
  const int nbc = nb_ok ? nb : 0;
  const u32x4* wp[NW];
  if (INT4) {
    wp[0] = reinterpret_cast<const u32x4*>(w0p) + ((size_t)nbc * KT) * 64 + lane;
    if (DUAL) wp[NW - 1] = reinterpret_cast<const u32x4*>(a.w1) + ((size_t)nbc * KT) * 64 + lane;
  } else {
    int n = min(nb * 16 + nn, N - 1);
    wp[0] = reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(w0p) + (size_t)n * K) + oct;
    if (DUAL) wp[NW - 1] = reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.w1) + (size_t)n * K) + oct;
  }
  // INT4: one u32x4 per tile; dense: four u32x4 per tile
  constexpr int LPT = INT4 ? 1 : 4;
  u32x4 cur[TPC][NW][LPT], nxt[TPC][NW][LPT];
  // scales / zero points ride along with the weight stream in the MFMA OUTPUT layout: each lane keeps
  // the 4 scales (one 8 B load, row-major [K/g, N]) and the AWQ zero word of its 4 output columns per group
  u32x2 csc[TPC][NW][SPT], nsc[TPC][NW][SPT];
  uint32_t czp[TPC][NW][SPT], nzp[TPC][NW][SPT];
  const int n4 = min(nbc * 16 + oct * 4, N - 4);
  // k / g as a shift for power-of-two groups (every real checkpoint): the division sat in the main loop, once per tile,
  // tensor and scale group (~25 VALU instructions each, in a kernel that is VALU / LDS bound)
  const int gsh = (g & (g - 1)) == 0 ? 31 - __builtin_clz((unsigned)g) : -1;
  const int G = grouped ? (gsh >= 0 ? K >> gsh : K / g) : 1;
  const bool awq = a.is_awq != 0 && qz0p != nullptr;
  auto load_chunk = [&](int c, u32x4 (&dst)[TPC][NW][LPT], u32x2 (&dsc)[TPC][NW][SPT], uint32_t (&dzp)[TPC][NW][SPT]) {
#pragma unroll
    for (int t = 0; t < TPC; t++) {
      const int kt = min(c * TPC + t, KT - 1);
#pragma unroll
      for (int w = 0; w < NW; w++) {
#pragma unroll
        for (int l = 0; l < LPT; l++) {
          if (INT4) dst[t][w][l] = __builtin_nontemporal_load(wp[w] + (size_t)kt * 64);
          else dst[t][w][l] = __builtin_nontemporal_load(wp[w] + kt * 16 + l * 4);
        }
        if (INT4) {
#pragma unroll
          for (int q = 0; q < SPT; q++) {
            const int k0 = kt * 128 + q * (128 / SPT);
            const int grp = min(grouped ? (gsh >= 0 ? k0 >> gsh : k0 / g) : 0, G - 1);
            const uint16_t* sp = static_cast<const uint16_t*>(w ? a.sc1 : sc0p);
            dsc[t][w][q] = *reinterpret_cast<const u32x2*>(sp + (size_t)grp * N + n4);
            if (SPT > 0 && awq) dzp[t][w][q] = (w ? a.qz1 : qz0p)[(size_t)grp * (N >> 3) + (n4 >> 3)];
            else dzp[t][w][q] = 0x88888888u;
          }
        }
      }
    }
  };

  if (c_begin < c_end) {
    u32x4 xr[XPT];
    x_load(c_begin, xr);
    load_chunk(c_begin, cur, csc, czp);
    x_store(0, xr);
  }
  __syncthreads();
  for (int c = c_begin; c < c_end; c++) {
    const int buf = (c - c_begin) & 1;
    const int cn = min(c + 1, c_end - 1);  // the last iteration re-loads its own chunk (never consumed)
    u32x4 xr[XPT];
    x_load(cn, xr);
    load_chunk(cn, nxt, nsc, nzp);
    const uint32_t* xb = xs + buf * XS_U32;
    const float* sxb = xsum + (size_t)buf * ROWS * FPC;
#pragma unroll
    for (int t = 0; t < TPC; t++) {
      const bool valid = nb_ok && (c * TPC + t) < KT;  // wave-uniform
      f32x4 ag[NW][MT];
#pragma unroll
      for (int w = 0; w < NW; w++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) ag[w][mt] = vra_zero_acc();
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int o = t * 16 + j * 4 + oct;  // octet within chunk
        s16x8 afrag[NW];
#pragma unroll
        for (int w = 0; w < NW; w++) {
          if (INT4) {
            afrag[w] = magic_word<DT>(cur[t][w][0][j]);
          } else {
            u32x4 wv = cur[t][w][INT4 ? 0 : j];
            if (!valid) wv = u32x4{0u, 0u, 0u, 0u};
            afrag[w] = __builtin_bit_cast(s16x8, wv);
          }
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
          const u32x4 xv = *reinterpret_cast<const u32x4*>(xb + (size_t)(mt * 16 + nn) * RS + o * 4);
          const s16x8 bfrag = __builtin_bit_cast(s16x8, xv);
#pragma unroll
          for (int w = 0; w < NW; w++) {
            if (INT4) DT::mfma(ag[w][mt], afrag[w], bfrag);
            else DT::mfma(acc[w][mt], afrag[w], bfrag);
          }
        }
        if (INT4 && (SPT == 4 || j == 3)) {  // end of a scale group: acc += s * (acc_g - (C+z) * Σx_g)
          VRA_MFMA_DRAIN();
          const int q = SPT == 4 ? j : 0;
#pragma unroll
          for (int w = 0; w < NW; w++) {
            float sc4[4], zc4[4];
            unpack_scale4<DT>(csc[t][w][q], czp[t][w][q], n4, sc4, zc4);
#pragma unroll
            for (int r = 0; r < 4; r++) sc4[r] = valid ? sc4[r] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
              const float sx = sxb[(size_t)(mt * 16 + nn) * FPC + t * SPT + q];
#pragma unroll
              for (int r = 0; r < 4; r++) {
                acc[w][mt][r] = fmaf(sc4[r], fmaf(-zc4[r], sx, ag[w][mt][r]), acc[w][mt][r]);
              }
              if (SPT == 4 && j < 3) ag[w][mt] = vra_zero_acc();
            }
          }
        }
      }
    }
    x_store(buf ^ 1, xr);  // the other buffer was last read before the previous barrier
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPC; t++)
#pragma unroll
      for (int w = 0; w < NW; w++) {
#pragma unroll
        for (int l = 0; l < LPT; l++) cur[t][w][l] = nxt[t][w][l];
#pragma unroll
        for (int q = 0; q < SPT; q++) {
          csc[t][w][q] = nsc[t][w][q];
          czp[t][w][q] = nzp[t][w][q];
        }
      }
  }

  if (!INT4) VRA_MFMA_DRAIN();  // the dense path accumulates straight into acc
  // ---- split-K: K slices meet through memory, as in kernel C (gemm_q4.cuh): partial tiles go out as agent-scope
  // (write-through, sc1) 16-byte stores — one contiguous KiB per wave —, every slice raises its own flag (one 64-byte
  // line each) once they are acknowledged, and the LAST slice (z = splitk-1, the "owner") polls the flags, sums the slabs in
  // fixed slice order (its own partial last: deterministic) and resets the flags.  No arrival counter (agent-scope
  // atomics on one address serialise, ~1.3 us each), no L2 write-back fences.  Progress: only owners wait and the
  // launcher keeps owners (tiles) below the number of resident workgroup slots, so some non-owner always runs.
  if (a.splitk > 1) {
    const int SK = a.splitk, zi = (int)blockIdx.z;
    const int tile = (int)(blockIdx.y * gridDim.x + blockIdx.x), ntiles = (int)(gridDim.x * gridDim.y);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, 0x7FFFFFF0, 0x00020000);
    auto slab_off = [&](int z, int w, int mt) {  // bytes: [slice][tile][tensor][m-tile][thread] x 16 B
      return (uint32_t)((((z * ntiles + tile) * NW + w) * MT + mt) * GB_THREADS + tid) * 16u;
    };
    uint32_t* fl = a.counters + (size_t)tile * SK * 16;
    if (zi != SK - 1) {
#pragma unroll
      for (int w = 0; w < NW; w++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[w][mt]), srs, slab_off(zi, w, mt), 0, 16);
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
    f32x4 sum[NW][MT];
#pragma unroll
    for (int w = 0; w < NW; w++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) sum[w][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int ZLD = SPT == 4 ? 4 : 16;  // loads of 16 B in flight (the fine-group kernels sit at the VGPR limit)
    constexpr int ZB = (ZLD / (NW * MT)) < 1 ? 1 : ZLD / (NW * MT);  // slices per round trip
    for (int z0 = 0; z0 < SK - 1; z0 += ZB) {
      u32x4 p[ZB][NW][MT];
#pragma unroll
      for (int j = 0; j < ZB; j++)
#pragma unroll
        for (int w = 0; w < NW; w++)
#pragma unroll
          for (int mt = 0; mt < MT; mt++) p[j][w][mt] = __builtin_amdgcn_raw_buffer_load_b128(srs, slab_off(min(z0 + j, SK - 2), w, mt), 0, 16);
#pragma unroll
      for (int j = 0; j < ZB; j++)
        if (z0 + j < SK - 1) {
#pragma unroll
          for (int w = 0; w < NW; w++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) sum[w][mt] += __builtin_bit_cast(f32x4, p[j][w][mt]);
        }
    }
#pragma unroll
    for (int w = 0; w < NW; w++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) acc[w][mt] = sum[w][mt] + acc[w][mt];
  }

  // ---- epilogue: D[row = 16-col index (lane>>4)*4 + r][col = m = lane&15]
  if (!nb_ok) return;
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const int m = m0 + mt * 16 + nn;
    if (m >= M) continue;
    const int n = nb * 16 + oct * 4;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float t = rnd_dt<DT>(acc[0][mt][r]);
      if (bias0p) t = rnd_dt<DT>(t + DT::to_f32(static_cast<const uint16_t*>(bias0p)[n + r]));
      if (DUAL) {
        float u = rnd_dt<DT>(acc[NW - 1][mt][r]);
        if (a.bias1) u = rnd_dt<DT>(u + DT::to_f32(static_cast<const uint16_t*>(a.bias1)[n + r]));
        float sl = rnd_dt<DT>(t / (1.0f + expf(-t)));
        t = sl * u;
      }
      if (a.residual) t = rnd_dt<DT>(t) + DT::to_f32(static_cast<const uint16_t*>(a.residual)[(size_t)m * a.res_ld + n + r]);
      v[r] = t;
    }
    if (a.out_f32) {
      f32x4 o = {rnd_dt<DT>(v[0]), rnd_dt<DT>(v[1]), rnd_dt<DT>(v[2]), rnd_dt<DT>(v[3])};
      *reinterpret_cast<f32x4*>(static_cast<float*>(outp) + (size_t)m * out_ld + n) = o;
    } else {
      u32x2 o = {DT::pack2(v[0], v[1]), DT::pack2(v[2], v[3])};
      *reinterpret_cast<u32x2*>(static_cast<uint16_t*>(outp) + (size_t)m * out_ld + n) = o;
    }
  }
}

static inline size_t gemm_skinny_lds_bytes(int mt, int spt) {
  return (size_t)2 * (GB_KC / 8 + 1) * mt * 16 * 16 + (size_t)2 * mt * 16 * (GB_KC / 128) * spt * 4 + 16;
}
