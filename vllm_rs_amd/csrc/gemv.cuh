// gemv.cuh — "kernel A": skinny GEMM (M <= 8 rows) that streams every weight byte exactly once.
//
// Roofline: HBM.  Algorithmic bytes per call: K*N/2 (packed int4) + (K/g)*N*2 (scales)
// [+ (K/g)*N/2 zeros] + M*K*2 + M*N*2   (dense variant: N*K*2 + ...).
//
// Shape of the launch:
//   PERSISTENT workgroups (grid = resident capacity), 512 threads = 8 waves.  A work item is one
//   16-column n-block (NBW=2: the same block of the gate AND the up tensor) over the FULL K range; the 8
//   waves of a workgroup split its k-tiles (wave w takes tiles w, w+8, ...: 1 KiB coalesced per tile,
//   the whole workgroup walks one contiguous K/128 KiB run) and meet in LDS — no inter-workgroup
//   traffic, no atomics, no second launch.  Work items are streamed: the loads of sub-step s+1 (4 tiles
//   per wave, possibly of the NEXT n-block) are issued before sub-step s is consumed, and the very first
//   loads go out BEFORE the prologue, so HBM latency hides under the x staging / RMSNorm.
//   x (optionally RMS-normalised on the fly: the reference's separate NormX launch, others.rs:11-29)
//   is staged in LDS ONCE per workgroup together with its per-group sums Σx (needed by the zero-point
//   fix-up, wna16.cuh), and reused for all work items.
//   The multiply is `v_mfma_f32_16x16x32` with A = (C + q) "magic" 16-bit floats (16 columns x 32 k,
//   one VALU op per two weights), B = xT (32 k x 16 rows; rows >= M read an all-zero LDS slot).  Per
//   scale group the partial result is folded as acc += s·(acc_g − (C+z)·Σx): 2 VALU ops per output.
#pragma once
#include "wna16.cuh"

#define GEMV_THREADS 512
#define GEMV_WAVES 8
#define GEMV_MAX_SEG 3
#ifndef GEMV_DENSE_NBUF
#define GEMV_DENSE_NBUF 3  // ring depth (sub-steps of 4 KiB per wave) of the dense single-tensor stream (the lm_head)
#endif
#define GEMV_AM_MAX_GRID 2048
#define GEMV_AM_COUNTER (8 * GEMV_AM_MAX_GRID)
// (value, index) -> one u64 whose unsigned order is "larger value first, then smaller index"; NaN orders below everything
__device__ __forceinline__ unsigned long long gemv_am_key(float v, uint32_t idx) {
  v += 0.0f;  // -0 -> +0: equal values must produce equal keys
  const uint32_t b = __float_as_uint(v);
  const uint32_t o = (v != v) ? 0u : ((b & 0x80000000u) ? ~b : (b | 0x80000000u));
  return ((unsigned long long)o << 32) | (unsigned long long)(0xffffffffu - idx);
}
__device__ __forceinline__ unsigned long long gemv_am_max(unsigned long long x, unsigned long long y) { return x > y ? x : y; }
__device__ __forceinline__ unsigned long long gemv_am_shfl_xor(unsigned long long x, int d) {
  const uint32_t lo = __shfl_xor((uint32_t)x, d, 64), hi = __shfl_xor((uint32_t)(x >> 32), d, 64);
  return ((unsigned long long)hi << 32) | lo;
}

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
  int n_items;    // work items (n-blocks, or gate/up pairs)
  int m_groups;   // int4 kernel: row-group families (1: all M rows in every workgroup)
  int rows_per_group;
  int single_red;  // int4 kernel: single-buffered cross-wave reduction (LDS-tight shapes)
  // int4 kernel: wave-uniform quotients precomputed by the launcher (an integer division is ~25 VALU instructions on the
  // way to the first load of every launch): steps per item, items per slot (quotient, remainder), x chunks per thread
  // (m_groups == 1), log2 of the octets per row (or -1)
  int steps_per_item, items_q, items_r, x_chunks, octs_shift;
  int dbg;        // experiment switches (VRA_EXP): 1 = prologue only, 2 = skip the x staging
  // dense single-tensor launches with f32 output (the lm_head of 1..8-row steps): the greedy token of every row comes out of
  // the same launch.  am_ws = [M][grid] candidate keys + the arrival counter at am_ws[GEMV_AM_COUNTER]; the last workgroup to
  // arrive reduces the candidates and writes am_out[m] (first maximal index, as vra_argmax_f32) and re-arms the counter.
  uint32_t* am_out;
  unsigned long long* am_ws;
  unsigned long long* ts;  // VRA_GEMV_TS builds: [grid][32] wall-clock stamps of wave 0
  // dense launches: seg[].w is the TILE-MAJOR copy vra_dense_tile_weights makes (u32x4 word ((nb*KT + kt)*4 + l)*64 + lane holds row
  // nb*16 + nn, columns kt*128 + l*32 + oct*8 ..: the MFMA operand order, one contiguous KiB per wave load) instead of row-major
  // [n, K].  A row-major wave load touches 16 rows x 64 bytes — half a 128-byte line of 16 different lines, the other halves by the
  // next load: the 1 GB lm_head streamed at 5.5 TB/s in that form whatever the ring depth (a plain contiguous read: 6.75)
  int dense_tiled;
};
#ifdef VRA_GEMV_TS
// (round 6) stamps are parked in LDS and leave in ONE store per thread at the end of the kernel (GEMV_STAMP_FLUSH).  Rounds 2-5 stored
// every stamp to global memory on the spot: on gfx9 stores count in vmcnt like the loads, hipcc then waits vmcnt(0) where it would
// have counted, and the stamped wave (wave 0) ran behind its peers — the "wave 0 ends 1.2 us late" of profiles/r05_timeline_kernel_e.txt
// was the instrument, not the kernel.
__device__ __forceinline__ unsigned long long* vra_ts_lds() {
  __shared__ unsigned long long t[32];
  return t;
}
#define GEMV_STAMP(i)                                                     \
  do {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                    \
    if ((i) == 0 && a.ts && tid < 32) vra_ts_lds()[tid] = 0ull; /* (stamps a launch never reaches read 0) */ \
    if (a.ts && tid == 0) vra_ts_lds()[(i)] = wall_clock64();             \
    __builtin_amdgcn_sched_barrier(0);                                    \
  } while (0)
#define GEMV_STAMP_FLUSH()                                                                        \
  do {                                                                                            \
    if (a.ts && tid < 32) a.ts[(size_t)blockIdx.x * 32 + tid] = vra_ts_lds()[tid]; /* (same wave as the writer: in order) */ \
  } while (0)
#elif defined(VRA_GEMV_SB)
#define GEMV_STAMP(i) __builtin_amdgcn_sched_barrier(0)
#define GEMV_STAMP_FLUSH() do {} while (0)
#else
#define GEMV_STAMP(i) do {} while (0)
#define GEMV_STAMP_FLUSH() do {} while (0)
#endif

// Stage x[M,K] into LDS (image xs[(oct*M + m)*4 .. +4] = x[m][oct*8 .. +8]), optionally RMS-normalised,
// and the sums of the staged (rounded) values over every run of OPG consecutive octets: xsum[m*NF + f].
template <class DT>
__device__ __forceinline__ void gemv_stage_x(const GemvArgs& a, uint32_t* xs, float* xsum, int opg, float* red8) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches, no exec masking
  const int octs = a.K >> 3;
  const int nf = octs / opg;
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
    for (int o0 = 0; o0 < octs; o0 += GEMV_THREADS) {  // uniform trip count: the shuffles below need all lanes
      const int o = o0 + tid;
      float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (o < octs) {
        u32x4 v = xr[o];
        unpack8<DT>(v, f);
        if (a.norm_w) {
          float g[8];
          unpack8<DT>(nw[o], g);
#pragma unroll
          for (int i = 0; i < 8; i++) f[i] = f[i] * rstd * g[i];
          v = pack8<DT>(f);
          unpack8<DT>(v, f);  // sums are taken over the ROUNDED values the MFMA will see
        }
        *reinterpret_cast<u32x4*>(xs + ((size_t)o * a.M + m) * 4) = v;
      }
      if (xsum) {
        float s8 = ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
        for (int d = 1; d < opg; d <<= 1) s8 += __shfl_xor(s8, d, 64);
        if (o < octs && (o % opg) == 0) xsum[(size_t)m * nf + o / opg] = s8;
      }
    }
  }
}

// INT4 = true: tiled int4 weights; false: dense row-major 16-bit weights.
// NBW = tensors per work item (2 = gate/up pair).  SPT = scale groups per k-tile (1: g >= 128, 4: 32/64).
// AWQ = per-group zero points (otherwise the zero point is the constant 8 and no zero words are loaded).
template <class DT, bool INT4, int NBW, int SPT, bool AWQ>
__global__ __launch_bounds__(GEMV_THREADS, 4) void gemv_kernel(const GemvArgs a) {  // 4 waves/SIMD = 2 workgroups per CU (<= 128 VGPRs)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int LPT = INT4 ? 1 : 4;  // 16 B loads per lane and tile
  constexpr int UK = INT4 ? 4 / NBW : 1;  // k-tiles per wave per sub-step (two sub-steps = 8 x 16 B loads in flight)
  constexpr int NSC = INT4 ? SPT : 1;
  constexpr int NZP = AWQ ? NSC : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches, no exec masking
  const int nn = lane & 15, oct = lane >> 4;
  const int K = a.K, M = a.M, KT = K >> 7;
  const int g = a.group_size > 0 ? a.group_size : K;
  const bool grouped = a.group_size > 0 && a.group_size < K;
  const int NF = KT * SPT;  // fix-up steps along K
  GEMV_STAMP(0);

  // ---- LDS carve-up
  uint32_t* xs = reinterpret_cast<uint32_t*>(smem);  // M*K*2 bytes + zero slot (16 B of zeros, also the Σx of missing rows)
  const uint32_t zero_slot = (uint32_t)((((size_t)M * K * 2 + 15) & ~(size_t)15) >> 2);
  size_t off = ((size_t)zero_slot << 2) + 16;
  float* xsum = reinterpret_cast<float*>(smem + off);  // [M][NF]
  off += INT4 ? (((size_t)M * NF * 4 + 15) & ~(size_t)15) : 0;
  f32x4* red = reinterpret_cast<f32x4*>(smem + off);  // [2][8][NBW][64] f32x4
  off += (size_t)2 * GEMV_WAVES * NBW * 64 * sizeof(f32x4);
  float* red8 = reinterpret_cast<float*>(smem + off);  // 8 floats

  // ---- this workgroup's stream of sub-steps
  const int nsub = ((KT + GEMV_WAVES - 1) / GEMV_WAVES + UK - 1) / UK;  // sub-steps per work item
  const int my_items = (a.n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int total_sub = (a.dbg & 1) ? 0 : my_items * nsub;

  // register ring of NBUF sub-steps: a buffer is refilled (for the sub-step NBUF ahead) right after it
  // has been consumed, so NBUF*UK*NBW KiB per wave are in flight all the time and nothing is copied
  constexpr int NBUF = (!INT4 && NBW == 2) ? 2 : (!INT4 ? GEMV_DENSE_NBUF : 3);  // (the dense gate/up pair holds 8 x 16 B per lane and sub-step: two sub-steps fit the 128 VGPRs)
  u32x4 wbuf[NBUF][UK][NBW][LPT];
  u32x2 sbuf[NBUF][UK][NBW][NSC];
  uint32_t zbuf[NBUF][UK][NBW][NZP];

  auto resolve = [&](int item, int b, int& segi, int& nb) {
    const int fb = (int)blockIdx.x + item * (int)gridDim.x;
    if (a.silu_dual) {
      segi = b;
      nb = fb;
    } else {
      int s = 0;
      if (a.nseg > 1 && fb >= a.seg[1].blk_start) s = 1;
      if (a.nseg > 2 && fb >= a.seg[2].blk_start) s = 2;
      segi = s;
      nb = fb - a.seg[s].blk_start;
    }
  };
  auto issue = [&](int item, int sub, u32x4 (&w)[UK][NBW][LPT], u32x2 (&sc)[UK][NBW][NSC], uint32_t (&zp)[UK][NBW][NZP]) {
#pragma unroll
    for (int b = 0; b < NBW; b++) {
      int segi, nb;
      resolve(item, b, segi, nb);
      const GemvSeg& sg = a.seg[segi];
#pragma unroll
      for (int u = 0; u < UK; u++) {
        const int kt = wave + GEMV_WAVES * (sub * UK + u);
        if (kt < KT) {
          if (INT4) {
            w[u][b][0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sg.w) + ((size_t)nb * KT + kt) * 64 + lane);
            const int n4 = min(nb * 16 + oct * 4, sg.n - 4);
#pragma unroll
            for (int q = 0; q < NSC; q++) {
              const int grp = min((kt * 128 + q * (128 / NSC)) / g, K / g - 1);
              uint32_t zdummy;
              load_scale4_raw<DT>(sg.scales, sg.qzeros, grp, n4, sg.n, a.scales_layout, grouped, AWQ, sc[u][b][q], AWQ ? zp[u][b][AWQ ? q : 0] : zdummy);
            }
          } else {
            const int n = min(nb * 16 + nn, sg.n - 1);
            // (one pointer, scalar-selected stride: the row-major and the tile-major form differ in base and step only)
            const u32x4* wp = a.dense_tiled ? reinterpret_cast<const u32x4*>(sg.w) + ((size_t)nb * KT + kt) * 256 + lane
                                            : reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(sg.w) + (size_t)n * K) + oct + kt * 16;
            const int lstep = a.dense_tiled ? 64 : 4;
#pragma unroll
            for (int l = 0; l < 4; l++) w[u][b][INT4 ? 0 : l] = __builtin_nontemporal_load(wp + l * lstep);
          }
        } else {
          // no such k-tile for this wave: zero weights / zero scales contribute exactly 0 and keep the
          // consume loop below free of branches (see common.cuh on MFMA register hazards)
#pragma unroll
          for (int l = 0; l < LPT; l++) w[u][b][l] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
          for (int q = 0; q < NSC; q++) sc[u][b][q] = u32x2{0u, 0u};
          if (AWQ) {
#pragma unroll
            for (int q = 0; q < NZP; q++) zp[u][b][q] = 0u;
          }
        }
      }
    }
  };

  // the first NBUF sub-steps go out BEFORE the prologue: HBM latency hides under the x staging
  int iitem = 0, isub = 0, issued = 0;  // issue cursor
  auto advance = [&](int& it, int& sb) {
    if (++sb == nsub) {
      sb = 0;
      ++it;
    }
  };
#pragma unroll
  for (int r = 0; r < NBUF; r++) {
    if (issued < total_sub) {
      issue(iitem, isub, wbuf[r], sbuf[r], zbuf[r]);
      advance(iitem, isub);
      ++issued;
    }
  }

  GEMV_STAMP(1);
  if (tid < 4) xs[zero_slot + tid] = 0u;
  if (!(a.dbg & 2)) gemv_stage_x<DT>(a, xs, INT4 ? xsum : nullptr, 16 / SPT, red8);
  __syncthreads();
  GEMV_STAMP(2);

  // B fragment addressing: lanes whose batch row does not exist read the zero slot (stride 0)
  const bool row_ok = nn < M;
  const uint32_t xbase = row_ok ? (uint32_t)(oct * M + nn) * 4u : zero_slot;
  const uint32_t xstride = row_ok ? (uint32_t)M * 4u : 0u;  // u32 per octet step
  const float* sxp = row_ok ? xsum + (size_t)nn * NF : reinterpret_cast<const float*>(xs + zero_slot);
  const int sxstride = row_ok ? 1 : 0;

  f32x4 acc[NBW];
#pragma unroll
  for (int b = 0; b < NBW; b++) acc[b] = vra_zero_acc();
  unsigned long long am_best = 0ull;  // this thread's (row, column-in-block) over its work items, ascending columns
  int item = 0, sub = 0, parity = 0;
  for (int s0 = 0; s0 < total_sub; s0 += NBUF) {
#pragma unroll
   for (int rb = 0; rb < NBUF; rb++) {
    if (s0 + rb >= total_sub) break;
    u32x4 (&cur)[UK][NBW][LPT] = wbuf[rb];
    u32x2 (&csc)[UK][NBW][NSC] = sbuf[rb];
    uint32_t (&czp)[UK][NBW][NZP] = zbuf[rb];
    // ---- consume sub-step (item, sub)
    int n4b[NBW];
#pragma unroll
    for (int b = 0; b < NBW; b++) {
      int segi, nb;
      resolve(item, b, segi, nb);
      n4b[b] = min(nb * 16 + oct * 4, a.seg[segi].n - 4);
    }
    // all MFMAs of a 32-row step (every tile u, every tensor b) are issued back to back into group
    // accumulators; ONE drain per scale group, then the fix-ups (VALU) of all of them
    // (tiles past the end of K carry zero weights and zero scales, so this region has no branches)
    f32x4 ag[UK][NBW];
#pragma unroll
    for (int u = 0; u < UK; u++)
#pragma unroll
      for (int b = 0; b < NBW; b++) ag[u][b] = vra_zero_acc();
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int u = 0; u < UK; u++) {
        const int kt = min(wave + GEMV_WAVES * (sub * UK + u), KT - 1);
        const u32x4 xb = *reinterpret_cast<const u32x4*>(xs + xbase + (uint32_t)(kt * 16 + j * 4) * xstride);
        const s16x8 bfrag = __builtin_bit_cast(s16x8, xb);
#pragma unroll
        for (int b = 0; b < NBW; b++) {
          if (INT4) DT::mfma(ag[u][b], magic_word<DT>(cur[u][b][0][j]), bfrag);
          else DT::mfma(acc[b], __builtin_bit_cast(s16x8, cur[u][b][INT4 ? 0 : j]), bfrag);
        }
      }
      if (INT4 && (SPT == 4 || j == 3)) {  // end of a scale group: acc += s * (acc_g - (C+z) * Σx_g)
        VRA_MFMA_DRAIN();
        const int q = SPT == 4 ? j : 0;
#pragma unroll
        for (int u = 0; u < UK; u++) {
          const int kt = min(wave + GEMV_WAVES * (sub * UK + u), KT - 1);
          const float sx = sxp[(kt * SPT + q) * sxstride];
#pragma unroll
          for (int b = 0; b < NBW; b++) {
            float sc4[4], zc4[4];
            unpack_scale4<DT>(csc[u][b][q], AWQ ? czp[u][b][AWQ ? q : 0] : 0x88888888u, n4b[b], sc4, zc4);
#pragma unroll
            for (int r = 0; r < 4; r++) acc[b][r] = fmaf(sc4[r], fmaf(-zc4[r], sx, ag[u][b][r]), acc[b][r]);
            if (SPT == 4 && j < 3) ag[u][b] = vra_zero_acc();
          }
        }
      }
    }
    GEMV_STAMP(3 + 2 * (s0 + rb < 5 ? s0 + rb : 5));
    // ---- end of a work item: cross-wave reduction in LDS, then the fused epilogue
    if (sub == nsub - 1) {
      f32x4* rbuf = red + (size_t)parity * GEMV_WAVES * NBW * 64;
      if (!INT4) VRA_MFMA_DRAIN();  // dense path accumulates straight into acc
#pragma unroll
      for (int b = 0; b < NBW; b++) {
        rbuf[(wave * NBW + b) * 64 + lane] = acc[b];
        acc[b] = vra_zero_acc();
      }
      __syncthreads();
      const int nout = a.silu_dual ? 16 * M : NBW * 16 * M;
      for (int idx = tid; idx < nout; idx += GEMV_THREADS) {
        // (b: the tensor of a multi-block item.  This kernel is launched with NBW = 1, or NBW = 2 as the gate/up PAIR whose epilogue
        // covers 16*M outputs: b is 0 either way — and must be a constant here: under a per-lane b hipcc selected the ADDRESS of
        // a.seg[b] inside the kernel-argument block and fetched its fields with vector loads + vmcnt(0), a memory round trip per
        // work item, 31 per workgroup of the lm_head launch; tools/check_mfma_overlap.py now fails the build on that pattern)
        const int nl = idx & 15, m = (idx >> 4) % M, b = NBW <= 2 ? 0 : idx / (16 * M);
        const int rl = (nl >> 2) * 16 + m, rr = nl & 3;  // D layout: row = (lane>>4)*4 + reg, col = lane&15
        float v = 0.f, v2 = 0.f;
#pragma unroll
        for (int w = 0; w < GEMV_WAVES; w++) {
          v += rbuf[(w * NBW + b) * 64 + rl][rr];
          if (NBW > 1) v2 += rbuf[(w * NBW + (NBW - 1)) * 64 + rl][rr];
        }
        int segi, nb;
        resolve(item, b, segi, nb);
        const GemvSeg& sg = a.seg[segi];
        const int n = nb * 16 + nl;
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
        if (!INT4 && NBW == 1 && a.am_out) am_best = gemv_am_max(am_best, gemv_am_key(rnd_dt<DT>(v), (uint32_t)n));
        if (a.out_f32) static_cast<float*>(sg.out)[(size_t)m * sg.out_ld + n] = rnd_dt<DT>(v);
        else static_cast<uint16_t*>(sg.out)[(size_t)m * sg.out_ld + n] = DT::from_f32(v);
      }
      parity ^= 1;  // the other buffer is only rewritten after the next barrier: no second barrier needed
    }
    GEMV_STAMP(4 + 2 * (s0 + rb < 5 ? s0 + rb : 5));
    advance(item, sub);
    // ---- refill this ring slot with the sub-step NBUF ahead
    if (issued < total_sub) {
      issue(iitem, isub, wbuf[rb], sbuf[rb], zbuf[rb]);
      advance(iitem, isub);
      ++issued;
    }
   }
  }
  if (!INT4 && NBW == 1) {
    if (a.am_out) {  // uniform.  nout = 16 * M <= 128 here (launcher), so thread t always held row t >> 4, column t & 15
      uint32_t* flag = reinterpret_cast<uint32_t*>(red8);
#pragma unroll
      for (int d = 8; d > 0; d >>= 1) am_best = gemv_am_max(am_best, gemv_am_shfl_xor(am_best, d));
      if (tid < 16 * M && (tid & 15) == 0)
        __hip_atomic_store(a.am_ws + (size_t)(tid >> 4) * gridDim.x + blockIdx.x, am_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // No C++-level release / acquire pair, on purpose: an agent-scope release is `buffer_wbl2 sc1` — it walks the whole
      // L2 once per workgroup: measured +150 us per launch.  The ordering rests on the ISA instead, and only on these four facts:
      //   (1) the candidate store, the counter RMW and the candidate loads are ALL agent-scope atomics (sc1): they are performed
      //       at the memory side of the XCD L2s, never served from or parked in a non-coherent line (one 8-byte word each: no tearing);
      //   (2) `s_waitcnt vmcnt(0)` returns once the write-through store has been ACKNOWLEDGED by that coherence point (gfx9 counts
      //       stores in vmcnt; the asm statement is opaque to hipcc, so it can neither be dropped nor moved — the guide's known
      //       failure is the compiler-generated wait behind a fence, which this is not);
      //   (3) the barrier below orders every thread's acknowledged store before thread 0's RMW; the RMW's returned value orders the
      //       last workgroup's loads behind every other workgroup's RMW (a returning atomic is complete when its value is back);
      //   (4) the workspace belongs to ONE Model and one stream: launches that share it are stream-ordered (the counter is re-armed
      //       by the last workgroup before the kernel ends).
      // tests/test_gpu_kernels.py::test_dense_gemm_argmax_hand_off_stress_over_all_xcds: 200 launches over all CUs, fresh x each.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const uint32_t old = __hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(a.am_ws + GEMV_AM_COUNTER), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = old == gridDim.x - 1 ? 1u : 0u;
      }
      __syncthreads();
      if (*flag) {  // the last workgroup to arrive: wave w reduces row w
        if (wave < M) {
          unsigned long long b = 0ull;
          for (int c = lane; c < (int)gridDim.x; c += 64)
            b = gemv_am_max(b, __hip_atomic_load(a.am_ws + (size_t)wave * gridDim.x + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
          for (int d = 32; d > 0; d >>= 1) b = gemv_am_max(b, gemv_am_shfl_xor(b, d));
          if (lane == 0) a.am_out[wave] = 0xffffffffu - (uint32_t)b;
        }
        if (tid == 0) __hip_atomic_store(reinterpret_cast<uint32_t*>(a.am_ws + GEMV_AM_COUNTER), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  GEMV_STAMP(15);
  GEMV_STAMP_FLUSH();
}

static inline size_t gemv_lds_bytes(bool int4, int nbw, int M, int K, int group_size) {
  const int spt = (int4 && group_size > 0 && group_size < 128) ? 4 : 1;
  size_t b = (((size_t)M * K * 2 + 15) & ~(size_t)15) + 16;
  if (int4) b += (((size_t)M * (K / 128) * spt * 4) + 15) & ~(size_t)15;
  b += (size_t)2 * GEMV_WAVES * nbw * 64 * 16 + 64;
  return b;
}
