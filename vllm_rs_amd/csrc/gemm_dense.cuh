// gemm_dense.cuh — "kernel X": the prefill GEMM on DEQUANTISED 16-bit weights, plus the pass that makes them.
//
// Why (round 6): kernel D (gemm_q4_big.cuh) converts int4 -> bf16 inside every 64-row workgroup and pays the exact per-group scale
// fix-up per accumulator: 3 VALU instructions per MFMA where one is free — 0.30-0.32 of the MFMA peak, and the same skeleton with
// the conversion removed runs 1.08 PFLOP/s (profiles/r06_kernel_d_probes.txt).  A prompt of M rows repeats that conversion M/64
// times per weight.  From ~700 rows on it is cheaper to dequantise a GEMM's weights ONCE into a scratch tensor (a streaming pass:
// 0.5 B read + 2 B written per weight) and run a plain 16-bit GEMM whose inner loop is nothing but LDS reads and MFMAs.  The
// dequantised weight is  w = round_dt((q - z) * s)  — what the reference's Marlin kernels feed their MMAs (src/utils/gptq.rs:116-178;
// oracle/vra_oracle.c orc_dequant + orc_gemm_wdense, orc_wna16_gemm_marlin) — so this path is the reference's arithmetic, f32
// accumulation, one rounding at the output and the epilogue roundings of the other kernels (bias, SiLU*mul, residual).
//
// Roofline: MFMA (2.5 PFLOP/s dense bf16).  Algorithmic FLOPs 2*M*K*N; bytes K*N*(0.5 + 2 + 2) + M*K*2 + M*N*2.
//
// Shape of the GEMM launch (256 x BN output tile, 8 waves = 2 (m) x 4 (n), one workgroup per CU, K-step 64):
//   * both operands travel global -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write pass) as 1 KiB
//     MFMA FRAGMENTS: a 16-row x 32-k block of x, or 16 columns x 32 k of w.  The dequantised weights are WRITTEN in fragment order
//     by the dequant pass (one wave load = one contiguous KiB in lane order: the LDS image is lane-linear and `ds_read_b128` at
//     lane*16 is conflict-free); x is row-major in memory, so a fragment's 64 chunks of 16 B are fetched row by row (4 lanes = 64
//     contiguous bytes of a row) with the chunk slot XOR-ed by (row >> 2) & 3 — the 16 lanes of a `ds_read_b128` group then cover
//     all 64 banks (rows r: slot (r & 3) * 64 + (q ^ (r >> 2)) * 16 bytes);
//   * a wave owns 128 rows x BN/4 columns = 8 x (BN/64) accumulator tiles; a K-step is FOUR phases (quadrants of the wave's tile:
//     4 m-frags x BN/128 n-frags x 2 k-halves = 16 MFMAs at BN = 256), each  { LDS reads of the quadrant's operands | one staging
//     unit of LDS-DMA issued | counted vmcnt | barrier | MFMAs | barrier }.  The two wave groups (m halves; one wave of each per
//     SIMD) run ONE barrier apart, so one group's MFMAs cover the other's LDS reads and DMA issue;
//   * two 64 KiB LDS buffers (even / odd K-step), recycled per STAGING UNIT (16 KiB at BN = 256): U0 = the first 64 rows of both
//     groups' x, U2 = the other 64, U3 = the first half of every wave's columns, U1 = the other half.  Phase 0 reads U0 + U3
//     (the U3 fragments stay in registers for phase 3), phase 1 reads U1, phase 2 U2, phase 3 nothing.  A unit is re-staged two
//     phases after its last read (so both wave groups have retired their reads behind a barrier): phase 0 stages U1(t+1), phase 1
//     U2(t+1), phase 2 U0(t+2), phase 3 U3(t+2) — every unit has 5-6 phases (1.25 K-steps of MFMAs) to land, with two buffers.
//     After issuing a phase's unit a wave waits `vmcnt` down to the four newest units; the unit that leaves the count was issued
//     four phases earlier and is read in the NEXT phase (one barrier later: LDS-DMA data is ordered for a reader only by the
//     issuing wave's count plus a barrier the reader has passed).  vmcnt never reaches 0 inside the loop;
//   * workgroup ids are remapped so that the workgroups of one XCD (ids = xcd mod 8) work on neighbouring tiles: 8 m-tiles x 4
//     n-tiles in flight per XCD share their x and w panels in that XCD's L2;
//   * DUAL (gate/up): the dequant pass interleaves the two tensors fragment by fragment (g0 u0 g1 u1 ...), a wave then holds gate
//     and up of the same output columns and the epilogue is silu(gate) * up (mlp.rs:451-469); q/k/v: one launch over the
//     concatenated columns, segments in the epilogue.
#pragma once
#include <type_traits>

#include "wna16.cuh"

#define GX_BM 256
#define GX_BK 64
#define GX_THREADS 512
#define GX_MAX_SEG 3

struct GemmXSeg {
  void* out;
  const void* bias;  // [columns of the segment] or null
  int out_ld;
  int vcol_start;  // first VIRTUAL column of the segment (DUAL: 2 virtual columns per output column), a multiple of 64
};
struct GemmXArgs {
  const void* x;  // [M, x_ld] 16-bit row-major
  int x_ld;
  const void* wd;  // dequantised weights in fragment order: fragment (n-frag nb, k-chunk kc) = 1 KiB at ((nb * K/32) + kc) KiB
  const void* residual;  // [M, res_ld] added after bias (single segment)
  int res_ld;
  GemmXSeg seg[GX_MAX_SEG];
  int nseg;
  int M, NV, K;  // NV = virtual columns (multiple of 16)
  int MT, NT;    // tiles along m and n
  // split-K (mid-size prompts: the output tiles alone leave half the chip idle — M = 2048: o_proj / down_proj are 128 tiles): workgroup
  // (tile, slice) = id / splitk, id % splitk walks K-steps [KS * slice / splitk, ...); slices 0..splitk-2 hand their f32 accumulators
  // to the last one through `slabs` (write-through stores, one flag line per slice, fixed summation order: the exchange of kernels
  // B / C / D).  The whole grid is co-resident (launcher: tiles * splitk <= CUs), so the owner's wait cannot starve its partners.
  int splitk;
  // tail split (large M whose tile count is 1.x rounds of the chip — q/k/v at 4096 rows: 384 tiles on 256 CUs): the first `tail_first`
  // tiles (whole rounds) run over the full K, only the tiles of the last, part-filled round are split `tail_sk` ways — it then takes
  // 1 / tail_sk of a round instead of ~0.8.  Every XCD gets its share of both kinds; a tile's slices are dispatched side by side, the
  // owner last (non-owners never wait, so the owner's wait cannot starve them).
  int tail_first, tail_sk;
  float* slabs;
  uint32_t* counters;
  uint32_t* err;
};

static inline size_t gemm_dense_lds_bytes(int bn) {  // the K loop's two buffers, or the epilogue's [256][bn (+ 8)] tile of outputs
  const size_t loop = (size_t)2 * (32 * 1024 + (size_t)(bn / 16) * 2 * 1024), epi = (size_t)GX_BM * ((size_t)bn * 2 + 16);
  return loop > epi ? loop : epi;
}

// ---- the dequant pass: int4 tile layout -> 16-bit fragments.  One thread = one 16-byte lane word of the int4 layout (8 codes of one
// column for each of 4 k-chunks of 32) -> four 16-byte lane words of four consecutive k-chunk fragments.
//   virtual n-frag of the tensor's n-block nb:  vfrag0 + nb * vstride   (q/k/v: the segment's first fragment, stride 1; gate / up: 0 / 1, stride 2)
// up to three tensors of one GEMM per launch (blockIdx.y: q | k | v, gate | up): same K, group size, layout and format
struct DequantFragBatch {
  const u32x4* tiled[3];
  const uint16_t* scales[3];
  const uint32_t* qzeros[3];
  int N[3], vfrag0[3], vstride[3];
};
template <class DT, bool AWQ>
__global__ __launch_bounds__(256) void dequant_frag_kernel(const DequantFragBatch b, u32x4* __restrict__ wd, int K, int group_size, int layout) {
  const int ti = (int)blockIdx.y;
  const u32x4* __restrict__ tiled = b.tiled[ti];
  const uint16_t* __restrict__ scales = b.scales[ti];
  const uint32_t* __restrict__ qzeros = b.qzeros[ti];
  const int N = b.N[ti], vfrag0 = b.vfrag0[ti], vstride = b.vstride[ti];
  const int KT = K >> 7;
  const int g = group_size > 0 ? group_size : K;
  const bool grouped = group_size > 0 && group_size < K;
  const size_t total = (size_t)(N >> 4) * KT * 64;
  const size_t o = blockIdx.x * (size_t)256 + threadIdx.x;
  if (o >= total) return;
  const int lane = (int)(o & 63);
  const size_t tile = o >> 6;
  const int kt = (int)(tile % KT), nb = (int)(tile / KT);
  const int n = nb * 16 + (lane & 15), oct = lane >> 4;
  const u32x4 w = __builtin_nontemporal_load(tiled + o);
  const size_t fbase = ((size_t)(vfrag0 + nb * vstride) * (K >> 5) + (size_t)kt * 4) * 64 + lane;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int k0 = kt * 128 + j * 32 + oct * 8;
    const int grp = k0 / g;
    const float s = DT::to_f32(scales[vra_scale_index(grp, n, N, layout, grouped)]);
    float z = 8.f;
    if (AWQ) z = (float)((qzeros[(size_t)grp * (N >> 3) + (n >> 3)] >> (4 * awq_rev(n & 7))) & 0xFu);
    const s16x8 f = dequant_word<DT>(w[j], s, -z * s);
    wd[fbase + (size_t)j * 64] = __builtin_bit_cast(u32x4, f);
  }
}

template <class DT, int BN, bool DUAL>
__global__ __launch_bounds__(GX_THREADS, 2) void gemm_dense_kernel(const GemmXArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NF = BN / 16;    // n-frags of the tile
  constexpr int WNF = NF / 4;    // n-frags of a wave (4 | 2)
  constexpr int HNF = WNF / 2;   // ... of one phase
  constexpr int XB = 32 * 1024;  // bytes of x per buffer: 16 m-frags x 2 k-chunks
  constexpr int WB = NF * 2 * 1024;
  constexpr int BUF = XB + WB;
  constexpr int WPU = HNF * 8 / 8;  // LDS-DMA instructions per wave for a unit of w: 4 waves' worth of HNF n-frags x 2 k-chunks over 8 waves
  constexpr int VMW = 2 * 2 + 2 * (HNF == 2 ? 2 : 1);  // DMA instructions of the four newest units (one of each kind)
  constexpr uint32_t RSRC3 = 0x00020000u;
  static_assert(BN == 256 || BN == 128, "tile width");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, q4 = lane >> 4;
  const int K = a.K, M = a.M;
  const int KS = K >> 6;  // K-steps (even: K % 128 == 0)
  int SK = a.splitk > 1 ? a.splitk : 1;  // (wave-uniform; per workgroup with a tail split)

  // ---- tile of this workgroup: ids of one XCD (id mod 8) take a contiguous run of the grouped tile order (8 m-tiles per super-row)
  int mt, nt, zi, xtile, xtiles;  // xtile of xtiles: the tile's index among the tiles that exchange partial sums
  {
    const int nwg = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    int lin;
    if (a.tail_sk > 1) {
      const int pos = bid >> 3, fq = a.tail_first >> 3, tq = (nwg - a.tail_first) >> 3;  // per XCD: full-K tiles, then tail workgroups
      if (pos < fq) {
        lin = xcd * fq + pos, SK = 1, zi = 0;
      } else {
        const int rr = xcd * tq + (pos - fq);
        SK = a.tail_sk;
        lin = a.tail_first + rr / SK;
        zi = rr - (rr / SK) * SK;
      }
      xtile = lin - a.tail_first, xtiles = a.MT * a.NT - a.tail_first;
    } else {
      const int lin_sk = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
      lin = lin_sk / SK;  // (the slices of a tile are neighbours in the remapped order: same XCD, the slabs stay in its L2)
      zi = lin_sk - lin * SK;
      xtile = lin, xtiles = a.MT * a.NT;
    }
    constexpr int GM = 8;
    const int per = GM * a.NT;
    const int grp = lin / per, rem = lin - grp * per;
    const int first = grp * GM, gsz = min(a.MT - first, GM);
    mt = first + rem % gsz;
    nt = rem / gsz;
  }
  const int m0 = mt * GX_BM;

  // ---- LDS-DMA sources.  x: lane p of a fragment fetch = row p >> 2, chunk slot p & 3 holding chunk (p & 3) ^ ((row >> 2) & 3)
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, 0x7FFFFFF0, RSRC3);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wd), 0, 0x7FFFFFF0, RSRC3);
  const int srow = lane >> 2, schunk = (lane & 3) ^ ((srow >> 2) & 3);
  //   this wave stages m-frag mfs (+4 for U2) of the tile: {0,1,2,3,8,9,10,11}[wave]
  const int mfs = wave < 4 ? wave : wave + 4;
  uint32_t vo_x[2];
#pragma unroll
  for (int u = 0; u < 2; u++)
    vo_x[u] = ((uint32_t)min(m0 + (mfs + 4 * u) * 16 + srow, M - 1) * (uint32_t)a.x_ld + (uint32_t)schunk * 8u) * 2u;
  const uint32_t vo_w = (uint32_t)lane * 16u;
  //   w units: U3 = the first HNF n-frags of every wave column, U1 = the others.  BN = 256: this wave stages n-frag (wave >> 1) * 4 + (wave & 1)
  //   (+2 for U1), both k-chunks; BN = 128: n-frag (wave >> 1) * 2 (+1), k-chunk wave & 1
  const int nfs = HNF == 2 ? (wave >> 1) * 4 + (wave & 1) : (wave >> 1) * 2;
  const int kcs = HNF == 2 ? 0 : (wave & 1);
  const int nfv = a.NV >> 4;  // virtual n-frags in all
  uint32_t so_w[2];  // scalar byte offset of the staged n-frag's K-run, U3 / U1
#pragma unroll
  for (int u = 0; u < 2; u++) so_w[u] = (uint32_t)min(nt * NF + nfs + u * HNF, nfv - 1) * (uint32_t)(K >> 5) * 1024u;

  typedef __attribute__((address_space(3))) void* lds_ptr;
  // (no instruction offsets on these loads: with `lds` set the hardware adds inst_offset to the LDS address as well as to the memory address)
  auto stage_x = [&](int buf, int u, int ks) {  // unit U0 (u = 0) / U2 (u = 1) of K-step ks
    const int kc = min(ks, KS - 1);
    unsigned char* d = smem + buf * BUF + (mfs + 4 * u) * 2048;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)d, 16, vo_x[u], (uint32_t)kc * 128u, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(d + 1024), 16, vo_x[u], (uint32_t)kc * 128u + 64u, 0, 0);
  };
  auto stage_w = [&](int buf, int u, int ks) {  // unit U3 (u = 0) / U1 (u = 1)
    const int kc = min(ks, KS - 1);
    unsigned char* d = smem + buf * BUF + XB + ((nfs + u * HNF) * 2 + kcs) * 1024;
    const uint32_t so = so_w[u] + (uint32_t)(kc * 2 + kcs) * 1024u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)d, 16, vo_w, so, 0, 0);
    if constexpr (HNF == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(d + 1024), 16, vo_w, so + 1024u, 0, 0);
  };

  // ---- LDS reads: x fragment (m-frag, kc) of this wave's rows, w fragment (n-frag, kc) of its columns
  const unsigned char* xrd = smem + (wr * 8) * 2048 + (r16 * 4 + (q4 ^ ((r16 >> 2) & 3))) * 16;
  const unsigned char* wrd = smem + XB + (wc * WNF) * 2048 + lane * 16;
  s16x8 xa[4][2], w0[HNF][2], w1[HNF][2];
  auto read_x = [&](int buf, int mh) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int kc = 0; kc < 2; kc++) xa[i][kc] = *reinterpret_cast<const s16x8*>(xrd + buf * BUF + ((mh * 4 + i) * 2 + kc) * 1024);
  };
  auto read_w = [&](int buf, int nh, s16x8 (&w)[HNF][2]) {
#pragma unroll
    for (int j = 0; j < HNF; j++)
#pragma unroll
      for (int kc = 0; kc < 2; kc++) w[j][kc] = *reinterpret_cast<const s16x8*>(wrd + buf * BUF + ((nh * HNF + j) * 2 + kc) * 1024);
  };

  f32x4 acc[8][WNF];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < WNF; j++) acc[i][j] = vra_zero_acc();
  auto mma = [&](int mh, int nh, const s16x8 (&w)[HNF][2]) {  // A = weights (16 columns x 32 k), B = x (16 rows x 32 k): D[column][row]
#pragma unroll
    for (int kc = 0; kc < 2; kc++)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < HNF; j++) DT::mfma(acc[mh * 4 + i][nh * HNF + j], w[j][kc], xa[i][kc]);
  };
#define GX_WAIT_VM()                                                   \
  do {                                                                 \
    if constexpr (VMW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); \
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");              \
  } while (0)
#define GX_COMPUTE(MH, NH, W)                        \
  do {                                               \
    __builtin_amdgcn_s_barrier();                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_setprio(1);                   \
    mma(MH, NH, W);                                  \
    __builtin_amdgcn_s_setprio(0);                   \
    __builtin_amdgcn_sched_barrier(0);               \
    __builtin_amdgcn_s_barrier();                    \
  } while (0)

  // this workgroup's K-steps (an even count: launcher) — all of them, or slice zi of SK
  const int ks0 = KS / SK * zi, ks1 = ks0 + KS / SK;
  // ---- prologue: the first K-step entirely, U0 / U3 of the second (the order the counted waits assume: oldest first)
  stage_x(0, 0, ks0);
  stage_w(0, 0, ks0);
  stage_w(0, 1, ks0);
  stage_x(0, 1, ks0);
  stage_x(1, 0, ks0 + 1);
  stage_w(1, 0, ks0 + 1);
  GX_WAIT_VM();
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind the first

  auto kstep = [&](auto BUFC, int ks) {
    constexpr int b = decltype(BUFC)::value;
    // phase 0: quadrant (rows 0..63, first column half)
    read_w(b, 0, w0);
    __builtin_amdgcn_sched_barrier(0);
    read_x(b, 0);
    stage_w(b ^ 1, 1, ks + 1);
    GX_WAIT_VM();
    GX_COMPUTE(0, 0, w0);
    // phase 1: (rows 0..63, second column half)
    read_w(b, 1, w1);
    stage_x(b ^ 1, 1, ks + 1);
    GX_WAIT_VM();
    GX_COMPUTE(0, 1, w1);
    // phase 2: (rows 64..127, second column half)
    read_x(b, 1);
    stage_x(b, 0, ks + 2);
    GX_WAIT_VM();
    GX_COMPUTE(1, 1, w1);
    // phase 3: (rows 64..127, first column half) — operands already in registers
    stage_w(b, 0, ks + 2);
    GX_WAIT_VM();
    GX_COMPUTE(1, 0, w0);
  };
  for (int ks = ks0; ks < ks1; ks += 2) {
    kstep(std::integral_constant<int, 0>{}, ks);
    kstep(std::integral_constant<int, 1>{}, ks + 1);
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped stagings of the last K-steps
  VRA_MFMA_DRAIN();
#undef GX_WAIT_VM
#undef GX_COMPUTE

  // ---- split-K: the slices of a tile meet through memory (gemm_q4_big.cuh): write-through 16-byte stores, one flag line per slice,
  // the last slice ("owner") polls, adds the slabs to its own partial in slice order and resets the flags
  if (SK > 1) {
    const int tile = xtile, ntiles = xtiles;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, 0x7FFFFFF0, RSRC3);
    auto slab_off = [&](int z, int i, int j) {  // bytes: [slice][tile][m-frag][n-frag][thread] x 16 B
      return (uint32_t)((((z * ntiles + tile) * 8 + i) * WNF + j) * GX_THREADS + tid) * 16u;
    };
    uint32_t* fl = a.counters + (size_t)tile * SK * 16;
    if (zi != SK - 1) {
#pragma unroll
      for (int i = 0; i < 8; i++)
#pragma unroll
        for (int j = 0; j < WNF; j++) {
          const u32x4 v = {__float_as_uint(acc[i][j][0]), __float_as_uint(acc[i][j][1]), __float_as_uint(acc[i][j][2]), __float_as_uint(acc[i][j][3])};
          __builtin_amdgcn_raw_buffer_store_b128(v, srs, slab_off(zi, i, j), 0, 16);
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
    for (int z = 0; z < SK - 1; z++) {  // fixed order; one m-frag row of accumulators per round trip
#pragma unroll
      for (int i = 0; i < 8; i++) {
        u32x4 pz[WNF];
#pragma unroll
        for (int j = 0; j < WNF; j++) pz[j] = __builtin_amdgcn_raw_buffer_load_b128(srs, slab_off(z, i, j), 0, 16);
#pragma unroll
        for (int j = 0; j < WNF; j++)
#pragma unroll
          for (int e = 0; e < 4; e++) acc[i][j][e] += __uint_as_float(pz[j][e]);
      }
    }
  }

  // ---- epilogue, through LDS (round 6): a lane holds 4 columns x 1 row of each accumulator tile — stored from there, every store
  // instruction touches 16 rows in 32-byte pieces (and the residual is read the same way): a residual cost 13-20 us per tile round
  // that way (tools/gemm_dense_overhead.py: 2 K-steps 22.6 -> 35.8 us; now 21.4 -> 26.8; K = 4096 with residual 132-138 -> 121-126 us).  The rounded outputs (bias / SiLU*mul applied) go to LDS as a [256 rows][columns] tile
  // instead (row stride + 16 bytes: the 64 lanes of a ds_write_b64 cover all banks), and the workgroup stores WHOLE ROWS, 16 bytes per lane,
  // adding the residual from equally contiguous loads.  Same roundings at the same points: rnd(acc) [+ bias, rnd] [SiLU*mul] are what LDS
  // holds, rnd(that + residual) is what leaves.
  constexpr int NOUT = DUAL ? WNF / 2 : WNF;  // output n-frags of the wave
  constexpr int OUTC = NOUT * 16 * 4;         // output columns of the tile
  constexpr int ROWB = OUTC * 2 + 16;         // bytes of a row of the LDS tile
  __builtin_amdgcn_s_barrier();  // every wave has drained its LDS-DMA (vmcnt(0) above) and finished its LDS reads: the buffers are free
  {
    const int vcol0 = nt * BN + wc * (WNF * 16);  // first virtual column of this wave (one segment: starts are multiples of 64)
    const bool s1 = a.nseg > 1 && vcol0 >= a.seg[1].vcol_start, s2 = a.nseg > 2 && vcol0 >= a.seg[2].vcol_start;
    const uint16_t* const biasp = static_cast<const uint16_t*>(s2 ? a.seg[2].bias : (s1 ? a.seg[1].bias : a.seg[0].bias));
    const int vrel = vcol0 - (s2 ? a.seg[2].vcol_start : (s1 ? a.seg[1].vcol_start : a.seg[0].vcol_start));
#pragma unroll
    for (int jo = 0; jo < NOUT; jo++) {
      const int j = DUAL ? 2 * jo : jo;  // DUAL: fragments (j, j + 1) = (gate, up) of the same 16 output columns
      const bool live = vcol0 + j * 16 < a.NV;
      float bs[4] = {0.f, 0.f, 0.f, 0.f};
      if (biasp && live) {
        const int n = (DUAL ? vrel / 2 : vrel) + jo * 16 + q4 * 4;
        const u32x2 bw = *reinterpret_cast<const u32x2*>(biasp + n);
        bs[0] = DT::to_f32((uint16_t)(bw[0] & 0xffffu)), bs[1] = DT::to_f32((uint16_t)(bw[0] >> 16));
        bs[2] = DT::to_f32((uint16_t)(bw[1] & 0xffffu)), bs[3] = DT::to_f32((uint16_t)(bw[1] >> 16));
      }
      unsigned char* const lp = smem + (size_t)(wr * 128 + r16) * ROWB + (wc * (NOUT * 16) + jo * 16 + q4 * 4) * 2;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float t = rnd_dt<DT>(acc[i][j][e]);
          if (biasp) t = rnd_dt<DT>(t + bs[e]);
          if (DUAL) {
            const float u = rnd_dt<DT>(acc[i][j + 1][e]);
            const float sl = rnd_dt<DT>(t / (1.0f + expf(-t)));
            t = sl * u;
          }
          v[e] = t;
        }
        *reinterpret_cast<u32x2*>(lp + (size_t)(i * 16) * ROWB) = u32x2{DT::pack2(v[0], v[1]), DT::pack2(v[2], v[3])};
      }
    }
  }
  __syncthreads();
  // whole rows out: thread -> (row, 16-byte chunk of 8 columns)
  constexpr int CPR = OUTC / 8;            // chunks per row
  constexpr int RPP = GX_THREADS / CPR;    // rows per pass of the workgroup
  const int crow = tid / CPR, cch = tid % CPR;
  const int ocol = (DUAL ? nt * (BN / 2) : nt * BN) + cch * 8;  // output column of the chunk (DUAL: of the half-width output)
  const int vcol = DUAL ? 2 * ocol : ocol;                      // its virtual column (segments, the NV bound)
  if (vcol >= a.NV) return;
  const bool c1 = a.nseg > 1 && vcol >= a.seg[1].vcol_start, c2 = a.nseg > 2 && vcol >= a.seg[2].vcol_start;
  uint16_t* const outp = static_cast<uint16_t*>(c2 ? a.seg[2].out : (c1 ? a.seg[1].out : a.seg[0].out));
  const int out_ld = c2 ? a.seg[2].out_ld : (c1 ? a.seg[1].out_ld : a.seg[0].out_ld);
  const int n = ocol - (c2 ? a.seg[2].vcol_start : (c1 ? a.seg[1].vcol_start : a.seg[0].vcol_start)) / (DUAL ? 2 : 1);
#pragma unroll 4
  for (int p = 0; p < GX_BM / RPP; p++) {
    const int row = p * RPP + crow, m = m0 + row;
    if (m >= M) break;
    u32x4 o = *reinterpret_cast<const u32x4*>(smem + (size_t)row * ROWB + cch * 16);
    if (a.residual) {
      const u32x4 rv = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.residual) + (size_t)m * a.res_ld + n);
      float f[8], g[8];
      unpack8<DT>(o, f);
      unpack8<DT>(rv, g);
#pragma unroll
      for (int e = 0; e < 8; e++) f[e] += g[e];
      o = pack8<DT>(f);
    }
    *reinterpret_cast<u32x4*>(outp + (size_t)m * out_ld + n) = o;  // (nt stores measured: no gain, profiles/r06_ab_gemm_dense_epilogue.txt)
  }
}
