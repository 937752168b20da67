// attn_prefill.cuh — causal varlen prefill attention over the paged KV cache, LDS-tiled for gfx950.
// Same operator as the prefill branch of attention_rs::PagedAttention::forward (src/models/layers/attention.rs:808-820):
// query rows of a chunk attend to every cached token of their sequence up to their own position (prefix-cache hits and
// chunked prefill included: keys come from the cache, which reshape_and_cache has already filled for this chunk).
//
// Why a second kernel: paged_attn_kernel gives every wave 16 query rows and lets it pull its own K/V fragments from
// global memory in MFMA layout (16 rows x 64 B per load).  That is right for decode (one row tile per kv head, HBM bound)
// but a 4096-token prefill ran at 60 TFLOP/s (2.3 ms per layer, 45 % of TTFT): every K/V byte crosses L1 once per 16 query
// rows, in quarter-line pieces.  Here a workgroup of 4 waves owns 64*MT query rows of one head and walks 64-key tiles:
//   global -> registers (16 B per lane, coalesced, issued one tile ahead) -> LDS (one copy per workgroup) -> fragments;
//   each wave holds MT row tiles, so one K/V fragment read from LDS feeds MT MFMAs.
// Data flow of one tile, per wave (rq = lane & 15, oct = lane >> 4):
//   S^T = K.Q^T   A = K rows from LDS (key kappa(u, rq), 8 channels of octet oct), B = Q fragments held in registers
//                 -> the lane owns query row rq and 4 keys of each 16-key sub-tile u;
//                 kappa(u, i) = 32*(u>>1) + (i>>2)*8 + (u&1)*4 + (i&3): the lane's 4+4 scores of sub-tiles 2h, 2h+1 are
//                 8 CONSECUTIVE tokens 32h + oct*8 .. +7, i.e. after exp and packing they are the B fragment of
//   O^T += V^T.P^T  A = V rows from LDS (channel t*16+rq, 8 consecutive tokens — the cache's own token-minor layout),
//                 -> the lane owns query row rq again (4 channels per 16-channel tile): the running max, the rescale
//                 factor and the final 1/l are all lane-local, no cross-lane traffic except the two xor-shuffles of the max.
// LDS: K tile [64 keys][D] and V tile [D][64 tokens], 16-byte chunks XOR-swizzled so that each ds_read_b128 lane group
// covers all 64 banks (MI355X_MICROARCH.md §LDS).  One buffer + register prefetch: 2 barriers per tile.
// (Round 6, measured and rejected: 8 waves = 256 query rows per workgroup with two LDS buffers and ONE barrier per tile — half the
// staging instructions per wave, parity-green, and slower: 618 against 738 TFLOP/s at 16 384 tokens, 448 against 486 at 8192
// (profiles/r06_ab_attn_prefill_8_waves.txt).  Two independent 4-wave workgroups per CU drift apart and cover each other's
// softmax; eight waves behind one barrier do not.)
// Roofline: MFMA.  FLOPs = 4 * D * (causal query-key pairs) per head.
#pragma once
#include "common.cuh"
#include "kvcache.cuh"

#define PF_THREADS 256
#define PF_WAVES 4
#define PF_KEYS 64

struct PrefillAttnArgs {
  void* out;                     // [Tq, Hq, D]
  const void* q;                 // [Tq, Hq, D]
  const void* kc;                // K cache [NB, Hkv, BS, D]
  const void* vc;                // V cache [NB, Hkv, D, BS]
  const uint32_t* block_tables;  // [B, max_blocks]
  const uint32_t* context_lens;  // [B]
  const uint32_t* cu_q;          // [B+1]
  int Hq, Hkv, BS, max_blocks;
  int bs_shift;  // log2(BS): this kernel takes power-of-two block sizes (others run paged_attn_kernel)
  float scale_log2e, softcap, scale;
  unsigned long long* ts;  // -DVRA_GEMV_TS builds: per-wave phase cycle sums (tools/attn_prefill_ts.py)
};
#ifdef VRA_GEMV_TS
#define PF_STAMP(v)                               \
  do {                                            \
    __builtin_amdgcn_sched_barrier(0);            \
    v = (long long)__builtin_readcyclecounter();  \
    __builtin_amdgcn_sched_barrier(0);            \
  } while (0)
#else
#define PF_STAMP(v) \
  do {              \
  } while (0)
#endif

template <int D>
__device__ __forceinline__ int pf_kswz(int key, int c) {  // K tile: chunk c of row `key` -> swizzled chunk
  // the 16 rows one fragment read touches are kappa(u, 0..15): distinct in ((key>>3)&3, key&3)
  const int f = (((key >> 3) & 3) << 2) | (key & 3);
  if (D == 128) return c ^ f;
  return c ^ (f >> 1);  // D = 64: 8 chunks per row, two rows per 256-byte bank row
}
__device__ __forceinline__ int pf_vswz(int ch, int c) { return c ^ ((ch >> 1) & 7); }  // V tile rows are 128 B (8 chunks)

template <class DT, int D, bool KV8, int MT>
__global__ __launch_bounds__(PF_THREADS, 2) void prefill_attn_kernel(const PrefillAttnArgs a) {
  typedef typename KVT<KV8>::elem kv_t;
  typedef typename std::conditional<KV8, u32x2, u32x4>::type raw_t;  // 8 cache elements as loaded
  constexpr int DJ = D / 32;             // k-steps of K.Q^T
  constexpr int DT16 = D / 16;           // channel tiles of O
  constexpr int KCH = D / 8;             // 16-byte chunks per K row
  constexpr int ROWS = PF_WAVES * MT * 16;
  constexpr int KLD = (PF_KEYS * D / 8) / PF_THREADS;  // 16-B chunks per thread per tile, K and V alike
  __shared__ __attribute__((aligned(16))) uint16_t Ks[PF_KEYS * D];
  __shared__ __attribute__((aligned(16))) uint16_t Vs[D * PF_KEYS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rq = lane & 15, oct = lane >> 4;
  // (round 6) dispatch order: workgroup ids run over the HEADS first and the row blocks — heaviest first — second, so the 512 co-resident
  // workgroups are always the heaviest blocks left, of every head.  Ids used to run over a head's row blocks first: the last head's
  // heaviest block (twice the average work) was dispatched when ~97 % of the grid had been, and the launch ended on that one workgroup —
  // 4096 tokens: 0.303 -> see profiles/r06_attn_prefill_dispatch_order.txt.  The head of id x is (x mod Hkv) * group + x / Hkv: ids are
  // dealt round-robin to the 8 XCDs, so with 8 kv heads every XCD's L2 holds the K / V panels of ONE kv head.
  const int b = blockIdx.z;
  const int group = a.Hq / a.Hkv;
  const int hk = (int)blockIdx.x % a.Hkv, qhead = hk * group + (int)blockIdx.x / a.Hkv;
  const int ctx = (int)a.context_lens[b];
  const int q0 = (int)a.cu_q[b];
  const int lq = (int)(a.cu_q[b + 1] - a.cu_q[b]);
  // heavy-first: the last row block of a sequence walks the most tiles and is dispatched first
  const int nxb = (lq + ROWS - 1) / ROWS;
  if ((int)blockIdx.y >= nxb) return;
  const int xb = nxb - 1 - (int)blockIdx.y;
  const int wg_i0 = xb * ROWS;
  const int wg_last_pos = ctx - lq + min(wg_i0 + ROWS, lq) - 1;
  const int ntiles = (wg_last_pos >> 6) + 1;
  const int w_i0 = wg_i0 + wave * (MT * 16);
  const bool wave_live = w_i0 < lq;
  const int w_first_pos = ctx - lq + w_i0;
  const int w_last_pos = ctx - lq + min(w_i0 + MT * 16, lq) - 1;

  // ---- Q fragments
  s16x8 qf[MT][DJ];
  int row_pos[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    const int qtok = w_i0 + mt * 16 + rq;
    const bool valid = qtok < lq;
    row_pos[mt] = valid ? ctx - lq + qtok : -1;
    const uint16_t* qp = static_cast<const uint16_t*>(a.q) + ((size_t)(q0 + (valid ? qtok : 0)) * a.Hq + qhead) * D;
#pragma unroll
    for (int j = 0; j < DJ; j++) {
      u32x4 v = *reinterpret_cast<const u32x4*>(qp + j * 32 + oct * 8);
      if (!valid) v = u32x4{0u, 0u, 0u, 0u};
      qf[mt][j] = __builtin_bit_cast(s16x8, v);
    }
  }
  f32x4 o[MT][DT16];
  float m_run[MT], l_run[MT];
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    m_run[mt] = -INFINITY, l_run[mt] = 0.f;
#pragma unroll
    for (int t = 0; t < DT16; t++) o[mt][t] = vra_zero_acc();
  }

  const kv_t* kcache = static_cast<const kv_t*>(a.kc);
  const kv_t* vcache = static_cast<const kv_t*>(a.vc);
  const uint32_t* bt = a.block_tables + (size_t)b * a.max_blocks;
  raw_t kst[KLD], vst[KLD];
  // block ids of a tile's two 32-token halves (a half never straddles a block: BS % 32 == 0); wave-uniform.  They are
  // fetched ONE TILE AHEAD of the loads that need them: looked up at issue time, the dependent block-table round trip
  // stood in front of every tile's K/V loads.  A half that starts past the context has no block: it reads block 0 (any
  // mapped memory will do — its scores are masked and store_tile zeroes its V tokens); nothing selects on loaded data, so
  // the loads stay in flight over the compute.
  // (round 6) The ids come out of a lane-held WINDOW of 64 table entries (one coalesced load per 64 blocks, v_readlane with the
  // wave-uniform index): looked up entry by entry they were two dependent global round trips per tile behind the barrier —
  // `global_load_dword; s_waitcnt vmcnt(0); v_readfirstlane` twice, 966 of the 5237 cycles of a tile (tools/attn_prefill_ts.py,
  // profiles/r06_attn_prefill_phases.txt).  Tiles are walked in ascending order, so one window suffices.
  auto block_of = [&](int tok) -> int { return tok >> a.bs_shift; };  // BS is a power of two (launcher)
  int bt_win = -1;
  uint32_t bt_lane = 0u;
  auto table_entry = [&](int idx) -> uint32_t {  // idx wave-uniform
    const int w = idx >> 6;
    if (w != bt_win) {
      bt_win = w;
      bt_lane = bt[min(w * 64 + lane, a.max_blocks - 1)];
    }
    return (uint32_t)__builtin_amdgcn_readlane((int)bt_lane, idx & 63);
  };
  auto load_blocks = [&](int tile, uint32_t (&blk)[2]) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int tok = (tile << 6) + 32 * h;
      blk[h] = tok < ctx ? table_entry(block_of(tok)) : 0u;
    }
  };
  // thread -> chunks of the tile.  K: chunk i = n*256 + tid of [64 keys][KCH chunks]; V: chunk i of [D channels][8 chunks].
  // The 2*KLD loads of the next tile are NOT issued as one burst behind the barrier: all waves of the workgroup would
  // queue 32 KB at the CU's texture-address unit at the same moment and sit in the issue stall (1600 of 5500 cycles per
  // tile in the phase timers).  next_bases() computes the wave-uniform bases, the loads go out one by one between the MFMA
  // groups of the S and P.V phases.
  size_t kb0 = 0, kb1 = 0, vb0 = 0, vb1 = 0;  // (scalars, not arrays: an array captured by the lambdas below lands in scratch)
  // (element offsets: block * Hkv*BS*D + head * BS*D is common to a block's K and V panels — one 64-bit product per half instead of
  // the four nested ones this used to spend ~300 scalar cycles per tile on)
  const uint32_t blk_elems = (uint32_t)a.Hkv * (uint32_t)a.BS * (uint32_t)D, head_off = (uint32_t)hk * (uint32_t)a.BS * (uint32_t)D;
  auto next_bases = [&](int tile, const uint32_t (&blk)[2]) {
    const int T0 = tile << 6;
    const uint32_t off0 = (uint32_t)(T0 & (a.BS - 1)), off1 = (uint32_t)((T0 + 32) & (a.BS - 1));  // BS is a power of two
    const size_t c0 = (size_t)blk[0] * blk_elems + head_off, c1 = (size_t)blk[1] * blk_elems + head_off;
    kb0 = c0 + off0 * (uint32_t)D;
    kb1 = c1 + off1 * (uint32_t)D;
    vb0 = c0 + off0;
    vb1 = c1 + off1;
  };
  auto issue_k = [&](int n) {
    const int i = n * PF_THREADS + tid;
    const int key = i / KCH, c = i % KCH;
    const bool h = ((n * PF_THREADS / KCH) >> 5) != 0;  // compile-time: 256 chunks never span two halves
    kst[n] = *reinterpret_cast<const raw_t*>(kcache + (h ? kb1 : kb0) + (size_t)(key & 31) * D + c * 8);
  };
  auto issue_v = [&](int n) {
    const int i = n * PF_THREADS + tid;
    const int ch = i >> 3, c = i & 7, h = c >> 2;
    vst[n] = *reinterpret_cast<const raw_t*>(vcache + (h ? vb1 : vb0) + (size_t)ch * a.BS + (c & 3) * 8);
  };
  auto issue_loads = [&](int tile, const uint32_t (&blk)[2]) {
    next_bases(tile, blk);
#pragma unroll
    for (int n = 0; n < KLD; n++) issue_k(n), issue_v(n);
  };
  auto widen = [&](const raw_t& r) -> u32x4 {
    if constexpr (KV8) return vra_unpack_e4m3x8<DT>(r);
    else return r;
  };
  auto store_tile = [&](int tile) {
    const int T0 = tile << 6;
    const bool tail = T0 + PF_KEYS > ctx;
#pragma unroll
    for (int n = 0; n < KLD; n++) {
      const int i = n * PF_THREADS + tid;
      {
        const int key = i / KCH, c = i % KCH;
        *reinterpret_cast<u32x4*>(Ks + key * D + pf_kswz<D>(key, c) * 8) = widen(kst[n]);
      }
      {
        const int ch = i >> 3, c = i & 7;
        u32x4 v = widen(vst[n]);
        if (tail) {  // slots past the context hold arbitrary bits (0 * NaN = NaN): zero them
#pragma unroll
          for (int e = 0; e < 8; e++)
            if (T0 + c * 8 + e >= ctx) v[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
        }
        *reinterpret_cast<u32x4*>(Vs + ch * PF_KEYS + pf_vswz(ch, c) * 8) = v;
      }
    }
  };

  long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0, t7 = 0, sA = 0, sB = 0, sC = 0, sD = 0, sE = 0, sF = 0, sG = 0;
  (void)t0, (void)t1, (void)t2, (void)t3, (void)t4, (void)t5, (void)t6, (void)t7;
  (void)sA, (void)sB, (void)sC, (void)sD, (void)sE, (void)sF, (void)sG;
  uint32_t blk_next[2];
  load_blocks(0, blk_next);
  issue_loads(0, blk_next);
  load_blocks(1, blk_next);
  for (int tile = 0; tile < ntiles; tile++) {
    const int T0 = tile << 6;
    PF_STAMP(t0);
    __syncthreads();  // every wave is done reading the previous tile
    PF_STAMP(t1);
    store_tile(tile);
    PF_STAMP(t2);
    __syncthreads();
    PF_STAMP(t3);
    const bool more = tile + 1 < ntiles;  // (the last tile re-requests itself: never stored)
    next_bases(more ? tile + 1 : tile, blk_next);
    if (more) load_blocks(tile + 2, blk_next);  // (past the last tile: positions >= ctx, no access)
    if (!wave_live || T0 > w_last_pos) {  // wave-uniform; the wave still takes part in loads and barriers
#pragma unroll
      for (int n = 0; n < KLD; n++) issue_k(n), issue_v(n);
      continue;
    }
    PF_STAMP(t4);
    sA += t1 - t0, sB += t2 - t1, sC += t3 - t2, sD += t4 - t3;

    // ---- S^T for the four 16-key sub-tiles.  The K fragments of sub-tile u+1 are read from LDS before the MFMAs of
    // sub-tile u are issued (the MFMAs are volatile asm in program order: hipcc does not hoist the reads by itself, and
    // read -> wait -> 2 MFMAs left the matrix pipe idle for the LDS latency: 1577 cycles for 32 MFMAs)
    f32x4 s[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int u = 0; u < 4; u++) s[mt][u] = vra_zero_acc();
    auto k_frags = [&](int u, u32x4 (&kf)[DJ]) {
      const int key = 32 * (u >> 1) + (rq >> 2) * 8 + (u & 1) * 4 + (rq & 3);
#pragma unroll
      for (int j = 0; j < DJ; j++) kf[j] = *reinterpret_cast<const u32x4*>(Ks + key * D + pf_kswz<D>(key, j * 4 + oct) * 8);
    };
    {
      u32x4 kfr[2][DJ];
      k_frags(0, kfr[0]);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (u + 1 < 4) k_frags(u + 1, kfr[(u + 1) & 1]);
        if (u * KLD / 4 < KLD && (u * KLD) % 4 == 0) issue_k(u * KLD / 4);  // KLD = 4: one per sub-tile; KLD = 2: u = 0, 2
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < DJ; j++)
#pragma unroll
          for (int mt = 0; mt < MT; mt++) DT::mfma(s[mt][u], __builtin_bit_cast(s16x8, kfr[u & 1][j]), qf[mt][j]);
      }
    }
    // the first V fragments are requested now: they land while the softmax runs
    auto v_frags = [&](int t, u32x4 (&vf)[2]) {
      const int ch = t * 16 + rq;
#pragma unroll
      for (int h = 0; h < 2; h++) vf[h] = *reinterpret_cast<const u32x4*>(Vs + ch * PF_KEYS + pf_vswz(ch, h * 4 + oct) * 8);
    };
    u32x4 vfr[3][2];
    v_frags(0, vfr[0]);
    v_frags(1, vfr[1]);
    VRA_MFMA_DRAIN();  // s (and the previous tile's O updates) are complete past this point
    PF_STAMP(t5);
    // ---- softmax update, row tile by row tile; scores in the log2 domain.  The maximum is taken over the raw scores
    // (scale > 0) and the scale rides in the FMA in front of the exp: exp2(s*c - m*c).
    const bool need_mask = T0 + PF_KEYS - 1 > w_first_pos || T0 + PF_KEYS > ctx;  // wave-uniform
    s16x8 pf[MT][2];
    float x[MT][16], tmax[MT];
    float cs = a.scale_log2e;  // x[] -> log2-domain logit: x * cs
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) x[mt][u * 4 + r] = s[mt][u][r];
    }
    if (a.softcap > 0.f) {  // wave-uniform, ONE branch around all elements
      const float inv = a.scale / a.softcap;
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int e = 0; e < 16; e++) x[mt][e] = tanhf(x[mt][e] * inv);
      cs = a.softcap * 1.44269504088896f;
    }
    if (need_mask) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int tok = T0 + 32 * (u >> 1) + oct * 8 + (u & 1) * 4 + r;
            if (tok > row_pos[mt] || tok >= ctx) x[mt][u * 4 + r] = -INFINITY;
          }
    }
    // row maxima of all row tiles together: in-lane, then the two cross-lane steps as VALU lane swaps (no LDS round trip)
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      float m = x[mt][0];
#pragma unroll
      for (int e = 1; e < 16; e++) m = fmaxf(m, x[mt][e]);
      tmax[mt] = m;
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) tmax[mt] = vra_xor16_max(tmax[mt]);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) tmax[mt] = vra_xor32_max(tmax[mt]);
    float alpha[MT];
    bool moved = false;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      const float m_new = fmaxf(m_run[mt], tmax[mt] * cs);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      alpha[mt] = __builtin_amdgcn_exp2f(m_run[mt] - m_safe);
      moved = moved || alpha[mt] != 1.0f;
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        x[mt][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[mt][e], cs, -m_safe));
        psum += x[mt][e];
      }
      l_run[mt] = l_run[mt] * alpha[mt] + psum;  // this lane's share of the row sum (its own 16 keys of every tile)
      m_run[mt] = m_new;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        u32x4 pa;
        pa[0] = DT::pack2(x[mt][h * 8 + 0], x[mt][h * 8 + 1]);
        pa[1] = DT::pack2(x[mt][h * 8 + 2], x[mt][h * 8 + 3]);
        pa[2] = DT::pack2(x[mt][h * 8 + 4], x[mt][h * 8 + 5]);
        pa[3] = DT::pack2(x[mt][h * 8 + 6], x[mt][h * 8 + 7]);
        pf[mt][h] = __builtin_bit_cast(s16x8, pa);
      }
    }
    // rescale O (row rq is this lane's own row): skipped when no row of the wave moved its maximum
    if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
      for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int t = 0; t < DT16; t++)
#pragma unroll
          for (int r = 0; r < 4; r++) o[mt][t][r] *= alpha[mt];
    }
    PF_STAMP(t6);
    // ---- O^T += V^T.P^T, the V fragments two channel tiles ahead of their MFMAs
#pragma unroll
    for (int t = 0; t < DT16; t++) {
      if (t + 2 < DT16) v_frags(t + 2, vfr[(t + 2) % 3]);
      if ((t * KLD) % DT16 == 0) issue_v(t * KLD / DT16);  // spread over the channel tiles
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) DT::mfma(o[mt][t], __builtin_bit_cast(s16x8, vfr[t % 3][h]), pf[mt][h]);
    }
    PF_STAMP(t7);
    sE += t5 - t4, sF += t6 - t5, sG += t7 - t6;
  }
#ifdef VRA_GEMV_TS
  if (a.ts && lane == 0) {
    const size_t wg = (size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (wg < 4096) {
      unsigned long long* t = a.ts + (wg * 4 + wave) * 8;
      t[0] = (unsigned long long)sA, t[1] = (unsigned long long)sB, t[2] = (unsigned long long)sC, t[3] = (unsigned long long)sD;
      t[4] = (unsigned long long)sE, t[5] = (unsigned long long)sF, t[6] = (unsigned long long)sG, t[7] = (unsigned long long)ntiles;
    }
  }
#endif
  VRA_MFMA_DRAIN();  // O is read by the VALU below
  if (!wave_live) return;
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    float l = vra_xor32_sum(vra_xor16_sum(l_run[mt]));
    const int qtok = w_i0 + mt * 16 + rq;
    if (qtok >= lq) continue;
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    uint16_t* op = static_cast<uint16_t*>(a.out) + ((size_t)(q0 + qtok) * a.Hq + qhead) * D + oct * 4;
#pragma unroll
    for (int t = 0; t < DT16; t++) {
      u32x2 w = {DT::pack2(o[mt][t][0] * inv, o[mt][t][1] * inv), DT::pack2(o[mt][t][2] * inv, o[mt][t][3] * inv)};
      *reinterpret_cast<u32x2*>(op + t * 16) = w;
    }
  }
}
