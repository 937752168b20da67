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
  float scale_log2e, softcap, scale;
};

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
  const int b = blockIdx.z, qhead = blockIdx.y;
  const int hk = qhead / (a.Hq / a.Hkv);
  const int ctx = (int)a.context_lens[b];
  const int q0 = (int)a.cu_q[b];
  const int lq = (int)(a.cu_q[b + 1] - a.cu_q[b]);
  // heavy-first: the last row block of a sequence walks the most tiles and is dispatched first
  const int nxb = (lq + ROWS - 1) / ROWS;
  if ((int)blockIdx.x >= nxb) return;
  const int xb = nxb - 1 - (int)blockIdx.x;
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
  // thread -> chunks of the tile.  K: chunk i = n*256 + tid of [64 keys][KCH chunks]; V: chunk i of [D channels][8 chunks]
  auto issue_loads = [&](int tile) {
    const int T0 = tile << 6;
    // the two 32-token halves of the tile: block and offset (a half never straddles a block: BS % 32 == 0); wave-uniform.
    // A half that starts past the context has no block: it reads block 0 (any mapped memory will do — its scores are masked
    // and store_tile zeroes its V tokens); nothing here selects on loaded data, so the loads stay in flight over the compute
    size_t kb[2], vb[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int tok = T0 + 32 * h;
      const uint32_t blk = tok < ctx ? bt[tok / a.BS] : 0u;
      const int off = tok % a.BS;
      kb[h] = (((size_t)blk * a.Hkv + hk) * a.BS + off) * D;
      vb[h] = (((size_t)blk * a.Hkv + hk) * D) * a.BS + off;
    }
#pragma unroll
    for (int n = 0; n < KLD; n++) {
      const int i = n * PF_THREADS + tid;
      {
        const int key = i / KCH, c = i % KCH;
        const int h = (n * PF_THREADS / KCH) >> 5;  // compile-time: 256 chunks never span two halves
        kst[n] = *reinterpret_cast<const raw_t*>(kcache + kb[h] + (size_t)(key & 31) * D + c * 8);
      }
      {
        const int ch = i >> 3, c = i & 7, h = c >> 2;
        vst[n] = *reinterpret_cast<const raw_t*>(vcache + (h ? vb[1] : vb[0]) + (size_t)ch * a.BS + (c & 3) * 8);
      }
    }
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

  issue_loads(0);
  for (int tile = 0; tile < ntiles; tile++) {
    const int T0 = tile << 6;
    __syncthreads();  // every wave is done reading the previous tile
    store_tile(tile);
    __syncthreads();
    if (tile + 1 < ntiles) issue_loads(tile + 1);
    if (!wave_live || T0 > w_last_pos) continue;  // wave-uniform; the wave still takes part in loads and barriers

    // ---- S^T for the four 16-key sub-tiles
    f32x4 s[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
      for (int u = 0; u < 4; u++) s[mt][u] = vra_zero_acc();
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int key = 32 * (u >> 1) + (rq >> 2) * 8 + (u & 1) * 4 + (rq & 3);
#pragma unroll
      for (int j = 0; j < DJ; j++) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + key * D + pf_kswz<D>(key, j * 4 + oct) * 8);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) DT::mfma(s[mt][u], __builtin_bit_cast(s16x8, kf), qf[mt][j]);
      }
    }
    VRA_MFMA_DRAIN();  // s (and the previous tile's O updates) are complete past this point
    // ---- softmax update, row tile by row tile; scores in the log2 domain
    const bool need_mask = T0 + PF_KEYS - 1 > w_first_pos || T0 + PF_KEYS > ctx;  // wave-uniform
    s16x8 pf[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
      float x[16];
      float tmax = -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float v = s[mt][u][r];
          if (a.softcap > 0.f) v = a.softcap * tanhf(v * a.scale / a.softcap) * 1.44269504088896f;
          else v *= a.scale_log2e;
          x[u * 4 + r] = v;
        }
      if (need_mask) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int tok = T0 + 32 * (u >> 1) + oct * 8 + (u & 1) * 4 + r;
            if (tok > row_pos[mt] || tok >= ctx) x[u * 4 + r] = -INFINITY;
          }
      }
#pragma unroll
      for (int e = 0; e < 16; e++) tmax = fmaxf(tmax, x[e]);
      tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
      const float m_new = fmaxf(m_run[mt], tmax);
      const float m_safe = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = __builtin_amdgcn_exp2f(m_run[mt] - m_safe);
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        x[e] = __builtin_amdgcn_exp2f(x[e] - m_safe);
        psum += x[e];
      }
      l_run[mt] = l_run[mt] * alpha + psum;  // this lane's share of the row sum (its own 16 keys of every tile)
      m_run[mt] = m_new;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        u32x4 pa;
        pa[0] = DT::pack2(x[h * 8 + 0], x[h * 8 + 1]);
        pa[1] = DT::pack2(x[h * 8 + 2], x[h * 8 + 3]);
        pa[2] = DT::pack2(x[h * 8 + 4], x[h * 8 + 5]);
        pa[3] = DT::pack2(x[h * 8 + 6], x[h * 8 + 7]);
        pf[mt][h] = __builtin_bit_cast(s16x8, pa);
      }
      // rescale O (row rq is this lane's own row): skipped when no row of the wave moved its maximum
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
        for (int t = 0; t < DT16; t++)
#pragma unroll
          for (int r = 0; r < 4; r++) o[mt][t][r] *= alpha;
      }
    }
    // ---- O^T += V^T.P^T
#pragma unroll
    for (int t = 0; t < DT16; t++) {
      const int ch = t * 16 + rq;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const u32x4 vf = *reinterpret_cast<const u32x4*>(Vs + ch * PF_KEYS + pf_vswz(ch, h * 4 + oct) * 8);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) DT::mfma(o[mt][t], __builtin_bit_cast(s16x8, vf), pf[mt][h]);
      }
    }
  }
  VRA_MFMA_DRAIN();  // O is read by the VALU below
  if (!wave_live) return;
#pragma unroll
  for (int mt = 0; mt < MT; mt++) {
    float l = l_run[mt];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
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
