// attention.hip — paged attention (decode + causal varlen prefill) on MFMA, no LDS staging, no
// transposes.  include/vllm_rs_amd.h §B; restates attention_rs::PagedAttention::forward as called
// from src/models/layers/attention.rs:808-820.
//
// Cache geometry: K [NB, Hkv, BS, D], V [NB, Hkv, D, BS]  (BS % 32 == 0).
// One wave owns a 16-row tile (rows = 16 query tokens of one q-head in prefill; the G q-heads of a
// kv-head in decode) and walks 32-token KV tiles:
//   Sᵀ = K·Qᵀ   A = K rows (lane: token l&15, 8 channels)   B = Qᵀ (held in registers)
//        -> lane (row l&15) holds 4+4 scores of tokens (l>>4)*4+r of the two 16-token halves,
//   softmax statistics per row: in-lane + 2 xor-shuffles (16, 32),
//   O += P·V    A = P: the lane's own 8 probabilities ARE the MFMA A fragment.  The 16 K rows of the first / second MFMA of
//        a tile are the tokens kappa(h, i) = (i>>2)*8 + h*4 + (i&3), so the lane's 4+4 scores belong to the 8 CONSECUTIVE
//        tokens oct*8 .. +7; B = V read token-minor from the transposed V cache with ONE load per channel tile (16 bytes
//        from a 16-bit cache, 8 from an FP8 one).  (Round 1 used rows 0..15 / 16..31: a lane's tokens were two runs of 4 and
//        V took two loads of half the size — the FP8 cache then read half the bytes in the same number of load instructions
//        and was only 9 % faster than the 16-bit one at bs 32 / ctx 4096; with this map it is 26 % faster.)
// Roofline: HBM (KV bytes) for decode; MFMA for long prefill.
#include "common.cuh"
#include "kvcache.cuh"
#include "attn_prefill.cuh"
#include "attn_launch.h"

#define PA_THREADS 256
#define PA_WAVES 4
// the fused decode kernel: a 32-token tile costs a wave one dependent K/V round trip (~2 us) and the waves of a workgroup
// split the tiles of their (sequence, kv head).  8 waves measured worse than 4 except at 8k context (bs 32: 3.84 vs 3.69
// ms/step): the wider merge and the larger workgroup cost more than the shorter tile chains save.
#define FD_WAVES 4
#define FD_THREADS (FD_WAVES * 64)

struct PagedAttnArgs {
  void* out;          // [Tq, Hq, D]
  const void* q;      // [Tq, Hq, D]
  const void* kc;     // K cache
  const void* vc;     // V cache
  const void* kflat;  // non-paged fallback: k [Tk, Hkv, D]
  const void* vflat;  //                     v [Tk, Hkv, D]
  const uint32_t* block_tables;  // [B, max_blocks] or null (fallback)
  const uint32_t* context_lens;  // [B] (paged)
  const uint32_t* cu_q;          // [B+1] or null (decode)
  const uint32_t* cu_k;          // [B+1] (fallback)
  int B, Hq, Hkv, BS, max_blocks;
  float scale_log2e;  // scale * log2(e)
  float softcap;      // 0 = off
  float scale;
  int decode;
  // split-KV across workgroups (decode, long contexts)
  int nsplit;
  float* ws_o;   // [B, Hq, nsplit, D]
  float* ws_ml;  // [B, Hq, nsplit, 2]
  // PagedAttention::new(.., sliding_window, ..) (attention.rs:607-616): > 0: a query at position p sees the keys p-W+1 .. p
  // (attention_rs::mask::causal_mask as restated by vra_causal_mask: j <= i && i - j < W); 0: off
  int sliding_window;
};

template <class DT, int D, bool KV8>
__global__ __launch_bounds__(PA_THREADS) void paged_attn_kernel(const PagedAttnArgs a) {
  typedef typename KVT<KV8>::elem kv_t;  // cache element: 16-bit model dtype, or one E4M3 byte (paged mode only)
  constexpr int DJ = D / 32;  // k-steps of QK^T
  constexpr int DT16 = D / 16;  // output channel tiles
  __shared__ __attribute__((aligned(16))) float lds_o[PA_WAVES][16][D + 4];
  __shared__ float lds_ml[PA_WAVES][16][2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: scalar branches, no exec masking
  const int rq = lane & 15, oct = lane >> 4;
  // token of K row rq within the first 16-key MFMA of a 32-token tile (the second: + 4): kappa(h, i) = (i>>2)*8 + h*4 + (i&3).
  // The lane then owns the scores of the 8 CONSECUTIVE tokens oct*8 .. +7 — its P fragment's k-order — and reads V as one
  // 16-byte (16-bit cache) or 8-byte (FP8) load per channel tile instead of two loads half that size.
  const int krow_tok = (rq >> 2) * 8 + (rq & 3);
  const int G = a.Hq / a.Hkv;
  const int b = blockIdx.z;

  // ---- geometry of this wave's row tile
  int ctx, lq, q0;
  if (a.block_tables) ctx = (int)a.context_lens[b];
  else ctx = (int)(a.cu_k[b + 1] - a.cu_k[b]);
  if (a.cu_q) {
    q0 = (int)a.cu_q[b];
    lq = (int)(a.cu_q[b + 1] - a.cu_q[b]);
  } else {
    q0 = b;
    lq = 1;
  }
  int hk, qtok, qhead;   // per-lane row identity (row rq)
  bool row_valid;
  int tile_first_pos, tile_last_pos;
  int kv_w0, kv_w1;      // this wave's KV tile range [kv_w0, kv_w1) in 32-token tiles
  const int split = a.decode ? blockIdx.x : 0;
  if (a.decode) {
    hk = blockIdx.y;
    qtok = 0;
    qhead = hk * G + rq;
    row_valid = rq < G;
    tile_first_pos = tile_last_pos = ctx - 1;
    const int ntiles = (ctx + 31) >> 5;
    const int t_lo = a.sliding_window > 0 ? max(ctx - a.sliding_window, 0) >> 5 : 0;  // tiles in front of the window are not walked
    // split across workgroups (nsplit) then across the 4 waves
    const int per_split = (ntiles - t_lo + a.nsplit - 1) / a.nsplit;
    const int s0 = min(ntiles, t_lo + split * per_split), s1 = min(ntiles, s0 + per_split);
    const int n_s = s1 - s0;
    kv_w0 = s0 + (n_s * wave) / PA_WAVES;
    kv_w1 = s0 + (n_s * (wave + 1)) / PA_WAVES;
  } else {
    qhead = blockIdx.y;
    hk = qhead / G;
    const int i0 = blockIdx.x * 64 + wave * 16;
    if (i0 >= lq) {
      // whole tile out of range; nothing to do (no cross-wave combine in prefill)
      return;
    }
    qtok = i0 + rq;
    row_valid = qtok < lq;
    tile_first_pos = ctx - lq + i0;
    tile_last_pos = ctx - lq + min(i0 + 15, lq - 1);
    kv_w0 = a.sliding_window > 0 ? max(tile_first_pos - a.sliding_window + 1, 0) >> 5 : 0;
    kv_w1 = (tile_last_pos >> 5) + 1;
  }
  const int row_pos = a.decode ? ctx - 1 : (row_valid ? ctx - lq + qtok : -1);
  const int sw = a.sliding_window;

  // ---- Q fragments (B operand of K·Qᵀ): lane (row rq, octet oct)
  s16x8 qf[DJ];
  {
    const uint16_t* qp = static_cast<const uint16_t*>(a.q) + ((size_t)(q0 + qtok) * a.Hq + qhead) * D;
#pragma unroll
    for (int j = 0; j < DJ; j++) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row_valid) v = *reinterpret_cast<const u32x4*>(qp + j * 32 + oct * 8);
      qf[j] = __builtin_bit_cast(s16x8, v);
    }
  }

  f32x4 o[DT16];
#pragma unroll
  for (int t = 0; t < DT16; t++) o[t] = vra_zero_acc();
  float m_run = -INFINITY, l_run = 0.f;  // per lane: row rq; l_run is this lane's partial sum

  const kv_t* kcache = static_cast<const kv_t*>(a.kc);
  const kv_t* vcache = static_cast<const kv_t*>(a.vc);
  for (int tile = kv_w0; tile < kv_w1; tile++) {
    const int T0 = tile << 5;
    // ---- Sᵀ for the two 16-token halves
    f32x4 s0 = vra_zero_acc(), s1 = vra_zero_acc();
    const uint16_t *krow0 = nullptr, *krow1 = nullptr;  // contiguous (non-paged) k rows
    const kv_t *kc0 = nullptr, *kc1 = nullptr;          // paged cache rows
    size_t vbase = 0;
    if (a.block_tables) {
      const uint32_t blk = a.block_tables[(size_t)b * a.max_blocks + T0 / a.BS];
      const int off = T0 % a.BS;
      kc0 = kcache + (((size_t)blk * a.Hkv + hk) * a.BS + off + krow_tok) * D;
      kc1 = kc0 + 4 * D;
      vbase = (((size_t)blk * a.Hkv + hk) * D) * a.BS + off;
    } else {
      // fallback: rows beyond ctx are clamped (their scores are masked below)
      const size_t kb = a.cu_k[b];
      int t0 = min(T0 + krow_tok, ctx - 1), t1 = min(T0 + krow_tok + 4, ctx - 1);
      krow0 = static_cast<const uint16_t*>(a.kflat) + ((kb + t0) * a.Hkv + hk) * D;
      krow1 = static_cast<const uint16_t*>(a.kflat) + ((kb + t1) * a.Hkv + hk) * D;
    }
    // all K fragments of the tile, then all V fragments, are REQUESTED before the first MFMA (paged mode): the wave then pays one
    // memory round trip per tile instead of one per fragment — with V loaded fragment by fragment inside the P·V loop a
    // 4096-token prefill ran at 60 TFLOP/s (2.3 ms per layer), latency bound on 8 dependent L2 round trips per tile
    u32x4 kf0[DJ], kf1[DJ];
#pragma unroll
    for (int j = 0; j < DJ; j++) {
      if (a.block_tables) {
        kf0[j] = kv_load8<DT, KV8>(kc0 + j * 32 + oct * 8);
        kf1[j] = kv_load8<DT, KV8>(kc1 + j * 32 + oct * 8);
      } else {
        kf0[j] = *reinterpret_cast<const u32x4*>(krow0 + j * 32 + oct * 8);
        kf1[j] = *reinterpret_cast<const u32x4*>(krow1 + j * 32 + oct * 8);
      }
    }
    u32x4 vfr[DT16];  // 8 consecutive tokens of a channel: one load per channel tile
    if (a.block_tables) {
#pragma unroll
      for (int t = 0; t < DT16; t++) {
        vfr[t] = kv_load8<DT, KV8>(vcache + vbase + (size_t)(t * 16 + rq) * a.BS + oct * 8);
      }
    }
#pragma unroll
    for (int j = 0; j < DJ; j++) {
      DT::mfma(s0, __builtin_bit_cast(s16x8, kf0[j]), qf[j]);
      DT::mfma(s1, __builtin_bit_cast(s16x8, kf1[j]), qf[j]);
    }
    VRA_MFMA_DRAIN();  // s0/s1 (and the previous tile's O updates) are complete past this point
    // ---- scale, softcap, causal/length mask; scores in log2 domain
    float sv[8];
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int tok = T0 + oct * 8 + e;
      float x = (e < 4 ? s0[e] : s1[e - 4]);
      if (a.softcap > 0.f) x = a.softcap * tanhf(x * a.scale / a.softcap) * 1.44269504088896f;
      else x *= a.scale_log2e;
      if (tok > row_pos || tok >= ctx || (sw > 0 && row_pos - tok >= sw)) x = -INFINITY;
      sv[e] = x;
      tmax = fmaxf(tmax, x);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_safe);
    float p[8], psum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      p[e] = exp2f(sv[e] - m_safe);
      psum += p[e];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // ---- rescale O: rows of O live at (lane>>4)*4 + r; fetch their alpha from lane (that row)
    float ar[4];
#pragma unroll
    for (int r = 0; r < 4; r++) ar[r] = __shfl(alpha, oct * 4 + r, 64);
    u32x4 pa;
    pa[0] = DT::pack2(p[0], p[1]);
    pa[1] = DT::pack2(p[2], p[3]);
    pa[2] = DT::pack2(p[4], p[5]);
    pa[3] = DT::pack2(p[6], p[7]);
    const s16x8 pfrag = __builtin_bit_cast(s16x8, pa);
    // tail tile: cache slots beyond ctx hold arbitrary bits (0 * NaN = NaN) -> zero those V lanes
    const bool tail = T0 + 32 > ctx;
    uint32_t vm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (tail) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int tok = T0 + oct * 8 + e;
        if (tok >= ctx) vm[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
      }
    }
    // ---- O += P·V, channel tile by channel tile
#pragma unroll
    for (int t = 0; t < DT16; t++) {
      u32x4 vv;
      if (a.block_tables) {
        vv = vfr[t];
      } else {
        const size_t kb = a.cu_k[b];
        uint16_t tmp[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          int tok = min(T0 + oct * 8 + e, ctx - 1);
          tmp[e] = static_cast<const uint16_t*>(a.vflat)[((kb + tok) * a.Hkv + hk) * D + t * 16 + rq];
        }
        vv = u32x4{(uint32_t)tmp[0] | ((uint32_t)tmp[1] << 16), (uint32_t)tmp[2] | ((uint32_t)tmp[3] << 16),
                   (uint32_t)tmp[4] | ((uint32_t)tmp[5] << 16), (uint32_t)tmp[6] | ((uint32_t)tmp[7] << 16)};
      }
      if (tail) {
#pragma unroll
        for (int i = 0; i < 4; i++) vv[i] &= vm[i];
      }
#pragma unroll
      for (int r = 0; r < 4; r++) o[t][r] *= ar[r];
      DT::mfma(o[t], pfrag, __builtin_bit_cast(s16x8, vv));
    }
  }
  VRA_MFMA_DRAIN();  // O is read by the VALU below
  // ---- finish: total l per row (sum the 4 lane groups), bring (m,l) to the O row layout
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);

  if (a.decode) {
    // combine the 4 waves' partial results in LDS
#pragma unroll
    for (int t = 0; t < DT16; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) lds_o[wave][oct * 4 + r][t * 16 + rq] = o[t][r];
    if (oct == 0) {
      lds_ml[wave][rq][0] = m_run;
      lds_ml[wave][rq][1] = l_run;
    }
    __syncthreads();
    // thread -> (row, channel pair...) : 16 rows x D channels over 256 threads
    for (int idx = tid; idx < 16 * D; idx += PA_THREADS) {
      const int row = idx / D, d = idx % D;
      if (row >= G) continue;
      float M = -INFINITY;
#pragma unroll
      for (int w = 0; w < PA_WAVES; w++) M = fmaxf(M, lds_ml[w][row][0]);
      const float Ms = M == -INFINITY ? 0.f : M;
      float L = 0.f, acc = 0.f;
#pragma unroll
      for (int w = 0; w < PA_WAVES; w++) {
        const float f = exp2f(lds_ml[w][row][0] - Ms);
        L += lds_ml[w][row][1] * f;
        acc += lds_o[w][row][d] * f;
      }
      const int head = hk * G + row;
      if (a.nsplit > 1) {
        a.ws_o[(((size_t)b * a.Hq + head) * a.nsplit + split) * D + d] = acc;
        if (d == 0) {
          a.ws_ml[(((size_t)b * a.Hq + head) * a.nsplit + split) * 2 + 0] = M;
          a.ws_ml[(((size_t)b * a.Hq + head) * a.nsplit + split) * 2 + 1] = L;
        }
      } else {
        static_cast<uint16_t*>(a.out)[((size_t)q0 * a.Hq + head) * D + d] = DT::from_f32(L > 0.f ? acc / L : 0.f);
      }
    }
  } else {
    float lr[4];
#pragma unroll
    for (int r = 0; r < 4; r++) lr[r] = __shfl(l_run, oct * 4 + r, 64);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int i = blockIdx.x * 64 + wave * 16 + oct * 4 + r;
      if (i >= lq) continue;
      uint16_t* op = static_cast<uint16_t*>(a.out) + ((size_t)(q0 + i) * a.Hq + qhead) * D;
      const float inv = lr[r] > 0.f ? 1.0f / lr[r] : 0.f;
#pragma unroll
      for (int t = 0; t < DT16; t++) op[t * 16 + rq] = DT::from_f32(o[t][r] * inv);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Fused decode step of one attention layer: RoPE(q, k) + KV-cache write + paged attention in ONE launch
// (the reference runs FusedRope::apply_inplace, reshape_and_cache and PagedAttention::forward as three
// launches, attention.rs:745-820).  Workgroup = (kv split, kv head, sequence); every workgroup rotates
// its own copy of q and of the new k in registers/LDS; the new token's K row and V column are taken from
// LDS instead of the cache, so no workgroup depends on another one's cache write.  Split 0 performs the
// cache write.  q and k are READ ONLY (several workgroups read them concurrently; an in-place update would
// race), i.e. unlike FusedRope::apply_inplace the rotated q / k never reach HBM — nothing reads them later.
// `out` and the caches are bit-identical to the three separate calls.
#ifdef VRA_ATTN_TS
static unsigned long long* g_attn_ts = nullptr;
static unsigned long long* attn_ts_buf() {
  if (!g_attn_ts) {
    (void)hipMalloc(&g_attn_ts, 4096 * 16 * 8);
    (void)hipMemset(g_attn_ts, 0, 4096 * 16 * 8);
  }
  return g_attn_ts;
}
extern "C" void vra_debug_attn_ts(unsigned long long* host, int n) { (void)hipMemcpy(host, attn_ts_buf(), (size_t)n * 8, hipMemcpyDeviceToHost); }
// (round 6) stamps are parked in LDS and written out once at the end: a global store per stamp counts in vmcnt like the loads, hipcc
// then waits vmcnt(0) where it would have counted, and the timeline showed the instrument (gemv.cuh GEMV_STAMP)
#define FD_STAMP(i)                                                  \
  do {                                                               \
    __builtin_amdgcn_sched_barrier(0);                               \
    if ((i) == 0 && a.ts && tid < 16) fd_ts_[tid] = 0ull;            \
    if (a.ts && tid == 0) fd_ts_[(i)] = wall_clock64();              \
    __builtin_amdgcn_sched_barrier(0);                               \
  } while (0)
#define FD_STAMP_DECL __shared__ unsigned long long fd_ts_[16];
#define FD_STAMP_FLUSH()                                                                                                                     \
  do {                                                                                                                                       \
    if (a.ts && tid < 16) a.ts[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + tid] = fd_ts_[tid];            \
  } while (0)
#else
#define FD_STAMP(i) do {} while (0)
#define FD_STAMP_DECL
#define FD_STAMP_FLUSH() do {} while (0)
#endif
struct FusedDecodeArgs {
  void* out;          // [B, Hq, D]
  const void* q;      // [B, Hq, D]   un-rotated, read only
  const void* k;      // [B, Hkv, D]  un-rotated, read only
  const void* v;      // [B, Hkv, D]
  void* kc;           // K cache [NB, Hkv, BS, D]
  void* vc;           // V cache [NB, Hkv, D, BS]
  const void* cosv;   // [n_pos, D/2] model dtype
  const void* sinv;
  const int64_t* positions;      // [B]
  const int64_t* slots;          // [B] (negative: padded lane, nothing is written)
  const uint32_t* block_tables;  // [B, max_blocks]
  const uint32_t* context_lens;  // [B] (includes the new token)
  int B, Hq, Hkv, BS, max_blocks;
  int bs_shift;  // log2(BS), or -1 when BS is not a power of two (set by the launcher)
  float scale_log2e;
  int nsplit;
  float* ws_o;   // [B, Hq, nsplit, D]
  float* ws_ml;  // [B, Hq, nsplit, 2]
  uint16_t* out_frag;  // sequences 0..31 also in kernel W's fragment order (gemv_q4s.cuh GemvSArgs::x_frag: the o_proj launch's x), or null
  unsigned long long* ts;  // VRA_ATTN_TS builds: per-workgroup wall-clock stamps
};
// element (row m < 32, column c) of a [rows, K] activation in kernel W's fragment order (16-bit index)
__device__ __forceinline__ size_t vra_frag_index16(int m, int c) {
  return ((size_t)((((c >> 7) * 2 + (m >> 4)) * 4 + ((c >> 5) & 3)) * 64 + ((c >> 3) & 3) * 16 + (m & 15))) * 8 + (c & 7);
}

// (Round 5 built a "latency form" of this kernel for small grids — the sequence's block ids fetched as one wave-wide load at kernel
// entry and picked with v_readlane, a wave's first K/V tile requested BEFORE the RoPE prologue, further tiles double-buffered, 240
// VGPRs — parity-green and SLOWER than this form at every point measured on one box (bs 1: 1.666 against 1.649 ms per step, ctx 1024
// 1.744 against 1.722, ctx 8000 2.063 against 2.054, bs 32 2.682 against 2.673; profiles/r05_ab_attention_lat.txt): the vector-memory
// path of a CU returns in order, so the early HBM loads stand in front of the prologue's L2 hits and the chain is no shorter.
// Removed; what stayed of it: raw loads first / conversion at use, the split rule and the merge kernel below.)
// (Round 6 built a form that requests a wave's first TWO tiles in the prologue — the second tile's load -> wait -> compute round trip of
// the one or two waves that own two tiles stands between "lds_o written" and the merge barrier, 1.35 us at ctx 150.  In the step's
// trace it LOST: 8.69 against 7.47 us per launch (profiles/r06_trace_attention.txt) — with 8..16 workgroups on the chip the K/V tiles
// arrive at the L1 fill rate of 8..16 CUs, and twice the requests (the clamped second tile of waves that own one included) is twice
// the time.  Removed.)
template <class DT, int D, bool KV8>
__global__ __launch_bounds__(FD_THREADS) void decode_attn_fused_kernel(const FusedDecodeArgs a) {
  typedef typename KVT<KV8>::elem kv_t;
  constexpr int DJ = D / 32, DT16 = D / 16, HALF = D / 2;
  constexpr int KR = KV8 ? D / 64 : DJ;  // 16-byte loads of a lane's share of one K row
  typedef typename std::conditional<KV8, u32x2, u32x4>::type vraw_t;  // a lane's 8 tokens of one V channel as loaded
  __shared__ __attribute__((aligned(16))) float lds_o[FD_WAVES][16][D + 4];
  __shared__ float lds_ml[FD_WAVES][16][2];
  __shared__ __attribute__((aligned(16))) kv_t knew[D];       // the new token's K row in CACHE format (read like a cache row)
  __shared__ __attribute__((aligned(16))) uint16_t vnew[D];   // its V column as the model-dtype values a cache read returns
  FD_STAMP_DECL
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rq = lane & 15, oct = lane >> 4;
  const int krow_tok = (rq >> 2) * 8 + (rq & 3);  // see paged_attn_kernel: K row -> token map that makes a lane's 8 scores consecutive tokens
  FD_STAMP(0);
  const int G = a.Hq / a.Hkv;
  const int b = blockIdx.z, hk = blockIdx.y, split = blockIdx.x;
  // The sequence's block ids as ONE wave-wide load at kernel entry (lane l: block l), in parallel with ctx / pos / slot; a tile's
  // block id is then a v_readlane.  (Rounds 1-4 fetched the id of a tile with a load of its own "one tile ahead" — and hipcc sank
  // that load to its use: every tile started with a dependent round trip, `global_load_dword; s_waitcnt vmcnt(0)` in front of its
  // K/V loads — profiles/r05_timeline_attn_decode.txt: 1.24 us between the prologue barrier and the first K/V load.)
  uint32_t blkvec = a.block_tables[(size_t)b * a.max_blocks + min(lane, a.max_blocks - 1)];
  const int ctx = (int)a.context_lens[b];
  const int64_t pos = a.positions[b];
  const int64_t slot = a.slots[b];
  // block / in-block offset of the new token's slot and of a tile start: shifts when BS is a power of two (a 64-bit
  // division is a ~130-instruction loop on the way to the first cache access)
  const int slot32 = (int)slot;  // slots are < 2^31 (blocks * BS tokens)
  const int slot_blk = a.bs_shift >= 0 ? slot32 >> a.bs_shift : slot32 / a.BS;
  const int slot_off = slot32 - slot_blk * a.BS;
  FD_STAMP(1);
  const uint16_t* cosp = static_cast<const uint16_t*>(a.cosv) + pos * HALF;
  const uint16_t* sinp = static_cast<const uint16_t*>(a.sinv) + pos * HALF;

  // ---- this wave's KV tiles (32 tokens each): split across workgroups, then across the 4 waves.  The block id of
  // the first tile is requested NOW (it only needs ctx): one dependent round trip less before the first K / V load
  const int ntiles = (ctx + 31) >> 5;
  const int per_split = a.nsplit > 1 ? (ntiles + a.nsplit - 1) / a.nsplit : ntiles;
  const int s0t = min(ntiles, split * per_split), s1t = min(ntiles, s0t + per_split);
  const int n_s = s1t - s0t;
  const int kv_w0 = s0t + ((n_s * wave) >> 2), kv_w1 = s0t + ((n_s * (wave + 1)) >> 2);
  static_assert(FD_WAVES == 4, "tile split assumes 4 waves");
  int vec_base = 0;  // the block index lane 0 of `blkvec` holds
  auto tile_blk = [&](int tile) -> uint32_t {
    const int T0 = tile << 5;
    const int bi = __builtin_amdgcn_readfirstlane(a.bs_shift >= 0 ? T0 >> a.bs_shift : T0 / a.BS);
    // (the vector is re-based on the wave's first block below, so bi - vec_base is in 0..63 for every tile of the wave; clamped rather
    // than backed by a load: a load on a never-taken path still put `s_waitcnt vmcnt` between the tile requests of the prologue)
    return (uint32_t)__builtin_amdgcn_readlane((int)blkvec, min(max(bi - vec_base, 0), 63));
  };
  const kv_t* kcache = static_cast<const kv_t*>(a.kc);
  const kv_t* vcache = static_cast<const kv_t*>(a.vc);
  // the raw loads of one tile (nothing is converted or consumed here: all of them go out back to back)
  auto load_tile = [&](int tile, uint32_t blk, u32x4 (&kr0)[KR], u32x4 (&kr1)[KR], vraw_t (&vr)[DT16]) {
    const int T0 = tile << 5;
    const int off = a.bs_shift >= 0 ? T0 & (a.BS - 1) : T0 % a.BS;
    const kv_t* krow0 = kcache + (((size_t)blk * a.Hkv + hk) * a.BS + off + krow_tok) * D;
    const kv_t* krow1 = krow0 + 4 * D;
    const size_t vbase = (((size_t)blk * a.Hkv + hk) * D) * a.BS + off;
#pragma unroll
    for (int i = 0; i < KR; i++) {
      kr0[i] = *reinterpret_cast<const u32x4*>(krow0 + (KV8 ? oct * (D / 4) + i * 16 : i * 32 + oct * 8));
      kr1[i] = *reinterpret_cast<const u32x4*>(krow1 + (KV8 ? oct * (D / 4) + i * 16 : i * 32 + oct * 8));
    }
    // V: they only depend on the block table; consumed after the softmax
#pragma unroll
    for (int t = 0; t < DT16; t++) vr[t] = *reinterpret_cast<const vraw_t*>(vcache + vbase + (size_t)(t * 16 + rq) * a.BS + oct * 8);
  };
  {
    // a wave whose tiles reach beyond block 63 (contexts above 4096 tokens at 64-token blocks) re-bases the vector on its own
    // first block: one dependent load, once per wave (a wave's share never spans more than 64 blocks: <= 16 tiles per wave)
    const int t_hi = min(kv_w1, ntiles - 1) << 5;
    if ((a.bs_shift >= 0 ? t_hi >> a.bs_shift : t_hi / a.BS) >= 64) {
      const int t_lo = min(kv_w0, max(ntiles - 1, 0)) << 5;
      vec_base = __builtin_amdgcn_readfirstlane(a.bs_shift >= 0 ? t_lo >> a.bs_shift : t_lo / a.BS);
      blkvec = a.block_tables[(size_t)b * a.max_blocks + min(vec_base + lane, a.max_blocks - 1)];
    }
  }
  // first tile of this wave, clamped to the sequence's last tile: the requests below are UNCONDITIONAL (straight-line code lets hipcc
  // count them in its s_waitcnt; inside branches it assumed none were issued and made the new token's staging wait for all of them).
  // A wave without tiles re-reads a tile its neighbours read; a padded lane (ctx 0: its table row is not validated by the engine,
  // ADVICE r5) reads block 0.
  const int t_first = min(kv_w0, max(ntiles - 1, 0));
  uint32_t blk_cur = ntiles > 0 ? tile_blk(t_first) : 0u;
  u32x4 ka0[KR], ka1[KR], kb0[KR], kb1[KR];
  vraw_t va[DT16], vb[DT16];
  // ---- (round 6) EVERY load of the prologue goes out before the first wait: the new token's k / v rows and their cos / sin, q and
  // its cos / sin, then this wave's first K/V tile(s) — L2 hits in front, the cache (HBM / Infinity Cache) behind them, so the
  // in-order return path of the CU delays nothing.  Rounds 1-5 staged the new k / v first (its own round trip, 0.46 -> 1.19 us on the
  // timeline) and only then requested q (a second one, -> 1.94 us).
  // new token: threads 0 .. D/16-1 rotate k, threads 64 .. 64+D/8-1 copy v (every thread loads — clamped, the same few lines)
  const bool kthr = tid < HALF / 8, vthr = tid >= 64 && tid < 64 + D / 8;
  const int kt8 = (tid & (HALF / 8 - 1)) * 8, vc8 = (tid & (D / 8 - 1)) * 8;
  const uint16_t* kp = static_cast<const uint16_t*>(a.k) + ((size_t)b * a.Hkv + hk) * D;
  const u32x4 nk_a = *reinterpret_cast<const u32x4*>(kp + kt8), nk_b = *reinterpret_cast<const u32x4*>(kp + HALF + kt8);
  const u32x4 nk_c = *reinterpret_cast<const u32x4*>(cosp + kt8), nk_s = *reinterpret_cast<const u32x4*>(sinp + kt8);
  const u32x4 nv_v = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.v) + ((size_t)b * a.Hkv + hk) * D + vc8);
  // Q fragments, rotated in registers: lane (row rq = q head of the group, octet oct)
  const bool row_valid = rq < G;
  const int qhead = hk * G + rq;
  s16x8 qf[DJ];
  // Channel map of the QK^T contraction (k-step j, lane octet oct, element e): the 16-bit cache uses j*32 + oct*8 + e (a K row is
  // DJ 16-byte loads per lane either way); the FP8 cache uses oct*(D/4) + j*8 + e — lane-contiguous, so that a lane's share of a
  // K row (D/4 bytes) is D/64 16-byte loads instead of DJ 8-byte ones (the FP8 cache read half the bytes in the same number
  // of load instructions).  q is laid out to match; the contraction order inside an MFMA changes, nothing else.
  const uint16_t* qp = static_cast<const uint16_t*>(a.q) + ((size_t)b * a.Hq + qhead) * D;
  constexpr int QJ = KV8 ? DJ : DJ / 2;
  u32x4 q_a[QJ], q_b[QJ], q_c[QJ], q_s[QJ];
  const bool first = oct < 2;  // (FP8 map) channels oct*(D/4) .. : the first half of the head for oct 0, 1
#pragma unroll
  for (int j = 0; j < QJ; j++) {
    const int c = KV8 ? oct * (D / 4) + j * 8 : j * 32 + oct * 8, cl = KV8 ? c & (HALF - 1) : c;
    q_a[j] = q_b[j] = u32x4{0u, 0u, 0u, 0u};
    if (row_valid) {
      q_a[j] = *reinterpret_cast<const u32x4*>(qp + c);
      q_b[j] = *reinterpret_cast<const u32x4*>(qp + (KV8 ? (first ? c + HALF : c - HALF) : HALF + c));
    }
    q_c[j] = *reinterpret_cast<const u32x4*>(cosp + cl);
    q_s[j] = *reinterpret_cast<const u32x4*>(sinp + cl);
  }
  __builtin_amdgcn_sched_barrier(0);
  load_tile(t_first, blk_cur, ka0, ka1, va);
  // a wave that owns a SECOND tile (5..8 tiles per workgroup: contexts of 129..256 tokens unsplit, the two-way split up to 512, ...)
  // requests it now as well — its load -> wait -> compute round trip used to stand between "lds_o written" and the merge barrier of
  // everybody else, 1.6 us of the 6.4 at ctx 150 (profiles/r06_timeline_attn_decode.txt).  Only that wave: requesting a clamped
  // second tile from every wave doubled the L1 fills of the 8..16 CUs that run a short context and lost 1.2 us per launch
  // (profiles/r06_trace_attention.txt).  (hipcc cannot count loads behind a branch: this wave's first tile waits for both.)
  const bool two = kv_w0 + 1 < kv_w1;  // (wave-uniform)
  if (two) load_tile(kv_w0 + 1, tile_blk(kv_w0 + 1), kb0, kb1, vb);
  __builtin_amdgcn_sched_barrier(0);
  // ---- new token: rotate k, copy v; stage both in LDS (and in the cache: split 0 only)
  u32x4 nk_r1 = {0u, 0u, 0u, 0u}, nk_r2 = {0u, 0u, 0u, 0u};
  if (kthr) {
    float x1[8], x2[8], cs[8], sn[8], y1[8], y2[8];
    unpack8<DT>(nk_a, x1);
    unpack8<DT>(nk_b, x2);
    unpack8<DT>(nk_c, cs);
    unpack8<DT>(nk_s, sn);
#pragma unroll
    for (int e = 0; e < 8; e++) {
      y1[e] = x1[e] * cs[e] - x2[e] * sn[e];
      y2[e] = x2[e] * cs[e] + x1[e] * sn[e];
    }
    nk_r1 = pack8<DT>(y1), nk_r2 = pack8<DT>(y2);
    kv_store8<DT, KV8>(knew + tid * 8, nk_r1);
    kv_store8<DT, KV8>(knew + HALF + tid * 8, nk_r2);
  } else if (vthr) {
    *reinterpret_cast<u32x4*>(vnew + (tid - 64) * 8) = kv_roundtrip8<DT, KV8>(nv_v);
  }
  // (the CACHE writes of the new token come after the q rotation: stores count in vmcnt like the loads, and with them queued here hipcc
  // made the rotation wait for every tile load — the barrier then stood behind the cache round trip instead of under it)
  auto store_new_token = [&]() {
  if (kthr) {
    if (split == 0 && slot >= 0) {
      kv_t* kcp = static_cast<kv_t*>(a.kc) + ((((size_t)slot_blk) * a.Hkv + hk) * a.BS + slot_off) * D;
      kv_store8<DT, KV8>(kcp + tid * 8, nk_r1);
      kv_store8<DT, KV8>(kcp + HALF + tid * 8, nk_r2);
    }
  } else if (vthr) {
    const int c = tid - 64;
    const u32x4 vv = nv_v;
    if (split == 0 && slot >= 0) {
      kv_t* vcp = static_cast<kv_t*>(a.vc) + (((size_t)slot_blk) * a.Hkv + hk) * D * a.BS + slot_off;
      if constexpr (KV8) {
        const u32x2 q8 = vra_pack_e4m3x8<DT>(vv);
#pragma unroll
        for (int e = 0; e < 8; e++) vcp[(size_t)(c * 8 + e) * a.BS] = (uint8_t)(q8[e >> 2] >> (8 * (e & 3)));
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          vcp[(size_t)(c * 8 + 2 * e) * a.BS] = (uint16_t)(vv[e] & 0xffffu);
          vcp[(size_t)(c * 8 + 2 * e + 1) * a.BS] = (uint16_t)(vv[e] >> 16);
        }
      }
    }
  }
  };
  FD_STAMP(2);
#pragma unroll
  for (int j = 0; j < QJ; j++) {
    float x1[8], x2[8], cs[8], sn[8];
    unpack8<DT>(q_a[j], x1);
    unpack8<DT>(q_b[j], x2);
    unpack8<DT>(q_c[j], cs);
    unpack8<DT>(q_s[j], sn);
    if constexpr (KV8) {
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; e++) y[e] = first ? x1[e] * cs[e] - x2[e] * sn[e] : x1[e] * cs[e] + x2[e] * sn[e];
      qf[j] = __builtin_bit_cast(s16x8, pack8<DT>(y));
    } else {
      float y1[8], y2[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        y1[e] = x1[e] * cs[e] - x2[e] * sn[e];
        y2[e] = x2[e] * cs[e] + x1[e] * sn[e];
      }
      qf[j] = __builtin_bit_cast(s16x8, pack8<DT>(y1));
      qf[j + DJ / 2] = __builtin_bit_cast(s16x8, pack8<DT>(y2));
    }
  }
  FD_STAMP(3);
  __syncthreads();  // knew / vnew staged
  FD_STAMP(4);
  store_new_token();

  const int last = ctx - 1;

  f32x4 o[DT16];
#pragma unroll
  for (int t = 0; t < DT16; t++) o[t] = vra_zero_acc();
  float m_run = -INFINITY, l_run = 0.f;
  // one tile: QK^T, online softmax, PV on the RAW registers of that tile (converted here: FP8 caches widen to 16 bits)
  // FIRST (compile time): the wave's first tile — o is still zero and alpha = exp2(-inf) = 0: the rescale of o (32 multiplies behind four
  // cross-lane reads of alpha) is skipped; 0 * 0 = 0, the same bits
  auto compute_tile = [&](int tile, const u32x4 (&kr0)[KR], const u32x4 (&kr1)[KR], const vraw_t (&vr)[DT16], auto first_c) {
    constexpr bool FIRST = decltype(first_c)::value;
    const int T0 = tile << 5;
    const bool has_new = last >= T0 && last < T0 + 32;  // wave-uniform: the tile that holds the new token
    u32x4 k0[DJ], k1[DJ];
    if constexpr (KV8) {
#pragma unroll
      for (int j = 0; j < DJ; j++) {
        k0[j] = vra_unpack_e4m3x8<DT>(u32x2{kr0[j >> 1][(j & 1) * 2], kr0[j >> 1][(j & 1) * 2 + 1]});
        k1[j] = vra_unpack_e4m3x8<DT>(u32x2{kr1[j >> 1][(j & 1) * 2], kr1[j >> 1][(j & 1) * 2 + 1]});
      }
    } else {
#pragma unroll
      for (int j = 0; j < DJ; j++) k0[j] = kr0[j], k1[j] = kr1[j];
    }
    u32x4 vfr[DT16];
#pragma unroll
    for (int t = 0; t < DT16; t++) {
      if constexpr (KV8) vfr[t] = vra_unpack_e4m3x8<DT>(vr[t]);
      else vfr[t] = vr[t];
    }
    FD_STAMP(5);
    // the new token's K row comes from LDS (the cache write of split 0 may not be visible yet).  Patched in AFTER the loads:
    // selecting the row POINTER (cache or LDS) made every K load of the loop a flat_load — slower to issue than global_load
    // and counted in both vmcnt and lgkmcnt
    if (has_new) {
      if (T0 + krow_tok == last) {
#pragma unroll
        for (int j = 0; j < DJ; j++) k0[j] = kv_load8<DT, KV8>(knew + (KV8 ? oct * (D / 4) + j * 8 : j * 32 + oct * 8));
      }
      if (T0 + krow_tok + 4 == last) {
#pragma unroll
        for (int j = 0; j < DJ; j++) k1[j] = kv_load8<DT, KV8>(knew + (KV8 ? oct * (D / 4) + j * 8 : j * 32 + oct * 8));
      }
    }
    f32x4 s0 = vra_zero_acc(), s1 = vra_zero_acc();
#pragma unroll
    for (int j = 0; j < DJ; j++) {
      DT::mfma(s0, __builtin_bit_cast(s16x8, k0[j]), qf[j]);
      DT::mfma(s1, __builtin_bit_cast(s16x8, k1[j]), qf[j]);
    }
    VRA_MFMA_DRAIN();
    FD_STAMP(6);
    float sv[8];
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int tok = T0 + oct * 8 + e;
      float x = (e < 4 ? s0[e] : s1[e - 4]) * a.scale_log2e;
      if (tok >= ctx) x = -INFINITY;
      sv[e] = x;
      tmax = fmaxf(tmax, x);
    }
    tmax = vra_xor16_max(tmax);  // (v_permlane16/32_swap: VALU, not the LDS round trip of __shfl_xor; the same maxima)
    tmax = vra_xor32_max(tmax);
    const float m_new = fmaxf(m_run, tmax);
    const float m_safe = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = exp2f(m_run - m_safe);
    float p[8], psum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      p[e] = exp2f(sv[e] - m_safe);
      psum += p[e];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    float ar[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (!FIRST) {
      // alpha of q row oct*4 + r.  Groups of <= 4 q heads per kv head (Llama-3: 4, Qwen2-7B: 7 takes the general path): only rows 0..3
      // are real and lanes 0..3 hold their alphas — v_readlane into SGPRs instead of four ds_bpermute round trips; the lanes of
      // oct > 0 scale rows that are never stored
      if (G <= 4) {
#pragma unroll
        for (int r = 0; r < 4; r++) ar[r] = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(alpha), r));
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) ar[r] = __shfl(alpha, oct * 4 + r, 64);
      }
    }
    u32x4 pa;
    pa[0] = DT::pack2(p[0], p[1]);
    pa[1] = DT::pack2(p[2], p[3]);
    pa[2] = DT::pack2(p[4], p[5]);
    pa[3] = DT::pack2(p[6], p[7]);
    const s16x8 pfrag = __builtin_bit_cast(s16x8, pa);
    FD_STAMP(7);
    const bool tail = T0 + 32 > ctx;
    uint32_t vm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    int new_e = -1;  // which of this lane's 8 tokens is the new one
    if (tail || has_new) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int tok = T0 + oct * 8 + e;
        if (tok >= ctx) vm[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
        if (tok == last) new_e = e;
      }
    }
#pragma unroll
    for (int t = 0; t < DT16; t++) {
      u32x4 vv = vfr[t];
      if (has_new && new_e >= 0) {
        const uint32_t nv = vnew[t * 16 + rq];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if ((new_e >> 1) == i) vv[i] = (new_e & 1) ? ((vv[i] & 0x0000ffffu) | (nv << 16)) : ((vv[i] & 0xffff0000u) | nv);
        }
      }
      if (tail) {
#pragma unroll
        for (int i = 0; i < 4; i++) vv[i] &= vm[i];
      }
      if constexpr (!FIRST) {
#pragma unroll
        for (int r = 0; r < 4; r++) o[t][r] *= ar[r];
      }
      DT::mfma(o[t], pfrag, __builtin_bit_cast(s16x8, vv));
    }
  };
  if (kv_w0 < kv_w1) compute_tile(kv_w0, ka0, ka1, va, std::true_type{});  // (its loads went out in front of the barrier)
  if (two) compute_tile(kv_w0 + 1, kb0, kb1, vb, std::false_type{});  // (requested in the prologue)
  for (int tile = kv_w0 + 2; tile < kv_w1; tile++) {
    load_tile(tile, tile_blk(tile), ka0, ka1, va);
    compute_tile(tile, ka0, ka1, va, std::false_type{});
  }
  FD_STAMP(8);
  VRA_MFMA_DRAIN();
  l_run = vra_xor16_sum(l_run);  // (a + b in either order: the same bits as the __shfl_xor form)
  l_run = vra_xor32_sum(l_run);
#pragma unroll
  for (int t = 0; t < DT16; t++)
#pragma unroll
    for (int r = 0; r < 4; r++) lds_o[wave][oct * 4 + r][t * 16 + rq] = o[t][r];
  if (oct == 0) {
    lds_ml[wave][rq][0] = m_run;
    lds_ml[wave][rq][1] = l_run;
  }
  FD_STAMP(9);
  __syncthreads();
  FD_STAMP(10);
  for (int idx = tid; idx < G * D; idx += FD_THREADS) {
    const int row = idx / D, d = idx % D;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < FD_WAVES; w++) M = fmaxf(M, lds_ml[w][row][0]);
    const float Ms = M == -INFINITY ? 0.f : M;
    float L = 0.f, acc = 0.f;
#pragma unroll
    for (int w = 0; w < FD_WAVES; w++) {
      const float f = exp2f(lds_ml[w][row][0] - Ms);
      L += lds_ml[w][row][1] * f;
      acc += lds_o[w][row][d] * f;
    }
    const int head = hk * G + row;
    if (a.nsplit > 1) {
      a.ws_o[(((size_t)b * a.Hq + head) * a.nsplit + split) * D + d] = acc;
      if (d == 0) {
        a.ws_ml[(((size_t)b * a.Hq + head) * a.nsplit + split) * 2 + 0] = M;
        a.ws_ml[(((size_t)b * a.Hq + head) * a.nsplit + split) * 2 + 1] = L;
      }
    } else {
      const uint16_t ov = DT::from_f32(L > 0.f ? acc / L : 0.f);
      static_cast<uint16_t*>(a.out)[((size_t)b * a.Hq + head) * D + d] = ov;
      if (a.out_frag && b < 32) a.out_frag[vra_frag_index16(b, head * D + d)] = ov;
    }
  }
  FD_STAMP(11);
  FD_STAMP_FLUSH();
}

// (Round 6 built the merge INTO the attention launch — write-through partial rows, an arrival ticket per (sequence, kv head), the last
// workgroup to arrive combines them — to save this launch (4.6 us per layer in the step's trace).  Same-box A/B, profiles/r06_ab_attention_merge.txt:
// -0.6 % of the bs-1 step at two splits, but +1.8 % at ctx 1024 (8 splits) and +4 % at ctx 8000 (32): the last arriver walks agent-scope
// loads of every split on ONE workgroup per kv head where this kernel has one per q head behind a kernel boundary.  Removed.)
// second pass for split-KV decode: merge nsplit partials per (b, head).  The (max, sum) pairs of the splits go through LDS once
// (thread s fetches split s: one round trip instead of a dependent pair per split), the partial rows are fetched eight splits at
// a time (independent loads, clamped index with weight 0) and added in split order — the same sums in the same order as the
// one-split-at-a-time loop of rounds 1-4.  nsplit <= 64 <= D.
template <class DT, int D>
__global__ void paged_attn_merge_kernel(uint16_t* out, const float* ws_o, const float* ws_ml, int Hq, int nsplit, uint16_t* out_frag = nullptr) {
  __shared__ float s_m[64], s_l[64];
  const int b = blockIdx.y, head = blockIdx.x, d = threadIdx.x;
  const size_t base = ((size_t)b * Hq + head) * nsplit;
  if (d < nsplit) {
    s_m[d] = ws_ml[(base + d) * 2];
    s_l[d] = ws_ml[(base + d) * 2 + 1];
  }
  __syncthreads();
  float M = -INFINITY;
  for (int s = 0; s < nsplit; s++) M = fmaxf(M, s_m[s]);
  const float Ms = M == -INFINITY ? 0.f : M;
  float L = 0.f, acc = 0.f;
  for (int s0 = 0; s0 < nsplit; s0 += 8) {
    float o8[8];
#pragma unroll
    for (int j = 0; j < 8; j++) o8[j] = ws_o[(base + min(s0 + j, nsplit - 1)) * D + d];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (s0 + j < nsplit) {
        const float f = exp2f(s_m[s0 + j] - Ms);
        L += s_l[s0 + j] * f;
        acc += o8[j] * f;
      }
    }
  }
  const uint16_t ov = DT::from_f32(L > 0.f ? acc / L : 0.f);
  out[((size_t)b * Hq + head) * D + d] = ov;
  if (out_frag && b < 32) out_frag[vra_frag_index16(b, head * D + d)] = ov;
}

static int num_cus_attn() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) ? p.multiProcessorCount : 256;
  }
  return n;
}
static int decode_nsplit(int batch, int kv_heads, int max_context_len) {
  int wg = batch * kv_heads;
  int tiles = (max_context_len + 31) / 32;
  int s = 1;
  static const char* e = getenv("VRA_ATTN_SPLIT_TILES");  // tuning aid: fewest tiles a split keeps (overrides both rules)
  const int forced = e && atoi(e) > 0 ? atoi(e) : 0;
  const int cus = num_cus_attn();
  if (wg * 2 <= cus) {
    // Small batches: as many splits as keep the grid within the CUs (one workgroup per CU at most) and leave a split at least
    // 4 tiles (one per wave).  bs 1: 2 splits from 256 tokens, ... 32 (256 workgroups) at 8k — rounds 1-4 stopped at 64
    // workgroups because their merge kernel walked the splits one dependent round trip at a time; measured on one box
    // (profiles/r05_ab_attention_splits.txt): keep 4 beats 2, 8 and 16 at ctx 150..400, 1024 and 8000.
    const int keep = forced ? forced : 4;
    // (more than one workgroup per CU loses: ctx 8000 2.04 -> 2.17 ms per step at 2 and at 4, profiles/r05_ab_attention_wg_per_cu.txt)
    static const char* pc = getenv("VRA_ATTN_WG_PER_CU");  // tuning aid: workgroups per CU the split count may reach
    const int cap = cus * (pc && atoi(pc) > 0 ? atoi(pc) : 1);
    while (wg * s * 2 <= cap && tiles / (s * 2) >= keep && s < 64) s *= 2;
    return s;
  }
  // enough workgroups to cover the chip (256 CUs); each split keeps >= 8 tiles (256 tokens): at bs 32 shorter splits do not pay
  // for the merge launch (3.46 -> 3.54 ms per step), and past 64 workgroups per ... the wider merge loses
  while (wg * s < 512 && tiles / (s * 2) >= (forced ? forced : 8) && s < 64) s *= 2;
  return s;
}

extern "C" size_t vra_paged_attention_decode_workspace_bytes(int32_t max_batch, int32_t q_heads, int32_t head_dim,
                                                             int32_t max_context_len) {
  (void)max_context_len;
  return (size_t)max_batch * q_heads * 64 * (head_dim + 2) * sizeof(float);
}

template <class DT>
static void launch_attn(const PagedAttnArgs& a, int D, bool kv8, dim3 grid, hipStream_t st) {
  if (D == 128) {
    if (kv8) paged_attn_kernel<DT, 128, true><<<grid, PA_THREADS, 0, st>>>(a);
    else paged_attn_kernel<DT, 128, false><<<grid, PA_THREADS, 0, st>>>(a);
  } else if (D == 64) {
    if (kv8) paged_attn_kernel<DT, 64, true><<<grid, PA_THREADS, 0, st>>>(a);
    else paged_attn_kernel<DT, 64, false><<<grid, PA_THREADS, 0, st>>>(a);
  } else {
    vra_set_error("paged attention: head_dim %d not supported (64, 128)", D);
  }
}
#ifdef VRA_GEMV_TS
static unsigned long long* g_pf_ts = nullptr;
static unsigned long long* vra_attn_pf_ts_buf() {
  if (!g_pf_ts) {
    (void)hipMalloc(&g_pf_ts, 4096 * 32 * 8);
    (void)hipMemset(g_pf_ts, 0, 4096 * 32 * 8);
  }
  return g_pf_ts;
}
extern "C" void vra_debug_attn_pf_ts(unsigned long long* host, int n) { (void)hipMemcpy(host, vra_attn_pf_ts_buf(), (size_t)n * 8, hipMemcpyDeviceToHost); }
#endif
// kv_dtype: the activation dtype (16-bit cache) or VRA_FP8_E4M3
static bool kv_dtype_ok(const char* who, int dtype, int kv_dtype) {
  if (kv_dtype == dtype || kv_dtype == VRA_FP8_E4M3) return true;
  vra_set_error("%s: kv_dtype must be the activation dtype or VRA_FP8_E4M3", who);
  return false;
}

extern "C" void vra_paged_attention_decode(void* out, const void* q, const void* k_cache, const void* v_cache,
                                           const uint32_t* block_tables, const uint32_t* context_lens, int32_t batch,
                                           int32_t q_heads, int32_t kv_heads, int32_t head_dim, int32_t block_size,
                                           int32_t max_blocks_per_seq, int32_t max_context_len, float scale, float softcap,
                                           void* workspace, int32_t dtype, int32_t kv_dtype, int64_t stream) {
  vra_paged_attention_decode_sw(out, q, k_cache, v_cache, block_tables, context_lens, batch, q_heads, kv_heads, head_dim, block_size,
                                max_blocks_per_seq, max_context_len, scale, softcap, 0, workspace, dtype, kv_dtype, stream);
}
extern "C" void vra_paged_attention_decode_sw(void* out, const void* q, const void* k_cache, const void* v_cache,
                                              const uint32_t* block_tables, const uint32_t* context_lens, int32_t batch,
                                              int32_t q_heads, int32_t kv_heads, int32_t head_dim, int32_t block_size,
                                              int32_t max_blocks_per_seq, int32_t max_context_len, float scale, float softcap,
                                              int32_t sliding_window, void* workspace, int32_t dtype, int32_t kv_dtype, int64_t stream) {
  VRA_CHECK_ARG(sliding_window >= 0, "vra_paged_attention_decode: sliding_window must be >= 0 (0 = off)");
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_paged_attention_decode: dtype must be bf16/f16");
  if (!kv_dtype_ok("vra_paged_attention_decode", dtype, kv_dtype)) return;
  VRA_CHECK_ARG(block_size % 32 == 0, "vra_paged_attention_decode: block_size must be a multiple of 32");
  VRA_CHECK_ARG(q_heads % kv_heads == 0 && q_heads / kv_heads <= 16, "vra_paged_attention_decode: need Hq %% Hkv == 0 and group <= 16");
  VRA_CHECK_ARG(out && q && k_cache && v_cache && block_tables && context_lens, "vra_paged_attention_decode: null pointer");
  if (batch <= 0) return;
  PagedAttnArgs a = {};
  a.out = out;
  a.q = q;
  a.kc = k_cache;
  a.vc = v_cache;
  a.block_tables = block_tables;
  a.context_lens = context_lens;
  a.B = batch;
  a.Hq = q_heads;
  a.Hkv = kv_heads;
  a.BS = block_size;
  a.max_blocks = max_blocks_per_seq;
  a.scale = scale;
  a.scale_log2e = scale * 1.44269504088896f;
  a.softcap = softcap;
  a.decode = 1;
  a.sliding_window = sliding_window;
  // (the split decision counts the tiles that are walked: a window shorter than the context shortens every chain)
  a.nsplit = workspace ? decode_nsplit(batch, kv_heads, sliding_window > 0 && sliding_window + 31 < max_context_len ? sliding_window + 31 : max_context_len) : 1;
  a.ws_o = static_cast<float*>(workspace);
  a.ws_ml = a.ws_o ? a.ws_o + (size_t)batch * q_heads * a.nsplit * head_dim : nullptr;
  dim3 grid(a.nsplit, kv_heads, batch);
  hipStream_t st = as_stream(stream);
  if (dtype == VRA_BF16) launch_attn<BF16>(a, head_dim, kv_dtype == VRA_FP8_E4M3, grid, st);
  else launch_attn<F16>(a, head_dim, kv_dtype == VRA_FP8_E4M3, grid, st);
  if (a.nsplit > 1) {
    dim3 mg(q_heads, batch);
#define VRA_MERGE(DT, DD) paged_attn_merge_kernel<DT, DD><<<mg, DD, 0, st>>>((uint16_t*)out, a.ws_o, a.ws_ml, q_heads, a.nsplit)
    if (dtype == VRA_BF16) {
      if (head_dim == 128) VRA_MERGE(BF16, 128);
      else if (head_dim == 64) VRA_MERGE(BF16, 64);
    } else {
      if (head_dim == 128) VRA_MERGE(F16, 128);
      else if (head_dim == 64) VRA_MERGE(F16, 64);
    }
#undef VRA_MERGE
  }
}

extern "C" void vra_paged_attention_prefill(void* out, const void* q, const void* k, const void* v, const void* k_cache,
                                            const void* v_cache, const uint32_t* block_tables, const uint32_t* context_lens,
                                            const uint32_t* cu_seqlens_q, const uint32_t* cu_seqlens_k, int32_t batch,
                                            int32_t total_q, int32_t max_seqlen_q, int32_t q_heads, int32_t kv_heads,
                                            int32_t head_dim, int32_t block_size, int32_t max_blocks_per_seq, float scale,
                                            float softcap, int32_t dtype, int32_t kv_dtype, int64_t stream) {
  vra_paged_attention_prefill_sw(out, q, k, v, k_cache, v_cache, block_tables, context_lens, cu_seqlens_q, cu_seqlens_k, batch, total_q,
                                 max_seqlen_q, q_heads, kv_heads, head_dim, block_size, max_blocks_per_seq, scale, softcap, 0, dtype, kv_dtype,
                                 stream);
}
extern "C" void vra_paged_attention_prefill_sw(void* out, const void* q, const void* k, const void* v, const void* k_cache,
                                               const void* v_cache, const uint32_t* block_tables, const uint32_t* context_lens,
                                               const uint32_t* cu_seqlens_q, const uint32_t* cu_seqlens_k, int32_t batch,
                                               int32_t total_q, int32_t max_seqlen_q, int32_t q_heads, int32_t kv_heads,
                                               int32_t head_dim, int32_t block_size, int32_t max_blocks_per_seq, float scale,
                                               float softcap, int32_t sliding_window, int32_t dtype, int32_t kv_dtype, int64_t stream) {
  (void)total_q;
  VRA_CHECK_ARG(sliding_window >= 0, "vra_paged_attention_prefill: sliding_window must be >= 0 (0 = off)");
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_paged_attention_prefill: dtype must be bf16/f16");
  if (!kv_dtype_ok("vra_paged_attention_prefill", dtype, kv_dtype)) return;
  VRA_CHECK_ARG(block_tables || kv_dtype == dtype, "vra_paged_attention_prefill: contiguous k/v are in the activation dtype");
  VRA_CHECK_ARG(out && q && cu_seqlens_q, "vra_paged_attention_prefill: null pointer");
  VRA_CHECK_ARG(q_heads % kv_heads == 0, "vra_paged_attention_prefill: need Hq %% Hkv == 0");
  if (block_tables) {
    VRA_CHECK_ARG(k_cache && v_cache && context_lens, "vra_paged_attention_prefill: paged mode needs caches and context_lens");
    VRA_CHECK_ARG(block_size % 32 == 0, "vra_paged_attention_prefill: block_size must be a multiple of 32");
  } else {
    VRA_CHECK_ARG(k && v && cu_seqlens_k, "vra_paged_attention_prefill: contiguous mode needs k, v, cu_seqlens_k");
  }
  if (batch <= 0 || max_seqlen_q <= 0) return;
  PagedAttnArgs a = {};
  a.out = out;
  a.q = q;
  a.kc = k_cache;
  a.vc = v_cache;
  a.kflat = k;
  a.vflat = v;
  a.block_tables = block_tables;
  a.context_lens = context_lens;
  a.cu_q = cu_seqlens_q;
  a.cu_k = cu_seqlens_k;
  a.B = batch;
  a.Hq = q_heads;
  a.Hkv = kv_heads;
  a.BS = block_size;
  a.max_blocks = max_blocks_per_seq;
  a.scale = scale;
  a.scale_log2e = scale * 1.44269504088896f;
  a.softcap = softcap;
  a.decode = 0;
  a.nsplit = 1;
  a.sliding_window = sliding_window;
  // (a sliding window runs on the generic kernel below — mask + skipped tiles; the LDS-tiled kernel is the full-causal fast path)
  if (sliding_window == 0 && block_tables && (head_dim == 128 || head_dim == 64) && (block_size & (block_size - 1)) == 0 && !getenv("VRA_NO_PREFILL_TILED")) {
    // the LDS-tiled prefill kernel (attn_prefill.cuh); 2 row tiles per wave once that still leaves >= 2 workgroups per CU
    PrefillAttnArgs p = {};
    p.out = out, p.q = q, p.kc = k_cache, p.vc = v_cache;
    p.block_tables = block_tables, p.context_lens = context_lens, p.cu_q = cu_seqlens_q;
    p.Hq = q_heads, p.Hkv = kv_heads, p.BS = block_size, p.max_blocks = max_blocks_per_seq;
    p.bs_shift = __builtin_ctz((unsigned)block_size);
    p.scale = scale, p.scale_log2e = a.scale_log2e, p.softcap = softcap;
#ifdef VRA_GEMV_TS
    p.ts = vra_attn_pf_ts_buf();
#endif
    const bool kv8 = kv_dtype == VRA_FP8_E4M3;
    const long wgs2 = (long)((max_seqlen_q + 127) / 128) * q_heads * batch;
    static const char* mt_env = getenv("VRA_PF_MT2_FROM");  // tuning aid: 128-row workgroups from this many of them
    const int mt = wgs2 >= (mt_env ? atol(mt_env) : 512) ? 2 : 1;
    dim3 pg(q_heads, (max_seqlen_q + 64 * mt - 1) / (64 * mt), batch);  // x: heads (fastest), y: row blocks, heaviest first
    hipStream_t st = as_stream(stream);
#define VRA_PF(DT, DD, K8, MM) prefill_attn_kernel<DT, DD, K8, MM><<<pg, PF_THREADS, 0, st>>>(p)
#define VRA_PF_MT(DT, DD, K8) \
  do {                        \
    if (mt == 2) VRA_PF(DT, DD, K8, 2); \
    else VRA_PF(DT, DD, K8, 1);         \
  } while (0)
#define VRA_PF_KV(DT, DD)            \
  do {                               \
    if (kv8) VRA_PF_MT(DT, DD, true); \
    else VRA_PF_MT(DT, DD, false);    \
  } while (0)
    if (dtype == VRA_BF16) {
      if (head_dim == 128) VRA_PF_KV(BF16, 128);
      else VRA_PF_KV(BF16, 64);
    } else {
      if (head_dim == 128) VRA_PF_KV(F16, 128);
      else VRA_PF_KV(F16, 64);
    }
#undef VRA_PF_KV
#undef VRA_PF_MT
#undef VRA_PF
    return;
  }
  dim3 grid((max_seqlen_q + 63) / 64, q_heads, batch);
  if (dtype == VRA_BF16) launch_attn<BF16>(a, head_dim, kv_dtype == VRA_FP8_E4M3, grid, as_stream(stream));
  else launch_attn<F16>(a, head_dim, kv_dtype == VRA_FP8_E4M3, grid, as_stream(stream));
}

extern "C" void vra_rope_cache_attention_decode(void* out, const void* q, const void* k, const void* v, void* k_cache, void* v_cache,
                                                const void* cos, const void* sin, const int64_t* positions,
                                                const int64_t* slot_mapping, const uint32_t* block_tables,
                                                const uint32_t* context_lens, int32_t batch, int32_t q_heads,
                                                int32_t kv_heads, int32_t head_dim, int32_t block_size,
                                                int32_t max_blocks_per_seq, int32_t max_context_len, float scale,
                                                void* workspace, int32_t dtype, int32_t kv_dtype, int64_t stream) {
  vra_rope_cache_attention_decode_frag(out, q, k, v, k_cache, v_cache, cos, sin, positions, slot_mapping, block_tables, context_lens, batch, q_heads,
                                       kv_heads, head_dim, block_size, max_blocks_per_seq, max_context_len, scale, workspace, dtype, kv_dtype, nullptr,
                                       stream);
}
// internal (native runtime): the same launch, the output ALSO in kernel W's fragment order (rows 0..31, K = q_heads * head_dim, a
// multiple of 128) for the o_proj launch that follows
void vra_rope_cache_attention_decode_frag(void* out, const void* q, const void* k, const void* v, void* k_cache, void* v_cache, const void* cos,
                                          const void* sin, const int64_t* positions, const int64_t* slot_mapping, const uint32_t* block_tables,
                                          const uint32_t* context_lens, int32_t batch, int32_t q_heads, int32_t kv_heads, int32_t head_dim,
                                          int32_t block_size, int32_t max_blocks_per_seq, int32_t max_context_len, float scale, void* workspace,
                                          int32_t dtype, int32_t kv_dtype, void* out_frag, int64_t stream) {
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_rope_cache_attention_decode: dtype must be bf16/f16");
  if (!kv_dtype_ok("vra_rope_cache_attention_decode", dtype, kv_dtype)) return;
  VRA_CHECK_ARG(out && q && k && v && k_cache && v_cache && cos && sin && positions && slot_mapping && block_tables && context_lens,
                "vra_rope_cache_attention_decode: null pointer");
  VRA_CHECK_ARG(block_size % 32 == 0, "vra_rope_cache_attention_decode: block_size must be a multiple of 32");
  VRA_CHECK_ARG(q_heads % kv_heads == 0 && q_heads / kv_heads <= 16, "vra_rope_cache_attention_decode: need Hq %% Hkv == 0 and group <= 16");
  VRA_CHECK_ARG(head_dim == 64 || head_dim == 128, "vra_rope_cache_attention_decode: head_dim %d not supported (64, 128)", head_dim);
  if (batch <= 0) return;
  FusedDecodeArgs a = {};
  a.out = out;
  a.q = q;
  a.k = k;
  a.v = v;
  a.kc = k_cache;
  a.vc = v_cache;
  a.cosv = cos;
  a.sinv = sin;
  a.positions = positions;
  a.slots = slot_mapping;
  a.block_tables = block_tables;
  a.context_lens = context_lens;
  a.B = batch;
  a.Hq = q_heads;
  a.Hkv = kv_heads;
  a.BS = block_size;
  a.max_blocks = max_blocks_per_seq;
  a.bs_shift = (block_size & (block_size - 1)) == 0 ? 31 - __builtin_clz((unsigned)block_size) : -1;
#ifdef VRA_ATTN_TS
  a.ts = attn_ts_buf();
#endif
  a.scale_log2e = scale * 1.44269504088896f;
  a.nsplit = workspace ? decode_nsplit(batch, kv_heads, max_context_len) : 1;
  a.ws_o = static_cast<float*>(workspace);
  a.ws_ml = a.ws_o ? a.ws_o + (size_t)batch * q_heads * a.nsplit * head_dim : nullptr;
  a.out_frag = (q_heads * head_dim) % 128 == 0 ? static_cast<uint16_t*>(out_frag) : nullptr;
  dim3 grid(a.nsplit, kv_heads, batch);
  hipStream_t st = as_stream(stream);
  const bool kv8 = kv_dtype == VRA_FP8_E4M3;
#define VRA_FD(DT, DD)                                                                   \
  do {                                                                                   \
    if (kv8) decode_attn_fused_kernel<DT, DD, true><<<grid, FD_THREADS, 0, st>>>(a);     \
    else decode_attn_fused_kernel<DT, DD, false><<<grid, FD_THREADS, 0, st>>>(a);        \
  } while (0)
  if (dtype == VRA_BF16) {
    if (head_dim == 128) VRA_FD(BF16, 128);
    else VRA_FD(BF16, 64);
  } else {
    if (head_dim == 128) VRA_FD(F16, 128);
    else VRA_FD(F16, 64);
  }
#undef VRA_FD
  if (a.nsplit > 1) {
    dim3 mg(q_heads, batch);
#define VRA_MERGE(DT, DD) paged_attn_merge_kernel<DT, DD><<<mg, DD, 0, st>>>((uint16_t*)out, a.ws_o, a.ws_ml, q_heads, a.nsplit, a.out_frag)
    if (dtype == VRA_BF16) {
      if (head_dim == 128) VRA_MERGE(BF16, 128);
      else VRA_MERGE(BF16, 64);
    } else {
      if (head_dim == 128) VRA_MERGE(F16, 128);
      else VRA_MERGE(F16, 64);
    }
#undef VRA_MERGE
  }
}
