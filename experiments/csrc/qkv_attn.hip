// qkv_attn.hip — decode steps of 1..4 sequences: RMSNorm + q/k/v GEMV (kernel E, gemv_q4s.cuh) + RoPE + KV-cache write + paged
// attention in ONE launch.  The reference runs them as norm, three WNA16 GEMMs, FusedRope::apply_inplace, reshape_and_cache and
// PagedAttention::forward (llama.rs:115-121, attention.rs:745-820); rounds 1-3 ran two launches (kernel E, then
// decode_attn_fused_kernel plus a merge launch when the context bucket split the KV range).
//
// Why: at one row the attention launch is pure latency — 7.9 us during which it fetches 0.9 MB — and it sits behind a kernel
// boundary (1.2-1.5 us) and four dependent round trips (context length / position -> block table -> K/V tile -> merge).
// Here the workgroup that owns the first unit of a kv head's K columns is also that head's attention workgroup:
//   * its GEMV part is kernel E unchanged, except that every workgroup's outputs leave as 8-byte {two values, tag} granules,
//     one write-through (sc1) store each (gemv_q4s_body<.., GRAN = true>): no flag, no counter, no fence — the tag IS the flag
//     (tag = (forward epoch << 8) + layer, the epoch word is bumped by the embedding launch of every forward, so a captured
//     graph replays with fresh tags);
//   * right behind its own epilogue the attention workgroup requests everything that does not depend on q/k/v — context
//     length, position, block-table entry, cos/sin rows and the K / V rows of each wave's first 32-token tile (16 waves: 512
//     tokens in flight) — and only then sweeps the (G + 2) x D / 2 granules of its head (sc1 loads, retried until every tag
//     matches): the HBM round trip of the tiles overlaps the tail of the other workgroups' GEMV streams;
//   * the attention itself is decode_attn_fused_kernel's arithmetic (Sᵀ = K·Qᵀ, lane-local softmax statistics, P as the A
//     fragment of P·V, the new token's K row / V column patched in from LDS) with the 32-token tiles dealt round-robin to 16
//     waves and ONE merge through LDS — no KV split across workgroups, no merge launch.
// All workgroups of the launch are co-resident (grid <= #CUs, one 16-wave workgroup per CU, as every kernel-E launch), so the
// wait cannot deadlock; it is bounded anyway (0.5 s) and raises the scratch error word the engine polls every step.
// Roofline: HBM (the q/k/v weights); algorithmic bytes as the q/k/v launch + ctx * Hkv * D * 4 per sequence.
#include "qkv_attn.h"

#include "kvcache.cuh"
#include "scratch.h"

void vra_gemv_s_plan(int n_units, int* grid, int* q, int* r);                               // wna16_gemm.hip
bool vra_gemv_s_fits(int ns, int M, int K, int group_size, int n_units, bool norm);  // wna16_gemm.hip

#define QA_MAX_G 8
#ifdef VRA_GEMV_TS
unsigned long long* vra_gemv_ts_buf_shared();  // wna16_gemm.hip
#define QA_STAMP(i)                                                            \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    if (ts && tid == 0) ts[(size_t)blockIdx.x * 32 + (i)] = wall_clock64();    \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#else
#define QA_STAMP(i) do {} while (0)
#endif
__host__ __device__ static inline size_t qa_al16(size_t x) { return (x + 15) & ~(size_t)15; }
static inline size_t qa_attn_lds_bytes(int G, int D) {
  // new K row + V column (256 B each), rotated q, per-wave (m, l), per-wave O partials
  return 512 + (size_t)G * D * 2 + qa_al16((size_t)GS_WAVES * G * 2 * 4) + (size_t)GS_WAVES * G * (D + 4) * 4;
}

// ---------------------------------------------------------------------------------------------- attention of one (sequence, kv head)
template <class DT, int D>
__device__ __forceinline__ void qa_attention(const QkvAttnTail& t, const void* gran, int gran_ld, int b, int hk, uint32_t tag,
                                             unsigned char* smem, unsigned long long* ts, int ctx, int pos32, int slot32, uint32_t blk_first) {
  constexpr int DJ = D / 32, DT16 = D / 16, HALF = D / 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  QA_STAMP(19);
  const int rq = lane & 15, oct = lane >> 4;
  const int krow_tok = (rq >> 2) * 8 + (rq & 3);  // K row -> token map of attention.hip: a lane's 8 scores are consecutive tokens
  const int G = t.Hq / t.Hkv;
  // ---- LDS carve-up (the GEMV part is done with it: the caller put a barrier in between)
  uint16_t* knew = reinterpret_cast<uint16_t*>(smem);        // the new token's rotated K row
  uint16_t* vnew = reinterpret_cast<uint16_t*>(smem + 256);  // its V column
  uint16_t* qr = reinterpret_cast<uint16_t*>(smem + 512);    // rotated q [G][D]
  float* lds_ml = reinterpret_cast<float*>(smem + 512 + (size_t)G * D * 2);  // [16 waves][G][2]
  float* lds_o = reinterpret_cast<float*>(smem + 512 + (size_t)G * D * 2 + qa_al16((size_t)GS_WAVES * G * 2 * 4));  // [16 waves][G][D + 4]

  const int64_t slot = slot32;
  const int slot_blk = t.bs_shift >= 0 ? slot32 >> t.bs_shift : slot32 / t.BS;
  const int slot_off = slot32 - slot_blk * t.BS;
  const int ntiles = (ctx + 31) >> 5;
  const int last = ctx - 1;
  const uint16_t* kcache = static_cast<const uint16_t*>(t.kc);
  const uint16_t* vcache = static_cast<const uint16_t*>(t.vc);
  auto tile_blk_index = [&](int tile) {
    const int T0 = tile << 5;
    return (size_t)b * t.max_blocks + (t.bs_shift >= 0 ? T0 >> t.bs_shift : T0 / t.BS);
  };
  u32x4 k0[DJ], k1[DJ], vfr[DT16];
  auto load_tile = [&](int tile, uint32_t blk) {
    const int T0 = tile << 5;
    const int off = t.bs_shift >= 0 ? T0 & (t.BS - 1) : T0 % t.BS;
    const uint16_t* krow0 = kcache + (((size_t)blk * t.Hkv + hk) * t.BS + off + krow_tok) * D;
    const uint16_t* krow1 = krow0 + 4 * D;
    const size_t vbase = (((size_t)blk * t.Hkv + hk) * D) * t.BS + off;
#pragma unroll
    for (int j = 0; j < DJ; j++) {
      k0[j] = *reinterpret_cast<const u32x4*>(krow0 + j * 32 + oct * 8);
      k1[j] = *reinterpret_cast<const u32x4*>(krow1 + j * 32 + oct * 8);
    }
#pragma unroll
    for (int tt = 0; tt < DT16; tt++) vfr[tt] = *reinterpret_cast<const u32x4*>(vcache + vbase + (size_t)(tt * 16 + rq) * t.BS + oct * 8);
  };
  // ---- requested first, before q / k / v exist: the K / V rows of this wave's first tile (its block id came in with the
  // kernel's first loads) and the cos / sin words of the rotating threads
  // tile i of the context belongs to wave (4 + i) % 16: the waves that carry the q/k/v hand-over (0 .. 3) get a tile only from
  // the 13th on, and they request it BEHIND their sweep — the CU's vector-memory queue returns in order, and a granule load
  // queued behind sixteen waves' K / V rows waited 1.6 .. 2.9 us (profiles/r04_timeline_qkv_attn.txt)
  const int tile0 = (wave + 12) & 15;
  const int NR = (G + 1) * (HALF / 2);
  const bool has_role = wave * 64 < NR + HALF;  // (wave-uniform)
  if (!has_role && tile0 < ntiles) load_tile(tile0, blk_first);
  QA_STAMP(21);
  // thread roles of the hand-over: r < NR rotates two channel pairs (c, c+1 | c+HALF, c+HALF+1) of q head hh < G or of the new
  // k (hh == G): it sweeps ITS two granules, rotates, and writes the rotated values — no staging of the raw q/k/v, no barrier
  // between sweep and rotation; the next D/2 threads carry the new v (one granule each)
  const bool is_rot = tid < NR, is_v = tid >= NR && tid < NR + HALF;
  const int hh = tid / (HALF / 2), c = (tid % (HALF / 2)) * 2;  // (rotating threads)
  const bool isk = hh == G;
  uint32_t cw = 0u, sw = 0u;
  if (is_rot) {
    cw = *reinterpret_cast<const uint32_t*>(static_cast<const uint16_t*>(t.cosv) + (size_t)pos32 * HALF + c);
    sw = *reinterpret_cast<const uint32_t*>(static_cast<const uint16_t*>(t.sinv) + (size_t)pos32 * HALF + c);
  }
  QA_STAMP(22);
  {
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(gran), 0, 0x7FFFFFF0, 0x00020000);
    const unsigned long long t_lim = wall_clock64() + 50000000ull;  // 0.5 s of the 100 MHz counter
    if (has_role) {
      const int colbase = isk ? t.Hq * D + hk * D : (hk * G + min(hh, G - 1)) * D;  // launch-wide column of the row's channel 0
      const int vcol = (t.Hq + t.Hkv) * D + hk * D + (tid - NR) * 2;
      const uint32_t ga = (uint32_t)(b * gran_ld + (is_v ? vcol : colbase + c) / 2);
      const uint32_t gb = (uint32_t)(b * gran_ld + (colbase + HALF + c) / 2);
      const bool two = is_rot, any = is_rot || is_v;
      u32x2 va, vb;
      for (;;) {
        va = __builtin_amdgcn_raw_buffer_load_b64(grs, (any ? ga : 0u) * 8u, 0, 16);  // sc1
        vb = __builtin_amdgcn_raw_buffer_load_b64(grs, (two ? gb : 0u) * 8u, 0, 16);  // sc1
        if (__all((!any || va[1] == tag) && (!two || vb[1] == tag))) break;
        if (wall_clock64() > t_lim) {
          if (lane == 0) __hip_atomic_store(t.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
      QA_STAMP(23);
      if (is_rot) {
        const float x1a = DT::to_f32((uint16_t)(va[0] & 0xffffu)), x1b = DT::to_f32((uint16_t)(va[0] >> 16));
        const float x2a = DT::to_f32((uint16_t)(vb[0] & 0xffffu)), x2b = DT::to_f32((uint16_t)(vb[0] >> 16));
        const float ca = DT::to_f32((uint16_t)(cw & 0xffffu)), cb = DT::to_f32((uint16_t)(cw >> 16));
        const float sa = DT::to_f32((uint16_t)(sw & 0xffffu)), sb = DT::to_f32((uint16_t)(sw >> 16));
        const uint32_t r1 = DT::pack2(x1a * ca - x2a * sa, x1b * cb - x2b * sb);
        const uint32_t r2 = DT::pack2(x2a * ca + x1a * sa, x2b * cb + x1b * sb);
        uint16_t* dst = isk ? knew : qr + (size_t)hh * D;
        *reinterpret_cast<uint32_t*>(dst + c) = r1;
        *reinterpret_cast<uint32_t*>(dst + HALF + c) = r2;
        if (isk && slot >= 0) {  // the rotated k also goes to the cache (reshape_and_cache)
          uint16_t* kcp = static_cast<uint16_t*>(t.kc) + ((((size_t)slot_blk) * t.Hkv + hk) * t.BS + slot_off) * D;
          *reinterpret_cast<uint32_t*>(kcp + c) = r1;
          *reinterpret_cast<uint32_t*>(kcp + HALF + c) = r2;
        }
      } else if (is_v) {
        const int d0 = (tid - NR) * 2;
        *reinterpret_cast<uint32_t*>(vnew + d0) = va[0];
        if (slot >= 0) {
          uint16_t* vcp = static_cast<uint16_t*>(t.vc) + (((size_t)slot_blk) * t.Hkv + hk) * D * t.BS + slot_off;
          vcp[(size_t)d0 * t.BS] = (uint16_t)(va[0] & 0xffffu);
          vcp[(size_t)(d0 + 1) * t.BS] = (uint16_t)(va[0] >> 16);
        }
      }
    }
  }
  if (has_role && tile0 < ntiles) load_tile(tile0, blk_first);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // qr / knew / vnew staged (LDS only: no wait for the cache stores)
  QA_STAMP(25);

  // ---- this wave's tiles: wave, wave + 16, ...  The running (m, l) stay in registers; the O partial of the wave lives in its
  // LDS slot (where the merge needs it anyway), so that no accumulator is carried across a tile's K / V registers
  const bool row_valid = rq < G;
  const uint16_t* qrow = qr + (size_t)min(rq, G - 1) * D;
  float* my_o = lds_o + (size_t)wave * G * (D + 4);
  float m_run = -INFINITY, l_run = 0.f;
  for (int tile = tile0; tile < ntiles; tile += GS_WAVES) {
    if (tile != tile0) load_tile(tile, t.block_tables[tile_blk_index(tile)]);  // (the first tile of every wave is already in flight)
    const int T0 = tile << 5;
    const bool has_new = last >= T0 && last < T0 + 32;  // wave-uniform: the tile that holds the new token
    if (has_new) {  // its K row comes from LDS (this launch's cache write may not be visible here)
      if (T0 + krow_tok == last) {
#pragma unroll
        for (int j = 0; j < DJ; j++) k0[j] = *reinterpret_cast<const u32x4*>(knew + j * 32 + oct * 8);
      }
      if (T0 + krow_tok + 4 == last) {
#pragma unroll
        for (int j = 0; j < DJ; j++) k1[j] = *reinterpret_cast<const u32x4*>(knew + j * 32 + oct * 8);
      }
    }
    f32x4 s0 = vra_zero_acc(), s1 = vra_zero_acc();
#pragma unroll
    for (int j = 0; j < DJ; j++) {
      u32x4 qv = *reinterpret_cast<const u32x4*>(qrow + j * 32 + oct * 8);
      if (!row_valid) qv = u32x4{0u, 0u, 0u, 0u};
      const s16x8 qf = __builtin_bit_cast(s16x8, qv);
      DT::mfma(s0, __builtin_bit_cast(s16x8, k0[j]), qf);
      DT::mfma(s1, __builtin_bit_cast(s16x8, k1[j]), qf);
    }
    VRA_MFMA_DRAIN();
    float sv[8];
    float tmax = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int tok = T0 + oct * 8 + e;
      float x = (e < 4 ? s0[e] : s1[e - 4]) * t.scale_log2e;
      if (tok >= ctx) x = -INFINITY;
      sv[e] = x;
      tmax = fmaxf(tmax, x);
    }
    tmax = vra_xor16_max(tmax);
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
    u32x4 pa;
    pa[0] = DT::pack2(p[0], p[1]);
    pa[1] = DT::pack2(p[2], p[3]);
    pa[2] = DT::pack2(p[4], p[5]);
    pa[3] = DT::pack2(p[6], p[7]);
    const s16x8 pfrag = __builtin_bit_cast(s16x8, pa);
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
    f32x4 o[DT16];
#pragma unroll
    for (int tt = 0; tt < DT16; tt++) {
      u32x4 vv = vfr[tt];
      if (has_new && new_e >= 0) {
        const uint32_t nv = vnew[tt * 16 + rq];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if ((new_e >> 1) == i) vv[i] = (new_e & 1) ? ((vv[i] & 0x0000ffffu) | (nv << 16)) : ((vv[i] & 0xffff0000u) | nv);
        }
      }
      if (tail) {
#pragma unroll
        for (int i = 0; i < 4; i++) vv[i] &= vm[i];
      }
      DT::mfma0(o[tt], pfrag, __builtin_bit_cast(s16x8, vv));
    }
    VRA_MFMA_DRAIN();
    // O of the wave so far: rows < G, (row oct*4 + r, channel tt*16 + rq); rescaled by alpha of ITS row when a tile came before
    if (tile == tile0) {
#pragma unroll
      for (int tt = 0; tt < DT16; tt++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (oct * 4 + r < G) my_o[(size_t)(oct * 4 + r) * (D + 4) + tt * 16 + rq] = o[tt][r];
    } else {
      float ar[4];
#pragma unroll
      for (int r = 0; r < 4; r++) ar[r] = __shfl(alpha, oct * 4 + r, 64);
#pragma unroll
      for (int tt = 0; tt < DT16; tt++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (oct * 4 + r < G) {
            float* op = my_o + (size_t)(oct * 4 + r) * (D + 4) + tt * 16 + rq;
            *op = *op * ar[r] + o[tt][r];
          }
    }
  }
  QA_STAMP(26);
  l_run = vra_xor16_sum(l_run);
  l_run = vra_xor32_sum(l_run);
  if (oct == 0 && rq < G && tile0 < ntiles) {
    lds_ml[(wave * G + rq) * 2 + 0] = m_run;
    lds_ml[(wave * G + rq) * 2 + 1] = l_run;
  }
  __syncthreads();
  QA_STAMP(27);
  // ---- ONE merge of the 16 wave partials, in wave order
  if (tid < G * D) {  // (G <= 8, D <= 128: one pass of the 1024 threads)
    const int idx = tid;
    const int row = idx / D, d = idx % D;
    // the waves that walked tiles, in tile order: wave (4 + i) % 16 for i < min(ntiles, 16)
    const int nw = min(ntiles, GS_WAVES);
    float Mx = -INFINITY;
    for (int i = 0; i < nw; i++) Mx = fmaxf(Mx, lds_ml[((((4 + i) & 15)) * G + row) * 2]);
    const float Ms = Mx == -INFINITY ? 0.f : Mx;
    float Ls = 0.f, acc = 0.f;
    for (int i = 0; i < nw; i++) {
      const int w = (4 + i) & 15;
      const float f = exp2f(lds_ml[(w * G + row) * 2] - Ms);
      Ls += lds_ml[(w * G + row) * 2 + 1] * f;
      acc += lds_o[((size_t)w * G + row) * (D + 4) + d] * f;
    }
    static_cast<uint16_t*>(t.out)[((size_t)b * t.Hq + hk * G + row) * D + d] = DT::from_f32(Ls > 0.f ? acc / Ls : 0.f);
  }
  QA_STAMP(28);
}

// ---------------------------------------------------------------------------------------------- kernel
template <class DT, bool AWQ, int D>
__global__ __launch_bounds__(GS_THREADS, GS_MIN_WAVES_PER_SIMD) void qkv_attn_decode_kernel(const GemvSArgs a, const QkvAttnTail t) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef VRA_GEMV_TS
  const unsigned long long t_entry = wall_clock64();
#endif
  // the attention job of this workgroup, if any: the owner of unit k_unit0 + hk*uph + s*attn_stride runs (sequence s < B, kv head
  // hk).  attn_stride >= the units of a workgroup and divides uph (launcher), so a workgroup owns at most ONE such unit: the
  // attention is straight-line code (inlined into a job LOOP, every lane-derived address of it was hoisted out and spilled)
  const int wg = (int)blockIdx.x;
  const int u0 = wg * a.units_q + min(wg, a.units_r);
  const int nu = a.units_q + (wg < a.units_r ? 1 : 0);
  int job_b = -1, job_hk = 0;
  for (int ui = 0; ui < nu; ui++) {
    const int rel = u0 + ui - t.k_unit0;
    if (rel < 0 || u0 + ui >= t.v_unit0) continue;
    const int hk = rel / t.uph, r = rel - hk * t.uph;
    if (r % t.attn_stride == 0 && r / t.attn_stride < t.B) job_b = r / t.attn_stride, job_hk = hk;
  }
  // its metadata chain (context length / position / slot, the block id of this wave's first KV tile) starts NOW, ahead of the
  // GEMV: behind the GEMV it cost two dependent round trips before the first K / V load (every workgroup loads: branch-free)
  const int jb = max(job_b, 0);
  const int wave_k = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int tile0_k = (wave_k + 12) & 15;  // this wave's first KV tile (qa_attention)
  const int tile0_blk = min(t.bs_shift >= 0 ? (tile0_k << 5) >> t.bs_shift : (tile0_k << 5) / t.BS, t.max_blocks - 1);
  const uint32_t m_ctx = t.context_lens[jb];
  const int64_t m_pos = t.positions[jb];
  const int64_t m_slot = t.slots[jb];
  const uint32_t m_blk = t.block_tables[(size_t)jb * t.max_blocks + tile0_blk];
  const uint32_t tag = ((uint32_t)__builtin_amdgcn_readfirstlane((int)*t.epoch) << 8) + (uint32_t)t.layer_tag;
  gemv_q4s_body<DT, 1, AWQ, 4, true>(a, smem, tag);
#ifdef VRA_GEMV_TS
  if (a.ts && threadIdx.x == 0) {
    a.ts[(size_t)blockIdx.x * 32 + 31] = t_entry;
    a.ts[(size_t)blockIdx.x * 32 + 30] = wall_clock64();
  }
#endif
  if (job_b < 0) return;
  // (wave-uniform values in SGPRs: the VGPRs are for the K / V tile; positions and slots are < 2^31)
  const int ctx = __builtin_amdgcn_readfirstlane((int)m_ctx), pos32 = __builtin_amdgcn_readfirstlane((int)m_pos);
  const int slot32 = __builtin_amdgcn_readfirstlane((int)m_slot);
  const uint32_t blk_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)m_blk);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the GEMV epilogue is done with the LDS (no wait for its granule stores)
  qa_attention<DT, D>(t, a.gran, a.gran_ld, job_b, job_hk, tag, smem, a.ts, ctx, pos32, slot32, blk_first);
}

// ---------------------------------------------------------------------------------------------- host
static int qa_num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}
int vra_qkv_attn_max_ctx() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VRA_QKV_ATTN_MAX_CTX");
    v = e ? atoi(e) : 2048;
  }
  return v;
}
size_t vra_qkv_attn_granule_bytes(int max_rows, int Hq, int Hkv, int D) { return (size_t)max_rows * (size_t)(Hq + 2 * Hkv) * D / 2 * 8; }

// unit stride between the attention workgroups of one kv head: the smallest divisor of the units per head that is >= the most
// units a workgroup owns (so that no workgroup owns two of them)
static int qa_attn_stride(int uph, int mu) {
  for (int s = mu < 1 ? 1 : mu; s <= uph; s++)
    if (uph % s == 0) return s;
  return uph + 1;
}
bool vra_qkv_attn_fits(int M, int K, int group_size, int n_units, int Hq, int Hkv, int D, int BS, int kv_dtype, int dtype, int max_context_len) {
  if (vra_qkv_attn_max_ctx() <= 0 || max_context_len > vra_qkv_attn_max_ctx()) return false;
  if (M < 1 || M > 4 || (D != 64 && D != 128) || Hkv < 1 || Hq % Hkv || Hq / Hkv > QA_MAX_G || BS % 32) return false;
  if (kv_dtype != dtype || (dtype != VRA_BF16 && dtype != VRA_F16)) return false;  // 16-bit KV cache only (FP8: the two-launch path)
  if (n_units != (Hq + 2 * Hkv) * (D / 16)) return false;
  if (!vra_gemv_s_fits(1, M, K, group_size, n_units, true)) return false;
  int grid, q, r;
  vra_gemv_s_plan(n_units, &grid, &q, &r);
  if (grid > qa_num_cus()) return false;  // every workgroup co-resident: the granule wait relies on it
  const int tpw = (K / 128 + 15) / 16, mu = q + (r ? 1 : 0);
  if (qa_attn_stride(D / 16, mu) * M > D / 16) return false;  // one attention job per workgroup, M of them per kv head
  if (gemv_q4s_lds_bytes(1, tpw, mu, 4) > (size_t)160 * 1024) return false;  // (the K > 16384 row-region variants are not instantiated here)
  return qa_attn_lds_bytes(Hq / Hkv, D) <= (size_t)160 * 1024;
}

template <class DT, bool AWQ, int D>
static void qa_launch(const GemvSArgs& a, const QkvAttnTail& t, int grid, size_t lds, hipStream_t st) {
  static uint64_t attr_devs = 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  auto kern = qkv_attn_decode_kernel<DT, AWQ, D>;
  if (!((attr_devs >> dev) & 1)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_devs |= (uint64_t)1 << dev;
  }
  kern<<<grid, GS_THREADS, lds, st>>>(a, t);
}

void vra_launch_qkv_attn(GemvSArgs a, QkvAttnTail t, void* gran, int group_size, bool awq, int dtype, int D, int64_t stream) {
  hipStream_t st = as_stream(stream);
  const bool grouped = group_size > 0 && group_size < a.K;
  a.gsh = grouped ? 31 - __builtin_clz((unsigned)group_size) : 31;
  a.KT = a.K / 128;
  a.TPW = (a.KT + 15) / 16;
  int grid;
  vra_gemv_s_plan(a.n_units, &grid, &a.units_q, &a.units_r);
  const int mu = a.units_q + (a.units_r ? 1 : 0);
  const int G = t.Hq / t.Hkv;
  size_t lds = gemv_q4s_lds_bytes(1, a.TPW, mu, 4);
  if (qa_attn_lds_bytes(G, D) > lds) lds = qa_attn_lds_bytes(G, D);
  a.dbg = 0;
#ifdef VRA_GEMV_TS
  a.ts = vra_gemv_ts_buf_shared();
#else
  a.ts = nullptr;
#endif
  a.gran = gran;
  a.gran_ld = (t.Hq + 2 * t.Hkv) * D / 2;
  t.uph = D / 16;
  t.k_unit0 = t.Hq * t.uph;
  t.v_unit0 = (t.Hq + t.Hkv) * t.uph;
  t.attn_stride = qa_attn_stride(t.uph, mu);
  t.bs_shift = (t.BS & (t.BS - 1)) == 0 ? 31 - __builtin_clz((unsigned)t.BS) : -1;
  t.err = vra_scratch_error_word();
  if (!t.err) {
    vra_set_error("qkv_attn: scratch not initialised (vra_scratch_init)");
    return;
  }
  const bool bf = dtype == VRA_BF16;
  if (D == 128) {
    if (awq) bf ? qa_launch<BF16, true, 128>(a, t, grid, lds, st) : qa_launch<F16, true, 128>(a, t, grid, lds, st);
    else bf ? qa_launch<BF16, false, 128>(a, t, grid, lds, st) : qa_launch<F16, false, 128>(a, t, grid, lds, st);
  } else {
    if (awq) bf ? qa_launch<BF16, true, 64>(a, t, grid, lds, st) : qa_launch<F16, true, 64>(a, t, grid, lds, st);
    else bf ? qa_launch<BF16, false, 64>(a, t, grid, lds, st) : qa_launch<F16, false, 64>(a, t, grid, lds, st);
  }
}
