// decode_step.hip — "kernel P": the persistent decode step for 1..2 sequences (see decode_step.h).
//
// Roofline: HBM.  Algorithmic bytes per launch = the int4 GEMV family of all layers (3 625 975 808 B for
// Llama-3-8B g128) + KV.  What the kernel is built around (profiles/r02_timeline_kernel_e.txt): a decode GEMV as its own
// launch spends 4-6 us outside its streaming phase — launch ramp, x round trip, norm, FIRST weight round trip, reduction,
// drain — and HBM idles through all of it.  Here the weight stream never stops at a dependency edge:
//   * wave 0 (LOADER) walks this workgroup's slots — (layer, GEMV, unit, 16 k-tiles, stream) in program order — and
//     copies each (16 KiB of tiles + their scales [+ AWQ zeros]) into the next position of an LDS ring with
//     `global_load_lds_dwordx4 ... nt` (no VGPRs, weights read once => non-temporal).  2-3 slots in flight, counted with
//     hand-placed `s_waitcnt vmcnt(N)`; a landed slot is published through an LDS word, a consumed one returned through
//     an LDS counter.  The loader runs ahead across phase boundaries until the ring is full (7 slots = 112 KiB per CU,
//     28 MiB on the chip = ~4.5 us of HBM time).
//   * waves 1..8 (CONSUMERS) run the phases.  GEMV arithmetic is kernel E's, bit for bit: consumer c plays E's waves c
//     and c+8 (k-tiles w, w+16, ... of every unit), partial tiles meet in LDS in wave order, same fused epilogue.
//   * phases are separated by a grid barrier: outputs are stored write-through (sc1), the last storing wave of a
//     workgroup adds 1 to its shard (workgroup % 8) of a device-scope counter, consumer 0 polls the 8 shards (relaxed,
//     s_sleep), x is read past the L1 (sc1).  Counters are monotonic (epoch = phases completed so far, kept in device
//     memory): nothing to reset, replayable from a hipGraph.  Every wait is bounded; a timeout raises the error word and
//     aborts all later waits of the launch.
//   * attention (RoPE + KV write + paged attention, one (sequence, kv head) per workgroup, tiles spread over the 8
//     consumers) runs while the other CUs' rings fill with o_proj / gate / up weights — the HBM-idle window of the
//     launch-per-op decode layer.
#include "decode_step.h"

#include <stdlib.h>

#include <type_traits>

#include "kvcache.cuh"
#include "wna16.cuh"

#define DP_CTL_LANDED 0   // slots landed in the ring so far (written by the loader; slots land in order)
#define DP_CTL_CONS 4     // [8]  slots consumer c has finished (single writer each)
#define DP_CTL_CBAR 16    // consumer barrier counter
#define DP_CTL_GRID 17    // phases released by the grid barrier (written by consumer 0)
#define DP_CTL_ABORT 18
#define DP_CTL_ARRIVE 19  // storing waves that have drained this phase
#define DP_CTL_PART 32    // float [16][4]: partial sums of x^2 per (E wave, row)
#define DP_SPIN_LIMIT (1u << 21)

#ifdef VRA_GEMV_TS
#define DP_STAMP(i)                                                                                                   \
  do {                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    if (a.ts && c == 0 && lane == 0) {                                                                                \
      a.ts[((size_t)blockIdx.x * 256 + (size_t)(ph - a.ph0)) * 8 + (i)] = wall_clock64();                             \
      if ((i) == 2 || (i) == 3) a.ts[((size_t)blockIdx.x * 256 + (size_t)(ph - a.ph0)) * 8 + 4 + (i)] = clock64();    \
    }                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  } while (0)
// per-slot stamps of one phase (a.ts_phase): 0 issued by the loader, 1 published, 2 consumer 0 starts its reads, 3 consumer 0 done
#define DP_SLOT_STAMP(phase, k, i)                                                                                      \
  do {                                                                                                                  \
    if (a.ts && (phase) == a.ts_phase && (k) < 64 && lane == 0)                                                          \
      a.ts[(size_t)256 * 256 * 8 + ((size_t)blockIdx.x * 64 + (k)) * 4 + (i)] = wall_clock64();                          \
  } while (0)
// shader-clock stamps inside the main loop of one workgroup (100), phase a.ts_phase: [consumer][step][8]
#define DP_CYC(k, i)                                                                                                     \
  do {                                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
    if (a.ts && ph == a.ts_phase && blockIdx.x == 100 && (k) < 16 && lane == 0)                                           \
      a.ts[(size_t)256 * 256 * 8 + 256 * 64 * 4 + ((size_t)c * 16 + (k)) * 8 + (i)] = clock64();                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                   \
  } while (0)
#else
#define DP_STAMP(i) do {} while (0)
#define DP_SLOT_STAMP(phase, k, i) do {} while (0)
#define DP_CYC(k, i) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------- LDS / sync primitives
__device__ __forceinline__ uint32_t dp_lds_ld(const uint32_t* p) {
  return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void dp_lds_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// one more round of a bounded spin; true = give up (timeout, or another wave already gave up)
__device__ __forceinline__ bool dp_give_up(uint32_t* ctl, uint32_t* err, uint32_t& n, uint32_t code) {
  __builtin_amdgcn_s_sleep(1);
  if (dp_lds_ld(ctl + DP_CTL_ABORT)) return true;
  if (++n > DP_SPIN_LIMIT) {
    dp_lds_st(ctl + DP_CTL_ABORT, 1u);
    __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  return false;
}
// LDS operations of one wave execute in issue order, so "write data, then bump a word" / "see the word, then read data"
// needs only compiler ordering
__device__ __forceinline__ void dp_wait_ge(uint32_t* ctl, uint32_t* err, int idx, uint32_t want, uint32_t code) {
  uint32_t n = 0;
  while ((int32_t)(dp_lds_ld(ctl + idx) - want) < 0)
    if (dp_give_up(ctl, err, n, code)) break;
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void dp_cbar(uint32_t* ctl, uint32_t* err, uint32_t& tgt, int lane) {
  tgt += DP_NC;
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_fetch_add(ctl + DP_CTL_CBAR, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  dp_wait_ge(ctl, err, DP_CTL_CBAR, tgt, 0x10u);
}
__device__ __forceinline__ int dp_shard_count(int grid, int s) { return (grid - s + 7) >> 3; }  // workgroups b with b % 8 == s

// a storing wave has issued its last write-through store of the phase: drain, and let the workgroup's last wave arrive
__device__ __forceinline__ void dp_arrive(const DPStepArgs& a, uint32_t* ctl, uint32_t& arrive_tgt, int lane) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  arrive_tgt += DP_NC;
  if (lane == 0) {
    const uint32_t old = __hip_atomic_fetch_add(ctl + DP_CTL_ARRIVE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (old + 1u == arrive_tgt)
      __hip_atomic_fetch_add(a.counters + (blockIdx.x & 7) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// wait until `done` phases (counted from the first launch ever) are complete on every workgroup
__device__ __forceinline__ void dp_grid_wait(const DPStepArgs& a, uint32_t* ctl, uint32_t done, int c, int lane) {
  if (c == 0) {
    const int grid = (int)gridDim.x;
    const uint32_t want = done * (uint32_t)dp_shard_count(grid, lane & 7);
    uint32_t n = 0;
    for (;;) {
      uint32_t v = want;
      if (lane < 8) v = __hip_atomic_load(a.counters + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all((int32_t)(v - want) >= 0)) break;
      if (dp_give_up(ctl, a.err, n, 0x20u)) break;
    }
    asm volatile("" ::: "memory");
    if (lane == 0) dp_lds_st(ctl + DP_CTL_GRID, done);
  } else {
    dp_wait_ge(ctl, a.err, DP_CTL_GRID, done, 0x21u);
  }
}

// the descriptor table is read-only for the whole launch: through the constant address space its (wave-uniform) reads are
// scalar loads, whatever the stores and asm statements around them
typedef const DPGemv __attribute__((address_space(4))) DPGemvC;
typedef const DPLayer __attribute__((address_space(4))) DPLayerC;
__device__ __forceinline__ DPLayerC* dp_layers(const DPStepArgs& a) { return (DPLayerC*)(uintptr_t)a.layers; }

// ---------------------------------------------------------------------------------------------- loader (wave 0)
// 16 bytes per lane from `gsrc` (per lane) to LDS byte address `lds_dst` + lane * 16 (wave-uniform base in M0).  The
// compiler neither counts nor waits for this load: the loader counts its own vmcnt.
__device__ __forceinline__ void dp_glds16(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// slots consumed by ALL consumers so far
__device__ __forceinline__ uint32_t dp_consumed(const uint32_t* ctl, int lane) {
  const uint32_t v = __hip_atomic_load(ctl + DP_CTL_CONS + (lane & (DP_NC - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  uint32_t m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
  for (int i = 1; i < DP_NC; i++) {
    const uint32_t x = __builtin_amdgcn_readlane(v, i);
    m = (int32_t)(x - m) < 0 ? x : m;
  }
  return m;
}
template <bool AWQ>
__device__ void dp_loader(const DPStepArgs& a, unsigned char* smem) {
  constexpr int LPS = AWQ ? 18 : 17;  // DMA instructions per slot
  const int lane = threadIdx.x & 63;
  uint32_t* ctl = reinterpret_cast<uint32_t*>(smem);
  const uint32_t ring = (uint32_t)(uintptr_t)(smem + a.ring_off);
  const int wg = (int)blockIdx.x, grid = (int)gridDim.x, nslot = a.nslot;
  uint32_t islot = 0, pub = 0, freed = 0;  // slots issued / published / known to be consumed by every consumer
  int pending = 0;
  for (int ph = a.ph0; ph < a.ph1; ph++) {
    const int l = ph / DP_PHASES_PER_LAYER, kind = ph % DP_PHASES_PER_LAYER;
    if (kind == 1) continue;  // attention streams no weights
    const DPGemvC& g = dp_layers(a)[l].g[kind == 0 ? 0 : kind - 1];
    const int rank = (wg + g.rot) % grid;
    const int nu = g.units_q + (rank < g.units_r ? 1 : 0);
    const int u0 = rank * g.units_q + min(rank, g.units_r);
    const int KT = g.KT, TPW = g.TPW, NS = g.NS, G = g.G, gsh = g.gsh;
    const uint32_t ph_slot0 = islot;
    (void)ph_slot0;
    for (int ui = 0; ui < nu; ui++) {
      const int unit = u0 + ui;
      for (int ti = 0; ti < TPW; ti++) {
        for (int b = 0; b < NS; b++) {
          const uint32_t pos = islot % (uint32_t)nslot;
          if (!(a.dbg & 1) && (int32_t)(freed + (uint32_t)nslot - islot) <= 0) {  // the position's previous slot may still be in use
            freed = dp_consumed(ctl, lane);
            if ((int32_t)(freed + (uint32_t)nslot - islot) <= 0) {
              // ring full: everything issued so far must be visible before this wave sleeps (the consumers may be waiting for it)
              if (pending) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                pub += (uint32_t)pending, pending = 0;
                if (lane == 0) dp_lds_st(ctl + DP_CTL_LANDED, pub);
              }
              uint32_t n = 0;
              while ((int32_t)(freed + (uint32_t)nslot - islot) <= 0) {
                if (dp_give_up(ctl, a.err, n, 0x30u)) break;
                freed = dp_consumed(ctl, lane);
              }
            }
          }
          const uint32_t dst = ring + pos * (uint32_t)DP_SLOT_BYTES;
          const unsigned char* wsrc = static_cast<const unsigned char*>(b ? g.w[1] : g.w[0]);
          const size_t ubase = (size_t)unit * KT;
          if (!(a.dbg & 4)) {
#pragma unroll
          for (int j = 0; j < 16; j++) {
            const int kt = min(16 * ti + j, KT - 1);
            dp_glds16(wsrc + ((ubase + kt) << 10) + lane * 16, dst + j * 1024);
          }
          }
          if (!(a.dbg & 4)) {  // scales of the groups the 16 tiles touch: 512 bytes from group (16*ti*128) >> gsh on (clamped into the stream)
            const int g0 = (16 * ti * 128) >> gsh;
            const int off = min((int)(((size_t)unit * G + g0) * 32) + lane * 16, g.sc_bytes - 16);
            const unsigned char* ssrc = static_cast<const unsigned char*>(b ? g.sc[1] : g.sc[0]) + off;
            if (lane < 32) dp_glds16(ssrc, dst + DP_SLOT_W);
            if (AWQ) {
              const int zoff = min((int)(((size_t)unit * G + g0) * 8) + lane * 16, g.zr_bytes - 16);
              const unsigned char* zsrc = reinterpret_cast<const unsigned char*>(b ? g.zr[1] : g.zr[0]) + zoff;
              if (lane < 8) dp_glds16(zsrc, dst + DP_SLOT_W + DP_SLOT_S);
            }
          }
          DP_SLOT_STAMP(ph, (int)(islot - ph_slot0), 0);
          islot++;
          if (++pending == 3) {  // two slots stay in flight; the oldest has landed
            if (a.dbg & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (AWQ) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(34)" ::: "memory");
            pub++, pending--;
            if (lane == 0) dp_lds_st(ctl + DP_CTL_LANDED, pub);
            DP_SLOT_STAMP(ph, (int)(pub - 1u - ph_slot0), 1);
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  pub += (uint32_t)pending;
  if (lane == 0) dp_lds_st(ctl + DP_CTL_LANDED, pub);
  (void)LPS;
}

// ---------------------------------------------------------------------------------------------- consumers: GEMV phase
struct DPState {
  uint32_t cslot;       // next slot of this workgroup's stream
  uint32_t landed;      // slots known to have landed (cached copy of the loader's count)
  uint32_t cbar_tgt;    // consumer-barrier target
  uint32_t arrive_tgt;  // storing-wave arrivals
};

template <class DT, bool AWQ>
__device__ __forceinline__ void dp_gemv(const DPStepArgs& a, DPGemvC& g, unsigned char* smem, DPState& st, uint32_t grid_done,
                                        bool wait_grid, int c, int lane, int ph) {
  constexpr int EW = 16 / DP_NC;  // E waves played by one consumer
  // everything below is derived from these two: opaque per phase, so that nothing of a phase's address arithmetic is hoisted
  // out of the phase loop (the hoisted values of all phases together spilled)
  asm volatile("" : "+v"(lane));
  asm volatile("" : "+s"(c));
  uint32_t* ctl = reinterpret_cast<uint32_t*>(smem);
  float* part = reinterpret_cast<float*>(smem) + DP_CTL_PART;
  unsigned char* const ringp = smem + a.ring_off;
  unsigned char* const xreg = smem + a.x_off;
  float* const red = reinterpret_cast<float*>(smem + a.red_off);
  const int nn = lane & 15, oct = lane >> 4;
  const int M = a.M, KT = g.KT, TPW = g.TPW, NS = g.NS, gsh = g.gsh, XT = a.xt;
  const int wg = (int)blockIdx.x, grid = (int)gridDim.x;
  const int rank = (wg + g.rot) % grid;
  const int nu = g.units_q + (rank < g.units_r ? 1 : 0);
  const int u0 = rank * g.units_q + min(rank, g.units_r);
  const bool norm = g.norm_w != nullptr;

  // ---- operands that do not depend on the previous phase, requested before the grid barrier is awaited:
  // epilogue operands of the first unit this wave will finish (unit c; thread = (row oct, column nn)), norm weights of its tiles
  const int e_m = oct, e_nl = nn;
  struct EpiOps {
    void* out;
    int ld, col;
    bool has_bias, has_bias2;
    uint32_t bias_w, bias2_w, res_w;
    size_t res_idx;
  };
  auto epi_ops = [&](int ue) {  // ue < nu
    EpiOps o;
    const int e_unit = u0 + ue;
    const bool e_s1 = g.nseg > 1 && e_unit >= g.unit_start[1], e_s2 = g.nseg > 2 && e_unit >= g.unit_start[2];
    o.out = e_s2 ? g.out[2] : (e_s1 ? g.out[1] : g.out[0]);
    const void* const biasp = e_s2 ? g.bias[2] : (e_s1 ? g.bias[1] : g.bias[0]);
    o.ld = e_s2 ? g.out_ld[2] : (e_s1 ? g.out_ld[1] : g.out_ld[0]);
    o.col = (e_unit - (e_s2 ? g.unit_start[2] : (e_s1 ? g.unit_start[1] : 0))) * 16 + e_nl;
    const void* const bias2p = NS == 2 ? g.bias[1] : nullptr;  // pair: bias[0] gate, bias[1] up (nseg = 1)
    o.has_bias = biasp != nullptr, o.has_bias2 = bias2p != nullptr;
    o.bias_w = o.bias2_w = o.res_w = 0u;
    o.res_idx = (size_t)e_m * g.res_ld + o.col;
    if (e_m < M) {
      // (the 32-bit word holding the value: a 16-bit load gets its zero extension — i.e. a wait — right behind the issue)
      if (biasp) o.bias_w = static_cast<const uint32_t*>(biasp)[o.col >> 1];
      if (bias2p) o.bias2_w = static_cast<const uint32_t*>(bias2p)[o.col >> 1];
      // the residual stream was written by other CUs of THIS launch (at least one grid barrier ago): read past the L1
      if (g.residual)
        o.res_w = __hip_atomic_load(static_cast<const uint32_t*>(g.residual) + (o.res_idx >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return o;
  };
  EpiOps e0 = {};
  if (c < nu) e0 = epi_ops(c);
  // staging lanes: the 4 rows of 16 lanes take (played wave, x row) pairs — M = 1: four played waves at once, M = 2: two
  // waves x two rows per pass — so no lane repeats another's work (kernel E's rows >= M alias row M-1: 3/4 of its staging
  // lanes at one row).  Per played wave the arithmetic and its order are kernel E's.
  const int erow = M == 1 ? 0 : (oct & 1);
  const int er = M == 1 ? oct : (oct >> 1);
  const int epp = M == 1 ? 4 : 2;  // played waves per pass
  constexpr int NPASS = EW >= 4 ? 2 : 1;  // passes at most (EW * M / 4, at least 1)
  const int npass = (EW + epp - 1) / epp;
  // the played wave of this lane in pass p: index p*epp + er (lanes past EW idle: they repeat the last played wave, stores masked)
  auto played = [&](int p, bool* act) {
    const int pi = p * epp + er;
    *act = pi < EW;
    return c + DP_NC * min(pi, EW - 1);
  };
  u32x4 nr[NPASS][4];
  if (norm) {
    const uint16_t* nwp = static_cast<const uint16_t*>(g.norm_w) + nn * 8;
#pragma unroll
    for (int p = 0; p < NPASS; p++)
      if (p < npass) {
        bool act;
        const int w = played(p, &act);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int kt = min(w + 16 * min(t, TPW - 1), KT - 1);
          nr[p][t] = *reinterpret_cast<const u32x4*>(nwp + (size_t)kt * 128);
        }
      }
  }
  DP_STAMP(0);
  if (wait_grid) dp_grid_wait(a, ctl, grid_done, c, lane);
  DP_STAMP(1);

  // ---- x slices of the E waves this consumer plays
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.x), 0, 0x7FFFFFF0, 0x00020000);
  const uint32_t xlane = (uint32_t)(((size_t)erow * g.x_ld + nn * 8) * 2);
  if (norm) {  // (TPW <= 4)
    u32x4 xv[NPASS][4];
#pragma unroll
    for (int p = 0; p < NPASS; p++)
      if (p < npass) {
        bool act;
        const int w = played(p, &act);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const int kt = min(w + 16 * min(t, TPW - 1), KT - 1);
          xv[p][t] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xlane + (uint32_t)kt * 256u, 0, 16);  // sc1
        }
      }
    // Σx² of the rows: partial sums per played wave (a wave without a k-tile adds zeros, as in kernel E), total in wave order
#pragma unroll
    for (int p = 0; p < NPASS; p++)
      if (p < npass) {
        bool act;
        const int w = played(p, &act);
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if (t < TPW) {
            if (w + 16 * t >= KT) xv[p][t] = u32x4{0u, 0u, 0u, 0u};
            float f[8];
            unpack8<DT>(xv[p][t], f);
#pragma unroll
            for (int e = 0; e < 8; e++) ss += f[e] * f[e];
          }
        }
        const float rsum = row16_sum(ss);
        if (nn == 0 && act) part[w * 4 + erow] = rsum;
      }
    dp_cbar(ctl, a.err, st.cbar_tgt, lane);
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; w++) tot += part[w * 4 + erow];
    const float rstd = 1.0f / sqrtf(tot / (float)g.K + a.eps);
#pragma unroll
    for (int p = 0; p < NPASS; p++)
      if (p < npass) {
        bool act;
        const int w = played(p, &act);
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if (t < TPW) {
            const bool valid = act && w + 16 * t < KT;
            unsigned char* tp = xreg + (size_t)min(w + 16 * t, KT - 1) * XT;
            float f[8], gw[8];
            unpack8<DT>(xv[p][t], f);
            unpack8<DT>(nr[p][t], gw);
#pragma unroll
            for (int e = 0; e < 8; e++) f[e] = f[e] * rstd * gw[e];
            const u32x4 v = pack8<DT>(f);
            if (valid) *reinterpret_cast<u32x4*>(tp + erow * 272 + nn * 16) = v;
            const float s8 = row16_sum(octet_sum<DT>(v));  // over the ROUNDED values the MFMA will see
            if (nn == 0 && valid) reinterpret_cast<float*>(tp + M * 272)[erow] = s8;
          }
        }
      }
    // (the partial table is written again only after the next phase's grid barrier)
  } else {
#pragma unroll
    for (int p = 0; p < NPASS; p++)
      if (p < npass) {
        bool act;
        const int w = played(p, &act);
        for (int t0 = 0; t0 < TPW; t0 += 4) {
          u32x4 xv[4];
#pragma unroll
          for (int t = 0; t < 4; t++) {
            const int kt = min(w + 16 * min(t0 + t, TPW - 1), KT - 1);
            xv[t] = __builtin_amdgcn_raw_buffer_load_b128(xrs, xlane + (uint32_t)kt * 256u, 0, 16);  // sc1
          }
#pragma unroll
          for (int t = 0; t < 4; t++) {
            if (t0 + t < TPW) {
              const bool valid = act && w + 16 * (t0 + t) < KT;
              unsigned char* tp = xreg + (size_t)min(w + 16 * (t0 + t), KT - 1) * XT;
              if (valid) *reinterpret_cast<u32x4*>(tp + erow * 272 + nn * 16) = xv[t];
              const float s8 = row16_sum(octet_sum<DT>(xv[t]));
              if (nn == 0 && valid) reinterpret_cast<float*>(tp + M * 272)[erow] = s8;
            }
          }
        }
      }
  }
  // (no barrier: a consumer reads only the slices of the waves it plays, which its own lanes staged)
  asm volatile("" ::: "memory");
  DP_STAMP(2);

  // ---- main loop: the slots of this workgroup's units, in the loader's order.  What bounds a consumer is instruction
  // issue (two waves per SIMD; measured ~1800 cycles per slot with a naive loop): everything that does not change from slot to
  // slot is a lane constant computed once per phase, a tile past KT re-reads the last tile with its scale zeroed (branch-free,
  // as kernel E), and the two tiles a consumer takes from a slot are independent MFMA chains, interleaved.
  // (Built and measured slower: a two-stage register pipeline over slots, and x fragments resident in registers for the
  // whole phase — both need more than the 168 VGPRs nine waves leave each wave and spill.)
  const int zsh = 4 * awq_rev(nn & 7);
  constexpr float CB = Magic<DT>::bias;
  const int arow = min(nn, M - 1);
  uint32_t cslot = __builtin_amdgcn_readfirstlane(st.cslot), landed = __builtin_amdgcn_readfirstlane(st.landed);
  uint32_t cpos = __builtin_amdgcn_readfirstlane(cslot % (uint32_t)a.nslot);  // ring position of slot `cslot`
  const uint32_t nslot_u = (uint32_t)a.nslot;
  // lane constants of the two played waves (i = 0, 1: E waves c, c + 8)
  uint32_t tile_l[EW], sc_l[EW], zr_l[EW];
#pragma unroll
  for (int i = 0; i < EW; i++) {
    const int w = c + DP_NC * i;
    const int gl = (w * 128) >> gsh;  // group of tile w + 16*ti relative to the slot's first group (the same for every ti)
    tile_l[i] = (uint32_t)(w * 1024 + lane * 16);
    sc_l[i] = (uint32_t)(DP_SLOT_W + gl * 32 + nn * 2);
    zr_l[i] = (uint32_t)(DP_SLOT_W + DP_SLOT_S + gl * 8 + (nn >> 3) * 4);
  }
  auto wait_landed = [&](uint32_t upto) {  // slots < upto have landed
    if ((a.dbg & 2) || (int32_t)(landed - upto) >= 0) return;
    uint32_t n = 0;
    for (;;) {
      landed = dp_lds_ld(ctl + DP_CTL_LANDED);
      if ((int32_t)(landed - upto) >= 0 || dp_give_up(ctl, a.err, n, 0x40u)) break;
    }
    asm volatile("" ::: "memory");
  };
  auto park = [&](int ui, const f32x4 (&acc)[EW][2]) {  // end of a unit: partial tiles of rows 0..M-1 (they live in lanes 0..15)
    if (oct == 0) {
#pragma unroll
      for (int i = 0; i < EW; i++)
#pragma unroll
        for (int b = 0; b < 2; b++)
          if (b < NS) {
            float* rp = red + ((size_t)((ui * NS + b) * 16 + c + DP_NC * i) * M) * 16 + nn;
            rp[0] = acc[i][b][0];
            if (M > 1) rp[16] = acc[i][b][1];
          }
    }
  };
  // One k-step of a unit at a time: its NS slots (a gate/up pair shares the x fragments) are read as one batch — the LDS
  // round trip is paid once per step and the partner wave on the SIMD runs under it — then 8 * NS MFMAs, one drain, the fix-ups.
  for (int ui = 0; ui < nu; ui++) {
    f32x4 acc[EW][2];
#pragma unroll
    for (int i = 0; i < EW; i++) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ti = 0; ti < TPW; ti++) {
      DP_CYC(ui * TPW + ti, 0);
      wait_landed(cslot + (uint32_t)NS);
      DP_CYC(ui * TPW + ti, 1);
      const unsigned char* slot0 = ringp + (size_t)cpos * DP_SLOT_BYTES;
      cpos = cpos + 1u == nslot_u ? 0u : cpos + 1u;
      const unsigned char* slot1 = ringp + (size_t)cpos * DP_SLOT_BYTES;
      if (NS == 2) cpos = cpos + 1u == nslot_u ? 0u : cpos + 1u;
      u32x4 wt[2][EW], xf[4][EW];
      f32x4 sx[EW];
      uint32_t sraw[2][EW], zraw[2][EW];
      bool valid[EW];
#pragma unroll
      for (int i = 0; i < EW; i++) {
        wt[0][i] = *reinterpret_cast<const u32x4*>(slot0 + tile_l[i]);
        sraw[0][i] = *reinterpret_cast<const uint16_t*>(slot0 + sc_l[i]);
        zraw[0][i] = AWQ ? *reinterpret_cast<const uint32_t*>(slot0 + zr_l[i]) : 0u;
        if (NS == 2) {
          wt[1][i] = *reinterpret_cast<const u32x4*>(slot1 + tile_l[i]);
          sraw[1][i] = *reinterpret_cast<const uint16_t*>(slot1 + sc_l[i]);
          zraw[1][i] = AWQ ? *reinterpret_cast<const uint32_t*>(slot1 + zr_l[i]) : 0u;
        } else {
          wt[1][i] = wt[0][i], sraw[1][i] = 0u, zraw[1][i] = 0u;
        }
        const int kt = c + DP_NC * i + 16 * ti;
        valid[i] = kt < KT;
        // a played wave without this k-tile multiplies the all-zero tile (its own tiles are the only ones it may read without a
        // barrier: another consumer's slices may not be staged yet)
        const unsigned char* xt = valid[i] ? xreg + (size_t)kt * XT : smem + a.zero_off;
        const unsigned char* xp = xt + arow * 272 + oct * 16;
#pragma unroll
        for (int j = 0; j < 4; j++) xf[j][i] = *reinterpret_cast<const u32x4*>(xp + j * 64);
        sx[i] = *reinterpret_cast<const f32x4*>(xt + M * 272);
      }
      // the step's slots go back to the loader right behind their reads: the LDS executes a wave's operations in order, so by
      // the time the loader can see this count the reads have taken their data (the MFMAs run out of registers)
      cslot += (uint32_t)NS;
      if (lane == 0) dp_lds_st(ctl + DP_CTL_CONS + c, cslot);
      __builtin_amdgcn_sched_barrier(0);  // every read of the step is requested before its first MFMA
      DP_CYC(ui * TPW + ti, 2);
      f32x4 ag[2][EW];
      if (NS == 2) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < EW; i++) {
              if (j == 0) DT::mfma0(ag[b][i], __builtin_bit_cast(s16x8, xf[j][i]), magic_word<DT>(wt[b][i][j]));
              else DT::mfma(ag[b][i], __builtin_bit_cast(s16x8, xf[j][i]), magic_word<DT>(wt[b][i][j]));
            }
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int i = 0; i < EW; i++) {
            if (j == 0) DT::mfma0(ag[0][i], __builtin_bit_cast(s16x8, xf[j][i]), magic_word<DT>(wt[0][i][j]));
            else DT::mfma(ag[0][i], __builtin_bit_cast(s16x8, xf[j][i]), magic_word<DT>(wt[0][i][j]));
          }
#pragma unroll
        for (int i = 0; i < EW; i++) ag[1][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      VRA_MFMA_DRAIN();
      DP_CYC(ui * TPW + ti, 3);
#pragma unroll
      for (int b = 0; b < 2; b++) {
        if (b < NS) {
#pragma unroll
          for (int i = 0; i < EW; i++) {
            const float sv = valid[i] ? DT::to_f32((uint16_t)sraw[b][i]) : 0.f;
            const float zc = AWQ ? CB + (float)((zraw[b][i] >> zsh) & 0xFu) : CB + 8.f;
#pragma unroll
            for (int e = 0; e < 4; e++) acc[i][b][e] = fmaf(sv, fmaf(-zc, sx[i][e], ag[b][i][e]), acc[i][b][e]);
          }
        }
      }
      DP_CYC(ui * TPW + ti, 4);
    }
    park(ui, acc);
  }
  st.cslot = cslot, st.landed = landed;
  DP_STAMP(3);
  dp_cbar(ctl, a.err, st.cbar_tgt, lane);
  DP_STAMP(4);

  // ---- epilogue: consumer c finishes units c, c+NC, ... (16 wave partials in wave order, then kernel E's fused epilogue)
  for (int ue = c; ue < nu; ue += DP_NC) {
    const EpiOps o = ue == c ? e0 : epi_ops(ue);
    uint32_t vbits = 0u;
    if (e_m < M) {
      float v = 0.f, v2 = 0.f;
#pragma unroll
      for (int w = 0; w < 16; w++) {
        v += red[((size_t)((ue * NS + 0) * 16 + w) * M + e_m) * 16 + e_nl];
        if (NS == 2) v2 += red[((size_t)((ue * NS + 1) * 16 + w) * M + e_m) * 16 + e_nl];
      }
      const float e_bias = DT::to_f32((uint16_t)((o.col & 1) ? o.bias_w >> 16 : o.bias_w));
      const float e_bias2 = DT::to_f32((uint16_t)((o.col & 1) ? o.bias2_w >> 16 : o.bias2_w));
      const float e_res = DT::to_f32((uint16_t)((o.res_idx & 1) ? o.res_w >> 16 : o.res_w));
      v = rnd_dt<DT>(v);
      if (o.has_bias) v = rnd_dt<DT>(v + e_bias);
      if (NS == 2) {
        v2 = rnd_dt<DT>(v2);
        if (o.has_bias2) v2 = rnd_dt<DT>(v2 + e_bias2);
        const float sl = rnd_dt<DT>(v / (1.0f + expf(-v)));
        v = sl * v2;
      }
      if (g.residual) v = rnd_dt<DT>(v) + e_res;
      vbits = DT::from_f32(v);
    }
    // 8 columns per store: lanes nn = 0 and 8 collect their 7 right neighbours (DPP row_shl) and issue ONE 16-byte
    // write-through store (a 2-byte sc1 store is a fabric write of its own: 12x the time per byte)
    uint32_t nb[8];
    nb[0] = vbits;
    nb[1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x101, 0xF, 0xF, true);
    nb[2] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x102, 0xF, 0xF, true);
    nb[3] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x103, 0xF, 0xF, true);
    nb[4] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x104, 0xF, 0xF, true);
    nb[5] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x105, 0xF, 0xF, true);
    nb[6] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x106, 0xF, 0xF, true);
    nb[7] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x107, 0xF, 0xF, true);
    if (e_m < M && (e_nl & 7) == 0) {
      const u32x4 pk = {nb[0] | (nb[1] << 16), nb[2] | (nb[3] << 16), nb[4] | (nb[5] << 16), nb[6] | (nb[7] << 16)};
      const __amdgpu_buffer_rsrc_t ors = __builtin_amdgcn_make_buffer_rsrc(o.out, 0, 0x7FFFFFF0, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(pk, ors, (uint32_t)(((size_t)e_m * o.ld + o.col) * 2), 0, 16);  // sc1
    }
  }
  dp_arrive(a, ctl, st.arrive_tgt, lane);
  DP_STAMP(5);
}

// ---------------------------------------------------------------------------------------------- consumers: attention phase
// RoPE(q, k) + KV-cache write + paged attention of one (sequence, kv head) per workgroup (attention.rs:745-820 in one
// phase; decode_attn_fused_kernel's arithmetic with the 32-token tiles dealt round-robin to 8 waves instead of in runs to 4).
template <class DT, int D, bool KV8>
__device__ __forceinline__ void dp_attn(const DPStepArgs& a, DPLayerC& L, unsigned char* smem, DPState& st, uint32_t grid_done,
                                        bool wait_grid, int c, int lane, int ph) {
  typedef typename KVT<KV8>::elem kv_t;
  constexpr int DJ = D / 32, DT16 = D / 16, HALF = D / 2;
  asm volatile("" : "+v"(lane));
  asm volatile("" : "+s"(c));
  uint32_t* ctl = reinterpret_cast<uint32_t*>(smem);
  const int wg = (int)blockIdx.x;
  const int G = a.Hq / a.Hkv;
  const bool active = wg < a.M * a.Hkv;
  DP_STAMP(0);
  if (wait_grid) dp_grid_wait(a, ctl, grid_done, c, lane);
  DP_STAMP(1);
  if (active) {
    unsigned char* xreg = smem + a.x_off;
    float* lds_o = reinterpret_cast<float*>(xreg);                                   // [4][G][D + 4]
    float* lds_ml = lds_o + (size_t)4 * G * (D + 4);                                 // [4][G][2]
    kv_t* knew = reinterpret_cast<kv_t*>(lds_ml + (size_t)4 * G * 2);                // [D] the new token's K row in CACHE format
    uint16_t* vnew = reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(knew) + 256);  // [D]
    const int rq = lane & 15, oct = lane >> 4;
    const int krow_tok = (rq >> 2) * 8 + (rq & 3);
    const int b = wg / a.Hkv, hk = wg % a.Hkv;
    const int ctx = (int)a.context_lens[b];
    const int64_t pos = a.positions[b];
    const int64_t slot = a.slots[b];
    const int slot32 = (int)slot;
    const int slot_blk = a.bs_shift >= 0 ? slot32 >> a.bs_shift : slot32 / a.BS;
    const int slot_off = slot32 - slot_blk * a.BS;
    const uint16_t* cosp = static_cast<const uint16_t*>(a.cosv) + pos * HALF;
    const uint16_t* sinp = static_cast<const uint16_t*>(a.sinv) + pos * HALF;
    const int ntiles = (ctx + 31) >> 5;
    auto tile_blk_index = [&](int tile) {
      const int T0 = tile << 5;
      return (size_t)b * a.max_blocks + (a.bs_shift >= 0 ? T0 >> a.bs_shift : T0 / a.BS);
    };
    // the tiles are dealt to the consumers in runs, exactly as decode_attn_fused_kernel deals them to its 4 waves (and merged
    // in the same order below): without a KV split the two produce bit-identical outputs
    constexpr int AW = 4;  // waves that walk tiles
    const int kv_w0 = c < AW ? (ntiles * c) >> 2 : 0, kv_w1 = c < AW ? (ntiles * (c + 1)) >> 2 : 0;
    uint32_t blk_cur = a.block_tables[tile_blk_index(min(kv_w0, max(ntiles - 1, 0)))];
    // q / k / v were written by other CUs in this launch: read past the L1 (buffer loads, sc1)
    const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.q), 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.k), 0, 0x7FFFFFF0, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.v), 0, 0x7FFFFFF0, 0x00020000);
    // ---- new token: consumer 0 rotates k (lanes 0 .. D/16-1), consumer 1 copies v (lanes 0 .. D/8-1); both staged in LDS
    if (c == 0 && lane < HALF / 8) {
      const uint32_t kb = (uint32_t)((((size_t)b * a.Hkv + hk) * D) * 2);
      const u32x4 xa = __builtin_amdgcn_raw_buffer_load_b128(krs, kb + lane * 16, 0, 16);
      const u32x4 xb = __builtin_amdgcn_raw_buffer_load_b128(krs, kb + HALF * 2 + lane * 16, 0, 16);
      float x1[8], x2[8], cs[8], sn[8], y1[8], y2[8];
      unpack8<DT>(xa, x1);
      unpack8<DT>(xb, x2);
      unpack8<DT>(*reinterpret_cast<const u32x4*>(cosp + lane * 8), cs);
      unpack8<DT>(*reinterpret_cast<const u32x4*>(sinp + lane * 8), sn);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        y1[e] = x1[e] * cs[e] - x2[e] * sn[e];
        y2[e] = x2[e] * cs[e] + x1[e] * sn[e];
      }
      const u32x4 r1 = pack8<DT>(y1), r2 = pack8<DT>(y2);
      kv_store8<DT, KV8>(knew + lane * 8, r1);
      kv_store8<DT, KV8>(knew + HALF + lane * 8, r2);
      if (slot >= 0) {
        kv_t* kcp = static_cast<kv_t*>(L.kc) + ((((size_t)slot_blk) * a.Hkv + hk) * a.BS + slot_off) * D;
        kv_store8<DT, KV8>(kcp + lane * 8, r1);
        kv_store8<DT, KV8>(kcp + HALF + lane * 8, r2);
      }
    } else if (c == 1 && lane < D / 8) {
      const u32x4 vv = __builtin_amdgcn_raw_buffer_load_b128(vrs, (uint32_t)((((size_t)b * a.Hkv + hk) * D) * 2) + lane * 16, 0, 16);
      *reinterpret_cast<u32x4*>(vnew + lane * 8) = kv_roundtrip8<DT, KV8>(vv);
      if (slot >= 0) {
        kv_t* vcp = static_cast<kv_t*>(L.vc) + (((size_t)slot_blk) * a.Hkv + hk) * D * a.BS + slot_off;
        if constexpr (KV8) {
          const u32x2 q8 = vra_pack_e4m3x8<DT>(vv);
#pragma unroll
          for (int e = 0; e < 8; e++) vcp[(size_t)(lane * 8 + e) * a.BS] = (uint8_t)(q8[e >> 2] >> (8 * (e & 3)));
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            vcp[(size_t)(lane * 8 + 2 * e) * a.BS] = (uint16_t)(vv[e] & 0xffffu);
            vcp[(size_t)(lane * 8 + 2 * e + 1) * a.BS] = (uint16_t)(vv[e] >> 16);
          }
        }
      }
    }
    // ---- Q fragments, rotated in registers: lane (row rq = q head of the group, octet oct)
    const bool row_valid = rq < G;
    const int qhead = hk * G + min(rq, G - 1);
    s16x8 qf[DJ];
    {
      const uint32_t qb = (uint32_t)((((size_t)b * a.Hq + qhead) * D) * 2);
#pragma unroll
      for (int j = 0; j < DJ / 2; j++) {
        const int c0 = j * 32 + oct * 8;
        u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(qrs, qb + c0 * 2, 0, 16);
        u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(qrs, qb + (HALF + c0) * 2, 0, 16);
        if (!row_valid) va = vb = u32x4{0u, 0u, 0u, 0u};
        float x1[8], x2[8], cs[8], sn[8], y1[8], y2[8];
        unpack8<DT>(va, x1);
        unpack8<DT>(vb, x2);
        unpack8<DT>(*reinterpret_cast<const u32x4*>(cosp + c0), cs);
        unpack8<DT>(*reinterpret_cast<const u32x4*>(sinp + c0), sn);
#pragma unroll
        for (int e = 0; e < 8; e++) {
          y1[e] = x1[e] * cs[e] - x2[e] * sn[e];
          y2[e] = x2[e] * cs[e] + x1[e] * sn[e];
        }
        const u32x4 r1 = pack8<DT>(y1), r2 = pack8<DT>(y2);
        qf[j] = __builtin_bit_cast(s16x8, r1);
        qf[j + DJ / 2] = __builtin_bit_cast(s16x8, r2);
      }
    }
    dp_cbar(ctl, a.err, st.cbar_tgt, lane);  // knew / vnew staged
    DP_STAMP(2);

    const int last = ctx - 1;
    f32x4 o[DT16];
#pragma unroll
    for (int t = 0; t < DT16; t++) o[t] = vra_zero_acc();
    float m_run = -INFINITY, l_run = 0.f;
    const kv_t* kcache = static_cast<const kv_t*>(L.kc);
    const kv_t* vcache = static_cast<const kv_t*>(L.vc);
    for (int tile = kv_w0; tile < kv_w1; tile++) {
      const int T0 = tile << 5;
      const uint32_t blk = blk_cur;
      const int off = a.bs_shift >= 0 ? T0 & (a.BS - 1) : T0 % a.BS;
      blk_cur = a.block_tables[tile_blk_index(min(tile + 1, ntiles - 1))];
      const kv_t* krow0 = kcache + (((size_t)blk * a.Hkv + hk) * a.BS + off + krow_tok) * D;
      const kv_t* krow1 = krow0 + 4 * D;
      const size_t vbase = (((size_t)blk * a.Hkv + hk) * D) * a.BS + off;
      const bool has_new = last >= T0 && last < T0 + 32;
      u32x4 k0[DJ], k1[DJ];
#pragma unroll
      for (int j = 0; j < DJ; j++) {
        k0[j] = kv_load8<DT, KV8>(krow0 + j * 32 + oct * 8);
        k1[j] = kv_load8<DT, KV8>(krow1 + j * 32 + oct * 8);
      }
      u32x4 vfr[DT16];
#pragma unroll
      for (int t = 0; t < DT16; t++) vfr[t] = kv_load8<DT, KV8>(vcache + vbase + (size_t)(t * 16 + rq) * a.BS + oct * 8);
      if (has_new) {  // the new token's K row comes from LDS (this launch's cache write may not be visible here yet)
        if (T0 + krow_tok == last) {
#pragma unroll
          for (int j = 0; j < DJ; j++) k0[j] = kv_load8<DT, KV8>(knew + j * 32 + oct * 8);
        }
        if (T0 + krow_tok + 4 == last) {
#pragma unroll
          for (int j = 0; j < DJ; j++) k1[j] = kv_load8<DT, KV8>(knew + j * 32 + oct * 8);
        }
      }
      f32x4 s0 = vra_zero_acc(), s1 = vra_zero_acc();
#pragma unroll
      for (int j = 0; j < DJ; j++) {
        DT::mfma(s0, __builtin_bit_cast(s16x8, k0[j]), qf[j]);
        DT::mfma(s1, __builtin_bit_cast(s16x8, k1[j]), qf[j]);
      }
      VRA_MFMA_DRAIN();
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
      float ar[4];
#pragma unroll
      for (int r = 0; r < 4; r++) ar[r] = __shfl(alpha, oct * 4 + r, 64);
      u32x4 pa;
      pa[0] = DT::pack2(p[0], p[1]);
      pa[1] = DT::pack2(p[2], p[3]);
      pa[2] = DT::pack2(p[4], p[5]);
      pa[3] = DT::pack2(p[6], p[7]);
      const s16x8 pfrag = __builtin_bit_cast(s16x8, pa);
      const bool tail = T0 + 32 > ctx;
      uint32_t vm[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      int new_e = -1;
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
#pragma unroll
        for (int r = 0; r < 4; r++) o[t][r] *= ar[r];
        DT::mfma(o[t], pfrag, __builtin_bit_cast(s16x8, vv));
      }
    }
    VRA_MFMA_DRAIN();
    l_run = vra_xor16_sum(l_run);
    l_run = vra_xor32_sum(l_run);
    // partial (m, l, O) of this consumer: rows < G only
    if (c < AW) {
#pragma unroll
      for (int t = 0; t < DT16; t++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (oct * 4 + r < G) lds_o[((size_t)c * G + oct * 4 + r) * (D + 4) + t * 16 + rq] = o[t][r];
      if (oct == 0 && rq < G) {
        lds_ml[(c * G + rq) * 2 + 0] = m_run;
        lds_ml[(c * G + rq) * 2 + 1] = l_run;
      }
    }
    DP_STAMP(3);
    dp_cbar(ctl, a.err, st.cbar_tgt, lane);
    DP_STAMP(4);
    // 64 consecutive channels of one row per wave and pass; 8 channels per 16-byte write-through store (DPP row_shl gather)
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(a.attn, 0, 0x7FFFFFF0, 0x00020000);
    for (int idx0 = c * 64; idx0 < G * D; idx0 += DP_NC * 64) {
      const int idx = idx0 + lane;
      const int row = idx / D, d = idx % D;
      float Mx = -INFINITY;
#pragma unroll
      for (int w = 0; w < AW; w++) Mx = fmaxf(Mx, lds_ml[(w * G + row) * 2]);
      const float Ms = Mx == -INFINITY ? 0.f : Mx;
      float Ls = 0.f, acc = 0.f;
#pragma unroll
      for (int w = 0; w < AW; w++) {
        const float f = exp2f(lds_ml[(w * G + row) * 2] - Ms);
        Ls += lds_ml[(w * G + row) * 2 + 1] * f;
        acc += lds_o[((size_t)w * G + row) * (D + 4) + d] * f;
      }
      const uint32_t vbits = DT::from_f32(Ls > 0.f ? acc / Ls : 0.f);
      uint32_t nb[8];
      nb[0] = vbits;
      nb[1] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x101, 0xF, 0xF, true);
      nb[2] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x102, 0xF, 0xF, true);
      nb[3] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x103, 0xF, 0xF, true);
      nb[4] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x104, 0xF, 0xF, true);
      nb[5] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x105, 0xF, 0xF, true);
      nb[6] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x106, 0xF, 0xF, true);
      nb[7] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)vbits, 0x107, 0xF, 0xF, true);
      if ((lane & 7) == 0) {
        const u32x4 pk = {nb[0] | (nb[1] << 16), nb[2] | (nb[3] << 16), nb[4] | (nb[5] << 16), nb[6] | (nb[7] << 16)};
        __builtin_amdgcn_raw_buffer_store_b128(pk, ars, (uint32_t)((((size_t)b * a.Hq + hk * G + row) * D + d) * 2), 0, 16);  // sc1
      }
    }
  }
  dp_arrive(a, ctl, st.arrive_tgt, lane);
  DP_STAMP(5);
}

// ---------------------------------------------------------------------------------------------- kernel
template <class DT, bool AWQ, int D, bool KV8>
__global__ __launch_bounds__(DP_THREADS) void decode_step_kernel(const DPStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t* ctl = reinterpret_cast<uint32_t*>(smem);
  if (tid < DP_CTL_BYTES / 4) ctl[tid] = 0u;
  for (int i = tid; i < (a.xt + 3) / 4; i += DP_THREADS) reinterpret_cast<uint32_t*>(smem + a.zero_off)[i] = 0u;
  const uint32_t base = *a.count;  // phases completed by earlier launches (written at the END of a launch)
  __syncthreads();
  if (wave == 0) {
    dp_loader<AWQ>(a, smem);
    return;
  }
  const int c = wave - 1;
  DPState st = {0u, 0u, 0u, 0u};
  for (int ph = a.ph0; ph < a.ph1; ph++) {
    const int l = ph / DP_PHASES_PER_LAYER, kind = ph % DP_PHASES_PER_LAYER;
    const uint32_t done = base + (uint32_t)(ph - a.ph0);  // phases that must be complete before this one reads its input
    const bool wait_grid = ph > a.ph0;
    DPLayerC& L = dp_layers(a)[l];
    if (kind == 1) dp_attn<DT, D, KV8>(a, L, smem, st, done, wait_grid, c, lane, ph);
    else dp_gemv<DT, AWQ>(a, L.g[kind == 0 ? 0 : kind - 1], smem, st, done, wait_grid, c, lane, ph);
  }
  if (blockIdx.x == 0 && c == 0 && lane == 0) *a.count = base + (uint32_t)(a.ph1 - a.ph0);
}

// ---------------------------------------------------------------------------------------------- host side
static const int kDpMaxLds = 160 * 1024;
static int dp_cur_dev() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d < 0 || d > 63 ? 0 : d;
}
struct DpDevState {
  uint32_t* mem = nullptr;  // 8 x 16 counters | count | err
  int cus = 0;
};
static DpDevState g_dp[64];
bool vra_decode_step_init() {
  DpDevState& s = g_dp[dp_cur_dev()];
  if (s.mem) return true;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dp_cur_dev()) != hipSuccess) return false;
  void* m = nullptr;
  if (hipMalloc(&m, 256 * sizeof(uint32_t)) != hipSuccess) return false;
  if (hipMemset(m, 0, 256 * sizeof(uint32_t)) != hipSuccess) return false;
  s.mem = static_cast<uint32_t*>(m);
  s.cus = p.multiProcessorCount;
  return hipDeviceSynchronize() == hipSuccess;
}
int vra_decode_step_grid() { return g_dp[dp_cur_dev()].cus; }
void vra_decode_step_sync_ptrs(uint32_t** counters, uint32_t** count, uint32_t** err) {
  uint32_t* m = g_dp[dp_cur_dev()].mem;
  *counters = m;
  *count = m ? m + 128 : nullptr;
  *err = m ? m + 144 : nullptr;
}
uint32_t* vra_decode_step_error_word() {
  uint32_t* m = g_dp[dp_cur_dev()].mem;
  return m ? m + 144 : nullptr;
}
// after a timeout the monotonic counters no longer agree with the phase count: start over (no launch may be in flight)
void vra_decode_step_reset() {
  DpDevState& s = g_dp[dp_cur_dev()];
  if (s.mem) (void)hipMemset(s.mem, 0, 256 * sizeof(uint32_t));
}
// OFF by default: measured slower than the launch-per-op decode (2.0 against 1.75 ms per step of Llama-3-8B at bs 1; DESIGN.md
// 3.1d has the timelines and why).  VRA_DECODE_STEP=1 or vra_debug_set_decode_step(1) turns it on (the parity tests do).
static int g_dp_enable = -1;  // -1: the environment decides
extern "C" void vra_debug_set_decode_step(int on) { g_dp_enable = on; }
bool vra_decode_step_enabled() {
  if (g_dp_enable >= 0) return g_dp_enable != 0;
  static const char* on = getenv("VRA_DECODE_STEP");
  return on && on[0] == '1';
}
bool vra_decode_step_plan(int M, int max_kt, int max_red, int group, int D, DPPlan* plan) {
  if (M < 1 || M > DP_MAX_ROWS || max_kt < 1 || max_red < 1 || group < 1 || group > 16 || (D != 64 && D != 128)) return false;
  const int xt = M * 272 + 16;
  size_t xbytes = (size_t)max_kt * xt;
  const size_t attn_bytes = (size_t)4 * group * (D + 4) * 4 + (size_t)4 * group * 8 + 512;
  if (attn_bytes > xbytes) xbytes = attn_bytes;
  xbytes = (xbytes + 15) & ~(size_t)15;
  const size_t zero_off = DP_CTL_BYTES;  // (filled in below: behind the ring, at the end of the x region)
  (void)zero_off;
  xbytes += (size_t)((xt + 15) & ~15);  // the all-zero k-tile
  const size_t red_bytes = (size_t)max_red * 16 * M * 16 * 4;
  const size_t fixed = DP_CTL_BYTES + xbytes + red_bytes;
  if (fixed + 3 * (size_t)DP_SLOT_BYTES > (size_t)kDpMaxLds) return false;
  int nslot = (int)(((size_t)kDpMaxLds - fixed) / DP_SLOT_BYTES);
  if (nslot > DP_MAX_SLOTS) nslot = DP_MAX_SLOTS;
  static const char* ns_env = getenv("VRA_DP_SLOTS");  // tuning aid
  if (ns_env && atoi(ns_env) >= 3 && atoi(ns_env) < nslot) nslot = atoi(ns_env);
  plan->nslot = nslot;
  plan->ring_off = DP_CTL_BYTES;
  plan->x_off = DP_CTL_BYTES + nslot * DP_SLOT_BYTES;
  plan->red_off = plan->x_off + (int)xbytes;
  plan->zero_off = plan->red_off - ((xt + 15) & ~15);
  plan->xt = xt;
  plan->lds_bytes = plan->red_off + (int)red_bytes;
  return true;
}

#ifdef VRA_GEMV_TS
static unsigned long long* g_dp_ts = nullptr;
extern "C" void vra_debug_decode_step_ts(unsigned long long* host, int n) {
  if (g_dp_ts) (void)hipMemcpy(host, g_dp_ts, (size_t)n * 8, hipMemcpyDeviceToHost);
}
#endif

template <class DT, bool AWQ, int D, bool KV8>
static void dp_launch_v(DPStepArgs a, size_t lds, hipStream_t st) {
  static uint64_t attr_devs = 0;
  auto kern = decode_step_kernel<DT, AWQ, D, KV8>;
  if (!((attr_devs >> dp_cur_dev()) & 1)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kDpMaxLds);
    attr_devs |= (uint64_t)1 << dp_cur_dev();
  }
#ifdef VRA_GEMV_TS
  if (!g_dp_ts) {
    (void)hipMalloc(&g_dp_ts, (size_t)(256 * 256 * 8 + 256 * 64 * 4 + 8 * 16 * 8) * 8);
    (void)hipMemset(g_dp_ts, 0, (size_t)(256 * 256 * 8 + 256 * 64 * 4 + 8 * 16 * 8) * 8);
  }
  a.ts = g_dp_ts;
  static const char* tsp = getenv("VRA_TS_PHASE");
  a.ts_phase = tsp ? atoi(tsp) : 8;
#else
  a.ts = nullptr;
#endif
  kern<<<vra_decode_step_grid(), DP_THREADS, lds, st>>>(a);
}
void vra_launch_decode_step(const DPStepArgs& a0, int dtype, bool awq, bool kv8, int head_dim, int64_t stream) {
  DPStepArgs a = a0;
  hipStream_t st = as_stream(stream);
  if (!g_dp[dp_cur_dev()].mem) {
    vra_set_error("decode_step: vra_decode_step_init() first");
    return;
  }
  if (a.M < 1 || a.M > DP_MAX_ROWS || a.ph0 < 0 || a.ph1 > a.n_layers * DP_PHASES_PER_LAYER || a.ph1 - a.ph0 > 256 || a.nslot < 3 ||
      a.nslot > DP_MAX_SLOTS) {
    vra_set_error("decode_step: bad arguments (rows %d, phases %d..%d, slots %d)", a.M, a.ph0, a.ph1, a.nslot);
    return;
  }
  vra_decode_step_sync_ptrs(&a.counters, &a.count, &a.err);
  static const char* dbg_env = getenv("VRA_DP_DBG");
  a.dbg = dbg_env ? atoi(dbg_env) : 0;
  const bool bf = dtype == VRA_BF16;
  // the whole LDS of the CU is requested: exactly one workgroup per CU (every workgroup of the grid must be resident)
#define DP_GO(DT, AW, DD, K8) dp_launch_v<DT, AW, DD, K8>(a, (size_t)kDpMaxLds, st)
  if (head_dim == 128) {
    if (bf) {
      if (awq) kv8 ? DP_GO(BF16, true, 128, true) : DP_GO(BF16, true, 128, false);
      else kv8 ? DP_GO(BF16, false, 128, true) : DP_GO(BF16, false, 128, false);
    } else {
      if (awq) kv8 ? DP_GO(F16, true, 128, true) : DP_GO(F16, true, 128, false);
      else kv8 ? DP_GO(F16, false, 128, true) : DP_GO(F16, false, 128, false);
    }
  } else if (head_dim == 64) {
    if (bf) {
      if (awq) kv8 ? DP_GO(BF16, true, 64, true) : DP_GO(BF16, true, 64, false);
      else kv8 ? DP_GO(BF16, false, 64, true) : DP_GO(BF16, false, 64, false);
    } else {
      if (awq) kv8 ? DP_GO(F16, true, 64, true) : DP_GO(F16, true, 64, false);
      else kv8 ? DP_GO(F16, false, 64, true) : DP_GO(F16, false, 64, false);
    }
  } else {
    vra_set_error("decode_step: head_dim %d not supported (64, 128)", head_dim);
  }
#undef DP_GO
}
