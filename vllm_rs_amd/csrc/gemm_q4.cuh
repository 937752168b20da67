// gemm_q4.cuh — "kernel C": int4 GEMM for 5..32 activation rows (decode batches) that streams every packed weight
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
//   KT/KS tile-steps with an 8 KiB register ring, branch-free, exact vmcnt (see gemv_q4.cuh for why that matters).
//   MT m-tiles (16 rows each) share every dequantised B fragment: 4*MT MFMAs per tile and tensor.
//   x reaches the MFMAs through LDS in K-chunks of KC = 512 or 1024 (double buffered, row-major with one octet of padding per row), staged
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
  int ks_shift, ktz, n_items;  // log2(ks), K/128/kz and ceil(n_blocks / (8/ks)): quotients the launcher precomputes (an
                               // integer division is ~25 VALU instructions; one of them sat in the compute waves' inner loop)
  int kc;        // k per staged chunk: 512 or 1024 ((K/kz) % kc == 0, (kc/128) % ks == 0)
  int kz;        // K slices across workgroups (grid.z): 1 = none.  Slice partials go to fp32 slabs and the LAST slice's
                 // workgroup of an (item, m-chunk) reduces them in slice order (deterministic) and runs the epilogue
  float* slabs;        // [kz][row tile][item][NBW][half][unit][4] f32 partial tiles of the K slices
  uint32_t* counters;  // arrival flags, 16 words apart, one per (item, row tile, slice): zero on entry, zero on exit
  uint32_t* err;       // scratch error word: set when a slice wait timed out (vra_scratch_error)
  unsigned long long* ts;  // VRA_GEMV_TS builds: [grid.z][grid.x][32] wall-clock stamps (compute wave 0: 0..15, producer wave 0: 16..31)
  // single-segment 16-bit launches of up to 32 rows: the outputs ALSO go to this buffer in kernel W's fragment order (gemv_q4w.cuh
  // x_frag: u32x4 word ((kt*2 + mt)*4 + j)*64 + oct*16 + nn = row mt*16 + nn, columns kt*128 + j*32 + oct*8 .. +7) — the next
  // launch that consumes them as x then fetches one contiguous KiB per wave load instead of 16 half lines
  void* out_frag;
};
#ifdef VRA_GEMV_TS
#define GC_STAMP(i)                                                                                   \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (a.ts && lane == 0 && (wave == 0 || wave == GC_CW))                                            \
      a.ts[((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 32 + (i)] = wall_clock64();               \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#define GC_CYC(v)                                 \
  do {                                            \
    __builtin_amdgcn_sched_barrier(0);            \
    v = (long long)__builtin_readcyclecounter();  \
    __builtin_amdgcn_sched_barrier(0);            \
  } while (0)
#else
#define GC_STAMP(i) do {} while (0)
#define GC_CYC(v) do {} while (0)
#endif

static inline size_t gemm_q4_lds_bytes(int nbw, int mt, int kc) {
  const int rows = 16 * mt;
  const size_t x2 = (size_t)2 * rows * (kc / 8 + 1) * 16, red = (size_t)GC_CW * nbw * mt * 64 * 16;  // red aliases the x buffers
  return (x2 > red ? x2 : red) + (size_t)2 * (kc / 128) * rows * 4 + 64;  // x (| red) | Σx | flag
}

template <class DT, int NBW, int MT, bool AWQ>
__global__ __launch_bounds__(GC_THREADS) void gemm_q4_kernel(const GemmCArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ROWS = 16 * MT;
  // logical wave / thread ids: the FIRST four hardware waves are the producers (logical waves 8..11).  A workgroup's waves are
  // launched in order and x is on the critical path of the start — the compute waves only have their ring to request.
  const int lane = (int)threadIdx.x & 63;
  const int hw_wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wave = hw_wave < GC_PW ? hw_wave + GC_CW : hw_wave - GC_PW;
  const int tid = wave * 64 + lane;
  const bool is_prod = wave >= GC_CW;
  const int nn = lane & 15, oct = lane >> 4;
  const int K = a.K, M = a.M, KT = K >> 7;
  const int KZ = (NBW == 1 && a.kz > 1) ? a.kz : 1, zi = (int)blockIdx.z;  // K slices across workgroups: narrow GEMMs only
  const int KTZ = a.ktz, kt0 = zi * KTZ;             // this workgroup's K slice, in tiles (= KT / KZ)
  const int KC = a.kc, TPC = KC >> 7, tpc_sh = KC == 1024 ? 3 : 2, NC = KTZ >> tpc_sh, OPC = KC >> 3;  // tiles, chunks, octets per chunk
  const int KS = a.ks, ks_sh = a.ks_shift, CGN = GC_CW >> ks_sh;
  const int SC = TPC >> ks_sh, sc_sh = tpc_sh - ks_sh;  // steps of one compute wave per chunk
  const int T = KTZ >> ks_sh;     // steps per item
  const bool grouped = a.group_size > 0 && a.group_size < K;
  const int gsh = grouped ? 31 - __builtin_clz(a.group_size) : 31;
  const int m0 = (int)blockIdx.y * ROWS;
  const int n_items = a.n_items;  // = ceil(n_blocks / CGN)

  // ---- LDS
  // x buffer: row-major with one octet of padding per row — the producers write 64 consecutive octets of a row per wave
  // (consecutive LDS addresses), the MFMA fragment reads (16 rows x 4 octets) hit 16 distinct bank quads per quarter wave.
  // (The [octet][row] XOR-swizzled layout of kernel B made every producer ds_write_b128 an 8-way bank conflict.)
  const int RS = (OPC + 1) * 4;       // row stride in u32
  const int XS_U32 = ROWS * RS;       // one x buffer in u32
  uint32_t* xs = reinterpret_cast<uint32_t*>(smem);
  float* xsum = reinterpret_cast<float*>(smem + (size_t)2 * XS_U32 * 4);  // [2][TPC][ROWS]
  // the partial tiles of an item ALIAS x buffer 0: they are written after the item's last chunk has been consumed and read by
  // the producers while the next item's FIRST chunk is being staged — which is why chunk c lives in buffer (c + 1) & 1: chunk 0
  // goes to buffer 1, and buffer 0 is first written (chunk 1) behind the chunk-0 barrier, which every producer wave only reaches
  // after its share of the epilogue.  (Round 3: with chunk 0 in buffer 0 a producer wave WITHOUT epilogue work — KS = 8 leaves
  // 64 units for 256 producer threads — staged the next item's x over partial tiles another producer wave was still summing:
  // non-reproducible logits at the TinyLlama widths, 17..32 rows, the only test shape with more items than CUs and KS = 8.)
  f32x4* red = reinterpret_cast<f32x4*>(smem);  // [GC_CW][NBW][MT][64] (<= 32 KiB <= one x buffer)
  int* flag = reinterpret_cast<int*>(smem + (size_t)2 * XS_U32 * 4 + (size_t)2 * TPC * ROWS * 4);

  const int nseg = a.nseg;
  const int blk1 = nseg > 1 ? a.seg[1].blk_start : 0x7fffffff, blk2 = nseg > 2 ? a.seg[2].blk_start : 0x7fffffff;

  // ---- CGN n-blocks x ROWS rows x 16 columns, handled as 8-column vectors (16 B stores; a scalar loop over single
  // outputs exposed one global round trip per output: ~35 us).  Sum the KS partial tiles; with K slices publish to the
  // slab and let the last slice's workgroup (the owner) finish.
  const int nunits = CGN * ROWS * 2;
  const float* rf = reinterpret_cast<const float*>(red);
  auto unit_geom = [&](int u, int& cgi, int& mrow, int& nl0) {
    cgi = u / (ROWS * 2);
    const int rem = u - cgi * ROWS * 2;
    mrow = rem >> 1;
    nl0 = (rem & 1) * 8;
  };
  auto lds_sum = [&](int cgi, int mrow, int nl0, float (&v)[8], float (&v2)[8]) {
    const int mt = mrow >> 4, mm = mrow & 15;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int lslot = ((mm >> 2) * 16 + nl0 + e) * 4 + (mm & 3);  // D layout: column = lane&15, row = (lane>>4)*4 + reg
      float s0 = 0.f, s1 = 0.f;
      for (int q = 0; q < KS; q++) {
        const int w = cgi * KS + q;
        s0 += rf[(size_t)((w * NBW + 0) * MT + mt) * 256 + lslot];
        if (NBW > 1) s1 += rf[(size_t)((w * NBW + (NBW - 1)) * MT + mt) * 256 + lslot];
      }
      v[e] = s0;
      v2[e] = s1;
    }
  };
  // slab of (slice z, this item, this row tile): [tensor][half][unit][4] f32 — a wave's 16-byte accesses are contiguous.
  // Accessed with buffer instructions carrying sc1 (agent scope: write through / re-fetch, the L2s of the XCDs are not
  // coherent with each other); 16-byte accesses — 8-byte agent-scope atomics re-fetched every line four times (+6 us).
  const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(a.slabs, 0, 0x7FFFFFF0, 0x00020000);
  auto slab_off = [&](int it, int z) {
    return (uint32_t)(((z * (int)gridDim.y + (int)blockIdx.y) * n_items + it) * NBW) * (uint32_t)nunits * 32u;  // bytes
  };
  // partial tiles of this slice -> its slab (agent-scope, write-through stores)
  auto store_units = [&](int it, int et, int ET) {
    for (int u = et; u < nunits; u += ET) {
      int cgi, mrow, nl0;
      unit_geom(u, cgi, mrow, nl0);
      if (it * CGN + cgi >= a.n_blocks) continue;
      float v[8], v2[8];
      lds_sum(cgi, mrow, nl0, v, v2);
      const uint32_t o = slab_off(it, zi) + (uint32_t)u * 16u;
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, srs, o, 0, 16);
      __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])}, srs,
                                             o + (uint32_t)nunits * 16u, 0, 16);
      if (NBW > 1) {
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v2[0]), __float_as_uint(v2[1]), __float_as_uint(v2[2]), __float_as_uint(v2[3])}, srs,
                                               o + (uint32_t)nunits * 32u, 0, 16);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(v2[4]), __float_as_uint(v2[5]), __float_as_uint(v2[6]), __float_as_uint(v2[7])}, srs,
                                               o + (uint32_t)nunits * 48u, 0, 16);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-through stores: acknowledged by memory
  };
  // sum (the KS partial tiles in LDS, plus the other slices' slabs), fused epilogue, store
  auto finish_units = [&](int it, int et, int ET) {
    for (int u = et; u < nunits; u += ET) {
      int cgi, mrow, nl0;
      unit_geom(u, cgi, mrow, nl0);
      const int fb = it * CGN + cgi;
      const int m = m0 + mrow;
      if (fb >= a.n_blocks || m >= M) continue;
      const int segi = NBW == 2 ? 0 : (fb >= blk2 ? 2 : (fb >= blk1 ? 1 : 0));
      const int nb = fb - (NBW == 2 ? 0 : a.seg[segi].blk_start);
      const GemvSeg& sg = a.seg[segi];
      const int n = nb * 16 + nl0;
      // every global load of this unit goes out before anything is consumed
      u32x4 bw = {0u, 0u, 0u, 0u}, bw2 = {0u, 0u, 0u, 0u}, rw = {0u, 0u, 0u, 0u};
      if (sg.bias) bw = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(sg.bias) + n);
      if (NBW == 2 && a.seg[1].bias) bw2 = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.seg[1].bias) + n);
      if (a.residual) rw = *reinterpret_cast<const u32x4*>(static_cast<const uint16_t*>(a.residual) + (size_t)m * a.res_ld + n);
      float v[8], v2[8];
      if (KZ > 1) {
        // the other K slices of the unit, all in flight at once (a loop of dependent loads costs a memory round trip
        // per slice), summed in slice order: deterministic
        auto slab_sum = [&](uint32_t toff, float (&acc)[8]) {
#pragma unroll
          for (int e = 0; e < 8; e++) acc[e] = 0.f;
          for (int z0 = 0; z0 < KZ - 1; z0 += 8) {
            u32x4 p[8][2];
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const uint32_t o = slab_off(it, min(z0 + j, KZ - 2)) + toff + (uint32_t)u * 16u;
              p[j][0] = __builtin_amdgcn_raw_buffer_load_b128(srs, o, 0, 16);
              p[j][1] = __builtin_amdgcn_raw_buffer_load_b128(srs, o + (uint32_t)nunits * 16u, 0, 16);
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
              if (z0 + j < KZ - 1) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                  acc[e] += __uint_as_float(p[j][0][e]);
                  acc[4 + e] += __uint_as_float(p[j][1][e]);
                }
              }
            }
          }
        };
        float own[8], own2[8];
        lds_sum(cgi, mrow, nl0, own, own2);
        slab_sum(0u, v);
        if (NBW > 1) slab_sum((uint32_t)nunits * 32u, v2);
#pragma unroll
        for (int e = 0; e < 8; e++) {
          v[e] += own[e];
          v2[e] = NBW > 1 ? v2[e] + own2[e] : 0.f;
        }
        GC_STAMP(27);
      } else {
        lds_sum(cgi, mrow, nl0, v, v2);
      }
      float bf[8], bf2[8], rf8[8], o8[8];
      unpack8<DT>(bw, bf);
      unpack8<DT>(bw2, bf2);
      unpack8<DT>(rw, rf8);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float t = rnd_dt<DT>(v[e]);
        if (sg.bias) t = rnd_dt<DT>(t + bf[e]);
        if (NBW == 2) {
          float t2 = rnd_dt<DT>(v2[e]);
          if (a.seg[1].bias) t2 = rnd_dt<DT>(t2 + bf2[e]);
          const float sl = rnd_dt<DT>(t / (1.0f + expf(-t)));
          t = sl * t2;
        }
        if (a.residual) t = rnd_dt<DT>(t) + rf8[e];
        o8[e] = t;
      }
      if (a.out_f32) {
        float* op = static_cast<float*>(sg.out) + (size_t)m * sg.out_ld + n;
        *reinterpret_cast<f32x4*>(op) = f32x4{rnd_dt<DT>(o8[0]), rnd_dt<DT>(o8[1]), rnd_dt<DT>(o8[2]), rnd_dt<DT>(o8[3])};
        *reinterpret_cast<f32x4*>(op + 4) = f32x4{rnd_dt<DT>(o8[4]), rnd_dt<DT>(o8[5]), rnd_dt<DT>(o8[6]), rnd_dt<DT>(o8[7])};
      } else {
        const u32x4 pk = pack8<DT>(o8);
        *reinterpret_cast<u32x4*>(static_cast<uint16_t*>(sg.out) + (size_t)m * sg.out_ld + n) = pk;
        if (a.out_frag && m < 32)
          static_cast<u32x4*>(a.out_frag)[(size_t)((((n >> 7) * 2 + (m >> 4)) * 4 + ((n >> 5) & 3)) * 64) + ((n >> 3) & 3) * 16 + (m & 15)] = pk;
      }
    }
  };
  // With K slices every thread of the workgroup takes part in the exchange (one item per workgroup: nothing to overlap it
  // with): team index of this thread, producers first
  const int ET_ALL = GC_THREADS, et_all = is_prod ? tid - GC_CW * 64 : tid + GC_PW * 64;
  const bool owner = zi == KZ - 1;
  const bool single = n_items <= (int)gridDim.x;  // one item per workgroup: the epilogue is not overlapped with a next item
  if (is_prod) {
    // ============================== producer waves: x chunks -> LDS, then the epilogue of every item
    const int pt = tid - GC_CW * 64;                   // 0..255
    constexpr int PTHREADS = GC_PW * 64;
    const int per = (ROWS * OPC) / PTHREADS;            // octets per thread and chunk (4 .. 16)
    const int osh = OPC == 128 ? 7 : 6;                 // OPC is 64 or 128
    // A chunk is REQUESTED two chunks ahead of the one the compute waves are on, into one of two register sets, and written
    // to LDS one chunk ahead: the first touch of x by an XCD is an L2 miss behind the weight streams of every workgroup —
    // tools/gemm_c_ts.py showed 3.7 µs from request to staged against 2.7 µs of MFMA work per chunk, i.e. the compute waves
    // waited for the producers at every chunk barrier.  All loads are unconditional (a thread with a smaller share re-reads
    // its first octet) so that hipcc's vmcnt for the older set stays exact while the younger one is in flight.
    constexpr int PER_MAX = ROWS * 128 / PTHREADS;      // 8 (16 rows) or 16 (32 rows)
    // (buffer loads: ONE per-lane offset for all of a thread's loads — its octet within the row — and a wave-uniform SGPR
    // offset per load for the row and the chunk; with flat addresses the two register sets spilled)
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, 0x7FFFFFF0, 0x00020000);
    const uint32_t vo = (uint32_t)(pt & (OPC - 1)) * 16u;
    const int prow = __builtin_amdgcn_readfirstlane(pt >> osh);  // a wave-load covers 64 consecutive octets of ONE row
    const int rstep = PTHREADS >> osh;                            // rows between consecutive loads of a thread
    auto request = [&](int c, u32x4 (&v)[PER_MAX]) {
#pragma unroll
      for (int r = 0; r < PER_MAX; r++) {
        const int m = min(m0 + prow + (r < per ? r : 0) * rstep, M - 1);  // rows >= M alias row M-1 (never stored)
        v[r] = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, (uint32_t)((m * a.x_ld + kt0 * 128 + c * KC) * 2), 0);
      }
    };
    auto commit = [&](int buf, const u32x4 (&v)[PER_MAX]) {
      // (the address walks down the rows behind an opaque copy: left to itself hipcc keeps the 16 addresses of both buffers
      // in registers across the chunk loop and spills)
      uint32_t off = (uint32_t)((buf ^ 1) * XS_U32 + prow * RS + (pt & (OPC - 1)) * 4);  // in u32; chunk parity `buf` -> buffer buf ^ 1 (see `red`)
      asm volatile("" : "+v"(off));
#pragma unroll
      for (int r = 0; r < PER_MAX; r++) {
        if (r < per) *reinterpret_cast<u32x4*>(xs + off) = v[r];
        off += (uint32_t)(rstep * RS);
      }
    };
    for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
      GC_STAMP(16);
      u32x4 va[PER_MAX], vb[PER_MAX];
      request(0, va);
      request(min(1, NC - 1), vb);
      commit(0, va);
      GC_STAMP(17);
      __syncthreads();  // chunk 0 is staged
      int c = 1;
      for (; c + 1 < NC; c += 2) {  // chunks c (held in vb) and c + 1 (va)
        request(c + 1, va);
        commit(1, vb);
        GC_STAMP(17 + (c < 6 ? c : 6));
        __syncthreads();
        request(min(c + 2, NC - 1), vb);
        commit(0, va);
        GC_STAMP(17 + (c + 1 < 6 ? c + 1 : 6));
        __syncthreads();
      }
      if (c < NC) {
        commit(1, vb);
        GC_STAMP(17 + (c < 6 ? c : 6));
        __syncthreads();
      }
      __syncthreads();  // every compute wave has consumed the last chunk
      __syncthreads();  // the item's partial tiles are in `red`
      GC_STAMP(24);
      if (NBW == 1 && KZ > 1) {
        // K slices meet through memory (the XCDs' L2s are not coherent with each other).  Measured on MI355X:
        //  * an arrival COUNTER serialises — agent-scope atomics of 8 slices on one address completed ~1.3 us apart;
        //  * plain stores + a release fence (buffer_wbl2: write back the whole L2) took 2..8 us per workgroup when every
        //    workgroup of the XCD does it at once.
        // So: partials go out as agent-scope (write-through, sc1) stores, every slice raises its own flag (one 64-byte line
        // each) after they are acknowledged, and the LAST slice (z = KZ-1, the "owner") polls the flags, sums the slabs
        // with agent-scope loads (its own partial comes straight from LDS, last in the fixed order) and resets the flags.
        // No fences.  Progress: only owners wait, and there are fewer owners (items x row tiles <= 48) than CUs, so in
        // any dispatch order some non-owner is resident, finishes without waiting and frees its slot (the launcher
        // checks owners < CUs; in practice workgroups are dispatched in linear-id order and owners start last).  A lost
        // slice ends the wait after ~1 s instead of hanging the device and raises the scratch error word.
        uint32_t* fl = a.counters + ((size_t)(it * gridDim.y + blockIdx.y) * KZ) * 16;
        if (!owner) {
          store_units(it, et_all, ET_ALL);
          __syncthreads();
          if (pt == 0) __hip_atomic_store(fl + zi * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          if (pt < KZ - 1) {
            const uint64_t t0 = __builtin_readcyclecounter();
            while (__hip_atomic_load(fl + pt * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
              __builtin_amdgcn_s_sleep(1);
              if (__builtin_readcyclecounter() - t0 > (1ull << 31)) {  // never hang the device on a lost slice
                __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
            }
            __hip_atomic_store(fl + pt * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          __syncthreads();
        }
        __syncthreads();
        GC_STAMP(26);
        if (owner) finish_units(it, et_all, ET_ALL);
      } else if (single) {
        finish_units(it, et_all, ET_ALL);  // one item per workgroup: nothing for the compute waves to run ahead into — all 12 waves finish it
      } else {
        finish_units(it, pt, PTHREADS);
      }
      GC_STAMP(25);
    }
    return;
  }

  // ============================== compute waves
  const int cg = wave >> ks_sh, ksi = wave & (KS - 1);
  const int zsh = 4 * awq_rev(nn & 7);
  const bool shalf = nn & 1;
  constexpr float CB = Magic<DT>::bias;
  u32x4 ones_w;
  ones_w[0] = ones_w[1] = ones_w[2] = ones_w[3] = DT::id == VRA_BF16 ? 0x3F803F80u : 0x3C003C00u;  // eight 1.0
  asm volatile("" : "+v"(ones_w));
  const s16x8 ones = __builtin_bit_cast(s16x8, ones_w);

  for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int fb = it * CGN + cg;
    const bool have = fb < a.n_blocks;
    const int fbc = have ? fb : a.n_blocks - 1;  // a wave without an n-block streams the last one with zeroed scales
    // wave-uniform tensor bases of this item as buffer resources: every ring load is (resource, ONE per-lane offset fixed for
    // the item, an SGPR offset for the tile / scale group) — no per-load 64-bit VALU address arithmetic.  The loop of this
    // kernel is VALU bound at 32 rows (110 VALU instructions per tile-step against 24 MFMAs; two compute waves per SIMD).
    __amdgpu_buffer_rsrc_t rw[NBW], rs[NBW], rz[NBW];
    int ncols[NBW], nbv;
    if (NBW == 2) {
      nbv = fbc;
#pragma unroll
      for (int b = 0; b < NBW; b++) {
        rw[b] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.seg[b].w), 0, 0x7FFFFFF0, 0x00020000);
        rs[b] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.seg[b].scales), 0, 0x7FFFFFF0, 0x00020000);
        rz[b] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(AWQ ? a.seg[b].qzeros : reinterpret_cast<const uint32_t*>(a.seg[b].scales)), 0, 0x7FFFFFF0, 0x00020000);
        ncols[b] = a.seg[b].n;
      }
    } else {
      const bool s1 = fbc >= blk1, s2 = fbc >= blk2;
      const void* w_ = s2 ? a.seg[2].w : (s1 ? a.seg[1].w : a.seg[0].w);
      const void* s_ = s2 ? a.seg[2].scales : (s1 ? a.seg[1].scales : a.seg[0].scales);
      const uint32_t* z_ = s2 ? a.seg[2].qzeros : (s1 ? a.seg[1].qzeros : a.seg[0].qzeros);
      nbv = fbc - (s2 ? blk2 : (s1 ? blk1 : 0));
      rw[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(w_), 0, 0x7FFFFFF0, 0x00020000);
      rs[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(s_), 0, 0x7FFFFFF0, 0x00020000);
      rz[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(AWQ ? z_ : reinterpret_cast<const uint32_t*>(s_)), 0, 0x7FFFFFF0, 0x00020000);
      ncols[0] = s2 ? a.seg[2].n : (s1 ? a.seg[1].n : a.seg[0].n);
    }
    const uint32_t vo_w = (uint32_t)lane * 16u;
    const uint32_t vo_s = (uint32_t)((nbv * 16 + nn) >> 1) * 4u;  // the 32-bit word that holds the lane's 16-bit scale
    const uint32_t vo_z = (uint32_t)(nbv * 2 + (nn >> 3)) * 4u;
    // ring depth: only 8 streaming waves per workgroup (and one workgroup per CU when LDS is full): 8 KiB per wave in
    // flight = 64 KiB per CU (a 2-step ring measured a ~34 us latency floor: every step waited for HBM)
#ifndef GC_RING_MT2
#define GC_RING_MT2 6
#endif
#ifndef GC_RING_MT2_PAIR
#define GC_RING_MT2_PAIR 3
#endif
    constexpr int D = MT == 2 ? (NBW == 2 ? GC_RING_MT2_PAIR : GC_RING_MT2) : 8 / NBW;  // (MT = 2 holds twice the accumulators and x fragments: a shorter ring avoids spills)
    u32x4 wb[D][NBW];
    uint32_t sb[D][NBW], zb[D][NBW];
    auto issue = [&](int i, u32x4 (&w)[NBW], uint32_t (&sc)[NBW], uint32_t (&zp)[NBW]) {
      const int kt = kt0 + ksi + KS * min(i, T - 1);  // steps past the end re-read the last tile (never consumed)
      const int grp = (kt * 128) >> gsh;
#pragma unroll
      for (int b = 0; b < NBW; b++) {
        w[b] = __builtin_amdgcn_raw_buffer_load_b128(rw[b], vo_w, (uint32_t)((nbv * KT + kt) * 1024), 2);  // nt
        sc[b] = __builtin_amdgcn_raw_buffer_load_b32(rs[b], vo_s, (uint32_t)(grp * ncols[b] * 2), 0);
        if (AWQ) zp[b] = __builtin_amdgcn_raw_buffer_load_b32(rz[b], vo_z, (uint32_t)(grp * (ncols[b] >> 3) * 4), 0);
        else zp[b] = 0;
      }
    };
    GC_STAMP(0);
#pragma unroll
    for (int r = 0; r < D; r++) issue(r, wb[r], sb[r], zb[r]);
    GC_STAMP(1);

    f32x4 acc[NBW][MT];
#pragma unroll
    for (int b = 0; b < NBW; b++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) acc[b][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int T_pad = (T + D - 1) / D * D;
    long long cA = 0, cB = 0, cyc[3] = {0, 0, 0};  // TS builds: cycles in chunk barriers | ring wait + dequant + MFMAs | fix-up + refill
    (void)cA, (void)cB, (void)cyc;
    for (int i0 = 0; i0 < T_pad; i0 += D) {
#pragma unroll
      for (int r = 0; r < D; r++) {
        const int i = i0 + r;
        if (i < T && (i & (SC - 1)) == 0) {
          GC_STAMP(2 + 2 * ((i >> sc_sh) < 5 ? (i >> sc_sh) : 5));
          GC_CYC(cA);
          __syncthreads();  // chunk i/SC is staged (and chunk i/SC - 2's buffer is free)
          GC_CYC(cB);
          cyc[0] += cB - cA;
          GC_STAMP(3 + 2 * ((i >> sc_sh) < 5 ? (i >> sc_sh) : 5));
        }
        if (i < T) {
          GC_CYC(cA);
          const int ktl = ksi + KS * i;  // tile within this workgroup's K slice
          const int c = ktl >> tpc_sh, tl = ktl & (TPC - 1);
          const uint32_t* xb = xs + (size_t)((c & 1) ^ 1) * XS_U32;
          f32x4 ag[NBW][MT];  // (every chain STARTS with the C = 0 form of the MFMA: no accumulator is zeroed on the VALU)
          // LDS reads run one k-step (j) ahead of the MFMAs that consume them.  The row sums Σx of the zero-point fix-up
          // come from one extra MFMA per (j, m-tile) against an all-ones B fragment: D[m][n] = Σ_k x[m][k] lands in
          // exactly the lanes that need it, and the producers only have to move bytes.
          u32x4 xv[2][MT];
          f32x4 sxv[MT];
#pragma unroll
          for (int mt = 0; mt < MT; mt++) {
            const int o = tl * 16 + oct;
            xv[0][mt] = *reinterpret_cast<const u32x4*>(xb + (size_t)(mt * 16 + nn) * RS + o * 4);
          }
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (j < 3) {
              const int o = tl * 16 + (j + 1) * 4 + oct;
#pragma unroll
              for (int mt = 0; mt < MT; mt++) xv[(j + 1) & 1][mt] = *reinterpret_cast<const u32x4*>(xb + (size_t)(mt * 16 + nn) * RS + o * 4);
            }
            s16x8 bfrag[NBW];
#pragma unroll
            for (int b = 0; b < NBW; b++) bfrag[b] = magic_word<DT>(wb[r][b][j]);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
              const s16x8 afrag = __builtin_bit_cast(s16x8, xv[j & 1][mt]);
#pragma unroll
              for (int b = 0; b < NBW; b++) {
                if (j == 0) DT::mfma0(ag[b][mt], afrag, bfrag[b]);
                else DT::mfma(ag[b][mt], afrag, bfrag[b]);
              }
              if (j == 0) DT::mfma0(sxv[mt], afrag, ones);
              else DT::mfma(sxv[mt], afrag, ones);
            }
          }
          VRA_MFMA_DRAIN();
          GC_CYC(cB);
          cyc[1] += cB - cA;
#pragma unroll
          for (int b = 0; b < NBW; b++) {
            float s = DT::to_f32((uint16_t)(shalf ? sb[r][b] >> 16 : sb[r][b]));
            s = have ? s : 0.f;
            const float zc = AWQ ? CB + (float)((zb[r][b] >> zsh) & 0xFu) : CB + 8.f;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
              const f32x4 sx = sxv[mt];
#pragma unroll
              for (int e = 0; e < 4; e++) acc[b][mt][e] = fmaf(s, fmaf(-zc, sx[e], ag[b][mt][e]), acc[b][mt][e]);
            }
          }
        }
        issue(i + D, wb[r], sb[r], zb[r]);  // unconditional refill (clamped)
        GC_CYC(cA);
        cyc[2] += cA - cB;
      }
    }
#ifdef VRA_GEMV_TS
    if (a.ts && lane == 0) {
      unsigned long long* o = a.ts + (size_t)2048 * 32 + (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * GC_CW + wave) * 4;
      o[0] = (unsigned long long)cyc[0], o[1] = (unsigned long long)cyc[1], o[2] = (unsigned long long)cyc[2], o[3] = (unsigned long long)T;
    }
#endif
    // ---- hand the partial tiles to the producer waves (they alias the x buffers: every wave must be done reading x)
    GC_STAMP(14);
    __syncthreads();
    GC_STAMP(15);
#pragma unroll
    for (int b = 0; b < NBW; b++)
#pragma unroll
      for (int mt = 0; mt < MT; mt++) red[(size_t)((wave * NBW + b) * MT + mt) * 64 + lane] = acc[b][mt];
    __syncthreads();
    if (NBW == 1 && KZ > 1) {  // the slab / flag exchange (see the producers' side): same two workgroup barriers
      if (!owner) store_units(it, et_all, ET_ALL);
      __syncthreads();
      __syncthreads();
      if (owner) finish_units(it, et_all, ET_ALL);
    } else if (single) {
      finish_units(it, et_all, ET_ALL);
    }
  }
}
