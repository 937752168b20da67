// wna16_gemm.hip — GPTQ/AWQ int4 repack + dequant-fused GEMM entry points (include/vllm_rs_amd.h §A/§B)
// and the dense 16-bit GEMM used for lm_head.  gfx950 only.
#include <stdio.h>

#include "gemm_skinny.cuh"
#include "gemv.cuh"
#include "gemv_q4.cuh"
#include "gemv_q4s.cuh"
#include "gemv_q4w.cuh"
#include "scratch.h"

// ------------------------------------------------------------------------------------------------
// repack: checkpoint layouts -> CDNA4 tile layout (common.cuh). One thread per output word.
// ------------------------------------------------------------------------------------------------
__global__ void gptq_repack_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int K, int N) {
  const int KT = K >> 7;
  const size_t total = (size_t)(K >> 3) * N;
  for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    int j = o & 3, lane = (o >> 2) & 63;
    size_t tile = o >> 8;
    int kt = tile % KT, nb = tile / KT;
    int nn = lane & 15, oct = lane >> 4;
    int n = nb * 16 + nn, krow = kt * 16 + j * 4 + oct;  // GPTQ word row (8 consecutive k)
    uint32_t w = in[(size_t)krow * N + n], r = 0;
#pragma unroll
    for (int p = 0; p < 8; p++) r |= ((w >> (4 * vra_tile_e_of_p(p))) & 0xFu) << (4 * p);
    out[o] = r;
  }
}
__global__ void awq_repack_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int K, int N) {
  const int KT = K >> 7;
  const size_t total = (size_t)(K >> 3) * N;
  for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    int j = o & 3, lane = (o >> 2) & 63;
    size_t tile = o >> 8;
    int kt = tile % KT, nb = tile / KT;
    int nn = lane & 15, oct = lane >> 4;
    int n = nb * 16 + nn, k0 = kt * 128 + j * 32 + oct * 8;
    int sh = 4 * awq_rev(n & 7);
    uint32_t r = 0;
#pragma unroll
    for (int p = 0; p < 8; p++) r |= ((in[(size_t)(k0 + vra_tile_e_of_p(p)) * (N >> 3) + (n >> 3)] >> sh) & 0xFu) << (4 * p);
    out[o] = r;
  }
}
__global__ void unpack_indices_kernel(const uint32_t* __restrict__ tiled, uint8_t* __restrict__ idx, int K, int N) {
  const int KT = K >> 7;
  const size_t total = (size_t)(K >> 3) * N;
  for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    int j = o & 3, lane = (o >> 2) & 63;
    size_t tile = o >> 8;
    int kt = tile % KT, nb = tile / KT;
    int n = nb * 16 + (lane & 15), k0 = kt * 128 + j * 32 + (lane >> 4) * 8;
    uint32_t w = tiled[o];
#pragma unroll
    for (int p = 0; p < 8; p++) idx[(size_t)(k0 + vra_tile_e_of_p(p)) * N + n] = (w >> (4 * p)) & 0xFu;
  }
}
template <class DT>
__global__ void dequant_kernel(const uint32_t* __restrict__ tiled, const uint16_t* __restrict__ scales,
                               const uint32_t* __restrict__ qzeros, uint16_t* __restrict__ wout, int K, int N,
                               int group_size, int is_awq, int layout) {
  const int KT = K >> 7;
  const int g = group_size > 0 ? group_size : K;
  const bool grouped = group_size > 0 && group_size < K;
  const size_t total = (size_t)(K >> 3) * N;
  for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
    int j = o & 3, lane = (o >> 2) & 63;
    size_t tile = o >> 8;
    int kt = tile % KT, nb = tile / KT;
    int n = nb * 16 + (lane & 15), k0 = kt * 128 + j * 32 + (lane >> 4) * 8;
    int grp = k0 / g;
    float s = DT::to_f32(scales[vra_scale_index(grp, n, N, layout, grouped)]);
    float z = 8.f;
    if (is_awq && qzeros) z = (float)((qzeros[(size_t)grp * (N >> 3) + (n >> 3)] >> (4 * awq_rev(n & 7))) & 0xFu);
    s16x8 f = dequant_word<DT>(tiled[o], s, -z * s);
#pragma unroll
    for (int e = 0; e < 8; e++) wout[(size_t)(k0 + e) * N + n] = (uint16_t)f[e];
  }
}

static inline int grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  return (int)(g > 4096 ? 4096 : (g == 0 ? 1 : g));
}

extern "C" void gptq_repack(const void* in, void* out, int32_t rows, int32_t cols, int64_t stream) {
  int K = rows * 8, N = cols;
  VRA_CHECK_ARG(in && out, "gptq_repack: null pointer");
  VRA_CHECK_ARG(K % 128 == 0 && N % 16 == 0, "gptq_repack: need K %% 128 == 0 and N %% 16 == 0 (K=%d N=%d)", K, N);
  size_t total = (size_t)rows * cols;
  gptq_repack_kernel<<<grid_for(total, 256), 256, 0, as_stream(stream)>>>((const uint32_t*)in, (uint32_t*)out, K, N);
}
extern "C" void awq_repack(const void* in, void* out, int32_t rows, int32_t cols, int32_t bits, int64_t stream) {
  int K = rows, N = cols * 8;
  VRA_CHECK_ARG(in && out, "awq_repack: null pointer");
  VRA_CHECK_ARG(bits == 4, "awq_repack: only 4-bit supported (bits=%d)", bits);
  VRA_CHECK_ARG(K % 128 == 0 && N % 16 == 0, "awq_repack: need K %% 128 == 0 and N %% 16 == 0 (K=%d N=%d)", K, N);
  size_t total = (size_t)rows * cols;
  awq_repack_kernel<<<grid_for(total, 256), 256, 0, as_stream(stream)>>>((const uint32_t*)in, (uint32_t*)out, K, N);
}
extern "C" void vra_wna16_unpack_indices(const void* qweight_tiled, uint8_t* idx, int32_t k, int32_t n, int64_t stream) {
  VRA_CHECK_ARG(k % 128 == 0 && n % 16 == 0, "unpack_indices: bad shape K=%d N=%d", k, n);
  size_t total = (size_t)(k / 8) * n;
  unpack_indices_kernel<<<grid_for(total, 256), 256, 0, as_stream(stream)>>>((const uint32_t*)qweight_tiled, idx, k, n);
}
extern "C" void vra_wna16_dequant(const void* qweight_tiled, const void* scales, const void* qzeros, void* w, int32_t k,
                                  int32_t n, int32_t group_size, int32_t is_awq, int32_t scales_layout, int32_t dtype,
                                  int64_t stream) {
  VRA_CHECK_ARG(k % 128 == 0 && n % 16 == 0, "dequant: bad shape K=%d N=%d", k, n);
  size_t total = (size_t)(k / 8) * n;
  if (dtype == VRA_BF16)
    dequant_kernel<BF16><<<grid_for(total, 256), 256, 0, as_stream(stream)>>>((const uint32_t*)qweight_tiled, (const uint16_t*)scales, (const uint32_t*)qzeros, (uint16_t*)w, k, n, group_size, is_awq, scales_layout);
  else if (dtype == VRA_F16)
    dequant_kernel<F16><<<grid_for(total, 256), 256, 0, as_stream(stream)>>>((const uint32_t*)qweight_tiled, (const uint16_t*)scales, (const uint32_t*)qzeros, (uint16_t*)w, k, n, group_size, is_awq, scales_layout);
  else vra_set_error("dequant: dtype must be bf16/f16");
}

// ------------------------------------------------------------------------------------------------
// launch helpers (shared with the native runtime through gemm_launch.h)
// ------------------------------------------------------------------------------------------------
#include "gemm_launch.h"
#include "gemm_dense_launch.h"

static const int kMaxDynLds = 160 * 1024;

static int cur_dev() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev & 63;
}
static bool dev_seen(const uint64_t& mask) { return (mask >> cur_dev()) & 1; }
static void dev_mark(uint64_t& mask) { mask |= (uint64_t)1 << cur_dev(); }
static int num_cus() {
  static int n[64] = {0};
  const int dev = cur_dev();
  if (!n[dev]) {
    if (hipDeviceGetAttribute(&n[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n[dev] <= 0) n[dev] = 256;
  }
  return n[dev];
}

#ifdef VRA_GEMV_TS
static unsigned long long* g_ts = nullptr;
static unsigned long long* vra_gemv_ts_buf() {
  if (!g_ts) {
    (void)hipMalloc(&g_ts, 4096 * 32 * 8);
    (void)hipMemset(g_ts, 0, 4096 * 32 * 8);
  }
  return g_ts;
}
unsigned long long* vra_gemv_ts_buf_shared() { return vra_gemv_ts_buf(); }
extern "C" void vra_debug_ts(unsigned long long* host, int n) { (void)hipMemcpy(host, vra_gemv_ts_buf(), (size_t)n * 8, hipMemcpyDeviceToHost); }
#endif
struct GemvArgs;
static void gemv_debug_args(GemvArgs& a);
template <class DT, bool INT4, int NBW, int SPT, bool AWQ>
static void launch_gemv_v(GemvArgs a, int nblocks, size_t lds, hipStream_t st) {
  static uint64_t attr_devs = 0;  // per device: function attributes belong to the device current at the call
  const bool attr_set = dev_seen(attr_devs);
  static size_t occ_lds = ~(size_t)0;
  static int occ_val = 1;
  auto kern = gemv_kernel<DT, INT4, NBW, SPT, AWQ>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    dev_mark(attr_devs);
  }
  // persistent grid = resident capacity (registers/LDS decide how many workgroups fit a CU), work
  // items split evenly so that there is no ragged last round
  if (occ_lds != lds) {
    int occ = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, GEMV_THREADS, lds);
    occ_val = (e == hipSuccess && occ > 0) ? (occ > 4 ? 4 : occ) : 1;
    occ_lds = lds;
  }
  a.n_items = nblocks;
  gemv_debug_args(a);
  const int cap = num_cus() * occ_val;
  const int per = (nblocks + cap - 1) / cap;
  const int grid = (nblocks + per - 1) / per;
  if (a.am_out && (INT4 || NBW != 1 || a.M > 8 || grid > GEMV_AM_MAX_GRID)) {
    vra_set_error("gemv: the fused argmax serves dense single-tensor launches of <= 8 rows (grid %d)", grid);
    return;
  }
  kern<<<grid, GEMV_THREADS, lds, st>>>(a);
}
static void gemv_debug_args(GemvArgs& a) {
  static const char* exp_env = getenv("VRA_EXP");
  a.dbg = exp_env ? atoi(exp_env) : 0;
  if (a.dbg & 4) a.norm_w = nullptr;
#ifdef VRA_GEMV_TS
  a.ts = vra_gemv_ts_buf();
#else
  a.ts = nullptr;
#endif
}

static int gemv_q4_rows_per_group(int nbw, int M, int K, int group_size);
// int4: persistent grid of (8 compute + 1 epilogue)-wave workgroups, two per CU, when there are at least
// two work items per CU; otherwise one (15 + 1)-wave workgroup per CU (about the same number of waves,
// twice the k-split per item)
template <class DT, int NBW, int SPT, bool AWQ>
static void launch_gemv_q4_v(GemvArgs a, int nblocks, hipStream_t st) {
  static uint64_t attr_devs = 0;  // per device: function attributes belong to the device current at the call
  const bool attr_set = dev_seen(attr_devs);
  static int occ9 = 0;  // resident 9-wave workgroups per CU (VGPR budget of this variant)
  auto kern = gemv_q4_kernel<DT, NBW, SPT, AWQ>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    int occ = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 9 * 64, 48 * 1024);
    occ9 = (e == hipSuccess && occ > 0) ? (occ > 2 ? 2 : occ) : 1;
    dev_mark(attr_devs);
  }
  const int cus = num_cus();
  const int rpg = gemv_q4_rows_per_group(NBW, a.M, a.K, a.group_size);
  const int mg = (a.M + rpg - 1) / rpg;  // row-group families (decode batches up to 32, or long K)
  const int rows = mg > 1 ? rpg : a.M;
  const int oc = rows * (a.K >> 3);
  // shape of the workgroup: two (8+1)-wave workgroups per CU when there is enough work and both fit LDS, else one
  // (15+1)-wave workgroup, shrinking to 14 / 12 compute waves and a single reduction buffer when 16 rows of x
  // leave too little LDS
  static const char* nw_env = getenv("VRA_GEMV_NW");
  int nw = (nblocks * mg >= 2 * cus && occ9 >= 2) ? 8 : 15;
  if (nw_env && (atoi(nw_env) == 8 || atoi(nw_env) == 15)) nw = atoi(nw_env);
  bool single = false;
  const size_t lim = (size_t)kMaxDynLds;
  if (nw == 8 && 2 * gemv_q4_lds_bytes(NBW, 8, rows, a.K, a.group_size) > lim) nw = 15;
  if (nw != 8) {
    const int cand[4] = {15, 15, 14, 12};
    for (int i = 0; i < 4; i++) {
      nw = cand[i];
      single = i > 0;
      if (gemv_q4_lds_bytes(NBW, nw, rows, a.K, a.group_size, single) <= lim) break;
    }
  }
  const size_t lds = gemv_q4_lds_bytes(NBW, nw, rows, a.K, a.group_size, single);
  a.single_red = single ? 1 : 0;
  int per_cu = nw == 8 ? occ9 : 1;
  a.n_items = nblocks;
  a.m_groups = mg;
  a.rows_per_group = rpg;
  gemv_debug_args(a);
  const int cap = cus * per_cu;
  int grid;
  if (mg > 1) {
    int slots = cap / mg / 8 * 8;  // the id -> (slot, family) map works on groups of 8 workgroups (one per XCD)
    if (slots < 8) slots = 8;
    const int need = (nblocks + 7) / 8 * 8;
    if (slots > need) slots = need;
    grid = slots * mg;
  } else {
    grid = nblocks < cap ? nblocks : cap;
  }
  {  // wave-uniform quotients the kernel would otherwise compute with VALU division sequences before its first load
    const int KT = a.K >> 7, nslots = grid / mg, octs = a.K >> 3, nthr = nw * 64;
    a.steps_per_item = (KT + nw - 1) / nw;
    a.items_q = nblocks / nslots;
    a.items_r = nblocks % nslots;
    a.x_chunks = (rows * octs + nthr - 1) / nthr;
    a.octs_shift = (octs & (octs - 1)) == 0 ? 31 - __builtin_clz((unsigned)octs) : -1;
  }
  kern<<<grid, (nw + 1) * 64, lds, st>>>(a);
}
template <class DT, int NBW>
static void launch_gemv_q4_t(const GemvArgs& a, int nblocks, hipStream_t st) {
  const bool fine = a.group_size > 0 && a.group_size < 128;
  if (fine && a.is_awq) launch_gemv_q4_v<DT, NBW, 4, true>(a, nblocks, st);
  else if (fine) launch_gemv_q4_v<DT, NBW, 4, false>(a, nblocks, st);
  else if (a.is_awq) launch_gemv_q4_v<DT, NBW, 1, true>(a, nblocks, st);
  else launch_gemv_q4_v<DT, NBW, 1, false>(a, nblocks, st);
}

// rows of x one workgroup of the int4 kernel can hold (register-staged prologue, LDS image): 0 = none
static int gemv_q4_rows_per_group(int nbw, int M, int K, int group_size) {
  static const char* rows_env = getenv("VRA_GEMV_ROWS");  // tuning aid: cap on the rows per row-group family
  const int cap = rows_env ? atoi(rows_env) : 8;  // measured: 4 families x 8 rows beat 2 x 16 at batch 32
  const int cand[5] = {16, 8, 4, 2, 1};
  for (int i = 0; i < 5; i++) {
    if (cand[i] > cap && cand[i] > 1) continue;
    const int r = M < cand[i] ? M : cand[i];
    if (r * (K >> 3) <= GQ_MAX_CHUNKS * 12 * 64 && gemv_q4_lds_bytes(nbw, 12, r, K, group_size, true) <= (size_t)kMaxDynLds) return r;
  }
  return 0;
}
bool vra_gemv_fits(bool int4, int nbw, int M, int K, int group_size) {
  if (M < 1) return false;
  if (!int4) return M <= 8 && gemv_lds_bytes(false, nbw, M, K, group_size) <= (size_t)72 * 1024;
  static const char* mm_env = getenv("VRA_GEMV_MAX_M");  // tuning aid: largest M routed to the streaming kernel
  if (M > (mm_env ? atoi(mm_env) : 32)) return false;
  if (vra_gemm_q4_fits(nbw, M, K, group_size)) return false;  // 5..32 rows: kernel C dequantises once for all rows
  if (K % 512) return false;  // a wave of the x staging must not straddle rows
  if (group_size > 0 && group_size < K && (group_size & (group_size - 1))) return false;  // power-of-two groups only
  const int rpg = gemv_q4_rows_per_group(nbw, M, K, group_size);
  static const char* mg_env = getenv("VRA_GEMV_MAX_MG");
  return rpg > 0 && (M + rpg - 1) / rpg <= (mg_env ? atoi(mg_env) : 4);  // row-group families each re-do the dequantisation (VALU bound beyond a few)
}

void vra_launch_gemv(const GemvArgs& a, bool int4, int dtype, int64_t stream) {
  hipStream_t st = as_stream(stream);
  int nblocks = 0;
  if (a.silu_dual) nblocks = (a.seg[0].n + 15) / 16;
  else
    for (int s = 0; s < a.nseg; s++) nblocks += (a.seg[s].n + 15) / 16;
  const bool bf = dtype == VRA_BF16;
  if (a.silu_dual && !int4) {  // dense gate/up pair + SiLU*mul (unquantised models: mlp.rs:451-469 in one launch)
    const size_t lds = gemv_lds_bytes(false, 2, a.M, a.K, a.group_size);
    if (bf) launch_gemv_v<BF16, false, 2, 1, false>(a, nblocks, lds, st);
    else launch_gemv_v<F16, false, 2, 1, false>(a, nblocks, lds, st);
  } else if (a.silu_dual) {
    if (bf) launch_gemv_q4_t<BF16, 2>(a, nblocks, st);
    else launch_gemv_q4_t<F16, 2>(a, nblocks, st);
  } else if (int4) {
    if (bf) launch_gemv_q4_t<BF16, 1>(a, nblocks, st);
    else launch_gemv_q4_t<F16, 1>(a, nblocks, st);
  } else {
    const size_t lds = gemv_lds_bytes(false, 1, a.M, a.K, a.group_size);
    if (bf) launch_gemv_v<BF16, false, 1, 1, false>(a, nblocks, lds, st);
    else launch_gemv_v<F16, false, 1, 1, false>(a, nblocks, lds, st);
  }
}

// ---- kernel E (gemv_q4s.cuh): decode batches of 1..4 rows
// distribution of `n_units` units over the workgroups of one launch: grid <= #CUs, the first `r` workgroups own q+1 units.
// A grid that divides the units evenly is preferred when it keeps at least 3/4 of the CUs busy (Llama-3-8B: 384 q/k/v
// units -> 192 x 2, 896 gate/up pairs -> 224 x 4): equal streams end together, which a ragged last unit does not.
//  The grid of kernels E / W never exceeds GW_PRE_PARTS workgroups: the producers of ready-made operands write one slot of
// the [GW_PRE_PARTS][32] partial-sum tables per workgroup (gemv_q4w.cuh) and the consumers add exactly that many.  MI355X has 256 CUs;
// a part with more (MI300X: 304) would otherwise write past the tables.
static int gemv_sw_cus() { return std::min(num_cus(), GW_PRE_PARTS); }
void vra_gemv_s_plan(int n_units, int* grid, int* q, int* r) {
  const int cus = gemv_sw_cus();
  int g = n_units < cus ? n_units : cus;
  static const char* mode = getenv("VRA_GS_GRID");  // tuning aid: "all" = always every CU, ragged
  if (!(mode && mode[0] == 'a')) {
    for (int c = g; c >= (cus * 3) / 4 && c >= 1; c--)
      if (n_units % c == 0) {
        g = c;
        break;
      }
  }
  *grid = g;
  *q = n_units / g;
  *r = n_units % g;
}
// skew of the k-tile shares of kernel E's waves (gemv_q4s.cuh GemvSArgs::skew): one tile more for the four waves that start first, one
// fewer for the four that start last — where the middle waves hold at least two tiles, a wave has at least four tile-steps in all
// (o_proj, one unit of two tiles per workgroup, measured 0.1 us SLOWER with it) and the longest share still fits.  Measured on one
// box (profiles/r05_ab_kernel_e_skew.txt): bs 1 1.709 -> 1.689 ms per step; a skew of two tiles for down_proj (7 tiles per wave): slower.
static int gemv_s_skew(int K, bool norm, int max_units) {
  static const char* e = getenv("VRA_GS_SKEW");  // tuning aid: 0 = equal shares
  const int KT = K / 128, tpw = (KT + 15) / 16;
  if ((e && atoi(e) == 0) || KT / 16 < 2 || max_units * (KT / 16) < 4) return 0;
  return tpw + 1 <= (norm ? GS_NORM_TPW : GS_MAX_TPW) ? 1 : 0;
}
bool vra_gemv_s_fits(int ns, int M, int K, int group_size, int n_units, bool norm) {
  static const char* off = getenv("VRA_NO_GEMV_S");
  if (off && off[0] == '1') return false;
  if (M < 1 || M > 4 || K % 128 || K > 128 * 16 * GS_MAX_TPW) return false;
  const int g = group_size > 0 && group_size < K ? group_size : K;
  if (g < K && (g < 128 || (g & (g - 1)))) return false;  // power-of-two groups of >= one k-tile, or channel-wise
  const int KT = K / 128, tpw0 = (KT + 15) / 16;
  if (norm && tpw0 > GS_NORM_TPW) return false;

  static const char* mu_env = getenv("VRA_GS_MIN_UNITS");  // tuning aid
  // too few n-blocks to spread over the chip without slicing K: kernel A / B.  (A quarter of the CUs, not half, since round 3:
  // the q/k/v launch of a Llama-3-70B TP=8 rank — K = 8192, 80 units — takes 8.7 us here against 12.3 in kernel A.)
  if (n_units < (mu_env ? atoi(mu_env) : num_cus() / 4)) return false;
  int grid, q, r;
  vra_gemv_s_plan(n_units, &grid, &q, &r);
  const int mu = q + (r ? 1 : 0);
  const int tpw = tpw0 + gemv_s_skew(K, norm, mu);  // (the longest share of a wave)
  // four row regions per tile in LDS, or only M of them when four do not fit (K > 16384 at 1..3 rows; single stream only)
  return mu <= GS_MAX_UNITS && (gemv_q4s_lds_bytes(ns, tpw, mu, 4) <= (size_t)kMaxDynLds || (ns == 1 && gemv_q4s_lds_bytes(ns, tpw, mu, M) <= (size_t)kMaxDynLds));
}
// the same predicate for the parity tests: the oracle mirrors "kernel E takes this fused-norm launch" (=> RMSNorm factor applied in
// the epilogue, gemv_q4s.cuh) from the model's shapes
extern "C" int32_t vra_debug_gemv_s_fits(int32_t ns, int32_t m, int32_t k, int32_t group_size, int32_t n_units, int32_t norm) {
  return vra_gemv_s_fits(ns, m, k, group_size, n_units, norm != 0) ? 1 : 0;
}
template <class DT, int NS, bool AWQ, int XR>
static void launch_gemv_s_x(const GemvSArgs& a, int grid, size_t lds, hipStream_t st) {
  static uint64_t attr_devs = 0;
  void (*kern)(const GemvSArgs);
  if constexpr (XR == 4) kern = gemv_q4s_kernel<DT, NS, AWQ>;
  else kern = gemv_q4s_rows_kernel<DT, AWQ, XR>;
  if (!dev_seen(attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    dev_mark(attr_devs);
  }
  kern<<<grid, GS_THREADS, lds, st>>>(a);
}
template <class DT, int NS, bool AWQ>
static void launch_gemv_s_v(GemvSArgs a, hipStream_t st) {
  const int g = a.K / 128;
  a.KT = g;
  int grid;
  vra_gemv_s_plan(a.n_units, &grid, &a.units_q, &a.units_r);
  const int mu = a.units_q + (a.units_r ? 1 : 0);
  a.skew = gemv_s_skew(a.K, a.norm_w != nullptr, mu);
  a.TPW = (a.KT + 15) / 16 + a.skew;
  const int xrows = gemv_q4s_lds_bytes(NS, a.TPW, mu, 4) <= (size_t)kMaxDynLds ? 4 : a.M;  // (vra_gemv_s_fits: one of the two fits)
  const size_t lds = gemv_q4s_lds_bytes(NS, a.TPW, mu, xrows);
  static const char* exp_env = getenv("VRA_EXP");
  a.dbg = exp_env ? atoi(exp_env) : 0;
#ifdef VRA_GEMV_TS
  a.ts = vra_gemv_ts_buf();
#else
  a.ts = nullptr;
#endif
  if (xrows == 4) return launch_gemv_s_x<DT, NS, AWQ, 4>(a, grid, lds, st);
  if constexpr (NS == 1) {  // K > 16384 (no fused norm, no pair there): only M row regions per tile
    if (xrows == 1) return launch_gemv_s_x<DT, NS, AWQ, 1>(a, grid, lds, st);
    if (xrows == 2) return launch_gemv_s_x<DT, NS, AWQ, 2>(a, grid, lds, st);
    if (xrows == 3) return launch_gemv_s_x<DT, NS, AWQ, 3>(a, grid, lds, st);
  }
  vra_set_error("gemv_s: no LDS layout for M=%d K=%d ns=%d", a.M, a.K, NS);
}
void vra_launch_gemv_s(GemvSArgs a, int ns, int group_size, bool awq, int dtype, int64_t stream) {
  hipStream_t st = as_stream(stream);
  const bool grouped = group_size > 0 && group_size < a.K;
  a.gsh = grouped ? 31 - __builtin_clz((unsigned)group_size) : 31;
  if ((a.s_grp_stride | a.s_unit_stride) & 1) {
    vra_set_error("gemv_s: scale strides must be even");
    return;
  }
  const bool bf = dtype == VRA_BF16;
  if (ns == 2) {
    if (awq) bf ? launch_gemv_s_v<BF16, 2, true>(a, st) : launch_gemv_s_v<F16, 2, true>(a, st);
    else bf ? launch_gemv_s_v<BF16, 2, false>(a, st) : launch_gemv_s_v<F16, 2, false>(a, st);
  } else {
    if (awq) bf ? launch_gemv_s_v<BF16, 1, true>(a, st) : launch_gemv_s_v<F16, 1, true>(a, st);
    else bf ? launch_gemv_s_v<BF16, 1, false>(a, st) : launch_gemv_s_v<F16, 1, false>(a, st);
  }
}
// ---- kernel W (gemv_q4w.cuh): 5..32 rows, K <= 4096; 33..256 rows (short prefills) as row blocks of 32
static int gemv_w_max_rows() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VRA_GEMV_W_MAX_ROWS");  // tuning aid: 32 puts the short prefills back on kernels B / D
    v = e ? atoi(e) : 256;
  }
  return v;
}
// unit distribution of a launch: up to 32 rows = kernel E's (one workgroup per CU); above, `rb` row blocks of 32 rows x `grid`
// column groups, about one workgroup per CU in total, at most GW_MAX_UNITS units per workgroup
// (gate/up pairs of 17+ rows run as PSEQ launches, gemv_q4w.cuh: n_units counts pairs, a pair is two units of the workgroup)
static void gemv_w_plan(int ns, int M, int n_units, bool plain, int* grid, int* rb, int* q, int* r) {
  const int rows = 32, max_units = (plain ? GW_MAX_UNITS_PLAIN : GW_MAX_UNITS) / (ns == 2 && M > 16 ? 2 : 1);
  if (M <= rows) {
    *rb = 1;
    vra_gemv_s_plan(n_units, grid, q, r);
    return;
  }
  *rb = (M + rows - 1) / rows;
  int cg = gemv_sw_cus() / *rb;
  if (cg < 1) cg = 1;
  const int need = (n_units + max_units - 1) / max_units;
  if (cg < need) cg = need;
  if (cg > n_units) cg = n_units;
  *grid = cg, *q = n_units / cg, *r = n_units % cg;
}
// K > 4096 (down_proj, 5..32 rows): kz K slices of at most 32 k-tiles (what 8 waves hold as x fragments), kz_groups = CUs / kz
// unit groups; workgroup (z, g) = blockIdx.x z * groups + g — the owners (z = kz - 1) carry the highest ids and start last
// Round 4 (last part): 33..128 rows (down_proj of short prefills) as `rb` row blocks of 32 rows in blockIdx.y, each with its own slabs
// and flags; groups = CUs / (kz * rb), so the whole launch is still one round of workgroups
static int gemv_w_kz_max_rows() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("VRA_GEMV_W_KZ_MAX_ROWS");  // tuning aid: 32 puts down_proj of short prefills back on kernel D
    v = e ? atoi(e) : 160;
  }
  return v;
}
static bool gemv_w_kz_plan(int K, int n_units, int rb, int* kz, int* ktz, int* groups) {
  const int KT = K / 128, per = GW_WAVES * GW_TPW;
  const int z = (KT + per - 1) / per;
  if (z < 2 || z > 8 || rb < 1) return false;
  const int g = gemv_sw_cus() / (z * rb);
  if (g < 1 || n_units < g) return false;
  const int t = (KT + z - 1) / z;
  if ((z - 1) * t >= KT) return false;  // every slice holds at least one tile
  *kz = z, *ktz = t, *groups = g;
  return true;
}
bool vra_gemv_w_fits(int ns, int M, int K, int group_size, int n_units, bool has_res, bool has_bias, bool norm_or_segments) {
  static const char* off = getenv("VRA_NO_GEMV_W");
  if (off && off[0] == '1') return false;
  if (K > 4096) {
    static const char* kz_off = getenv("VRA_GEMV_W_KZ");  // tuning aid: 0 puts down_proj of 5..32 rows back on kernel C
    if ((kz_off && kz_off[0] == '0') || ns != 1 || norm_or_segments || M < 5 || M > std::max(32, gemv_w_kz_max_rows()) || K % 128) return false;
    const int g = group_size > 0 && group_size < K ? group_size : K;
    if (g < K && (g < 128 || (g & (g - 1)))) return false;
    int kz, ktz, groups;
    const int rb = (M + 31) / 32;
    // which row counts: measured against kernel D (+ its row-sum pass) on the Llama-3-8B down projection, us per launch
    // (profiles/r04_ab_kernel_w_k_slices_row_blocks.txt): 64 rows 38.6 -> 25.5, 96 rows 49.3 -> 36.4, 160 rows 77.7 -> 50.9 — but
    // 128 rows 36.4 -> 39.6 and 200 rows 57.0 -> 65.6 (two / four full 64-row tiles per column block are kernel D's good cases)
    static const char* any_rb = getenv("VRA_GEMV_W_KZ_ANY_RB");  // tuning aid: 1 = every row-block count up to the row limit
    if (rb > 3 && rb != 5 && !(any_rb && any_rb[0] == '1')) return false;
    if (!gemv_w_kz_plan(K, n_units, rb, &kz, &ktz, &groups)) return false;
    const int mu = (n_units + groups - 1) / groups, mt = M > 16 ? 2 : 1;
    if (mu > (rb > 1 ? GW_MAX_UNITS_PLAIN : GW_MAX_UNITS)) return false;  // (row blocks: epilogue operands straight from memory, gemv_q4w.cuh)
    if ((size_t)rb * kz * n_units * mt * 256 * 4 > vra_scratch_slab_bytes() - ((size_t)16 << 20) || (size_t)rb * groups * kz * 16 > vra_scratch_counter_count()) return false;
    return gemv_q4w_lds_bytes(1, mt, mu, has_res, true) <= (size_t)kMaxDynLds;
  }
  static const char* pair_env = getenv("VRA_GEMV_W_PAIR_MAX_ROWS");  // tuning aid: 16 puts 17+-row gate/up launches back on kernels C / D
  const int pair_max = pair_env ? atoi(pair_env) : gemv_w_max_rows();
  if (M < 5 || M > (ns == 1 ? gemv_w_max_rows() : (pair_max > 16 ? pair_max : 16)) || K % 128 || K > 4096) return false;
  const int g = group_size > 0 && group_size < K ? group_size : K;
  if (g < K && (g < 128 || (g & (g - 1)))) return false;
  if (n_units < num_cus() / 2) return false;
  const bool plain = !has_res && !has_bias;
  int grid, rb, q, r;
  gemv_w_plan(ns, M, n_units, plain, &grid, &rb, &q, &r);
  const bool pseq = ns == 2 && M > 16;
  if (pseq && !plain) return false;  // (the sequential pair form carries no bias / residual: kernels C / D)
  const int mu = (q + (r ? 1 : 0)) * (pseq ? 2 : 1);
  if (mu > (plain ? GW_MAX_UNITS_PLAIN : GW_MAX_UNITS)) return false;
  // row blocks only while all of them run at once: with a second round of workgroups (q/k/v at 200+ rows: 336..384 workgroups)
  // the launch measured no faster than kernels B / D (tools/short_prefill_gemm_times.py: 48.9 against 51.3 us at 200 rows, 49.8
  // against 41.2 at 256)
  if (rb > 1 && grid * rb > num_cus()) return false;
  return gemv_q4w_lds_bytes(pseq ? 1 : ns, M > 16 ? 2 : 1, mu, has_res) <= (size_t)kMaxDynLds;
}
template <class DT, int NS, int MT, bool AWQ, bool NORM, bool PSEQ = false, bool XF = false>
static void launch_gemv_w_n(GemvSArgs a, hipStream_t st) {
  if constexpr (!XF) {  // fragment-order x (rows <= 32 only): the same launch, other x loads
    if (a.x_frag && a.M <= 32 && a.nseg >= 1) return launch_gemv_w_n<DT, NS, MT, AWQ, NORM, PSEQ, true>(a, st);
  }
  static uint64_t attr_devs = 0;
  auto kern = gemv_q4w_kernel<DT, NS, MT, AWQ, NORM, PSEQ, XF>;
  if (!dev_seen(attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    dev_mark(attr_devs);
  }
  a.KT = a.K / 128;
  a.TPW = GW_TPW;
  int grid, rb;
  const bool plain = !a.residual && !a.seg[0].bias && !(a.nseg > 1 && a.seg[1].bias) && !(a.nseg > 2 && a.seg[2].bias);
  gemv_w_plan(PSEQ ? 2 : NS, a.M, a.n_units, plain, &grid, &rb, &a.units_q, &a.units_r);
  const size_t lds = gemv_q4w_lds_bytes(NS, MT, (a.units_q + (a.units_r ? 1 : 0)) * (PSEQ ? 2 : 1), a.residual != nullptr);
  a.dbg = 0;
#ifdef VRA_GEMV_TS
  a.ts = vra_gemv_ts_buf();
#else
  a.ts = nullptr;
#endif
  if (a.pre_norm_w && grid > GW_PRE_PARTS) {
    vra_set_error("gemv_w: %d workgroups exceed the %d slots of the partial-sum table", grid, GW_PRE_PARTS);
    return;
  }
  kern<<<dim3(grid, rb), GW_THREADS, lds, st>>>(a);
}
template <class DT, int MT, bool AWQ, bool XF>
static void launch_gemv_w_kz(GemvSArgs a, hipStream_t st) {
  static uint64_t attr_devs = 0;
  auto kern = gemv_q4w_kernel<DT, 1, MT, AWQ, false, false, XF, true>;
  if (!dev_seen(attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    dev_mark(attr_devs);
  }
  a.KT = a.K / 128;
  a.TPW = GW_TPW;
  const int rb = (a.M + 31) / 32;
  if (a.norm_w || a.nseg != 1 || !gemv_w_kz_plan(a.K, a.n_units, rb, &a.kz, &a.ktz, &a.kz_groups)) {
    vra_set_error("gemv_w: K = %d needs the K-sliced form (single segment, no fused norm, 2..8 slices)", a.K);
    return;
  }
  a.units_q = a.n_units / a.kz_groups, a.units_r = a.n_units % a.kz_groups;
  a.slabs = vra_scratch_slabs(), a.counters = vra_scratch_counters(), a.err = vra_scratch_error_word();
  if (!a.slabs || !a.counters) {
    vra_set_error("gemv_w: scratch not initialised (vra_scratch_init)");
    return;
  }
  const size_t lds = gemv_q4w_lds_bytes(1, MT, a.units_q + (a.units_r ? 1 : 0), a.residual != nullptr, true);
  a.dbg = 0;
#ifdef VRA_GEMV_TS
  a.ts = vra_gemv_ts_buf();
#else
  a.ts = nullptr;
#endif
  if (a.pre_norm_w && a.kz * a.kz_groups > GW_PRE_PARTS) {
    vra_set_error("gemv_w: %d workgroups exceed the %d slots of the partial-sum table", a.kz * a.kz_groups, GW_PRE_PARTS);
    return;
  }
  kern<<<dim3(a.kz * a.kz_groups, rb), GW_THREADS, lds, st>>>(a);
}
template <class DT, int NS, int MT, bool AWQ>
static void launch_gemv_w_v(const GemvSArgs& a, hipStream_t st) {
  if (a.norm_w) launch_gemv_w_n<DT, NS, MT, AWQ, true>(a, st);
  else launch_gemv_w_n<DT, NS, MT, AWQ, false>(a, st);
}
void vra_launch_gemv_w(GemvSArgs a, int ns, int group_size, bool awq, int dtype, int64_t stream) {
  hipStream_t st = as_stream(stream);
  const bool grouped = group_size > 0 && group_size < a.K;
  a.gsh = grouped ? 31 - __builtin_clz((unsigned)group_size) : 31;
  if ((a.s_grp_stride | a.s_unit_stride) & 1) {
    vra_set_error("gemv_w: scale strides must be even");
    return;
  }
  const bool bf = dtype == VRA_BF16, two = a.M > 16;
  if (a.K > 4096) {  // K slices across workgroups (down_proj)
    if (ns != 1 || a.M > std::max(32, gemv_w_kz_max_rows())) {
      vra_set_error("gemv_w: K = %d takes single-stream launches of up to %d rows", a.K, std::max(32, gemv_w_kz_max_rows()));
      return;
    }
    const bool xf = a.x_frag != nullptr && a.M <= 32;
#define VRA_WKZ(DT_, MT_)                                                                                                                     \
  do {                                                                                                                                        \
    if (awq) xf ? launch_gemv_w_kz<DT_, MT_, true, true>(a, st) : launch_gemv_w_kz<DT_, MT_, true, false>(a, st);                              \
    else xf ? launch_gemv_w_kz<DT_, MT_, false, true>(a, st) : launch_gemv_w_kz<DT_, MT_, false, false>(a, st);                                \
  } while (0)
    if (bf) {
      if (two) VRA_WKZ(BF16, 2);
      else VRA_WKZ(BF16, 1);
    } else {
      if (two) VRA_WKZ(F16, 2);
      else VRA_WKZ(F16, 1);
    }
#undef VRA_WKZ
    return;
  }
  if (ns == 2 && two) {  // gate/up pair of 17+ rows: alternating gate / up units of the single-stream two-m-tile kernel (PSEQ)
    if (!a.norm_w) {
      vra_set_error("gemv_w: the sequential pair form is built with the fused RMSNorm only");
      return;
    }
    if (awq) bf ? launch_gemv_w_n<BF16, 1, 2, true, true, true>(a, st) : launch_gemv_w_n<F16, 1, 2, true, true, true>(a, st);
    else bf ? launch_gemv_w_n<BF16, 1, 2, false, true, true>(a, st) : launch_gemv_w_n<F16, 1, 2, false, true, true>(a, st);
    return;
  }
#define VRA_W(NS_, AWQ_)                                                                                   \
  do {                                                                                                     \
    if (bf) two ? launch_gemv_w_v<BF16, NS_, 2, AWQ_>(a, st) : launch_gemv_w_v<BF16, NS_, 1, AWQ_>(a, st);  \
    else two ? launch_gemv_w_v<F16, NS_, 2, AWQ_>(a, st) : launch_gemv_w_v<F16, NS_, 1, AWQ_>(a, st);       \
  } while (0)
  if (ns == 2) {
    if (awq) VRA_W(2, true);
    else VRA_W(2, false);
  } else {
    if (awq) VRA_W(1, true);
    else VRA_W(1, false);
  }
#undef VRA_W
}
// one-time copies for kernel E: scales [G, N] row-major -> unit-major [N/16][G][16] at unit offset `unit0` of `out`
// (G groups per unit in the destination), AWQ zeros [G, N/8] -> [N/16][G][2]
__global__ void scales_unit_major_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int G, int N, int unit0) {
  const size_t total = (size_t)G * N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i / N), n = (int)(i % N);
    out[((size_t)(unit0 + (n >> 4)) * G + g) * 16 + (n & 15)] = in[i];
  }
}
__global__ void zeros_unit_major_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int G, int N8, int unit0) {
  const size_t total = (size_t)G * N8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i / N8), w = (int)(i % N8);
    out[((size_t)(unit0 + (w >> 1)) * G + g) * 2 + (w & 1)] = in[i];
  }
}
void vra_scales_to_unit_major(const void* scales, void* out, int G, int N, int unit0, int64_t stream) {
  scales_unit_major_kernel<<<grid_for((size_t)G * N, 256), 256, 0, as_stream(stream)>>>((const uint16_t*)scales, (uint16_t*)out, G, N, unit0);
}
void vra_zeros_to_unit_major(const uint32_t* zeros, uint32_t* out, int G, int N, int unit0, int64_t stream) {
  zeros_unit_major_kernel<<<grid_for((size_t)G * (N / 8), 256), 256, 0, as_stream(stream)>>>(zeros, out, G, N / 8, unit0);
}

// columns of the flattened n-block space (all segments)
static inline int skinny_cols(const GemmBArgs& a) { return a.nseg > 1 ? a.xseg[a.nseg - 2].blk_start * 16 + a.xseg[a.nseg - 2].n : a.N; }
template <class DT, bool INT4, bool DUAL, int MT>
static void launch_skinny_t(GemmBArgs a, hipStream_t st) {
  const bool fine = INT4 && a.group_size > 0 && a.group_size < 128;  // several groups per k-tile
  size_t lds = gemm_skinny_lds_bytes(MT, fine ? 4 : 1);
  dim3 grid((skinny_cols(a) + 127) / 128, (a.M + MT * 16 - 1) / (MT * 16), a.splitk);
  // fine scale groups (SPT=4) carry 4x the scale registers: (DUAL, MT>=2) and (single, MT=4) would need > 256
  // VGPRs (spills next to MFMAs) and are never instantiated — vra_launch_skinny caps MT accordingly
  constexpr bool kHasFine = INT4 && !(DUAL && MT >= 2) && !(!DUAL && MT == 4);
  static uint64_t attr_devs = 0;  // per device: function attributes belong to the device current at the call
  const bool attr_set = dev_seen(attr_devs);
  if (!attr_set) {  // MT=4 needs 64 KiB + 16 B of dynamic LDS, just past the default limit
    if constexpr (kHasFine)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<DT, INT4, DUAL, MT, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<DT, INT4, DUAL, MT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    dev_mark(attr_devs);
  }
  if constexpr (kHasFine) {
    if (fine) {
      gemm_skinny_kernel<DT, INT4, DUAL, MT, 4><<<grid, GB_THREADS, lds, st>>>(a);
      return;
    }
  }
  gemm_skinny_kernel<DT, INT4, DUAL, MT, 1><<<grid, GB_THREADS, lds, st>>>(a);
}

// choose split-K so that the grid has roughly >= 2 workgroups per CU, bounded by the slab scratch
static int choose_splitk(int M, int N, int K, int mt, bool dual) {
  int gx = (N + 127) / 128, gy = (M + mt * 16 - 1) / (mt * 16);
  int wg = gx * gy;
  int nchunk = (K + GB_KC - 1) / GB_KC;
  int s = 1;
  // slices only while the tiles (= owners of the exchange, the only workgroups that wait) stay well below the 2 resident
  // workgroups per CU: some non-owner always runs, whatever the dispatch order
  const int cus = num_cus();
  while (wg * s < cus * 3 / 2 && s * 2 <= nchunk && s < 16) s *= 2;
  static const char* force = getenv("VRA_FORCE_SPLITK");  // debugging aid
  if (force) s = atoi(force) < 1 ? 1 : (atoi(force) > nchunk ? nchunk : atoi(force));
  size_t slab = (size_t)(dual ? 2 : 1) * gy * mt * 16 * gx * 128 * 4;
  while (s > 1 && slab * s > vra_scratch_slab_bytes()) s /= 2;
  while (s > 1 && (size_t)wg * s * 16 > vra_scratch_counter_count()) s /= 2;
  return s;
}

void vra_launch_skinny(GemmBArgs a, bool int4, bool dual, int dtype, int64_t stream) {
  hipStream_t st = as_stream(stream);
  int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
  if (int4 && a.group_size > 0 && a.group_size < 128) mt = dual ? 1 : (mt > 2 ? 2 : mt);  // see launch_skinny_t
  a.splitk = choose_splitk(a.M, skinny_cols(a), a.K, mt, dual);
  a.slabs = a.splitk > 1 ? vra_scratch_slabs() : nullptr;
  a.counters = a.splitk > 1 ? vra_scratch_counters() : nullptr;
  a.err = vra_scratch_error_word();
  if (a.splitk > 1 && (!a.slabs || !a.counters)) a.splitk = 1;
  const bool bf = dtype == VRA_BF16;
#define VRA_SK(DT, I4, DU)                                     \
  do {                                                         \
    if (mt == 1) launch_skinny_t<DT, I4, DU, 1>(a, st);        \
    else if (mt == 2) launch_skinny_t<DT, I4, DU, 2>(a, st);   \
    else launch_skinny_t<DT, I4, DU, 4>(a, st);                \
  } while (0)
  if (int4 && dual) {
    if (bf) VRA_SK(BF16, true, true);
    else VRA_SK(F16, true, true);
  } else if (int4) {
    if (bf) VRA_SK(BF16, true, false);
    else VRA_SK(F16, true, false);
  } else {
    if (bf) VRA_SK(BF16, false, false);
    else VRA_SK(F16, false, false);
  }
#undef VRA_SK
}

// ---- kernel D launcher
int vra_gemm_q4_big_fits(bool dual, int M, int cols, int K, int group_size, const GemmDArgs* segs) {
  static const char* off = getenv("VRA_NO_KERNEL_D");  // tuning aid
  if (off && atoi(off)) return 0;
  if (M < 64 || K % 128 || cols % 16) return 0;
  static const char* mb_env = getenv("VRA_GD_MB");  // tuning / test aid: force kernel D with 2 or 4 m-tiles per wave
  if (group_size > 0 && group_size < K && (group_size < 128 || (group_size & (group_size - 1)))) return 0;
  if (segs && segs->nseg > 1) {  // a wave's 4 n-blocks must not straddle tensors
    if (dual || segs->residual || segs->N % 64) return 0;
    for (int i = 0; i + 1 < segs->nseg; i++)
      if (segs->xseg[i].n % 64) return 0;
  }
  // Which of kernel B (slices K, every CU busy on narrow / short problems), kernel D with 64-row tiles (mb 4) or with 32-row
  // tiles (mb 2)?  A small cost model fitted to the Llama-3-8B shapes at M = 128..4096 (tools/gemv_s_microbench.py, us):
  //   kernel D: workgroups are co-resident in pairs (2 per CU, 4 waves each); per K = 4096 a pair of 64-row workgroups takes
  //             82 us and a lone one 47; 32-row workgroups 46 and 28; + 15 us of launch, row-sum pass and epilogue;
  //   kernel B: 17 us + FLOPs at 490 TFLOP/s; the gate/up pair form at 420 (round 5: it was priced at 650 and therefore picked for
  //             ~143..221 rows, where it really takes 110-129 us against kernel D's 85-92: a 200-token prompt was slower than a
  //             256-token one, profiles/r04_probe_kernel_d_tile_choice_160_256_rows.txt).
  // e.g. M = 512: o 50.9 (B) / 39.8 (D, mb 2) / 58.3 (mb 4); down 143 / 117 / 180; gate/up 169 (mb 4) / 189 (mb 2).
  const int gx = dual ? (cols + 127) / 128 : (cols + 255) / 256;
  const int cus = num_cus();
  if (mb_env && (atoi(mb_env) == 2 || atoi(mb_env) == 4) && !getenv("VRA_GD_SPLITK")) return atoi(mb_env);
  const double kf = (double)K / 4096.0;
  auto est_d = [&](int mb) {
    const long w = (long)gx * ((M + 16 * mb - 1) / (16 * mb));
    const long q = (w + cus - 1) / cus;
    const double pair = mb == 4 ? 82.0 : 46.0, lone = mb == 4 ? 47.0 : 28.0;
    // (round 6: the gate/up pair on 32-row tiles runs ~15 % behind this model above 256 rows — 320 rows: 152 us against the 64-row tiles' 142,
    // profiles/r06_llama_midm_gemm_shapes.txt)
    const double pen = dual && mb == 2 && M > 256 ? 1.15 : 1.0;
    return 15.0 + ((double)(q / 2) * pair + (double)(q % 2) * lone) * kf * pen;
  };
  const double flops = 2.0 * (double)M * (double)cols * (dual ? 2.0 : 1.0) * (double)K;
  // (round 6: the rates above were fitted at K = 4096 / 14336.  Where K is not a multiple of 1024 — Qwen2-7B: 3584, 18944 — kernel B measures
  // 195..350 TFLOP/s, tools/gemm_shape_probe.py, profiles/r06_qwen2_gemm_shapes.txt: priced at 490 it took the q/k/v GEMM of a 512-row prefill,
  // 76 us against kernel D's 59, and o_proj at 640 rows, 75 against 58)
  // ... and at the Llama shapes kernel B runs 240..400 TFLOP/s between 257 and 767 rows (profiles/r06_llama_midm_gemm_shapes.txt): priced at 490
  // it took q/k/v at 384 rows (81 us against kernel D's 60), o_proj at 576 / 640 rows (79 against 58) and down_proj at 576 (218 against ~145);
  // at 400 those go to kernel D and the launches where B does win (down_proj at 288 / 320 rows: 120 against 140) stay
  const bool k_odd = (K % 1024) != 0;
  const double est_b = 17.0 + flops / ((dual ? (k_odd ? 215.0 : 345.0) : (k_odd ? 250.0 : 400.0)) * 1e6);
  double best = est_b;
  int pick = 0;
  const double d4 = est_d(4), d2 = est_d(2);
  if (d2 < best) best = d2, pick = 2;
  if (d4 <= best) best = d4, pick = 4;
  // split-K (short prefills: the output tiles alone leave most CUs idle): S slices of >= 4 k-tiles each, one exchange
  // (write-through slabs + flags, ~6 us + the owner's S-1 slab reads) on top of a slice's share of the k loop
  static const char* sk_env = getenv("VRA_GD_SPLITK");  // tuning / test aid: 0 = never, N = force N slices where legal
  const int sk_force = sk_env ? atoi(sk_env) : -1;
  const int KT = K / 128;
  if (vra_scratch_slabs() && vra_scratch_counters() && sk_force != 0) {
    for (int mb = 4; mb >= 2; mb -= 2)
      for (int S = 2; S <= 16; S *= 2) {
        if (KT % S || KT / S < 4) continue;
        const long tiles = (long)gx * ((M + 16 * mb - 1) / (16 * mb));
        if (tiles * S > 2L * cus) continue;  // owners must find their slices resident: the whole grid is co-resident
        if (sk_force > 1) {
          if (S == sk_force) return mb | (S << 8);
          continue;
        }
        const long q = (tiles * S + cus - 1) / cus;
        const double pair = mb == 4 ? 82.0 : 46.0, lone = mb == 4 ? 47.0 : 28.0;
        const double body = ((double)(q / 2) * pair + (double)(q % 2) * lone) * kf / S;
        const double est = 15.0 + 3.0 + body + 6.0 + (mb == 4 ? 0.5 : 0.25) * (S - 1);
        if (est < best) best = est, pick = mb | (S << 8);
      }
  }
  return pick;
}
template <class DT, bool DUAL, int MB>
static void launch_gemm_q4_big_t(const GemmDArgs& a, bool awq, dim3 grid, hipStream_t st) {
  const size_t lds = gemm_q4_big_lds_bytes(MB);
  static uint64_t attr_devs = 0;  // per device: function attributes belong to the device current at the call
  const bool attr_set = dev_seen(attr_devs);
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_q4_big_kernel<DT, DUAL, false, MB>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_q4_big_kernel<DT, DUAL, true, MB>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    dev_mark(attr_devs);
  }
  if (awq) gemm_q4_big_kernel<DT, DUAL, true, MB><<<grid, GD_THREADS, lds, st>>>(a);
  else gemm_q4_big_kernel<DT, DUAL, false, MB><<<grid, GD_THREADS, lds, st>>>(a);
}
static const size_t kXsumBytes = (size_t)16 << 20;  // the row-sum table lives in the LAST 16 MiB of the slab region
float* vra_gemm_q4_big_xsum_table(int M, int K) {
  float* tbl = vra_scratch_slabs() ? vra_scratch_slabs() + (vra_scratch_slab_bytes() - kXsumBytes) / 4 : nullptr;
  return tbl && (size_t)M * (K >> 7) * 4 <= kXsumBytes ? tbl : nullptr;
}
void vra_launch_gemm_q4_big(const GemmDArgs& a0, bool dual, bool awq, int mb_sk, int dtype, int64_t stream) {
  GemmDArgs a = a0;
  const int mb = mb_sk & 0xff, sk = mb_sk >> 8 > 1 ? mb_sk >> 8 : 1;  // vra_gemm_q4_big_fits: m-tiles | split-K << 8
  if (!a.xsum) {  // row sums of x per k-tile, once per GEMM (scratch: the tail of the split-K slab region) — unless the caller's
                  // norm launch already left them there (vra_gemm_q4_big_xsum_table + vra_rms_norm_xsum)
    float* tbl = vra_gemm_q4_big_xsum_table(a.M, a.K);
    const int KT = a.K >> 7;
    if (!tbl) {
      vra_set_error("gemm_q4_big: scratch for the row sums unavailable (%d x %d)", a.M, KT);
      return;
    }
    const int64_t total = (int64_t)a.M * KT * 16;
    if (dtype == VRA_BF16) xsum_rows_kernel<BF16><<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(a.x, a.x_ld, a.M, KT, tbl);
    else xsum_rows_kernel<F16><<<(unsigned)((total + 255) / 256), 256, 0, as_stream(stream)>>>(a.x, a.x_ld, a.M, KT, tbl);
    a.xsum = tbl;
  }
#ifdef VRA_GEMV_TS
  a.ts = vra_gemv_ts_buf();
#endif
  const int cols = dual ? a.N : (a.nseg > 1 ? a.xseg[a.nseg - 2].blk_start * 16 + a.xseg[a.nseg - 2].n : a.N);
  dim3 grid(dual ? (cols + 127) / 128 : (cols + 255) / 256, (a.M + 16 * GD_WM * mb - 1) / (16 * GD_WM * mb), sk);
  a.splitk = sk;
  if (sk > 1) {
    a.slabs = vra_scratch_slabs();
    a.counters = vra_scratch_counters();
    a.err = vra_scratch_error_word();
    const size_t need = (size_t)sk * grid.x * grid.y * GD_NB * mb * GD_THREADS * 16;
    if (!a.slabs || !a.counters || need > vra_scratch_slab_bytes() - kXsumBytes || (size_t)grid.x * grid.y * sk * 16 > vra_scratch_counter_count()) {
      vra_set_error("gemm_q4_big: split-K scratch unavailable");
      return;
    }
  }
  hipStream_t st = as_stream(stream);
  const bool bf = dtype == VRA_BF16;
#define VRA_GD(DU, MBV)                                                   \
  do {                                                                    \
    if (bf) launch_gemm_q4_big_t<BF16, DU, MBV>(a, awq, grid, st);        \
    else launch_gemm_q4_big_t<F16, DU, MBV>(a, awq, grid, st);            \
  } while (0)
  if (dual) {
    if (mb == 4) VRA_GD(true, 4);
    else VRA_GD(true, 2);
  } else {
    if (mb == 4) VRA_GD(false, 4);
    else VRA_GD(false, 2);
  }
#undef VRA_GD
}

// ---- kernel C launcher
bool vra_gemm_q4_fits(int nbw, int M, int K, int group_size) {
  static const char* off = getenv("VRA_NO_KERNEL_C");  // tuning aid
  if (off && atoi(off)) return false;
  // 5..32 rows (measured on Llama-3-8B decode, ms/step, kernel A row groups vs kernel C: bs 4 2.76 / 2.82, bs 6 3.24 / 2.81,
  // bs 8 3.73 / 2.92)
  static const char* mn_env = getenv("VRA_GC_MIN_M");  // tuning aid: fewest rows routed to kernel C
  if (M < (mn_env ? atoi(mn_env) : 5) || M > 32) return false;
  if (group_size > 0 && group_size < K && (group_size < 128 || (group_size & (group_size - 1)))) return false;
  const int KT = K >> 7;
  if (KT % 4) return false;  // x chunks of 512 (4 k-tiles) or 1024
  if (nbw == 1) {  // narrow GEMMs slice K across workgroups (slices of a multiple of 4 k-tiles): e.g. not K = 18944 (148 tiles)
    if (!vra_scratch_slabs() || !vra_scratch_counters()) return false;
    bool sliceable = false;
    for (int z = 2; z <= 16; z++) sliceable = sliceable || (KT % z == 0 && (KT / z) % 4 == 0);
    if (!sliceable) return false;
  }
  return true;
}
template <class DT, int NBW, int MT>
static void launch_gemm_q4_t(const GemmCArgs& a, bool awq, dim3 grid, size_t lds, hipStream_t st) {
  static uint64_t attr_devs = 0;  // per device: function attributes belong to the device current at the call
  const bool attr_set = dev_seen(attr_devs);
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_q4_kernel<DT, NBW, MT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_q4_kernel<DT, NBW, MT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    dev_mark(attr_devs);
  }
  if (awq) gemm_q4_kernel<DT, NBW, MT, true><<<grid, GC_THREADS, lds, st>>>(a);
  else gemm_q4_kernel<DT, NBW, MT, false><<<grid, GC_THREADS, lds, st>>>(a);
}
void vra_launch_gemm_q4(GemmCArgs a, bool awq, int dtype, int64_t stream) {
  const int nbw = a.silu_dual ? 2 : 1;
  const int mt = a.M <= 16 ? 1 : 2;
  const int cus = num_cus();
  const int KT = a.K >> 7;
  int items, ks = 1, kz = 1;
  if (nbw == 2) {
    // wide gate/up pair: n-block-major, k split inside the workgroup until there is ~a workgroup per CU
    a.kc = (KT % 8 == 0) ? 1024 : 512;  // fewer, larger x chunks: the producers' staging is latency bound
    const int tpc = a.kc >> 7;
    while (ks < 8 && ks * 2 <= tpc && (a.n_blocks + (GC_CW / ks) - 1) / (GC_CW / ks) < cus * 3 / 4) ks *= 2;
    items = (a.n_blocks + (GC_CW / ks) - 1) / (GC_CW / ks);
  } else {
    // narrow GEMMs (N = 4096..6144): every wave owns an n-block (CG = 8), K is sliced across workgroups so that each
    // stages only its slice of x; slices of >= 4 tiles, ~a workgroup per CU
    items = (a.n_blocks + GC_CW - 1) / GC_CW;
    for (int z = 2; z <= 16; z++) {
      if (KT % z || (KT / z) % 4) continue;
      if (items * z > cus + cus / 4) break;
      kz = z;
    }
    static const char* kz_env = getenv("VRA_GC_KZ");  // tuning aid
    if (kz_env && atoi(kz_env) >= 1 && KT % atoi(kz_env) == 0 && (KT / atoi(kz_env)) % 4 == 0) kz = atoi(kz_env);
    const int ktz = KT / kz;
    a.kc = (ktz % 8 == 0) ? 1024 : 512;
    static const char* kc_env = getenv("VRA_GC_KC");  // tuning aid
    if (kc_env && (atoi(kc_env) == 512 || (atoi(kc_env) == 1024 && ktz % 8 == 0))) a.kc = atoi(kc_env);
    size_t slab = (size_t)kz * ((a.M + 16 * mt - 1) / (16 * mt)) * items * nbw * (GC_CW * 16 * mt * 2) * 32;  // [slice][row tile][item][tensor][unit] x 32 B
    if (kz > 1 && (slab > vra_scratch_slab_bytes() || (size_t)items * ((a.M + 16 * mt - 1) / (16 * mt)) * kz * 16 > vra_scratch_counter_count())) kz = 1;
    if (items * ((a.M + 16 * mt - 1) / (16 * mt)) >= cus) kz = 1;  // owners (the only workgroups that wait) must stay below the CU count
    if (kz == 1) a.kc = (KT % 8 == 0) ? 1024 : 512;
  }
  a.ks = ks;
  a.kz = kz;
  a.ks_shift = ks == 8 ? 3 : ks == 4 ? 2 : ks == 2 ? 1 : 0;
  a.ktz = KT / kz;
  a.n_items = items;
  a.slabs = kz > 1 ? vra_scratch_slabs() : nullptr;
  a.counters = kz > 1 ? vra_scratch_counters() : nullptr;
  a.err = vra_scratch_error_word();
#ifdef VRA_GEMV_TS
  a.ts = vra_gemv_ts_buf();
#else
  a.ts = nullptr;
#endif
  dim3 grid(items < cus || kz > 1 ? items : cus, (a.M + 16 * mt - 1) / (16 * mt), kz);
  const size_t lds = gemm_q4_lds_bytes(nbw, mt, a.kc);
  hipStream_t st = as_stream(stream);
  const bool bf = dtype == VRA_BF16;
#define VRA_GC(NBWV, MTV)                                                          \
  do {                                                                             \
    if (bf) launch_gemm_q4_t<BF16, NBWV, MTV>(a, awq, grid, lds, st);              \
    else launch_gemm_q4_t<F16, NBWV, MTV>(a, awq, grid, lds, st);                  \
  } while (0)
  if (nbw == 2) {
    if (mt == 1) VRA_GC(2, 1);
    else VRA_GC(2, 2);
  } else {
    if (mt == 1) VRA_GC(1, 1);
    else VRA_GC(1, 2);
  }
#undef VRA_GC
}

// ------------------------------------------------------------------------------------------------
// Marlin-permuted scales (wna16.rs:180-218, what a reference-format caller passes to marlin_*): the GEMM
// kernels read scales row-major only (one load per lane and tile, no index arithmetic in the stream), so a
// permuted tensor is first copied row-major into per-process scratch on the caller's stream.  The native
// runtime keeps its scales row-major and never takes this path.
// ------------------------------------------------------------------------------------------------
__global__ void unpermute_scales_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int G, int N, int grouped) {
  const int64_t total = (int64_t)G * N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int grp = (int)(i / N), n = (int)(i - (int64_t)grp * N);
    out[i] = in[vra_scale_index(grp, n, N, VRA_SCALES_MARLIN, grouped)];
  }
}
static const void* rowmajor_scales(const void* sc, int32_t& layout, int k, int n, int group_size, int which, int64_t stream) {
  if (layout == VRA_SCALES_ROWMAJOR || !sc) return sc;
  const bool grouped = group_size > 0 && group_size < k;
  const int G = grouped ? k / group_size : 1;
  const size_t bytes = (size_t)G * n * 2;
  void* dst = vra_scratch_scales(which);
  if (!dst || bytes > vra_scratch_scale_bytes()) {
    vra_set_error("marlin-permuted scales: scratch unavailable or too small (%zu bytes)", bytes);
    return nullptr;
  }
  unpermute_scales_kernel<<<grid_for((size_t)G * n, 256), 256, 0, as_stream(stream)>>>((const uint16_t*)sc, (uint16_t*)dst, G, n, grouped ? 1 : 0);
  return dst;
}

// ------------------------------------------------------------------------------------------------
// public entry points
// ------------------------------------------------------------------------------------------------
static bool check_gemm_shape(const char* who, int m, int k, int n, int group_size) {
  if (m < 1 || k < 128 || n < 16 || k % 128 || n % 16) {
    vra_set_error("%s: need m>=1, k %% 128 == 0, n %% 16 == 0 (m=%d k=%d n=%d)", who, m, k, n);
    return false;
  }
  if (!(group_size == -1 || (group_size >= 32 && group_size % 32 == 0 && k % group_size == 0))) {
    vra_set_error("%s: group_size must be -1 or a multiple of 32 dividing k (got %d)", who, group_size);
    return false;
  }
  return true;
}

// decode batches of 1..4 rows on kernel E straight from the caller's tensors: row-major scales / zeros as in the
// checkpoint, or the Marlin-permuted scales the reference passes (grouped form: a per-lane index inside 64 columns, no copy)
static bool gemv_s_direct(int ns, const void* in, const void* w0, const void* sc0, const void* qz0, const void* w1, const void* sc1, const void* qz1,
                          const void* bias, const void* residual, void* out, int m, int k, int n, int group_size, int is_awq, int scales_layout,
                          int dtype, int64_t stream, const void* norm_w = nullptr, float eps = 0.f) {
  const bool use_w = n % 16 == 0 && vra_gemv_w_fits(ns, m, k, group_size, n / 16, residual != nullptr, bias != nullptr) &&
                     !(ns == 2 && m > 16 && !norm_w) &&  // (the 17+-row pair form of kernel W exists with the fused norm only)
                     !(k > 4096 && norm_w);              // (K slices do not see whole rows: no fused norm)
  // (this entry point: one output segment)
  if (n % 16 || !(use_w || vra_gemv_s_fits(ns, m, k, group_size, n / 16, norm_w != nullptr))) return false;
  const bool grouped = group_size > 0 && group_size < k;
  if (scales_layout == VRA_SCALES_MARLIN && (!grouped || n % 64)) return false;  // channel-wise permutation: converted copy (rowmajor_scales)
  const bool awq = is_awq != 0 && qz0 != nullptr;
  GemvSArgs a = {};
  a.w[0] = w0, a.scales[0] = sc0, a.zeros[0] = awq ? (const uint32_t*)qz0 : nullptr;
  a.w[1] = w1, a.scales[1] = sc1, a.zeros[1] = awq ? (const uint32_t*)qz1 : nullptr;
  a.s_grp_stride = n, a.s_unit_stride = 16;
  a.z_grp_stride = n / 8, a.z_unit_stride = 2;
  a.marlin = scales_layout == VRA_SCALES_MARLIN ? 1 : 0;
  a.x = in, a.x_ld = k;
  a.norm_w = norm_w, a.eps = eps;
  a.residual = residual, a.res_ld = n;
  a.nseg = ns;
  a.seg[0] = GemvSSeg{out, bias, n, 0};
  a.seg[1] = GemvSSeg{out, nullptr, n, 0x7fffffff};
  a.M = m, a.K = k, a.n_units = n / 16;
  if (use_w) vra_launch_gemv_w(a, ns, group_size, awq, dtype, stream);
  else vra_launch_gemv_s(a, ns, group_size, awq, dtype, stream);
  return true;
}

extern "C" void vra_wna16_gemm(const void* in, const void* qweight_tiled, const void* scales, const void* qzeros,
                               const void* bias, const void* residual, void* out, int32_t m, int32_t k, int32_t n,
                               int32_t group_size, int32_t is_awq, int32_t scales_layout, int32_t dtype,
                               int64_t stream) {
  VRA_CHECK_ARG(in && qweight_tiled && scales && out, "vra_wna16_gemm: null pointer");
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_wna16_gemm: dtype must be bf16/f16");
  if (!check_gemm_shape("vra_wna16_gemm", m, k, n, group_size)) return;
  if (gemv_s_direct(1, in, qweight_tiled, scales, qzeros, nullptr, nullptr, nullptr, bias, residual, out, m, k, n, group_size, is_awq, scales_layout,
                    dtype, stream))
    return;
  // long prefills: the weights dequantised once (Marlin's 16-bit weight, gptq.rs:116-178) + the 256-row dense GEMM (gemm_dense.cuh)
  if (vra_dense_prefill_fits(m, k, n, group_size)) {
    if (void* wd = vra_dense_scratch((size_t)k * n * 2, stream)) {
      vra_launch_dequant_frag(qweight_tiled, scales, qzeros, wd, k, n, group_size, is_awq != 0 && qzeros != nullptr, scales_layout, dtype, 0, 1, stream);
      GemmXArgs a = {};
      a.x = in, a.x_ld = k, a.wd = wd, a.residual = residual, a.res_ld = n;
      a.seg[0] = GemmXSeg{out, bias, n, 0};
      a.nseg = 1;
      a.M = m, a.NV = n, a.K = k;
      vra_launch_gemm_dense(a, false, dtype, vra_gemm_dense_tile(m, n, k), stream);
      return;
    }
  }
  {
    const void* rs = rowmajor_scales(scales, scales_layout, k, n, group_size, 0, stream);
    if (!rs) return;
    scales = rs;
    scales_layout = VRA_SCALES_ROWMAJOR;
  }
  if (vra_gemv_fits(true, 1, m, k, group_size)) {
    GemvArgs a = {};
    a.nseg = 1;
    a.seg[0] = GemvSeg{qweight_tiled, scales, (const uint32_t*)qzeros, bias, out, n, n, 0};
    a.x = in;
    a.x_ld = k;
    a.residual = residual;
    a.res_ld = n;
    a.M = m;
    a.K = k;
    a.group_size = group_size;
    a.is_awq = is_awq;
    a.scales_layout = scales_layout;
    vra_launch_gemv(a, true, dtype, stream);
  } else if (vra_gemm_q4_fits(1, m, k, group_size)) {
    GemmCArgs c = {};
    c.nseg = 1;
    c.seg[0] = GemvSeg{qweight_tiled, scales, (const uint32_t*)qzeros, bias, out, n, n, 0};
    c.x = in;
    c.x_ld = k;
    c.residual = residual;
    c.res_ld = n;
    c.M = m;
    c.K = k;
    c.group_size = group_size;
    c.n_blocks = n / 16;
    vra_launch_gemm_q4(c, is_awq != 0 && qzeros != nullptr, dtype, stream);
  } else if (int mb = vra_gemm_q4_big_fits(false, m, n, k, group_size, nullptr)) {
    GemmDArgs d = {};
    d.w0 = qweight_tiled, d.sc0 = scales, d.qz0 = (const uint32_t*)qzeros, d.bias0 = bias;
    d.x = in, d.x_ld = k, d.residual = residual, d.res_ld = n, d.out = out, d.out_ld = n;
    d.M = m, d.N = n, d.K = k, d.group_size = group_size;
    vra_launch_gemm_q4_big(d, false, is_awq != 0 && qzeros != nullptr, mb, dtype, stream);
  } else {
    GemmBArgs b = {};
    b.w0 = qweight_tiled;
    b.sc0 = scales;
    b.qz0 = (const uint32_t*)qzeros;
    b.bias0 = bias;
    b.x = in;
    b.x_ld = k;
    b.residual = residual;
    b.res_ld = n;
    b.out = out;
    b.out_ld = n;
    b.M = m;
    b.N = n;
    b.K = k;
    b.group_size = group_size;
    b.is_awq = is_awq;
    b.scales_layout = scales_layout;
    vra_launch_skinny(b, true, false, dtype, stream);
  }
}

extern "C" void vra_wna16_gate_up_silu(const void* in, const void* qw_gate, const void* sc_gate, const void* qz_gate,
                                       const void* qw_up, const void* sc_up, const void* qz_up, void* out, int32_t m,
                                       int32_t k, int32_t n, int32_t group_size, int32_t is_awq, int32_t scales_layout,
                                       int32_t dtype, int64_t stream) {
  VRA_CHECK_ARG(in && qw_gate && sc_gate && qw_up && sc_up && out, "vra_wna16_gate_up_silu: null pointer");
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_wna16_gate_up_silu: dtype must be bf16/f16");
  if (!check_gemm_shape("vra_wna16_gate_up_silu", m, k, n, group_size)) return;
  if (gemv_s_direct(2, in, qw_gate, sc_gate, qz_gate, qw_up, sc_up, qz_up, nullptr, nullptr, out, m, k, n, group_size, is_awq, scales_layout, dtype,
                    stream))
    return;
  if (vra_dense_prefill_fits(m, k, 2 * n, group_size)) {
    if (void* wd = vra_dense_scratch((size_t)k * n * 4, stream)) {
      const bool awq = is_awq != 0 && qz_gate != nullptr && qz_up != nullptr;
      const void* tw[2] = {qw_gate, qw_up};
      const void* ts[2] = {sc_gate, sc_up};
      const void* tz[2] = {qz_gate, qz_up};
      const int tn[2] = {n, n}, tf[2] = {0, 1}, tst[2] = {2, 2};
      vra_launch_dequant_frag_batch(2, tw, ts, tz, tn, tf, tst, wd, k, group_size, awq, scales_layout, dtype, stream);  // gate | up interleaved, one launch
      GemmXArgs a = {};
      a.x = in, a.x_ld = k, a.wd = wd;
      a.seg[0] = GemmXSeg{out, nullptr, n, 0};
      a.nseg = 1;
      a.M = m, a.NV = 2 * n, a.K = k;
      vra_launch_gemm_dense(a, true, dtype, vra_gemm_dense_tile(m, 2 * n, k), stream);
      return;
    }
  }
  {
    int32_t l0 = scales_layout, l1 = scales_layout;
    const void* g0 = rowmajor_scales(sc_gate, l0, k, n, group_size, 0, stream);
    const void* u0 = rowmajor_scales(sc_up, l1, k, n, group_size, 1, stream);
    if (!g0 || !u0) return;
    sc_gate = g0;
    sc_up = u0;
    scales_layout = VRA_SCALES_ROWMAJOR;
  }
  if (vra_gemv_fits(true, 2, m, k, group_size)) {
    GemvArgs a = {};
    a.nseg = 2;
    a.seg[0] = GemvSeg{qw_gate, sc_gate, (const uint32_t*)qz_gate, nullptr, out, n, n, 0};
    a.seg[1] = GemvSeg{qw_up, sc_up, (const uint32_t*)qz_up, nullptr, out, n, n, 0};
    a.silu_dual = 1;
    a.x = in;
    a.x_ld = k;
    a.M = m;
    a.K = k;
    a.group_size = group_size;
    a.is_awq = is_awq;
    a.scales_layout = scales_layout;
    vra_launch_gemv(a, true, dtype, stream);
  } else if (vra_gemm_q4_fits(2, m, k, group_size)) {
    GemmCArgs c = {};
    c.nseg = 2;
    c.seg[0] = GemvSeg{qw_gate, sc_gate, (const uint32_t*)qz_gate, nullptr, out, n, n, 0};
    c.seg[1] = GemvSeg{qw_up, sc_up, (const uint32_t*)qz_up, nullptr, out, n, n, 0};
    c.silu_dual = 1;
    c.x = in;
    c.x_ld = k;
    c.M = m;
    c.K = k;
    c.group_size = group_size;
    c.n_blocks = n / 16;
    vra_launch_gemm_q4(c, is_awq != 0 && qz_gate != nullptr, dtype, stream);
  } else if (int mb = vra_gemm_q4_big_fits(true, m, n, k, group_size, nullptr)) {
    GemmDArgs d = {};
    d.w0 = qw_gate, d.w1 = qw_up, d.sc0 = sc_gate, d.sc1 = sc_up, d.qz0 = (const uint32_t*)qz_gate, d.qz1 = (const uint32_t*)qz_up;
    d.x = in, d.x_ld = k, d.out = out, d.out_ld = n;
    d.M = m, d.N = n, d.K = k, d.group_size = group_size;
    vra_launch_gemm_q4_big(d, true, is_awq != 0 && qz_gate != nullptr, mb, dtype, stream);
  } else {
    GemmBArgs b = {};
    b.w0 = qw_gate;
    b.w1 = qw_up;
    b.sc0 = sc_gate;
    b.sc1 = sc_up;
    b.qz0 = (const uint32_t*)qz_gate;
    b.qz1 = (const uint32_t*)qz_up;
    b.x = in;
    b.x_ld = k;
    b.out = out;
    b.out_ld = n;
    b.M = m;
    b.N = n;
    b.K = k;
    b.group_size = group_size;
    b.is_awq = is_awq;
    b.scales_layout = scales_layout;
    vra_launch_skinny(b, true, true, dtype, stream);
  }
}

bool vra_gemv_dw_fits(int M, int K, int N) {
  static const char* off = getenv("VRA_NO_GEMV_DW");
  if (off && off[0] == '1') return false;
  static const char* mn_env = getenv("VRA_GDW_MIN_M");  // tuning aid: fewest rows routed here
  const int min_m = mn_env ? atoi(mn_env) : 4;  // (below: kernel A, which keeps x in LDS; vra_dense_gemm asks kernel A first up to 7 rows)
  if (M < min_m || M > 32 || K % 128 || K > 128 * GW_WAVES * GW_TPW || N % 16) return false;
  const int units = N / 16;
  if (units < num_cus() / 2) return false;
  int grid, q, r;
  vra_gemv_s_plan(units, &grid, &q, &r);
  const int mu = q + (r ? 1 : 0);
  return mu <= GDW_MAX_UNITS && gemv_dw_lds_bytes(M > 16 ? 2 : 1, mu) <= (size_t)kMaxDynLds;
}
template <class DT, int MT, bool NORM>
static void launch_gemv_dw_n(GemvDWArgs a, hipStream_t st) {
  static uint64_t attr_devs = 0;
  auto kern = gemv_dw_kernel<DT, MT, NORM>;
  if (!dev_seen(attr_devs)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynLds);
    dev_mark(attr_devs);
  }
  a.KT = a.K / 128;
  a.n_units = a.n_units > 0 ? a.n_units : 0;
  int grid;
  vra_gemv_s_plan(a.n_units, &grid, &a.units_q, &a.units_r);
  const size_t lds = gemv_dw_lds_bytes(MT, a.units_q + (a.units_r ? 1 : 0));
  if (a.am_out && (!a.am_ws || !a.out_f32 || (size_t)grid * a.M > (size_t)GEMV_AM_COUNTER)) {
    vra_set_error("gemv_dw: fused argmax needs a workspace, f32 logits and grid * M <= %d (grid %d, M %d)", GEMV_AM_COUNTER, grid, a.M);
    return;
  }
  kern<<<grid, GW_THREADS, lds, st>>>(a);
}
void vra_launch_gemv_dw(GemvDWArgs a, int dtype, int64_t stream) {
  hipStream_t st = as_stream(stream);
  const bool two = a.M > 16, norm = a.norm_w != nullptr;
  if (dtype == VRA_BF16) {
    if (two) norm ? launch_gemv_dw_n<BF16, 2, true>(a, st) : launch_gemv_dw_n<BF16, 2, false>(a, st);
    else norm ? launch_gemv_dw_n<BF16, 1, true>(a, st) : launch_gemv_dw_n<BF16, 1, false>(a, st);
  } else {
    if (two) norm ? launch_gemv_dw_n<F16, 2, true>(a, st) : launch_gemv_dw_n<F16, 2, false>(a, st);
    else norm ? launch_gemv_dw_n<F16, 1, true>(a, st) : launch_gemv_dw_n<F16, 1, false>(a, st);
  }
}
extern "C" void vra_dense_gemm(const void* x, const void* w, const void* bias, void* out, int32_t m, int32_t k, int32_t n,
                               int32_t dtype, int32_t out_dtype, int64_t stream) {
  VRA_CHECK_ARG(x && w && out, "vra_dense_gemm: null pointer");
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_dense_gemm: dtype must be bf16/f16");
  VRA_CHECK_ARG(out_dtype == dtype || out_dtype == VRA_F32, "vra_dense_gemm: out_dtype must equal dtype or be f32");
  VRA_CHECK_ARG(m >= 1 && k % 128 == 0 && n % 16 == 0, "vra_dense_gemm: need k %% 128 == 0, n %% 16 == 0 (m=%d k=%d n=%d)", m, k, n);
  if (vra_gemv_fits(false, 1, m, k, -1)) {
    GemvArgs a = {};
    a.nseg = 1;
    a.seg[0] = GemvSeg{w, nullptr, nullptr, bias, out, n, n, 0};
    a.x = x;
    a.x_ld = k;
    a.M = m;
    a.K = k;
    a.group_size = -1;
    a.out_f32 = out_dtype == VRA_F32;
    vra_launch_gemv(a, false, dtype, stream);
  } else if (vra_gemv_dw_fits(m, k, n)) {
    GemvDWArgs a = {};
    a.x = x, a.x_ld = k, a.w = w, a.bias = bias, a.out = out, a.out_ld = n, a.out_f32 = out_dtype == VRA_F32;
    a.M = m, a.K = k, a.n_units = n / 16;
    vra_launch_gemv_dw(a, dtype, stream);
  } else {
    GemmBArgs b = {};
    b.w0 = w;
    b.bias0 = bias;
    b.x = x;
    b.x_ld = k;
    b.out = out;
    b.out_ld = n;
    b.M = m;
    b.N = n;
    b.K = k;
    b.group_size = -1;
    b.out_f32 = out_dtype == VRA_F32;
    vra_launch_skinny(b, false, false, dtype, stream);
  }
}

extern "C" int64_t vra_dense_gemm_argmax_workspace_bytes(void) { return ((int64_t)GEMV_AM_COUNTER + 8) * 8; }
extern "C" void vra_dense_gemm_argmax(const void* x, const void* w, const void* bias, float* logits, uint32_t* tokens, void* workspace,
                                      int32_t m, int32_t k, int32_t n, int32_t dtype, int64_t stream) {
  VRA_CHECK_ARG(x && w && logits && tokens && workspace, "vra_dense_gemm_argmax: null pointer");
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_dense_gemm_argmax: dtype must be bf16/f16");
  VRA_CHECK_ARG(m >= 1 && k % 128 == 0 && n % 16 == 0, "vra_dense_gemm_argmax: need k %% 128 == 0, n %% 16 == 0 (m=%d k=%d n=%d)", m, k, n);
  if (m <= 8 && vra_gemv_fits(false, 1, m, k, -1)) {
    GemvArgs a = {};
    a.nseg = 1;
    a.seg[0] = GemvSeg{w, nullptr, nullptr, bias, logits, n, n, 0};
    a.x = x;
    a.x_ld = k;
    a.M = m;
    a.K = k;
    a.group_size = -1;
    a.out_f32 = 1;
    a.am_out = tokens;
    a.am_ws = static_cast<unsigned long long*>(workspace);
    vra_launch_gemv(a, false, dtype, stream);
    return;
  }
  if (!vra_gemv_fits(false, 1, m, k, -1) && vra_gemv_dw_fits(m, k, n)) {  // 4..32 rows: the dense W kernel, same hand-off
    GemvDWArgs a = {};
    a.x = x, a.x_ld = k, a.w = w, a.bias = bias, a.out = logits, a.out_ld = n, a.out_f32 = 1;
    a.M = m, a.K = k, a.n_units = n / 16;
    a.am_out = tokens, a.am_ws = static_cast<unsigned long long*>(workspace);
    vra_launch_gemv_dw(a, dtype, stream);
    return;
  }
  vra_dense_gemm(x, w, bias, logits, m, k, n, dtype, VRA_F32, stream);
  vra_argmax_f32(logits, tokens, m, n, stream);
}

// ---- Section A: the seven symbols of src/utils/gptq.rs:3-6 ------------------------------------
static void marlin_common(const char* who, const void* in, const int32_t* qweight, const void* scales, const void* qzeros,
                          void* out, int m, int k, int n, int group_size, int is_awq, int dtype, int64_t stream) {
  if (!in || !qweight || !scales || !out) {
    vra_set_error("%s: null pointer", who);
    return;
  }
  vra_wna16_gemm(in, qweight, scales, is_awq ? qzeros : nullptr, nullptr, nullptr, out, m, k, n, group_size, is_awq,
                 VRA_SCALES_MARLIN, dtype, stream);
}
extern "C" void marlin_4bit_bf16(const void* in, const int32_t* qweight, const void* scales, const void* qzeros,
                                 const void* g_idx, void* out, int32_t m, int32_t k, int32_t n, const void* workspace,
                                 int32_t group_size, int64_t stream) {
  (void)g_idx;
  (void)workspace;
  marlin_common("marlin_4bit_bf16", in, qweight, scales, qzeros, out, m, k, n, group_size, 0, VRA_BF16, stream);
}
extern "C" void marlin_4bit_f16(const void* in, const int32_t* qweight, const void* scales, const void* qzeros,
                                const void* g_idx, void* out, int32_t m, int32_t k, int32_t n, const void* workspace,
                                int32_t group_size, int64_t stream) {
  (void)g_idx;
  (void)workspace;
  marlin_common("marlin_4bit_f16", in, qweight, scales, qzeros, out, m, k, n, group_size, 0, VRA_F16, stream);
}
extern "C" void marlin_awq_4bit_bf16(const void* in, const int32_t* qweight, const void* scales, const void* qzeros,
                                     const void* g_idx, void* out, int32_t m, int32_t k, int32_t n, const void* workspace,
                                     int32_t group_size, int64_t stream) {
  (void)g_idx;
  (void)workspace;
  marlin_common("marlin_awq_4bit_bf16", in, qweight, scales, qzeros, out, m, k, n, group_size, 1, VRA_BF16, stream);
}
extern "C" void marlin_awq_4bit_f16(const void* in, const int32_t* qweight, const void* scales, const void* qzeros,
                                    const void* g_idx, void* out, int32_t m, int32_t k, int32_t n, const void* workspace,
                                    int32_t group_size, int64_t stream) {
  (void)g_idx;
  (void)workspace;
  marlin_common("marlin_awq_4bit_f16", in, qweight, scales, qzeros, out, m, k, n, group_size, 1, VRA_F16, stream);
}

// gemm_half_q_half_alt: plain GPTQ checkpoint layout, f16 only (src/utils/gptq.rs:181-198).  Not a
// hot path (sym=false / odd group sizes): one thread block per 64 columns, straight from the
// checkpoint layout, zero = stored+1, g_idx honoured (desc_act), f32 accumulate, one rounding.
// BITS = 4 | 8 (wna16.rs:154-176 routes every non-Marlin checkpoint here, bits included: 32 / BITS values per word along k in
// qweight [k * BITS / 32, n], along n in qzeros [k/g, n * BITS / 32], zero = stored + 1 in both widths)
template <int BITS>
__global__ __launch_bounds__(256) void gptq_alt_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                       const uint32_t* __restrict__ qz, const uint16_t* __restrict__ sc,
                                                       const int32_t* __restrict__ g_idx, uint16_t* __restrict__ out,
                                                       int M, int N, int K, int group_size) {
  // block: 64 columns x 4 k-slices; grid.y = row m
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, ks = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c, m = blockIdx.y;
  float acc = 0.f;
  if (n < N) {
    constexpr int PW = 32 / BITS;  // values per word
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const int rows = K / PW;
    for (int r = ks; r < rows; r += 4) {
      uint32_t w = qw[(size_t)r * N + n];
#pragma unroll
      for (int e = 0; e < PW; e++) {
        int k = r * PW + e;
        int grp = g_idx ? g_idx[k] : k / group_size;
        float s = F16::to_f32(sc[(size_t)grp * N + n]);
        int z = (int)((qz[(size_t)grp * (N / PW) + (n / PW)] >> (BITS * (n % PW))) & MASK) + 1;
        float wv = (float)((int)((w >> (BITS * e)) & MASK) - z) * s;  // exact (q - z)*s, one rounding at the output
        acc += F16::to_f32(x[(size_t)m * K + k]) * wv;
      }
    }
  }
  part[ks][c] = acc;
  __syncthreads();
  if (ks == 0 && n < N) out[(size_t)m * N + n] = F16::from_f32(part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}
extern "C" void gemm_half_q_half_alt(const void* in, const uint32_t* qweight, const uint32_t* qzeros, const void* scales,
                                     const int32_t* g_idx, void* out, int32_t m, int32_t n, int32_t k, int32_t bits,
                                     int64_t stream) {
  VRA_CHECK_ARG(in && qweight && qzeros && scales && out, "gemm_half_q_half_alt: null pointer");
  VRA_CHECK_ARG(bits == 4 || bits == 8, "gemm_half_q_half_alt: 4- or 8-bit GPTQ only (bits=%d)", bits);
  VRA_CHECK_ARG(k % 8 == 0 && n % 8 == 0, "gemm_half_q_half_alt: k,n must be multiples of 8");
  // The reference's signature carries no group size: with g_idx (every GPTQ checkpoint has one, and wna16.rs:127-148 always
  // passes it on this path) the group of a row is g_idx[k].  Without it the group size is read off the EXTENT of the scales
  // allocation ([k/g, n] f16: g = k * n * 2 / bytes), which is only meaningful when `scales` IS its own allocation.  A view into
  // a larger one (candle's layout.start_offset(), gptq.rs:67) would give another group size silently: refused.
  int group = 128;
  if (!g_idx) {
    void* base = nullptr;
    size_t bytes = 0;
    const bool known = hipMemGetAddressRange((hipDeviceptr_t*)&base, &bytes, (hipDeviceptr_t)scales) == hipSuccess;
    if (!known) (void)hipGetLastError();
    size_t groups = 0;
    if (known && base == scales && bytes >= (size_t)n * 2 && bytes % ((size_t)n * 2) == 0) groups = bytes / ((size_t)n * 2);
    if (!groups || groups > (size_t)k || (size_t)k % groups != 0) {
      vra_set_error("gemm_half_q_half_alt: without g_idx the group size comes from the extent of the scales allocation — `scales` must be the base of "
                    "its own [k/g, n] allocation (it is %s); pass g_idx", !known ? "not a device allocation this process knows" : (base != scales ? "a view into a larger one" : "of another size"));
      return;
    }
    group = (int)((size_t)k / groups);
  }
  dim3 grid((n + 63) / 64, m);
  if (bits == 8) gptq_alt_kernel<8><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)in, qweight, qzeros, (const uint16_t*)scales, g_idx, (uint16_t*)out, m, n, k, group);
  else gptq_alt_kernel<4><<<grid, 256, 0, as_stream(stream)>>>((const uint16_t*)in, qweight, qzeros, (const uint16_t*)scales, g_idx, (uint16_t*)out, m, n, k, group);
}

// NormX::forward + QLinear::forward in one call (others.rs:11-29 in front of wna16.rs:263-306): out = rmsnorm(in)·W (+ bias).
// 1..4 rows run fused on kernel E (the normalised activations never reach HBM); otherwise `xn_workspace` [m, k] receives
// rmsnorm(in) and the general GEMM follows — same rounding points either way (the normalised x is rounded to `dtype`).
extern "C" void vra_rms_norm_wna16_gemm(const void* in, const void* norm_weight, float eps, const void* qweight_tiled, const void* scales,
                                        const void* qzeros, const void* bias, void* out, void* xn_workspace, int32_t m, int32_t k, int32_t n,
                                        int32_t group_size, int32_t is_awq, int32_t scales_layout, int32_t dtype, int64_t stream) {
  VRA_CHECK_ARG(in && norm_weight && qweight_tiled && scales && out, "vra_rms_norm_wna16_gemm: null pointer");
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_rms_norm_wna16_gemm: dtype must be bf16/f16");
  if (!check_gemm_shape("vra_rms_norm_wna16_gemm", m, k, n, group_size)) return;
  if (gemv_s_direct(1, in, qweight_tiled, scales, qzeros, nullptr, nullptr, nullptr, bias, nullptr, out, m, k, n, group_size, is_awq, scales_layout, dtype,
                    stream, norm_weight, eps))
    return;
  VRA_CHECK_ARG(xn_workspace != nullptr, "vra_rms_norm_wna16_gemm: this shape needs xn_workspace [m, k]");
  vra_rms_norm(in, norm_weight, xn_workspace, m, k, eps, dtype, stream);
  vra_wna16_gemm(xn_workspace, qweight_tiled, scales, qzeros, bias, nullptr, out, m, k, n, group_size, is_awq, scales_layout, dtype, stream);
}
// NormX + MLP gate/up + SiLU·mul (llama.rs:127-129, mlp.rs:451-469): out = silu(rmsnorm(in)·Wg) * (rmsnorm(in)·Wu)
extern "C" void vra_rms_norm_wna16_gate_up_silu(const void* in, const void* norm_weight, float eps, const void* qw_gate, const void* sc_gate,
                                                const void* qz_gate, const void* qw_up, const void* sc_up, const void* qz_up, void* out,
                                                void* xn_workspace, int32_t m, int32_t k, int32_t n, int32_t group_size, int32_t is_awq,
                                                int32_t scales_layout, int32_t dtype, int64_t stream) {
  VRA_CHECK_ARG(in && norm_weight && qw_gate && sc_gate && qw_up && sc_up && out, "vra_rms_norm_wna16_gate_up_silu: null pointer");
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "vra_rms_norm_wna16_gate_up_silu: dtype must be bf16/f16");
  if (!check_gemm_shape("vra_rms_norm_wna16_gate_up_silu", m, k, n, group_size)) return;
  if (gemv_s_direct(2, in, qw_gate, sc_gate, qz_gate, qw_up, sc_up, qz_up, nullptr, nullptr, out, m, k, n, group_size, is_awq, scales_layout, dtype,
                    stream, norm_weight, eps))
    return;
  VRA_CHECK_ARG(xn_workspace != nullptr, "vra_rms_norm_wna16_gate_up_silu: this shape needs xn_workspace [m, k]");
  vra_rms_norm(in, norm_weight, xn_workspace, m, k, eps, dtype, stream);
  vra_wna16_gate_up_silu(xn_workspace, qw_gate, sc_gate, qz_gate, qw_up, sc_up, qz_up, out, m, k, n, group_size, is_awq, scales_layout, dtype, stream);
}
