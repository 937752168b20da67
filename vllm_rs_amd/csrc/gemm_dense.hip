// gemm_dense.hip — launchers and C entry points of kernel X (gemm_dense.cuh): the dequant pass into fragment order and the
// 256-row dense GEMM the prefill of long prompts runs on (reference call being replaced: src/utils/gptq.rs:116-178, the Marlin GEMM
// of a prefill chunk — same arithmetic: weights rounded to 16 bits, f32 accumulation).
#include "gemm_dense.cuh"

#include <algorithm>

#include "gemm_dense_launch.h"
#include "scratch.h"

static int gx_cur_dev() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev & 63;
}

template <class DT, int BN, bool DUAL>
static void launch_gemm_dense_t(const GemmXArgs& a, hipStream_t st) {
  const size_t lds = gemm_dense_lds_bytes(BN);
  static uint64_t attr_devs = 0;  // per device: function attributes belong to the device current at the call
  if (!((attr_devs >> gx_cur_dev()) & 1)) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dense_kernel<DT, BN, DUAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_devs |= (uint64_t)1 << gx_cur_dev();
  }
  const int tiles = a.MT * a.NT;
  const unsigned grid = a.tail_sk > 1 ? (unsigned)(a.tail_first + (tiles - a.tail_first) * a.tail_sk) : (unsigned)(tiles * (a.splitk > 1 ? a.splitk : 1));
  gemm_dense_kernel<DT, BN, DUAL><<<dim3(grid), GX_THREADS, lds, st>>>(a);
}

static int gx_num_cus() {
  static int n[64] = {0};
  const int dev = gx_cur_dev();
  if (!n[dev] && (hipDeviceGetAttribute(&n[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n[dev] <= 0)) n[dev] = 256;
  return n[dev];
}

// tile width (256 | 128) and split-K factor (<< 16) for an M x nv x K problem
int vra_gemm_dense_tile(int M, int nv, int K) {
  static const char* env = getenv("VRA_GX_BN");  // tuning aid: width | splitk << 16
  if (env && ((atoi(env) & 0xffff) == 128 || (atoi(env) & 0xffff) == 256)) return atoi(env);
  const int cus = gx_num_cus();
  const long mt = (M + GX_BM - 1) / GX_BM;
  const long t256 = mt * ((nv + 255) / 256), t128 = mt * ((nv + 127) / 128);
  // time ~ rounds of `cus` workgroups x tile work.  Measured (tools/gemm_dense_microbench.py --tile): a 128-wide tile does half the work
  // at ~0.8 of the wide tile's rate (o_proj, 4096 rows: 256 wide tiles 107 us, 512 narrow ones 135), and a last round that fills
  // half the chip takes ~0.8 of a full one (q/k/v, 4096 rows: 384 wide tiles 191 us = 1.78 rounds; narrow 202)
  auto rounds = [&](long t) {
    const long full = t / cus, rem = t % cus;
    return (double)full + (rem == 0 ? 0.0 : (rem <= cus / 2 ? 0.8 : 1.0));
  };
  const double c256 = rounds(t256) * 1.0, c128 = rounds(t128) * 0.5 / 0.8;
  int best = c128 < 0.95 * c256 ? 128 : 256;
  double cost = std::min(c128, c256);
  // split-K on the wide tile: S slices of a tile side by side (all co-resident), one exchange of ~8 + 4 (S - 1) us against a wide tile's
  // ~105 us per 4096 of K
  static const char* sk_off = getenv("VRA_GX_NO_SPLITK");
  if (!(sk_off && atoi(sk_off)) && vra_scratch_slabs() && vra_scratch_counters()) {
    const int KS = K / 64;
    const double t_tile = 105.0 * (double)K / 4096.0;
    for (int S = 2; S <= 4; S *= 2) {
      if (t256 * S > cus || KS % (2 * S)) continue;
      if ((size_t)(S - 1) * t256 * 32 * GX_THREADS * 16 > vra_scratch_slab_bytes() / 2 || (size_t)t256 * S * 16 > vra_scratch_counter_count()) continue;
      const double c = 1.0 / S + (8.0 + 4.0 * (S - 1)) / t_tile;
      if (c < 0.92 * cost) cost = c, best = 256 | (S << 16);
    }
    // tail split: whole rounds over the full K, the tiles of a last round that fills at most half the chip split S ways (<< 24)
    static const char* tail_off = getenv("VRA_GX_NO_TAIL_SPLIT");
    const long rem = t256 % cus, full = t256 - rem;
    // (measured, tools/gemm_dense_microbench.py: q/k/v of 3072 rows, 288 tiles, 182 -> 148 us; of 4096 rows, 384 tiles, 189 -> 173;
    // gate/up of 2048 rows — 3.5 rounds — gains nothing: only problems of ONE whole round + a tail take it)
    if (!(tail_off && atoi(tail_off)) && full == cus && rem > 0 && rem <= cus / 2 && cus % 8 == 0) {
      for (int S = 4; S >= 2; S /= 2) {
        if (rem * S > cus || (rem * S) % 8 || KS % (2 * S)) continue;
        if ((size_t)(S - 1) * rem * 32 * GX_THREADS * 16 > vra_scratch_slab_bytes() / 2 || (size_t)rem * S * 16 > vra_scratch_counter_count()) continue;
        const double c = (double)(full / cus) + 1.0 / S + (8.0 + 4.0 * (S - 1)) / t_tile;
        if (c < 0.95 * cost) cost = c, best = 256 | (S << 24);
        break;
      }
    }
  }
  return best;
}

void vra_launch_gemm_dense(GemmXArgs a, bool dual, int dtype, int bn_sk, int64_t stream) {
  const int bn = bn_sk & 0xffff, sk = ((bn_sk >> 16) & 0xff) > 1 ? ((bn_sk >> 16) & 0xff) : 1, tsk = (bn_sk >> 24) > 1 ? (bn_sk >> 24) : 1;
  a.MT = (a.M + GX_BM - 1) / GX_BM;
  a.NT = (a.NV + bn - 1) / bn;
  a.splitk = sk;
  a.tail_sk = 0, a.tail_first = 0;
  if (tsk > 1) {  // (encoded only by vra_gemm_dense_tile, which checked the shape)
    const int cus = gx_num_cus(), tiles = a.MT * a.NT;
    a.tail_first = tiles / cus * cus;
    a.tail_sk = tsk;
    a.splitk = 1;
    if (bn != 256 || a.tail_first <= 0 || a.tail_first >= tiles || (a.tail_first & 7) || ((tiles - a.tail_first) * tsk) % 8 || (a.K / 64) % (2 * tsk)) {
      vra_set_error("gemm_dense: tail split does not fit the shape (tiles=%d, slices=%d)", tiles, tsk);
      return;
    }
  }
  if (sk > 1 || tsk > 1) {
    a.slabs = vra_scratch_slabs();
    a.counters = vra_scratch_counters();
    a.err = vra_scratch_error_word();
    if (!a.slabs || !a.counters || bn != 256 || (a.K / 64) % (2 * std::max(sk, tsk))) {
      vra_set_error("gemm_dense: split-K scratch unavailable or K not divisible (K=%d, slices=%d)", a.K, sk);
      return;
    }
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool bf = dtype == VRA_BF16;
#define VRA_GX(BNV, DU)                                   \
  do {                                                    \
    if (bf) launch_gemm_dense_t<BF16, BNV, DU>(a, st);    \
    else launch_gemm_dense_t<F16, BNV, DU>(a, st);        \
  } while (0)
  if (bn == 256) {
    if (dual) VRA_GX(256, true);
    else VRA_GX(256, false);
  } else {
    if (dual) VRA_GX(128, true);
    else VRA_GX(128, false);
  }
#undef VRA_GX
}

void vra_launch_dequant_frag_batch(int n, const void* const* tiled, const void* const* scales, const void* const* qzeros, const int* N, const int* vfrag0,
                                   const int* vstride, void* wd, int K, int group_size, bool awq, int layout, int dtype, int64_t stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DequantFragBatch b = {};
  size_t most = 0;
  bool zeros = awq;
  for (int i = 0; i < n && i < 3; i++) {
    b.tiled[i] = static_cast<const u32x4*>(tiled[i]), b.scales[i] = static_cast<const uint16_t*>(scales[i]);
    b.qzeros[i] = static_cast<const uint32_t*>(qzeros ? qzeros[i] : nullptr);
    b.N[i] = N[i], b.vfrag0[i] = vfrag0[i], b.vstride[i] = vstride[i];
    most = std::max(most, (size_t)(N[i] >> 4) * (K >> 7) * 64);
    zeros = zeros && b.qzeros[i] != nullptr;
  }
  const dim3 grid((unsigned)((most + 255) / 256), (unsigned)std::min(n, 3));
  const bool bf = dtype == VRA_BF16;
#define VRA_DQ(DT, AW) dequant_frag_kernel<DT, AW><<<grid, 256, 0, st>>>(b, (u32x4*)wd, K, group_size, layout)
  if (bf) {
    if (zeros) VRA_DQ(BF16, true);
    else VRA_DQ(BF16, false);
  } else {
    if (zeros) VRA_DQ(F16, true);
    else VRA_DQ(F16, false);
  }
#undef VRA_DQ
}
void vra_launch_dequant_frag(const void* tiled, const void* scales, const void* qzeros, void* wd, int K, int N, int group_size, bool awq,
                             int layout, int dtype, int vfrag0, int vstride, int64_t stream) {
  vra_launch_dequant_frag_batch(1, &tiled, &scales, &qzeros, &N, &vfrag0, &vstride, wd, K, group_size, awq, layout, dtype, stream);
}

// ---- when the prefill GEMMs take this path, and the scratch tensor the dequantised weights of ONE GEMM live in
// Rows from which dequant pass + dense GEMM beats kernel D (tools/gemm_dense_microbench.py, Llama-3-8B shapes, one layer's four GEMMs,
// us: 512 rows 376 (D) / 420 (X + pass); 768: 570 / 523; 1024: 661 / 573; 2048: 1179 / 875; 4096: 2290 / 1500; 8192: 4377 / 2845 —
// profiles/r06_gemm_dense_microbench.txt).  VRA_DENSE_PREFILL_MIN_ROWS overrides (0 = never); tests lower it through
// vra_debug_set_dense_prefill_min_rows to reach the path with small models.
static int g_dense_min_rows = -1;
int vra_dense_prefill_min_rows() {
  if (g_dense_min_rows < 0) {
    const char* e = getenv("VRA_DENSE_PREFILL_MIN_ROWS");
    g_dense_min_rows = e ? std::max(0, atoi(e)) : 768;
  }
  return g_dense_min_rows;
}
extern "C" void vra_debug_set_dense_prefill_min_rows(int32_t rows) { g_dense_min_rows = rows < 0 ? 0 : rows; }
extern "C" int32_t vra_debug_dense_prefill_min_rows(void) { return vra_dense_prefill_min_rows(); }

bool vra_dense_prefill_fits(int M, int K, int nv, int group_size) {
  const int mr = vra_dense_prefill_min_rows();
  if (mr <= 0 || M < mr || K % 128 || nv % 16) return false;
  if (!(group_size <= 0 || (group_size % 8 == 0 && K % group_size == 0))) return false;
  // (buffer resources of 0x7FFFFFF0 bytes: the dequantised tensor and — at the engine's 16 384-row cap — x stay below that)
  return (uint64_t)K * (uint64_t)nv * 2u < 0x7FFFFFF0ull && (uint64_t)M * (uint64_t)K * 2u < 0x7FFFFFF0ull;
}

// One region per device, grown on demand (never inside a stream capture: prefill steps are not captured, and a capturing caller
// gets null and takes the int4 kernels).  Launches that use it are ordered by the caller's stream; a second stream using the dense
// path at the same time would need its own region — the engine and the FFI entry points run one stream per process.
static void* g_dense_scr[64] = {};
static size_t g_dense_scr_bytes[64] = {};
void* vra_dense_scratch(size_t bytes, int64_t stream) {
  const int dev = gx_cur_dev();
  if (g_dense_scr_bytes[dev] >= bytes) return g_dense_scr[dev];
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(reinterpret_cast<hipStream_t>(stream), &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;
  }
  (void)hipDeviceSynchronize();  // the old region may still be read by launches in flight
  if (g_dense_scr[dev]) (void)hipFree(g_dense_scr[dev]);
  g_dense_scr[dev] = nullptr, g_dense_scr_bytes[dev] = 0;
  const size_t want = (bytes + ((size_t)16 << 20) - 1) & ~(((size_t)16 << 20) - 1);
  void* p = nullptr;
  if (hipMalloc(&p, want) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  g_dense_scr[dev] = p, g_dense_scr_bytes[dev] = want;
  return p;
}

// ---- C entry points (tests, microbenchmarks; the engine calls the launchers)
extern "C" void vra_wna16_dequant_frag(const void* qweight_tiled, const void* scales, const void* qzeros, void* wd, int32_t k, int32_t n,
                                       int32_t group_size, int32_t is_awq, int32_t scales_layout, int32_t dtype, int32_t vfrag0,
                                       int32_t vstride, int64_t stream) {
  VRA_CHECK_ARG(qweight_tiled && scales && wd, "dequant_frag: null pointer");
  VRA_CHECK_ARG(k % 128 == 0 && n % 16 == 0, "dequant_frag: bad shape K=%d N=%d", k, n);
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "dequant_frag: dtype must be bf16/f16");
  VRA_CHECK_ARG(group_size <= 0 || (group_size % 8 == 0 && k % group_size == 0), "dequant_frag: bad group size %d", group_size);
  VRA_CHECK_ARG(vstride >= 1 && vfrag0 >= 0, "dequant_frag: bad fragment placement");
  vra_launch_dequant_frag(qweight_tiled, scales, qzeros, wd, k, n, group_size, is_awq != 0, scales_layout, dtype, vfrag0, vstride, stream);
}
extern "C" void vra_dense_frag_gemm(const void* x, const void* wd, const void* bias, const void* residual, void* out, int32_t m, int32_t k,
                                    int32_t nv, int32_t dual, int32_t dtype, int32_t tile_n, int64_t stream) {
  VRA_CHECK_ARG(x && wd && out, "dense_frag_gemm: null pointer");
  VRA_CHECK_ARG(m >= 1 && k % 128 == 0 && nv % 16 == 0 && (!dual || nv % 32 == 0), "dense_frag_gemm: bad shape M=%d K=%d NV=%d", m, k, nv);
  VRA_CHECK_ARG(dtype == VRA_BF16 || dtype == VRA_F16, "dense_frag_gemm: dtype must be bf16/f16");
  VRA_CHECK_ARG(!(dual && bias), "dense_frag_gemm: the gate/up form takes no bias");
  VRA_CHECK_ARG(tile_n == 0 || (tile_n & 0xffff) == 128 || (tile_n & 0xffff) == 256, "dense_frag_gemm: tile_n must be 0 (auto), 128 or 256 (| split-K slices << 16 | tail slices << 24)");
  if (tile_n >> 16 > 1) vra_scratch_init();
  GemmXArgs a{};
  a.x = x, a.x_ld = k, a.wd = wd, a.residual = residual;
  const int n_out = dual ? nv / 2 : nv;
  a.res_ld = n_out;
  a.seg[0] = GemmXSeg{out, bias, n_out, 0};
  a.nseg = 1;
  a.M = m, a.NV = nv, a.K = k;
  vra_launch_gemm_dense(a, dual != 0, dtype, tile_n ? tile_n : vra_gemm_dense_tile(m, nv, k), stream);
}
