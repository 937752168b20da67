// overlap_probe — does a chain of dependent decode-sized kernels run faster when consecutive kernels are CO-RESIDENT
// (alternating over two streams, the successor launched early, prefetching its weights and polling data-tagged
// granules of its predecessor) than as ordinary dependent launches on one stream?  And does a hipGraph captured from
// the two streams keep the concurrency?  (numbers in profiles/r04_overlap_probe.txt)
//
// Every stage: 256 workgroups x 16 waves.  stamp START | request 2 KiB of weights per wave | sweep the predecessor's
// output (n_gran 8-byte {2 x bf16, tag} granules, sc1 loads, retried until every tag matches) into LDS | barrier |
// stamp READY | stream `steps` more KiB per wave | barrier | publish own n_gran granules (sc1 stores, spread over the
// workgroups) | stamp END.
//   build: hipcc -O3 --offload-arch=gfx950 -o tools/overlap_probe tools/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

struct StageArgs {
  const void* gin;   // granules of the predecessor
  void* gout;        // own granules
  const u32x4* w;    // weights of this stage: [256 wg][steps+2][16 waves][64 lanes] u32x4
  unsigned long long* ts;  // [wg][4]: start, ready, end, polls
  uint32_t* err;
  uint32_t* sink;
  const uint32_t* epoch;  // device word, bumped by the last kernel of a chain: tags are (epoch*1024 + stage), so a captured
                          // graph can be replayed
  int seq;
  int n_gran;  // granules per edge (2048 = a 4096-wide bf16 row)
  int steps;   // KiB per wave streamed after READY
  int polite;  // > 0: s_sleep argument of the single watcher wave (64 cycles each)
  int early;   // 1: weights requested before the sweep (co-resident mode); 0: after it (as the serial kernels do)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

__global__ __launch_bounds__(1024) void stage_kernel(const StageArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
  const unsigned long long t_start = wall_clock64();
  const uint32_t ep = __builtin_amdgcn_readfirstlane(*a.epoch);
  const uint32_t tag_in = ep * 1024u + (uint32_t)a.seq, tag_out = tag_in + 1u;
  if (tid == 0) a.ts[wg * 4 + 0] = t_start;
  const u32x4* wp = a.w + ((size_t)wg * (a.steps + 2) * 16 + wave) * 64 + lane;
  u32x4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0};
  if (a.early) {
    r0 = __builtin_nontemporal_load(wp);
    r1 = __builtin_nontemporal_load(wp + 16 * 64);
  }
  // ---- sweep: wave w takes granules [w*128, w*128+128) (two per lane) of every 2048; all 16 waves cover one row
  const auto gr = rsrc(a.gin, (uint32_t)a.n_gran * 8u);
  uint32_t* xl = reinterpret_cast<uint32_t*>(smem);
  unsigned polls = 0;
  const unsigned long long t_lim = wall_clock64() + 200000000ull;  // 2 s
  if (a.polite && a.seq > 0) {
    // ONE wave watches ONE granule pair with long sleeps (16 poller waves next to the predecessor's weight stream doubled its
    // time); the other 15 wait at the barrier and sweep once the watched pair has flipped
    if (wave == 0) {
      for (;;) {
        const u32x2 g0 = __builtin_amdgcn_raw_buffer_load_b64(gr, (uint32_t)lane * 8u, 0, 16);  // sc1
        polls++;
        if (__all(g0[1] == tag_in) || wall_clock64() > t_lim) break;
        if (a.polite >= 48) __builtin_amdgcn_s_sleep(60);
        else if (a.polite >= 16) __builtin_amdgcn_s_sleep(24);
        else __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
  }
  for (int base = wave * 128; base < a.n_gran; base += 2048) {
    for (;;) {
      const u32x2 g0 = __builtin_amdgcn_raw_buffer_load_b64(gr, (uint32_t)(base + lane) * 8u, 0, 16);       // sc1
      const u32x2 g1 = __builtin_amdgcn_raw_buffer_load_b64(gr, (uint32_t)(base + 64 + lane) * 8u, 0, 16);  // sc1
      const bool ok = a.seq == 0 || (g0[1] == tag_in && g1[1] == tag_in);
      polls++;
      if (__all(ok)) {
        xl[base + lane] = g0[0];
        xl[base + 64 + lane] = g1[0];
        break;
      }
      if (wall_clock64() > t_lim) {
        if (lane == 0) atomicExch(a.err, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  if (!a.early) {
    r0 = __builtin_nontemporal_load(wp);
    r1 = __builtin_nontemporal_load(wp + 16 * 64);
  }
  __syncthreads();
  if (tid == 0) {
    a.ts[wg * 4 + 1] = wall_clock64();
    a.ts[wg * 4 + 3] = polls;
  }
  // ---- stream: `steps` more KiB per wave, 2 in flight
  uint32_t acc = xl[(tid * 2) % a.n_gran];
  for (int s = 0; s < a.steps; s += 2) {
    acc ^= r0[0] ^ r0[1] ^ r0[2] ^ r0[3];
    r0 = __builtin_nontemporal_load(wp + (size_t)(s + 2) * 16 * 64);
    acc ^= r1[0] ^ r1[1] ^ r1[2] ^ r1[3];
    r1 = __builtin_nontemporal_load(wp + (size_t)(s + 3) * 16 * 64);
  }
  acc ^= r0[0] ^ r0[1] ^ r0[2] ^ r0[3] ^ r1[0] ^ r1[1] ^ r1[2] ^ r1[3];
  uint32_t* red = reinterpret_cast<uint32_t*>(smem + 16384);
  red[tid] = acc;
  __syncthreads();
  // ---- publish: workgroup wg owns granules [wg*n_gran/256, ...): one 8-byte sc1 store per granule
  const int per = a.n_gran / (int)gridDim.x;
  if (tid < per) {
    uint32_t v = 0;
    for (int w = 0; w < 16; w++) v ^= red[w * 64 + tid];
    const auto go = rsrc(a.gout, (uint32_t)a.n_gran * 8u);
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{v | 1u, tag_out}, go, (uint32_t)(wg * per + tid) * 8u, 0, 16);  // sc1
  }
  if (acc == 0x12345678u) a.sink[0] = acc;
  if (tid == 0) a.ts[wg * 4 + 2] = wall_clock64();
}

__global__ void bump_kernel(uint32_t* e) { *e += 1u; }
static double ticks_us(unsigned long long t) { return (double)t / 100.0; }  // s_memrealtime: 100 MHz

int main(int argc, char** argv) {
  const int N = 40, WG = 256;
  int n_gran = argc > 1 ? atoi(argv[1]) : 2048;
  int lds = argc > 2 ? atoi(argv[2]) : 40960;
  int polite = argc > 3 ? atoi(argv[3]) : 0;
  const int step_list[5] = {2, 0, 2, 14, 7};  // q/k/v, attention (no stream), o_proj, gate/up, down: KiB per wave after READY
  size_t wbytes_stage = (size_t)WG * 16 * 16 * 1024;  // 64 MiB per stage slot
  u32x4* w;
  CK(hipMalloc(&w, wbytes_stage * 8));
  CK(hipMemset(w, 1, wbytes_stage * 8));
  void* g[3];
  for (int i = 0; i < 3; i++) {
    CK(hipMalloc(&g[i], (size_t)n_gran * 8));
    CK(hipMemset(g[i], 0, (size_t)n_gran * 8));
  }
  unsigned long long* ts;
  CK(hipMalloc(&ts, (size_t)N * WG * 4 * 8));
  uint32_t *err, *sink;
  CK(hipMalloc(&err, 4));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(err, 0, 4));
  CK(hipFuncSetAttribute((const void*)stage_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t ef, ej, t0, t1;
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  CK(hipEventCreate(&t0));
  CK(hipEventCreate(&t1));
  uint32_t epoch = 1;
  uint32_t* d_epoch;
  CK(hipMalloc(&d_epoch, 4));
  CK(hipMemcpy(d_epoch, &epoch, 4, hipMemcpyHostToDevice));

  auto enqueue = [&](int mode, uint32_t ep) {  // mode 0: one stream, plain order; 1: two streams alternating
    if (mode == 1) {
      CK(hipEventRecord(ef, s0));
      CK(hipStreamWaitEvent(s1, ef, 0));
    }
    for (int i = 0; i < N; i++) {
      StageArgs a;
      a.gin = g[(i + 2) % 3];
      a.gout = g[i % 3];
      a.w = w + ((size_t)(i % 8) * wbytes_stage) / 16;
      a.ts = ts + (size_t)i * WG * 4;
      a.err = err;
      a.sink = sink;
      a.epoch = d_epoch;
      a.seq = i;
      a.n_gran = n_gran;
      a.steps = step_list[i % 5];
      a.early = mode;
      a.polite = mode ? polite : 0;
      hipStream_t st = (mode == 1 && (i & 1)) ? s1 : s0;
      hipLaunchKernelGGL(stage_kernel, dim3(WG), dim3(1024), lds, st, a);
    }
    if (mode == 1) {
      CK(hipEventRecord(ej, s1));
      CK(hipStreamWaitEvent(s0, ej, 0));
    }
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, s0, d_epoch);
  };
  auto report = [&](const char* name, float ms) {
    std::vector<unsigned long long> h((size_t)N * WG * 4);
    CK(hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost));
    uint32_t e;
    CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    std::vector<double> st_min(N), st_max(N), rd_max(N), en_max(N), en_min(N), polls(N);
    for (int i = 0; i < N; i++) {
      unsigned long long a0 = ~0ull, a1 = 0, b1 = 0, c1 = 0, c0 = ~0ull, p = 0;
      for (int wgi = 0; wgi < WG; wgi++) {
        const unsigned long long* r = &h[((size_t)i * WG + wgi) * 4];
        a0 = std::min(a0, r[0]), a1 = std::max(a1, r[0]), b1 = std::max(b1, r[1]), c1 = std::max(c1, r[2]), c0 = std::min(c0, r[2]);
        p = std::max(p, r[3]);
      }
      st_min[i] = ticks_us(a0), st_max[i] = ticks_us(a1), rd_max[i] = ticks_us(b1), en_max[i] = ticks_us(c1), en_min[i] = ticks_us(c0), polls[i] = (double)p;
    }
    const double T0 = st_min[0];
    printf("== %s: host-timed %.1f us per chain of %d = %.2f us per stage; device first start -> last end %.1f us = %.2f per stage; err %u\n", name,
           ms * 1e3, N, ms * 1e3 / N, en_max[N - 1] - T0, (en_max[N - 1] - T0) / N, e);
    double ho = 0, ov = 0, span = 0;
    int cnt = 0;
    for (int i = 5; i < N; i++) {
      ho += rd_max[i] - en_max[i - 1];
      ov += en_max[i - 1] - st_max[i];
      span += en_max[i] - en_max[i - 1];
      cnt++;
    }
    printf("   mean over stages 5..: last store of predecessor -> all consumers READY %.2f us; successor's last start BEFORE predecessor's end by %.2f us; end-to-end period %.2f us\n",
           ho / cnt, ov / cnt, span / cnt);
    for (int i = 10; i < 20; i++)
      printf("   stage %2d (steps %2d): start %7.2f..%7.2f ready<=%7.2f end %7.2f..%7.2f  (pred end %7.2f) max polls %.0f\n", i, step_list[i % 5], st_min[i] - T0,
             st_max[i] - T0, rd_max[i] - T0, en_min[i] - T0, en_max[i] - T0, en_max[i - 1] - T0, polls[i]);
  };
  auto timed = [&](const char* name, auto&& fn) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; rep++) {
      CK(hipEventRecord(t0, s0));
      fn(epoch);
      CK(hipEventRecord(t1, s0));
      CK(hipEventSynchronize(t1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, t0, t1));
      best = std::min(best, ms);
      epoch++;
      if (rep == 5) report(name, ms);
    }
    printf("   best of 6: %.2f us per stage\n", best * 1e3 / N);
  };

  printf("# overlap_probe: %d stages x %d workgroups x 1024 threads, %d granules per edge, %d B dynamic LDS, watcher sleep %d\n", N, WG, n_gran, lds, polite);
  timed("serial eager (one stream, weights requested after x)", [&](uint32_t ep) { enqueue(0, ep); });
  timed("two streams eager (successor co-resident, weights requested before x)", [&](uint32_t ep) { enqueue(1, ep); });
  for (int mode = 0; mode < 2; mode++) {
    hipGraph_t gr;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
    enqueue(mode, 0);
    CK(hipStreamEndCapture(s0, &gr));
    CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    timed(mode ? "two streams captured into ONE hipGraph, replayed" : "serial captured into a hipGraph, replayed", [&](uint32_t) { CK(hipGraphLaunch(ge, s0)); });
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(gr));
  }
  return 0;
}
