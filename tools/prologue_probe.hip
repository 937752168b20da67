// prologue_probe — where does the time between "kernel N ended" and "kernel N+1 has its first weight tile" go, for the geometry of
// kernel E (gemv_q4s.cuh: one 16-wave workgroup per CU, 2 KiB of weights in flight per wave, 8 KB of x written by the previous
// launch)?  A chain producer -> consumer is captured in a hipGraph and replayed; every wave of the consumer stamps
//   t0 first instruction | t1 kernel arguments in SGPRs | t2 x landed | t3 first weight tile landed | t4 stream done
// with s_memrealtime (100 MHz).  MODE picks the order of the prologue:
//   0  args -> x (vector loads) -> wait -> weight ring            (what kernel E does)
//   1  args -> x (vector), ring right behind, no wait between
//   2  args -> ring, x through the SCALAR cache (s_load_dwordx16 x 8 = the wave's 512 B of one row)
//   3  args -> ring, then x (vector)
//   4  as 0 with 8-wave workgroups and 4 KiB in flight per wave
//   5  as 2 with 8-wave workgroups and 4 KiB in flight per wave
//   hipcc -O3 --offload-arch=gfx950 -o tools/prologue_probe tools/prologue_probe.hip
// Run with HIP_FORCE_DEV_KERNARG=0 / 1 to see what the placement of the argument block costs (t1 - t0).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

struct Args {  // the size of GemvSArgs: the argument block is fetched by scalar loads at the top of the kernel
  const uint32_t* w;
  const uint16_t* x;
  uint16_t* out;
  unsigned long long* ts;
  uint32_t* sink;
  int steps, K, pad0;
  long long filler[32];
};

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memrealtime(); }

__global__ __launch_bounds__(1024) void producer(Args a) {  // writes x (8 KB) like the epilogue of the previous GEMV
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.K) a.out[i] = (uint16_t)(0x3c00 + (i & 63));
  if (threadIdx.x == 0) a.ts[(size_t)blockIdx.x] = now();
}

template <int MODE, int WAVES, int RING>
__global__ __launch_bounds__(WAVES * 64) void consumer(Args a) {
  const unsigned long long t0 = now();
  __builtin_amdgcn_sched_barrier(0);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int steps = a.steps;  // forces the argument block
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(steps), "s"(a.w), "s"(a.x) : "memory");
  const unsigned long long t1 = now();
  const u32x4* wp = reinterpret_cast<const u32x4*>(a.w) + ((size_t)(blockIdx.x * WAVES + wave) * (steps + RING)) * 64 + lane;
  u32x4 ring[RING];
  auto fill = [&]() {
#pragma unroll
    for (int r = 0; r < RING; r++) ring[r] = __builtin_nontemporal_load(wp + (size_t)r * 64);
  };
  // the wave's slice of x: 2 k-tiles x 128 columns of one row = 512 B
  const uint16_t* xw = a.x + (size_t)wave * 256;
  uint32_t xacc = 0;
  unsigned long long t2, t3;
  if (MODE == 0 || MODE == 4) {
    const u32x4 xv = *reinterpret_cast<const u32x4*>(xw + (lane & 31) * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(xv) : "memory");
    t2 = now();
    xacc = xv[0] ^ xv[3];
    fill();
  } else if (MODE == 1) {
    const u32x4 xv = *reinterpret_cast<const u32x4*>(xw + (lane & 31) * 8);
    fill();
    asm volatile("s_waitcnt vmcnt(%1)" ::"v"(xv), "n"(RING) : "memory");
    t2 = now();
    xacc = xv[0] ^ xv[3];
  } else if (MODE == 2 || MODE == 5) {
    fill();
    // (one k-tile of one row = 256 B = 64 SGPRs at a time: two dependent scalar round trips, the conservative case)
    u32x16 s[4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int i = 0; i < 4; i++) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(s[i]) : "s"(xw + h * 128), "n"(i * 64) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; i++) asm volatile("" ::"s"(s[i]));
      xacc ^= s[0][0] ^ s[3][15];
    }
    t2 = now();
  } else {
    fill();
    const u32x4 xv = *reinterpret_cast<const u32x4*>(xw + (lane & 31) * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::"v"(xv) : "memory");
    t2 = now();
    xacc = xv[0] ^ xv[3];
  }
  uint32_t acc = xacc;
  bool first = true;
  for (int s = 0; s < steps; s += RING) {
#pragma unroll
    for (int r = 0; r < RING; r++) {
      const u32x4 v = ring[r];
      acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
      if (first && r == 0) {
        asm volatile("" ::"v"(acc));
        t3 = now();
        first = false;
      }
      ring[r] = __builtin_nontemporal_load(wp + (size_t)(s + r + RING) * 64);
    }
  }
#pragma unroll
  for (int r = 0; r < RING; r++) acc ^= ring[r][0];
  asm volatile("" ::"v"(acc));
  const unsigned long long t4 = now();
  if (acc == 0x12345u) a.sink[0] = acc;
  if (lane == 0) {
    unsigned long long* t = a.ts + 1024 + ((size_t)blockIdx.x * 16 + wave) * 8;
    t[0] = t0, t[1] = t1, t[2] = t2, t[3] = t3, t[4] = t4;
  }
  __syncthreads();
  if (threadIdx.x == 0) a.ts[1024 + (size_t)1024 * 16 * 8 + blockIdx.x] = now();
}

static double med(std::vector<double>& v) {
  std::sort(v.begin(), v.end());
  return v.empty() ? 0 : v[v.size() / 2];
}

template <int MODE, int WAVES, int RING>
void run(const char* name, Args a, int grid, int steps, hipStream_t st) {
  a.steps = steps;
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 8; i++) {  // 8 x 64 MB of different weights per replay: more than the 256 MB Infinity Cache keeps
    Args b = a;
    b.w = a.w + (size_t)i * (64u << 20) / 4;
    producer<<<256, 1024, 0, st>>>(b);
    consumer<MODE, WAVES, RING><<<grid, WAVES * 64, 0, st>>>(b);
  }
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  std::vector<unsigned long long> h(1024 + 1024 * 16 * 8 + 1024);
  std::vector<double> d_args, d_x, d_w, d_end, d_w0, d_x0, d_gap, d_ramp, d_tot;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 30; it++) {
    hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    if (it < 5) continue;
    hipMemcpy(h.data(), a.ts, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long pend = 0, first = ~0ull, last = 0;
    for (int b = 0; b < 256; b++) pend = std::max(pend, h[b]);
    for (int b = 0; b < grid; b++)
      for (int w = 0; w < WAVES; w++) {
        const unsigned long long* t = &h[1024 + ((size_t)b * 16 + w) * 8];
        first = std::min(first, t[0]);
      }
    for (int b = 0; b < grid; b++) {
      unsigned long long wg0 = ~0ull, wgl = 0;
      for (int w = 0; w < WAVES; w++) {
        const unsigned long long* t = &h[1024 + ((size_t)b * 16 + w) * 8];
        wg0 = std::min(wg0, t[0]), wgl = std::max(wgl, t[0]);
        d_args.push_back((t[1] - t[0]) * 0.01);
        d_x.push_back((t[2] - first) * 0.01);
        d_w.push_back((t[3] - first) * 0.01);
        d_end.push_back((t[4] - first) * 0.01);
        if (w == 0) d_x0.push_back((t[2] - first) * 0.01), d_w0.push_back((t[3] - first) * 0.01);
      }
      d_ramp.push_back((wgl - wg0) * 0.01);
      last = std::max(last, h[1024 + (size_t)1024 * 16 * 8 + b]);
    }
    d_gap.push_back(((long long)first - (long long)pend) * 0.01);
    d_tot.push_back((last - first) * 0.01);
  }
  hipEventRecord(e0, st);
  for (int it = 0; it < 50; it++) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s grid %3d steps %2d: producer end -> first wave %5.2f | wave ramp in a WG %5.2f | args %5.2f | x landed %5.2f (wave0 %5.2f) | first tile %5.2f (wave0 %5.2f) | stream done %5.2f | kernel %5.2f us | pair by events %6.2f us\n",
         name, grid, steps, med(d_gap), med(d_ramp), med(d_args), med(d_x), med(d_x0), med(d_w), med(d_w0), med(d_end), med(d_tot), ms * 1e3 / 400);
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
}

int main() {
  Args a{};
  const size_t wbytes = (size_t)640 << 20;
  hipMalloc((void**)&a.w, wbytes);
  hipMemset((void*)a.w, 1, wbytes);
  hipMalloc((void**)&a.out, 1 << 16);
  a.x = a.out;
  a.K = 4096;
  hipMalloc((void**)&a.ts, (1024 + 1024 * 16 * 8 + 1024) * 8);
  hipMalloc((void**)&a.sink, 64);
  hipStream_t st;
  hipStreamCreate(&st);
  const char* e = getenv("HIP_FORCE_DEV_KERNARG");
  printf("HIP_FORCE_DEV_KERNARG=%s\n", e ? e : "(unset)");
  for (int steps : {4, 14}) {
    for (int grid : {256, 192}) {
      run<0, 16, 2>("0 args->x->wait->ring (kernel E)", a, grid, steps, st);
      run<1, 16, 2>("1 args->x,ring (no wait between)", a, grid, steps, st);
      run<2, 16, 2>("2 args->ring, x by s_load", a, grid, steps, st);
      run<3, 16, 2>("3 args->ring->x (vector)", a, grid, steps, st);
      run<4, 8, 4>("4 as 0, 8 waves x 4 KiB", a, grid, steps * 2, st);
      run<5, 8, 4>("5 as 2, 8 waves x 4 KiB", a, grid, steps * 2, st);
      run<0, 16, 3>("6 as 0, ring 3 KiB", a, grid, steps, st);
      run<2, 16, 3>("7 as 2, ring 3 KiB", a, grid, steps, st);
    }
  }
  return 0;
}
