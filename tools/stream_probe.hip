// burst-read experiment: how fast can a SHORT kernel pull 60 MB from HBM?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int D, bool NT>
__global__ __launch_bounds__(1024) void rd(const u32x4* __restrict__ p, size_t n16, unsigned* out, int per_wave_tiles) {
  // each wave reads per_wave_tiles tiles of 1 KiB, contiguous per workgroup
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  size_t wg_tiles = (size_t)per_wave_tiles * nw;
  const u32x4* base = p + ((size_t)blockIdx.x * wg_tiles) * 64 + lane;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[D];
#pragma unroll
  for (int d = 0; d < D; d++) {
    size_t t = (size_t)wave + (size_t)nw * d;
    if (d < per_wave_tiles) buf[d] = NT ? __builtin_nontemporal_load(base + t * 64) : base[t * 64];
  }
  for (int i0 = 0; i0 < per_wave_tiles; i0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      if (i0 + d < per_wave_tiles) {
        acc ^= buf[d];
        int nx = i0 + d + D;
        if (nx < per_wave_tiles) {
          size_t t = (size_t)wave + (size_t)nw * nx;
          buf[d] = NT ? __builtin_nontemporal_load(base + t * 64) : base[t * 64];
        }
      }
    }
  }
  unsigned v = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (v == 0x12345678u) out[0] = v;
}
__global__ void empty_k(unsigned* out) { if (threadIdx.x == 9999) out[0] = 1; }

template <int D, bool NT>
float run(const char* name, char** bufs, int nbuf, size_t bytes, int grid, int threads, unsigned* out, int iters) {
  int nw = threads / 64;
  size_t tiles = bytes / 1024;
  int per_wave = (int)(tiles / ((size_t)grid * nw));
  size_t used = (size_t)per_wave * grid * nw * 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 8; i++) rd<D, NT><<<grid, threads>>>((const u32x4*)bufs[i % nbuf], 0, out, per_wave);
  hipEventRecord(e0);
  for (int i = 0; i < iters; i++) rd<D, NT><<<grid, threads>>>((const u32x4*)bufs[i % nbuf], 0, out, per_wave);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  float us = ms * 1e3f / iters;
  printf("%-28s grid %5d thr %4d D %2d nt %d tiles/wave %4d  %7.2f us  %7.1f GB/s\n", name, grid, threads, D, (int)NT, per_wave, us, used / us / 1e3);
  return us;
}
int main(int argc, char** argv) {
  size_t bytes = (argc > 1 ? atol(argv[1]) : 60) * (1ull << 20);
  const int nbuf = 8;
  char* bufs[nbuf];
  for (int i = 0; i < nbuf; i++) { hipMalloc(&bufs[i], bytes); hipMemset(bufs[i], i + 1, bytes); }
  unsigned* out; hipMalloc(&out, 64);
  hipDeviceSynchronize();
  { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; i++) empty_k<<<256, 64>>>(out);
    hipEventRecord(e0); for (int i = 0; i < 500; i++) empty_k<<<256, 64>>>(out); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("empty kernel: %.2f us per launch\n", ms * 1e3 / 500); }
  int it = 200;
  run<8, true>("512x512 D8 nt", bufs, nbuf, bytes, 512, 512, out, it);
  run<8, false>("512x512 D8", bufs, nbuf, bytes, 512, 512, out, it);
  run<4, true>("512x512 D4 nt", bufs, nbuf, bytes, 512, 512, out, it);
  run<2, true>("512x512 D2 nt", bufs, nbuf, bytes, 512, 512, out, it);
  run<8, true>("256x1024 D8 nt", bufs, nbuf, bytes, 256, 1024, out, it);
  run<4, true>("256x1024 D4 nt", bufs, nbuf, bytes, 256, 1024, out, it);
  run<4, true>("1024x256 D4 nt", bufs, nbuf, bytes, 1024, 256, out, it);
  run<4, true>("2048x256 D4 nt", bufs, nbuf, bytes, 2048, 256, out, it);
  run<2, true>("4096x256 D2 nt", bufs, nbuf, bytes, 4096, 256, out, it);
  run<2, false>("4096x256 D2", bufs, nbuf, bytes, 4096, 256, out, it);
  run<1, true>("8192x256 D1 nt", bufs, nbuf, bytes, 8192, 256, out, it);
  run<4, true>("1 buf 512x512 D4 nt", bufs, 1, bytes, 512, 512, out, it);
  // kernel W's shape: one 8-wave workgroup per CU, a 4-deep ring of 1 KiB tiles per wave
  run<2, true>("256x512 D2 nt (W/2)", bufs, nbuf, bytes, 256, 512, out, it);
  run<4, true>("256x512 D4 nt (W)", bufs, nbuf, bytes, 256, 512, out, it);
  run<8, true>("256x512 D8 nt (2W)", bufs, nbuf, bytes, 256, 512, out, it);
  run<16, true>("256x512 D16 nt (4W)", bufs, nbuf, bytes, 256, 512, out, it);
  return 0;
}
