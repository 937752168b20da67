// How fast does the weight stream of the decode GEMV kernels (E: 16 waves, W: 8 waves; one workgroup per CU; a ring of D 1-KiB
// tiles per wave; a workgroup walks `q` units of KT tiles) come out of HBM with NO compute — by unit -> workgroup assignment,
// ring depth and waves per workgroup?  58.7 MB = gate + up of Llama-3-8B (1792 units of 32 tiles).
//   mode 0: workgroup b owns units b*q .. b*q+q-1 (what the kernels do)      mode 1: units b, b+G, b+2G, ... (all workgroups inside one moving window)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(1024) void rd(const u32x4* __restrict__ p, int KT, int q, int mode, int swz, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  const int TPW = KT / W, S = q * TPW, G = gridDim.x, b = blockIdx.x;
  auto addr = [&](int s) {
    const int i = s / TPW, ti = s - i * TPW;
    int unit = mode == 0 ? b * q + i : b + G * i;
    if (swz) unit = (unit & ~255) | ((unit * swz) & 255);  // odd multiplier: a permutation of 256 consecutive units
    return p + ((size_t)unit * KT + wave + W * ti) * 64 + lane;
  };
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[D];
#pragma unroll
  for (int d = 0; d < D; d++) buf[d] = __builtin_nontemporal_load(addr(d < S ? d : S - 1));
  for (int s0 = 0; s0 < S; s0 += D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      acc ^= buf[d];
      const int nx = s0 + d + D;
      buf[d] = __builtin_nontemporal_load(addr(nx < S ? nx : S - 1));
    }
  }
  unsigned v = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  if (v == 0x12345678u) out[0] = v;
}
template <int D>
void run(char** bufs, int nbuf, int units, int KT, int grid, int waves, int mode, int swz, unsigned* out) {
  const int q = units / grid;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int i = 0; i < 8; i++) rd<D><<<grid, waves * 64>>>((const u32x4*)bufs[i % nbuf], KT, q, mode, swz, out);
  hipEventRecord(e0);
  const int iters = 200;
  for (int i = 0; i < iters; i++) rd<D><<<grid, waves * 64>>>((const u32x4*)bufs[i % nbuf], KT, q, mode, swz, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const float us = ms * 1e3f / iters;
  const double bytes = (double)q * grid * KT * 1024;
  printf("units %4d KT %3d grid %4d waves %2d D %2d mode %d swz %3d : %6.2f us  %6.1f GB/s (in flight %3d KiB per workgroup)\n", units, KT, grid, waves, D, mode, swz, us, bytes / us / 1e3,
         waves * D);
}
int main() {
  const size_t bytes = 64ull << 20;
  const int nbuf = 8;
  char* bufs[nbuf];
  for (int i = 0; i < nbuf; i++) { hipMalloc(&bufs[i], bytes); hipMemset(bufs[i], i + 1, bytes); }
  unsigned* out;
  hipMalloc(&out, 64);
  hipDeviceSynchronize();
  for (int mode = 0; mode < 2; mode++) {
    for (int waves : {8, 16}) {
      run<1>(bufs, nbuf, 1792, 32, 256, waves, mode, 0, out);
      run<2>(bufs, nbuf, 1792, 32, 256, waves, mode, 0, out);
      run<4>(bufs, nbuf, 1792, 32, 256, waves, mode, 0, out);
      run<8>(bufs, nbuf, 1792, 32, 256, waves, mode, 0, out);
    }
    run<2>(bufs, nbuf, 1792, 32, 512, 8, mode, 0, out);
    run<4>(bufs, nbuf, 1792, 32, 512, 8, mode, 0, out);
  }
  // a permutation of the units inside every 256 (what a different channel spread would look like)
  for (int swz : {3, 5, 37, 101}) run<4>(bufs, nbuf, 1792, 32, 256, 8, 0, swz, out), run<2>(bufs, nbuf, 1792, 32, 256, 16, 0, swz, out);
  // down_proj: 256 units of 112 tiles
  for (int mode = 0; mode < 2; mode++) {
    run<4>(bufs, nbuf, 256, 112, 256, 16, mode, 0, out);
    run<2>(bufs, nbuf, 256, 112, 256, 16, mode, 0, out);
  }
  // o_proj / q/k/v: 256 / 384 units of 32 tiles
  run<2>(bufs, nbuf, 256, 32, 256, 16, 0, 0, out);
  run<4>(bufs, nbuf, 256, 32, 256, 8, 0, 0, out);
  return 0;
}
