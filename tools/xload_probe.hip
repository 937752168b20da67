// xload_probe — how fast does ONE workgroup per CU (8 waves, as kernel W) pull the same 256 KB of activations (32 rows x 4096
// bf16, L2 resident) into registers: (A) row-major, 16 rows x 64 bytes per wave load (kernel W's x fragments, with its odd-row
// swizzle), (B) the same bytes laid out in fragment order, one contiguous KiB per wave load?  
//   hipcc -O3 --offload-arch=gfx950 -o tools/xload_probe tools/xload_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void xload(const uint16_t* __restrict__ x, uint32_t* sink, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nn = lane & 15, oct = lane >> 4;
  u32x4 f[4][4][2];
#pragma unroll
  for (int ti = 0; ti < 4; ti++) {
    const int kt = wave + 8 * ti;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (MODE == 0) f[ti][j][mt] = *reinterpret_cast<const u32x4*>(x + (size_t)(mt * 16 + nn) * K + kt * 128 + (j ^ (nn & 1)) * 32 + oct * 8);
        else f[ti][j][mt] = *(reinterpret_cast<const u32x4*>(x) + ((size_t)((kt * 2 + mt) * 4 + j) * 64 + lane));
      }
  }
  uint32_t a = 0;
#pragma unroll
  for (int ti = 0; ti < 4; ti++)
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
      for (int j = 0; j < 4; j++) a ^= f[ti][j][mt][0] ^ f[ti][j][mt][1] ^ f[ti][j][mt][2] ^ f[ti][j][mt][3];
  if (a == 0x12345u) sink[0] = a;
}
int main() {
  const int K = 4096, M = 32;
  uint16_t* x;
  uint32_t* sink;
  hipMalloc(&x, (size_t)M * K * 2);
  hipMemset(x, 1, (size_t)M * K * 2);
  hipMalloc(&sink, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++)
    for (int grid : {256, 64}) {
      for (int i = 0; i < 20; i++) mode ? xload<1><<<grid, 512>>>(x, sink, K) : xload<0><<<grid, 512>>>(x, sink, K);
      hipEventRecord(e0);
      for (int i = 0; i < 400; i++) mode ? xload<1><<<grid, 512>>>(x, sink, K) : xload<0><<<grid, 512>>>(x, sink, K);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s grid %3d: %.2f us per launch (launch overhead ~2.4 included), 256 KB per workgroup\n", mode ? "fragment-major (1 KiB contiguous per wave load)" : "row-major (16 rows x 64 B per wave load)      ", grid, ms * 1e3 / 400);
    }
  return 0;
}
