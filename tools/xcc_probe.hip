// Which XCD does workgroup (x, y, z) of a grid land on?  (tools: same-XCD split-K exchange)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* out) {
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xf;  // HW_REG_XCC_ID[3:0]
  if (threadIdx.x == 0) out[lin] = xcc;
}
static void run(dim3 g, int threads) {
  const unsigned n = g.x * g.y * g.z;
  unsigned* d;
  hipMalloc(&d, n * 4);
  hipMemset(d, 0xff, n * 4);
  probe<<<g, threads>>>(d);
  std::vector<unsigned> h(n);
  hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (unsigned i = 0; i < n; i++) bad += h[i] != (i % 8);
  printf("grid (%u,%u,%u) x %d threads: %d of %u workgroups NOT on XCD (linear id mod 8); first 24:", g.x, g.y, g.z, threads, bad, n);
  for (unsigned i = 0; i < 24 && i < n; i++) printf(" %u", h[i]);
  printf("\n");
  hipFree(d);
}
int main() {
  for (int rep = 0; rep < 3; rep++) {
    run(dim3(256), 768);
    run(dim3(32, 1, 7), 768);
    run(dim3(224), 768);
    run(dim3(1000), 256);
    run(dim3(24, 2, 8), 256);
    run(dim3(7, 3, 5), 1024);
  }
  return 0;
}
