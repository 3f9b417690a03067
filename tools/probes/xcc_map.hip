// which XCD a workgroup lands on: blockIdx -> XCC_ID (hardware register 20) for a grid of one-wavefront workgroups
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/xcc_map.hip -o gpurun_out/xcc_map ; run: gpurun_out/xcc_map 1024
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void __launch_bounds__(64) k(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u);
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  int* d; hipMalloc(&d, n * sizeof(int));
  hipLaunchKernelGGL(k, dim3(n), dim3(64), 0, 0, d);
  std::vector<int> h(n); hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
  int agree = 0; for (int i = 0; i < n; ++i) agree += (h[i] == h[i % 8]);
  printf("first 32:"); for (int i = 0; i < 32 && i < n; ++i) printf(" %d", h[i]); printf("\nblocks whose XCC equals that of block (i mod 8): %d of %d\n", agree, n);
  return 0;
}
