// Issue rate of the fp64 vector instructions of one wavefront alone on its SIMD (gfx950): v_fma_f64, v_mul_f64, v_add_f64, and
// for comparison v_fma_f32 -- shader clocks per instruction from 8 independent chains per lane (no dependency stall).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/probes/f64_issue_probe.hip -o tools/probes/f64_issue_probe
// The fp64 kernels of this repo run one wavefront per SIMD; DESIGN.md prices their instruction streams with these figures.
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T, int OP>
__global__ void __launch_bounds__(64) chains(T* out, long long* clk, int iters, T a, T b) {
  T x[8];
  for (int i = 0; i < 8; ++i) x[i] = (T)threadIdx.x + (T)i;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) x[i] = sizeof(T) == 4 ? (T)__builtin_fmaf((float)x[i], (float)a, (float)b) : (T)__builtin_fma((double)x[i], (double)a, (double)b);
        else if (OP == 1) x[i] = x[i] * a;
        else x[i] = x[i] + b;
      }
    }
  }
  const long long t1 = clock64();
  T s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <typename T, int OP> static void run(const char* name, int blocks) {
  T* d; long long* c;
  hipMalloc(&d, sizeof(T) * 64 * blocks); hipMalloc(&c, sizeof(long long) * blocks);
  const int iters = 4096;
  hipLaunchKernelGGL((chains<T, OP>), dim3(blocks), dim3(64), 0, 0, d, c, 16, (T)0.999, (T)0.001);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((chains<T, OP>), dim3(blocks), dim3(64), 0, 0, d, c, iters, (T)0.999, (T)0.001);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
  long long h = 0; hipMemcpy(&h, c, sizeof(h), hipMemcpyDeviceToHost);
  const double n = (double)iters * 64;
  printf("%-12s %5d wavefront(s): %6.2f clocks per instruction (wave 0), %8.3f ms, %.2f GHz implied\n", name, blocks, h / n, ms, h / (ms * 1e6));
  hipFree(d); hipFree(c);
}

int main() {
  for (int blocks : {1, 1024, 2048}) {
    run<float, 0>("v_fma_f32", blocks);
    run<double, 0>("v_fma_f64", blocks);
    run<double, 1>("v_mul_f64", blocks);
    run<double, 2>("v_add_f64", blocks);
  }
  return 0;
}
