// Accuracy of lfsd::t_sin / t_cos (fp32, cpdp_common.h) against fp64 on 2^24 arguments in [-100, 100] and on a sweep of
// [-pi, pi], beside the device library's sinf / cosf; and the instruction counts of the four (see the build line).
//   hipcc --offload-arch=gfx950 -O3 -I learning-from-sparse-demonstrations_amd/csrc tools/probes/trig_probe.hip -o /tmp/trig_probe && /tmp/trig_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "cpdp_common.h"
__global__ void k(const float* x, float* s1, float* c1, float* s2, float* c2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  s1[i] = lfsd::t_sin(x[i]); c1[i] = lfsd::t_cos(x[i]); s2[i] = sinf(x[i]); c2[i] = cosf(x[i]);
}
static double ulp_err(float got, double ref) {
  const double a = std::fabs(ref);
  int e; std::frexp(a > 1e-30 ? a : 1e-30, &e);
  return std::fabs((double)got - ref) / std::ldexp(1.0, e - 24);
}
int main() {
  const int n = 1 << 24;
  for (int pass = 0; pass < 2; ++pass) {
    const double lo = pass ? -3.14159265358979 : -100.0, hi = -lo;
    std::vector<float> x(n), s1(n), c1(n), s2(n), c2(n);
    for (int i = 0; i < n; ++i) x[i] = (float)(lo + (hi - lo) * (i + 0.37) / n);
    float *dx, *d[4];
    hipMalloc(&dx, n * 4); for (auto& p : d) hipMalloc(&p, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d[0], d[1], d[2], d[3], n);
    hipMemcpy(s1.data(), d[0], n * 4, hipMemcpyDeviceToHost); hipMemcpy(c1.data(), d[1], n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s2.data(), d[2], n * 4, hipMemcpyDeviceToHost); hipMemcpy(c2.data(), d[3], n * 4, hipMemcpyDeviceToHost);
    double u[4] = {0, 0, 0, 0}, ab[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
      const double rs = std::sin((double)x[i]), rc = std::cos((double)x[i]);
      const float g[4] = {s1[i], c1[i], s2[i], c2[i]};
      const double r[4] = {rs, rc, rs, rc};
      for (int j = 0; j < 4; ++j) { u[j] = std::fmax(u[j], ulp_err(g[j], r[j])); ab[j] = std::fmax(ab[j], std::fabs((double)g[j] - r[j])); }
    }
    printf("[%g, %g], %d points: max error  t_sin %.2f ulp (abs %.2e)  t_cos %.2f ulp (abs %.2e) | sinf %.2f ulp (abs %.2e)  cosf %.2f ulp (abs %.2e)\n",
           lo, hi, n, u[0], ab[0], u[1], ab[1], u[2], ab[2], u[3], ab[3]);
    hipFree(dx); for (auto& p : d) hipFree(p);
  }
  return 0;
}
