// Vector issue rate of one SIMD with one and with two resident wavefronts (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/probes/valu_issue_probe.hip -o tools/probes/valu_issue_probe
// MI355X_MICROARCH.md gives v_fma_f32 (wave64) as 2 cycles per SIMD but 4 cycles for one wave alone.  The OC solve of
// this repo runs ONE wavefront per SIMD at the benchmark batch (4096 trajectories / 1024 SIMDs, 4 per wavefront): if two
// co-resident wavefronts really issue at twice the rate, a solve split into two cooperating wavefronts per SIMD (nominal
// roll-out | tangent sweep) can halve the issue-bound phase.  Measured here: 8 independent FMA chains per lane (no
// dependency stall at 4 cycles per issue), scalar and packed, grids of 1, 2 and 4 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void __launch_bounds__(64) chains(float* out, int iters, float a, float b) {
  if (PK) {
    f2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f2{(float)threadIdx.x + i, (float)i};
    const f2 av = {a, a}, bv = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
      }
    }
    f2 s = {0.f, 0.f};
    for (int i = 0; i < 8; ++i) s += x[i];
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = s.x + s.y;
  } else {
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fmaf(x[i], a, b);
      }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
  }
}

template <int PK> static float run(int blocks, int iters, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(chains<PK>, dim3(blocks), dim3(64), 0, 0, d, 16, 0.999f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(chains<PK>, dim3(blocks), dim3(64), 0, 0, d, iters, 0.999f, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int simds = p.multiProcessorCount * 4;
  printf("device %s, %d CUs, %d SIMDs, clock %d kHz\n", p.name, p.multiProcessorCount, simds, p.clockRate);
  float* d;
  hipMalloc(&d, (size_t)simds * 8 * 64 * sizeof(float));
  const int iters = 20000;                       // x 64 vector instructions per lane
  const double insts = (double)iters * 64;
  for (int pk = 0; pk < 2; ++pk) {
    for (int w : {1, 2, 4}) {
      const int blocks = simds * w;
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) { const float ms = pk ? run<1>(blocks, iters, d) : run<0>(blocks, iters, d); if (ms < best) best = ms; }
      // all SIMDs hold w wavefronts; each issues `insts` vector instructions
      printf("%s  %d wave(s)/SIMD  %8.3f ms   %.2f ns per instruction per wave   SIMD aggregate %.2f ns per instruction\n",
             pk ? "v_pk_fma_f32" : "v_fma_f32   ", w, best, best * 1e6 / insts, best * 1e6 / insts / w);
    }
  }
  hipFree(d);
  return 0;
}
