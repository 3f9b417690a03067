// Probe (run on the GPU): lane/register layout of v_mfma_f32_16x16x1_4b_f32 and of the permlane swaps, checked with exact
// integer data.  build: hipcc --offload-arch=gfx950 -O2 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include "../../learning-from-sparse-demonstrations_amd/csrc/cpdp_kernels.h"
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { printf("%s: %s\n", #call, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out) {
  const int l = threadIdx.x;
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // D_b[i][j] = a(row lane) * b(col lane):  a = 1000*b + 10*i encoded by the row provider, b = 1
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32((float)(1000 * (l >> 4) + 10 * (l & 15)), 1.0f, acc, 0, 0, 0);
  f32x16 acc2; for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(1.0f, (float)(1000 * (l >> 4) + (l & 15)), acc2, 0, 0, 0);
  for (int r = 0; r < 16; ++r) { out[l * 16 + r] = acc[r]; out[1024 + l * 16 + r] = acc2[r]; }
  // permlane swaps on a = 100 + lane, b = 200 + lane
  unsigned a = 100 + l, b = 200 + l;
  auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[2048 + l * 4 + 0] = (float)r32[0]; out[2048 + l * 4 + 1] = (float)r32[1];
  out[2048 + l * 4 + 2] = (float)r16[0]; out[2048 + l * 4 + 3] = (float)r16[1];
}
// the helpers of the MFMA backward sweep exactly as the kernels use them: K rank-1 updates per block, then the transposition
__global__ void chain(const float* a, const float* b, float* out) {
  lfsd::f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float av[13], bv[13];
  for (int i = 0; i < 13; ++i) { av[i] = a[threadIdx.x * 13 + i]; bv[i] = b[threadIdx.x * 13 + i]; }
#pragma unroll
  for (int kk = 0; kk < 13; ++kk) lfsd::mfma4b(av[kk], bv[kk], acc);
  lfsd::tile_transpose(acc);
  for (int r = 0; r < 16; ++r) out[threadIdx.x * 16 + r] = acc[r];
}
static int check_chain() {
  static float ha[64 * 13], hb[64 * 13], ho[64 * 16];
  srand(7);
  for (int i = 0; i < 64 * 13; ++i) { ha[i] = (float)(rand() % 17 - 8); hb[i] = (float)(rand() % 13 - 6); }      // exact in fp32
  float *da, *db, *dout;
  HIP_OK(hipMalloc(&da, sizeof(ha))); HIP_OK(hipMalloc(&db, sizeof(hb))); HIP_OK(hipMalloc(&dout, sizeof(ho)));
  HIP_OK(hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice)); HIP_OK(hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, da, db, dout);
  HIP_OK(hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int blk = 0; blk < 4; ++blk) for (int j = 0; j < 16; ++j) for (int i = 0; i < 16; ++i) {
    float ref = 0.f;                                  // D_blk[i][j] = sum_k a(lane 16 blk + i)[k] * b(lane 16 blk + j)[k]
    for (int k = 0; k < 13; ++k) ref += ha[(16 * blk + i) * 13 + k] * hb[(16 * blk + j) * 13 + k];
    const float got = ho[(16 * blk + j) * 16 + i];    // after tile_transpose: lane 16 blk + j, register i
    if (got != ref) { if (bad < 8) printf("chain block %d i %d j %d: got %g expected %g\n", blk, i, j, got, ref); ++bad; }
  }
  printf("lfsd::mfma4b x13 + lfsd::tile_transpose vs host: %s (%d mismatches of 1024)\n", bad ? "WRONG" : "CONFIRMED", bad);
  return bad;
}
int main() {
  check_chain();
  float* d; HIP_OK(hipMalloc(&d, 4096 * 4)); HIP_OK(hipMemset(d, 0, 4096 * 4));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  static float h[4096]; HIP_OK(hipMemcpy(h, d, 4096 * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
    // hypothesis: register 4*b + (i%4), lane 16*(i/4) + j  holds  D_b[i][j]
    const int b = r / 4, i = 4 * (l >> 4) + r % 4, j = l & 15;
    if (h[l * 16 + r] != 1000.f * b + 10.f * i) { if (bad < 8) printf("A-probe lane %d reg %d: got %g expected %g\n", l, r, h[l * 16 + r], 1000.f * b + 10.f * i); ++bad; }
    if (h[1024 + l * 16 + r] != 1000.f * b + j) { if (bad < 8) printf("B-probe lane %d reg %d: got %g expected %g\n", l, r, h[1024 + l * 16 + r], 1000.f * b + j); ++bad; }
  }
  printf("mfma_f32_16x16x1f32 (4 blocks) layout hypothesis [reg 4b + i%%4, lane 16(i/4) + j]: %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
  printf("lane: permlane32_swap(a=100+l, b=200+l) -> (r0, r1) | permlane16_swap -> (r0, r1)\n");
  for (int l = 0; l < 64; l += 1) printf("%2d: %3.0f %3.0f | %3.0f %3.0f\n", l, h[2048 + l * 4], h[2048 + l * 4 + 1], h[2048 + l * 4 + 2], h[2048 + l * 4 + 3]);
  return 0;
}
