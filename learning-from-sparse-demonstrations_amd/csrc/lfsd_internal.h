// Shared prologue of the translation units of one model library (lfsd_capi.cpp, lfsd_riccati.cpp).
#pragma once
#include "cpdp_kernels.h"
#include LFSD_MODEL_HEADER
#define LFSD_API extern "C" __attribute__((visibility("default")))
#include "../../include/lfsd_cpdp.h"

#ifndef LFSD_G
#error "LFSD_G (lanes per trajectory) must be defined by the build"
#endif

using Model = LFSD_MODEL_NS::Model;
static constexpr int G = LFSD_G;
static constexpr int GPB = 64 / G;

#if defined(LFSD_EMU)
#define LFSD_LAUNCH(kern, grid, block, stream, args) emu::launch(dim3(grid), dim3(block), [&] { kern(args); })
#define LFSD_ZERO(ptr, bytes, stream) memset((ptr), 0, (bytes))
static int launch_status() { return 0; }
static int device_cu_count() { return 256; }
#else
#include <cstdio>
#include <cstdlib>
// A launch error shows up in hipGetLastError() at once; a fault INSIDE the kernel (e.g. a memory aperture violation) only at
// the caller's next synchronisation, without a name attached.  LFSD_SYNC_CHECK=1 in the environment makes every launch of
// this library synchronise its stream and report the kernel by name (debugging aid: it serialises the stream).
static bool lfsd_sync_check() {
  static const int on = [] { const char* e = getenv("LFSD_SYNC_CHECK"); return (e && atoi(e) != 0) ? 1 : 0; }();
  return on != 0;
}
#define LFSD_LAUNCH(kern, grid, block, stream, args)                                                          \
  do {                                                                                                        \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, (hipStream_t)(stream), args);                        \
    if (lfsd_sync_check()) {                                                                                  \
      const hipError_t e1_ = hipPeekAtLastError();                                                            \
      const hipError_t e2_ = hipStreamSynchronize((hipStream_t)(stream));                                     \
      if (e1_ != hipSuccess || e2_ != hipSuccess)                                                             \
        fprintf(stderr, "lfsd: kernel %s (grid %u) failed: launch %s, execution %s\n", #kern, (unsigned)(grid), \
                hipGetErrorString(e1_), hipGetErrorString(e2_));                                              \
    }                                                                                                         \
  } while (0)
#define LFSD_ZERO(ptr, bytes, stream) (void)hipMemsetAsync((ptr), 0, (bytes), (hipStream_t)(stream))
static int launch_status() { return (int)hipGetLastError(); }
static int device_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
  }();
  return n;
}
#endif


// Compiler settings per kernel (profiles/r01_tune_compiler_flags.txt).  clang's SLP vectoriser pairs scalar fp32 operations into
// v_pk_* instructions at the price of adjacent-register constraints: every kernel of this library is faster without it (oc_solve
// 11.4 -> 8.2 ms, aux_forward 3.35 -> 2.97 ms; the Riccati sweep needs so few registers without it that it runs two waves per
// SIMD: 8.5 -> 7.1 ms).  The Riccati sweep keeps its own translation unit (runtime.hipcc_commands, -DLFSD_SPLIT_RICCATI): rounds
// 1-5 compiled everything else with the max-ILP instruction scheduler, which cost the Riccati sweep 25 %; round 6 dropped that
// scheduler (two wrong builds of the wide OC kernel under it, no gain left: runtime.py, profiles/r06_f_wide_stale_cost.txt; the
// cause -- register spills placed before the exec restore of a join block, also seen under the default scheduler -- is what
// runtime.py scans every build's assembly for: lfsd_amd/isa_check.py).
// Single-unit builds (emulator, sanitizer, tuning tools) include the launcher below instead.
namespace lfsd_detail {
int launch_riccati_f32(unsigned grid, void* stream, const lfsd::AuxArgs<float>& a);
int launch_riccati_f64(unsigned grid, void* stream, const lfsd::AuxArgs<double>& a);
}
