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
static int launch_status() { return 0; }
#else
#define LFSD_LAUNCH(kern, grid, block, stream, args) \
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, (hipStream_t)(stream), args)
static int launch_status() { return (int)hipGetLastError(); }
#endif


// The Riccati sweep is the one kernel that gains from clang's SLP vectoriser (it pairs 44 % of the kernel's scalar math
// into v_pk_* ops: 8.6 vs 9.7 ms); every other kernel loses to the register pressure and operand shuffling the pairing
// costs (oc_solve 11.4 -> 8.3 ms, aux_forward 3.35 -> 2.97 ms without it; profiles/r01_tune_compiler_flags.txt).  The
// product build therefore compiles it in its own translation unit with SLP on and the rest with -fno-slp-vectorize
// (runtime.build_library, -DLFSD_SPLIT_RICCATI); single-TU builds (emulator, sanitizer, tuning tools) include the launcher.
namespace lfsd_detail {
int launch_riccati_f32(unsigned grid, void* stream, const lfsd::AuxArgs<float>& a);
int launch_riccati_f64(unsigned grid, void* stream, const lfsd::AuxArgs<double>& a);
}
