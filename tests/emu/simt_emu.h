// TEST INFRASTRUCTURE ONLY — a minimal CPU SIMT emulator so the HIP kernels in
// learning-from-sparse-demonstrations_amd/csrc can be compiled with g++ and run
// (slowly) without a GPU.  Every lane of a workgroup is a ucontext fiber;
// __syncthreads() yields to a round-robin scheduler; __shared__ memory is a
// function-local static (workgroups run one after another on one OS thread).
// The product path never loads a library built from this header.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
struct Lane {
  ucontext_t ctx;
  std::vector<char> stack;
  dim3 tid;
  bool done = false;
};
struct State {
  Lane* cur = nullptr;
  dim3 block_idx, block_dim, grid_dim;
  ucontext_t sched;
  std::function<void()> body;
};
inline State& st() { static State s; return s; }

inline void trampoline() {
  State& s = st();
  s.body();
  s.cur->done = true;
  swapcontext(&s.cur->ctx, &s.sched);
}
inline void barrier() {
  State& s = st();
  swapcontext(&s.cur->ctx, &s.sched);
}
template <class F> void launch(dim3 grid, dim3 block, F f) {
  State& s = st();
  s.grid_dim = grid; s.block_dim = block; s.body = f;
  const unsigned nl = block.x;
  std::vector<Lane> lanes(nl);
  for (auto& l : lanes) l.stack.resize(1 << 20);
  for (unsigned b = 0; b < grid.x; ++b) {
    s.block_idx = dim3(b);
    for (unsigned i = 0; i < nl; ++i) {
      Lane& l = lanes[i];
      l.tid = dim3(i); l.done = false;
      getcontext(&l.ctx);
      l.ctx.uc_stack.ss_sp = l.stack.data();
      l.ctx.uc_stack.ss_size = l.stack.size();
      l.ctx.uc_link = &s.sched;
      makecontext(&l.ctx, (void (*)())trampoline, 0);
    }
    bool any = true;
    while (any) {          // each sweep runs every live lane up to its next barrier
      any = false;
      for (unsigned i = 0; i < nl; ++i) {
        if (lanes[i].done) continue;
        s.cur = &lanes[i];
        swapcontext(&s.sched, &lanes[i].ctx);
        if (!lanes[i].done) any = true;
      }
    }
  }
  s.cur = nullptr;
}
}  // namespace emu

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define LFSD_DEV inline
#define threadIdx (emu::st().cur->tid)
#define blockIdx (emu::st().block_idx)
#define blockDim (emu::st().block_dim)
#define gridDim (emu::st().grid_dim)
inline void __syncthreads() { emu::barrier(); }
inline long long clock64() { return 0; }      // (the diagnostic clock builds of the kernels compile on the CPU too)
using std::sin; using std::cos; using std::tan; using std::exp; using std::log; using std::sqrt; using std::pow;
