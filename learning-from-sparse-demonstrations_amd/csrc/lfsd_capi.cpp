// C ABI (include/lfsd_cpdp.h) over the kernels of cpdp_kernels.h for ONE model.
// Build:  runtime.build_library (two translation units, see lfsd_internal.h), or as a single one:
//         hipcc -x hip --offload-arch=gfx950 -DLFSD_MODEL_HEADER='"gen/<hash>.h"' -DLFSD_G=<lanes> ...
// (tests/emu builds the same file with g++ -DLFSD_EMU for the CPU SIMT emulator.)
#include "lfsd_internal.h"
#include <cmath>
#include <cstdlib>
#include <algorithm>
#ifndef LFSD_WIDE_MAX_BATCH
#define LFSD_WIDE_MAX_BATCH 1536
#endif
#if !defined(LFSD_SPLIT_RICCATI)
#include "lfsd_riccati.inc"
#endif

LFSD_API int lfsd_get_model_info(lfsd_model_info* out) {
  if (!out) return LFSD_EINVAL;
  out->abi_version = LFSD_ABI_VERSION;
  out->n_state = Model::NX; out->n_control = Model::NU; out->n_auxvar = Model::NP; out->n_const = Model::NC_REAL;
  out->time_varying = Model::TIME_VARYING ? 1 : 0;
  out->lanes_per_trajectory = G;
#if defined(LFSD_EMU)
  out->is_emulator = 1;
#else
  out->is_emulator = 0;
#endif
  out->name = Model::name();
  out->hash = Model::hash();
  return 0;
}

LFSD_API int lfsd_interface_dim(void) { return Model::NIF; }

LFSD_API double lfsd_const_default(int i) {
  if (i < 0 || i >= Model::NC_REAL) return 0.0;
  return Model::const_default(i);
}

// fp32 lean OC kernel of the 32-lane models: packed roll-out, four trajectories per wavefront (cpdp_kernels.h, PK)
static constexpr bool OC_PK = (G == 32) && (Model::NX + Model::NU + 1) / 2 <= 16;
static constexpr int OC_GPB = OC_PK ? 4 : GPB;      // scratch slots are padded to whole workgroups of either mapping
static long long padded_batch(int batch) { return ((long long)(batch + OC_GPB - 1) / OC_GPB) * OC_GPB; }

// Which mapping solves a batch.  The lock-step kernels (several trajectories per wavefront, intervals in sequence) fill the
// machine from ~4096 trajectories up; below ~1500 most of the 1024 SIMDs would have no wavefront, and where an iteration is
// expensive -- exact stage Hessians: robot arm, rocket -- the WIDE kernel (one trajectory per wavefront, intervals in
// parallel: oc_solve_wide_kernel) is 3.5-6x faster (robot arm 1024 seeds 108 -> 18.6 ms, rocket n_grid 100 763 -> 215 ms).
// Models that converge in a handful of cheap iterations (quadrotor: 5, pendulum, cart-pole) are faster on the lock-step
// mapping from a few thousand trajectories up, the fp32 packed / MFMA lean kernel at every batch size; models that spend
// most iterations on exact stage Hessians are faster on the wide mapping at every batch size measured (robot arm 4096:
// 113 -> 41 ms, 8192: 202 -> 75 ms; rocket 4096: 1315 -> 525 ms; profiles/r02_d_wide_vs_lockstep.txt).  The caller can say
// so (`mapping`: LFSD_MAP_AUTO / _LOCKSTEP / _WIDE; lfsd_amd.models does for the robot arm).  A bounded problem
// (control_lb / control_ub) always runs the wide kernel.
static bool use_wide(int dtype, int batch, int exact_after, int mapping, bool bounded) {
  if (bounded) return true;
  if (mapping == LFSD_MAP_LOCKSTEP) return false;
  if (mapping == LFSD_MAP_WIDE) return true;
  if (exact_after == 0) return true;      // Newton from the first iteration (rocket): wide wins at every batch size measured
  const bool lean_mfma = OC_PK && dtype == LFSD_F32;
  return batch <= LFSD_WIDE_MAX_BATCH && !lean_mfma;
}

// fp64 on the lock-step mapping, 32-lane models (the class with a mesh continuation), no bounds, not Newton-from-start: a cold
// start is solved in fp32 first (the lean MFMA kernel, a quarter of the fp64 kernel's time per iteration) and the fp64 kernel
// starts from those controls with the Hamiltonian model -- 3-4 fp64 iterations on the reference's grid instead of 8-9 (5 of
// them coarse).  Every returned number is the fp64 kernel's, every convergence test runs in fp64 on the NLP of CPDP.py:110-175:
// the fp32 solve only changes the path to its KKT point, as the mesh continuation does.  Measured on the benchmark's fp64 leg:
// DESIGN.md section 3.1, profiles/r04_w_mixed_precision_probe.txt.
// LFSD_F64_SEED=0 in the environment switches it off (the kernel-level tests that compare the fp64 kernel's own iterates).
static bool seeded_f64(int dtype, int batch, int exact_after, int mapping, bool bounded) {
  const char* e = getenv("LFSD_F64_SEED");
  if (e && atoi(e) == 0) return false;
  return (LFSD_F64_SEED != 0) && OC_PK && dtype == LFSD_F64 && !bounded && exact_after != 0 &&
         !use_wide(dtype, batch, exact_after, mapping, bounded);
}
// staging area behind the fp64 scratch (byte offsets from its end, 256-byte aligned)
struct SeedLayout {
  size_t ws32, x0, hz, th, cs, xs, us, ls, cost, it, st, u0, ui, end;
  SeedLayout(int batch, int n_grid, int const_rows) {
    const size_t B = (size_t)batch, N1 = (size_t)n_grid + 1;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) / 256 * 256; return at; };
    ws32 = take((size_t)padded_batch(batch) * (size_t)lfsd::OcLayout<Model>::template ws_elems<G>(n_grid) * 4);
    x0 = take(B * Model::NX * 4); hz = take(B * 4); th = take(B * Model::NP * 4);
    cs = take(B * (size_t)(Model::NC > 0 ? Model::NC : 1) * 4);      // (sized for per-trajectory constants whatever the call passes)
    xs = take(B * N1 * Model::NX * 4); us = take(B * N1 * Model::NU * 4); ls = take(B * N1 * Model::NX * 4);
    cost = take(B * 4); it = take(B * 4); st = take(B * 4);
    u0 = take(B * (size_t)n_grid * Model::NU * 8);
    ui = take(B * (size_t)n_grid * Model::NU * 4);      // the caller's initial guess in fp32 (rows of zeros = cold start)
    end = o; (void)const_rows;
  }
  size_t total(size_t f64_bytes) const { return (f64_bytes + 255) / 256 * 256 + end; }
};

// ---- wide mapping with several wavefronts per trajectory (oc_solve_wide_kernel<..., W>; fp32, the models whose interval-parallel
// phases take several rounds of one wavefront: rocket, quadrotor) ----
// A wide launch lasts as long as its slowest trajectory (rocket learner step: median 53 iterations, slowest 160), and for most of that
// time most SIMDs hold a finished workgroup.  A workgroup of WIDE_W wavefronts runs the two interval-parallel phases of ONE trajectory
// (exact stage Hessians + linearisation: 53 % of a rocket iteration) on the four SIMDs of a CU; a CU holds one such workgroup (512
// registers per wavefront, 116 KB of LDS).  So:  batch <= CUs -- every trajectory gets such a workgroup from the start;  otherwise the
// solve is TWO launches on the stream: one wavefront per trajectory until all but `CUs` trajectories are finished (a device counter;
// the rest park their solver state in the workspace), then the rest with WIDE_W wavefronts each.  No host read in between: the second
// launch is enqueued unconditionally, one workgroup per entry of the hand-over list the first one wrote (the others leave at once).  Same arithmetic per item whoever runs it:
// results do not depend on when a trajectory was handed over (asserted by the GPU tier: bit-identical to the one-launch solve).
// Environment (test hooks / A-B): LFSD_WIDE_WAVES=1 one wavefront per trajectory always, =4 WIDE_W from the start at any batch;
// LFSD_WIDE_CAPACITY=<n> in place of the CU count; LFSD_WIDE_SUSPEND_IT=<k> hand over at iteration k instead of by the counter.
static constexpr int WIDE_W = 4;
template <typename T> static constexpr bool wide_multi() { return sizeof(T) == 4 && !lfsd::OcLayout<Model>::HALL; }
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
static size_t wide_sched_offset(size_t solver_bytes) { return (solver_bytes + 255) / 256 * 256; }      // the counters + hand-over list behind the solver scratch
static size_t wide_sched_bytes(int batch) { return ((size_t)(2 + batch) * sizeof(int) + 255) / 256 * 256; }

// exactly what lfsd_coc_solve needs for the mapping the same arguments select (ABI 6; ABI 5 returned the larger of the two
// layouts whatever the batch: 9.6 GB instead of 6.4 GB at 32768 quadrotor trajectories)
LFSD_API size_t lfsd_coc_workspace_bytes(int dtype, int batch, int n_grid, int exact_after, int mapping, int bounded) {
  if (batch <= 0 || n_grid <= 0 || (dtype != LFSD_F32 && dtype != LFSD_F64)) return 0;
  if (mapping < LFSD_MAP_AUTO || mapping > LFSD_MAP_WIDE) return 0;
  const size_t es = dtype == LFSD_F32 ? 4 : 8;
  if (use_wide(dtype, batch, exact_after, mapping, bounded != 0))
    return wide_sched_offset((size_t)batch * (size_t)lfsd::OcLayout<Model>::ws_elems_wide(n_grid) * es) + wide_sched_bytes(batch);
  const size_t lock = (size_t)padded_batch(batch) * (size_t)lfsd::OcLayout<Model>::template ws_elems<G>(n_grid) * es;
  if (seeded_f64(dtype, batch, exact_after, mapping, bounded != 0)) return SeedLayout(batch, n_grid, 1).total(lock);
  return lock;
}

static int coc_solve_seeded(int batch, int n_grid, int steps_per_grid, const void* ini_state, const void* horizon,
                            const void* auxvar, const void* consts, int const_per_traj, const void* u_init, void* state_grid,
                            void* control_grid, void* costate_grid, void* cost, int* iters, int* status, int max_iter, double tol,
                            int exact_after, int mapping, void* workspace, size_t workspace_bytes, void* stream);

template <typename T>
static int coc_solve_t(int batch, int n_grid, int steps_per_grid, const void* ini_state, const void* horizon,
                       const void* auxvar, const void* consts, int const_per_traj, const void* u_init,
                       const void* control_lb, const void* control_ub,
                       const void* state_lb, const void* state_ub, const void* state_mult, double state_rho,
                       void* state_grid, void* control_grid, void* costate_grid, void* cost, int* iters, int* status,
                       int max_iter, double tol, int exact_after, int mapping, void* workspace, size_t workspace_bytes,
                       void* stream, int start_mode = 0) {
  if constexpr (sizeof(T) == 8) {
    // (start_mode 1 marks the fp64 half of a seeded solve: it must not seed itself again)
    // (and a caller-held workspace sized while LFSD_F64_SEED=0 was in the environment has no staging area: that solve runs
    //  unseeded instead of failing with LFSD_ENOSPC -- the seed changes the path to the KKT point, not the point)
    if (start_mode == 0 && seeded_f64(LFSD_F64, batch, exact_after, mapping, control_lb != nullptr) &&
        workspace_bytes >= SeedLayout(batch, n_grid, 1).total((size_t)padded_batch(batch) * (size_t)lfsd::OcLayout<Model>::template ws_elems<G>(n_grid) * 8))
      return coc_solve_seeded(batch, n_grid, steps_per_grid, ini_state, horizon, auxvar, consts, const_per_traj, u_init, state_grid,
                              control_grid, costate_grid, cost, iters, status, max_iter, tol, exact_after, mapping, workspace,
                              workspace_bytes, stream);
  }
  lfsd::OcArgs<T> a;
  a.sched = nullptr; a.suspend_at = 0; a.suspend_it = -1;
  a.batch = batch; a.n_grid = n_grid; a.steps_per_grid = steps_per_grid; a.max_iter = max_iter;
  a.ini_state = (const T*)ini_state; a.horizon = (const T*)horizon; a.auxvar = (const T*)auxvar;
  a.consts = consts ? (const T*)consts : (const T*)horizon;      // NC_REAL == 0: any readable word, never used
  a.const_stride = (consts && const_per_traj) ? Model::NC : 0;
  a.u_init = (const T*)u_init;
  a.u_lb = (const T*)control_lb; a.u_ub = (const T*)control_ub;
  a.x_lb = (const T*)state_lb; a.x_ub = (const T*)state_ub; a.x_mult = (const T*)state_mult; a.x_rho = (T)state_rho;
  a.state_grid = (T*)state_grid; a.control_grid = (T*)control_grid; a.costate_grid = (T*)costate_grid;
  a.cost = (T*)cost; a.iters = iters; a.status = status;
  a.ws = (T*)workspace; a.ws_stride = lfsd::OcLayout<Model>::template ws_elems<G>(n_grid);
  a.tol = (T)tol;
  a.exact_after = exact_after;
  a.mu_stage_frac = (exact_after == 0) ? (T)(LFSD_MU_STAGE_FRAC_NEWTON) : T(0);
  a.start_mode = start_mode;
  if (use_wide(sizeof(T) == 4 ? LFSD_F32 : LFSD_F64, batch, exact_after, mapping, control_lb != nullptr)) {       // bounded problems: the wide kernel at every batch size
    a.ws_stride = lfsd::OcLayout<Model>::ws_elems_wide(n_grid);
    const size_t solver_bytes = (size_t)batch * (size_t)a.ws_stride * sizeof(T);
    if (workspace_bytes < solver_bytes) return LFSD_ENOSPC;
    a.it_start = 0; a.resume = 0; a.max_iter_total = max_iter;
    a.sched = nullptr; a.suspend_at = 0; a.suspend_it = -1;
    if (control_lb) { LFSD_LAUNCH((lfsd::oc_solve_wide_kernel<Model, T, true, true>), (unsigned)batch, 64, stream, a); }
    else if (exact_after < 0) { LFSD_LAUNCH((lfsd::oc_solve_wide_kernel<Model, T, false>), (unsigned)batch, 64, stream, a); }
    else {
      if constexpr (wide_multi<T>()) {
        const int waves = env_int("LFSD_WIDE_WAVES", 0);
        const int cap = std::max(1, env_int("LFSD_WIDE_CAPACITY", device_cu_count()));
        if (waves == WIDE_W || (waves == 0 && batch <= cap)) {
          LFSD_LAUNCH((lfsd::oc_solve_wide_kernel<Model, T, true, false, WIDE_W>), (unsigned)batch, 64 * WIDE_W, stream, a);
          return launch_status();
        }
        if (waves == 0 && max_iter > 8 && workspace_bytes >= wide_sched_offset(solver_bytes) + wide_sched_bytes(batch)) {
          a.sched = (int*)((char*)workspace + wide_sched_offset(solver_bytes));
          a.suspend_at = batch - cap;
          a.suspend_it = env_int("LFSD_WIDE_SUSPEND_IT", -1);
          LFSD_ZERO(a.sched, 2 * sizeof(int), stream);
          LFSD_LAUNCH((lfsd::oc_solve_wide_kernel<Model, T, true>), (unsigned)batch, 64, stream, a);
          int rc = launch_status();
          if (rc) return rc;
          // (by the counter at most `cap` trajectories are handed over: all but suspend_at = batch - cap were finished; the test hook may hand over all)
          a.resume = 2;
          LFSD_LAUNCH((lfsd::oc_solve_wide_kernel<Model, T, true, false, WIDE_W>), (unsigned)(a.suspend_it >= 0 ? batch : cap), 64 * WIDE_W, stream, a);
          return launch_status();
        }
      }
      LFSD_LAUNCH((lfsd::oc_solve_wide_kernel<Model, T, true>), (unsigned)batch, 64, stream, a);
    }
    return launch_status();
  }
  const size_t need = (size_t)padded_batch(batch) * (size_t)a.ws_stride * sizeof(T);
  if (workspace_bytes < need) return LFSD_ENOSPC;
  const unsigned grid = (unsigned)(padded_batch(batch) / GPB);
  a.it_start = 0; a.resume = 0; a.max_iter_total = max_iter;
  // fp32: packed roll-out (+ MFMA sweep); fp64: one live column per lane of a 16-lane group (OcSolver::rollout_sens_live)
  constexpr bool PK = OC_PK && (sizeof(T) == 4 || lfsd::OcLayout<Model>::LIVE <= 16);
  const unsigned grid_lean = PK ? (unsigned)(padded_batch(batch) / OC_GPB) : grid;
  if (exact_after < 0) {                       // Gauss-Newton / Hamiltonian models only
    LFSD_LAUNCH((lfsd::oc_solve_kernel<Model, T, G, false, PK>), grid_lean, 64, stream, a);
    return launch_status();
  }
  if (exact_after == 0) {                      // Newton from the first iteration
    LFSD_LAUNCH((lfsd::oc_solve_kernel<Model, T, G, true>), grid, 64, stream, a);
    return launch_status();
  }
  // default: lean kernel for the first `exact_after` iterations, then the exact-capable kernel resumes
  // (warm-started from control_grid) the trajectories that are still at MAXITER
  a.max_iter = max_iter < exact_after ? max_iter : exact_after;
  if (max_iter <= exact_after) a.max_iter_total = 0;      // no phase 2: never hand over
  LFSD_LAUNCH((lfsd::oc_solve_kernel<Model, T, G, false, PK>), grid_lean, 64, stream, a);
  int rc = launch_status();
  if (rc || max_iter <= exact_after) return rc;
  a.max_iter = max_iter; a.it_start = exact_after; a.resume = 1;
  LFSD_LAUNCH((lfsd::oc_solve_kernel<Model, T, G, true>), grid, 64, stream, a);
  return launch_status();
}

// `u_init` (may be NULL): the caller's initial guess.  The fp32 solve starts from it too -- an all-zero row is a cold start for both
// kernels, so a learner that hands zeros for every row but the ones it continues (SparseDemoLearner with skip_unconverged) gets
// every row seeded -- and a row the fp32 solve failed on starts the fp64 kernel from the caller's row.
static int coc_solve_seeded(int batch, int n_grid, int steps_per_grid, const void* ini_state, const void* horizon,
                            const void* auxvar, const void* consts, int const_per_traj, const void* u_init, void* state_grid,
                            void* control_grid, void* costate_grid, void* cost, int* iters, int* status, int max_iter, double tol,
                            int exact_after, int mapping, void* workspace, size_t workspace_bytes, void* stream) {
  const size_t f64_bytes = (size_t)padded_batch(batch) * (size_t)lfsd::OcLayout<Model>::template ws_elems<G>(n_grid) * 8;
  const SeedLayout L(batch, n_grid, 1);
  if (workspace_bytes < L.total(f64_bytes)) return LFSD_ENOSPC;
  char* base = (char*)workspace + (f64_bytes + 255) / 256 * 256;
  auto cast = [&](const void* src, void* dst, long long n, int to_f32) {
    lfsd::CastArgs c{src, dst, n, to_f32};
    LFSD_LAUNCH(lfsd::cast_kernel<void>, (unsigned)((n + 255) / 256), 256, stream, c);
  };
  cast(ini_state, base + L.x0, (long long)batch * Model::NX, 1);
  cast(horizon, base + L.hz, batch, 1);
  cast(auxvar, base + L.th, (long long)batch * Model::NP, 1);
  if (consts) cast(consts, base + L.cs, const_per_traj ? (long long)batch * Model::NC : (long long)Model::NC, 1);
  if (u_init) cast(u_init, base + L.ui, (long long)batch * n_grid * Model::NU, 1);
  int rc = launch_status();
  if (rc) return rc;
  const double tol32 = tol > 1e-6 ? tol : 1e-6;      // (runtime.py's default for an fp32 solve)
  rc = coc_solve_t<float>(batch, n_grid, steps_per_grid, base + L.x0, base + L.hz, base + L.th, consts ? base + L.cs : nullptr,
                          const_per_traj, u_init ? base + L.ui : nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0, base + L.xs, base + L.us,
                          base + L.ls, base + L.cost, (int*)(base + L.it), (int*)(base + L.st), max_iter, tol32, exact_after,
                          LFSD_MAP_LOCKSTEP, base + L.ws32, L.x0 - L.ws32, stream);
  if (rc) return rc;
  {
    lfsd::SeedArgs sa{(const float*)(base + L.us), (const float*)(base + L.cost), (const int*)(base + L.st), (const double*)u_init,
                      (double*)(base + L.u0), batch, n_grid, Model::NU};
    const long long n = (long long)batch * n_grid * Model::NU;
    LFSD_LAUNCH(lfsd::seed_controls_kernel<void>, (unsigned)((n + 255) / 256), 256, stream, sa);
  }
  rc = coc_solve_t<double>(batch, n_grid, steps_per_grid, ini_state, horizon, auxvar, consts, const_per_traj, base + L.u0, nullptr,
                           nullptr, nullptr, nullptr, nullptr, 0.0, state_grid, control_grid, costate_grid, cost, iters, status,
                           max_iter, tol, exact_after, mapping, workspace, f64_bytes, stream, /*start_mode=*/1);
  if (rc) return rc;
  lfsd::AddItersArgs ia{iters, (const int*)(base + L.it), batch};      // iterations reported: both solves
  LFSD_LAUNCH(lfsd::add_iters_kernel<void>, (unsigned)((batch + 255) / 256), 256, stream, ia);
  return launch_status();
}

LFSD_API int lfsd_coc_solve(int dtype, int batch, int n_grid, int steps_per_grid, const void* ini_state,
                              const void* horizon, const void* auxvar, const void* consts, int const_per_traj,
                              const void* u_init, const void* control_lb, const void* control_ub,
                              const void* state_lb, const void* state_ub, const void* state_mult, double state_rho,
                              void* state_grid, void* control_grid, void* costate_grid, void* cost,
                              int* iters, int* status, int max_iter, double tol, int exact_after, int mapping,
                              void* workspace, size_t workspace_bytes, void* stream) {
  if (batch <= 0 || n_grid <= 0 || steps_per_grid <= 0 || max_iter < 0 || !(tol >= 0)) return LFSD_EINVAL;
  if (mapping < LFSD_MAP_AUTO || mapping > LFSD_MAP_WIDE) return LFSD_EINVAL;
  if (steps_per_grid > lfsd::OcLayout<Model>::SMAX) return LFSD_EINVAL;
  if (!ini_state || !horizon || !auxvar || !state_grid || !control_grid || !costate_grid || !cost || !iters ||
      !status || !workspace)
    return LFSD_EINVAL;
  if (Model::NC_REAL > 0 && !consts) return LFSD_EINVAL;
  if ((control_lb == nullptr) != (control_ub == nullptr)) return LFSD_EINVAL;
  // state bounds: all three arrays or none, a positive penalty, and the control-bound arrays beside them (the bounded
  // kernel reads both boxes; entries of +-1e20 mean "no bound")
  if (state_lb || state_ub || state_mult) {
    if (!state_lb || !state_ub || !state_mult || !control_lb || !(state_rho > 0)) return LFSD_EINVAL;
  }
  if (dtype == LFSD_F32)
    return coc_solve_t<float>(batch, n_grid, steps_per_grid, ini_state, horizon, auxvar, consts, const_per_traj, u_init,
                              control_lb, control_ub, state_lb, state_ub, state_mult, state_rho, state_grid, control_grid, costate_grid,
                              cost, iters, status, max_iter, tol, exact_after, mapping, workspace, workspace_bytes, stream);
  if (dtype == LFSD_F64)
    return coc_solve_t<double>(batch, n_grid, steps_per_grid, ini_state, horizon, auxvar, consts, const_per_traj,
                               u_init, control_lb, control_ub, state_lb, state_ub, state_mult, state_rho, state_grid, control_grid,
                               costate_grid, cost, iters, status, max_iter, tol, exact_after, mapping, workspace, workspace_bytes, stream);
  return LFSD_EINVAL;
}

template <typename T>
static int aux_phase_t(int phases, int batch, int n_grid, const void* horizon, const void* auxvar, const void* consts,
                       int const_per_traj, const void* state_grid, const void* control_grid, const void* costate_grid,
                       void* Z_grid, int n_waypoints, int n_iface, const int* iface_idx, const void* taus,
                       const void* waypoints, void* loss, void* grad, void* auxX_grid, void* auxU_grid, int substeps,
                       double rtol, int* stats, const int* oc_status, int skip_mask, void* stream) {
  lfsd::AuxArgs<T> a;
  // rtol > 0: error-controlled sub-stepping from `substeps` (default 1) units per interval upwards;
  // rtol = 0: a fixed minimum of `substeps` (default 4) units, refined for stiffness only (the round-1 behaviour)
  a.batch = batch; a.n_grid = n_grid; a.substeps = substeps > 0 ? substeps : (rtol > 0 ? 1 : 4);
  a.rtol = (T)rtol;
  // Stiffness cap of a split unit: dt * |Huu^-1 fu^T P fu| <= 8 / substeps (substeps = 4 -> 2), and at most max_refine x
  // `substeps` units per interval.  The last interval before a heavy final cost is the hard one: on the robot arm
  // (h_xx = 200, control weight 0.5) dgrid * rate is ~2000 there, the splitting is stiffly accurate to ~1e-4 at any unit
  // count below a few hundred, and leaves that plateau only once dt * rate < 1 -- thousands of units (profiles/
  // r04_d_dudtheta_refinement.txt).  The error control sees it (such an interval is reported in `stats` as accepted above
  // rtol when the cap binds), but with the default cap a tolerance below 1e-3 could not be met there: a tighter rtol therefore
  // tightens the stiffness cap (fourth root: the extrapolated pair is fourth order in dt * rate) and widens max_refine with
  // it.  At rtol >= 1e-3 -- the reference's own tolerance, the default -- nothing changes.
  const double tight = (rtol > 0 && rtol < 1e-3) ? std::pow(1e-3 / rtol, 0.25) : 1.0;
  a.rate_max = (T)(8.0 / a.substeps / tight);
  a.max_refine = 256;
  while (a.max_refine < 256 * tight && a.max_refine < 16384) a.max_refine *= 2;
  // a budget of 24 units per interval on average (twice what a well-posed robot-arm row spends, ten times a quadrotor row; 64 until
  // round 5: the diverged rows of one learner step then held both sweeps at 12-14 ms against 2.1 / 1.2) for rows whose OC solve did
  // not converge (status not 1 / 2; needs oc_status): rows
  // a fixed learning rate has driven to parameters of 1e14 refined EVERY interval to the cap (8 864 units against the 565 of
  // a well-posed robot-arm row: Riccati launch 35 ms against 2.1, profiles/r04_l_steps_robotarm_fast_trig.txt); what they
  // return is flagged in `stats`.  Converged rows are never budgeted (a stiff last interval alone may need thousands of units).
  a.unit_budget = (int)std::min<double>(24.0 * n_grid * a.substeps * tight, (double)(1LL << 30));      // (in double: the fourth-root factor must not truncate to 1)
  a.horizon = (const T*)horizon; a.auxvar = (const T*)auxvar;
  a.consts = consts ? (const T*)consts : (const T*)horizon;
  a.const_stride = (consts && const_per_traj) ? Model::NC : 0;
  a.state_grid = (const T*)state_grid; a.control_grid = (const T*)control_grid; a.costate_grid = (const T*)costate_grid;
  a.Z_grid = (T*)Z_grid;
  a.n_waypoints = n_waypoints; a.n_iface = n_iface; a.iface_idx = iface_idx;
  a.taus = (const T*)taus; a.waypoints = (const T*)waypoints;
  a.loss = (T*)loss; a.grad = (T*)grad; a.auxX_grid = (T*)auxX_grid; a.auxU_grid = (T*)auxU_grid;
  a.stats = stats;
  a.oc_status = oc_status; a.skip_mask = skip_mask;      // (the status also decides which rows the unit budget applies to)
  const unsigned grid = (unsigned)(((long long)batch + GPB - 1) / GPB);
  if (phases & 1) {
    int rc;
    if constexpr (sizeof(T) == 4) rc = lfsd_detail::launch_riccati_f32(grid, stream, a);
    else rc = lfsd_detail::launch_riccati_f64(grid, stream, a);
    if (rc) return rc;
  }
  if (phases & 2) {
    // the forward sweep packs more trajectories into a wavefront than the Riccati sweep (lfsd::fwd_lanes)
    constexpr int GF = lfsd::fwd_lanes<Model>() < G ? lfsd::fwd_lanes<Model>() : G;
    constexpr int GPBF = 64 / GF;
    const unsigned grid_f = (unsigned)(((long long)batch + GPBF - 1) / GPBF);
    LFSD_LAUNCH((lfsd::aux_forward_kernel<Model, T, GF>), grid_f, 64, stream, a);
    return launch_status();
  }
  return 0;
}

static int aux_dispatch(int phases, int dtype, int batch, int n_grid, const void* horizon, const void* auxvar,
                        const void* consts, int const_per_traj, const void* state_grid, const void* control_grid,
                        const void* costate_grid, void* Z_grid, int n_waypoints, int n_iface, const int* iface_idx,
                        const void* taus, const void* waypoints, void* loss, void* grad, void* auxX_grid,
                        void* auxU_grid, int substeps, double rtol, int* stats, const int* oc_status, int skip_mask,
                        void* stream) {
  if (batch <= 0 || n_grid <= 0 || n_waypoints < 0 || n_iface < 0 || substeps < 0 || !(rtol >= 0)) return LFSD_EINVAL;
  if (!horizon || !auxvar || !state_grid || !control_grid || !costate_grid || !Z_grid) return LFSD_EINVAL;
  if ((phases & 2) && (!loss || !grad)) return LFSD_EINVAL;
  if ((phases & 2) && n_waypoints > 0 && (n_iface <= 0 || !taus || !waypoints)) return LFSD_EINVAL;
  // iface_idx == NULL selects the interface function compiled into the library: it must exist and n_iface must be its size
  if ((phases & 2) && n_waypoints > 0 && !iface_idx && (Model::NIF == 0 || n_iface != Model::NIF)) return LFSD_EINVAL;
  if (Model::NC_REAL > 0 && !consts) return LFSD_EINVAL;
  if (skip_mask < 0 || (skip_mask != 0 && !oc_status)) return LFSD_EINVAL;
  if (dtype == LFSD_F32)
    return aux_phase_t<float>(phases, batch, n_grid, horizon, auxvar, consts, const_per_traj, state_grid, control_grid,
                              costate_grid, Z_grid, n_waypoints, n_iface, iface_idx, taus, waypoints, loss, grad,
                              auxX_grid, auxU_grid, substeps, rtol, stats, oc_status, skip_mask, stream);
  if (dtype == LFSD_F64)
    return aux_phase_t<double>(phases, batch, n_grid, horizon, auxvar, consts, const_per_traj, state_grid,
                               control_grid, costate_grid, Z_grid, n_waypoints, n_iface, iface_idx, taus, waypoints,
                               loss, grad, auxX_grid, auxU_grid, substeps, rtol, stats, oc_status, skip_mask, stream);
  return LFSD_EINVAL;
}

LFSD_API int lfsd_aux_solve(int dtype, int batch, int n_grid, const void* horizon, const void* auxvar,
                            const void* consts, int const_per_traj, const void* state_grid, const void* control_grid,
                            const void* costate_grid, void* Z_grid, int n_waypoints, int n_iface,
                            const int* iface_idx, const void* taus, const void* waypoints, void* loss, void* grad,
                            void* auxX_grid, void* auxU_grid, int substeps, double rtol, int* stats,
                            const int* oc_status, int skip_status_mask, void* stream) {
  return aux_dispatch(3, dtype, batch, n_grid, horizon, auxvar, consts, const_per_traj, state_grid, control_grid,
                      costate_grid, Z_grid, n_waypoints, n_iface, iface_idx, taus, waypoints, loss, grad, auxX_grid,
                      auxU_grid, substeps, rtol, stats, oc_status, skip_status_mask, stream);
}

LFSD_API int lfsd_aux_riccati(int dtype, int batch, int n_grid, const void* horizon, const void* auxvar,
                              const void* consts, int const_per_traj, const void* state_grid,
                              const void* control_grid, const void* costate_grid, void* Z_grid, int substeps,
                              double rtol, int* stats, const int* oc_status, int skip_status_mask, void* stream) {
  return aux_dispatch(1, dtype, batch, n_grid, horizon, auxvar, consts, const_per_traj, state_grid, control_grid,
                      costate_grid, Z_grid, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                      substeps, rtol, stats, oc_status, skip_status_mask, stream);
}

LFSD_API int lfsd_aux_forward(int dtype, int batch, int n_grid, const void* horizon, const void* auxvar,
                              const void* consts, int const_per_traj, const void* state_grid,
                              const void* control_grid, const void* costate_grid, const void* Z_grid, int n_waypoints,
                              int n_iface, const int* iface_idx, const void* taus, const void* waypoints, void* loss,
                              void* grad, void* auxX_grid, void* auxU_grid, int substeps, double rtol, int* stats,
                              const int* oc_status, int skip_status_mask, void* stream) {
  return aux_dispatch(2, dtype, batch, n_grid, horizon, auxvar, consts, const_per_traj, state_grid, control_grid,
                      costate_grid, const_cast<void*>(Z_grid), n_waypoints, n_iface, iface_idx, taus, waypoints, loss,
                      grad, auxX_grid, auxU_grid, substeps, rtol, stats, oc_status, skip_status_mask, stream);
}

template <typename T>
static int opt_step_t(int method, int batch, int n_param, int iter_idx, double lr, double mu, double beta1,
                      double beta2, double eps, void* theta, const void* grad, void* m, void* v, void* vhat,
                      const void* proj_lo, const int* row_active, void* stream) {
  lfsd::OptArgs<T> a;
  a.batch = batch; a.n_param = n_param; a.method = method; a.iter_idx = iter_idx;
  a.lr = (T)lr; a.mu = (T)mu; a.beta1 = (T)beta1; a.beta2 = (T)beta2; a.eps = (T)eps;
  a.theta = (T*)theta; a.grad = (const T*)grad; a.m = (T*)m; a.v = (T*)v; a.vhat = (T*)vhat;
  a.proj_lo = (const T*)proj_lo; a.row_active = row_active;
  const long long n = (long long)batch * n_param;
  const unsigned grid = (unsigned)((n + 63) / 64);
  LFSD_LAUNCH((lfsd::optimizer_kernel<T>), grid, 64, stream, a);
  return launch_status();
}

LFSD_API int lfsd_optimizer_step(int dtype, int method, int batch, int n_param, int iter_idx, double lr, double mu,
                                   double beta1, double beta2, double eps, void* theta, const void* grad, void* m,
                                   void* v, void* vhat, const void* proj_lo, const int* row_active, void* stream) {
  if (batch <= 0 || n_param <= 0 || iter_idx < 0 || !theta || !grad) return LFSD_EINVAL;
  if (method < LFSD_OPT_VANILLA || method > LFSD_OPT_AMSGRAD) return LFSD_EINVAL;
  if (method == LFSD_OPT_NESTEROV && !m) return LFSD_EINVAL;
  if (method >= LFSD_OPT_ADAM && (!m || !v)) return LFSD_EINVAL;
  if (method == LFSD_OPT_AMSGRAD && !vhat) return LFSD_EINVAL;
  if (dtype == LFSD_F32)
    return opt_step_t<float>(method, batch, n_param, iter_idx, lr, mu, beta1, beta2, eps, theta, grad, m, v, vhat,
                             proj_lo, row_active, stream);
  if (dtype == LFSD_F64)
    return opt_step_t<double>(method, batch, n_param, iter_idx, lr, mu, beta1, beta2, eps, theta, grad, m, v, vhat,
                              proj_lo, row_active, stream);
  return LFSD_EINVAL;
}

LFSD_API int lfsd_lookahead(int dtype, long long n, double mu, const void* theta, const void* v, void* out,
                              void* stream) {
  if (n <= 0 || !theta || !v || !out) return LFSD_EINVAL;
  const unsigned grid = (unsigned)((n + 63) / 64);
  if (dtype == LFSD_F32) {
    const float muf = (float)mu; const float* th = (const float*)theta; const float* vv = (const float*)v; float* o = (float*)out;
#if defined(LFSD_EMU)
    emu::launch(dim3(grid), dim3(64), [&] { lfsd::lookahead_kernel<float>(n, muf, th, vv, o); });
#else
    hipLaunchKernelGGL((lfsd::lookahead_kernel<float>), dim3(grid), dim3(64), 0, (hipStream_t)stream, n, muf, th, vv, o);
#endif
    return launch_status();
  }
  if (dtype == LFSD_F64) {
    const double* th = (const double*)theta; const double* vv = (const double*)v; double* o = (double*)out;
#if defined(LFSD_EMU)
    emu::launch(dim3(grid), dim3(64), [&] { lfsd::lookahead_kernel<double>(n, mu, th, vv, o); });
#else
    hipLaunchKernelGGL((lfsd::lookahead_kernel<double>), dim3(grid), dim3(64), 0, (hipStream_t)stream, n, mu, th, vv, o);
#endif
    return launch_status();
  }
  return LFSD_EINVAL;
}
