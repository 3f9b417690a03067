// Build switches, packed/MFMA primitives and the small dense helpers shared by every kernel.
// Part of the kernel sources collected by cpdp_kernels.h (include that header, not this one).
#pragma once


#include <type_traits>
#if defined(LFSD_EMU)
#include "simt_emu.h"
#define LFSD_LAMBDA_INLINE
#define LFSD_LAMBDA_BW
#define LFSD_LAMBDA_RO
#define LFSD_HD
#else
#include <hip/hip_runtime.h>
#define LFSD_DEV __device__ __forceinline__
#define LFSD_HD __host__ __device__
// lambdas inside kernels must be inlined as well: a real call passes their by-reference captures through scratch
#define LFSD_LAMBDA_INLINE __attribute__((always_inline))
#define LFSD_LAMBDA_BW __attribute__((always_inline))
#define LFSD_LAMBDA_RO __attribute__((always_inline))
#endif

// register budget: measured on MI355X (tools/tune.py) 1 wave/SIMD with all 512 VGPR+AGPR beats 2-3 waves with scratch spills
#ifndef LFSD_WAVES_PER_SIMD
#define LFSD_WAVES_PER_SIMD 1
#endif
#ifndef LFSD_WAVES_OC
#define LFSD_WAVES_OC LFSD_WAVES_PER_SIMD
#endif
// aux kernels and occupancy, measured on MI355X.  Compiled with clang's SLP vectoriser (profiles/r01_tune_aux_occupancy.txt)
// a second wave per SIMD never paid: held to 256 VGPRs the kernels spilled 0.5-1.3 KB/lane (Riccati 8.8 -> 9.0 ms, forward
// 5.8 -> 11.1 ms).  Without SLP (profiles/r01_tune_compiler_flags.txt) the fp32 Riccati sweep needs 256 + 84 registers: with
// the small column cache (LFSD_RIC_CACHE 2) it runs two waves per SIMD -- the 2048 waves of the benchmark batch in one
// round instead of two, 8.2 -> 7.1 ms with 0.4 KB/lane of spills, and 5.7 ms with none (246 VGPRs) once the coarse and the
// fine Richardson chain run in place with the other column parked in LDS.  The forward sweep stays at one wave per SIMD
// (256 + 256 registers) and keeps its two chains as independent instruction streams: in place it is 20 % slower.
// Error-controlled sub-stepping of the auxiliary sweeps: the next interval starts with HALF the units of this one when
// this one's worst per-unit estimate is at most 1/LFSD_AUX_DOWN of its tolerance.  The estimate is the local error of
// one split unit, O(h^3): measured on the headline workload it grows 7.4-8x when the units are halved
// (profiles/r02_h_aux_units.txt), so an 8-fold margin predicts <= 0.93-1.0 of the tolerance after the halving.  (It was
// 32 until that trace showed every trajectory integrating at 2-4x the units its tolerance asked for.  Measured on the
// benchmark: 10 -> Riccati 2.28 ms, 8 -> 2.14 ms, 6 -> 2.28 ms again: below 8 the halved intervals fail their test and are
// redone.)
// Non-stiff part of a split unit of the auxiliary sweeps: 4 = classical RK4, 2 = explicit midpoint rule.  The Strang
// splitting around it is second order either way; with RK4 the Richardson pair is (nearly) symmetric and the extrapolation
// gains two orders, with the midpoint rule one -- at half the right-hand sides per unit.  Measured on MI355X
// (profiles/r02_m_substeps_accuracy.txt, r02_m_*): in fp32 the sweeps' rounding floor (1e-5 relative) hides the
// difference -- gradient error vs the tight oracle 1.2e-4 at rtol 1e-3 with both, Riccati 2.12 -> 1.47 ms, forward
// 1.08 -> 0.91 ms on the benchmark -- so the fp32 kernels use the midpoint rule.  In fp64 (parity reference; the rocket's
// auxiliary pass) RK4 stays: the midpoint rule cost the robot arm's large-sensitivity seeds a factor 3 in accuracy and
// the rocket 12 % in time (more units).
#ifndef LFSD_AUX_RK32
#define LFSD_AUX_RK32 2
#endif
#ifndef LFSD_AUX_RK64
#define LFSD_AUX_RK64 4
#endif
#ifndef LFSD_AUX_DOWN
#define LFSD_AUX_DOWN 8
#endif
// Riccati sweep: intervals that need at least this many uniform units are integrated with step-size control inside the interval
// (cpdp_aux.h, aux_riccati_kernel); a very large value switches it off
#ifndef LFSD_RIC_ADAPT
#define LFSD_RIC_ADAPT 16
#endif
// outer per-node loops of the once-per-unit preparation (ric_cols; fwd_prep, fwd_cols): rolled.  Measured: 6 % faster in
// the Riccati sweep; the forward sweep preferred them unrolled (7 %) until it was compiled with the max-ILP scheduler,
// since then rolled is 5 % faster there too (profiles/r01_tune_aux_occupancy.txt, r01_tune_compiler_flags.txt)
#ifndef LFSD_RIC_NODE_LOOP
#define LFSD_RIC_NODE_LOOP _Pragma("unroll 1")
#endif
#ifndef LFSD_FWD_NODE_LOOP
#define LFSD_FWD_NODE_LOOP _Pragma("unroll 1")
#endif
// forward auxiliary sweep: operands of the stiff step / right-hand side fetched into registers with back-to-back LDS reads
// (lds_fetch) before the sparse operators run -- bit 0: fp32 instantiation, bit 1: fp64.  Measured on MI355X (profiles/
// r04_b): fp32 0.849 -> 0.625 ms (185 exposed LDS round trips per split unit become 7 batches); fp64 1.99 -> 2.30 ms (the
// 96 + 13 doubles of a right-hand side do not fit beside the chain's state: scratch 144 -> 212 B/lane), so fp64 keeps
// reading its operands where it uses them.
// (LFSD_FWD_FETCH: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
#ifndef LFSD_WAVES_RIC
#define LFSD_WAVES_RIC 2
#endif
#ifndef LFSD_WAVES_FWD
#define LFSD_WAVES_FWD 1
#endif

// Levenberg shift ladder of the OC solve: factor up after a failed backward sweep / line search, factor down after an
// accepted full step, and how many accepted full steps to hold before the shift returns to a level that has just failed.
// Measured on BASELINE configs[1] (robot arm, 1024 seeds, tools/tune_arm.py, profiles/r01_tune_step_control.txt): with
// x10 / x0.1 the accepted steps land on shifts up to 10x larger than necessary (over-damped) and every other backward
// sweep fails; sqrt(10) rungs + a one-step hold cut the slowest seed from 87 to 65 iterations and the seeds that run
// out of iterations at the step-1 parameters from 98 to 65 of 1024.  Handing over to the exact model as soon as
// Gauss-Newton crawls (the oracle's rule, LFSD_GN_CRAWL) costs iterations here (27 -> 40 on average) and stays off.
#ifndef LFSD_MU_UP
#define LFSD_MU_UP 3.1623
#endif
#ifndef LFSD_MU_DOWN
#define LFSD_MU_DOWN 0.31623
#endif
// generic backward sweep of a solve that runs Newton from its first iteration (exact_after == 0, the rocket): fraction of the
// Levenberg shift a stage keeps when its Q_uu factorises with it (0: one shift for all stages).  Measured on the rocket learner
// step with the coarse time grid: 0 -> 207 ms, 0.01 -> 191, 0.003 -> 149, 0.001 -> 150, 1e-4 -> 214 (profiles/r04_s_*).  Solves
// that start with Gauss-Newton sweeps keep one shift: the robot arm's step went 28.4 -> 42.9 ms with a per-stage fraction.
#ifndef LFSD_MU_STAGE_FRAC_NEWTON
#define LFSD_MU_STAGE_FRAC_NEWTON 0.001
#endif
#ifndef LFSD_MU_HOLD
#define LFSD_MU_HOLD 1
#endif
// (LFSD_GN_CRAWL: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// Levenberg shift of the Newton modes, measured on BASELINE configs[1] (robot arm, 1024 seeds, emulator + MI355X,
// profiles/r02_arm_step_control.txt).  LFSD_REG_CONSISTENT: the value recursion continues with the SHIFTED Q_uu, i.e. the
// sweep is the block LDL^T factorisation of (Lagrangian Hessian + mu I_u) -- the model the step actually minimises, and
// what IPOPT's inertia correction delta_w does to the KKT matrix (CPDP.py:177-184).  With the unshifted Q_uu in the
// recursion (Tassa's form) an indefinite Q_uu feeds -mu K^T K into V_xx, the sweep needs shifts 30x larger, and 64 of
// 1024 seeds ran out of 100 iterations.  LFSD_HAM_SHIFT: shift the cheap Hamiltonian model as well instead of falling
// back to Gauss-Newton (measured: no gain).
// backward sweep: issue the loads of interval k-1 while interval k is processed (1; costs NX+1 + NX+NU registers), or load
// each interval when the sweep gets there (0).  Round 2 measured no difference and kept 1 -- the compiler had sunk the
// prefetch to its first use (LFSD_ISSUE_FENCE below); with the loads really issued ahead (r03_g_ab_prefetch_pin.txt)
// oc_solve takes 2.65 ms against 2.59 ms without the prefetch: the 23 registers it holds through the stage cost more
// accumulator-register traffic than the loads' latency does beside 7000 cycles of stage work.
// the same look-ahead in the GENERIC backward sweep (OcSolver::backward: wide kernel, lock-step kernels without MFMA, fp64):
// 1 = fp32 instantiations, 3 = fp64 too
#ifndef LFSD_BW_PREFETCH_GEN
#define LFSD_BW_PREFETCH_GEN 1
#endif
// stages of look-ahead of the wide kernel's costate sweep (0: load where used)
#ifndef LFSD_CS_AHEAD
#define LFSD_CS_AHEAD 3
#endif
// lean fp32 kernel of the 32-lane models: backward sweep on the matrix cores (1) or relayed on the vector pipe (0)
#ifndef LFSD_MFMA_BACKWARD
#define LFSD_MFMA_BACKWARD 1
#endif
// Gauss-Newton -> Hamiltonian (cheap Newton-like) stage-Hessian model: after the first accepted full step whose gain is
// below this fraction of the new cost ("past the first big drops").  Measured on the benchmark: 0.9 changes nothing (6
// iterations per solve either way), 3.0 -- switching right after the first step -- costs 2 % of the trajectories a 7th.
#ifndef LFSD_HAM_SWITCH
#define LFSD_HAM_SWITCH 0.3
#endif
// lean fp32 kernel of the 32-lane models: leave the structurally constant tangent columns of the model (Model::NZC, e.g.
// the quadrotor's position) out of the roll-out, the workspace and the matrix products (1), or treat every column alike (0)
#ifndef LFSD_STRUCT_COLS
#define LFSD_STRUCT_COLS 1
#endif
// Mesh continuation of the lean fp32 OC kernel (cpdp_oc.h, oc_solve_kernel): one RK4 step per grid interval while full
// steps gain more than LFSD_COARSE_SWITCH of the cost, then the reference's steps_per_grid.  LFSD_COARSE_START 0 switches it
// off.  Measured on the benchmark (profiles/r03_d_ab_coarse_exit.txt, r03_e_ab_coarse_switch.txt), oc_solve per launch: off
// 3.79 ms; switch at 3.0 (after the first step) 3.20; 0.3 (with the Gauss-Newton -> Hamiltonian hand-over) 2.93; 1e-2 2.94;
// 1e-3 (one Newton-like step more on the coarse grid) 2.85; 1e-4 2.98; 0 (only when the coarse problem has converged) 3.66.
#ifndef LFSD_COARSE_START
#define LFSD_COARSE_START 1
#endif
#ifndef LFSD_COARSE_SWITCH
#define LFSD_COARSE_SWITCH 1e-3
#endif
// lean matrix-core kernels: level 0 of the mesh continuation merges LFSD_LEAN_TC control intervals (1: off) with LFSD_LEAN_TC_S RK4
// steps per merged interval for the first LFSD_LEAN_TC_ITERS iterations of a cold start; never below LFSD_LEAN_TC_MIN intervals.
// Measured on BOTH workloads of the benchmark (oc_solve per launch, 20 steps; profiles/r04_z_ab_lean_time_coarsening.txt) --
// independent seeds (one start / goal, 4 096 parameter vectors: a homogeneous batch whose trajectories all walk the same path) |
// shared parameters (4 096 random starts / goals / waypoints, the N > 1 workload: the launch lasts as long as its slowest path):
//   off 2.57 | 2.70;   2 x 1 step for 2 / 3 / 4 iterations 2.34 / 2.40 / 2.62 | 2.48 / 2.40 / 2.96;
//   5 x 2 for 2 / 3 iterations 2.45 / 2.16 | 2.43 / 2.90;   5 x 1 for 2 / 3 iterations 2.46 / 2.10 | 2.40 / 2.85.
// 5 merged intervals for 3 iterations is the best on the homogeneous batch and WORSE than no level 0 on the diverse one; shipped is
// the setting that gains on both, 2 x 1 x 3.
#ifndef LFSD_LEAN_TC
#define LFSD_LEAN_TC 2
#endif
#ifndef LFSD_LEAN_TC_S
#define LFSD_LEAN_TC_S 1
#endif
#ifndef LFSD_LEAN_TC_ITERS
#define LFSD_LEAN_TC_ITERS 3
#endif
#ifndef LFSD_LEAN_TC_MIN
#define LFSD_LEAN_TC_MIN 12      // (round 5, held-out A/B: with 10 merged intervals -- n_grid 20 -- level 0 cost 14 % instead of saving; 13 and 15 save: profiles/r05_held_out_schedule_ab.jsonl)
#endif
// lean kernels: the convergence histories (last gradient norm, last predicted decrease) survive the step that leaves the coarse grid
// when the coarse problem had converged (cpdp_oc.h); 0 = they always start over on the reference's grid
#ifndef LFSD_EXIT_KEEP_HISTORY
#define LFSD_EXIT_KEEP_HISTORY 2      // 1: only when the coarse problem had passed a convergence test; 2: also when the exit step predicted a decrease below the cost's resolution
#endif
// lean kernels: accepted steps after the transfer from level 0 during which a refused or shortened full step is answered by a line
// search on the one-step-per-interval level instead of by leaving for the reference's grid (profiles/r04_bc_ab_grace_after_transfer.txt)
#ifndef LFSD_LEAN_TC_GRACE
#define LFSD_LEAN_TC_GRACE 2
#endif
// (a coarse phase for the wide kernel's models below 32 lanes was measured on the robot arm and removed: 21.2 -> 12.7 ms with the same
//  minima on 1 023 of 1 024 seeds, but one seed follows the coarse discretisation to a minimum the reference's does not have:
//  profiles/HISTORY.md, r04_ah_robotarm_coarse_phase_ab.txt)
// wide kernel: the interval-parallel (multiple-shooting) iteration of OcWide::ms_* (cpdp_oc.h) -- the reference's own lifted
// formulation, CPDP.py:136-172 -- for unbounded problems with at least LFSD_MS_MIN_GRID intervals; 0: single shooting only.
// The phase hands over to the single-shooting iteration through a closed-loop roll-out, so every convergence test is unchanged.
#ifndef LFSD_MS
#define LFSD_MS 1
#endif
#ifndef LFSD_MS_MIN_GRID
#define LFSD_MS_MIN_GRID 40
#endif
// ... consecutive accepted SHORT steps (step length < 1) after which the phase ends on the current level (cpdp_oc.h)
// wide kernel, fp32, models with at most 8 columns of [A B]: all columns of an interval's exact stage Hessian on one lane (1,
// OcSolver::stage_hessian_all) or one (interval, column) item per lane (0)
// (LFSD_HESS_ALL: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// (LFSD_LEAN_CTL_PREFETCH: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// ... shorter multiple-shooting steps: after a refused full step, at most LFSD_MS_HALF HALF steps in a row are tried before the
// closed-loop roll-outs (0: none).  A trial costs 25 k clocks against a roll-out's 580 k.  Measured on the robot-arm learner
// (1 024 seeds, MI355X, oc_solve of outer iterations 0 / 1 / 4 / 5 / 6 / 7; profiles/r05_o_*): none 12.5 / 11.1 / 17.8 / 7.5 / 7.9 / 8.2 ms,
// one 10.9 / 10.2 / 8.7 / 7.2 / 5.5 / 5.2, two in a row 9.6 / 9.8 / 9.6 / 8.2 / 5.7 / 5.5.  (A full line search along the linear
// direction -- built first, removed -- lets the gaps pile up: 1 % of the trajectories then need 45-50 iterations.)
// (LFSD_MS_HALF: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// (LFSD_MS_JFEAS: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// (LFSD_MS_NEWTON: shipped as off since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// wide kernel (one trajectory per wavefront): smallest n_grid with a coarse phase
#ifndef LFSD_COARSE_MIN_GRID
#define LFSD_COARSE_MIN_GRID 40
#endif
// wide kernel: control intervals merged in the coarse phase (1: none, the coarse phase only takes one RK4 step per interval), and
// the smallest coarse grid it may produce
#ifndef LFSD_COARSE_TIME
#define LFSD_COARSE_TIME 4
#endif
// wide kernel: 1 = between the merged-interval coarse level and the reference's grid, a level with one RK4 step per interval
// (LFSD_COARSE_MID_LEVEL: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// RK4 steps per merged interval of that phase (0: as many as merged intervals, i.e. the step of one step per interval)
#ifndef LFSD_COARSE_TIME_S
#define LFSD_COARSE_TIME_S 2
#endif
#ifndef LFSD_COARSE_TIME_MIN
#define LFSD_COARSE_TIME_MIN 20
#endif
// lfsd_coc_solve in fp64, lock-step mapping, 32-lane models, cold start: solve in fp32 first and start the fp64 kernel from those
// controls with the Hamiltonian model (lfsd_capi.cpp, coc_solve_seeded).  Quadrotor headline in fp64: 9.8 -> see DESIGN section 3.1
#ifndef LFSD_F64_SEED
#define LFSD_F64_SEED 1
#endif
// wide kernel: when an accepted step that gains less than LFSD_COARSE_SWITCH of the cost ends the coarse phase (cpdp_oc.h)
#ifndef LFSD_COARSE_EXIT_RULE
#define LFSD_COARSE_EXIT_RULE 2
#endif
#ifndef LFSD_COARSE_EXIT_MU
#define LFSD_COARSE_EXIT_MU 1e-2
#endif
// (LFSD_REG_CONSISTENT: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)

// ---- fp64 OC kernels (round 3; profiles/r03_o_fp64_backward.txt, r03_q_fp64_live_park.txt) --------------------------------
// fp64 lean OC kernel of the 32-lane models on 16-lane groups (four trajectories per wavefront); 0: 32-lane groups (round 2)
// (LFSD_FP64_LIVE: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// ... with the between-stage values of its tangent RK4 step parked in LDS (OcSolver::rk4_step_parked)
// (LFSD_FP64_PARK: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// ... and with the one-pass backward sweep of the structural-column layout (backward_sc, LDS-fed products) instead of two
// passes of the generic sweep on 32-lane groups
// (LFSD_FP64_SC: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// LFSD_FENCE64, bits: 1 = pin64 (row sums of the backward sweep's dense products are materialised where they are computed),
// 2 = LFSD_SCHED_FENCE64 (a scheduling barrier per row), 4 = two row buffers (a row's LDS reads are issued while the previous
// row is multiplied).  The pathology they remove: Q = [A B]^T Y is first used behind a branch; the compiler sank its FMAs
// there, left the 221 LDS reads of their operands where they were, spilled every operand to scratch and reloaded it one at a
// time -- 95 000 clocks for a stage that computes for 5 000, 66 % of oc_solve<double> (tools/oc_clock64.py).
#ifndef LFSD_FENCE64
#define LFSD_FENCE64 7
#endif
// ---- the generic backward sweep in fp32 (wide kernel, lock-step kernels without MFMA; profiles/r03_t_generic_backward.txt) --
// wide kernel, models with at most 16 columns of [A B] and NX >= 8 (rocket): rows of the backward sweep's dense products
// split over the four 16-lane quarters of the wavefront (OcSolver::backward, QS)
// wide kernel, small fp32 models (at most 8 columns, NX * NXU <= 32): the backward sweep on LDS-staged operands, every lane running
// the whole recursion (OcWide::backward_small)
#ifndef LFSD_BW_SMALL
#define LFSD_BW_SMALL 1
#endif
// (LFSD_BW_QSPLIT: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// two row buffers for the dense products, fp32 instantiations with NX >= 8 (fp64: LFSD_FENCE64 & 4)
// (LFSD_BW_ROWBUF32: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
// LFSD_ROW_FENCE / LFSD_SCHED_FENCE64(T): the instruction scheduler moves nothing across this point (bounds the live ranges of
// the fully unrolled contractions); LFSD_SCHED_FENCE: the same between the RK4 stages, off unless LFSD_USE_SCHED_FENCE is
// defined.  No-ops in the emulator build.
#if defined(LFSD_EMU)
#define LFSD_ROW_FENCE()
#define LFSD_SCHED_FENCE64(T)
#else
#define LFSD_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)
#define LFSD_SCHED_FENCE64(T) do { if constexpr (((LFSD_FENCE64) & 2) != 0 && sizeof(T) == 8) __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

// LFSD_WAVE_SYNC: LDS hand-over between lanes of ONE wavefront (kernels whose workgroup is a single wavefront: the
// auxiliary sweeps).  Release/acquire fences restricted to the LDS address space (s_waitcnt lgkmcnt(0); global loads in
// flight are not drained) around a wave_barrier, which emits nothing and only keeps the compiler from moving LDS accesses
// across it.  Unlike __syncthreads() it is well defined under control flow that differs between the lanes of the wavefront.
// The CPU emulator runs every lane as a fiber and needs a real rendez-vous there.
#if defined(LFSD_EMU)
#define LFSD_WAVE_SYNC() __syncthreads()
#else
#define LFSD_WAVE_SYNC()                                            \
  do {                                                              \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); \
    __builtin_amdgcn_wave_barrier();                                \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); \
  } while (0)
#endif

// LFSD_STAGE_SYNC: the hand-overs INSIDE a stage of the lean kernels' backward sweep (one wavefront per workgroup; every one of
// them goes through LDS).  __syncthreads() is a workgroup fence over ALL address spaces: on this hardware it compiles to
// `s_waitcnt vmcnt(0) lgkmcnt(0)` (the s_barrier of a one-wave workgroup is elided), so the sync that follows the stage's
// global stores of the gains parks the wavefront until those stores are acknowledged by the L2 -- once per stage, ~1 600 of a
// stage's 7 200 cycles by the phase clocks (profiles/r03_g_ab_prefetch_pin.txt, "K+Vupdate"), and it drains whatever was
// prefetched for the next stage with it.  The gains are read by the NEXT roll-out, behind the block-wide votes of the main
// loop (real __syncthreads); nothing inside the sweep reads them back.  LFSD_OC_LDS_SYNC 1: LDS-scoped fence there instead,
// and the stage's loads are waited for at its top (cpdp_oc.h, backward_sc).  MEASURED (profiles/r04_j_ab_stage_sync.txt): the
// `s_waitcnt vmcnt(0)` behind the stores is gone from the stage's ISA and oc_solve takes exactly as long -- 2.587 ms against
// 2.589 with the old syncs, 2.629 with the next stage's loads prefetched on top (LFSD_BW_PREFETCH 1): the stores are long
// acknowledged when the wait is reached; the sweep is bound by issue, as round 3 concluded.  Kept: it is the narrower fence.
// (LFSD_OC_LDS_SYNC: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
#if defined(LFSD_EMU)
#define LFSD_STAGE_SYNC() __syncthreads()
#else
#define LFSD_STAGE_SYNC() LFSD_WAVE_SYNC()
#endif
// ... the same inside a stage of the GENERIC backward sweep (wide kernel, lock-step kernels without MFMA): there the next
// stage's global loads ARE issued a stage ahead (LFSD_BW_PREFETCH_GEN), and every __syncthreads() of the stage drained them.
// Measured (profiles/r04_k_ab_generic_stage_sync.txt): rocket 107.8 / 107.5 ms, robot arm 17.8 / 17.7 ms with / without: nothing.
// (LFSD_OC_LDS_SYNC_GEN: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
#if defined(LFSD_EMU)
#define LFSD_STAGE_SYNC_GEN() __syncthreads()
#else
#define LFSD_STAGE_SYNC_GEN() LFSD_WAVE_SYNC()
#endif

namespace lfsd {

// debug aid for the emulator build: start every kernel with NaN-filled LDS so that a read of
// never-written shared memory cannot go unnoticed (tests build with -DLFSD_POISON_LDS)
template <typename T> LFSD_DEV void poison_lds(T* p, int n) {
#if defined(LFSD_POISON_LDS)
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = T(0) / T(0);
  __syncthreads();
#else
  (void)p; (void)n;
#endif
}

enum Status { ST_RUNNING = 0, ST_CONVERGED = 1, ST_STALLED = 2, ST_MAXITER = 3, ST_FAILED = 4 };
enum OptMethod { OPT_VANILLA = 0, OPT_NESTEROV = 1, OPT_ADAM = 2, OPT_NADAM = 3, OPT_AMSGRAD = 4 };

// Two values per lane, for the kernels that carry two tangent columns on one lane: native 2-vectors on the GPU (the
// tangent code is linear, so it compiles to v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 at the scalar issue rate), a plain
// struct in the CPU emulator build.
#if defined(LFSD_EMU)
template <typename T> struct pk2 {
  T x, y;
  pk2() = default;
  pk2(T s) : x(s), y(s) {}
  pk2(T a, T b) : x(a), y(b) {}
};
template <typename T> inline pk2<T> operator+(pk2<T> a, pk2<T> b) { return pk2<T>(a.x + b.x, a.y + b.y); }
template <typename T> inline pk2<T> operator-(pk2<T> a, pk2<T> b) { return pk2<T>(a.x - b.x, a.y - b.y); }
template <typename T> inline pk2<T> operator-(pk2<T> a) { return pk2<T>(-a.x, -a.y); }
template <typename T> inline pk2<T> operator*(pk2<T> a, pk2<T> b) { return pk2<T>(a.x * b.x, a.y * b.y); }
template <typename T> inline pk2<T> operator*(T a, pk2<T> b) { return pk2<T>(a * b.x, a * b.y); }
template <typename T> inline pk2<T> operator*(pk2<T> a, T b) { return pk2<T>(a.x * b, a.y * b); }
template <typename T> inline pk2<T> operator+(T a, pk2<T> b) { return pk2<T>(a + b.x, a + b.y); }
template <typename T> inline pk2<T> operator+(pk2<T> a, T b) { return pk2<T>(a.x + b, a.y + b); }
template <typename T> inline pk2<T> operator/(pk2<T> a, T b) { return pk2<T>(a.x / b, a.y / b); }
template <typename T> inline pk2<T> operator/(pk2<T> a, pk2<T> b) { return pk2<T>(a.x / b.x, a.y / b.y); }
template <typename T> inline pk2<T> operator-(pk2<T> a, T b) { return pk2<T>(a.x - b, a.y - b); }
template <typename T> inline pk2<T> operator-(T a, pk2<T> b) { return pk2<T>(a - b.x, a - b.y); }
template <typename T> inline pk2<T>& operator+=(pk2<T>& a, pk2<T> b) { a.x += b.x; a.y += b.y; return a; }
template <typename T> inline pk2<T>& operator-=(pk2<T>& a, pk2<T> b) { a.x -= b.x; a.y -= b.y; return a; }
#else
template <typename T> using pk2 = T __attribute__((ext_vector_type(2)));
#endif
template <typename T> LFSD_DEV pk2<T> mk2(T a, T b) { pk2<T> v; v.x = a; v.y = b; return v; }

// ---- wave-level matrix helpers of the MFMA backward sweep (fp32, four 16-lane trajectories per wavefront) --------------
// mfma4b: v_mfma_f32_16x16x1_4b_f32 -- four independent 16x16 rank-1 updates, block b fed by the lanes of 16-lane group b:
//   D_b[i][j] += a(lane 16b+i) * b(lane 16b+j);  D_b[i][j] lives in register 4b + i%4 of lane 16(i/4) + j.
// tile_transpose: afterwards lane 16b+j holds D_b[i][j] in register i (column j of ITS block: the column-per-lane layout
//   of the rest of the kernel), by 8 v_permlane32_swap + 8 v_permlane16_swap.
// Both must be reached by all 64 lanes.  The emulator build restates them with an exchange buffer.
#if defined(LFSD_EMU)
struct f32x16 {
  float v[16];
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
};
// (exchange buffers of the emulated cross-lane operations: one slot per thread of the workgroup, a wavefront = 64 consecutive
//  threads -- workgroups of several wavefronts, oc_solve_wide_kernel<..., W>, exchange inside their own 64 slots)
constexpr int EMU_MAXT = 256;
inline void mfma4b(float a, float b, f32x16& acc) {
  static float sa[EMU_MAXT], sb[EMU_MAXT];
  const int t = threadIdx.x, w0 = t & ~63, l = t & 63;
  sa[t] = a; sb[t] = b;
  __syncthreads();
  for (int r = 0; r < 16; ++r) {
    const int blk = r / 4, i = 4 * (l >> 4) + r % 4, j = l & 15;
    acc[r] = std::fmaf(sa[w0 + 16 * blk + i], sb[w0 + 16 * blk + j], acc[r]);      // the hardware's k-ordered fmaf chain
  }
  __syncthreads();
}
inline void tile_transpose(f32x16& acc) {
  static float sx[EMU_MAXT][16];
  const int t = threadIdx.x, w0 = t & ~63, l = t & 63, b = l >> 4, j = l & 15;
  for (int r = 0; r < 16; ++r) sx[t][r] = acc[r];
  __syncthreads();
  for (int q = 0; q < 4; ++q)
    for (int r = 0; r < 4; ++r) acc[4 * q + r] = sx[w0 + 16 * q + j][4 * b + r];
  __syncthreads();
}
#else
typedef float f32x16 __attribute__((ext_vector_type(16)));
LFSD_DEV void mfma4b(float a, float b, f32x16& acc) { acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc, 0, 0, 0); }
LFSD_DEV void tile_transpose(f32x16& acc) {
  // exchange the block index (register group 4R..4R+3) with the lane-group index, one bit per stage
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int R0 = 0; R0 < 2; ++R0) {        // bit 1: register groups R0 / R0+2  <->  lane halves
      const float lo = acc[4 * R0 + r], hi = acc[4 * (R0 + 2) + r];      // (scalars first: __builtin_bit_cast applied to a
      const auto v = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lo),      //  vector ELEMENT reads element 0)
                                                      __builtin_bit_cast(unsigned, hi), false, false);
      const unsigned v0 = v[0], v1 = v[1];
      acc[4 * R0 + r] = __builtin_bit_cast(float, v0);
      acc[4 * (R0 + 2) + r] = __builtin_bit_cast(float, v1);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int R1 = 0; R1 < 2; ++R1) {        // bit 0: register groups 2R1 / 2R1+1  <->  odd / even 16-lane rows
      const float lo = acc[8 * R1 + r], hi = acc[8 * R1 + 4 + r];
      const auto v = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, lo),
                                                      __builtin_bit_cast(unsigned, hi), false, false);
      const unsigned v0 = v[0], v1 = v[1];
      acc[8 * R1 + r] = __builtin_bit_cast(float, v0);
      acc[8 * R1 + 4 + r] = __builtin_bit_cast(float, v1);
    }
  }
}
#endif

// lane_get(v, j): the value lane j of the wavefront holds in v, on every lane (j is the same on all of them: v_readlane_b32).  Must be
// reached by all 64 lanes.
#if defined(LFSD_EMU)
template <typename T> inline T lane_get(T v, int j) {
  static T sb[EMU_MAXT];
  sb[threadIdx.x] = v;
  __syncthreads();
  const T r = sb[(threadIdx.x & ~63) + j];
  __syncthreads();
  return r;
}
#else
LFSD_DEV float lane_get(float v, int j) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); }
LFSD_DEV double lane_get(double v, int j) { return __shfl(v, j); }
#endif

// quarter_sum(v): sum of v over the four lanes l, l^16, l^32, l^48 of the wavefront, in every one of them (the wide kernel's
// backward sweep splits the rows of its dense products over the four 16-lane quarters).  Must be reached by all 64 lanes.
#if defined(LFSD_EMU)
template <typename T> inline T quarter_sum(T v) {
  static T sq[EMU_MAXT];
  const int t = threadIdx.x, w0 = t & ~63, l = t & 15;
  sq[t] = v;
  __syncthreads();
  const T r = (sq[w0 + l] + sq[w0 + l + 16]) + (sq[w0 + l + 32] + sq[w0 + l + 48]);
  __syncthreads();
  return r;
}
#else
LFSD_DEV float quarter_sum(float v) {
  // v_permlane32_swap(a, b): a' = [a.lo32, b.lo32], b' = [a.hi32, b.hi32]; with a = b = v their sum is v.lo + v.hi everywhere
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned p0 = p[0], p1 = p[1];
  const float w = __builtin_bit_cast(float, p0) + __builtin_bit_cast(float, p1);
  const unsigned uw = __builtin_bit_cast(unsigned, w);
  const auto q = __builtin_amdgcn_permlane16_swap(uw, uw, false, false);
  const unsigned q0 = q[0], q1 = q[1];
  return __builtin_bit_cast(float, q0) + __builtin_bit_cast(float, q1);
}
LFSD_DEV double quarter_sum(double v) {
  const double w = v + __shfl_xor(v, 32);
  return w + __shfl_xor(w, 16);
}
#endif

// LFSD_ISSUE_FENCE(): the loads written above it are ISSUED above it.  Without it the compiler sinks a software prefetch
// ("load stage k-1 while stage k is processed") down to the first use of its values -- the copy at the end of the loop
// body -- i.e. it undoes the prefetch, and the next stage starts by waiting for its own operands (seen in the ISA of the
// backward sweeps: s_waitcnt vmcnt(15) ... vmcnt(3) between the first thirteen MFMAs of a stage; round 2 measured
// "prefetch on / off: no difference" for exactly that reason).  A compiler-level memory clobber: no instruction, and no
// wait either -- the s_waitcnt still goes where the values are first used.
#if defined(LFSD_EMU)
#define LFSD_ISSUE_FENCE()
#else
#define LFSD_ISSUE_FENCE() asm volatile("" ::: "memory")
#endif

// sched_load / sched_add: the counter of finished trajectories of a two-launch wide solve (OcArgs::sched): relaxed device-scope
// operations performed in the L2 (a plain load could be served by the scalar cache or the vector L1 for ever).  The emulator runs the
// workgroups of a launch one after the other.
#if defined(LFSD_EMU)
inline int sched_load(const int* p) { return *p; }
inline int sched_load_uniform(const int* p) { return *p; }
inline int sched_add(int* p, int v) { const int o = *p; *p += v; return o; }
#else
LFSD_DEV int sched_load(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ... by all lanes of a wavefront at once, the same value on every lane (no LDS hand-over, no barrier)
LFSD_DEV int sched_load_uniform(int* p) { return __builtin_amdgcn_readfirstlane(sched_load(p)); }
LFSD_DEV int sched_add(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#endif

// pin(x): the value is materialised HERE, unconditionally.  Needed where a load feeds one arm of a per-lane select
//   r = own_column ? f(lds[addr]) : other;
// -- clang turns that into a divergent branch around the load, and a sequence of thirteen of them into thirteen serialised
// LDS round trips (measured in the backward sweep of the OC solve, profiles/r03_g_ab_prefetch_pin.txt: 1600 of 8300 cycles
// per stage for the symmetrisation of V_xx alone).  With the loads issued back to back and pinned, the select is a v_cndmask.
// (Tried and dropped in the forward auxiliary sweep, which runs at 256 + 256 registers with spills: batching the loads of
// its stiff steps this way doubled its scratch traffic, 1.01 -> 2.13 ms; batching only its node staging cost 3 %.)
#if defined(LFSD_EMU)
template <typename T> LFSD_DEV void pin(T&) {}
#else
template <typename T> LFSD_DEV void pin(T& x) { asm volatile("" : "+v"(x)); }
#endif

// lds_fetch<N>(src, dst): N consecutive LDS words into registers, ALL reads issued back to back, the values materialised
// here.  Where a phase applies staged coefficients through the generated sparse operators the compiler, left to itself,
// emits read - wait - four FMAs - read - wait ...: one exposed LDS round trip per 16 bytes (185 of them per split unit of
// the forward auxiliary sweep, ~100 cycles each on a wavefront that is alone on its SIMD: two thirds of the kernel's time).
// Fetched first, the operators run on registers and the phase waits once.
// lds_issue / lds_land: the two halves, for several arrays fetched together (issue all, then land all).
template <int N, typename T> LFSD_DEV void lds_issue(const T* src, T* dst) {
#pragma unroll
  for (int i = 0; i < N; ++i) dst[i] = src[i];
}
template <int N, typename T> LFSD_DEV void lds_issue_strided(const T* src, int stride, T* dst) {      // N words `stride` apart
#pragma unroll
  for (int i = 0; i < N; ++i) dst[i] = src[i * stride];
}
template <int N, typename T> LFSD_DEV void lds_land(T* dst) {
#pragma unroll
  for (int i = 0; i < N; ++i) pin(dst[i]);
}
template <int N, typename T> LFSD_DEV void lds_fetch(const T* src, T* dst) { lds_issue<N>(src, dst); lds_land<N>(dst); }

// pin64(x): pin() in the fp64 instantiations only.  There a dense product whose result is first used beyond a branch had
// its FMAs sunk to that use by the compiler while its LDS reads stayed put (they cannot cross the barrier in between): all
// 221 operands of Q = [A B]^T Y of the backward sweep were read, spilled to scratch and reloaded one by one -- 157 scratch
// round trips per stage that miss the L2 (116 KB of scratch per wavefront), 95 000 cycles of a stage that computes for
// about 5 000 (tools/oc_clock64.py, profiles/r03_o_fp64_backward.txt).  A pinned result keeps the FMAs with their reads.
template <typename T> LFSD_DEV void pin64(T& x) { if constexpr (sizeof(T) == 8 && ((LFSD_FENCE64) & 1) != 0) pin(x); }

template <typename T> struct Eps;
template <> struct Eps<float> { static LFSD_DEV float v() { return 1.1920929e-07f; } };
template <> struct Eps<double> { static LFSD_DEV double v() { return 2.220446049250313e-16; } };

template <typename T> LFSD_HD constexpr int aux_rk() { return sizeof(T) == 4 ? LFSD_AUX_RK32 : LFSD_AUX_RK64; }

template <typename T> LFSD_DEV T t_abs(T a) { return a < T(0) ? -a : a; }
template <typename T> LFSD_DEV T t_max(T a, T b) { return a > b ? a : b; }
template <typename T> LFSD_DEV T t_min(T a, T b) { return a < b ? a : b; }
template <typename T> LFSD_DEV bool t_finite(T a) { return (a - a) == T(0); }
LFSD_DEV float t_sqrt(float a) { return sqrtf(a); }
LFSD_DEV double t_sqrt(double a) { return sqrt(a); }
// Reciprocal and reciprocal square root of a PIVOT (positive, well inside the normal range): the hardware estimate
// (v_rcp_f32 / v_rsq_f32, 1 ulp) refined by one Newton step -- three to four instructions where the IEEE division and
// square root of the compiler are ten-instruction dependent sequences each.  fp64 and the CPU emulator divide.
#if defined(LFSD_EMU)
LFSD_DEV float t_rcp(float a) { return 1.0f / a; }
LFSD_DEV float t_rsqrt(float a) { return 1.0f / sqrtf(a); }
#else
LFSD_DEV float t_rcp(float a) { float r = __builtin_amdgcn_rcpf(a); return fmaf(fmaf(-a, r, 1.0f), r, r); }
LFSD_DEV float t_rsqrt(float a) { float r = __builtin_amdgcn_rsqf(a); return r * fmaf(-0.5f * a * r, r, 1.5f); }
#endif
LFSD_DEV double t_rcp(double a) { return 1.0 / a; }
LFSD_DEV double t_rsqrt(double a) { return 1.0 / sqrt(a); }
// sin / cos of the generated model code (codegen.py prints lfsd::t_sin / lfsd::t_cos).  The device library's sinf / cosf are
// ~140-instruction routines each (150 for both through sincosf; tools/probes/trig_probe.hip), and the robot arm's dynamics
// call four of them per evaluation -- 800 evaluations per iteration of its roll-outs, the largest phase of its solve
// (profiles/r04_g_arm_wide_clock.txt).  fp32 on the GPU: one Cody-Waite reduction by multiples of pi/2 (three constants, exact
// products for |x| <= 100) + the two degree-7 / degree-8 minimax polynomials of cephes' sinf / cosf on [-pi/4, pi/4]; t_sin(a)
// and t_cos(a) of the same argument share the reduction after inlining.  Measured against fp64 on 2^24 arguments in
// [-100, 100] (trig_probe): see profiles/r04_l_trig_probe.txt.  Beyond |x| = 100, in fp64 and in the CPU emulator: the library.
// (LFSD_FAST_TRIG: shipped as on since the round it was measured in; the switch was removed in round 6, the alternative is in the history)
LFSD_DEV double t_sin(double a) { return sin(a); }
LFSD_DEV double t_cos(double a) { return cos(a); }
#if defined(LFSD_EMU)
LFSD_DEV float t_sin(float a) { return sinf(a); }
LFSD_DEV float t_cos(float a) { return cosf(a); }
#else
LFSD_DEV void sincos_pio4(float x, float& sr, float& cr, int& quad) {
  const float q = __builtin_rintf(x * 0.63661977236758134f);          // nearest multiple of pi/2
  float r = fmaf(q, -1.57073974609375f, x);                             // pi/2 = A + B + C, A and B with short mantissas
  r = fmaf(q, -5.657970905303955078125e-05f, r);
  r = fmaf(q, -9.920936294705029468e-10f, r);
  const float z = r * r;
  sr = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  cr = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.0f));
  quad = (int)q;
}
LFSD_DEV float t_sin(float x) {
  float sr, cr; int n;
  sincos_pio4(x, sr, cr, n);
  float v = (n & 1) ? cr : sr;
  v = (n & 2) ? -v : v;
  if (!(__builtin_fabsf(x) <= 100.0f)) v = sinf(x);
  return v;
}
LFSD_DEV float t_cos(float x) {
  float sr, cr; int n;
  sincos_pio4(x, sr, cr, n);
  float v = (n & 1) ? sr : cr;
  v = ((n + 1) & 2) ? -v : v;
  if (!(__builtin_fabsf(x) <= 100.0f)) v = cosf(x);
  return v;
}
#endif
LFSD_DEV float t_floor(float a) { return floorf(a); }
LFSD_DEV double t_floor(double a) { return floor(a); }
LFSD_DEV float t_pow(float a, float b) { return powf(a, b); }
LFSD_DEV double t_pow(double a, double b) { return pow(a, b); }

// ---- tiny dense helpers on group-uniform n x n matrices (row-major, in registers) -----------
// Cholesky A = L L^T in place (lower); false if not positive definite.  The DIAGONAL of the result holds 1 / L_jj: the
// substitutions of chol_solve then multiply instead of divide (an IEEE fp32 division is a ten-instruction dependent
// sequence on this hardware, and the backward sweep of the OC solve runs two solves per stage on every lane).
template <int n, typename T> LFSD_DEV bool chol_factor(T* A, T& dmin) {
  bool ok = true;
#pragma unroll
  for (int j = 0; j < n; ++j) {
    T d = A[j * n + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > T(0))) { if (ok && d < dmin) dmin = d; ok = false; d = T(1); }    // only the first failing pivot is meaningful
    const T inv = t_rsqrt(d);
    A[j * n + j] = inv;
#pragma unroll
    for (int i = j + 1; i < n; ++i) {
      T s = A[i * n + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s * inv;
    }
  }
  return ok;
}
template <int n, typename T> LFSD_DEV void chol_solve(const T* Lm, T* b) {
#pragma unroll
  for (int i = 0; i < n; ++i) {
    T s = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= Lm[i * n + k] * b[k];
    b[i] = s * Lm[i * n + i];
  }
#pragma unroll
  for (int i = n - 1; i >= 0; --i) {
    T s = b[i];
#pragma unroll
    for (int k = i + 1; k < n; ++k) s -= Lm[k * n + i] * b[k];
    b[i] = s * Lm[i * n + i];
  }
}
// Box-constrained stage problem of the control-limited backward sweep (finite control_lb / control_ub of
// COCSys.setControlVariable, CPDP.py:33-46, which the reference hands to IPOPT as lbw / ubw):
//     min_x  1/2 x^T Q x + q^T x   s.t.  lo <= x <= hi            (Q = Q_uu + mu I positive definite, n <= 4 controls)
// by a primal active-set iteration on the masked system (clamped rows / columns replaced by identity): solve, clamp the
// components that left the box, re-solve; when nothing moves, release the clamped component whose multiplier has the
// wrong sign most.  Returns the Cholesky factor of the final masked matrix in Lf (the feedback gains of the free
// components are solved with it; clamped components get zero gain) and the clamp mask.
template <int n, typename T> LFSD_DEV bool box_qp(const T* Q, const T* q, const T* lo, const T* hi, T* x, unsigned& mask, T* Lf, T& dmin) {
  mask = 0u;
  T xc[n];
#pragma unroll
  for (int i = 0; i < n; ++i) xc[i] = T(0);
  bool ok = true;
  for (int round = 0; round < 3 * n + 2; ++round) {
    T b[n];
#pragma unroll
    for (int i = 0; i < n; ++i) {
      const bool ci = (mask >> i) & 1u;
      T s = ci ? xc[i] : -q[i];
#pragma unroll
      for (int j = 0; j < n; ++j) {
        const bool cj = (mask >> j) & 1u;
        Lf[i * n + j] = (ci || cj) ? ((i == j) ? T(1) : T(0)) : Q[i * n + j];
        if (!ci && cj) s -= Q[i * n + j] * xc[j];
      }
      b[i] = s;
    }
    T dd = T(0);
    if (!chol_factor<n>(Lf, dd)) { ok = false; if (dd < dmin) dmin = dd; break; }
    chol_solve<n>(Lf, b);
    unsigned newmask = mask;
#pragma unroll
    for (int i = 0; i < n; ++i) {
      if ((mask >> i) & 1u) continue;
      if (b[i] < lo[i]) { newmask |= 1u << i; xc[i] = lo[i]; }
      else if (b[i] > hi[i]) { newmask |= 1u << i; xc[i] = hi[i]; }
    }
#pragma unroll
    for (int i = 0; i < n; ++i) x[i] = ((newmask >> i) & 1u) ? xc[i] : b[i];
    if (newmask != mask) { mask = newmask; continue; }
    // stationary for this active set: multipliers g_i = (Q x + q)_i of the clamped components must push outward
    int worst = -1;
    T wv = T(0);
#pragma unroll
    for (int i = 0; i < n; ++i) {
      if (!((mask >> i) & 1u)) continue;
      T g = q[i];
#pragma unroll
      for (int j = 0; j < n; ++j) g += Q[i * n + j] * x[j];
      const T viol = (xc[i] <= lo[i]) ? -g : g;         // at the lower bound the gradient must be >= 0, at the upper <= 0
      if (viol > wv) { wv = viol; worst = i; }
    }
    if (worst < 0) break;
    mask &= ~(1u << worst);
  }
  return ok;
}
// LU without pivoting, in place (unit lower + upper).  For I + small and SPD-like matrices.  As in chol_factor the
// diagonal of the result holds the RECIPROCAL pivots, and lu_solve multiplies by them.
template <int n, typename T> LFSD_DEV void lu_factor(T* A) {
#pragma unroll
  for (int j = 0; j < n; ++j) {
    const T inv = t_rcp(A[j * n + j]);
    A[j * n + j] = inv;
#pragma unroll
    for (int i = j + 1; i < n; ++i) {
      const T f = A[i * n + j] * inv;
      A[i * n + j] = f;
#pragma unroll
      for (int k = j + 1; k < n; ++k) A[i * n + k] -= f * A[j * n + k];
    }
  }
}
template <int n, typename T> LFSD_DEV void lu_solve(const T* A, T* b) {
#pragma unroll
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < i; ++k) b[i] -= A[i * n + k] * b[k];
  }
#pragma unroll
  for (int i = n - 1; i >= 0; --i) {
#pragma unroll
    for (int k = i + 1; k < n; ++k) b[i] -= A[i * n + k] * b[k];
    b[i] *= A[i * n + i];
  }
}
template <int n, typename T> LFSD_DEV void mat_inverse(const T* A, T* Ainv) {
  T F[n * n];
#pragma unroll
  for (int i = 0; i < n * n; ++i) F[i] = A[i];
  lu_factor<n>(F);
#pragma unroll
  for (int c = 0; c < n; ++c) {
    T b[n];
#pragma unroll
    for (int i = 0; i < n; ++i) b[i] = (i == c) ? T(1) : T(0);
    lu_solve<n>(F, b);
#pragma unroll
    for (int i = 0; i < n; ++i) Ainv[i * n + c] = b[i];
  }
}
template <int n, typename T> LFSD_DEV void matmul(const T* A, const T* B, T* C) {
#pragma unroll
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < n; ++j) {
      T s = T(0);
#pragma unroll
      for (int k = 0; k < n; ++k) s += A[i * n + k] * B[k * n + j];
      C[i * n + j] = s;
    }
  }
}
// P = phi1(M) = M^-1 (I - exp(-M)) for a small matrix with non-negative spectrum
// (scaling and squaring: Taylor of degree 8 on M/2^s, then phi1(2A) = (I + e^-A) phi1(A) / 2).
template <typename T> struct PhiDeg;
template <> struct PhiDeg<float> { static constexpr int v = 6; };
template <> struct PhiDeg<double> { static constexpr int v = 9; };
// P = phi1(M), P2 = phi1(2M)
template <int n, typename T> LFSD_DEV void phi1_neg(const T* M, T* P, T* P2) {
  T nrm = T(0);
#pragma unroll
  for (int i = 0; i < n; ++i) {
    T r = T(0);
#pragma unroll
    for (int j = 0; j < n; ++j) r += t_abs(M[i * n + j]);
    nrm = t_max(nrm, r);
  }
  int sq = 0;
  T sc = T(1);
  while (nrm * sc > T(0.25) && sq < 60) { sc *= T(0.5); ++sq; }
  T A[n * n], E[n * n], W[n * n];
#pragma unroll
  for (int i = 0; i < n * n; ++i) { A[i] = M[i] * sc; P[i] = T(0); }
  const T ck[10] = {T(1), T(1) / T(2), T(1) / T(6), T(1) / T(24), T(1) / T(120), T(1) / T(720), T(1) / T(5040),
                    T(1) / T(40320), T(1) / T(362880), T(1) / T(3628800)};
  constexpr int DEG = PhiDeg<T>::v;       // |A| <= 1/4: truncation 4^-(DEG+1)/(DEG+2)! below round-off
#pragma unroll
  for (int i = 0; i < n; ++i) P[i * n + i] = ck[DEG];
#pragma unroll
  for (int k = DEG - 1; k >= 0; --k) {
    matmul<n>(A, P, W);
#pragma unroll
    for (int i = 0; i < n * n; ++i) P[i] = -W[i];
#pragma unroll
    for (int i = 0; i < n; ++i) P[i * n + i] += ck[k];
  }
  matmul<n>(A, P, W);
#pragma unroll
  for (int i = 0; i < n * n; ++i) E[i] = -W[i];
#pragma unroll
  for (int i = 0; i < n; ++i) E[i * n + i] += T(1);
  for (int it = 0; it < sq; ++it) {
    matmul<n>(E, P, W);
#pragma unroll
    for (int i = 0; i < n * n; ++i) P[i] = T(0.5) * (P[i] + W[i]);
    matmul<n>(E, E, W);
#pragma unroll
    for (int i = 0; i < n * n; ++i) E[i] = W[i];
  }
  matmul<n>(E, P, W);
#pragma unroll
  for (int i = 0; i < n * n; ++i) P2[i] = T(0.5) * (P[i] + W[i]);
}
// ---- a small matrix spread ROW PER LANE over a sub-group of Q = 1 / 2 / 4 adjacent lanes (Q | 4, sub-group aligned) ------
// sub_bcast<Q>(v, k): v of lane k of the caller's sub-group, in every lane of it.  On the GPU a DPP move (quad_perm), i.e. a
// register-to-register instruction without LDS; k must be a constant after unrolling.  Every lane that could be a SOURCE
// has to be active: the callers run whole lane groups through the same control flow.  The CPU emulator restates it with an
// exchange buffer (all lanes of the sub-group must call it together).
template <int Q> constexpr bool sub_ok() { return Q == 1 || Q == 2 || Q == 4; }
#if defined(LFSD_EMU)
template <int Q, typename T> inline T sub_bcast(T v, int k) {
  static_assert(sub_ok<Q>(), "sub-group of 1, 2 or 4 lanes");
  if (Q == 1) return v;
  static T sb[EMU_MAXT];
  const int l = threadIdx.x;
  sb[l] = v;
  __syncthreads();
  const T r = sb[(l & ~(Q - 1)) | k];
  __syncthreads();
  return r;
}
#else
template <int CTRL> LFSD_DEV float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> LFSD_DEV double dpp_mov(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)u, CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
template <int Q, typename T> LFSD_DEV T sub_bcast(T v, int k) {
  static_assert(sub_ok<Q>(), "sub-group of 1, 2 or 4 lanes");
  if constexpr (Q == 1) return v;
  else if constexpr (Q == 2) return k == 0 ? dpp_mov<0xA0>(v) : dpp_mov<0xF5>(v);            // quad_perm [0,0,2,2] / [1,1,3,3]
  else return k == 0 ? dpp_mov<0x00>(v) : (k == 1 ? dpp_mov<0x55>(v) : (k == 2 ? dpp_mov<0xAA>(v) : dpp_mov<0xFF>(v)));
}
#endif
// smallest sub-group that holds one row per lane of an n x n matrix (n <= 4)
template <int n> constexpr int sub_lanes() { return n <= 1 ? 1 : (n == 2 ? 2 : 4); }
// W = A B, row a of each matrix in lane a of the sub-group (lanes a >= n carry zeros); same summation order as matmul<n>
template <int n, int Q, typename T> LFSD_DEV void matmul_rows(const T* Arow, const T* Brow, T* Wrow) {
#pragma unroll
  for (int j = 0; j < n; ++j) Wrow[j] = T(0);
#pragma unroll
  for (int k = 0; k < n; ++k) {
#pragma unroll
    for (int j = 0; j < n; ++j) Wrow[j] += Arow[k] * sub_bcast<Q>(Brow[j], k);
  }
}
// phi1_neg with the matrix spread row per lane: lane a of the sub-group holds row a of M and receives row a of
// P = phi1(M) and P2 = phi1(2M).  The scaling (sc = 2^-sq with |M| sc <= 1/4) comes from the caller, who makes it uniform
// over everything that runs through here together (over-scaling is harmless: more squarings of a smaller argument).
template <int n, int Q, typename T> LFSD_DEV void phi1_neg_rows(const T* Mrow, int a, int sq, T sc, T* P, T* P2) {
  T A[n], E[n], W[n];
  const T ck[10] = {T(1), T(1) / T(2), T(1) / T(6), T(1) / T(24), T(1) / T(120), T(1) / T(720), T(1) / T(5040),
                    T(1) / T(40320), T(1) / T(362880), T(1) / T(3628800)};
  constexpr int DEG = PhiDeg<T>::v;
#pragma unroll
  for (int j = 0; j < n; ++j) { A[j] = Mrow[j] * sc; P[j] = (j == a) ? ck[DEG] : T(0); }
#pragma unroll
  for (int k = DEG - 1; k >= 0; --k) {
    matmul_rows<n, Q>(A, P, W);
#pragma unroll
    for (int j = 0; j < n; ++j) P[j] = ((j == a) ? ck[k] : T(0)) - W[j];
  }
  matmul_rows<n, Q>(A, P, W);
#pragma unroll
  for (int j = 0; j < n; ++j) E[j] = ((j == a) ? T(1) : T(0)) - W[j];
  for (int it = 0; it < sq; ++it) {
    matmul_rows<n, Q>(E, P, W);
#pragma unroll
    for (int j = 0; j < n; ++j) P[j] = T(0.5) * (P[j] + W[j]);
    matmul_rows<n, Q>(E, E, W);
#pragma unroll
    for (int j = 0; j < n; ++j) E[j] = W[j];
  }
  matmul_rows<n, Q>(E, P, W);
#pragma unroll
  for (int j = 0; j < n; ++j) P2[j] = T(0.5) * (P[j] + W[j]);
}
template <int n, typename T> LFSD_DEV void matvec(const T* A, const T* v, T* y) {
#pragma unroll
  for (int i = 0; i < n; ++i) {
    T s = T(0);
#pragma unroll
    for (int k = 0; k < n; ++k) s += A[i * n + k] * v[k];
    y[i] = s;
  }
}

}  // namespace lfsd
