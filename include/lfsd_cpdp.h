/* lfsd_cpdp.h — C ABI of the MI355X-native batched Continuous-PDP solver.
 *
 * One shared library is built per optimal-control model (dimensions and the
 * model's derivative code are compiled in); every library exports exactly the
 * symbols below.  All array arguments are DEVICE pointers (hipMalloc'd memory,
 * e.g. torch tensors' data_ptr()), row-major, batch-major; `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  dtype: 0 = float32,
 * 1 = float64 (the arithmetic type of every array in the call).
 * Every function returns 0 on success, a negative LFSD_E* code on bad
 * arguments, or a positive hipError_t if a launch failed.
 *
 * Reference interface each entry point replaces (wanxinjin/Learning-from-
 * Sparse-Demonstrations @ v1):
 *   lfsd_coc_solve        CPDP/CPDP.py:92-198   COCSys.cocSolver (and :486-594 time-varying)
 *   lfsd_aux_solve        CPDP/CPDP.py:301-381  COCSys.auxSysSolver (and :706-786)   [= lfsd_aux_riccati + lfsd_aux_forward]
 *                         + lib/QuadAlgorithm.py:616-673 getloss_pos_corrections / getloss_corrections
 *                           (Examples/*.py getloss_corrections)
 *   lfsd_optimizer_step   lib/QuadAlgorithm.py:454-578 Vanilla/Nesterov/Adam/Nadam/AMSGrad
 *   lfsd_lookahead        lib/QuadAlgorithm.py:478 (Nesterov look-ahead point)
 */
#ifndef LFSD_CPDP_H
#define LFSD_CPDP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LFSD_ABI_VERSION 9
#define LFSD_F32 0
#define LFSD_F64 1
#define LFSD_EINVAL (-1)   /* bad argument (null pointer, non-positive size, unknown enum) */
#define LFSD_ENOSPC (-2)   /* workspace too small */

/* status[] values written by lfsd_coc_solve */
#define LFSD_ST_CONVERGED 1   /* max |dJ/du| < tol * (1 + |J|)                         */
#define LFSD_ST_STALLED   2   /* converged to working precision: no step improves J beyond rounding, or the Newton
                                 decrement is below the resolution of J with the gradient at its rounding floor */
#define LFSD_ST_MAXITER   3
#define LFSD_ST_FAILED    4   /* non-finite cost / regularisation exhausted            */

/* mapping of the optimal-control solve onto the machine (lfsd_coc_solve) */
#define LFSD_MAP_AUTO     0   /* by batch size / model / exact_after */
#define LFSD_MAP_LOCKSTEP 1   /* several trajectories per wavefront, shooting intervals in sequence */
#define LFSD_MAP_WIDE     2   /* one trajectory per wavefront, intervals / step lengths in parallel */

/* optimizer methods (lib/QuadAlgorithm.py:164-188) */
#define LFSD_OPT_VANILLA  0
#define LFSD_OPT_NESTEROV 1
#define LFSD_OPT_ADAM     2
#define LFSD_OPT_NADAM    3
#define LFSD_OPT_AMSGRAD  4

typedef struct lfsd_model_info {
  int abi_version;
  int n_state, n_control, n_auxvar, n_const;   /* n_const may be 0 */
  int time_varying;                            /* 1: COCSys_TimeVarying semantics */
  int lanes_per_trajectory;                    /* lane-group width G the kernels were built with */
  int is_emulator;                             /* 1 only for the CPU SIMT-emulator test build */
  const char* name;
  const char* hash;
} lfsd_model_info;

int lfsd_get_model_info(lfsd_model_info* out);
/* outputs of the interface function y = g(x) compiled into this library (ABI 9), 0 if the model was generated without one.  The
 * reference's interface functions are arbitrary CasADi expressions of the state (lib/QuadAlgorithm.py:616-639, Examples/*.py:
 * `Function('interface', [oc.state], [...])` and its `jacobian`); every example selects state components, which `iface_idx` covers
 * without a recompilation; a model generated with `setInterface(expr)` carries g and (dg/dx)^T r as generated code. */
int lfsd_interface_dim(void);
/* default value of runtime constant i (the number the reference would have baked into the CasADi graph) */
double lfsd_const_default(int i);

/* bytes of device scratch lfsd_coc_solve needs for `batch` trajectories when called with the same exact_after / mapping
 * and with (bounded != 0) or without control bounds: the two mappings of the solve lay their scratch out differently, and an
 * fp64 solve that is seeded by an fp32 one (see lfsd_coc_solve) stages the fp32 problem behind its own scratch */
size_t lfsd_coc_workspace_bytes(int dtype, int batch, int n_grid, int exact_after, int mapping, int bounded);

/* Solve `batch` independent optimal-control problems (the NLP of CPDP.py:110-175:
 * n_grid shooting intervals, steps_per_grid RK4 steps each, piecewise-constant control).
 *   ini_state [B][n_state]   horizon [B]   auxvar [B][n_auxvar]
 *   consts    [B][n_const] (const_per_traj=1) or [n_const] (const_per_traj=0); may be NULL if n_const==0
 *   u_init    [B][n_grid][n_control] initial guess, or NULL for zeros
 * outputs (CPDP.py:186-196):
 *   state_grid [B][n_grid+1][n_state], control_grid [B][n_grid+1][n_control] (last row repeated),
 *   costate_grid [B][n_grid+1][n_state] (== IPOPT lam_g), cost [B], iters [B], status [B]
 * solver: Gauss-Newton steps first, then Newton steps; `exact_after` = iteration from which the exact
 *   Lagrangian Hessian of the RK4 stages (what IPOPT gets from CasADi) is forced: 16 is the default policy,
 *   0 = exact from the first iteration, <0 = never (Gauss-Newton / Hamiltonian model only).
 *   steps_per_grid <= 8.
 *   control_lb / control_ub [n_control] (shared by the batch) or NULL: finite control bounds, the reference's
 *   setControlVariable(control, control_lb, control_ub) -> lbw / ubw of the NLP (CPDP.py:33-46, 150-153).  Both or
 *   neither; entries beyond +-1e19 mean "unbounded in that direction".  Solved by a control-limited backward sweep
 *   (box QP per stage, zero feedback gain on clamped components, clamped roll-out); the initial guess is the midpoint of
 *   finite bounds as in the reference.
 *   state_lb / state_ub [n_state] (shared by the batch), state_mult [B][n_grid][2][n_state], state_rho > 0, or all NULL:
 *   finite STATE bounds, the reference's setStateVariable(state, state_lb, state_ub) -> lbw / ubw of the shooting nodes
 *   X_1..X_N (CPDP.py:20-31, 140-147).  One call solves ONE augmented-Lagrangian subproblem: the nodes' terms
 *   [max(0, lu + rho (x - ub))^2 - lu^2 + max(0, ll + rho (lb - x))^2 - ll^2] / (2 rho) with the multipliers lu = state_mult
 *   [b][k-1][0][:], ll = [b][k-1][1][:] of node k are added to the cost; the caller updates the multipliers
 *   (lu <- max(0, lu + rho (x_k - ub)), ll likewise) and the penalty between calls until the nodes are feasible -- the outer
 *   loop is host code (COCSys.cocSolverBatch).  The returned costates include the bound multipliers, as IPOPT's lam_g
 *   do.  With state bounds the control-bound arrays must be given too (entries of +-1e20 where there is none).
 *   mapping: LFSD_MAP_AUTO, or force one of the two mappings of the same algorithm (same KKT points either way).
 *   dtype LFSD_F64, lock-step mapping, 32-lane models (quadrotor class), no bounds, exact_after != 0: the problem is solved in
 *   fp32 first -- from u_init as well when one is given: an all-zero row of u_init is a cold start, so a caller that hands zeros
 *   for every row it does not continue gets every row seeded -- and the fp64 kernel starts from those controls (a trajectory
 *   the fp32 solve failed on starts from its own row of u_init, or cold).  Every output and every convergence test is the fp64
 *   kernel's; iters[] counts both solves (so it may exceed max_iter, which bounds each of them); lfsd_coc_workspace_bytes
 *   includes the staging area.  LFSD_F64_SEED=0 in the environment switches the seeding off; a workspace without room for the
 *   staging area (sized while the switch was off) makes the call solve unseeded rather than fail.
 *   Wide mapping, dtype LFSD_F32, no bounds, exact_after >= 0, models whose interval-parallel phases take several rounds of one
 *   wavefront (more than 8 columns of [A B]: quadrotor, rocket): a trajectory may get a workgroup of FOUR wavefronts -- from the start
 *   when batch <= the number of CUs, else in a SECOND launch on the same stream that takes over the trajectories still running once
 *   all but one-per-CU are finished (their solver state is parked in the workspace; a device counter decides, the host reads nothing
 *   in between; lfsd_coc_workspace_bytes includes the counters and the hand-over list).  The reference solves every seed on its own
 *   (Examples/robotarm_random.py:60-73): a trajectory's outputs do not depend on the scheme or on the moment of the hand-over, bit
 *   for bit.  Environment (test hooks): LFSD_WIDE_WAVES=1 never more than one wavefront per trajectory, =4 four from the start at any
 *   batch; LFSD_WIDE_CAPACITY=<n> in place of the CU count; LFSD_WIDE_SUSPEND_IT=<k> hand over at iteration k.   */
int lfsd_coc_solve(int dtype, int batch, int n_grid, int steps_per_grid,
                   const void* ini_state, const void* horizon, const void* auxvar,
                   const void* consts, int const_per_traj, const void* u_init,
                   const void* control_lb, const void* control_ub,
                   const void* state_lb, const void* state_ub, const void* state_mult, double state_rho,
                   void* state_grid, void* control_grid, void* costate_grid,
                   void* cost, int* iters, int* status,
                   int max_iter, double tol, int exact_after, int mapping,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Differentiate the maximum principle along the solved trajectories and evaluate the
 * sparse-demonstration loss and its gradient.
 *   Z_grid  [B][n_grid+1][n_state+n_auxvar][n_state]  out: Riccati pair [P W], column-major
 *           (P_k, W_k of CPDP.py:329-338)
 *   iface_idx [n_iface] int32: state components exposed by the interface function; NULL (ABI 9): the interface function
 *           compiled into the library, n_iface = lfsd_interface_dim() (waypoints [B][n_waypoints][n_iface] as before)
 *   taus [B][n_waypoints], waypoints [B][n_waypoints][n_iface]
 *   loss [B], grad [B][n_auxvar]:  loss = sum_k |y(tau_k)-wp_k|^2, grad = sum_k (y-wp)^T dy/dx dx/dtheta
 *           (no factor 2, exactly as lib/QuadAlgorithm.py:630-637)
 *   auxX_grid [B][n_grid+1][n_auxvar][n_state], auxU_grid [B][n_grid+1][n_auxvar][n_control]:
 *           optional (NULL to skip) grids of dx/dtheta and du/dtheta (CPDP.py:352-381), column-major
 *   substeps: minimum coarse split-steps ("units") per grid interval (a 2x finer sweep is run alongside and
 *           Richardson-extrapolated; stiff intervals are refined further); 0 selects the default (1 with rtol > 0, else 4).
 *   rtol:   > 0: error-controlled sub-stepping -- an interval is redone with twice the units while the Richardson estimate
 *           |fine - coarse| / 3 of a block of columns exceeds rtol x that block's magnitude (the reference integrates the
 *           same ODEs with scipy's solve_ivp at its default rtol 1e-3, CPDP.py:335, 368, which is the host side's default here too:
 *           measured gradient error 1e-5..1e-4 of the exact ODE solution against the reference integrator's 2.6e-3).
 *           0: fixed `substeps`.  With oc_status given, a row whose solve did NOT end converged / at working precision (its grids
 *           are not a KKT point) stops refining once it has spent 24 x n_grid x substeps split units in a sweep (64 x until round 5); reported in `stats`.
 *   stats   [B][4] int32 or NULL: per trajectory, {split units executed by the Riccati sweep (rejected attempts included),
 *           intervals of it that were accepted ABOVE rtol because refinement stopped gaining (next to a conjugate point) or hit
 *           its cap, the same two numbers of the forward sweep}.  A non-zero second or fourth entry marks a loss / gradient
 *           whose error estimate exceeds the tolerance asked for.
 *   oc_status [B] int32 or NULL, skip_status_mask (ABI 8): the status[] lfsd_coc_solve wrote for these trajectories and a
 *           bit mask over its values (bit s set = skip rows with status s, e.g. 1 << LFSD_ST_FAILED).  A skipped row costs
 *           nothing: neither sweep runs for it, its loss, gradient and its Z_grid / auxX_grid / auxU_grid rows are NaN
 *           (never what the caller's buffers held before), its stats 0.  Without it a solve that FAILED (non-finite grids) or ran out of iterations on a problem
 *           without a minimiser still goes through the error-controlled sweeps, refines to the cap and holds its launch
 *           many times longer than the well-posed batch needs.  NULL (or mask 0): every row is differentiated, as the
 *           reference does.                                                                   */
int lfsd_aux_solve(int dtype, int batch, int n_grid,
                   const void* horizon, const void* auxvar, const void* consts, int const_per_traj,
                   const void* state_grid, const void* control_grid, const void* costate_grid,
                   void* Z_grid,
                   int n_waypoints, int n_iface, const int* iface_idx,
                   const void* taus, const void* waypoints,
                   void* loss, void* grad, void* auxX_grid, void* auxU_grid,
                   int substeps, double rtol, int* stats,
                   const int* oc_status, int skip_status_mask, void* stream);

/* The two phases of lfsd_aux_solve as separate launches (same arguments; lfsd_aux_solve == riccati then forward):
 *   lfsd_aux_riccati  CPDP/CPDP.py:316-338  backward Riccati sweep, fills Z_grid
 *   lfsd_aux_forward  CPDP/CPDP.py:340-381  forward sensitivity sweep from Z_grid + loss/gradient          */
int lfsd_aux_riccati(int dtype, int batch, int n_grid,
                     const void* horizon, const void* auxvar, const void* consts, int const_per_traj,
                     const void* state_grid, const void* control_grid, const void* costate_grid,
                     void* Z_grid, int substeps, double rtol, int* stats,
                     const int* oc_status, int skip_status_mask, void* stream);
int lfsd_aux_forward(int dtype, int batch, int n_grid,
                     const void* horizon, const void* auxvar, const void* consts, int const_per_traj,
                     const void* state_grid, const void* control_grid, const void* costate_grid,
                     const void* Z_grid,
                     int n_waypoints, int n_iface, const int* iface_idx,
                     const void* taus, const void* waypoints,
                     void* loss, void* grad, void* auxX_grid, void* auxU_grid,
                     int substeps, double rtol, int* stats,
                     const int* oc_status, int skip_status_mask, void* stream);

/* theta <- update(theta, grad) for every trajectory; m/v/vhat are optimizer state [B][n_param]
 * (m: Nesterov velocity or first moment; v: second moment; vhat: AMSGrad max; unused ones may be NULL).
 * proj_lo [n_param] or NULL: theta <- max(theta, proj_lo) after the step (the examples' projection
 * current_parameter[0] = fmax(current_parameter[0], 1e-8)).  iter_idx counts from 0.
 * row_active [B] int32 or NULL: rows with 0 are frozen for this step -- theta and m/v/vhat stay untouched (a
 * learner that skips trajectories whose solve did not converge; the reference has no such case, pass NULL).  */
int lfsd_optimizer_step(int dtype, int method, int batch, int n_param, int iter_idx,
                        double lr, double mu, double beta1, double beta2, double eps,
                        void* theta, const void* grad, void* m, void* v, void* vhat,
                        const void* proj_lo, const int* row_active, void* stream);

/* out = theta + mu * v   (Nesterov look-ahead, [B][n_param]) */
int lfsd_lookahead(int dtype, long long n, double mu, const void* theta, const void* v, void* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
