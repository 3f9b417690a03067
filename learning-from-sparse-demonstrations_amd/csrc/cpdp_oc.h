// Optimal-control solve: OcSolver / OcWide and the two oc_solve kernels (COCSys.cocSolver, CPDP.py:92-198).
// Part of the kernel sources collected by cpdp_kernels.h (include that header, not this one).
#pragma once
#include "cpdp_common.h"

namespace lfsd {

// =====================================================================================
//  Optimal-control solve
// =====================================================================================
template <typename T> struct OcArgs {
  int batch, n_grid, steps_per_grid, max_iter;
  const T* ini_state;   // [B][NX]
  const T* horizon;     // [B]
  const T* auxvar;      // [B][NP]
  const T* consts;      // [B or 1][NC]
  int const_stride;     // NC or 0
  const T* u_init;      // [B][N][NU] or nullptr (zeros, the reference's w0 for unbounded controls)
  T* state_grid;        // [B][N+1][NX]
  T* control_grid;      // [B][N+1][NU]  (last row repeats row N-1, CPDP.py:191)
  T* costate_grid;      // [B][N+1][NX]
  T* cost;              // [B]
  int* iters;           // [B]
  int* status;          // [B]
  T* ws;                // per-trajectory scratch, ws_stride elements each
  long long ws_stride;
  T tol;                // stop when max|dJ/du| < tol*(1+|J|)
  int exact_after;      // iteration from which the exact stage Hessian is forced (0: from the start, <0: never)
  int it_start;         // iteration counter to start from (phase 2 of a two-launch solve)
  int max_iter_total;   // overall iteration limit of the solve (phase 1 only hands over if a phase 2 follows)
  int resume;           // 1: continue only trajectories whose status is ST_MAXITER, warm-started from control_grid
  const T* u_lb;        // [NU] finite control bounds (CPDP.py:33-46) or nullptr; handled by the wide kernel
  const T* u_ub;
  // state bounds on the shooting nodes 1..N (CPDP.py:20-31, 140-147: lbw / ubw of the X_k), as the augmented-Lagrangian
  // term of ONE outer iteration (the host updates multipliers and penalty between solves, COCSys.cocSolverBatch):
  //   sum_k sum_i [ max(0, lu + rho (x_i - ub_i))^2 - lu^2 + max(0, ll + rho (lb_i - x_i))^2 - ll^2 ] / (2 rho)
  const T* x_lb;        // [NX] or nullptr (no state bounds); entries beyond +-1e19: unbounded in that direction
  const T* x_ub;
  const T* x_mult;      // [B][N][2][NX] multipliers lu (upper), ll (lower) of node k = 1..N at index k-1
  T x_rho;
  int start_mode;       // lean kernel: stage-Hessian model a trajectory WITH an initial guess starts from (0 Gauss-Newton, 1 Hamiltonian: a guess next to the answer)
  T mu_stage_frac;      // > 0: a stage whose Q_uu factorises with this fraction of the Levenberg shift keeps only that fraction (generic sweep)
  // wide kernel, two-launch solves (lfsd_capi.cpp, coc_solve_t): sched[0] counts the trajectories of the launch that are finished; a
  // trajectory that finds suspend_at or more of them finished at the top of an iteration parks its solver state in the workspace
  // (OcLayout::WIDE_STATE) and leaves with status ST_RUNNING -- the next launch (resume == 2, several wavefronts per trajectory)
  // takes it up exactly there.  sched == nullptr: no suspension.  suspend_it >= 0 (test hook): suspend at that iteration instead.
  int* sched;
  int suspend_at;
  int suspend_it;
};

template <class M> struct OcLayout {
  static constexpr int NX = M::NX, NU = M::NU, NXU = NX + NU;
  // scratch per trajectory (elements)
  static constexpr int SMAX = 8;       // RK4 sub-steps per grid interval supported by the exact-Hessian sweep
  // The linearisation [A B; q] of interval k and its exact stage Hessian are stored ROW-major with the column index
  // fastest ([k][row][column], rows padded to an even length): lane j owns column j, so one store / load instruction of
  // the tangent sweep (writer) and of the backward sweep (reader) touches NXU consecutive words -- one or two cache lines
  // -- where the column-major layout of round 1 touched NXU lines at a 56-byte stride (83x the algorithmic traffic in
  // the round-1 profile).  An even row length keeps the two-column stores of the packed roll-out 8-byte aligned.
  static constexpr int NXUP = (NXU + 1) / 2 * 2;
  static constexpr int M_ELEMS = (NX + 1) * NXUP;                // [A B] rows + the cost-gradient row q, per interval
  static constexpr int H_ELEMS = NXU * NXUP;
  // Lean kernel with structurally constant tangent columns (M::NZC leading state components the dynamics do not depend
  // on, codegen.py): only the LIVE = NXU - ZC columns are propagated and stored -- rows 0..NX of them, then the cost-row
  // entries q of the constant columns -- inside the same scratch region (MS_ELEMS <= M_ELEMS words per interval).
  static constexpr int ZC = M::NZC, LIVE = NXU - ZC, LIVEP = (LIVE + 1) / 2 * 2, ZCP = (ZC + 1) / 2 * 2;
  static constexpr int MS_ELEMS = (NX + 1) * LIVEP + ZCP;
  // (the lean kernel takes this path only where MS_ELEMS <= M_ELEMS, 16-lane groups hold the live columns, and the
  //  constant states find a lane among the control lanes: sc_ok)
  static constexpr bool sc_ok = ZC > 0 && MS_ELEMS <= M_ELEMS && LIVE <= 16 && ZC <= NU && NX <= 16;
  template <int G> LFSD_HD static long long ws_elems(int N) {
    const long long n = 2LL * (N + 1) * NX + 2LL * N * NU + 2LL * N * M_ELEMS + 1LL * N * NX * NU + 1LL * N * NU +
           1LL * (N + 1) * NX + 1LL * SMAX * NX * (1 + G) +     // + sub-step start states (uniform | per lane)
           1LL * N * H_ELEMS;                                    // + exact stage Hessians of the current nominal
    return (n + 1) / 2 * 2;
  }
  // wide mapping (one trajectory per wavefront): the above for 64 lanes + the parked roll-outs of the 16 step lengths
  // + per-lane sub-step start states of the exact-Hessian sweeps
  static constexpr int WIDE_NAL = 16;
  // wide mapping, fp32, models with at most 8 columns of [A B] (robot arm 6, cart-pole 5, pendulum 3): ALL columns of an interval's
  // exact stage Hessian on ONE lane, as NVH packed pairs that share the nominal part of every evaluation (OcSolver::stage_hessian_all)
  static constexpr bool HALL = NXU <= 8;
  static constexpr int NVH = (NXU + 1) / 2;
  // ... + the gaps of a multiple-shooting iterate (own words since round 6: the parked roll-outs of a refused step used to overwrite them)
  // + the per-lane scratch of the exact-Hessian sweeps for the wavefronts 1..WIDE_WMAX-1 of a workgroup that gives one trajectory
  // several wavefronts (oc_solve_wide_kernel<..., W>: the tail of a launch, re-launched) + the parked solver state of a suspended solve
  static constexpr int WIDE_WMAX = 4;
  static constexpr int WIDE_STATE = 64;
  static constexpr long long WIDE_XW = 1LL * SMAX * NX * (1 + 64) + 1LL * SMAX * NX * 64;      // exws-shaped + exwu-shaped region of one extra wavefront
  LFSD_HD static long long ws_elems_wide(int N) {
    const long long n = ws_elems<64>(N) + 1LL * WIDE_NAL * ((N + 1) * NX + N * NU) + 1LL * SMAX * NX * 64 +
                        (HALL ? 2LL * NVH * SMAX * NX * 64 + 2 : 0LL) +      // + per-lane tangent sub-step starts of stage_hessian_all (8-byte aligned)
                        2LL * N * NX + (HALL ? 0LL : (WIDE_WMAX - 1) * WIDE_XW) + WIDE_STATE;
    return (n + 1) / 2 * 2;
  }
  // LDS per group (elements)
  static constexpr int LDS_V = 0;
  static constexpr int LDS_M = LDS_V + NX * NX;
  static constexpr int LDS_K = LDS_M + NXU * NX;
  static constexpr int LDS_QUX = LDS_K + NX * NU;
  static constexpr int LDS_QUU = LDS_QUX + NX * NU;
  static constexpr int LDS_QU = LDS_QUU + NU * NU;
  static constexpr int LDS_VX = LDS_QU + NU;
  static constexpr int LDS_LAM = LDS_VX + NX;
  static constexpr int LDS_RED = LDS_LAM + NX;       // G entries
  // cold per-trajectory state kept in LDS rather than in (spilling) registers
  template <int G> static constexpr int lds_e() { return LDS_RED + G; }
  template <int G> static constexpr int lds_c() { return lds_e<G>() + M::NP; }
  template <int G> static constexpr int lds_x0() { return lds_c<G>() + M::NCX; }
  // exact-Hessian sweep: per-lane slots for the 4 RK4 stage points (uniform copy + this lane's tangent)
  template <int G> static constexpr int lds_ex() { return (lds_x0<G>() + NX + 1) / 2 * 2; }
  // (stage_hessian_all, 64-lane groups: 4 stage points of the nominal + of NVH packed tangent pairs)
  template <int G, int ES = 8> static constexpr int lds_ex_size() { return (G == 64 && ES == 4 && HALL && (4 + 8 * NVH) > 8) ? (4 + 8 * NVH) * NX * G : 8 * NX * G; }
  template <int G, int ES = 8> static constexpr int lds_elems() { return ((lds_ex<G>() + lds_ex_size<G, ES>() + 3) / 4) * 4; }
};

template <class M, typename T, int G, bool EXACT, bool BND = false> struct OcSolver {
  static constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NC = M::NC, NXU = NX + NU;
  T ulb[BND ? NU : 1], uub[BND ? NU : 1];      // BND: box on the controls (clamped roll-out, box-QP backward sweep)
  // BND, state bounds: augmented-Lagrangian node terms (OcArgs::x_lb ...); xm == nullptr: none
  T xlb[BND ? NX : 1], xub[BND ? NX : 1], xrho = T(1);
  const T* xm = nullptr;
  // value of the node term at node k (1..N) for the state x
  LFSD_DEV T node_pen(int k, const T* x) const {
    if (!BND || xm == nullptr) return T(0);
    const T* mk = xm + (long long)(k - 1) * 2 * NX;
    T sacc = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const T lu = mk[i], ll = mk[NX + i];
      const T tu = t_max(lu + xrho * (x[i] - xub[i]), T(0)), tl = t_max(ll + xrho * (xlb[i] - x[i]), T(0));
      sacc += (tu * tu - lu * lu) + (tl * tl - ll * ll);
    }
    return sacc * (T(0.5) / xrho);
  }
  // its gradient and (diagonal, Gauss-Newton) Hessian with respect to component i
  LFSD_DEV void node_pen_d(int k, int i, T xi, T lbi, T ubi, T& g, T& h) const {
    const T* mk = xm + (long long)(k - 1) * 2 * NX;
    const T tu = t_max(mk[i] + xrho * (xi - ubi), T(0)), tl = t_max(mk[NX + i] + xrho * (lbi - xi), T(0));
    g = tu - tl;
    h = xrho * ((tu > T(0) ? T(1) : T(0)) + (tl > T(0) ? T(1) : T(0)));
  }
  static constexpr int NALPHA = (G < 10) ? G : 10;
  using Lay = OcLayout<M>;

  int lane, N, S;
  const T *e, *c, *x0;      // [NP], [NC], [NX] in LDS
  T horizon, dgrid, DT;
  T mu_stage_frac = T(0);      // OcArgs::mu_stage_frac
  T *xb[2], *ub[2], *Mws[2], *Kws, *kws, *lds, *exws, *Hws;
  T *xa = nullptr, *ua = nullptr, *exwu = nullptr;      // wide mapping only: per-step-length roll-outs, per-lane sub-step states
  T *exwm = nullptr;                                     // ... and the per-lane tangent sub-step starts of stage_hessian_all
  // the double-buffered arrays are picked by a select, not by indexing the pointer arrays with a run-time value: that would
  // put the arrays in scratch (it was the lean kernel's last 168 B/lane of scratch)
  LFSD_DEV T* xbp(int i) const { return i ? xb[1] : xb[0]; }
  LFSD_DEV T* ubp(int i) const { return i ? ub[1] : ub[0]; }
  LFSD_DEV T* Mwp(int i) const { return i ? Mws[1] : Mws[0]; }
  T *pkx = nullptr, *pkm = nullptr;      // rk4_step_parked: LDS homes of (x, sum of k) of the group and of this lane's (m, sum of dk)
  T* yz64 = nullptr;                     // backward_sc without matrix cores: [16][ZC] exchange rows (the LDS_M region holds the live columns there)
  bool reuse_hess = false;   // exact stage Hessians in Hws belong to the nominal being swept (a retry with another shift)
  // Multiple-shooting iterate (wide kernel, OcWide::ms_*): the node states of buffer `cur` are variables of their own and
  // gap[k] = F(x_k, u_k) - x_k+1 ([N][NX]) is what the shooting constraints of CPDP.py:166-169 still miss; nullptr: a roll-out (no gaps)
  const T* gap = nullptr;
  T lam_max = T(0);          // backward(): largest |costate| of the sweep (sizes the penalty of the multiple-shooting merit function)
  T* lam_out;   // costate grid of this trajectory (or scratch when invalid)

  LFSD_DEV T tk(int k) const { return M::TIME_VARYING ? dgrid * T(k) : T(0); }

  // One RK4 step of (x, q) with frozen control; optionally with the per-lane tangent (m, mq).
  // V: tangent type -- T (one column per lane) or pk2<T> (two columns per lane, packed math)
  // NZ > 0 (lean kernel with structurally constant columns): qz[i] += RK4 quadrature of dc/dx_i along the stages, i < NZ
  template <bool SENS, class V = T, int NZ = 0>
  LFSD_DEV void rk4_step(T t, T* x, T& q, const T* u, V* m, V& mq, const V* du, T* qz = nullptr) const {
    T xs[NX], ax[NX], f[NX], cq, aq;
    V ms[NX], am[NX], d[NX], dq, adq;
    T cz[NZ ? NZ : 1], az[NZ ? NZ : 1];
    const T hh = DT * T(0.5);
    if (SENS) M::dyn_cost_jvp(t, x, u, e, c, m, du, f, cq, d, dq); else M::dyn_cost(t, x, u, e, c, f, cq);
    aq = cq; if (SENS) adq = dq;
    if (NZ) {
      M::cost_grad_zc(t, x, u, e, c, cz);
#pragma unroll
      for (int i = 0; i < NZ; ++i) az[i] = cz[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) { ax[i] = f[i]; xs[i] = x[i] + hh * f[i]; if (SENS) { am[i] = d[i]; ms[i] = m[i] + hh * d[i]; } }
    LFSD_SCHED_FENCE64(T);
    if (SENS) M::dyn_cost_jvp(t, xs, u, e, c, ms, du, f, cq, d, dq); else M::dyn_cost(t, xs, u, e, c, f, cq);
    aq += T(2) * cq; if (SENS) adq += T(2) * dq;
    if (NZ) {
      M::cost_grad_zc(t, xs, u, e, c, cz);
#pragma unroll
      for (int i = 0; i < NZ; ++i) az[i] += T(2) * cz[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) { ax[i] += T(2) * f[i]; xs[i] = x[i] + hh * f[i]; if (SENS) { am[i] += T(2) * d[i]; ms[i] = m[i] + hh * d[i]; } }
    LFSD_SCHED_FENCE64(T);
    if (SENS) M::dyn_cost_jvp(t, xs, u, e, c, ms, du, f, cq, d, dq); else M::dyn_cost(t, xs, u, e, c, f, cq);
    aq += T(2) * cq; if (SENS) adq += T(2) * dq;
    if (NZ) {
      M::cost_grad_zc(t, xs, u, e, c, cz);
#pragma unroll
      for (int i = 0; i < NZ; ++i) az[i] += T(2) * cz[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) { ax[i] += T(2) * f[i]; xs[i] = x[i] + DT * f[i]; if (SENS) { am[i] += T(2) * d[i]; ms[i] = m[i] + DT * d[i]; } }
    LFSD_SCHED_FENCE64(T);
    if (SENS) M::dyn_cost_jvp(t, xs, u, e, c, ms, du, f, cq, d, dq); else M::dyn_cost(t, xs, u, e, c, f, cq);
    aq += cq; if (SENS) adq += dq;
    const T h6 = DT / T(6);
    if (NZ) {
      M::cost_grad_zc(t, xs, u, e, c, cz);
#pragma unroll
      for (int i = 0; i < NZ; ++i) qz[i] += h6 * (az[i] + cz[i]);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) { x[i] += h6 * (ax[i] + f[i]); if (SENS) m[i] += h6 * (am[i] + d[i]); }
    q += h6 * aq; if (SENS) mq += h6 * adq;
  }

  // The tangent step of the fp64 live-column roll-out (rollout_sens_live).  Same arithmetic as rk4_step<true, T, NZ>, but the
  // values that are only needed BETWEEN the stages -- x_n and the running sum of the k_i (group-uniform: one LDS copy per
  // group, written by its lane 0), this lane's m_n and its running sum -- live in LDS while a stage is evaluated: 52 doubles
  // = 104 registers of a kernel that runs at 256 + 256 and spilled its roll-out loop to scratch (56 reloads per step that
  // miss the L2: 4 wavefronts x 39 KB per CU).  pkm is indexed [i * 64] (one word per lane: conflict-free).
  template <int NZ>
  LFSD_DEV void rk4_step_parked(T t, T* x, T& q, const T* u, T* m, T& mq, const T* du, T* qz) const {
    T xs[NX], f[NX], ms[NX], d[NX], cq, aq, dq, adq;
    T cz[NZ ? NZ : 1], az[NZ ? NZ : 1];
    const T hh = DT * T(0.5);
    const bool l0 = lane == 0;
    M::dyn_cost_jvp(t, x, u, e, c, m, du, f, cq, d, dq);
    aq = cq; adq = dq;
    if (NZ) {
      M::cost_grad_zc(t, x, u, e, c, cz);
#pragma unroll
      for (int i = 0; i < NZ; ++i) az[i] = cz[i];
    }
    LFSD_WAVE_SYNC();                    // (the previous step's readers of pkx are done)
    if (l0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) { pkx[i] = x[i]; pkx[NX + i] = f[i]; }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      pkm[i * 64] = m[i]; pkm[(NX + i) * 64] = d[i];
      xs[i] = x[i] + hh * f[i]; ms[i] = m[i] + hh * d[i];
    }
    LFSD_WAVE_SYNC();
#pragma unroll
    for (int stage = 1; stage < 3; ++stage) {
      const T hs = (stage == 1) ? hh : DT;
      M::dyn_cost_jvp(t, xs, u, e, c, ms, du, f, cq, d, dq);
      aq += T(2) * cq; adq += T(2) * dq;
      if (NZ) {
        M::cost_grad_zc(t, xs, u, e, c, cz);
#pragma unroll
        for (int i = 0; i < NZ; ++i) az[i] += T(2) * cz[i];
      }
      if (l0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) pkx[NX + i] += T(2) * f[i];
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        pkm[(NX + i) * 64] += T(2) * d[i];
        xs[i] = pkx[i] + hs * f[i]; ms[i] = pkm[i * 64] + hs * d[i];
      }
      LFSD_WAVE_SYNC();
    }
    M::dyn_cost_jvp(t, xs, u, e, c, ms, du, f, cq, d, dq);
    aq += cq; adq += dq;
    const T h6 = DT / T(6);
    if (NZ) {
      M::cost_grad_zc(t, xs, u, e, c, cz);
#pragma unroll
      for (int i = 0; i < NZ; ++i) qz[i] += h6 * (az[i] + cz[i]);
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) { x[i] = pkx[i] + h6 * (pkx[NX + i] + f[i]); m[i] = pkm[i * 64] + h6 * (pkm[(NX + i) * 64] + d[i]); }
    q += h6 * aq; mq += h6 * adq;
  }

  // closed-loop control  u = ubar + alpha*kff + K (x - xbar)   (all operands group-uniform loads)
  LFSD_DEV void control(int cur, int k, const T* x, T alpha, bool gains, T* u) const {
    const T* ubk = ubp(cur) + k * NU;
#pragma unroll
    for (int a = 0; a < NU; ++a) u[a] = ubk[a];
    if (gains) {
      const T* xbk = xbp(cur) + k * NX;
      const T* Kk = Kws + k * NX * NU;
      const T* kk = kws + k * NU;
#pragma unroll
      for (int a = 0; a < NU; ++a) u[a] += alpha * kk[a];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        const T dx = x[i] - xbk[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) u[a] += Kk[i * NU + a] * dx;
      }
    }
    if (BND) {
#pragma unroll
      for (int a = 0; a < NU; ++a) u[a] = t_min(t_max(u[a], ulb[a]), uub[a]);
    }
  }

  // Roll the (closed-loop) nominal out into buffer `nxt` and linearise the shooting map along it:
  // lane j < NX+NU propagates column j of d(x_{k+1}, Q_k)/d(x_k, u_k) through the RK4 stages.
  LFSD_DEV T rollout_sens(int cur, int nxt, T alpha, bool gains) {
    T x[NX], u[NU], J = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = x0[i];
    for (int k = 0; k < N; ++k) {
      control(cur, k, x, alpha, gains, u);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xbp(nxt)[k * NX + i] = x[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) ubp(nxt)[k * NU + a] = u[a];
      }
      T m[NX], du[NU], mq = T(0), q = T(0);
#pragma unroll
      for (int i = 0; i < NX; ++i) m[i] = (lane == i) ? T(1) : T(0);
#pragma unroll
      for (int a = 0; a < NU; ++a) du[a] = (lane == NX + a) ? T(1) : T(0);
      const T t = tk(k);
      for (int s = 0; s < S; ++s) rk4_step<true>(t, x, q, u, m, mq, du);
      J += q;
      if (lane < NXU) {
        T* Mk = Mwp(nxt) + (long long)k * Lay::M_ELEMS + lane;
#pragma unroll
        for (int i = 0; i < NX; ++i) Mk[i * Lay::NXUP] = m[i];
        Mk[NX * Lay::NXUP] = mq;
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xbp(nxt)[N * NX + i] = x[i];
    }
    J += M::final_cost(tk(N), x, e, c);
    return J;
  }

  // The same sweep with two columns per lane: lane l < (NX+NU+1)/2 propagates columns 2l and 2l+1 as one packed
  // tangent, so a trajectory needs half the lanes (quadrotor: 9 of a 16-lane group, four trajectories per wavefront).
  LFSD_DEV T rollout_sens_pk(int cur, int nxt, T alpha, bool gains) {
    using V = pk2<T>;
    T x[NX], u[NU], J = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = x0[i];
    const int c0 = 2 * lane, c1 = 2 * lane + 1;
    for (int k = 0; k < N; ++k) {
      control(cur, k, x, alpha, gains, u);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xbp(nxt)[k * NX + i] = x[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) ubp(nxt)[k * NU + a] = u[a];
      }
      V m[NX], du[NU], mq = V(T(0));
      T q = T(0);
#pragma unroll
      for (int i = 0; i < NX; ++i) m[i] = mk2<T>((c0 == i) ? T(1) : T(0), (c1 == i) ? T(1) : T(0));
#pragma unroll
      for (int a = 0; a < NU; ++a) du[a] = mk2<T>((c0 == NX + a) ? T(1) : T(0), (c1 == NX + a) ? T(1) : T(0));
      const T t = tk(k);
      for (int s = 0; s < S; ++s) rk4_step<true, V>(t, x, q, u, m, mq, du);
      J += q;
      if (c0 < NXU) {          // columns c0, c0+1 of every row as one 8-byte store (c1 < NXUP: the pad column of an odd NXU)
        V* Mk = reinterpret_cast<V*>(Mwp(nxt) + (long long)k * Lay::M_ELEMS + c0);
#pragma unroll
        for (int i = 0; i < NX; ++i) Mk[i * (Lay::NXUP / 2)] = m[i];
        Mk[NX * (Lay::NXUP / 2)] = mq;
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xbp(nxt)[N * NX + i] = x[i];
    }
    J += M::final_cost(tk(N), x, e, c);
    return J;
  }

  // The same sweep of the fp64 lean kernel (round 3): ONE column per lane on 16-lane groups, four trajectories per wavefront
  // like the packed fp32 roll-out (fp64 has no packed math; two columns per lane would double the tangent registers of a
  // kernel that already runs at 256 + 256).  Only the LIVE = NXU - ZC columns the dynamics act on are propagated -- lane l
  // carries column ZC + l; the quadrotor's 14 and the rocket's 13 fit the 16 lanes -- and the structurally constant ones
  // (unit vectors through every RK4 stage, see rollout_sens_sc) are written as such, with the quadrature of dc/dx_z along the
  // stages (rk4_step's NZ) as their cost-row entry.  The stored [A B; q] has the layout of rollout_sens: the backward sweep
  // keeps its 32-lane mapping (two passes per wavefront through the mailbox, as for the packed kernel without MFMA).
  // SCL: store [A B; q] in the structural layout of rollout_sens_sc (live columns only + the cost-row entries of the constant
  // states) for the one-pass sweep backward_sc; else in the full layout of rollout_sens for the two-pass generic sweep.
  template <bool SCL = false>
  LFSD_DEV T rollout_sens_live(int cur, int nxt, T alpha, bool gains) {
    constexpr int ZC_ = Lay::ZC, LIVE_ = Lay::LIVE;
    static_assert(LIVE_ <= G || G != 16, "live columns: one per lane of the group");
    T x[NX], u[NU], J = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = x0[i];
    const int col = ZC_ + lane;          // lanes >= LIVE: col >= NXU, a zero tangent that is never stored
#if defined(LFSD_OC_CLOCK) && !defined(LFSD_EMU)      // diagnostic build (tools/oc_clock64.py): clocks of control law | RK4 steps | stores, per roll-out
    long long rck[3] = {0, 0, 0}, rck_t = clock64();
#define LFSD_RCK(i) { const long long t_ = clock64(); rck[i] += t_ - rck_t; rck_t = t_; }
#else
#define LFSD_RCK(i)
#endif
    for (int k = 0; k < N; ++k) {
      control(cur, k, x, alpha, gains, u);
      LFSD_RCK(0)
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xbp(nxt)[k * NX + i] = x[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) ubp(nxt)[k * NU + a] = u[a];
      }
      T m[NX], du[NU], mq = T(0), q = T(0), qz[ZC_ ? ZC_ : 1];
#pragma unroll
      for (int i = 0; i < (ZC_ ? ZC_ : 1); ++i) qz[i] = T(0);
#pragma unroll
      for (int i = 0; i < NX; ++i) m[i] = (col == i) ? T(1) : T(0);
#pragma unroll
      for (int a = 0; a < NU; ++a) du[a] = (col == NX + a) ? T(1) : T(0);
      const T t = tk(k);
      for (int s = 0; s < S; ++s) {
        rk4_step_parked<ZC_>(t, x, q, u, m, mq, du, qz);
      }
      LFSD_RCK(1)
      J += q;
      if constexpr (SCL) {
        T* Ms = Mwp(nxt) + (long long)k * Lay::MS_ELEMS;
        if (lane < LIVE_) {
#pragma unroll
          for (int i = 0; i < NX; ++i) Ms[i * Lay::LIVEP + lane] = m[i];
          Ms[NX * Lay::LIVEP + lane] = mq;
        }
        if (lane == 0) {
#pragma unroll
          for (int z = 0; z < ZC_; ++z) Ms[(NX + 1) * Lay::LIVEP + z] = qz[z];
        }
        LFSD_RCK(2)
        continue;
      }
      T* Mk = Mwp(nxt) + (long long)k * Lay::M_ELEMS;
      if (lane < LIVE_) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Mk[i * Lay::NXUP + col] = m[i];
        Mk[NX * Lay::NXUP + col] = mq;
      }
      if constexpr (ZC_ > 0) {
        if (lane < ZC_) {
#pragma unroll
          for (int i = 0; i < NX; ++i) Mk[i * Lay::NXUP + lane] = (i == lane) ? T(1) : T(0);
#pragma unroll
          for (int z = 0; z < ZC_; ++z) { if (lane == z) Mk[NX * Lay::NXUP + z] = qz[z]; }
        }
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xbp(nxt)[N * NX + i] = x[i];
    }
    J += M::final_cost(tk(N), x, e, c);
#if defined(LFSD_OC_CLOCK) && !defined(LFSD_EMU)
    if (threadIdx.x == 0 && blockIdx.x == 0) printf("rollout clock (wave 0): S %d intervals %d control %lld steps %lld stores %lld\n", S, N, rck[0], rck[1], rck[2]);
#endif
#undef LFSD_RCK
    return J;
  }

  // Column `lane` of the exact Hessian of the stage Lagrangian  Q_k(x,u) + lam'^T F_k(x,u)  w.r.t. (x_k,u_k):
  // second-order adjoint sweep through the S x 4 RK4 stages (tangent forward, adjoint + its tangent backward).
  // No cross-lane traffic: every lane recomputes the group-uniform stage states and parks them, with its own
  // tangents, in its private LDS slots (index (slot)*G + lane), so the routine may run under a divergent branch.
  // WIDE (one trajectory per wavefront, lanes work on different intervals): `col` is the column, `lane` only names the
  // private slots, and the sub-step start states are per lane too (exwu).
  template <bool WIDE = false>
  LFSD_DEV void stage_hessian_col(int k, const T* xk, const T* uk, const T* lam_next, T* hx, T* hu, int col = -1) {
    constexpr int SMAX = Lay::SMAX;
    if (!WIDE) col = lane;
    T* ex = lds + Lay::template lds_ex<G>();
    T* exu = WIDE ? exwu + lane : exws;      // [S][NX] uniform (all lanes store the same value) | [S][NX][G] per lane
    constexpr int XS = WIDE ? G : 1;
    T* exl = exws + SMAX * NX;               // [S][NX][G] per lane
    const T t = tk(k);
    const T h = DT;
    T x[NX], m[NX], du[NU], q = T(0), mq = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) { x[i] = xk[i]; m[i] = (col == i) ? T(1) : T(0); }
#pragma unroll
    for (int a = 0; a < NU; ++a) du[a] = (col == NX + a) ? T(1) : T(0);
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int i = 0; i < NX; ++i) { exu[(s * NX + i) * XS] = x[i]; exl[(s * NX + i) * G + lane] = m[i]; }
      rk4_step<true>(t, x, q, uk, m, mq, du);
    }
    T lam[NX], dlam[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) { lam[i] = lam_next[i]; dlam[i] = T(0); }
#pragma unroll
    for (int a = 0; a < NU; ++a) hu[a] = T(0);
    const T wgt[4] = {h / T(6), h / T(3), h / T(3), h / T(6)};
    const T car[4] = {h * T(0.5), h * T(0.5), h, T(0)};      // kappa_i = w_i*lam + car_i * ybar_{i+1}
    const T adv[3] = {h * T(0.5), h * T(0.5), h};
    for (int s = S - 1; s >= 0; --s) {
      // recompute the four stage points of this sub-step and park them
      T x0s[NX], m0s[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) { x0s[i] = exu[(s * NX + i) * XS]; m0s[i] = exl[(s * NX + i) * G + lane]; }
#pragma unroll
      for (int i = 0; i < NX; ++i) { x[i] = x0s[i]; m[i] = m0s[i]; }
      for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { ex[(st * NX + i) * G + lane] = x[i]; ex[((4 + st) * NX + i) * G + lane] = m[i]; }
        if (st < 3) {
          T f[NX], d[NX], cq, dq;
          M::dyn_cost_jvp(t, x, uk, e, c, m, du, f, cq, d, dq);
#pragma unroll
          for (int i = 0; i < NX; ++i) { x[i] = x0s[i] + adv[st] * f[i]; m[i] = m0s[i] + adv[st] * d[i]; }
        }
      }
      T yb[NX], dyb[NX], lam_new[NX], dlam_new[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) { yb[i] = T(0); dyb[i] = T(0); lam_new[i] = lam[i]; dlam_new[i] = dlam[i]; }
      for (int st = 3; st >= 0; --st) {
        T kap[NX], dkap[NX], ls[NX], xs[NX], ms[NX], y1[NX], t2x[NX], t2u[NU], gx[NX], gu[NU];
        const T w = wgt[st], cc = car[st], iw = T(1) / w;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          kap[i] = w * lam[i] + cc * yb[i];
          dkap[i] = w * dlam[i] + cc * dyb[i];
          ls[i] = kap[i] * iw;
          xs[i] = ex[(st * NX + i) * G + lane];
          ms[i] = ex[((4 + st) * NX + i) * G + lane];
        }
        M::dyn_vjp2(t, xs, uk, e, c, kap, w, dkap, y1, t2x, t2u);
        M::ham_hess_mul(t, xs, uk, ls, e, c, ms, du, gx, gu);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          yb[i] = y1[i];
          dyb[i] = t2x[i] + w * gx[i];
          lam_new[i] += yb[i];
          dlam_new[i] += dyb[i];
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) hu[a] += t2u[a] + w * gu[a];
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) { lam[i] = lam_new[i]; dlam[i] = dlam_new[i]; }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) hx[i] = dlam[i];
  }

  // One interval integrated and linearised with ALL its tangent columns on this lane -- NV packed pairs side by side that share the
  // nominal part of every model call, as in stage_hessian_all below (wide kernel, fp32, OcLayout::HALL: one item per interval
  // instead of one per (interval, column pair): the robot arm's 150 items = 3 rounds of the wavefront become 50 = one round that
  // costs less than two).  x is advanced in place, q is the interval's cost, [A B; q] of the interval goes to Mk ([row][column]).
  template <int NV>
  LFSD_DEV void interval_sens_all(T t, T* x, T& q, const T* u, T* Mk) const {
    using V = pk2<T>;
    const T hh = DT * T(0.5), h6 = DT / T(6);
    T el[NP > 0 ? NP : 1], cl[M::NCX > 0 ? M::NCX : 1];      // (in registers: see stage_hessian_all)
#pragma unroll
    for (int i = 0; i < NP; ++i) el[i] = e[i];
#pragma unroll
    for (int i = 0; i < M::NCX; ++i) cl[i] = c[i];
    V m[NV][NX], du[NV][NU], mq[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      mq[v] = V(T(0));
#pragma unroll
      for (int i = 0; i < NX; ++i) m[v][i] = mk2<T>((2 * v == i) ? T(1) : T(0), (2 * v + 1 == i) ? T(1) : T(0));
#pragma unroll
      for (int a = 0; a < NU; ++a) du[v][a] = mk2<T>((2 * v == NX + a) ? T(1) : T(0), (2 * v + 1 == NX + a) ? T(1) : T(0));
    }
    q = T(0);
    for (int s = 0; s < S; ++s) {
      T xs[NX], ax[NX], f[NX], cq, aq = T(0);
      V ms[NV][NX], am[NV][NX], d[NV][NX], dq[NV], adq[NV];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const T* xe = (st == 0) ? x : xs;
        M::template dyn_cost_jvp_n<NV>(t, xe, u, el, cl, (st == 0) ? &m[0][0] : &ms[0][0], &du[0][0], f, cq, &d[0][0], dq);
        const T wgt = (st == 0 || st == 3) ? T(1) : T(2), adv = (st < 2) ? hh : DT;
        aq = (st == 0) ? cq : aq + wgt * cq;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          ax[i] = (st == 0) ? f[i] : ax[i] + wgt * f[i];
          if (st < 3) xs[i] = x[i] + adv * f[i];
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          adq[v] = (st == 0) ? dq[v] : adq[v] + wgt * dq[v];
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            am[v][i] = (st == 0) ? d[v][i] : am[v][i] + wgt * d[v][i];
            if (st < 3) ms[v][i] = m[v][i] + adv * d[v][i];
          }
        }
      }
      q += h6 * aq;
#pragma unroll
      for (int i = 0; i < NX; ++i) x[i] += h6 * ax[i];
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        mq[v] += h6 * adq[v];
#pragma unroll
        for (int i = 0; i < NX; ++i) m[v][i] += h6 * am[v][i];
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      V* mc = reinterpret_cast<V*>(Mk + 2 * v);
#pragma unroll
      for (int i = 0; i < NX; ++i) mc[i * (Lay::NXUP / 2)] = m[v][i];
      mc[NX * (Lay::NXUP / 2)] = mq[v];
    }
  }

  // The same second-order adjoint for ALL columns of interval k on this one lane (wide kernel, fp32, OcLayout::HALL): NV packed
  // pairs of tangent columns run through the S x 4 stages side by side.  A column-per-item sweep re-evaluates the nominal part of
  // every model call -- for the robot arm four sin / cos, the mass matrix and its inverse: the larger part of a call -- once per
  // column; here the generated multi-tangent functions (codegen.py _body_multi: dyn_cost_jvp_n, dyn_vjp2_n, ham_hess_mul_n) emit the
  // nominal part once and the tangent part per pair.  (NV inlined calls of the one-tangent functions in one basic block do NOT get
  // there: measured on gfx950, the all-columns routines then cost exactly NV times the one-pair item.)  Robot arm (6 columns, 50 intervals): 300 (interval, column) items = 5 rounds of the wavefront become
  // 50 items = one round that costs about twice a single-column item.  Hk: the interval's Hessian [row][column] in the workspace.
  template <int NV>
  LFSD_DEV void stage_hessian_all(int k, const T* xk, const T* uk, const T* lam_next, T* Hk) {
    using V = pk2<T>;
    static_assert(G == 64, "one trajectory per wavefront");
    T* ex = lds + Lay::template lds_ex<G>();                           // [4][NX][G] stage points of the nominal, per lane
    V* exv = reinterpret_cast<V*>(ex + 4 * NX * G);                    // [4][NX][NV][G] ... of the tangent pairs
    T* exu = exwu + lane;                                              // [S][NX][G] sub-step starts of the nominal
    V* exl = reinterpret_cast<V*>(exwm) + lane;                        // [S][NX][NV][G] ... of the tangent pairs
    const T t = tk(k);
    const T h = DT, hh = DT * T(0.5), h6 = DT / T(6);
    // (parameters and constants in registers: read through the LDS pointers the compiler has to assume that the stores between two
    //  model calls may have changed them, and keeps a copy of the nominal part per call -- the sharing this routine exists for)
    T el[NP > 0 ? NP : 1], cl[M::NCX > 0 ? M::NCX : 1];
#pragma unroll
    for (int i = 0; i < NP; ++i) el[i] = e[i];
#pragma unroll
    for (int i = 0; i < M::NCX; ++i) cl[i] = c[i];
    T x[NX];
    V m[NV][NX], du[NV][NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = xk[i];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int i = 0; i < NX; ++i) m[v][i] = mk2<T>((2 * v == i) ? T(1) : T(0), (2 * v + 1 == i) ? T(1) : T(0));
#pragma unroll
      for (int a = 0; a < NU; ++a) du[v][a] = mk2<T>((2 * v == NX + a) ? T(1) : T(0), (2 * v + 1 == NX + a) ? T(1) : T(0));
    }
    // forward: the nominal and its NV tangent pairs through the S RK4 steps, sub-step starts parked
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        exu[(s * NX + i) * G] = x[i];
#pragma unroll
        for (int v = 0; v < NV; ++v) exl[((s * NX + i) * NV + v) * G] = m[v][i];
      }
      T xs[NX], ax[NX], f[NX], cq;
      V ms[NV][NX], am[NV][NX], d[NV][NX], dq[NV];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const T* xe = (st == 0) ? x : xs;
        M::template dyn_cost_jvp_n<NV>(t, xe, uk, el, cl, (st == 0) ? &m[0][0] : &ms[0][0], &du[0][0], f, cq, &d[0][0], dq);
        const T wgt = (st == 0 || st == 3) ? T(1) : T(2), adv = (st < 2) ? hh : h;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          ax[i] = (st == 0) ? f[i] : ax[i] + wgt * f[i];
          if (st < 3) xs[i] = x[i] + adv * f[i];
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            am[v][i] = (st == 0) ? d[v][i] : am[v][i] + wgt * d[v][i];
            if (st < 3) ms[v][i] = m[v][i] + adv * d[v][i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        x[i] += h6 * ax[i];
#pragma unroll
        for (int v = 0; v < NV; ++v) m[v][i] += h6 * am[v][i];
      }
    }
    T lam[NX];
    V dlam[NV][NX], hu[NV][NU];
#pragma unroll
    for (int i = 0; i < NX; ++i) lam[i] = lam_next[i];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int i = 0; i < NX; ++i) dlam[v][i] = V(T(0));
#pragma unroll
      for (int a = 0; a < NU; ++a) hu[v][a] = V(T(0));
    }
    const T wgt[4] = {h / T(6), h / T(3), h / T(3), h / T(6)};
    const T car[4] = {h * T(0.5), h * T(0.5), h, T(0)};      // kappa_i = w_i*lam + car_i * ybar_{i+1}
    const T adv[3] = {h * T(0.5), h * T(0.5), h};
    for (int s = S - 1; s >= 0; --s) {
      // recompute the four stage points of this sub-step and park them
      T x0s[NX];
      V m0s[NV][NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        x0s[i] = exu[(s * NX + i) * G]; x[i] = x0s[i];
#pragma unroll
        for (int v = 0; v < NV; ++v) { m0s[v][i] = exl[((s * NX + i) * NV + v) * G]; m[v][i] = m0s[v][i]; }
      }
      for (int st = 0; st < 4; ++st) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          ex[(st * NX + i) * G + lane] = x[i];
#pragma unroll
          for (int v = 0; v < NV; ++v) exv[((st * NX + i) * NV + v) * G + lane] = m[v][i];
        }
        if (st < 3) {
          T f[NX], cq;
          V d[NV][NX], dq[NV];
          M::template dyn_cost_jvp_n<NV>(t, x, uk, el, cl, &m[0][0], &du[0][0], f, cq, &d[0][0], dq);
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            x[i] = x0s[i] + adv[st] * f[i];
#pragma unroll
            for (int v = 0; v < NV; ++v) m[v][i] = m0s[v][i] + adv[st] * d[v][i];
          }
        }
      }
      T yb[NX], lam_new[NX];
      V dyb[NV][NX], dlam_new[NV][NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        yb[i] = T(0); lam_new[i] = lam[i];
#pragma unroll
        for (int v = 0; v < NV; ++v) { dyb[v][i] = V(T(0)); dlam_new[v][i] = dlam[v][i]; }
      }
      for (int st = 3; st >= 0; --st) {
        T kap[NX], ls[NX], xs[NX], y1[NX];
        const T w = wgt[st], cc = car[st], iw = T(1) / w;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          kap[i] = w * lam[i] + cc * yb[i];
          ls[i] = kap[i] * iw;
          xs[i] = ex[(st * NX + i) * G + lane];
        }
        V dkap[NV][NX], msv[NV][NX], t2x[NV][NX], t2u[NV][NU], gx[NV][NX], gu[NV][NU];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            dkap[v][i] = w * dlam[v][i] + cc * dyb[v][i];
            msv[v][i] = exv[((st * NX + i) * NV + v) * G + lane];
          }
        }
        // (one call each for the NV tangent pairs: the nominal part of the evaluation is computed once, codegen.py _body_multi)
        M::template dyn_vjp2_n<NV>(t, xs, uk, el, cl, kap, w, &dkap[0][0], y1, &t2x[0][0], &t2u[0][0]);
        M::template ham_hess_mul_n<NV>(t, xs, uk, ls, el, cl, &msv[0][0], &du[0][0], &gx[0][0], &gu[0][0]);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
#pragma unroll
          for (int i = 0; i < NX; ++i) { dyb[v][i] = t2x[v][i] + w * gx[v][i]; dlam_new[v][i] += dyb[v][i]; }
#pragma unroll
          for (int a = 0; a < NU; ++a) hu[v][a] += t2u[v][a] + w * gu[v][a];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) { yb[i] = y1[i]; lam_new[i] += yb[i]; }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        lam[i] = lam_new[i];
#pragma unroll
        for (int v = 0; v < NV; ++v) dlam[v][i] = dlam_new[v][i];
      }
    }
    // columns 2v, 2v+1 of every row as one 8-byte store ([row][column], rows of NXUP words)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      V* hc = reinterpret_cast<V*>(Hk + 2 * v);
#pragma unroll
      for (int i = 0; i < NX; ++i) hc[i * (Lay::NXUP / 2)] = dlam[v][i];
#pragma unroll
      for (int a = 0; a < NU; ++a) hc[(NX + a) * (Lay::NXUP / 2)] = hu[v][a];
    }
  }

  // Backward sweep on buffer `cur`: DDP gains + exact discrete costate (== IPOPT's lam_g).
  // Stage Hessian model: mode 0 Gauss-Newton (cost curvature), 1 interval * Hamiltonian Hessian (cheap Newton-like),
  // 2 exact Lagrangian Hessian of the RK4 stage (stage_hessian_col).
  LFSD_DEV bool backward(int cur, int mode, T mu, T& gnorm, T& dV1, T& dV2, T& dmin) {
    T* ldsV = lds + Lay::LDS_V;  T* ldsM = lds + Lay::LDS_M;  T* ldsK = lds + Lay::LDS_K;
    T* ldsQux = lds + Lay::LDS_QUX;  T* ldsQuu = lds + Lay::LDS_QUU;  T* ldsQu = lds + Lay::LDS_QU;
    T* ldsVx = lds + Lay::LDS_VX;  T* ldsLam = lds + Lay::LDS_LAM;  T* ldsRed = lds + Lay::LDS_RED;
    T Vx[NX], lam[NX], vcol[NX], xk[NX], uk[NU];
    bool ok = true;
    T gl_max = T(0);
    dV1 = T(0); dV2 = T(0); dmin = T(0);
    {
      const T* xN = xbp(cur) + N * NX;
#pragma unroll
      for (int i = 0; i < NX; ++i) xk[i] = xN[i];
      M::final_grad(tk(N), xk, e, c, Vx);
      T ox[NX], oe[NP];
      lam_max = T(0);
#pragma unroll
      for (int i = 0; i < NX; ++i) { ox[i] = (lane == i) ? T(1) : T(0); lam[i] = Vx[i]; lam_max = t_max(lam_max, t_abs(lam[i])); }
#pragma unroll
      for (int i = 0; i < NP; ++i) oe[i] = T(0);
      M::final_hess_mul(tk(N), xk, e, c, ox, oe, vcol);
      if (BND && xm != nullptr) {            // state-bound term of node N: gradient into V_x and the costate, Hessian onto the diagonal
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          T g_, h_;
          node_pen_d(N, i, xk[i], xlb[i], xub[i], g_, h_);
          Vx[i] += g_; lam[i] += g_;
          if (lane == i) vcol[i] += h_;
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) lam_out[N * NX + i] = lam[i];
      }
    }
    // this lane's column of [A B; q] and the nominal (x_k, u_k) of one interval.  The loads of interval k-1 are issued
    // while interval k is being processed (one wave per SIMD: nothing else would hide their latency)
    // QS (wide kernel, at most 16 columns): the backward sweep is one column per lane on 16 of the wavefront's 64 lanes.  Its
    // dense products  Y = V_xx m_j,  Q = [A B]^T Y  -- half of a stage's instructions -- are split four ways instead: every
    // 16-lane quarter holds the columns, quarter p computes rows p, p+4, p+8, (p+12) of Y and its share of the sum over
    // those rows in Q, and two cross-quarter exchanges per entry of Q add the shares up (quarter_sum).
    constexpr bool QS = (G == 64) && NX >= 8 && NXU <= 16 && !BND;
    const int jcol = QS ? (lane & 15) : lane;
    auto load_stage = [&](int k_, T* m_, T& mq_, T* xk_, T* uk_, T* dk_) LFSD_LAMBDA_INLINE {
      if (gap != nullptr) {                      // (the gap of the interval: group-uniform, loaded with the rest of the stage -- a stage ahead)
        const T* dp = gap + (long long)k_ * NX;
#pragma unroll
        for (int i = 0; i < NX; ++i) dk_[i] = dp[i];
      }
      if (jcol < NXU) {
        const T* Mk = Mwp(cur) + (long long)k_ * Lay::M_ELEMS + jcol;
#pragma unroll
        for (int i = 0; i < NX; ++i) m_[i] = Mk[i * Lay::NXUP];
        mq_ = Mk[NX * Lay::NXUP];
      } else {
#pragma unroll
        for (int i = 0; i < NX; ++i) m_[i] = T(0);
        mq_ = T(0);
      }
      const T* xp = xbp(cur) + k_ * NX;  const T* up = ubp(cur) + k_ * NU;
#pragma unroll
      for (int i = 0; i < NX; ++i) xk_[i] = xp[i];
#pragma unroll
      for (int a = 0; a < NU; ++a) uk_[a] = up[a];
    };
    // (the wide kernel and the lock-step kernels without a partner wavefront on their SIMD: one stage of look-ahead on the
    //  global loads of a stage -- rocket, wide fp32, n_grid 100: 128.8 -> 126.0 ms together with the costate sweep's,
    //  profiles/r03_t_generic_backward.txt; fp64 lean quadrotor: 9.4 -> 11.9 ms, the registers are not there: off)
    constexpr bool PFG = (LFSD_BW_PREFETCH_GEN) != 0 && (sizeof(T) == 4 || ((LFSD_BW_PREFETCH_GEN) & 2) != 0);
    T m[NX], mq = T(0), mN[NX], mqN = T(0), xkN[NX], ukN[NU], dv[NX], dvN[NX];
    T hC[(EXACT && PFG) ? NXU : 1], hN[(EXACT && PFG) ? NXU : 1];
#pragma unroll
    for (int i = 0; i < NX; ++i) { dv[i] = T(0); dvN[i] = T(0); }
    load_stage(N - 1, m, mq, xk, uk, dv);
    if (EXACT && PFG && mode == 2 && reuse_hess) {
      const T* hn = Hws + (long long)(N - 1) * Lay::H_ELEMS + (lane < NXU ? lane : 0);
#pragma unroll
      for (int i = 0; i < NXU; ++i) hC[i] = hn[i * Lay::NXUP];
    }
    for (int k = N - 1; k >= 0; --k) {
      if (lane < NX) {
#pragma unroll
        for (int i = 0; i < NX; ++i) ldsV[lane * NX + i] = vcol[i];
      }
      if (lane < NXU) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { if (QS) ldsM[i * NXU + lane] = m[i]; else ldsM[lane * NX + i] = m[i]; }      // QS: transposed, row i = entries (i, all columns)
      }
      LFSD_STAGE_SYNC_GEN();
      if (gap != nullptr) {
        // multiple shooting: the linearised interval ends d_k away from node k+1, so the value function of that node is entered
        // at delta x_k+1 = A dx + B du + d_k:  V_x <- V_x + V_xx d_k  (the costate recursion below stays the exact adjoint one)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          T sacc = T(0);
#pragma unroll
          for (int j = 0; j < NX; ++j) sacc += ldsV[j * NX + i] * dv[j];
          Vx[i] += sacc;
        }
      }
      if (PFG && k > 0) {
        load_stage(k - 1, mN, mqN, xkN, ukN, dvN);
        if (EXACT && mode == 2 && reuse_hess) {      // ... and its column of the cached stage Hessian
          const T* hn = Hws + (long long)(k - 1) * Lay::H_ELEMS + (lane < NXU ? lane : 0);
#pragma unroll
          for (int i = 0; i < NXU; ++i) hN[i] = hn[i * Lay::NXUP];
        }
        LFSD_ISSUE_FENCE();
      }
      // Y = Vxx' m_j ;  Qcol = [A B]^T Y
      T Y[NX], Qcol[NXU];
      if constexpr (QS) {
        constexpr int RP = (NX + 3) / 4;
        const int part = lane >> 4;
        T Yp[RP], Qp[NXU];
#pragma unroll
        for (int r = 0; r < NXU; ++r) Qp[r] = T(0);
        // (row addresses depend on the lane: the reads of all of a quarter's rows are issued back to back, then used)
        T vrow[RP][NX], mrow[RP][NXU];
#pragma unroll
        for (int q = 0; q < RP; ++q) {
          const int i = part + 4 * q;
          const T* vr = ldsV + ((i < NX) ? i : 0) * NX;
#pragma unroll
          for (int kk = 0; kk < NX; ++kk) vrow[q][kk] = vr[kk];
        }
#pragma unroll
        for (int q = 0; q < RP; ++q) {
#pragma unroll
          for (int kk = 0; kk < NX; ++kk) pin(vrow[q][kk]);
        }
#pragma unroll
        for (int q = 0; q < RP; ++q) {
          const int i = part + 4 * q;
          const T* mr = ldsM + ((i < NX) ? i : 0) * NXU;
#pragma unroll
          for (int r = 0; r < NXU; ++r) mrow[q][r] = mr[r];
        }
#pragma unroll
        for (int q = 0; q < RP; ++q) {
          T s = T(0);
#pragma unroll
          for (int kk = 0; kk < NX; ++kk) s += vrow[q][kk] * m[kk];
          Yp[q] = (part + 4 * q < NX) ? s : T(0);
        }
#pragma unroll
        for (int q = 0; q < RP; ++q) {
#pragma unroll
          for (int r = 0; r < NXU; ++r) pin(mrow[q][r]);
        }
#pragma unroll
        for (int q = 0; q < RP; ++q) {
#pragma unroll
          for (int r = 0; r < NXU; ++r) Qp[r] += mrow[q][r] * Yp[q];
        }
#pragma unroll
        for (int r = 0; r < NXU; ++r) Qcol[r] = quarter_sum(Qp[r]);
        (void)Y;
      } else if constexpr (sizeof(T) == 8 ? ((LFSD_FENCE64) & 4) != 0 : NX >= 8) {
        // a row's reads are issued while the previous row is multiplied (two row buffers, constant indices after the
        // unrolling).  fp64: with one buffer every row waited a full LDS round trip between the scheduling barriers; fp32
        // (wide kernel, lock-step kernels without MFMA): left to itself the compiler issued the 104 ds_read_b128 of a stage one
        // at a time, each followed by its wait, on a SIMD with nothing else to run (rocket, wide, n_grid 100, 1024 seeds: 126.0 ->
        // 113.7 ms).  Not for the small models: the robot arm's rows of 4 lose more to the barriers than they gain (17.8 -> 23.5 ms)
        T rowb[2][NX];
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) rowb[0][kk] = ldsV[kk];
#pragma unroll
        for (int r = 0; r < NX + NXU; ++r) {
          const T* nxt = (r + 1 < NX) ? ldsV + (r + 1) * NX : ldsM + (r + 1 - NX) * NX;
          if (r + 1 < NX + NXU) {
#pragma unroll
            for (int kk = 0; kk < NX; ++kk) rowb[(r + 1) & 1][kk] = nxt[kk];
          }
          T s = T(0);
          if (r < NX) {
#pragma unroll
            for (int kk = 0; kk < NX; ++kk) s += rowb[r & 1][kk] * m[kk];
          } else {
#pragma unroll
            for (int kk = 0; kk < NX; ++kk) s += rowb[r & 1][kk] * Y[kk];
          }
          pin(s);
          if (r < NX) Y[r] = s; else Qcol[r - NX] = s;
          LFSD_ROW_FENCE();
        }
      } else {
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        T s = T(0);
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) s += ldsV[i * NX + kk] * m[kk];
        pin64(s);
        Y[i] = s;
        LFSD_SCHED_FENCE64(T);
      }
#pragma unroll
      for (int r = 0; r < NXU; ++r) {
        T s = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) s += ldsM[r * NX + i] * Y[i];
        pin64(s);
        Qcol[r] = s;
        LFSD_SCHED_FENCE64(T);
      }
      }
      if (EXACT && mode == 2) {
        // column `lane` of the exact stage Hessian depends on the nominal and its costates only, not on the shift: a
        // retry of the sweep with a larger shift reads it back instead of repeating the second-order adjoint
        T hx[NX], hu[NU];
        T* hcol = Hws + (long long)k * Lay::H_ELEMS + (lane < NXU ? lane : 0);      // [row][column], column = lane
        if (reuse_hess && PFG) {
#pragma unroll
          for (int i = 0; i < NX; ++i) hx[i] = hC[i];
#pragma unroll
          for (int a = 0; a < NU; ++a) hu[a] = hC[NX + a];
        } else if (reuse_hess) {
#pragma unroll
          for (int i = 0; i < NX; ++i) hx[i] = hcol[i * Lay::NXUP];
#pragma unroll
          for (int a = 0; a < NU; ++a) hu[a] = hcol[(NX + a) * Lay::NXUP];
        } else {
          stage_hessian_col(k, xk, uk, lam, hx, hu);
          if (lane < NXU) {
#pragma unroll
            for (int i = 0; i < NX; ++i) hcol[i * Lay::NXUP] = hx[i];
#pragma unroll
            for (int a = 0; a < NU; ++a) hcol[(NX + a) * Lay::NXUP] = hu[a];
          }
        }
#if defined(LFSD_TRACE_HCOL)
        if (lane < NXU && blockIdx.x == 0 && threadIdx.x < G) {
          printf("HCOL k %d lane %d x", k, lane); for (int i = 0; i < NX; ++i) printf(" %.17g", (double)xk[i]);
          printf(" u"); for (int a = 0; a < NU; ++a) printf(" %.17g", (double)uk[a]);
          printf(" l"); for (int i = 0; i < NX; ++i) printf(" %.17g", (double)lam[i]);
          printf(" h"); for (int i = 0; i < NX; ++i) printf(" %.17g", (double)hx[i]); for (int a = 0; a < NU; ++a) printf(" %.17g", (double)hu[a]);
          printf("\n");
        }
#endif
#pragma unroll
        for (int i = 0; i < NX; ++i) Qcol[i] += hx[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) Qcol[NX + a] += hu[a];
      } else {
        T ox[NX], ou[NU], ls[NX], hx[NX], hu[NU];
        const T HL = (mode == 1) ? T(1) : T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) { ox[i] = (lane == i) ? T(1) : T(0); ls[i] = HL * lam[i]; }
#pragma unroll
        for (int a = 0; a < NU; ++a) ou[a] = (lane == NX + a) ? T(1) : T(0);
        M::ham_hess_mul(tk(k), xk, uk, ls, e, c, ox, ou, hx, hu);
#pragma unroll
        for (int i = 0; i < NX; ++i) Qcol[i] += dgrid * hx[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) Qcol[NX + a] += dgrid * hu[a];
      }
      T Qg = mq, gl = mq;
#pragma unroll
      for (int i = 0; i < NX; ++i) { Qg += m[i] * Vx[i]; gl += m[i] * lam[i]; }
      T Quxj[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) Quxj[a] = Qcol[NX + a];
      if (lane < NX) {
#pragma unroll
        for (int a = 0; a < NU; ++a) ldsQux[lane * NU + a] = Quxj[a];
      } else if (lane < NXU) {
        const int b = lane - NX;
#pragma unroll
        for (int a = 0; a < NU; ++a) ldsQuu[b * NU + a] = Quxj[a];
        ldsQu[b] = Qg;
        T glp = gl;
        if (BND) {
          // projected gradient: a control sitting on a bound with the descent direction pointing out of the box is stationary
          T ukb = T(0), lbb = T(0), ubb = T(0);
#pragma unroll
          for (int a = 0; a < NU; ++a) { if (a == b) { ukb = uk[a]; lbb = ulb[a]; ubb = uub[a]; } }
          if ((ukb <= lbb && gl > T(0)) || (ukb >= ubb && gl < T(0))) glp = T(0);
        }
        gl_max = t_max(gl_max, t_abs(glp));
      }
      LFSD_STAGE_SYNC_GEN();
      T Quu0[NU * NU], Lc[NU * NU], Qu[NU], kff[NU], Kj[NU], t1[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        Qu[a] = ldsQu[a];
#pragma unroll
        for (int b = 0; b < NU; ++b) Quu0[a * NU + b] = T(0.5) * (ldsQuu[b * NU + a] + ldsQuu[a * NU + b]);
      }
      // The Levenberg shift PER STAGE (round 4; solves that run on exact stage Hessians from the first iteration, lfsd_coc_solve):
      // a stage whose Q_uu is positive definite with mu_stage_frac of the shift keeps only that fraction; the full shift goes
      // where the factorisation needs it.  The rocket's slowest solves were accepted full steps gaining 1-2 % each at shifts
      // 20x the size of Q_uu (profiles/r04_s_rocket_trace.txt): negative curvature sits in a few stages, one shift for all
      // damps every stage.  Still the block LDL^T of (Lagrangian Hessian + diag(mu_k)): the value recursion stays consistent.
      T mu_k = mu;
      if constexpr (!BND) {
        if (mu > T(0) && mu_stage_frac > T(0)) {
          T Lt[NU * NU], dd = T(0);
#pragma unroll
          for (int i = 0; i < NU * NU; ++i) Lt[i] = Quu0[i];
#pragma unroll
          for (int a = 0; a < NU; ++a) Lt[a * NU + a] += mu * mu_stage_frac;
          if (chol_factor<NU>(Lt, dd)) mu_k = mu * mu_stage_frac;
        }
      }
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Lc[i] = Quu0[i];
#pragma unroll
      for (int a = 0; a < NU; ++a) Lc[a * NU + a] += mu_k;
      {      // (the value recursion continues with the SHIFTED Q_uu: cpdp_common.h, "Levenberg shift of the Newton modes")
#pragma unroll
        for (int a = 0; a < NU; ++a) Quu0[a * NU + a] += mu_k;
      }
      if (BND) {
        // control-limited step: box QP for the feed-forward part, zero feedback gain on the clamped components
        T lo[NU], hi[NU], Lf[NU * NU], dd = T(0);
#pragma unroll
        for (int a = 0; a < NU; ++a) { lo[a] = ulb[a] - uk[a]; hi[a] = uub[a] - uk[a]; }
        unsigned cmask = 0u;
        const bool okq = box_qp<NU>(Lc, Qu, lo, hi, kff, cmask, Lf, dd);
        if (ok && !okq) { ok = false; if (dd < dmin) dmin = dd; }
#pragma unroll
        for (int a = 0; a < NU; ++a) Kj[a] = ((cmask >> a) & 1u) ? T(0) : -Quxj[a];
        chol_solve<NU>(Lf, Kj);
      } else {
        if (ok) ok = chol_factor<NU>(Lc, dmin); else { T dd = T(0); chol_factor<NU>(Lc, dd); }   // first failing pivot sizes the shift
        // wide kernel (the wavefront is one trajectory, `ok` is uniform; its stage Hessians are computed before the sweep): a sweep
        // whose Q_uu + mu I is not positive definite is over -- the caller only reads the failing pivot -- and the stages below it
        // are the larger part of a retry ladder's cost (several shifts in a row on the way up)
        if constexpr (G == 64) { if (!ok) break; }
#pragma unroll
        for (int a = 0; a < NU; ++a) { kff[a] = -Qu[a]; Kj[a] = -Quxj[a]; }
        chol_solve<NU>(Lc, kff);
        chol_solve<NU>(Lc, Kj);
      }
      T qk[NU];
      matvec<NU>(Quu0, kff, qk);
#pragma unroll
      for (int a = 0; a < NU; ++a) { dV1 += kff[a] * Qu[a]; dV2 += T(0.5) * kff[a] * qk[a]; }
      T Vxj = Qg;
#pragma unroll
      for (int a = 0; a < NU; ++a) { Vxj += Kj[a] * (qk[a] + Qu[a]); Vxj += Quxj[a] * kff[a]; }
      if (lane < NX) {
#pragma unroll
        for (int a = 0; a < NU; ++a) ldsK[lane * NU + a] = Kj[a];
        ldsVx[lane] = Vxj;
        ldsLam[lane] = gl;
        T* Kout = Kws + ((long long)k * NX + lane) * NU;
#pragma unroll
        for (int a = 0; a < NU; ++a) Kout[a] = Kj[a];
      }
      if (lane == 0) {
#pragma unroll
        for (int a = 0; a < NU; ++a) kws[k * NU + a] = kff[a];
      }
      LFSD_STAGE_SYNC_GEN();
      matvec<NU>(Quu0, Kj, t1);
#pragma unroll
      for (int a = 0; a < NU; ++a) t1[a] += Quxj[a];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        T s = Qcol[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) { s += ldsK[i * NU + a] * t1[a]; s += ldsQux[i * NU + a] * Kj[a]; }
        vcol[i] = s;
        Vx[i] = ldsVx[i];
        lam[i] = ldsLam[i];
        lam_max = t_max(lam_max, t_abs(lam[i]));
      }
      if (BND && xm != nullptr && k > 0) {   // state-bound term of node k (x_0 is given, not bounded: CPDP.py:131-134)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          T g_, h_;
          node_pen_d(k, i, xk[i], xlb[i], xub[i], g_, h_);
          Vx[i] += g_; lam[i] += g_;
          if (lane == i) vcol[i] += h_;
        }
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) lam_out[k * NX + i] = lam[i];
      }
      // symmetrise V_xx through LDS (ldsV of this stage has been fully consumed above)
      if (lane < NX) {
#pragma unroll
        for (int i = 0; i < NX; ++i) ldsV[lane * NX + i] = vcol[i];
      }
      LFSD_STAGE_SYNC_GEN();
      if (lane < NX) {
#pragma unroll
        for (int i = 0; i < NX; ++i) vcol[i] = T(0.5) * (vcol[i] + ldsV[i * NX + lane]);
      }
      LFSD_STAGE_SYNC_GEN();
      if (k > 0) {
        if (PFG) {
#pragma unroll
          for (int i = 0; i < NX; ++i) { m[i] = mN[i]; xk[i] = xkN[i]; dv[i] = dvN[i]; }
#pragma unroll
          for (int a = 0; a < NU; ++a) uk[a] = ukN[a];
          mq = mqN;
          if (EXACT) {
#pragma unroll
            for (int i = 0; i < NXU; ++i) hC[i] = hN[i];
          }
        } else {
          load_stage(k - 1, m, mq, xk, uk, dv);
        }
      }
    }
    ldsRed[lane] = gl_max;
    __syncthreads();
    gnorm = T(0);
#pragma unroll
    for (int a = 0; a < NU; ++a) gnorm = t_max(gnorm, ldsRed[NX + a]);
    __syncthreads();
    if (!t_finite(gnorm) || !t_finite(dV1) || !t_finite(dV2)) ok = false;
    return ok;
  }

  // ---- the backward sweep on the matrix cores (fp32, 16-lane groups: four trajectories per wavefront, ONE pass) -------
  // The two dense products of a stage,  Y = V_xx [A B]  and  Q = [A B]^T Y,  run as rank-1 updates of the 4-block MFMA
  // v_mfma_f32_16x16x1_4b_f32 (mfma4b): block b = trajectory b of the wavefront, operands straight from the registers
  // of the column-per-lane layout (lane i of a group owns column i of V_xx -- symmetric, so also row i -- and column i
  // of [A B]); the result comes back column-per-lane after one register/lane-group transposition (tile_transpose).
  // 2 x NX MFMAs per stage for FOUR trajectories replace 2 x (NX*NX + NXU*NX) LDS-fed FMAs per lane for two of them.
  // The 16 lanes carry columns 0..15; a 17th column (quadrotor: NX + NU = 17) is carried as a row-per-lane vector in LDS
  // and enters through NX-term dot products; its rows of Q and of the stage Hessian follow from symmetry, its diagonal
  // element from M::ham_huu.  Stage-Hessian models 0 / 1 only (lean kernel); the exact model keeps the 32-lane sweep.
  static constexpr int NCL = (NXU < 16) ? NXU : 16;      // columns carried by lanes
  static constexpr int NUL = NCL - NX;                   // ... of them control columns
  static constexpr int NEXT = NXU - NCL;                 // columns carried in LDS (0 or 1)
  LFSD_DEV void mf_load_stage(int cur, int k, T* m, T& mq, T& me, T* xk, T* uk) const {
    const T* Mk = Mwp(cur) + (long long)k * Lay::M_ELEMS;
    if (lane < NCL) {
#pragma unroll
      for (int i = 0; i < NX; ++i) m[i] = Mk[i * Lay::NXUP + lane];
      mq = Mk[NX * Lay::NXUP + lane];
    } else {
#pragma unroll
      for (int i = 0; i < NX; ++i) m[i] = T(0);
      mq = T(0);
    }
    me = (NEXT && lane <= NX) ? Mk[lane * Lay::NXUP + NCL] : T(0);
    const T* xp = xbp(cur) + k * NX;  const T* up = ubp(cur) + k * NU;
#pragma unroll
    for (int i = 0; i < NX; ++i) xk[i] = xp[i];
#pragma unroll
    for (int a = 0; a < NU; ++a) uk[a] = up[a];
  }
  LFSD_DEV bool backward_mf(int cur, int mode, T mu, bool live, T& gnorm, T& dV1, T& dV2, T& dmin) {
    static_assert(G == 16 && NX <= 16 && NEXT <= 1, "MFMA backward sweep: 16-lane groups, at most one column beyond 16");
    T* ldsV = lds + Lay::LDS_V;  T* ldsK = lds + Lay::LDS_K;
    T* ldsQux = lds + Lay::LDS_QUX;  T* ldsQuu = lds + Lay::LDS_QUU;  T* ldsQu = lds + Lay::LDS_QU;
    T* ldsVx = lds + Lay::LDS_VX;  T* ldsLam = lds + Lay::LDS_LAM;  T* ldsRed = lds + Lay::LDS_RED;
    T* ldsME = lds + Lay::LDS_M;              // [NX] extra column of [A B], [NX] its q entry   (LDS_M region: NXU*NX words)
    T* ldsYE = ldsME + NX + 1;                // [NX] V_xx times that column
    T Vx[NX], lam[NX], vcol[NX], xk[NX], uk[NU];
    bool ok = true;
    T gl_max = T(0);
    dV1 = T(0); dV2 = T(0); dmin = T(0);
    {
      const T* xN = xbp(cur) + N * NX;
#pragma unroll
      for (int i = 0; i < NX; ++i) xk[i] = xN[i];
      M::final_grad(tk(N), xk, e, c, Vx);
      T ox[NX], oe[NP];
#pragma unroll
      for (int i = 0; i < NX; ++i) { ox[i] = (lane == i) ? T(1) : T(0); lam[i] = Vx[i]; }
#pragma unroll
      for (int i = 0; i < NP; ++i) oe[i] = T(0);
      M::final_hess_mul(tk(N), xk, e, c, ox, oe, vcol);      // lanes >= NX: ox = 0  ->  vcol = 0
      if (lane == 0 && live) {
#pragma unroll
        for (int i = 0; i < NX; ++i) lam_out[N * NX + i] = lam[i];
      }
    }
    // this lane's column of [A B; q], its row of the LDS-carried column and the nominal (x_k, u_k) of one interval: the
    // loads of interval k-1 are issued while interval k is being processed (one wave per SIMD: nothing else would hide
    // the latency of the workspace, which does not fit the caches)
    T m[NX], mq = T(0), me = T(0);
    mf_load_stage(cur, N - 1, m, mq, me, xk, uk);
    for (int k = N - 1; k >= 0; --k) {
      if (NEXT) {                                  // column NCL of [A B; q]: row `lane` of it, parked in LDS
        if (lane <= NX) ldsME[lane] = me;
      }
      // Y = V_xx [A B](:, 0..15)
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) mfma4b(vcol[kk], m[kk], acc);
      // stage-Hessian column of this lane (vector pipe; independent of the products in flight on the matrix pipe)
      T hx[NX], hu[NU], huu[NU * NU];
      {
        T ox[NX], ou[NU], ls[NX];
        const T HL = (mode == 1) ? T(1) : T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) { ox[i] = (lane == i) ? T(1) : T(0); ls[i] = HL * lam[i]; }
#pragma unroll
        for (int a = 0; a < NU; ++a) ou[a] = (lane == NX + a) ? T(1) : T(0);
        M::ham_hess_mul(tk(k), xk, uk, ls, e, c, ox, ou, hx, hu);
        if (NEXT) M::ham_huu(tk(k), xk, uk, ls, e, c, huu);
      }
      tile_transpose(acc);
      T ycol[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) ycol[r] = acc[r];
      __syncthreads();                             // ldsME visible
      T q_e = T(0), q_ee = T(0), Qg_e = T(0), gl_e = T(0);      // row NCL of Q at this lane's column; its diagonal; gradient terms
      if (NEXT) {
        T ye = T(0);
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) { ye += vcol[kk] * ldsME[kk]; q_e += ldsME[kk] * ycol[kk]; }
        if (lane < NX) ldsYE[lane] = ye;           // (V_xx m_e)_lane : V_xx row `lane` == this lane's column
      }
      // Q(0..15, 0..15) = [A B]^T Y
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) mfma4b(m[kk], ycol[kk], acc);
      T Qg = mq, gl = mq;
#pragma unroll
      for (int i = 0; i < NX; ++i) { Qg += m[i] * Vx[i]; gl += m[i] * lam[i]; }
      tile_transpose(acc);
      T Qcol[NXU];
#pragma unroll
      for (int r = 0; r < NCL; ++r) Qcol[r] = acc[r];
      __syncthreads();                             // ldsYE visible
      if (NEXT) {
        Qg_e = ldsME[NX]; gl_e = ldsME[NX];
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) { q_ee += ldsME[kk] * ldsYE[kk]; Qg_e += ldsME[kk] * Vx[kk]; gl_e += ldsME[kk] * lam[kk]; }
        // symmetry of the stage Hessian: row NCL of column j is the last control entry of column j's own product
        q_e += dgrid * hu[NU - 1];
        q_ee += dgrid * huu[NU * NU - 1];
        Qcol[NCL] = q_e;
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) Qcol[i] += dgrid * hx[i];
#pragma unroll
      for (int a = 0; a < NUL; ++a) Qcol[NX + a] += dgrid * hu[a];
      T Quxj[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) Quxj[a] = Qcol[NX + a];
      if (lane < NX) {
#pragma unroll
        for (int a = 0; a < NU; ++a) ldsQux[lane * NU + a] = Quxj[a];
      } else if (lane < NCL) {
        const int b = lane - NX;
#pragma unroll
        for (int a = 0; a < NU; ++a) ldsQuu[b * NU + a] = Quxj[a];
        ldsQu[b] = Qg;
        gl_max = t_max(gl_max, t_abs(gl));
        if (NEXT) ldsQuu[(NU - 1) * NU + b] = q_e;           // column NCL of Q_uu from its row (symmetric)
      }
      if (NEXT) {
        if (lane == 0) { ldsQuu[NU * NU - 1] = q_ee; ldsQu[NU - 1] = Qg_e; }
        gl_max = t_max(gl_max, (lane == NX) ? t_abs(gl_e) : T(0));
      }
      __syncthreads();
      T Quu0[NU * NU], Lc[NU * NU], Qu[NU], kff[NU], Kj[NU], t1[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        Qu[a] = ldsQu[a];
#pragma unroll
        for (int b = 0; b < NU; ++b) Quu0[a * NU + b] = T(0.5) * (ldsQuu[b * NU + a] + ldsQuu[a * NU + b]);
      }
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Lc[i] = Quu0[i];
#pragma unroll
      for (int a = 0; a < NU; ++a) Lc[a * NU + a] += mu;
      if (ok) ok = chol_factor<NU>(Lc, dmin); else { T dd = T(0); chol_factor<NU>(Lc, dd); }
      {      // (the value recursion continues with the SHIFTED Q_uu: cpdp_common.h, "Levenberg shift of the Newton modes")
#pragma unroll
        for (int a = 0; a < NU; ++a) Quu0[a * NU + a] += mu;
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) { kff[a] = -Qu[a]; Kj[a] = -Quxj[a]; }
      chol_solve<NU>(Lc, kff);
      chol_solve<NU>(Lc, Kj);
      T qk[NU];
      matvec<NU>(Quu0, kff, qk);
#pragma unroll
      for (int a = 0; a < NU; ++a) { dV1 += kff[a] * Qu[a]; dV2 += T(0.5) * kff[a] * qk[a]; }
      T Vxj = Qg;
#pragma unroll
      for (int a = 0; a < NU; ++a) { Vxj += Kj[a] * (qk[a] + Qu[a]); Vxj += Quxj[a] * kff[a]; }
      if (lane < NX) {
#pragma unroll
        for (int a = 0; a < NU; ++a) ldsK[lane * NU + a] = Kj[a];
        ldsVx[lane] = Vxj;
        ldsLam[lane] = gl;
        if (live) {
          T* Kout = Kws + ((long long)k * NX + lane) * NU;
#pragma unroll
          for (int a = 0; a < NU; ++a) Kout[a] = Kj[a];
        }
      }
      if (lane == 0 && live) {
#pragma unroll
        for (int a = 0; a < NU; ++a) kws[k * NU + a] = kff[a];
      }
      __syncthreads();
      matvec<NU>(Quu0, Kj, t1);
#pragma unroll
      for (int a = 0; a < NU; ++a) t1[a] += Quxj[a];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        T sacc = Qcol[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) { sacc += ldsK[i * NU + a] * t1[a]; sacc += ldsQux[i * NU + a] * Kj[a]; }      // (two FMAs; one statement compiles to mul + fma + add)
        vcol[i] = sacc;
        Vx[i] = ldsVx[i];
        lam[i] = ldsLam[i];
      }
      if (lane == 0 && live) {
#pragma unroll
        for (int i = 0; i < NX; ++i) lam_out[k * NX + i] = lam[i];
      }
      // symmetrise V_xx through LDS (the rank-1 feeds above rely on row i == column i); lanes >= NX carry no column
      if (lane < NX) {
#pragma unroll
        for (int i = 0; i < NX; ++i) ldsV[lane * NX + i] = vcol[i];
      }
      __syncthreads();
      {
        const int lv = lane < NX ? lane : 0;
        T vt[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) vt[i] = ldsV[i * NX + lv];
#pragma unroll
        for (int i = 0; i < NX; ++i) pin(vt[i]);
#pragma unroll
        for (int i = 0; i < NX; ++i) vcol[i] = (lane < NX) ? T(0.5) * (vcol[i] + vt[i]) : T(0);
      }
      __syncthreads();
      if (k > 0) mf_load_stage(cur, k - 1, m, mq, me, xk, uk);
    }
    ldsRed[lane] = gl_max;
    __syncthreads();
    gnorm = T(0);
#pragma unroll
    for (int a = 0; a < NUL; ++a) gnorm = t_max(gnorm, ldsRed[NX + a]);      // (lane NX also carries the LDS column's entry)
    __syncthreads();
    if (!t_finite(gnorm) || !t_finite(dV1) || !t_finite(dV2)) ok = false;
    return ok;
  }

  // Costates and gradient norm of nominal `cur` WITHOUT the value recursion (16-lane groups of the lean fp32 kernel):
  //   lam_k = A_k^T lam_k+1 + l_x,   dJ/du_k = B_k^T lam_k+1 + l_u
  // from the stored [A B; q] alone -- two dot products per lane and stage where the full sweep (backward_mf) does two
  // matrix products, a factorisation and two solves.  A solve that is about to pass its convergence test needs nothing
  // else from its last sweep: oc_solve_kernel tries this first when the previous Newton step already predicted a decrease
  // below the resolution of the cost, and pays for the full sweep only if the gradient test then fails.
  LFSD_DEV void costate_sweep_mf(int cur, bool live, T& gnorm) {
    static_assert(G == 16 && NX <= 16 && NEXT <= 1, "16-lane groups, at most one column beyond 16");
    T* ldsME = lds + Lay::LDS_M;  T* ldsLam = lds + Lay::LDS_LAM;  T* ldsRed = lds + Lay::LDS_RED;
    T lam[NX], xk[NX], uk[NU];
    {
      const T* xN = xbp(cur) + N * NX;
#pragma unroll
      for (int i = 0; i < NX; ++i) xk[i] = xN[i];
      M::final_grad(tk(N), xk, e, c, lam);
      if (lane == 0 && live) {
#pragma unroll
        for (int i = 0; i < NX; ++i) lam_out[N * NX + i] = lam[i];
      }
    }
    T gl_max = T(0);
    T m[NX], mq = T(0), me = T(0), mN[NX], mqN = T(0), meN = T(0);
    mf_load_stage(cur, N - 1, m, mq, me, xk, uk);
    for (int k = N - 1; k >= 0; --k) {
      if (NEXT) { if (lane <= NX) ldsME[lane] = me; }
      if (k > 0) { mf_load_stage(cur, k - 1, mN, mqN, meN, xk, uk); LFSD_ISSUE_FENCE(); }
      T gl = mq;
#pragma unroll
      for (int i = 0; i < NX; ++i) gl += m[i] * lam[i];
      if (lane < NX) ldsLam[lane] = gl;
      else if (lane < NCL) gl_max = t_max(gl_max, t_abs(gl));
      __syncthreads();
      if (NEXT) {
        T gl_e = ldsME[NX];
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) gl_e += ldsME[kk] * lam[kk];
        gl_max = t_max(gl_max, (lane == NX) ? t_abs(gl_e) : T(0));
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) lam[i] = ldsLam[i];
      if (lane == 0 && live) {
#pragma unroll
        for (int i = 0; i < NX; ++i) lam_out[k * NX + i] = lam[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NX; ++i) m[i] = mN[i];
      mq = mqN; me = meN;
    }
    ldsRed[lane] = gl_max;
    __syncthreads();
    gnorm = T(0);
#pragma unroll
    for (int a = 0; a < NUL; ++a) gnorm = t_max(gnorm, ldsRed[NX + a]);
    __syncthreads();
  }

  // =====================================================================================================================
  //  Lean fp32 kernel with STRUCTURALLY CONSTANT tangent columns (M::NZC > 0: quadrotor / rocket position, round 3)
  // =====================================================================================================================
  // The dynamics do not depend on the leading ZC state components, so their columns of [A_k B_k] are exact unit vectors
  // through every RK4 stage: they are neither propagated (the packed roll-out carries the LIVE = NXU - ZC other columns, 7
  // lanes of 16 for the quadrotor instead of 9), nor stored (14 x 14 + 4 words per interval instead of 14 x 18), nor
  // multiplied: with  M = [E_Z  M_live]
  //     Y = V M       ->  Y(:, Z) = V(:, Z)                      (free)
  //     Q = M^T Y     ->  Q(Z, live) = Y(Z, live),  Q(Z, Z) = V(Z, Z)      (free; rows of what the live lanes hold anyway)
  // and all 17 columns of the quadrotor fit the 16 lanes of a group WITHOUT the LDS-carried 17th column of backward_mf:
  //     lane L < LX = NX - ZC     M-role: live state column ZC + L       V-role: the same state
  //     lane LX + a, a < NU       M-role: control column a               V-role (a < ZC): constant state a
  // A lane's two roles share every instruction; the constant state's column of Q is gathered from the rows Y(Z, :) the
  // other lanes publish in LDS.  The stage Hessian of a constant state couples to nothing outside Z (codegen checks), so
  // one ham_hess_mul with BOTH one-hots set returns the control column's entries and the constant state's side by side.
  static constexpr int ZC = Lay::ZC, LIVE = Lay::LIVE, LX = NX - Lay::ZC, MS = Lay::MS_ELEMS, LIVEP = Lay::LIVEP;
  LFSD_DEV T rollout_sens_sc(int cur, int nxt, T alpha, bool gains) {
    using V = pk2<T>;
    T x[NX], u[NU], J = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = x0[i];
    const int c0 = ZC + 2 * lane, c1 = c0 + 1;             // original column indices of this lane's pair of live columns
    // the operands of the control law (nominal, feed-forward, gains of interval k: group-uniform global loads) are fetched while
    // interval k-1 is integrated: on the one-step-per-interval levels of the mesh continuation an interval is ~4 000 clocks of
    // arithmetic, and its control law otherwise starts by waiting a global round trip for 73 words
    T ubN[NU], xbN[NX], KN[NX * NU], kN[NU];
    auto load_ctl = [&](int k_) LFSD_LAMBDA_INLINE {
      const T* ubk = ubp(cur) + k_ * NU;
#pragma unroll
      for (int a = 0; a < NU; ++a) ubN[a] = ubk[a];
      if (gains) {
        const T* xbk = xbp(cur) + k_ * NX;
        const T* Kk = Kws + (long long)k_ * NX * NU;
        const T* kk = kws + k_ * NU;
#pragma unroll
        for (int i = 0; i < NX; ++i) xbN[i] = xbk[i];
#pragma unroll
        for (int i = 0; i < NX * NU; ++i) KN[i] = Kk[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) kN[a] = kk[a];
      }
    };
    load_ctl(0);
    for (int k = 0; k < N; ++k) {
#pragma unroll
      for (int a = 0; a < NU; ++a) u[a] = ubN[a];
      if (gains) {
#pragma unroll
        for (int a = 0; a < NU; ++a) u[a] += alpha * kN[a];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const T dx = x[i] - xbN[i];
#pragma unroll
          for (int a = 0; a < NU; ++a) u[a] += KN[i * NU + a] * dx;
        }
      }
      if (k + 1 < N) { load_ctl(k + 1); LFSD_ISSUE_FENCE(); }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xbp(nxt)[k * NX + i] = x[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) ubp(nxt)[k * NU + a] = u[a];
      }
      V m[NX], du[NU], mq = V(T(0));
      T q = T(0), qz[ZC ? ZC : 1];
#pragma unroll
      for (int i = 0; i < ZC; ++i) qz[i] = T(0);
#pragma unroll
      for (int i = 0; i < NX; ++i) m[i] = mk2<T>((c0 == i) ? T(1) : T(0), (c1 == i) ? T(1) : T(0));
#pragma unroll
      for (int a = 0; a < NU; ++a) du[a] = mk2<T>((c0 == NX + a) ? T(1) : T(0), (c1 == NX + a) ? T(1) : T(0));
      const T t = tk(k);
      for (int s = 0; s < S; ++s) rk4_step<true, V, ZC>(t, x, q, u, m, mq, du, qz);
      J += q;
      T* Mk = Mwp(nxt) + (long long)k * MS;
      if (c0 < NXU) {          // live columns 2 lane, 2 lane + 1 of every row as one 8-byte store
        V* Mv = reinterpret_cast<V*>(Mk + 2 * lane);
#pragma unroll
        for (int i = 0; i < NX; ++i) Mv[i * (LIVEP / 2)] = m[i];
        Mv[NX * (LIVEP / 2)] = mq;
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < ZC; ++i) Mk[(NX + 1) * LIVEP + i] = qz[i];
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xbp(nxt)[N * NX + i] = x[i];
    }
    J += M::final_cost(tk(N), x, e, c);
    return J;
  }
  // this lane's live column of [A B; q], the cost-row entry of its constant state, the nominal of one interval
  LFSD_DEV void sc_load_stage(int cur, int k, T* m, T& mq, T& qzl, T* xk, T* uk) const {
    const T* Mk = Mwp(cur) + (long long)k * MS;
    if (lane < LIVE) {
#pragma unroll
      for (int i = 0; i < NX; ++i) m[i] = Mk[i * LIVEP + lane];
      mq = Mk[NX * LIVEP + lane];
    } else {
#pragma unroll
      for (int i = 0; i < NX; ++i) m[i] = T(0);
      mq = T(0);
    }
    qzl = (lane >= LX && lane < NX) ? Mk[(NX + 1) * LIVEP + (lane - LX)] : T(0);
    const T* xp = xbp(cur) + k * NX;  const T* up = ubp(cur) + k * NU;
#pragma unroll
    for (int i = 0; i < NX; ++i) xk[i] = xp[i];
#pragma unroll
    for (int a = 0; a < NU; ++a) uk[a] = up[a];
  }
  // diagnostic build (-DLFSD_BW_CLOCK, tools/bw_clock.py): shader clocks of the six phases of a stage, summed over a sweep
#if defined(LFSD_BW_CLOCK) && !defined(LFSD_EMU)
  long long bwc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define LFSD_BWC(i) { const long long t_ = clock64(); bwc[i] += t_ - bwc_t; bwc_t = t_; }
#else
#define LFSD_BWC(i)
#endif
  LFSD_DEV bool backward_sc(int cur, int mode, T mu, bool live, T& gnorm, T& dV1, T& dV2, T& dmin) {
    static_assert(!Lay::sc_ok || (G == 16 && LIVE <= 16 && ZC <= NU && NX <= 16), "structural sweep: 16-lane groups");
    T* ldsV = lds + Lay::LDS_V;  T* ldsK = lds + Lay::LDS_K;
    T* ldsQux = lds + Lay::LDS_QUX;  T* ldsQuu = lds + Lay::LDS_QUU;  T* ldsQu = lds + Lay::LDS_QU;
    T* ldsVx = lds + Lay::LDS_VX;  T* ldsLam = lds + Lay::LDS_LAM;  T* ldsRed = lds + Lay::LDS_RED;
    // MM: the two dense products on the matrix cores (fp32).  Otherwise (fp64, round 3) they are LDS-fed FMAs in the same lane
    // roles: V_xx (symmetric, natural row order) and this stage's live columns are published in LDS, every lane reads the rows
    // it needs two row buffers ahead -- ONE pass for the wavefront's four trajectories where the generic sweep took two.
    constexpr bool MM = sizeof(T) == 4;
    constexpr bool VXR = MM;      // fp32 (matrix cores): V_x and the costate carried in registers through the stage; fp64: read from their LDS images where used
    T* ldsMl = lds + Lay::LDS_M;              // !MM: [LIVE][NX] live column r of [A B] as row r   (LDS_M region: NXU*NX words)
    T* ldsYZ = MM ? lds + Lay::LDS_M : yz64;  // [16 lanes][ZC] rows Y(Z, column of the lane)
    const bool has_v = lane < NX;                       // V-role: lanes < LX live state ZC + lane, lanes LX.. constant state lane - LX
    const bool st_zc = lane >= LX && lane < NX;
    const int sv = has_v ? (lane < LX ? ZC + lane : lane - LX) : 0;       // natural state index of the V-role
    const int zi = st_zc ? lane - LX : 0;
    const bool ctl = lane >= LX && lane < LIVE;         // M-role: control column lane - LX
    T Vx[NX], lam[NX], vcol[NX], xk[NX], uk[NU];
    bool ok = true;
    T gl_max = T(0);
    dV1 = T(0); dV2 = T(0); dmin = T(0);
    {
      const T* xN = xbp(cur) + N * NX;
#pragma unroll
      for (int i = 0; i < NX; ++i) xk[i] = xN[i];
      M::final_grad(tk(N), xk, e, c, Vx);
      T ox[NX], oe[NP];
#pragma unroll
      for (int i = 0; i < NX; ++i) { ox[i] = (has_v && sv == i) ? T(1) : T(0); lam[i] = Vx[i]; }
#pragma unroll
      for (int i = 0; i < NP; ++i) oe[i] = T(0);
      M::final_hess_mul(tk(N), xk, e, c, ox, oe, vcol);      // lanes >= NX: ox = 0  ->  vcol = 0
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { ldsVx[i] = Vx[i]; ldsLam[i] = lam[i]; }
        if (live) {
#pragma unroll
          for (int i = 0; i < NX; ++i) lam_out[N * NX + i] = lam[i];
        }
      }
      if (!MM && has_v) {
#pragma unroll
        for (int i = 0; i < NX; ++i) ldsV[sv * NX + i] = vcol[i];
      }
    }
    __syncthreads();
    T m[NX], mq = T(0), qzl = T(0);
    sc_load_stage(cur, N - 1, m, mq, qzl, xk, uk);
#if defined(LFSD_BW_CLOCK) && !defined(LFSD_EMU)
    long long bwc_t = clock64();
#endif
    for (int k = N - 1; k >= 0; --k) {
      {
        // every global load of the stage lands HERE.  Loads and stores share one in-order counter on this hardware: the stage's
        // nominal control, first used in the middle of the stage, was waited for behind the global stores of the gains -- i.e.
        // the wavefront sat out the stores' round trip to the L2 once per stage (`s_waitcnt vmcnt(0)` in the ISA)
#pragma unroll
        for (int a = 0; a < NU; ++a) pin(uk[a]);
#pragma unroll
        for (int i = 0; i < NX; ++i) pin(xk[i]);
        pin(mq); pin(qzl);
      }
      // Y(V-lane r, live column of this lane) = sum_kk V[kk][state of lane r] M[kk][column]
      typename std::conditional<MM, f32x16, T[16]>::type acc;
      T yn[NX];                                    // Y(:, column of this lane) in NATURAL row order
      if constexpr (!MM) {
        if (lane < LIVE) {
#pragma unroll
          for (int i = 0; i < NX; ++i) ldsMl[lane * NX + i] = m[i];
        }
        LFSD_STAGE_SYNC();                           // V_xx (published at the end of the previous stage) and the columns are visible
        T rowb[2][NX];
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) rowb[0][kk] = ldsV[kk];
#pragma unroll
        for (int r = 0; r < NX + LIVE; ++r) {
          const T* nxt = (r + 1 < NX) ? ldsV + (r + 1) * NX : ldsMl + (r + 1 - NX) * NX;
          if (r + 1 < NX + LIVE) {
#pragma unroll
            for (int kk = 0; kk < NX; ++kk) rowb[(r + 1) & 1][kk] = nxt[kk];
          }
          T sacc = T(0);
          if (r < NX) {
#pragma unroll
            for (int kk = 0; kk < NX; ++kk) sacc += rowb[r & 1][kk] * m[kk];
          } else {
#pragma unroll
            for (int kk = 0; kk < NX; ++kk) sacc += rowb[r & 1][kk] * yn[kk];
          }
          pin(sacc);
          if (r < NX) yn[r] = sacc; else acc[r - NX] = sacc;
          LFSD_ROW_FENCE();
        }
      }
      if constexpr (MM) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int kk = 0; kk < NX; ++kk) mfma4b(vcol[kk], m[kk], acc);
      }
      // stage-Hessian column(s) of this lane on the vector pipe, beside the products in flight on the matrix pipe
      T hx[NX], hu[NU];
      {
        T ox[NX], ou[NU], ls[NX];          // one-hots of BOTH roles: the V-role's state and the M-role's control
        const T HL = (mode == 1) ? T(1) : T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) { ox[i] = (has_v && sv == i) ? T(1) : T(0); ls[i] = HL * (VXR ? lam[i] : ldsLam[i]); }
#pragma unroll
        for (int a = 0; a < NU; ++a) ou[a] = (lane == LX + a) ? T(1) : T(0);
        M::ham_hess_mul(tk(k), xk, uk, ls, e, c, ox, ou, hx, hu);
      }
      if constexpr (MM) {
        tile_transpose(acc);
        LFSD_BWC(0)                                // loads issued, MFMA #1, stage Hessian column, transpose
#pragma unroll
        for (int s_ = 0; s_ < NX; ++s_) yn[s_] = acc[s_ < ZC ? LX + s_ : s_ - ZC];
      }
      if (lane < LIVE) {
#pragma unroll
        for (int z = 0; z < ZC; ++z) ldsYZ[lane * ZC + z] = yn[z];
      }
      // Q(live, live) = M_live^T Y
      if constexpr (MM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) mfma4b(m[kk], yn[kk], acc);
      }
      T Qg = mq, gl = mq;
      if constexpr (VXR) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { Qg += m[i] * Vx[i]; gl += m[i] * lam[i]; }
      } else {                                     // fp64: V_x and the costate stay in LDS (52 registers less through the stage)
        T vxl[NX], lml[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { vxl[i] = ldsVx[i]; lml[i] = ldsLam[i]; }
#pragma unroll
        for (int i = 0; i < NX; ++i) { Qg += m[i] * vxl[i]; gl += m[i] * lml[i]; }
      }
      if constexpr (MM) tile_transpose(acc);
      LFSD_STAGE_SYNC();                             // ldsYZ visible
      LFSD_BWC(1)                                  // MFMA #2, gradient dot products, transpose
      // column of Q (natural row order) of this lane's V-role, control rows of its M-role
      T Qcol[NX], mu_rows[NU], Quxj[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) mu_rows[a] = acc[LX + a] + dgrid * hu[a];
      T gzv[LIVE];                                 // row zi of Y, every live column: issued back to back for ALL lanes (pin)
#pragma unroll
      for (int j = 0; j < LIVE; ++j) gzv[j] = ldsYZ[j * ZC + zi];
      // (Vx and the costate of the constant state: from their LDS images -- still the incoming values here; a per-lane
      //  select over the register copies makes the compiler park both arrays in scratch)
      T Vx_z = ldsVx[zi], lam_z = ldsLam[zi];
#pragma unroll
      for (int j = 0; j < LIVE; ++j) pin(gzv[j]);
      pin(Vx_z); pin(lam_z);
#pragma unroll
      for (int j = 0; j < LX; ++j)                 // Q(live state ZC + j, constant state zi) = Y(zi, that column)
        Qcol[ZC + j] = st_zc ? gzv[j] : acc[j] + dgrid * hx[ZC + j];
#pragma unroll
      for (int z = 0; z < ZC; ++z) Qcol[z] = st_zc ? vcol[z] + dgrid * hx[z] : yn[z];      // Q(Z, Z) = V(Z, Z) + H_ZZ ; Q(Z, live) = Y(Z, live)
#pragma unroll
      for (int a = 0; a < NU; ++a) Quxj[a] = st_zc ? gzv[LX + a] : mu_rows[a];
      const T Qg_v = st_zc ? qzl + Vx_z : Qg;
      const T gl_v = st_zc ? qzl + lam_z : gl;
      if (has_v) {
#pragma unroll
        for (int a = 0; a < NU; ++a) ldsQux[sv * NU + a] = Quxj[a];
      }
      if (ctl) {
        const int b = lane - LX;
#pragma unroll
        for (int a = 0; a < NU; ++a) ldsQuu[b * NU + a] = mu_rows[a];
        ldsQu[b] = Qg;
        gl_max = t_max(gl_max, t_abs(gl));
      }
      LFSD_STAGE_SYNC();
      LFSD_BWC(2)                                  // gather of the constant states' columns, Q columns, Q_ux / Q_uu to LDS
      T Quu0[NU * NU], Lc[NU * NU], Qu[NU], kff[NU], Kj[NU], t1[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        Qu[a] = ldsQu[a];
#pragma unroll
        for (int b = 0; b < NU; ++b) Quu0[a * NU + b] = T(0.5) * (ldsQuu[b * NU + a] + ldsQuu[a * NU + b]);
      }
#pragma unroll
      for (int i = 0; i < NU * NU; ++i) Lc[i] = Quu0[i];
#pragma unroll
      for (int a = 0; a < NU; ++a) Lc[a * NU + a] += mu;
      if (ok) ok = chol_factor<NU>(Lc, dmin); else { T dd = T(0); chol_factor<NU>(Lc, dd); }
      {      // (the value recursion continues with the SHIFTED Q_uu: cpdp_common.h, "Levenberg shift of the Newton modes")
#pragma unroll
        for (int a = 0; a < NU; ++a) Quu0[a * NU + a] += mu;
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) { kff[a] = -Qu[a]; Kj[a] = -Quxj[a]; }
      chol_solve<NU>(Lc, kff);
      chol_solve<NU>(Lc, Kj);
      T qk[NU];
      matvec<NU>(Quu0, kff, qk);
#pragma unroll
      for (int a = 0; a < NU; ++a) { dV1 += kff[a] * Qu[a]; dV2 += T(0.5) * kff[a] * qk[a]; }
      T Vxj = Qg_v;
#pragma unroll
      for (int a = 0; a < NU; ++a) { Vxj += Kj[a] * (qk[a] + Qu[a]); Vxj += Quxj[a] * kff[a]; }
      LFSD_BWC(3)                                  // Q_uu from LDS, Cholesky, gains
      if (has_v) {
#pragma unroll
        for (int a = 0; a < NU; ++a) ldsK[sv * NU + a] = Kj[a];
        ldsVx[sv] = Vxj;
        ldsLam[sv] = gl_v;
        if (live) {
          T* Kout = Kws + ((long long)k * NX + sv) * NU;
#pragma unroll
          for (int a = 0; a < NU; ++a) Kout[a] = Kj[a];
        }
      }
      if (lane == 0 && live) {
#pragma unroll
        for (int a = 0; a < NU; ++a) kws[k * NU + a] = kff[a];
      }
      LFSD_STAGE_SYNC();
      matvec<NU>(Quu0, Kj, t1);
#pragma unroll
      for (int a = 0; a < NU; ++a) t1[a] += Quxj[a];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        T sacc = Qcol[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) { sacc += ldsK[i * NU + a] * t1[a]; sacc += ldsQux[i * NU + a] * Kj[a]; }      // (two FMAs; one statement compiles to mul + fma + add)
        vcol[i] = sacc;
        if (VXR) { Vx[i] = ldsVx[i]; lam[i] = ldsLam[i]; }
      }
      if (lane == 0 && live) {
#pragma unroll
        for (int i = 0; i < NX; ++i) lam_out[k * NX + i] = VXR ? lam[i] : ldsLam[i];
      }
      LFSD_BWC(4)                                  // gains to LDS / HBM, V_xx update
      // symmetrise V_xx through LDS (the rank-1 feeds rely on row i == column i)
      if (has_v) {
#pragma unroll
        for (int i = 0; i < NX; ++i) ldsV[sv * NX + i] = vcol[i];
      }
      LFSD_STAGE_SYNC();
      {
        T vt[NX];                                  // row sv of V_xx: thirteen loads in flight together (pin), then the select
#pragma unroll
        for (int i = 0; i < NX; ++i) vt[i] = ldsV[i * NX + sv];
#pragma unroll
        for (int i = 0; i < NX; ++i) pin(vt[i]);
#pragma unroll
        for (int i = 0; i < NX; ++i) vcol[i] = has_v ? T(0.5) * (vcol[i] + vt[i]) : T(0);
      }
      LFSD_STAGE_SYNC();
      if (!MM && has_v) {                          // publish the symmetric V_xx for the next stage's products
#pragma unroll
        for (int i = 0; i < NX; ++i) ldsV[sv * NX + i] = vcol[i];
      }
      LFSD_BWC(5)                                  // symmetrisation
      if (k > 0) sc_load_stage(cur, k - 1, m, mq, qzl, xk, uk);
    }
    ldsRed[lane] = gl_max;
    __syncthreads();
    gnorm = T(0);
#pragma unroll
    for (int a = 0; a < NU; ++a) gnorm = t_max(gnorm, ldsRed[LX + a]);
    __syncthreads();
    if (!t_finite(gnorm) || !t_finite(dV1) || !t_finite(dV2)) ok = false;
#if defined(LFSD_BW_CLOCK) && !defined(LFSD_EMU)
    if (threadIdx.x == 0 && blockIdx.x == 0)
      printf("bw clock (wave 0, one sweep of %d stages): mfma1+hess %lld  mfma2+dots %lld  gather+Qcol %lld  chol+gains %lld  K+Vupdate %lld  symm %lld\n",
             N, bwc[0], bwc[1], bwc[2], bwc[3], bwc[4], bwc[5]);
#endif
    return ok;
  }
  // costates and gradient norm without the value recursion (see costate_sweep_mf), structural layout
  LFSD_DEV void costate_sweep_sc(int cur, bool live, T& gnorm) {
    T* ldsLam = lds + Lay::LDS_LAM;  T* ldsRed = lds + Lay::LDS_RED;
    const bool has_v = lane < NX, st_zc = lane >= LX && lane < NX, ctl = lane >= LX && lane < LIVE;
    const int sv = has_v ? (lane < LX ? ZC + lane : lane - LX) : 0;
    const int zi = st_zc ? lane - LX : 0;
    T lam[NX], xk[NX], uk[NU];
    {
      const T* xN = xbp(cur) + N * NX;
#pragma unroll
      for (int i = 0; i < NX; ++i) xk[i] = xN[i];
      M::final_grad(tk(N), xk, e, c, lam);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) ldsLam[i] = lam[i];
        if (live) {
#pragma unroll
          for (int i = 0; i < NX; ++i) lam_out[N * NX + i] = lam[i];
        }
      }
    }
    __syncthreads();
    T gl_max = T(0);
    T m[NX], mq = T(0), qzl = T(0), mN[NX], mqN = T(0), qzN = T(0);
    sc_load_stage(cur, N - 1, m, mq, qzl, xk, uk);
    for (int k = N - 1; k >= 0; --k) {
      if (k > 0) { sc_load_stage(cur, k - 1, mN, mqN, qzN, xk, uk); LFSD_ISSUE_FENCE(); }
      T gl = mq;
#pragma unroll
      for (int i = 0; i < NX; ++i) gl += m[i] * lam[i];
      if (ctl) gl_max = t_max(gl_max, t_abs(gl));
      T lam_z = ldsLam[zi];                        // (incoming costate of the constant state, from its LDS image)
      pin(lam_z);
      const T glz = qzl + lam_z;
      __syncthreads();
      if (has_v) ldsLam[sv] = st_zc ? glz : gl;
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NX; ++i) lam[i] = ldsLam[i];
      if (lane == 0 && live) {
#pragma unroll
        for (int i = 0; i < NX; ++i) lam_out[k * NX + i] = lam[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NX; ++i) m[i] = mN[i];
      mq = mqN; qzl = qzN;
    }
    ldsRed[lane] = gl_max;
    __syncthreads();
    gnorm = T(0);
#pragma unroll
    for (int a = 0; a < NU; ++a) gnorm = t_max(gnorm, ldsRed[LX + a]);
    __syncthreads();
  }

  // Lane l tries step length 2^-l (all candidate roll-outs run concurrently in the group).
  // Returns the index of the largest accepted step (or -1) and the best cost seen.
  LFSD_DEV int linesearch(int cur, T J, T dV1, T dV2, T& alpha_out, T& Jmin, bool& flat_full) {
    T* ldsRed = lds + Lay::LDS_RED;
    T alpha = T(0);
    if (lane < NALPHA) { alpha = T(1); for (int i = 0; i < lane; ++i) alpha *= T(0.5); }
    T x[NX], u[NU], Ja = T(0), dummy = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = x0[i];
    for (int k = 0; k < N; ++k) {
      control(cur, k, x, alpha, true, u);
      T q = T(0);
      const T t = tk(k);
      for (int s = 0; s < S; ++s) rk4_step<false>(t, x, q, u, x, dummy, u);
      Ja += q;
    }
    Ja += M::final_cost(tk(N), x, e, c);
    ldsRed[lane] = Ja;
    __syncthreads();
    int ia = -1;
    Jmin = J;
    T a = T(1);
    alpha_out = T(0);
    const T flat = T(8) * Eps<T>::v() * t_abs(J);
    flat_full = t_finite(ldsRed[0]) && t_abs(ldsRed[0] - J) <= T(64) * Eps<T>::v() * t_abs(J);
    for (int l = 0; l < NALPHA; ++l) {
      const T Jl = ldsRed[l];
      const T expected = -(a * dV1 + a * a * dV2);
      const bool okl = t_finite(Jl) && ((J - Jl) >= T(1e-4) * expected - flat) && (Jl < J);
      if (okl && ia < 0) { ia = l; alpha_out = a; }
      if (t_finite(Jl)) Jmin = t_min(Jmin, Jl);
      a *= T(0.5);
    }
    __syncthreads();
    return ia;
  }
};

// The same steps for the WIDE mapping -- one trajectory per wavefront (OcSolver<M, T, 64, EXACT>), for batches that leave most
// of the machine idle under the lock-step mapping (robot arm / rocket at 1024 trajectories per GPU: 256-512 wavefronts
// for 1024 SIMDs, and an iteration as long as N sequential intervals).  What depends only on the nominal of ONE interval
// is done for all intervals at once, spread over the 64 lanes:
//   rollout_alphas      the closed-loop roll-outs of 16 step lengths, one per lane (the line search IS the roll-out)
//   linearise_parallel  [A_k B_k; q_k] for every interval k: lane <- (k, column pair), N*ceil(NXU/2)/64 rounds
//   costate_sweep       lambda_k = q_x + A_k^T lambda_k+1 (sequential, NX FMAs per interval)
//   hessians_parallel   exact stage Hessian columns for every (k, column): second-order adjoint sweeps, N*NXU/64 rounds
// which leaves only the cheap Riccati-type recursion of OcSolver::backward sequential in k.
// W > 1 (the re-launched tail of a wide launch, small batches): W wavefronts per trajectory in one workgroup.  Each wavefront keeps
// its own LDS region and runs the SEQUENTIAL phases (roll-outs, backward sweep, step control) redundantly on identical inputs --
// identical values, identical control flow, the same words of the workspace written with the same bits --; the interval-parallel
// phases (linearisation, exact stage Hessians) are split: wavefront w takes the items w*64 + lane, + 64 W, ... and a workgroup
// barrier publishes the results.  The items are computed by the same code whoever runs them: results do not depend on W.
template <class M, typename T, bool EXACT, bool BND = false, int W = 1> struct OcWide : OcSolver<M, T, 64, EXACT, BND> {
  using Base = OcSolver<M, T, 64, EXACT, BND>;
  using Lay = OcLayout<M>;
  static constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NXU = NX + NU;
  static constexpr int NAL = 16;                       // step lengths 2^0 .. 2^-15
  static constexpr int NW = W, PSTRIDE = 64 * W;       // wavefronts per trajectory; stride of the item loops of the parallel phases
  int wave = 0;                                        // this wavefront's index in the workgroup
  LFSD_DEV int pitem() const { return W == 1 ? lane : wave * 64 + lane; }
  using Base::lane; using Base::N; using Base::S; using Base::e; using Base::c; using Base::x0; using Base::xb; using Base::ub; using Base::xbp; using Base::ubp; using Base::Mwp;
  using Base::Mws; using Base::Hws; using Base::lds; using Base::xa; using Base::ua; using Base::lam_out; using Base::DT;

  // lane l < NAL rolls the closed loop out with step length 2^-l and parks states / controls at [k][component][l]
  // (the operands of the control law -- nominal, feed-forward, gains: group-uniform global loads -- are fetched one interval ahead:
  //  nothing else runs on the SIMD to hide their ~1 500 clocks behind, and there are n_grid of them in a row)
  LFSD_DEV T rollout_alphas(int cur, bool gains, T& alpha) {
    alpha = T(0);
    if (lane < NAL) { alpha = T(1); for (int i = 0; i < lane; ++i) alpha *= T(0.5); }
    T x[NX], u[NU], Ja = T(0), dummy = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) x[i] = x0[i];
    T ubC[NU], ubN[NU], xbC[NX], xbN[NX], KC[NX * NU], KN[NX * NU], kC[NU], kN[NU];
    auto load_ctl = [&](int k_, T* ub_, T* xb_, T* K_, T* k_ff) LFSD_LAMBDA_INLINE {
      const T* ubk = ubp(cur) + k_ * NU;
#pragma unroll
      for (int a = 0; a < NU; ++a) ub_[a] = ubk[a];
      if (gains) {
        const T* xbk = xbp(cur) + k_ * NX;
        const T* Kk = this->Kws + (long long)k_ * NX * NU;
        const T* kk = this->kws + k_ * NU;
#pragma unroll
        for (int i = 0; i < NX; ++i) xb_[i] = xbk[i];
#pragma unroll
        for (int i = 0; i < NX * NU; ++i) K_[i] = Kk[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) k_ff[a] = kk[a];
      }
    };
    load_ctl(0, ubN, xbN, KN, kN);
    for (int k = 0; k < N; ++k) {
#pragma unroll
      for (int a = 0; a < NU; ++a) { ubC[a] = ubN[a]; kC[a] = kN[a]; }
#pragma unroll
      for (int i = 0; i < NX; ++i) xbC[i] = xbN[i];
#pragma unroll
      for (int i = 0; i < NX * NU; ++i) KC[i] = KN[i];
      if (k + 1 < N) { load_ctl(k + 1, ubN, xbN, KN, kN); LFSD_ISSUE_FENCE(); }
      // closed-loop control  u = ubar + alpha*kff + K (x - xbar)  (OcSolver::control, on the operands fetched above)
#pragma unroll
      for (int a = 0; a < NU; ++a) u[a] = ubC[a];
      if (gains) {
#pragma unroll
        for (int a = 0; a < NU; ++a) u[a] += alpha * kC[a];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const T dx = x[i] - xbC[i];
#pragma unroll
          for (int a = 0; a < NU; ++a) u[a] += KC[i * NU + a] * dx;
        }
      }
      if (BND) {
#pragma unroll
        for (int a = 0; a < NU; ++a) u[a] = t_min(t_max(u[a], this->ulb[a]), this->uub[a]);
      }
      if (lane < NAL) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xa[(k * NX + i) * NAL + lane] = x[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) ua[(k * NU + a) * NAL + lane] = u[a];
      }
      T q = T(0);
      const T t = this->tk(k);
      for (int s = 0; s < S; ++s) this->template rk4_step<false>(t, x, q, u, x, dummy, u);
      Ja += q;
      if (BND) Ja += this->node_pen(k + 1, x);          // state bounds on node k+1 (augmented-Lagrangian term)
    }
    if (lane < NAL) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xa[(N * NX + i) * NAL + lane] = x[i];
    }
    Ja += M::final_cost(this->tk(N), x, e, c);
    return Ja;
  }
  // the roll-out of step length index ia becomes nominal `nxt`
  LFSD_DEV void adopt_alpha(int ia, int nxt) {
    __syncthreads();
    for (int i = lane; i < (N + 1) * NX; i += 64) xbp(nxt)[i] = xa[i * NAL + ia];
    for (int i = lane; i < N * NU; i += 64) ubp(nxt)[i] = ua[i * NAL + ia];
    __syncthreads();
  }
  // linearise the shooting map along nominal `nxt`, all intervals at once
  LFSD_DEV void linearise_parallel(int nxt) {
    if constexpr (sizeof(T) == 4 && Lay::HALL) {
      for (int k = pitem(); k < N; k += PSTRIDE) {
        T x[NX], u[NU], q;
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = xbp(nxt)[k * NX + i];
#pragma unroll
        for (int a = 0; a < NU; ++a) u[a] = ubp(nxt)[k * NU + a];
        this->template interval_sens_all<Lay::NVH>(this->tk(k), x, q, u, Mwp(nxt) + (long long)k * Lay::M_ELEMS);
      }
      __syncthreads();
      return;
    }
    if constexpr (sizeof(T) == 4) {
      using V = pk2<T>;
      constexpr int NCT = (NXU + 1) / 2;
      for (int t = pitem(); t < N * NCT; t += PSTRIDE) {
        const int k = t / NCT, c0 = 2 * (t % NCT), c1 = c0 + 1;
        T x[NX], u[NU], q = T(0);
        V m[NX], du[NU], mq = V(T(0));
#pragma unroll
        for (int i = 0; i < NX; ++i) { x[i] = xbp(nxt)[k * NX + i]; m[i] = mk2<T>((c0 == i) ? T(1) : T(0), (c1 == i) ? T(1) : T(0)); }
#pragma unroll
        for (int a = 0; a < NU; ++a) { u[a] = ubp(nxt)[k * NU + a]; du[a] = mk2<T>((c0 == NX + a) ? T(1) : T(0), (c1 == NX + a) ? T(1) : T(0)); }
        const T tt = this->tk(k);
        for (int s = 0; s < S; ++s) this->template rk4_step<true, V>(tt, x, q, u, m, mq, du);
        V* Mk = reinterpret_cast<V*>(Mwp(nxt) + (long long)k * Lay::M_ELEMS + c0);
#pragma unroll
        for (int i = 0; i < NX; ++i) Mk[i * (Lay::NXUP / 2)] = m[i];
        Mk[NX * (Lay::NXUP / 2)] = mq;
      }
    } else {
      for (int t = pitem(); t < N * NXU; t += PSTRIDE) {
        const int k = t / NXU, col = t % NXU;
        T x[NX], u[NU], m[NX], du[NU], q = T(0), mq = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) { x[i] = xbp(nxt)[k * NX + i]; m[i] = (col == i) ? T(1) : T(0); }
#pragma unroll
        for (int a = 0; a < NU; ++a) { u[a] = ubp(nxt)[k * NU + a]; du[a] = (col == NX + a) ? T(1) : T(0); }
        const T tt = this->tk(k);
        for (int s = 0; s < S; ++s) this->template rk4_step<true>(tt, x, q, u, m, mq, du);
        T* Mk = Mwp(nxt) + (long long)k * Lay::M_ELEMS + col;
#pragma unroll
        for (int i = 0; i < NX; ++i) Mk[i * Lay::NXUP] = m[i];
        Mk[NX * Lay::NXUP] = mq;
      }
    }
    __syncthreads();
  }
  // exact discrete costates (== IPOPT's lam_g) of nominal `cur` into lam_out; returns max |dJ/du|
  LFSD_DEV T costate_sweep(int cur) {
    T* ldsLam = lds + Lay::LDS_LAM;  T* ldsRed = lds + Lay::LDS_RED;
    T lam[NX], xN[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) xN[i] = xbp(cur)[N * NX + i];
    M::final_grad(this->tk(N), xN, e, c, lam);
    if (BND && this->xm != nullptr) {
#pragma unroll
      for (int i = 0; i < NX; ++i) { T g_, h_; this->node_pen_d(N, i, xN[i], this->xlb[i], this->xub[i], g_, h_); lam[i] += g_; }
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NX; ++i) lam_out[N * NX + i] = lam[i];
    }
    T gl_max = T(0);
    // this lane's column of [A_k B_k; q_k], loaded LFSD_CS_AHEAD stages ahead of its use: the recursion itself is NX FMAs and
    // an LDS exchange per stage, a global load is ~1 500 clocks and nothing else runs on the SIMD
    constexpr int AH = LFSD_CS_AHEAD;
    T mc[AH > 0 ? AH : 1][NX + 1];
    auto load_col = [&](int k_, T* dst) LFSD_LAMBDA_INLINE {
      const T* Mk = Mwp(cur) + (long long)(k_ < 0 ? 0 : k_) * Lay::M_ELEMS + (lane < NXU ? lane : 0);
#pragma unroll
      for (int i = 0; i <= NX; ++i) dst[i] = Mk[i * Lay::NXUP];
    };
    if (AH > 0) {
#pragma unroll
      for (int j = 0; j < AH; ++j) load_col(N - 1 - j, mc[j]);
    }
    for (int k = N - 1; k >= 0; --k) {
      T gl = T(0);
      T col[NX + 1];
      if (AH > 0) {
#pragma unroll
        for (int i = 0; i <= NX; ++i) col[i] = mc[0][i];
#pragma unroll
        for (int j = 0; j + 1 < AH; ++j) {
#pragma unroll
          for (int i = 0; i <= NX; ++i) mc[j][i] = mc[j + 1][i];
        }
        load_col(k - AH, mc[AH - 1]);
        LFSD_ISSUE_FENCE();
      } else {
        load_col(k, col);
      }
      if (lane < NXU) {
        gl = col[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) gl += col[i] * lam[i];
        if (BND && this->xm != nullptr && k > 0 && lane < NX) {      // + the state-bound term of node k
          T xi = T(0), lbi = T(0), ubi = T(0), g_, h_;
#pragma unroll
          for (int i = 0; i < NX; ++i) { if (lane == i) { xi = xbp(cur)[k * NX + i]; lbi = this->xlb[i]; ubi = this->xub[i]; } }
          this->node_pen_d(k, lane, xi, lbi, ubi, g_, h_);
          gl += g_;
        }
        if (lane < NX) { ldsLam[lane] = gl; lam_out[k * NX + lane] = gl; }
        else gl_max = t_max(gl_max, t_abs(gl));
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NX; ++i) lam[i] = ldsLam[i];
      __syncthreads();
    }
    ldsRed[lane] = gl_max;
    __syncthreads();
    T g = T(0);
#pragma unroll
    for (int a = 0; a < NU; ++a) g = t_max(g, ldsRed[NX + a]);
    __syncthreads();
    return g;
  }
  // exact stage Hessians of nominal `cur` (costates must be on lam_out), every (interval, column) at once
  LFSD_DEV void hessians_parallel(int cur) {
    if constexpr (sizeof(T) == 4 && Lay::HALL) {
      for (int k = pitem(); k < N; k += PSTRIDE) {
        T xk[NX], uk[NU], ln[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xk[i] = xbp(cur)[k * NX + i]; ln[i] = lam_out[(k + 1) * NX + i]; }
#pragma unroll
        for (int a = 0; a < NU; ++a) uk[a] = ubp(cur)[k * NU + a];
        this->template stage_hessian_all<Lay::NVH>(k, xk, uk, ln, Hws + (long long)k * Lay::H_ELEMS);
      }
      __syncthreads();
      return;
    }
    for (int t = pitem(); t < N * NXU; t += PSTRIDE) {
      const int k = t / NXU, col = t % NXU;
      T xk[NX], uk[NU], ln[NX], hx[NX], hu[NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) { xk[i] = xbp(cur)[k * NX + i]; ln[i] = lam_out[(k + 1) * NX + i]; }
#pragma unroll
      for (int a = 0; a < NU; ++a) uk[a] = ubp(cur)[k * NU + a];
      this->template stage_hessian_col<true>(k, xk, uk, ln, hx, hu, col);
      T* hcol = Hws + (long long)k * Lay::H_ELEMS + col;
#pragma unroll
      for (int i = 0; i < NX; ++i) hcol[i * Lay::NXUP] = hx[i];
#pragma unroll
      for (int a = 0; a < NU; ++a) hcol[(NX + a) * Lay::NXUP] = hu[a];
    }
    __syncthreads();
  }

  // ---- the backward sweep of the small models (fp32; robot arm 4 x 6, cart-pole 4 x 5, pendulum 2 x 3) ------------------------
  // OcSolver::backward keeps one column per lane and hands rows over through LDS five times per stage; with six columns on a
  // wavefront that is alone on its SIMD a stage is ~2 900 clocks, most of them waiting (LDS round trips, the model call of the
  // stage Hessian in modes 0 / 1, global operands).  Here
  //   * all operands of all stages ([A_k B_k; q_k], gaps, cached exact Hessians) are copied into the idle LDS region of the
  //     Hessian sweeps first, by all 64 lanes, eight loads in flight per lane (as ms_forward does);
  //   * the stage Hessians of modes 0 / 1 -- one model call per stage in the generic sweep, in sequence -- are computed for all
  //     intervals at once, one interval per lane, all columns side by side (ham_hess_mul_n); mode 1 needs the costates of the sweep:
  //     a costate recursion on the staged operands runs first (NX^2 FMAs per stage);
  //   * the recursion keeps column j of [A B; q], of the stage Hessian and of Q on lane j and V_xx, V_x, the costate uniform on every
  //     lane; what a lane needs of another lane's column comes through v_readlane (lane_get): no LDS hand-over, no barrier.  410
  //     instructions per stage.  (First version: the whole recursion on EVERY lane with the dense products packed two columns per
  //     instruction -- no exchange at all, 525 instructions per stage: oc_solve of the bench's five steps 11.05 ms against 10.38.)
  // Same recursion, same outputs (gains, feed-forward, costates, predicted decrease, failing pivot) as OcSolver::backward.
  static constexpr bool SMALL_BW = EXACT && !BND && sizeof(T) == 4 && Lay::HALL && NX * NXU <= 32 && (LFSD_BW_SMALL) != 0;
  static constexpr int BWS_STG = Lay::M_ELEMS + Lay::H_ELEMS + 2 * NX;      // words per stage
  LFSD_DEV bool small_bw_fits() const { return SMALL_BW && (N + 1) * BWS_STG <= Lay::template lds_ex_size<64, (int)sizeof(T)>(); }
  LFSD_DEV bool backward_small(int cur, int mode, T mu, T& gnorm, T& dV1, T& dV2, T& dmin) {
    using V = pk2<T>;
    constexpr int NP2 = Lay::NXUP / 2, RS = Lay::NXUP, ME = Lay::M_ELEMS, HE = Lay::H_ELEMS, NVH = Lay::NVH;
    T* st = lds + Lay::template lds_ex<64>();
    T* sM = st;                           // [N][M_ELEMS]   as in the workspace
    T* sH = sM + N * ME;                  // [N][H_ELEMS]   stage Hessians [row][column]
    T* sd = sH + N * HE;                  // [N][NX]        gaps (zero for a roll-out)
    T* sl = sd + N * NX;                  // [N + 1][NX]    costates (mode 1)
    auto copy = [&](T* dst, const T* src, int n) LFSD_LAMBDA_INLINE {
      for (int b = lane; b < n; b += 64 * 8) {
        T v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (src != nullptr && b + 64 * j < n) ? src[b + 64 * j] : T(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) { if (b + 64 * j < n) dst[b + 64 * j] = v[j]; }
      }
    };
    copy(sM, Mwp(cur), N * ME);
    copy(sd, this->gap, N * NX);
    if (mode == 2) copy(sH, Hws, N * HE);
    T Vx[NX], lam[NX], xN[NX], Vxx[NX][NX];
    {
      const T* xp = xbp(cur) + N * NX;
#pragma unroll
      for (int i = 0; i < NX; ++i) xN[i] = xp[i];
      M::final_grad(this->tk(N), xN, e, c, Vx);
      T oe[NP > 0 ? NP : 1];
#pragma unroll
      for (int i = 0; i < NP; ++i) oe[i] = T(0);
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        T ox[NX], hc[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) ox[i] = (i == j) ? T(1) : T(0);
        M::final_hess_mul(this->tk(N), xN, e, c, ox, oe, hc);
#pragma unroll
        for (int i = 0; i < NX; ++i) Vxx[i][j] = hc[i];
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) lam[i] = Vx[i];
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) lam_out[N * NX + i] = lam[i];
      }
    }
    __syncthreads();
    if (mode != 2) {
      if (mode == 1) {
        // the costates of the sweep (the same recursion as in the loop below, on the staged rows)
        T l[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) l[i] = lam[i];
        for (int k = N - 1; k >= 0; --k) {
          const T* mk = sM + k * ME;
          if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NX; ++i) sl[(k + 1) * NX + i] = l[i];
          }
          T ln[NX];
#pragma unroll
          for (int j = 0; j < NX; ++j) {
            T s_ = mk[NX * RS + j];
#pragma unroll
            for (int i = 0; i < NX; ++i) s_ += mk[i * RS + j] * l[i];
            ln[j] = s_;
          }
#pragma unroll
          for (int i = 0; i < NX; ++i) l[i] = ln[i];
        }
        __syncthreads();
      }
      T el[NP > 0 ? NP : 1], cl[M::NCX > 0 ? M::NCX : 1];
#pragma unroll
      for (int i = 0; i < NP; ++i) el[i] = e[i];
#pragma unroll
      for (int i = 0; i < M::NCX; ++i) cl[i] = c[i];
      for (int k = lane; k < N; k += 64) {
        T xk[NX], uk[NU], ls[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) { xk[i] = xbp(cur)[k * NX + i]; ls[i] = (mode == 1) ? sl[(k + 1) * NX + i] : T(0); }
#pragma unroll
        for (int a = 0; a < NU; ++a) uk[a] = ubp(cur)[k * NU + a];
        V ox[NVH][NX], ou[NVH][NU], gx[NVH][NX], gu[NVH][NU];
#pragma unroll
        for (int v = 0; v < NVH; ++v) {
#pragma unroll
          for (int i = 0; i < NX; ++i) ox[v][i] = mk2<T>((2 * v == i) ? T(1) : T(0), (2 * v + 1 == i) ? T(1) : T(0));
#pragma unroll
          for (int a = 0; a < NU; ++a) ou[v][a] = mk2<T>((2 * v == NX + a) ? T(1) : T(0), (2 * v + 1 == NX + a) ? T(1) : T(0));
        }
        M::template ham_hess_mul_n<NVH>(this->tk(k), xk, uk, ls, el, cl, &ox[0][0], &ou[0][0], &gx[0][0], &gu[0][0]);
        V* hk = reinterpret_cast<V*>(sH + k * HE);
#pragma unroll
        for (int v = 0; v < NVH; ++v) {
#pragma unroll
          for (int i = 0; i < NX; ++i) hk[i * NP2 + v] = this->dgrid * gx[v][i];
#pragma unroll
          for (int a = 0; a < NU; ++a) hk[(NX + a) * NP2 + v] = this->dgrid * gu[v][a];
        }
      }
      __syncthreads();
    }
    bool ok = true;
    T gl_max = T(0), lmax = T(0);
    dV1 = T(0); dV2 = T(0); dmin = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) lmax = t_max(lmax, t_abs(lam[i]));
    const bool gaps = this->gap != nullptr;
    {
      // Column j of [A B; q], of the stage Hessian and of Q on lane j (< NXU); V_xx, V_x and the costate uniform on every lane; what a
      // lane needs of another lane's column comes through v_readlane (lane_get): no LDS hand-over, no barrier, and the dense products
      // cost one column's worth of instructions instead of all of them.
      constexpr int RS2 = RS;
      const int jc = lane < NXU ? lane : 0;
      const bool act = lane < NXU;
      for (int k = N - 1; k >= 0; --k) {
        const T* mk = sM + k * ME;
        const T* hk = sH + k * HE;
        const T* dk = sd + k * NX;
        T mj[NX], mqj, hj[NXU], Mu[NX][NXU];
        // (read unconditionally -- the index is safe on every lane -- and materialised before the select: no branch around a load)
#pragma unroll
        for (int i = 0; i < NX; ++i) mj[i] = mk[i * RS2 + jc];
        mqj = mk[NX * RS2 + jc];
#pragma unroll
        for (int r = 0; r < NXU; ++r) hj[r] = hk[r * RS2 + jc];
#pragma unroll
        for (int i = 0; i < NX; ++i) { pin(mj[i]); mj[i] = act ? mj[i] : T(0); }
        pin(mqj); mqj = act ? mqj : T(0);
#pragma unroll
        for (int r = 0; r < NXU; ++r) { pin(hj[r]); hj[r] = act ? hj[r] : T(0); }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
          for (int r = 0; r < NXU; ++r) Mu[i][r] = mk[i * RS2 + r];
        }
        if (gaps) {
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            T sacc = T(0);
#pragma unroll
            for (int j = 0; j < NX; ++j) sacc += Vxx[i][j] * dk[j];
            Vx[i] += sacc;
          }
        }
        T Y[NX], Qc[NXU], Qg = mqj, gl = mqj;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          T s_ = Vxx[i][0] * mj[0];
#pragma unroll
          for (int kk = 1; kk < NX; ++kk) s_ += Vxx[i][kk] * mj[kk];
          Y[i] = s_;
          Qg += mj[i] * Vx[i]; gl += mj[i] * lam[i];
        }
#pragma unroll
        for (int r = 0; r < NXU; ++r) {
          T s_ = hj[r];
#pragma unroll
          for (int i = 0; i < NX; ++i) s_ += Mu[i][r] * Y[i];
          Qc[r] = s_;
        }
        T Quu0[NU * NU], Lc[NU * NU], Qu[NU], kff[NU], Kj[NU], t1[NU], Qr[NU * NU];
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          Qu[a] = lane_get(Qg, NX + a);
          gl_max = t_max(gl_max, t_abs(lane_get(gl, NX + a)));
#pragma unroll
          for (int b = 0; b < NU; ++b) Qr[a * NU + b] = lane_get(Qc[NX + a], NX + b);      // Q[NX + a][NX + b]
        }
#pragma unroll
        for (int a = 0; a < NU; ++a) {
#pragma unroll
          for (int b = 0; b < NU; ++b) Quu0[a * NU + b] = T(0.5) * (Qr[a * NU + b] + Qr[b * NU + a]);
        }
        T mu_k = mu;
        if (mu > T(0) && this->mu_stage_frac > T(0)) {
          T Lt[NU * NU], dd = T(0);
#pragma unroll
          for (int i = 0; i < NU * NU; ++i) Lt[i] = Quu0[i];
#pragma unroll
          for (int a = 0; a < NU; ++a) Lt[a * NU + a] += mu * this->mu_stage_frac;
          if (chol_factor<NU>(Lt, dd)) mu_k = mu * this->mu_stage_frac;
        }
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Lc[i] = Quu0[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) Lc[a * NU + a] += mu_k;
        {      // (the value recursion continues with the SHIFTED Q_uu: cpdp_common.h, "Levenberg shift of the Newton modes")
#pragma unroll
          for (int a = 0; a < NU; ++a) Quu0[a * NU + a] += mu_k;
        }
        ok = chol_factor<NU>(Lc, dmin);
        if (!ok) break;
#pragma unroll
        for (int a = 0; a < NU; ++a) { kff[a] = -Qu[a]; Kj[a] = -Qc[NX + a]; }
        chol_solve<NU>(Lc, kff);
        chol_solve<NU>(Lc, Kj);
        T qk[NU];
        matvec<NU>(Quu0, kff, qk);
        matvec<NU>(Quu0, Kj, t1);
        T Vxj = Qg;
#pragma unroll
        for (int a = 0; a < NU; ++a) {
          dV1 += kff[a] * Qu[a]; dV2 += T(0.5) * kff[a] * qk[a];
          Vxj += Kj[a] * (qk[a] + Qu[a]); Vxj += Qc[NX + a] * kff[a];
          t1[a] += Qc[NX + a];
        }
        T Vn[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          T s_ = Qc[i];
#pragma unroll
          for (int a = 0; a < NU; ++a) { s_ += lane_get(Kj[a], i) * t1[a]; s_ += lane_get(Qc[NX + a], i) * Kj[a]; }
          Vn[i] = s_;
        }
        T U[NX][NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
          for (int j = 0; j < NX; ++j) U[i][j] = lane_get(Vn[i], j);      // row i of column j
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
#pragma unroll
          for (int j = 0; j < NX; ++j) Vxx[i][j] = T(0.5) * (U[i][j] + U[j][i]);
          Vx[i] = lane_get(Vxj, i);
          lam[i] = lane_get(gl, i);
          lmax = t_max(lmax, t_abs(lam[i]));
        }
        if (lane < NX) {
          T* Kout = this->Kws + ((long long)k * NX + lane) * NU;
#pragma unroll
          for (int a = 0; a < NU; ++a) Kout[a] = Kj[a];
          lam_out[k * NX + lane] = gl;
        }
        if (lane == 0) {
#pragma unroll
          for (int a = 0; a < NU; ++a) this->kws[k * NU + a] = kff[a];
        }
      }
    }
    this->lam_max = lmax;
    __syncthreads();                          // (gains / costates visible to every lane; the staging region is free again)
    gnorm = gl_max;
    if (!t_finite(gnorm) || !t_finite(dV1) || !t_finite(dV2)) ok = false;
    return ok;
  }

  // ---- the interval-parallel iteration: multiple shooting, the reference's own formulation (CPDP.py:136-172) ----------------
  // The reference hands IPOPT the LIFTED problem: every node state X_k is a variable, every interval an equality constraint
  // F(X_k, U_k) - X_k+1 = 0.  Given an iterate of that problem -- node states xb, controls ub, gaps d_k = F(x_k, u_k) - x_k+1 --
  // all N intervals integrate their RK4 steps and sensitivities CONCURRENTLY from the node states (ms_trial: the 64 lanes take
  // (interval, column pair) items, as linearise_parallel does), and the only sequential parts of an iteration are the
  // Riccati-type recursion with gap terms (OcSolver::backward with `gap` set) and a LINEAR forward pass (ms_forward):
  //     du_k = k_k + K_k dx_k,   dx_k+1 = A_k dx_k + B_k du_k + d_k,   dx_0 = 0
  // -- the Newton step of the lifted KKT system (Gauss-Newton multiple shooting / the step of an SQP method on the NLP above).
  // Where the single-shooting iteration of this kernel pays N x S sequentially dependent RK4 steps for its roll-outs, this one
  // pays S (per round of 64 items).  The full Newton step (it closes the linearised gaps entirely) or ONE half step is taken this way,
  // and only when the augmented Lagrangian  J + lambda^T d + rho/2 |d|^2  (costates of the iterate held fixed, rho = max(|lambda|_inf, 1);
  // oc_solve_wide_kernel, "Merit function") accepts it by an Armijo test; otherwise the step is the closed-loop nonlinear roll-out
  // around the node states, which closes every gap by construction.  Convergence is never declared on an iterate with gaps.
  T *gapb[2] = {nullptr, nullptr}, *dxw = nullptr, *duw = nullptr;      // [N][NX] gaps of the two nominal buffers; Newton step [N+1][NX], [N][NU]
  LFSD_DEV T* gapp(int i) const { return i ? gapb[1] : gapb[0]; }
  LFSD_DEV T wave_sum(T v) {
    T* ldsRed = lds + Lay::LDS_RED;
    ldsRed[lane] = v;
    __syncthreads();
    T r = T(0);
    for (int l = 0; l < 64; ++l) r += ldsRed[l];
    __syncthreads();
    return r;
  }
  LFSD_DEV T wave_max(T v) {
    T* ldsRed = lds + Lay::LDS_RED;
    ldsRed[lane] = v;
    __syncthreads();
    T r = ldsRed[0];
    for (int l = 1; l < 64; ++l) r = (ldsRed[l] > r || !t_finite(ldsRed[l])) ? ldsRed[l] : r;      // (a NaN / inf wins: the caller tests finiteness)
    __syncthreads();
    return r;
  }
  // The Newton step (for step length 1) of the lifted problem from the gains of the last backward sweep, into dxw / duw.
  // Returns its first-order change of the cost,  sum_k q_k^T (dx_k, du_k) + h_x^T dx_N.
  LFSD_DEV T ms_forward(int cur, T& lamd, T& lamabs) {
    // small models (robot arm 4 x 6, cart-pole 4 x 5, pendulum 2 x 3) whose whole recursion fits the idle LDS region of the Hessian
    // sweeps: ALL operands of all stages -- [A_k B_k; q_k], gains, feed-forward, gap, costate -- are copied there first, by all 64
    // lanes at once (one global round trip instead of one per stage: the recursion was waiting ~1 300 clocks per stage for operands
    // it had asked for one stage earlier); then every lane carries the whole dx and runs the recursion on group-uniform LDS reads,
    // without any exchange between lanes.
    constexpr int STG = Lay::M_ELEMS + NX * NU + NU + 2 * NX;      // words per stage
    if constexpr (EXACT && NX * NXU <= 32) {      // (EXACT: the instantiations that have that LDS region)
      if (N * STG <= Lay::template lds_ex_size<64, (int)sizeof(T)>()) {
        T* st = lds + Lay::template lds_ex<64>();
        const T* gp = this->gap;
        // five straight copies, eight loads per lane in flight (a copy loop that stores what it has just loaded waits one global
        // round trip per element and lane: 38 of them in a row cost as much as the recursion it was meant to feed)
        T* sM = st;                                   // [N][M_ELEMS]  as in the workspace
        T* sK = sM + N * Lay::M_ELEMS;                // [N][NX][NU]
        T* sk = sK + N * NX * NU;                     // [N][NU]
        T* sd = sk + N * NU;                          // [N][NX] gaps (zero for a roll-out)
        T* sl = sd + N * NX;                          // [N][NX] costates of nodes 1..N
        auto copy = [&](T* dst, const T* src, int n) LFSD_LAMBDA_INLINE {
          for (int b = lane; b < n; b += 64 * 8) {
            T v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (src != nullptr && b + 64 * j < n) ? src[b + 64 * j] : T(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) { if (b + 64 * j < n) dst[b + 64 * j] = v[j]; }
          }
        };
        copy(sM, Mwp(cur), N * Lay::M_ELEMS);
        copy(sK, this->Kws, N * NX * NU);
        copy(sk, this->kws, N * NU);
        copy(sd, gp, N * NX);
        copy(sl, lam_out + NX, N * NX);
        if (lane < NX) dxw[lane] = T(0);
        __syncthreads();
        T dxv[NX], dl = T(0), ld = T(0), la = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) dxv[i] = T(0);
        for (int k = 0; k < N; ++k) {
          const T* mk = sM + k * Lay::M_ELEMS;       // (rows of NXUP words)
          const T* Kk = sK + k * NX * NU;
          const T* kk = sk + k * NU;
          const T* dk = sd + k * NX;
          const T* lk = sl + k * NX;
          T duv[NU], nx[NX];
#pragma unroll
          for (int a = 0; a < NU; ++a) duv[a] = kk[a];
#pragma unroll
          for (int i = 0; i < NX; ++i) {
#pragma unroll
            for (int a = 0; a < NU; ++a) duv[a] += Kk[i * NU + a] * dxv[i];
          }
#pragma unroll
          for (int i = 0; i < NX; ++i) {
            T sacc = dk[i];
#pragma unroll
            for (int j = 0; j < NX; ++j) sacc += mk[i * Lay::NXUP + j] * dxv[j];
#pragma unroll
            for (int a = 0; a < NU; ++a) sacc += mk[i * Lay::NXUP + NX + a] * duv[a];
            nx[i] = sacc;
            const T lg = lk[i] * dk[i];
            ld += lg; la += t_abs(lg);
          }
#pragma unroll
          for (int j = 0; j < NX; ++j) dl += mk[NX * Lay::NXUP + j] * dxv[j];
#pragma unroll
          for (int a = 0; a < NU; ++a) dl += mk[NX * Lay::NXUP + NX + a] * duv[a];
          if (lane < NX) {
            T v = T(0);
#pragma unroll
            for (int i = 0; i < NX; ++i) { if (lane == i) v = nx[i]; }
            dxw[(k + 1) * NX + lane] = v;
          } else if (lane < NXU) {
            T v = T(0);
#pragma unroll
            for (int a = 0; a < NU; ++a) { if (lane == NX + a) v = duv[a]; }
            duw[k * NU + (lane - NX)] = v;
          }
#pragma unroll
          for (int i = 0; i < NX; ++i) dxv[i] = nx[i];
        }
        {
          T xN[NX], hx[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) xN[i] = xbp(cur)[N * NX + i];
          M::final_grad(this->tk(N), xN, e, c, hx);
#pragma unroll
          for (int i = 0; i < NX; ++i) dl += hx[i] * dxv[i];
        }
        __syncthreads();                               // (dxw / duw visible to every lane; the staging region is free again)
        lamd = ld; lamabs = la;
        return dl;
      }
    }
    T* ldsDx = lds + Lay::LDS_VX;                 // (NX words, free between backward sweeps)
    const T* Mc = Mwp(cur);
    const T* gp = this->gap;                       // (nullptr: buffer `cur` is a roll-out, no gaps)
    T dxv[NX], dl = T(0), ld = T(0), la = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) dxv[i] = T(0);
    if (lane < NX) dxw[lane] = T(0);
    const int row = lane < NX ? lane : 0, col = lane < NXU ? lane : 0;
    // operands of one stage, loaded one stage ahead (nothing else runs on the SIMD to hide a global load behind)
    T ar[2][NXU], qj[2], Kk[2][NX * NU], kk[2][NU], dk[2];
    auto load = [&](int k_, T* ar_, T& qj_, T* Kk_, T* kk_, T& dk_) LFSD_LAMBDA_INLINE {
      const T* Mk = Mc + (long long)k_ * Lay::M_ELEMS;
#pragma unroll
      for (int j = 0; j < NXU; ++j) ar_[j] = Mk[row * Lay::NXUP + j];
      qj_ = Mk[NX * Lay::NXUP + col];
      const T* Kg = this->Kws + (long long)k_ * NX * NU;
#pragma unroll
      for (int i = 0; i < NX * NU; ++i) Kk_[i] = Kg[i];
#pragma unroll
      for (int a = 0; a < NU; ++a) kk_[a] = this->kws[k_ * NU + a];
      dk_ = gp ? gp[k_ * NX + row] : T(0);
    };
    // (two operand sets used alternately: every index is a constant, nothing lands in scratch)
    auto stage = [&](int k, const T* arC, T qjC, const T* KkC, const T* kkC, T dkC, T* arN, T& qjN, T* KkN, T* kkN, T& dkN) LFSD_LAMBDA_INLINE {
      if (k + 1 < N) { load(k + 1, arN, qjN, KkN, kkN, dkN); LFSD_ISSUE_FENCE(); }
      T duv[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) duv[a] = kkC[a];
#pragma unroll
      for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int a = 0; a < NU; ++a) duv[a] += KkC[i * NU + a] * dxv[i];
      }
      T nx = dkC, zj = T(0);
#pragma unroll
      for (int j = 0; j < NX; ++j) { nx += arC[j] * dxv[j]; if (lane == j) zj = dxv[j]; }
#pragma unroll
      for (int a = 0; a < NU; ++a) { nx += arC[NX + a] * duv[a]; if (lane == NX + a) zj = duv[a]; }
      if (lane < NXU) dl += qjC * zj;
      if (lane < NX) { ldsDx[lane] = nx; dxw[(k + 1) * NX + lane] = nx; const T lk = lam_out[(k + 1) * NX + lane]; ld += lk * dkC; la += t_abs(lk * dkC); }
      else if (lane < NXU) duw[k * NU + (lane - NX)] = zj;
      LFSD_STAGE_SYNC_GEN();
#pragma unroll
      for (int i = 0; i < NX; ++i) dxv[i] = ldsDx[i];
      LFSD_STAGE_SYNC_GEN();
    };
    load(0, ar[0], qj[0], Kk[0], kk[0], dk[0]);
    for (int k = 0; k < N; k += 2) {
      stage(k, ar[0], qj[0], Kk[0], kk[0], dk[0], ar[1], qj[1], Kk[1], kk[1], dk[1]);
      if (k + 1 < N) stage(k + 1, ar[1], qj[1], Kk[1], kk[1], dk[1], ar[0], qj[0], Kk[0], kk[0], dk[0]);
    }
    {
      T xN[NX], hx[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) xN[i] = xbp(cur)[N * NX + i];
      M::final_grad(this->tk(N), xN, e, c, hx);
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) dl += hx[i] * dxv[i];
      }
    }
    const T D = wave_sum(dl);        // (its barriers also publish dxw / duw to every lane)
    lamd = wave_sum(ld);
    lamabs = wave_sum(la);
    return D;
  }
  // The iterate  (xb, ub)[cur] + alpha (dxw, duw)  into buffer `nxt`: every interval integrated and linearised from its own node
  // state, all intervals at once (lane <- (interval, column pair) as in linearise_parallel), its gaps into gapb[nxt].
  // J: cost of the iterate (interval costs + final cost of its last node); g1 / gm: l1 norm / largest entry of its gaps.
  // lamabs: sum_k sum_i |lambda_k+1,i| |d_k,i| with the costates of the CURRENT iterate (lam_out: the weights of the merit function)
  LFSD_DEV void ms_trial(int cur, int nxt, T alpha, T& J, T& lamabs, T& lamd, T& g2, T& g1, T& gm) {
    T Jl = T(0), g1l = T(0), gml = T(0), lal = T(0), ldl = T(0), g2l = T(0);
    constexpr bool PK2 = sizeof(T) == 4;
    constexpr bool ALLC = PK2 && Lay::HALL;      // one item per interval, all column pairs on its lane
    constexpr int NCT = ALLC ? 1 : (PK2 ? (NXU + 1) / 2 : NXU);
    using V = typename std::conditional<PK2, pk2<T>, T>::type;
    for (int t = lane; t < N * NCT; t += 64) {
      const int k = t / NCT, c0 = PK2 ? 2 * (t % NCT) : (t % NCT), c1 = c0 + 1;
      T x[NX], u[NU], q = T(0);
      V m[NX], du[NU], mq = V(T(0));
#pragma unroll
      for (int i = 0; i < NX; ++i) {
        x[i] = xbp(cur)[k * NX + i] + alpha * dxw[k * NX + i];
        if constexpr (PK2) m[i] = mk2<T>((c0 == i) ? T(1) : T(0), (c1 == i) ? T(1) : T(0)); else m[i] = (c0 == i) ? T(1) : T(0);
      }
#pragma unroll
      for (int a = 0; a < NU; ++a) {
        u[a] = ubp(cur)[k * NU + a] + alpha * duw[k * NU + a];
        if constexpr (PK2) du[a] = mk2<T>((c0 == NX + a) ? T(1) : T(0), (c1 == NX + a) ? T(1) : T(0)); else du[a] = (c0 == NX + a) ? T(1) : T(0);
      }
      if (c0 == 0) {
#pragma unroll
        for (int i = 0; i < NX; ++i) xbp(nxt)[k * NX + i] = x[i];
#pragma unroll
        for (int a = 0; a < NU; ++a) ubp(nxt)[k * NU + a] = u[a];
      }
      const T tt = this->tk(k);
      if constexpr (ALLC) {
        this->template interval_sens_all<Lay::NVH>(tt, x, q, u, Mwp(nxt) + (long long)k * Lay::M_ELEMS);
      } else {
      for (int s = 0; s < S; ++s) this->template rk4_step<true, V>(tt, x, q, u, m, mq, du);
      if constexpr (PK2) {
        V* Mk = reinterpret_cast<V*>(Mwp(nxt) + (long long)k * Lay::M_ELEMS + c0);
#pragma unroll
        for (int i = 0; i < NX; ++i) Mk[i * (Lay::NXUP / 2)] = m[i];
        Mk[NX * (Lay::NXUP / 2)] = mq;
      } else {
        T* Mk = Mwp(nxt) + (long long)k * Lay::M_ELEMS + c0;
#pragma unroll
        for (int i = 0; i < NX; ++i) Mk[i * Lay::NXUP] = m[i];
        Mk[NX * Lay::NXUP] = mq;
      }
      }
      if (c0 == 0) {
        Jl += q;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const T xn = xbp(cur)[(k + 1) * NX + i] + alpha * dxw[(k + 1) * NX + i];
          const T g = x[i] - xn;
          gapp(nxt)[k * NX + i] = g;
          g1l += t_abs(g); g2l += g * g; { const T lg = lam_out[(k + 1) * NX + i] * g; lal += t_abs(lg); ldl += lg; }
          gml = (t_abs(g) > gml || !t_finite(g)) ? t_abs(g) : gml;
          if (k == N - 1) xbp(nxt)[N * NX + i] = xn;
        }
        if (k == N - 1) {
          T xN[NX];
#pragma unroll
          for (int i = 0; i < NX; ++i) xN[i] = xbp(cur)[N * NX + i] + alpha * dxw[N * NX + i];
          Jl += M::final_cost(this->tk(N), xN, e, c);
        }
      }
    }
    J = wave_sum(Jl);
    lamabs = wave_sum(lal);
    lamd = wave_sum(ldl);
    g2 = wave_sum(g2l);
    g1 = wave_sum(g1l);
    gm = wave_max(gml);
  }
};

// Point a solver view at one trajectory: its LDS region, its scratch slot and its costate rows.  GL is the lane-group
// size the LDS / workspace layouts were sized for (the Riccati-style "one column per lane" mapping).
template <class M, typename T, int GL, class Sol>
LFSD_DEV void oc_bind(Sol& s, const OcArgs<T>& a, T* region, long long slot, bool valid, long long traj) {
  using Lay = OcLayout<M>;
  constexpr int NX = M::NX, NU = M::NU;
  s.N = a.n_grid; s.S = a.steps_per_grid;
  s.mu_stage_frac = a.mu_stage_frac;
  s.lds = region;
  s.e = region + Lay::template lds_e<GL>();
  s.c = region + Lay::template lds_c<GL>();
  s.x0 = region + Lay::template lds_x0<GL>();
  s.horizon = a.horizon[traj];
  s.dgrid = s.horizon / T(s.N);
  s.DT = s.dgrid / T(s.S);
  const int N = s.N;
  T* w = a.ws + slot * a.ws_stride;
  s.xb[0] = w; w += (N + 1) * NX;
  s.xb[1] = w; w += (N + 1) * NX;
  s.ub[0] = w; w += N * NU;
  s.ub[1] = w; w += N * NU;
  s.Mws[0] = w; w += (long long)N * Lay::M_ELEMS;
  s.Mws[1] = w; w += (long long)N * Lay::M_ELEMS;
  s.Kws = w; w += (long long)N * NX * NU;
  s.kws = w; w += N * NU;
  s.exws = w; w += (long long)Lay::SMAX * NX * (1 + GL);
  s.Hws = w; w += (long long)N * Lay::H_ELEMS;
  // padding groups (slot >= batch) clone the last trajectory and keep their costates in scratch
  s.lam_out = valid ? a.costate_grid + traj * (N + 1) * NX : w;
}

// The gradient test  |Q_u|_inf < tol (1 + |J|)  sits at the rounding floor of the gradient in fp32 (tol 1e-6).  Measured on the
// benchmark (profiles/r02_h_oc_straggler.txt): the typical trajectory reaches 1.2e-5 against a threshold of 1.1e-5 after
// five steps and pays a sixth for 0.5e-5; one whose gradient norm stops at 1.0-1.2e-5 against 0.8e-5 wandered through
// every fall-back of the step control for 6 more iterations -- and one such trajectory holds its whole launch.
// Second test, for the Newton-like models without a Levenberg shift.  `dec` = the decrease the full step predicts = half
// the squared Newton decrement; below 2 eps |J| it is under the resolution of the cost itself.  The iterate is at working
// precision when that holds now AND either
//   - it already held at the nominal the last accepted step left: the first such step leaves an error of sqrt(eps) size
//     in the iterate (which the fp32 parity tests do see), the Newton step after it squares that
//     -- accepted when the gradient is within a factor 2 of the tolerance (flat problems such as the robot arm reach a
//     small decrement long before their iterate has settled: there the gradient test stays in charge); or
//   - the last accepted step did not contract the gradient (g > g_last / 2): the quadratic phase has ended at the floor
//     -- accepted when the gradient is within a factor 32 of the tolerance.
// Reported as ST_STALLED ("converged to working precision").
template <typename T> LFSD_DEV bool at_working_precision(int mode, T mu, T gnorm, T g_last, T dec_last, T dV1, T dV2, T J, T tol) {
  const T res = T(2) * Eps<T>::v() * t_abs(J), gtol = tol * (T(1) + t_abs(J));
  if (!(mode >= 1 && mu == T(0) && g_last >= T(0) && -(dV1 + dV2) <= res)) return false;
  return (dec_last <= res && gnorm < T(2) * gtol) || (gnorm > T(0.5) * g_last && gnorm < T(32) * gtol);
}

// EXACT = false: lean instantiation without the exact-Hessian code (Gauss-Newton / Hamiltonian models only);
// EXACT = true: may switch to the exact stage Hessians.  lfsd_coc_solve runs the lean kernel for the first
// `exact_after` iterations and resumes the unfinished trajectories in the exact-capable one.
// PK = true (lean kernel of the 32-lane models only): the roll-out + linearisation and the line search run on
// 16-lane groups -- two tangent columns per lane on packed math (rollout_sens_pk) -- so a wavefront carries four
// trajectories instead of two; the backward sweep keeps its one-column-per-lane mapping on 32-lane groups and is run
// in two passes.  The phases already meet in the per-trajectory scratch ([A B q], gains, nominal), so only the
// few scalars of the step control cross between the mappings, through an LDS mailbox.
template <class M, typename T, int G, bool EXACT, bool PK = false>
__global__ void __launch_bounds__(64, LFSD_WAVES_OC) oc_solve_kernel(OcArgs<T> a) {
  constexpr int GR = PK ? 16 : G;                 // lanes per trajectory of the roll-out / line-search mapping
  using Sol = OcSolver<M, T, GR, EXACT>;
  using SolB = OcSolver<M, T, G, EXACT>;          // backward-sweep mapping
  using Lay = OcLayout<M>;
  constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NC = M::NC;
  constexpr int GPB = 64 / GR;
  // backward sweep of the packed kernel on the matrix cores (OcSolver::backward_mf) where the model fits 16-lane groups
  constexpr bool MF = PK && (LFSD_MFMA_BACKWARD != 0) && sizeof(T) == 4 && NX <= 16 && NX + NU <= 17;
  // ... without the structurally constant tangent columns of the model (OcSolver::backward_sc, rollout_sens_sc)
  constexpr bool SC = MF && (LFSD_STRUCT_COLS != 0) && Lay::sc_ok;
  constexpr int RS = EXACT ? Lay::template lds_elems<G>() : ((Lay::template lds_ex<G>() + 3) / 4) * 4;
  constexpr int MB = 12;                          // mailbox floats per trajectory
  static_assert(64 % G == 0 && G >= NX + NU, "lane group must hold one column of [A B] per lane");
  // PK in fp64: no packed math -- one LIVE column per lane of the 16-lane group instead (OcSolver::rollout_sens_live)
  constexpr bool LV = PK && sizeof(T) == 8;
  static_assert(!PK || (!EXACT && G == 32 && (LV ? Lay::LIVE <= GR : 2 * GR >= NX + NU)), "packed / live-column roll-out: lean kernel of a 32-lane model");
  __shared__ T lds_all[GPB * RS];
  __shared__ T mbox[PK ? GPB * MB : 1];
  __shared__ T park[LV ? 2 * NX * 64 + GPB * 2 * NX : 1];      // rk4_step_parked: [2 NX][64 lanes], then [GPB][2 NX]
  // ... and the ONE-pass backward sweep of the structural layout (backward_sc with LDS-fed products) where the model has it
  constexpr bool SC64 = LV && Lay::sc_ok;
  __shared__ T yzx[SC64 ? GPB * 16 * (Lay::ZC ? Lay::ZC : 1) : 1];
  __shared__ int vote[3];
  poison_lds(lds_all, GPB * RS);
  Sol s;
  const int gib = threadIdx.x / GR;
  s.lane = threadIdx.x % GR;
  const long long slot = (long long)blockIdx.x * GPB + gib;      // scratch slot (padded batch)
  const bool in_batch = slot < a.batch;
  const long long traj = in_batch ? slot : (long long)a.batch - 1;
  // phase 2 of a two-launch solve: only trajectories the first launch left at MAXITER are continued; the others
  // ride along as clones whose outputs are not written
  const bool valid = in_batch && (!a.resume || a.status[traj] == ST_MAXITER);
  if (a.resume) {
    if (threadIdx.x == 0) vote[0] = 0;
    __syncthreads();
    if (valid) vote[0] = 1;
    __syncthreads();
    if (!vote[0]) return;              // nothing to do in this workgroup
    __syncthreads();
  }
  oc_bind<M, T, G>(s, a, lds_all + gib * RS, slot, valid, traj);
  if constexpr (LV) { s.pkm = park + threadIdx.x; s.pkx = park + 2 * NX * 64 + gib * 2 * NX; }
  if constexpr (SC64) s.yz64 = yzx + gib * 16 * Lay::ZC;
  {
    T* le = s.lds + Lay::template lds_e<G>();
    T* lc = s.lds + Lay::template lds_c<G>();
    T* lx = s.lds + Lay::template lds_x0<G>();
    for (int i = s.lane; i < NP; i += GR) le[i] = a.auxvar[traj * NP + i];
    for (int i = s.lane; i < NC; i += GR) lc[i] = a.consts[traj * a.const_stride + i];
    for (int i = s.lane; i < NX; i += GR) lx[i] = a.ini_state[traj * NX + i];
    if constexpr (M::ND > 0) { __syncthreads(); if (s.lane == 0) M::derive_consts(lc); }
  }
  __syncthreads();
  const int N = s.N;

  // backward sweep of this lane's trajectory; `force`: also when the trajectory is no longer running (final refresh
  // of the costates).  Returns whether the sweep ran (PK skips a pass none of whose trajectories wants it).
  auto do_backward = [&](int cur_, int mode_, T mu_, bool want_, T& gnorm_, T& dV1_, T& dV2_, T& dmin_, bool& ok_) LFSD_LAMBDA_BW -> bool {
    if constexpr (MF || SC64) {
      // all four trajectories of the wavefront sweep together on the matrix cores; a group that does not want the sweep
      // rides along without writing anything (the MFMAs need every lane)
      if (threadIdx.x == 0) vote[2] = 0;
      __syncthreads();
      if (want_) vote[2] = 1;
      __syncthreads();
      const bool any = vote[2] != 0;
      __syncthreads();
      if (!any) return false;
      T g_, d1_, d2_, dm_;
      bool okb;
      if constexpr (SC || SC64) okb = s.backward_sc(cur_, mode_, mu_, want_, g_, d1_, d2_, dm_);
      else okb = s.backward_mf(cur_, mode_, mu_, want_, g_, d1_, d2_, dm_);
      if (want_) { ok_ = okb; gnorm_ = g_; dV1_ = d1_; dV2_ = d2_; dmin_ = dm_; }
      return want_;
    } else if constexpr (!PK) {
      ok_ = s.backward(cur_, mode_, mu_, gnorm_, dV1_, dV2_, dmin_);
      return true;
    } else {
      T* mb = mbox + gib * MB;
      if (s.lane == 0) { mb[0] = T(cur_); mb[1] = T(mode_); mb[2] = mu_; mb[3] = want_ ? T(1) : T(0); }
      __syncthreads();
      SolB sb;
      sb.lane = threadIdx.x % G;
      for (int pass = 0; pass < GPB * G / 64; ++pass) {
        const int t0 = pass * (64 / G);
        bool want_pass = false;
        for (int j = 0; j < 64 / G; ++j) want_pass = want_pass || (mbox[(t0 + j) * MB + 3] != T(0));
        const int tb = t0 + threadIdx.x / G;
        T* mo = mbox + tb * MB;
        if (want_pass) {
          const long long slot_b = (long long)blockIdx.x * GPB + tb;
          const bool valid_b = slot_b < a.batch;
          oc_bind<M, T, G>(sb, a, lds_all + tb * RS, slot_b, valid_b, valid_b ? slot_b : (long long)a.batch - 1);
          T g_, d1_, d2_, dm_;
          const bool okb = sb.backward((int)mo[0], (int)mo[1], mo[2], g_, d1_, d2_, dm_);
          if (sb.lane == 0) { mo[4] = okb ? T(1) : T(0); mo[5] = g_; mo[6] = d1_; mo[7] = d2_; mo[8] = dm_; mo[9] = T(1); }
        } else if (sb.lane == 0) {
          mo[9] = T(0);
        }
        __syncthreads();
      }
      const bool ran = mb[9] != T(0);
      if (ran) { ok_ = mb[4] != T(0); gnorm_ = mb[5]; dV1_ = mb[6]; dV2_ = mb[7]; dmin_ = mb[8]; }
      __syncthreads();
      return ran;
    }
  };
  auto do_rollout = [&](int cur_, int nxt_, T alpha_, bool gains_) LFSD_LAMBDA_RO -> T {
    if constexpr (SC) return s.rollout_sens_sc(cur_, nxt_, alpha_, gains_);
    else if constexpr (LV) return s.template rollout_sens_live<SC64>(cur_, nxt_, alpha_, gains_);
    else if constexpr (PK) return s.rollout_sens_pk(cur_, nxt_, alpha_, gains_);
    else return s.rollout_sens(cur_, nxt_, alpha_, gains_);
  };

  // initial guess into buffer 1, then "roll out" 1 -> 0 without gains
  // (`warm`: the caller's initial guess of THIS trajectory is not all zero -- an all-zero row of u_init is the cold start, e.g.
  //  the rows of a learner that only continues the solves that ran out of iterations)
  bool warm = false;
  {
    T umax = T(0);
    for (int i = s.lane; i < N * NU; i += GR) {
      const T u0 = a.resume ? a.control_grid[traj * (N + 1) * NU + i] : (a.u_init ? a.u_init[traj * N * NU + i] : T(0));
      s.ub[1][i] = u0;
      umax = t_max(umax, t_abs(u0));
    }
    if (a.u_init != nullptr && !a.resume) {
      T* ldsRed0 = s.lds + Lay::LDS_RED;
      ldsRed0[s.lane] = umax;
      __syncthreads();
      for (int l = 0; l < GR; ++l) warm = warm || !(ldsRed0[l] == T(0));
    }
  }
  __syncthreads();
  int cur = 0;
#if defined(LFSD_OC_CLOCK)
  long long clk_bw = 0, clk_ro = 0, clk_ls = 0, clk_t0 = clock64();
#define LFSD_CLK(acc, stmt) { const long long c0_ = clock64(); stmt; acc += clock64() - c0_; }
#else
#define LFSD_CLK(acc, stmt) { stmt; }
#endif
  // Mesh continuation (lean fp32 kernel of the 32-lane models, round 3).  The first iterations of a cold start only have to
  // get near the optimum -- the zero-control roll-out of the quadrotor starts at J = 2.5e4 for an optimum of 10 -- and do
  // not need the reference's 4 RK4 steps per grid interval for that: while full steps keep gaining more than
  // LFSD_COARSE_SWITCH of the cost, roll-outs and linearisations run with ONE RK4 step per interval (a quarter of the work
  // of the phase that is 53 % of the kernel).  Then the nominal is rolled out and linearised once on the reference's
  // discretisation (`relin`, an iteration without a backward sweep) and the solve continues there: every convergence
  // test, every returned number belongs to the NLP of CPDP.py:110-175 with steps_per_grid sub-steps; the coarse phase only
  // changes the path to its KKT point, as the choice of DDP over IPOPT does.
  // (also the fp64 lean kernel of the same 32-lane models -- the reference-precision path of the benchmark; smaller models
  //  keep the reference's grid throughout: their parity cases include problems with several local minima, cart-pole swing-up,
  //  where another path may end in another KKT point than the oracle's)
  constexpr bool CS = !EXACT && (PK || (G == 32 && sizeof(T) == 8)) && (LFSD_COARSE_START != 0);
  bool coarse = CS && a.steps_per_grid > 1 && !a.resume && a.max_iter > 4 && !warm;      // (a caller's initial guess starts next to its answer)
  bool relin = false;       // leave the coarse grid at the next iteration ...
  bool relin_hard = false;      // ... by a roll-out + linearisation of the nominal without a step (else: with the step)
  if (coarse) { s.S = 1; s.DT = s.dgrid; }
  // Level 0 of the mesh continuation (round 4; the matrix-core kernels only): the first LFSD_LEAN_TC_ITERS iterations of a
  // workgroup whose trajectories are ALL cold also merge LFSD_LEAN_TC control intervals into one (n_grid / tc stages in the
  // backward sweep, n_grid / tc * LFSD_LEAN_TC_S RK4 steps per roll-out), then the controls are prolongated -- each held over its
  // tc intervals -- and rolled out + linearised on the coarse level above (an iteration's roll-out without a step).  The
  // schedule is a fixed iteration count, so what a trajectory does never depends on its partners in the wavefront; a
  // trajectory that wants to leave the coarse phase earlier waits for it (the tests that set `relin` are deferred).
  constexpr bool TCL = CS && (MF || SC64) && (LFSD_LEAN_TC > 1);
  const int N_full = s.N;
  const T dgrid_full = s.dgrid;
  int tc = 1;
  if constexpr (TCL) {
    if (threadIdx.x == 0) vote[0] = 0;
    __syncthreads();
    if (!coarse) vote[0] = 1;
    __syncthreads();
    const bool all_cold = vote[0] == 0;
    __syncthreads();
    if (all_cold && N_full % (LFSD_LEAN_TC) == 0 && N_full / (LFSD_LEAN_TC) >= LFSD_LEAN_TC_MIN && a.max_iter > LFSD_LEAN_TC_ITERS + 6) {
      tc = LFSD_LEAN_TC;
      s.N = N_full / tc; s.dgrid = dgrid_full * T(tc); s.S = LFSD_LEAN_TC_S; s.DT = s.dgrid / T(s.S);
    }
  }
  T J = do_rollout(1, 0, T(0), false);
  __syncthreads();
  if constexpr (CS) {
    // A coarse roll-out (ONE RK4 step per interval) that overflows says nothing about the reference's discretisation: before
    // a trajectory is declared FAILED the initial guess is rolled out on the reference grid, as the wide kernel does.  The
    // roll-out is a block-wide phase, so the decision goes through the vote; the partners of such a trajectory simply
    // start on the reference grid as well.
    if (threadIdx.x == 0) vote[0] = 0;
    __syncthreads();
    if (coarse && !t_finite(J)) vote[0] = 1;
    __syncthreads();
    const bool redo = vote[0] != 0;
    __syncthreads();
    if (redo) {
      coarse = false; tc = 1; s.N = N_full; s.dgrid = dgrid_full;
      s.S = a.steps_per_grid; s.DT = s.dgrid / T(s.S);
      J = do_rollout(1, 0, T(0), false);
      __syncthreads();
    }
  }
  T mu = T(0);
  int mode = (warm && a.start_mode == 1) ? 1 : 0;      // stage Hessian model: 0 Gauss-Newton, 1 Hamiltonian (cheap Newton-like), 2 exact
  bool ham_ok = true;       // the cheap Newton-like model has not failed on this trajectory yet
  bool optimistic = true;   // try the full step directly (skips the parallel line search while alpha = 1 keeps working)
  const int it_off = (a.resume && in_batch) ? (a.iters[traj] - a.it_start) : 0;   // iterations already spent in phase 1
  int status = ST_RUNNING, it = a.it_start, my_iters = a.it_start;
  bool need_bw = true;      // costates on `lam_out` are stale
  T gnorm = T(0), dV1 = T(0), dV2 = T(0);
  T g_flat = T(-1);         // gradient norm at the last accepted noise-level ("flat") step; <0: none yet
  T g_last = T(-1);         // gradient norm of the nominal the last accepted step left; <0: none yet
  T dec_last = T(1e30);     // ... and the decrease its full step predicted
  T J_ref = J;              // cost 4 accepted steps ago (stagnation window)
  int n_acc = 0;
  bool hess_ok = false;     // Hws holds the exact stage Hessians of nominal `cur`
  bool gn_crawl = false;    // Gauss-Newton is all that is left (Hamiltonian model failed) and its full steps gain < 1 %
  T mu_bad = T(-1);         // largest Levenberg shift that failed recently (<0: none)
  int mu_hold = 0, mu_hold_need = LFSD_MU_HOLD;   // accepted full steps to wait before the shift returns to a level <= mu_bad
  // accepted steps after the transfer from level 0 during which a refused or shortened full step is NOT read as "past the big
  // drops, go to the reference's grid": there it only says that the level-0 model and this level disagree, and the line search
  // it asks for is four times cheaper here
  int grace = 0;
  bool conv_coarse = false;      // the coarse problem passed a convergence test (which, there, only asks for the reference's grid)
  if (!t_finite(J)) status = ST_FAILED;
  for (; it < a.max_iter; ++it) {
    if (threadIdx.x == 0) vote[0] = 0;
    __syncthreads();
    if (status == ST_RUNNING) vote[0] = 1;
    __syncthreads();
    if (!vote[0]) break;
    __syncthreads();
    if constexpr (TCL) {
      if (tc > 1 && (it >= a.it_start + LFSD_LEAN_TC_ITERS || it + 5 >= a.max_iter)) {      // (uniform in the workgroup)
        // prolongation in place: control k of the finer level = control k / tc (chunks from the back: the source of element i is at
        // an index <= i, so no chunk overwrites the source of an earlier one)
        T* uc = s.ubp(cur);
        const int tot = N_full * NU;
        for (int c0 = ((tot - 1) / GR) * GR; c0 >= 0; c0 -= GR) {
          const int i = c0 + s.lane;
          const T keep = (i < tot) ? uc[((i / NU) / tc) * NU + (i % NU)] : T(0);
          __syncthreads();
          if (i < tot) uc[i] = keep;
          __syncthreads();
        }
        s.N = N_full; s.dgrid = dgrid_full; tc = 1;
        // a trajectory that already asked for the reference's grid goes there directly
        if (relin && status == ST_RUNNING) { s.S = a.steps_per_grid; coarse = false; relin = false; } else { s.S = 1; }
        s.DT = s.dgrid / T(s.S);
        T Jt;
        LFSD_CLK(clk_ro, Jt = do_rollout(cur, cur ^ 1, T(0), false));
        __syncthreads();
        cur ^= 1; J = Jt;
        need_bw = true; hess_ok = false; optimistic = true;
        grace = coarse ? (LFSD_LEAN_TC_GRACE) : 0;
        g_last = T(-1); dec_last = T(1e30); g_flat = T(-1); J_ref = J; n_acc = 0;
        if (!t_finite(J) && status == ST_RUNNING) {
          if (coarse) { relin = true; relin_hard = true; }      // the reference's discretisation decides
          else status = ST_FAILED;
        }
      }
    }
    if (coarse && tc == 1 && it + 3 >= a.max_iter) relin = true;      // never leave a launch on the coarse grid
    // Leaving the coarse grid.  Normally WITH this iteration's step (`fine_step`): the backward sweep still runs on the coarse
    // linearisation, its full step is rolled out and linearised on the reference's discretisation and taken unless the cost
    // rises beyond what the two discretisations can differ by (their costs differ by ~1e-5 relative, about the gain a
    // Newton-like step predicts at this point, so the Armijo test cannot referee this one step).  Otherwise -- the full step
    // was refused, or a line search is due -- WITHOUT a step (`relin_now`): the nominal is rolled out and linearised again.
    const bool fine_step = CS && relin && tc == 1 && !relin_hard && optimistic && status == ST_RUNNING;
    const bool relin_now = CS && relin && tc == 1 && !fine_step && status == ST_RUNNING;
    if constexpr (MF) {
      // A solve whose last Newton step already predicted a decrease below the resolution of the cost is about to pass its
      // convergence test: try the costate-only sweep (gradient norm + costates, a few per cent of a full sweep) first.
      // Accepted on the gradient test itself, or on the first clause of at_working_precision (the decrement of the nominal
      // just left stands in for this one's, which only the full sweep knows: the gradient must not have grown).
      const bool want_c = status == ST_RUNNING && !coarse && mode >= 1 && mu == T(0) && g_last >= T(0) && dec_last <= T(2) * Eps<T>::v() * t_abs(J);
      if (threadIdx.x == 0) vote[1] = 0;
      __syncthreads();
      if (want_c) vote[1] = 1;
      __syncthreads();
      const bool any_c = vote[1] != 0;
      __syncthreads();
      if (any_c) {
        T gc = T(0);
        if constexpr (SC) s.costate_sweep_sc(cur, want_c, gc);
        else s.costate_sweep_mf(cur, want_c, gc);
        if (want_c) {
          const T gtol = a.tol * (T(1) + t_abs(J));
          if (gc < gtol) status = ST_CONVERGED;
          else if (gc < T(2) * gtol && gc < g_last) status = ST_STALLED;
          if (status != ST_RUNNING) { gnorm = gc; need_bw = false; my_iters = it + 1 + it_off; }
        }
      }
    }
    // hand over to the exact stage Hessians at the iteration limit of the cheap models.  (Measured: handing over
    // earlier, e.g. after three full Gauss-Newton steps, costs more regularised Newton steps than it saves.)
    // ... or as soon as Gauss-Newton is reduced to crawling (the oracle's "close: switch to Newton" rule)
    const bool want_exact = a.exact_after >= 0 && mode < 2 && (it >= a.exact_after || gn_crawl);
    if (EXACT) { if (want_exact) mode = 2; }
    else if (want_exact && status == ST_RUNNING && it < a.max_iter_total - 1) { status = ST_MAXITER; my_iters = it + it_off; }
    T dmin = T(0);
    bool bw_ok = false;
    s.reuse_hess = EXACT && mode == 2 && hess_ok;
    bool bw_ran = false;
    LFSD_CLK(clk_bw, bw_ran = do_backward(cur, mode, mu, status == ST_RUNNING && !relin_now, gnorm, dV1, dV2, dmin, bw_ok));
    if (bw_ran) { need_bw = false; hess_ok = EXACT && mode == 2; }
    bool try_step = false;
    if (relin_now) my_iters = it + 1 + it_off;
    if (status == ST_RUNNING && !relin_now) {
      my_iters = it + 1 + it_off;
      if (!bw_ok) {
        // indefinite Q_uu: the cheap Newton-like model hands over to the exact one; otherwise Levenberg shift
        if (mode == 1) { mode = 0; ham_ok = false; }     // back to Gauss-Newton until the exact model takes over
        else {
          mu_bad = mu; mu_hold = 0;
          if (mu == T(0) && mode >= 1 && t_finite(dmin))
            mu = t_min(t_max(T(-2) * dmin, T(1e-4)), T(1e6));     // first shift: the size of the negative pivot
          else
            mu = t_max(mu * T(LFSD_MU_UP), mode == 2 ? T(1e-4) : T(1e-6));
          if (mu > T(1e12)) status = ST_FAILED;
        }
      } else if (gnorm < a.tol * (T(1) + t_abs(J))) {
        status = ST_CONVERGED;
      } else if (at_working_precision(mode, mu, gnorm, g_last, dec_last, dV1, dV2, J, a.tol)) {
        status = ST_STALLED;
      } else {
        try_step = true;
      }
      if (coarse) {
        // convergence is only ever declared on the reference's discretisation; a failed sweep is retried there as well
        if (status == ST_CONVERGED || status == ST_STALLED) { status = ST_RUNNING; relin = true; try_step = fine_step; conv_coarse = true; }
        else if (!bw_ok) { relin = true; relin_hard = true; }
      }
    }
    // Step selection.  Optimistic groups roll the full step (alpha = 1) out together with its linearisation and
    // keep it if it passes the Armijo test; only after a failure do they pay for the parallel line search.
    const bool opt_try = try_step && optimistic;
    const bool ls_try = try_step && !optimistic;
    T alpha = T(0), Jmin = J;
    bool flat_full = false;
    int ia = -1;
    if (threadIdx.x == 0) vote[1] = 0;
    __syncthreads();
    if (ls_try) vote[1] = 1;
    __syncthreads();
    const bool any_ls = vote[1] != 0;
    __syncthreads();
    if (any_ls) LFSD_CLK(clk_ls, ia = s.linesearch(cur, J, dV1, dV2, alpha, Jmin, flat_full));
    bool accept = false;
    if (ls_try) {
      if (ia >= 0) {
        accept = true;
      } else if (mode >= 1 && flat_full && (g_flat < T(0) || gnorm < T(0.7) * g_flat)) {
        // Newton-like step whose cost change is below rounding noise: take it as long as the
        // gradient norm keeps contracting (this is what lets fp32 reach its gradient floor)
        accept = true; ia = 0; alpha = T(1); g_flat = gnorm;
      } else if (mode == 1) {
        mode = 0; ham_ok = false;
      } else if (mu > T(1e10) ||
                 ((J - Jmin) <= T(8) * Eps<T>::v() * t_abs(J) && (mode == 0 || flat_full || mu > T(1e6)))) {
        status = ST_STALLED;      // no step length gains more than rounding noise
      } else {                    // (exact model far from the optimum: indefinite direction -> larger shift)
        mu_bad = mu; mu_hold = 0;
        mu = t_max(mu * T(LFSD_MU_UP), mode == 2 ? T(1e-4) : T(1e-6));
      }
    }
#if defined(LFSD_TRACE)
    if (s.lane == 0 && slot == 0) printf("it %d st %d mode %d bw_ok %d g %.6e J %.12e opt %d ia %d alpha %g accept %d mu %g dV1 %.4e dV2 %.4e Jmin %.12e flat %d\n", it, status, mode, (int)bw_ok, (double)gnorm, (double)J, (int)opt_try, ia, (double)alpha, (int)accept, (double)mu, (double)dV1, (double)dV2, (double)Jmin, (int)flat_full);
#endif
    const bool roll = accept || opt_try;
    if (threadIdx.x == 0) vote[1] = 0;
    __syncthreads();
    if (roll || relin_now) vote[1] = 1;
    __syncthreads();
    if (vote[1]) {
      T Jn;
      if (relin_now || (fine_step && opt_try)) { s.S = a.steps_per_grid; s.DT = s.dgrid / T(s.S); coarse = false; }
      LFSD_CLK(clk_ro, Jn = do_rollout(cur, cur ^ 1, opt_try ? T(1) : (accept ? alpha : T(0)), roll));
      if (relin_now) {
        // the same controls on the reference's discretisation: new nominal, new cost, fresh linearisation; the histories of
        // the convergence tests start over
        cur ^= 1; J = Jn; relin = false;
        need_bw = true; hess_ok = false;
        g_last = T(-1); dec_last = T(1e30); g_flat = T(-1); J_ref = J; n_acc = 0;
        if (!t_finite(J)) status = ST_FAILED;
      }
      if (opt_try && fine_step) {
        relin = false;
        if (t_finite(Jn) && Jn <= J + T(1e-3) * t_abs(J)) {
          accept = true; ia = 0;
        } else {
          relin = true; relin_hard = true;      // refused: re-linearise the (coarse) nominal on the fine grid, then carry on there
        }
      } else if (opt_try) {
        const T flat = T(8) * Eps<T>::v() * t_abs(J);
        const bool fin = t_finite(Jn);
        if (fin && (J - Jn) >= T(1e-4) * (-(dV1 + dV2)) - flat && Jn < J) {
          accept = true; ia = 0;
        } else if (fin && mode >= 1 && t_abs(Jn - J) <= T(64) * Eps<T>::v() * t_abs(J) &&
                   (g_flat < T(0) || gnorm < T(0.7) * g_flat)) {
          accept = true; ia = 0; g_flat = gnorm;
        } else {
          optimistic = false;     // same gains, parallel line search next round
          if (coarse && grace == 0) { relin = true; relin_hard = true; }      // ... on the fine grid (right after level 0: on this level)
        }
      }
      if (accept) {
        if (coarse && ((ia != 0 && grace == 0) || (J - Jn) < T(LFSD_COARSE_SWITCH) * t_abs(Jn))) { relin = true; if (ia != 0) relin_hard = true; }      // past the first big drops
        if (grace > 0) --grace;
        cur ^= 1;
        g_last = gnorm; dec_last = (mode >= 1 && mu == T(0)) ? -(dV1 + dV2) : T(1e30);
        // histories start over on the fine grid -- except after a coarse phase that ended because the coarse problem had CONVERGED:
        // the step just taken was a Newton-like step from a nominal whose decrement was below the resolution of the cost, which is
        // what the first clause of at_working_precision asks of the previous nominal.  Without it a trajectory whose first
        // gradient on the reference's grid lands between 1x and 2x the tolerance (the fp32 floor of this problem is 0.5x) pays
        // one more iteration for nothing the cost can resolve, and holds its launch (2 % of the benchmark's trajectories, 0.35 ms
        // of 2.4 for all of them: profiles/r04_av_headline_eighth_iteration.txt)
        // (... or, the same situation reached by the gain rule: the step that left the coarse grid was a Newton-like step without a
        //  shift that itself predicted a decrease below the resolution of the cost)
        const bool below_res = mode >= 1 && mu == T(0) && -(dV1 + dV2) <= T(2) * Eps<T>::v() * t_abs(J);
        if (fine_step && !((LFSD_EXIT_KEEP_HISTORY) != 0 && (conv_coarse || ((LFSD_EXIT_KEEP_HISTORY) > 1 && below_res)))) { g_last = T(-1); dec_last = T(1e30); }
        if (fine_step) { g_flat = T(-1); J_ref = Jn; n_acc = 0; }
        need_bw = true;
        hess_ok = false;
        optimistic = (ia == 0);
        if (ia == 0) {
          // relax the shift after a full step -- but not straight back to a level that has just failed: hold for
          // mu_hold_need accepted steps first, and twice as long after every failed return (a shift that bounces
          // between a failing and a working level wastes every other backward sweep)
          const T mu_next = (mu > T(1e-8)) ? mu * T(LFSD_MU_DOWN) : T(0);
          if (mu > T(0) && mu_bad >= T(0) && mu_next <= mu_bad && mu_hold < mu_hold_need) {
            ++mu_hold;
          } else {
            mu = mu_next; mu_hold = 0;
          }
          if (mode == 0 && ham_ok && (J - Jn) < T(LFSD_HAM_SWITCH) * t_abs(Jn)) mode = 1;      // past the first big drops: Newton-like
          else if (mode == 0 && !ham_ok && (J - Jn) < T(1e-2) * t_abs(Jn)) gn_crawl = true;
        }
        J = Jn;
        if (++n_acc >= 4) {
          // four accepted steps that together gain less than rounding noise: converged to working precision
          if (status == ST_RUNNING && J_ref - J <= T(16) * Eps<T>::v() * t_abs(J)) status = ST_STALLED;
          J_ref = J; n_acc = 0;
        }
      }
    }
    __syncthreads();
  }
  if (status == ST_RUNNING) { status = ST_MAXITER; }
  if (threadIdx.x == 0) vote[0] = 0;
  __syncthreads();
  if (need_bw) vote[0] = 1;
  __syncthreads();
  { T dmin = T(0); bool okf = false; if (vote[0]) LFSD_CLK(clk_bw, do_backward(cur, 0, T(0), need_bw, gnorm, dV1, dV2, dmin, okf)); }   // refresh costates on the final nominal
#if defined(LFSD_OC_CLOCK)
  if (threadIdx.x == 0 && blockIdx.x < LFSD_OC_CLOCK)
    printf("oc clock wave %d: iterations %d total %lld backward %lld rollout %lld linesearch %lld (shader clocks)\n", (int)blockIdx.x, it, clock64() - clk_t0, clk_bw, clk_ro, clk_ls);
#endif
#undef LFSD_CLK
  __syncthreads();
  if (valid) {
    T* xo = a.state_grid + traj * (N + 1) * NX;
    T* uo = a.control_grid + traj * (N + 1) * NU;
    for (int i = s.lane; i < (N + 1) * NX; i += GR) xo[i] = s.xbp(cur)[i];
    for (int i = s.lane; i < (N + 1) * NU; i += GR) uo[i] = s.ubp(cur)[(i < N * NU) ? i : i - NU];
    if (s.lane == 0) { a.cost[traj] = J; a.iters[traj] = my_iters; a.status[traj] = status; }
  }
}

// The WIDE solver: one trajectory per wavefront (grid = batch), same discretisation, same stage-Hessian models and the same
// step control as oc_solve_kernel (one backward sweep = one iteration; the step lengths 2^0..2^-15 are all rolled out at
// once, so "optimistic full step, then line search" is a single phase: the largest step length that passes the Armijo
// test is taken).  Control flow is uniform per workgroup -- no votes, no lock-step partners.  lfsd_coc_solve picks this
// kernel when the lock-step mapping would leave most SIMDs without a wavefront (the `mapping` argument overrides).
// W: wavefronts per trajectory (OcWide).  A launch with W > 1 is either the whole solve of a small batch or the second launch of a
// two-launch solve (a.resume == 2): then workgroup b takes entry b of the hand-over list the first launch wrote (a.sched) and continues
// from the solver state that launch parked in the workspace; workgroups beyond the list leave at once.
template <class M, typename T, bool EXACT, bool BND = false, int W = 1>
__global__ void __launch_bounds__(64 * W, 1) oc_solve_wide_kernel(OcArgs<T> a) {
  using Sol = OcWide<M, T, EXACT, BND, W>;
  using Lay = OcLayout<M>;
  constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NC = M::NC, NAL = Sol::NAL;
  static_assert(W == 1 || (!Lay::HALL && W <= Lay::WIDE_WMAX), "several wavefronts per trajectory: the models whose parallel phases take several rounds");
  constexpr int RS = EXACT ? Lay::template lds_elems<64, (int)sizeof(T)>() : ((Lay::template lds_ex<64>() + 3) / 4) * 4;
  __shared__ __attribute__((aligned(16))) T lds_all[W * RS];
  if (blockDim.x != 64 * W) return;
  const bool resuming = W > 1 && a.resume == 2;      // (only launches with several wavefronts per trajectory take solves over)
  if (resuming && (int)blockIdx.x >= a.sched[1]) return;      // (uniform per workgroup: the hand-over list is shorter than the grid)
  const long long traj = resuming ? a.sched[2 + blockIdx.x] : (long long)blockIdx.x;
  poison_lds(lds_all, W * RS);
  Sol s;
  s.lane = W == 1 ? (int)threadIdx.x : (int)(threadIdx.x & 63);
  s.wave = W == 1 ? 0 : (int)(threadIdx.x >> 6);
  oc_bind<M, T, 64>(s, a, lds_all + s.wave * RS, traj, true, traj);
  const int N = s.N;
  T* wstate;      // [WIDE_STATE] parked solver state
  {
    T* w2 = a.ws + traj * a.ws_stride + Lay::template ws_elems<64>(N);
    s.xa = w2; s.ua = s.xa + (long long)NAL * (N + 1) * NX; s.exwu = s.ua + (long long)NAL * N * NU;
    s.exwm = s.exwu + (long long)Lay::SMAX * NX * 64;
    if ((((long long)(s.exwm - a.ws)) & 1) != 0) ++s.exwm;      // (8-byte aligned: it holds packed pairs; the region has a word to spare)
    T* w3 = s.exwu + (long long)Lay::SMAX * NX * 64 + (Lay::HALL ? 2LL * Lay::NVH * Lay::SMAX * NX * 64 + 2 : 0LL);      // (the order of OcLayout::ws_elems_wide)
    // the gaps of the two nominal buffers have words of their own; the Newton step of a multiple-shooting iterate lives in the
    // region of the parked step-length roll-outs (dead by the time a roll-out is parked)
    s.gapb[0] = w3; s.gapb[1] = w3 + (long long)N * NX; w3 += 2LL * N * NX;
    s.dxw = s.xa; s.duw = s.dxw + (long long)(N + 1) * NX;
    if constexpr (W > 1) {
      if (s.wave > 0) {      // per-lane scratch of the exact-Hessian sweeps: every wavefront its own
        T* xw = w3 + (long long)(s.wave - 1) * Lay::WIDE_XW;
        s.exws = xw; s.exwu = xw + (long long)Lay::SMAX * NX * (1 + 64);
      }
    }
    wstate = w3 + (Lay::HALL ? 0LL : (Lay::WIDE_WMAX - 1) * Lay::WIDE_XW);
    T* le = s.lds + Lay::template lds_e<64>();
    T* lc = s.lds + Lay::template lds_c<64>();
    T* lx = s.lds + Lay::template lds_x0<64>();
    for (int i = s.lane; i < NP; i += 64) le[i] = a.auxvar[traj * NP + i];
    for (int i = s.lane; i < NC; i += 64) lc[i] = a.consts[traj * a.const_stride + i];
    for (int i = s.lane; i < NX; i += 64) lx[i] = a.ini_state[traj * NX + i];
    if constexpr (M::ND > 0) { __syncthreads(); if (s.lane == 0) M::derive_consts(lc); }
  }
  T* ldsRed = s.lds + Lay::LDS_RED;
  if constexpr (BND) {
#pragma unroll
    for (int b = 0; b < NU; ++b) { s.ulb[b] = a.u_lb[b]; s.uub[b] = a.u_ub[b]; }
    if (a.x_lb != nullptr) {
#pragma unroll
      for (int i = 0; i < NX; ++i) { s.xlb[i] = a.x_lb[i]; s.xub[i] = a.x_ub[i]; }
      s.xm = a.x_mult + traj * (long long)N * 2 * NX;
      s.xrho = a.x_rho;
    }
  }
  // ---- the solver state that is carried from iteration to iteration (and, parked in the workspace, from the launch that suspends a
  // solve to the launch that resumes it: `park` / `unpark` below list every one of them) ----
  int cur = 0;
  T alpha_l = T(0);
  // Mesh continuation (see oc_solve_kernel): the 32-lane models only (quadrotor, rocket -- the rocket's cold start needs
  // ~40-100 regularised Newton iterations of 2-5 % gain each before its last five quadratic ones, profiles/
  // r03_f_rocket_trace_head.txt, and every one of them pays a second-order adjoint sweep through S x 4 RK4 stages per
  // interval); never with an initial guess from the caller or with bounds (warm-started subproblems start next to their
  // answer).  Smaller models keep the reference's grid: their parity cases follow the oracle's path into one of several minima.
  constexpr bool CSW = !BND && (LFSD_COARSE_START != 0) && (NX + (NU > NP ? NU : NP) > 16);
  // ... and only where the coarse grid still has as many RK4 steps as the reference's example grids have in all (n_grid
  // 10-15 x 4): at n_grid 15 the rocket's coarse path ends in ANOTHER stationary point than the fine one (one with a
  // conjugate point inside the horizon, where the Riccati sweep of the auxiliary pass has a finite escape; emulator tier,
  // test_rocket_newton_mode_vs_oracle) -- a step of 0.2 s is too long for its attitude dynamics under aggressive controls.
  bool coarse = false, relin = false;
  // ... and on a coarser CONTROL grid as well (LFSD_COARSE_TIME, round 4): `tc` grid intervals share one control, i.e. the coarse
  // phase solves the problem on n_grid / tc intervals of length tc * dgrid with tc RK4 steps each -- the same RK4 step as one step
  // per original interval, but tc times fewer stages in the backward recursion and tc times fewer exact stage Hessians, the two
  // phases that are 63 % of an iteration (profiles/r04_e_rocket_wide_clock.txt).  Leaving the coarse phase prolongates the
  // controls (each held over its tc intervals) and goes through the same re-linearisation on the reference's discretisation.
  const int N_full = s.N;
  const T dgrid_full = s.dgrid;
  int tc = 1;
  // Multiple-shooting steps (OcWide::ms_*; the step logic is in the loop below): every level of the mesh continuation starts from a
  // roll-out (no gaps); an iterate with gaps is closed by a closed-loop roll-out before any convergence test applies to it
  // (`ms_check`: the cheap costate-only gradient test comes first then).
  // Not for solves that run Newton from their first iteration (exact_after == 0, the rocket): their steps are regularised Newton
  // steps at Levenberg shifts of 10^2 - 10^4 through strongly curved attitude dynamics; the linear prediction of the node states
  // then leaves gaps as large as the step closes, the line search settles on steps of 1/4, and the phase needs as many or more
  // iterations than the closed-loop nonlinear roll-out while saving only the roll-outs' 20 % of an iteration (measured:
  // DESIGN.md).
  const bool ms_on = !BND && (LFSD_MS) != 0 && a.n_grid >= (LFSD_MS_MIN_GRID) && a.max_iter > 8 && a.exact_after != 0;
  bool ms = ms_on, ms_check = false, ms_floor = false;
  int n_acc_need = 4, n_ms = 0, n_half = 0;
  T g1c = T(0), g2c = T(0), gmc = T(0);      // l1 norm, sum of squares and largest entry of the gaps of the current iterate (0: a roll-out)
  T J = T(0), J_feas = T(0);                 // cost of the iterate; cost of the last iterate WITHOUT gaps (a roll-out)
  T mu = T(0);
  int mode = (EXACT && a.exact_after == 0) ? 2 : 0;
  bool ham_ok = true, hess_ok = false, gn_crawl = false, costates_ok = false;
  int status = ST_RUNNING, it = 0;
  T gnorm = T(0), dV1 = T(0), dV2 = T(0), g_flat = T(-1), g_last = T(-1), dec_last = T(1e30), J_ref = T(0), mu_bad = T(-1);
  int n_acc = 0, mu_hold = 0;
#if defined(LFSD_OC_CLOCK)      // diagnostic build (tools/wide_clock.py): shader clocks of the phases of the slowest solves
  long long wck[7] = {0, 0, 0, 0, 0, 0, 0}, wck_exit = 0;
  int wck_it_exit = -1;
  const long long wck_t0 = clock64();
#define LFSD_WCK(i, stmt) { const long long c0_ = clock64(); stmt; wck[i] += clock64() - c0_; }
#else
#define LFSD_WCK(i, stmt) { stmt; }
#endif
  if (!resuming) {
  // initial guess into buffer 1 (the reference's w0: zero, or the midpoint of finite control bounds, CPDP.py:153), rolled out
  // without gains into buffer 0, linearised
  // (`warm`: the caller's initial guess of THIS trajectory is not all zero.  An all-zero row of u_init is the cold start -- a
  //  learner that only continues the solves that ran out of iterations hands zeros for every other row -- and keeps the
  //  mesh continuation below, exactly as in oc_solve_kernel)
  T umax = T(0);
  for (int i = s.lane; i < N * NU; i += 64) {
    T u0 = a.u_init ? a.u_init[traj * N * NU + i] : T(0);
    umax = t_max(umax, t_abs(u0));
    if constexpr (BND) {
      const T lb = a.u_lb[i % NU], ub_ = a.u_ub[i % NU];
      if (!a.u_init && t_abs(lb) < T(1e19) && t_abs(ub_) < T(1e19)) u0 = T(0.5) * (lb + ub_);
      u0 = t_min(t_max(u0, lb), ub_);
    }
    s.ub[1][i] = u0;
  }
  bool warm = false;
  if (a.u_init != nullptr) {
    ldsRed[s.lane] = umax;
    __syncthreads();
    for (int l = 0; l < 64; ++l) warm = warm || !(ldsRed[l] == T(0));
  }
  __syncthreads();
  coarse = CSW && a.steps_per_grid > 1 && !warm && a.max_iter > 8 && a.n_grid >= LFSD_COARSE_MIN_GRID;
  if (coarse) {
    constexpr int TC0 = (NX + (NU > NP ? NU : NP) > 16) ? (LFSD_COARSE_TIME) : 1;      // (the small models: one RK4 step per interval only, measured)
    for (int f = TC0; f > 1; f /= 2) {
      if (f <= Lay::SMAX && N_full % f == 0 && N_full / f >= LFSD_COARSE_TIME_MIN) { tc = f; break; }
    }
    s.N = N_full / tc; s.dgrid = dgrid_full * T(tc);
    s.S = ((LFSD_COARSE_TIME_S) > 0 && (LFSD_COARSE_TIME_S) < tc) ? (LFSD_COARSE_TIME_S) : tc; s.DT = s.dgrid / T(s.S);
  }
  J = s.rollout_alphas(1, false, alpha_l);             // (every lane rolls the same controls out; lane 0's copy is adopted)
  ldsRed[s.lane] = J;
  __syncthreads();
  J = ldsRed[0];
  __syncthreads();
  if (coarse && !t_finite(J)) {                          // the coarse grid cannot even integrate the initial guess: reference grid
    coarse = false; tc = 1; s.N = N_full; s.dgrid = dgrid_full;      // (the initial guess in buffer 1 is zero / the midpoint of the bounds on every interval)
    s.S = a.steps_per_grid; s.DT = s.dgrid / T(s.S);
    J = s.rollout_alphas(1, false, alpha_l);
    ldsRed[s.lane] = J;
    __syncthreads();
    J = ldsRed[0];
    __syncthreads();
  }
  s.adopt_alpha(0, 0);
  s.linearise_parallel(0);
  J_feas = J; J_ref = J;
  status = t_finite(J) ? ST_RUNNING : ST_FAILED;
  }
  if (resuming) {
    const T* w = wstate;
    int q = 0;
    auto gi = [&]() LFSD_LAMBDA_INLINE { return (int)w[q++]; };
    cur = gi(); coarse = gi() != 0; relin = gi() != 0; tc = gi(); s.N = gi(); s.S = gi(); ms = gi() != 0; ms_check = gi() != 0; ms_floor = gi() != 0;
    n_acc_need = gi(); n_ms = gi(); n_half = gi(); mode = gi(); ham_ok = gi() != 0; hess_ok = gi() != 0; gn_crawl = gi() != 0; costates_ok = gi() != 0;
    it = gi(); n_acc = gi(); mu_hold = gi();
    const bool has_gap = gi() != 0;
    g1c = w[q++]; g2c = w[q++]; gmc = w[q++]; J = w[q++]; J_feas = w[q++]; mu = w[q++];
    g_flat = w[q++]; g_last = w[q++]; dec_last = w[q++]; J_ref = w[q++]; mu_bad = w[q++];
    s.dgrid = dgrid_full * T(tc); s.DT = s.dgrid / T(s.S);
    s.gap = has_gap ? s.gapp(cur) : nullptr;
    status = ST_RUNNING;
  }
  const int mu_hold_need = LFSD_MU_HOLD;
  bool suspended = false;
#if defined(LFSD_TEST_REFUSE_GAPPED)
  int n_test_refused = 0;
#endif
  for (; it < a.max_iter && status == ST_RUNNING; ++it) {
    if constexpr (W == 1) {
      // two-launch solves: once enough trajectories of the launch are finished for the rest to have a workgroup of several
      // wavefronts each, the rest park their state and leave (uniform per workgroup: every lane loads the same word, the first lane's value counts)
      if (a.sched != nullptr && !resuming) {
        bool susp;
        if (a.suspend_it >= 0) susp = it >= a.suspend_it;
        else susp = sched_load_uniform(a.sched) >= a.suspend_at;
        if (__builtin_expect(susp, 0)) { suspended = true; break; }      // (the state is parked behind the loop: nothing of it is kept alive longer for that)
      }
    }
    // (... and the LAST iteration of a solve whose iterate still has gaps -- the iteration limit after sweeps that kept failing while
    //  the gaps were to be closed -- takes the same path: what is returned is always a trajectory of the reference's discretisation,
    //  the controls rolled out open loop with the cost and the costates of THAT trajectory, as the reference returns IPOPT's last
    //  iterate; round 5 returned the node states of the lifted iterate beside a cost no feasible trajectory has)
    const bool close_last = s.gap != nullptr && it + 1 >= a.max_iter;
    if ((coarse && (relin || it + 4 >= a.max_iter)) || close_last) {
      // leave the coarse grid: the same controls rolled out (open loop) and linearised on the reference's discretisation; an
      // iteration without a sweep.  Every convergence test below only ever passes on this grid.
      // (a coarse phase with merged intervals first hands over to the full control grid with ONE RK4 step per
      //  interval -- the lean kernels' coarse level -- and leaves that one by the same rules)
      const bool to_mid = coarse && tc > 1 && a.steps_per_grid > 1 && it + 8 < a.max_iter;
      coarse = to_mid; relin = false;
#if defined(LFSD_OC_CLOCK)
      if (!to_mid) { wck_exit = clock64() - wck_t0; wck_it_exit = it; }
#endif
      if (tc > 1) {
        // prolongation: control k of the reference grid = control k / tc of the coarse one (in place: sources into registers first)
        // (chunks of 64 from the back: the source of element i is at an index <= i, so no chunk overwrites a later chunk's source)
        T* uc = s.ubp(cur);
        const int tot = N_full * NU;
        for (int c0 = ((tot - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
          const int i = c0 + s.lane;
          const T keep = (i < tot) ? uc[((i / NU) / tc) * NU + (i % NU)] : T(0);
          __syncthreads();
          if (i < tot) uc[i] = keep;
          __syncthreads();
        }
        s.N = N_full; s.dgrid = dgrid_full; tc = 1;
      }
      s.S = to_mid ? 1 : a.steps_per_grid; s.DT = s.dgrid / T(s.S);
      const T Jr = s.rollout_alphas(cur, false, alpha_l);
      ldsRed[s.lane] = Jr;
      __syncthreads();
      J = ldsRed[0];
      __syncthreads();
      s.adopt_alpha(0, cur ^ 1);
      s.linearise_parallel(cur ^ 1);
      cur ^= 1;
      hess_ok = false; costates_ok = false;
      g_last = T(-1); dec_last = T(1e30); g_flat = T(-1); J_ref = J; n_acc = 0;
      ms = ms_on; g1c = T(0); g2c = T(0); gmc = T(0); s.gap = nullptr; J_feas = J;      // (a roll-out: the new level starts without gaps)
      if (!t_finite(J)) { if (coarse) { relin = true; continue; } status = ST_FAILED; break; }      // (the mid level cannot integrate it: the reference's grid decides)
      continue;
    }
    if (ms_check) {
      // first test after the gaps of a finished multiple-shooting iterate were closed: the gradient of the closed trajectory from a
      // costate sweep alone (no gains, no Hessians)
      ms_check = false;
      T gn;
      LFSD_WCK(2, gn = s.costate_sweep(cur));
      costates_ok = true;
      if (gn < a.tol * (T(1) + t_abs(J))) { gnorm = gn; if (coarse) { relin = true; continue; } status = ST_CONVERGED; break; }
    }
    if (EXACT && a.exact_after >= 0 && mode < 2 && (it >= a.exact_after || gn_crawl)) mode = 2;
    if (EXACT && mode == 2 && !hess_ok) {
      LFSD_WCK(2, s.costate_sweep(cur));
      LFSD_WCK(3, s.hessians_parallel(cur));
      hess_ok = true;
    }
    s.reuse_hess = EXACT && mode == 2;
    T dmin = T(0);
    bool bw_ok;
    if constexpr (std::remove_reference<decltype(s)>::type::SMALL_BW) {
      LFSD_WCK(4, bw_ok = s.small_bw_fits() ? s.backward_small(cur, mode, mu, gnorm, dV1, dV2, dmin) : s.backward(cur, mode, mu, gnorm, dV1, dV2, dmin));
    } else {
      LFSD_WCK(4, bw_ok = s.backward(cur, mode, mu, gnorm, dV1, dV2, dmin));
    }
    costates_ok = bw_ok;                              // (a sweep that failed stopped at the failing stage)
    if (!bw_ok) {
      if (mode == 1) { mode = 0; ham_ok = false; }
      else {
        mu_bad = mu; mu_hold = 0;
        if (mu == T(0) && mode >= 1 && t_finite(dmin)) mu = t_min(t_max(T(-2) * dmin, T(1e-4)), T(1e6));
        else mu = t_max(mu * T(LFSD_MU_UP), mode == 2 ? T(1e-4) : T(1e-6));
        if (mu > T(1e12)) status = ST_FAILED;
      }
      continue;
    }
    // ---- the step.  With the multiple-shooting iteration enabled (`ms`) the CHEAP step is tried first: the full Newton step of the
    // lifted problem (linear forward pass, every interval re-integrated and linearised in parallel: OcWide::ms_forward / ms_trial),
    // accepted on the merit function below.  It leaves an iterate with gaps.  When it is refused -- or the iterate has to be closed
    // because its level is done -- the step is the closed-loop NONLINEAR roll-out of the 16 step lengths around the node states
    // (feed-forward and gains of the same gap-aware sweep: the roll-out closes every gap by construction, as a feasibility-driven
    // DDP step does), judged against the merit value of the iterate it starts from.  An iterate without gaps is a single-shooting
    // nominal: the convergence tests of this kernel apply to it and to nothing else. ----
    const T epsT = Eps<T>::v();
    const bool have_gaps = ms && g1c > T(0);
    bool close_now = have_gaps && it + 3 >= a.max_iter, ms_off = false;
    T phi0 = J, pred_ms = T(0);
    if (ms) {
      const T lam_mx = s.lam_max;
      // stationary, and the gaps closed to the resolution of the cost (a gap d costs lambda^T d to first order): this level is done
      if (have_gaps && !close_now && gnorm < a.tol * (T(1) + t_abs(J)) && lam_mx * g1c <= T(8) * epsT * (T(1) + t_abs(J))) {
        if (coarse) { relin = true; continue; }
        close_now = true; ms_floor = true; ms_off = true;
      }
      if (!close_now) {
        T lamd0, lamabs0, Dlin;
        LFSD_WCK(5, Dlin = s.ms_forward(cur, lamd0, lamabs0));
        // Merit function of the multiple-shooting step: the augmented Lagrangian of the lifted NLP with the costates of THIS iterate,
        //     m = J + lambda^T d + rho/2 |d|^2,   rho = max(|lambda|_inf, 1)
        // (lambda^T d is what closing the gaps costs to first order).  The full Newton step closes the linearised gaps entirely:
        // m'(0) = Dlin - lambda^T d - rho |d|^2.  Measured against it (128 robot-arm seeds, CPU emulator, cost in units of 10^6
        // clocks with the GPU's per-phase clocks: mean / slowest): this function 13.6 / 23.9; no penalty at all 14.0 / 21.5; the l1
        // penalty with one weight |lambda_i| per constraint x 0.1 / 0.25 / 1: 16.2 / 18.7 / 27.9 (single shooting: 29.3 / 39.6) -- a
        // function that counts every gap as a cost refuses most of the steps that work.  Two safeguards make up for its optimism
        // about gaps of the "right" sign: a step is only taken to an iterate whose cost is not above the last CLOSED trajectory's
        // (`J_feas`), and the roll-outs that close an iterate are judged against that closed cost, not against an estimate.
        const T rho = t_max(lam_mx, T(1));
        const T dl0 = Dlin - lamd0;
        pred_ms = rho * g2c - dl0;                     // first-order decrease of the merit function along the full step
        phi0 = J + lamd0 + T(0.5) * rho * g2c;
        (void)lamabs0;
        if (!t_finite(pred_ms) || !(pred_ms > T(2) * epsT * t_abs(phi0))) {
          // nothing the merit function resolves is left to gain: an iterate with gaps is closed now (and the single-shooting tests
          // take over for good); one without gaps IS a single-shooting nominal, the tests below decide
          if (have_gaps) {
            if (coarse) { relin = true; continue; }
            close_now = true; ms_floor = t_finite(pred_ms); ms_off = true;
          }
        } else {
          T Jt, lat, ldt, g2t, g1t, gmt;
          LFSD_WCK(6, s.ms_trial(cur, cur ^ 1, T(1), Jt, lat, ldt, g2t, g1t, gmt));
          T phit = Jt + ldt + T(0.5) * rho * g2t;
          const T flat = T(8) * epsT * t_abs(phi0);
          bool accept = t_finite(phit) && t_finite(g1t) && (phi0 - phit) >= T(1e-4) * pred_ms - flat && phit < phi0 &&
                        Jt <= J_feas + T(8) * epsT * t_abs(J_feas);
          bool half = false;
          if (!accept && n_half < 1) {
            // ONE shorter step along the same linear direction before the closed-loop roll-outs (an item of 25 k clocks against
            // 580 k): half the Newton step, same tests; it leaves half of the old gaps, so no second one follows it directly
            LFSD_WCK(6, s.ms_trial(cur, cur ^ 1, T(0.5), Jt, lat, ldt, g2t, g1t, gmt));
            phit = Jt + ldt + T(0.5) * rho * g2t;
            accept = t_finite(phit) && t_finite(g1t) && (phi0 - phit) >= T(0.5e-4) * pred_ms - flat && phit < phi0 &&
                     Jt <= J_feas + T(8) * epsT * t_abs(J_feas);
            half = accept;
          }
#if defined(LFSD_TRACE)
          if (s.lane == 0 && traj == 0) printf("wide ms it %d mode %d g %.6e J %.12e gap1 %.4e gapmax %.3e lam %.3e rho %.3e phi0 %.10e pred %.4e accept %d mu %g -> J %.12e gap1 %.4e phi %.10e\n", it, mode, (double)gnorm, (double)J, (double)g1c, (double)gmc, (double)lam_mx, (double)rho, (double)phi0, (double)pred_ms, (int)accept, (double)mu, (double)Jt, (double)g1t, (double)phit);
#endif
          if (accept) {
            const T mu_taken = mu;
            const T gain = phi0 - phit;
            cur ^= 1;
            g_last = gnorm; dec_last = T(1e30);
            hess_ok = false; costates_ok = false;
            // (no hold on the way down after a multiple-shooting step: a shift that turns out too small costs one backward sweep on
            //  cached Hessians, a held rung costs a whole iteration -- 128 seeds, emulator: mean 21.1 -> 19.8, slowest 33 -> 30
            //  iterations; dropping by 10 after steps that gain half their prediction: one seed at 52)
            if (!half) { mu = (mu > T(1e-8)) ? mu * T(LFSD_MU_DOWN) : T(0); mu_hold = 0; }
            n_half = half ? n_half + 1 : 0;
            if (mode == 0 && ham_ok && gain < T(LFSD_HAM_SWITCH) * t_abs(Jt)) mode = 1;
            else if (mode == 0 && !ham_ok && gain < T(1e-2) * t_abs(Jt)) gn_crawl = true;
            if (coarse && gain < T(LFSD_COARSE_SWITCH) * t_abs(Jt)) {
              if ((LFSD_COARSE_EXIT_RULE) <= 1 || ((LFSD_COARSE_EXIT_RULE) == 2 && mu_taken <= T(LFSD_COARSE_EXIT_MU))) relin = true;
            }
            J = Jt; g1c = g1t; g2c = g2t; gmc = gmt;
            s.gap = (g1c > T(0)) ? s.gapp(cur) : nullptr;
            ++n_ms;
            continue;
          }
        }
      }
    }
    if (!have_gaps) {
      if (gnorm < a.tol * (T(1) + t_abs(J))) { if (coarse) { relin = true; continue; } status = ST_CONVERGED; break; }
      if (at_working_precision(mode, mu, gnorm, g_last, dec_last, dV1, dV2, J, a.tol)) {
        if (coarse) { relin = true; continue; }
        status = ST_STALLED; ++it; break;      // (this iteration's sweep counts)
      }
      // first sweep after a multiple-shooting iterate that was closed because its Newton step could gain nothing the merit function
      // resolves: when the Newton-like step of the closed trajectory predicts a decrease below the resolution of the cost as well,
      // TWO accepted steps that gain nothing the cost resolves end the solve as "at working precision" (the rule below waits for
      // four; flat problems -- pendulum, robot arm -- keep their gradient tests: no step is skipped on the prediction alone).
      // (Round 5: ONE.  On the robot arm's flat valley the cost resolves nothing while the states are still 3e-4 from the KKT point:
      //  1 024 seeds at theta_1, fp32 against fp64, state error median / 90th percentile 3.2e-4 / 1.2e-3 with one, 1.6e-4 / 9.3e-4 with
      //  two and the closing roll-out's gradient history reset below; the single-shooting build 8.5e-5 / 7.7e-4; cold solve 9.0 -> 9.4 ms:
      //  profiles/r06_k_robotarm_accuracy_ab.txt)
      if (ms_floor) {
        ms_floor = false;
        if (mode >= 1 && mu <= T(0.1) && -(dV1 + dV2) <= T(2) * epsT * t_abs(J)) n_acc_need = 2;      // (a shift left over from the ladder's way down: small against Q_uu)
      }
    }
    // all step lengths at once; the largest one that passes the Armijo test is taken
    T Ja;
    LFSD_WCK(0, Ja = s.rollout_alphas(cur, true, alpha_l));
    ldsRed[s.lane] = Ja;
    __syncthreads();
    int ia = -1, ib = -1;
    const T Jr = have_gaps ? J_feas : J;                // roll-outs that close an iterate with gaps: progress is measured from the last CLOSED trajectory
    T Jmin = Jr, aa = T(1);
    const T flat = T(8) * epsT * t_abs(Jr);
    const bool flat_full = t_finite(ldsRed[0]) && t_abs(ldsRed[0] - Jr) <= T(64) * epsT * t_abs(Jr);
    // (a ROLLED loop that only picks indices; the costs are read back at the indices it picked.  Round 6: fully unrolled, with the
    //  costs carried in selects beside the indices, one build of this kernel -- rocket, fp32, one wavefront per trajectory -- adopted
    //  the right roll-out and kept the OLD cost whenever a step shorter than the full one was taken; the same source with a printf in
    //  it, with another scheduler or with an empty asm statement elsewhere did not (profiles/r06_f_wide_stale_cost.txt).  The cause,
    //  found later on a build of THIS form of the loop: the register allocator's spills of J, mu, dV1, ... placed before the exec
    //  restore behind adopt_alpha's copy loop, so that lanes >= 3 N kept the previous iterate's values -- nothing this loop's shape
    //  decides (profiles/r06_v_spill_before_exec_restore.txt).  Every build is scanned for that placement (lfsd_amd/isa_check.py),
    //  and the GPU tier solves the rocket under every launch scheme and compares bit for bit.)
#pragma unroll 1
    for (int l = 0; l < NAL; ++l) {
      const T Jl = ldsRed[l];
      const T expected = have_gaps ? T(0) : -(aa * dV1 + aa * aa * dV2);
      const bool okl = t_finite(Jl) && ((Jr - Jl) >= T(1e-4) * expected - flat) && (Jl < Jr);
      if (okl && ia < 0) ia = l;
      if (t_finite(Jl)) { Jmin = t_min(Jmin, Jl); if (ib < 0 || Jl < ldsRed[ib]) ib = l; }
      aa *= T(0.5);
    }
    T Jn = (ia >= 0) ? ldsRed[ia] : Jr, Jb = (ib >= 0) ? ldsRed[ib] : T(0);
    __syncthreads();
    if (have_gaps && ia >= 0) { ia = ib; Jn = Jb; }     // (closing an iterate: the cheapest of the roll-outs below the last closed cost, not the longest)
#if defined(LFSD_TEST_REFUSE_GAPPED)      // test hook (tests/test_emu_kernels.py): the first LFSD_TEST_REFUSE_GAPPED roll-outs from an iterate WITH gaps are refused
    if (have_gaps && !close_now && n_test_refused < (LFSD_TEST_REFUSE_GAPPED)) { ia = -1; ++n_test_refused; }
#endif
    bool accept = ia >= 0;
    if (!accept) {
      if (mode >= 1 && flat_full && (g_flat < T(0) || gnorm < T(0.7) * g_flat)) {
        accept = true; ia = 0; Jn = ldsRed[0]; g_flat = gnorm;      // Newton-like step below rounding noise, gradient still contracting
      } else if (close_now) {
        // (the gaps of a finished iterate have to go whatever the roll-outs cost: the cheapest one)
        if (ib >= 0) { accept = true; ia = ib; Jn = Jb; } else status = ST_FAILED;
      } else if (mode == 1) {
        mode = 0; ham_ok = false;
      } else if (mu > T(1e10) || ((Jr - Jmin) <= T(8) * epsT * t_abs(Jr) && ((mode == 0 && !BND) || flat_full || mu > T(1e6)))) {
        // (with a box on the controls the clamped closed loop can fail every step length although the Gauss-Newton
        //  direction is a descent direction of the unclamped model: that is a reason to shorten the step, not to stop)
        if (coarse) relin = true;
        else if (!have_gaps) status = ST_STALLED;
        else if (ib >= 0) { accept = true; ia = ib; Jn = Jb; ms_off = true; }      // (an iterate with gaps cannot be the answer: close it, single shooting from here)
        else status = ST_FAILED;
      } else {
        mu_bad = mu; mu_hold = 0;
        mu = t_max(mu * T(LFSD_MU_UP), mode == 2 ? T(1e-4) : T(1e-6));
      }
    }
#if defined(LFSD_TRACE)
    if (s.lane == 0 && traj == 0) printf("wide it %d st %d mode %d g %.6e J %.12e (ref %.10e gaps %d close %d) ia %d accept %d mu %g dV1 %.4e dV2 %.4e Jmin %.12e\n", it, status, mode, (double)gnorm, (double)J, (double)Jr, (int)have_gaps, (int)close_now, ia, (int)accept, (double)mu, (double)dV1, (double)dV2, (double)Jmin);
#endif
    if (accept) {
      const T mu_taken = mu;
      s.adopt_alpha(ia, cur ^ 1);
      LFSD_WCK(1, s.linearise_parallel(cur ^ 1));
      cur ^= 1;
      // (the roll-out that CLOSES an iterate with gaps is no Newton step: that the gradient did not contract over it says nothing
      //  about convergence -- round 5 let the "gradient stopped contracting" test end such solves one sweep later, 32x above the
      //  gradient tolerance, with the states still a gap's width from the KKT point: profiles/r06_k_robotarm_accuracy_ab.txt)
      g_last = have_gaps ? T(-1) : gnorm; dec_last = (mode >= 1 && mu == T(0) && !have_gaps) ? -(dV1 + dV2) : T(1e30);
      hess_ok = false; costates_ok = false;
      g1c = T(0); g2c = T(0); gmc = T(0); s.gap = nullptr; n_half = 0;      // a roll-out has no gaps
      if (ms_off) { ms = false; ms_check = true; }
      if (ia == 0) {
        // (a ratio-tested faster descent of the shift was measured on the rocket and needs MORE iterations in every variant --
        //  a shift that falls faster fails the next factorisation more often: profiles/HISTORY.md, r04_e_rocket_gain_ratio_ab.txt)
        const T mu_next = (mu > T(1e-8)) ? mu * T(LFSD_MU_DOWN) : T(0);
        if (mu > T(0) && mu_bad >= T(0) && mu_next <= mu_bad && mu_hold < mu_hold_need) ++mu_hold;
        else { mu = mu_next; mu_hold = 0; }
        if (mode == 0 && ham_ok && (Jr - Jn) < T(LFSD_HAM_SWITCH) * t_abs(Jn)) mode = 1;
        else if (mode == 0 && !ham_ok && (Jr - Jn) < T(1e-2) * t_abs(Jn)) gn_crawl = true;
      }
      // past the big drops: the reference's grid.  LFSD_COARSE_EXIT_RULE 0: any accepted step that gains less than the switch;
      // 1: full steps only (a short step through a badly modelled patch gains little too, and is no sign of convergence);
      // 2: full steps at a shift below LFSD_COARSE_EXIT_MU; 3: never by the gain (only the convergence tests above)
      if (coarse && (Jr - Jn) < T(LFSD_COARSE_SWITCH) * t_abs(Jn)) {
        if ((LFSD_COARSE_EXIT_RULE) == 0 || ((LFSD_COARSE_EXIT_RULE) == 1 && ia == 0) ||
            ((LFSD_COARSE_EXIT_RULE) == 2 && ia == 0 && mu_taken <= T(LFSD_COARSE_EXIT_MU))) relin = true;
      }
      J = Jn;
      J_feas = J;
      if (have_gaps) { J_ref = J; n_acc = 0; }             // (the stall test compares costs of roll-outs)
      else if (++n_acc >= n_acc_need) {
        if (J_ref - J <= T(16) * epsT * t_abs(J)) { if (coarse) relin = true; else status = ST_STALLED; }
        J_ref = J; n_acc = 0; n_acc_need = 4;
      }
    }
  }
  // The parked state: every variable above, as numbers of the solve's own type (the integers are small: exact).  `it` is where
  // the iteration loop continues; the buffers the state refers to (both nominal buffers, linearisations, gains, cached stage
  // Hessians, costates, gaps) already live in the workspace / the output rows.
  if (__builtin_expect(suspended, 0)) {
    if (threadIdx.x == 0) {
      T* w = wstate;
      const T iv[] = {T(cur), T(coarse), T(relin), T(tc), T(s.N), T(s.S), T(ms), T(ms_check), T(ms_floor), T(n_acc_need), T(n_ms), T(n_half),
                      T(mode), T(ham_ok), T(hess_ok), T(gn_crawl), T(costates_ok), T(it), T(n_acc), T(mu_hold), T(s.gap != nullptr)};
      const T fv[] = {g1c, g2c, gmc, J, J_feas, mu, g_flat, g_last, dec_last, J_ref, mu_bad};      // (gnorm, dV1, dV2: outputs of the sweep every iteration starts with)
      constexpr int NI = sizeof(iv) / sizeof(T), NF = sizeof(fv) / sizeof(T);
      static_assert(NI + NF <= Lay::WIDE_STATE, "parked solver state");
#pragma unroll
      for (int i = 0; i < NI; ++i) w[i] = iv[i];
#pragma unroll
      for (int i = 0; i < NF; ++i) w[NI + i] = fv[i];
      a.iters[traj] = it; a.status[traj] = ST_RUNNING;
      // the hand-over list of the launch: the next launch runs one workgroup per entry (sched[1] counts them)
      a.sched[2 + sched_add(a.sched + 1, 1)] = (int)traj;
    }
    return;
  }
  if (status == ST_RUNNING) status = ST_MAXITER;
#if defined(LFSD_MS_STATS)      // development aid (tools/ms_dev.py): iterations and multiple-shooting steps per trajectory
  if (threadIdx.x == 0) printf("msstat %d %d %d\n", (int)blockIdx.x, it, n_ms);
#endif
#if defined(LFSD_OC_CLOCK)
#ifndef LFSD_OC_CLOCK_TOTAL
#define LFSD_OC_CLOCK_TOTAL (1LL << 62)
#endif
  if (threadIdx.x == 0 && (it >= LFSD_OC_CLOCK || blockIdx.x == 0 || clock64() - wck_t0 >= (long long)(LFSD_OC_CLOCK_TOTAL)))
    printf("wide clock traj %d: iterations %d (%d multiple-shooting steps) total %lld rollout_alphas %lld linearise %lld costates %lld hessians %lld backward %lld ms_forward %lld ms_trial %lld | left the coarse grid at iteration %d, clock %lld\n",
           (int)blockIdx.x, it, n_ms, clock64() - wck_t0, wck[0], wck[1], wck[2], wck[3], wck[4], wck[5], wck[6], wck_it_exit, wck_exit);
#endif
#undef LFSD_WCK
  if (!costates_ok) s.costate_sweep(cur);                 // costates of the final nominal
  __syncthreads();
  {
    T* xo = a.state_grid + traj * (N + 1) * NX;
    T* uo = a.control_grid + traj * (N + 1) * NU;
    for (int i = s.lane; i < (N + 1) * NX; i += 64) xo[i] = s.xbp(cur)[i];
    for (int i = s.lane; i < (N + 1) * NU; i += 64) uo[i] = s.ubp(cur)[(i < N * NU) ? i : i - NU];
    if (s.lane == 0) { a.cost[traj] = J; a.iters[traj] = (status == ST_CONVERGED) ? it + 1 : it; a.status[traj] = status; }
    if (a.sched != nullptr && threadIdx.x == 0) sched_add(a.sched, 1);      // one more trajectory of the launch finished
  }
}


// =====================================================================================
//  fp64 solve seeded by the fp32 solve of the same problem (lfsd_capi.cpp, coc_solve_seeded): conversions between the two
// =====================================================================================
struct CastArgs { const void* src; void* dst; long long n; int to_f32; };
template <typename Tag> __global__ void cast_kernel(CastArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  if (a.to_f32) ((float*)a.dst)[i] = (float)((const double*)a.src)[i];
  else ((double*)a.dst)[i] = (double)((const float*)a.src)[i];
}

// controls of the fp32 solve [B][N+1][NU] -> initial guess of the fp64 solve [B][N][NU]; a trajectory the fp32 solve FAILED
// on (or whose cost is not finite) gets the caller's own row (`uc`, may be NULL) or else the all-zero row, which the kernels read
// as "cold start" (mesh continuation and all)
struct SeedArgs { const float* u32; const float* cost32; const int* status32; const double* uc; double* u0; int batch, n_grid, nu; };
template <typename Tag> __global__ void seed_controls_kernel(SeedArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long row = (long long)a.n_grid * a.nu;
  if (i >= (long long)a.batch * row) return;
  const long long b = i / row, j = i % row;
  const bool usable = a.status32[b] != ST_FAILED && t_finite(a.cost32[b]);
  a.u0[i] = usable ? (double)a.u32[b * (row + a.nu) + j] : (a.uc ? a.uc[i] : 0.0);
}

struct AddItersArgs { int* iters; const int* more; int batch; };
template <typename Tag> __global__ void add_iters_kernel(AddItersArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.batch) a.iters[i] += a.more[i];
}

}  // namespace lfsd
