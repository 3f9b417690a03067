// Auxiliary control system: Riccati and forward-sensitivity sweeps + waypoint loss (COCSys.auxSysSolver, CPDP.py:316-381).
// Part of the kernel sources collected by cpdp_kernels.h (include that header, not this one).
#pragma once
#include "cpdp_common.h"

namespace lfsd {

// =====================================================================================
//  Auxiliary control system (differentiated maximum principle)
// =====================================================================================
// (Measured and removed: the Z-dependent product fx^T Z of the Riccati right-hand side on the matrix cores -- +12 % -- and the stiff
//  update's Gram matrix by m lanes + an LDS hand-over -- +1.5 %: profiles/HISTORY.md, profiles/r03_mfma_aux.txt, r04_h_*.)

template <typename T> struct AuxArgs {
  int batch, n_grid, substeps;    // substeps = minimum coarse split-steps per grid interval (fine = 2x, Richardson)
  T rate_max;                     // refine an interval until  dt * |Huu^-1 fu^T P fu|_inf <= rate_max
  int max_refine;                 // cap on that refinement (factor over `substeps`)
  int unit_budget;                // > 0, for rows whose OC solve did NOT end converged / at working precision (oc_status given and not
                                  // 1 or 2): once such a trajectory has spent this many split units in a sweep its remaining intervals
                                  // run at `substeps` units without refinement (and count as accepted above rtol when they are).  Its
                                  // grids are not a KKT point, so its sensitivities are approximate whatever the sweeps do; and a row whose
                                  // parameters have left the well-posed region must not hold its launch at the cap of EVERY interval
  T rtol;                         // > 0: error-controlled sub-stepping -- an interval is redone with twice the units while the
                                  // Richardson estimate |fine - coarse| / 3 of a column exceeds rtol * (its magnitude + floor)
  const T* horizon;               // [B]
  const T* auxvar;                // [B][NP]
  const T* consts; int const_stride;
  const T* state_grid;            // [B][N+1][NX]
  const T* control_grid;          // [B][N+1][NU]
  const T* costate_grid;          // [B][N+1][NX]
  T* Z_grid;                      // [B][N+1][NX+NP][NX]   column-major Z = [P W]
  // forward sweep / loss
  int n_waypoints, n_iface;
  const int* iface_idx;           // [n_iface] state components the interface exposes; nullptr: the interface function compiled into
                                  // the model (Model::NIF outputs; n_iface must equal it)
  const T* taus;                  // [B][n_waypoints]
  const T* waypoints;             // [B][n_waypoints][n_iface]
  T* loss;                        // [B]
  T* grad;                        // [B][NP]
  T* auxX_grid;                   // [B][N+1][NP][NX] or nullptr  (dx/dtheta, column-major)
  T* auxU_grid;                   // [B][N+1][NP][NU] or nullptr
  int* stats;                     // [B][4] or nullptr: split units executed and intervals accepted ABOVE rtol, Riccati | forward sweep
  const int* oc_status;           // [B] or nullptr: status of the OC solve; rows whose status bit is set in skip_mask are not
  int skip_mask;                  // differentiated (no sweep, NaN loss / gradient): include/lfsd_cpdp.h, ABI 8
  LFSD_DEV bool skipped(long long traj) const { return oc_status != nullptr && ((skip_mask >> (oc_status[traj] & 31)) & 1) != 0; }
  LFSD_DEV int budget(long long traj) const {
    if (oc_status == nullptr || unit_budget <= 0) return 0;
    const int st = oc_status[traj];
    return (st == ST_CONVERGED || st == ST_STALLED) ? 0 : unit_budget;
  }
};

// LAY: which packing of the staged coefficients the kernel uses (codegen: 0 Riccati sweep, every matrix; 1 forward sweep,
// without Hxx / Hxe) -- the node stride NCOEF and every offset behind it follow
template <class M, int LAY = 0> struct AuxLayout {
  static constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NZ = NX + NP, NNODE = 5;
  static constexpr int NCOEF = M::NCOEF_L[LAY];
  static constexpr int LDS_L = 0;
  static constexpr int LDS_S = LDS_L + NNODE * NCOEF;
  static constexpr int LDS_T = LDS_S + NX * NU;
  // LDS_T (transposed exchange of the Riccati right-hand side) and LDS_KN/LDS_PSI (forward sweep) share one region:
  // each kernel uses only its own
  static constexpr int LDS_KN = LDS_T;                        // 3 stiff nodes x (NX x NU) feedback rows K^T
  static constexpr int LDS_PSI = LDS_KN + 3 * NX * NU;        // 3 stiff nodes x {phi1(h/4 K fu), phi1(h/2 K fu)}
  static constexpr int LDS_TSZ0 = (LAY == 0) ? NX * NZ : 3 * NX * NU + 6 * NU * NU;
  static constexpr int LDS_TSZ = LDS_TSZ0 > 128 ? LDS_TSZ0 : 128;      // (>= 2 x 64: the error reduction of the step control)
  static constexpr int LDS_E = LDS_T + LDS_TSZ;
                                                              // cold per-trajectory state: auxvar, consts,
  static constexpr int LDS_C = LDS_E + NP;                    // and the (x,u,lambda) grid values at both interval ends
  static constexpr int LDS_GA = LDS_C + M::NCX;               // [x_k u_k l_k]
  static constexpr int LDS_GB = LDS_GA + 2 * NX + NU;         // [x_k+1 u_k+1 l_k+1]
  static constexpr int LDS_END = LDS_GB + 2 * NX + NU;
  static constexpr int lds_elems() { return ((LDS_END + 3) / 4) * 4; }
  // forward kernel only, behind LDS_END:
  static constexpr int FWD_XPREV = LDS_END;                   // parking slot for X(t_k) (row i of column j at [i*NP + j])
  // ... per node and column, the X-independent part of the right-hand side (fe - fu Huu^-1 (fu^T W + Hue)) e_j
  static constexpr int FWD_BC = FWD_XPREV + NX * NP;
  // ... the P columns at both ends of the interval ([end][column][row], and one zero row that lanes without a P column
  // point at): they are needed three times per unit only, too cold for 2*NX registers per lane
  static constexpr int FWD_P = FWD_BC + NNODE * NX * NP;
  // ... the hand-over of the two Richardson chains, which run on different lanes ([chain][row][column])
  static constexpr int FWD_XCH = FWD_P + 2 * NX * NX + NX;
  // ... and the reductions of the step control (one word per lane, twice)
  static constexpr int FWD_RED = FWD_XCH + 2 * NX * NP;
  static constexpr int lds_elems_fwd() { return ((FWD_RED + 128 + 3) / 4) * 4; }
  // Riccati kernel only: one parking slot per lane for a column of Z (row i of lane l at [i*G + l]): of the unit's start value and
  // the coarse Richardson result only one has to be in registers at a time
  template <int G> static constexpr int ric_park() { return LDS_END; }
  // ... and a second one: the start value of a STEP of the step-size control inside a stiff interval (aux_riccati_kernel, "adaptive")
  // (compiled into the fp32 instantiations with lane groups of at most 16 lanes -- pendulum, robot arm, cart-pole: the small models,
  //  whose sweeps have registers to spare; the 32-lane sweeps of the quadrotor / rocket run two waves per SIMD at 252 registers, spilled
  //  60 B per lane with it (headline aux_riccati 1.50 -> 1.545 ms) and stay exactly what they were)
  template <int G> static constexpr bool ric_adaptive() { return G <= 16; }
  template <int G> static constexpr int ric_park2() { return ric_park<G>() + NX * G; }
  template <int G> static constexpr int lds_elems_ric() { return ((ric_park2<G>() + (ric_adaptive<G>() ? NX * G : 0) + 3) / 4) * 4; }
};

// Lanes per trajectory of the forward sweep.  Only the NP columns of X = dx/dtheta advance there; the NX columns of P are
// merely interpolated.  The two chains of a Richardson pair run side by side on the two halves of the lane group -- lanes
// j < NP carry column j through the two fine Strang steps, lanes G/2 + j carry the same column through the coarse one --
// and lanes < NX double as the owners of P column `lane` where the feedback gains are formed, so max(NX, 2 NP) lanes are
// needed (quadrotor: 16, four trajectories per wavefront; the Riccati sweep needs 32 for its NX+NP columns).  The m x m
// matrix functions of the three stiff nodes run row per lane on sub-groups of sub_lanes<NU>() lanes.
template <class M> constexpr int fwd_lanes() {
  int need = M::NX > 2 * M::NP ? M::NX : 2 * M::NP;
  if (need < 5) need = 5;                                      // AuxLayout::NNODE staging lanes
  if (need < 3 * sub_lanes<M::NU>()) need = 3 * sub_lanes<M::NU>();
  int g = 8;
  while (g < need) g *= 2;
  return g;
}

template <class M, typename T, int G, int LAY> struct AuxCtx {
  static constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NC = M::NC, NZ = NX + NP;
  using Lay = AuxLayout<M, LAY>;
  static constexpr int H = G / 2;      // forward sweep: first lane of the coarse Richardson chain
  int lane;
  int xcol;                  // forward sweep: column of X = dx/dtheta (and of W) this lane carries (fine chain: lanes < NP,
  bool xlane, coarse;        // coarse chain: lanes H .. H+NP-1), else 0
  const T *e, *c;            // [NP], [NC] in LDS
  const T *xa_, *ua_, *la_, *xb_, *ub_, *lb_;   // grid values at both ends of the interval, in LDS
  T t_a, dgrid;
  T* lds;
  T ox[NX], oe[NP];    // one-hot selectors of this lane's column
  // Riccati sweep: where this lane's column of [Hxx Hxe] (NX rows) and of [Hux Hue] (NU rows) sits in node 0 (structural
  // zeros point at the node's zero word): `y += H e_lane` is a gather of NX words, not a product with a one-hot vector
  const T* hcol[LAY == 0 ? NX : 1];
  const T* ucol[LAY == 0 ? NU : 1];

  // Grid values (x, u, lambda) of the two nodes of interval k into LDS.  Consecutive intervals share a node, and the sweeps walk
  // the grid in one direction (`dir` = +1 forward sweep, -1 Riccati sweep): only the NEW node is read from memory, and it is
  // fetched into registers ONE INTERVAL AHEAD (`gpf`, one or two words per lane) -- an interval is a few thousand clocks of work
  // on a wavefront that is (nearly) alone on its SIMD, and used to start by waiting a global round trip for 60 words.
  static constexpr int NR = 2 * NX + NU, GPF = (NR + G - 1) / G;
  static constexpr bool PF = sizeof(T) == 4;      // (fp64: the registers are not there -- aux_forward 2.03 -> 2.57 ms, aux_riccati 4.60 -> 4.78 with it)
  T gpf[GPF];
  T *gA_ = nullptr, *gB_ = nullptr;
  LFSD_DEV void fetch_node(const AuxArgs<T>& a, long long traj, int node, int N) {
    const T* xs = a.state_grid + (traj * (N + 1) + node) * NX;
    const T* us = a.control_grid + (traj * (N + 1) + node) * NU;
    const T* ls = a.costate_grid + (traj * (N + 1) + node) * NX;
#pragma unroll
    for (int j = 0; j < GPF; ++j) {
      const int idx = lane + j * G;
      gpf[j] = (idx < NX) ? xs[idx] : ((idx < NX + NU) ? us[idx - NX] : ((idx < NR) ? ls[idx - NX - NU] : T(0)));
    }
  }
  LFSD_DEV void load_interval(const AuxArgs<T>& a, long long traj, int k, int N, int dir) {
    LFSD_WAVE_SYNC();                         // previous interval's readers are done
    if (gA_ == nullptr || !PF) {              // first interval of the sweep: both nodes straight from memory
      gA_ = lds + Lay::LDS_GA; gB_ = lds + Lay::LDS_GB;
      const T* xs = a.state_grid + (traj * (N + 1) + k) * NX;
      const T* us = a.control_grid + (traj * (N + 1) + k) * NU;
      const T* ls = a.costate_grid + (traj * (N + 1) + k) * NX;
      for (int i = lane; i < NX; i += G) { gA_[i] = xs[i]; gB_[i] = xs[NX + i]; gA_[NX + NU + i] = ls[i]; gB_[NX + NU + i] = ls[NX + i]; }
      for (int i = lane; i < NU; i += G) { gA_[NX + i] = us[i]; gB_[NX + i] = us[NU + i]; }
    } else {                                  // the shared node changes ends, the new node comes out of the registers
      T* t_ = gA_; gA_ = gB_; gB_ = t_;
      T* dst = (dir > 0) ? gB_ : gA_;
#pragma unroll
      for (int j = 0; j < GPF; ++j) { const int idx = lane + j * G; if (idx < NR) dst[idx] = gpf[j]; }
    }
    xa_ = gA_; ua_ = gA_ + NX; la_ = gA_ + NX + NU; xb_ = gB_; ub_ = gB_ + NX; lb_ = gB_ + NX + NU;
    t_a = M::TIME_VARYING ? dgrid * T(k) : T(0);
    LFSD_WAVE_SYNC();
    const int nn = (dir > 0) ? k + 2 : k - 1;     // the node the NEXT interval adds
    if (PF && nn >= 0 && nn <= N) fetch_node(a, traj, nn, N);
  }
  // Lane `node` (< 5) evaluates the packed PMP coefficients at its own time node s (fraction of the
  // interval) on the reference's linear interpolant of (x,u,lambda) (CPDP.py:320-323) and stages them in LDS.
  LFSD_DEV void stage_nodes(T s_first, T s_step) {
    if (lane < Lay::NNODE) {
      const T s = s_first + s_step * T(lane);
      T x[NX], u[NU], l[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) { x[i] = xa_[i] + s * (xb_[i] - xa_[i]); l[i] = la_[i] + s * (lb_[i] - la_[i]); }
#pragma unroll
      for (int i = 0; i < NU; ++i) u[i] = ua_[i] + s * (ub_[i] - ua_[i]);
      T* L = lds + Lay::LDS_L + lane * Lay::NCOEF;
      M::template pmp_coeffs<LAY>(t_a + s * dgrid, x, u, l, e, c, L);
      if constexpr (!M::IHUU_CLOSED) {      // (a diagonal Huu is inverted in closed form by pmp_coeffs itself)
        T Huu[NU * NU], iH[NU * NU];
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Huu[i] = L[M::OFF_HUU_L[LAY] + i];
        mat_inverse<NU>(Huu, iH);       // casadi.pinv(ddHuu) of a nonsingular Huu (CPDP.py:262)
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) L[M::OFF_IHUU_L[LAY] + i] = iH[i];
      }
    }
    LFSD_WAVE_SYNC();
  }
  LFSD_DEV const T* node(int i) const { return lds + Lay::LDS_L + i * Lay::NCOEF; }

  // |Huu^-1 fu^T P fu|_inf : rate of the stiff closed-loop modes at one node (P = first NX lanes' columns)
  LFSD_DEV T stiff_rate(const T* zt, const T* L) {
    T* ldsS = lds + Lay::LDS_S;
    T s[NU], kj[NU];
    M::template fu_mulT<false, LAY>(L, zt, s);
    M::template ihuu_mul<LAY>(L, s, kj);
    if (lane < NX) {
#pragma unroll
      for (int a = 0; a < NU; ++a) ldsS[lane * NU + a] = kj[a];
    }
    LFSD_WAVE_SYNC();
    T Mx[NU * NU];
    M::template fu_gram<false, LAY>(L, ldsS, Mx);
    T nrm = T(0);
#pragma unroll
    for (int a = 0; a < NU; ++a) {
      T r = T(0);
#pragma unroll
      for (int b = 0; b < NU; ++b) r += t_abs(Mx[a * NU + b]);
      nrm = t_max(nrm, r);
    }
    LFSD_WAVE_SYNC();
    return nrm;
  }
  LFSD_DEV int units_for(T rate, int Sa, T rate_max, int max_refine) const {
    T want = rate * dgrid / rate_max;
    if (!t_finite(want)) want = T(Sa);
    int u = Sa;
    const long long cap = (long long)Sa * max_refine;
    while ((T)u < want && (long long)u * 2 <= cap) u *= 2;
    return u;
  }
  // ---- Riccati (backward in time; tau = -t) -------------------------------------------------
  // stiff sub-flow  dZ/dtau = -P R Z,  R = fu Huu^-1 fu^T :   Z <- Z - P fu (Huu/dt + fu^T P fu)^-1 fu^T Z
  LFSD_DEV void ric_stiff(T* z, const T* L, T dt) {
    T* ldsS = lds + Lay::LDS_S;
    T s[NU];
    M::template fu_mulT<false, LAY>(L, z, s);
    if (lane < NX) {
#pragma unroll
      for (int a = 0; a < NU; ++a) ldsS[lane * NU + a] = s[a];
    }
    LFSD_WAVE_SYNC();
    T Gm[NU * NU];
    const T idt = T(1) / dt;
#pragma unroll
    for (int i = 0; i < NU * NU; ++i) Gm[i] = L[M::OFF_HUU_L[LAY] + i] * idt;
    M::template fu_gram<true, LAY>(L, ldsS, Gm);
    lu_factor<NU>(Gm);
    lu_solve<NU>(Gm, s);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      T d = T(0);
#pragma unroll
      for (int a = 0; a < NU; ++a) d += ldsS[i * NU + a] * s[a];
      z[i] -= d;
    }
    LFSD_WAVE_SYNC();
  }
  // non-stiff part  dZ/dtau = [Qt qt] + A^T Z + P [A rt]   (A, Qt, rt, qt of CPDP.py:265-269)
  LFSD_DEV void ric_rhs(const T* z, int nd, T* y) {
    T* ldsT = lds + Lay::LDS_T;
    const T* L = node(nd);
    T s[NU], v[NU], w[NU], nv[NU], r[NP], wq[NU];
    {
      // (this lane's column of Huu^-1 [Hux Hue] is recomputed per right-hand side: with the midpoint rule's six per unit that is
      //  cheaper than parking it per node and lane -- measured 1.37 against 1.47 / 2.30 ms, profiles/HISTORY.md)
      T hu[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) hu[a] = ucol[a][nd * Lay::NCOEF];      // (Hux | Hue) e_lane
      M::template ihuu_mul<LAY>(L, hu, wq);
    }
    M::template fu_mulT<false, LAY>(L, z, s);
    M::template ihuu_mul<LAY>(L, s, v);
#pragma unroll
    for (int a = 0; a < NU; ++a) { nv[a] = -v[a]; w[a] = -(wq[a] + v[a]); }
    M::template fx_mulT<false, LAY>(L, z, y);
    if (lane < NX) {
      T tv[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) tv[i] = y[i];
      M::template Hxu_mul<true, LAY>(L, nv, tv);
      M::template fe_mulT<false, LAY>(L, z, r);
      M::template Hue_mulT<true, LAY>(L, nv, r);
#pragma unroll
      for (int i = 0; i < NX; ++i) ldsT[lane * NZ + i] = tv[i];
#pragma unroll
      for (int i = 0; i < NP; ++i) ldsT[lane * NZ + NX + i] = r[i];
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) y[i] += hcol[i][nd * Lay::NCOEF];      // [Hxx Hxe] e_lane
    M::template Hxu_mul<true, LAY>(L, w, y);
    LFSD_WAVE_SYNC();
    if (lane < NZ) {
#pragma unroll
      for (int i = 0; i < NX; ++i) y[i] += ldsT[i * NZ + lane];
    }
    LFSD_WAVE_SYNC();
  }
  // non-stiff RK4 step of length h over nodes (n0, n1, n2)
  LFSD_DEV void ric_rk4(T* z, int n0, int n1, int n2, T h) {
    T k[NX], acc[NX], zs[NX];
    if constexpr (aux_rk<T>() == 2) {
      // explicit midpoint rule: the Strang splitting around it is second order anyway, and in fp32 the rounding floor of
      // the sweep hides the two orders the Richardson pair gains with RK4 (cpdp_common.h, LFSD_AUX_RK32)
      (void)n2; (void)acc;
      ric_rhs(z, n0, k);
#pragma unroll
      for (int i = 0; i < NX; ++i) zs[i] = z[i] + T(0.5) * h * k[i];
      ric_rhs(zs, n1, k);
#pragma unroll
      for (int i = 0; i < NX; ++i) z[i] += h * k[i];
      return;
    }
    ric_rhs(z, n0, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] = k[i]; zs[i] = z[i] + T(0.5) * h * k[i]; }
    ric_rhs(zs, n1, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] += T(2) * k[i]; zs[i] = z[i] + T(0.5) * h * k[i]; }
    ric_rhs(zs, n1, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] += T(2) * k[i]; zs[i] = z[i] + h * k[i]; }
    ric_rhs(zs, n2, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) z[i] += h / T(6) * (acc[i] + k[i]);
  }
  // Strang step: stiff h/2, non-stiff h, stiff h/2
  LFSD_DEV void ric_strang(T* z, int n0, int n1, int n2, T h) {
    ric_stiff(z, node(n0), h * T(0.5));
    ric_rk4(z, n0, n1, n2, h);
    ric_stiff(z, node(n2), h * T(0.5));
  }
  // two Strang steps of length h/2 over nodes (0,1,2) and (2,3,4): the two adjacent stiff quarter-steps at the middle
  // node are the exact flow of the same frozen system, so they compose exactly into one half-step
  LFSD_DEV void ric_strang2(T* z, T h) {
    ric_stiff(z, node(0), h * T(0.25));
    ric_rk4(z, 0, 1, 2, h * T(0.5));
    ric_stiff(z, node(2), h * T(0.5));
    ric_rk4(z, 2, 3, 4, h * T(0.5));
    ric_stiff(z, node(4), h * T(0.25));
  }

  // ---- forward auxiliary state -----------------------------------------------------------------
  // Lane roles.  Lanes < NX own P column `lane` (both interval ends in LDS) wherever feedback gains are formed.  Lanes
  // j < NP carry X / W column j through the FINE chain of a Richardson pair (two Strang steps), lanes H + j carry the same
  // column through the COARSE one (one Strang step): the pair costs the instructions of the fine chain alone, and a lane
  // holds one chain's state, not two.
  // stiff sub-flow  X' = -fu K X,  K = Huu^-1 fu^T P (frozen over the sub-step), solved exactly:
  //   X(dt) = X - dt fu phi1(dt K fu) K X,   phi1(M) = M^-1 (I - e^-M)   (m x m matrix function).
  // (An A-stable rational step is not enough here: with a cheap control cost dt*|K fu| reaches O(10^2).)
  // fwd_prep: for the three stiff nodes (0, 2, 4) of a unit the P lanes gather K(t_node); then the matrix function of
  // each node (quarter and half step) is evaluated once per unit, ROW PER LANE on a sub-group of Q lanes per node
  // (phi1_neg_rows: the m x m products cost m FMAs + m DPP moves per lane instead of m^3 FMAs on one lane of the group).
  LFSD_DEV void fwd_prep(const T* zA, const T* zB, T s0, T ds, T hq) {
    T* ldsK = lds + Lay::LDS_KN;
    T* ldsP = lds + Lay::LDS_PSI;
    LFSD_FWD_NODE_LOOP
    for (int r = 0; r < 3; ++r) {
      const T* L = node(2 * r);
      const T sr = s0 + T(2 * r) * ds;
      T zt[NX], sv[NU], kj[NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) zt[i] = zA[i] + sr * (zB[i] - zA[i]);
      M::template fu_mulT<false, LAY>(L, zt, sv);
      M::template ihuu_mul<LAY>(L, sv, kj);
      if (lane < NX) {
#pragma unroll
        for (int a = 0; a < NU; ++a) ldsK[(r * NX + lane) * NU + a] = kj[a];
      }
    }
    LFSD_WAVE_SYNC();
    if constexpr (NU <= 4) {
      constexpr int Q = sub_lanes<NU>();
      // every lane of the group runs the same instructions (sub_bcast needs its sources active): sub-groups beyond the
      // third repeat node 4 and drop the result
      const int sg = lane / Q, a = lane % Q;
      const int r = sg < 2 ? sg : 2;
      const bool row = a < NU;
      const bool mine = sg < 3 && row;
      const int ar = row ? a : 0;
      T sa[NX], Mrow[NU], Pq[NU], Ph[NU];
#pragma unroll
      for (int i = 0; i < NX; ++i) sa[i] = ldsK[(r * NX + i) * NU + ar];
      M::template fu_mulT<false, LAY>(node(2 * r), sa, Mrow);                  // row a of K fu = fu^T (row a of K)
      T rs = T(0);
#pragma unroll
      for (int j = 0; j < NU; ++j) { Mrow[j] = row ? Mrow[j] * hq : T(0); rs += t_abs(Mrow[j]); }
      // one scaling for the three nodes (|.|_inf over all rows of the group): uniform trip count of the squaring loop
      T* red = lds + Lay::FWD_RED;
      red[lane] = mine ? rs : T(0);
      LFSD_WAVE_SYNC();
      T nrm = T(0);
#pragma unroll
      for (int l = 0; l < 3 * Q; ++l) nrm = t_max(nrm, red[l]);
      LFSD_WAVE_SYNC();
      int sq = 0;
      T sc = T(1);
      while (nrm * sc > T(0.25) && sq < 60) { sc *= T(0.5); ++sq; }
      phi1_neg_rows<NU, Q>(Mrow, a, sq, sc, Pq, Ph);
      if (mine) {
#pragma unroll
        for (int j = 0; j < NU; ++j) { ldsP[(r * 2) * NU * NU + a * NU + j] = Pq[j]; ldsP[(r * 2 + 1) * NU * NU + a * NU + j] = Ph[j]; }
      }
    } else {
      if (lane < 3) {
        T Mx[NU * NU], Pq[NU * NU], Ph[NU * NU];
        M::template fu_gram<false, LAY>(node(2 * lane), ldsK + lane * NX * NU, Mx);       // K fu
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) Mx[i] *= hq;
        phi1_neg<NU>(Mx, Pq, Ph);
#pragma unroll
        for (int i = 0; i < NU * NU; ++i) { ldsP[lane * 2 * NU * NU + i] = Pq[i]; ldsP[(lane * 2 + 1) * NU * NU + i] = Ph[i]; }
      }
    }
    LFSD_WAVE_SYNC();
  }
  // apply the prepared exact stiff step of stiff node r (0..2) over dt = hq (half == 0) or 2 hq (half == 1); r and half
  // may differ between the lanes (the two chains sit at different nodes)
  static constexpr bool FETCH = sizeof(T) == 4;      // (fp64: the registers are not there, cpdp_common.h)
  LFSD_DEV void fwd_stiff(T* xa, int r, int half, T dt) {
    if constexpr (!FETCH) {
      const T* Kn = lds + Lay::LDS_KN + r * NX * NU;
      const T* Psi = lds + Lay::LDS_PSI + (r * 2 + half) * NU * NU;
      T kx[NU], y[NU];
#pragma unroll
      for (int a = 0; a < NU; ++a) kx[a] = T(0);
#pragma unroll
      for (int i = 0; i < NX; ++i) {
#pragma unroll
        for (int a = 0; a < NU; ++a) kx[a] += Kn[i * NU + a] * xa[i];
      }
      matvec<NU>(Psi, kx, y);
#pragma unroll
      for (int a = 0; a < NU; ++a) y[a] *= -dt;
      M::template fu_mul<true, LAY>(node(2 * r), y, xa);
      return;
    }
    // gain rows, matrix function and fu of the node: fetched into registers with back-to-back reads, one wait (lds_fetch)
    T Kn[NX * NU], Psi[NU * NU], Lf[M::FU_N_L[LAY]];
    lds_issue<NX * NU>(lds + Lay::LDS_KN + r * NX * NU, Kn);
    lds_issue<NU * NU>(lds + Lay::LDS_PSI + (r * 2 + half) * NU * NU, Psi);
    lds_issue<M::FU_N_L[LAY]>(node(2 * r), Lf);
    lds_land<NX * NU>(Kn); lds_land<NU * NU>(Psi); lds_land<M::FU_N_L[LAY]>(Lf);
    T kx[NU], y[NU];
#pragma unroll
    for (int a = 0; a < NU; ++a) kx[a] = T(0);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
      for (int a = 0; a < NU; ++a) kx[a] += Kn[i * NU + a] * xa[i];
    }
    matvec<NU>(Psi, kx, y);
#pragma unroll
    for (int a = 0; a < NU; ++a) y[a] *= -dt;
    M::template fu_mul<true, LAY>(Lf, y, xa);
  }
  // non-stiff part  X' = fx X + fe - fu Huu^-1 (Hux X + Hue + fu^T W).  Its X-independent part
  //   b_j(t) = (fe - fu Huu^-1 (fu^T W(t) + Hue)) e_j
  // is evaluated once per staged node (fwd_cols) and parked per column; the right-hand sides of a unit then cost
  //   y = fx x - fu Huu^-1 Hux x + b_j.
  // Three rounds for the five nodes: the fine lanes of a column take nodes 0, 1, 2, its coarse lanes nodes 3, 4 (and 4 again).
  LFSD_DEV void fwd_cols(const T* zA, const T* zB, T s0, T ds) {
    T* bc = lds + Lay::FWD_BC;
    LFSD_FWD_NODE_LOOP
    for (int rd = 0; rd < 3; ++rd) {
      const int nd = coarse ? (rd < 2 ? 3 + rd : 4) : rd;
      const T* L = node(nd);
      const T sr = s0 + T(nd) * ds;
      T wt[NX], sv[NU], v[NU], b[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) wt[i] = zA[i] + sr * (zB[i] - zA[i]);
      M::template fu_mulT<false, LAY>(L, wt, sv);
      M::template Hue_mul<true, LAY>(L, oe, sv);
      M::template ihuu_mul<LAY>(L, sv, v);
#pragma unroll
      for (int a = 0; a < NU; ++a) v[a] = -v[a];
      M::template fe_mul<false, LAY>(L, oe, b);
      M::template fu_mul<true, LAY>(L, v, b);
      if (xlane) {
#pragma unroll
        for (int i = 0; i < NX; ++i) bc[(nd * NX + i) * NP + xcol] = b[i];
      }
    }
  }
  LFSD_DEV void fwd_rhs(const T* xa, int nd, T* y) {
    if constexpr (!FETCH) {
      const T* L = node(nd);
      const T* bc = lds + Lay::FWD_BC + nd * NX * NP;     // lanes without an X column read column 0; their result is dropped
      T s[NU], v[NU];
      M::template Hxu_mulT<false, LAY>(L, xa, s);
      M::template ihuu_mul<LAY>(L, s, v);
#pragma unroll
      for (int a = 0; a < NU; ++a) v[a] = -v[a];
      M::template fx_mul<false, LAY>(L, xa, y);
      M::template fu_mul<true, LAY>(L, v, y);
#pragma unroll
      for (int i = 0; i < NX; ++i) y[i] += bc[i * NP + xcol];
      return;
    }
    // the node's fu, fx, Hxu, Huu^-1 (a prefix of the forward packing) and this column's b_j: registers first (lds_fetch)
    T L[M::RHS_N_L[LAY]], b[NX];
    lds_issue<M::RHS_N_L[LAY]>(node(nd), L);
    lds_issue_strided<NX>(lds + Lay::FWD_BC + nd * NX * NP + xcol, NP, b);     // lanes without an X column read column 0; their result is dropped
    lds_land<M::RHS_N_L[LAY]>(L); lds_land<NX>(b);
    T s[NU], v[NU];
    M::template Hxu_mulT<false, LAY>(L, xa, s);
    M::template ihuu_mul<LAY>(L, s, v);
#pragma unroll
    for (int a = 0; a < NU; ++a) v[a] = -v[a];
    M::template fx_mul<false, LAY>(L, xa, y);
    M::template fu_mul<true, LAY>(L, v, y);
#pragma unroll
    for (int i = 0; i < NX; ++i) y[i] += b[i];
  }
  // non-stiff RK4 step of length h over nodes (n0, n1, n2) -- per-lane values
  LFSD_DEV void fwd_rk4(T* xa, int n0, int n1, int n2, T h) {
    T k[NX], acc[NX], xs[NX];
    if constexpr (aux_rk<T>() == 2) {
      (void)n2; (void)acc;
      fwd_rhs(xa, n0, k);
#pragma unroll
      for (int i = 0; i < NX; ++i) xs[i] = xa[i] + T(0.5) * h * k[i];
      fwd_rhs(xs, n1, k);
#pragma unroll
      for (int i = 0; i < NX; ++i) xa[i] += h * k[i];
      return;
    }
    fwd_rhs(xa, n0, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] = k[i]; xs[i] = xa[i] + T(0.5) * h * k[i]; }
    fwd_rhs(xs, n1, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] += T(2) * k[i]; xs[i] = xa[i] + T(0.5) * h * k[i]; }
    fwd_rhs(xs, n1, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) { acc[i] += T(2) * k[i]; xs[i] = xa[i] + h * k[i]; }
    fwd_rhs(xs, n2, k);
#pragma unroll
    for (int i = 0; i < NX; ++i) xa[i] += h / T(6) * (acc[i] + k[i]);
  }
  // One split unit of length hc for this lane's chain:
  //   fine    stiff(node 0, hq)   RK(nodes 0 1 2, hc/2)  stiff(node 2, 2 hq)  RK(nodes 2 3 4, hc/2)  stiff(node 4, hq)
  //   coarse  stiff(node 0, 2 hq) RK(nodes 0 2 4, hc)    stiff(node 4, 2 hq)
  // (hq = hc/4; the two adjacent fine stiff quarter-steps at the middle node compose exactly into one half-step.)  The
  // first three operations are the same instructions with per-lane nodes and step lengths; the coarse lanes sit out the
  // last two.
  LFSD_DEV void fwd_chain(T* x, T hc, T hq) {
    const int c = coarse ? 1 : 0;
    fwd_stiff(x, 0, c, coarse ? T(2) * hq : hq);
    fwd_rk4(x, 0, 1 + c, 2 + 2 * c, coarse ? hc : hc * T(0.5));
    fwd_stiff(x, 1 + c, 1, T(2) * hq);
    if (!coarse) {
      fwd_rk4(x, 2, 3, 4, hc * T(0.5));
      fwd_stiff(x, 2, 0, hq);
    }
  }
  // auxiliary control at a grid point (CPDP.py:295):  U = -Huu^-1((Hux + fu^T P) X + fu^T W + Hue)
  LFSD_DEV void aux_control(const T* xa, const T* pt, const T* wt, const T* L, T* uo) {
    T* ldsS = lds + Lay::LDS_S;
    T s[NU];
    M::template fu_mulT<false, LAY>(L, pt, s);          // P role: row `lane` of (fu^T P)^T
    if (lane < NX) {
#pragma unroll
      for (int a = 0; a < NU; ++a) ldsS[lane * NU + a] = s[a];
    }
    LFSD_WAVE_SYNC();
    M::template fu_mulT<false, LAY>(L, wt, s);          // X role: s = fu^T w_j + Hue e_j + Hux x_j + fu^T P x_j
    M::template Hue_mul<true, LAY>(L, oe, s);
    M::template Hxu_mulT<true, LAY>(L, xa, s);
#pragma unroll
    for (int i = 0; i < NX; ++i) {
#pragma unroll
      for (int a = 0; a < NU; ++a) s[a] += ldsS[i * NU + a] * xa[i];
    }
    M::template ihuu_mul<LAY>(L, s, uo);
#pragma unroll
    for (int a = 0; a < NU; ++a) uo[a] = -uo[a];
    LFSD_WAVE_SYNC();
  }
};

template <class M, typename T, int G, int LAY> LFSD_DEV void aux_setup(AuxCtx<M, T, G, LAY>& s, const AuxArgs<T>& a, long long traj,
                                                            T* lds_all, int lds_stride) {
  constexpr int NX = M::NX, NP = M::NP, NC = M::NC;
  using Lay = AuxLayout<M, LAY>;
  const int gib = threadIdx.x / G;
  s.lane = threadIdx.x % G;
  s.coarse = (LAY == 1) && (s.lane >= G / 2);
  const int xl = s.coarse ? s.lane - G / 2 : s.lane;
  s.xlane = (LAY == 1) && (xl < NP);
  s.xcol = s.xlane ? xl : 0;
  s.lds = lds_all + gib * lds_stride;
  {
    T* le = s.lds + Lay::LDS_E;
    T* lc = s.lds + Lay::LDS_C;
    for (int i = s.lane; i < NP; i += G) le[i] = a.auxvar[traj * NP + i];
    for (int i = s.lane; i < NC; i += G) lc[i] = a.consts[traj * a.const_stride + i];
    if constexpr (M::ND > 0) { LFSD_WAVE_SYNC(); if (s.lane == 0) M::derive_consts(lc); }
    s.e = le; s.c = lc;
  }
  LFSD_WAVE_SYNC();
  s.dgrid = a.horizon[traj] / T(a.n_grid);
#pragma unroll
  for (int i = 0; i < NX; ++i) s.ox[i] = (s.lane == i) ? T(1) : T(0);
#pragma unroll
  for (int i = 0; i < NP; ++i) s.oe[i] = (LAY == 1 ? (s.xlane && s.xcol == i) : (s.lane == NX + i)) ? T(1) : T(0);
}

// Synchronisation in the two auxiliary sweeps: the number of split units per interval (`units`) follows each trajectory's
// own stiffness, so the lane groups of one wavefront pass DIFFERENT numbers of synchronisation points.  A workgroup barrier
// (__syncthreads / s_barrier) under such data-dependent control flow is undefined; what the sweeps need is less than a
// barrier anyway: a workgroup is exactly ONE wavefront (launched with 64 threads, checked on entry), every lane group
// works on its own private LDS slice, and the only hand-over is between lanes of the same wavefront.  LFSD_WAVE_SYNC
// (cpdp_common.h) is exactly that: an LDS-scoped release/acquire fence pair (s_waitcnt lgkmcnt(0) -- the DS queue of a
// wavefront is in order) around a wave_barrier (a scheduling fence for the compiler, no instruction).  A port to multi-wave
// workgroups would have to make `units` block-uniform first (as oc_solve_kernel does with its votes).
template <class M, typename T, int G>
__global__ void __launch_bounds__(64, (sizeof(T) == 4 ? LFSD_WAVES_RIC : 1)) aux_riccati_kernel(AuxArgs<T> a) {
  if (blockDim.x != 64) return;                   // one wavefront per workgroup: see the note on barriers above
  using Ctx = AuxCtx<M, T, G, 0>;
  using Lay = AuxLayout<M>;
  constexpr int NX = M::NX, NP = M::NP, NZ = NX + NP;
  constexpr int GPB = 64 / G;
  static_assert(64 % G == 0 && G >= NZ && G >= Lay::NNODE, "lane group must hold one column of [P W] per lane");
  __shared__ T lds_all[GPB * Lay::template lds_elems_ric<G>()];
  poison_lds(lds_all, GPB * Lay::template lds_elems_ric<G>());
  const long long slot = (long long)blockIdx.x * GPB + threadIdx.x / G;
  const bool valid = slot < a.batch;
  const long long traj = valid ? slot : (long long)a.batch - 1;
  if (a.skipped(traj)) {                          // a solve the caller does not want differentiated: this lane group is done
    if (valid && a.stats && threadIdx.x % G == 0) { a.stats[traj * 4 + 0] = 0; a.stats[traj * 4 + 1] = 0; }
    if (valid) {                                  // its [P W] is NaN, not whatever the buffer held (include/lfsd_cpdp.h)
      T* Zs = a.Z_grid + traj * (long long)(a.n_grid + 1) * NZ * NX;
      const T nan = T(0) / T(0);
      for (int i = threadIdx.x % G; i < (a.n_grid + 1) * NZ * NX; i += G) Zs[i] = nan;
    }
    return;                                       // (its lanes leave together; the other groups of the wavefront share nothing with it)
  }
  Ctx s;
  aux_setup<M, T, G, 0>(s, a, traj, lds_all, Lay::template lds_elems_ric<G>());
  const int N = a.n_grid, Sa = a.substeps;
  const int lane = s.lane;
  {
    const int col = lane < NZ ? lane : 0;
    const T* n0 = s.lds + Lay::LDS_L;
#pragma unroll
    for (int i = 0; i < NX; ++i) s.hcol[i] = n0 + (lane < NZ ? M::hcol_off0(col, i) : M::OFF_ZERO);
#pragma unroll
    for (int a2 = 0; a2 < M::NU; ++a2) s.ucol[a2] = n0 + (lane < NZ ? M::ucol_off0(col, a2) : M::OFF_ZERO);
  }
  T* Zt = a.Z_grid + traj * (long long)(N + 1) * NZ * NX;
  T z[NX];
  {
    T xN[NX];
    const T* xs = a.state_grid + (traj * (N + 1) + N) * NX;
#pragma unroll
    for (int i = 0; i < NX; ++i) xN[i] = xs[i];
    const T tN = M::TIME_VARYING ? s.dgrid * T(N) : T(0);
    M::final_hess_mul(tN, xN, s.e, s.c, s.ox, s.oe, z);      // [ddhxx ddhxe], CPDP.py:330-331
    if (lane >= NZ) {
#pragma unroll
      for (int i = 0; i < NX; ++i) z[i] = T(0);
    }
    if (valid && lane < NZ) {
#pragma unroll
      for (int i = 0; i < NX; ++i) Zt[((long long)N * NZ + lane) * NX + i] = z[i];
    }
  }
  T* ldsT = s.lds + Lay::LDS_T;
  int units_hint = Sa;
  int n_units = 0, n_unmet = 0;      // (group-uniform) units executed incl. rejected attempts; intervals accepted above tolerance
  const int budget = a.budget(traj);
  for (int k = N - 1; k >= 0; --k) {
    s.load_interval(a, traj, k, N, -1);
    // stiffness-aware sub-stepping: P is largest at the later end of the interval (terminal transient).  The coefficients
    // are staged for the first unit of the expected unit count at once: node 0 sits at the interval end either way, and
    // when the stiffness estimate confirms the count the first unit need not stage again
    const int units_guess = units_hint;
    s.stage_nodes(T(1), T(-1) / T(4 * units_guess));
    const int refine_k = (budget > 0 && n_units >= budget) ? 1 : a.max_refine;      // budget spent: no refinement
    int units = s.units_for(s.stiff_rate(z, s.node(0)), Sa, a.rate_max, refine_k);
    // error-driven refinement stops at max_refine x the minimum units -- and as soon as a doubling fails to halve the
    // estimate: next to a conjugate point (finite escape of the Riccati solution) no step size meets a relative tolerance,
    // and one such trajectory must not stall the batch
    const long long units_cap = (long long)Sa * refine_k;
    if (units < units_hint) units = (int)t_min((long long)units_hint, units_cap);
    T ratio_prev = T(-1);
    bool staged = (units == units_guess);
    // Step-size control INSIDE a stiff interval (round 6).  The uniform refinement below sizes every unit of an interval for its
    // stiffest point.  The interval before a heavy final cost is a transient: P starts at h_xx and decays like 1 / (1/P_0 + R tau) --
    // on the robot arm dgrid x rate falls from 2 000 to 1 inside that one interval, which took 500 of the sweep's 565 units (and
    // 2.6 ms of a 15 ms learner step).  From LFSD_RIC_ADAPT units up the interval is integrated with steps of its own: every step is a
    // Richardson pair as before, judged on ITS estimate against the same tolerances; a refused step is redone from its parked start
    // value with half the length; after a step whose estimate leaves an 8-fold margin the length doubles when the stiffness at the
    // NEW position allows it (dt x rate <= rate_max, evaluated on the coefficients the step has just staged at its far node).
    // Positions are integers on the interval's finest admissible grid (units_cap ticks), steps powers of two: the last step lands on
    // the grid node exactly.  Quiet intervals (the headline's 1-4 units) keep the uniform path, bit for bit.
    // fp32 only: the fp64 sweep is the parity reference; its uniform intervals over-deliver by orders (steps sized for the stiffest point
    // everywhere), its floors against the tight oracle were set on that, and it stays bit for bit what it was.
    bool adaptive_done = false;
    if constexpr (sizeof(T) == 4 && Lay::template ric_adaptive<G>()) {
    if (a.rtol > T(0) && units >= (LFSD_RIC_ADAPT) && units_cap % units == 0) {
      adaptive_done = true;
      const int R = (int)units_cap;
      int stp = R / units, pos = R;
      bool unmet = false;
      T* zpark = s.lds + Lay::template ric_park<G>();
      T* zstart = s.lds + Lay::template ric_park2<G>();
      while (pos > 0) {
        const T f = T(stp) / T(R), s_hi = T(pos) / T(R);
        if (!(staged && pos == R && stp * units_guess == R)) s.stage_nodes(s_hi, -f * T(0.25));
        staged = false;
        const T hc = s.dgrid * f;
#pragma unroll
        for (int i = 0; i < NX; ++i) { zpark[i * G + lane] = z[i]; zstart[i * G + lane] = z[i]; }
        s.ric_strang(z, 0, 2, 4, hc);
#pragma unroll
        for (int i = 0; i < NX; ++i) { const T z0 = zpark[i * G + lane]; zpark[i * G + lane] = z[i]; z[i] = z0; }
        s.ric_strang2(z, hc);
        T err_l = T(0), scl_l = T(0);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const T zc = zpark[i * G + lane];
          err_l = t_max(err_l, t_abs(z[i] - zc));
          z[i] = (T(4) * z[i] - zc) * (T(1) / T(3));
          scl_l = t_max(scl_l, t_abs(z[i]));
        }
        ++n_units;
        ldsT[lane] = err_l; ldsT[G + lane] = scl_l;
        LFSD_WAVE_SYNC();
        T eP = T(0), sP = T(0), eW = T(0), sW = T(0);
        for (int l = 0; l < NZ; ++l) {
          if (l < NX) { eP = t_max(eP, ldsT[l]); sP = t_max(sP, ldsT[G + l]); }
          else { eW = t_max(eW, ldsT[l]); sW = t_max(sW, ldsT[G + l]); }
        }
        LFSD_WAVE_SYNC();
        const T tolP = T(3) * a.rtol * sP, tolW = T(3) * a.rtol * (sW + T(1e-3) * sP);
        const bool fine_enough = (eP <= tolP && eW <= tolW) || !(t_finite(eP) && t_finite(eW));
        const T ratio = t_max(eP / t_max(tolP, T(1e-30)), eW / t_max(tolW, T(1e-30)));
        const bool no_gain = ratio_prev >= T(0) && ratio > T(0.5) * ratio_prev;      // (the halved step did not halve the estimate: next to a conjugate point)
#if defined(LFSD_AUX_TRACE)
        if (lane == 0 && slot < LFSD_AUX_TRACE) printf("ric traj %d k %d adaptive pos %d / %d step %d ratio %.3e\n", (int)slot, k, pos, R, stp, (double)ratio);
#endif
        if (fine_enough || no_gain || stp == 1 || !valid) {
          if (!fine_enough) unmet = true;
          pos -= stp;
          ratio_prev = T(-1);
          // (the margin of the estimate before a step may grow is the uniform path's before it halves the next interval's units.  Steps
          //  sized for the LOCAL stiffness each contribute what only the stiffest units of a uniform interval did: measured in fp64 -- robot
          //  arm n_grid 30, [P W] against the tight oracle -- 1e-8 uniform, 6.1e-7 with this margin, 7.0e-8 with a 256-fold one; in fp32
          //  the rounding floor of the sweep, 1e-5, hides the difference: [P W] 2.4e-6 / 8.7e-6 either way)
          constexpr int MARGIN = LFSD_AUX_DOWN;
          if (pos > 0 && eP * T(MARGIN) <= tolP && eW * T(MARGIN) <= tolW && pos % (2 * stp) == 0 && (long long)2 * stp * Sa <= R) {
            // (the stiffness where the NEXT step starts: node 4 of this step's staging sits exactly there)
            const int need = s.units_for(s.stiff_rate(z, s.node(4)), Sa, a.rate_max, refine_k);
            if ((long long)need * 2 * stp <= R) stp *= 2;
          }
        } else {
          ratio_prev = ratio;
          stp /= 2;
#pragma unroll
          for (int i = 0; i < NX; ++i) z[i] = zstart[i * G + lane];
        }
      }
      if (unmet) ++n_unmet;
      units_hint = (int)t_max((long long)Sa, (long long)(R / stp));      // the next interval starts with the step this one ended on
    }
    }
    if (!adaptive_done)
    // Error-controlled sub-stepping (a.rtol > 0): the Richardson pair gives |fine - coarse| / 3 as an estimate of the
    // second-order error that the extrapolation removes; while it exceeds rtol relative to the column's size the interval
    // is redone from its stored start value Z(t_k+1) with twice the units.  (solve_ivp's rtol of the reference, CPDP.py:335,
    // is 1e-3 on the un-extrapolated estimate of its pair, and so is the default here.)
    for (;;) {
      const T hc = s.dgrid / T(units);
      const T ds = T(1) / T(4 * units);
      T err_l = T(0), scl_l = T(0);
      for (int unit = 0; unit < units; ++unit) {
        const T s_hi = T(1) - T(unit) / T(units);
        if (!(staged && unit == 0)) s.stage_nodes(s_hi, -ds);      // node i sits at fraction s_hi - i/(4 units)
        staged = false;
        // coarse chain in place, then the fine chain in place from the parked start value (the barriers inside the chains
        // keep the compiler from carrying the parked column in registers)
        T* zpark = s.lds + Lay::template ric_park<G>();
#pragma unroll
        for (int i = 0; i < NX; ++i) zpark[i * G + lane] = z[i];
        s.ric_strang(z, 0, 2, 4, hc);
#pragma unroll
        for (int i = 0; i < NX; ++i) { const T z0 = zpark[i * G + lane]; zpark[i * G + lane] = z[i]; z[i] = z0; }
        s.ric_strang2(z, hc);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const T zc = zpark[i * G + lane];
          err_l = t_max(err_l, t_abs(z[i] - zc));
          z[i] = (T(4) * z[i] - zc) * (T(1) / T(3));     // Richardson (Strang is O(h^2), symmetric); constant reciprocal: no division
          scl_l = t_max(scl_l, t_abs(z[i]));
        }
      }
      n_units += units;
      if (!(a.rtol > T(0))) break;
      // per block of columns (P: lanes < NX, W: the rest) the worst estimate against that block's magnitude
      ldsT[lane] = err_l; ldsT[G + lane] = scl_l;
      LFSD_WAVE_SYNC();
      T eP = T(0), sP = T(0), eW = T(0), sW = T(0);
      for (int l = 0; l < NZ; ++l) {
        if (l < NX) { eP = t_max(eP, ldsT[l]); sP = t_max(sP, ldsT[G + l]); }
        else { eW = t_max(eW, ldsT[l]); sW = t_max(sW, ldsT[G + l]); }
      }
      LFSD_WAVE_SYNC();
      const T tolP = T(3) * a.rtol * sP, tolW = T(3) * a.rtol * (sW + T(1e-3) * sP);
      const bool fine_enough = (eP <= tolP && eW <= tolW) || !(t_finite(eP) && t_finite(eW));
      const T ratio = t_max(eP / t_max(tolP, T(1e-30)), eW / t_max(tolW, T(1e-30)));
      const bool no_gain = ratio_prev >= T(0) && ratio > T(0.5) * ratio_prev;
#if defined(LFSD_AUX_TRACE)
      if (lane == 0 && slot < LFSD_AUX_TRACE) printf("ric traj %d k %d units %d ratio %.3e (P %.3e W %.3e) prior %d\n", (int)slot, k, units, (double)ratio, (double)(eP / t_max(tolP, T(1e-30))), (double)(eW / t_max(tolW, T(1e-30))), units_guess);
#endif
      ratio_prev = ratio;
      if (fine_enough || no_gain || !valid || (long long)units * 2 > units_cap) {
        if (!fine_enough) ++n_unmet;       // refinement gave up (next to a conjugate point, or at the cap): reported, not hidden
        // next interval: start from this interval's units, or half of them when the estimate leaves room for it
        units_hint = (eP * T(LFSD_AUX_DOWN) <= tolP && eW * T(LFSD_AUX_DOWN) <= tolW && units > Sa) ? units / 2 : units;
        break;
      }
      units *= 2;
      if (lane < NZ) {
#pragma unroll
        for (int i = 0; i < NX; ++i) z[i] = Zt[((long long)(k + 1) * NZ + lane) * NX + i];
      }
    }
    // keep P symmetric (the closed-form stiff update relies on it) and store the grid value
    if (lane < NX) {
#pragma unroll
      for (int i = 0; i < NX; ++i) ldsT[lane * NZ + i] = z[i];
    }
    LFSD_WAVE_SYNC();
    if (lane < NX) {
#pragma unroll
      for (int i = 0; i < NX; ++i) z[i] = T(0.5) * (z[i] + ldsT[i * NZ + lane]);
    }
    LFSD_WAVE_SYNC();
    if (valid && lane < NZ) {
#pragma unroll
      for (int i = 0; i < NX; ++i) Zt[((long long)k * NZ + lane) * NX + i] = z[i];
    }
  }
  if (valid && a.stats && lane == 0) { a.stats[traj * 4 + 0] = n_units; a.stats[traj * 4 + 1] = n_unmet; }
}

template <class M, typename T, int G>
__global__ void __launch_bounds__(64, (sizeof(T) == 4 ? LFSD_WAVES_FWD : 1)) aux_forward_kernel(AuxArgs<T> a) {
  if (blockDim.x != 64) return;                   // one wavefront per workgroup: see the note above aux_riccati_kernel
  using Ctx = AuxCtx<M, T, G, 1>;
  using Lay = AuxLayout<M, 1>;
  constexpr int NX = M::NX, NU = M::NU, NP = M::NP, NZ = NX + NP;
  constexpr int GPB = 64 / G;
  static_assert(64 % G == 0 && G >= NX && G >= 2 * NP && G >= Lay::NNODE && G >= 3 * sub_lanes<NU <= 4 ? NU : 1>(),
                "forward lane group: one P column per lane, and one X column per lane in each half (fine / coarse chain)");
  __shared__ T lds_all[GPB * Lay::lds_elems_fwd()];
  poison_lds(lds_all, GPB * Lay::lds_elems_fwd());
  const long long slot = (long long)blockIdx.x * GPB + threadIdx.x / G;
  const bool valid = slot < a.batch;
  const long long traj = valid ? slot : (long long)a.batch - 1;
  if (a.skipped(traj)) {                          // not differentiated: NaN loss / gradient / sensitivity grids
    const int l = threadIdx.x % G;
    const T nan = T(0) / T(0);
    if (valid) {
      if (l == 0) { a.loss[traj] = nan; if (a.stats) { a.stats[traj * 4 + 2] = 0; a.stats[traj * 4 + 3] = 0; } }
      if (l < NP) a.grad[traj * NP + l] = nan;
      if (a.auxX_grid) { T* o = a.auxX_grid + traj * (long long)(a.n_grid + 1) * NP * NX; for (int i = l; i < (a.n_grid + 1) * NP * NX; i += G) o[i] = nan; }
      if (a.auxU_grid) { T* o = a.auxU_grid + traj * (long long)(a.n_grid + 1) * NP * NU; for (int i = l; i < (a.n_grid + 1) * NP * NU; i += G) o[i] = nan; }
    }
    return;
  }
  Ctx s;
  aux_setup<M, T, G, 1>(s, a, traj, lds_all, Lay::lds_elems_fwd());
  const int N = a.n_grid, Sa = a.substeps;
  const int lane = s.lane;
  const bool xlane = s.xlane, coarse = s.coarse;
  const bool fine_x = xlane && !coarse;           // the lanes that own the results of a column
  const T* Zt = a.Z_grid + traj * (long long)(N + 1) * NZ * NX;
  T xa[NX], wA[NX], wB[NX];     // X column; W column at both ends of the interval (the P columns live in LDS)
#pragma unroll
  for (int i = 0; i < NX; ++i) { xa[i] = T(0); wA[i] = T(0); wB[i] = T(0); }     // X(0) = 0, CPDP.py:355
  T* ldsPA = s.lds + Lay::FWD_P;
  T* ldsPB = ldsPA + NX * NX;
  T* ldsP0 = ldsPB + NX * NX;                                  // zero row
  for (int i = lane; i < NX; i += G) ldsP0[i] = T(0);
  const T* pA = (lane < NX) ? ldsPA + lane * NX : ldsP0;
  const T* pB = (lane < NX) ? ldsPB + lane * NX : ldsP0;
  T pN[NX], wN[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) { pN[i] = T(0); wN[i] = T(0); }
  T* xprev = s.lds + Lay::FWD_XPREV;                 // X(t_k), parked in LDS: start value of a redone interval, and the loss needs it
  T* xch = s.lds + Lay::FWD_XCH;
  T* ldsR = s.lds + Lay::FWD_RED;
  T loss = T(0), gacc = T(0);
  int units_hint = Sa;
  int n_units = 0, n_unmet = 0;
  const int budget = a.budget(traj);
  T* Xo = a.auxX_grid ? a.auxX_grid + traj * (long long)(N + 1) * NP * NX : nullptr;
  T* Uo = a.auxU_grid ? a.auxU_grid + traj * (long long)(N + 1) * NP * NU : nullptr;
  if (valid && Xo && fine_x) {
#pragma unroll
    for (int i = 0; i < NX; ++i) Xo[(long long)s.xcol * NX + i] = T(0);
  }
  for (int k = 0; k < N; ++k) {
    s.load_interval(a, traj, k, N, +1);
    // [P W] at the two ends of the interval: the later end of interval k is the earlier end of interval k+1, and the new end is
    // fetched into registers one interval ahead (pN: this lane's column of P, wN: its column of W) -- as the grid rows above
    if (k == 0 || !Ctx::PF) {
      if (lane < NX) {           // (load_interval's barriers fence the previous interval's readers; stage_nodes' the writers)
#pragma unroll
        for (int i = 0; i < NX; ++i) { ldsPA[lane * NX + i] = Zt[((long long)k * NZ + lane) * NX + i]; ldsPB[lane * NX + i] = Zt[((long long)(k + 1) * NZ + lane) * NX + i]; }
      }
      if (xlane) {
#pragma unroll
        for (int i = 0; i < NX; ++i) { wA[i] = Zt[((long long)k * NZ + NX + s.xcol) * NX + i]; wB[i] = Zt[((long long)(k + 1) * NZ + NX + s.xcol) * NX + i]; }
      }
    } else {
      { T* t_ = ldsPA; ldsPA = ldsPB; ldsPB = t_; }
      pA = (lane < NX) ? ldsPA + lane * NX : ldsP0;
      pB = (lane < NX) ? ldsPB + lane * NX : ldsP0;
      if (lane < NX) {
#pragma unroll
        for (int i = 0; i < NX; ++i) ldsPB[lane * NX + i] = pN[i];
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) { wA[i] = wB[i]; wB[i] = wN[i]; }
    }
    if (Ctx::PF && k + 2 <= N) {
      if (lane < NX) {
#pragma unroll
        for (int i = 0; i < NX; ++i) pN[i] = Zt[((long long)(k + 2) * NZ + lane) * NX + i];
      }
      if (xlane) {
#pragma unroll
        for (int i = 0; i < NX; ++i) wN[i] = Zt[((long long)(k + 2) * NZ + NX + s.xcol) * NX + i];
      }
      LFSD_ISSUE_FENCE();
    }
    if (fine_x) {
#pragma unroll
      for (int i = 0; i < NX; ++i) xprev[i * NP + s.xcol] = xa[i];
    }
    s.stage_nodes(T(0), T(0.25));          // nodes at 0, 1/4 .. 1 of the interval: the stiffness at both ends -- and exactly
    const T rate = t_max(s.stiff_rate(pA, s.node(0)), s.stiff_rate(pB, s.node(4)));      // the staging of a single unit
    const int refine_k = (budget > 0 && n_units >= budget) ? 1 : a.max_refine;      // budget spent: no refinement
    int units = s.units_for(rate, Sa, a.rate_max, refine_k);
    const long long units_cap = (long long)Sa * refine_k;
    if (units < units_hint) units = (int)t_min((long long)units_hint, units_cap);
    T ratio_prev = T(-1);
    bool staged = (units == 1);
    for (;;) {                         // error-controlled sub-stepping, as in the Riccati sweep; the start value X(t_k) is `xprev`
      const T hc = s.dgrid / T(units);
      const T ds = T(1) / T(4 * units);
      T err_l = T(0), scl_l = T(0);
      for (int unit = 0; unit < units; ++unit) {
        const T s_lo = T(unit) / T(units);
        if (!(staged && unit == 0)) s.stage_nodes(s_lo, ds);
        staged = false;
        if (Uo && unit == 0) {
          T uo[NU];
          s.aux_control(xa, pA, wA, s.node(0), uo);
          if (valid && fine_x) {
#pragma unroll
            for (int b = 0; b < NU; ++b) Uo[((long long)k * NP + s.xcol) * NU + b] = uo[b];
          }
        }
        const T hq = hc * T(0.25);
        s.fwd_prep(pA, pB, s_lo, ds, hq);
        s.fwd_cols(wA, wB, s_lo, ds);
        LFSD_WAVE_SYNC();      // the parked columns b_j of every node are in place
        // the Richardson pair: coarse Strang step on the upper half of the lane group, two fine ones on the lower half,
        // both in place from the same start value
        s.fwd_chain(xa, hc, hq);
        if (xlane) {
#pragma unroll
          for (int i = 0; i < NX; ++i) xch[((coarse ? NX : 0) + i) * NP + s.xcol] = xa[i];
        }
        LFSD_WAVE_SYNC();      // (also: all reads of this unit's staged coefficients are done before the next staging)
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const T xo = xlane ? xch[((coarse ? 0 : NX) + i) * NP + s.xcol] : T(0);      // the other chain's result
          const T xf = coarse ? xo : xa[i], xc = coarse ? xa[i] : xo;
          err_l = t_max(err_l, t_abs(xf - xc));
          xa[i] = xlane ? (T(4) * xf - xc) * (T(1) / T(3)) : T(0);      // both halves continue from the extrapolated value
          scl_l = t_max(scl_l, t_abs(xa[i]));
        }
        if (Uo && k == N - 1 && unit == units - 1) {
          T uo[NU];
          s.aux_control(xa, pB, wB, s.node(4), uo);
          if (valid && fine_x) {
#pragma unroll
            for (int b = 0; b < NU; ++b) Uo[((long long)N * NP + s.xcol) * NU + b] = uo[b];
          }
        }
      }
      n_units += units;
#if defined(LFSD_AUX_TRACE)
      if (lane == 0 && slot < LFSD_AUX_TRACE) printf("fwd traj %d k %d ran units %d hc %.6e xa %.9e %.9e %.9e\n", (int)slot, k, units, (double)hc, (double)xa[0], (double)xa[1], (double)xa[2]);
#endif
      if (!(a.rtol > T(0))) break;
      LFSD_WAVE_SYNC();
      ldsR[lane] = err_l; ldsR[G + lane] = scl_l;
      LFSD_WAVE_SYNC();
      T eX = T(0), sX = T(0);
      for (int l = 0; l < NP; ++l) { eX = t_max(eX, ldsR[l]); sX = t_max(sX, ldsR[G + l]); }
      LFSD_WAVE_SYNC();
      const T tolX = T(3) * a.rtol * (sX + T(1e-2));       // dx/dtheta starts from zero: absolute floor 1e-2 * rtol
      const T ratio = eX / tolX;
      const bool no_gain = ratio_prev >= T(0) && ratio > T(0.5) * ratio_prev;
#if defined(LFSD_AUX_TRACE)
      if (lane == 0 && slot < LFSD_AUX_TRACE) printf("fwd traj %d k %d units %d ratio %.3e\n", (int)slot, k, units, (double)ratio);
#endif
      ratio_prev = ratio;
      if (eX <= tolX || no_gain || !t_finite(eX) || (long long)units * 2 > units_cap) {
        if (!(eX <= tolX)) ++n_unmet;
        units_hint = (eX * T(LFSD_AUX_DOWN) <= tolX && units > Sa) ? units / 2 : units;
        break;
      }
      units *= 2;
#pragma unroll
      for (int i = 0; i < NX; ++i) xa[i] = xlane ? xprev[i * NP + s.xcol] : T(0);
    }
    if (valid && Xo && fine_x) {
#pragma unroll
      for (int i = 0; i < NX; ++i) Xo[((long long)(k + 1) * NP + s.xcol) * NX + i] = xa[i];
    }
    // loss and gradient contributions of the waypoints that fall into this interval
    // (linear interpolation of the grid values, exactly what opt_sol(t)/auxsys_sol(t) do: CPDP.py:386)
    for (int w = 0; w < a.n_waypoints; ++w) {
      const T tau = a.taus[traj * a.n_waypoints + w];
      int kw = (int)t_floor(tau / s.dgrid);
      kw = kw < 0 ? 0 : (kw > N - 1 ? N - 1 : kw);
      if (kw != k) continue;
      const T sw = (tau - T(k) * s.dgrid) / s.dgrid;
      T rvec[NX];
#pragma unroll
      for (int i = 0; i < NX; ++i) rvec[i] = T(0);
      bool compiled_iface = false;
      if constexpr (M::NIF > 0) {
        // the interface function compiled into the model (iface_idx == NULL): y = g(x(tau)) on the interpolated state, residual
        // r = y - waypoint, and r^T dg/dx as the vector the sensitivity is contracted with (lib/QuadAlgorithm.py:625-637: an
        // arbitrary CasADi expression of the state and its jacobian)
        if (a.iface_idx == nullptr) {
          compiled_iface = true;
          T cur[NX], y[M::NIF], r[M::NIF];
#pragma unroll
          for (int i = 0; i < NX; ++i) cur[i] = s.xa_[i] + sw * (s.xb_[i] - s.xa_[i]);
          M::iface(cur, s.c, y);
#pragma unroll
          for (int q = 0; q < M::NIF; ++q) {
            r[q] = y[q] - a.waypoints[(traj * a.n_waypoints + w) * M::NIF + q];
            loss += r[q] * r[q];
          }
          M::iface_vjp(cur, s.c, r, rvec);
        }
      }
      for (int q = 0; q < (compiled_iface ? 0 : a.n_iface); ++q) {
        const int idx = a.iface_idx[q];
        const T target = a.waypoints[(traj * a.n_waypoints + w) * a.n_iface + q];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          if (i == idx) {
            const T cur = s.xa_[i] + sw * (s.xb_[i] - s.xa_[i]);
            const T r = cur - target;
            rvec[i] += r;
            loss += r * r;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NX; ++i) { const T xp = xprev[i * NP + s.xcol]; gacc += rvec[i] * (xp + sw * (xa[i] - xp)); }
    }
  }
  if (valid) {
    if (lane == 0) a.loss[traj] = loss;
    if (fine_x) a.grad[traj * NP + s.xcol] = gacc;
    if (a.stats && lane == 0) { a.stats[traj * 4 + 2] = n_units; a.stats[traj * 4 + 3] = n_unmet; }
  }
}

}  // namespace lfsd
