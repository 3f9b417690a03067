"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU / fp64 restatement of the reference's Continuous-PDP hot path,
/root/reference/CPDP/CPDP.py (class ``COCSys``; ``COCSys_TimeVarying`` is the
same code with an explicit time argument and is covered by ``time_var=``).

Third-party pieces the reference leans on that are NOT in this image:

* CasADi 3.5.5 (symbolic SX graph + ``jacobian``)  -> sympy + lambdify here.
* IPOPT 3.11.9 via ``casadi.nlpsol`` (CPDP.py:177-184) -> the NLP built at
  CPDP.py:110-175 (multiple shooting, RK4, piecewise-constant control) is an
  equality-constrained problem whose KKT point IPOPT returns to tol 1e-8.  The
  oracle finds the same KKT point with an fp64 DDP/Newton iteration on the
  identical discretisation, and ``kkt_certificate`` proves stationarity with
  derivative-free (complex-step) arithmetic that shares no code with the
  solver.  ``lam_g`` (CPDP.py:193) is the multiplier of ``Xk_end - X_{k+1}``,
  i.e. the discrete costate lambda_k = dJ/dx_k, which is what we return.
* scipy ``solve_ivp`` / ``interp1d`` ARE here and are called exactly as the
  reference calls them (CPDP.py:335, 368, 386).

Parity pin: tests/golden/uav_golden.npz (extracted from the reference's own
saved run data/uav_results_random_20210308113016.mat) — see
tests/test_oracle_golden.py.
"""
import numpy as np
import sympy as sp
import scipy.interpolate as ip
from scipy.integrate import solve_ivp


def _lam(args, expr):
    expr = sp.Matrix(expr) if not isinstance(expr, sp.MatrixBase) else expr
    fn = sp.lambdify(args, expr, modules='numpy', cse=True)
    shape = expr.shape

    def call(*vals):
        out = fn(*vals)
        out = np.asarray(out)
        if out.dtype == object:       # mixed scalar/array entries
            out = np.array(out.tolist(), dtype=np.result_type(*[np.asarray(v).dtype for v in vals]))
        return out.reshape(shape)
    return call


class COCSys:
    """Restates CPDP.COCSys (CPDP.py:9-390) / COCSys_TimeVarying (CPDP.py:394-786)."""

    def __init__(self, project_name="myOc", time_varying=False):
        self.sys_name = project_name
        self.time_varying = time_varying
        self.time = sp.Symbol('time', real=True)

    # ---- CPDP.py:15-87 ----------------------------------------------------
    def setAuxvarVariable(self, auxvar):
        self.auxvar = list(auxvar)
        self.n_auxvar = len(self.auxvar)

    def setStateVariable(self, state, state_lb=[], state_ub=[]):
        """CPDP.py:20-31: bounds default to -1e20 / +1e20 per component; finite ones become lbw / ubw of the node
        states X_1..X_N of the NLP (CPDP.py:140-147; X_0 is pinned to ini_state, CPDP.py:131-134)."""
        self.state = list(state)
        self.n_state = len(self.state)
        self.state_lb = list(state_lb) if len(state_lb) == self.n_state else self.n_state * [-1e20]
        self.state_ub = list(state_ub) if len(state_ub) == self.n_state else self.n_state * [1e20]

    def setControlVariable(self, control, control_lb=[], control_ub=[]):
        """CPDP.py:33-46: the same for the controls (lbw / ubw of the U_k, CPDP.py:150-153)."""
        self.control = list(control)
        self.n_control = len(self.control)
        self.control_lb = list(control_lb) if len(control_lb) == self.n_control else self.n_control * [-1e20]
        self.control_ub = list(control_ub) if len(control_ub) == self.n_control else self.n_control * [1e20]

    def has_state_bounds(self):
        return any(abs(v) < 1e19 for v in list(getattr(self, "state_lb", [])) + list(getattr(self, "state_ub", [])))

    def has_control_bounds(self):
        return any(abs(v) < 1e19 for v in list(getattr(self, "control_lb", [])) + list(getattr(self, "control_ub", [])))

    def setTimeVariable(self, t):
        self.time = t

    def setDyn(self, ode):
        self.dyn = sp.Matrix(ode)

    def setPathCost(self, path_cost):
        self.path_cost = sp.sympify(path_cost)

    def setFinalCost(self, final_cost):
        self.final_cost = sp.sympify(final_cost)

    def setIntegrator(self, n_grid=10, steps_per_grid=4):
        self.n_grid = n_grid
        self.steps_per_grid = steps_per_grid

    def setInterface(self, interface):
        """The examples' interface function and its Jacobian (Examples/robotarm_random.py:35-36:
        ``Function('interface', [oc.state], [expr])`` and ``Function('diff_interface', [oc.state], [jacobian(expr, oc.state)])``),
        as lambdified sympy expressions of the state."""
        g = sp.Matrix(list(interface))
        X = sp.Matrix(self.state)
        self.interface_fn = sp.lambdify([self.state], list(g), modules="numpy")
        self.diff_interface_fn = sp.lambdify([self.state], g.jacobian(X), modules="numpy")

    # ---- CPDP.py:201-248 --------------------------------------------------
    def diffPMP(self):
        if hasattr(self, '_fn'):
            return
        X, U, E = sp.Matrix(self.state), sp.Matrix(self.control), sp.Matrix(self.auxvar)
        self.costate = [sp.Symbol('lambda_%d' % i, real=True) for i in range(self.n_state)]
        L = sp.Matrix(self.costate)
        f, c, h = self.dyn, self.path_cost, self.final_cost
        H = c + (f.T * L)[0, 0]                      # CPDP.py:218
        dHx = sp.Matrix([H]).jacobian(X).T
        dHu = sp.Matrix([H]).jacobian(U).T
        dhx = sp.Matrix([h]).jacobian(X).T
        a_xue = [self.time] + self.state + self.control + self.auxvar
        a_xule = [self.time] + self.state + self.control + self.costate + self.auxvar
        a_xe = [self.time] + self.state + self.auxvar
        self._fn = dict(
            dyn=_lam(a_xue, f), cost=_lam(a_xue, [c]), final=_lam(a_xe, [h]),
            dfx=_lam(a_xue, f.jacobian(X)), dfu=_lam(a_xue, f.jacobian(U)), dfe=_lam(a_xue, f.jacobian(E)),
            dcx=_lam(a_xue, sp.Matrix([c]).jacobian(X)), dcu=_lam(a_xue, sp.Matrix([c]).jacobian(U)),
            dHx=_lam(a_xule, dHx), dHu=_lam(a_xule, dHu),
            ddHxx=_lam(a_xule, dHx.jacobian(X)), ddHxu=_lam(a_xule, dHx.jacobian(U)),
            ddHxe=_lam(a_xule, dHx.jacobian(E)), ddHux=_lam(a_xule, dHu.jacobian(X)),
            ddHuu=_lam(a_xule, dHu.jacobian(U)), ddHue=_lam(a_xule, dHu.jacobian(E)),
            dhx=_lam(a_xe, dhx), ddhxx=_lam(a_xe, dhx.jacobian(X)), ddhxe=_lam(a_xe, dhx.jacobian(E)),
        )

    def _call(self, name, t, *vecs):
        return self._fn[name](t, *np.concatenate([np.atleast_1d(v) for v in vecs]))

    # ---- the RK4 shooting map of CPDP.py:110-124 ----------------------------
    def grid_map(self, tk, x, u, e, DT, derivs=False):
        """(xf, qf) = 'grid_fc' of CPDP.py:117-124; with derivs also d(xf,qf)/d(x,u).

        Note (reference behaviour, kept): the time-varying variant evaluates the
        integrand at the fixed interval start ``tk`` for all sub-steps (CPDP.py:513-519).
        """
        n, m = self.n_state, self.n_control
        S = self.steps_per_grid
        dt_ = np.result_type(np.asarray(x).dtype, np.asarray(u).dtype, np.float64)
        y = np.zeros(n + 1, dtype=dt_)
        y[:n] = x

        def g(yy):
            return np.concatenate([self._call('dyn', tk, yy[:n], u, e).ravel(),
                                   self._call('cost', tk, yy[:n], u, e).ravel()])

        if derivs:
            M = np.zeros((n + 1, n + m))
            M[:n, :n] = np.eye(n)

            def dg(yy, dY):
                xx = yy[:n]
                Gx = np.vstack([self._call('dfx', tk, xx, u, e), self._call('dcx', tk, xx, u, e)])
                Gu = np.vstack([self._call('dfu', tk, xx, u, e), self._call('dcu', tk, xx, u, e)])
                out = Gx @ dY[:n, :]
                out[:, n:] += Gu
                return out
        for _ in range(S):
            k1 = g(y)
            k2 = g(y + DT / 2 * k1)
            k3 = g(y + DT / 2 * k2)
            k4 = g(y + DT * k3)
            if derivs:
                d1 = dg(y, M)
                d2 = dg(y + DT / 2 * k1, M + DT / 2 * d1)
                d3 = dg(y + DT / 2 * k2, M + DT / 2 * d2)
                d4 = dg(y + DT * k3, M + DT * d3)
                M = M + DT / 6 * (d1 + 2 * d2 + 2 * d3 + d4)
            y = y + DT / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
        if derivs:
            return y[:n], y[n], M
        return y[:n], y[n]

    def stage_hessian(self, tk, x, u, e, DT, lam_next):
        """Exact Hessian w.r.t. (x_k, u_k) of the stage Lagrangian  Q_k(x,u) + lam_{k+1}^T F_k(x,u)
        (the block of the NLP's Lagrangian Hessian that IPOPT gets from CasADi), by a second-order
        adjoint sweep through the RK4 stages.  Also returns the first-order adjoint lam_k."""
        n, m, S = self.n_state, self.n_control, self.steps_per_grid
        h = DT
        w = (h / 6, h / 3, h / 3, h / 6)
        Ud = np.hstack([np.zeros((m, n)), np.eye(m)])
        f = lambda xx: self._call('dyn', tk, xx, u, e).ravel()
        fx = lambda xx: self._call('dfx', tk, xx, u, e)
        fu = lambda xx: self._call('dfu', tk, xx, u, e)

        def stages(x0s, X0):
            xs, Xs, ks, ds = [x0s], [X0], [], []
            for a in (h / 2, h / 2, h):
                ks.append(f(xs[-1])); ds.append(fx(xs[-1]) @ Xs[-1] + fu(xs[-1]) @ Ud)
                xs.append(x0s + a * ks[-1]); Xs.append(X0 + a * ds[-1])
            ks.append(f(xs[-1])); ds.append(fx(xs[-1]) @ Xs[-1] + fu(xs[-1]) @ Ud)
            return xs, Xs, ks, ds
        x = np.asarray(x, dtype=float)
        Xd = np.hstack([np.eye(n), np.zeros((n, m))])
        starts = []
        for _ in range(S):
            starts.append((x.copy(), Xd.copy()))
            xs, Xs, ks, ds = stages(x, Xd)
            x = x + h / 6 * (ks[0] + 2 * ks[1] + 2 * ks[2] + ks[3])
            Xd = Xd + h / 6 * (ds[0] + 2 * ds[1] + 2 * ds[2] + ds[3])
        lam = np.asarray(lam_next, dtype=float).copy()
        dlam = np.zeros((n, n + m))
        du = np.zeros((m, n + m))
        carry = (0.0, h, h / 2, h / 2)          # kappa_i = w_i*lam + carry_i * ybar_{i+1}
        for s in range(S - 1, -1, -1):
            xs, Xs, _, _ = stages(*starts[s])
            yb_next, dyb_next = np.zeros(n), np.zeros((n, n + m))
            lam_new, dlam_new = lam.copy(), dlam.copy()
            for i in (3, 2, 1, 0):
                c_i = carry[3 - i] if i < 3 else 0.0
                kap = w[i] * lam + c_i * yb_next
                dkap = w[i] * dlam + c_i * dyb_next
                Fx, Fu = fx(xs[i]), fu(xs[i])
                l = kap / w[i]
                Hxx = self._call('ddHxx', tk, xs[i], u, l, e); Hxu = self._call('ddHxu', tk, xs[i], u, l, e)
                Huu = self._call('ddHuu', tk, xs[i], u, l, e)
                cx = self._call('dcx', tk, xs[i], u, e).ravel()
                yb = Fx.T @ kap + w[i] * cx
                dyb = Fx.T @ dkap + w[i] * (Hxx @ Xs[i] + Hxu @ Ud)
                du = du + Fu.T @ dkap + w[i] * (Hxu.T @ Xs[i] + Huu @ Ud)
                lam_new = lam_new + yb
                dlam_new = dlam_new + dyb
                yb_next, dyb_next = yb, dyb
            lam, dlam = lam_new, dlam_new
        Hz = np.vstack([dlam, du])
        return 0.5 * (Hz + Hz.T), lam

    def rollout_cost(self, ini_state, horizon, e, U):
        """Objective J of the NLP (CPDP.py:157-175) for a control sequence (single shooting)."""
        N = self.n_grid
        DT = horizon / N / self.steps_per_grid
        tg = np.linspace(0, horizon, N + 1)
        x = np.asarray(ini_state, dtype=np.result_type(U.dtype, np.float64))
        J = 0
        X = [x]
        for k in range(N):
            x, q = self.grid_map(tg[k], x, U[k], e, DT)
            J = J + q
            X.append(x)
        J = J + self._call('final', tg[-1], x, e)[0, 0]
        return J, np.array(X)

    # ---- CPDP.py:92-198 -----------------------------------------------------
    def cocSolver(self, ini_state, horizon, auxvar_value=1, interplation_level=1, print_level=0,
                  tol=1e-10, max_iter=300, U_init=None, return_grids=False, exact_after=16):
        self.diffPMP()
        if not hasattr(self, 'n_grid'):
            self.setIntegrator()
        n, m, N = self.n_state, self.n_control, self.n_grid
        if self.has_state_bounds() or self.has_control_bounds():
            # finite lbw / ubw (CPDP.py:140-153): the same NLP handed to scipy's bounded solvers below instead of the DDP iteration
            # (the reference hands it to IPOPT unchanged, CPDP.py:177-184)
            if self.has_state_bounds():
                cb = self.has_control_bounds()
                tg, X, Uc, J = self.cocSolverStateBounded(ini_state, horizon, auxvar_value, self.state_lb, self.state_ub,
                                                         control_lb=self.control_lb if cb else None,
                                                         control_ub=self.control_ub if cb else None, U_init=U_init)
                lam = self.costates_along(ini_state, horizon, auxvar_value, X, Uc[:N])
            else:
                tg, X, Uc, lam, J = self.cocSolverBounded(ini_state, horizon, auxvar_value, self.control_lb, self.control_ub, U_init=U_init)
            self.last_info = dict(iters=-1, converged=True, bounded=True)
            opt_sol = self.interpolation(tg, np.concatenate((X, Uc, lam), axis=1), interplation_level)
            return (tg, opt_sol, X, Uc, lam) if return_grids else (tg, opt_sol)
        e = np.asarray(auxvar_value, dtype=float).ravel()
        x0 = np.asarray(ini_state, dtype=float).ravel()
        DT = horizon / N / self.steps_per_grid
        dgrid = horizon / N
        time_grid = np.linspace(0, horizon, N + 1)     # == [horizon/N*k] of CPDP.py:192

        U = np.zeros((N, m)) if U_init is None else np.array(U_init, dtype=float)
        J, X = self.rollout_cost(x0, horizon, e, U)
        mu = 0.0
        lam = np.zeros((N + 1, n))
        info = dict(iters=0, converged=False)
        # Stage Hessian model.  exact=False: Gauss-Newton (cost curvature only: robust far from the optimum);
        # exact=True: exact Lagrangian Hessian of the RK4 shooting stage (second-order adjoint) with a
        # Levenberg shift mu on Q_uu when it is indefinite -- Newton's method on the NLP, as IPOPT runs it.
        exact = False
        gn_patience = exact_after       # same knob as lfsd_coc_solve's `exact_after`
        zero_l = np.zeros(n)
        for it in range(max_iter):
            A, B, qx, qu = [], [], [], []
            for k in range(N):
                _, _, M = self.grid_map(time_grid[k], X[k], U[k], e, DT, derivs=True)
                A.append(M[:n, :n]); B.append(M[:n, n:]); qx.append(M[n, :n]); qu.append(M[n, n:])
            # exact discrete costates  lambda_k = q_x + A^T lambda_{k+1}
            lam[N] = self._call('dhx', time_grid[-1], X[N], e).ravel()
            gnorm = 0.0
            for k in range(N - 1, -1, -1):
                gnorm = max(gnorm, np.max(np.abs(qu[k] + B[k].T @ lam[k + 1])))
                lam[k] = qx[k] + A[k].T @ lam[k + 1]
            info.update(iters=it, grad_inf=gnorm)
            if gnorm < tol * (1 + abs(J)):
                info['converged'] = True
                break
            if gn_patience >= 0 and it >= gn_patience:
                exact = True
            Hs = []
            for k in range(N):
                if exact:
                    Hs.append(self.stage_hessian(time_grid[k], X[k], U[k], e, DT, lam[k + 1])[0])
                else:
                    Hs.append(dgrid * np.block([
                        [self._call('ddHxx', time_grid[k], X[k], U[k], zero_l, e),
                         self._call('ddHxu', time_grid[k], X[k], U[k], zero_l, e)],
                        [self._call('ddHxu', time_grid[k], X[k], U[k], zero_l, e).T,
                         self._call('ddHuu', time_grid[k], X[k], U[k], zero_l, e)]]))
            while True:      # backward sweep (retry with a larger shift when Q_uu is not positive definite)
                Vx = lam[N].copy()
                Vxx = self._call('ddhxx', time_grid[-1], X[N], e)
                kff = np.zeros((N, m)); K = np.zeros((N, m, n))
                dV1 = dV2 = 0.0
                ok = True
                for k in range(N - 1, -1, -1):
                    Qx = qx[k] + A[k].T @ Vx
                    Qu = qu[k] + B[k].T @ Vx
                    Qxx = Hs[k][:n, :n] + A[k].T @ Vxx @ A[k]
                    Qux = Hs[k][n:, :n] + B[k].T @ Vxx @ A[k]
                    Quu0 = Hs[k][n:, n:] + B[k].T @ Vxx @ B[k]
                    Quu0 = 0.5 * (Quu0 + Quu0.T)
                    try:
                        Lc = np.linalg.cholesky(Quu0 + mu * np.eye(m))
                    except np.linalg.LinAlgError:
                        ok = False
                        break
                    sol = lambda r: np.linalg.solve(Lc.T, np.linalg.solve(Lc, r))
                    kff[k] = -sol(Qu)
                    K[k] = -sol(Qux)
                    dV1 += kff[k] @ Qu
                    dV2 += 0.5 * kff[k] @ Quu0 @ kff[k]
                    Vx = Qx + K[k].T @ Quu0 @ kff[k] + K[k].T @ Qu + Qux.T @ kff[k]
                    Vxx = Qxx + K[k].T @ Quu0 @ K[k] + K[k].T @ Qux + Qux.T @ K[k]
                    Vxx = 0.5 * (Vxx + Vxx.T)
                if ok:
                    break
                mu = max(10 * mu, 1e-4)
                if mu > 1e12:
                    break
            if not ok:
                break
            # forward line search on the true NLP objective
            alpha = 1.0
            accepted = False
            while alpha > 1e-10:
                Xn = [x0]; Un = np.zeros_like(U); Jn = 0.0
                x = x0
                for k in range(N):
                    Un[k] = U[k] + alpha * kff[k] + K[k] @ (x - X[k])
                    x, q = self.grid_map(time_grid[k], x, Un[k], e, DT)
                    Jn += q
                    Xn.append(x)
                Jn += self._call('final', time_grid[-1], x, e)[0, 0]
                expected = -(alpha * dV1 + alpha * alpha * dV2)
                if np.isfinite(Jn) and (J - Jn) >= 1e-4 * expected - 1e-14 * abs(J):
                    accepted = True
                    break
                alpha *= 0.5
            if not accepted:
                mu = max(10 * mu, 1e-4)
                if mu > 1e12:
                    break
                continue
            if alpha == 1.0:
                mu = mu / 10 if mu > 1e-8 else 0.0
                if (J - Jn) < 1e-2 * (abs(Jn) + 1e-12):
                    exact = True        # close: switch to Newton for the quadratic tail
            X, U, J = np.array(Xn), Un, Jn
        self.last_info = info
        self.last_cost = J

        # CPDP.py:186-196: grids, with the last control repeated
        state_grid = np.asarray(X)
        control_grid = np.vstack([U, U[-1:]])
        costate_grid = lam.copy()
        opt_sol = self.interpolation(time_grid, np.concatenate((state_grid, control_grid, costate_grid), axis=1),
                                     interplation_level)
        if return_grids:
            return time_grid, opt_sol, state_grid, control_grid, costate_grid
        return time_grid, opt_sol

    def costates_along(self, ini_state, horizon, auxvar_value, X, U):
        """Adjoint recursion lambda_k = q_x + A_k^T lambda_k+1 along given grids (the multipliers of the shooting constraints
        where no state bound is active; with an active bound IPOPT's lam_g carries the bound multiplier as well)."""
        n, N = self.n_state, self.n_grid
        e = np.asarray(auxvar_value, dtype=float).ravel()
        DT = horizon / N / self.steps_per_grid
        tg = np.linspace(0, horizon, N + 1)
        lam = np.zeros((N + 1, n))
        lam[N] = self._call('dhx', tg[-1], X[N], e).ravel()
        for k in range(N - 1, -1, -1):
            _, _, M = self.grid_map(tg[k], X[k], U[k], e, DT, derivs=True)
            lam[k] = M[n, :n] + M[:n, :n].T @ lam[k + 1]
        return lam

    def cocSolverBounded(self, ini_state, horizon, auxvar_value, control_lb, control_ub, U_init=None, tol=1e-12):
        """The NLP of CPDP.py:110-175 with finite control bounds lbw / ubw (CPDP.py:150-153), solved in single-shooting form
        by scipy's L-BFGS-B with complex-step gradients of the plain RK4 roll-out -- no code shared with the HIP solver's
        control-limited sweep or with this oracle's own DDP.  Initial guess: the midpoint of the bounds, as the reference's
        w0.  Returns (time_grid, state_grid, control_grid [N+1, last row repeated], costate_grid, J)."""
        from scipy.optimize import minimize
        self.diffPMP()
        n, m, N = self.n_state, self.n_control, self.n_grid
        e = np.asarray(auxvar_value, dtype=float).ravel()
        x0 = np.asarray(ini_state, dtype=float).ravel()
        lb, ub = np.asarray(control_lb, dtype=float), np.asarray(control_ub, dtype=float)
        DT = horizon / N / self.steps_per_grid
        tg = np.linspace(0, horizon, N + 1)

        def fun(u):
            return float(self.rollout_cost(x0, horizon, e, u.reshape(N, m))[0])

        def grad(u):
            g = np.zeros(N * m)
            for i in range(N * m):
                uc = u.astype(complex)
                uc[i] += 1e-30j
                g[i] = self.rollout_cost(x0, horizon, e, uc.reshape(N, m))[0].imag / 1e-30
            return g
        u0 = np.tile(np.where((np.abs(lb) < 1e19) & (np.abs(ub) < 1e19), 0.5 * (lb + ub), np.clip(0.0, lb, ub)), N) \
            if U_init is None else np.clip(np.asarray(U_init, dtype=float), lb, ub).ravel()
        r = minimize(fun, u0, jac=grad, method="L-BFGS-B", bounds=list(zip(np.tile(lb, N), np.tile(ub, N))),
                     options=dict(maxiter=5000, ftol=1e-16, gtol=tol, maxcor=30))
        U = r.x.reshape(N, m)
        J, X = self.rollout_cost(x0, horizon, e, U)
        lam = np.zeros((N + 1, n))
        lam[N] = self._call('dhx', tg[-1], X[N], e).ravel()
        for k in range(N - 1, -1, -1):
            _, _, M = self.grid_map(tg[k], X[k], U[k], e, DT, derivs=True)
            lam[k] = M[n, :n] + M[:n, :n].T @ lam[k + 1]
        self.last_cost = float(J)
        return tg, np.asarray(X), np.vstack([U, U[-1:]]), lam, float(J)

    def cocSolverStateBounded(self, ini_state, horizon, auxvar_value, state_lb, state_ub, control_lb=None, control_ub=None,
                              U_init=None, tol=1e-12):
        """The NLP of CPDP.py:110-175 with finite STATE bounds on the shooting nodes X_1..X_N (lbw / ubw of the X_k,
        CPDP.py:140-147; X_0 is pinned to ini_state) -- and optionally control bounds -- solved in single-shooting form by
        scipy's SLSQP: objective and node states from the plain RK4 roll-out, all derivatives by complex steps.  No code
        shared with the HIP solver's augmented-Lagrangian loop or with this oracle's own DDP.
        Returns (time_grid, state_grid, control_grid [N+1, last row repeated], J)."""
        from scipy.optimize import minimize
        self.diffPMP()
        n, m, N = self.n_state, self.n_control, self.n_grid
        e = np.asarray(auxvar_value, dtype=float).ravel()
        x0 = np.asarray(ini_state, dtype=float).ravel()
        xlb, xub = np.asarray(state_lb, dtype=float), np.asarray(state_ub, dtype=float)
        idx_l = [i for i in range(n) if abs(xlb[i]) < 1e19]
        idx_u = [i for i in range(n) if abs(xub[i]) < 1e19]
        tg = np.linspace(0, horizon, N + 1)

        def roll(u):
            J, X = self.rollout_cost(x0, horizon, e, u.reshape(N, m))
            return J, np.asarray(X)

        def cons_of(X):
            X = np.asarray(X)[1:]
            return np.concatenate([(X[:, idx_l] - xlb[idx_l]).ravel(), (xub[idx_u] - X[:, idx_u]).ravel()])

        def fun(u):
            return float(roll(u)[0])

        def both_jac(u):
            g = np.zeros(N * m)
            nc = N * (len(idx_l) + len(idx_u))
            Jc = np.zeros((nc, N * m))
            for i in range(N * m):
                uc = u.astype(complex)
                uc[i] += 1e-30j
                J, X = roll(uc)
                g[i] = J.imag / 1e-30
                Jc[:, i] = cons_of(X).imag / 1e-30
            return g, Jc
        cache = {}

        def cached(u):
            key = u.tobytes()
            if key not in cache:
                cache.clear()
                cache[key] = both_jac(u)
            return cache[key]
        bounds = None
        if control_lb is not None:
            bounds = list(zip(np.tile(np.asarray(control_lb, dtype=float), N), np.tile(np.asarray(control_ub, dtype=float), N)))
        u0 = np.zeros(N * m) if U_init is None else np.asarray(U_init, dtype=float).ravel()
        r = minimize(fun, u0, jac=lambda u: cached(u)[0], method="SLSQP", bounds=bounds,
                     constraints=[dict(type="ineq", fun=lambda u: cons_of(roll(u)[1]).real, jac=lambda u: cached(u)[1])],
                     options=dict(maxiter=2000, ftol=tol))
        if not r.success:
            raise RuntimeError("SLSQP: %s" % r.message)
        U = r.x.reshape(N, m)
        J, X = roll(r.x)
        self.last_cost = float(J)
        return tg, np.asarray(X), np.vstack([U, U[-1:]]), float(J)

    def kkt_certificate(self, ini_state, horizon, e, state_grid, control_grid, costate_grid, h=1e-30):
        """Solver-independent check that (X,U,lambda) is the KKT point of the NLP at CPDP.py:126-179.

        Uses only plain RK4 roll-outs with complex-step differentiation.  Returns
        (max dynamics defect, max |dJ/dU|, max |lambda_k - dJ_k/dx_k|).
        """
        self.diffPMP()
        N, n, m = self.n_grid, self.n_state, self.n_control
        e = np.asarray(e, dtype=float).ravel()
        U = np.array(control_grid[:N], dtype=float)
        DT = horizon / N / self.steps_per_grid
        tg = np.linspace(0, horizon, N + 1)
        _, X = self.rollout_cost(ini_state, horizon, e, U)
        defect = np.max(np.abs(X - state_grid))
        gmax = 0.0
        for k in range(N):
            for j in range(m):
                Uc = U.astype(complex)
                Uc[k, j] += 1j * h
                Jc, _ = self.rollout_cost(ini_state, horizon, e, Uc)
                gmax = max(gmax, abs(Jc.imag / h))
        lmax = 0.0
        for k in range(0, N + 1, max(1, N // 5)):
            for i in range(n):
                xc = np.array(state_grid[k], dtype=complex)
                xc[i] += 1j * h
                Jt = 0
                x = xc
                for kk in range(k, N):
                    x, q = self.grid_map(tg[kk], x, U[kk].astype(complex), e, DT)
                    Jt = Jt + q
                Jt = Jt + self._call('final', tg[-1], x, e)[0, 0]
                lmax = max(lmax, abs(Jt.imag / h - costate_grid[k, i]))
        return defect, gmax, lmax

    # ---- CPDP.py:253-276 ----------------------------------------------------
    def raccati_fn(self, t, x, u, lam, e, P, W):
        dfx = self._call('dfx', t, x, u, e)
        dfu = self._call('dfu', t, x, u, e)
        dfe = self._call('dfe', t, x, u, e)
        Hxx = self._call('ddHxx', t, x, u, lam, e)
        Hxu = self._call('ddHxu', t, x, u, lam, e)
        Hxe = self._call('ddHxe', t, x, u, lam, e)
        Huu = self._call('ddHuu', t, x, u, lam, e)
        Hue = self._call('ddHue', t, x, u, lam, e)
        invHuu = np.linalg.pinv(Huu)
        GinvHuu = dfu @ invHuu
        HxuinvHuu = Hxu @ invHuu
        A = dfx - GinvHuu @ Hxu.T
        R = GinvHuu @ dfu.T
        Q = Hxx - HxuinvHuu @ Hxu.T
        r = dfe - GinvHuu @ Hue
        q = Hxe - HxuinvHuu @ Hue
        P_dot = -(Q + A.T @ P + P @ A - P @ R @ P)
        W_dot = P @ R @ W - A.T @ W - P @ r - q
        return P_dot, W_dot

    # ---- CPDP.py:281-298 ----------------------------------------------------
    def auxsys_controller_fn(self, t, x, u, lam, e, P, W, Xa):
        dfu = self._call('dfu', t, x, u, e)
        Hux = self._call('ddHux', t, x, u, lam, e)
        Huu = self._call('ddHuu', t, x, u, lam, e)
        Hue = self._call('ddHue', t, x, u, lam, e)
        return -np.linalg.pinv(Huu) @ ((Hux + dfu.T @ P) @ Xa + dfu.T @ W + Hue)

    def auxsys_state_dot_fn(self, t, x, u, lam, e, P, W, Xa):
        dfx = self._call('dfx', t, x, u, e)
        dfu = self._call('dfu', t, x, u, e)
        dfe = self._call('dfe', t, x, u, e)
        Ua = self.auxsys_controller_fn(t, x, u, lam, e, P, W, Xa)
        return dfx @ Xa + dfu @ Ua + dfe

    # ---- CPDP.py:301-381 ----------------------------------------------------
    def auxSysSolver(self, time_grid, opt_sol, auxvar_value=1, riccati_method=None, ivp_kwargs=None,
                     return_grids=False):
        """Defaults reproduce the reference calls exactly: Riccati sweep with
        ``method='BDF'`` for COCSys (CPDP.py:335) / default RK45 for
        COCSys_TimeVarying (CPDP.py:740); aux state with default RK45
        (CPDP.py:368); scipy default rtol=1e-3, atol=1e-6.  ``ivp_kwargs`` (e.g.
        ``dict(rtol=1e-11, atol=1e-13)``) gives the 'tight' oracle the HIP path
        is compared against."""
        self.diffPMP()
        n, m, p, N = self.n_state, self.n_control, self.n_auxvar, self.n_grid
        e = np.asarray(auxvar_value, dtype=float).ravel()
        kw = dict(ivp_kwargs or {})
        if riccati_method is None:
            riccati_method = 'RK45' if self.time_varying else 'BDF'
        tq = lambda t: t if self.time_varying else 0.0

        def split(xulam):
            return xulam[0:n], xulam[n:n + m], xulam[n + m:]

        def vec_PW_ode(t, vec_PW):
            P = vec_PW[0:n * n].reshape(n, n)
            W = vec_PW[n * n:].reshape(n, -1)
            x, u, lam = split(opt_sol(t))
            P_dot, W_dot = self.raccati_fn(tq(t), x, u, lam, e, P, W)
            return np.concatenate((P_dot.ravel(), W_dot.ravel()))

        x, _, _ = split(opt_sol(float(time_grid[-1])))
        vec_PW_grid = np.zeros((N + 1, n * n + n * p))
        vec_PW_grid[-1, :] = np.concatenate((self._call('ddhxx', tq(time_grid[-1]), x, e).ravel(),
                                             self._call('ddhxe', tq(time_grid[-1]), x, e).ravel()))
        for k in range(N, 0, -1):
            t_span = [time_grid[k], time_grid[k - 1]]
            sol = solve_ivp(vec_PW_ode, t_span, vec_PW_grid[k, :], t_eval=[t_span[1]], method=riccati_method, **kw)
            if not sol.success:        # finite escape of the Riccati solution (a conjugate point): the reference would crash here too
                raise RuntimeError("Riccati sweep failed on interval %d: %s" % (k, sol.message))
            vec_PW_grid[k - 1, :] = sol.y.ravel()
        vec_PW_sol = self.interpolation(time_grid, vec_PW_grid)

        def PW_at(t):
            v = vec_PW_sol(t)
            return v[0:n * n].reshape(n, n), v[n * n:].reshape(n, -1)

        def vec_auxsys_state_ode(t, v):
            Xa = v.reshape(n, p)
            x, u, lam = split(opt_sol(t))
            P, W = PW_at(t)
            return self.auxsys_state_dot_fn(tq(t), x, u, lam, e, P, W, Xa).ravel()

        vec_X = np.zeros((N + 1, n * p))
        vec_U = np.zeros((N + 1, m * p))
        x, u, lam = split(opt_sol(0))
        P, W = PW_at(0)
        vec_U[0, :] = self.auxsys_controller_fn(tq(0.0), x, u, lam, e, P, W, vec_X[0].reshape(n, p)).ravel()
        for k in range(N):
            t_span = [time_grid[k], time_grid[k + 1]]
            sol = solve_ivp(vec_auxsys_state_ode, t_span, vec_X[k, :], t_eval=[time_grid[k + 1]], **kw)
            vec_X[k + 1, :] = sol.y.ravel()
            t1 = float(time_grid[k + 1])
            x, u, lam = split(opt_sol(t1))
            P, W = PW_at(t1)
            vec_U[k + 1, :] = self.auxsys_controller_fn(tq(t1), x, u, lam, e, P, W,
                                                        vec_X[k + 1].reshape(n, p)).ravel()
        aux = self.interpolation(time_grid, np.concatenate((vec_X, vec_U), axis=1))
        if return_grids:
            return aux, vec_PW_grid, vec_X, vec_U
        return aux

    # ---- CPDP.py:384-390 ----------------------------------------------------
    def interpolation(self, x, y, method=1):
        if method == 1:
            return ip.interp1d(x, y, axis=0)
        if method == 2:
            return ip.interp1d(x, y, axis=0, kind='cubic')


def COCSys_TimeVarying(project_name="myOc"):
    return COCSys(project_name, time_varying=True)


# ---- loss of the examples (lib/QuadAlgorithm.py:616-673, Examples/*.py) -------
def getloss_corrections(oc, time_grid, target_waypoints, opt_sol, auxsys_sol, interface_idx):
    """loss = sum_k ||y(tau_k) - waypoint_k||^2,  diff_loss = sum_k (y - wp)^T dy/dx dx/dtheta.

    ``interface_idx``: which state components the interface function exposes
    ([0,1,2] for getloss_pos_corrections; [0,1,2,6,7,8,9] for the rocket's
    getloss_corrections; [0] pendulum; [0,1] robot arm).  The reference's
    gradient carries no factor 2 (dl_dpos = current - target); kept."""
    n, p = oc.n_state, oc.n_auxvar
    loss = 0.0
    diff_loss = np.zeros(p)
    idx = None if interface_idx is None else list(interface_idx)      # None: the general interface function of oc.setInterface
    for k, t in enumerate(time_grid):
        target = np.asarray(target_waypoints[k], dtype=float).ravel()
        x = opt_sol(t)[0:n]
        if idx is None:
            cur = np.asarray(oc.interface_fn(x), dtype=float).ravel()
            dy_dx = np.asarray(oc.diff_interface_fn(x), dtype=float).reshape(len(cur), n)
        else:
            cur = x[idx]
            dy_dx = np.eye(n)[idx]
        loss += np.linalg.norm(target - cur) ** 2
        dl_dy = cur - target
        dx_dp = auxsys_sol(t)[0:n * p].reshape(n, p)
        diff_loss += dl_dy @ dy_dx @ dx_dp
    return loss, diff_loss


# ---- parameter update rules (lib/QuadAlgorithm.py:454-578) ---------------------
class Optimizer:
    """State + update rule of QuadAlgorithm.{Vanilla_gradient_descent,Nesterov,Adam,Nadam,AMSGrad}.

    ``lookahead(theta)`` is the point where (loss, grad) must be evaluated
    (theta itself, except Nesterov: theta + mu*v, QuadAlgorithm.py:478);
    ``step(theta, grad, idx)`` returns the new theta."""

    def __init__(self, method, n, learning_rate, mu=0.9, beta_1=0.9, beta_2=0.999, epsilon=1e-8):
        self.method, self.lr = method, learning_rate
        self.mu, self.b1, self.b2, self.eps = mu, beta_1, beta_2, epsilon
        self.v = np.zeros(n)
        self.m = np.zeros(n)
        self.vhat = np.zeros(n)
        if method not in ("Vanilla", "Nesterov", "Adam", "Nadam", "AMSGrad"):
            raise Exception("Wrong optimization method type!")

    def lookahead(self, theta):
        return theta + self.mu * self.v if self.method == "Nesterov" else theta

    def step(self, theta, g, iter_idx_now):
        g = np.asarray(g, dtype=float)
        idx = iter_idx_now + 1
        if self.method == "Vanilla":
            return theta - self.lr * g
        if self.method == "Nesterov":
            self.v = self.mu * self.v - self.lr * g
            return theta + self.v
        self.m = self.b1 * self.m + (1 - self.b1) * g
        self.v = self.b2 * self.v + (1 - self.b2) * g ** 2
        if self.method == "AMSGrad":
            self.vhat = np.maximum(self.vhat, self.v)
            return theta - self.lr * self.m / (np.sqrt(self.vhat) + self.eps)
        mhat = self.m / (1 - self.b1 ** idx)
        vhat = self.v / (1 - self.b2 ** idx)
        if self.method == "Adam":
            return theta - self.lr * mhat / (np.sqrt(vhat) + self.eps)
        # Nadam
        return theta - self.lr * (self.b1 * mhat + (1 - self.b1) / (1 - self.b1 ** idx) * g) / (np.sqrt(vhat) + self.eps)
