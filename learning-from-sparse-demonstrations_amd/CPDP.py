"""Host-side mirror of the reference's ``CPDP`` module (CPDP/CPDP.py).

Same classes, method names and argument meaning as the reference —
``COCSys`` (CPDP.py:9-390) and ``COCSys_TimeVarying`` (CPDP.py:394-786) —
with the numerics executed by the model's HIP library on the GPU:

=====================  =====================================================
reference               here
=====================  =====================================================
setDyn/setPathCost/...  keep sympy expressions (symbolic.SX instead of casadi.SX)
diffPMP/raccatiODE/     derivative code is *generated and compiled*
auxSysODE               (codegen.py) instead of building CasADi Functions
cocSolver               lfsd_coc_solve  (IPOPT -> batched DDP, same NLP)
auxSysSolver            lfsd_aux_solve  (solve_ivp BDF/RK45 -> split-step RK4
                        + Richardson on the same ODEs and interpolants)
=====================  =====================================================

Beyond the reference's one-trajectory calls there are ``*Batch`` methods that
keep everything in HBM for thousands of trajectories, and ``SparseDemoLearner``
that runs the outer learning iteration (loss, gradient, parameter update) of
the examples / lib/QuadAlgorithm.py on the device.
"""
import numpy as np
import sympy as sp
import scipy.interpolate as ip
import torch

from . import symbolic, codegen, runtime
from .runtime import LfsdError, ModelLibrary


class COCSys:
    time_varying = False
    # tests / tuning tools: force one mapping of the OC solve for every instance ("lockstep" | "wide"), whatever the
    # instance or the batch size would pick (the C ABI takes the choice as an argument; nothing reads the environment)
    mapping_override = None

    def __init__(self, project_name="myOc"):
        self.sys_name = project_name
        self.time = sp.Symbol('time', real=True)
        self.device = torch.device("cuda", 0) if torch.cuda.is_available() else None
        self.dtype = torch.float32
        self._lib = None
        self._lib_override = None
        self.aux_substeps = 0           # minimum split units per grid interval; 0 = library default (1 with aux_rtol > 0)
        self.aux_rtol = 1e-3            # error-controlled sub-stepping of the auxiliary sweeps at the reference's own
                                        # solve_ivp tolerance (CPDP.py:335, 368: scipy's default rtol); 0: fixed aux_substeps
        self.max_iter = 300              # (IPOPT's default is 3000; an iteration here is one DDP sweep)
        self.tol = None
        self.exact_after = 16            # iteration from which the exact stage Hessian is forced
        self.aux_dtype = None            # None: same as dtype; torch.float64: fp64 auxiliary (Riccati/sensitivity) pass
        self.mapping = "auto"            # "auto" | "lockstep" | "wide": mapping of the OC solve onto the machine
        # state bounds (augmented Lagrangian): first penalty, its growth, feasibility tolerance (None: 1e-7 fp64 / 1e-4 fp32,
        # relative to 1 + the largest finite bound), limit of outer iterations
        self.state_rho0, self.state_rho_growth, self.state_tol, self.state_max_outer = 10.0, 10.0, None, 40

    # ---- model definition (CPDP.py:15-87) ------------------------------------------------------
    def setAuxvarVariable(self, auxvar=None):
        if auxvar is None:
            auxvar = symbolic.SX.sym('auxvar')
        self.auxvar = symbolic._flat([auxvar])
        self.n_auxvar = len(self.auxvar)
        self._lib = None

    def setStateVariable(self, state, state_lb=[], state_ub=[]):
        """CPDP.py:20-31: bounds are used only when their length equals n_state (otherwise +-1e20, as the reference).
        Finite state bounds are the variable bounds lbw / ubw of the shooting nodes X_1..X_N in the reference's NLP
        (CPDP.py:140-147; X_0 is pinned to ini_state, :131-134), which IPOPT's interior point handles; here they are
        enforced by an augmented-Lagrangian outer loop around the batched solve (cocSolverBatch): every node gets a
        multiplier pair and a quadratic penalty, the kernels solve the subproblem, the host updates multipliers and penalty
        until the nodes are feasible to `state_tol`."""
        self.state = symbolic._flat([state])
        self.n_state = len(self.state)
        self.state_lb = [float(v) for v in state_lb] if len(state_lb) == self.n_state else self.n_state * [-1e20]
        self.state_ub = [float(v) for v in state_ub] if len(state_ub) == self.n_state else self.n_state * [1e20]
        if any(l > u for l, u in zip(self.state_lb, self.state_ub)):
            raise ValueError("state_lb > state_ub")
        self._lib = None

    def _state_bounds(self):
        """(lb, ub) device tensors, or (None, None) when no bound is finite."""
        lb, ub = getattr(self, "state_lb", None), getattr(self, "state_ub", None)
        if lb is None or not any(abs(v) < 1e19 for v in lb + ub):
            return None, None
        return self._t(lb), self._t(ub)

    def setControlVariable(self, control, control_lb=[], control_ub=[]):
        """CPDP.py:33-46: bounds are used only when their length equals n_control (otherwise +-1e20, as the reference).
        Finite control bounds are honoured by the control-limited backward sweep of the HIP solver."""
        self.control = symbolic._flat([control])
        self.n_control = len(self.control)
        self.control_lb = [float(v) for v in control_lb] if len(control_lb) == self.n_control else self.n_control * [-1e20]
        self.control_ub = [float(v) for v in control_ub] if len(control_ub) == self.n_control else self.n_control * [1e20]
        if any(l > u for l, u in zip(self.control_lb, self.control_ub)):
            raise ValueError("control_lb > control_ub")
        self._lib = None

    def _control_bounds(self):
        """(lb, ub) device tensors, or (None, None) when no bound is finite."""
        lb, ub = getattr(self, "control_lb", None), getattr(self, "control_ub", None)
        if lb is None or not any(abs(v) < 1e19 for v in lb + ub):
            return None, None
        return self._t(lb), self._t(ub)

    def setTimeVariable(self, t=None):
        self.time = t if t is not None else sp.Symbol('time', real=True)

    def setDyn(self, ode):
        if not hasattr(self, 'auxvar'):
            self.setAuxvarVariable()
        self.dyn = sp.Matrix(symbolic._flat([ode]))
        self._lib = None

    def setPathCost(self, path_cost):
        if not hasattr(self, 'auxvar'):
            self.setAuxvarVariable()
        self.path_cost = sp.sympify(path_cost)
        self._lib = None

    def setFinalCost(self, final_cost):
        if not hasattr(self, 'auxvar'):
            self.setAuxvarVariable()
        self.final_cost = sp.sympify(final_cost)
        self._lib = None

    def setInterface(self, interface=None):
        """The interface function y = g(x) of the sparse-demonstration loss as a symbolic expression of the state -- what the
        reference's examples build as ``Function('interface', [oc.state], [expr])`` together with its ``jacobian``
        (lib/QuadAlgorithm.py:616-639, Examples/robotarm_random.py:35-36).  It is compiled into the model library (g and
        (dg/dx)^T r as generated code); ``auxSysSolverBatch`` / ``SparseDemoLearner`` called with ``interface_idx=None`` use
        it.  Interfaces that merely select state components need none of this: pass ``interface_idx``."""
        self.interface = None if interface is None else symbolic._flat([interface])
        self._lib = None

    def setIntegrator(self, n_grid=10, steps_per_grid=4):
        self.n_grid = n_grid
        self.steps_per_grid = steps_per_grid

    # ---- extensions ---------------------------------------------------------------------------
    def setDevice(self, device=None, dtype=None, aux_dtype=None):
        """dtype: arithmetic of the OC solve; aux_dtype (optional): arithmetic of the differentiated-PMP pass, e.g.
        fp32 solve + fp64 Riccati/sensitivity sweeps (BASELINE configs[4])."""
        if device is not None:
            self.device = torch.device(device)
        if dtype is not None:
            self.dtype = dtype
        if aux_dtype is not None:
            self.aux_dtype = aux_dtype

    def setSolverOptions(self, max_iter=None, tol=None, aux_substeps=None, exact_after=None, aux_rtol=None, mapping=None):
        if max_iter is not None:
            self.max_iter = int(max_iter)
        if tol is not None:
            self.tol = float(tol)
        if aux_substeps is not None:
            self.aux_substeps = int(aux_substeps)
        if exact_after is not None:
            self.exact_after = int(exact_after)
        if aux_rtol is not None:
            self.aux_rtol = float(aux_rtol)
        if mapping is not None:
            if mapping not in runtime.MAPPINGS:
                raise LfsdError("mapping must be one of %s" % sorted(runtime.MAPPINGS))
            self.mapping = mapping

    def use_library(self, path_or_lib):
        """Bind an already built model library (tests use this to inject the SIMT-emulator build)."""
        self._lib_override = path_or_lib if isinstance(path_or_lib, ModelLibrary) else ModelLibrary(path_or_lib)
        self._lib = None

    # ---- model -> HIP library -------------------------------------------------------------------
    def model_spec(self, name=None):
        for attr in ('state', 'control', 'dyn', 'path_cost', 'final_cost'):
            assert hasattr(self, attr), {"state": "Define the state variable first!",
                                         "control": "Define the control variable first!",
                                         "dyn": "Define the system dynamics first!",
                                         "path_cost": "Define the running cost/reward function first!",
                                         "final_cost": "Define the final cost/reward function first!"}[attr]
        known = set(self.state) | set(self.control) | set(self.auxvar) | {self.time}
        iface = getattr(self, "interface", None)
        ifree = set().union(*[sp.sympify(g).free_symbols for g in iface]) if iface else set()
        free = (self.dyn.free_symbols | self.path_cost.free_symbols | self.final_cost.free_symbols | ifree) - known
        bad = [s for s in free if not symbolic.is_const(s)]
        if bad:
            raise LfsdError("free symbols that are neither state, control, auxvar nor const(): %s" % bad)
        consts = sorted(free, key=lambda s: int(str(s).rsplit('__k', 1)[1]))
        tv = self.time_varying and (self.time in (self.dyn.free_symbols | self.path_cost.free_symbols |
                                                  self.final_cost.free_symbols))
        return codegen.ModelSpec(self.state, self.control, self.auxvar, consts, self.time, self.dyn, self.path_cost,
                                 self.final_cost, time_varying=tv,
                                 const_defaults=[symbolic.const_default(s) for s in consts],
                                 name=name or self.sys_name, interface=iface)

    def compile(self, force=False, verbose=False):
        if self._lib is not None and not force:
            return self._lib
        spec = self.model_spec()
        if self._lib_override is not None:
            lib = self._lib_override
            if lib.hash != spec.hash():
                raise LfsdError("bound library was built for a different model (%s vs %s)" % (lib.hash, spec.hash()))
        else:
            lib = ModelLibrary(runtime.build_library(spec, force=force, verbose=verbose))
        self._spec = spec
        self._lib = lib
        self.const_values = list(spec.const_defaults)     # this instance's values; the library only fixes the structure
        return lib

    # the reference builds CasADi Functions here (CPDP.py:201-298); for us that is the code generator
    def diffPMP(self):
        self.compile()

    def raccatiODE(self):
        self.compile()

    def auxSysODE(self):
        self.compile()

    # ---- batched device API ------------------------------------------------------------------------
    def _dev(self):
        lib = self.compile()
        if lib.is_emulator:
            return torch.device("cpu")
        if self.device is None or self.device.type != "cuda":
            raise LfsdError("no GPU: the HIP solver has no CPU fallback")
        return self.device

    def _t(self, a, shape=None):
        t = torch.as_tensor(np.asarray(a, dtype=np.float64) if not isinstance(a, torch.Tensor) else a)
        t = t.to(device=self._dev(), dtype=self.dtype)
        if shape is not None:
            t = t.expand(shape) if t.dim() == len(shape) else t.reshape(shape)
        return t.contiguous()

    def consts_tensor(self, batch=None, overrides=None):
        """Runtime constants: shared [n_const] (batch=None) or per-trajectory [B][n_const]."""
        lib = self.compile()
        base = torch.tensor(self.const_values, dtype=torch.float64)
        if overrides:
            names = [str(s).rsplit('__k', 1)[0] for s in self._spec.consts]
            for k, v in overrides.items():
                idx = [i for i, nm in enumerate(names) if nm == k]
                if not idx:
                    raise LfsdError("unknown constant %r (have %s)" % (k, names))
                v = torch.as_tensor(v, dtype=torch.float64)
                if v.dim() == 0:
                    base[idx[0]] = v
                else:
                    if batch is None:
                        batch = v.shape[0]
                    if base.dim() == 1:
                        base = base.repeat(batch, 1)
                    base[:, idx[0]] = v
        if batch is not None and base.dim() == 1:
            base = base.repeat(batch, 1)
        if lib.n_const == 0:
            return None
        return base.to(device=self._dev(), dtype=self.dtype).contiguous()

    def cocSolverBatch(self, ini_state, horizon, auxvar, consts=None, u_init=None, workspace=None, out=None):
        """Solve B problems. ini_state [B,n], horizon scalar or [B], auxvar [B,p] (or [p]) -> dict of device tensors."""
        lib = self.compile()
        if not hasattr(self, 'n_grid'):
            self.setIntegrator()
        x0 = self._t(ini_state)
        if x0.dim() == 1:
            x0 = x0.unsqueeze(0)
        B = x0.shape[0]
        th = self._t(auxvar)
        if th.dim() == 1:
            th = th.unsqueeze(0).expand(B, -1).contiguous()
        hz = self._t(horizon)
        if hz.dim() == 0:
            hz = hz.expand(B).contiguous()
        if consts is None:
            consts = self.consts_tensor()
        clb, cub = self._control_bounds()
        slb, sub = self._state_bounds()
        kw = dict(max_iter=self.max_iter, tol=self.tol, exact_after=self.exact_after,
                  mapping=COCSys.mapping_override or self.mapping)
        if slb is None:
            sol = lib.coc_solve(x0, hz, th, consts, self.n_grid, self.steps_per_grid, u_init=u_init, workspace=workspace,
                                out=out, control_lb=clb, control_ub=cub, **kw)
        else:
            sol = self._solve_state_bounded(lib, x0, hz, th, consts, u_init, workspace, out, clb, cub, slb, sub, kw)
        sol.update(horizon=hz, auxvar=th, consts=consts, ini_state=x0, n_grid=self.n_grid)
        return sol

    def _solve_state_bounded(self, lib, x0, hz, th, consts, u_init, workspace, out, clb, cub, slb, sub, kw):
        """Augmented-Lagrangian outer loop for finite state bounds on the shooting nodes X_1..X_N (CPDP.py:140-147).
        Subproblem k (one lfsd_coc_solve, warm-started from the controls of subproblem k-1): the NLP plus, per node and state
        component, [max(0, lu + rho (x - ub))^2 - lu^2 + max(0, ll + rho (lb - x))^2 - ll^2] / (2 rho).  Then
        lu <- max(0, lu + rho (x - ub)), ll <- max(0, ll + rho (lb - x)); rho grows when the worst violation of the batch
        shrinks by less than 4x.  Done when every node of every trajectory is feasible to `state_tol` and the multipliers
        have stopped moving: the last subproblem's stationarity is then the KKT condition of the bounded NLP, with
        lu - ll the bound multipliers (IPOPT's lam_x) and the returned costates the dynamics multipliers (lam_g).

        Failure is reported per trajectory, as IPOPT would (infeasible problem / iteration limit): when the loop ends
        without meeting that test, every row that is still infeasible beyond `state_tol`, or whose last subproblem did not
        end CONVERGED / STALLED, gets status 3 (MAXITER) -- `mask_unconverged`, `cocSolver` and the tests decide on
        `status` alone.  The reference puts the bounds on X_0 as well (CPDP.py:126-134), so an `ini_state` outside a
        finite box makes its NLP infeasible: that raises here.  The returned `cost` is the AUGMENTED value of the last
        subproblem -- objective + multiplier terms, which at a node that violates its bound by g estimates the optimal value
        to O(g^2) where the plain objective of the (slightly infeasible) iterate is off by lambda g (measured against the
        independent SLSQP solve: 6e-12 against 1.5e-8); `cost_objective` is the plain objective of the returned grids."""
        B, n, N = x0.shape[0], lib.n_state, self.n_grid
        if bool(((x0 < slb) | (x0 > sub)).any()):
            raise LfsdError("ini_state violates the state bounds: the reference's NLP bounds X_0 too (CPDP.py:126-134) "
                            "and is infeasible")
        if clb is None:                     # the bounded kernel reads both boxes
            clb, cub = self._t(lib.n_control * [-1e20]), self._t(lib.n_control * [1e20])
        mult = torch.zeros((B, N, 2, n), dtype=x0.dtype, device=x0.device)
        rho = float(self.state_rho0)
        finite = torch.cat([slb[slb.abs() < 1e19], sub[sub.abs() < 1e19]])
        scale = 1.0 + float(finite.abs().max())
        tol = self.state_tol if self.state_tol is not None else (1e-7 if x0.dtype == torch.float64 else 1e-4)
        viol_prev, iters_total, sol = None, None, None
        met, viol_b = False, None
        for outer in range(int(self.state_max_outer)):
            sol = lib.coc_solve(x0, hz, th, consts, self.n_grid, self.steps_per_grid, u_init=u_init, workspace=workspace,
                                out=out, control_lb=clb, control_ub=cub, state_lb=slb, state_ub=sub, state_mult=mult,
                                state_rho=rho, **kw)
            workspace, out = sol["workspace"], {k: sol[k] for k in ("state_grid", "control_grid", "costate_grid", "cost",
                                                                   "iters", "status")}
            iters_total = sol["iters"].clone() if iters_total is None else iters_total + sol["iters"]
            X = sol["state_grid"][:, 1:, :]
            gu, gl = X - sub, slb - X                                        # <= 0 when feasible
            new_u = torch.clamp(mult[:, :, 0] + rho * gu, min=0.0)
            new_l = torch.clamp(mult[:, :, 1] + rho * gl, min=0.0)
            moved_b = torch.maximum((new_u - mult[:, :, 0]).abs().amax(dim=(1, 2)), (new_l - mult[:, :, 1]).abs().amax(dim=(1, 2)))      # per trajectory
            moved = float(moved_b.max())
            # the plain objective: the node terms the kernel added, at the multipliers and penalty it was called with
            pen = ((torch.clamp(mult[:, :, 0] + rho * gu, min=0.0) ** 2 - mult[:, :, 0] ** 2
                    + torch.clamp(mult[:, :, 1] + rho * gl, min=0.0) ** 2 - mult[:, :, 1] ** 2) / (2.0 * rho)).sum(dim=(1, 2))
            mult = torch.stack((new_u, new_l), dim=2).contiguous()
            viol_b = torch.clamp(torch.maximum(gu, gl), min=0.0).amax(dim=(1, 2))       # per trajectory
            viol = float(viol_b.max())
            solved = bool(((sol["status"] == 1) | (sol["status"] == 2)).all())
            rho_last = rho
            if solved and viol <= tol * scale and moved <= tol * scale * rho:
                met = True
                break
            if viol_prev is not None and viol > 0.25 * viol_prev and rho < 1e8:
                rho *= float(self.state_rho_growth)
            viol_prev = viol
            u_init = sol["control_grid"][:, :-1].contiguous()
        sol["iters"] = iters_total
        sol["cost_objective"] = sol["cost"] - pen.to(sol["cost"].dtype)
        if not met:         # outer-iteration limit or penalty cap: infeasible / non-KKT rows must not pass for solved
            # (a row whose own multipliers were still moving in the last update is not a KKT point of the bounded NLP either,
            #  however feasible and converged its last subproblem was)
            bad = (viol_b > tol * scale) | (moved_b > tol * scale * rho_last) | ~((sol["status"] == 1) | (sol["status"] == 2))
            sol["status"] = torch.where(bad & (sol["status"] != 4), torch.full_like(sol["status"], 3), sol["status"])
        sol["state_mult"], sol["state_rho"], sol["al_outer"], sol["state_violation"] = mult, rho, outer + 1, viol
        sol["state_violation_rows"] = viol_b
        return sol

    def check_waypoints(self, taus, horizon, interface_idx):
        """Host-side validation the kernels do not repeat.  The reference's opt_sol(t) is scipy's interp1d (CPDP.py:386),
        which raises ValueError for t outside [0, horizon]; its interface functions are arbitrary CasADi expressions,
        ours select state components only."""
        lib = self.compile()
        if interface_idx is None:
            if lib.n_interface == 0:
                raise LfsdError("no interface_idx given and no interface function set (COCSys.setInterface)")
        else:
            idx = [int(i) for i in interface_idx]
            if any(i < 0 or i >= lib.n_state for i in idx):
                raise LfsdError("interface_idx %s outside [0, n_state=%d)" % (idx, lib.n_state))
        tt = torch.as_tensor(taus, dtype=torch.float64).cpu() if not isinstance(taus, torch.Tensor) else taus.double().cpu()
        hz = torch.as_tensor(horizon, dtype=torch.float64).cpu() if not isinstance(horizon, torch.Tensor) else horizon.double().cpu()
        hz = hz.reshape(-1, 1) if (hz.dim() >= 1 and tt.dim() == 2 and hz.numel() == tt.shape[0]) else hz.min()
        if tt.numel() and (bool((tt < 0).any()) or bool((tt > hz * (1 + 1e-12)).any())):
            raise ValueError("A value in taus is outside the interpolation range [0, horizon].")

    def auxSysSolverBatch(self, sol, taus=None, waypoints=None, interface_idx=None, auxvar=None, want_grids=False,
                          Z_grid=None, out=None, phase_hook=None, validate=True, skip_status=None):
        """Differentiate the PMP along ``sol`` and (optionally) evaluate the sparse-waypoint loss + gradient.
        ``skip_status``: OC-solve statuses whose rows are NOT differentiated (NaN loss / gradient, no sweep).  Default:
        FAILED (4) only -- a solve that ended with non-finite grids has nothing to differentiate and would otherwise hold
        the launch at the refinement cap; a solve at the iteration limit (3) is differentiated as the reference does,
        unless the caller (SparseDemoLearner with skip_unconverged) says otherwise."""
        lib = self.compile()
        B = sol["state_grid"].shape[0]
        th = sol["auxvar"] if auxvar is None else self._t(auxvar, (B, lib.n_auxvar))
        tt = wp = ii = None
        if taus is not None:
            tt = self._t(taus)
            if tt.dim() == 1:
                tt = tt.unsqueeze(0).expand(B, -1).contiguous()
            wp = self._t(waypoints)
            if wp.dim() == 2:
                wp = wp.unsqueeze(0).expand(B, -1, -1).contiguous()
            ii = None if interface_idx is None else torch.as_tensor(list(interface_idx), dtype=torch.int32, device=self._dev())
            if validate:          # (a device->host read: callers that validated at setup switch it off)
                self.check_waypoints(tt, sol["horizon"], interface_idx)
        hz, cs, X, U, Lm = sol["horizon"], sol["consts"], sol["state_grid"], sol["control_grid"], sol["costate_grid"]
        ad = self.aux_dtype
        if ad is not None and ad != X.dtype:          # mixed precision: promote the solved grids for the aux pass
            cv = lambda t: None if t is None else t.to(ad).contiguous()
            hz, th, cs, X, U, Lm, tt, wp = (cv(t) for t in (hz, th, cs, X, U, Lm, tt, wp))
        status = sol.get("status")
        if skip_status is None:
            skip_status = (4,)
        if status is None:
            skip_status = ()
        return lib.aux_solve(hz, th, cs, X, U, Lm, tt, wp, ii, substeps=self.aux_substeps, want_grids=want_grids,
                             Z_grid=Z_grid, out=out, phase_hook=phase_hook, rtol=self.aux_rtol, oc_status=status,
                             skip_status=skip_status)

    # ---- the reference's one-trajectory calls --------------------------------------------------------
    def cocSolver(self, ini_state, horizon, auxvar_value=1, interplation_level=1, print_level=0):
        """CPDP.py:92-198: returns (time_grid, opt_sol) with opt_sol(t) -> [x, u, lambda]."""
        if not hasattr(self, 'n_grid'):
            self.setIntegrator()
        if type(ini_state) is list:
            ini_state = np.array(ini_state).flatten()
        e = np.atleast_1d(np.asarray(auxvar_value, dtype=np.float64)).ravel()
        sol = self.cocSolverBatch(np.asarray(ini_state, dtype=np.float64)[None, :], float(horizon), e[None, :])
        self.last_solution = sol
        st = int(sol["status"][0])
        if print_level:
            print("lfsd coc_solve: status=%s iters=%d cost=%g" % (runtime.STATUS.get(st, st), int(sol["iters"][0]),
                                                                 float(sol["cost"][0])))
        time_grid = np.linspace(0, float(horizon), self.n_grid + 1)
        grids = np.concatenate([sol[k][0].double().cpu().numpy() for k in ("state_grid", "control_grid",
                                                                           "costate_grid")], axis=1)
        return time_grid, self.interpolation(time_grid, grids, interplation_level)

    def auxSysSolver(self, time_grid, opt_sol, auxvar_value=1):
        """CPDP.py:301-381: returns auxsys_sol(t) -> [vec(dx/dtheta) (n*p, row-major), vec(du/dtheta) (m*p)]."""
        lib = self.compile()
        n, m, p = lib.n_state, lib.n_control, lib.n_auxvar
        # The reference integrates the auxiliary ODEs along WHATEVER interpolant it is handed (CPDP.py:320, 347); the sweeps here
        # differentiate along the LINEAR interpolant of the grid values (interplation_level 1, what every example uses).  A cubic
        # opt_sol (cocSolver(..., interplation_level=2), CPDP.py:388-390) would silently be resampled to that -- refuse it instead.
        # (interpolation() tags what it returns with `lfsd_level`; an interpolant from elsewhere is probed instead: a piecewise-linear
        #  one is reproduced by the linear interpolant of its own grid values at the interval midpoints.  Nothing relies on scipy's
        #  private attributes.)
        level = getattr(opt_sol, "lfsd_level", None)
        if level is None:
            tg_ = np.asarray(time_grid, dtype=np.float64)
            mid = 0.5 * (tg_[:-1] + tg_[1:])
            gv = np.asarray(opt_sol(tg_), dtype=np.float64)
            gm = np.asarray(opt_sol(mid), dtype=np.float64)
            lin = 0.5 * (gv[:-1] + gv[1:])
            level = 1 if np.abs(gm - lin).max() <= 1e-9 * max(1.0, np.abs(gv).max()) else "non-linear"
        if level != 1:
            raise LfsdError("auxSysSolver: opt_sol is not the linear interpolant of its grid (interplation level %r); the HIP sweeps "
                            "integrate along the linear interpolant (interplation_level=1, CPDP.py:386) only" % (level,))
        time_grid = np.asarray(time_grid, dtype=np.float64)
        N = len(time_grid) - 1
        g = np.asarray(opt_sol(time_grid), dtype=np.float64)
        e = np.atleast_1d(np.asarray(auxvar_value, dtype=np.float64)).ravel()
        sol = dict(state_grid=self._t(g[None, :, 0:n]), control_grid=self._t(g[None, :, n:n + m]),
                   costate_grid=self._t(g[None, :, n + m:]), horizon=self._t([time_grid[-1] - time_grid[0]]),
                   auxvar=self._t(e[None, :]), consts=self.consts_tensor())
        aux = self.auxSysSolverBatch(sol, want_grids=True)
        self.last_aux = aux
        X = aux["auxX_grid"][0].double().cpu().numpy().transpose(0, 2, 1).reshape(N + 1, n * p)
        U = aux["auxU_grid"][0].double().cpu().numpy().transpose(0, 2, 1).reshape(N + 1, m * p)
        return self.interpolation(time_grid, np.concatenate((X, U), axis=1))

    def interpolation(self, x, y, method=1):
        """CPDP.py:384-390."""
        if method == 1:
            f = ip.interp1d(x, y, axis=0)
        elif method == 2:
            f = ip.interp1d(x, y, axis=0, kind='cubic')
        else:
            return None                     # (the reference falls off the end of its ifs as well)
        f.lfsd_level = method              # read by auxSysSolver: level 2 is returned, as in the reference, but not differentiated along
        return f


class COCSys_TimeVarying(COCSys):
    """CPDP.py:394-786 — dynamics / costs may depend on the time symbol given to ``setTimeVariable``."""
    time_varying = True


class SparseDemoLearner:
    """The outer learning iteration of the examples, batched on the device.

    One iteration (Examples/robotarm_random.py:67-73, lib/QuadAlgorithm.py:454-578):
        solve OC at theta -> differentiate PMP -> waypoint loss & gradient -> parameter update -> projection.
    ``mode='independent'``: every trajectory (seed) owns its theta and optimizer state.
    ``mode='shared'``: one theta for all demonstrations; the gradient is summed over the batch and
    all-reduced over ``process_group`` (RCCL) before a single update.

    ``skip_unconverged`` (default: OFF in ``independent`` mode = the reference's behaviour, every gradient is applied
    to its own seed; ON in ``shared`` mode, where ONE non-finite or unconverged demonstration would otherwise be summed,
    all-reduced and applied to the single theta of every rank).  When switched on, a
    trajectory whose optimal-control solve ended at the iteration limit or failed, or whose loss / gradient is not
    finite, is frozen for that step: its row is masked out of the update kernel (parameters AND optimizer state stay
    untouched, for every update rule), in shared mode it is left out of the summed loss / gradient, and the number of
    dropped demonstrations is all-reduced and reported as ``n_unconverged``.  At the next outer iteration such a solve
    is continued from the controls it stopped at; every other trajectory cold-starts as in the reference.

    ``event_hook(name)``, if set, is called right before each device phase of ``step`` ("oc_solve", "aux_riccati",
    "aux_forward", "update") and once after the last one ("end"), so a caller can bracket the kernels with HIP events
    (bench.py) without re-implementing the iteration.
    """

    def __init__(self, oc, ini_state, horizon, taus, waypoints, interface_idx, theta0, method="Vanilla",
                 learning_rate=1e-2, mu=0.9, beta_1=0.9, beta_2=0.999, epsilon=1e-8, proj_lo=None, consts=None,
                 mode="independent", process_group=None, true_loss_print_flag=False, warm_start=False,
                 skip_unconverged=None):
        self.oc, self.method, self.lr, self.mu = oc, method, learning_rate, mu
        self.b1, self.b2, self.eps = beta_1, beta_2, epsilon
        if method not in runtime.OPT_METHODS:
            raise Exception("Wrong optimization method type!")
        self.mode, self.pg = mode, process_group
        self.lib = oc.compile()
        x0 = oc._t(ini_state)
        self.x0 = x0.unsqueeze(0) if x0.dim() == 1 else x0
        B = self.B = self.x0.shape[0]
        p = self.lib.n_auxvar
        hz = oc._t(horizon)
        self.hz = hz.expand(B).contiguous() if hz.dim() == 0 else hz
        tt = oc._t(taus)
        self.taus = tt.unsqueeze(0).expand(B, -1).contiguous() if tt.dim() == 1 else tt
        wp = oc._t(waypoints)
        self.wps = wp.unsqueeze(0).expand(B, -1, -1).contiguous() if wp.dim() == 2 else wp
        self.iface = None if interface_idx is None else torch.as_tensor(list(interface_idx), dtype=torch.int32, device=self.x0.device)
        oc.check_waypoints(self.taus, self.hz, interface_idx)
        th = oc._t(theta0)
        th = th.unsqueeze(0) if th.dim() == 1 else th
        if mode == "shared":
            assert th.shape[0] == 1, "shared mode keeps a single parameter vector"
            self.theta = th.clone()
        else:
            self.theta = th.expand(B, p).contiguous().clone()
        self.consts = consts if consts is not None else oc.consts_tensor()
        z = lambda: torch.zeros_like(self.theta)
        self.m, self.v, self.vhat = z(), z(), z()
        lo = torch.full((p,), -float("inf"), dtype=torch.float64)
        for k, val in (proj_lo or {0: 1e-8}).items():      # examples: current_parameter[0] = fmax(., 1e-8)
            lo[k] = val
        self.proj_lo = lo.to(device=self.x0.device, dtype=self.theta.dtype)
        self.iter_idx = 0
        self.true_loss = true_loss_print_flag
        # warm_start: start every OC solve from the previous iteration's controls (theta moves little per step).
        # The reference cold-starts IPOPT every time; the converged KKT point is the same, only the path to it is shorter.
        self.warm_start = warm_start
        self.skip_unconverged = (mode == "shared") if skip_unconverged is None else bool(skip_unconverged)
        self.count_unconverged = True      # one small device->host read per step; switch off inside timed loops
        self.n_unconverged = 0
        self.n_bad_device = None
        self.event_hook = None
        self._ok = None
        self._ws = None
        self._sol = None
        self._aux = None
        self._Z = None

    def evaluate(self, theta):
        """(loss [B], grad [B,p]) of every trajectory at parameters theta ([B,p] or [1,p])."""
        th = theta if theta.shape[0] == self.B else theta.expand(self.B, -1).contiguous()
        u_init = None
        if self.warm_start and self._sol is not None:
            u_init = self._sol["control_grid"][:, :-1].contiguous()
        elif self.skip_unconverged and self._sol is not None:
            # a solve that ran out of iterations is continued from where it stopped instead of restarted (its parameters
            # did not move, a cold start would fail the same way); all others start from zero controls = cold start
            cont = (self._sol["status"] == 3).reshape(-1, 1, 1)
            prev = self._sol["control_grid"][:, :-1]
            u_init = torch.where(cont & torch.isfinite(prev), prev, torch.zeros_like(prev)).contiguous()
        hook = self.event_hook
        if hook is not None:
            hook("oc_solve")
        self._sol = self.oc.cocSolverBatch(self.x0, self.hz, th, consts=self.consts, u_init=u_init,
                                           workspace=self._ws, out=self._sol_out())
        self._ws = self._sol["workspace"]
        phase = None if hook is None else (lambda nm: hook("aux_" + nm) if nm != "end" else None)
        # a learner that freezes unconverged rows anyway does not pay for differentiating them (they are masked by their
        # status below): the diverged seeds of a fixed learning rate otherwise hold the Riccati launch 20x longer
        self._aux = self.oc.auxSysSolverBatch(self._sol, self.taus, self.wps, self.iface, Z_grid=self._Z,
                                              out=self._aux_out(), phase_hook=phase, validate=False,
                                              skip_status=(3, 4) if self.skip_unconverged else None)
        self._Z = self._aux["Z_grid"]
        loss, grad = self._aux["loss"].to(self.theta.dtype), self._aux["grad"].to(self.theta.dtype)
        if self.skip_unconverged:
            loss, grad = self.mask_unconverged(self._sol["status"], loss, grad)
        return loss, grad

    def mask_unconverged(self, status, loss, grad):
        """Rows whose OC solve neither converged (1) nor stalled at working precision (2), whose loss / gradient is
        not finite (parameters that have left the region where the problem is well posed, e.g. a cost weight driven
        negative), or whose sensitivity sweeps report an interval accepted above `aux_rtol` (the `stats` output of
        lfsd_aux_solve: refinement stopped gaining next to a conjugate point) are frozen for this step: ``self._ok`` masks them out of the update kernel; their gradient (and, in
        shared mode, their loss, which enters a sum) is zeroed."""
        ok = ((status == 1) | (status == 2)) & torch.isfinite(loss) & torch.isfinite(grad).all(dim=1)
        st = self._aux.get("stats") if self._aux is not None else None
        if st is not None:      # ... or whose auxiliary sweeps accepted an interval above their tolerance (next to a conjugate point)
            ok = ok & ((st[:, 1] + st[:, 3]) == 0)
        self._ok = ok
        grad = torch.where(ok.unsqueeze(1), grad, torch.zeros_like(grad))
        if self.mode == "shared":
            loss = torch.where(ok, loss, torch.zeros_like(loss))
        return loss, grad

    def _sol_out(self):
        if self._sol is None:
            return None
        return {k: self._sol[k] for k in ("state_grid", "control_grid", "costate_grid", "cost", "iters", "status")}

    def _aux_out(self):
        if self._aux is None:
            return None
        return {k: self._aux[k] for k in ("loss", "grad", "stats")}

    def step(self):
        """One outer iteration; returns (loss, grad) evaluated where the update rule needs them."""
        theta_eval = self.theta
        if self.method == "Nesterov":
            theta_eval = self.lib.lookahead(self.theta, self.m, self.mu)      # QuadAlgorithm.py:478
        self._ok = None
        loss, grad = self.evaluate(theta_eval)
        hook = self.event_hook
        if hook is not None:
            hook("update")
        row_active = None
        if self.mode == "shared":
            g = grad.sum(dim=0, keepdim=True)
            l = loss.sum().reshape(1)
            n_bad = (self.B - self._ok.sum()).to(g.dtype).reshape(1) if self._ok is not None else torch.zeros_like(l)
            if self.pg is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
                buf = torch.cat([g.reshape(-1), l, n_bad])
                torch.distributed.all_reduce(buf, group=self.pg)              # RCCL over xGMI on the GPU
                g, l, n_bad = buf[:-2].reshape(1, -1), buf[-2:-1], buf[-1:]
            grad_used, loss_out = g.contiguous(), l
            self.n_bad_device = n_bad                                         # (device tensor: read it outside timed loops)
            if self.skip_unconverged and self.count_unconverged:
                self.n_unconverged = int(round(n_bad.item()))                 # over all ranks
        else:
            grad_used, loss_out = grad, loss
            if self._ok is not None:
                row_active = self._ok.to(torch.int32)
                if self.count_unconverged:
                    self.n_unconverged = int(self.B - self._ok.sum().item())
        self.lib.optimizer_step(self.method, self.theta, grad_used, self.iter_idx, self.lr, self.mu, self.b1,
                                self.b2, self.eps, m=self.m, v=self.v, vhat=self.vhat, proj_lo=self.proj_lo,
                                row_active=row_active)
        self.iter_idx += 1
        if hook is not None:
            hook("end")
        if self.method == "Nesterov" and self.true_loss:
            loss_out, grad_used = self.evaluate(self.theta)                   # QuadAlgorithm.py:487-492
            if self.mode == "shared":
                loss_out, grad_used = loss_out.sum().reshape(1), grad_used.sum(dim=0, keepdim=True)
        return loss_out, grad_used
