"""CPU tier: the HIP kernel sources compiled against the SIMT emulator (tests/emu) vs the oracle.

This exercises the exact kernel logic (lane-group layout, LDS exchanges, DDP iteration, split-step
Riccati / sensitivity sweeps, loss, optimizers) without a GPU.  The -m gpu tier repeats the same
comparisons through the real gfx950 libraries."""
import os

import numpy as np
import pytest
import torch

import lfsd_amd  # noqa: F401
from lfsd_amd import CPDP, JinEnv, models, runtime
from lfsd_amd.symbolic import SX, vertcat
from conftest import make_oracle, oracle_loss_grad

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "uav_golden.npz"))

# tolerances (relative to the largest component), written out as the task requires:
#   fp64: OC grids 1e-6, loss 1e-7, gradient 1e-4, aux grids (Z, dx/dtheta, du/dtheta) 1e-3 at 16 substeps
#         (the split-step + Richardson sweeps converge at 4th order in `substeps`: see test_aux_sweeps_converge)
#   fp32: loss 1e-4, gradient 5e-3, aux grids 1e-2  (fp64 -> fp32 tolerance of the whole pipeline)
TOL = {torch.float64: dict(grid=1e-6, loss=1e-7, grad=1e-4, aux=1e-3),
       torch.float32: dict(grid=5e-3, loss=1e-4, grad=5e-3, aux=1e-2)}


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def run_case(emu, kind, dtype, thetas, taus, wps, n_grid, batch_pad=0, substeps=8, rtol=None):
    oc, env, d = models.ZOO[kind](n_grid=n_grid)
    emu(oc)
    oc.setDevice(dtype=dtype)
    oc.setSolverOptions(aux_substeps=substeps, aux_rtol=rtol)
    thetas = np.atleast_2d(np.asarray(thetas, dtype=np.float64))
    B = thetas.shape[0]
    sol = oc.cocSolverBatch(np.tile(d["ini_state"], (B, 1)), d["horizon"], thetas)
    aux = oc.auxSysSolverBatch(sol, taus, wps, d["interface"], want_grids=True)
    return oc, d, sol, aux


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_pendulum_full_pipeline_vs_oracle(emu, dtype, oc_mapping):
    thetas = [[1.0, 0.5, 1.5], [2.0, 1.0, 1.0], [0.7, 1.3, 0.6]]      # B=3: ragged vs 8 groups per block
    taus, wps = [0.0, 0.3, 0.6, 0.7, 1.0], [[0.0], [1.2], [2.1], [2.4], [2.9]]   # endpoints + a grid point
    oc, d, sol, aux = run_case(emu, "pendulum", dtype, thetas, taus, wps, 10, substeps=16)
    o = make_oracle("pendulum", 10)
    t = TOL[dtype]
    assert set(sol["status"].tolist()) <= {1, 2}
    for b, th in enumerate(thetas):
        r = oracle_loss_grad(o, d["ini_state"], d["horizon"], th, taus, wps, d["interface"])
        assert rel(sol["state_grid"][b], r["X"]) < t["grid"]
        assert rel(sol["control_grid"][b], r["U"]) < t["grid"]
        assert rel(sol["costate_grid"][b], r["L"]) < 10 * t["grid"]
        n, p = 2, 3
        Zo = np.concatenate([r["PW"][:, :n * n].reshape(-1, n, n), r["PW"][:, n * n:].reshape(-1, n, p)], axis=2)
        assert rel(aux["Z_grid"][b].numpy().transpose(0, 2, 1), Zo) < t["aux"]
        assert rel(aux["auxX_grid"][b].numpy().transpose(0, 2, 1).reshape(-1, n * p), r["vX"]) < t["aux"]
        assert rel(aux["auxU_grid"][b].numpy().transpose(0, 2, 1).reshape(-1, 1 * p), r["vU"]) < t["aux"]
        assert abs(aux["loss"][b].item() - r["loss"]) < t["loss"] * max(1.0, r["loss"])
        assert rel(aux["grad"][b], r["grad"]) < t["grad"]


def test_aux_sweeps_converge_at_fourth_order(emu):
    """fp64 (classical RK4 inside the Strang split): halving the split-step size must cut the gradient error ~16x (Strang +
    Richardson) until round-off.  (The fp32 kernels use the explicit midpoint rule, LFSD_AUX_RK32: third order, below
    their rounding floor either way.)"""
    thetas, taus, wps = [[2.0, 1.0, 1.0]], [0.1, 0.3, 0.6, 0.7, 0.9], [[0.4], [1.2], [2.1], [2.4], [2.9]]
    o = make_oracle("pendulum", 10)
    r = None
    errs = []
    for sub in (4, 8, 16):
        oc, d, sol, aux = run_case(emu, "pendulum", torch.float64, thetas, taus, wps, 10, substeps=sub, rtol=0.0)   # fixed units
        r = r or oracle_loss_grad(o, d["ini_state"], d["horizon"], thetas[0], taus, wps, d["interface"])
        errs.append(rel(aux["grad"][0], r["grad"]))
    assert errs[0] / errs[1] > 8 and errs[1] / errs[2] > 8 and errs[2] < 2e-5, errs


def test_error_controlled_substepping_beats_fixed_units(emu):
    """aux_rtol > 0 (the default is 1e-3, scipy's default rtol with which the reference calls solve_ivp): an interval is redone with twice the split units while the Richardson estimate
    |fine - coarse| / 3 exceeds rtol.  On the coarse pendulum grid (10 intervals of 0.1-0.2 s) one unit per interval is far
    too few and four -- the fixed round-1 default -- leave 1e-3; the controlled sweep must deliver its tolerance class from
    the library defaults (no aux_substeps), and a tighter rtol must not be less accurate."""
    thetas, taus, wps = [[2.0, 1.0, 1.0]], [0.1, 0.3, 0.6, 0.7, 0.9], [[0.4], [1.2], [2.1], [2.4], [2.9]]
    o = make_oracle("pendulum", 10)
    err = {}
    r = None
    for key, sub, rtol in (("fixed4", 4, 0.0), ("default", 0, None), ("rtol1e-4", 0, 1e-4), ("rtol1e-6", 0, 1e-6)):
        oc, d, sol, aux = run_case(emu, "pendulum", torch.float64, thetas, taus, wps, 10, substeps=sub, rtol=rtol)
        r = r or oracle_loss_grad(o, d["ini_state"], d["horizon"], thetas[0], taus, wps, d["interface"])
        n, p = 2, 3
        err[key] = (rel(aux["grad"][0], r["grad"]), rel(aux["auxX_grid"][0].numpy().transpose(0, 2, 1).reshape(-1, n * p), r["vX"]))
    assert err["fixed4"][0] > 5e-4                                    # what the fixed default leaves on this grid
    # library default, rtol 1e-3: both errors inside the requested tolerance, the gradient well inside (the controller does
    # not integrate finer than the tolerance asks for -- LFSD_AUX_DOWN -- so dx/dtheta sits at 6e-4 here, not far below)
    assert err["default"][0] < 2e-4 and err["default"][1] < 1e-3, err
    assert err["rtol1e-4"][0] < 1e-5 and err["rtol1e-4"][1] < 5e-5, err
    assert err["rtol1e-6"][0] <= err["rtol1e-4"][0] * 1.5 and err["rtol1e-6"][1] < 1e-5, err


def test_robotarm_and_cartpole_fp32(emu):
    for kind, n_grid, thetas, taus, wps in (
            ("robotarm", 12, [[5., 1, 1, 1, 1], [3., 0.5, 2, 1.5, 0.2]], [0.3], [[-np.pi / 4, 2 * np.pi / 3]]),
            ("cartpole", 10, [[1.0, 0.5, 0.5, 0.5, 0.5], [0.8, 2, 0.3, 1, 1]], [0.25, 0.8], [[0.1, 0.5], [0.0, 2.5]])):
        oc, d, sol, aux = run_case(emu, kind, torch.float32, thetas, taus, wps, n_grid)
        o = make_oracle(kind, n_grid)
        for b, th in enumerate(thetas):
            r = oracle_loss_grad(o, d["ini_state"], d["horizon"], th, taus, wps, d["interface"])
            # fp32 tolerance on these ill-conditioned (flat-cost) problems: loss 2e-3, gradient 2e-2;
            # the same cases in fp64 meet 1e-7 / 1e-4 (test below)
            assert abs(aux["loss"][b].item() - r["loss"]) < 2e-3 * max(1.0, r["loss"]), kind
            assert rel(aux["grad"][b], r["grad"]) < 2e-2, kind
        oc, d, sol, aux = run_case(emu, kind, torch.float64, thetas, taus, wps, n_grid, substeps=16)
        for b, th in enumerate(thetas):
            r = oracle_loss_grad(o, d["ini_state"], d["horizon"], th, taus, wps, d["interface"])
            assert abs(aux["loss"][b].item() - r["loss"]) < 1e-6 * max(1.0, r["loss"]), kind
            assert rel(aux["grad"][b], r["grad"]) < 1e-4, kind


def test_quadrotor_against_reference_golden_run(emu):
    """HIP kernels (emulated, fp64) vs the numbers the reference itself produced (CasADi+IPOPT+solve_ivp).
    Tolerance 1e-2 on the gradient: the reference's own solve_ivp error (rtol=1e-3) is 5e-4 .. 5e-3."""
    oc, env, d = models.quadrotor(n_grid=int(G["n_grid"]))
    emu(oc)
    oc.setDevice(dtype=torch.float64)
    consts = oc.consts_tensor(overrides=dict(goal_r0=G["goal_r"][0], goal_r1=G["goal_r"][1], goal_r2=G["goal_r"][2]))
    idx = [0, 60, 99]
    sol = oc.cocSolverBatch(np.tile(G["ini_state"], (3, 1)), float(G["horizon"]), G["lookahead_theta"][idx],
                            consts=consts)
    aux = oc.auxSysSolverBatch(sol, G["taus"], G["waypoints"], [0, 1, 2])
    for k, j in enumerate(idx):
        assert abs(aux["loss"][k].item() - G["loss_trace"][j]) < 1e-6 * G["loss_trace"][j]
        assert rel(aux["grad"][k], G["grad_trace"][j]) < 1e-2
    # and the final optimal trajectory the reference saved (QuadAlgorithm.py:306-317)
    oc.const_values = consts.tolist()
    tg, opt = oc.cocSolver(G["ini_state"], float(G["horizon"]), G["theta_trace"][-1])
    tr = opt(G["time_steps"])
    assert np.abs(tr[:, :13] - G["opt_state_traj"]).max() < 1e-6
    assert np.abs(tr[:, 13:17] - G["opt_control_traj"]).max() < 1e-6


def test_quadrotor_fp32_packed_rollout_matches_fp64(emu, monkeypatch):
    """The fp32 lean OC kernel of the 32-lane models rolls out on 16-lane groups with two tangent columns per lane
    (oc_solve_kernel<..., PK=true>, four trajectories per workgroup, backward sweep on the (emulated) matrix cores); the fp64
    lean kernel rolls out on 16-lane groups too, one LIVE column per lane with the between-stage values of the tangent step
    parked in LDS (rollout_sens_live, rk4_step_parked).  Same inputs, ragged batch of 5 (one full workgroup + one partial),
    stated fp32 tolerances: loss 5e-4, gradient 2e-2, grids 5e-3; the fp64 lock-step kernel against the fp64 wide kernel
    (another mapping, the plain rk4_step, all 17 columns propagated): 1e-6, both stop on the same gradient test."""
    monkeypatch.setattr(CPDP.COCSys, "mapping_override", "lockstep")          # this test is about the lock-step kernels
    oc, env, d = models.quadrotor(n_grid=int(G["n_grid"]))
    emu(oc)
    consts = oc.consts_tensor(overrides=dict(goal_r0=G["goal_r"][0], goal_r1=G["goal_r"][1], goal_r2=G["goal_r"][2]))
    idx = [0, 20, 40, 60, 99]
    out = {}
    for dt in (torch.float64, torch.float32):
        oc.setDevice(dtype=dt)
        sol = oc.cocSolverBatch(np.tile(G["ini_state"], (len(idx), 1)), float(G["horizon"]), G["lookahead_theta"][idx],
                                consts=consts.to(dt))
        aux = oc.auxSysSolverBatch(sol, G["taus"], G["waypoints"], [0, 1, 2])
        out[dt] = (sol, aux)
    s64, a64 = out[torch.float64]
    s32, a32 = out[torch.float32]
    monkeypatch.setattr(CPDP.COCSys, "mapping_override", "wide")
    oc.setDevice(dtype=torch.float64)
    w64 = oc.cocSolverBatch(np.tile(G["ini_state"], (len(idx), 1)), float(G["horizon"]), G["lookahead_theta"][idx],
                            consts=consts.to(torch.float64))
    assert np.isin(s64["status"].numpy(), (1, 2)).all() and np.isin(w64["status"].numpy(), (1, 2)).all()
    assert rel(s64["state_grid"], w64["state_grid"]) < 1e-6 and rel(s64["control_grid"], w64["control_grid"]) < 1e-6
    assert rel(s64["costate_grid"], w64["costate_grid"]) < 1e-6
    assert (s32["status"] != 4).all() and (s32["status"] != 3).all(), s32["status"]      # neither failed nor at the limit
    assert rel(s32["state_grid"], s64["state_grid"]) < 5e-3
    assert rel(s32["costate_grid"], s64["costate_grid"]) < 5e-3
    for k, j in enumerate(idx):
        assert abs(a32["loss"][k].item() - G["loss_trace"][j]) < 5e-4 * G["loss_trace"][j]
        assert rel(a32["grad"][k], a64["grad"][k]) < 2e-2


def test_reference_shaped_single_trajectory_api(emu):
    """cocSolver / auxSysSolver with the reference's signatures and return types (CPDP.py:92,301)."""
    oc, env, d = models.pendulum(n_grid=10)
    emu(oc)
    oc.setDevice(dtype=torch.float64)
    oc.setSolverOptions(aux_substeps=16)
    th = [2, 1, 1]
    time_grid, opt_sol = oc.cocSolver(d["ini_state"], d["horizon"], th)
    auxsys_sol = oc.auxSysSolver(time_grid, opt_sol, th)
    assert time_grid.shape == (11,) and opt_sol(0.37).shape == (2 + 1 + 2,)
    assert auxsys_sol(0.37).shape == (2 * 3 + 1 * 3,)
    o = make_oracle("pendulum", 10)
    tg, osol = o.cocSolver(d["ini_state"], d["horizon"], th)
    oaux = o.auxSysSolver(tg, osol, th, riccati_method='Radau', ivp_kwargs=dict(rtol=1e-10, atol=1e-12))
    for t in (0.0, 0.05, 0.37, 1.0):
        assert rel(opt_sol(t), osol(t)) < 1e-6
        assert rel(auxsys_sol(t), oaux(t)) < 1e-3
    assert np.allclose(opt_sol(1.0)[2], opt_sol(0.9)[2])        # last control repeated (CPDP.py:191)


def test_time_varying_model_vs_oracle(emu):
    """COCSys_TimeVarying with v(t) = b1 + 2 b2 t (Examples/pendulum_timewarping.py:35-38)."""
    import sympy as sp
    from oracle import jinenv_sym as J
    from oracle.cpdp_oracle import COCSys_TimeVarying, getloss_corrections
    oc, env, d = models.pendulum_poly2(n_grid=10)
    emu(oc)
    assert oc.compile().time_varying
    oc.setDevice(dtype=torch.float64)
    oc.setSolverOptions(aux_substeps=8)
    th = [1.0, 0.8, 1.0, 1.2]
    sol = oc.cocSolverBatch([d["ini_state"]], d["horizon"], [th])
    aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"])
    e2 = J.SinglePendulum(); e2.initDyn(l=1, m=1, damping_ratio=0.1); e2.initCost(wu=.01)
    o = COCSys_TimeVarying()
    t, b1, b2 = sp.symbols('t beta1 beta2', real=True)
    o.setTimeVariable(t)
    o.setAuxvarVariable([b1, b2] + e2.cost_auxvar); o.setStateVariable(e2.X); o.setControlVariable(e2.U)
    v = b1 + 2 * b2 * t
    o.setDyn(v * e2.f); o.setPathCost(v * e2.path_cost); o.setFinalCost(e2.final_cost); o.setIntegrator(10)
    tg, osol, X, U, L = o.cocSolver(d["ini_state"], d["horizon"], th, return_grids=True)
    oaux = o.auxSysSolver(tg, osol, th, riccati_method='Radau', ivp_kwargs=dict(rtol=1e-10, atol=1e-12))
    l_o, g_o = getloss_corrections(o, d["taus"], d["waypoints"], osol, oaux, d["interface"])
    assert rel(sol["state_grid"][0], X) < 1e-6 and rel(sol["costate_grid"][0], L) < 1e-5
    assert abs(aux["loss"][0].item() - l_o) < 1e-7 * max(1, l_o)
    assert rel(aux["grad"][0], g_o) < 1e-4


@pytest.mark.parametrize("method", ["Vanilla", "Nesterov", "Adam", "Nadam", "AMSGrad"])
def test_optimizer_kernels_vs_reference_rules(emu, method):
    """lib/QuadAlgorithm.py:454-578 incl. the projection theta[0] >= 1e-8 (QuadAlgorithm.py:250)."""
    from oracle.cpdp_oracle import Optimizer
    oc, _, _ = models.pendulum()
    lib = emu(oc).compile()
    rng = np.random.default_rng(3)
    B, p = 5, 3
    th0 = rng.standard_normal((B, p))
    theta = torch.tensor(th0.copy())
    m, v, vh = torch.zeros_like(theta), torch.zeros_like(theta), torch.zeros_like(theta)
    lo = torch.tensor([1e-8, -np.inf, -np.inf], dtype=torch.float64)
    ref = [Optimizer(method, p, 0.05) for _ in range(B)]
    th_ref = th0.copy()
    for it in range(6):
        g = rng.standard_normal((B, p)) * (3.0 if it == 2 else 1.0)
        if method == "Nesterov":
            la = lib.lookahead(theta, m, 0.9)
            assert np.allclose(la.numpy(), np.stack([ref[b].lookahead(th_ref[b]) for b in range(B)]), atol=1e-14)
        lib.optimizer_step(method, theta, torch.tensor(g), it, 0.05, m=m, v=v, vhat=vh, proj_lo=lo)
        for b in range(B):
            th_ref[b] = ref[b].step(th_ref[b], g[b], it)
            th_ref[b][0] = max(th_ref[b][0], 1e-8)
        assert np.allclose(theta.numpy(), th_ref, rtol=1e-12, atol=1e-13)


def test_error_behaviour_matches_reference(emu):
    oc = CPDP.COCSys()
    x, u = SX.sym('x'), SX.sym('u')
    with pytest.raises(AssertionError, match="Define the state variable first!"):
        oc.model_spec()
    oc.setStateVariable(x, state_lb=[-1.0], state_ub=[1.0])                                      # state bounds: supported (round 3)
    assert oc.state_lb == [-1.0] and oc.state_ub == [1.0]
    with pytest.raises(ValueError):
        oc.setStateVariable(x, state_lb=[1.0], state_ub=[-1.0])
    oc.setControlVariable(u, control_lb=[-1.0], control_ub=[1.0])                                 # control bounds: supported
    assert oc.control_lb == [-1.0] and oc.control_ub == [1.0]
    with pytest.raises(ValueError):
        oc.setControlVariable(u, control_lb=[1.0], control_ub=[-1.0])
    oc.setStateVariable(x); oc.setControlVariable(u, control_lb=[-1e20], control_ub=[1e20])   # the reference's "no bound"
    with pytest.raises(AssertionError, match="Define the system dynamics first!"):
        oc.model_spec()
    oc.setDyn(u + SX.sym('stray'))
    oc.setPathCost(u * u); oc.setFinalCost(x * x)
    with pytest.raises(runtime.LfsdError):
        oc.model_spec()
    oc2, env, d = models.pendulum()
    emu(oc2)
    with pytest.raises(Exception, match="Wrong optimization method type!"):
        CPDP.SparseDemoLearner(oc2, d["ini_state"], 1.0, [0.5], [[1.0]], [0], d["theta0"], method="SGD")


def test_learning_loop_decreases_loss_like_the_example(emu):
    """Examples/pendulum_groundtruth.py: waypoints from a ground-truth theta; a few vanilla-GD iterations."""
    oc, env, d = models.pendulum(n_grid=10)
    emu(oc)
    oc.setDevice(dtype=torch.float64)
    o = make_oracle("pendulum", 10)
    tg, osol = o.cocSolver(d["ini_state"], 1.0, d["true_theta"])
    taus = tg[[1, 3, 6, 7, 9]]
    wps = [[osol(t)[0]] for t in taus]
    L = CPDP.SparseDemoLearner(oc, np.tile(d["ini_state"], (2, 1)), 1.0, taus, wps, [0],
                               np.array([d["theta0"], [1.2, 0.8, 1.1]]), method="Vanilla", learning_rate=1e-2)
    from oracle.cpdp_oracle import getloss_corrections
    oc.setSolverOptions(aux_substeps=8)
    losses = []
    for it in range(4):
        th = L.theta[0].numpy().copy()                 # parameters this iteration is evaluated at
        loss, grad = L.step()
        # replay trajectory 0 with the oracle pipeline of the example at the same parameters
        tg, s = o.cocSolver(d["ini_state"], 1.0, th)
        a = o.auxSysSolver(tg, s, th, riccati_method='Radau', ivp_kwargs=dict(rtol=1e-10, atol=1e-12))
        l_o, g_o = getloss_corrections(o, taus, wps, s, a, [0])
        assert abs(loss[0].item() - l_o) < 1e-6 * max(1, l_o) and rel(grad[0], g_o) < 5e-4
        th_new = th - 1e-2 * grad[0].numpy()           # current_parameter -= lr * diff_loss ; projection
        th_new[0] = max(th_new[0], 1e-8)
        assert np.allclose(L.theta[0].numpy(), th_new, rtol=0, atol=1e-13)
        losses.append(loss[0].item())
    assert losses[-1] < losses[0]


def test_quadalgorithm_driver_replays_reference_run_start(emu):
    """lib/QuadAlgorithm.py driver mirror with the settings of Examples/quad_example_human_input.py: the first
    Nesterov iterations must follow the reference's saved loss / parameter traces."""
    from lfsd_amd.QuadAlgorithm import QuadAlgorithm, QuadPara, DemoSparse
    from lfsd_amd.JinEnv import QuadStates
    from conftest import build_emu_library
    cfg = {"QUAD_AVERAGE_SPEED": 1.0, "LAB_SPACE_LIMIT": {"LIMIT_X": [-3.2, 3.2], "LIMIT_Y": [-1.6, 1.6], "LIMIT_Z": [0.0, 2.2]}}
    S = QuadAlgorithm(cfg, QuadPara([1.0, 1.0, 1.0], 1.0, 1.0, 0.02), int(G["n_grid"]), dtype=torch.float64)
    S.library = build_emu_library(models.quadrotor(n_grid=int(G["n_grid"]))[0])
    S.load_optimization_function({"learning_rate": 0.01, "iter_num": 3, "method": "Nesterov", "mu": 0.9,
                                  "true_loss_print_flag": False})
    ini = QuadStates(position=[-2.0, -1.0, 0.6])
    goal = QuadStates(position=[2.5, 1.0, 1.5])
    demo = DemoSparse(waypoints=G["waypoints"].tolist(), time_list=G["taus"].tolist(), time_horizon=1.0)
    res = S.run(ini, goal, demo, ObsList=[], print_flag=False, save_flag=False)
    assert res["loss_trace"].shape[0] == 3
    assert abs(res["loss_trace"][0, 0] - G["loss_trace"][0]) < 1e-7 * G["loss_trace"][0]
    # later iterations inherit the reference's own gradient error (its solve_ivp rtol is 1e-3)
    assert np.allclose(res["loss_trace"][:, 0], G["loss_trace"][:3], rtol=1e-3)
    assert np.allclose(res["parameter_trace"][1:4, 0], G["theta_trace"][1:4], rtol=2e-3, atol=2e-4)
    assert res["opt_state_traj"].shape == (101, 13) and res["opt_control_traj"].shape == (101, 4)
    with pytest.raises(Exception, match="Wrong optimization method type!"):
        S.load_optimization_function({"learning_rate": 0.01, "iter_num": 3, "method": "RMSprop"})


def test_rocket_lean_kernels_fp64_vs_fp32_same_iterates(emu, monkeypatch):
    """The lean lock-step kernels on the rocket's dimensions (13 live columns of 16, 3 constant states, 3 controls -- the
    quadrotor has 14 / 3 / 4; the rocket's default solve is Newton from the first iteration on the wide kernel and never
    gets here): the same six Gauss-Newton / Hamiltonian iterations from a cold start, no exact-Hessian phase, in fp64
    (live-column roll-out, parked RK4 step, one-pass structural sweep with LDS-fed products) and in fp32 (packed roll-out,
    structural sweep on the emulated matrix cores).  Same step control, same path: after six iterations -- far from
    converged, cost 1.1e4 and 3.8e2 -- the iterates agree to the fp32 class (measured: cost 1.7e-6, states 1e-5)."""
    monkeypatch.setattr(CPDP.COCSys, "mapping_override", "lockstep")
    monkeypatch.setenv("LFSD_F64_SEED", "0")      # the fp64 kernel's own six iterations, not the fp32-seeded solve
    oc, env, d = models.rocket(n_grid=15)
    emu(oc)
    oc.setSolverOptions(max_iter=6, exact_after=-1)
    th = np.array([d["true_theta"], d["theta0"]], dtype=float)
    x0 = np.tile(d["ini_state"], (2, 1))
    out = {}
    for dt in (torch.float64, torch.float32):
        oc.setDevice(dtype=dt)
        out[dt] = oc.cocSolverBatch(x0, d["horizon"], th)
        assert (out[dt]["iters"] == 6).all() and (out[dt]["status"] == 3).all(), (dt, out[dt]["iters"], out[dt]["status"])
    l64, l32 = out[torch.float64], out[torch.float32]
    assert rel(l32["cost"].double(), l64["cost"]) < 2e-5
    assert rel(l32["state_grid"].double(), l64["state_grid"]) < 2e-4 and rel(l32["control_grid"].double(), l64["control_grid"]) < 2e-3


def test_rocket_newton_mode_vs_oracle(emu):
    """Examples/rocket_groundtruth.py (6-DoF powered landing, T=3): needs the exact stage Hessians (second-order
    adjoint through the RK4 stages) from the first iteration.  The problem has several local minima and which one a
    cold start reaches depends on the globalisation (IPOPT's would differ from both ours and the oracle's), so parity is
    checked basin-independently: the oracle certifies the kernel's cold-start answer as a KKT point of the reference's
    NLP and differentiates the PMP along it; and the oracle's own KKT point is a fixed point of the kernel."""
    from conftest import oracle_check_solution
    oc, env, d = models.rocket(n_grid=15)
    emu(oc)
    oc.setDevice(dtype=torch.float64)
    oc.setSolverOptions(aux_substeps=8)
    th = d["true_theta"]
    sol = oc.cocSolverBatch([d["ini_state"]], d["horizon"], [th])
    assert sol["status"].tolist() == [1]
    o = make_oracle("rocket", 15)
    tg = np.linspace(0, d["horizon"], 16)
    taus = tg[[1, 3, 6, 10, 13]]                                    # rocket_groundtruth.py:78
    X, U, Lm = (sol[k][0].numpy() for k in ("state_grid", "control_grid", "costate_grid"))
    wps = [np.concatenate([X[k, 0:3], X[k, 6:10]]) + 0.05 for k in (1, 3, 6, 10, 13)]
    r = oracle_check_solution(o, d["ini_state"], d["horizon"], th, X, U, Lm, taus, wps, d["interface"])
    assert r["defect"] < 1e-9 and r["gmax"] < 1e-6 and r["lmax"] < 1e-6 * np.abs(Lm).max(), (r["defect"], r["gmax"], r["lmax"])
    aux = oc.auxSysSolverBatch(sol, taus, wps, d["interface"])
    assert abs(aux["loss"][0].item() - r["loss"]) < 1e-7 * max(1.0, r["loss"])
    assert rel(aux["grad"][0], r["grad"]) < 1e-3
    # the oracle's cold-start KKT point, handed over as the initial guess, is where the kernel stays
    r0 = o.cocSolver(d["ini_state"], d["horizon"], th, return_grids=True, exact_after=0, max_iter=400)
    assert o.last_info["converged"]
    sol2 = oc.cocSolverBatch([d["ini_state"]], d["horizon"], [th], u_init=torch.as_tensor(r0[3][None, :-1].copy()))
    assert sol2["status"].tolist() == [1] and abs(o.last_cost - sol2["cost"][0].item()) < 1e-8 * abs(o.last_cost)
    assert rel(sol2["state_grid"][0], r0[2]) < 1e-6 and rel(sol2["costate_grid"][0], r0[4]) < 1e-5


def test_edge_cases_ragged_empty_and_per_trajectory_inputs(emu, oc_mapping):
    """Ragged batch sizes vs the lane-group packing, no waypoints, waypoint times outside [0, T] (clamped to the end
    intervals like the interpolant's end points), per-trajectory horizons and per-trajectory constants."""
    oc, env, d = models.pendulum(n_grid=10)
    emu(oc)
    oc.setDevice(dtype=torch.float64)
    oc.setSolverOptions(aux_substeps=4)
    rng = np.random.default_rng(7)
    th_all = np.array([1.0, 0.5, 1.5]) + 0.2 * rng.standard_normal((17, 3))
    th_all[:, 0] = np.abs(th_all[:, 0]) + 0.3
    hz_all = rng.uniform(0.6, 1.4, 17)
    taus = [0.25, 0.5]
    wps = [[0.5], [1.0]]
    ref = None
    for B in (1, 7, 8, 9, 17):                        # 8 trajectories per wavefront for this model (G = 8)
        sol = oc.cocSolverBatch(np.tile(d["ini_state"], (B, 1)), hz_all[:B], th_all[:B])
        aux = oc.auxSysSolverBatch(sol, taus, wps, d["interface"])
        assert sol["state_grid"].shape == (B, 11, 2) and set(sol["status"].tolist()) <= {1, 2}
        if ref is None:
            ref = (sol["state_grid"][0].clone(), aux["loss"][0].clone(), aux["grad"][0].clone())
        # trajectory 0 must not depend on who shares its wavefront
        assert torch.equal(sol["state_grid"][0], ref[0]) and torch.equal(aux["loss"][0], ref[1])
        assert torch.equal(aux["grad"][0], ref[2])
    # no waypoints: loss 0, gradient 0, Riccati grid still produced
    sol = oc.cocSolverBatch(np.tile(d["ini_state"], (3, 1)), 1.0, th_all[:3])
    aux = oc.auxSysSolverBatch(sol)
    assert float(aux["loss"].abs().max()) == 0 and float(aux["grad"].abs().max()) == 0
    assert torch.isfinite(aux["Z_grid"]).all()
    # waypoint exactly at t = 0 contributes no gradient (X(0) = 0) and loss (x0 - wp)^2; t beyond T clamps to t = T
    a0 = oc.auxSysSolverBatch(sol, [0.0], [[0.3]], d["interface"])
    assert np.allclose(a0["loss"].numpy(), 0.09, atol=1e-12) and float(a0["grad"].abs().max()) < 1e-12
    aT = oc.auxSysSolverBatch(sol, [1.0], [[0.3]], d["interface"])
    with pytest.raises(ValueError):                   # scipy's interp1d (CPDP.py:386) refuses t > T; so does the host check
        oc.auxSysSolverBatch(sol, [1.0 + 1e-9], [[0.3]], d["interface"])
    aT2 = oc.auxSysSolverBatch(sol, [1.0 + 1e-9], [[0.3]], d["interface"], validate=False)      # the kernel itself clamps
    assert torch.allclose(aT["loss"], aT2["loss"], rtol=1e-6) and torch.allclose(aT["grad"], aT2["grad"], rtol=1e-5, atol=1e-9)
    # per-trajectory constants: a batch with different pendulum lengths equals the individual solves
    lens = np.array([0.8, 1.0, 1.3])
    cb = oc.consts_tensor(batch=3, overrides=dict(l=lens))
    solb = oc.cocSolverBatch(np.tile(d["ini_state"], (3, 1)), 1.0, th_all[:3], consts=cb)
    for b in range(3):
        c1 = oc.consts_tensor(overrides=dict(l=float(lens[b])))
        s1 = oc.cocSolverBatch([d["ini_state"]], 1.0, th_all[b:b + 1], consts=c1)
        assert torch.equal(s1["state_grid"][0], solb["state_grid"][b])
    assert not torch.equal(solb["state_grid"][0], solb["state_grid"][2])


def test_warm_start_and_mixed_precision(emu, monkeypatch):
    """u_init warm start reaches the same KKT point in fewer iterations; fp32 solve + fp64 auxiliary pass
    (BASELINE configs[4]) returns fp64-accurate sweeps on the fp32 trajectory.  Robot arm on the lock-step kernels (the other
    robot-arm tests run the wide mapping, lfsd_coc_solve's choice for small batches)."""
    monkeypatch.setattr(CPDP.COCSys, "mapping_override", "lockstep")
    oc, env, d = models.robotarm(n_grid=12)
    emu(oc)
    th = [[3., 0.5, 2, 1.5, 0.2]]
    oc.setDevice(dtype=torch.float64)
    cold = oc.cocSolverBatch([d["ini_state"]], 1.0, th)
    warm = oc.cocSolverBatch([d["ini_state"]], 1.0, th, u_init=cold["control_grid"][:, :12].contiguous())
    assert warm["iters"].item() <= 2 and cold["iters"].item() > 10
    assert torch.allclose(warm["state_grid"], cold["state_grid"], atol=1e-8)
    taus, wps = [0.3], [[-np.pi / 4, 2 * np.pi / 3]]
    a64 = oc.auxSysSolverBatch(cold, taus, wps, d["interface"])
    oc.setDevice(dtype=torch.float32, aux_dtype=torch.float64)
    s32 = oc.cocSolverBatch([d["ini_state"]], 1.0, th)
    amix = oc.auxSysSolverBatch(s32, taus, wps, d["interface"])
    assert s32["state_grid"].dtype == torch.float32 and amix["grad"].dtype == torch.float64
    oc.aux_dtype = None
    a32 = oc.auxSysSolverBatch(s32, taus, wps, d["interface"])
    e_mix = rel(amix["grad"][0], a64["grad"][0].numpy())
    e_32 = rel(a32["grad"][0], a64["grad"][0].numpy())
    assert e_mix < 2e-2 and e_mix <= e_32 * 1.5 + 1e-6


def test_learner_warm_start_same_iterates_fewer_solver_iterations(emu):
    oc, env, d = models.pendulum(n_grid=10)
    emu(oc)
    oc.setDevice(dtype=torch.float64)
    args = (oc, np.tile(d["ini_state"], (2, 1)), 1.0, [0.2, 0.5, 0.8], [[0.4], [1.5], [2.6]], [0],
            np.array([[1.0, 0.5, 1.5], [1.4, 0.8, 1.0]]))
    cold = CPDP.SparseDemoLearner(*args, method="Vanilla", learning_rate=1e-2)
    warm = CPDP.SparseDemoLearner(*args, method="Vanilla", learning_rate=1e-2, warm_start=True)
    it_c = it_w = 0
    for _ in range(3):
        lc, gc = cold.step(); it_c += int(cold._sol["iters"].sum())
        lw, gw = warm.step(); it_w += int(warm._sol["iters"].sum())
        assert torch.allclose(lc, lw, rtol=1e-8) and torch.allclose(gc, gw, rtol=1e-5, atol=1e-8)
    assert torch.allclose(cold.theta, warm.theta, rtol=1e-7) and it_w < it_c


def test_learner_skips_unconverged_trajectories(emu):
    """skip_unconverged=True (opt-in; the default is the reference's unguarded behaviour): a trajectory whose OC solve ran
    out of iterations has a meaningless sensitivity, so its row is frozen for that step -- parameters AND optimizer state,
    for every update rule (lfsd_optimizer_step's row_active mask) -- and counted."""
    from lfsd_amd import CPDP
    oc, env, d = models.ZOO["pendulum"](n_grid=10)
    emu(oc)
    oc.setDevice(dtype=torch.float64)
    th = np.array([[1.0, 0.5, 1.5], [2.0, 1.0, 1.0]])
    args = (np.tile(d["ini_state"], (2, 1)), d["horizon"], [0.3, 0.6], [[0.5], [1.0]], d["interface"], th)
    oc.setSolverOptions(max_iter=3)                       # every solve stops at the limit
    for method in ("Vanilla", "Nesterov", "Adam", "Nadam", "AMSGrad"):
        L = CPDP.SparseDemoLearner(oc, *args, method=method, learning_rate=0.1, skip_unconverged=True)
        L.m += 0.25; L.v += 0.5; L.vhat += 0.75        # state a masked update must leave alone (it would decay / move it)
        th0 = L.theta.clone()
        loss, grad = L.step()
        assert (L._sol["status"] == 3).all() and L.n_unconverged == 2
        assert (grad == 0).all() and torch.equal(L.theta, th0), method
        assert (L.m == 0.25).all() and (L.v == 0.5).all() and (L.vhat == 0.75).all(), method
    L = CPDP.SparseDemoLearner(oc, *args, learning_rate=0.1, skip_unconverged=True)
    th0 = L.theta.clone()
    # ... and the solve is continued at the next outer iteration rather than restarted: with 3 iterations per step a cold
    # start could never finish, the continued solves converge after a few steps and the update then goes ahead
    converged_once = False
    for _ in range(15):
        L.step()
        converged_once = converged_once or bool((L._sol["status"] == 1).any())
        if not torch.equal(L.theta[0], th0[0]):
            break
    assert converged_once and not torch.equal(L.theta[0], th0[0])
    L2 = CPDP.SparseDemoLearner(oc, *args, learning_rate=0.1)          # default: every gradient is applied
    assert L2.skip_unconverged is False
    L2.step()
    assert not torch.equal(L2.theta, th0)
    oc.setSolverOptions(max_iter=100)                     # converged solves are untouched by the guard
    L3 = CPDP.SparseDemoLearner(oc, *args, learning_rate=0.1, skip_unconverged=True)
    L4 = CPDP.SparseDemoLearner(oc, *args, learning_rate=0.1)
    L3.step(); L4.step()
    assert L3.n_unconverged == 0 and torch.equal(L3.theta, L4.theta)


def test_shared_mode_masks_a_bad_demonstration_by_default(emu):
    """mode='shared' sums every demonstration's gradient into ONE theta (and all-reduces it): a single failed solve or
    non-finite gradient would destroy the whole job, so the status / finiteness mask is ON by default there.  A batch
    with one poisoned demonstration must take exactly the step of the batch without it, and count it."""
    from lfsd_amd import CPDP
    oc, env, d = models.ZOO["pendulum"](n_grid=10)
    emu(oc)
    oc.setDevice(dtype=torch.float64)
    x0 = np.array([[0.0, 0.0], [0.1, 0.0], [np.nan, 0.0]])
    taus, wps, th0 = [0.3, 0.6], [[0.5], [1.0]], [1.5, 0.8, 1.2]
    L = CPDP.SparseDemoLearner(oc, x0, d["horizon"], taus, wps, d["interface"], th0, learning_rate=1e-2, mode="shared")
    assert L.skip_unconverged is True
    Lg = CPDP.SparseDemoLearner(oc, x0[:2], d["horizon"], taus, wps, d["interface"], th0, learning_rate=1e-2, mode="shared")
    for _ in range(2):
        loss, grad = L.step()
        loss_g, grad_g = Lg.step()
        assert L.n_unconverged == 1 and Lg.n_unconverged == 0
        assert torch.isfinite(L.theta).all() and torch.equal(L.theta, Lg.theta)
        assert torch.equal(loss, loss_g) and torch.equal(grad, grad_g)
    Lu = CPDP.SparseDemoLearner(oc, x0, d["horizon"], taus, wps, d["interface"], th0, learning_rate=1e-2, mode="shared",
                                skip_unconverged=False)            # the unguarded sum is what the default protects against
    Lu.step()
    assert not torch.isfinite(Lu.theta).all()


def test_waypoints_outside_the_horizon_raise_like_interp1d(emu):
    """The reference's opt_sol(t) is scipy's interp1d (CPDP.py:386): a tau outside [0, horizon] raises ValueError there;
    the kernels would extrapolate silently, so the host checks.  Same for an interface index that selects no state."""
    from lfsd_amd import CPDP, runtime
    oc, env, d = models.ZOO["pendulum"](n_grid=10)
    emu(oc)
    oc.setDevice(dtype=torch.float64)
    sol = oc.cocSolverBatch([d["ini_state"]], d["horizon"], [[1.0, 0.5, 1.5]])
    with pytest.raises(ValueError):
        oc.auxSysSolverBatch(sol, [0.3, 1.2], [[0.5], [1.0]], d["interface"])
    with pytest.raises(ValueError):
        oc.auxSysSolverBatch(sol, [-0.1], [[0.5]], d["interface"])
    with pytest.raises(runtime.LfsdError):
        oc.auxSysSolverBatch(sol, [0.3], [[0.5]], [2])
    with pytest.raises(ValueError):
        CPDP.SparseDemoLearner(oc, [d["ini_state"]], d["horizon"], [0.3, 1.5], [[0.5], [1.0]], d["interface"], [1.0, 0.5, 1.5])
    oc.auxSysSolverBatch(sol, [0.0, 1.0], [[0.5], [1.0]], d["interface"])           # both ends are inside


def test_control_bounds_vs_independent_bounded_solve(emu):
    """Finite control_lb / control_ub (CPDP.py:33-46, 150-153): control-limited sweep vs the oracle's L-BFGS-B solve of the
    same bounded NLP (tests/parity_cases.control_bounds)."""
    import parity_cases as pc

    def prepare(oc, dtype):
        emu(oc)
        oc.setDevice(dtype=dtype)
        return oc
    pc.control_bounds(prepare, torch.float64)


def test_configs0_pendulum_horizon50_single_seed(emu):
    """BASELINE configs[0] (SinglePendulum, horizon 50, 1 seed) on the emulated kernels; the -m gpu tier repeats it."""
    import parity_cases as pc

    def prepare(oc, dtype):
        emu(oc)
        oc.setDevice(dtype=dtype)
        return oc
    pc.configs0_pendulum(prepare)


def test_state_bounds_vs_independent_bounded_solve(emu):
    """Finite state_lb / state_ub (CPDP.py:20-31, 140-147): augmented-Lagrangian loop around the (emulated) kernels vs the
    oracle's SLSQP solve of the same bounded NLP (tests/parity_cases.state_bounds)."""
    import parity_cases as pc

    def prepare(oc, dtype):
        emu(oc)
        oc.setDevice(dtype=dtype)
        return oc
    pc.state_bounds(prepare, torch.float64)
    oc, env, d = models.pendulum(n_grid=10)
    oc.setStateVariable(env.X, [-1e20, -1e20], [1e20, 1e20])          # the reference's defaults: no bounded path
    assert oc._lib is None and oc.state_lb == [-1e20, -1e20]
    with pytest.raises(ValueError):
        oc.setStateVariable(env.X, [1.0, 0.0], [0.0, 1.0])


def test_aux_pass_skips_rows_by_oc_status(emu):
    """ABI 8: rows whose OC-solve status is in `skip_status` are not differentiated -- NaN loss / gradient / [P W] / dx/dtheta /
    du/dtheta rows (never what the caller's buffers happened to hold), zero stats -- and every other row is bit-identical to the unskipped call (Examples/robotarm_random.py:60-73
    applies every gradient; a learner that freezes unconverged rows does not pay for their sweeps)."""
    oc, env, d = models.pendulum(n_grid=10)
    emu(oc)
    oc.setDevice(dtype=torch.float64)
    th = np.array([[1.0, 0.5, 1.5], [2.0, 1.0, 1.0], [0.7, 1.3, 0.6], [1.5, 0.8, 1.2], [1.2, 0.9, 0.7]])
    taus, wps = [0.1, 0.3, 0.6], [[0.4], [1.2], [2.1]]
    sol = oc.cocSolverBatch(np.tile(d["ini_state"], (5, 1)), d["horizon"], th)
    ref = oc.auxSysSolverBatch(sol, taus, wps, d["interface"], skip_status=())
    ref_l, ref_g, ref_Z = ref["loss"].clone(), ref["grad"].clone(), ref["Z_grid"].clone()
    sol2 = dict(sol)
    st = sol["status"].clone()
    st[1], st[3] = 4, 3                            # pretend: row 1 failed, row 3 ran out of iterations
    sol2["status"] = st
    Z = torch.full_like(ref_Z, 7.0)
    keep0 = [0, 2, 3, 4]
    a4 = oc.auxSysSolverBatch(sol2, taus, wps, d["interface"], Z_grid=Z, want_grids=True)             # default: FAILED rows only
    assert torch.isnan(a4["loss"][1]) and torch.isnan(a4["grad"][1]).all() and bool(torch.isnan(a4["Z_grid"][1]).all())
    assert bool(torch.isnan(a4["auxX_grid"][1]).all()) and bool(torch.isnan(a4["auxU_grid"][1]).all())
    assert bool(torch.isfinite(a4["auxX_grid"][keep0]).all()) and bool(torch.isfinite(a4["auxU_grid"][keep0]).all())
    assert a4["stats"][1].tolist() == [0, 0, 0, 0]
    keep = [0, 2, 3, 4]
    assert torch.equal(a4["loss"][keep], ref_l[keep]) and torch.equal(a4["grad"][keep], ref_g[keep])
    assert torch.equal(a4["Z_grid"][keep], ref_Z[keep])
    a34 = oc.auxSysSolverBatch(sol2, taus, wps, d["interface"], skip_status=(3, 4))
    assert torch.isnan(a34["loss"][[1, 3]]).all() and torch.isnan(a34["grad"][[1, 3]]).all()
    assert torch.equal(a34["loss"][[0, 2, 4]], ref_l[[0, 2, 4]]) and torch.equal(a34["grad"][[0, 2, 4]], ref_g[[0, 2, 4]])
    # a mask without the status array is an argument error, not a silent no-op
    with pytest.raises(runtime.LfsdError):
        oc.compile().aux_solve(sol["horizon"], sol["auxvar"], sol["consts"], sol["state_grid"], sol["control_grid"],
                               sol["costate_grid"], None, None, None, skip_status=(4,))


def test_dudtheta_error_is_the_last_interval_times_one_gain(emu):
    from parity_cases import dudtheta_refinement

    def prepare(oc, dtype):
        emu(oc)
        oc.setDevice(dtype=dtype)
    dudtheta_refinement(prepare, with_rtol=False)      # (the rtol law is asserted in the GPU tier)


def test_general_interface_function_vs_oracle(emu):
    from parity_cases import general_interface

    def prepare(oc, dtype):
        emu(oc)
        oc.setDevice(dtype=dtype)
    general_interface(prepare, torch.float64)


def test_fp64_solve_seeded_by_fp32_reaches_the_same_kkt_point(emu):
    from parity_cases import seeded_f64_same_kkt_point

    def prepare(oc, dtype):
        emu(oc)
        oc.setDevice(dtype=dtype)
    seeded_f64_same_kkt_point(prepare, n_grid=8, batch=5)


def test_level0_of_the_mesh_continuation_same_kkt_point_and_partner_independent(emu, monkeypatch):
    """Level 0 of the lean kernels' mesh continuation (cpdp_oc.h TCL: the first three iterations of a cold start on n_grid / 2
    merged intervals, then prolongation + a roll-out on the one-step-per-interval level).  Quadrotor, fp32, lock-step mapping,
    n_grid 20 (level 0 needs an even grid of >= 20 intervals): the same KKT points as a build without it (-DLFSD_LEAN_TC=1) to
    the fp32 class; and the schedule is a fixed iteration count, so a trajectory's result does not depend on its partners in
    the wavefront -- the same problem placed in two different wavefronts (rows 1 and 6 of a ragged batch of 7, four per
    wavefront) comes back bit-identical, also next to a row that is warm-started (its workgroup then skips level 0)."""
    from conftest import build_emu_library
    monkeypatch.setattr(CPDP.COCSys, "mapping_override", "lockstep")
    rng = np.random.default_rng(5)
    out = {}
    for tag, flags in (("", ()), ("notc", ("-DLFSD_LEAN_TC=1",))):
        oc, env, d = models.quadrotor(n_grid=20)
        oc.use_library(build_emu_library(oc, flags, tag)); oc.compile(); oc.setDevice(dtype=torch.float32)
        if not tag:
            th = np.asarray(d["theta0"])[None, :] * (1 + 0.1 * rng.standard_normal((7, len(d["theta0"]))))
            th[:, 0] = np.abs(th[:, 0]) + 0.2
            x0 = np.tile(d["ini_state"], (7, 1)); x0[:, :3] += 0.3 * rng.standard_normal((7, 3))
            th[6], x0[6] = th[1], x0[1]
        out[tag] = oc.cocSolverBatch(x0, d["horizon"], th)
        assert ((out[tag]["status"] == 1) | (out[tag]["status"] == 2)).all(), out[tag]["status"]
    a, b = out[""], out["notc"]
    assert rel(a["cost"].double(), b["cost"].double()) < 2e-5
    assert rel(a["state_grid"].double(), b["state_grid"].double()) < 5e-4 and rel(a["control_grid"].double(), b["control_grid"].double()) < 5e-3
    assert (a["iters"] <= b["iters"] + 1).all()      # one counted iteration is the transfer roll-out
    for k in ("cost", "state_grid", "control_grid", "costate_grid"):
        assert torch.equal(a[k][1], a[k][6]), k
    # a warm-started row in the first wavefront: that workgroup runs without level 0, the duplicate in the second one with it --
    # both end in the same KKT point
    oc, env, d = models.quadrotor(n_grid=20)
    emu(oc); oc.setDevice(dtype=torch.float32)
    u0 = torch.zeros(7, 20, 4)
    u0[0] = a["control_grid"][0, :20]
    w = oc.cocSolverBatch(x0, d["horizon"], th, u_init=u0)
    assert ((w["status"] == 1) | (w["status"] == 2)).all()
    assert rel(w["cost"].double(), a["cost"].double()) < 2e-5


@pytest.mark.parametrize("kind,n_grid,dtype", [("robotarm", 40, torch.float64), ("robotarm", 40, torch.float32), ("cartpole", 40, torch.float64),
                                               ("pendulum", 40, torch.float32)])
def test_multiple_shooting_steps_end_at_the_single_shooting_kkt_point(kind, n_grid, dtype):
    """Wide kernel at >= 40 intervals: the product build (multiple-shooting steps of the lifted problem tried first, closed-loop
    roll-outs as fallback and to close the gaps, cpdp_oc.h OcWide::ms_*) against a build with -DLFSD_MS=0 (single shooting only) on
    the CPU emulator: both end at KKT points (status 1 / 2) of the NLP of CPDP.py:110-175 and at the SAME one, the answer does not
    depend on where a problem sits in the batch (duplicates bit-identical), and the costates returned are those of a gap-free
    trajectory (the shooting constraints hold to rounding when the product's controls are rolled out by the other build's nodes)."""
    from conftest import build_emu_library
    rng = np.random.default_rng(5)
    sols = []
    for flags, tag in (((), ""), (("-DLFSD_MS=0",), "noms")):
        oc, env, d = models.ZOO[kind](n_grid=n_grid)
        oc.use_library(build_emu_library(oc, extra_flags=flags, tag=tag))
        oc.compile()
        oc.setDevice(dtype=dtype)
        oc.setSolverOptions(mapping="wide")
        p = len(d["theta0"])
        th = np.array(d["theta0"])[None, :] * (1 + 0.05 * np.random.default_rng(5).standard_normal((3, p)))
        th[:, 0] = np.abs(th[:, 0]) + 0.1
        th = np.concatenate([th, th[:1]])                 # a duplicate of problem 0 at the end of the batch
        sols.append(oc.cocSolverBatch(np.tile(d["ini_state"], (4, 1)), d["horizon"], th))
    a, b = sols
    assert set(a["status"].tolist()) <= {1, 2} and set(b["status"].tolist()) <= {1, 2}, (a["status"], b["status"])
    assert torch.equal(a["state_grid"][0], a["state_grid"][3]) and torch.equal(a["costate_grid"][0], a["costate_grid"][3])
    f64 = dtype == torch.float64
    xt, jt = (2e-6, 1e-9) if f64 else (3e-3, 2e-5)
    assert float((a["state_grid"] - b["state_grid"]).abs().max() / b["state_grid"].abs().max()) < xt
    assert float((a["control_grid"] - b["control_grid"]).abs().max() / b["control_grid"].abs().max()) < 10 * xt
    assert float((a["costate_grid"] - b["costate_grid"]).abs().max() / b["costate_grid"].abs().max()) < 20 * xt
    assert float(((a["cost"] - b["cost"]).abs() / b["cost"].abs()).max()) < jt


@pytest.mark.parametrize("kind,n_grid", [("robotarm", 40), ("cartpole", 40), ("pendulum", 20)])
def test_small_model_backward_sweep_matches_the_generic_one(kind, n_grid):
    """Wide kernel, fp32, the small models: the backward sweep on LDS-staged operands with one column per lane and v_readlane
    exchanges (cpdp_oc.h OcWide::backward_small; stage Hessians of modes 0 / 1 precomputed for all intervals, costates of mode 1 by
    a recursion on the staged rows) against a build with -DLFSD_BW_SMALL=0 (OcSolver::backward: LDS hand-overs, the model called
    per stage) on the CPU emulator: the same recursion, so the same iterates up to the order of the fp32 sums -- same class of
    status, iteration counts within three (robot arm, measured: 18 18 25 against 18 20 25), same trajectory at the fp32 tolerance of the solve, with and without multiple-shooting gaps
    (n_grid 40 runs them, 20 does not)."""
    from conftest import build_emu_library
    sols = []
    for flags, tag in (((), ""), (("-DLFSD_BW_SMALL=0",), "nobws")):
        oc, env, d = models.ZOO[kind](n_grid=n_grid)
        oc.use_library(build_emu_library(oc, extra_flags=flags, tag=tag))
        oc.compile()
        oc.setDevice(dtype=torch.float32)
        oc.setSolverOptions(mapping="wide")
        p = len(d["theta0"])
        th = np.array(d["theta0"])[None, :] * (1 + 0.05 * np.random.default_rng(11).standard_normal((3, p)))
        th[:, 0] = np.abs(th[:, 0]) + 0.1
        sols.append(oc.cocSolverBatch(np.tile(d["ini_state"], (3, 1)), d["horizon"], th))
    a, b = sols
    assert set(a["status"].tolist()) <= {1, 2} and set(b["status"].tolist()) <= {1, 2}, (a["status"], b["status"])
    assert int((a["iters"] - b["iters"]).abs().max()) <= 3, (a["iters"], b["iters"])
    assert float((a["state_grid"] - b["state_grid"]).abs().max() / b["state_grid"].abs().max()) < 3e-3
    assert float((a["costate_grid"] - b["costate_grid"]).abs().max() / b["costate_grid"].abs().max()) < 6e-2
    assert float(((a["cost"] - b["cost"]).abs() / b["cost"].abs()).max()) < 2e-5


def test_wide_solve_with_several_wavefronts_and_two_launches_is_bit_identical(monkeypatch):
    """Wide kernel, fp32, a model whose interval-parallel phases take several rounds of a wavefront (rocket: Newton from the first
    iteration, mesh continuation with merged intervals at n_grid 40): the launch schemes of lfsd_capi.cpp's coc_solve_t -- one
    wavefront per trajectory; four per trajectory from the start (each wavefront its own LDS region, sequential phases redundant,
    parallel phases split, OcWide<..., W>); TWO launches with the solver state parked in the workspace in between, handed over by
    the counter of finished trajectories (the emulator runs the workgroups one after the other: with a capacity of 2 the last two of
    four trajectories are handed over before their first iteration) and at a fixed iteration (3: in the coarse phase; 30: later) --
    return the same bits: states, controls, costates, cost, iteration count, status."""
    from conftest import build_emu_library
    oc, env, d = models.rocket(n_grid=40)
    oc.use_library(build_emu_library(oc))
    oc.compile()
    oc.setDevice(dtype=torch.float32)
    p = len(d["theta0"])
    th = np.array(d["theta0"])[None, :] * (1 + 0.05 * np.random.default_rng(3).standard_normal((4, p)))
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    x0 = np.tile(d["ini_state"], (4, 1))
    keys = ("state_grid", "control_grid", "costate_grid", "cost", "iters", "status")
    ref = None
    for envs in (dict(LFSD_WIDE_WAVES="1"), dict(LFSD_WIDE_WAVES="4"), dict(LFSD_WIDE_CAPACITY="2"),
                 dict(LFSD_WIDE_CAPACITY="2", LFSD_WIDE_SUSPEND_IT="3"), dict(LFSD_WIDE_CAPACITY="2", LFSD_WIDE_SUSPEND_IT="30")):
        for k in ("LFSD_WIDE_WAVES", "LFSD_WIDE_CAPACITY", "LFSD_WIDE_SUSPEND_IT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in envs.items():
            monkeypatch.setenv(k, v)
        sol = oc.cocSolverBatch(x0, d["horizon"], th)
        assert set(sol["status"].tolist()) <= {1, 2}, (envs, sol["status"])
        if ref is None:
            ref = sol
            assert int(sol["iters"].max()) > 30          # (the hand-over at iteration 30 really happens)
        else:
            for key in keys:
                assert torch.equal(sol[key], ref[key]), (envs, key)


@pytest.mark.parametrize("kind,n_grid", [("robotarm", 40)])
def test_refused_rollout_from_an_iterate_with_gaps_keeps_the_gaps(kind, n_grid):
    """Wide kernel, multiple-shooting steps (advisor finding of round 5): the gaps of an iterate of the lifted problem used to live in
    the parking region of the 16 step-length roll-outs -- when ALL roll-outs from an iterate with gaps were refused, the loop went on
    with overwritten gaps (wrong directions; the emulator scan found no such refusal in 143 gapped roll-outs, so nothing saw it).
    Since round 6 they have words of their own.  A build with a test hook (-DLFSD_TEST_REFUSE_GAPPED=2) refuses the first two
    roll-outs that start from an iterate with gaps; the solve must go on from intact gaps: same KKT point as the product build, and no
    more than a few iterations more (each refusal costs one sweep with a larger shift or a change of the Hessian model)."""
    from conftest import build_emu_library
    sols = []
    for flags, tag in (((), ""), (("-DLFSD_TEST_REFUSE_GAPPED=2",), "refuse2")):
        oc, env, d = models.ZOO[kind](n_grid=n_grid)
        oc.use_library(build_emu_library(oc, extra_flags=flags, tag=tag))
        oc.compile()
        oc.setDevice(dtype=torch.float64)
        oc.setSolverOptions(mapping="wide")
        p = len(d["theta0"])
        th = np.array(d["theta0"])[None, :] * (1 + 0.05 * np.random.default_rng(7).standard_normal((4, p)))
        th[:, 0] = np.abs(th[:, 0]) + 0.1
        sols.append(oc.cocSolverBatch(np.tile(d["ini_state"], (4, 1)), d["horizon"], th))
    a, b = sols
    assert set(a["status"].tolist()) <= {1, 2} and set(b["status"].tolist()) <= {1, 2}, (a["status"], b["status"])
    assert int((b["iters"] - a["iters"]).max()) <= 8, (a["iters"], b["iters"])
    assert int((b["iters"] - a["iters"]).min()) >= 1, (a["iters"], b["iters"])      # (the hook fired on every seed: measured 19 20 19 18 -> 21 22 21 21)
    assert float((a["state_grid"] - b["state_grid"]).abs().max() / a["state_grid"].abs().max()) < 2e-6
    assert float(((a["cost"] - b["cost"]).abs() / a["cost"].abs()).max()) < 1e-9
