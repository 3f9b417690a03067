"""-m gpu tier: the real gfx950 libraries, called through the C ABI, vs the oracle and the golden fixture."""
import os

import numpy as np
import pytest
import torch

import lfsd_amd  # noqa: F401
from lfsd_amd import CPDP, models, runtime
from conftest import make_oracle, oracle_loss_grad

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "uav_golden.npz"))


def rel(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def gpu_model(kind, dtype, n_grid, substeps=8):
    oc, env, d = models.ZOO[kind](n_grid=n_grid)
    oc.setDevice("cuda:0", dtype)
    oc.setSolverOptions(aux_substeps=substeps)
    lib = oc.compile()
    assert not lib.is_emulator, "the GPU tier must run the HIP library"
    return oc, d


# stated tolerances: fp64 (loss 1e-7, grad 1e-4 at 8-16 substeps); fp32 = fp64->fp32 tolerance of the
# pipeline (loss 1e-4 well-conditioned / 2e-3 flat-cost problems, gradient 5e-3 / 2e-2)
@pytest.mark.parametrize("dtype,ltol,gtol", [(torch.float64, 1e-7, 1e-4), (torch.float32, 1e-4, 5e-3)])
def test_pendulum_vs_oracle(dtype, ltol, gtol):
    oc, d = gpu_model("pendulum", dtype, 10, substeps=16)
    thetas = np.array([[1.0, 0.5, 1.5], [2.0, 1.0, 1.0], [0.7, 1.3, 0.6]])
    taus, wps = [0.0, 0.3, 0.6, 0.7, 1.0], [[0.0], [1.2], [2.1], [2.4], [2.9]]
    sol = oc.cocSolverBatch(np.tile(d["ini_state"], (3, 1)), d["horizon"], thetas)
    aux = oc.auxSysSolverBatch(sol, taus, wps, d["interface"], want_grids=True)
    o = make_oracle("pendulum", 10)
    assert set(sol["status"].tolist()) <= {1, 2}
    for b in range(3):
        r = oracle_loss_grad(o, d["ini_state"], d["horizon"], thetas[b], taus, wps, d["interface"])
        assert rel(sol["state_grid"][b], r["X"]) < (1e-6 if dtype == torch.float64 else 5e-3)
        assert abs(aux["loss"][b].item() - r["loss"]) < ltol * max(1.0, r["loss"])
        assert rel(aux["grad"][b], r["grad"]) < gtol
        n, p = 2, 3
        assert rel(aux["auxX_grid"][b].permute(0, 2, 1).reshape(-1, n * p), r["vX"]) < (1e-3 if dtype == torch.float64 else 1e-2)


@pytest.mark.parametrize("kind,n_grid,thetas,taus,wps", [
    ("robotarm", 12, [[5., 1, 1, 1, 1], [3., 0.5, 2, 1.5, 0.2]], [0.3], [[-np.pi / 4, 2 * np.pi / 3]]),
    ("cartpole", 10, [[1.0, 0.5, 0.5, 0.5, 0.5], [0.8, 2, 0.3, 1, 1]], [0.25, 0.8], [[0.1, 0.5], [0.0, 2.5]])])
def test_robotarm_cartpole_vs_oracle(kind, n_grid, thetas, taus, wps):
    o = make_oracle(kind, n_grid)
    refs = None
    for dtype, ltol, gtol in ((torch.float64, 1e-6, 1e-4), (torch.float32, 2e-3, 2e-2)):
        oc, d = gpu_model(kind, dtype, n_grid, substeps=16)
        refs = refs or [oracle_loss_grad(o, d["ini_state"], d["horizon"], th, taus, wps, d["interface"]) for th in thetas]
        B = len(thetas)
        sol = oc.cocSolverBatch(np.tile(d["ini_state"], (B, 1)), d["horizon"], np.array(thetas))
        aux = oc.auxSysSolverBatch(sol, taus, wps, d["interface"])
        for b in range(B):
            assert abs(aux["loss"][b].item() - refs[b]["loss"]) < ltol * max(1.0, refs[b]["loss"]), (kind, dtype)
            assert rel(aux["grad"][b], refs[b]["grad"]) < gtol, (kind, dtype)


@pytest.mark.parametrize("dtype,ltol", [(torch.float64, 1e-6), (torch.float32, 5e-4)])
def test_quadrotor_vs_reference_golden_run(dtype, ltol):
    """(theta, loss, dtheta) triples produced by the reference's own CasADi+IPOPT+solve_ivp run.
    Gradient tolerance 1e-2: the reference's solve_ivp (rtol 1e-3) is itself 5e-4..5e-3 off the exact ODE."""
    oc, d = gpu_model("quadrotor", dtype, int(G["n_grid"]), substeps=4)
    consts = oc.consts_tensor(overrides=dict(goal_r0=G["goal_r"][0], goal_r1=G["goal_r"][1], goal_r2=G["goal_r"][2]))
    idx = list(range(0, 100, 3)) + [99]
    sol = oc.cocSolverBatch(np.tile(G["ini_state"], (len(idx), 1)), float(G["horizon"]), G["lookahead_theta"][idx],
                            consts=consts)
    aux = oc.auxSysSolverBatch(sol, G["taus"], G["waypoints"], [0, 1, 2])
    assert set(sol["status"].tolist()) <= {1, 2}
    loss = aux["loss"].double().cpu().numpy()
    assert np.all(np.abs(loss - G["loss_trace"][idx]) < ltol * G["loss_trace"][idx])
    gtol = 1e-2 if dtype == torch.float64 else 2e-2     # fp32: near-stationary points (|dtheta| ~ 0.05) amplify rounding
    for k, j in enumerate(idx):
        assert rel(aux["grad"][k], G["grad_trace"][j]) < gtol, j
    if dtype == torch.float64:
        oc.const_values = consts.tolist()
        tg, opt = oc.cocSolver(G["ini_state"], float(G["horizon"]), G["theta_trace"][-1])
        tr = opt(G["time_steps"])
        assert np.abs(tr[:, :13] - G["opt_state_traj"]).max() < 1e-6
        assert np.abs(tr[:, 13:17] - G["opt_control_traj"]).max() < 1e-6


def test_quadrotor_loss_trace_parity_with_nesterov():
    """Loss-vs-iteration parity with the CasADi CPU path: replay the reference's 100 Nesterov iterations
    (lib/QuadAlgorithm.py:469-495, lr 0.01, mu 0.9) on the GPU in fp64 and compare the whole loss trace."""
    oc, d = gpu_model("quadrotor", torch.float64, int(G["n_grid"]), substeps=8)
    consts = oc.consts_tensor(overrides=dict(goal_r0=G["goal_r"][0], goal_r1=G["goal_r"][1], goal_r2=G["goal_r"][2]))
    L = CPDP.SparseDemoLearner(oc, G["ini_state"], float(G["horizon"]), G["taus"], G["waypoints"], [0, 1, 2],
                               G["theta_trace"][0], method="Nesterov", learning_rate=float(G["learning_rate"]),
                               mu=float(G["mu"]), consts=consts)
    losses = []
    for j in range(40):
        loss, grad = L.step()
        losses.append(loss[0].item())
    losses = np.array(losses)
    # the trajectories of two optimisers fed slightly different gradients drift apart slowly: 2% over 40 iterations
    assert np.all(np.abs(losses - G["loss_trace"][:40]) < 2e-2 * G["loss_trace"][:40]), losses
    assert rel(L.theta[0], G["theta_trace"][40]) < 2e-2


def test_full_size_properties_quadrotor_batch4096():
    """BASELINE size (horizon 50, batch 4096): size-independent properties instead of the (slow) oracle."""
    oc, d = gpu_model("quadrotor", torch.float32, 50, substeps=4)
    B = 4096
    rng = np.random.default_rng(0)
    th = np.array(d["theta0"])[None, :] + 0.05 * rng.standard_normal((B, 7))
    th[:, 0] = np.abs(th[:, 0]) + 0.5
    th[B // 2:] = th[:B // 2]                       # duplicated seeds must give identical results
    x0 = np.tile(d["ini_state"], (B, 1))
    sol = oc.cocSolverBatch(x0, 1.0, th)
    aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"])
    st = sol["status"].cpu().numpy()
    assert np.all((st == 1) | (st == 2)), np.bincount(st)
    assert torch.isfinite(aux["loss"]).all() and torch.isfinite(aux["grad"]).all()
    assert torch.equal(aux["loss"][:B // 2], aux["loss"][B // 2:]) and torch.equal(aux["grad"][:B // 2], aux["grad"][B // 2:])
    X, U, Lm = sol["state_grid"], sol["control_grid"], sol["costate_grid"]
    assert torch.allclose(X[:, 0], torch.as_tensor(x0, dtype=torch.float32, device=X.device))       # x(0) = ini_state
    assert torch.equal(U[:, -1], U[:, -2])                                                         # CPDP.py:191
    # P(t_k) symmetric, terminal W = d2h/dxde = 0 for this cost; X(0) = 0 implies grad finite
    Z = aux["Z_grid"]
    P = Z[:, :, :13, :]
    assert (P - P.transpose(2, 3)).abs().max() < 1e-3 * P.abs().max()
    assert Z[:, -1, 13:, :].abs().max() == 0
    # a random subset against the fp64 path (fp64 is itself pinned to the oracle/golden above)
    sub = rng.choice(B // 2, 64, replace=False)
    oc64, _ = gpu_model("quadrotor", torch.float64, 50, substeps=4)
    sol64 = oc64.cocSolverBatch(x0[sub], 1.0, th[sub])
    aux64 = oc64.auxSysSolverBatch(sol64, d["taus"], d["waypoints"], d["interface"])
    l32, l64 = aux["loss"][sub].double().cpu().numpy(), aux64["loss"].cpu().numpy()
    assert np.all(np.abs(l32 - l64) < 1e-3 * np.maximum(1.0, l64))
    g32, g64 = aux["grad"][sub].double().cpu().numpy(), aux64["grad"].cpu().numpy()
    assert np.all(np.abs(g32 - g64).max(axis=1) < 2e-2 * np.abs(g64).max(axis=1))


def test_shared_theta_gradient_is_sum_over_demonstrations():
    oc, d = gpu_model("pendulum", torch.float64, 10, substeps=8)
    B = 37                                            # ragged: not a multiple of the 8 groups per wavefront
    rng = np.random.default_rng(1)
    x0 = np.tile(d["ini_state"], (B, 1)) + 0.1 * rng.standard_normal((B, 2))
    taus = np.tile([0.2, 0.5, 0.8], (B, 1))
    wps = rng.uniform(0.2, 2.5, (B, 3, 1))
    th0 = np.array([1.5, 0.8, 1.2])
    Ls = CPDP.SparseDemoLearner(oc, x0, 1.0, taus, wps, [0], th0, method="Vanilla", learning_rate=1e-3, mode="shared")
    Li = CPDP.SparseDemoLearner(oc, x0, 1.0, taus, wps, [0], th0, method="Vanilla", learning_rate=1e-3)
    ls, gs = Ls.step()
    li, gi = Li.step()
    assert torch.allclose(ls, li.sum().reshape(1), rtol=1e-12) and torch.allclose(gs, gi.sum(0, keepdim=True), rtol=1e-12)
    assert torch.allclose(Ls.theta, torch.as_tensor(th0, device=gs.device)[None] - 1e-3 * gs, rtol=1e-12)


@pytest.mark.parametrize("method", ["Vanilla", "Nesterov", "Adam", "Nadam", "AMSGrad"])
def test_optimizer_kernels(method):
    from oracle.cpdp_oracle import Optimizer
    oc, d = gpu_model("pendulum", torch.float64, 10)
    lib = oc.compile()
    rng = np.random.default_rng(3)
    B, p = 5, 3
    th_ref = rng.standard_normal((B, p))
    theta = torch.tensor(th_ref.copy(), device="cuda:0")
    m, v, vh = torch.zeros_like(theta), torch.zeros_like(theta), torch.zeros_like(theta)
    lo = torch.tensor([1e-8, -np.inf, -np.inf], dtype=torch.float64, device="cuda:0")
    ref = [Optimizer(method, p, 0.05) for _ in range(B)]
    for it in range(6):
        g = rng.standard_normal((B, p))
        lib.optimizer_step(method, theta, torch.tensor(g, device="cuda:0"), it, 0.05, m=m, v=v, vhat=vh, proj_lo=lo)
        for b in range(B):
            th_ref[b] = ref[b].step(th_ref[b], g[b], it)
            th_ref[b][0] = max(th_ref[b][0], 1e-8)
        assert np.allclose(theta.cpu().numpy(), th_ref, rtol=1e-12, atol=1e-13)


def test_rocket_newton_mode_vs_oracle():
    """BASELINE configs[4] robot (Examples/rocket_groundtruth.py): exact stage Hessians from the first iteration."""
    o = make_oracle("rocket", 15)
    oc, d = gpu_model("rocket", torch.float64, 15, substeps=8)
    th = d["true_theta"]
    taus = np.linspace(0, d["horizon"], 16)[[1, 3, 6, 10, 13]]
    r0 = o.cocSolver(d["ini_state"], d["horizon"], th, return_grids=True, exact_after=0, max_iter=400)
    wps = [np.concatenate([r0[1](t)[0:3], r0[1](t)[6:10]]) + 0.05 for t in taus]
    r = oracle_loss_grad(o, d["ini_state"], d["horizon"], th, taus, wps, d["interface"], exact_after=0, max_iter=400)
    sol = oc.cocSolverBatch([d["ini_state"]] * 3, d["horizon"], [th] * 3)
    aux = oc.auxSysSolverBatch(sol, taus, wps, d["interface"])
    assert sol["status"].tolist() == [1, 1, 1]
    assert abs(sol["cost"][0].item() - o.last_cost) < 1e-8 * abs(o.last_cost)
    assert rel(sol["state_grid"][2], r["X"]) < 1e-6
    assert abs(aux["loss"][1].item() - r["loss"]) < 1e-7 * max(1.0, r["loss"]) and rel(aux["grad"][1], r["grad"]) < 1e-3
    oc32, _ = gpu_model("rocket", torch.float32, 15, substeps=8)
    sol32 = oc32.cocSolverBatch([d["ini_state"]], d["horizon"], [th])
    aux32 = oc32.auxSysSolverBatch(sol32, taus, wps, d["interface"])
    assert sol32["status"].item() in (1, 2)
    assert abs(aux32["loss"][0].item() - r["loss"]) < 2e-3 * max(1.0, r["loss"]) and rel(aux32["grad"][0], r["grad"]) < 2e-2


def test_time_varying_model_vs_oracle():
    """COCSys_TimeVarying with v(t) = b1 + 2 b2 t (Examples/pendulum_timewarping.py:35-38) on the GPU."""
    import sympy as sp
    from oracle import jinenv_sym as J
    from oracle.cpdp_oracle import COCSys_TimeVarying, getloss_corrections
    oc, d = gpu_model("pendulum_poly2", torch.float64, 10, substeps=8)
    assert oc.compile().time_varying
    th = [1.0, 0.8, 1.0, 1.2]
    sol = oc.cocSolverBatch([d["ini_state"]] * 2, d["horizon"], [th] * 2)
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
    assert rel(sol["state_grid"][1], X) < 1e-6 and rel(sol["costate_grid"][1], L) < 1e-5
    assert abs(aux["loss"][1].item() - l_o) < 1e-7 * max(1, l_o) and rel(aux["grad"][1], g_o) < 1e-4


def test_quadalgorithm_driver_fp32_follows_reference_run():
    """lib/QuadAlgorithm.py mirror on the GPU in fp32 (the precision of the benchmark): the first Nesterov iterations
    of Examples/quad_example_human_input.py must follow the reference's saved loss trace."""
    from lfsd_amd.QuadAlgorithm import QuadAlgorithm, QuadPara, DemoSparse
    from lfsd_amd.JinEnv import QuadStates
    cfg = {"QUAD_AVERAGE_SPEED": 1.0, "LAB_SPACE_LIMIT": {"LIMIT_X": [-3.2, 3.2], "LIMIT_Y": [-1.6, 1.6], "LIMIT_Z": [0.0, 2.2]}}
    S = QuadAlgorithm(cfg, QuadPara([1.0, 1.0, 1.0], 1.0, 1.0, 0.02), int(G["n_grid"]), device="cuda:0", dtype=torch.float32)
    S.load_optimization_function({"learning_rate": 0.01, "iter_num": 10, "method": "Nesterov", "mu": 0.9,
                                  "true_loss_print_flag": False})
    demo = DemoSparse(waypoints=G["waypoints"].tolist(), time_list=G["taus"].tolist(), time_horizon=1.0)
    res = S.run(QuadStates(position=[-2.0, -1.0, 0.6]), QuadStates(position=[2.5, 1.0, 1.5]), demo, ObsList=[])
    assert res["loss_trace"].shape[0] == 10
    assert np.allclose(res["loss_trace"][:, 0], G["loss_trace"][:10], rtol=5e-3)
    assert np.allclose(res["parameter_trace"][10, 0], G["theta_trace"][10], rtol=1e-2, atol=2e-3)


def test_robotarm_batch1024_random_seeds_configs1():
    """BASELINE configs[1]: robot arm, n_grid 50, 1024 random seeds, fp32.  At the perturbed initial guesses every cold-started
    solve has to finish (converged, or stalled at fp32 working precision) and the fp32 loss / gradient have to agree with the
    fp64 kernels on every seed (stated fp32 tolerances for the flat-cost arm: loss 2e-3, gradient 2e-2; measured 4e-4).
    Then five plain gradient steps at the example's learning rate (Examples/robotarm_random.py): no trajectory may be lost
    to a non-finite parameter -- an unconverged solve is skipped and continued, not applied."""
    from lfsd_amd import CPDP
    B = 1024
    rng = np.random.default_rng(0)
    th = np.array([5.0, 1, 1, 1, 1])[None, :] * (1 + 0.05 * rng.standard_normal((B, 5)))     # around the example's initial guess
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    res = {}
    for dt in (torch.float32, torch.float64):
        oc, d = gpu_model("robotarm", dt, 50, substeps=4)
        sol = oc.cocSolverBatch(np.tile(d["ini_state"], (B, 1)), d["horizon"], th)
        aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"])
        st = sol["status"].cpu().numpy()
        assert np.isin(st, (1, 2)).all(), (dt, np.bincount(st, minlength=5))
        res[dt] = (aux["loss"].double().cpu().numpy(), aux["grad"].double().cpu().numpy())
    l32, g32 = res[torch.float32]
    l64, g64 = res[torch.float64]
    assert (np.abs(l32 - l64) < 2e-3 * np.maximum(1.0, l64)).all()
    assert (np.abs(g32 - g64).max(1) < 2e-2 * np.abs(g64).max(1)).all()
    oc, d = gpu_model("robotarm", torch.float32, 50, substeps=4)
    L = CPDP.SparseDemoLearner(oc, np.tile(d["ini_state"], (B, 1)), d["horizon"], d["taus"], d["waypoints"], d["interface"], th,
                               method="Vanilla", learning_rate=d["lr"])
    for _ in range(5):
        loss, grad = L.step()
    assert torch.isfinite(L.theta).all() and torch.isfinite(grad).all()
    st = L._sol["status"].cpu().numpy()
    assert (st == 4).mean() < 0.01 and np.isin(st, (1, 2)).mean() > 0.85, np.bincount(st, minlength=5)
