"""Pins the oracle to the reference's own saved run (tests/golden/uav_golden.npz, extracted by
tests/golden/make_uav_golden.py from data/uav_results_random_20210308113016.mat): the real
CasADi+IPOPT+solve_ivp pipeline produced these (theta, loss, dtheta) triples."""
import os

import numpy as np
import pytest

from conftest import make_oracle

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "uav_golden.npz"))


@pytest.fixture(scope="module")
def quad():
    oc = make_oracle("quadrotor", int(G["n_grid"]), goal=tuple(G["goal_r"]))
    oc.diffPMP()
    return oc


@pytest.mark.parametrize("j", [0, 60])
def test_oracle_reproduces_reference_loss_and_gradient(quad, j):
    """Reference-faithful mode (solve_ivp BDF / RK45 at scipy default tolerances, CPDP.py:335,368)."""
    from oracle.cpdp_oracle import getloss_corrections
    th = G["lookahead_theta"][j]
    tg, sol = quad.cocSolver(G["ini_state"], float(G["horizon"]), th)
    assert quad.last_info["converged"]
    aux = quad.auxSysSolver(tg, sol, th)
    loss, grad = getloss_corrections(quad, G["taus"], G["waypoints"], sol, aux, [0, 1, 2])
    assert abs(loss - G["loss_trace"][j]) <= 1e-8 * G["loss_trace"][j]
    assert np.abs(grad - G["grad_trace"][j]).max() <= 1e-5 * np.abs(G["grad_trace"][j]).max()


def test_oracle_reproduces_reference_final_trajectory(quad):
    """opt_state_traj / opt_control_traj saved by QuadAlgorithm.py:306-317 at the last parameter."""
    th = G["theta_trace"][-1]
    tg, sol, X, U, L = quad.cocSolver(G["ini_state"], float(G["horizon"]), th, return_grids=True)
    tr = sol(G["time_steps"])
    assert np.abs(tr[:, :13] - G["opt_state_traj"]).max() < 1e-8
    assert np.abs(tr[:, 13:17] - G["opt_control_traj"]).max() < 1e-8
    # solver-independent KKT certificate (complex-step roll-outs only)
    defect, gmax, lmax = quad.kkt_certificate(G["ini_state"], float(G["horizon"]), th, X, U, L)
    assert defect < 1e-10 and gmax < 1e-7 and lmax < 1e-9


def test_nesterov_rule_replays_reference_parameter_trace():
    """lib/QuadAlgorithm.py:469-495 with the golden gradients must replay parameter_trace."""
    from oracle.cpdp_oracle import Optimizer
    opt = Optimizer("Nesterov", 7, float(G["learning_rate"]), mu=float(G["mu"]))
    th = G["theta_trace"][0].copy()
    for j in range(100):
        assert np.allclose(opt.lookahead(th), G["lookahead_theta"][j], rtol=0, atol=1e-12)
        th = opt.step(th, G["grad_trace"][j], j)
        th[0] = max(th[0], 1e-8)
        assert np.allclose(th, G["theta_trace"][j + 1], rtol=0, atol=1e-10)


def test_kkt_certificate_other_robots():
    for kind, n_grid, x0, T, th in (("pendulum", 10, [0.0, 0.0], 1.0, [2, 1, 1]),
                                    ("robotarm", 12, [-np.pi / 2, 0, 0, 0], 1.0, [5., 1, 1, 1, 1]),
                                    ("cartpole", 10, [0, 0, 0, 0], 1.0, [2., 0.5, 0.5, 0.5, 0.5])):
        oc = make_oracle(kind, n_grid)
        tg, sol, X, U, L = oc.cocSolver(x0, T, th, return_grids=True)
        assert oc.last_info["converged"], kind
        defect, gmax, lmax = oc.kkt_certificate(x0, T, th, X, U, L)
        assert defect < 1e-10 and gmax < 1e-6 and lmax < 1e-7, (kind, defect, gmax, lmax)


# ---- second pipeline pin: the pendulum of Examples/pendulum_groundtruth.py (needs no CasADi: the example defines its own truth) -----
def _pendulum_example(n_grid):
    """Examples/pendulum_groundtruth.py:15-33, 59-68: the environment, the true parameter [2, 1, 1], waypoints = the angle of the
    true parameter's own solution at grid nodes 1, 3, 6, 7, 9 (of 10; scaled with a finer grid)."""
    from oracle.cpdp_oracle import getloss_corrections
    o = make_oracle("pendulum", n_grid)
    x0, T = [0.0, 0.0], 1.0
    tg, sol = o.cocSolver(x0, T, [2.0, 1.0, 1.0])
    taus = tg[[k * n_grid // 10 for k in (1, 3, 6, 7, 9)]]
    wps = np.array([[sol(t)[0]] for t in taus])

    def loss_grad(th, **aux_kw):
        tg2, s2 = o.cocSolver(x0, T, th)
        aux = o.auxSysSolver(tg2, s2, th, **aux_kw)
        return getloss_corrections(o, taus, wps, s2, aux, [0])
    return loss_grad


def test_oracle_pipeline_on_the_pendulum_example_gradient_vs_finite_differences():
    """The whole oracle pipeline (OC solve -> Riccati sweep -> forward sensitivity sweep -> waypoint loss, CPDP.py:92-381) on a
    SECOND model, against checks that share no code with the auxiliary system:
      * at the example's true parameter the waypoints are reproduced: loss = 0 and d(theta) = 0;
      * away from it the PDP gradient (no factor 2: lib/QuadAlgorithm.py:655-660) is half the derivative of the loss, which
        central differences through complete re-solves give directly.  The PDP differentiates the CONTINUOUS maximum principle
        along the interpolated grids (CPDP.py:320-323, 347), the loss is that of the discrete NLP: the two agree to O(dgrid^~1.7)
        -- measured 1.8e-2 / 6.0e-3 / 1.7e-3 at n_grid 10 / 20 / 40 -- so the error must shrink under refinement."""
    from conftest import TIGHT
    err = {}
    for n_grid in (10, 20):
        lg = _pendulum_example(n_grid)
        l0, g0 = lg([2.0, 1.0, 1.0], **TIGHT)
        assert l0 < 1e-20 and np.abs(g0).max() < 1e-10, (l0, g0)
        th = np.array([1.0, 0.5, 1.5])                      # the example's initial guess (pendulum_groundtruth.py:76)
        l, g = lg(th, **TIGHT)
        fd = np.zeros(3)
        for i in range(3):
            a, b = th.copy(), th.copy()
            a[i] += 1e-5; b[i] -= 1e-5
            fd[i] = (lg(a, **TIGHT)[0] - lg(b, **TIGHT)[0]) / 2e-5
        err[n_grid] = np.abs(g - fd / 2).max() / np.abs(fd / 2).max()
    assert err[10] < 3e-2 and err[20] < 1e-2 and err[20] < 0.5 * err[10], err


def test_oracle_replays_the_pendulum_example_loop():
    """Examples/pendulum_groundtruth.py:73-88 in the reference-faithful mode (BDF / RK45 at scipy's defaults): 100 plain gradient
    steps at lr 1e-2 from [1, 0.5, 1.5] with the projection beta >= 1e-8.  The loss falls monotonically from 10.35 to 6e-4 and the
    parameters move towards the truth [2, 1, 1] (measured end point [1.81, 1.27, 1.03]: the five waypoints identify beta best); the test
    replays the first 50 of them (loss 2.9e-2)."""
    lg = _pendulum_example(10)
    th = np.array([1.0, 0.5, 1.5])
    trace = []
    for j in range(50):          # (the example runs 100: 6e-4 at the end; the first 50 hold the CPU tier's time)
        l, g = lg(th)
        trace.append(l)
        th = th - 1e-2 * g
        th[0] = max(th[0], 1e-8)
    assert abs(trace[0] - 10.348190669827167) < 1e-6      # (the tight and the reference-mode integrators agree on the loss: it only needs the OC solve)
    assert np.all(np.diff(trace) < 0) and trace[-1] < 5e-2, (trace[0], trace[-1])      # (measured 2.9e-2 after 50, 6.3e-4 after 100)
    assert np.abs(th - np.array([2.0, 1.0, 1.0])).max() < 0.45, th


# ---- the robot models against the reference's own source text (tests/golden/jinenv_points.npz) --------------------------
def _jinenv_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_jinenv_golden", os.path.join(os.path.dirname(__file__), "golden",
                                                                                      "make_jinenv_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)             # (defines CASES / GOAL / evaluate; its main() is not run and nothing reads /root/reference)
    return mod


_JG = _jinenv_cases()
_JP = np.load(os.path.join(os.path.dirname(__file__), "golden", "jinenv_points.npz"))


@pytest.mark.parametrize("case", _JG.CASES, ids=[c[0] for c in _JG.CASES])
@pytest.mark.parametrize("which", ["oracle", "product"])
def test_jinenv_models_equal_the_reference_text(case, which):
    """f, path_cost and final_cost of every initDyn / initCost* variant of the five robots, at 16 random points each, equal
    what the reference's JinEnv.py (JinEnv/JinEnv.py:40-1575, evaluated through a sympy stand-in for casadi by
    tests/golden/make_jinenv_golden.py) gives -- for the oracle's restatement (oracle/jinenv_sym.py) and for the product's
    (learning-from-sparse-demonstrations_amd/JinEnv.py), which were written separately.  Same ordering of the learnable
    parameters included: the points carry the parameter vector in the reference's order."""
    name, cls, dyn_kw, cost_fn, cost_kw, wants_goal = case
    subs = None
    if which == "oracle":
        from oracle import jinenv_sym as J
        env = getattr(J, cls)()
        env.initDyn(**dyn_kw)
        if wants_goal:
            g = _JG.GOAL
            getattr(env, cost_fn)(g["position"], g["velocity"], g["attitude_quaternion"], g["angular_velocity"], **cost_kw)
        else:
            getattr(env, cost_fn)(**cost_kw)
    else:
        import lfsd_amd  # noqa: F401
        from lfsd_amd import JinEnv as J, symbolic
        env = getattr(J, cls)()
        env.initDyn(**dyn_kw)
        if wants_goal:
            getattr(env, cost_fn)(J.QuadStates(**_JG.GOAL), **cost_kw)
        else:
            getattr(env, cost_fn)(**cost_kw)
        free = set().union(*[e.free_symbols for e in list(env.f) + [env.path_cost, env.final_cost]])
        subs = {s: symbolic.const_default(s) for s in free if symbolic.is_const(s)}      # run-time constants at their values
    pts = {k: _JP["%s/%s" % (name, k)] for k in ("X", "U", "D", "E")}
    f, pc, fc = _JG.evaluate(env, pts, subs=subs)
    for got, key in ((f, "f"), (pc, "path_cost"), (fc, "final_cost")):
        ref = _JP["%s/%s" % (name, key)]
        err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
        assert err < 1e-12, "%s %s %s: %.3e" % (which, name, key, err)
