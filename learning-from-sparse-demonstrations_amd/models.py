"""The optimal-control systems the reference's examples build, ready to compile.

Each factory repeats the model-construction lines of one example script
(Examples/*.py, lib/QuadAlgorithm.py:74-98) against the HIP-backed classes and
returns ``(oc, env, defaults)``; ``defaults`` carries the example's horizon,
initial state, initial parameter guess, learning rate and interface indices.
"""
import math

import numpy as np

from . import CPDP, JinEnv
from .symbolic import SX, vertcat


def _warp(env, oc, name, n_grid):
    """Time-warped system with scalar beta: dyn = beta*f, path cost = beta*c (e.g. robotarm_random.py:20-28)."""
    beta = SX.sym('beta')
    oc.setAuxvarVariable(vertcat(beta, env.cost_auxvar))
    oc.setStateVariable(env.X)
    oc.setControlVariable(env.U)
    oc.setDyn(beta * env.f)
    oc.setPathCost(beta * env.path_cost)
    oc.setFinalCost(env.final_cost)
    oc.setIntegrator(n_grid)
    oc.sys_name = name
    return oc


def pendulum(n_grid=10):
    """Examples/pendulum_random.py / pendulum_groundtruth.py."""
    env = JinEnv.SinglePendulum()
    env.initDyn(l=1, m=1, damping_ratio=0.1)
    env.initCost(wu=.01)
    oc = _warp(env, CPDP.COCSys(), "pendulum_tw", n_grid)
    return oc, env, dict(ini_state=[0.0, 0.0], horizon=1.0, theta0=[1, 0.5, 1.5], lr=1e-2, interface=[0],
                         true_theta=[2, 1, 1])


def pendulum_poly2(n_grid=10):
    """Examples/pendulum_timewarping.py with the second-order polynomial time-warp v = b1 + 2 b2 t."""
    env = JinEnv.SinglePendulum()
    env.initDyn(l=1, m=1, damping_ratio=0.1)
    env.initCost(wu=.01)
    oc = CPDP.COCSys_TimeVarying()
    t = SX.sym('t')
    oc.setTimeVariable(t)
    b1, b2 = SX.sym('beta1'), SX.sym('beta2')
    oc.setAuxvarVariable(vertcat(b1, b2, env.cost_auxvar))
    oc.setStateVariable(env.X)
    oc.setControlVariable(env.U)
    v = b1 + 2 * b2 * t
    oc.setDyn(v * env.f)
    oc.setPathCost(v * env.path_cost)
    oc.setFinalCost(env.final_cost)
    oc.setIntegrator(n_grid)
    oc.sys_name = "pendulum_poly2"
    return oc, env, dict(ini_state=[0.0, 0.0], horizon=0.2, theta0=[1., 1., 1, 1], lr=5e-3, interface=[0],
                         taus=(np.array([0.1, 0.3, 0.6, 0.7, 0.9]) * 0.2).tolist(),
                         waypoints=[[0.5], [1.8], [2.0], [2.9], [3.1]])


def robotarm(n_grid=30):
    """Examples/robotarm_random.py."""
    env = JinEnv.RobotArm()
    env.initDyn(l1=1, m1=1, l2=1, m2=1, g=0)
    env.initCost_Polynomial(wu=.5)
    oc = _warp(env, CPDP.COCSys(), "robotarm_poly_tw", n_grid)
    # most of this problem's iterations run on exact stage Hessians: the wide mapping (one trajectory per wavefront) is
    # 2.7-6x faster than the lock-step kernels at every batch size measured (profiles/r02_d_wide_vs_lockstep.txt)
    # ... and 120 iterations are more than twice what any well-posed seed of the example needs (51 over 12 learner steps of
    # 1024 seeds, tests/test_gpu_parity.py); the seeds a fixed learning rate has thrown out of the well-posed region have
    # no minimiser and would otherwise hold every launch for 300 (profiles/r03_b_arm_j0.txt)
    oc.setSolverOptions(mapping="wide", max_iter=120)
    return oc, env, dict(ini_state=[-math.pi / 2, 0, 0, 0], horizon=1.0, theta0=[5., 1, 1, 1, 1], lr=1e-1,
                         interface=[0, 1], taus=[0.3], waypoints=[[-math.pi / 4, 2 * math.pi / 3]])


def cartpole(n_grid=20):
    env = JinEnv.CartPole()
    env.initDyn(mc=0.5, mp=0.5, l=1)
    env.initCost(wu=0.1)
    oc = _warp(env, CPDP.COCSys(), "cartpole_tw", n_grid)
    return oc, env, dict(ini_state=[0, 0, 0, 0], horizon=1.0, theta0=[2., 0.5, 0.5, 0.5, 0.5], lr=1e-2,
                         interface=[0, 1])


def quadrotor(n_grid=25, goal=(3, 3, 1.5)):
    """lib/QuadAlgorithm.py:74-98 with the parameters of Examples/quad_example.py."""
    env = JinEnv.Quadrotor()
    env.initDyn(Jx=1.0, Jy=1.0, Jz=1.0, mass=1.0, l=1.0, c=0.02)
    env.initCost_Polynomial(JinEnv.QuadStates(position=list(goal)), w_thrust=0.1)
    oc = _warp(env, CPDP.COCSys(), "quadrotor_poly_tw", n_grid)
    wps = [[0.5, 0.5, 0.6], [1.0, 1.0, 0.8], [1.5, 1.5, 1.0], [2.0, 2.0, 1.2], [2.5, 2.5, 1.5]]
    return oc, env, dict(ini_state=[0, 0, 0.6, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], horizon=1.0,
                         theta0=[1, 0.1, 0.1, 0.1, 0.1, 0.1, -1], lr=1e-2, interface=[0, 1, 2],
                         taus=(np.array([1.0, 2.0, 3.0, 4.0, 5.0]) / 6).tolist(), waypoints=wps)


def rocket(n_grid=15):
    """Examples/rocket_groundtruth.py."""
    env = JinEnv.Rocket()
    env.initDyn(Jx=1, Jy=1, Jz=1, mass=1, l=1)
    env.initCost2(wthrust=0.1)
    oc = _warp(env, CPDP.COCSys(), "rocket_cost2_tw", n_grid)
    ini = [10, -8, 3.] + [0.1, 0.0, -0.0] + JinEnv.toQuaternion(1, [0, -1, 1]) + [0, -0.0, 0.0]
    # far initial state + 3 s horizon: Gauss-Newton stalls in a poor basin, Newton (exact stage Hessians) from the
    # first iteration reaches the optimum
    oc.setSolverOptions(exact_after=0, max_iter=600)
    return oc, env, dict(ini_state=ini, horizon=3.0, theta0=[1.0] + [0.5] * 11, lr=1e-3,
                         interface=[0, 1, 2, 6, 7, 8, 9], true_theta=[2] + [1] * 11)


ZOO = dict(pendulum=pendulum, pendulum_poly2=pendulum_poly2, robotarm=robotarm, cartpole=cartpole,
           quadrotor=quadrotor, rocket=rocket)


def build_all(verbose=False, force=False):
    """Compile every standard model for gfx950 (in-tree .so under csrc/build/).  Generated headers / libraries that an
    EARLIER build_all produced for the zoo and that are no longer reachable (code-generator version bump) are removed;
    files of models a user compiled through COCSys.compile() are never touched (csrc/gen/ZOO_MANIFEST.json records
    which hashes belong to the zoo)."""
    import json
    import os
    from . import runtime
    from concurrent.futures import ThreadPoolExecutor
    out, keep, specs = {}, set(), {}
    for name, fac in ZOO.items():
        oc, _, _ = fac()
        specs[name] = oc.model_spec()
        keep.add(specs[name].hash())
    # (hipcc runs as child processes, every build in a work directory of its own: a few at a time -- a fresh clone builds the zoo in
    #  ~4 minutes instead of ~11)
    with ThreadPoolExecutor(max_workers=max(1, min(4, (os.cpu_count() or 2) // 2))) as pool:
        futs = {name: pool.submit(runtime.build_library, spec, force, verbose) for name, spec in specs.items()}
        for name, fut in futs.items():
            out[name] = fut.result()
            if verbose:
                print("built", name, specs[name].hash(), out[name])
    manifest = os.path.join(runtime.GEN_DIR, "ZOO_MANIFEST.json")
    old = set()
    if os.path.exists(manifest):
        try:
            old = set(json.load(open(manifest)).get("hashes", []))
        except Exception:
            old = set()
    for h in old - keep:
        for path in (runtime.header_path(h), runtime.library_path(h), runtime.isa_record_path(runtime.library_path(h))):
            if os.path.exists(path):
                os.remove(path)
    with open(manifest, "w") as f:
        json.dump({"hashes": sorted(keep)}, f, indent=1)
    return out
