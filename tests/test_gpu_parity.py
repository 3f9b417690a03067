"""-m gpu tier: the real gfx950 libraries, called through the C ABI, vs the oracle and the golden fixture."""
import os

import numpy as np
import pytest
import torch

import lfsd_amd  # noqa: F401
from lfsd_amd import CPDP, models, runtime
from conftest import make_oracle, oracle_loss_grad, oracle_parallel, oracle_check_solution, assert_grids_match, parity_record
import parity_cases as pc

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


def gpu_prepare(oc, dtype):
    oc.setDevice("cuda:0", dtype)
    assert not oc.compile().is_emulator, "the GPU tier must run the HIP library"
    return oc


@pytest.mark.parametrize("kind", ["pendulum", "robotarm", "cartpole", "quadrotor"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_every_output_grid_vs_oracle(kind, dtype, oc_mapping):
    """state / control / costate grids, Z = [P W], dx/dtheta, du/dtheta, loss, gradient -- all of them, on the GPU, against
    the tight oracle (tolerances: parity_cases.TOL)."""
    pc.all_grids_vs_oracle(gpu_prepare, kind, dtype)


@pytest.mark.parametrize("kind", ["pendulum", "robotarm"])
def test_reference_shaped_single_trajectory_api(kind):
    """oc.cocSolver(ini_state, horizon, theta) / oc.auxSysSolver(time_grid, opt_sol, theta) with the reference's signatures
    and return types (CPDP.py:92, 301), through the HIP library."""
    pc.single_trajectory_api(gpu_prepare, kind)


@pytest.mark.parametrize("substeps", [0, 4])        # 0: the library default bench.py runs (error-controlled, rtol 1e-3); 4: fixed units
@pytest.mark.parametrize("dtype,ltol,gtol,xtol", [(torch.float64, 2e-7, 1e-5, 2e-6), (torch.float32, 5e-5, 1e-3, 3e-4)])
def test_quadrotor_bench_seeds_vs_tight_oracle(dtype, ltol, gtol, xtol, substeps):
    """The headline configuration itself (n_grid 50, the first seeds bench.py draws; library defaults and fixed 4 units)
    against the TIGHT oracle (Radau, rtol 1e-10), so the floor of the shipped fp32 path is known apart from the reference
    integrator's own 5e-3.  Measured on MI355X (profiles/r03_a_parity_floors.jsonl), fp64 / fp32: state 9e-7 / 1.1e-5, control
    4e-7 / 2.9e-5, costate 3e-7 / 9e-6, [P W] 1e-5 / 1.4e-4, dx/dth 2.7e-4 / 2.7e-4, loss 2e-8 / 3.6e-6, gradient 5e-7 / 1.1e-4;
    asserted at about 10x (fp64 grids at 2x: they sit at the solver's tolerance): see `tol` below.  du/dth 1.6e-2 in both
    precisions -- the discretisation of its value at t = T, 10^3 x the error of dx/dth."""
    oc, d = gpu_model("quadrotor", dtype, 50, substeps=substeps)
    oc.setSolverOptions(aux_rtol=1e-3 if substeps == 0 else 0.0)
    rng = np.random.default_rng(1234)
    th = np.array(d["theta0"])[None, :] + 0.05 * rng.standard_normal((4096, 7))
    th[:, 0] = np.abs(th[:, 0]) + 0.5
    th = th[:4]
    sol = oc.cocSolverBatch(np.tile(d["ini_state"], (4, 1)), d["horizon"], th)
    aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"], want_grids=True)
    assert set(sol["status"].tolist()) <= {1, 2}
    refs = oracle_parallel([dict(kind="quadrotor", n_grid=50, ini_state=d["ini_state"], horizon=d["horizon"], theta=list(t),
                                 taus=d["taus"], wps=d["waypoints"], iface=d["interface"]) for t in th])
    # fp64 (round 4): loss 1.6e-8 -> 2e-7, gradient 4.6e-7 -> 1e-5, costate 2.8e-7 -> 3e-6, [P W] 1.0e-5 -> 1e-4 (measured -> asserted)
    f64 = dtype == torch.float64
    tol = dict(grid=xtol, costate=(1.5 if f64 else 10) * xtol, Z=1e-4 if f64 else 3e-3, aux=3e-3, auxU=5e-2, loss=ltol, grad=gtol)
    for b in range(4):
        assert_grids_match(sol, aux, b, refs[b], 13, 4, 7, tol, what="bench seed %d %s" % (b, dtype))


@pytest.mark.parametrize("dtype,ltol", [(torch.float64, 1e-6), (torch.float32, 5e-4)])
def test_quadrotor_vs_reference_golden_run(dtype, ltol, oc_mapping):
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


def test_level0_of_the_mesh_continuation_on_the_saved_runs_problem_even_grid():
    """The reference's saved run was made at n_grid 25 -- a grid on which level 0 of the lean kernels' mesh continuation (merged
    intervals: even n_grid >= 20) never runs.  The same PROBLEM (the run's start, goal, horizon, waypoints, and 12 parameter vectors
    of its trace) at n_grid 26, fp32 at the library defaults on the lock-step mapping (level 0, the coarse level, the working-
    precision stops, error-controlled sweeps), against (a) the tight oracle at n_grid 26 -- headline tolerances -- and (b) the
    reference's own saved loss at n_grid 25: the two discretisations of one continuous problem differ by 5e-5 ... 5e-3 of the loss
    (oracle at 25 vs 26: 4e-5 at theta_0, 4.7e-3 at iteration 60), asserted at 1e-2."""
    n_grid = 26
    oc, d = gpu_model("quadrotor", torch.float32, n_grid, substeps=0)
    oc.setSolverOptions(mapping="lockstep")
    consts = oc.consts_tensor(overrides=dict(goal_r0=G["goal_r"][0], goal_r1=G["goal_r"][1], goal_r2=G["goal_r"][2]))
    idx = [0, 3, 7, 12, 20, 30, 40, 50, 60, 75, 90, 99]
    pad = 4096                                                    # (a batch the lock-step lean kernel is the library's own choice for)
    th = np.tile(G["lookahead_theta"][idx], (pad // len(idx) + 1, 1))[:pad]
    sol = oc.cocSolverBatch(np.tile(G["ini_state"], (pad, 1)), float(G["horizon"]), th, consts=consts)
    aux = oc.auxSysSolverBatch(sol, G["taus"], G["waypoints"], [0, 1, 2], want_grids=True)
    assert set(sol["status"].tolist()) <= {1, 2}
    assert float(sol["iters"].float().mean()) >= 5.0              # (cold starts: level 0 + coarse + fine iterations, not a warm path)
    refs = oracle_parallel([dict(kind="quadrotor", n_grid=n_grid, ini_state=list(G["ini_state"]), horizon=float(G["horizon"]),
                                 theta=list(G["lookahead_theta"][j]), taus=list(G["taus"]), wps=G["waypoints"].tolist(), iface=[0, 1, 2],
                                 make_kw=dict(goal=tuple(float(v) for v in G["goal_r"]))) for j in idx])
    # (du/dtheta: its value at T carries the last interval's discretisation error times one gain, parity_cases.dudtheta_refinement; an
    #  interval of this grid is twice as long as the headline's -- measured 5.7e-2 against the headline's 1.6e-2)
    # (gradient: late trace points are near-stationary, |dtheta| ~ 0.05, which amplifies fp32 rounding relative to its size --
    #  measured 2.1e-3 at trace point 50, 1e-4 ... 1e-3 elsewhere; the golden-run test above allows 2e-2 for the same reason)
    tol = dict(grid=3e-4, costate=5e-3, Z=3e-3, aux=3e-3, auxU=1.5e-1, loss=5e-5, grad=6e-3)
    for k, j in enumerate(idx):
        assert_grids_match(sol, aux, k, refs[k], 13, 4, 7, tol, what="saved run's problem at n_grid 26, trace point %d" % j)
        parity_record("saved run's problem at n_grid 26 vs the reference's loss at n_grid 25, trace point %d" % j, "loss",
                      abs(float(aux["loss"][k]) - G["loss_trace"][j]) / G["loss_trace"][j], 1e-2)
        # duplicates of a problem anywhere in the batch come back bit-identical (level 0's schedule does not depend on partners)
        assert torch.equal(sol["state_grid"][k], sol["state_grid"][k + len(idx)])


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


def test_headline_workload_has_no_stragglers_and_stops_at_the_fp64_answer():
    """The benchmark's own learner (bench.build_learner: 4096 seeds, Nesterov), 12 outer iterations.  A launch lasts as long
    as its slowest wavefront, so one trajectory wandering at the fp32 gradient floor used to double oc_solve's time
    (profiles/r02_h_oc_straggler.txt): the working-precision test must end every solve within 8 iterations at every outer
    iteration -- and what it accepts must be the fp64 solve's answer: loss 5e-4, gradient 1e-2 (the stated fp32 tolerance of
    the pipeline) on a random subset at the parameters of the last outer iteration."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    args = bench.parse_args(["--no-cpu-baseline"])
    oc, env, d = models.quadrotor(n_grid=args.n_grid)
    oc = gpu_prepare(oc, torch.float32)
    L, theta0, x0 = bench.build_learner(args, oc, d, oc.compile(), 0, 1, "independent")
    L.count_unconverged = False
    for k in range(12):
        th = oc.compile().lookahead(L.theta, L.m, L.mu).clone()
        L.step()
        it, st = L._sol["iters"].cpu().numpy(), L._sol["status"].cpu().numpy()
        assert np.isin(st, (1, 2)).all(), (k, np.bincount(st, minlength=5))
        assert it.max() <= 8, (k, np.bincount(it))
    sub = np.random.default_rng(3).choice(args.batch, 48, replace=False)
    oc64, _, _ = models.quadrotor(n_grid=args.n_grid)
    oc64 = gpu_prepare(oc64, torch.float64)
    sol64 = oc64.cocSolverBatch(x0[sub], d["horizon"], th[sub].double())
    aux64 = oc64.auxSysSolverBatch(sol64, d["taus"], d["waypoints"], d["interface"])
    l32, l64 = L._aux["loss"][sub].double().cpu().numpy(), aux64["loss"].cpu().numpy()
    g32, g64 = L._aux["grad"][sub].double().cpu().numpy(), aux64["grad"].cpu().numpy()
    # (the comparison with the ORACLE at these parameters is test_headline_defaults_vs_tight_oracle_at_outer_iteration_12;
    #  this one covers 48 more seeds against fp64 HIP.  Measured r03: loss 3e-6, gradient 4e-4)
    parity_record("headline fp32 vs fp64 HIP, 48 seeds at outer iteration 12", "loss", (np.abs(l32 - l64) / np.maximum(1.0, l64)).max(), 5e-5)
    parity_record("headline fp32 vs fp64 HIP, 48 seeds at outer iteration 12", "grad", (np.abs(g32 - g64).max(axis=1) / np.abs(g64).max(axis=1)).max(), 3e-3)


@pytest.mark.parametrize("n_outer", [12, 25])
def test_headline_defaults_vs_tight_oracle_at_outer_iteration_12(n_outer):
    """(n_outer = 25, round 5: the regime the DRIVER times -- `bench.py --warmup 5 --steps 20` ends at outer iteration 25, where a
    third of the batch ends at working precision (status 2) against a few per cent at iteration 12; there the sample is 16 rows of
    each status, at the same tolerances.)
    Where the work-saving shortcuts of the shipped fp32 path act -- the working-precision stop of the OC solve (status 2,
    cpdp_oc.h at_working_precision / the costate-only last sweep), the error-controlled unit count and the midpoint rule
    of the auxiliary sweeps (library defaults: rtol 1e-3 from one unit) -- they are compared with the TIGHT ORACLE, not
    with fp64 HIP: the benchmark's own learner runs 12 outer iterations, and at the parameters of the 12th (later
    iterations need more split units than theta_0) six trajectories that ended CONVERGED and six that ended at WORKING
    PRECISION go through the fp64 oracle (IPOPT-equivalent solve, Radau rtol 1e-10 sweeps) -- round 4: up to sixteen of the
    latter and 32 trajectories in all on the oracle fan-out (conftest.oracle_parallel, one worker per host core).
    Measured floors (profiles/r03_a_parity_floors.jsonl): state 2e-5, loss 3e-6, gradient 3e-4; asserted with a margin of
    ~5x: state 2e-4, costate 5e-3, loss 5e-5, gradient 2e-3."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    args = bench.parse_args(["--no-cpu-baseline"])
    oc, env, d = models.quadrotor(n_grid=args.n_grid)
    oc = gpu_prepare(oc, torch.float32)
    oc.setSolverOptions(aux_substeps=args.substeps, aux_rtol=args.aux_rtol)        # the library defaults bench.py runs
    L, theta0, x0 = bench.build_learner(args, oc, d, oc.compile(), 0, 1, "independent")
    for k in range(n_outer):
        th = oc.compile().lookahead(L.theta, L.m, L.mu).clone()
        L.step()
    st = L._sol["status"].cpu().numpy()
    assert np.isin(st, (1, 2)).all(), np.bincount(st, minlength=5)
    rng = np.random.default_rng(n_outer)
    pick = []
    n2 = int((st == 2).sum())
    assert n2 >= (3 if n_outer == 12 else 16) and int((st == 1).sum()) >= 26, np.bincount(st, minlength=5)        # both exits are really taken on this workload
    k2 = min(16, n2)                                                                   # (9 of 4096 ended at status 2 in the round-4 run, 5 with level 0 of the mesh continuation)
    for code, cnt in ((2, k2), (1, 32 - k2)):
        idx = np.where(st == code)[0]
        pick += list(rng.choice(idx, cnt, replace=False))
    th64 = th.double().cpu().numpy()
    refs = oracle_parallel([dict(kind="quadrotor", n_grid=args.n_grid, ini_state=d["ini_state"], horizon=d["horizon"],
                                 theta=list(th64[b]), taus=d["taus"], wps=d["waypoints"], iface=d["interface"]) for b in pick])
    tol = dict(grid=2e-4, costate=5e-3, loss=5e-5, grad=2e-3)
    for b, r in zip(pick, refs):
        assert_grids_match(L._sol, L._aux, int(b), r, 13, 4, 7, tol,
                           what="headline fp32 defaults, outer iteration %d, seed %d, status %d" % (n_outer, b, st[b]))


def test_configs0_pendulum_horizon50_single_seed():
    """BASELINE configs[0]: SinglePendulum (JinEnv), horizon (n_grid) 50, ONE seed -- the reference's own CPU-runnable
    case -- in fp64 against the tight oracle, through the batch API and through the reference-shaped calls."""
    pc.configs0_pendulum(gpu_prepare)


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_bench_two_ranks_share_theta_through_the_allreduce(backend):
    """N > 1 on hardware: `bench.py --gpus 2` as a FRESH child process (its parent never touches a GPU and starts
    torch.distributed.run; nothing is re-executed from this pytest process).  With gloo both ranks land on the one GPU of
    the box (local_rank % device_count); with nccl (= RCCL) the test needs two GPUs and is skipped otherwise.  Rank 0's
    JSON line must say n_gpus 2 / mode shared, and the single theta every rank holds after warmup + steps iterations must
    equal a single-process mode='shared' run over the UNION of the two ranks' demonstrations (fp64, 1e-10): the summed
    gradient really went through the all-reduce and drove the update (QuadAlgorithm.py:469-495 / robotarm_random.py:60-73
    is the loop being sharded)."""
    import json
    import subprocess
    import sys
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL run needs two GPUs (the driver's SCALE run covers it)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    argv = ["--gpus", "2", "--backend", backend, "--batch", "256", "--steps", "2", "--warmup", "1", "--dtype", "f64",
            "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["mode"] == "shared" and out["scaling"] == "weak"
    assert out["config"]["n_unconverged_last_step"] == 0
    assert out["value"] > 0 and abs(out["value"] - 2 * 256 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    # single process, union of the two shards, the same three iterations
    args = bench.parse_args(argv)
    oc, env_, d = models.quadrotor(n_grid=args.n_grid)
    oc = gpu_prepare(oc, torch.float64)
    oc.setSolverOptions(aux_substeps=args.substeps, aux_rtol=args.aux_rtol)
    parts = [bench.demo_set(args, d, rk, "shared") for rk in (0, 1)]
    union = dict(x0=np.concatenate([p_["x0"] for p_ in parts]), goal=np.concatenate([p_["goal"] for p_ in parts]),
                 wps=np.concatenate([p_["wps"] for p_ in parts]), theta0=parts[0]["theta0"])
    L = bench.shared_learner(oc, d, union, 2 * args.batch)
    for _ in range(args.warmup + args.steps):
        L.step()
    th = L.theta.double().cpu().numpy().ravel()
    assert L.n_unconverged == 0
    assert np.abs(np.array(out["config"]["theta"]) - th).max() < 1e-10 * np.abs(th).max(), (out["config"]["theta"], th.tolist())


def test_rccl_one_rank_allreduce_of_the_summed_gradient_is_bit_identical():
    """First RCCL bytes on the box's one GPU: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --mode shared
    --backend nccl` as a FRESH child (this pytest process has touched the GPU; nothing is re-executed from it).  Under a launcher
    bench.py initialises the process group whatever the world size, and SparseDemoLearner.step takes the all-reduce branch whenever a
    group is initialised (CPDP.py `torch.distributed.all_reduce(buf, group=self.pg)`): RCCL loads, creates a communicator on the
    device and reduces the p+2 numbers (summed d(theta), loss, masked-row count) every outer iteration.  A one-rank sum must leave the
    buffer unchanged, so theta after warmup + steps iterations equals -- bit for bit -- the same command without a process group."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv = ["--gpus", "1", "--mode", "shared", "--backend", "nccl", "--batch", "256", "--steps", "2", "--warmup", "1", "--dtype", "f64",
            "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    outs = []
    for cmd in (launcher + [os.path.join(root, "bench.py")] + argv, [sys.executable, os.path.join(root, "bench.py")] + argv):
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(env, NCCL_DEBUG="VERSION"), timeout=900)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        outs.append((json.loads(lines[0]), r.stdout + r.stderr))
    (with_pg, log_pg), (without, _) = outs
    assert with_pg["n_gpus"] == 1 and with_pg["config"]["mode"] == "shared" and with_pg["config"]["process_group"] == "nccl"
    assert without["config"]["process_group"] is None
    assert "RCCL" in log_pg or "NCCL version" in log_pg, log_pg[-2000:]          # the communicator really came up (NCCL_DEBUG=VERSION banner)
    assert with_pg["config"]["n_unconverged_last_step"] == 0
    assert with_pg["config"]["theta"] == without["config"]["theta"], (with_pg["config"]["theta"], without["config"]["theta"])


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


def test_rocket_newton_mode_vs_oracle(oc_mapping):
    """BASELINE configs[4] robot (Examples/rocket_groundtruth.py): exact stage Hessians from the first iteration.  Several
    local minima: which one a cold start reaches depends on the globalisation (IPOPT's would differ from ours and from the
    oracle's), so parity is basin-independent -- the oracle certifies the kernel's cold-start answer as a KKT point of the
    reference's NLP with complex-step arithmetic and differentiates the PMP along it; and the oracle's own KKT point,
    handed over as the initial guess, is where the kernel stays."""
    o = make_oracle("rocket", 15)
    oc, d = gpu_model("rocket", torch.float64, 15, substeps=8)
    th = d["true_theta"]
    idx = [1, 3, 6, 10, 13]
    taus = np.linspace(0, d["horizon"], 16)[idx]
    sol = oc.cocSolverBatch([d["ini_state"]] * 3, d["horizon"], [th] * 3)
    assert sol["status"].tolist() == [1, 1, 1]
    X, U, Lm = (sol[k][2].cpu().numpy() for k in ("state_grid", "control_grid", "costate_grid"))
    wps = [np.concatenate([X[k, 0:3], X[k, 6:10]]) + 0.05 for k in idx]
    aux = oc.auxSysSolverBatch(sol, taus, wps, d["interface"])
    r = oracle_check_solution(o, d["ini_state"], d["horizon"], th, X, U, Lm, taus, wps, d["interface"])
    assert r["defect"] < 1e-9 and r["gmax"] < 1e-6 and r["lmax"] < 1e-6 * np.abs(Lm).max(), (r["defect"], r["gmax"], r["lmax"])
    assert abs(aux["loss"][1].item() - r["loss"]) < 1e-7 * max(1.0, r["loss"]) and rel(aux["grad"][1], r["grad"]) < 1e-3
    r0 = o.cocSolver(d["ini_state"], d["horizon"], th, return_grids=True, exact_after=0, max_iter=400)
    assert o.last_info["converged"]
    u0 = torch.as_tensor(r0[3][None, :-1].copy(), device="cuda:0")
    sol2 = oc.cocSolverBatch([d["ini_state"]], d["horizon"], [th], u_init=u0)
    assert sol2["status"].tolist() == [1] and abs(sol2["cost"][0].item() - o.last_cost) < 1e-8 * abs(o.last_cost)
    assert rel(sol2["state_grid"][0], r0[2]) < 1e-6 and rel(sol2["costate_grid"][0], r0[4]) < 1e-5


def test_rocket_lean_kernels_fp64_vs_fp32_same_iterates(monkeypatch):
    """The lean lock-step kernels on the rocket's dimensions (13 live columns, 3 constant states, 3 controls; the rocket's
    default solve never gets here): six Gauss-Newton / Hamiltonian iterations from a cold start in fp64 (live-column roll-out,
    parked RK4 step, one-pass structural sweep with LDS-fed products) and in fp32 (packed roll-out, structural sweep on the
    matrix cores) walk the same path -- ragged batch of 6 (one full wavefront of four + a partial one)."""
    monkeypatch.setattr(CPDP.COCSys, "mapping_override", "lockstep")
    monkeypatch.setenv("LFSD_F64_SEED", "0")      # the fp64 kernel's own six iterations, not the fp32-seeded solve
    oc, env, d = models.rocket(n_grid=15)
    oc.setSolverOptions(max_iter=6, exact_after=-1)
    rng = np.random.default_rng(3)
    th = np.array([d["true_theta"], d["theta0"]] * 3, dtype=float) * (1 + 0.03 * rng.standard_normal((6, len(d["theta0"]))))
    x0 = np.tile(d["ini_state"], (6, 1))
    out = {}
    for dt in (torch.float64, torch.float32):
        gpu_prepare(oc, dt)
        out[dt] = oc.cocSolverBatch(x0, d["horizon"], th)
        assert (out[dt]["iters"] == 6).all() and (out[dt]["status"] == 3).all(), (dt, out[dt]["iters"], out[dt]["status"])
    l64, l32 = out[torch.float64], out[torch.float32]
    what = "rocket lean kernels, fp32 vs fp64 after 6 iterations"
    # (measured on MI355X and, identically, in the CPU emulator: cost 3.1e-4, states 2.5e-4 -- one of the six trajectories,
    #  at cost 1.4e4 after iteration 4, takes a marginally different step in fp32; the other five agree to 6e-6)
    parity_record(what, "cost", rel(l32["cost"].double(), l64["cost"].cpu().numpy()), 3e-3)
    parity_record(what, "state grid", rel(l32["state_grid"].double(), l64["state_grid"].cpu().numpy()), 3e-3)
    parity_record(what, "control grid", rel(l32["control_grid"].double(), l64["control_grid"].cpu().numpy()), 3e-2)
    c64, c32 = l64["cost"].cpu().numpy(), l32["cost"].double().cpu().numpy()
    assert np.median(np.abs(c32 - c64) / np.abs(c64)) < 2e-5


def test_rocket_fp32_solve_fp64_aux_vs_oracle(oc_mapping):
    """The mixed-precision path of BASELINE configs[4] (setDevice(aux_dtype=float64), CPDP.py:247-250 of this package)."""
    pc.rocket_mixed_precision(gpu_prepare)


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


def _arm_seeds(B=1024):
    rng = np.random.default_rng(0)
    th = np.array([5.0, 1, 1, 1, 1])[None, :] * (1 + 0.05 * rng.standard_normal((B, 5)))     # around the example's initial guess
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    return th


def test_robotarm_batch1024_random_seeds_configs1():
    """BASELINE configs[1]: robot arm, n_grid 50, 1024 random seeds.  EVERY cold-started solve has to end converged (or
    stalled at fp32 working precision) -- at the perturbed initial guesses theta_0 and at theta_1 = theta_0 - lr*grad_0
    (Examples/robotarm_random.py:67-73), where the round-1 step control left 64 of 1024 at the iteration limit -- in
    fp32 and in fp64, with the fp32 loss within 2e-3 of fp64 on every seed and the fp32 gradient within 2e-2 (flat-cost
    arm) on every seed at theta_0 and on >= 98 % of them at theta_1 (see below)."""
    B = 1024
    th0 = _arm_seeds(B)
    res = {}
    thetas = {"theta0": th0}
    for name in ("theta0", "theta1"):
        for dt in (torch.float64, torch.float32):
            oc, d = gpu_model("robotarm", dt, 50, substeps=4)
            sol = oc.cocSolverBatch(np.tile(d["ini_state"], (B, 1)), d["horizon"], thetas[name])
            aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"])
            st = sol["status"].cpu().numpy()
            assert np.isin(st, (1, 2)).mean() == 1.0, (dt, name, np.bincount(st, minlength=5))
            res[(dt, name)] = (aux["loss"].double().cpu().numpy(), aux["grad"].double().cpu().numpy())
            if dt == torch.float64:
                x64 = sol["state_grid"].clone()           # (the fp32 solve follows: `sol` is then the fp32 one)
        if name == "theta0":
            th1 = th0 - d["lr"] * res[(torch.float64, "theta0")][1]
            th1[:, 0] = np.maximum(th1[:, 0], 1e-8)
            thetas["theta1"] = th1
        l32, g32 = res[(torch.float32, name)]
        l64, g64 = res[(torch.float64, name)]
        parity_record("robot arm 1024 seeds fp32 vs fp64 HIP at " + name, "loss", (np.abs(l32 - l64) / np.maximum(1.0, l64)).max(), 5e-4)      # measured 4e-5 / 8e-5
        gerr = np.abs(g32 - g64).max(1) / np.abs(g64).max(1)
        parity_record("robot arm 1024 seeds fp32 vs fp64 HIP at " + name, "grad, median over seeds", float(np.median(gerr)), 1e-3)      # measured 8e-5 / 1.4e-4
        if name == "theta0":
            parity_record("robot arm 1024 seeds fp32 vs fp64 HIP at theta0", "grad, worst seed", gerr.max(), 2e-2)      # measured 3.7e-3
        else:
            # (the share depends on the fp32 rounding of the build: 1.46 % in round 3, 1.27 % in round 4 after sin / cos changed by an
            #  ulp -- asserted with room for that, 3 %; the 95th percentile below holds the bulk)
            parity_record("robot arm 1024 seeds fp32 vs fp64 HIP at theta1", "share of seeds with grad error > 2e-2", float((gerr >= 2e-2).mean()), 0.03)
            # theta_1: about 1 % of the seeds sit next to a conjugate point of the optimal-control problem (the tight oracle's
            # Riccati integration has a finite escape there, next test); the same KKT point is found (loss above), but its
            # sensitivity is ill-conditioned with respect to the trajectory itself: fp32 round-off of the SOLVE moves it by
            # O(1), whichever precision the auxiliary pass runs in.  Stated: >= 97 % of the seeds within 2e-2.
            assert (gerr < 2e-2).mean() >= 0.97, (name, (gerr < 2e-2).mean())
            parity_record("robot arm 1024 seeds fp32 vs fp64 HIP at theta1", "grad, 95th percentile", float(np.quantile(gerr, 0.95)), 2e-2)
            # A/B over the whole population against the build with every schedule off (single shooting only): the product's fp32
            # state and gradient errors may not be worse than 2x that build's on the SAME seeds (or the stated floor) -- round 5's
            # product was 3.8x worse in the median state error (profiles/r06_k_robotarm_accuracy_ab.txt); measured round 6:
            # state p50 1.5e-4 / p90 9.6e-4 (plain 8.1e-5 / 7.4e-4), gradient p90 8.0e-4 (7.3e-4), share beyond 2e-2 1.3 % (1.1 %)
            from conftest import build_variant_library, PLAIN_SCHEDULE
            ocp, _ = gpu_model("robotarm", torch.float32, 50, substeps=4)
            ocp.use_library(build_variant_library(ocp, "plain", PLAIN_SCHEDULE))
            ocp.setDevice("cuda:0", torch.float32)
            solp = ocp.cocSolverBatch(np.tile(d["ini_state"], (B, 1)), d["horizon"], thetas[name])
            auxp = ocp.auxSysSolverBatch(solp, d["taus"], d["waypoints"], d["interface"])
            assert np.isin(solp["status"].cpu().numpy(), (1, 2)).all()
            gp = auxp["grad"].double().cpu().numpy()
            gerr_p = np.abs(gp - g64).max(1) / np.abs(g64).max(1)
            xerr = lambda s_: ((s_["state_grid"].double() - x64).abs().flatten(1).max(1)[0] / x64.abs().flatten(1).max(1)[0]).cpu().numpy()
            xe, xe_p = xerr(sol), xerr(solp)
            for what_, mine, ref_, floor in (("state, median", np.median(xe), np.median(xe_p), 3e-4), ("state, 90th percentile", np.quantile(xe, .9), np.quantile(xe_p, .9), 1e-3),
                                             ("grad, 90th percentile", np.quantile(gerr, .9), np.quantile(gerr_p, .9), 1e-3),
                                             ("grad, 95th percentile", np.quantile(gerr, .95), np.quantile(gerr_p, .95), 3e-3),
                                             # (the 99th percentile of 1 024 seeds is the tenth-worst of the ~13 that sit next to a conjugate
                                             #  point, a figure of the rounding -- measured 2.5e-2 and 8.1e-2 on two builds of the product, 3.2e-2 /
                                             #  3.4e-2 on the plain one: the class is held through its SHARE instead)
                                             ("share of seeds with grad error > 2e-2", (gerr >= 2e-2).mean(), (gerr_p >= 2e-2).mean(), 0.02)):
                parity_record("robot arm 1024 seeds fp32 vs fp64 HIP at theta1, PLAIN build", what_, float(ref_), float("inf"))
                parity_record("robot arm 1024 seeds fp32 vs fp64 HIP at theta1, product vs plain", what_, float(mine), max(floor, 2 * float(ref_)))


def test_robotarm_theta1_vs_oracle_16_seeds():
    """HIP vs the tight oracle at theta_1 on >= 16 of the 1024 seeds: 12 drawn at random and the 12 with the largest
    gradients below 100x the typical one (sensitivities of 10-100x: these parameters sit next to a conjugate point of the
    optimal-control problem; the oracle confirms the numbers).  Where the tight Riccati integration itself runs into the
    pole (finite escape: scipy gives up, as the reference's BDF call would or would step across) there is no number to
    compare with and the seed is skipped; at least 16 comparisons must remain."""
    B = 1024
    th0 = _arm_seeds(B)
    oc, d = gpu_model("robotarm", torch.float64, 50, substeps=8)
    x0 = np.tile(d["ini_state"], (B, 1))
    sol = oc.cocSolverBatch(x0, d["horizon"], th0)
    aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"])
    th1 = th0 - d["lr"] * aux["grad"].cpu().numpy()
    th1[:, 0] = np.maximum(th1[:, 0], 1e-8)
    sol = oc.cocSolverBatch(x0, d["horizon"], th1)
    aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"])
    g = aux["grad"].cpu().numpy()
    gmax = np.abs(g).max(1)
    cand = np.where(gmax < 100 * np.median(gmax))[0]
    pick = list(cand[np.argsort(-gmax[cand])[:12]]) + list(np.random.default_rng(1).choice(B, 12, replace=False))
    refs = oracle_parallel([dict(kind="robotarm", n_grid=50, ini_state=d["ini_state"], horizon=d["horizon"],
                                 theta=list(th1[b]), taus=d["taus"], wps=d["waypoints"], iface=d["interface"],
                                 allow_fail=True) for b in pick])
    oc32, _ = gpu_model("robotarm", torch.float32, 50, substeps=8)
    sol32 = oc32.cocSolverBatch(x0[pick], d["horizon"], th1[pick])
    aux32 = oc32.auxSysSolverBatch(sol32, d["taus"], d["waypoints"], d["interface"])
    # A/B of the SAME seeds on the build with every schedule off (single shooting, no coarse levels: `plain`): the fp32 bounds below
    # are anchored to what that build measures on the same seeds, so a tolerance cannot simply follow the product any more
    from conftest import build_variant_library, PLAIN_SCHEDULE
    ocp, _ = gpu_model("robotarm", torch.float32, 50, substeps=8)
    ocp.use_library(build_variant_library(ocp, "plain", PLAIN_SCHEDULE))
    ocp.setDevice("cuda:0", torch.float32)
    assert not ocp.compile().is_emulator
    solp = ocp.cocSolverBatch(x0[pick], d["horizon"], th1[pick])
    auxp = ocp.auxSysSolverBatch(solp, d["taus"], d["waypoints"], d["interface"])
    compared = 0
    large_err, large_err_plain, typ_err, typ_err_plain = [], [], [], []
    for k, b in enumerate(pick):
        r = refs[k]
        if "error" in r:
            assert "Riccati" in r["error"], r["error"]
            continue
        compared += 1
        big = np.abs(r["grad"]).max() > 10 * np.median(gmax)
        assert rel(sol["state_grid"][b], r["X"]) < 1e-6 and rel(sol["costate_grid"][b], r["L"]) < 1e-5, b
        assert abs(aux["loss"][b].item() - r["loss"]) < 1e-6 * max(1.0, r["loss"]), b
        assert rel(aux["grad"][b], r["grad"]) < (1e-3 if big else 1e-4), (b, g[b], r["grad"])
        assert abs(aux32["loss"][k].item() - r["loss"]) < 2e-3 * max(1.0, r["loss"]), b
        # fp32: 2e-2 where the sensitivity is of typical size; the seeds picked for their large gradients sit next to a
        # conjugate point, where fp32 round-off of the solve itself moves the gradient by tens of percent
        typical = np.abs(r["grad"]).max() < 3 * np.median(gmax)
        e32 = rel(aux32["grad"][k], r["grad"])
        e32p = rel(auxp["grad"][k], r["grad"])
        what = "robot arm theta1 seed %d fp32 vs oracle (%s sensitivity)" % (b, "typical" if typical else "large")
        parity_record(what + ", PLAIN build", "grad", e32p, float("inf"))      # (recorded, not asserted: the yardstick)
        if typical:
            # round 4's bound, 5e-3 -- or, for a seed on which the plain build itself is above it, twice what the plain build measures
            # (where the fp32 solve stops inside its tolerance decides: seed 889 measured 4.1e-4 and 6.2e-3 on two builds whose populations agree)
            parity_record(what, "grad", e32, max(5e-3, 2 * e32p))
            typ_err.append(e32); typ_err_plain.append(e32p)
        else:
            # next to a conjugate point the fp32 gradient is a property of the rounding, not of the kernel: the SAME seed measured 0.27
            # (round 3), 0.45 and 0.66 (round 4), 2.5 - 3.5 (round 5) -- the figure moves by its own size with any change of the fp32 path.
            # Per seed: finite; the class is asserted below, against round 4's bounds AND against the plain build on the same seeds
            assert np.isfinite(e32), (b, e32)
            parity_record(what, "grad", e32, max(5.0, 2 * e32p))
            large_err.append(e32); large_err_plain.append(e32p)
    assert compared >= 16, compared
    assert len(large_err) >= 6
    # the product's class figures may not be worse than the plain build's on the same seeds by more than 2x, nor than round 4's bounds
    # (measured in round 4, profiles/r04_final_parity_floors.jsonl: 0.0 0.0 0.001 0.004 0.013 0.016 0.039 0.14 0.15 0.66)
    for name, q, floor in (("median", 0.5, 0.1), ("80th percentile", 0.8, 0.7)):
        mine, ref_ = float(np.quantile(large_err, q)), float(np.quantile(large_err_plain, q))
        parity_record("robot arm theta1 fp32 vs oracle, large-sensitivity seeds, PLAIN build", name + " gradient error", ref_, float("inf"))
        parity_record("robot arm theta1 fp32 vs oracle, large-sensitivity seeds", name + " gradient error", mine, max(floor, 2 * ref_))
    # ... and a FIXED count: at most one seed of the class beyond 1 more than the plain build has
    n_mine, n_ref = int((np.array(large_err) > 1).sum()), int((np.array(large_err_plain) > 1).sum())
    assert n_mine <= n_ref + 1, (n_mine, n_ref, sorted(large_err), sorted(large_err_plain))
    if typ_err:
        parity_record("robot arm theta1 fp32 vs oracle, typical seeds", "median gradient error", float(np.median(typ_err)),
                      max(1e-3, 2 * float(np.median(typ_err_plain))))


def test_robotarm_12_vanilla_steps_every_gradient_applied():
    """Twelve plain gradient steps at the example's learning rate 0.1 with skip_unconverged=False (the reference's loop,
    Examples/robotarm_random.py:60-73), 1024 seeds, fp32.  The fixed learning rate throws the few large-sensitivity seeds
    (previous test) out of the region where the problem is well posed -- negative quadratic state weights make the
    running cost non-convex; no solver has a KKT point to return there (J runs to -10^3 ... -10^4) -- so the assertion
    is on the admissible seeds: parameters finite and below 1e3 (they start at 1..5; a seed at 1e17 has left the problem
    for good), beta > 0, both quadratic weights > 0.05.  Every one of them has to converge at every step."""
    from lfsd_amd import CPDP
    B = 1024
    th0 = _arm_seeds(B)
    oc, d = gpu_model("robotarm", torch.float32, 50, substeps=4)
    L = CPDP.SparseDemoLearner(oc, np.tile(d["ini_state"], (B, 1)), d["horizon"], d["taus"], d["waypoints"], d["interface"], th0,
                               method="Vanilla", learning_rate=d["lr"], skip_unconverged=False)
    for k in range(13):
        th = L.theta.double().cpu().numpy()
        adm = np.isfinite(th).all(1) & (np.abs(th) < 1e3).all(1) & (th[:, 0] > 0) & (th[:, 1] > 0.05) & (th[:, 3] > 0.05)
        L.step()
        st = L._sol["status"].cpu().numpy()
        assert np.isin(st[adm], (1, 2)).mean() == 1.0, (k, np.bincount(st[adm], minlength=5))
        assert adm.mean() > 0.96, (k, adm.sum())      # (993-1000 of 1024 after 12 steps, depending on fp32 rounding of the build)
        if k <= 1:
            assert adm.all(), k
        # ... and the exclusion is not a convenience of this test: the solves that do NOT converge are exactly those whose
        # cost is running away below anything a well-posed seed reaches (admissible seeds end at J >= -40 from J_0 = +980,
        # asserted > -50; the diverged ones, asserted < -50, were measured at -59 ... -1440 when the iteration limit of 120
        # stops them and at -230 ... -2200 after 300 iterations, profiles/r03_b_arm_j0.txt) or whose parameters have
        # blown up: the reference's IPOPT would be iterating on an NLP without a minimiser as well
        J = L._sol["cost"].double().cpu().numpy()
        bad = ~np.isin(st, (1, 2))                     # every solve that did not converge (all of them excluded above) ...
        blown = ~np.isfinite(th).all(1) | (np.abs(th) >= 1e3).any(1)
        # (round 5 moved the bound between the two populations with the product, -50 -> -42 / -40: how far a diverging cost gets within
        #  the iteration limit depends on the path.  Round 6: no constant of the product's -- every solve that did not converge has a
        #  cost BELOW every admissible seed's of the same step (or blown-up parameters), and the admissible ones stay above round 4's -50)
        assert (blown[bad] | ~np.isfinite(J[bad]) | (J[bad] < J[adm].min())).all(), (k, J[bad], st[bad], J[adm].min())      # ... is a cost running away
        assert J[adm].min() > -50.0, (k, J[adm].min())




def test_full_size_properties_quadrotor_batch32768():
    """BASELINE configs[3]'s per-node batch (32768 random demonstrations) on one GPU: size-independent properties."""
    oc, d = gpu_model("quadrotor", torch.float32, 50, substeps=4)
    B = 32768
    rng = np.random.default_rng(3)
    x0 = np.tile(d["ini_state"], (B, 1))
    x0[:, 0:3] += 0.2 * rng.standard_normal((B, 3))
    goal = np.array([3.0, 3.0, 1.5])[None, :] + 0.2 * rng.standard_normal((B, 3))
    wps = np.array(d["waypoints"])[None, :, :] + 0.1 * rng.standard_normal((B, 5, 3))
    th = np.array(d["theta0"])[None, :] + 0.03 * rng.standard_normal((B, 7))
    for a in (x0, goal, wps, th):
        a[B // 2:] = a[:B // 2]                       # duplicated demonstrations must give identical results
    consts = oc.consts_tensor(batch=B, overrides=dict(goal_r0=goal[:, 0], goal_r1=goal[:, 1], goal_r2=goal[:, 2]))
    sol = oc.cocSolverBatch(x0, d["horizon"], th, consts=consts)
    aux = oc.auxSysSolverBatch(sol, np.tile(d["taus"], (B, 1)), wps, d["interface"])
    st = sol["status"].cpu().numpy()
    assert np.isin(st, (1, 2)).all(), np.bincount(st, minlength=5)
    assert torch.isfinite(aux["loss"]).all() and torch.isfinite(aux["grad"]).all()
    h = B // 2
    assert torch.equal(aux["loss"][:h], aux["loss"][h:]) and torch.equal(aux["grad"][:h], aux["grad"][h:])
    assert torch.equal(sol["state_grid"][:h], sol["state_grid"][h:])
    assert torch.allclose(sol["state_grid"][:, 0], torch.as_tensor(x0, dtype=torch.float32, device="cuda:0"))
    assert torch.equal(sol["control_grid"][:, -1], sol["control_grid"][:, -2])                     # CPDP.py:191
    Z = aux["Z_grid"]
    P = Z[:, :, :13, :]
    assert (P - P.transpose(2, 3)).abs().max() < 1e-3 * P.abs().max()
    # shared-theta gradient of the whole node batch == sum over the demonstrations (what the ranks all-reduce)
    sub = rng.choice(h, 32, replace=False)
    oc64, _ = gpu_model("quadrotor", torch.float64, 50, substeps=4)
    c64 = oc64.consts_tensor(batch=32, overrides=dict(goal_r0=goal[sub, 0], goal_r1=goal[sub, 1], goal_r2=goal[sub, 2]))
    s64 = oc64.cocSolverBatch(x0[sub], d["horizon"], th[sub], consts=c64)
    a64 = oc64.auxSysSolverBatch(s64, np.tile(d["taus"], (32, 1)), wps[sub], d["interface"])
    l32, l64 = aux["loss"][sub].double().cpu().numpy(), a64["loss"].cpu().numpy()
    g32, g64 = aux["grad"][sub].double().cpu().numpy(), a64["grad"].cpu().numpy()
    assert np.all(np.abs(l32 - l64) < 1e-3 * np.maximum(1.0, l64))
    assert np.all(np.abs(g32 - g64).max(axis=1) < 2e-2 * np.abs(g64).max(axis=1))


def test_full_size_properties_rocket_n100_mixed_precision():
    """BASELINE configs[4]: Rocket (6-DoF), horizon (n_grid) 100, fp32 solve + fp64 auxiliary Riccati / sensitivity pass,
    1024 seeds on one GPU (8192 over 8).  No solve may end at the iteration limit or fail; duplicates bit-identical;
    P symmetric; and one trajectory certified by the oracle as a KKT point of the reference's NLP (fp32 floor) with the
    fp64 auxiliary pass along it reproduced to fp64 tolerance."""
    oc, env, d = models.rocket(n_grid=100)
    oc.setDevice("cuda:0", torch.float32, aux_dtype=torch.float64)
    oc.setSolverOptions(aux_substeps=4)
    assert not oc.compile().is_emulator
    B = 1024
    rng = np.random.default_rng(0)
    th = np.array(d["theta0"])[None, :] * (1 + 0.05 * rng.standard_normal((B, 12)))
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    th[B // 2:] = th[:B // 2]
    x0 = np.tile(d["ini_state"], (B, 1))
    sol = oc.cocSolverBatch(x0, d["horizon"], th)
    st = sol["status"].cpu().numpy()
    assert np.isin(st, (1, 2)).all(), np.bincount(st, minlength=5)
    idx = [7, 20, 40, 67, 87]
    taus = np.linspace(0, d["horizon"], 101)[idx]
    X0 = sol["state_grid"][0].double().cpu().numpy()
    wps = [np.concatenate([X0[k, 0:3], X0[k, 6:10]]) + 0.05 for k in idx]
    aux = oc.auxSysSolverBatch(sol, taus, wps, d["interface"])
    assert aux["loss"].dtype == torch.float64
    assert torch.isfinite(aux["loss"]).all() and torch.isfinite(aux["grad"]).all()
    h = B // 2
    assert torch.equal(aux["loss"][:h], aux["loss"][h:]) and torch.equal(aux["grad"][:h], aux["grad"][h:])
    assert torch.equal(sol["control_grid"][:, -1], sol["control_grid"][:, -2])
    Z = aux["Z_grid"]
    P = Z[:, :, :13, :]
    assert (P - P.transpose(2, 3)).abs().max() < 1e-6 * P.abs().max()
    # EIGHT trajectories (round 4; one in round 3) certified by the oracle, fanned out over the host cores: each is a KKT point
    # of the reference's NLP at the fp32 floor, and the fp64 auxiliary pass along it is reproduced to fp64 tolerance
    picks = [5, 17, 101, 233, 310, 404, 467, 511]
    jobs = []
    for b in picks:
        X, U, Lm = (sol[k][b].double().cpu().numpy() for k in ("state_grid", "control_grid", "costate_grid"))
        jobs.append(dict(kind="rocket", n_grid=100, ini_state=d["ini_state"], horizon=d["horizon"], theta=list(th[b]),
                         taus=list(taus), wps=wps, iface=d["interface"], check=(X, U, Lm), tight=True))
    for b, r in zip(picks, oracle_parallel(jobs)):
        X, Lm = sol["state_grid"][b].double().cpu().numpy(), sol["costate_grid"][b].double().cpu().numpy()
        J = float(sol["cost"][b])
        assert r["defect"] < 1e-4 * np.abs(X).max() and r["gmax"] < 2e-4 * (1 + abs(J)) and r["lmax"] < 5e-3 * np.abs(Lm).max(), \
            (b, r["defect"], r["gmax"], r["lmax"], J)
        parity_record("rocket n_grid 100 mixed precision, trajectory %d" % b, "loss", abs(aux["loss"][b].item() - r["loss"]) / max(1.0, r["loss"]), 1e-5)
        parity_record("rocket n_grid 100 mixed precision, trajectory %d" % b, "grad", rel(aux["grad"][b], r["grad"]), 5e-3)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_control_bounds_vs_independent_bounded_solve(dtype):
    """Finite control bounds of setControlVariable (CPDP.py:33-46) on the GPU: control-limited backward sweep + clamped
    roll-out vs the oracle's L-BFGS-B solve of the same bounded NLP (basin-independent, both directions)."""
    pc.control_bounds(gpu_prepare, dtype)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_state_bounds_vs_independent_bounded_solve(dtype):
    """Finite state bounds of setStateVariable (CPDP.py:20-31, 140-147) on the GPU: augmented-Lagrangian loop around
    lfsd_coc_solve vs the oracle's SLSQP solve of the same bounded NLP."""
    pc.state_bounds(gpu_prepare, dtype)


def test_dudtheta_error_is_the_last_interval_times_one_gain():
    """du/dtheta(T) of auxSysSolver (CPDP.py:370-381) against the tight oracle under refinement, on the GPU: the whole
    discrepancy is the last interval's dx/dtheta error times one constant gain (parity_cases.dudtheta_refinement)."""
    pc.dudtheta_refinement(gpu_prepare)


@pytest.mark.parametrize("dtype,xtol,jtol", [(torch.float64, 2e-6, 1e-9), (torch.float32, 3e-3, 2e-5)])
def test_mesh_continuation_ends_at_the_same_kkt_point(dtype, xtol, jtol):
    """The lean OC kernels start a cold solve on a coarse mesh (one RK4 step per interval) and leave it WITH a step that is
    accepted although the cost may rise by up to 1e-3 |J| (cpdp_oc.h, oc_solve_kernel: `fine_step`, the two discretisations
    cannot be refereed by the Armijo test).  Direct check of that exit: the same 64 headline seeds solved by the product
    library and by a build without the coarse phase (-DLFSD_COARSE_START=0, compiled here if the tree does not carry it)
    reach the same KKT point of the NLP of CPDP.py:110-175 -- states, controls, costates and cost to the precision class of
    the arithmetic -- and both report convergence on the reference's discretisation."""
    from conftest import build_variant_library
    oc, env, d = models.quadrotor(n_grid=50)
    variant = build_variant_library(oc, "nocoarse", ["-DLFSD_COARSE_START=0"])
    rng = np.random.default_rng(1234)
    th = np.array(d["theta0"])[None, :] + 0.05 * rng.standard_normal((4096, 7))
    th[:, 0] = np.abs(th[:, 0]) + 0.5
    th = th[:64]
    x0 = np.tile(d["ini_state"], (64, 1))
    sols = []
    for lib in (None, variant):
        o2, _, _ = models.quadrotor(n_grid=50)
        if lib:
            o2.use_library(lib)
        o2 = gpu_prepare(o2, dtype)
        o2.setSolverOptions(mapping="lockstep")
        sols.append(o2.cocSolverBatch(x0, d["horizon"], th))
    a, b = sols
    assert set(a["status"].tolist()) <= {1, 2} and set(b["status"].tolist()) <= {1, 2}
    # the coarse phase is really taken by the product build: it changes the iteration path (not necessarily the count)
    for k, t in (("state_grid", xtol), ("control_grid", 5 * xtol), ("costate_grid", 10 * xtol)):
        parity_record("mesh continuation on/off %s" % dtype, k, rel(a[k], b[k].double().cpu().numpy()), t)
    ja, jb = a["cost"].double().cpu().numpy(), b["cost"].double().cpu().numpy()
    parity_record("mesh continuation on/off %s" % dtype, "cost", float(np.abs(ja - jb).max() / np.abs(jb).max()), jtol)


HELD_OUT = {
    # workloads NO schedule constant of csrc/cpdp_common.h was tuned on (the tuning workloads are bench.py's: the quad_example start /
    # goal at n_grid 50, its random demonstrations, the robot arm and the rocket of the examples at n_grid 50 / 100)
    "quadrotor_n20": dict(kind="quadrotor", n_grid=20, B=256, mapping="lockstep", horizon=1.4, spread=0.08,
                          ini=[-1.0, 0.5, 0.3, 0.2, 0, 0, 1, 0, 0, 0, 0, 0, 0], theta0=[1.5, 0.3, 0.2, 0.1, 0.2, 0.05, -0.5],
                          consts=dict(goal_r0=2.0, goal_r1=-2.0, goal_r2=2.0, Jx=0.5, Jy=0.8, Jz=1.2, mass=1.4)),
    "quadrotor_n30": dict(kind="quadrotor", n_grid=30, B=256, mapping="lockstep", horizon=0.8, spread=0.08,
                          ini=[0.5, 2.0, 1.5, 0, -0.3, 0, 1, 0, 0, 0, 0, 0, 0], theta0=[2.0, 0.2, 0.2, 0.3, 0.1, 0.1, 0.5],
                          consts=dict(goal_r0=-1.0, goal_r1=-1.0, goal_r2=0.5, Jx=1.5, Jy=1.5, Jz=0.7, mass=0.8)),
    "quadrotor_n100": dict(kind="quadrotor", n_grid=100, B=256, mapping="lockstep", horizon=1.0, spread=0.08,
                           ini=[1.0, -1.0, 1.0, 0, 0, 0.2, 1, 0, 0, 0, 0, 0, 0], theta0=[1.2, 0.15, 0.15, 0.15, 0.05, 0.05, -1.5],
                           consts=dict(goal_r0=4.0, goal_r1=1.0, goal_r2=2.5, Jx=1.0, Jy=0.6, Jz=1.4, mass=1.1)),
    "cartpole_n40": dict(kind="cartpole", n_grid=40, B=256, mapping="wide", horizon=1.0, spread=0.05, ini=[0, 0.3, 0, 0], theta0=None, consts={}),
    "rocket_other_landing_n40": dict(kind="rocket", n_grid=40, B=128, mapping="auto", horizon=3.0, spread=0.05,
                                     ini=[8.0, 5.0, -4.0, -0.2, 0.1, 0.0, 0.9238795, 0.0, 0.2705981, -0.2705981, 0, 0, 0], theta0=None, consts={}),
}


@pytest.mark.parametrize("name", sorted(HELD_OUT))
def test_schedule_constants_on_held_out_workloads(name):
    """The solver's schedules -- level 0 / coarse phase of the mesh continuation, the merged-interval levels of the wide kernel, the
    multiple-shooting steps -- were chosen by A/B on bench.py's own workloads.  Here: workloads never used for that (other starts,
    goals, horizons, inertias and masses, other grids; a cart-pole; a rocket with another landing approach), the PRODUCT build
    against a build with every schedule switched off (-DLFSD_LEAN_TC=1 -DLFSD_COARSE_START=0 -DLFSD_COARSE_TIME=1 -DLFSD_MS=0):
    both must end at KKT points (status 1 / 2) and at the SAME one -- asserted at the mesh-continuation test's fp32 tolerances on
    the quadrotor workloads; recorded where the problem has several minima (cart-pole, rocket: which one a cold start reaches
    depends on the path) --, and the time ratio of the two is RECORDED (LFSD_PARITY_REPORT's
    directory, held_out_schedule_ab.jsonl -> profiles/), not asserted: a workload where the tuned schedule is slower than none
    is named in DESIGN.md."""
    import json
    import time
    from conftest import build_variant_library, PLAIN_SCHEDULE
    w = HELD_OUT[name]
    oc0, env, d = models.ZOO[w["kind"]](n_grid=w["n_grid"])
    variant = build_variant_library(oc0, "plain", PLAIN_SCHEDULE)
    B = w["B"]
    p = len(d["theta0"])
    rng = np.random.default_rng(77)
    th0 = np.array(w["theta0"] if w["theta0"] is not None else d["theta0"], dtype=np.float64)
    th = th0[None, :] * (1 + w["spread"] * rng.standard_normal((B, p)))
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    x0 = np.tile(np.array(w["ini"], dtype=np.float64), (B, 1))
    out, ms = [], []
    for lib in (None, variant):
        o2, _, _ = models.ZOO[w["kind"]](n_grid=w["n_grid"])
        if lib:
            o2.use_library(lib)
        o2 = gpu_prepare(o2, torch.float32)
        if w["mapping"] != "auto":
            o2.setSolverOptions(mapping=w["mapping"])
        consts = o2.consts_tensor(overrides=w["consts"]) if w["consts"] else None
        sol = o2.cocSolverBatch(x0, w["horizon"], th, consts=consts)            # (first call: allocations, code object load)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            sol = o2.cocSolverBatch(x0, w["horizon"], th, consts=consts, workspace=sol["workspace"])
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / 3 * 1e3)
        out.append(sol)
    a, b = out
    sa, sb = a["status"].cpu().numpy(), b["status"].cpu().numpy()
    ok = np.isin(sa, (1, 2)) & np.isin(sb, (1, 2))
    ja, jb = a["cost"].double().cpu().numpy(), b["cost"].double().cpu().numpy()
    xa, xb = a["state_grid"].double().cpu().numpy(), b["state_grid"].double().cpu().numpy()
    xerr = np.abs(xa - xb).max(axis=(1, 2)) / np.maximum(np.abs(xb).max(axis=(1, 2)), 1e-300)
    jerr = np.abs(ja - jb) / np.maximum(np.abs(jb), 1e-300)
    same = ok & (xerr < 3e-3) & (jerr < 2e-5)
    differ = ok & ~same
    rec = dict(workload=name, batch=B, n_grid=w["n_grid"], ms_product=ms[0], ms_plain=ms[1], ratio_product_over_plain=ms[0] / ms[1],
               other_minimum=float(differ.mean()), product_cost_lower_where_other=float((ja[differ] < jb[differ]).mean()) if differ.any() else None,
               cost_rel_diff_mean_where_other=float(((ja[differ] - jb[differ]) / np.abs(jb[differ])).mean()) if differ.any() else None,
               iters_product=float(a["iters"].double().mean()), iters_plain=float(b["iters"].double().mean()),
               iters_max_product=int(a["iters"].max()), iters_max_plain=int(b["iters"].max()),
               status_product=np.bincount(sa, minlength=5).tolist(), status_plain=np.bincount(sb, minlength=5).tolist(),
               same_kkt_point=float(same.mean()), cost_rel_diff_max_where_same=float(jerr[same].max()) if same.any() else None)
    rep = os.environ.get("LFSD_PARITY_REPORT")
    if rep:
        with open(os.path.join(os.path.dirname(rep), "held_out_schedule_ab.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    print(rec)
    # Problems with several minima (cart-pole swing-up, rocket landing): which one a cold start reaches depends on the path, under
    # ANY two globalisations -- measured here on the rocket with another landing approach: 37 % of the
    # trajectories end in another KKT point than with every schedule off.  There both answers must be KKT points (status; the
    # oracle certifies the kernel's rocket answers in test_rocket_* / test_full_size_properties_rocket_*), the share of equal
    # answers and which side's cost is lower are RECORDED; on the single-minimum workloads equality is asserted.
    multi_minima = w["kind"] in ("cartpole", "rocket")
    assert ok.mean() >= (0.97 if multi_minima else 1.0), rec
    if not multi_minima:
        assert same.mean() == 1.0, rec


@pytest.mark.parametrize("cfg,words", [("robotarm", ("configs[1]", "RobotArm")), ("rocket", ("configs[4]", "Rocket"))])
def test_bench_config_lines_have_the_full_schema(cfg, words):
    """`bench.py --config robotarm|rocket` (BASELINE configs[1] / configs[4]) as a fresh child at a small batch: one JSON line
    with the headline's schema -- metric, value, roofline (with the kernel it names and its arithmetic), config.workload naming
    the configuration -- and a CPU leg from the oracle on the run's own seeds."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", cfg, "--batch", "64", "--steps", "2", "--warmup", "1",
                        "--cpu-seeds", "4"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    o = json.loads(lines[0])
    assert o["metric"].startswith("CPDP outer iterations/sec") and o["unit"] == "trajectory outer-iterations/s"
    assert o["n_gpus"] == 1 and o["steps"] == 2 and o["value"] > 0 and o["higher_is_better"] is True and o["vs_baseline"] is None
    assert all(wd in o["config"]["workload"] for wd in words) and o["config"]["name"] == cfg and o["config"]["batch_per_gpu"] == 64
    rf = o["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] in ("oc_solve", "aux_riccati", "aux_forward") and rf["achieved"] > 0 and 0 < rf["frac"] < 1
    assert rf["kernel_dtype"] in ("f32", "f64") and rf["valu_useful_tflops"] > 0
    cb = o["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] and cb["value"] > 0, cb
    assert (cfg == "rocket") == ("f64 (auxiliary pass)" in o["dtype"])


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_general_interface_function_vs_oracle(dtype):
    """An interface function that is an arbitrary expression of the state (lib/QuadAlgorithm.py:616-639), compiled into the model
    library: robot-arm end-effector position and a nonlinear pendulum observation against the oracle's general-interface loss."""
    pc.general_interface(gpu_prepare, dtype)


def test_fp64_solve_seeded_by_fp32_reaches_the_same_kkt_point():
    """lfsd_coc_solve in fp64 (lock-step mapping, quadrotor class) runs the fp32 lean kernel first and starts the fp64 kernel from
    its controls: same KKT point as the fp64 kernel from the cold start, a row the fp32 solve overflows on is started cold."""
    pc.seeded_f64_same_kkt_point(gpu_prepare, n_grid=50, batch=70)


@pytest.mark.parametrize("kind,n_grid", [("rocket", 100), ("quadrotor", 50)])
def test_wide_launch_schemes_bit_identical(kind, n_grid, monkeypatch):
    """The wide OC solve under every launch scheme of lfsd_capi.cpp's coc_solve_t, 1 024 seeds, fp32: one wavefront per trajectory;
    four per trajectory from the start; TWO launches -- the solver state parks in the workspace once all but one-CU-each
    trajectories are finished (a device counter, no host read) and the tail resumes with four wavefronts per trajectory -- handed
    over by the counter and at fixed iterations (3: inside the coarse phase, 10, 37).  Every output must be the same bits: an item of
    an interval-parallel phase is computed by the same code whoever runs it, so a trajectory's result may not depend on when it
    was handed over (nor on its partners: duplicated seeds included).  Rocket = BASELINE configs[4]'s per-GPU shard (Newton from the
    first iteration, mesh continuation with merged intervals); quadrotor on the wide mapping runs the multiple-shooting steps.
    Also the guard of profiles/r06_f_wide_stale_cost.txt: a build whose one-wavefront kernel kept a stale cost differed here."""
    oc, d = gpu_model(kind, torch.float32, n_grid, substeps=4)
    if kind == "quadrotor":
        oc.setSolverOptions(mapping="wide")
    B = 1024
    p = len(d["theta0"])
    rng = np.random.default_rng(0)
    th = np.array(d["theta0"])[None, :] * (1 + 0.05 * rng.standard_normal((B, p)))
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    th[B // 2:] = th[:B // 2]
    x0 = np.tile(d["ini_state"], (B, 1))
    keys = ("state_grid", "control_grid", "costate_grid", "cost", "iters", "status")
    ref = None
    for envs in (dict(LFSD_WIDE_WAVES="1"), dict(LFSD_WIDE_WAVES="4"), dict(), dict(LFSD_WIDE_SUSPEND_IT="3"),
                 dict(LFSD_WIDE_SUSPEND_IT="10"), dict(LFSD_WIDE_SUSPEND_IT="37")):
        for k in ("LFSD_WIDE_WAVES", "LFSD_WIDE_CAPACITY", "LFSD_WIDE_SUSPEND_IT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in envs.items():
            monkeypatch.setenv(k, v)
        sol = oc.cocSolverBatch(x0, d["horizon"], th)
        st = sol["status"].cpu().numpy()
        assert np.isin(st, (1, 2)).all(), (envs, np.bincount(st, minlength=5))
        if ref is None:
            ref = sol
            h = B // 2
            assert torch.equal(sol["state_grid"][:h], sol["state_grid"][h:]) and torch.equal(sol["iters"][:h], sol["iters"][h:])
        else:
            for key in keys:
                assert torch.equal(sol[key], ref[key]), (envs, key, int((sol[key] != ref[key]).sum()))


def test_the_libraries_this_box_loads_passed_the_assembly_check():
    """Every model library the tier can load on this box -- the ones that travelled with the tree and the ones built here -- went through
    runtime._checked_build: a record of ITS bytes (sha256) says both translation units are free of VGPR spills before an exec restore
    (lfsd_amd/isa_check.py; the cause of round 6's wrong builds of the wide kernel, profiles/r06_v_spill_before_exec_restore.txt).  And
    the rocket's wide fp32 solve -- the kernel two of those builds got wrong -- gives the SAME iterates with one and with four wavefronts
    per trajectory on cold starts that shorten steps on the coarse level (the wrong builds needed ~60 iterations to status 2 here)."""
    import glob
    from lfsd_amd import runtime
    for kind in models.ZOO:
        oc, env, d = models.ZOO[kind]()
        oc.setDevice("cuda:0", torch.float32)
        lib = oc.compile()
        path = runtime.library_path(oc.model_spec().hash())
        assert runtime.isa_record_clean(path), (kind, path)
    for path in glob.glob(os.path.join(runtime.BUILD_DIR, "liblfsd_*.so")):      # (+ the models other tests of the tier compiled on this box)
        assert runtime.isa_record_clean(path), path
    for kind, n_grid, tag, flags in runtime.GPU_TIER_VARIANTS:                      # the variants the tier's A/B tests load
        path = runtime.variant_library_path(models.ZOO[kind](n_grid=n_grid)[0].model_spec(), tag)
        assert not os.path.exists(path) or runtime.isa_record_clean(path), path
    oc, env, d = models.ZOO["rocket"](n_grid=100)
    oc.setDevice("cuda:0", torch.float32)
    rng = np.random.default_rng(0)
    B = 256
    th = np.array(d["theta0"])[None, :] * (1 + 0.05 * rng.standard_normal((B, len(d["theta0"]))))
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    x0 = np.tile(d["ini_state"], (B, 1))
    res = {}
    try:
        for waves in ("1", "4"):
            os.environ["LFSD_WIDE_WAVES"] = waves
            s = oc.cocSolverBatch(x0, d["horizon"], th)
            res[waves] = {k: s[k].clone() for k in ("state_grid", "control_grid", "cost", "iters", "status")}
    finally:
        os.environ.pop("LFSD_WIDE_WAVES", None)
    for k in res["1"]:
        assert torch.equal(res["1"][k], res["4"][k]), k
    st, it = res["1"]["status"].cpu().numpy(), res["1"]["iters"].cpu().numpy()
    assert (st == 1).mean() > 0.95 and it.mean() < 50, (np.bincount(st, minlength=5), it.mean())
