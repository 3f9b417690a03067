#!/usr/bin/env python3
"""Headline benchmark: CPDP outer iterations/sec over a batch of trajectories (BASELINE.json metric).

One "step" = one complete outer iteration of ``SparseDemoLearner.step`` (the product's own iteration, bracketed with
HIP events through its ``event_hook``) for every trajectory of the batch (lib/QuadAlgorithm.py:469-486, Nesterov,
lr 0.01, mu 0.9):
    look-ahead point -> optimal-control solve (cold start, as the reference) -> differentiated PMP
    (Riccati + sensitivity sweeps) -> waypoint loss and d(theta) -> parameter update + projection.

Workloads (Quadrotor CPDP with time-warping, n_grid "horizon" 50, 4 RK4 steps per grid interval, 4096 per GPU):
  --mode independent (default at N=1; BASELINE configs[2]): 4096 random initial-guess seeds on the quad_example
      waypoints, every seed with its own theta and optimizer state.
  --mode shared (default at N>1; BASELINE configs[3]): ONE theta for all ranks, 4096 random demonstrations per GPU
      (start position, goal and waypoints drawn around the quad_example's); per outer iteration the summed d(theta)
      and loss are all-reduced over the ranks (RCCL over xGMI) and the all-reduced gradient drives the single update.
`value` = trajectories * K / wall seconds over all ranks (trajectory outer-iterations per second, whole job).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype f32|f64] [--mode independent|shared]
                    [--config quadrotor|robotarm|rocket]

`--config` selects another BASELINE configuration with the same JSON schema (WORKLOADS below): `robotarm` = configs[1]
(Examples/robotarm_random.py:14-73: n_grid 50, 1024 random seeds, fp32, plain gradient steps at lr 0.1), `rocket` =
configs[4] at one GPU's shard (Examples/rocket_groundtruth.py:14-111: n_grid 100, 8192 / 8 = 1024 seeds, fp32 solve +
fp64 auxiliary pass, ground-truth waypoints, lr 1e-3).  The default line (quadrotor, fp32, N = 1) also carries an
`"f64"` object: the same workload run for 5 steps at the reference's own precision after the timed region.

`--gpus N` with N > 1 and no torch.distributed.run environment: this process touches no GPU, starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, relays rank 0's JSON line and exits
with the child's code.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
VALU_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}
PROFILE_TAG = "r06"            # profiles/<tag>_hbm_traffic.json, <tag>_issue_counters.json feed the roofline object

# The workloads of BASELINE.json's configs that bench.py can emit a line for.  `batch` is per GPU.
WORKLOADS = {
    "quadrotor": dict(kind="quadrotor", n_grid=50, batch=4096, dtype="f32", aux_dtype=None, method="Nesterov", lr=1e-2,
                      spread=None, baseline="configs[2]", cpu_seeds=64, cpu_workers=64,
                      what="Quadrotor (JinEnv, initCost_Polynomial, beta time-warp) CPDP"),
    "robotarm": dict(kind="robotarm", n_grid=50, batch=1024, dtype="f32", aux_dtype=None, method="Vanilla", lr=1e-1,
                     spread=0.05, baseline="configs[1]", cpu_seeds=64, cpu_workers=64,
                     what="RobotArm 2-link (JinEnv, initCost_Polynomial, beta time-warp) CPDP, Examples/robotarm_random.py:14-73"),
    "rocket": dict(kind="rocket", n_grid=100, batch=1024, dtype="f32", aux_dtype="f64", method="Vanilla", lr=1e-3,
                   spread=0.05, baseline="configs[4] (one GPU's shard of 8192 / 8)", cpu_seeds=16, cpu_workers=16,
                   what="Rocket 6-DoF (JinEnv, initCost2, beta time-warp) CPDP, Examples/rocket_groundtruth.py:14-111, "
                        "fp32 OC solve + fp64 auxiliary (Riccati / sensitivity) pass, waypoints from the true parameters"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults = the window the round driver times: every BENCH_r0*.json records `--steps 20 --warmup 5`; until round 4 the defaults
    #  were 10 / 2, a cheaper window -- earlier outer iterations need fewer split units -- that the docs then called the driver's)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="quadrotor", choices=sorted(WORKLOADS),
                    help="BASELINE configuration: quadrotor = configs[2] / [3] (headline), robotarm = configs[1], rocket = configs[4]")
    ap.add_argument("--batch", type=int, default=None, help="trajectories per GPU (default: the configuration's)")
    ap.add_argument("--n-grid", type=int, default=None)
    ap.add_argument("--dtype", default=None, choices=["f32", "f64"], help="arithmetic of the OC solve (default: the configuration's)")
    ap.add_argument("--no-f64-leg", action="store_true", help="skip the 5 fp64 steps appended to the default fp32 headline line")
    ap.add_argument("--substeps", type=int, default=0,
                    help="minimum split units per grid interval of the auxiliary sweeps (0: library default = 1 with error control)")
    ap.add_argument("--aux-rtol", type=float, default=1e-3,
                    help="error-controlled sub-stepping of the auxiliary sweeps at the reference's own solve_ivp tolerance "
                         "(scipy default rtol 1e-3, CPDP.py:335,368); 0 = fixed --substeps (round 1: --substeps 4 --aux-rtol 0)")
    ap.add_argument("--mode", default=None, choices=["independent", "shared"],
                    help="default: independent seeds at N=1 (configs[2]), shared theta + gradient all-reduce at N>1 (configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seeds", type=int, default=0, help="seeds of the CPU baseline sample (default 64)")
    ap.add_argument("--library", default=None, help="tuning only: path of an alternative build of the model library")
    ap.add_argument("--warm-start", action="store_true",
                    help="NOT the headline: start each OC solve from the previous iteration's controls")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm)")
    args = ap.parse_args(argv)
    w = WORKLOADS[args.config]
    args.batch = args.batch or w["batch"]
    args.n_grid = args.n_grid or w["n_grid"]
    args.dtype = args.dtype or w["dtype"]
    return args


def spawn_ranks(args, argv):
    """N > 1 without a launcher: become the launcher.  Nothing in this process has touched a GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def algorithmic_bytes(kernel, n, m, p, nc, N, nw, ni, es):
    """Bytes one trajectory's launch MUST move (inputs read once + outputs written once), DESIGN.md section 4."""
    grids = (N + 1) * (2 * n + m)
    if kernel == "oc_solve":
        return es * (n + 1 + p + nc + grids + 1) + 8
    if kernel == "aux_riccati":
        return es * (1 + p + nc + grids + (N + 1) * n * (n + p))
    if kernel == "aux_forward":
        return es * (1 + p + nc + grids + (N + 1) * n * (n + p) + nw * (1 + ni) + 1 + p)
    raise KeyError(kernel)


# ---- CPU baseline: the oracle (port of the reference pipeline) on ALL host cores -----------------------------------
def cpu_worker(kind, n_grid):
    """`bench.py --cpu-worker KIND N_GRID`: build the oracle, say READY, read one JSON line of jobs, answer one JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import make_oracle
    from oracle.cpdp_oracle import getloss_corrections
    o = make_oracle(kind, n_grid)
    o.diffPMP()
    print("READY", flush=True)
    jobs = json.loads(sys.stdin.readline())
    out = []
    for ini, hz, taus, wps, iface, theta in jobs:
        tg, sol = o.cocSolver(ini, hz, theta)
        aux = o.auxSysSolver(tg, sol, theta)            # reference settings: BDF + RK45 at scipy defaults
        l, g = getloss_corrections(o, taus, wps, sol, aux, iface)
        out.append([float(l), [float(x) for x in g]])
    print(json.dumps(out), flush=True)


def cpu_baseline(kind, d, n_grid, thetas, n_sample, max_workers=64, timeout=240.0):
    """The oracle (fp64 numpy/scipy port of the reference pipeline; CasADi/IPOPT cannot be installed here) on a
    bounded sample of the benchmark's own seeds, one worker PROCESS per host core (plain child processes of this
    script: nothing is forked from the process that holds the GPU context).  Every worker has imported and lambdified
    its model before the clock starts; seeds are dealt round-robin; a worker that does not answer in time is killed."""
    import threading
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    # 64 seeds on at most 64 workers: one seed costs 2.5-6 s of one core, so the leg stays inside ~10-30 s on any host
    # (on the 256-core GPU box 256 workers x 2 seeds took 154 s: the processes fight over memory bandwidth)
    n_sample = min(len(thetas), n_sample if n_sample > 0 else 64)
    workers = min(cores, n_sample, max_workers)
    jobs = [(list(map(float, d["ini_state"])), float(d["horizon"]), list(map(float, d["taus"])),
             [list(map(float, w)) for w in d["waypoints"]], list(map(int, d["interface"])),
             [float(x) for x in thetas[b]]) for b in range(n_sample)]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")   # the processes are the parallelism
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", kind, str(n_grid)], env=env,
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
             for _ in range(workers)]
    ready, go = [threading.Event() for _ in procs], threading.Event()
    answers = [None] * workers

    def serve(w):
        if procs[w].stdout.readline().strip() != "READY":
            return
        ready[w].set()
        go.wait()
        procs[w].stdin.write(json.dumps(jobs[w::workers]) + "\n")
        procs[w].stdin.flush()
        line = procs[w].stdout.readline()
        answers[w] = json.loads(line) if line.strip() else None
    threads = [threading.Thread(target=serve, args=(w,), daemon=True) for w in range(workers)]
    try:
        for t in threads:
            t.start()
        deadline = time.time() + timeout
        for e in ready:
            if not e.wait(max(0.0, deadline - time.time())):
                raise RuntimeError("a CPU-baseline worker did not start within %.0f s" % timeout)
        t0 = time.time()
        go.set()
        deadline = t0 + timeout
        for t in threads:
            t.join(max(0.0, deadline - time.time()))
        dt = time.time() - t0
        if any(a is None for a in answers):
            raise RuntimeError("a CPU-baseline worker did not answer within %.0f s" % timeout)
    finally:
        go.set()
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    results = [None] * n_sample
    for w in range(workers):
        results[w::workers] = [(a[0], a[1]) for a in answers[w]]
    return dict(value=n_sample / dt, unit="trajectory outer-iterations/s", cores=workers, kind="port",
                sample="%d of the %d seeds of rank 0, 1 outer iteration each, oracle/cpdp_oracle.py (numpy/scipy fp64, "
                       "solve_ivp BDF+RK45 as CPDP.py:335,368) on %d single-threaded worker processes (host has %d "
                       "cores), %.1f s; the reference itself needs CasADi 3.5.5 + IPOPT 3.11.9, which are not installed "
                       "and cannot be (no network)" % (n_sample, len(thetas), workers, cores, dt)), results


def demo_set(args, d, rank, mode, w=None, p=None):
    """The problem set rank `rank` draws (weak scaling: every rank its own `--batch` trajectories, seeded by the rank)."""
    import numpy as np
    w = w or WORKLOADS[args.config]
    p = p or len(d["theta0"])
    B = args.batch
    rng = np.random.default_rng(1234 + rank)
    x0 = np.tile(d["ini_state"], (B, 1))
    if mode == "independent":
        if w["spread"] is None:       # quadrotor (configs[2]): additive spread around the example's initial guess
            theta0 = np.array(d["theta0"])[None, :] + 0.05 * rng.standard_normal((B, len(d["theta0"])))
            theta0[:, 0] = np.abs(theta0[:, 0]) + 0.5
        else:                         # robot arm / rocket: random seeds = the example's initial guess, 5 % relative spread
            theta0 = np.array(d["theta0"])[None, :] * (1 + w["spread"] * rng.standard_normal((B, p)))
            theta0[:, 0] = np.abs(theta0[:, 0]) + 0.1
        return dict(x0=x0, theta0=theta0)
    # shared theta, random demonstrations: start position, goal and waypoints perturbed per trajectory
    x0[:, 0:3] += 0.2 * rng.standard_normal((B, 3))
    goal = np.array([3.0, 3.0, 1.5])[None, :] + 0.2 * rng.standard_normal((B, 3))
    wps = np.array(d["waypoints"])[None, :, :] + 0.1 * rng.standard_normal((B, len(d["waypoints"]), 3))
    return dict(x0=x0, goal=goal, wps=wps, theta0=np.array(d["theta0"], dtype=np.float64))


def shared_learner(oc, d, demos, n_total, pg=None, warm_start=False):
    """SparseDemoLearner(mode='shared') over `demos` (one rank's demo_set, or the concatenation of several ranks');
    n_total = demonstrations over ALL ranks: lr 1e-2 is the example's rate for ONE demonstration, the summed gradient
    is scaled back by it."""
    import numpy as np
    from lfsd_amd import CPDP
    x0, goal, wps = demos["x0"], demos["goal"], demos["wps"]
    B = x0.shape[0]
    consts = oc.consts_tensor(batch=B, overrides=dict(goal_r0=goal[:, 0], goal_r1=goal[:, 1], goal_r2=goal[:, 2]))
    return CPDP.SparseDemoLearner(oc, x0, d["horizon"], np.tile(d["taus"], (B, 1)), wps, d["interface"], demos["theta0"],
                                  method="Nesterov", learning_rate=1e-2 / n_total, mu=0.9, consts=consts,
                                  mode="shared", process_group=pg, warm_start=warm_start)


def demonstration(oc, d, n_grid):
    """(taus, waypoints) of the workload.  Examples with fixed waypoints carry them; the ground-truth examples
    (rocket_groundtruth.py:77-86) take them from the optimal trajectory at the true parameters, at the same fractions of
    the horizon as the example's grid indices [1, 3, 6, 10, 13] of 15."""
    import numpy as np
    if "taus" in d:
        return list(d["taus"]), [list(w) for w in d["waypoints"]]
    sol = oc.cocSolverBatch([d["ini_state"]], d["horizon"], [d["true_theta"]])
    assert int(sol["status"][0]) in (1, 2), "the ground-truth solve of the demonstration did not converge"
    tg = np.linspace(0, d["horizon"], n_grid + 1)
    idx = sorted(set(int(round(f * n_grid)) for f in (1 / 15, 3 / 15, 6 / 15, 10 / 15, 13 / 15)))
    wps = sol["state_grid"][0, idx][:, list(d["interface"])].double().cpu().numpy()
    return tg[idx].tolist(), wps.tolist()


def build_learner(args, oc, d, lib, rank, world, mode, w=None, pg=None):
    """The benchmark's learner for rank `rank`."""
    from lfsd_amd import CPDP
    w = w or WORKLOADS[args.config]
    demos = demo_set(args, d, rank, mode, w, lib.n_auxvar)
    if mode == "independent":
        L = CPDP.SparseDemoLearner(oc, demos["x0"], d["horizon"], d["taus"], d["waypoints"], d["interface"], demos["theta0"],
                                   method=w["method"], learning_rate=w["lr"], mu=0.9, warm_start=args.warm_start)
        return L, demos["theta0"], demos["x0"]
    L = shared_learner(oc, d, demos, args.batch * world, pg=pg, warm_start=args.warm_start)
    return L, demos["theta0"][None, :], demos["x0"]


def timed_steps(L, steps, warmup, sync, torch):
    """`warmup` untimed steps, then exactly `steps` steps between two synchronisations, every kernel of a step bracketed
    by HIP events on the stream it is launched on (torch's current stream).  -> (seconds, {kernel: ms per step}, last loss)"""
    import numpy as np
    names = ("oc_solve", "aux_riccati", "aux_forward", "update")
    ev = {k: [] for k in names}
    cur = {}

    def hook(name):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if cur.get("prev"):
            ev[cur["prev"][0]].append((cur["prev"][1], e))
        cur["prev"] = None if name == "end" else (name, e)
    for _ in range(warmup):
        L.step()
    sync()
    L.event_hook = hook
    t0 = time.perf_counter()
    loss = None
    for _ in range(steps):
        loss, _g = L.step()
    sync()
    elapsed = time.perf_counter() - t0
    L.event_hook = None
    ktime = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}    # ms per step
    return elapsed, ktime, loss


def lean_level0(args):
    """(iterations, merged intervals, RK4 steps per merged interval) of level 0 of the lean kernels' mesh continuation (csrc/
    cpdp_common.h LFSD_LEAN_TC_ITERS / LFSD_LEAN_TC / LFSD_LEAN_TC_S and the rule of cpdp_oc.h), None where it does not apply.
    With it the solve's 1 + 6.3 roll-outs are: 1 + 3 on level 0, the transfer on the one-step-per-interval level, the rest (2.3, as
    before level 0 existed) on the reference's grid -- the mean iteration count did not grow by the transfer iteration."""
    return (3, 2, 1) if (args.n_grid % 2 == 0 and args.n_grid // 2 >= 10) else None


def seeded_f64(args, dtype_name, kernel, warm):
    """lfsd_coc_solve in fp64 on the lock-step mapping of the quadrotor class solves the problem in fp32 first and starts the
    fp64 kernel from those controls (csrc/lfsd_capi.cpp, coc_solve_seeded): `oc_solve` is then two kernels in two precisions.
    Since round 5 that holds with an initial guess from the caller as well (an all-zero row is a cold start for both kernels), i.e.
    for `--mode shared` (whose learner hands zeros for every row it does not continue) and for `--warm-start`."""
    return (kernel == "oc_solve" and dtype_name == "f64" and args.config == "quadrotor"
            and os.environ.get("LFSD_F64_SEED", "1") != "0")


def kernel_model(perf_model, spec, args, dtype_name, aux_dtype_name, kernel, ktime, it_mean, units, B, warm, it_seed=None):
    """(useful vector TFLOP/s, useful matrix-core TFLOP/s, useful flops per launch, dtype, fraction of the vector roof) of
    `kernel` by the operation-count model.  The fraction is (time the useful work takes at the vector peak of its arithmetic) /
    (measured time); for the fp32-seeded fp64 solve the fp32 part (`it_seed` iterations, 5 of its roll-outs on the coarse grid)
    is priced at the fp32 peak and the fp64 part (the remaining iterations on the reference's grid) at the fp64 peak."""
    sub = max(1, args.substeps) if args.aux_rtol > 0 else (args.substeps or 4)
    kd = dtype_name if kernel == "oc_solve" else (aux_dtype_name or dtype_name)
    t = ktime[kernel] * 1e-3
    if seeded_f64(args, kd, kernel, warm) and it_seed:
        f32, m32 = perf_model.kernel_flops(spec, kernel, args.n_grid, 4, sub, mean_iters=it_seed, split=True, midpoint=True,
                                           coarse_rollouts=1 if lean_level0(args) else 5, level0=lean_level0(args))
        f64, m64 = perf_model.kernel_flops(spec, kernel, args.n_grid, 4, sub, mean_iters=max(it_mean - it_seed, 0.0), split=True,
                                           midpoint=False, coarse_rollouts=0)
        f64 += m64
        flops = f32 + m32 + f64
        frac = ((f32 + m32) / VALU_PEAK_TFLOPS["f32"] + f64 / VALU_PEAK_TFLOPS["f64"]) * B / 1e12 / t
        return flops * B / t / 1e12, 0.0, flops * B, kd, frac
    # mesh continuation: 5 of the roll-outs of a cold lean fp32 / fp64 solve run on the coarse grid; none when
    # the solve is warm-started or the model has no coarse phase
    cold_lean = not (warm or kernel != "oc_solve" or args.config != "quadrotor")
    lvl0 = lean_level0(args) if cold_lean else None
    coarse = 0 if not cold_lean else (1 if lvl0 else 5)
    flops, mflops = perf_model.kernel_flops(spec, kernel, args.n_grid, 4, sub, mean_iters=it_mean,
                                            units_per_interval=units.get(kernel), split=True, midpoint=(kd == "f32"),
                                            coarse_rollouts=coarse, level0=lvl0)
    if kd == "f64" or args.config != "quadrotor":      # matrix cores only in the lean fp32 kernel of the 13-state models' lock-step mapping
        flops, mflops = flops + mflops, 0.0
    return flops * B / t / 1e12, mflops * B / t / 1e12, flops * B, kd, flops * B / t / 1e12 / VALU_PEAK_TFLOPS[kd]


def seed_iterations(models, torch, w, args, L, dev):
    """mean iterations of the fp32 solve that seeds a cold fp64 solve at the learner's current parameters (outside the timed
    region; only feeds the operation-count model of the two-precision `oc_solve`); None if it cannot be had"""
    try:
        oc32, _, _ = models.ZOO[w["kind"]](n_grid=args.n_grid)
        oc32.setDevice(dev, torch.float32)
        th = L.theta.detach().float()
        consts = None if L.consts is None else L.consts.float()
        s = oc32.cocSolverBatch(L.x0.float(), L.hz, th, consts=consts)
        return float(s["iters"].double().mean().item())
    except Exception:
        return None


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) == 3 and argv[0] == "--cpu-worker":
        return cpu_worker(argv[1], int(argv[2]))
    args = parse_args(argv)
    w = WORKLOADS[args.config]
    in_launcher = "RANK" in os.environ and "MASTER_PORT" in os.environ      # started by torch.distributed.run
    if args.gpus > 1 and not in_launcher:
        spawn_ranks(args, argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if in_launcher else 1
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import numpy as np
    import torch
    import torch.distributed as dist
    use_dist = in_launcher and world > 1
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)                         # one process per GPU; the device is bound before the communicator exists
    if in_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)       # "nccl" == RCCL on ROCm

    import lfsd_amd  # noqa: F401
    from lfsd_amd import models, perf_model
    TD = {"f32": torch.float32, "f64": torch.float64}
    dtype = TD[args.dtype]
    aux_name = w["aux_dtype"] if args.dtype == w["dtype"] else None      # (an explicit --dtype f64 runs everything in fp64)
    mode = args.mode or ("shared" if world > 1 else "independent")
    if mode == "shared" and args.config != "quadrotor":
        print("bench.py: --mode shared is the quadrotor's configs[3] workload", file=sys.stderr)
        sys.exit(2)
    oc, env, d = models.ZOO[w["kind"]](n_grid=args.n_grid)
    if args.library:
        oc.use_library(args.library)
    oc.setDevice(dev, dtype, aux_dtype=TD[aux_name] if aux_name else None)
    oc.setSolverOptions(aux_substeps=args.substeps, aux_rtol=args.aux_rtol)
    if use_dist and rank != 0:
        dist.barrier()                                 # rank 0 makes sure the model library exists (it is normally prebuilt)
    lib = oc.compile()
    if use_dist and rank == 0:
        dist.barrier()
    assert not lib.is_emulator
    B = args.batch
    d = dict(d)
    d["taus"], d["waypoints"] = demonstration(oc, d, args.n_grid)
    L, theta0, x0 = build_learner(args, oc, d, lib, rank, world, mode, w)
    L.count_unconverged = False                       # no device->host read inside the timed loop

    # HIP results of the first seeds at theta_0, kept for the cross-check against the CPU baseline's oracle results
    n_chk = 4
    chk_loss = chk_grad = None
    if mode == "independent" and rank == 0:
        sol_c = oc.cocSolverBatch(x0[:n_chk], d["horizon"], theta0[:n_chk])
        aux_c = oc.auxSysSolverBatch(sol_c, d["taus"], d["waypoints"], d["interface"])
        chk_loss, chk_grad = aux_c["loss"].double().cpu().numpy(), aux_c["grad"].double().cpu().numpy()

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    elapsed, ktime, loss = timed_steps(L, args.steps, args.warmup, sync, torch)
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    st = L._sol["status"].cpu().numpy()
    it = L._sol["iters"].cpu().numpy()
    if rank == 0:
        n, m, p, nc = lib.n_state, lib.n_control, lib.n_auxvar, lib.n_const
        dom = max(("oc_solve", "aux_riccati", "aux_forward"), key=lambda k: ktime[k])
        nw, ni = L.taus.shape[1], len(d["interface"])
        stats = L._aux["stats"].double().cpu().numpy()
        units = {"aux_riccati": float(stats[:, 0].mean()) / args.n_grid, "aux_forward": float(stats[:, 2].mean()) / args.n_grid}
        it_seed = seed_iterations(models, torch, w, args, L, dev) if seeded_f64(args, args.dtype, dom, args.warm_start) else None
        useful_tflops, mfma_tflops, useful_flops, dom_dtype, valu_frac = kernel_model(perf_model, oc.model_spec(), args, args.dtype, aux_name, dom,
                                                                                      ktime, float(it.mean()), units, B, args.warm_start, it_seed)
        es = 4 if dom_dtype == "f32" else 8
        abytes = B * algorithmic_bytes(dom, n, m, p, nc, args.n_grid, nw, ni, es)
        achieved = abytes / (ktime[dom] * 1e-3) / 1e9
        # PMC passes are separate rocprofv3 runs (tools/hbm_traffic.py, tools/issue_counters.py); the figures only apply
        # to the workload they were collected on, so they are attached to the configuration's default command and null otherwise
        default_cmd = (B == w["batch"] and args.n_grid == w["n_grid"] and args.dtype == w["dtype"] and args.substeps == 0
                       and args.aux_rtol == 1e-3 and mode == "independent" and not args.warm_start and not args.library)
        tag = PROFILE_TAG if args.config == "quadrotor" else "%s_%s" % (PROFILE_TAG, args.config)
        sources = {}

        def profile(name, key=None):
            """Counter figures that THIS run does not measure: read from a committed fold of separate rocprofv3 --pmc passes
            of the same command, and named as such in the line (file + git blob hash of the file read, and the commit the
            fold was collected at when the fold records it)."""
            import hashlib
            rel = "profiles/%s_%s.json" % (tag, name)
            path = os.path.join(ROOT, rel)
            if not (default_cmd and os.path.exists(path)):
                return {}
            try:
                raw = open(path, "rb").read()
                js = json.loads(raw)
                sources[name] = {"file": rel, "git_blob": hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest(),
                                 "collected_at_commit": js.get("collected_at_commit")}
                if key:
                    return js.get(key) or {}
                return js.get(dom) or (js.get("oc_solve_wide", {}) if dom == "oc_solve" else {})      # (models solved on the wide mapping)
            except Exception:
                return {}
        traffic = profile("hbm_traffic").get("hbm_bytes_per_launch")
        if traffic is not None and it_seed is not None:      # the fp32-seeded fp64 solve: both kernels of `oc_solve` move bytes
            traffic += profile("hbm_traffic", "oc_solve_seed_f32").get("hbm_bytes_per_launch") or 0.0
        issue = profile("issue_counters")
        executed = issue.get("valu_flops_executed_per_launch")
        if dom == "oc_solve":      # a two-launch wide solve (round 6): `oc_solve` is both launches, the folds keep the four-wavefront one apart
            w4 = profile("hbm_traffic", "oc_solve_wide_w4").get("hbm_bytes_per_launch")
            if traffic is not None and w4:
                traffic += w4
            w4 = profile("issue_counters", "oc_solve_wide_w4").get("valu_flops_executed_per_launch")
            if executed is not None and w4:
                executed += w4
        peak = VALU_PEAK_TFLOPS[dom_dtype]
        out = {
            "metric": "CPDP outer iterations/sec (batch trajectories)",
            "value": B * world * args.steps / elapsed,
            "unit": "trajectory outer-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype if not aux_name else "%s (OC solve) + %s (auxiliary pass)" % (args.dtype, aux_name), "data": "synthetic",
            "config": {"workload": ("%s, n_grid(horizon) %d x 4 RK4 steps, batch %d per GPU, %s%s, %s OC solve every iteration; " %
                                    (w["what"], args.n_grid, B, w["method"] if mode == "independent" else "Nesterov",
                                     " mu 0.9" if (mode == "shared" or w["method"] == "Nesterov") else "",
                                     "WARM-started (not the headline configuration)" if args.warm_start else "cold-start")) +
                                   ("BASELINE %s: random initial-guess seeds, one theta and optimizer state per seed, lr %g" %
                                    (w["baseline"], w["lr"]) if mode == "independent" else
                                    "BASELINE configs[3] at 4096 per GPU: random demonstrations (start, goal, waypoints), ONE "
                                    "shared theta, summed d(theta)+loss all-reduced over the ranks every iteration and "
                                    "driving the update (SparseDemoLearner mode='shared')"),
                       "name": args.config, "mode": mode, "batch_per_gpu": B, "n_grid": args.n_grid, "steps_per_grid": 4,
                       # backend of the initialised torch.distributed group the shared-mode all-reduce went through (None: no group)
                       "process_group": (str(dist.get_backend()) if dist.is_initialized() else None),
                       "aux_substeps": args.substeps, "aux_rtol": args.aux_rtol,
                       "aux_integration": ("error-controlled split-step + Richardson sweeps, rtol %g on the un-extrapolated "
                                           "estimate (the reference integrates the same ODEs with solve_ivp at rtol 1e-3), from %d unit(s) per interval" %
                                           (args.aux_rtol, max(1, args.substeps))) if args.aux_rtol > 0 else
                                          ("fixed %d units per interval" % (args.substeps or 4)),
                       "oc_status_hist": np.bincount(st, minlength=5).tolist(), "oc_iters_mean": float(it.mean()),
                       "oc_iters_max": int(it.max()), "loss_mean": float(torch.nanmean(loss.double()).item()) / (1 if mode == "independent" else B * world),
                       "kernel_ms": {k: round(v, 3) for k, v in ktime.items()},
                       "aux_units_per_interval": {k: round(v, 3) for k, v in units.items()},
                       "aux_intervals_accepted_above_rtol": int(stats[:, 1].sum() + stats[:, 3].sum()),
                       # shared mode: the one parameter vector every rank holds after warmup + steps iterations (17 digits: the
                       # N>1 test compares it with a single-process run over the union of the ranks' demonstrations)
                       "theta": [float(x) for x in L.theta.double().cpu().numpy().ravel()] if mode == "shared" else None,
                       "n_unconverged_last_step": (int(round(float(L.n_bad_device.item()))) if (mode == "shared" and L.n_bad_device is not None) else None)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": ktime[dom],
                         # the fraction that actually bounds this path: vector issue, not HBM
                         "valu_useful_tflops": useful_tflops,
                         "valu_frac": valu_frac, "valu_peak_tflops": peak, "kernel_dtype": dom_dtype,
                         "seed_iters_mean_f32": it_seed,
                         # what the vector pipe EXECUTED per launch by the counters (every enabled lane, redundant group-uniform
                         # work included) at this run's launch time, and the modelled useful share of it
                         "valu_executed_tflops": None if executed is None else executed / (ktime[dom] * 1e-3) / 1e12,
                         "valu_useful_over_executed": None if executed is None else useful_flops / executed,
                         "mfma_useful_tflops": mfma_tflops,
                         "valu_issue_util": issue.get("valu_issue_util"),
                         "valu_lane_util": issue.get("valu_lane_util"),
                         "mfma_busy": issue.get("mfma_busy"),
                         "source": sources or None,
                         "note": "the per-trajectory recursions are latency/VALU-issue bound, not HBM bound: "
                                 "algorithmic bytes are O(10 KB) per trajectory against O(10^7) FLOP of sequential "
                                 "vector work per solve; valu_frac = useful VECTOR flops (perf_model.py, checked against the "
                                 "counters in round 3) / the vector peak of the kernel's arithmetic (157.3 fp32, 78.6 fp64 TFLOP/s); "
                                 "traffic, valu_issue_util, valu_lane_util (EXEC-enabled lanes per vector instruction) and "
                                 "mfma_busy are NOT measured by this run: they come from the committed fold of separate rocprofv3 "
                                 "--pmc passes of this command named in `source` (null when the run is not the configuration's "
                                 "default command or no fold is committed for it); see DESIGN.md section 4"},
        }
        # the reference computes in fp64 throughout (CPDP.py:183, :329): the same workload at that precision, 5 steps after the
        # timed region of the default fp32 headline line -- a reported figure beside `value`, which stays the fp32 rate
        if (args.config == "quadrotor" and default_cmd and world == 1 and not args.no_f64_leg):
            try:
                oc.setDevice(dev, torch.float64)
                L64, _, _ = build_learner(args, oc, d, lib, rank, world, mode, w)
                L64.count_unconverged = False
                e64, k64, _ = timed_steps(L64, 5, 1, sync, torch)
                st64 = L64._aux["stats"].double().cpu().numpy()
                u64 = {"aux_riccati": float(st64[:, 0].mean()) / args.n_grid, "aux_forward": float(st64[:, 2].mean()) / args.n_grid}
                d64 = max(("oc_solve", "aux_riccati", "aux_forward"), key=lambda k: k64[k])
                it64 = float(L64._sol["iters"].double().mean().item())
                a64 = argparse.Namespace(**dict(vars(args), dtype="f64"))
                seed64 = float(it.mean()) if seeded_f64(a64, "f64", d64, False) else None      # (the fp32 leg's own solves: same seeds, same rule)
                uf64, _, _, _, frac64 = kernel_model(perf_model, oc.model_spec(), a64, "f64", None, d64, k64, it64, u64, B, False, seed64)
                out["f64"] = {"value": B * 5 / e64, "unit": "trajectory outer-iterations/s", "steps": 5, "warmup": 1,
                              "ms_per_step": e64 / 5 * 1e3, "kernel_ms": {k: round(v, 3) for k, v in k64.items()},
                              "aux_units_per_interval": {k: round(v, 3) for k, v in u64.items()},
                              "oc_status_hist": np.bincount(L64._sol["status"].cpu().numpy(), minlength=5).tolist(),
                              "kernel": d64, "valu_useful_tflops": uf64, "valu_frac": frac64,
                              "oc_iters_mean": it64, "oc_seed_iters_mean_f32": seed64,
                              "note": "same seeds, every returned number and every convergence test in fp64 (the reference's precision); "
                                      "the cold OC solve is seeded by the fp32 solve of the same problem (lfsd_capi.cpp, coc_solve_seeded; "
                                      "oc_iters_mean counts both), so `oc_solve` is an fp32 and an fp64 kernel and valu_frac prices each "
                                      "part at the vector peak of its arithmetic; model-only utilisation"}
                del L64
            except Exception as exc:
                out["f64"] = {"value": None, "note": "failed: %r" % (exc,)}
            oc.setDevice(dev, dtype)
        if world == 1 and mode == "independent" and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"], ores = cpu_baseline(w["kind"], d, args.n_grid, theta0, args.cpu_seeds or w["cpu_seeds"],
                                                         max_workers=w["cpu_workers"])
            except Exception as exc:          # the GPU measurement must not be lost to a host-side problem
                out["cpu_baseline"], ores = dict(value=None, unit="trajectory outer-iterations/s", cores=0, kind="port",
                                                 sample="failed: %r" % (exc,)), []
            # parity of the HIP path with the oracle on the benchmark's own seeds (oracle in reference mode: its
            # solve_ivp tolerance 1e-3 limits the agreement of the gradient to ~5e-3)
            k = min(n_chk, len(ores))
            out["parity_vs_oracle"] = None if k == 0 else {
                "seeds": k,
                "loss_rel_err_max": max(abs(chk_loss[i] - ores[i][0]) / abs(ores[i][0]) for i in range(k)),
                "grad_rel_err_max": max(float(np.abs(chk_grad[i] - np.array(ores[i][1])).max() / np.abs(ores[i][1]).max())
                                        for i in range(k)),
                "note": None if args.config != "rocket" else
                        "the rocket's cold starts end in DIFFERENT stationary points under different globalisations (DESIGN.md "
                        "section 8): this compares the kernel's KKT point with the oracle's own, not a parity figure; the test tier "
                        "certifies the kernel's answer with the oracle instead (tests/test_gpu_parity.py, 8 trajectories at n_grid 100)"}
        print(json.dumps(out), flush=True)
    if in_launcher:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
