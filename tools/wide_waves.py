"""Wide OC solve with several wavefronts per trajectory (oc_solve_wide_kernel<..., W>; lfsd_capi.cpp, "two launches"): for every
outer iteration of bench.py's learner of one configuration, the SAME solve under every launch scheme -- one wavefront per
trajectory only (LFSD_WIDE_WAVES=1), four from the start (=4), two launches handed over by the device counter (default) and at
fixed iterations (LFSD_WIDE_SUSPEND_IT) -- timed by HIP events, outputs compared bit for bit with the one-wavefront solve.

    python tools/wide_waves.py <rocket|quadrotor> [steps] [batch]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import lfsd_amd  # noqa: F401
from lfsd_amd import models

cfg = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
argv = ["--config", cfg, "--no-cpu-baseline"] + (["--batch", sys.argv[3]] if len(sys.argv) > 3 else [])
args = bench.parse_args(argv)
w = bench.WORKLOADS[cfg]
TD = {"f32": torch.float32, "f64": torch.float64}
oc, env, d = models.ZOO[w["kind"]](n_grid=args.n_grid)
if cfg == "quadrotor":
    oc.setSolverOptions(mapping="wide")
oc.setDevice("cuda:0", TD[args.dtype], aux_dtype=TD[w["aux_dtype"]] if w["aux_dtype"] else None)
lib = oc.compile()
d = dict(d)
d["taus"], d["waypoints"] = bench.demonstration(oc, d, args.n_grid)
L, theta0, x0 = bench.build_learner(args, oc, d, lib, 0, 1, "independent", w)
L.count_unconverged = False

SCHEMES = [("one wavefront", dict(LFSD_WIDE_WAVES="1")), ("four from the start", dict(LFSD_WIDE_WAVES="4")),
           ("two launches (counter)", dict()), ("two launches, hand-over at it 10", dict(LFSD_WIDE_SUSPEND_IT="10")),
           ("two launches, hand-over at it 37", dict(LFSD_WIDE_SUSPEND_IT="37"))]
KEYS = ("state_grid", "control_grid", "costate_grid", "cost", "iters", "status")


def solve(envs, theta):
    for k in ("LFSD_WIDE_WAVES", "LFSD_WIDE_SUSPEND_IT", "LFSD_WIDE_CAPACITY"):
        os.environ.pop(k, None)
    os.environ.update(envs)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for rep in range(2):
        e0.record()
        sol = oc.cocSolverBatch(L.x0, L.hz, theta, consts=L.consts)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return sol, best


for k in range(steps):
    theta = L.theta.clone()
    ref = None
    line = []
    for name, envs in SCHEMES:
        sol, ms = solve(envs, theta)
        if ref is None:
            ref = {key: sol[key].clone() for key in KEYS}
            it = sol["iters"].cpu().numpy()
            line.append("iters p50 %d p75 %d p99 %d max %d" % (np.median(it), np.quantile(it, 0.75), np.quantile(it, 0.99), it.max()))
        same = all(torch.equal(sol[key], ref[key]) for key in KEYS)
        ndiff = int((sol["state_grid"] != ref["state_grid"]).flatten(1).any(1).sum())
        line.append("%s %.2f ms %s" % (name, ms, "identical" if same else "DIFFERENT (%d trajectories)" % ndiff))
    print("step %d: %s" % (k, " | ".join(line)), flush=True)
    for kk in ("LFSD_WIDE_WAVES", "LFSD_WIDE_SUSPEND_IT", "LFSD_WIDE_CAPACITY"):
        os.environ.pop(kk, None)
    L.step()
