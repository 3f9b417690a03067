"""Slowest OC solves ALONG the learner's path (not at the theta_0 seeds of tools/oc_trace.py): run bench.py's learner of a
configuration for `steps` outer iterations, save the parameters of the `n_slow` trajectories with the most iterations in the
last solve (gpurun_out/<tag>_theta.npz: they reproduce on the CPU emulator), and re-solve them one at a time with the trace
variant (tools/oc_trace.py build <model>).

    python tools/path_trace.py <robotarm|rocket|quadrotor> <steps> [n_slow] [tag]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import bench
import lfsd_amd  # noqa: F401
from lfsd_amd import models
import oc_trace

cfg, steps = sys.argv[1], int(sys.argv[2])
n_slow = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tag = sys.argv[4] if len(sys.argv) > 4 else "path_%s" % cfg
args = bench.parse_args(["--config", cfg, "--no-cpu-baseline"])
w = bench.WORKLOADS[cfg]
TD = {"f32": torch.float32, "f64": torch.float64}
oc, env, d = models.ZOO[w["kind"]](n_grid=args.n_grid)
oc.setDevice("cuda:0", TD[args.dtype], aux_dtype=TD[w["aux_dtype"]] if w["aux_dtype"] else None)
lib = oc.compile()
d = dict(d)
d["taus"], d["waypoints"] = bench.demonstration(oc, d, args.n_grid)
L, theta0, x0 = bench.build_learner(args, oc, d, lib, 0, 1, "independent", w)
L.count_unconverged = False
for k in range(steps):
    th_in = L.theta.detach().clone()
    el, kt, loss = bench.timed_steps(L, 1, 0, torch.cuda.synchronize, torch)
    it, st = L._sol["iters"].cpu().numpy(), L._sol["status"].cpu().numpy()
    print("step %d: oc %.2f ms, status %s, iterations mean %.1f p50 %d p90 %d p99 %d max %d" %
          (k, kt["oc_solve"], np.bincount(st, minlength=5).tolist(), it.mean(), np.median(it), np.quantile(it, .9), np.quantile(it, .99), it.max()), flush=True)
order = np.argsort(-it)
# the parameters the last solve actually used: a Nesterov learner solves at the look-ahead point (its own array); a vanilla learner
# hands the solver its parameter array itself, which the update that followed has already changed -- take the copy made before the step
th = (L._sol["auxvar"] if w["method"] == "Nesterov" else th_in).double().cpu().numpy()
x0n = np.asarray(x0.double().cpu().numpy() if hasattr(x0, "cpu") else x0)
if x0n.ndim == 1:
    x0n = np.tile(x0n, (len(th), 1))
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/%s_theta.npz" % tag, theta=th[order], x0=x0n[order], iters=it[order], status=st[order], horizon=float(d["horizon"]))      # slowest first
print("iteration histogram (bins of 20):", np.bincount(it // 20).tolist())
if it.max() < 40:      # the lean kernels: exact counts, and the slowest trajectory of every group of four (one wavefront)
    print("iterations:", np.bincount(it).tolist(), "| per wavefront of four, the maximum:", np.bincount(it[:len(it) // 4 * 4].reshape(-1, 4).max(1)).tolist())
    print("status by iterations:", {int(k): np.bincount(st[it == k], minlength=5).tolist() for k in np.unique(it)})
oc2, _, _ = models.ZOO[w["kind"]](n_grid=args.n_grid)
oc2.use_library(oc_trace.variant_path(oc2.model_spec(), "trace")); oc2.setDevice("cuda:0", TD[args.dtype])
oc2.setSolverOptions(mapping=oc.mapping if oc.mapping != "auto" else ("wide" if oc.exact_after == 0 else "lockstep"))
for j in order[:n_slow]:
    print("=== trajectory %d (%d iterations, status %d in the batch), theta %s" % (j, it[j], st[j], np.array2string(th[j], precision=5)), flush=True)
    s1 = oc2.cocSolverBatch(x0n[j:j + 1], d["horizon"], th[j:j + 1]); torch.cuda.synchronize()
    print("=== iterations %d status %d cost %.8g" % (int(s1["iters"][0]), int(s1["status"][0]), float(s1["cost"][0])), flush=True)
