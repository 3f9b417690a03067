"""Per outer iteration of bench.py's learner for one configuration: kernel times (HIP events), OC status / iteration
histogram, and the distribution over the batch of the split units each auxiliary sweep spent (lfsd_aux_solve `stats`).
Shows whether a slow launch is the whole batch getting stiffer or a few trajectories holding it.

    python tools/config_steps.py <robotarm|rocket|quadrotor> [steps]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import lfsd_amd  # noqa: F401
from lfsd_amd import models

cfg = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
args = bench.parse_args(["--config", cfg, "--no-cpu-baseline"])
w = bench.WORKLOADS[cfg]
TD = {"f32": torch.float32, "f64": torch.float64}
oc, env, d = models.ZOO[w["kind"]](n_grid=args.n_grid)
if os.environ.get("LFSD_TOOL_LIBRARY"):      # a build variant (tools/model_ab.py build <model> <tag> ...): csrc/build/trace_<hash>_<tag>.so
    oc.use_library(os.environ["LFSD_TOOL_LIBRARY"])
oc.setDevice("cuda:0", TD[args.dtype], aux_dtype=TD[w["aux_dtype"]] if w["aux_dtype"] else None)
lib = oc.compile()
d = dict(d)
d["taus"], d["waypoints"] = bench.demonstration(oc, d, args.n_grid)
L, theta0, x0 = bench.build_learner(args, oc, d, lib, 0, 1, "independent", w)
L.count_unconverged = False
q = lambda a: "mean %.1f p50 %d p99 %d max %d" % (a.mean(), np.median(a), np.quantile(a, 0.99), a.max())
for k in range(steps):
    el, kt, loss = bench.timed_steps(L, 1, 0, torch.cuda.synchronize, torch)
    st = L._aux["stats"].cpu().numpy()
    s, it = L._sol["status"].cpu().numpy(), L._sol["iters"].cpu().numpy()
    print("step %d: oc %.2f ric %.2f fwd %.2f ms | OC status %s iters %s | Riccati units/traj %s unmet %d | forward units/traj %s unmet %d | theta |max| %.3g"
          % (k, kt["oc_solve"], kt["aux_riccati"], kt["aux_forward"], np.bincount(s, minlength=5).tolist(), q(it), q(st[:, 0]), int((st[:, 1] > 0).sum()),
             q(st[:, 2]), int((st[:, 3] > 0).sum()), float(L.theta.abs().max())), flush=True)
    worst = np.argsort(-st[:, 0])[:3]
    print("        slowest Riccati rows %s: units %s status %s theta %s" % (worst.tolist(), st[worst, 0].tolist(), s[worst].tolist(),
          np.array2string(L.theta[worst].double().cpu().numpy(), precision=2)), flush=True)
