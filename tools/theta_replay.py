"""Re-solve saved parameter sets (tools/path_trace.py: <tag>_theta.npz, slowest first) with build variants of a model library:
launch time and iteration statistics per variant.  One launch = the whole file (the learner's batch at that outer iteration).

    python tools/theta_replay.py <robotarm|rocket|quadrotor> <file.npz> <product|variant tag> [...]     (variants: tools/model_ab.py build)
    LFSD_REPLAY_ROWS=16  only the first rows (the slowest trajectories); LFSD_REPLAY_CLOCK=1 prints the kernels' own output
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import bench
import lfsd_amd  # noqa: F401
from lfsd_amd import models
import oc_trace

cfg, path = sys.argv[1], sys.argv[2]
args = bench.parse_args(["--config", cfg, "--no-cpu-baseline"])
w = bench.WORKLOADS[cfg]
TD = {"f32": torch.float32, "f64": torch.float64}
z = np.load(path)
rows = int(os.environ.get("LFSD_REPLAY_ROWS", "0")) or len(z["theta"])
th, x0 = z["theta"][:rows], z["x0"][:rows]
print("%s: %d parameter sets of %s (iterations when saved: mean %.1f max %d)" % (cfg, rows, path, z["iters"][:rows].mean(), z["iters"][:rows].max()))
for tag in sys.argv[3:]:
    oc, env, d = models.ZOO[w["kind"]](n_grid=args.n_grid)
    if tag != "product":
        oc.use_library(oc_trace.variant_path(oc.model_spec(), tag))
    oc.setDevice("cuda:0", TD[args.dtype])
    oc.compile()
    oc.cocSolverBatch(x0, float(z["horizon"]), th); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        t0 = time.perf_counter(); sol = oc.cocSolverBatch(x0, float(z["horizon"]), th); torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    it, st, J = sol["iters"].cpu().numpy(), sol["status"].cpu().numpy(), sol["cost"].double().cpu().numpy()
    print("%-14s %8.2f ms | status %s | iterations mean %.1f p50 %d p90 %d p99 %d max %d | sum J (status 1, 2) %.6f | first rows %s" %
          (tag, min(ms), np.bincount(st, minlength=5).tolist(), it.mean(), np.median(it), np.quantile(it, .9), np.quantile(it, .99), it.max(),
           J[(st == 1) | (st == 2)].sum(), it[:8].tolist()), flush=True)
