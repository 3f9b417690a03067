"""Shader-clock split of one stage of the structural backward sweep (OcSolver::backward_sc) on the headline workload.
`python tools/ab_variants.py build bwclock` (no GPU) compiles the quadrotor library with -DLFSD_BW_CLOCK;
`python tools/bw_clock.py` solves the benchmark's 4096 seeds once with it: wavefront 0 prints, per backward sweep, the clocks
spent in each of the six phases of the 50 stages (s_memtime stamps cost ~10 %: shares, not absolutes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lfsd_amd  # noqa: F401
from lfsd_amd import models
import bench
from ab_variants import variant_path

args = bench.parse_args(["--no-cpu-baseline"])
oc, env, d = models.quadrotor(n_grid=args.n_grid)
oc.use_library(variant_path(oc.model_spec(), sys.argv[1] if len(sys.argv) > 1 else "bwclock"))
oc.setDevice("cuda:0", torch.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else torch.float32)
L, theta0, x0 = bench.build_learner(args, oc, d, oc.compile(), 0, 1, "independent")
for rep in range(2):
    print("=== solve %d" % rep, flush=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); sol = oc.cocSolverBatch(L.x0, L.hz, L.theta, consts=L.consts); b.record(); torch.cuda.synchronize()
    print("kernel %.3f ms, iterations mean %.2f" % (a.elapsed_time(b), sol["iters"].float().mean().item()), flush=True)
