"""Shader-clock split (backward sweep | roll-out + linearisation | line search) of the lean OC kernel in fp64 or fp32 on the
headline seeds.  `python tools/ab_variants.py build occlock` first; `python tools/oc_clock64.py [f64|f32]`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lfsd_amd  # noqa: F401
from lfsd_amd import models
import bench
from ab_variants import variant_path

dt = torch.float64 if (len(sys.argv) < 2 or sys.argv[1] == "f64") else torch.float32
args = bench.parse_args(["--no-cpu-baseline"])
oc, env, d = models.quadrotor(n_grid=args.n_grid)
oc.use_library(variant_path(oc.model_spec(), "occlock"))
oc.setDevice("cuda:0", dt)
L, theta0, x0 = bench.build_learner(args, oc, d, oc.compile(), 0, 1, "independent")
for rep in range(2):
    print("=== solve %d" % rep, flush=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); sol = oc.cocSolverBatch(L.x0, L.hz, L.theta, consts=L.consts); b.record(); torch.cuda.synchronize()
    print("kernel %.3f ms, iterations mean %.2f" % (a.elapsed_time(b), sol["iters"].float().mean().item()), flush=True)
