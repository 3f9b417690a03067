"""Iteration trace (-DLFSD_TRACE) of the slowest trajectories of the headline workload.

`python tools/oc_straggler.py build` compiles the quadrotor library with -DLFSD_TRACE; `python tools/oc_straggler.py`
runs the bench learner until an outer iteration has a trajectory with >= 10 OC iterations, then re-solves the slowest
three one at a time (lock-step mapping, batch 1) with the trace variant."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfsd_amd  # noqa: F401
from lfsd_amd import models, runtime


def variant_path(spec):
    return os.path.join(runtime.BUILD_DIR, "trace_%s_oc.so" % spec.hash())


def build():
    oc, env, d = models.quadrotor(n_grid=50)
    spec = oc.model_spec(); runtime.write_header(spec)
    out = variant_path(spec)
    runtime.build_checked(spec, out, ["-DLFSD_TRACE"])      # (assembly-checked like the product build: lfsd_amd/isa_check.py)
    print(out)


def run(max_steps):
    import torch
    import bench
    args = bench.parse_args(["--no-cpu-baseline"])
    oc, env, d = models.quadrotor(n_grid=args.n_grid)
    oc.setDevice("cuda:0", torch.float32)
    lib = oc.compile()
    L, theta0, x0 = bench.build_learner(args, oc, d, lib, 0, 1, "independent")
    L.count_unconverged = False
    for k in range(max_steps):
        th = lib.lookahead(L.theta, L.m, L.mu).clone()
        L.step(); torch.cuda.synchronize()
        it = L._sol["iters"].cpu().numpy()
        if it.max() >= 10:
            break
    slow = np.argsort(-it)[:3]
    print("step %d: slowest %s iterations %s" % (k, slow.tolist(), it[slow].tolist()), flush=True)
    oc2, _, _ = models.quadrotor(n_grid=args.n_grid)
    oc2.use_library(variant_path(oc2.model_spec())); oc2.setDevice("cuda:0", torch.float32)
    oc2.setSolverOptions(mapping="lockstep")
    for j in slow:
        print("=== trajectory %d, theta %s" % (j, th[j].cpu().numpy().tolist()), flush=True)
        sol = oc2.cocSolverBatch(L.x0[j:j + 1], L.hz[j:j + 1], th[j:j + 1], consts=L.consts)
        torch.cuda.synchronize()
        print("=== iterations %d status %d" % (int(sol["iters"][0]), int(sol["status"][0])), flush=True)


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 14)
