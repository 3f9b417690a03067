"""Shader-clock split of the lock-step oc_solve kernel on the headline workload: backward sweep vs roll-out +
linearisation vs line search, for the first wavefronts of a full-batch launch.

`python tools/oc_clock.py build` (no GPU needed) compiles the quadrotor library with -DLFSD_OC_CLOCK=<n>;
`python tools/oc_clock.py [steps]` runs the bench learner with the product library, then one cold-started solve of the
full batch with the instrumented variant."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfsd_amd  # noqa: F401
from lfsd_amd import models, runtime

NWAVES = 6


def variant_path(spec):
    return os.path.join(runtime.BUILD_DIR, "trace_%s_occlock.so" % spec.hash())


def build():
    oc, env, d = models.quadrotor(n_grid=50)
    spec = oc.model_spec(); runtime.write_header(spec)
    out = variant_path(spec)
    runtime.build_checked(spec, out, ["-DLFSD_OC_CLOCK=%d" % NWAVES])      # (assembly-checked like the product build: lfsd_amd/isa_check.py)
    print(out)


def run(steps):
    import torch
    import bench
    args = bench.parse_args(["--no-cpu-baseline"])
    oc, env, d = models.quadrotor(n_grid=args.n_grid)
    oc.setDevice("cuda:0", torch.float32)
    lib = oc.compile()
    L, theta0, x0 = bench.build_learner(args, oc, d, lib, 0, 1, "independent")
    L.count_unconverged = False
    for _ in range(steps):
        L.step()
    th = lib.lookahead(L.theta, L.m, L.mu).clone()
    torch.cuda.synchronize()
    oc2, _, _ = models.quadrotor(n_grid=args.n_grid)
    oc2.use_library(variant_path(oc2.model_spec())); oc2.setDevice("cuda:0", torch.float32)
    for rep in range(2):
        print("=== launch %d" % rep, flush=True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); sol = oc2.cocSolverBatch(L.x0, L.hz, th, consts=L.consts); b.record(); torch.cuda.synchronize()
        print("kernel %.3f ms (with printf), iterations of the first trajectories %s" % (a.elapsed_time(b), sol["iters"][:4 * NWAVES].cpu().numpy().tolist()), flush=True)


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
