"""Split units per grid interval chosen by the error-controlled auxiliary sweeps on the headline workload.

`python tools/aux_units.py build` (no GPU needed) compiles a variant of the quadrotor library with
-DLFSD_AUX_TRACE=<n>: the first n trajectories print every attempt (units, estimate / tolerance) of every interval.
`python tools/aux_units.py` runs the bench learner for a few outer iterations with the product library, then the
auxiliary pass of the first trajectories with the trace variant, and summarises attempts per interval."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfsd_amd  # noqa: F401
from lfsd_amd import models, runtime

NTRACE = 4


def variant_path(spec):
    return os.path.join(runtime.BUILD_DIR, "trace_%s_aux.so" % spec.hash())


def build():
    oc, env, d = models.quadrotor(n_grid=50)
    spec = oc.model_spec(); runtime.write_header(spec)
    out = variant_path(spec)
    runtime.build_checked(spec, out, ["-DLFSD_AUX_TRACE=%d" % NTRACE])      # (assembly-checked like the product build: lfsd_amd/isa_check.py)
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
    torch.cuda.synchronize()
    sl = slice(0, 8)
    sol = {k: (v[sl].contiguous() if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == L.B else v) for k, v in L._sol.items()}
    sol.pop("workspace", None)
    oc2, _, _ = models.quadrotor(n_grid=args.n_grid)
    oc2.use_library(variant_path(oc2.model_spec())); oc2.setDevice("cuda:0", torch.float32)
    sys.stdout.flush()
    print("=== trace begin", flush=True)
    oc2.auxSysSolverBatch(sol, L.taus[sl], L.wps[sl], L.iface, validate=False)
    torch.cuda.synchronize()
    print("=== trace end", flush=True)


def summarise(path):
    att = {}
    for ln in open(path):
        m = re.match(r"(ric|fwd) traj (\d+) k (\d+) units (\d+) ratio (\S+)", ln)
        if m:
            att.setdefault((m.group(1), int(m.group(2))), {}).setdefault(int(m.group(3)), []).append((int(m.group(4)), float(m.group(5))))
    for (kern, traj), iv in sorted(att.items()):
        final = sum(a[-1][0] for a in iv.values()); spent = sum(u for a in iv.values() for u, _ in a)
        print("%s trajectory %d: %d intervals, units of the accepted attempts %d, units executed incl. rejected attempts %d" % (kern, traj, len(iv), final, spent))
        print("   k: attempts  ", "  ".join("%d:%s" % (k, "/".join("%d(%.2g)" % a for a in iv[k])) for k in sorted(iv) if len(iv[k]) > 1 or iv[k][0][0] > 1))


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        build()
    elif len(sys.argv) > 2 and sys.argv[1] == "summarise":
        summarise(sys.argv[2])
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
