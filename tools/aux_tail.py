"""Which trajectories set the duration of the two auxiliary sweeps on the headline workload?

After a few outer iterations of the bench learner the Riccati and forward sweeps are timed (HIP events) on the full
batch and on every wavefront-sized slice of it by itself (2 trajectories for the Riccati sweep, 4 for the forward
sweep): the distribution of the per-wavefront times against the full-batch time shows how much of a launch is the
tail of its slowest wavefronts (error-controlled sub-stepping gives every trajectory its own number of split units).
Fixed one-unit sweeps (rtol 0) give the floor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd  # noqa: F401
from lfsd_amd import models
import bench


def phases(oc, sol, L, sl, reps=3):
    s = {k: (v[sl].contiguous() if isinstance(v, torch.Tensor) and v.dim() >= 1 and v.shape[0] == L.B else v) for k, v in sol.items()}
    s.pop("workspace", None)
    best = {"riccati": 1e9, "forward": 1e9}
    for _ in range(reps):
        ev = []
        oc.auxSysSolverBatch(s, L.taus[sl], L.wps[sl], L.iface, phase_hook=lambda nm: (ev.append((nm, torch.cuda.Event(enable_timing=True))), ev[-1][1].record()),
                             validate=False)
        torch.cuda.synchronize()
        t = {ev[i][0]: ev[i][1].elapsed_time(ev[i + 1][1]) for i in range(len(ev) - 1)}
        for k in best:
            best[k] = min(best[k], t[k])
    return best


def main(steps=12):
    args = bench.parse_args(["--no-cpu-baseline"])
    oc, env, d = models.quadrotor(n_grid=args.n_grid)
    oc.setDevice("cuda:0", torch.float32)
    lib = oc.compile()
    L, theta0, x0 = bench.build_learner(args, oc, d, lib, 0, 1, "independent")
    L.count_unconverged = False
    for _ in range(steps):
        L.step()
    sol = dict(L._sol)
    full = slice(0, L.B)
    for rtol, sub in ((1e-3, 0), (0.0, 1), (0.0, 2), (0.0, 4)):
        oc.setSolverOptions(aux_rtol=rtol, aux_substeps=sub)
        t = phases(oc, sol, L, full)
        print("full batch, rtol %g substeps %d: riccati %.3f ms  forward %.3f ms" % (rtol, sub, t["riccati"], t["forward"]), flush=True)
    oc.setSolverOptions(aux_rtol=1e-3, aux_substeps=0)
    n = L.B // 4
    ric = np.zeros(2 * n); fwd = np.zeros(n)
    for w in range(n):
        t = phases(oc, sol, L, slice(4 * w, 4 * w + 4), reps=2)
        fwd[w] = t["forward"]
        for h in range(2):
            ric[2 * w + h] = phases(oc, sol, L, slice(4 * w + 2 * h, 4 * w + 2 * h + 2), reps=2)["riccati"]
    for nm, v in (("riccati (2 trajectories)", ric), ("forward (4 trajectories)", fwd)):
        q = np.percentile(v, [0, 10, 50, 90, 99, 100])
        print("one wavefront alone, %s: min %.3f p10 %.3f median %.3f p90 %.3f p99 %.3f max %.3f ms; mean %.3f" % ((nm,) + tuple(q) + (v.mean(),)))
        h, edges = np.histogram(v / np.median(v), bins=[0, 1.05, 1.25, 1.5, 2, 3, 5, 1e9])
        print("   time / median histogram, edges %s: %s" % (edges[:-1].tolist(), h.tolist()))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
