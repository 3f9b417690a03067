"""How much of the headline oc_solve launch is the tail of its slowest wavefronts?

Runs the bench learner for a few outer iterations, then re-solves the same parameter set with the iteration budget
capped at 4..12 (diagnosis only: a capped solve is not converged) and prints kernel time, status and iteration
histograms.  If the time with the cap at the typical iteration count is well below the uncapped time, the launch is
waiting for a few slow trajectories."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd  # noqa: F401
from lfsd_amd import models
import bench


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return min(ts), r


def main(steps=12):
    args = bench.parse_args(["--no-cpu-baseline"])
    oc, env, d = models.quadrotor(n_grid=args.n_grid)
    oc.setDevice("cuda:0", torch.float32)
    lib = oc.compile()
    L, theta0, x0 = bench.build_learner(args, oc, d, lib, 0, 1, "independent")
    L.count_unconverged = False
    for _ in range(steps):
        L.step()
    th = lib.lookahead(L.theta, L.m, L.mu).clone()
    it_full = None
    for cap in (300, 12, 10, 8, 7, 6, 5, 4):
        oc.setSolverOptions(max_iter=cap)
        ms, sol = timed(lambda: oc.cocSolverBatch(L.x0, L.hz, th, consts=L.consts))
        st = sol["status"].cpu().numpy(); it = sol["iters"].cpu().numpy()
        if it_full is None:
            it_full = it
            print("uncapped iteration histogram:", np.bincount(it).tolist())
            w = it.reshape(-1, 4).max(axis=1)            # four trajectories share a wavefront of the packed kernel
            print("per-wavefront max histogram :", np.bincount(w).tolist(), " mean of wave max %.2f, mean %.2f" % (w.mean(), it.mean()))
        print("max_iter %3d: oc_solve %.3f ms  status %s  iters mean %.2f max %d" %
              (cap, ms, np.bincount(st, minlength=5).tolist(), it.mean(), it.max()), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 12)
