"""Per outer iteration of the headline workload: oc_solve time, iteration histogram and the slowest trajectories."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd  # noqa: F401
from lfsd_amd import models
import bench


def main(steps=14):
    args = bench.parse_args(["--no-cpu-baseline"])
    oc, env, d = models.quadrotor(n_grid=args.n_grid)
    oc.setDevice("cuda:0", torch.float32)
    lib = oc.compile()
    L, theta0, x0 = bench.build_learner(args, oc, d, lib, 0, 1, "independent")
    L.count_unconverged = False
    for k in range(steps):
        ev = {}
        L.event_hook = lambda nm: ev.setdefault(nm, torch.cuda.Event(enable_timing=True)).record()
        L.step(); torch.cuda.synchronize()
        it = L._sol["iters"].cpu().numpy(); st = L._sol["status"].cpu().numpy()
        slow = np.argsort(-it)[:6]
        print("step %2d: oc_solve %.3f ms riccati %.3f forward %.3f | iters hist %s | slowest %s status %s" %
              (k, ev["oc_solve"].elapsed_time(ev["aux_riccati"]), ev["aux_riccati"].elapsed_time(ev["aux_forward"]),
               ev["aux_forward"].elapsed_time(ev["update"]), np.bincount(it).tolist()[4:], list(zip(slow.tolist(), it[slow].tolist())), st[slow].tolist()), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 14)
