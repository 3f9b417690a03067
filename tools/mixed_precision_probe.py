"""Would an fp32 solve hand the fp64 solve a useful start?  Headline workload (quadrotor, n_grid 50, 4 096 seeds): the fp64
OC solve cold (the product path of bench.py --dtype f64) against fp32 cold solve + fp64 solve started from its controls.

    python tools/mixed_precision_probe.py [batch]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import lfsd_amd  # noqa: F401
from lfsd_amd import models

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
args = bench.parse_args(["--no-cpu-baseline", "--batch", str(B)])
w = bench.WORKLOADS["quadrotor"]


def learner(dt):
    a = bench.parse_args(["--no-cpu-baseline", "--batch", str(B), "--dtype", dt])
    oc, env, d = models.ZOO[w["kind"]](n_grid=a.n_grid)
    oc.setDevice("cuda:0", torch.float32 if dt == "f32" else torch.float64)
    lib = oc.compile()
    d = dict(d)
    d["taus"], d["waypoints"] = bench.demonstration(oc, d, a.n_grid)
    L, theta0, x0 = bench.build_learner(a, oc, d, lib, 0, 1, "independent", w)
    return oc, d, L, x0


def timed(f, n=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), r


oc32, d, L32, x0 = learner("f32")
oc64, _, L64, _ = learner("f64")
for step in range(3):
    th = L64.theta.detach().clone()
    x0t = torch.as_tensor(np.asarray(x0.cpu() if hasattr(x0, "cpu") else x0), device="cuda:0")
    ms64, s64 = timed(lambda: oc64.cocSolverBatch(x0t.double(), d["horizon"], th.double()))
    ms32, s32 = timed(lambda: oc32.cocSolverBatch(x0t.float(), d["horizon"], th.float()))
    N = oc64.n_grid
    u0 = s32["control_grid"][:, :N].double().contiguous()
    msw, sw = timed(lambda: oc64.cocSolverBatch(x0t.double(), d["horizon"], th.double(), u_init=u0))
    q = lambda s: "iters mean %.2f max %d status %s" % (s["iters"].double().mean().item(), int(s["iters"].max()), np.bincount(s["status"].cpu().numpy(), minlength=5).tolist())
    dJ = ((sw["cost"] - s64["cost"]).abs() / s64["cost"].abs()).max().item()
    du = (sw["control_grid"] - s64["control_grid"]).abs().max().item()
    dl = (sw["costate_grid"] - s64["costate_grid"]).abs().max().item() / s64["costate_grid"].abs().max().item()
    print("outer iteration %d: fp64 cold %.2f ms (%s) | fp32 cold %.2f ms (%s) | fp64 from the fp32 controls %.2f ms (%s) | sum %.2f ms | vs cold fp64: cost %.1e controls %.1e costates %.1e"
          % (step, ms64, q(s64), ms32, q(s32), msw, q(sw), ms32 + msw, dJ, du, dl), flush=True)
    L64.step()
