"""Long-run check of bench.py's quadrotor learner: run it until the first seed whose parameters jump (max |theta_new - theta_old| > `jump`),
and save what that seed's outer iteration saw -- the parameters its gradient was evaluated at, the loss, the gradient, the OC status -- so
that the oracle can be asked for the gradient at the same parameters (tests-style check, run by hand: profiles/r06_w_long_runs.txt).

    python tools/first_jump.py <f32|f64> [max_steps] [jump]      -> gpurun_out/first_jump_<dtype>.npz
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import lfsd_amd  # noqa: F401
from lfsd_amd import models

dt = sys.argv[1]
max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
jump = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
args = bench.parse_args(["--no-cpu-baseline", "--dtype", dt])
w = bench.WORKLOADS["quadrotor"]
TD = {"f32": torch.float32, "f64": torch.float64}
oc, env, d = models.ZOO["quadrotor"](n_grid=args.n_grid)
oc.setDevice("cuda:0", TD[dt])
lib = oc.compile()
d = dict(d)
d["taus"], d["waypoints"] = bench.demonstration(oc, d, args.n_grid)
L, theta0, x0 = bench.build_learner(args, oc, d, lib, 0, 1, "independent", w)
L.count_unconverged = False
for k in range(max_steps):
    th_old = L.theta.clone()
    th_eval = lib.lookahead(L.theta, L.m, L.mu).clone()
    loss, grad = L.step()
    dth = (L.theta - th_old).abs().max(dim=1).values
    bad = torch.nonzero(~torch.isfinite(dth) | (dth > jump)).flatten()
    if len(bad):
        j = int(bad[0])
        st = L._sol["status"].cpu().numpy(); it = L._sol["iters"].cpu().numpy()
        print("step %d: %d seed(s) jump; first: seed %d, |dtheta| %.3g, status %d, iterations %d, loss %.6g" % (k, len(bad), j, float(dth[j]), st[j], it[j], float(loss[j])))
        print("theta_eval", th_eval[j].double().cpu().numpy()); print("grad", grad[j].double().cpu().numpy()); print("theta_old", th_old[j].double().cpu().numpy())
        stats = L._aux["stats"][j].cpu().numpy() if "stats" in L._aux else None
        print("aux stats (riccati units, unmet, forward units, unmet):", stats)
        os.makedirs("gpurun_out", exist_ok=True)
        np.savez("gpurun_out/first_jump_%s.npz" % dt, step=k, seed=j, theta_eval=th_eval[j].double().cpu().numpy(), theta_old=th_old[j].double().cpu().numpy(),
                 m_old=0, grad=grad[j].double().cpu().numpy(), loss=float(loss[j]), status=int(st[j]), iters=int(it[j]),
                 x0=np.asarray(x0.double().cpu().numpy() if hasattr(x0, "cpu") else x0)[j] if np.ndim(x0) > 1 else np.asarray(x0),
                 taus=np.asarray(d["taus"]), waypoints=np.asarray(d["waypoints"]), horizon=float(d["horizon"]),
                 state=L._sol["state_grid"][j].double().cpu().numpy(), control=L._sol["control_grid"][j].double().cpu().numpy(),
                 cost=float(L._sol["cost"][j]), stats=stats if stats is not None else 0)
        break
    if k % 20 == 0:
        print("step %d: max |dtheta| %.3g, |theta| max %.3g" % (k, float(dth.max()), float(L.theta.abs().max())), flush=True)
else:
    print("no jump in %d steps" % max_steps)
