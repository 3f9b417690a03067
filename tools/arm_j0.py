"""How far below its initial cost does an OC solve of a diverged robot-arm seed fall, and how fast?  For the seeds of
BASELINE configs[1] that need >= 60 iterations along 6 Vanilla steps: J_0 (cost of the zero-control roll-out = a solve with
max_iter 0), J after 10 / 20 / 40 / 80 iterations and at the end, beside the admissible seeds' range of (J_0 - J_end) / (1 + |J_0|).
Sizes the divergence test of the OC kernels (cpdp_oc.h, LFSD_DIVERGED)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd  # noqa: F401
from lfsd_amd import models, CPDP
from arm_steps import seeds, admissible

B = 1024
oc, env, d = models.ZOO["robotarm"](n_grid=50)
oc.setDevice("cuda:0", torch.float32)
L = CPDP.SparseDemoLearner(oc, np.tile(d["ini_state"], (B, 1)), d["horizon"], d["taus"], d["waypoints"], d["interface"], seeds(B),
                           method="Vanilla", learning_rate=d["lr"], skip_unconverged=False)
x0 = np.tile(d["ini_state"], (B, 1))


def cost_after(th, iters):
    o2, _, _ = models.ZOO["robotarm"](n_grid=50)
    o2.setDevice("cuda:0", torch.float32)
    o2.setSolverOptions(max_iter=iters)
    s = o2.cocSolverBatch(x0, d["horizon"], th)
    return s["cost"].double().cpu().numpy(), s["status"].cpu().numpy()


for k in range(6):
    th = L.theta.clone()
    adm = admissible(th)
    L.step(); torch.cuda.synchronize()
    st = L._sol["status"].cpu().numpy(); it = L._sol["iters"].cpu().numpy(); J = L._sol["cost"].double().cpu().numpy()
    J0, _ = cost_after(th, 0)
    drop = (J0 - J) / (1 + np.abs(J0))
    print("step %d: status %s; admissible seeds: (J0 - J_end)/(1+|J0|) max %.3g, J0 range %.3g..%.3g, iterations max %d" %
          (k, np.bincount(st, minlength=5).tolist(), drop[adm & np.isfinite(drop)].max(), J0[adm].min(), J0[adm].max(), it[adm].max()), flush=True)
    long_ = np.where(it >= 60)[0][:10]
    if len(long_):
        snaps = {n: cost_after(th, n)[0] for n in (10, 20, 40, 80)}
        for b in long_:
            print("   seed %4d adm %d status %d iters %3d theta %s  J0 %.4g  J@10 %.4g  J@20 %.4g  J@40 %.4g  J@80 %.4g  J_end %.4g  drop %.3g" %
                  (b, adm[b], st[b], it[b], np.array2string(th[b].double().cpu().numpy(), precision=2), J0[b], snaps[10][b], snaps[20][b],
                   snaps[40][b], snaps[80][b], J[b], drop[b]), flush=True)
