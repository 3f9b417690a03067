"""What do the seeds that exhaust the iteration limit on BASELINE configs[1] look like?  (theta, cost, iterate size)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd  # noqa: F401
from lfsd_amd import models, CPDP
from arm_steps import seeds, admissible

B = 1024
oc, env, d = models.ZOO["robotarm"](n_grid=50)
oc.setDevice("cuda:0", torch.float32)
oc.setSolverOptions(aux_substeps=4)
L = CPDP.SparseDemoLearner(oc, np.tile(d["ini_state"], (B, 1)), d["horizon"], d["taus"], d["waypoints"], d["interface"], seeds(B),
                           method="Vanilla", learning_rate=d["lr"], skip_unconverged=False)
for k in range(5):
    th = L.theta.double().cpu().numpy().copy()
    adm = admissible(L.theta)
    L.step(); torch.cuda.synchronize()
    st = L._sol["status"].cpu().numpy(); it = L._sol["iters"].cpu().numpy()
    X = L._sol["state_grid"].double().cpu().numpy(); U = L._sol["control_grid"].double().cpu().numpy(); J = L._sol["cost"].double().cpu().numpy()
    long_ = np.where(it >= 60)[0]
    print("step %d: status %s, %d trajectories with >= 60 iterations (admissible among them: %d)" % (k, np.bincount(st, minlength=5).tolist(), len(long_), adm[long_].sum()))
    for b in long_[:12]:
        print("   seed %4d status %d iters %3d admissible %d theta %s J %.4g max|x| %.3g max|u| %.3g" %
              (b, st[b], it[b], adm[b], np.array2string(th[b], precision=3), J[b], np.abs(X[b]).max(), np.abs(U[b]).max()))
    print("   admissible seeds: iterations max %d, J range %.3g .. %.3g, max|x| %.3g, max|u| %.3g" %
          (it[adm].max(), J[adm].min(), J[adm].max(), np.abs(X[adm]).max(), np.abs(U[adm]).max()), flush=True)
