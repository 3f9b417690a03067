"""Diagnostic: lock-step waste of the OC solve on the bench workload -- a wavefront of 4 trajectories runs as long as its
slowest member.  Prints the iteration histogram, the mean of per-wave maxima in natural and in sorted order, and how
well the previous outer iteration's counts predict the next ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models, CPDP
B = 4096
oc, env, d = models.quadrotor(n_grid=50)
oc.setDevice("cuda:0", torch.float32)
rng = np.random.default_rng(1234)
th = np.array(d["theta0"])[None, :] + 0.05 * rng.standard_normal((B, 7)); th[:, 0] = np.abs(th[:, 0]) + 0.5
L = CPDP.SparseDemoLearner(oc, np.tile(d["ini_state"], (B, 1)), d["horizon"], d["taus"], d["waypoints"], d["interface"], th,
                           method="Nesterov", learning_rate=1e-2, mu=0.9)
prev = None
for k in range(8):
    L.step()
    it = L._sol["iters"].cpu().numpy()
    nat = it.reshape(-1, 4).max(1).mean()
    srt = np.sort(it).reshape(-1, 4).max(1).mean()
    msg = "step %d iters hist %s mean %.3f | per-wave max: natural %.3f sorted %.3f" % (k, np.bincount(it).tolist(), it.mean(), nat, srt)
    if prev is not None:
        order = np.argsort(prev, kind="stable")
        pred = it[order].reshape(-1, 4).max(1).mean()
        msg += " ordered by previous step's counts %.3f  (corr %.2f)" % (pred, np.corrcoef(prev, it)[0, 1])
    print(msg, flush=True)
    prev = it
