"""BASELINE configs[1] (robot arm, n_grid 50, 1024 random seeds): OC status histogram at theta_0, at theta_1 = theta_0 - lr*grad_0
and along 12 Vanilla steps at the example's learning rate 0.1 (Examples/robotarm_random.py:60-73) with every gradient
applied (skip_unconverged=False), library-default auxiliary sweeps (error-controlled from one unit per interval).  A seed is *admissible* while its parameters keep the problem well posed: finite and
below 1e3, time-warp beta > 0 and both quadratic state weights > 0.05 (convex running cost); the few seeds whose (correct, oracle-checked)
sensitivity at theta_1 is 20-100x the typical one are thrown out of that region by the fixed learning rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models, CPDP


def seeds(B=1024):
    rng = np.random.default_rng(0)
    th = np.array([5.0, 1, 1, 1, 1])[None, :] * (1 + 0.05 * rng.standard_normal((B, 5)))
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    return th


def admissible(theta):
    th = theta.detach().double().cpu().numpy() if isinstance(theta, torch.Tensor) else np.asarray(theta)
    return np.isfinite(th).all(1) & (np.abs(th) < 1e3).all(1) & (th[:, 0] > 0) & (th[:, 1] > 0.05) & (th[:, 3] > 0.05)


if __name__ == "__main__":
    dev = sys.argv[1] if len(sys.argv) > 1 else "cuda:0"
    B = 1024
    th0 = seeds(B)
    for dt in (torch.float32, torch.float64):
        oc, env, d = models.ZOO["robotarm"](n_grid=50)
        oc.setDevice(dev, dt)
        x0 = np.tile(d["ini_state"], (B, 1))
        L = CPDP.SparseDemoLearner(oc, x0, d["horizon"], d["taus"], d["waypoints"], d["interface"], th0, method="Vanilla",
                                   learning_rate=d["lr"], skip_unconverged=False)
        for k in range(13):
            adm = admissible(L.theta)
            th_prev = L.theta.double().cpu().numpy().copy()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loss, grad = L.step()
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
            st = L._sol["status"].cpu().numpy(); it = L._sol["iters"].cpu().numpy()
            g = grad.double().cpu().numpy()
            th_before = None
            bad = np.where(adm & ~np.isin(st, (1, 2)))[0]
            for b in bad[:4]:
                print("   admissible but not converged: seed %d status %d iters %d theta(before step) %s" % (b, st[b], it[b], np.array2string(th_prev[b], precision=6)), flush=True)
            print("%s step %2d: %7.1f ms | status %s | admissible %4d, of them not converged %d | iters mean %.1f max %d (admissible max %d) | "
                  "|grad| median %.3g max(admissible) %.3g" %
                  (str(dt)[6:], k, ms, np.bincount(st, minlength=5).tolist(), adm.sum(), (~np.isin(st[adm], (1, 2))).sum(), it.mean(),
                   it.max(), it[adm].max(), np.median(np.abs(g).max(1)), np.nanmax(np.abs(g[adm]))), flush=True)
