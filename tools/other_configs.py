"""Timing of the other BASELINE.json configurations (secondary to bench.py's headline workload)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models, CPDP

def run(kind, n_grid, B, dtype, steps=5, aux_dtype=None, lr=None):
    oc, env, d = models.ZOO[kind](n_grid=n_grid)
    oc.setDevice("cuda:0", dtype, aux_dtype=aux_dtype)
    rng = np.random.default_rng(0)
    p = oc.compile().n_auxvar
    th = np.array(d["theta0"])[None, :] * (1 + 0.05 * rng.standard_normal((B, p)))
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    x0 = np.tile(d["ini_state"], (B, 1))
    if "taus" in d:
        taus, wps = d["taus"], d["waypoints"]
    else:   # ground-truth style demonstrations (rocket / pendulum): waypoints from the true parameters
        sol = oc.cocSolverBatch([d["ini_state"]], d["horizon"], [d["true_theta"]])
        tg = np.linspace(0, d["horizon"], n_grid + 1)
        idx = sorted(set(int(round(f * n_grid)) for f in (0.07, 0.2, 0.4, 0.67, 0.87)))
        taus = tg[idx]
        wps = sol["state_grid"][0, idx][:, d["interface"]].double().cpu().numpy()
    L = CPDP.SparseDemoLearner(oc, x0, d["horizon"], taus, wps, d["interface"], th, method="Vanilla",
                               learning_rate=lr or d["lr"])
    L.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, grad = L.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = L._sol["status"].cpu().numpy(); it = L._sol["iters"].cpu().numpy()
    print("%-10s n_grid %3d batch %5d %s%s: %.1f ms/iteration = %.0f trajectory outer-iterations/s | OC status hist %s iters mean %.1f max %d | loss mean %.4g"
          % (kind, n_grid, B, str(dtype)[6:], ("+aux " + str(aux_dtype)[6:]) if aux_dtype else "", dt * 1e3, B / dt,
             np.bincount(st, minlength=5).tolist(), it.mean(), it.max(), float(loss.mean())), flush=True)

if __name__ == "__main__":
    run("pendulum", 50, 4096, torch.float32)
    run("robotarm", 50, 1024, torch.float32)                       # BASELINE configs[1]
    run("quadrotor", 50, 4096, torch.float32)                      # configs[2] (bench.py headline, here with Vanilla GD)
    run("quadrotor", 50, 32768, torch.float32, steps=2)            # configs[3] per-node batch (one GPU's worth x8)
    run("rocket", 100, 1024, torch.float32, steps=2, aux_dtype=torch.float64)   # configs[4] robot/horizon, mixed precision
