"""Lock-step vs wide mapping of the OC solve (setSolverOptions(mapping=...)) on the small-batch BASELINE configurations:
kernel time of one cold-started solve, status histogram, iteration counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models

def run(kind, n_grid, B, dtype, reps=3):
    for wide in ("0", "1"):
        oc, env, d = models.ZOO[kind](n_grid=n_grid)
        oc.setSolverOptions(mapping="wide" if wide == "1" else "lockstep")
        oc.setDevice("cuda:0", dtype)
        p = oc.compile().n_auxvar
        rng = np.random.default_rng(0)
        th = np.array(d["theta0"])[None, :] * (1 + 0.05 * rng.standard_normal((B, p)))
        th[:, 0] = np.abs(th[:, 0]) + 0.1
        x0 = np.tile(d["ini_state"], (B, 1))
        oc.cocSolverBatch(x0, d["horizon"], th); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); sol = oc.cocSolverBatch(x0, d["horizon"], th); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        st = sol["status"].cpu().numpy(); it = sol["iters"].cpu().numpy()
        print("%-9s n_grid %3d batch %5d %s %-9s: %8.2f ms | status %s | iters mean %.1f max %d | cost mean %.6f" %
              (kind, n_grid, B, str(dtype)[6:], "wide" if wide == "1" else "lock-step", min(ts), np.bincount(st, minlength=5).tolist(),
               it.mean(), it.max(), sol["cost"].double().mean().item()), flush=True)

if __name__ == "__main__":
    run("robotarm", 50, 1024, torch.float32)
    run("robotarm", 50, 1024, torch.float64)
    run("quadrotor", 50, 512, torch.float32)
    run("quadrotor", 50, 1024, torch.float32)
    run("quadrotor", 50, 2048, torch.float32)
    run("quadrotor", 50, 4096, torch.float32)
    run("rocket", 100, 1024, torch.float32)
    run("pendulum", 50, 1024, torch.float32)
