"""Diagnostic: robot arm (BASELINE configs[1]) -- OC solve at the step-1 parameters theta1 = theta0 - lr*grad0, fp32 vs fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models, CPDP
B = 1024
rng = np.random.default_rng(0)
oc, env, d = models.ZOO["robotarm"](n_grid=50)
th0 = np.array(d["theta0"])[None, :] * (1 + 0.05 * rng.standard_normal((B, 5)))
th0[:, 0] = np.abs(th0[:, 0]) + 0.1
x0 = np.tile(d["ini_state"], (B, 1))
oc.setDevice("cuda:0", torch.float64)
sol = oc.cocSolverBatch(x0, d["horizon"], th0)
aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"])
th1 = th0 - d["lr"] * aux["grad"].cpu().numpy()
th1[:, 0] = np.maximum(th1[:, 0], 1e-8)
print("theta1 range", th1.min(0).round(3).tolist(), th1.max(0).round(3).tolist())
res = {}
for dt in (torch.float32, torch.float64):
    oc, env, d = models.ZOO["robotarm"](n_grid=50)
    oc.setDevice("cuda:0", dt)
    for ea in (16, 0, -1):
        oc.setSolverOptions(exact_after=ea)
        s = oc.cocSolverBatch(x0, d["horizon"], th1)
        a = oc.auxSysSolverBatch(s, d["taus"], d["waypoints"], d["interface"])
        st = s["status"].cpu().numpy(); it = s["iters"].cpu().numpy()
        g = a["grad"].double().cpu().numpy()
        print(dt, "exact_after", ea, "status", np.bincount(st, minlength=5).tolist(), "iters mean %.1f max %d" % (it.mean(), it.max()),
              "|grad| max: median %.3g max %.3g" % (np.median(np.abs(g).max(1)), np.abs(g).max()), "cost mean %.6f" % s["cost"].double().mean().item())
        res[(dt, ea)] = (s, a)
s32 = res[(torch.float32, 16)][0]; s64 = res[(torch.float64, 16)][0]
bad = np.where(s32["status"].cpu().numpy() == 3)[0]
print("fp32 MAXITER seeds:", bad[:12].tolist())
for b in bad[:5]:
    print("seed", b, "theta1", th1[b].round(4).tolist(), "fp32 cost %.8f iters %d | fp64 cost %.8f status %d iters %d" %
          (s32["cost"][b].item(), int(s32["iters"][b]), s64["cost"][b].item(), int(s64["status"][b]), int(s64["iters"][b])))
np.save("gpurun_out/arm_theta1.npy", th1)
