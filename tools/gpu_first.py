"""First GPU contact: smoke, quadrotor parity against the golden fixture, rough timing."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models
import __graft_entry__ as ge

t0 = time.time(); ge.smoke(); print("smoke %.1fs" % (time.time() - t0), flush=True)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "uav_golden.npz"))
for dt in (torch.float64, torch.float32):
    oc, env, d = models.quadrotor(n_grid=25)
    oc.setDevice("cuda:0", dt)
    consts = oc.consts_tensor(overrides=dict(goal_r0=2.5, goal_r1=1.0, goal_r2=1.5))
    idx = [0, 1, 5, 20, 60, 80, 99]
    th = g["lookahead_theta"][idx]
    sol = oc.cocSolverBatch(np.tile(g["ini_state"], (len(idx), 1)), 1.0, th, consts=consts)
    aux = oc.auxSysSolverBatch(sol, g["taus"], g["waypoints"], [0, 1, 2])
    torch.cuda.synchronize()
    loss = aux["loss"].double().cpu().numpy(); grad = aux["grad"].double().cpu().numpy()
    print(dt, "status", sol["status"].tolist(), "iters", sol["iters"].tolist())
    for k, j in enumerate(idx):
        print("  j=%d loss %.8f golden %.8f rel %.2e | grad rel-to-max err %.2e" % (
            j, loss[k], g["loss_trace"][j], abs(loss[k] - g["loss_trace"][j]) / g["loss_trace"][j],
            np.abs(grad[k] - g["grad_trace"][j]).max() / np.abs(g["grad_trace"][j]).max()), flush=True)
# timing
for dt in (torch.float32, torch.float64):
    for B, N in ((4096, 25), (4096, 50)):
        oc, env, d = models.quadrotor(n_grid=N)
        oc.setDevice("cuda:0", dt)
        rng = np.random.default_rng(0)
        th = np.array(d["theta0"])[None, :] + 0.05 * rng.standard_normal((B, 7)); th[:, 0] = np.abs(th[:, 0]) + 0.5
        x0 = np.tile(d["ini_state"], (B, 1))
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.time()
            sol = oc.cocSolverBatch(x0, 1.0, th)
            torch.cuda.synchronize(); t1 = time.time()
            aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"])
            torch.cuda.synchronize(); t2 = time.time()
        st = sol["status"].cpu().numpy(); it = sol["iters"].cpu().numpy()
        print(dt, "B", B, "N", N, "coc %.1f ms aux %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), "status hist",
              np.bincount(st, minlength=5).tolist(), "iters mean %.1f max %d" % (it.mean(), it.max()),
              "loss mean", float(aux["loss"].mean()), flush=True)
