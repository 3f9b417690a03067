"""GPU repro helper: the inputs of tests/test_gpu_parity.py::test_robotarm_cartpole_vs_oracle, one launch at a time."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models
CASES = [("robotarm", 12, [[5., 1, 1, 1, 1], [3., 0.5, 2, 1.5, 0.2]], [0.3], [[-np.pi / 4, 2 * np.pi / 3]]),
         ("cartpole", 10, [[1.0, 0.5, 0.5, 0.5, 0.5], [0.8, 2, 0.3, 1, 1]], [0.25, 0.8], [[0.1, 0.5], [0.0, 2.5]])]
for kind, n_grid, thetas, taus, wps in CASES:
    for dt in (torch.float32,):
        oc, env, d = models.ZOO[kind](n_grid=n_grid)
        oc.setDevice("cuda:0", dt)
        oc.setSolverOptions(aux_substeps=16)
        if len(sys.argv) > 1:
            oc.setSolverOptions(exact_after=int(sys.argv[1]))
        B = len(thetas)
        print(kind, dt, "coc ...", flush=True)
        sol = oc.cocSolverBatch(np.tile(d["ini_state"], (B, 1)), d["horizon"], np.array(thetas))
        torch.cuda.synchronize()
        print("   status", sol["status"].tolist(), "iters", sol["iters"].tolist(), flush=True)
        aux = oc.auxSysSolverBatch(sol, taus, wps, d["interface"])
        torch.cuda.synchronize()
        print("   loss", aux["loss"].tolist(), flush=True)
