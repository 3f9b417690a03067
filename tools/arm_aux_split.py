"""Robot arm (configs[1]) learner step split by kernel (HIP events), library-default auxiliary sweeps vs a forced
minimum of four units per interval."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models, CPDP
from arm_steps import seeds, admissible
B = 1024
for sub in (0, 4):
    oc, env, d = models.ZOO["robotarm"](n_grid=50)
    oc.setDevice("cuda:0", torch.float32)
    oc.setSolverOptions(aux_substeps=sub)
    L = CPDP.SparseDemoLearner(oc, np.tile(d["ini_state"], (B, 1)), d["horizon"], d["taus"], d["waypoints"], d["interface"], seeds(B), method="Vanilla", learning_rate=d["lr"], skip_unconverged=False)
    for k in range(9):
        ev = {}
        L.event_hook = lambda nm: ev.setdefault(nm, torch.cuda.Event(enable_timing=True)).record()
        torch.cuda.synchronize(); t0 = time.perf_counter(); L.step(); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
        if k >= 5:
            print("aux_substeps %d step %d: %.1f ms | oc %.1f riccati %.1f forward %.1f" % (sub, k, ms, ev["oc_solve"].elapsed_time(ev["aux_riccati"]), ev["aux_riccati"].elapsed_time(ev["aux_forward"]), ev["aux_forward"].elapsed_time(ev["update"])), flush=True)
