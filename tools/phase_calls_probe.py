"""One OC solve + aux pass with an alternative build of a model library, in a process of its own (a faulting build must not
take the caller down).  usage: phase_calls_probe.py <model> <library.so> <f32|f64> <batch> <n_grid> [wide]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models
kind, lib, dt, B, N = sys.argv[1], sys.argv[2], torch.float32 if sys.argv[3] == "f32" else torch.float64, int(sys.argv[4]), int(sys.argv[5])
oc, env, d = models.ZOO[kind](n_grid=N)
oc.setSolverOptions(mapping="wide" if len(sys.argv) > 6 else "lockstep")
if lib != "default":
    oc.use_library(lib)
oc.setDevice("cuda:0", dt)
p = oc.compile().n_auxvar
rng = np.random.default_rng(0)
th = np.array(d["theta0"])[None, :] * (1 + 0.05 * rng.standard_normal((B, p))); th[:, 0] = np.abs(th[:, 0]) + 0.1
x0 = np.tile(d["ini_state"], (B, 1))
sol = oc.cocSolverBatch(x0, d["horizon"], th); torch.cuda.synchronize()
t0 = time.perf_counter(); sol = oc.cocSolverBatch(x0, d["horizon"], th); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
st = sol["status"].cpu().numpy()
print("OK %s %s %s B %d N %d: %.2f ms status %s cost mean %.6f" % (kind, os.path.basename(lib), sys.argv[3], B, N, ms, np.bincount(st, minlength=5).tolist(), sol["cost"].double().mean().item()), flush=True)
