"""Robot arm (BASELINE configs[1]) at theta_1, 1 024 seeds: the fp32 solve of the product build and of build variants against the fp64
solve of the product build -- state / gradient error quantiles over ALL seeds and over the seeds the GPU tier compares with the oracle
(tests/test_gpu_parity.py::test_robotarm_theta1_vs_oracle_16_seeds), iterations, solve time.

    python tools/arm_accuracy_ab.py [tag ...]      ("product", "plain" = csrc/build/ab_<hash>_plain.so, else trace_<hash>_<tag>.so)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import lfsd_amd, oc_trace
from lfsd_amd import models, runtime
B = 1024
rng = np.random.default_rng(0)
th0 = np.array([5.0, 1, 1, 1, 1])[None, :] * (1 + 0.05 * rng.standard_normal((B, 5))); th0[:, 0] = np.abs(th0[:, 0]) + 0.1
def mk(dt, lib=None):
    oc, env, d = models.robotarm(n_grid=50)
    if lib: oc.use_library(lib)
    oc.setDevice("cuda:0", dt); oc.setSolverOptions(aux_substeps=8)
    return oc, d
oc64, d = mk(torch.float64)
x0 = np.tile(d["ini_state"], (B, 1))
s = oc64.cocSolverBatch(x0, d["horizon"], th0); a = oc64.auxSysSolverBatch(s, d["taus"], d["waypoints"], d["interface"])
th1 = th0 - d["lr"] * a["grad"].cpu().numpy(); th1[:, 0] = np.maximum(th1[:, 0], 1e-8)
s64 = oc64.cocSolverBatch(x0, d["horizon"], th1); a64 = oc64.auxSysSolverBatch(s64, d["taus"], d["waypoints"], d["interface"])
pick = [801, 467, 702, 241, 885, 933, 978, 861, 428, 192, 35, 518, 889]
spec = oc64.model_spec()
q = lambda v: "p50 %.2e p90 %.2e p99 %.2e max %.2e" % tuple(np.quantile(v, [0.5, 0.9, 0.99, 1.0]))
for tag in (sys.argv[1:] or ["product", "plain"]):
    lib = None if tag == "product" else (runtime.variant_library_path(spec, "plain") if tag == "plain" else oc_trace.variant_path(spec, tag))
    oc, _ = mk(torch.float32, lib)
    oc.cocSolverBatch(x0, d["horizon"], th1); torch.cuda.synchronize()
    t0 = time.perf_counter(); s32 = oc.cocSolverBatch(x0, d["horizon"], th1); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    a32 = oc.auxSysSolverBatch(s32, d["taus"], d["waypoints"], d["interface"])
    dx = ((s32["state_grid"].double() - s64["state_grid"]).abs().flatten(1).max(1)[0] / s64["state_grid"].abs().flatten(1).max(1)[0]).cpu().numpy()
    g32, g64 = a32["grad"].double().cpu().numpy(), a64["grad"].cpu().numpy()
    dg = np.abs(g32 - g64).max(1) / np.abs(g64).max(1)
    it = s32["iters"].cpu().numpy(); st = s32["status"].cpu().numpy()
    print("%-8s solve %.2f ms status %s iters mean %.1f max %d | state err %s | grad err %s | share grad err > 2e-2: %.4f" %
          (tag, ms, np.bincount(st, minlength=5).tolist(), it.mean(), it.max(), q(dx), q(dg), (dg > 2e-2).mean()))
    print("         tier seeds: dx " + " ".join("%.1e" % dx[b] for b in pick))
    print("                     dg " + " ".join("%.1e" % dg[b] for b in pick))
