"""Accuracy of the auxiliary sweeps vs their `aux_substeps` knob on the headline configuration (quadrotor, n_grid 50, the first
bench seeds): gradient / loss error against the TIGHT oracle (Radau, rtol 1e-10) and against the reference-mode oracle
(solve_ivp BDF + RK45 at scipy's default rtol 1e-3, i.e. what CPDP.py:335,368 computes), with the aux kernel times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models
from conftest import oracle_parallel

oc, env, d = models.quadrotor(n_grid=50)
rng = np.random.default_rng(1234)
th = np.array(d["theta0"])[None, :] + 0.05 * rng.standard_normal((4096, 7)); th[:, 0] = np.abs(th[:, 0]) + 0.5
K = 8
jobs = [dict(kind="quadrotor", n_grid=50, ini_state=d["ini_state"], horizon=d["horizon"], theta=list(th[b]), taus=d["taus"],
             wps=d["waypoints"], iface=d["interface"], tight=t) for t in (True, False) for b in range(K)]
refs = oracle_parallel(jobs)
tight, loose = refs[:K], refs[K:]
gerr = lambda g, r: np.abs(g - r["grad"]).max() / np.abs(r["grad"]).max()
print("reference-mode oracle (the reference's own integrator settings) vs tight oracle: gradient error max %.2e" %
      max(gerr(loose[b]["grad"], tight[b]) for b in range(K)))
x0 = np.tile(d["ini_state"], (4096, 1))
for dt in (torch.float64, torch.float32):
    oc.setDevice("cuda:0", dt)
    sol = oc.cocSolverBatch(x0, d["horizon"], th)
    for sub, rtol in ((1, 0.0), (2, 0.0), (4, 0.0), (8, 0.0), (0, 1e-3), (0, 1e-4), (0, 1e-5), (0, 1e-6)):
        oc.setSolverOptions(aux_substeps=sub, aux_rtol=rtol)
        aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"]); torch.cuda.synchronize()
        t0 = time.perf_counter(); aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"]); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        g = aux["grad"][:K].double().cpu().numpy(); l = aux["loss"][:K].double().cpu().numpy()
        what = ("fixed %d units" % sub) if rtol == 0 else ("error-controlled rtol %.0e" % rtol)
        print("%s %-28s: aux pass %6.2f ms (batch 4096) | gradient error vs tight oracle max %.2e | vs reference-mode oracle max %.2e | loss error %.1e" %
              (str(dt)[6:], what, ms, max(gerr(g[b], tight[b]) for b in range(K)), max(gerr(g[b], loose[b]) for b in range(K)),
               max(abs(l[b] - tight[b]["loss"]) / tight[b]["loss"] for b in range(K))), flush=True)
