"""Step-control variants of the OC solve on BASELINE configs[1] (robot arm, 1024 seeds) at the step-1 parameters
theta1 = theta0 - lr*grad0 (usage: tune_arm.py build | run).  Prints status histogram, iterations and kernel time."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models, runtime
FINE = ["-DLFSD_MU_UP=3.1623", "-DLFSD_MU_DOWN=0.31623", "-DLFSD_MU_HOLD_BACKOFF=0"]
# (round 6: the LFSD_GN_CRAWL switch of this sweep is gone from csrc/ -- shipped on since round 1; the crawl_* rows now equal the fine* rows)
VARIANTS = [("base", ["-DLFSD_MU_HOLD=0", "-DLFSD_MU_UP=10", "-DLFSD_MU_DOWN=0.1"]),   # the rule before this sweep
            ("fine0", ["-DLFSD_MU_HOLD=0"] + FINE),
            ("fine1", ["-DLFSD_MU_HOLD=1"] + FINE),
            ("crawl_fine0", ["-DLFSD_MU_HOLD=0"] + FINE),
            ("crawl_fine1", ["-DLFSD_MU_HOLD=1"] + FINE)]
def lib(kind, tag):
    oc, _, _ = models.ZOO[kind]()
    return os.path.join(runtime.BUILD_DIR, "tune_%s_%s.so" % (oc.model_spec().hash(), tag))
if sys.argv[1] == "build":
    for kind in ("robotarm", "quadrotor"):
        oc, _, _ = models.ZOO[kind]()
        spec = oc.model_spec(); runtime.write_header(spec)
        for tag, extra in VARIANTS:
            r = subprocess.run(runtime.hipcc_command(spec, lib(kind, tag), extra), cwd=runtime.CSRC_DIR, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            print("built", kind, tag, flush=True)
    sys.exit(0)
B = 1024
rng = np.random.default_rng(0)
oc, env, d = models.ZOO["robotarm"](n_grid=50)
th0 = np.array(d["theta0"])[None, :] * (1 + 0.05 * rng.standard_normal((B, 5)))
th0[:, 0] = np.abs(th0[:, 0]) + 0.1
x0 = np.tile(d["ini_state"], (B, 1))
th1 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "arm_theta1.npy"))   # theta0 - lr*grad0 (fp64)
for tag, _ in VARIANTS:
    for dt in (torch.float32, torch.float64):
        oc, env, d = models.ZOO["robotarm"](n_grid=50)
        oc.use_library(lib("robotarm", tag)); oc.setDevice("cuda:0", dt)
        for th, nm in ((th0, "theta0"), (th1, "theta1")):
            oc.cocSolverBatch(x0, d["horizon"], th); torch.cuda.synchronize()
            t0 = time.perf_counter(); s = oc.cocSolverBatch(x0, d["horizon"], th); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
            a = oc.auxSysSolverBatch(s, d["taus"], d["waypoints"], d["interface"])
            st = s["status"].cpu().numpy(); it = s["iters"].cpu().numpy(); g = a["grad"].double().cpu().numpy()
            print("%-8s %s %s: %7.1f ms status %s iters mean %.1f max %d  cost mean %.6f  |grad| max %.3g" %
                  (tag, str(dt)[6:], nm, ms, np.bincount(st, minlength=5).tolist(), it.mean(), it.max(), s["cost"].double().mean().item(), np.abs(g).max()), flush=True)
# the headline workload must not change: quadrotor bench seeds, first solve
oc, env, d = models.quadrotor(n_grid=50)
rng = np.random.default_rng(1234)
thq = np.array(d["theta0"])[None, :] + 0.05 * rng.standard_normal((4096, 7)); thq[:, 0] = np.abs(thq[:, 0]) + 0.5
for tag, _ in VARIANTS:
    oc, env, d = models.quadrotor(n_grid=50)
    oc.use_library(lib("quadrotor", tag)); oc.setDevice("cuda:0", torch.float32)
    xq = np.tile(d["ini_state"], (4096, 1))
    oc.cocSolverBatch(xq, d["horizon"], thq); torch.cuda.synchronize()
    t0 = time.perf_counter(); s = oc.cocSolverBatch(xq, d["horizon"], thq); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    print("%-8s quadrotor f32: %6.2f ms status %s iters mean %.2f cost mean %.6f" % (tag, ms, np.bincount(s["status"].cpu().numpy(), minlength=5).tolist(), s["iters"].float().mean().item(), s["cost"].double().mean().item()), flush=True)
