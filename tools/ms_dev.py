"""CPU emulator: the wide kernel with and without the multiple-shooting phase on a handful of seeds (development aid).
usage: python tools/ms_dev.py <robotarm|rocket|quadrotor|cartpole|pendulum> <n_grid> <n_seeds> [f32|f64] [trace]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lfsd_amd
from lfsd_amd import models, CPDP
from conftest import build_emu_library

kind, n_grid, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
dt = torch.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else torch.float32
trace = len(sys.argv) > 5 and sys.argv[5] == "trace"
off = int(sys.argv[6]) if len(sys.argv) > 6 else 0
rng = np.random.default_rng(1234)
res = {}
VAR = {"ms": (), "ss": ("-DLFSD_MS=0",), "clk": ("-DLFSD_OC_CLOCK=1000",)}
import os as _os
for tag in _os.environ.get("VARIANTS", "ms ss").split():
  flags = VAR[tag]
  if True:
    oc, env, d = models.ZOO[kind](n_grid=n_grid)
    fl = list(flags) + (["-DLFSD_TRACE"] if trace else ["-DLFSD_MS_STATS"])
    oc.use_library(build_emu_library(oc, extra_flags=fl, tag="dev_" + tag + ("_tr" if trace else "")))
    oc.compile()
    oc.setDevice(dtype=dt)
    oc.setSolverOptions(mapping="wide")
    p = len(d["theta0"])
    th = np.array(d["theta0"])[None, :] * (1 + 0.05 * np.random.default_rng(1234).standard_normal((B + off, p)))
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    th = th[off:]
    t0 = time.time()
    sol = oc.cocSolverBatch(np.tile(d["ini_state"], (B, 1)), d["horizon"], th)
    res[tag] = sol
    print(tag, "time %.1fs" % (time.time() - t0), "iters", sol["iters"].tolist(), "status", sol["status"].tolist())
    print("   cost", ["%.8g" % v for v in sol["cost"].tolist()])
tags = list(res)
for t in tags[1:]:
    print("max |dx| %s vs %s:" % (tags[0], t), (res[tags[0]]["state_grid"] - res[t]["state_grid"]).abs().amax(dim=(1, 2)).tolist())
