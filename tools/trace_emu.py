"""CPU-side debugging aid (test infrastructure): build the SIMT-emulator library of a model with the kernel's
per-iteration trace enabled and solve single trajectories.  usage: trace_emu.py <model> <theta.npy> <seed> [f32|f64]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models, runtime
import conftest
kind, thf, seed = sys.argv[1], sys.argv[2], int(sys.argv[3])
dt = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.float64
oc, env, d = models.ZOO[kind](n_grid=50)
spec = oc.model_spec(); runtime.write_header(spec)
out = os.path.join(conftest.EMU_BUILD, "liblfsd_%s_emu_trace.so" % spec.hash())
g = runtime.lanes_for(spec.n, spec.m, spec.p)
cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-DLFSD_EMU", "-DLFSD_TRACE", "-fvisibility=hidden", "-DLFSD_G=%d" % g,
       '-DLFSD_MODEL_HEADER="gen/%s.h"' % spec.hash(), "-I" + conftest.EMU_DIR, "-I" + runtime.CSRC_DIR,
       os.path.join(runtime.CSRC_DIR, "lfsd_capi.cpp"), "-o", out]
os.makedirs(conftest.EMU_BUILD, exist_ok=True)
r = subprocess.run(cmd, cwd=runtime.CSRC_DIR, capture_output=True, text=True); assert r.returncode == 0, r.stderr[-2000:]
oc.use_library(out); oc.setDevice(dtype=dt)
if len(sys.argv) > 5: oc.setSolverOptions(exact_after=int(sys.argv[5]))
th = np.load(thf)[seed:seed + 1]
sol = oc.cocSolverBatch(np.asarray(d["ini_state"])[None, :], d["horizon"], th)
print("status", sol["status"].tolist(), "iters", sol["iters"].tolist(), "cost %.10f" % sol["cost"][0].item())
