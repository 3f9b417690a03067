"""Iteration trace (-DLFSD_TRACE, wide mapping) of single robot-arm solves at the configs[1] seeds (fp32)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfsd_amd  # noqa: F401
from lfsd_amd import models, runtime
from arm_steps import seeds


def variant_path(spec):
    return os.path.join(runtime.BUILD_DIR, "trace_%s_oc.so" % spec.hash())


def build():
    oc, env, d = models.ZOO["robotarm"](n_grid=50)
    spec = oc.model_spec(); runtime.write_header(spec)
    out = variant_path(spec)
    runtime.build_checked(spec, out, ["-DLFSD_TRACE"])      # (assembly-checked like the product build: lfsd_amd/isa_check.py)
    print(out)


def run():
    import torch
    oc, env, d = models.ZOO["robotarm"](n_grid=50)
    oc.use_library(variant_path(oc.model_spec())); oc.setDevice("cuda:0", torch.float32)
    th = seeds(1024)
    for j in (0, 1, 2):
        print("=== seed %d theta %s" % (j, th[j].tolist()), flush=True)
        sol = oc.cocSolverBatch([d["ini_state"]], d["horizon"], th[j:j + 1]); torch.cuda.synchronize()
        print("=== iterations %d status %d" % (int(sol["iters"][0]), int(sol["status"][0])), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
