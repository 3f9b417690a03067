"""Iteration trace (-DLFSD_TRACE, wide mapping) of the slowest trajectories of BASELINE configs[4]'s solve (rocket,
n_grid 100, 1024 perturbed initial guesses).  `build` compiles the trace variant (no GPU needed)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfsd_amd  # noqa: F401
from lfsd_amd import models, runtime

N_GRID = 100


def variant_path(spec):
    return os.path.join(runtime.BUILD_DIR, "trace_%s_oc.so" % spec.hash())


def build():
    oc, env, d = models.ZOO["rocket"](n_grid=N_GRID)
    spec = oc.model_spec(); runtime.write_header(spec)
    out = variant_path(spec)
    cmds, objs = runtime.hipcc_commands(spec, out, ["-DLFSD_TRACE"])
    for c in cmds:
        r = subprocess.run(c, cwd=runtime.CSRC_DIR, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    for o in objs:
        os.remove(o)
    print(out)


def run():
    import torch
    B = 1024
    oc, env, d = models.ZOO["rocket"](n_grid=N_GRID)
    oc.setDevice("cuda:0", torch.float32)
    rng = np.random.default_rng(0)
    p = oc.compile().n_auxvar
    th = np.array(d["theta0"])[None, :] * (1 + 0.05 * rng.standard_normal((B, p)))
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    x0 = np.tile(d["ini_state"], (B, 1))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    oc.cocSolverBatch(x0, d["horizon"], th); a.record(); sol = oc.cocSolverBatch(x0, d["horizon"], th); b.record(); torch.cuda.synchronize()
    it = sol["iters"].cpu().numpy(); st = sol["status"].cpu().numpy()
    print("solve %.1f ms, status %s, iterations mean %.1f, percentiles 50/90/99/100: %s" %
          (a.elapsed_time(b), np.bincount(st, minlength=5).tolist(), it.mean(), np.percentile(it, [50, 90, 99, 100]).tolist()))
    slow = np.argsort(-it)[:2]
    oc2, _, _ = models.ZOO["rocket"](n_grid=N_GRID)
    oc2.use_library(variant_path(oc2.model_spec())); oc2.setDevice("cuda:0", torch.float32)
    for j in slow:
        print("=== trajectory %d (%d iterations, status %d), theta %s" % (j, it[j], st[j], th[j].tolist()), flush=True)
        s2 = oc2.cocSolverBatch(x0[j:j + 1], d["horizon"], th[j:j + 1]); torch.cuda.synchronize()
        print("=== iterations %d status %d" % (int(s2["iters"][0]), int(s2["status"][0])), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
