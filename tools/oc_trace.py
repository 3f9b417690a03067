"""Iteration trace (-DLFSD_TRACE) of the slowest OC solves of any model of the zoo.

    python tools/oc_trace.py build <model> [extra hipcc flags...]      (no GPU needed)
    python tools/oc_trace.py run <model> <n_grid> <batch> <f32|f64> [n_slowest]

`run` solves `batch` perturbed seeds with the product library (iteration histogram, kernel time), then re-solves the
slowest ones one at a time (batch 1: trajectory 0 is the one the kernels trace) with the trace variant."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lfsd_amd  # noqa: F401
from lfsd_amd import models, runtime


def variant_path(spec, tag="trace"):
    return os.path.join(runtime.BUILD_DIR, "trace_%s_%s.so" % (spec.hash(), tag))


def build(kind, extra, tag="trace"):
    oc, env, d = models.ZOO[kind]()
    spec = oc.model_spec(); runtime.write_header(spec)
    out = variant_path(spec, tag)
    runtime.build_checked(spec, out, ["-DLFSD_TRACE"] + list(extra))      # (assembly-checked like the product build: lfsd_amd/isa_check.py)
    print(out)


def seeds(d, p, B):
    rng = np.random.default_rng(0)
    th = np.array(d["theta0"])[None, :] * (1 + 0.05 * rng.standard_normal((B, p)))
    th[:, 0] = np.abs(th[:, 0]) + 0.1
    return th


def run(kind, n_grid, B, dt, n_slow=2, tag="trace", library=None):
    import torch
    dtype = torch.float32 if dt == "f32" else torch.float64
    oc, env, d = models.ZOO[kind](n_grid=n_grid)
    if library:
        oc.use_library(library)
    oc.setDevice("cuda:0", dtype)
    p = oc.compile().n_auxvar
    th = seeds(d, p, B)
    x0 = np.tile(d["ini_state"], (B, 1))
    oc.cocSolverBatch(x0, d["horizon"], th); torch.cuda.synchronize()
    t0 = time.perf_counter(); sol = oc.cocSolverBatch(x0, d["horizon"], th); torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    it, st = sol["iters"].cpu().numpy(), sol["status"].cpu().numpy()
    print("%s n_grid %d batch %d %s: %.2f ms, status %s, iterations mean %.1f median %d p90 %d max %d" %
          (kind, n_grid, B, dt, ms, np.bincount(st, minlength=5).tolist(), it.mean(), np.median(it), np.quantile(it, 0.9), it.max()), flush=True)
    if n_slow <= 0:
        return
    oc2, _, _ = models.ZOO[kind](n_grid=n_grid)
    oc2.use_library(variant_path(oc2.model_spec(), tag)); oc2.setDevice("cuda:0", dtype)
    oc2.setSolverOptions(mapping=oc.mapping if oc.mapping != "auto" else ("wide" if oc.exact_after == 0 else "lockstep"))
    pick = [int(v) for v in os.environ["LFSD_TRACE_IDX"].split(",")] if os.environ.get("LFSD_TRACE_IDX") else np.argsort(-it)[:n_slow]
    for j in pick:
        print("=== trajectory %d (%d iterations in the batch), theta %s" % (j, it[j], np.array2string(th[j], precision=4)), flush=True)
        s1 = oc2.cocSolverBatch(x0[j:j + 1], d["horizon"], th[j:j + 1]); torch.cuda.synchronize()
        print("=== iterations %d status %d cost %.8g" % (int(s1["iters"][0]), int(s1["status"][0]), float(s1["cost"][0])), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        run(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]) if len(sys.argv) > 6 else 2)
