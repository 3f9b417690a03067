"""A/B timing of build variants of the quadrotor library on the headline workload (bench.py, 4096 seeds).

    python tools/ab_variants.py build [name ...]     no GPU needed: csrc/build/ab_<hash>_<name>.so (they travel with gpurun)
    python tools/ab_variants.py run [name ...] [--steps K] [--batch B ...]

Each variant is a set of extra hipcc flags (VARIANTS below; `base` = the product build).  `run` starts `bench.py
--library <variant> --no-cpu-baseline` as a fresh child per (variant, batch) -- a DVFS / first-launch effect of one run
cannot leak into the next -- and prints the per-kernel HIP-event times of each.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# name -> (extra flags for both units, capi-unit flags or None for the tuned default, drop -fno-slp-vectorize from the capi unit)
VARIANTS = {
    "base": ((), None, False),
    "noldssync": (("-DLFSD_OC_LDS_SYNC=0",), None, False),
    "pf1": (("-DLFSD_BW_PREFETCH=1",), None, False),
    "vupfetch": (("-DLFSD_SC_VUP_FETCH=1",), None, False),
    "synclight": (("-DLFSD_SYNC_LIGHT=1",), None, False),
    "gramrows": (("-DLFSD_RIC_GRAM_ROWS=1",), None, False),
    "synclight_gram": (("-DLFSD_SYNC_LIGHT=1", "-DLFSD_RIC_GRAM_ROWS=1"), None, False),
    "noprefetch": (("-DLFSD_BW_PREFETCH=0",), None, False),
    "slp": ((), None, True),
    "nomfma": (("-DLFSD_MFMA_BACKWARD=0",), None, False),
    "ricmfma": (("-DLFSD_RIC_MFMA=1",), None, False),
    "dual": (("-DLFSD_OC_DUAL=1",), None, False),
    "nodual": (("-DLFSD_OC_DUAL=0",), None, False),
    "nostruct": (("-DLFSD_STRUCT_COLS=0",), None, False),
    "nocoarse": (("-DLFSD_COARSE_START=0",), None, False),
    "nowidecoarse": (("-DLFSD_COARSE_MIN_GRID=100000",), None, False),
    "r02like": (("-DLFSD_COARSE_START=0", "-DLFSD_STRUCT_COLS=0"), None, False),
    "relinhard": (("-DLFSD_COARSE_RELIN=1",), None, False),
    "cs1e2": (("-DLFSD_COARSE_SWITCH=0.01",), None, False),
    "cs1e3": (("-DLFSD_COARSE_SWITCH=0.001",), None, False),
    "cs3": (("-DLFSD_COARSE_SWITCH=3.0",), None, False),
    "cs1e4": (("-DLFSD_COARSE_SWITCH=0.0001",), None, False),
    "cs0": (("-DLFSD_COARSE_SWITCH=0.0",), None, False),
    "bwclock": (("-DLFSD_BW_CLOCK=1",), None, False),
    "occlock": (("-DLFSD_OC_CLOCK=3",), None, False),
    "nofence64": (("-DLFSD_FENCE64=0",), None, False),
    "pin64only": (("-DLFSD_FENCE64=1",), None, False),
    "fence3": (("-DLFSD_FENCE64=3",), None, False),
    "nolive64": (("-DLFSD_FP64_LIVE=0",), None, False),
    "nopark64": (("-DLFSD_FP64_PARK=0",), None, False),
    "nosc64": (("-DLFSD_FP64_SC=0",), None, False),
    "vxlds": (("-DLFSD_SC_VX_LDS=1",), None, False),
    "pflate": (("-DLFSD_BW_PREFETCH=2",), None, False),
    "pflateclock": (("-DLFSD_BW_PREFETCH=2", "-DLFSD_BW_CLOCK=1"), None, False),
    # level 0 of the lean kernel's mesh continuation: tc merged control intervals x RK4 steps per merged interval x iterations
    "tc2k3": (("-DLFSD_LEAN_TC=2", "-DLFSD_LEAN_TC_ITERS=3"), None, False),
    "tc2k2": (("-DLFSD_LEAN_TC=2", "-DLFSD_LEAN_TC_ITERS=2"), None, False),
    "tc2k4": (("-DLFSD_LEAN_TC=2", "-DLFSD_LEAN_TC_ITERS=4"), None, False),
    "tc5k3": (("-DLFSD_LEAN_TC=5", "-DLFSD_LEAN_TC_S=2", "-DLFSD_LEAN_TC_ITERS=3"), None, False),
    "tc5s1k2": (("-DLFSD_LEAN_TC=5", "-DLFSD_LEAN_TC_S=1", "-DLFSD_LEAN_TC_ITERS=2"), None, False),
    "tc2s2k3": (("-DLFSD_LEAN_TC=2", "-DLFSD_LEAN_TC_S=2", "-DLFSD_LEAN_TC_ITERS=3"), None, False),
    "tc5k2": (("-DLFSD_LEAN_TC=5", "-DLFSD_LEAN_TC_S=2", "-DLFSD_LEAN_TC_ITERS=2"), None, False),
    "tc5k4": (("-DLFSD_LEAN_TC=5", "-DLFSD_LEAN_TC_S=2", "-DLFSD_LEAN_TC_ITERS=4"), None, False),
    "tc5k5": (("-DLFSD_LEAN_TC=5", "-DLFSD_LEAN_TC_S=2", "-DLFSD_LEAN_TC_ITERS=5"), None, False),
    "tc5s3k3": (("-DLFSD_LEAN_TC=5", "-DLFSD_LEAN_TC_S=3", "-DLFSD_LEAN_TC_ITERS=3"), None, False),
    "tc10s4k3": (("-DLFSD_LEAN_TC=10", "-DLFSD_LEAN_TC_S=4", "-DLFSD_LEAN_TC_ITERS=3", "-DLFSD_LEAN_TC_MIN=5"), None, False),
    "tc10s2k3": (("-DLFSD_LEAN_TC=10", "-DLFSD_LEAN_TC_S=2", "-DLFSD_LEAN_TC_ITERS=3", "-DLFSD_LEAN_TC_MIN=5"), None, False),
    "tc5k4fine": (("-DLFSD_LEAN_TC=5", "-DLFSD_LEAN_TC_S=2", "-DLFSD_LEAN_TC_ITERS=4", "-DLFSD_LEAN_TC_TO_FINE=1"), None, False),
    "tc5k5fine": (("-DLFSD_LEAN_TC=5", "-DLFSD_LEAN_TC_S=2", "-DLFSD_LEAN_TC_ITERS=5", "-DLFSD_LEAN_TC_TO_FINE=1"), None, False),
    # on top of the shipped level 0 (5 x 2 x 3 iterations)
    "tc5k3fine": (("-DLFSD_LEAN_TC_TO_FINE=1",), None, False),
    "ham3": (("-DLFSD_HAM_SWITCH=3.0",), None, False),
    "ham09": (("-DLFSD_HAM_SWITCH=0.9",), None, False),
    "ham01": (("-DLFSD_HAM_SWITCH=0.1",), None, False),
    "tc5s2k3min5": (("-DLFSD_LEAN_TC_MIN=5",), None, False),
    "notc": (("-DLFSD_LEAN_TC=1",), None, False),
    "nokeephist": (("-DLFSD_EXIT_KEEP_HISTORY=0",), None, False),
    "keephist1": (("-DLFSD_EXIT_KEEP_HISTORY=1",), None, False),
    "tc2k3b": (("-DLFSD_LEAN_TC=2", "-DLFSD_LEAN_TC_S=1", "-DLFSD_LEAN_TC_ITERS=3"), None, False),
    "tc5k2b": (("-DLFSD_LEAN_TC_ITERS=2",), None, False),
    "tc5s3": (("-DLFSD_LEAN_TC_S=3",), None, False),
    "occ_k3": (("-DLFSD_OC_CLOCK=1024",), None, False),
    "occ_k2": (("-DLFSD_OC_CLOCK=1024", "-DLFSD_LEAN_TC_ITERS=2"), None, False),
    "grace1": (("-DLFSD_LEAN_TC_GRACE=1",), None, False),
    "grace2": (("-DLFSD_LEAN_TC_GRACE=2",), None, False),
    "grace1s1": (("-DLFSD_LEAN_TC_GRACE=1", "-DLFSD_LEAN_TC_S=1"), None, False),
    "tc2k2b": (("-DLFSD_LEAN_TC=2", "-DLFSD_LEAN_TC_S=1", "-DLFSD_LEAN_TC_ITERS=2"), None, False),
    "tc2k4b": (("-DLFSD_LEAN_TC=2", "-DLFSD_LEAN_TC_S=1", "-DLFSD_LEAN_TC_ITERS=4"), None, False),
    "tc5s1k2b": (("-DLFSD_LEAN_TC_S=1", "-DLFSD_LEAN_TC_ITERS=2"), None, False),
    "tc5s1k3b": (("-DLFSD_LEAN_TC_S=1", "-DLFSD_LEAN_TC_ITERS=3"), None, False),
    "k4ham01": (("-DLFSD_LEAN_TC_ITERS=4", "-DLFSD_HAM_SWITCH=0.05"), None, False),
}


def variant_path(spec, name):
    from lfsd_amd import runtime
    return os.path.join(runtime.BUILD_DIR, "ab_%s_%s.so" % (spec.hash(), name))


def build(names):
    import lfsd_amd  # noqa: F401
    from lfsd_amd import models, runtime
    oc, env, d = models.quadrotor(n_grid=50)
    spec = oc.model_spec()
    runtime.write_header(spec)
    os.makedirs(runtime.BUILD_DIR, exist_ok=True)
    for name in names:
        extra, capi, slp = VARIANTS[name]
        out = variant_path(spec, name)
        cmds, objs = runtime.hipcc_commands(spec, out, list(extra), **({} if capi is None else {"extra_capi": capi}))
        if slp:
            cmds[0] = [c for c in cmds[0] if c != "-fno-slp-vectorize"]
        try:
            for c in cmds:
                r = subprocess.run(c, cwd=runtime.CSRC_DIR, capture_output=True, text=True)
                assert r.returncode == 0, r.stderr[-3000:]
        finally:
            for o in objs:
                if os.path.exists(o):
                    os.remove(o)
        print("built", name, out, flush=True)


def run(names, steps, batches, extra_args):
    import lfsd_amd  # noqa: F401
    from lfsd_amd import models
    oc, env, d = models.quadrotor(n_grid=50)
    spec = oc.model_spec()
    for name in names:
        for B in batches:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", str(steps), "--warmup", "3",
                   "--batch", str(B)] + extra_args
            if name != "base":
                cmd += ["--library", variant_path(spec, name)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                print("%-12s batch %5d FAILED rc %d: %s" % (name, B, r.returncode, r.stderr[-400:]), flush=True)
                continue
            o = json.loads(line[-1])
            c = o["config"]
            print("%-12s batch %5d  %9.0f it/s  %7.3f ms/step  kernels %s  iters %.2f/%d  status %s" %
                  (name, B, o["value"], o["ms_per_step"], c["kernel_ms"], c["oc_iters_mean"], c["oc_iters_max"], c["oc_status_hist"]), flush=True)


if __name__ == "__main__":
    argv = sys.argv[1:]
    mode = argv.pop(0)
    steps, batches, names, extra = 20, [], [], []
    while argv:
        a = argv.pop(0)
        if a == "--steps":
            steps = int(argv.pop(0))
        elif a == "--batch":
            batches.append(int(argv.pop(0)))
        elif a == "--":
            extra = argv[:]
            argv = []
        else:
            names.append(a)
    names = names or ["base"]
    if mode == "build":
        build(names)
    else:
        run(names, steps, batches or [4096], extra)
