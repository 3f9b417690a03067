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
    # round 5 (the switches of rounds 3-4 whose branches were measured negative are gone from csrc/; their results: profiles/HISTORY.md)
    "nocoarse": (("-DLFSD_COARSE_START=0",), None, False),
    "notc": (("-DLFSD_LEAN_TC=1",), None, False),
    "slp": ((), None, True),
    "nomfma": (("-DLFSD_MFMA_BACKWARD=0",), None, False),
    "nostruct": (("-DLFSD_STRUCT_COLS=0",), None, False),
    # round 6 (verdict item 4): the OC kernels held to 256 registers, two wavefronts per SIMD
    "waves2": (("-DLFSD_WAVES_OC=2",), None, False),
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
        if capi is None and not slp:      # (the common case: assembly-checked like the product build, lfsd_amd/isa_check.py)
            runtime.build_checked(spec, out, list(extra))
            print("built", name, out, flush=True)
            continue
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
