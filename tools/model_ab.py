"""A/B timing of build variants of ANY model library on one cold-started OC solve of `batch` perturbed seeds (the quadrotor's
headline A/B is tools/ab_variants.py).

    python tools/model_ab.py build <model> <tag> [extra hipcc flags...]      (no GPU needed; csrc/build/trace_<hash>_<tag>.so)
    python tools/model_ab.py run <model> <n_grid> <batch> <f32|f64> [tag ...]   ("product" = the shipped library)
"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oc_trace

if __name__ == "__main__":
    mode, kind = sys.argv[1], sys.argv[2]
    from lfsd_amd import models, runtime
    if mode == "build":
        oc, env, d = models.ZOO[kind]()
        spec = oc.model_spec(); runtime.write_header(spec)
        out = oc_trace.variant_path(spec, sys.argv[3])
        runtime.build_checked(spec, out, sys.argv[4:])      # (assembly-checked like the product build: lfsd_amd/isa_check.py)
        print(out)
    else:
        n_grid, B, dt = int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
        oc, env, d = models.ZOO[kind]()
        for tag in (sys.argv[6:] or ["product"]):
            print("--- %s" % tag, flush=True)
            oc_trace.run(kind, n_grid, B, dt, 0, library=None if tag == "product" else oc_trace.variant_path(oc.model_spec(), tag))
