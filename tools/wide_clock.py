"""Shader-clock split of the WIDE OC kernel (roll-outs of the 16 step lengths | linearisation | costates | exact stage
Hessians | backward recursion) for the solves that take at least `min_iters` iterations, any model of the zoo.

    python tools/wide_clock.py build <model> <min_iters> [extra hipcc flags...]      (no GPU needed)
    python tools/wide_clock.py run <model> <n_grid> <batch> <f32|f64>
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oc_trace

if __name__ == "__main__":
    mode, kind = sys.argv[1], sys.argv[2]
    if mode == "build":
        import subprocess
        from lfsd_amd import models, runtime
        oc, env, d = models.ZOO[kind]()
        spec = oc.model_spec(); runtime.write_header(spec)
        out = oc_trace.variant_path(spec, "wclock")
        runtime.build_checked(spec, out, ["-DLFSD_OC_CLOCK=%d" % int(sys.argv[3])] + sys.argv[4:])      # (assembly-checked like the product build: lfsd_amd/isa_check.py)
        print(out)
    else:
        from lfsd_amd import models
        oc, env, d = models.ZOO[kind]()
        oc_trace.run(kind, int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], 0, library=oc_trace.variant_path(oc.model_spec(), "wclock"))
