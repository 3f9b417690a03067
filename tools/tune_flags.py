"""Per-translation-unit compiler-flag variants of the quadrotor library (product-style split build).
usage: tune_flags.py build   then   bench.py --no-cpu-baseline --library csrc/build/tune_<hash>_<tag>.so"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lfsd_amd
from lfsd_amd import models, runtime
# tag, flags for lfsd_capi.cpp (everything but the Riccati sweep), flags for lfsd_riccati.cpp
T1, T2 = list(runtime.TUNED_CAPI), list(runtime.TUNED_RICCATI)
NS = ["-fno-slp-vectorize"]
R2 = ["-DLFSD_RIC_CACHE=2", "-DLFSD_WAVES_RIC=2"]
ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
VARIANTS = [("r2_ns", T1, NS + R2),
            ("r2_slp4", T1, T2 + R2),
            ("r2_ns_ilp", T1, NS + ILP + R2),
            ("r2_ns_unr", T1, NS + R2 + ['-DLFSD_RIC_NODE_LOOP=_Pragma("unroll")']),
            ("r2_ns_memcl", T1, NS + ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"] + R2)]
oc, _, _ = models.quadrotor()
spec = oc.model_spec(); runtime.write_header(spec)
for tag, e1, e2 in VARIANTS:
    out = os.path.join(runtime.BUILD_DIR, "tune_%s_%s.so" % (spec.hash(), tag))
    cmds, objs = runtime.hipcc_commands(spec, out, ["-Rpass-analysis=kernel-resource-usage"], e1, e2)
    ok = True
    for cmd in cmds:
        r = subprocess.run(cmd, cwd=runtime.CSRC_DIR, capture_output=True, text=True)
        if r.returncode != 0:
            print("build failed:", tag, r.stderr[-200:].replace("\n", " ")); ok = False; break
        name = None
        for ln in r.stderr.splitlines():
            m = re.search(r"remark: [^:]*:?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]): (\S+)", ln)
            if not m: continue
            if m.group(1) == "Function Name":
                name = m.group(2) if re.search(r"(oc_solve_kernel.*fLi32ELb0ELb1|aux_riccati_kernel.*fLi32|aux_forward_kernel.*fLi16)", m.group(2)) else None
                if name: print("   ", tag, re.sub(r"IN\d+lfsd_gen_[0-9a-f]+5ModelE", "", name)[9:34], end=" ")
            elif name:
                print(m.group(1).split(" ")[0], m.group(2), end="  " if not m.group(1).startswith("Scratch") else "\n")
    for o in objs:
        if os.path.exists(o): os.remove(o)
    if ok: print("built", tag, flush=True)
