"""Build kernel variants (register budget / lanes per trajectory) and time them on the bench workload."""
import os, re, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models, runtime

def build(spec, tag, extra):
    out = os.path.join(runtime.BUILD_DIR, "tune_%s_%s.so" % (spec.hash(), tag))
    cmd = runtime.hipcc_command(spec, out, list(extra) + ["-Rpass-analysis=kernel-resource-usage"])
    r = subprocess.run(cmd, cwd=runtime.CSRC_DIR, capture_output=True, text=True)
    if r.returncode != 0:
        print("  build failed:", tag, r.stderr[-300:].replace("\n", " "))
        return None
    # one line per kernel: registers, spills, LDS, occupancy
    cur = None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark: [^:]*:?\s*(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = m.group(2)
            if "lfsd" in cur and ("aux_" in cur or "oc_solve" in cur) and "Id" not in cur.split("Model")[-1][:3]:
                print("  ", re.sub(r"IN\d+lfsd_gen_[0-9a-f]+5ModelE", "", cur)[:48], end=" ")
            else:
                cur = None
        elif cur:
            print(m.group(1).split(" ")[0], m.group(2), end="  " if not m.group(1).startswith("LDS") else "\n")
    return out

if __name__ == "__main__":
    variants = [("base", []), ("noslp", ["-fno-slp-vectorize"]), ("O2", ["-O2"]),
                ("nomisched", ["-mllvm", "-enable-misched=0"]), ("relaxocc", ["-mllvm", "-amdgpu-schedule-relaxed-occupancy=1"])]
    if sys.argv[1:] == ["build"]:
        oc, env, d = models.quadrotor(n_grid=50)
        spec = oc.model_spec(); runtime.write_header(spec)
        for tag, extra in variants:
            print(build(spec, tag, extra))
        sys.exit(0)
    B, N = 4096, 50
    for dt in (torch.float32,):
        for tag, _ in variants:
            oc, env, d = models.quadrotor(n_grid=N)
            spec = oc.model_spec()
            oc.use_library(os.path.join(runtime.BUILD_DIR, "tune_%s_%s.so" % (spec.hash(), tag)))
            oc.setDevice("cuda:0", dt)
            rng = np.random.default_rng(1234)
            th = np.array(d["theta0"])[None, :] + 0.05 * rng.standard_normal((B, 7)); th[:, 0] = np.abs(th[:, 0]) + 0.5
            x0 = np.tile(d["ini_state"], (B, 1))
            ts = []
            for rep in range(3):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                ev[0].record()
                sol = oc.cocSolverBatch(x0, 1.0, th)
                hook = lambda nm: {"riccati": ev[1], "forward": ev[2], "end": ev[3]}[nm].record()
                aux = oc.auxSysSolverBatch(sol, d["taus"], d["waypoints"], d["interface"], phase_hook=hook)
                torch.cuda.synchronize()
                ts.append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
            t = np.array(ts)[1:].mean(0)
            print(tag, str(dt), "oc %.2f ms  riccati %.2f ms  forward %.2f ms  total %.2f" % (t[0], t[1], t[2], t.sum()),
                  "iters %.2f" % sol["iters"].float().mean().item(), "loss", aux["loss"].mean().item(), flush=True)
