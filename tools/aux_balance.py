"""Load balance of the three kernels of the headline step: per-trajectory work (OC iterations; split units of the two
auxiliary sweeps, from their statistics output) against what a wavefront / a SIMD has to wait for.

    python tools/aux_balance.py [steps]        (GPU; the benchmark's learner, 4096 quadrotor seeds)
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lfsd_amd  # noqa: F401
from lfsd_amd import models
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
args = bench.parse_args(["--no-cpu-baseline"])
oc, env, d = models.quadrotor(n_grid=args.n_grid)
oc.setDevice("cuda:0", torch.float32)
L, theta0, x0 = bench.build_learner(args, oc, d, oc.compile(), 0, 1, "independent")
for s in range(steps):
    L.step()
torch.cuda.synchronize()
st = L._aux["stats"].double().cpu().numpy()
it = L._sol["iters"].cpu().numpy().astype(float)


def report(name, w, per_wave, waves_per_simd):
    B = len(w)
    wave = w.reshape(B // per_wave, per_wave).max(axis=1)          # lock-step partners wait for the slowest
    nw = len(wave)
    print("%-12s per trajectory: mean %.1f p50 %.0f p90 %.0f p99 %.0f max %.0f | per wavefront (max of %d): mean %.1f max %.0f -> mean/max %.2f"
          % (name, w.mean(), np.median(w), np.quantile(w, .9), np.quantile(w, .99), w.max(), per_wave, wave.mean(), wave.max(), wave.mean() / wave.max()))
    if waves_per_simd == 2:
        simd = wave[:nw // 2] + wave[nw // 2:]                      # wave i and i + 1024 share a SIMD if dispatch fills slot by slot
        srt = np.sort(wave)[::-1]
        best = srt[:nw // 2] + srt[nw // 2:][::-1]
        print("%-12s per SIMD (2 waves, work adds up): as dispatched mean %.1f max %.0f (mean/max %.2f); heavy paired with light: max %.0f (mean/max %.2f)"
              % ("", simd.mean(), simd.max(), simd.mean() / simd.max(), best.max(), best.mean() / best.max()))


report("oc iters", it, 4, 1)
report("ric units", st[:, 0], 2, 2)
report("fwd units", st[:, 2], 4, 1)
# how stable is a trajectory's work from one outer iteration to the next (could the previous step's statistics order the next?)
prev = st.copy()
L.step(); torch.cuda.synchronize()
st2 = L._aux["stats"].double().cpu().numpy()
for c, n in ((0, "ric"), (2, "fwd")):
    print("%s units, step k vs k+1: correlation %.3f, mean |diff| %.2f" % (n, np.corrcoef(prev[:, c], st2[:, c])[0, 1], np.abs(prev[:, c] - st2[:, c]).mean()))
