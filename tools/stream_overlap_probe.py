"""Do the launches of independent sub-batches overlap usefully?  Every kernel of this path lasts as long as its slowest wavefront
(one wavefront per SIMD); the headline batch split into C chunks, each with its own learner on its own HIP stream, lets the
wavefronts of one chunk's kernel run on the SIMDs another chunk's kernel has already left.

    python tools/stream_overlap_probe.py [chunks ...]      (default 1 2 4)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import lfsd_amd  # noqa: F401
from lfsd_amd import models, CPDP

chunks = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
args = bench.parse_args(["--no-cpu-baseline"])
w = bench.WORKLOADS["quadrotor"]
oc, env, d = models.quadrotor(n_grid=args.n_grid)
oc.setDevice("cuda:0", torch.float32)
lib = oc.compile()
d = dict(d); d["taus"], d["waypoints"] = bench.demonstration(oc, d, args.n_grid)
demos = bench.demo_set(args, d, 0, "independent", w, lib.n_auxvar)
B = args.batch
for C in chunks:
    streams = [torch.cuda.Stream() for _ in range(C)]
    Ls = []
    for c in range(C):
        sl = slice(c * B // C, (c + 1) * B // C)
        with torch.cuda.stream(streams[c]):
            L = CPDP.SparseDemoLearner(oc, demos["x0"][sl], d["horizon"], d["taus"], d["waypoints"], d["interface"], demos["theta0"][sl],
                                       method=w["method"], learning_rate=w["lr"], mu=0.9)
            L.count_unconverged = False
        Ls.append(L)

    def step():
        for c in range(C):
            with torch.cuda.stream(streams[c]):
                Ls[c].step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    loss = float(np.mean([float(torch.nanmean(L._aux["loss"].double()).item()) for L in Ls]))
    print("%d chunk(s) of %d on %d stream(s): %.3f ms per step of the whole batch, %.0f it/s, mean loss %.6f" % (C, B // C, C, el / 10 * 1e3, B * 10 / el, loss), flush=True)
