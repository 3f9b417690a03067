"""N>1 path on CPU: two ranks (gloo), each with its own shard of demonstrations; shared-theta mode all-reduces the
summed gradient every outer iteration (the RCCL all-reduce of the GPU run) and both ranks must hold the same theta,
equal to a single-process run over the union of the shards.  Uses the SIMT-emulator build of the kernels."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem():
    rng = np.random.default_rng(5)
    B = 6
    x0 = np.tile([0.0, 0.0], (B, 1)) + 0.1 * rng.standard_normal((B, 2))
    taus = np.tile([0.2, 0.5, 0.8], (B, 1))
    wps = rng.uniform(0.2, 2.5, (B, 3, 1))
    return x0, taus, wps, np.array([1.5, 0.8, 1.2])


def _learner(lo, hi, emu_path):
    import lfsd_amd  # noqa: F401
    from lfsd_amd import CPDP, models
    oc, env, d = models.pendulum(n_grid=10)
    oc.use_library(emu_path)
    oc.setDevice(dtype=torch.float64)
    x0, taus, wps, th0 = _problem()
    return CPDP.SparseDemoLearner(oc, x0[lo:hi], 1.0, taus[lo:hi], wps[lo:hi], [0], th0, method="Adam",
                                  learning_rate=1e-2, mode="shared")


def _worker(rank, world, port, emu_path, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = _learner(rank * 3, rank * 3 + 3, emu_path)
    out = []
    for it in range(3):
        loss, grad = L.step()
        out.append((loss.item(), grad.numpy().copy()))
    q.put((rank, L.theta.numpy().copy(), out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process():
    sys.path.insert(0, ROOT)
    from conftest import build_emu_library
    import lfsd_amd  # noqa: F401
    from lfsd_amd import models
    emu_path = build_emu_library(models.pendulum(n_grid=10)[0])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, emu_path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    L = _learner(0, 6, emu_path)
    ref = []
    for it in range(3):
        loss, grad = L.step()
        ref.append((loss.item(), grad.numpy().copy()))
    for rank, theta, out in res:
        assert np.allclose(theta, L.theta.numpy(), rtol=1e-12, atol=1e-14)
        for (l, g), (lr_, gr) in zip(out, ref):
            assert abs(l - lr_) < 1e-12 * max(1, abs(lr_)) and np.allclose(g, gr, rtol=1e-11, atol=1e-13)
