"""bench.py's launcher logic (no GPU): `--gpus N` without a torch.distributed.run environment must start N ranks itself
BEFORE anything touches a GPU, relay the child's exit code, and refuse a world size that contradicts `--gpus`."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_gpus_n_spawns_torch_distributed_run(monkeypatch):
    import bench
    for k in ("RANK", "WORLD_SIZE", "MASTER_PORT", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setitem(sys.modules, "torch", None)          # importing torch in the launcher process would be a bug
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert e.value.code == 7                                   # the child's exit code is relayed
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_world_size_must_match_gpus(monkeypatch):
    import bench
    monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("MASTER_PORT", "29999")
    monkeypatch.setitem(sys.modules, "torch", None)
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "8"])
    assert e.value.code == 2


def test_default_workload_per_world_size():
    import bench
    a = bench.parse_args([])
    assert a.gpus == 1 and a.steps == 20 and a.warmup == 5 and a.batch == 4096 and a.mode is None
    # mode None -> independent seeds (configs[2]) at N = 1, shared theta + all-reduce (configs[3]) at N > 1: see bench.main
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'args.mode or ("shared" if world > 1 else "independent")' in src


def test_config_selects_the_baseline_workload():
    """`--config` picks BASELINE configs[1] / configs[4] with their own batch, grid and arithmetic; explicit flags still win."""
    import bench
    a = bench.parse_args(["--config", "robotarm"])
    assert (a.batch, a.n_grid, a.dtype) == (1024, 50, "f32") and bench.WORKLOADS["robotarm"]["method"] == "Vanilla"
    r = bench.parse_args(["--config", "rocket"])
    assert (r.batch, r.n_grid, r.dtype) == (1024, 100, "f32") and bench.WORKLOADS["rocket"]["aux_dtype"] == "f64"
    assert bench.parse_args(["--config", "rocket", "--batch", "64", "--n-grid", "15"]).batch == 64
    q = bench.parse_args([])
    assert (q.config, q.batch, q.n_grid, q.dtype, q.no_f64_leg) == ("quadrotor", 4096, 50, "f32", False)
    with pytest.raises(SystemExit):
        bench.parse_args(["--config", "pendulum"])
