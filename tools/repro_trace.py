"""GPU repro helper: robot arm fp32 with the kernel's per-iteration trace (usage: build | run)."""
import sys, os, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import lfsd_amd
from lfsd_amd import models, runtime
oc, env, d = models.ZOO["robotarm"](n_grid=12)
spec = oc.model_spec()
out = os.path.join(runtime.BUILD_DIR, "trace_%s.so" % spec.hash())
if sys.argv[1] == "build":
    runtime.write_header(spec)
    r = subprocess.run(runtime.hipcc_command(spec, out, ["-DLFSD_TRACE"]), cwd=runtime.CSRC_DIR, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    print(out)
    sys.exit(0)
if sys.argv[1] == "run":
    oc.use_library(out)
oc.setDevice("cuda:0", torch.float32)
oc.setSolverOptions(exact_after=int(sys.argv[2]) if len(sys.argv) > 2 else 16)
thetas = np.array([[5., 1, 1, 1, 1], [3., 0.5, 2, 1.5, 0.2]])
sol = oc.cocSolverBatch(np.tile(d["ini_state"], (2, 1)), d["horizon"], thetas)
torch.cuda.synchronize()
print("status", sol["status"].tolist(), "iters", sol["iters"].tolist(), "cost", sol["cost"].tolist())
print("x[0,:3]", sol["state_grid"][0, :3].tolist())
