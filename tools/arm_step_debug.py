"""The robot-arm learner of tests/test_gpu_parity.py::test_robotarm_12_vanilla_steps_every_gradient_applied up to outer iteration
`k`: the admissible rows that did not converge there, re-solved one at a time with the trace variant (tools/oc_trace.py build robotarm).

    python tools/arm_step_debug.py <k>
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import lfsd_amd  # noqa: F401
from lfsd_amd import models, CPDP
import oc_trace

K = int(sys.argv[1])
B = 1024
rng = np.random.default_rng(0)
th0 = np.array([5.0, 1, 1, 1, 1])[None, :] * (1 + 0.05 * rng.standard_normal((B, 5)))
th0[:, 0] = np.abs(th0[:, 0]) + 0.1
oc, env, d = models.robotarm(n_grid=50)
oc.setDevice("cuda:0", torch.float32); oc.setSolverOptions(aux_substeps=4); oc.compile()
L = CPDP.SparseDemoLearner(oc, np.tile(d["ini_state"], (B, 1)), d["horizon"], d["taus"], d["waypoints"], d["interface"], th0,
                           method="Vanilla", learning_rate=d["lr"], skip_unconverged=False)
for k in range(K + 1):
    th = L.theta.double().cpu().numpy()
    adm = np.isfinite(th).all(1) & (np.abs(th) < 1e3).all(1) & (th[:, 0] > 0) & (th[:, 1] > 0.05) & (th[:, 3] > 0.05)
    L.step()
    st, it = L._sol["status"].cpu().numpy(), L._sol["iters"].cpu().numpy()
    print("step %d: status %s, iterations mean %.1f max %d, admissible %d, admissible not converged %s" %
          (k, np.bincount(st, minlength=5).tolist(), it.mean(), it.max(), adm.sum(), np.nonzero(adm & ~np.isin(st, (1, 2)))[0].tolist()), flush=True)
bad = np.nonzero(adm & ~np.isin(st, (1, 2)))[0]
oc2, _, _ = models.robotarm(n_grid=50)
oc2.use_library(oc_trace.variant_path(oc2.model_spec(), "trace")); oc2.setDevice("cuda:0", torch.float32)
oc2.setSolverOptions(mapping=oc.mapping, max_iter=oc.max_iter)
print("max_iter", oc.max_iter, "mapping", oc.mapping, "exact_after", oc.exact_after)
for j in bad[:2]:
    print("=== row %d theta %s iterations %d status %d" % (j, np.array2string(th[j], precision=6), it[j], st[j]), flush=True)
    s1 = oc2.cocSolverBatch(np.asarray(d["ini_state"], dtype=float)[None, :], d["horizon"], th[j:j + 1]); torch.cuda.synchronize()
    print("=== alone: iterations %d status %d cost %.8g" % (int(s1["iters"][0]), int(s1["status"][0]), float(s1["cost"][0])), flush=True)
