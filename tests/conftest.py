import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_BUILD = os.path.join(EMU_DIR, "_build")


def pytest_addoption(parser):
    parser.addoption("--sanitize-all", action="store_true", help="run the AddressSanitizer build for every model (slow)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def build_emu_library(oc, extra_flags=(), tag=""):
    """g++ build of the SAME kernel sources against the CPU SIMT emulator (tests/emu/simt_emu.h).

    Test infrastructure only: lets the CPU suite exercise the kernels' logic; never used by the product.
    `extra_flags` + `tag`: a variant build (an experiment switch of csrc/cpdp_common.h set on the command line)."""
    import lfsd_amd
    from lfsd_amd import runtime
    spec = oc.model_spec()
    runtime.write_header(spec)
    os.makedirs(EMU_BUILD, exist_ok=True)
    out = os.path.join(EMU_BUILD, "liblfsd_%s_emu%s.so" % (spec.hash(), ("_" + tag) if tag else ""))
    deps = [runtime.header_path(spec.hash()), os.path.join(EMU_DIR, "simt_emu.h")] + \
        [os.path.join(runtime.CSRC_DIR, f) for f in runtime.KERNEL_SOURCES]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    g = runtime.lanes_for(spec.n, spec.m, spec.p)
    cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-DLFSD_EMU", "-DLFSD_POISON_LDS", "-fvisibility=hidden", "-DLFSD_G=%d" % g,
           '-DLFSD_MODEL_HEADER="gen/%s.h"' % spec.hash(), "-I" + EMU_DIR, "-I" + runtime.CSRC_DIR] + list(extra_flags) + [
           os.path.join(runtime.CSRC_DIR, "lfsd_capi.cpp"), "-o", out + ".tmp"]
    r = subprocess.run(cmd, cwd=runtime.CSRC_DIR, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    os.replace(out + ".tmp", out)
    return out


def build_variant_library(oc, tag, flags):
    """A variant build of a model library (lfsd_amd.runtime.build_variant_library: csrc/build/ab_<hash>_<tag>.so)."""
    from lfsd_amd import runtime
    return runtime.build_variant_library(oc.model_spec(), tag, flags)


def _tier_variants():
    import lfsd_amd  # noqa: F401
    from lfsd_amd import runtime
    return runtime.PLAIN_SCHEDULE, tuple((k, t, f) for k, _, t, f in runtime.GPU_TIER_VARIANTS)


# build variants the -m gpu tier compares the product with (kind, tag, flags): defined next to the build code, lfsd_amd/runtime.py
PLAIN_SCHEDULE, GPU_TIER_VARIANTS = _tier_variants()


@pytest.fixture(params=["lockstep", "wide"])
def oc_mapping(request, monkeypatch):
    """Both mappings of the OC solve: several trajectories per wavefront with the intervals in sequence (lfsd_coc_solve's
    choice from a few thousand trajectories up) and one trajectory per wavefront with the intervals in parallel (its choice
    below that).  COCSys.mapping_override forces one or the other whatever the instance or the batch size asks for (it is
    handed to lfsd_coc_solve as its `mapping` argument)."""
    from lfsd_amd import CPDP
    monkeypatch.setattr(CPDP.COCSys, "mapping_override", request.param)
    return request.param


@pytest.fixture(scope="session")
def emu():
    def bind(oc):
        oc.use_library(build_emu_library(oc))
        oc.compile()
        return oc
    return bind


def make_oracle(kind, n_grid, **kw):
    """Oracle twin of lfsd_amd.models.<kind> (separately written model definitions in oracle/jinenv_sym.py)."""
    import sympy as sp
    from oracle import jinenv_sym as J
    from oracle.cpdp_oracle import COCSys
    if kind == "pendulum":
        env = J.SinglePendulum(); env.initDyn(l=1, m=1, damping_ratio=0.1); env.initCost(wu=.01)
    elif kind == "robotarm":
        env = J.RobotArm(); env.initDyn(l1=1, m1=1, l2=1, m2=1, g=0); env.initCost_Polynomial(wu=.5)
    elif kind == "cartpole":
        env = J.CartPole(); env.initDyn(mc=0.5, mp=0.5, l=1); env.initCost(wu=0.1)
    elif kind == "quadrotor":
        goal = kw.get("goal", (3, 3, 1.5))
        env = J.Quadrotor(); env.initDyn(1.0, 1.0, 1.0, 1.0, 1.0, 0.02)
        env.initCost_Polynomial(list(goal), [0, 0, 0], [1, 0, 0, 0], [0, 0, 0], w_thrust=0.1)
    elif kind == "rocket":
        env = J.Rocket(); env.initDyn(Jx=1, Jy=1, Jz=1, mass=1, l=1); env.initCost2(wthrust=0.1)
    else:
        raise KeyError(kind)
    oc = COCSys()
    beta = sp.Symbol('beta', real=True)
    oc.setAuxvarVariable([beta] + list(env.cost_auxvar))
    # finite bounds (CPDP.py:20-46) go through the oracle's reference-shaped setters: its cocSolver then solves the bounded NLP
    oc.setStateVariable(env.X, kw.get("state_lb", []), kw.get("state_ub", []))
    oc.setControlVariable(env.U, kw.get("control_lb", []), kw.get("control_ub", []))
    oc.setDyn(beta * env.f); oc.setPathCost(beta * env.path_cost); oc.setFinalCost(env.final_cost)
    oc.setIntegrator(n_grid)
    return oc


TIGHT = dict(riccati_method='Radau', ivp_kwargs=dict(rtol=1e-10, atol=1e-12))


_REF_CACHE = {}


def oracle_loss_grad(oc, ini_state, horizon, theta, taus, wps, iface, tight=True, **solver_kw):
    """(cached per session: several tests and parametrisations compare against the same oracle solution)"""
    import numpy as np
    key = (id(type(oc)), tuple(map(str, oc.state)), str(oc.dyn), str(oc.path_cost), str(oc.final_cost), oc.n_grid,
           tuple(np.ravel(ini_state).tolist()), float(horizon), tuple(np.ravel(theta).tolist()),
           tuple(np.ravel(taus).tolist()), tuple(np.ravel(wps).tolist()), tuple(iface), tight, tuple(sorted(solver_kw.items())))
    if key not in _REF_CACHE:
        _REF_CACHE[key] = _oracle_loss_grad(oc, ini_state, horizon, theta, taus, wps, iface, tight=tight, **solver_kw)
    return _REF_CACHE[key]


def _oracle_loss_grad(oc, ini_state, horizon, theta, taus, wps, iface, tight=True, **solver_kw):
    from oracle.cpdp_oracle import getloss_corrections
    tg, sol, X, U, L = oc.cocSolver(ini_state, horizon, theta, return_grids=True, **solver_kw)
    aux, PW, vX, vU = oc.auxSysSolver(tg, sol, theta, return_grids=True, **(TIGHT if tight else {}))
    loss, grad = getloss_corrections(oc, taus, wps, sol, aux, iface)
    return dict(loss=loss, grad=grad, X=X, U=U, L=L, PW=PW, vX=vX, vU=vU, info=oc.last_info)


def oracle_check_solution(oc, ini_state, horizon, theta, X, U, L, taus, wps, iface, tight=True):
    """Basin-independent parity check for problems with several local minima (rocket, cart-pole swing-up): the oracle
    does not solve -- it (1) certifies with complex-step arithmetic that the given grids ARE a KKT point of the
    reference's NLP (CPDP.py:126-179) and (2) differentiates the PMP along exactly these grids (CPDP.py:301-381) and
    evaluates the loss / gradient.  X, U, L: [N+1][n], [N+1][m], [N+1][n] numpy arrays."""
    import numpy as np
    from oracle.cpdp_oracle import getloss_corrections
    oc.diffPMP()
    X, U, L = (np.asarray(a, dtype=np.float64) for a in (X, U, L))
    defect, gmax, lmax = oc.kkt_certificate(ini_state, horizon, theta, X, U, L)
    tg = np.linspace(0, horizon, oc.n_grid + 1)
    sol = oc.interpolation(tg, np.concatenate((X, U, L), axis=1))
    aux, PW, vX, vU = oc.auxSysSolver(tg, sol, theta, return_grids=True, **(TIGHT if tight else {}))
    loss, grad = getloss_corrections(oc, taus, wps, sol, aux, iface)
    return dict(defect=defect, gmax=gmax, lmax=lmax, loss=loss, grad=grad, PW=PW, vX=vX, vU=vU)


# ---- the slow fp64 oracle, fanned out over the host cores (plain child processes, one BLAS thread each) -------------
_ORACLES = {}


def oracle_job(job):
    """job: dict(kind, n_grid, ini_state, horizon, theta, taus, wps, iface, tight=True, solver_kw={}, make_kw={},
    check=None).  check = (X, U, L): do not solve, certify + differentiate these grids (oracle_check_solution)."""
    key = (job["kind"], job["n_grid"], tuple(sorted(job.get("make_kw", {}).items())))
    if key not in _ORACLES:
        _ORACLES[key] = make_oracle(job["kind"], job["n_grid"], **job.get("make_kw", {}))
    o = _ORACLES[key]
    if job.get("check") is not None:
        X, U, L = job["check"]
        return oracle_check_solution(o, job["ini_state"], job["horizon"], job["theta"], X, U, L, job["taus"], job["wps"],
                                     job["iface"], tight=job.get("tight", True))
    try:
        r = oracle_loss_grad(o, job["ini_state"], job["horizon"], job["theta"], job["taus"], job["wps"], job["iface"],
                             tight=job.get("tight", True), **job.get("solver_kw", {}))
    except RuntimeError as exc:          # e.g. finite escape of the Riccati solution (conjugate point) under the tight integrator
        if job.get("allow_fail"):
            return dict(error=str(exc))
        raise
    r["cost"] = o.last_cost
    return r


def oracle_parallel(jobs, timeout=1500.0):
    """Results of oracle_job for every job, computed by min(len(jobs), cores) child processes."""
    import pickle
    import tempfile
    import time
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    w = max(1, min(len(jobs), cores))
    if w == 1:
        return [oracle_job(j) for j in jobs]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for i in range(w):
            jp, op = os.path.join(td, "j%d.pkl" % i), os.path.join(td, "o%d.pkl" % i)
            with open(jp, "wb") as f:
                pickle.dump(jobs[i::w], f)
            procs.append((subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "oracle_worker.py"), jp, op], env=env), op))
        deadline = time.time() + timeout
        try:
            for pr, _ in procs:
                pr.wait(max(1.0, deadline - time.time()))
                assert pr.returncode == 0, "oracle worker failed"
        finally:
            for pr, _ in procs:
                if pr.poll() is None:
                    pr.kill()
        out = [None] * len(jobs)
        for i, (_, op) in enumerate(procs):
            out[i::w] = pickle.load(open(op, "rb"))
    return out


def parity_record(what, metric, measured, asserted):
    """Every parity comparison leaves its measured error beside the asserted bound: in the failure message, and -- when
    LFSD_PARITY_REPORT names a file -- as one JSON line per comparison (profiles/*_parity_floors.jsonl are such runs on the
    GPU; the asserted bounds of the fp32 cases are set from them with a stated margin, not chosen blanket)."""
    import json
    path = os.environ.get("LFSD_PARITY_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(json.dumps(dict(what=what, metric=metric, measured=float(measured), asserted=float(asserted))) + "\n")
    assert measured < asserted, "%s: %s measured %.3e, asserted < %.1e" % (what, metric, measured, asserted)


def assert_grids_match(sol, aux, b, r, n, m, p, tol, what=""):
    """Every output of the path vs the oracle: state / control / costate grids (CPDP.py:186-196), Riccati pair [P W]
    (CPDP.py:329-338), dx/dtheta and du/dtheta (CPDP.py:352-381), loss and gradient.  tol: dict(grid, costate, aux, loss, grad)."""
    import numpy as np

    def rel(a, ref):
        a = a.detach().double().cpu().numpy() if hasattr(a, "detach") else np.asarray(a, dtype=np.float64)
        ref = np.asarray(ref, dtype=np.float64)
        return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-300)
    parity_record(what, "state", rel(sol["state_grid"][b], r["X"]), tol["grid"])
    parity_record(what, "control", rel(sol["control_grid"][b], r["U"]), tol["grid"])
    parity_record(what, "costate", rel(sol["costate_grid"][b], r["L"]), tol["costate"])
    if aux.get("auxX_grid") is not None:
        N1 = r["PW"].shape[0]
        Zo = np.concatenate([r["PW"][:, :n * n].reshape(N1, n, n), r["PW"][:, n * n:].reshape(N1, n, p)], axis=2)
        parity_record(what, "Z_grid", rel(aux["Z_grid"][b].permute(0, 2, 1), Zo), tol.get("Z", tol["aux"]))
        parity_record(what, "auxX_grid", rel(aux["auxX_grid"][b].permute(0, 2, 1).reshape(N1, n * p), r["vX"]), tol["aux"])
        # du/dtheta(T) = -Huu^-1 (fu^T h_xx) dx/dtheta(T) + ...  amplifies the error of dx/dtheta by |Huu^-1 fu^T h_xx| (10^3 for
        # the arm's final-cost weight 100), hence its own tolerance
        parity_record(what, "auxU_grid", rel(aux["auxU_grid"][b].permute(0, 2, 1).reshape(N1, m * p), r["vU"]), tol.get("auxU", tol["aux"]))
    parity_record(what, "loss", abs(float(aux["loss"][b]) - r["loss"]) / max(1.0, r["loss"]), tol["loss"])
    parity_record(what, "grad", rel(aux["grad"][b], r["grad"]), tol["grad"])
