import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_BUILD = os.path.join(EMU_DIR, "_build")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def build_emu_library(oc):
    """g++ build of the SAME kernel sources against the CPU SIMT emulator (tests/emu/simt_emu.h).

    Test infrastructure only: lets the CPU suite exercise the kernels' logic; never used by the product."""
    import lfsd_amd
    from lfsd_amd import runtime
    spec = oc.model_spec()
    runtime.write_header(spec)
    os.makedirs(EMU_BUILD, exist_ok=True)
    out = os.path.join(EMU_BUILD, "liblfsd_%s_emu.so" % spec.hash())
    deps = [runtime.header_path(spec.hash()), os.path.join(runtime.CSRC_DIR, "cpdp_kernels.h"),
            os.path.join(runtime.CSRC_DIR, "lfsd_capi.cpp"), os.path.join(runtime.CSRC_DIR, "lfsd_internal.h"),
            os.path.join(runtime.CSRC_DIR, "lfsd_riccati.inc"), os.path.join(EMU_DIR, "simt_emu.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    g = runtime.lanes_for(spec.n, spec.m, spec.p)
    cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-DLFSD_EMU", "-DLFSD_POISON_LDS", "-fvisibility=hidden", "-DLFSD_G=%d" % g,
           '-DLFSD_MODEL_HEADER="gen/%s.h"' % spec.hash(), "-I" + EMU_DIR, "-I" + runtime.CSRC_DIR,
           os.path.join(runtime.CSRC_DIR, "lfsd_capi.cpp"), "-o", out + ".tmp"]
    r = subprocess.run(cmd, cwd=runtime.CSRC_DIR, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    os.replace(out + ".tmp", out)
    return out


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
    oc.setStateVariable(env.X); oc.setControlVariable(env.U)
    oc.setDyn(beta * env.f); oc.setPathCost(beta * env.path_cost); oc.setFinalCost(env.final_cost)
    oc.setIntegrator(n_grid)
    return oc


TIGHT = dict(riccati_method='Radau', ivp_kwargs=dict(rtol=1e-10, atol=1e-12))


def oracle_loss_grad(oc, ini_state, horizon, theta, taus, wps, iface, tight=True, **solver_kw):
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
