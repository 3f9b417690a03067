"""The C-ABI libraries export every symbol include/lfsd_cpdp.h declares and reject bad arguments
(no compute: there is no GPU in this tier)."""
import ctypes
import os
import re

import pytest

import lfsd_amd  # noqa: F401
from lfsd_amd import models, runtime

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lfsd_cpdp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lfsd_[a-z_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    assert declared_symbols() == sorted(runtime.ModelLibrary.EXPORTS)


@pytest.mark.parametrize("name", sorted(models.ZOO))
def test_hip_library_exports_all_symbols(name):
    oc, _, _ = models.ZOO[name]()
    spec = oc.model_spec()
    path = runtime.build_library(spec)        # no-op when the in-tree .so is newer than its sources
    lib = ctypes.CDLL(path)
    for sym in declared_symbols():
        assert hasattr(lib, sym), (name, sym)
    ml = runtime.ModelLibrary(path)
    assert not ml.is_emulator
    assert (ml.n_state, ml.n_control, ml.n_auxvar) == (spec.n, spec.m, spec.p)
    assert ml.hash == spec.hash() and ml.lanes == runtime.lanes_for(spec.n, spec.m, spec.p)
    assert ml.time_varying == spec.time_varying
    # argument validation happens before any launch
    assert lib.lfsd_coc_solve(0, 0, 10, 4, None, None, None, None, 0, None, None, None, None, None, None, ctypes.c_double(0.0),
                              None, None, None, None, None, None,
                              10, ctypes.c_double(1e-6), 10, 0, None, ctypes.c_size_t(0), None) == -1
    assert lib.lfsd_coc_solve(7, 1, 10, 4, None, None, None, None, 0, None, None, None, None, None, None, ctypes.c_double(0.0),
                              None, None, None, None, None, None,
                              10, ctypes.c_double(1e-6), 10, 0, None, ctypes.c_size_t(0), None) == -1
    # ABI 8: a skip mask without the status array (or a negative mask) is an argument error -- caught before any launch, so
    # host dummies stand in for the device arrays here
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.lfsd_aux_solve.argtypes = [ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, ci, cd, vp, vp, ci, vp]
    buf = (ctypes.c_double * 8)()
    d = ctypes.cast(buf, vp)
    for mask in (1 << 4, -1):
        assert lib.lfsd_aux_solve(0, 1, 10, d, d, d, 0, d, d, d, d, 0, 0, None, None, None, d, d, None, None, 0, 1e-3, None,
                                  None, mask, None) == -1
    lib.lfsd_coc_workspace_bytes.restype = ctypes.c_size_t
    assert lib.lfsd_coc_workspace_bytes(0, 4096, 50, 16, 0, 0) > 0
    assert lib.lfsd_coc_workspace_bytes(3, 4096, 50, 16, 0, 0) == 0
    assert lib.lfsd_coc_workspace_bytes(0, 4096, 50, 16, 7, 0) == 0        # unknown mapping
    # the wide mapping (forced, or implied by control bounds) needs more scratch per trajectory than the lock-step one
    lock, wide = lib.lfsd_coc_workspace_bytes(0, 4096, 50, 16, 1, 0), lib.lfsd_coc_workspace_bytes(0, 4096, 50, 16, 2, 0)
    assert 0 < lock < wide and lib.lfsd_coc_workspace_bytes(0, 4096, 50, 16, 1, 1) == wide


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the HIP library must fail loudly when handed host memory."""
    import torch
    oc, _, d = models.pendulum()
    lib = oc.compile()
    x0 = torch.zeros((1, 2), dtype=torch.float32)
    with pytest.raises(runtime.LfsdError):
        lib.coc_solve(x0, torch.ones(1), torch.ones((1, 3)), torch.tensor(lib.const_defaults, dtype=torch.float32), 10)
    if not torch.cuda.is_available():
        with pytest.raises(runtime.LfsdError):
            oc.cocSolver(d["ini_state"], 1.0, [1, 1, 1])


def test_missing_library_is_loud(tmp_path):
    with pytest.raises(runtime.LfsdError):
        runtime.ModelLibrary(str(tmp_path / "liblfsd_missing.so"))


def test_binding_argument_counts_match_the_header():
    """Every entry point's ctypes signature (runtime.ModelLibrary, and the stub INTEGRATION.md shows a reference maintainer) has
    exactly as many arguments as include/lfsd_cpdp.h declares."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "lfsd_cpdp.h")).read(), flags=re.S)
    declared = {}
    for name in runtime.ModelLibrary.EXPORTS:
        m = re.search(name + r"\s*\((.*?)\);", hdr, flags=re.S)
        assert m, name
        args = [a.strip() for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        declared[name] = len(args)
    count = lambda body: len([a for a in body.replace("\n", " ").split(",") if a.strip()])
    src = open(os.path.join(ROOT, "learning-from-sparse-demonstrations_amd", "runtime.py")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    seen = 0
    for text in (src, doc):
        for m in re.finditer(r"L\.(lfsd_\w+)\.argtypes = \[(.*?)\]", text, flags=re.S):
            assert count(m.group(2)) == declared[m.group(1)], (m.group(1), count(m.group(2)), declared[m.group(1)])
            seen += 1
    assert seen >= 9
