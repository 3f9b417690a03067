"""ctypes binding of the C ABI in include/lfsd_cpdp.h + the hipcc build of one model library.

PyTorch is plumbing here: tensors own the HBM buffers and the HIP stream; every
numeric step of the hot path runs in the model's shared library.  There is no
CPU fallback: a missing library is built with hipcc or the call fails loudly.
"""
import ctypes
import os
import shutil
import subprocess

import torch

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(PKG_DIR, "csrc")
GEN_DIR = os.path.join(CSRC_DIR, "gen")
BUILD_DIR = os.path.join(CSRC_DIR, "build")
INCLUDE_DIR = os.path.join(os.path.dirname(PKG_DIR), "include")

LFSD_F32, LFSD_F64 = 0, 1
STATUS = {1: "converged", 2: "stalled", 3: "maxiter", 4: "failed"}
OPT_METHODS = {"Vanilla": 0, "Nesterov": 1, "Adam": 2, "Nadam": 3, "AMSGrad": 4}
MAPPINGS = {"auto": 0, "lockstep": 1, "wide": 2}


class LfsdError(RuntimeError):
    pass


class _ModelInfo(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_int), ("n_state", ctypes.c_int), ("n_control", ctypes.c_int),
                ("n_auxvar", ctypes.c_int), ("n_const", ctypes.c_int), ("time_varying", ctypes.c_int),
                ("lanes_per_trajectory", ctypes.c_int), ("is_emulator", ctypes.c_int),
                ("name", ctypes.c_char_p), ("hash", ctypes.c_char_p)]


def lanes_for(n, m, p):
    need = max(n + m, n + p, 8)
    g = 8
    while g < need:
        g *= 2
    if g > 64:
        raise LfsdError("model too wide for one wavefront per trajectory: n+max(m,p) = %d > 64" % need)
    return g


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise LfsdError("hipcc not found: the HIP model library cannot be built")


def library_path(model_hash):
    return os.path.join(BUILD_DIR, "liblfsd_%s.so" % model_hash)


def header_path(model_hash):
    return os.path.join(GEN_DIR, "%s.h" % model_hash)


def write_header(spec, force=False):
    from . import codegen
    os.makedirs(GEN_DIR, exist_ok=True)
    hp = header_path(spec.hash())
    if force or not os.path.exists(hp):
        src = codegen.emit_header(spec)
        tmp = hp + ".tmp%d" % os.getpid()
        with open(tmp, "w") as f:
            f.write(src)
        os.replace(tmp, hp)
    return hp


def _hipcc_flags(spec, extra=()):
    g = lanes_for(spec.n, spec.m, spec.p)
    return [find_hipcc(), "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC",
            "-fno-signed-zeros", "-fvisibility=hidden",
            "-DLFSD_G=%d" % g, '-DLFSD_MODEL_HEADER="gen/%s.h"' % spec.hash(), "-I" + CSRC_DIR] + list(extra) + \
        os.environ.get("LFSD_EXTRA_HIPCC_FLAGS", "").split()       # tuning experiments (tools/tune.py)


def hipcc_command(spec, out, extra=()):
    """The library as ONE translation unit (tuning tools; the product build is hipcc_commands)."""
    return _hipcc_flags(spec, extra) + ["-shared", os.path.join(CSRC_DIR, "lfsd_capi.cpp"), "-o", out]


# Per translation unit.  Rounds 1-5 compiled the first unit with `-mllvm -amdgpu-sched-strategy=max-ilp` (round 1: 5 % off oc_solve;
# the Riccati sweep lost 25 % with it, hence its own unit).  Round 6 dropped it: two builds of the wide OC kernel were WRONG under
# that scheduler and right under the default one with the same source (profiles/r06_f_wide_stale_cost.txt), and it no longer pays:
# headline 966 300 -> 966 700 it/s, rocket step 40.8 -> 38.8 ms, robot arm 15.31 -> 15.41 ms, fp64 headline 13.5 -> 13.9 ms
# (profiles/r06_i_max_ilp_ab.txt).  The cause turned out to be independent of the scheduler -- VGPR spills placed before the exec restore
# of a join block, which a third wrong build showed under the DEFAULT one -- and every build is now scanned for it (_checked_build below).
TUNED_CAPI = ()
TUNED_RICCATI = ()


def hipcc_commands(spec, out, extra=(), extra_capi=TUNED_CAPI, extra_riccati=TUNED_RICCATI):
    """Product build: two translation units -- the Riccati sweep apart from everything else -- so that each gets the
    compiler settings it measured best with (csrc/lfsd_internal.h, profiles/r01_tune_compiler_flags.txt), then the link.
    Both are compiled without clang's SLP vectoriser."""
    flags = _hipcc_flags(spec, extra) + ["-fno-slp-vectorize"]
    o1, o2 = out + ".capi.o", out + ".riccati.o"
    return ([flags + ["-c", "-DLFSD_SPLIT_RICCATI", os.path.join(CSRC_DIR, "lfsd_capi.cpp"), "-o", o1] + list(extra_capi),
             flags + ["-c", os.path.join(CSRC_DIR, "lfsd_riccati.cpp"), "-o", o2] + list(extra_riccati),
             [find_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", o1, o2, "-o", out]], [o1, o2])


# every hand-written file a model library is compiled from (rebuild when any of them is newer than the .so)
KERNEL_SOURCES = ("cpdp_kernels.h", "cpdp_common.h", "cpdp_oc.h", "cpdp_aux.h", "cpdp_opt.h", "lfsd_capi.cpp", "lfsd_internal.h",
                  "lfsd_riccati.inc", "lfsd_riccati.cpp")


# Every build is compiled with -save-temps and its gfx950 assembly scanned by lfsd_amd.isa_check (VGPR spills placed before the
# exec restore of a join block: two wrong builds of the wide OC kernel in round 6 -- isa_check.py, profiles/
# r06_v_spill_before_exec_restore.txt).  A translation unit whose assembly shows the pattern is recompiled with the next entry of
# this list -- other instruction schedulers: the placement moves with any perturbation of the kernel -- until the scan is clean;
# if none is, the build FAILS (no library is better than a library whose upper lanes read stale spill slots).
SCHEDULE_ALTERNATES = ((), ("-mllvm", "-amdgpu-sched-strategy=max-ilp"), ("-mllvm", "-amdgpu-sched-strategy=max-memory-clause"),
                       ("-mllvm", "-amdgpu-use-amdgpu-trackers"), ("-mllvm", "-greedy-reverse-local-assignment"))
ISA_CHECK_VERSION = 3      # (2: loop exits without a skip branch are examined too; 3: the entries of else-regions count as restores)


def isa_record_path(lib_path):
    return lib_path + ".isa.json"


def isa_record_clean(lib_path):
    """The library was built by _checked_build of this version and both units passed (a library without such a record -- built by an
    older tree or by hand -- counts as stale: build_library compiles it again)."""
    import json
    try:
        rec = json.load(open(isa_record_path(lib_path)))
    except (OSError, ValueError):
        return False
    units = rec.get("units", {})
    return rec.get("isa_check") == ISA_CHECK_VERSION and set(units) == {"capi", "riccati"} and all(u.get("hazards") == 0 for u in units.values()) \
        and rec.get("sha256") == _sha256(lib_path)      # (the record belongs to THIS file, not to one a later hand build replaced)


def _sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()


def _checked_build(spec, out, extra=(), verbose=False, what="model"):
    """Compile the two translation units into `out` (csrc/build/...), each with the first entry of SCHEDULE_ALTERNATES whose device
    assembly passes isa_check; writes `<out>.isa.json` (what was scanned, which flags each unit was built with).  Raises LfsdError."""
    import json, shutil, tempfile
    from . import isa_check
    os.makedirs(BUILD_DIR, exist_ok=True)
    work = tempfile.mkdtemp(prefix="obj%d_" % os.getpid(), dir=BUILD_DIR)
    try:
        tmp = os.path.join(work, "lib.so")
        record = {"isa_check": ISA_CHECK_VERSION, "units": {}}
        objs = []
        for unit in ("capi", "riccati"):
            tuned = TUNED_CAPI if unit == "capi" else TUNED_RICCATI
            tried, built = [], False
            for alt in SCHEDULE_ALTERNATES:
                kw = {"extra_capi": tuple(tuned) + tuple(alt)} if unit == "capi" else {"extra_riccati": tuple(tuned) + tuple(alt)}
                cmds, obj_paths = hipcc_commands(spec, tmp, list(extra), **kw)
                cmd = cmds[0 if unit == "capi" else 1] + ["-save-temps=obj"]
                if verbose:
                    print(" ".join(cmd))
                for f in os.listdir(work):      # (the temporaries of the previous attempt)
                    if f.endswith(".s"):
                        os.remove(os.path.join(work, f))
                r = subprocess.run(cmd, cwd=CSRC_DIR, capture_output=True, text=True)
                if r.returncode != 0:
                    if alt:      # a toolchain that does not know this -mllvm option: next
                        tried.append({"flags": list(alt), "error": r.stderr[-300:]})
                        continue
                    raise LfsdError("hipcc failed for %s %s:\n%s\n%s" % (what, spec.name, r.stdout[-4000:], r.stderr[-4000:]))
                asm = [f for f in os.listdir(work) if f.endswith("gfx950.s")]
                if len(asm) != 1:
                    raise LfsdError("no device assembly to check for %s %s (%s): %s" % (what, spec.name, unit, sorted(os.listdir(work))))
                text = open(os.path.join(work, asm[0])).read()
                hz = isa_check.find_exec_hazards(text)
                if hz:
                    tried.append({"flags": list(alt), "hazards": len(hz), "first": hz[0]})
                    if verbose:
                        print("isa_check: %d spill(s) before an exec restore in %s (%s, flags %s): %s" % (len(hz), spec.name, unit, list(alt), hz[0]))
                    continue
                record["units"][unit] = dict(isa_check.summary(text), flags=list(tuned) + list(alt), hazards=0, rejected=tried)
                objs.append(obj_paths[0 if unit == "capi" else 1])
                built = True
                break
            if not built:
                raise LfsdError("every build of %s %s (%s) has VGPR spills before an exec restore (lfsd_amd/isa_check.py): %s"
                                % (what, spec.name, unit, tried))
        link = hipcc_commands(spec, tmp, list(extra))[0][2]
        r = subprocess.run(link, cwd=CSRC_DIR, capture_output=True, text=True)
        if r.returncode != 0:
            raise LfsdError("link failed for %s %s:\n%s" % (what, spec.name, r.stderr[-3000:]))
        record["sha256"] = _sha256(tmp)
        with open(isa_record_path(out) + ".tmp%d" % os.getpid(), "w") as f:
            json.dump(record, f, indent=1, sort_keys=True, default=str)
        os.replace(tmp, out)
        os.replace(isa_record_path(out) + ".tmp%d" % os.getpid(), isa_record_path(out))
        return record
    finally:
        shutil.rmtree(work, ignore_errors=True)


def build_checked(spec, out, flags=(), verbose=False):
    """A diagnostic / experiment build (tools/): the product's two-unit build with extra flags into `out`, assembly-checked the same way
    -- an instrumented build that is silently wrong misleads (round 6: a clock-reading build of the wide kernel was)."""
    write_header(spec)
    return _checked_build(spec, out, extra=list(flags), verbose=verbose, what="diagnostic build of model")


def build_library(spec, force=False, verbose=False):
    """Generate the model header and compile the gfx950 shared library in-tree (csrc/build/), assembly checked (above)."""
    os.makedirs(BUILD_DIR, exist_ok=True)
    out = library_path(spec.hash())
    write_header(spec, force=force)
    deps = [header_path(spec.hash()), os.path.join(INCLUDE_DIR, "lfsd_cpdp.h")] + \
        [os.path.join(CSRC_DIR, f) for f in KERNEL_SOURCES]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps) and isa_record_clean(out):
        return out
    _checked_build(spec, out, verbose=verbose)
    return out


def variant_library_path(spec, tag):
    return os.path.join(BUILD_DIR, "ab_%s_%s.so" % (spec.hash(), tag))


def build_variant_library(spec, tag, flags):
    """hipcc build of a VARIANT of a model library (experiment switches of csrc/cpdp_common.h set on the command line) into
    csrc/build/ab_<hash>_<tag>.so, rebuilt when a kernel source is newer.  The A/B tests of the GPU tier compare the product
    build with such variants; never loaded by the product path.  Same assembly check as the product build."""
    write_header(spec)
    out = variant_library_path(spec, tag)
    deps = [header_path(spec.hash())] + [os.path.join(CSRC_DIR, f) for f in KERNEL_SOURCES]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(p) for p in deps) and isa_record_clean(out):
        return out
    _checked_build(spec, out, extra=list(flags), what="variant %s of model" % tag)
    return out


# build variants the -m gpu tier compares the product with: (model kind, n_grid, tag, flags).  "plain" = every schedule off.
PLAIN_SCHEDULE = ("-DLFSD_LEAN_TC=1", "-DLFSD_COARSE_START=0", "-DLFSD_COARSE_TIME=1", "-DLFSD_MS=0")
GPU_TIER_VARIANTS = (("quadrotor", 50, "nocoarse", ("-DLFSD_COARSE_START=0",)),
                     ("quadrotor", 50, "plain", PLAIN_SCHEDULE), ("cartpole", 40, "plain", PLAIN_SCHEDULE),
                     ("rocket", 40, "plain", PLAIN_SCHEDULE), ("robotarm", 50, "plain", PLAIN_SCHEDULE))


def build_gpu_tier_variants(verbose=False):
    """Prebuild the variants above so that they travel with the tree (optional: a failure here is reported, not raised --
    the product libraries do not depend on them)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import models
    built = {}
    with ThreadPoolExecutor(max_workers=max(1, min(4, (os.cpu_count() or 2) // 2))) as pool:
        futs = {}
        for kind, n_grid, tag, flags in GPU_TIER_VARIANTS:
            oc = models.ZOO[kind](n_grid=n_grid)[0]
            futs[(kind, tag)] = pool.submit(build_variant_library, oc.model_spec(), tag, flags)
        for (kind, tag), fut in futs.items():
            try:
                built[(kind, tag)] = fut.result()
                if verbose:
                    print("variant", kind, tag, built[(kind, tag)])
            except Exception as exc:      # noqa: BLE001
                print("WARNING: test variant %s/%s not built: %s" % (kind, tag, str(exc)[:300]))
    return built


_DT = {torch.float32: LFSD_F32, torch.float64: LFSD_F64}


class ModelLibrary:
    """One loaded model library (all entry points of include/lfsd_cpdp.h)."""

    EXPORTS = ("lfsd_get_model_info", "lfsd_interface_dim", "lfsd_const_default", "lfsd_coc_workspace_bytes", "lfsd_coc_solve",
               "lfsd_aux_solve", "lfsd_aux_riccati", "lfsd_aux_forward", "lfsd_optimizer_step", "lfsd_lookahead")

    def __init__(self, path):
        if not os.path.exists(path):
            raise LfsdError("model library %s does not exist (build it with __graft_entry__.build() or "
                            "COCSys.compile())" % path)
        self.path = path
        self.lib = ctypes.CDLL(path)
        for sym in self.EXPORTS:
            if not hasattr(self.lib, sym):
                raise LfsdError("%s does not export %s" % (path, sym))
        L = self.lib
        vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.lfsd_get_model_info.argtypes = [ctypes.POINTER(_ModelInfo)]
        L.lfsd_const_default.argtypes = [ci]
        L.lfsd_const_default.restype = cd
        L.lfsd_coc_workspace_bytes.argtypes = [ci, ci, ci, ci, ci, ci]
        L.lfsd_coc_workspace_bytes.restype = ctypes.c_size_t
        L.lfsd_coc_solve.argtypes = [ci, ci, ci, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, cd, vp, vp, vp, vp, vp, vp, ci, cd, ci,
                                     ci, vp, ctypes.c_size_t, vp]
        L.lfsd_aux_solve.argtypes = [ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp,
                                     ci, cd, vp, vp, ci, vp]
        L.lfsd_aux_riccati.argtypes = [ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp, ci, cd, vp, vp, ci, vp]
        L.lfsd_aux_forward.argtypes = [ci, ci, ci, vp, vp, vp, ci, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp,
                                       ci, cd, vp, vp, ci, vp]
        L.lfsd_optimizer_step.argtypes = [ci, ci, ci, ci, ci, cd, cd, cd, cd, cd, vp, vp, vp, vp, vp, vp, vp, vp]
        L.lfsd_lookahead.argtypes = [ci, ctypes.c_longlong, cd, vp, vp, vp, vp]
        info = _ModelInfo()
        rc = L.lfsd_get_model_info(ctypes.byref(info))
        if rc != 0 or info.abi_version != 9:
            raise LfsdError("ABI mismatch in %s" % path)
        self.n_state, self.n_control, self.n_auxvar, self.n_const = (info.n_state, info.n_control, info.n_auxvar,
                                                                      info.n_const)
        self.time_varying = bool(info.time_varying)
        self.lanes = info.lanes_per_trajectory
        self.is_emulator = bool(info.is_emulator)
        self.name = info.name.decode()
        self.hash = info.hash.decode()
        self.const_defaults = [L.lfsd_const_default(i) for i in range(self.n_const)]
        self.n_interface = int(L.lfsd_interface_dim())      # outputs of the compiled interface function (0: none)

    # ---- argument plumbing ---------------------------------------------------------------
    def _check(self, t, shape, dtype, name, optional=False):
        if t is None:
            if optional:
                return None
            raise LfsdError("%s is required" % name)
        if not isinstance(t, torch.Tensor):
            raise LfsdError("%s must be a torch tensor" % name)
        if tuple(t.shape) != tuple(shape):
            raise LfsdError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
        if t.dtype != dtype:
            raise LfsdError("%s has dtype %s, expected %s" % (name, t.dtype, dtype))
        if not t.is_contiguous():
            raise LfsdError("%s must be contiguous" % name)
        if self.is_emulator:
            if t.device.type != "cpu":
                raise LfsdError("the SIMT-emulator test library only takes CPU tensors (%s)" % name)
        elif t.device.type != "cuda":
            raise LfsdError("%s lives on %s: the HIP library needs device memory (no CPU fallback)" % (name, t.device))
        return t

    @staticmethod
    def _p(t):
        return None if t is None else ctypes.c_void_p(t.data_ptr())

    def _stream(self, ref):
        if self.is_emulator:
            return None
        return ctypes.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)

    def _on(self, ref):
        """Context that makes the tensors' GPU the current HIP device (kernel launches go to the current device)."""
        import contextlib
        return contextlib.nullcontext() if self.is_emulator else torch.cuda.device(ref.device)

    @staticmethod
    def _rc(rc, what):
        if rc != 0:
            raise LfsdError("%s failed with code %d" % (what, rc))

    # ---- entry points ----------------------------------------------------------------------
    def coc_workspace_bytes(self, dtype, batch, n_grid, exact_after=16, mapping="auto", bounded=False):
        return int(self.lib.lfsd_coc_workspace_bytes(_DT[dtype], batch, n_grid, int(exact_after), MAPPINGS[mapping],
                                                      1 if bounded else 0))

    def coc_solve(self, ini_state, horizon, auxvar, consts, n_grid, steps_per_grid=4, u_init=None, max_iter=100,
                  tol=None, workspace=None, out=None, exact_after=16, control_lb=None, control_ub=None, mapping="auto",
                  state_lb=None, state_ub=None, state_mult=None, state_rho=0.0):
        """state_lb / state_ub [n], state_mult [B][n_grid][2][n], state_rho: ONE augmented-Lagrangian subproblem of the
        state-bounded NLP (include/lfsd_cpdp.h); the multiplier loop is COCSys.cocSolverBatch."""
        dt = ini_state.dtype
        B = ini_state.shape[0]
        n, m, p, nc = self.n_state, self.n_control, self.n_auxvar, self.n_const
        self._check(ini_state, (B, n), dt, "ini_state")
        self._check(horizon, (B,), dt, "horizon")
        self._check(auxvar, (B, p), dt, "auxvar")
        per_traj = 0
        if nc:
            if consts is None:
                raise LfsdError("consts is required (n_const=%d)" % nc)
            if consts.dim() == 2:
                self._check(consts, (B, nc), dt, "consts")
                per_traj = 1
            else:
                self._check(consts, (nc,), dt, "consts")
        else:
            consts = None
        self._check(u_init, (B, n_grid, m), dt, "u_init", optional=True)
        self._check(control_lb, (m,), dt, "control_lb", optional=True)
        self._check(control_ub, (m,), dt, "control_ub", optional=True)
        if (control_lb is None) != (control_ub is None):
            raise LfsdError("control_lb and control_ub go together")
        if state_lb is not None:
            self._check(state_lb, (n,), dt, "state_lb")
            self._check(state_ub, (n,), dt, "state_ub")
            self._check(state_mult, (B, n_grid, 2, n), dt, "state_mult")
            if control_lb is None:
                raise LfsdError("state bounds need the control-bound arrays beside them (+-1e20 where there is none)")
        dev = ini_state.device
        if out is None:
            out = dict(state_grid=torch.empty((B, n_grid + 1, n), dtype=dt, device=dev),
                       control_grid=torch.empty((B, n_grid + 1, m), dtype=dt, device=dev),
                       costate_grid=torch.empty((B, n_grid + 1, n), dtype=dt, device=dev),
                       cost=torch.empty((B,), dtype=dt, device=dev),
                       iters=torch.zeros((B,), dtype=torch.int32, device=dev),
                       status=torch.zeros((B,), dtype=torch.int32, device=dev))
        need = self.coc_workspace_bytes(dt, B, n_grid, exact_after, mapping, control_lb is not None)
        if workspace is None or workspace.numel() * workspace.element_size() < need:
            workspace = torch.empty((need + 7) // 8, dtype=torch.int64, device=dev)
        if tol is None:
            tol = 1e-6 if dt == torch.float32 else 1e-9
        args = (_DT[dt], B, n_grid, steps_per_grid, self._p(ini_state), self._p(horizon), self._p(auxvar),
                self._p(consts), per_traj, self._p(u_init), self._p(control_lb), self._p(control_ub),
                self._p(state_lb), self._p(state_ub), self._p(state_mult), float(state_rho),
                self._p(out["state_grid"]), self._p(out["control_grid"]),
                self._p(out["costate_grid"]), self._p(out["cost"]), self._p(out["iters"]), self._p(out["status"]),
                int(max_iter), float(tol), int(exact_after), MAPPINGS[mapping], self._p(workspace),
                workspace.numel() * workspace.element_size(), self._stream(ini_state))
        with self._on(ini_state):
            self._rc(self.lib.lfsd_coc_solve(*args), "lfsd_coc_solve")
        out["workspace"] = workspace
        return out

    def aux_solve(self, horizon, auxvar, consts, state_grid, control_grid, costate_grid, taus, waypoints, iface_idx,
                  substeps=0, want_grids=False, Z_grid=None, out=None, phase_hook=None, rtol=1e-3, oc_status=None,
                  skip_status=()):
        """``phase_hook(name)``, if given, is called before/after each of the two launches
        ("riccati", "forward") so a caller can bracket them with HIP events (bench.py).
        ``oc_status`` [B] int32 (the status the OC solve wrote) + ``skip_status`` (status values, e.g. (3, 4)): rows with
        one of these statuses are not differentiated -- NaN loss / gradient, no sweep (include/lfsd_cpdp.h, ABI 8)."""
        dt = state_grid.dtype
        B, N1, n = state_grid.shape
        N = N1 - 1
        m, p, nc = self.n_control, self.n_auxvar, self.n_const
        dev = state_grid.device
        self._check(horizon, (B,), dt, "horizon")
        self._check(auxvar, (B, p), dt, "auxvar")
        self._check(state_grid, (B, N + 1, self.n_state), dt, "state_grid")
        self._check(control_grid, (B, N + 1, m), dt, "control_grid")
        self._check(costate_grid, (B, N + 1, n), dt, "costate_grid")
        per_traj = 0
        if nc:
            if consts is None:
                raise LfsdError("consts is required (n_const=%d)" % nc)
            if consts.dim() == 2:
                self._check(consts, (B, nc), dt, "consts")
                per_traj = 1
            else:
                self._check(consts, (nc,), dt, "consts")
        else:
            consts = None
        nw = 0 if taus is None else taus.shape[1]
        ni = 0 if iface_idx is None else iface_idx.shape[0]
        if nw and iface_idx is None:      # the interface function compiled into the library (COCSys.setInterface)
            ni = self.n_interface
            if ni == 0:
                raise LfsdError("no interface_idx given and the model library carries no compiled interface function")
        if nw:
            self._check(taus, (B, nw), dt, "taus")
            self._check(waypoints, (B, nw, ni), dt, "waypoints")
            self._check(iface_idx, (ni,), torch.int32, "iface_idx", optional=True)
        if Z_grid is None:
            Z_grid = torch.empty((B, N + 1, n + p, n), dtype=dt, device=dev)
        if out is None:
            out = dict(loss=torch.zeros((B,), dtype=dt, device=dev), grad=torch.zeros((B, p), dtype=dt, device=dev))
        if out.get("stats") is None:
            # [B][4]: split units executed / intervals accepted above rtol, Riccati sweep | forward sweep (include/lfsd_cpdp.h)
            out["stats"] = torch.zeros((B, 4), dtype=torch.int32, device=dev)
        self._check(out["stats"], (B, 4), torch.int32, "stats")
        skip_mask = 0
        for st in skip_status:
            if not 0 <= int(st) < 31:
                raise LfsdError("skip_status values must be OC-solve statuses (0..30)")
            skip_mask |= 1 << int(st)
        if skip_mask or oc_status is not None:      # (without a mask the status still tells the sweeps which rows to budget)
            self._check(oc_status, (B,), torch.int32, "oc_status")
        auxX = auxU = None
        if want_grids:
            auxX = torch.empty((B, N + 1, p, n), dtype=dt, device=dev)
            auxU = torch.empty((B, N + 1, p, m), dtype=dt, device=dev)
        common = (_DT[dt], B, N, self._p(horizon), self._p(auxvar), self._p(consts), per_traj,
                  self._p(state_grid), self._p(control_grid), self._p(costate_grid), self._p(Z_grid))
        tail = (nw, ni, self._p(iface_idx), self._p(taus), self._p(waypoints), self._p(out["loss"]),
                self._p(out["grad"]), self._p(auxX), self._p(auxU), int(substeps), float(rtol), self._p(out["stats"]),
                self._p(oc_status), skip_mask, self._stream(state_grid))
        with self._on(state_grid):
            if phase_hook is None:
                self._rc(self.lib.lfsd_aux_solve(*common, *tail), "lfsd_aux_solve")
            else:
                phase_hook("riccati")
                self._rc(self.lib.lfsd_aux_riccati(*common, int(substeps), float(rtol), self._p(out["stats"]),
                                                   self._p(oc_status), skip_mask, self._stream(state_grid)), "lfsd_aux_riccati")
                phase_hook("forward")
                self._rc(self.lib.lfsd_aux_forward(*common, *tail), "lfsd_aux_forward")
                phase_hook("end")
        out["Z_grid"] = Z_grid
        out["auxX_grid"], out["auxU_grid"] = auxX, auxU
        return out

    def optimizer_step(self, method, theta, grad, iter_idx, lr, mu=0.9, beta1=0.9, beta2=0.999, eps=1e-8, m=None,
                       v=None, vhat=None, proj_lo=None, row_active=None):
        dt = theta.dtype
        B, p = theta.shape
        for nm, t in (("theta", theta), ("grad", grad)):
            self._check(t, (B, p), dt, nm)
        for nm, t in (("m", m), ("v", v), ("vhat", vhat)):
            self._check(t, (B, p), dt, nm, optional=True)
        self._check(proj_lo, (p,), dt, "proj_lo", optional=True)
        self._check(row_active, (B,), torch.int32, "row_active", optional=True)
        with self._on(theta):
            rc = self.lib.lfsd_optimizer_step(_DT[dt], OPT_METHODS[method] if isinstance(method, str) else int(method),
                                              B, p, int(iter_idx), float(lr), float(mu), float(beta1), float(beta2),
                                              float(eps), self._p(theta), self._p(grad), self._p(m), self._p(v),
                                              self._p(vhat), self._p(proj_lo), self._p(row_active),
                                              self._stream(theta))
        self._rc(rc, "lfsd_optimizer_step")

    def lookahead(self, theta, v, mu, out=None):
        if out is None:
            out = torch.empty_like(theta)
        with self._on(theta):
            rc = self.lib.lfsd_lookahead(_DT[theta.dtype], theta.numel(), float(mu), self._p(theta), self._p(v),
                                         self._p(out), self._stream(theta))
        self._rc(rc, "lfsd_lookahead")
        return out
