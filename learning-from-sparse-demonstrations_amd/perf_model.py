"""Useful-flop model of the three kernels (bench.py's ``roofline.valu_useful_tflops`` / ``valu_frac``).

"Useful" = arithmetic the algorithm needs, counted on the same CSE'd expressions the code generator emits
(codegen.py) times how often each kernel evaluates them, for the columns that exist (NX+NU tangent columns, NX+NP
Riccati columns, NP forward columns) -- not for idle lanes, not for operand moves.  One FMA counts as 2.
The model is deliberately simple: it is the numerator of a utilisation figure, good to ~10 %.
"""
import functools

import sympy as sp


def _ops(e):
    """flops of one CSE'd right-hand side"""
    n = 0
    for node in sp.preorder_traversal(e):
        if isinstance(node, (sp.Add, sp.Mul)):
            n += len(node.args) - 1
        elif isinstance(node, sp.Pow):
            ex = node.exp
            n += (abs(int(ex)) - 1 + (1 if ex < 0 else 0)) if ex.is_Integer else 4
        elif isinstance(node, sp.Function):
            n += 4                                   # sin / cos / sqrt ...: quarter-rate instruction
    return n


def _count(exprs, tangent=()):
    """(flops not depending on `tangent` symbols, flops depending on them) of a jointly CSE'd expression list"""
    repl, red = sp.cse([sp.sympify(e) for e in exprs], order='none')
    vt = set(tangent)
    uni = vec = 0
    for s, r in repl:
        k = _ops(r)
        if r.free_symbols & vt:
            vt.add(s)
            vec += k
        else:
            uni += k
    for r in red:
        k = _ops(r)
        if r.free_symbols & vt:
            vec += k
        else:
            uni += k
    return uni, vec


@functools.lru_cache(maxsize=None)
def _model_counts(spec_hash, spec):
    n, m, p = spec.n, spec.m, spec.p
    X, U, E, L = sp.Matrix(spec.X), sp.Matrix(spec.U), sp.Matrix(spec.E), sp.Matrix(spec.L)
    f, c = spec.f, spec.c
    fx, fu, fe = f.jacobian(X), f.jacobian(U), f.jacobian(E)
    cx, cu = sp.Matrix([c]).jacobian(X), sp.Matrix([c]).jacobian(U)
    H = c + (f.T * L)[0, 0]
    Hx = sp.Matrix([H]).jacobian(X).T
    Hu = sp.Matrix([H]).jacobian(U).T
    Hxx, Hxu, Hxe, Huu, Hue = Hx.jacobian(X), Hx.jacobian(U), Hx.jacobian(E), Hu.jacobian(U), Hu.jacobian(E)
    dX = sp.Matrix([sp.Symbol('dx%d' % i, real=True) for i in range(n)])
    dU = sp.Matrix([sp.Symbol('du%d' % i, real=True) for i in range(m)])
    tang = list(dX) + list(dU)
    out = {}
    out["dyn_cost"] = sum(_count(list(f) + [c]))
    out["jvp_uniform"], out["jvp_column"] = _count(list(f) + [c] + list(fx * dX + fu * dU) +
                                                   [(cx * dX)[0, 0] + (cu * dU)[0, 0]], tang)
    _, out["ham_column"] = _count(list(Hxx * dX + Hxu * dU) + list(Hxu.T * dX + Huu * dU), tang)
    out["pmp_coeffs"] = sum(_count([e for M in (fx, fu, fe, Hxx, Hxu, Hxe, Hue, Huu) for e in M if e != 0]))
    nz = lambda M: sum(1 for e in M if e != 0)
    out["nnz"] = {k: nz(M) for k, M in (("fx", fx), ("fu", fu), ("fe", fe), ("Hxx", Hxx), ("Hxu", Hxu), ("Hxe", Hxe),
                                        ("Hue", Hue))}
    out["nnz_cx"], out["nnz_cu"] = nz(cx), nz(cu)
    # structurally constant leading tangent columns (same rule as codegen.emit_header)
    cxx, cxu = cx.jacobian(X), cx.jacobian(U)
    k = 0
    for i in range(n):
        if any(fx[r, i] != 0 for r in range(n)):
            break
        k += 1
    while k > 0 and not (all(cxx[i, j] == 0 for i in range(k) for j in range(k, n)) and
                         all(cxu[i, a] == 0 for i in range(k) for a in range(m))):
        k -= 1
    out["nzc"] = k
    return out


def model_counts(spec):
    return _model_counts(spec.hash(), spec)


def kernel_flops(spec, kernel, n_grid, steps_per_grid, substeps, mean_iters=1.0, units_per_interval=None, midpoint=False,
                 coarse_rollouts=0, split=False, level0=None):
    """Useful flops of ONE trajectory in one launch of `kernel` ("oc_solve" | "aux_riccati" | "aux_forward").

    Round 3: the figures are checked against the counters (SQ_INSTS_VALU_FLOPS_FP32 x 64 lanes x EXEC share,
    profiles/r03_issue_counters.json, `valu_flops_executed_per_launch`) times the share of the executing lanes that carry
    needed work, and were corrected where they disagreed by more than 20 %: the round-2 model counted the RK4 combination
    twice and took the tangent operations from sympy's CSE of the directional derivative, which keeps group-uniform
    coefficient products inside the per-column count (2.2x too many for the quadrotor's roll-out).  A tangent column now
    costs what the linear map costs: one FMA per structural non-zero of [f_x f_u; c_x c_u] plus the RK4 combination.

    oc_solve: `mean_iters` solver iterations, each = tangent roll-out (RK4, 4 stages x S steps x N intervals, one uniform
    evaluation + the LIVE tangent columns: structurally constant columns are not computed, codegen NZC) + backward sweep
    (dense products V_xx [A B], [A B]^T Y -- on the matrix cores in the lean fp32 kernel, reported apart with split=True --
    stage Hessian column, gains, V_xx update) -- plus the initial roll-out.  `coarse_rollouts` of the 1 + mean_iters
    roll-outs run with ONE RK4 step per interval (mesh continuation; 5 on the benchmark).  `level0` = (iterations, merged
    intervals, RK4 steps per merged interval) of the lean kernels' level 0 (round 4: 3, 2, 1): the initial roll-out and those
    iterations' roll-outs and sweeps run on n_grid / merged stages, and ONE of the counted iterations is the transfer to the
    next level -- a roll-out (counted in `coarse_rollouts`) without a backward sweep.
    aux kernels: per interval `units_per_interval` split units (default `substeps`; bench.py passes the measured mean of the
    error-controlled sweeps), each 12 right-hand sides (coarse RK4 + two fine RK4 steps; 6 with the explicit midpoint rule of
    the fp32 kernels, `midpoint`) per column + 5 coefficient nodes."""
    c = model_counts(spec)
    n, m, p = spec.n, spec.m, spec.p
    nxu, N, S = n + m, n_grid, steps_per_grid
    nn = c["nnz"]
    if kernel == "oc_solve":
        live = nxu - c.get("nzc", 0)
        glue = 4 * (n + 1)                                             # x_s = x + a f, acc += w f: two FMAs per component
        col = 2 * (nn["fx"] + nn["fu"] + c["nnz_cx"] + c["nnz_cu"]) + glue      # one tangent column, one RK4 stage
        uni = c["jvp_uniform"] + glue                                  # nominal f, c and the Jacobian coefficients, once
        rollout = N * S * 4 * (uni + live * col) + N * 2 * n * m       # (+ closed-loop control)
        dense = N * (2 * n * n * live + 2 * live * n * live)           # Y = V_xx M_live, Q = M_live^T Y
        backward = N * (nxu * c["ham_column"]                           # stage Hessian model, one column each
                        + 4 * n * nxu + m * m * m // 3 + 4 * m * m * nxu      # Q_u / lambda, Cholesky, gains K, k
                        + n * (4 * m * n + 2 * m * m) + n * n)          # V_xx update + symmetrisation
        n_ro = 1.0 + mean_iters
        if level0 and S > 1:
            k0, tc0, s0 = level0
            n0 = min(1.0 + k0, n_ro)                                   # roll-outs on level 0
            rollout0 = (N / tc0) * s0 * 4 * (uni + live * col) + (N / tc0) * 2 * n * m
            nc = min(float(coarse_rollouts), n_ro - n0)
            sweeps = max(mean_iters - 1.0, 1.0)                        # (the transfer iteration has none)
            sw0 = min(float(k0), sweeps)
            valu = n0 * rollout0 + nc * rollout / S + (n_ro - n0 - nc) * rollout + (sweeps - sw0) * backward + sw0 * backward / tc0
            mfma = (sweeps - sw0) * dense + sw0 * dense / tc0
            return (valu, mfma) if split else valu + mfma
        nc = min(float(coarse_rollouts), n_ro) if S > 1 else 0.0
        valu = (n_ro - nc) * rollout + nc * rollout / S + max(mean_iters, 1.0) * backward
        mfma = max(mean_iters, 1.0) * dense
        return (valu, mfma) if split else valu + mfma
    units = units_per_interval or substeps
    nrhs = 6 if midpoint else 12
    if kernel == "aux_riccati":
        cols = n + p
        # one right-hand side, one column -- as implemented (cpdp_aux.h ric_rhs): fx^T z, fu^T z, Huu^-1 (.), Hxu w; the P
        # columns also produce their row of P [A r] (Hxu nv, fe^T z, Hue^T nv), which the other lanes read back transposed
        rhs = 2 * (nn["fx"] + nn["fu"] + nn["Hxu"]) + 2 * m * m + (n / cols) * 2 * (nn["Hxu"] + nn["fe"] + nn["Hue"]) + 2 * n
        stiff = 2 * nn["fu"] + 2 * m * m + 2 * n * m + (2 * m * nn["fu"] + m * m * m) / cols
        per_unit = 5 * (c["pmp_coeffs"] + m ** 3) / cols + nrhs * (rhs + 3 * n) + 5 * stiff
        v = N * units * cols * per_unit
        return (v, 0.0) if split else v
    if kernel == "aux_forward":
        cols = p
        rhs = 2 * (nn["fx"] + nn["Hxu"] + nn["fu"]) + 2 * m * m + n
        stiff = 2 * n * m + 2 * m * m + 2 * nn["fu"]
        prep = 3 * (2 * nn["fu"] + 2 * m * m) * n / cols + 3 * 40 * m ** 3 / cols
        per_unit = 5 * (c["pmp_coeffs"] + m ** 3) / cols + prep + nrhs * (rhs + 3 * n) + 8 * stiff \
            + 5 * (2 * nn["fu"] + 2 * nn["fe"] + 2 * m * m)
        v = N * units * cols * per_unit
        return (v, 0.0) if split else v
    raise KeyError(kernel)
