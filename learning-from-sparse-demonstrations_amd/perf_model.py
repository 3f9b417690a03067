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
    return out


def model_counts(spec):
    return _model_counts(spec.hash(), spec)


def kernel_flops(spec, kernel, n_grid, steps_per_grid, substeps, mean_iters=1.0, units_per_interval=None):
    """Useful flops of ONE trajectory in one launch of `kernel` ("oc_solve" | "aux_riccati" | "aux_forward").
    oc_solve: `mean_iters` solver iterations, each = tangent roll-out (RK4, 4 stages x S steps x N intervals, one
    uniform evaluation + NX+NU tangent columns) + backward sweep (V_xx m, [A B]^T Y dense products, stage Hessian
    column, gains) -- plus the initial roll-out.  aux kernels: per interval `units` split units (default = substeps),
    each 12 right-hand sides (coarse RK4 + two fine RK4 steps) per column + 5 coefficient nodes."""
    c = model_counts(spec)
    n, m, p = spec.n, spec.m, spec.p
    nxu, N, S = n + m, n_grid, steps_per_grid
    rk = 4 * (n + 1)                                                   # RK4 combination per stage, state + cost
    if kernel == "oc_solve":
        rollout = N * S * 4 * (c["jvp_uniform"] + 2 * rk + nxu * (c["jvp_column"] + 2 * rk)) + N * 2 * n * m
        backward = N * (2 * n * n * nxu + 2 * nxu * n * nxu            # Y = V_xx m, Q = [A B]^T Y
                        + nxu * c["ham_column"]                         # stage Hessian model, one column each
                        + 2 * n * nxu + m * m * m // 3 + 4 * m * m * nxu   # Q_u / lambda, Cholesky, gains K, k
                        + n * (4 * m * n + 2 * m * m))                  # V_xx update
        return (1.0 + mean_iters) * rollout + max(mean_iters, 1.0) * backward
    nn = c["nnz"]
    units = units_per_interval or substeps
    if kernel == "aux_riccati":
        cols = n + p
        rhs = 2 * (2 * nn["fx"] + 2 * nn["fu"] + nn["Hxx"] + 2 * nn["Hxu"] + nn["fe"] + nn["Hxe"] + nn["Hue"]) // 1 \
            + 2 * m * m * 2 + 2 * n
        stiff = 2 * (2 * nn["fu"]) + 2 * n * m + m * m * m
        per_unit = 5 * (c["pmp_coeffs"] + m ** 3) / cols + 12 * (rhs + 3 * n) + 5 * stiff
        return N * units * cols * per_unit
    if kernel == "aux_forward":
        cols = p
        rhs = 2 * (nn["fx"] + nn["Hxu"] + nn["fu"]) + 2 * m * m + n
        stiff = 2 * n * m + 2 * m * m + 2 * nn["fu"]
        prep = 3 * (2 * nn["fu"] + 2 * m * m) * n / cols + 3 * 40 * m ** 3 / cols
        per_unit = 5 * (c["pmp_coeffs"] + m ** 3) / cols + prep + 12 * (rhs + 3 * n) + 8 * stiff \
            + 5 * (2 * nn["fu"] + 2 * nn["fe"] + 2 * m * m)
        return N * units * cols * per_unit
    raise KeyError(kernel)
