"""sympy -> HIP device code for one optimal-control model.

Replaces what CasADi does for the reference at CPDP/CPDP.py:49-79 (Function
objects for dyn / cost / final cost and their Jacobians) and CPDP.py:201-248
(``diffPMP``: first and second derivatives of the Hamiltonian) — but instead of
an interpreted expression graph evaluated one trajectory at a time on the CPU,
the derivatives are emitted as straight-line, CSE'd, *sparse* device code that
the kernels in ``csrc/cpdp_kernels.h`` inline.

Emitted per model (``struct Model`` in namespace ``lfsd_gen``):

* ``dyn_cost``        f(x,u), c(x,u)                       (line-search roll-outs)
* ``dyn_cost_jvp``    f, c and their directional derivative along a per-lane
                      tangent (dx,du): the RK4 sensitivity sweep of the shooting map
* ``final_cost / final_grad / final_hess_mul``
* ``ham_hess_mul``    [Hxx Hxu; Hux Huu] applied to a per-lane vector (DDP backward)
* ``ham_huu``         the dense control block Huu alone (MFMA backward sweep)
* ``pmp_coeffs``      every structurally non-zero entry of fx, fu, fe, Hxx, Hxu,
                      Hxe, Hue (+ dense Huu) packed into one array that is staged
                      in LDS, and ``<mat>_mul / <mat>_mulT`` operators that apply
                      those packed matrices with compile-time offsets (auxiliary
                      Riccati / sensitivity pass, CPDP.py:253-298).
"""
import hashlib

import sympy as sp
from sympy.printing.c import C99CodePrinter

CODEGEN_VERSION = 18


class ModelSpec:
    def __init__(self, state, control, auxvar, consts, time, dyn, path_cost, final_cost,
                 time_varying=False, const_defaults=None, name="model", interface=None):
        self.state, self.control, self.auxvar, self.consts = list(state), list(control), list(auxvar), list(consts)
        self.time = time
        self.dyn = sp.Matrix(dyn)
        self.path_cost = sp.sympify(path_cost)
        self.final_cost = sp.sympify(final_cost)
        self.time_varying = bool(time_varying)
        self.const_defaults = list(const_defaults or [0.0] * len(self.consts))
        self.name = name
        # interface function y = g(x) of the sparse-demonstration loss (lib/QuadAlgorithm.py:616-639: an arbitrary CasADi
        # expression of the state; None: the loss selects state components by index, which every example does)
        self.interface = None if interface is None else sp.Matrix(list(interface))
        self._canon()

    def _canon(self):
        n, m, p, nc = len(self.state), len(self.control), len(self.auxvar), len(self.consts)
        self.X = [sp.Symbol('x%d' % i, real=True) for i in range(n)]
        self.U = [sp.Symbol('u%d' % i, real=True) for i in range(m)]
        self.E = [sp.Symbol('e%d' % i, real=True) for i in range(p)]
        self.C = [sp.Symbol('c%d' % i, real=True) for i in range(nc)]
        self.L = [sp.Symbol('l%d' % i, real=True) for i in range(n)]
        self.Tt = sp.Symbol('tt', real=True)
        sub = dict(zip(self.state, self.X))
        sub.update(zip(self.control, self.U))
        sub.update(zip(self.auxvar, self.E))
        sub.update(zip(self.consts, self.C))
        sub[self.time] = self.Tt
        self.f = self.dyn.xreplace(sub)
        self.c = self.path_cost.xreplace(sub)
        self.h = self.final_cost.xreplace(sub)
        self.g = None if self.interface is None else self.interface.xreplace(sub)
        if self.g is not None and (self.g.free_symbols - set(self.X + self.C)):
            raise ValueError("the interface function may depend on the state (and constants) only: %s"
                             % sorted(map(str, self.g.free_symbols - set(self.X + self.C))))
        allowed = set(self.X + self.U + self.E + self.C + [self.Tt])
        stray = (self.f.free_symbols | self.c.free_symbols | self.h.free_symbols) - allowed
        if stray:
            raise ValueError("model expressions contain undeclared symbols: %s" % sorted(map(str, stray)))
        if self.h.free_symbols & set(self.U):
            raise ValueError("final cost must not depend on the control")
        self.n, self.m, self.p, self.nc = n, m, p, nc

    def hash(self):
        # structure and expressions only.  Name and constant DEFAULTS are deliberately left out: a driver such as
        # lib/QuadAlgorithm.py builds the same model with another goal (a constant's default) for every run, and that must
        # not cost a recompilation.  Consequently lfsd_get_model_info().name and lfsd_const_default() report the values of
        # whichever instance generated the header first -- informational only: COCSys always passes its own instance's
        # constant values to the kernels (COCSys.const_values / consts_tensor), never the library's defaults.
        key = "v%d|%d %d %d %d %d|%s|%s|%s" % (CODEGEN_VERSION, self.n, self.m, self.p, self.nc, self.time_varying,
                                               sp.srepr(self.f), sp.srepr(self.c), sp.srepr(self.h))
        if self.g is not None:
            key += "|g:" + sp.srepr(self.g)
        return hashlib.sha1(key.encode()).hexdigest()[:16]


class _HipPrinter(C99CodePrinter):
    def _print_Float(self, e):
        return 'T(%r)' % float(e)

    def _print_Integer(self, e):
        return 'T(%d)' % int(e)

    def _print_Rational(self, e):
        return '(T(%d)/T(%d))' % (e.p, e.q)

    def _print_Pi(self, e):
        return 'T(3.141592653589793)'

    def _print_sin(self, e):
        return 'lfsd::t_sin(%s)' % self._print(e.args[0])

    def _print_cos(self, e):
        return 'lfsd::t_cos(%s)' % self._print(e.args[0])

    def _print_Pow(self, e):
        b, ex = e.base, e.exp
        if ex.is_Integer:
            k = int(ex)
            bs = self.parenthesize(b, 1000)
            if k == -1:
                return '(T(1)/%s)' % bs
            if 0 < k <= 4:
                return '(' + '*'.join([bs] * k) + ')'
            if -4 <= k < 0:
                return '(T(1)/(' + '*'.join([bs] * (-k)) + '))'
        if ex == sp.Rational(1, 2):
            return 'sqrt(%s)' % self._print(b)
        if ex == sp.Rational(-1, 2):
            return '(T(1)/sqrt(%s))' % self._print(b)
        return 'pow(%s, %s)' % (self._print(b), self._print(ex))


_P = _HipPrinter()


def _pr(e):
    return _P.doprint(e)


def _body(inputs, outputs, indent='    ', vec=(), vec_out=None):
    """inputs: list of (symbol, c_expr_to_load); outputs: list of (c_lvalue, sympy_expr).
    Returns C statements computing all outputs with common sub-expression elimination.

    vec: input symbols of the tangent type `V` (a scalar `T`, or a pair of them that the packed-math kernels carry per
    lane); every temporary that depends on one is typed `V` as well.  The expressions are linear in those symbols, so
    only T*V, V+V and -V occur.  vec_out(lvalue) -> True marks outputs that are `V` even when identically constant."""
    exprs = [sp.sympify(e) for _, e in outputs]
    repl, red = sp.cse(exprs, symbols=sp.numbered_symbols('w_'), order='none')
    used = set()
    for _, r in repl:
        used |= r.free_symbols
    for r in red:
        used |= r.free_symbols
    vt = set(vec)
    lines = []
    for sym, load in inputs:
        if sym in used:
            lines.append('%sconst %s %s = %s;' % (indent, 'V' if sym in vt else 'T', sym, load))
    for s, r in repl:
        isv = bool(r.free_symbols & vt)
        if isv:
            vt.add(s)
        lines.append('%sconst %s %s = %s;' % (indent, 'V' if isv else 'T', s, _pr(r)))
    for (lv, _), r in zip(outputs, red):
        if vec_out is not None and vec_out(lv) and not (r.free_symbols & vt):
            lines.append('%s%s = V(%s);' % (indent, lv, _pr(r)))
        else:
            lines.append('%s%s = %s;' % (indent, lv, _pr(r)))
    return '\n'.join(lines)


def _body_multi(inputs, outputs, vec, indent='    '):
    """The same straight-line code for NV tangents at once (template parameter NV of the emitted function): every statement that
    does not depend on a tangent symbol -- the NOMINAL part: for the robot arm four sin / cos, the mass matrix and its inverse,
    the larger part of a call -- is emitted once, the tangent-dependent statements inside `for (v < NV)`.  (Calling the
    one-tangent function NV times from one basic block does not get there: measured on gfx950, the compiler keeps a copy of the
    nominal part per call.)  inputs: (symbol, load) with `v` free in the loads of tangent symbols; outputs: (lvalue, expr) with `v`
    free in tangent lvalues; vec: the tangent symbols."""
    exprs = [sp.sympify(e) for _, e in outputs]
    repl, red = sp.cse(exprs, symbols=sp.numbered_symbols('w_'), order='none')
    used = set()
    for _, r in repl:
        used |= r.free_symbols
    for r in red:
        used |= r.free_symbols
    vt = set(vec)
    nom, tan = [], []
    for sym, load in inputs:
        if sym in used:
            (tan if sym in vt else nom).append('const %s %s = %s;' % ('V' if sym in vt else 'T', sym, load))
    for s_, r in repl:
        isv = bool(r.free_symbols & vt)
        if isv:
            vt.add(s_)
        (tan if isv else nom).append('const %s %s = %s;' % ('V' if isv else 'T', s_, _pr(r)))
    for (lv, _), r in zip(outputs, red):
        if 'v' in lv.replace('dv', ''):           # a per-tangent output
            tan.append('%s = %s;' % (lv, _pr(r)) if (r.free_symbols & vt) else '%s = V(%s);' % (lv, _pr(r)))
        else:
            assert not (r.free_symbols & vt), lv
            nom.append('%s = %s;' % (lv, _pr(r)))
    lines = [indent + l for l in nom]
    lines.append('#pragma unroll')
    lines.append(indent + 'for (int v = 0; v < NV; ++v) {')
    lines += [indent + '  ' + l for l in tan]
    lines.append(indent + '}')
    return '\n'.join(lines)


def _loads(spec, with_u=True, with_l=False):
    ins = [(spec.Tt, 't')]
    ins += [(s, 'x[%d]' % i) for i, s in enumerate(spec.X)]
    if with_u:
        ins += [(s, 'u[%d]' % i) for i, s in enumerate(spec.U)]
    if with_l:
        ins += [(s, 'l[%d]' % i) for i, s in enumerate(spec.L)]
    ins += [(s, 'e[%d]' % i) for i, s in enumerate(spec.E)]
    ins += [(s, 'c[%d]' % i) for i, s in enumerate(spec.C)]
    ins += [(s, 'c[%d]' % (max(spec.nc, 1) + j)) for j, (s, _) in enumerate(getattr(spec, '_derived', ()))]
    return ins


def _has_division(e):
    return any(isinstance(q, sp.Pow) and q.exp.is_negative for q in sp.preorder_traversal(e))


def _hoist_const_quotients(exprs, consts):
    """Quotients of the model's run-time constants (1/m, (Jy - Jz)/Jx, ...) are the same for every call of a solve, but
    the compiler cannot hoist them out of the kernels' loops (the constants live in LDS, across barriers), and an IEEE
    division is a ten-instruction dependent sequence.  Every maximal sub-expression that depends on constants only and
    contains a division is replaced by a DERIVED constant cd<j>; the kernels evaluate Model::derive_consts once per
    trajectory when they stage the constants.  Returns (new expressions, [(symbol, expression)])."""
    cset = set(consts)
    table = {}

    def sym_for(e):
        if e not in table:
            table[e] = sp.Symbol('cd%d' % len(table), real=True)
        return table[e]

    def rec(e):
        if e.is_Atom:
            return e
        fs = e.free_symbols
        if fs and fs <= cset:
            return sym_for(e) if _has_division(e) else e
        if isinstance(e, sp.Mul):
            cargs = [a for a in e.args if a.free_symbols and a.free_symbols <= cset]
            if cargs:
                cm = sp.Mul(*cargs)
                if _has_division(cm):
                    rest = [a if not a.free_symbols else rec(a) for a in e.args if not (a.free_symbols and a.free_symbols <= cset)]
                    return sp.Mul(sym_for(cm), *rest)
        return e.func(*[rec(a) for a in e.args])

    out = [x.applyfunc(rec) if isinstance(x, sp.MatrixBase) else rec(sp.sympify(x)) for x in exprs]
    return out, [(v, k) for k, v in sorted(table.items(), key=lambda kv: int(str(kv[1])[2:]))]


def _nz(e):
    return sp.sympify(e) != 0


class _Sparse:
    """Structurally sparse matrix packed into the coefficient array, start aligned to 16 bytes.  Two packings: layout 0
    (Riccati sweep) and layout 1 (forward sweep) each store the matrix in the order ('r'ow- or 'c'olumn-major) in which
    that kernel's hot operator walks it, so the reads are contiguous LDS words -- and each at its own offset: a layout
    only holds the matrices its kernel applies ('-' in `orders`: not staged; the forward sweep never touches Hxx / Hxe)."""

    def __init__(self, name, mat, offs, orders):
        self.name, self.mat = name, mat
        rc = [(r, c) for r in range(mat.shape[0]) for c in range(mat.shape[1]) if _nz(mat[r, c])]
        self.nnz = len(rc)
        self.lay, self.off, self.end = [], [], []
        for off, o in zip(offs, orders):
            if o == '-' or off is None:
                self.lay.append(None); self.off.append(None); self.end.append(None)
                continue
            assert off % 4 == 0
            seq = rc if o == 'r' else sorted(rc, key=lambda t: (t[1], t[0]))
            self.lay.append([(r, c, off + i, mat[r, c]) for i, (r, c) in enumerate(seq)])
            self.off.append(off)
            self.end.append(off + len(rc))

    def stores(self, lay):
        return [('L[%d]' % o, ex) for (_, _, o, ex) in (self.lay[lay] or [])]

    def _lines(self, lay, transposed):
        if self.lay[lay] is None:
            return '      static_assert(LAY != %d, "%s is not staged in this layout");' % (lay, self.name)
        nr, ncol = self.mat.shape
        rows = {}
        for (r, cc, o, _) in sorted(self.lay[lay], key=lambda t: t[2]):
            i, k = (cc, r) if transposed else (r, cc)
            rows.setdefault(i, []).append('L[%d]*v[%d]' % (o, k))
        nout = ncol if transposed else nr
        lines = []
        for i in range(nout):
            terms = rows.get(i)
            if terms:
                lines.append('      y[%d] = (ACC ? y[%d] : T(0)) + %s;' % (i, i, ' + '.join(terms)))
            else:
                lines.append('      if (!ACC) y[%d] = T(0);' % i)
        return '\n'.join(lines)

    def emit_ops(self):
        nr, ncol = self.mat.shape
        out = []
        for suffix, transposed in (('mul', False), ('mulT', True)):
            nout = ncol if transposed else nr
            out.append('  // y[%d] (+)= %s%s * v[%d]   (%d non-zeros; LAY picks the packing the calling kernel staged)\n'
                       '  template<bool ACC, int LAY, class T> static LFSD_DEV void %s_%s(const T* L, const T* v, T* y) {\n'
                       '    if constexpr (LAY == 0) {\n%s\n    } else {\n%s\n    }\n  }'
                       % (nout, self.name, "^T" if transposed else "", nr if transposed else ncol, self.nnz,
                          self.name, suffix, self._lines(0, transposed), self._lines(1, transposed)))
        return '\n'.join(out)


# packing order per matrix: (layout 0 = Riccati sweep, layout 1 = forward sweep)
_ORDERS = {'fx': 'cr', 'fu': 'cr', 'fe': 'cr', 'Hxx': 'r-', 'Hxu': 'rc', 'Hxe': 'r-', 'Hue': 'cr'}


def emit_header(spec):
    n, m, p, nc = spec.n, spec.m, spec.p, spec.nc
    X, U, E, L = sp.Matrix(spec.X), sp.Matrix(spec.U), sp.Matrix(spec.E), sp.Matrix(spec.L)
    (f, c, h), derived = _hoist_const_quotients([spec.f, spec.c, spec.h], spec.C)
    spec._derived = derived
    fx, fu, fe = f.jacobian(X), f.jacobian(U), f.jacobian(E)
    cx, cu = sp.Matrix([c]).jacobian(X), sp.Matrix([c]).jacobian(U)
    H = c + (f.T * L)[0, 0]
    Hx = sp.Matrix([H]).jacobian(X).T
    Hu = sp.Matrix([H]).jacobian(U).T
    Hxx, Hxu, Hxe = Hx.jacobian(X), Hx.jacobian(U), Hx.jacobian(E)
    Huu, Hue = Hu.jacobian(U), Hu.jacobian(E)
    hx = sp.Matrix([h]).jacobian(X).T
    hxx, hxe = hx.jacobian(X), hx.jacobian(E)

    dX = sp.Matrix([sp.Symbol('dx%d' % i, real=True) for i in range(n)])
    dU = sp.Matrix([sp.Symbol('du%d' % i, real=True) for i in range(m)])
    oE = sp.Matrix([sp.Symbol('oe%d' % i, real=True) for i in range(p)])
    tang = [(s, 'dx[%d]' % i) for i, s in enumerate(dX)] + [(s, 'du[%d]' % i) for i, s in enumerate(dU)]
    tang_e = [(s, 'dx[%d]' % i) for i, s in enumerate(dX)] + [(s, 'oe[%d]' % i) for i, s in enumerate(oE)]

    S = []
    S.append('// AUTO-GENERATED by codegen.py (version %d) — do not edit.  model "%s" hash %s'
             % (CODEGEN_VERSION, spec.name, spec.hash()))
    S.append('#pragma once')
    # one namespace per model: several model libraries live in one process and must not share
    # weak/unique symbols (template instantiations, function-local statics)
    S.append('#define LFSD_MODEL_NS lfsd_gen_%s' % spec.hash())
    S.append('namespace LFSD_MODEL_NS {')
    S.append('struct Model {')
    S.append('  static constexpr int NX = %d, NU = %d, NP = %d, NC = %d;' % (n, m, p, max(nc, 1)))
    S.append('  static constexpr int NC_REAL = %d;' % nc)
    S.append('  // derived constants (quotients of the run-time constants): c[NC .. NCX), filled by derive_consts at staging')
    S.append('  static constexpr int ND = %d, NCX = NC + ND;' % len(derived))
    S.append('  template<class T> static LFSD_DEV void derive_consts(T* c) {')
    if derived:
        S.append(_body([(s_, 'c[%d]' % i) for i, s_ in enumerate(spec.C)],
                       [('c[%d]' % (max(nc, 1) + j), ex) for j, (_, ex) in enumerate(derived)]))
    else:
        S.append('    (void)c;')
    S.append('  }')
    S.append('  static constexpr bool TIME_VARYING = %s;' % ('true' if spec.time_varying else 'false'))
    S.append('  static const char* name() { return "%s"; }' % spec.name)
    S.append('  static const char* hash() { return "%s"; }' % spec.hash())
    S.append('  static double const_default(int i) { static const double v[%d] = {%s}; return v[i]; }'
             % (max(nc, 1), ', '.join(repr(float(v)) for v in (spec.const_defaults or [0.0])) if nc else '0.0'))

    sig_xu = 'T t, const T* x, const T* u, const T* e, const T* c'
    # 0. structurally constant tangent columns.  The leading NZC state components enter neither the dynamics (f_x e_i == 0:
    #    position of the quadrotor / rocket, cart position) nor any mixed second derivative of the running cost with the
    #    other variables.  Their columns of the shooting sensitivity [A_k B_k] are exact unit vectors through every RK4
    #    stage -- only their entry of the cost row q_k is non-trivial, and that is an RK4 quadrature of dc/dx_i along the
    #    nominal (cost_grad_zc) -- and their rows / columns of the stage Hessian couple to nothing else.  The lean OC kernel
    #    neither propagates nor stores nor multiplies them (cpdp_oc.h, backward_sc).
    nzc = 0
    for i in range(n):
        if any(_nz(fx[r, i]) for r in range(n)):
            break
        nzc += 1
    cxx, cxu = cx.jacobian(X), cx.jacobian(U)

    def _separable(k):
        return all(not _nz(cxx[i, j]) for i in range(k) for j in range(k, n)) and \
            all(not _nz(cxu[i, a]) for i in range(k) for a in range(m))
    while nzc > 0 and not _separable(nzc):
        nzc -= 1
    S.append('  // leading state components with f_x e_i == 0 and no mixed cost curvature with the rest: structurally constant tangent columns')
    S.append('  static constexpr int NZC = %d;' % nzc)
    S.append('  // cz[i] = dc/dx_i, i < NZC')
    S.append('  template<class T> static LFSD_DEV void cost_grad_zc(%s, T* cz) {' % sig_xu)
    if nzc:
        S.append(_body(_loads(spec), [('cz[%d]' % i, cx[0, i]) for i in range(nzc)]))
    else:
        S.append('    (void)t; (void)x; (void)u; (void)e; (void)c; (void)cz;')
    S.append('  }')
    # 1. dynamics + running cost
    S.append('  template<class T> static LFSD_DEV void dyn_cost(%s, T* f, T& q) {' % sig_xu)
    S.append(_body(_loads(spec), [('f[%d]' % i, f[i]) for i in range(n)] + [('q', c)]))
    S.append('  }')
    # 2. + directional derivative
    df = fx * dX + fu * dU
    dq = (cx * dX)[0, 0] + (cu * dU)[0, 0]
    # V = T: one tangent column per lane; V = lfsd::pk2<T>: two columns per lane on packed math (v_pk_fma_f32)
    S.append('  template<class T, class V> static LFSD_DEV void dyn_cost_jvp(%s, const V* dx, const V* du, T* f, T& q, V* df, V& dq) {' % sig_xu)
    S.append(_body(_loads(spec) + tang, [('f[%d]' % i, f[i]) for i in range(n)] + [('q', c)] +
                   [('df[%d]' % i, df[i]) for i in range(n)] + [('dq', dq)],
                   vec=[s_ for s_, _ in tang], vec_out=lambda lv: lv.startswith('df[') or lv == 'dq'))
    S.append('  }')
    # 2b. two vector-Jacobian products at one point (second-order adjoint sweep through the RK4 stages):
    #     y1x = fx^T v1 + w1*cx   (first-order stage adjoint, group-uniform)
    #     y2x = fx^T v2, y2u = fu^T v2   (its per-lane tangent)
    V1 = sp.Matrix([sp.Symbol('va%d' % i, real=True) for i in range(n)])
    V2 = sp.Matrix([sp.Symbol('vb%d' % i, real=True) for i in range(n)])
    W1 = sp.Symbol('wa', real=True)
    y1x = fx.T * V1 + W1 * cx.T
    y2x = fx.T * V2
    y2u = fu.T * V2
    vin = [(s_, 'v1[%d]' % i) for i, s_ in enumerate(V1)] + [(s_, 'v2[%d]' % i) for i, s_ in enumerate(V2)] + [(W1, 'w1')]
    # (V as in dyn_cost_jvp: the tangent v2 and its two products may be packed pairs; v1, w1, y1x are group-uniform scalars)
    S.append('  template<class T, class V> static LFSD_DEV void dyn_vjp2(%s, const T* v1, T w1, const V* v2, T* y1x, V* y2x, V* y2u) {' % sig_xu)
    S.append(_body(_loads(spec) + vin, [('y1x[%d]' % i, y1x[i]) for i in range(n)] +
                   [('y2x[%d]' % i, y2x[i]) for i in range(n)] + [('y2u[%d]' % i, y2u[i]) for i in range(m)],
                   vec=list(V2), vec_out=lambda lv: lv.startswith('y2x[') or lv.startswith('y2u[')))
    S.append('  }')
    # 2c. the same two for NV tangents at once, nominal part shared (wide kernel, all columns of an interval on one lane)
    tang_n = [(s_, 'dx[v * %d + %d]' % (n, i)) for i, s_ in enumerate(dX)] + [(s_, 'du[v * %d + %d]' % (m, i)) for i, s_ in enumerate(dU)]
    S.append('  template<int NV, class T, class V> static LFSD_DEV void dyn_cost_jvp_n(%s, const V* dx, const V* du, T* f, T& q, V* df, V* dq) {' % sig_xu)
    S.append(_body_multi(_loads(spec) + tang_n, [('f[%d]' % i, f[i]) for i in range(n)] + [('q', c)] +
                         [('df[v * %d + %d]' % (n, i), df[i]) for i in range(n)] + [('dq[v]', dq)], vec=[s_ for s_, _ in tang_n]))
    S.append('  }')
    vin_n = [(s_, 'v1[%d]' % i) for i, s_ in enumerate(V1)] + [(s_, 'v2[v * %d + %d]' % (n, i)) for i, s_ in enumerate(V2)] + [(W1, 'w1')]
    S.append('  template<int NV, class T, class V> static LFSD_DEV void dyn_vjp2_n(%s, const T* v1, T w1, const V* v2, T* y1x, V* y2x, V* y2u) {' % sig_xu)
    S.append(_body_multi(_loads(spec) + vin_n, [('y1x[%d]' % i, y1x[i]) for i in range(n)] +
                         [('y2x[v * %d + %d]' % (n, i), y2x[i]) for i in range(n)] + [('y2u[v * %d + %d]' % (m, i), y2u[i]) for i in range(m)],
                         vec=list(V2)))
    S.append('  }')
    # 3. final cost
    sig_x = 'T t, const T* x, const T* e, const T* c'
    S.append('  template<class T> static LFSD_DEV T final_cost(%s) {\n    T hval;' % sig_x)
    S.append(_body(_loads(spec, with_u=False), [('hval', h)]))
    S.append('    return hval;\n  }')
    S.append('  template<class T> static LFSD_DEV void final_grad(%s, T* hx) {' % sig_x)
    S.append(_body(_loads(spec, with_u=False), [('hx[%d]' % i, hx[i]) for i in range(n)]))
    S.append('  }')
    yfin = hxx * dX + hxe * oE
    S.append('  // y = hxx*dx + hxe*oe')
    S.append('  template<class T> static LFSD_DEV void final_hess_mul(%s, const T* dx, const T* oe, T* y) {' % sig_x)
    S.append(_body(_loads(spec, with_u=False) + tang_e, [('y[%d]' % i, yfin[i]) for i in range(n)]))
    S.append('  }')
    # 3b. interface function of the waypoint loss (optional): y = g(x), and rx = (dg/dx)^T r
    nif = 0 if spec.g is None else spec.g.shape[0]
    S.append('  static constexpr int NIF = %d;      // outputs of the compiled interface function (0: none, the loss selects components)' % nif)
    S.append('  template<class T> static LFSD_DEV void iface(const T* x, const T* c, T* y) {')
    if nif:
        S.append(_body(_loads(spec, with_u=False), [('y[%d]' % i, spec.g[i]) for i in range(nif)]))
    else:
        S.append('    (void)x; (void)c; (void)y;')
    S.append('  }')
    S.append('  template<class T> static LFSD_DEV void iface_vjp(const T* x, const T* c, const T* r, T* rx) {')
    if nif:
        R = sp.Matrix([sp.Symbol('ri%d' % i, real=True) for i in range(nif)])
        gx = spec.g.jacobian(X)
        rx = gx.T * R
        S.append(_body(_loads(spec, with_u=False) + [(s_, 'r[%d]' % i) for i, s_ in enumerate(R)],
                       [('rx[%d]' % i, rx[i]) for i in range(n)]))
    else:
        S.append('    (void)x; (void)c; (void)r; (void)rx;')
    S.append('  }')
    # 4. Hamiltonian Hessian applied to a vector
    sig_xul = 'T t, const T* x, const T* u, const T* l, const T* e, const T* c'
    yx = Hxx * dX + Hxu * dU
    yu = Hxu.T * dX + Huu * dU
    S.append('  // yx = Hxx*dx + Hxu*du ; yu = Hux*dx + Huu*du   (H = c + l.f, CPDP.py:218)')
    S.append('  template<class T, class V> static LFSD_DEV void ham_hess_mul(%s, const V* dx, const V* du, V* yx, V* yu) {' % sig_xul)
    S.append(_body(_loads(spec, with_l=True) + tang, [('yx[%d]' % i, yx[i]) for i in range(n)] +
                   [('yu[%d]' % i, yu[i]) for i in range(m)],
                   vec=[s_ for s_, _ in tang], vec_out=lambda lv: True))
    S.append('  }')
    S.append('  template<int NV, class T, class V> static LFSD_DEV void ham_hess_mul_n(%s, const V* dx, const V* du, V* yx, V* yu) {' % sig_xul)
    S.append(_body_multi(_loads(spec, with_l=True) + tang_n, [('yx[v * %d + %d]' % (n, i), yx[i]) for i in range(n)] +
                         [('yu[v * %d + %d]' % (m, i), yu[i]) for i in range(m)], vec=[s_ for s_, _ in tang_n]))
    S.append('  }')
    # 4b. the control block of the Hamiltonian Hessian alone (dense, row-major): the MFMA backward sweep of the 16-lane
    #     mapping gets the columns no lane owns from the symmetry of H and needs only their diagonal block on top
    S.append('  template<class T> static LFSD_DEV void ham_huu(%s, T* Huu_out) {' % sig_xul)
    S.append(_body(_loads(spec, with_l=True), [('Huu_out[%d]' % (a * m + b), Huu[a, b]) for a in range(m) for b in range(m)]))
    S.append('  }')
    # 5. packed PMP coefficients, one packing per layout (0: Riccati sweep, every matrix; 1: forward sweep, without Hxx / Hxe
    #    and ordered so that what its inner loops read is a PREFIX of the node: fu (stiff step), then fx, Hxu, Huu^-1 (right-hand
    #    side) -- the kernel fetches that prefix into registers with back-to-back LDS reads, FU_N / RHS_N words)
    named = dict(fx=fx, fu=fu, fe=fe, Hxx=Hxx, Hxu=Hxu, Hxe=Hxe, Hue=Hue)
    seqs = [['fx', 'fu', 'fe', 'Hxx', 'Hxu', 'Hxe', 'Hue', 'HUU', 'IHUU'], ['fu', 'fx', 'Hxu', 'IHUU', 'fe', 'Hue', 'HUU']]
    place = [dict(), dict()]                       # name -> (offset, end) per layout
    off_huu, off_ihuu, off_zero, ncoef, fu_n, rhs_n = [0, 0], [0, 0], [0, 0], [0, 0], [0, 0], [0, 0]
    for lay, seq in enumerate(seqs):
        off = 0
        for nm in seq:
            off = (off + 3) // 4 * 4
            if nm == 'HUU':
                off_huu[lay] = off; off += m * m
            elif nm == 'IHUU':
                off_ihuu[lay] = off; off += m * m
            else:
                cnt = sum(1 for e in named[nm] if _nz(e))
                place[lay][nm] = off
                off += cnt
            if nm == 'fu':
                fu_n[lay] = (off + 3) // 4 * 4
            if nm == 'IHUU' and lay == 1:
                rhs_n[lay] = (off + 3) // 4 * 4
        off_zero[lay] = off                        # one word that always holds 0 (gather target of structural zeros)
        ncoef[lay] = ((off + 1 + 3) // 4) * 4
    rhs_n[0] = ncoef[0]
    mats = []
    for nm in ('fx', 'fu', 'fe', 'Hxx', 'Hxu', 'Hxe', 'Hue'):
        mats.append(_Sparse(nm, named[nm], [place[0].get(nm), place[1].get(nm)], _ORDERS[nm]))
    # A diagonal Huu (control cost separable in the controls, dynamics affine in them: every model of the zoo) is inverted
    # in closed form by the staging lane's own pmp_coeffs; a general Huu is inverted by the kernel (mat_inverse)
    ihuu_closed = all(not _nz(Huu[a, b]) for a in range(m) for b in range(m) if a != b)
    pair = lambda v: '{%d, %d}' % (v[0], v[1])
    S.append('  static constexpr int OFF_HUU_L[2] = %s, OFF_IHUU_L[2] = %s, OFF_ZERO_L[2] = %s, NCOEF_L[2] = %s;'
             % (pair(off_huu), pair(off_ihuu), pair(off_zero), pair(ncoef)))
    S.append('  static constexpr int OFF_HUU = %d, OFF_IHUU = %d, OFF_ZERO = %d, NCOEF = %d;      // layout 0'
             % (off_huu[0], off_ihuu[0], off_zero[0], ncoef[0]))
    S.append('  // leading words of a node that hold fu | fu, fx, Hxu, Huu^-1 (layout 1; layout 0: the whole node)')
    S.append('  static constexpr int FU_N_L[2] = %s, RHS_N_L[2] = %s;' % (pair(fu_n), pair(rhs_n)))
    S.append('  static constexpr bool IHUU_CLOSED = %s;      // Huu diagonal: pmp_coeffs stores Huu^-1 as well' % ('true' if ihuu_closed else 'false'))
    for lay in (0, 1):
        S.append('  // layout %d, packed (16-byte aligned starts): ' % lay +
                 ', '.join('%s[%d..%d)' % (sm.name, sm.off[lay], sm.end[lay]) for sm in mats if sm.lay[lay] is not None) +
                 ', Huu dense @%d, Huu^-1 dense @%d' % (off_huu[lay], off_ihuu[lay]))
    S.append('  template<int LAY, class T> static LFSD_DEV void pmp_coeffs(%s, T* L) {' % sig_xul)
    for lay in (0, 1):
        outs = []
        for sm in mats:
            outs += sm.stores(lay)
        outs += [('L[%d]' % (off_huu[lay] + a * m + b), Huu[a, b]) for a in range(m) for b in range(m)]
        if ihuu_closed:
            outs += [('L[%d]' % (off_ihuu[lay] + a * m + b), (1 / Huu[a, a]) if a == b else sp.Integer(0))
                     for a in range(m) for b in range(m)]
        outs += [('L[%d]' % off_zero[lay], sp.Integer(0))]
        S.append('    %s (LAY == %d) {' % ('if constexpr' if lay == 0 else '} else', lay) if lay == 0 else '    } else {')
        S.append(_body(_loads(spec, with_l=True), outs, indent='      '))
    S.append('    }')
    S.append('  }')
    # y = Huu^-1 v from the staged inverse: m products when Huu is diagonal, a dense m x m product otherwise
    S.append('  template<int LAY, class T> static LFSD_DEV void ihuu_mul(const T* L, const T* v, T* y) {')
    S.append('    const T* iH = L + OFF_IHUU_L[LAY];')
    if ihuu_closed:
        S.append('\n'.join('    y[%d] = iH[%d]*v[%d];' % (a, a * m + a, a) for a in range(m)))
    else:
        S.append('\n'.join('    y[%d] = %s;' % (a, ' + '.join('iH[%d]*v[%d]' % (a * m + b, b) for b in range(m))) for a in range(m)))
    S.append('  }')
    for sm in mats:
        S.append(sm.emit_ops())
    # packed offset of fx[r][c] per layout, OFF_ZERO for a structural zero: lets a lane gather ITS column of fx as a dense
    # vector -- the operand layout of the matrix-core experiment in the Riccati sweep (cpdp_aux.h, LFSD_RIC_MFMA)
    for lay in (0, 1):
        tab = [off_zero[lay]] * (n * n)
        for (r, cc, o, _) in mats[0].lay[lay]:
            tab[r * n + cc] = o
        S.append('  static LFSD_DEV int fx_off%d(int r, int c) { constexpr short tab[%d] = {%s}; return tab[r * %d + c]; }'
                 % (lay, n * n, ', '.join(map(str, tab)), n))
    # Column gathers.  The auxiliary sweeps carry one column per lane and add "their" column of a staged matrix to a
    # vector: y += Hxx e_j, hu = Hux e_j ...  As products with a one-hot vector those cost one FMA per non-zero of the whole
    # matrix; as gathers they cost one LDS read per ROW through a per-lane offset (OFF_ZERO where the entry is structurally
    # zero).  hcol_off(c, i): packed offset of [Hxx Hxe][i][c]; ucol_off(c, a): of [Hux Hue][a][c]; ecol_off(c, i): of fe[i][c]
    # (c over the NX + NP columns of Z = [P W], resp. the NP parameters).
    by = {sm.name: sm for sm in mats}

    def _tab(lay, rows, cols, entry):
        t = []
        for c in range(cols):
            for i in range(rows):
                t.append(entry(lay, i, c))
        return t

    def _or(v, dflt):
        return dflt if v is None else v

    def _find(sm, lay, r, c):
        if sm.lay[lay] is None:
            return None
        for (rr, cc, o, _) in sm.lay[lay]:
            if rr == r and cc == c:
                return o
        return None
    for lay in (0, 1):
        z = off_zero[lay]
        if by['Hxx'].lay[lay] is not None:
            t = _tab(lay, n, n + p, lambda l, i, c: _or(_find(by['Hxx'], l, i, c) if c < n else _find(by['Hxe'], l, i, c - n), z))
            S.append('  static LFSD_DEV int hcol_off%d(int c, int i) { constexpr short tab[%d] = {%s}; return tab[c * %d + i]; }'
                     % (lay, len(t), ', '.join(map(str, t)), n))
        t = _tab(lay, m, n + p, lambda l, a, c: _or(_find(by['Hxu'], l, c, a) if c < n else _find(by['Hue'], l, a, c - n), z))
        S.append('  static LFSD_DEV int ucol_off%d(int c, int a) { constexpr short tab[%d] = {%s}; return tab[c * %d + a]; }'
                 % (lay, len(t), ', '.join(map(str, t)), m))
        t = _tab(lay, n, p, lambda l, i, c: _or(_find(by['fe'], l, i, c), z))
        S.append('  static LFSD_DEV int ecol_off%d(int c, int i) { constexpr short tab[%d] = {%s}; return tab[c * %d + i]; }'
                 % (lay, len(t), ', '.join(map(str, t)), n))
    # G[a*NU+b] (+)= sum_i S[i*NU+a] * fu[i][b]    (S = rows of B^T P gathered in LDS)
    fu_sm = mats[1]
    S.append('  // G = S^T fu  with S an NX x NU row-major matrix (e.g. S = P fu  ->  G = fu^T P fu)')
    S.append('  template<bool ACC, int LAY, class T> static LFSD_DEV void fu_gram(const T* L, const T* S, T* G) {')
    for lay in (0, 1):
        cols = {}
        for (r, cc, o, _) in fu_sm.lay[lay]:
            cols.setdefault(cc, []).append((r, o))
        lines = []
        for a in range(m):
            for b in range(m):
                terms = ['S[%d]*L[%d]' % (r * m + a, o) for (r, o) in sorted(cols.get(b, []))]
                lines.append('      G[%d] = (ACC ? G[%d] : T(0))%s;' % (a * m + b, a * m + b,
                                                                      (' + ' + ' + '.join(terms)) if terms else ''))
        S.append('    if constexpr (LAY == %d) {' % lay if lay == 0 else '    } else {')
        S.append('\n'.join(lines))
    S.append('    }')
    S.append('  }')
    S.append('};')
    S.append('}  // namespace LFSD_MODEL_NS')
    return '\n'.join(S) + '\n'
