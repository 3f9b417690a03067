"""Small CasADi-flavoured symbolic layer (sympy underneath).

The reference scripts build their models with ``casadi.SX`` (e.g.
Examples/quad_example.py via lib/QuadAlgorithm.py:80-98:
``beta = SX.sym('beta'); dyn = beta * env.f; vertcat(beta, env.cost_auxvar)``).
CasADi is only the *expression front-end* of the hot path; here the same
expressions are sympy objects that ``codegen.py`` turns into HIP device code.

Extension over the reference: ``const(name, value)`` makes a *runtime constant*
(physical parameter, goal state ...).  The reference bakes such numbers into
the CasADi graph; here they are kernel inputs so that one compiled model
serves every goal / every demonstration of a batch.
"""
import itertools

import sympy as sp

_uid = itertools.count()
_CONST_DEFAULT = {}


class SX:
    """``SX.sym(name[, n])`` as in CasADi: a scalar symbol, or an n-vector of them."""

    @staticmethod
    def sym(name, n=1, m=None):
        if m is not None:
            return sp.Matrix(n, m, lambda i, j: sp.Symbol('%s_%d_%d' % (name, i, j), real=True))
        if n == 1:
            return sp.Symbol(name, real=True)
        return sp.Matrix([sp.Symbol('%s_%d' % (name, i), real=True) for i in range(n)])


def const(name, value):
    """Runtime constant with a default value (uniquely named so two envs never clash)."""
    s = sp.Symbol('%s__k%d' % (name, next(_uid)), real=True)
    _CONST_DEFAULT[s] = float(value)
    return s


def is_const(s):
    return s in _CONST_DEFAULT


def const_default(s):
    return _CONST_DEFAULT[s]


def _flat(args):
    out = []
    for a in args:
        if isinstance(a, sp.MatrixBase):
            out.extend(list(a))
        elif isinstance(a, (list, tuple)):
            out.extend(_flat(a))
        else:
            out.append(sp.sympify(a))
    return out


def vertcat(*args):
    return sp.Matrix(_flat(args))


def vcat(args):
    return sp.Matrix(_flat(args))


def horzcat(*args):
    return sp.Matrix([_flat(args)])


def mtimes(a, b):
    return sp.Matrix(a) * sp.Matrix(b)


def transpose(a):
    return sp.Matrix(a).T


def dot(a, b):
    a, b = _flat([a]), _flat([b])
    return sum(x * y for x, y in zip(a, b))


def trace(a):
    return sp.Matrix(a).trace()


def diag(v):
    return sp.diag(*_flat([v]))


def pinv(a):
    """CasADi ``pinv`` of the square nonsingular matrices the robots use == inverse."""
    a = sp.Matrix(a)
    if a.shape[0] != a.shape[1]:
        raise NotImplementedError("pinv of non-square symbolic matrices")
    if a.is_diagonal():
        return sp.diag(*[1 / a[i, i] for i in range(a.shape[0])])
    return a.adjugate() / a.det()


def jacobian(expr, wrt):
    e = sp.Matrix(_flat([expr]))
    return e.jacobian(sp.Matrix(_flat([wrt])))


sin, cos, tan, exp, log, sqrt = sp.sin, sp.cos, sp.tan, sp.exp, sp.log, sp.sqrt


def fmax(a, b):
    """Numeric fmax (the examples only use it for the projection step on numbers)."""
    import numpy as np
    return np.maximum(a, b)
