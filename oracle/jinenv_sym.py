"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product path).

Symbolic (sympy) restatement of the reference's robot environments
(/root/reference/JinEnv/JinEnv.py).  The reference builds CasADi ``SX``
expressions; CasADi 3.5.5 is not installed in this image, so the same
expressions are restated here with sympy and differentiated symbolically the
way ``CPDP.COCSys.diffPMP`` does (CPDP/CPDP.py:201-248).

Each class follows the reference class of the same name:

* ``SinglePendulum``  JinEnv.py:40-107
* ``RobotArm``        JinEnv.py:178-326
* ``CartPole``        JinEnv.py:499-574
* ``Quadrotor``       JinEnv.py:662-953 (+ dir_cosine/skew/omega 1182-1205)
* ``Rocket``          JinEnv.py:1248-1576

Convention kept from the reference: an argument left ``None`` becomes a
learnable symbol (collected in ``dyn_auxvar`` / ``cost_auxvar``); a numeric
argument is baked into the expression.  Animation/plot helpers are UI and are
out of scope.
"""
import math

import sympy as sp


def _sym_or(value, name, bag):
    """JinEnv idiom: ``None`` -> fresh learnable symbol appended to ``bag``."""
    if value is None:
        s = sp.Symbol(name, real=True)
        bag.append(s)
        return s
    return sp.Float(value) if isinstance(value, float) else sp.Integer(value) if isinstance(value, int) else value


def dir_cosine(q):
    """JinEnv.py:1182-1188 / 1553-1559 (inertial -> body DCM from quaternion)."""
    q0, q1, q2, q3 = q
    return sp.Matrix([
        [1 - 2 * (q2 ** 2 + q3 ** 2), 2 * (q1 * q2 + q0 * q3), 2 * (q1 * q3 - q0 * q2)],
        [2 * (q1 * q2 - q0 * q3), 1 - 2 * (q1 ** 2 + q3 ** 2), 2 * (q2 * q3 + q0 * q1)],
        [2 * (q1 * q3 + q0 * q2), 2 * (q2 * q3 - q0 * q1), 1 - 2 * (q1 ** 2 + q2 ** 2)],
    ])


def skew(v):
    """JinEnv.py:1190-1196."""
    return sp.Matrix([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def omega(w):
    """JinEnv.py:1198-1205."""
    return sp.Matrix([
        [0, -w[0], -w[1], -w[2]],
        [w[0], 0, w[2], -w[1]],
        [w[1], -w[2], 0, w[0]],
        [w[2], w[1], -w[0], 0],
    ])


def toQuaternion(angle, direction):
    """JinEnv.py:1730-1737."""
    n = math.sqrt(sum(d * d for d in direction))
    return [math.cos(angle / 2)] + [math.sin(angle / 2) * d / n for d in direction]


class SinglePendulum:
    """JinEnv.py:40-107."""

    def initDyn(self, l=None, m=None, damping_ratio=None):
        g = 10
        par = []
        self.l = _sym_or(l, 'l', par)
        self.m = _sym_or(m, 'm', par)
        self.damping_ratio = _sym_or(damping_ratio, 'damping_ratio', par)
        self.dyn_auxvar = par
        self.q, self.dq = sp.symbols('q dq', real=True)
        self.X = sp.Matrix([self.q, self.dq])
        u = sp.Symbol('u', real=True)
        self.U = sp.Matrix([u])
        I = sp.Rational(1, 3) * self.m * self.l * self.l
        self.f = sp.Matrix([self.dq,
                            (u - self.m * g * self.l * sp.sin(self.q) - self.damping_ratio * self.dq) / I])

    def initCost(self, wq=None, wdq=None, wu=0.001):
        par = []
        self.wq = _sym_or(wq, 'wq', par)
        self.wdq = _sym_or(wdq, 'wdq', par)
        self.cost_auxvar = par
        cost_q = (self.q - math.pi) ** 2
        cost_dq = (self.dq - 0) ** 2
        cost_u = self.U[0] * self.U[0]
        self.path_cost = self.wq * cost_q + self.wdq * cost_dq + wu * cost_u
        self.final_cost = self.wq * cost_q + self.wdq * cost_dq


class RobotArm:
    """JinEnv.py:178-326."""

    def initDyn(self, l1=None, m1=None, l2=None, m2=None, g=10):
        par = []
        self.l1 = _sym_or(l1, 'l1', par)
        self.m1 = _sym_or(m1, 'm1', par)
        self.l2 = _sym_or(l2, 'l2', par)
        self.m2 = _sym_or(m2, 'm2', par)
        self.dyn_auxvar = par
        self.q1, self.dq1, self.q2, self.dq2 = sp.symbols('q1 dq1 q2 dq2', real=True)
        self.X = sp.Matrix([self.q1, self.q2, self.dq1, self.dq2])
        u1, u2 = sp.symbols('u1 u2', real=True)
        self.U = sp.Matrix([u1, u2])
        r1 = self.l1 / 2
        r2 = self.l2 / 2
        I1 = self.l1 * self.l1 * self.m1 / 12
        I2 = self.l2 * self.l2 * self.m2 / 12
        M11 = self.m1 * r1 * r1 + I1 + self.m2 * (self.l1 * self.l1 + r2 * r2 + 2 * self.l1 * r2 * sp.cos(self.q2)) + I2
        M12 = self.m2 * (r2 * r2 + self.l1 * r2 * sp.cos(self.q2)) + I2
        M22 = self.m2 * r2 * r2 + I2
        M = sp.Matrix([[M11, M12], [M12, M22]])
        h = self.m2 * self.l1 * r2 * sp.sin(self.q2)
        C = sp.Matrix([-h * self.dq2 * self.dq2 - 2 * h * self.dq1 * self.dq2, h * self.dq1 * self.dq1])
        G = sp.Matrix([
            self.m1 * r1 * g * sp.cos(self.q1) + self.m2 * g * (r2 * sp.cos(self.q1 + self.q2) + self.l1 * sp.cos(self.q1)),
            self.m2 * g * r2 * sp.cos(self.q1 + self.q2)])
        # reference: mtimes(pinv(M), -C - G + U); M is a nonsingular 2x2 so pinv == inverse
        det = M11 * M22 - M12 * M12
        Minv = sp.Matrix([[M22, -M12], [-M12, M11]]) / det
        ddq = Minv * (-C - G + self.U)
        self.f = sp.Matrix([self.dq1, self.dq2, ddq[0], ddq[1]])

    def initCost_WeightedDistance(self, wq1=None, wq2=None, wdq1=None, wdq2=None, wu=0.1):
        par = []
        self.wq1 = _sym_or(wq1, 'wq1', par)
        self.wq2 = _sym_or(wq2, 'wq2', par)
        self.wdq1 = _sym_or(wdq1, 'wdq1', par)
        self.wdq2 = _sym_or(wdq2, 'wdq2', par)
        self.cost_auxvar = par
        goal = [math.pi / 2, 0, 0, 0]
        cq1 = (self.q1 - goal[0]) ** 2
        cq2 = (self.q2 - goal[1]) ** 2
        cdq1 = (self.dq1 - goal[2]) ** 2
        cdq2 = (self.dq2 - goal[3]) ** 2
        cu = self.U.dot(self.U)
        self.path_cost = self.wq1 * cq1 + self.wq2 * cq2 + self.wdq1 * cdq1 + self.wdq2 * cdq2 + wu * cu
        self.final_cost = self.wq1 * cq1 + self.wq2 * cq2 + self.wdq1 * cdq1 + self.wdq2 * cdq2

    def initCost_Polynomial(self, wu=0.1):
        goal = [math.pi / 2, 0, 0, 0]
        cq1 = (self.q1 - goal[0]) ** 2
        cq2 = (self.q2 - goal[1]) ** 2
        cdq1 = (self.dq1 - goal[2]) ** 2
        cdq2 = (self.dq2 - goal[3]) ** 2
        cu = self.U.dot(self.U)
        w_q1_sq, w_q1, w_q2_sq, w_q2 = sp.symbols('w_q1_sq w_q1 w_q2_sq w_q2', real=True)
        self.path_cost = (w_q1 * self.q1 + w_q1_sq * sp.Rational(1, 2) * self.q1 * self.q1 +
                          w_q2 * self.q2 + w_q2_sq * sp.Rational(1, 2) * self.q2 * self.q2 + wu * cu)
        self.final_cost = 100 * cq1 + 100 * cq2 + 100 * cdq1 + 100 * cdq2
        self.cost_auxvar = [w_q1_sq, w_q1, w_q2_sq, w_q2]


class CartPole:
    """JinEnv.py:499-574."""

    def initDyn(self, mc=None, mp=None, l=None):
        g = 10
        par = []
        self.mc = _sym_or(mc, 'mc', par)
        self.mp = _sym_or(mp, 'mp', par)
        self.l = _sym_or(l, 'l', par)
        self.dyn_auxvar = par
        self.x, self.q, self.dx, self.dq = sp.symbols('x q dx dq', real=True)
        self.X = sp.Matrix([self.x, self.q, self.dx, self.dq])
        u = sp.Symbol('u', real=True)
        self.U = sp.Matrix([u])
        s, c = sp.sin(self.q), sp.cos(self.q)
        ddx = (u + self.mp * s * (self.l * self.dq * self.dq + g * c)) / (self.mc + self.mp * s * s)
        ddq = (-u * c - self.mp * self.l * self.dq * self.dq * s * c - (self.mc + self.mp) * g * s) / (
            self.l * self.mc + self.l * self.mp * s * s)
        self.f = sp.Matrix([self.dx, self.dq, ddx, ddq])

    def initCost(self, wx=None, wq=None, wdx=None, wdq=None, wu=0.001):
        par = []
        self.wx = _sym_or(wx, 'wx', par)
        self.wq = _sym_or(wq, 'wq', par)
        self.wdx = _sym_or(wdx, 'wdx', par)
        self.wdq = _sym_or(wdq, 'wdq', par)
        self.cost_auxvar = par
        goal = [0.0, math.pi, 0.0, 0.0]
        state_cost = (self.wx * (self.x - goal[0]) ** 2 + self.wq * (self.q - goal[1]) ** 2 +
                      self.wdx * (self.dx - goal[2]) ** 2 + self.wdq * (self.dq - goal[3]) ** 2)
        self.path_cost = state_cost + wu * (self.U[0] * self.U[0])
        self.final_cost = state_cost


class _RigidBody6Dof:
    def _declare_states(self, n_u, u_names):
        self.r_I = sp.Matrix(sp.symbols('rx ry rz', real=True))
        self.v_I = sp.Matrix(sp.symbols('vx vy vz', real=True))
        self.q = sp.Matrix(sp.symbols('q0 q1 q2 q3', real=True))
        self.w_B = sp.Matrix(sp.symbols('wx wy wz', real=True))
        self.T_B = sp.Matrix(sp.symbols(u_names, real=True))


class Quadrotor(_RigidBody6Dof):
    """JinEnv.py:662-953."""

    def __init__(self):
        self._declare_states(4, 'f1 f2 f3 f4')

    def initDyn(self, Jx=None, Jy=None, Jz=None, mass=None, l=None, c=None):
        g = 9.81
        par = []
        self.Jx = _sym_or(Jx, 'Jx', par)
        self.Jy = _sym_or(Jy, 'Jy', par)
        self.Jz = _sym_or(Jz, 'Jz', par)
        self.mass = _sym_or(mass, 'mass', par)
        self.l = _sym_or(l, 'l', par)
        self.c = _sym_or(c, 'c', par)
        self.dyn_auxvar = par
        J = sp.diag(self.Jx, self.Jy, self.Jz)
        Jinv = sp.diag(1 / self.Jx, 1 / self.Jy, 1 / self.Jz)   # pinv of a diagonal J
        g_I = sp.Matrix([0, 0, -g])
        T = self.T_B
        thrust_B = sp.Matrix([0, 0, T[0] + T[1] + T[2] + T[3]])
        M_B = sp.Matrix([-T[1] * self.l / 2 + T[3] * self.l / 2,
                         -T[0] * self.l / 2 + T[2] * self.l / 2,
                         (T[0] - T[1] + T[2] - T[3]) * self.c])
        C_I_B = dir_cosine(self.q).T
        dr = self.v_I
        dv = (1 / self.mass) * (C_I_B * thrust_B) + g_I
        dq = sp.Rational(1, 2) * (omega(self.w_B) * self.q)
        dw = Jinv * (M_B - skew(self.w_B) * J * self.w_B)
        self.X = sp.Matrix.vstack(self.r_I, self.v_I, self.q, self.w_B)
        self.U = self.T_B
        self.f = sp.Matrix.vstack(dr, dv, dq, dw)

    def _goal_terms(self, goal_r, goal_v, goal_q, goal_w):
        goal_r, goal_v, goal_w = sp.Matrix(goal_r), sp.Matrix(goal_v), sp.Matrix(goal_w)
        cr = (self.r_I - goal_r).dot(self.r_I - goal_r)
        cv = (self.v_I - goal_v).dot(self.v_I - goal_v)
        cw = (self.w_B - goal_w).dot(self.w_B - goal_w)
        cq = (sp.eye(3) - dir_cosine(list(goal_q)).T * dir_cosine(self.q)).trace()
        return cr, cv, cq, cw

    def initCost(self, goal_r, goal_v, goal_q, goal_w, wr=None, wv=None, wq=None, ww=None, wthrust=0.1):
        par = []
        self.wr = _sym_or(wr, 'wr', par)
        self.wv = _sym_or(wv, 'wv', par)
        self.wq = _sym_or(wq, 'wq', par)
        self.ww = _sym_or(ww, 'ww', par)
        self.cost_auxvar = par
        cr, cv, cq, cw = self._goal_terms(goal_r, goal_v, goal_q, goal_w)
        ct = self.T_B.dot(self.T_B)
        self.final_cost = self.wr * cr + self.wv * cv + self.ww * cw + self.wq * cq
        self.path_cost = self.final_cost + wthrust * ct

    def initCost2(self, goal_r, goal_v, goal_q, goal_w, wthrust=0.1):
        names = 'wrx wry wrz wvx wvy wvz wwx wwy wwz wq'
        w = sp.symbols(names, real=True)
        self.cost_auxvar = list(w)
        _, _, cq, _ = self._goal_terms(goal_r, goal_v, goal_q, goal_w)
        state_cost = sum(w[i] * (self.r_I[i] - goal_r[i]) ** 2 for i in range(3))
        state_cost += sum(w[3 + i] * (self.v_I[i] - goal_v[i]) ** 2 for i in range(3))
        state_cost += sum(w[6 + i] * (self.w_B[i] - goal_w[i]) ** 2 for i in range(3))
        state_cost += w[9] * cq
        self.path_cost = state_cost + wthrust * self.T_B.dot(self.T_B)
        self.final_cost = state_cost

    def initCost_Polynomial(self, goal_r, goal_v, goal_q, goal_w, w_thrust=0.1):
        cr, cv, cq, cw = self._goal_terms(goal_r, goal_v, goal_q, goal_w)
        ct = self.T_B.dot(self.T_B)
        w_xsq, w_x, w_ysq, w_y, w_zsq, w_z = sp.symbols('w_xsq w_x w_ysq w_y w_zsq w_z', real=True)
        half = sp.Rational(1, 2)
        r = self.r_I
        self.path_cost = (w_xsq * half * r[0] * r[0] + w_x * r[0] + w_ysq * half * r[1] * r[1] + w_y * r[1] +
                          w_zsq * half * r[2] * r[2] + w_z * r[2] + w_thrust * ct)
        self.final_cost = 1 * cr + 11 * cv + 100 * cq + 10 * cw
        self.cost_auxvar = [w_xsq, w_x, w_ysq, w_y, w_zsq, w_z]


class Rocket(_RigidBody6Dof):
    """JinEnv.py:1248-1551."""

    def __init__(self):
        self._declare_states(3, 'ux uy uz')

    def initDyn(self, Jx=None, Jy=None, Jz=None, mass=None, l=None):
        g = 10
        par = []
        self.Jx = _sym_or(Jx, 'Jx', par)
        self.Jy = _sym_or(Jy, 'Jy', par)
        self.Jz = _sym_or(Jz, 'Jz', par)
        self.mass = _sym_or(mass, 'mass', par)
        self.l = _sym_or(l, 'l', par)
        self.dyn_auxvar = par
        J = sp.diag(self.Jx, self.Jy, self.Jz)
        Jinv = sp.diag(1 / self.Jx, 1 / self.Jy, 1 / self.Jz)
        g_I = sp.Matrix([-g, 0, 0])
        r_T_B = sp.Matrix([-self.l / 2, 0, 0])
        C_I_B = dir_cosine(self.q).T
        dr = self.v_I
        dv = (1 / self.mass) * (C_I_B * self.T_B) + g_I
        dq = sp.Rational(1, 2) * (omega(self.w_B) * self.q)
        dw = Jinv * (skew(r_T_B) * self.T_B - skew(self.w_B) * J * self.w_B)
        self.X = sp.Matrix.vstack(self.r_I, self.v_I, self.q, self.w_B)
        self.U = self.T_B
        self.f = sp.Matrix.vstack(dr, dv, dq, dw)

    def _terms(self):
        C_I_B = dir_cosine(self.q).T
        nx = sp.Matrix([1, 0, 0])
        b = C_I_B * nx
        tilt = b[1] ** 2 + b[2] ** 2
        side = self.T_B[1] ** 2 + self.T_B[2] ** 2
        thrust = self.T_B.dot(self.T_B)
        return tilt, side, thrust

    def initCost(self, wr=None, wv=None, wtilt=None, ww=None, wsidethrust=None, wthrust=1.0):
        par = []
        self.wr = _sym_or(wr, 'wr', par)
        self.wv = _sym_or(wv, 'wv', par)
        self.wtilt = _sym_or(wtilt, 'wtilt', par)
        self.wsidethrust = _sym_or(wsidethrust, 'wsidethrust', par)
        self.ww = _sym_or(ww, 'ww', par)
        self.cost_auxvar = par
        tilt, side, thrust = self._terms()
        cr, cv, cw = self.r_I.dot(self.r_I), self.v_I.dot(self.v_I), self.w_B.dot(self.w_B)
        self.final_cost = self.wr * cr + self.wv * cv + self.ww * cw + self.wtilt * tilt
        self.path_cost = self.final_cost + self.wsidethrust * side + wthrust * thrust

    def _per_axis(self, w):
        s = sum(w[i] * self.r_I[i] ** 2 for i in range(3))
        s += sum(w[3 + i] * self.v_I[i] ** 2 for i in range(3))
        s += sum(w[6 + i] * self.w_B[i] ** 2 for i in range(3))
        return s

    def initCost2(self, wthrust=0.1):
        w = sp.symbols('wrx wry wrz wvx wvy wvz wwx wwy wwz wsidethrust wtilt', real=True)
        self.cost_auxvar = list(w)
        tilt, side, thrust = self._terms()
        self.final_cost = self._per_axis(w) + w[10] * tilt
        self.path_cost = self.final_cost + w[9] * side + wthrust * thrust

    def initCost_Ex(self, wthrust=0.1):
        w = sp.symbols('wrx wry wrz wvx wvy wvz wwx wwy wwz wtilt wsidethrust', real=True)
        self.cost_auxvar = list(w)
        tilt, side, thrust = self._terms()
        self.final_cost = self._per_axis(w) + w[9] * tilt + w[10] * side
        self.path_cost = self.final_cost + wthrust * thrust
