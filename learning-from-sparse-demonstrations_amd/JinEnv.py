"""Robot environments of the reference's ``JinEnv`` package, as symbolic models.

Mirrors the *model-definition* half of /root/reference/JinEnv/JinEnv.py (the
``initDyn`` / ``initCost*`` methods and the attributes ``X, U, f, path_cost,
final_cost, dyn_auxvar, cost_auxvar`` the examples consume).  Same call
signatures and the same convention: an argument left ``None`` becomes a
learnable symbol; a number is a fixed parameter.  Fixed parameters are emitted
as *runtime constants* (``symbolic.const``) rather than baked literals, so one
compiled HIP model serves every goal state / mass / arm length of a batch.

Animation / plotting (``play_animation`` ...) is UI and is out of scope here.
"""
import math
from dataclasses import dataclass, field

import numpy as np
import sympy as sp

from .symbolic import SX, const


@dataclass
class QuadStates:
    """lib/QuadStates.py:5-14."""
    position: list = field(default_factory=lambda: [0, 0, 0])
    velocity: list = field(default_factory=lambda: [0, 0, 0])
    attitude_quaternion: list = field(default_factory=lambda: [1, 0, 0, 0])
    angular_velocity: list = field(default_factory=lambda: [0, 0, 0])


def _param(value, name, learnable):
    if value is None:
        s = SX.sym(name)
        learnable.append(s)
        return s
    if isinstance(value, sp.Basic):
        return value
    return const(name, value)


def _vec(names):
    return sp.Matrix([SX.sym(nm) for nm in names])


def _dcm(q):
    """Direction cosine matrix, inertial -> body (JinEnv.py:1182-1188)."""
    a, b, c, d = q[0], q[1], q[2], q[3]
    return sp.Matrix(3, 3, [
        1 - 2 * (c * c + d * d), 2 * (b * c + a * d), 2 * (b * d - a * c),
        2 * (b * c - a * d), 1 - 2 * (b * b + d * d), 2 * (c * d + a * b),
        2 * (b * d + a * c), 2 * (c * d - a * b), 1 - 2 * (b * b + c * c)])


def _cross_mat(v):
    """JinEnv.py:1190-1196."""
    return sp.Matrix(3, 3, [0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0])


def _quat_rate_mat(w):
    """JinEnv.py:1198-1205."""
    return sp.Matrix(4, 4, [0, -w[0], -w[1], -w[2],
                            w[0], 0, w[2], -w[1],
                            w[1], -w[2], 0, w[0],
                            w[2], w[1], -w[0], 0])


def toQuaternion(angle, dir):
    """JinEnv.py:1730-1737."""
    d = np.asarray(dir, dtype=float)
    d = d / np.linalg.norm(d)
    return [math.cos(angle / 2)] + (math.sin(angle / 2) * d).tolist()


class SinglePendulum:
    """JinEnv.py:40-107.  state (q, dq), control u."""

    def __init__(self, project_name='single pendlumn system'):
        self.project_name = project_name

    def initDyn(self, l=None, m=None, damping_ratio=None):
        g = 10
        learn = []
        self.l = _param(l, 'l', learn)
        self.m = _param(m, 'm', learn)
        self.damping_ratio = _param(damping_ratio, 'damping_ratio', learn)
        self.dyn_auxvar = sp.Matrix(learn)
        self.q, self.dq = SX.sym('q'), SX.sym('dq')
        self.X = sp.Matrix([self.q, self.dq])
        self.U = sp.Matrix([SX.sym('u')])
        inertia = self.m * self.l ** 2 / 3
        torque = self.U[0] - self.m * g * self.l * sp.sin(self.q) - self.damping_ratio * self.dq
        self.f = sp.Matrix([self.dq, torque / inertia])

    def initCost(self, wq=None, wdq=None, wu=0.001):
        learn = []
        self.wq = _param(wq, 'wq', learn)
        self.wdq = _param(wdq, 'wdq', learn)
        wu = _param(wu, 'wu', learn)
        self.cost_auxvar = sp.Matrix(learn)
        err = self.wq * (self.q - math.pi) ** 2 + self.wdq * self.dq ** 2
        self.path_cost = err + wu * self.U[0] ** 2
        self.final_cost = err


class RobotArm:
    """JinEnv.py:178-326.  state (q1, q2, dq1, dq2), control (u1, u2)."""

    def __init__(self, project_name='two-link robot arm'):
        self.project_name = project_name

    def initDyn(self, l1=None, m1=None, l2=None, m2=None, g=10):
        learn = []
        self.l1 = _param(l1, 'l1', learn)
        self.m1 = _param(m1, 'm1', learn)
        self.l2 = _param(l2, 'l2', learn)
        self.m2 = _param(m2, 'm2', learn)
        grav = _param(g, 'g', learn)
        self.dyn_auxvar = sp.Matrix(learn)
        self.q1, self.dq1, self.q2, self.dq2 = SX.sym('q1'), SX.sym('dq1'), SX.sym('q2'), SX.sym('dq2')
        self.X = sp.Matrix([self.q1, self.q2, self.dq1, self.dq2])
        self.U = _vec(['u1', 'u2'])
        l1, l2, m1, m2 = self.l1, self.l2, self.m1, self.m2
        rc1, rc2 = l1 / 2, l2 / 2
        J1, J2 = l1 * l1 * m1 / 12, l2 * l2 * m2 / 12
        c2 = sp.cos(self.q2)
        a11 = m1 * rc1 * rc1 + J1 + m2 * (l1 * l1 + rc2 * rc2 + 2 * l1 * rc2 * c2) + J2
        a12 = m2 * (rc2 * rc2 + l1 * rc2 * c2) + J2
        a22 = m2 * rc2 * rc2 + J2
        hh = m2 * l1 * rc2 * sp.sin(self.q2)
        cor = sp.Matrix([-hh * self.dq2 ** 2 - 2 * hh * self.dq1 * self.dq2, hh * self.dq1 ** 2])
        c12 = sp.cos(self.q1 + self.q2)
        grv = sp.Matrix([m1 * rc1 * grav * sp.cos(self.q1) + m2 * grav * (rc2 * c12 + l1 * sp.cos(self.q1)),
                         m2 * grav * rc2 * c12])
        rhs = -cor - grv + self.U
        # pinv(M) * rhs with M = [[a11,a12],[a12,a22]] nonsingular (JinEnv.py:236)
        det = a11 * a22 - a12 * a12
        acc = sp.Matrix([(a22 * rhs[0] - a12 * rhs[1]) / det, (a11 * rhs[1] - a12 * rhs[0]) / det])
        self.f = sp.Matrix([self.dq1, self.dq2, acc[0], acc[1]])

    def _goal_errs(self):
        return ((self.q1 - math.pi / 2) ** 2, self.q2 ** 2, self.dq1 ** 2, self.dq2 ** 2)

    def initCost_WeightedDistance(self, wq1=None, wq2=None, wdq1=None, wdq2=None, wu=0.1):
        learn = []
        self.wq1 = _param(wq1, 'wq1', learn)
        self.wq2 = _param(wq2, 'wq2', learn)
        self.wdq1 = _param(wdq1, 'wdq1', learn)
        self.wdq2 = _param(wdq2, 'wdq2', learn)
        wu = _param(wu, 'wu', learn)
        self.cost_auxvar = sp.Matrix(learn)
        e1, e2, e3, e4 = self._goal_errs()
        self.final_cost = self.wq1 * e1 + self.wq2 * e2 + self.wdq1 * e3 + self.wdq2 * e4
        self.path_cost = self.final_cost + wu * self.U.dot(self.U)

    def initCost_Polynomial(self, wu=0.1):
        wu = _param(wu, 'wu', [])
        self.w_q1_sq, self.w_q1 = SX.sym('w_q1_sq'), SX.sym('w_q1')
        self.w_q2_sq, self.w_q2 = SX.sym('w_q2_sq'), SX.sym('w_q2')
        self.cost_auxvar = sp.Matrix([self.w_q1_sq, self.w_q1, self.w_q2_sq, self.w_q2])
        half = sp.Rational(1, 2)
        self.path_cost = (self.w_q1 * self.q1 + self.w_q1_sq * half * self.q1 ** 2 +
                          self.w_q2 * self.q2 + self.w_q2_sq * half * self.q2 ** 2 + wu * self.U.dot(self.U))
        self.final_cost = 100 * sum(self._goal_errs())


class CartPole:
    """JinEnv.py:499-574.  state (x, q, dx, dq), control u."""

    def __init__(self, project_name='cart-pole-system'):
        self.project_name = project_name

    def initDyn(self, mc=None, mp=None, l=None):
        g = 10
        learn = []
        self.mc = _param(mc, 'mc', learn)
        self.mp = _param(mp, 'mp', learn)
        self.l = _param(l, 'l', learn)
        self.dyn_auxvar = sp.Matrix(learn)
        self.x, self.q, self.dx, self.dq = SX.sym('x'), SX.sym('q'), SX.sym('dx'), SX.sym('dq')
        self.X = sp.Matrix([self.x, self.q, self.dx, self.dq])
        self.U = sp.Matrix([SX.sym('u')])
        u = self.U[0]
        sq, cq = sp.sin(self.q), sp.cos(self.q)
        den = self.mc + self.mp * sq * sq
        ddx = (u + self.mp * sq * (self.l * self.dq ** 2 + g * cq)) / den
        ddq = (-u * cq - self.mp * self.l * self.dq ** 2 * sq * cq - (self.mc + self.mp) * g * sq) / (self.l * den)
        self.f = sp.Matrix([self.dx, self.dq, ddx, ddq])

    def initCost(self, wx=None, wq=None, wdx=None, wdq=None, wu=0.001):
        learn = []
        self.wx = _param(wx, 'wx', learn)
        self.wq = _param(wq, 'wq', learn)
        self.wdx = _param(wdx, 'wdx', learn)
        self.wdq = _param(wdq, 'wdq', learn)
        wu = _param(wu, 'wu', learn)
        self.cost_auxvar = sp.Matrix(learn)
        err = (self.wx * self.x ** 2 + self.wq * (self.q - math.pi) ** 2 +
               self.wdx * self.dx ** 2 + self.wdq * self.dq ** 2)
        self.path_cost = err + wu * self.U[0] ** 2
        self.final_cost = err


class _SixDof:
    def _states(self, controls):
        self.r_I = _vec(['rx', 'ry', 'rz'])
        self.v_I = _vec(['vx', 'vy', 'vz'])
        self.q = _vec(['q0', 'q1', 'q2', 'q3'])
        self.w_B = _vec(['wx', 'wy', 'wz'])
        self.T_B = _vec(controls)

    def _inertia(self, Jx, Jy, Jz, mass, l, learn):
        self.Jx = _param(Jx, 'Jx', learn)
        self.Jy = _param(Jy, 'Jy', learn)
        self.Jz = _param(Jz, 'Jz', learn)
        self.mass = _param(mass, 'mass', learn)
        self.l = _param(l, 'l', learn)

    def _euler(self, moment):
        Jd = [self.Jx, self.Jy, self.Jz]
        Jw = sp.Matrix([Jd[i] * self.w_B[i] for i in range(3)])
        gyro = _cross_mat(self.w_B) * Jw
        return sp.Matrix([(moment[i] - gyro[i]) / Jd[i] for i in range(3)])

    def _assemble(self, force_B, moment_B, g_I):
        dr = self.v_I
        dv = (_dcm(self.q).T * force_B) / self.mass + g_I
        dq = _quat_rate_mat(self.w_B) * self.q / 2
        dw = self._euler(moment_B)
        self.X = sp.Matrix.vstack(self.r_I, self.v_I, self.q, self.w_B)
        self.U = self.T_B
        self.f = sp.Matrix.vstack(dr, dv, dq, dw)


class Quadrotor(_SixDof):
    """JinEnv.py:662-953.  13 states (r, v, quaternion, body rate), 4 rotor thrusts."""

    def __init__(self, project_name='my UAV'):
        self.project_name = 'my uav'
        self._states(['f1', 'f2', 'f3', 'f4'])

    def initDyn(self, Jx=None, Jy=None, Jz=None, mass=None, l=None, c=None):
        learn = []
        self._inertia(Jx, Jy, Jz, mass, l, learn)
        self.c = _param(c, 'c', learn)
        self.dyn_auxvar = sp.Matrix(learn)
        T = self.T_B
        thrust = sp.Matrix([0, 0, T[0] + T[1] + T[2] + T[3]])
        moment = sp.Matrix([(T[3] - T[1]) * self.l / 2, (T[2] - T[0]) * self.l / 2,
                            (T[0] - T[1] + T[2] - T[3]) * self.c])
        self._assemble(thrust, moment, sp.Matrix([0, 0, -sp.Float(9.81)]))

    def _goal(self, Q):
        gr = sp.Matrix([const('goal_r%d' % i, Q.position[i]) for i in range(3)])
        gv = sp.Matrix([const('goal_v%d' % i, Q.velocity[i]) for i in range(3)])
        gq = [const('goal_q%d' % i, Q.attitude_quaternion[i]) for i in range(4)]
        gw = sp.Matrix([const('goal_w%d' % i, Q.angular_velocity[i]) for i in range(3)])
        return gr, gv, gq, gw

    def _attitude_err(self, gq):
        return (sp.eye(3) - _dcm(gq).T * _dcm(self.q)).trace()

    def initCost(self, QuadDesiredStates, wr=None, wv=None, wq=None, ww=None, wthrust=0.1):
        gr, gv, gq, gw = self._goal(QuadDesiredStates)
        learn = []
        self.wr = _param(wr, 'wr', learn)
        self.wv = _param(wv, 'wv', learn)
        self.wq = _param(wq, 'wq', learn)
        self.ww = _param(ww, 'ww', learn)
        wthrust = _param(wthrust, 'wthrust', [])
        self.cost_auxvar = sp.Matrix(learn)
        dr, dv, dw = self.r_I - gr, self.v_I - gv, self.w_B - gw
        self.final_cost = (self.wr * dr.dot(dr) + self.wv * dv.dot(dv) + self.ww * dw.dot(dw) +
                           self.wq * self._attitude_err(gq))
        self.path_cost = self.final_cost + wthrust * self.T_B.dot(self.T_B)

    def initCost2(self, QuadDesiredStates, wthrust=0.1):
        gr, gv, gq, gw = self._goal(QuadDesiredStates)
        wthrust = _param(wthrust, 'wthrust', [])
        names = ['wrx', 'wry', 'wrz', 'wvx', 'wvy', 'wvz', 'wwx', 'wwy', 'wwz', 'wq']
        w = [SX.sym(nm) for nm in names]
        self.cost_auxvar = sp.Matrix(w)
        s = 0
        for i in range(3):
            s += w[i] * (self.r_I[i] - gr[i]) ** 2 + w[3 + i] * (self.v_I[i] - gv[i]) ** 2
            s += w[6 + i] * (self.w_B[i] - gw[i]) ** 2
        s += w[9] * self._attitude_err(gq)
        self.final_cost = s
        self.path_cost = s + wthrust * self.T_B.dot(self.T_B)

    def initCost_Polynomial(self, QuadDesiredStates, w_thrust=0.1):
        gr, gv, gq, gw = self._goal(QuadDesiredStates)
        w_thrust = _param(w_thrust, 'w_thrust', [])
        names = ['w_xsq', 'w_x', 'w_ysq', 'w_y', 'w_zsq', 'w_z']
        w = [SX.sym(nm) for nm in names]
        self.cost_auxvar = sp.Matrix(w)
        half = sp.Rational(1, 2)
        pc = w_thrust * self.T_B.dot(self.T_B)
        for i in range(3):
            pc += w[2 * i] * half * self.r_I[i] ** 2 + w[2 * i + 1] * self.r_I[i]
        self.path_cost = pc
        dr, dv, dw = self.r_I - gr, self.v_I - gv, self.w_B - gw
        # hand-tuned terminal weights of the reference (JinEnv.py:947-950)
        self.final_cost = 1 * dr.dot(dr) + 11 * dv.dot(dv) + 100 * self._attitude_err(gq) + 10 * dw.dot(dw)


class Rocket(_SixDof):
    """JinEnv.py:1248-1551.  13 states, 3-axis gimballed thrust."""

    def __init__(self, project_name='rocket powered landing'):
        self.project_name = project_name
        self._states(['ux', 'uy', 'uz'])

    def initDyn(self, Jx=None, Jy=None, Jz=None, mass=None, l=None):
        learn = []
        self._inertia(Jx, Jy, Jz, mass, l, learn)
        self.dyn_auxvar = sp.Matrix(learn)
        arm = sp.Matrix([-self.l / 2, 0, 0])
        self._assemble(self.T_B, _cross_mat(arm) * self.T_B, sp.Matrix([-10, 0, 0]))

    def _pieces(self):
        nose = _dcm(self.q).T * sp.Matrix([1, 0, 0])
        tilt = nose[1] ** 2 + nose[2] ** 2
        side = self.T_B[1] ** 2 + self.T_B[2] ** 2
        return tilt, side, self.T_B.dot(self.T_B)

    def initCost(self, wr=None, wv=None, wtilt=None, ww=None, wsidethrust=None, wthrust=1.0):
        learn = []
        self.wr = _param(wr, 'wr', learn)
        self.wv = _param(wv, 'wv', learn)
        self.wtilt = _param(wtilt, 'wtilt', learn)
        self.wsidethrust = _param(wsidethrust, 'wsidethrust', learn)
        self.ww = _param(ww, 'ww', learn)
        wthrust = _param(wthrust, 'wthrust', [])
        self.cost_auxvar = sp.Matrix(learn)
        tilt, side, thr = self._pieces()
        self.final_cost = (self.wr * self.r_I.dot(self.r_I) + self.wv * self.v_I.dot(self.v_I) +
                           self.ww * self.w_B.dot(self.w_B) + self.wtilt * tilt)
        self.path_cost = self.final_cost + self.wsidethrust * side + wthrust * thr

    def _axis_weights(self, names):
        w = [SX.sym(nm) for nm in names]
        s = 0
        for i in range(3):
            s += w[i] * self.r_I[i] ** 2 + w[3 + i] * self.v_I[i] ** 2 + w[6 + i] * self.w_B[i] ** 2
        return w, s

    def initCost2(self, wthrust=0.1):
        wthrust = _param(wthrust, 'wthrust', [])
        w, s = self._axis_weights(['wrx', 'wry', 'wrz', 'wvx', 'wvy', 'wvz', 'wwx', 'wwy', 'wwz',
                                   'wsidethrust', 'wtilt'])
        self.cost_auxvar = sp.Matrix(w)
        tilt, side, thr = self._pieces()
        self.final_cost = s + w[10] * tilt
        self.path_cost = self.final_cost + w[9] * side + wthrust * thr

    def initCost_Ex(self, wthrust=0.1):
        wthrust = _param(wthrust, 'wthrust', [])
        w, s = self._axis_weights(['wrx', 'wry', 'wrz', 'wvx', 'wvy', 'wvz', 'wwx', 'wwy', 'wwz',
                                   'wtilt', 'wsidethrust'])
        self.cost_auxvar = sp.Matrix(w)
        tilt, side, thr = self._pieces()
        self.final_cost = s + w[9] * tilt + w[10] * side
        self.path_cost = self.final_cost + wthrust * thr
