"""Mirror of the reference's quadrotor driver (lib/QuadAlgorithm.py + lib/QuadPara.py, QuadStates.py,
DemoSparse.py, ObsInfo.py) on top of the HIP path.

Same constructor, ``load_optimization_function(para_dict)``, ``run(...)`` and ``getloss_pos_corrections``;
the learning loop (QuadAlgorithm.py:231-257) runs on the device through ``CPDP.SparseDemoLearner``.
Plotting / animation (QuadAlgorithm.py:260-281, 354-451, 581-613) is UI and is not reproduced; ``run`` returns the
dictionary the reference saves to ``data/uav_results_random_*.mat`` (QuadAlgorithm.py:324-333) and writes it only if
``save_flag`` is set.

Extension: ``run(..., initial_parameters=[B,7])`` learns B independent seeds in lock-step.
"""
import os
import time
from dataclasses import dataclass, field

import numpy as np
import torch

from . import CPDP, JinEnv
from .JinEnv import QuadStates  # noqa: F401  (lib/QuadStates.py)
from .symbolic import SX, vertcat


@dataclass
class QuadPara:
    """lib/QuadPara.py."""
    inertial_x: float = 1
    inertial_y: float = 1
    inertial_z: float = 1
    mass: float = 1
    l: float = 1
    c: float = 1

    def __init__(self, inertial_list, mass, l, c):
        self.inertial_x, self.inertial_y, self.inertial_z = inertial_list
        self.mass, self.l, self.c = mass, l, c


@dataclass
class DemoSparse:
    """lib/DemoSparse.py."""
    waypoints: list = field(default_factory=lambda: [[0, 0, 0], [0, 0, 0], [0, 0, 0]])
    time_list: list = field(default_factory=lambda: [1, 2, 3])
    time_horizon: float = 4


@dataclass
class ObsInfo:
    """lib/ObsInfo.py (only carried along; obstacles are plotted, never used by the algorithm)."""
    length: float = 1
    width: float = 1
    height: float = 1
    center: list = field(default_factory=lambda: [0, 0, 0])

    def __init__(self, center_pisition, size_list):
        self.center = center_pisition
        self.length, self.width, self.height = size_list


class QuadAlgorithm(object):
    def __init__(self, config_data, QuadParaInput, n_grid, device=None, dtype=torch.float32):
        self.QuadPara = QuadParaInput
        self.n_grid = n_grid
        self.space_limit_x = config_data["LAB_SPACE_LIMIT"]["LIMIT_X"]
        self.space_limit_y = config_data["LAB_SPACE_LIMIT"]["LIMIT_Y"]
        self.space_limit_z = config_data["LAB_SPACE_LIMIT"]["LIMIT_Z"]
        self.quad_average_speed = float(config_data["QUAD_AVERAGE_SPEED"])
        self.device, self.dtype = device, dtype
        self.library = None          # tests may bind a prebuilt library (oc.use_library)

    def settings(self, QuadDesiredStates):
        """QuadAlgorithm.py:74-130: environment, time-warped OC system, interface = position."""
        self.env = JinEnv.Quadrotor()
        Q = self.QuadPara
        self.env.initDyn(Jx=Q.inertial_x, Jy=Q.inertial_y, Jz=Q.inertial_z, mass=Q.mass, l=Q.l, c=Q.c)
        self.env.initCost_Polynomial(QuadDesiredStates, w_thrust=0.1)
        self.oc = CPDP.COCSys()
        beta = SX.sym('beta')
        self.oc.setAuxvarVariable(vertcat(beta, self.env.cost_auxvar))
        self.oc.setStateVariable(self.env.X)
        self.oc.setControlVariable(self.env.U)
        self.oc.setDyn(beta * self.env.f)
        self.oc.setPathCost(beta * self.env.path_cost)
        self.oc.setFinalCost(self.env.final_cost)
        self.oc.setIntegrator(self.n_grid)
        self.oc.sys_name = "quadrotor_poly_tw"
        if self.library is not None:
            self.oc.use_library(self.library)
        self.oc.setDevice(self.device, self.dtype)
        self.interface_pos_idx = [0, 1, 2]
        self.interface_ori_idx = [6, 7, 8, 9]
        if self.optimization_method_str not in ("Vanilla", "Nesterov", "Adam", "Nadam", "AMSGrad"):
            raise Exception("Wrong optimization method type!")

    def load_optimization_function(self, para_input):
        """QuadAlgorithm.py:133-191 (same dictionary keys)."""
        self.learning_rate = para_input["learning_rate"]
        self.iter_num = para_input["iter_num"]
        self.optimization_method_str = para_input["method"]
        self.opt_kwargs = {}
        m = para_input["method"]
        if m == "Vanilla":
            pass
        elif m == "Nesterov":
            self.mu_momentum = para_input["mu"]
            self.actual_loss_print_nesterov_flag = para_input["true_loss_print_flag"]
            self.opt_kwargs = dict(mu=self.mu_momentum, true_loss_print_flag=self.actual_loss_print_nesterov_flag)
        elif m in ("Adam", "Nadam", "AMSGrad"):
            self.opt_kwargs = dict(beta_1=para_input["beta_1"], beta_2=para_input["beta_2"],
                                   epsilon=para_input["epsilon"])
        else:
            raise Exception("Wrong optimization method type!")

    def run(self, QuadInitialCondition, QuadDesiredStates, SparseInput, ObsList=(), print_flag=False, save_flag=False,
            initial_parameters=None, save_dir=None):
        t0 = time.time()
        self.ObsList = ObsList
        self.settings(QuadDesiredStates)
        self.ini_state = (list(QuadInitialCondition.position) + list(QuadInitialCondition.velocity) +
                          list(QuadInitialCondition.attitude_quaternion) + list(QuadInitialCondition.angular_velocity))
        # QuadAlgorithm.py:221-223: the reference normalises the horizon to 1 and the waypoint times with it
        self.time_horizon = 1.0
        self.time_list_sparse = np.array(SparseInput.time_list) / SparseInput.time_horizon
        self.waypoints = np.array(SparseInput.waypoints)
        theta0 = np.array([1, 0.1, 0.1, 0.1, 0.1, 0.1, -1], dtype=float) if initial_parameters is None else \
            np.asarray(initial_parameters, dtype=float)                    # QuadAlgorithm.py:235
        self.learner = CPDP.SparseDemoLearner(self.oc, self.ini_state if theta0.ndim == 1 else
                                              np.tile(self.ini_state, (theta0.shape[0], 1)),
                                              self.time_horizon, self.time_list_sparse, self.waypoints,
                                              self.interface_pos_idx, theta0, method=self.optimization_method_str,
                                              learning_rate=self.learning_rate, **self.opt_kwargs)
        self.loss_trace, self.parameter_trace = [], [self.learner.theta.cpu().numpy().copy()]
        loss, diff_loss_norm = 100.0, 100.0
        for j in range(self.iter_num):
            if (loss > 0.9) and (diff_loss_norm > 0.05):                    # QuadAlgorithm.py:242
                l, g = self.learner.step()
                loss = float(l.max())                                       # every seed must pass the stop test
                diff_loss_norm = float(torch.linalg.norm(g, dim=1).max())
                self.loss_trace.append(l.cpu().numpy().copy())
                self.parameter_trace.append(self.learner.theta.cpu().numpy().copy())
                if print_flag:
                    print('iter:', j, ', loss:', self.loss_trace[-1], ', loss gradient norm:', diff_loss_norm)
            else:
                if print_flag:
                    print("The loss is less than threshold, stop the iteration.")
                break
        horizon = self.time_horizon
        current_parameter = self.parameter_trace[-1][0]
        _, opt_sol = self.oc.cocSolver(self.ini_state, horizon, current_parameter)
        time_steps = np.linspace(0, horizon, num=100 + 1)                   # QuadAlgorithm.py:309
        opt_traj = opt_sol(time_steps)
        n, m = self.oc.n_state, self.oc.n_control
        results = {'parameter_trace': np.array(self.parameter_trace), 'loss_trace': np.array(self.loss_trace),
                   'learning_rate': self.learning_rate, 'waypoints': self.waypoints,
                   'time_grid': self.time_list_sparse, 'time_steps': time_steps,
                   'opt_state_traj': opt_traj[:, :n], 'opt_control_traj': opt_traj[:, n:n + m],
                   'horizon': horizon, 'T': self.time_horizon, 'seconds': time.time() - t0}
        if save_flag:
            import scipy.io as sio
            d = save_dir or os.path.join(os.getcwd(), 'data')
            os.makedirs(d, exist_ok=True)
            sio.savemat(os.path.join(d, 'uav_results_random_' + time.strftime("%Y%m%d%H%M%S") + '.mat'),
                        {'results': results})
        return results

    def getloss_pos_corrections(self, time_grid, target_waypoints, opt_sol, auxsys_sol):
        """QuadAlgorithm.py:616-639 on host objects returned by cocSolver / auxSysSolver (same formula the kernel fuses)."""
        n, p = self.oc.n_state, self.oc.n_auxvar
        loss, diff_loss = 0.0, np.zeros(p)
        for k, t in enumerate(time_grid):
            target = np.asarray(target_waypoints[k])[0:3]
            cur = opt_sol(t)[0:n][0:3]
            loss += np.linalg.norm(target - cur) ** 2
            dxpos_dp = auxsys_sol(t)[0:n * p].reshape((n, p))
            diff_loss += (cur - target) @ dxpos_dp[0:3]
        return loss, diff_loss
