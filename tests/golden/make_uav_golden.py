"""Extract the golden vectors the reference ships with its own saved run.

Source (data, not code): /root/reference/data/uav_results_random_20210308113016.mat
written by lib/QuadAlgorithm.py:322-339 from Examples/quad_example_human_input.py
(Quadrotor J=(1,1,1) m=1 l=1 c=0.02, n_grid=25, T=1, ini r=(-2,-1,0.6),
goal r=(2.5,1,1.5); Nesterov lr=0.01 mu=0.9, true_loss_print_flag=False,
100 iterations).

With Nesterov (QuadAlgorithm.py:469-495, flag False) the saved traces satisfy
    v_{j+1} = theta_{j+1} - theta_j          (projection never active: beta >> 1e-8)
    v_{j+1} = mu v_j - lr g_j
so   g_j = (mu v_j - v_{j+1}) / lr   is the reference's diff_loss at the
look-ahead point   theta_j + mu v_j,   and loss_trace[j] is its loss there.
That gives 100 (theta, loss, dtheta) triples produced by the real CasADi/IPOPT
pipeline, plus the final optimal trajectory.

Run here (needs /root/reference):  python tests/golden/make_uav_golden.py
"""
import os
import numpy as np
import scipy.io as sio

SRC = '/root/reference/data/uav_results_random_20210308113016.mat'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'uav_golden.npz')

r = sio.loadmat(SRC)['results'][0, 0]
theta = r['parameter_trace'].astype(np.float64)          # (101, 7)
loss = r['loss_trace'].ravel().astype(np.float64)        # (100,)
lr, mu = float(r['learning_rate'][0, 0]), 0.9
v = np.vstack([np.zeros((1, theta.shape[1])), np.diff(theta, axis=0)])   # v_0 = 0, v_{j+1}
lookahead = theta[:-1] + mu * v[:-1]
grad = (mu * v[:-1] - v[1:]) / lr
np.savez(OUT,
         theta_trace=theta, loss_trace=loss, lookahead_theta=lookahead, grad_trace=grad,
         learning_rate=lr, mu=mu,
         waypoints=r['waypoints'].astype(np.float64), taus=r['time_grid'].ravel().astype(np.float64),
         time_steps=r['time_steps'].ravel(), opt_state_traj=r['opt_state_traj'], opt_control_traj=r['opt_control_traj'],
         horizon=float(r['horizon'][0, 0]), n_grid=25,
         ini_state=np.array([-2.0, -1.0, 0.6, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float64),
         goal_r=np.array([2.5, 1.0, 1.5]), goal_v=np.zeros(3), goal_q=np.array([1.0, 0, 0, 0]), goal_w=np.zeros(3),
         quad_para=np.array([1.0, 1.0, 1.0, 1.0, 1.0, 0.02]))
print('wrote', OUT, os.path.getsize(OUT), 'bytes')
print('grad[0] =', grad[0]); print('grad[50] =', grad[50])
