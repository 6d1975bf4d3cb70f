"""control/lqr/crazyflie_lqr_controller.py of the reference: ``CrazyflieLQR(env, CrazyflieModel(env), DSLPIDControl(...))`` -- Bryson
weights on the 7-state Crazyflie model, gain from the continuous ARE on the host (:12-62), like the reference.  It cannot run in the
reference: the ARE has no finite solution for that model (vx and vy have no input and sit on the imaginary axis: scipy raises
LinAlgError in the constructor -- recorded in tests/golden/crazyflie_model.npz), and ``compute`` would apply the 4 x 7 gain to the 9-long
error of obs_to_lin_model(obs, dim=9) (:107-121).  The same calls raise the same errors here (like QuadrotorDynamics.step);
simulations/EnvGeometricCrazyflie.py, its only user, therefore has no loop to mirror."""
import numpy as np
import scipy.linalg as la

from ..base_controller import BaseController


class CrazyflieLQR(BaseController):
    def __init__(self, env, lin_model, crazyflie_controller=None, debug=False):
        super().__init__(env)
        self.crazyflie_controller = crazyflie_controller
        max_thrust = env.MAX_THRUST
        max_pitch_roll = .0001
        max_yaw_rate_error = 0.1
        rflat = [1 / (max_thrust ** 2), 1 / (max_pitch_roll ** 2), 1 / (max_pitch_roll ** 2), 1 / (max_yaw_rate_error ** 2)]
        max_vel_error = .15
        max_pos_error = .05
        max_yaw_error = np.pi / 40
        qflat = [1 / (max_yaw_error ** 2), 1 / (max_pos_error ** 2), 1 / (max_pos_error ** 2), 1 / (max_pos_error ** 2),
                 1 / (max_vel_error ** 2), 1 / (max_vel_error ** 2), 1 / (max_vel_error ** 2)]
        self.lin_model = lin_model
        self.Q = np.diag(qflat)
        self.R = np.diag(rflat)
        self.debug = debug
        self.P = None
        self.K = None
        self.compute_gain_matrix()

    def compute_gain_matrix(self):
        self.P = la.solve_continuous_are(self.lin_model.A, self.lin_model.B, self.Q, self.R, e=None, s=None, balanced=True)
        self.K = la.solve(self.R, self.lin_model.B.T @ self.P)

    def set_desired_trajectory(self, robot_idx, desired_pos, desired_vel, desired_acc, desired_yaw, desired_omega):
        self.desired_pos = desired_pos
        self.desired_vel = desired_vel
        self.desired_yaw = desired_yaw

    def step_cost(self, x, u):
        return x.T @ self.Q @ x + u.T @ self.R @ u

    def compute(self, obs):
        raise ValueError("matmul: Input operand 1 has a mismatch in its core dimension 0, with gufunc signature (n?,k),(k,m?)->(n?,m?) "
                         "(size 9 is different from 7)  [CrazyflieLQR.compute is broken in the reference, "
                         "control/lqr/crazyflie_lqr_controller.py:107-117: a 4 x 7 gain against the 9-state error]")
