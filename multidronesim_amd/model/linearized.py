"""model/linearized.py of the reference: ``LinearizedModel(env)`` -- hover linearisation with the 12-state
x = [r, p, y, r_dot, p_dot, y_dot, vx, vy, vz, px, py, pz] and input u = [F, tau_x, tau_y, tau_z] (:50-75), plus the
deliberately wrong (Ahat, Bhat) the scripts use as a 'noisy model' (inertia and mass off by 0.75).  Only the constant
matrices are on the hot path: they parameterise the LQR gain of control/lqr/lqr_controller.py."""
import numpy as np


class LinearizedModel:
    def __init__(self, env, debug=False):
        self.mass = env.M
        self.Ixx, self.Iyy, self.Izz = env.J[0, 0], env.J[1, 1], env.J[2, 2]
        self.g = env.G
        self.env = env
        self.A = np.zeros((12, 12))
        self.B = np.zeros((12, 4))
        self.C = np.eye(12)
        self.D = np.zeros((12, 6))
        self.init_matrices()

    def init_matrices(self):
        self.A[0:3, 3:6] = np.eye(3)
        self.A[9:, 6:9] = np.eye(3)
        self.A[6, 1] = self.g
        self.A[7, 0] = -self.g
        self.B[8, 0] = 1.0 / self.mass
        self.B[3:6, 1:] = np.diag([1 / self.Ixx, 1 / self.Iyy, 1 / self.Izz])
        self.D[:, 2:] = self.B.copy()
        self.D[7, 1] = 1.0 / self.mass
        self.D[6, 0] = 1.0 / self.mass
        self.Ahat = self.A.copy()
        self.Bhat = self.B.copy()
        self.Bhat[3:6, 1:] = np.diag([1 / self.Ixx, 1 / self.Iyy, 1 / self.Izz]) * 0.75
        self.Bhat[8, 0] = 1.0 / (self.mass * .75)
