"""model/linear_crazyflie.py of the reference: the 7-state model x = [yaw, x, y, z, vx, vy, vz], u = [f, pitch, roll, yaw_rate] the
Crazyflie firmware interface allows (:27-52).  Constant (A, B) only; ``calc_xdot`` is broken in the reference (the 12-long
obs_to_lin_model(obs) against the 7 x 7 A, :54-72) and raises the same ValueError here."""
import numpy as np


class CrazyflieModel:
    def __init__(self, env, debug=False):
        self.mass = env.M
        self.g = env.G
        self.A = np.zeros((7, 7))
        self.B = np.zeros((7, 4))
        self.env = env
        self.Ahat = np.zeros((7, 7))
        self.Bhat = np.zeros((7, 4))
        self.init_matrices()
        if debug:
            print("A matrix: ")
            print(self.A)
            print("B matrix: ")
            print(self.B)

    def init_matrices(self):
        self.A[1:4, 4:7] = np.eye(3)
        self.B[0, -1] = 1
        self.B[-1, 0] = 1.0 / self.mass
        self.B[1:3, 1:3] = np.array([[0, self.g], [-self.g, 0]])
        self.Ahat = self.A.copy()
        self.Bhat = self.B.copy()

    def calc_xdot_from_obs(self, obs):
        return self.calc_xdot(None, None)

    def calc_xdot(self, x, action):
        raise ValueError("matmul: Input operand 1 has a mismatch in its core dimension 0, with gufunc signature (n?,k),(k,m?)->(n?,m?) "
                         "(size 12 is different from 7)  [CrazyflieModel.calc_xdot is broken in the reference, model/linear_crazyflie.py:62-72]")
