"""model/linear_crazyflie.py of the reference: the 7-state model x = [yaw, x, y, z, vx, vy, vz], u = [f, pitch, roll, yaw_rate] the
Crazyflie firmware interface allows (:27-52).  Constant (A, B) only; ``calc_xdot`` is broken in the reference (the 12-long
obs_to_lin_model(obs) against the 7 x 7 A, :54-72) and raises the same ValueError here."""
import numpy as np


class CrazyflieModel:
    def __init__(self, env, debug=False):
        self.env, self.mass, self.g = env, env.M, env.G
        self.init_matrices()
        if debug:
            print("A matrix: ", self.A, "B matrix: ", self.B, sep="\n")

    def init_matrices(self):
        """x = [yaw, x, y, z, vx, vy, vz], u = [f, pitch, roll, yaw_rate] (:40-52): positions integrate velocities, yaw integrates the
        yaw-rate input, vz the thrust; pitch / roll act on the x / y rows through g (the reference's `B[1:3, 1:3] = [[0, g], [-g, 0]]`)."""
        A, B = np.zeros((7, 7)), np.zeros((7, 4))
        A[[1, 2, 3], [4, 5, 6]] = 1.0
        B[0, 3], B[6, 0] = 1.0, 1.0 / self.mass
        B[1, 2], B[2, 1] = self.g, -self.g
        self.A, self.B = A, B
        self.Ahat, self.Bhat = A.copy(), B.copy()

    def calc_xdot_from_obs(self, obs):
        return self.calc_xdot(None, None)

    def calc_xdot(self, x, action):
        raise ValueError("matmul: Input operand 1 has a mismatch in its core dimension 0, with gufunc signature (n?,k),(k,m?)->(n?,m?) "
                         "(size 12 is different from 7)  [CrazyflieModel.calc_xdot is broken in the reference, model/linear_crazyflie.py:62-72]")
