"""model/linear_yank_omega.py of the reference: state [r,p,y,F,vx,vy,vz,x,y,z], input
[Y, wx, wy, wz] (:45-51)."""
import numpy as np


class LinearizedYankOmegaModel:
    def __init__(self, env, debug=False):
        self.mass = env.M
        self.g = env.G
        self.env = env
        self.A = np.zeros((10, 10))
        self.B = np.zeros((10, 4))
        self.C = np.eye(12)
        self.m, self.n = 10, 4
        self.init_matrices()

    def init_matrices(self):
        self.A[7:, 4:7] = np.eye(3)
        self.A[4, 1] = self.g
        self.A[5, 0] = -self.g
        self.A[6, 3] = 1.0 / self.mass
        self.B[:3, 1:] = np.eye(3)
        self.B[3, 0] = 1.0
        self.Ahat = self.A.copy()
        self.Ahat[4, 1] = self.g * 1.2
        self.Ahat[5, 0] = -self.g * 1.2
        self.Ahat[6, 3] = 1.0 / (self.mass * 0.8)
        self.Bhat = self.B.copy()

    def calc_xdot_from_obs(self, obs):
        return self.calc_xdot(None, None)

    def calc_xdot(self, x, action):
        """Broken in the reference (linear_yank_omega.py:62-79: the 12-long obs_to_lin_model(obs) against this model's own smaller A); raises the
        same ValueError here.  The 12-state LinearizedModel is the one simulations/CompareModels.py uses."""
        n = self.A.shape[0]
        raise ValueError(f"matmul: Input operand 1 has a mismatch in its core dimension 0, with gufunc signature (n?,k),(k,m?)->(n?,m?) "
                         f"(size 12 is different from {n})  [LinearizedYankOmegaModel.calc_xdot is broken in the reference]")
