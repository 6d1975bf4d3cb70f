"""model/linear_omega.py of the reference: hover linearisation with state
[r,p,y,vx,vy,vz,x,y,z] and input [F, wx, wy, wz] (:46-53).  Only the constant (A, B) pair is on
the hot path -- it parameterises the ECBF rows."""
import numpy as np


class LinearizedOmegaModel:
    def __init__(self, env, debug=False):
        self.mass = env.M
        self.g = env.G
        self.env = env
        self.A = np.zeros((9, 9))
        self.B = np.zeros((9, 4))
        self.C = np.eye(12)
        self.init_matrices()

    def init_matrices(self):
        self.A[6:, 3:6] = np.eye(3)
        self.A[3, 1] = self.g
        self.A[4, 0] = -self.g
        self.B[5, 0] = 1.0 / self.mass
        self.B[:3, 1:] = np.eye(3)
        self.Ahat = self.A.copy()               # the reference's deliberately perturbed model (:56-61)
        self.Ahat[3, 1] = self.g * 1.2
        self.Ahat[4, 0] = -self.g * 1.2
        self.Bhat = self.B.copy()
        self.Bhat[5, 0] = 1.0 / (self.mass * 0.8)

    def calc_xdot_from_obs(self, obs):
        return self.calc_xdot(None, None)

    def calc_xdot(self, x, action):
        """Broken in the reference (linear_omega.py:63-80: the 12-long obs_to_lin_model(obs) against this model's own smaller A); raises the
        same ValueError here.  The 12-state LinearizedModel is the one simulations/CompareModels.py uses."""
        n = self.A.shape[0]
        raise ValueError(f"matmul: Input operand 1 has a mismatch in its core dimension 0, with gufunc signature (n?,k),(k,m?)->(n?,m?) "
                         f"(size 12 is different from {n})  [LinearizedOmegaModel.calc_xdot is broken in the reference]")
