"""``from model import ...`` of the reference (model/__init__.py) for the models on the hot path."""
from .dynamics import QuadrotorDynamics  # noqa: F401
from .linear_omega import LinearizedOmegaModel  # noqa: F401
from .linear_yank_omega import LinearizedYankOmegaModel  # noqa: F401
from .linearized import LinearizedModel  # noqa: F401


class CrazyflieModel:
    """Out of scope: the reference's 7-state model cannot run there either (model/linear_crazyflie.py:62-72)."""
    def __init__(self, *a, **k):
        raise NotImplementedError("CrazyflieModel is outside this build's hot path (broken in the reference: calc_xdot multiplies a 12-long state by a 7 x 7 A)")
