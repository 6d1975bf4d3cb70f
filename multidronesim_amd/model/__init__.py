"""``from model import ...`` of the reference (model/__init__.py) for the models on the hot path."""
from .dynamics import QuadrotorDynamics  # noqa: F401
from .linear_omega import LinearizedOmegaModel  # noqa: F401
from .linear_yank_omega import LinearizedYankOmegaModel  # noqa: F401
from .linearized import LinearizedModel  # noqa: F401
from .linear_crazyflie import CrazyflieModel  # noqa: F401
