"""``from control import ...`` of the reference (control/__init__.py) for the controllers on the hot path."""
from .geometric import GeometricControl  # noqa: F401
from .low_level.thrust_omega_ctrl import ThrustOmegaController  # noqa: F401
from .low_level.yank_omega_ctrl import YankOmegaController  # noqa: F401
from .lqr.lqr_omega_controller import LQROmegaController  # noqa: F401
from .lqr.lqr_YO_controller import LQRYankOmegaController  # noqa: F401
from .lqr.lqr_controller import LQRController  # noqa: F401


class CrazyflieLQR:
    """Out of scope: the reference's constructor raises LinAlgError (its Riccati equation has no finite solution)."""
    def __init__(self, *a, **k):
        raise NotImplementedError("CrazyflieLQR is outside this build's hot path (its constructor cannot complete in the reference either)")
