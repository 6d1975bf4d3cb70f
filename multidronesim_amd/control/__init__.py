"""``from control import ...`` of the reference (control/__init__.py) for the controllers on the hot path."""
from .geometric import GeometricControl  # noqa: F401
from .low_level.thrust_omega_ctrl import ThrustOmegaController  # noqa: F401
from .low_level.yank_omega_ctrl import YankOmegaController  # noqa: F401
from .lqr.lqr_omega_controller import LQROmegaController  # noqa: F401
from .lqr.lqr_YO_controller import LQRYankOmegaController  # noqa: F401
from .lqr.lqr_controller import LQRController  # noqa: F401
from .lqr.crazyflie_lqr_controller import CrazyflieLQR  # noqa: F401
