"""trajectories/ of the reference: same class names and ``__call__(t)`` 5-tuples
(trajectories/__init__.py star-imports all of them)."""
from .base import TrajectoryBase  # noqa: F401
from .Lemniscate import Lemniscate  # noqa: F401
from .Circle import CircleTrajectory  # noqa: F401
from .LineTrajectory import LineTrajectory, WaitTrajectory  # noqa: F401
from .CompoundTrajectory import CompoundTrajectory  # noqa: F401
from .RotateTrajectory import RotateTrajectory  # noqa: F401
