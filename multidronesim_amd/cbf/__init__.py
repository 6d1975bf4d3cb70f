"""``from cbf import ...`` of the reference (cbf/__init__.py)."""
from .cbf import DroneCBF  # noqa: F401
from .qptracker import DroneQPTracker  # noqa: F401
