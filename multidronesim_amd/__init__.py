"""multidronesim_amd -- MI355X-native batched multi-drone step.

Drop-in for the one hot path of JasonTStanley/MultiDroneSim: ``CtrlAviary`` (the env the
reference drives through gym-pybullet-drones), ``GeometricControl``, ``Lemniscate``,
``model_conversions`` and ``MultiDroneEnv``, all backed by hand-written HIP kernels
reached through the C-ABI in ``include/mds.h`` (``libmds.so``).  There is no CPU path."""
from ._capi import MdsError, load_library  # noqa: F401

__version__ = "0.2.0"
