"""[UPSTREAM] gym_pybullet_drones.utils.enums -- names the reference imports
(PIDEnv.py:10, utils/env_builder.py:2)."""
from enum import Enum


class DroneModel(Enum):
    CF2X = "cf2x"
    CF2P = "cf2p"
    RACE = "racer"


class Physics(Enum):
    PYB = "pyb"
    DYN = "dyn"
    PYB_GND = "pyb_gnd"
    PYB_DRAG = "pyb_drag"
    PYB_DW = "pyb_dw"
    PYB_GND_DRAG_DW = "pyb_gnd_drag_dw"
