"""[UPSTREAM] gym_pybullet_drones.envs.CtrlAviary as the reference uses it
(PIDEnv.py:106-116, simulations/EnvGeometric.py:89-100): RPM actions in, 20-float
per-drone state vectors out, constant reward -1, never terminates."""
from __future__ import annotations

import numpy as np

from .BaseAviary import BaseAviary, DroneModel, Physics

__all__ = ["CtrlAviary", "DroneModel", "Physics"]


class CtrlAviary(BaseAviary):
    def __init__(self, drone_model: DroneModel = DroneModel.CF2X, num_drones: int = 1, neighbourhood_radius: float = np.inf,
                 initial_xyzs=None, initial_rpys=None, physics: Physics = Physics.PYB, pyb_freq: int = 240,
                 ctrl_freq: int = 240, gui=False, record=False, obstacles=False, user_debug_gui=True,
                 output_folder="results", **batch_kwargs):
        super().__init__(drone_model=drone_model, num_drones=num_drones, neighbourhood_radius=neighbourhood_radius,
                         initial_xyzs=initial_xyzs, initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq,
                         ctrl_freq=ctrl_freq, gui=gui, record=record, obstacles=obstacles,
                         user_debug_gui=user_debug_gui, output_folder=output_folder, **batch_kwargs)

    def _computeReward(self):
        return -1

    def _computeTerminated(self):
        return False

    def _computeTruncated(self):
        return False

    def _computeInfo(self):
        return {"answer": 42}
