"""utils/env_builder.py of the reference (:4-10): ``Environment(G, M, MAX_THRUST, CTRL_TIMESTEP, DRONE_MODEL)``, the plain-attribute
stand-in for an env that the reference fills from a yaml file and hands to its models and controllers.  Host data only."""
from dataclasses import dataclass

from .enums import DroneModel


@dataclass
class Environment:
    G: float
    M: float
    MAX_THRUST: float
    CTRL_TIMESTEP: float
    DRONE_MODEL: DroneModel = DroneModel.CF2X
