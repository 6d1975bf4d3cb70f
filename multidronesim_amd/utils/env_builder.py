"""utils/env_builder.py of the reference: ``Environment(G, M, MAX_THRUST, CTRL_TIMESTEP, DRONE_MODEL)`` -- the plain-attribute stand-in
for an env the reference builds from a yaml file (:4-10).  Host data only."""
from .enums import DroneModel


class Environment:
    def __init__(self, G, M, MAX_THRUST, CTRL_TIMESTEP, DRONE_MODEL=DroneModel.CF2X):
        self.G = G
        self.M = M
        self.MAX_THRUST = MAX_THRUST
        self.CTRL_TIMESTEP = CTRL_TIMESTEP
        self.DRONE_MODEL = DRONE_MODEL
