"""``from utils import ...`` of the reference (utils/__init__.py), restricted to the hot path:
model conversions and the loop helpers.  Logging / plotting helpers stay with the reference."""
from .model_conversions import *  # noqa: F401,F403
from .model_conversions import action_to_input, calc_z_thrust, input_to_action, obs_to_geo_model, obs_to_lin_model  # noqa: F401
from .utils import str2bool, sync  # noqa: F401
from .env_builder import Environment  # noqa: F401
