"""[UPSTREAM] gym_pybullet_drones.utils.utils: sync / str2bool as the reference imports
them (PIDEnv.py:12, simulations/EnvGeometric.py:11)."""
import argparse
import time


def sync(i, start_time, timestep):
    """Throttle a loop to wall-clock real time (the reference calls this every step:
    PIDEnv.py:179).  Batched GPU runs normally skip it."""
    if timestep > .04 or i % (int(1 / (24 * timestep))) == 0:
        elapsed = time.time() - start_time
        if elapsed < (i * timestep):
            time.sleep(timestep * i - elapsed)


def str2bool(val):
    if isinstance(val, bool):
        return val
    elif val.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    elif val.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    else:
        raise argparse.ArgumentTypeError("[ERROR] in str2bool(), a Boolean value is expected")
