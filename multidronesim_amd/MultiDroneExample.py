"""MultiDroneExample.py of the reference (BASELINE config 1: N drones hovering 1 m above their start
under the halved-gain DSLPID): same ``parse_args()`` flags (:27-53), ``create_env(args)`` and
``do_control(args, env)`` (:74-126).  Differences: ``create_env`` returns the env (the reference prints
M, J and calls ``exit()``, :68-70), ``--num_envs`` batches the scene, ``--realtime`` keeps or drops the
wall-clock ``sync`` and rendering, and the per-drone PID runs fused with the physics step on the GPU."""
import argparse
import time

import numpy as np

from .control.DSLPIDControl import DSLPIDControl
from .envs.BaseAviary import DroneModel, Physics
from .envs.CtrlAviary import CtrlAviary
from .utils.utils import str2bool, sync

DEFAULT_DRONES = DroneModel("cf2p")
DEFAULT_PHYSICS = Physics("pyb")
DEFAULT_GUI = True
DEFAULT_PLOT = True
DEFAULT_RECORD = False
DEFAULT_USER_DEBUG_GUI = False
DEFAULT_SIMULATION_FREQ_HZ = 100
DEFAULT_CONTROL_FREQ_HZ = 100
DEFAULT_DURATION_SEC = 30
DEFAULT_OUTPUT_FOLDER = 'results'
DEFAULT_NUM_DRONES = 2


# flag, default, type, help -- the reference's flags (:27-53) followed by the two this package adds
_FLAGS = (
    ("drone", DEFAULT_DRONES, DroneModel, "drone model (cf2p / cf2x)"),
    ("num_drones", DEFAULT_NUM_DRONES, int, "drones per env"),
    ("physics", DEFAULT_PHYSICS, Physics, "physics mode"),
    ("gui", DEFAULT_GUI, str2bool, "accepted for compatibility (there is no GUI)"),
    ("plot", DEFAULT_PLOT, str2bool, "accepted for compatibility"),
    ("user_debug_gui", DEFAULT_USER_DEBUG_GUI, str2bool, "accepted for compatibility"),
    ("simulation_freq_hz", DEFAULT_SIMULATION_FREQ_HZ, int, "physics rate"),
    ("control_freq_hz", DEFAULT_CONTROL_FREQ_HZ, int, "control rate"),
    ("duration_sec", DEFAULT_DURATION_SEC, int, "simulated seconds"),
    ("output_folder", DEFAULT_OUTPUT_FOLDER, str, "kept for compatibility (nothing is logged to disk)"),
    ("init_rad", 1.0, float, "radius of the start circle"),
    ("num_envs", 1, int, "independent copies of the scene stepped together"),
    ("realtime", True, str2bool, "throttle to the wall clock and render like the reference"),
)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Multi-drone hover example on the batched MI355X step")
    for flag, default, kind, text in _FLAGS:
        extra = {"choices": kind} if kind in (DroneModel, Physics) else {}
        parser.add_argument("--" + flag, default=default, type=kind, help=text, metavar="", **extra)
    return parser.parse_args(argv)


def initial_conditions(args):
    """The __main__ block of the reference (:129-148): drone 0 at the origin, the rest on a circle; targets 1 m above."""
    init = np.zeros((args.num_drones, 3))
    for i in range(1, args.num_drones):
        init[i, 0] = args.init_rad * np.cos((i / args.num_drones) * 2 * np.pi)
        init[i, 1] = args.init_rad * np.sin((i / args.num_drones) * 2 * np.pi)
    targets = init + np.array([0.0, 0.0, 1.0])
    return init, np.zeros((args.num_drones, 3)), targets, np.zeros((args.num_drones, 3))


def create_env(args, init_xyzs, init_rpys):
    return CtrlAviary(drone_model=args.drone, num_drones=args.num_drones, initial_xyzs=init_xyzs, initial_rpys=init_rpys,
                      physics=args.physics, pyb_freq=args.simulation_freq_hz, ctrl_freq=args.control_freq_hz, gui=args.gui,
                      user_debug_gui=args.user_debug_gui, output_folder=args.output_folder, num_envs=args.num_envs)


def do_control(args, env, target_positions, target_rpys):
    env.getPyBulletClient()
    env.getDroneIds()
    env._showDroneLocalAxes(0)
    ctrl = []
    if args.drone in [DroneModel.CF2X, DroneModel.CF2P]:
        for i in range(args.num_drones):
            ctrl.append(DSLPIDControl(drone_model=args.drone))
            ctrl[i].P_COEFF_FOR = 0.5 * np.array([.4, .4, 1.25])
            ctrl[i].I_COEFF_FOR = 0.5 * np.array([.05, .05, .05])
            ctrl[i].D_COEFF_FOR = 0.5 * np.array([.2, .2, .5])
            ctrl[i].P_COEFF_TOR = 0.5 * np.array([70000., 70000., 60000.])
            ctrl[i].I_COEFF_TOR = 0.5 * np.array([.0, .0, 500.])
            ctrl[i].D_COEFF_TOR = 0.5 * np.array([20000., 20000., 12000.])
        env.set_dslpid_gains(ctrl[0])
    start = time.time()
    shape = (args.num_drones, 4) if args.num_envs == 1 else (args.num_envs, args.num_drones, 4)
    obs, _, _, _, _ = env.step(np.zeros(shape))
    ctrl_steps = int(args.duration_sec * env.CTRL_FREQ)
    if args.realtime:
        for i in range(ctrl_steps):
            obs = env.step_dslpid(target_positions, target_rpys)  # per-drone computeControlFromState + env.step, fused
            env.render()
            sync(i, start, env.CTRL_TIMESTEP)
    elif ctrl_steps > 0:
        obs = env.rollout_dslpid(target_positions, target_rpys, ctrl_steps)   # the same loop, issued by one C call
    final = obs.double().cpu().numpy()
    env.close()
    return final


if __name__ == "__main__":
    ARGS = parse_args()
    INIT_XYZS, INIT_RPYS, TARGET_POSITIONS, TARGET_RPYS = initial_conditions(ARGS)
    ENV = create_env(ARGS, INIT_XYZS, INIT_RPYS)
    FINAL = do_control(ARGS, ENV, TARGET_POSITIONS, TARGET_RPYS)
    print("final positions (env 0):\n", FINAL.reshape(ARGS.num_envs, ARGS.num_drones, 20)[0, :, 0:3])
