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


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Multi-drone hover example (batched MI355X step)')
    parser.add_argument('--drone', default=DEFAULT_DRONES, type=DroneModel, help='Drone model', metavar='', choices=DroneModel)
    parser.add_argument('--num_drones', default=DEFAULT_NUM_DRONES, type=int, help='Number of drones', metavar='')
    parser.add_argument('--physics', default=DEFAULT_PHYSICS, type=Physics, help='Physics updates', metavar='', choices=Physics)
    parser.add_argument('--gui', default=DEFAULT_GUI, type=str2bool, help='accepted for compatibility (no GUI here)', metavar='')
    parser.add_argument('--plot', default=DEFAULT_PLOT, type=str2bool, help='accepted for compatibility', metavar='')
    parser.add_argument('--user_debug_gui', default=DEFAULT_USER_DEBUG_GUI, type=str2bool, help='accepted for compatibility', metavar='')
    parser.add_argument('--simulation_freq_hz', default=DEFAULT_SIMULATION_FREQ_HZ, type=int, help='Simulation frequency in Hz', metavar='')
    parser.add_argument('--control_freq_hz', default=DEFAULT_CONTROL_FREQ_HZ, type=int, help='Control frequency in Hz', metavar='')
    parser.add_argument('--duration_sec', default=DEFAULT_DURATION_SEC, type=int, help='Duration of the simulation in seconds', metavar='')
    parser.add_argument('--output_folder', default=DEFAULT_OUTPUT_FOLDER, type=str, help='Folder where to save logs', metavar='')
    parser.add_argument('--init_rad', default=1.0, type=float, help='Initial radius of the drones', metavar='')
    parser.add_argument('--num_envs', default=1, type=int, help='independent copies of the scene stepped together', metavar='')
    parser.add_argument('--realtime', default=True, type=str2bool, help='throttle to wall-clock and render like the reference', metavar='')
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
    for i in range(ctrl_steps):
        obs = env.step_dslpid(target_positions, target_rpys)      # per-drone computeControlFromState + env.step, fused
        if args.realtime:
            env.render()
            sync(i, start, env.CTRL_TIMESTEP)
    final = obs.double().cpu().numpy()
    env.close()
    return final


if __name__ == "__main__":
    ARGS = parse_args()
    INIT_XYZS, INIT_RPYS, TARGET_POSITIONS, TARGET_RPYS = initial_conditions(ARGS)
    ENV = create_env(ARGS, INIT_XYZS, INIT_RPYS)
    FINAL = do_control(ARGS, ENV, TARGET_POSITIONS, TARGET_RPYS)
    print("final positions (env 0):\n", FINAL.reshape(ARGS.num_envs, ARGS.num_drones, 20)[0, :, 0:3])
