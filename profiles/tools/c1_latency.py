import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from multidronesim_amd import MultiDroneExample as M
args = M.parse_args(["--num_drones", "2", "--duration_sec", "10", "--simulation_freq_hz", "240", "--control_freq_hz", "240"])
ix, ir, tp, tr = M.initial_conditions(args)
env = M.create_env(args, ix, ir)
t0 = time.perf_counter(); final = M.do_control(args, env, tp, tr); el = time.perf_counter() - t0
print("C1: MultiDroneExample 2 drones, 240 Hz x 10 s = 2400 control steps: %.3f s wall = %.0f drone-steps/s; final z %s" % (el, 2 * 2400 / el, np.round(final.reshape(-1, 20)[:, 2], 3)))
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics
env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=2, initial_xyzs=ix, initial_rpys=ir, physics=Physics.PYB, pyb_freq=240, ctrl_freq=240)
a = np.full((2, 4), env.HOVER_RPM)
for _ in range(50): env.step(a)
t0 = time.perf_counter()
for _ in range(2000): obs, *_ = env.step(a)
el = time.perf_counter() - t0
print("reference-shaped env.step(np.ndarray[2,4]) -> np.ndarray[2,20]: %.1f us per call (H2D action + launch + D2H obs + sync)" % (el / 2000 * 1e6))
