"""Launch form 1 (one launch of k_step_geometric per control step; the library's stream policy) against form 2 (k_rollout_geometric, 50 control
steps per launch, the state in registers) at and ABOVE BASELINE config 3's size, per dtype: 65 536 / 262 144 / 524 288 envs x 8 drones,
float32 / float32c / float64, calls of 20 and 200 control steps, every step's observation written in both.  One untimed call through the
same branch, then `reps` timed calls; HIP events on the launch stream, median.
Run on the GPU box from the repo root:  python3 profiles/tools/r04_form_sweep.py [out.json [reps]]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04_form_sweep.json"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
D = 8
rec = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "reps": reps, "cases": []}
for E in (65536, 262144, 524288):
    for dtype in ("float32", "float32c", "float64"):
        xyz, rpy, P = bench.make_inputs(E, D, "c3", 1000)
        env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100,
                         num_envs=E, dtype=dtype, device=0)
        env.set_trajectories(P)
        env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
        dev, st = env.device, torch.cuda.current_stream(env.device)
        case = {"envs": E, "drones": E * D, "dtype": dtype, "forms": {}}
        for name, form in (("form1", 1), ("form2_50", 2)):
            env.set_rollout_form(form, 50)
            env.set_rollout_streams(0)
            res = {}
            for steps in (20, 200):
                env.rollout_geometric(0.0, steps, want_obs=True, obs_every_step=True)
                torch.cuda.synchronize(dev)
                ev = []
                for r in range(reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    env.rollout_geometric(0.01 * steps * (r + 1), steps, want_obs=True, obs_every_step=True)
                    e1.record(st)
                    torch.cuda.synchronize(dev)
                    ev.append(e0.elapsed_time(e1) * 1e3 / steps)
                assert env.last_rollout_form() == form
                res[str(steps)] = {"us_per_step": float(np.median(ev)), "min": float(min(ev)), "streams": env.last_rollout_streams()}
            case["forms"][name] = res
        print(json.dumps(case), flush=True)
        rec["cases"].append(case)
        env.close()
        del env
        torch.cuda.empty_cache()
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(rec, open(out_path, "w"), indent=1)
