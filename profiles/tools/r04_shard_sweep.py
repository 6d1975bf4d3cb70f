"""The strong-scaling curve of BASELINE config 3 predicted on ONE MI355X (GPU g owns envs [g E / G, (g + 1) E / G) of the 65 536).
For G in 1, 2, 4, 8 the shard of rank 0 (bench.shard_inputs: the same envs the G-GPU job gives that rank) runs alone on this GPU in each
launch form -- (a) one launch per control step on the caller's stream, (b) the same as two half-shard chains, (c) the whole-rollout kernel
in launches of 50 (and 20 / 100) control steps, every step's observation written in all of them -- for calls of 20, 200 and 2000 control
steps: one untimed call of the same length through the same branch, then `reps` timed calls; HIP events on the launch stream and wall clock
(enqueue .. synchronize), median.  Also what the library's auto policy picks (mds_rollout_form_for / mds_rollout_streams_for).
Implied efficiency of a G-GPU run = t(G = 1) / (G x t(shard of G)), best form at each size, same call length.
Run on the GPU box from the repo root:  python3 profiles/tools/r04_shard_sweep.py [out.json [reps]]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics

out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r04_shard_sweep.json"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
E_job, D = 65536, 8
CALLS = (20, 200, 2000)
FORMS = (("per_step_one_stream", 1, 1, 0), ("per_step_two_chains", 1, 2, 0), ("rollout_kernel_50", 2, 1, 50), ("rollout_kernel_20", 2, 1, 20),
         ("rollout_kernel_100", 2, 1, 100))
rec = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "envs_job": E_job, "drones_per_env": D, "dtype": "float32", "reps": reps, "shards": []}
for G in (1, 2, 4, 8):
    xyz, rpy, P, sl = bench.shard_inputs(E_job, D, "c3", 0, G, "strong")
    E = xyz.shape[0]
    env = CtrlAviary(drone_model=DroneModel.CF2P, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.DYN, pyb_freq=100, ctrl_freq=100,
                     num_envs=E, dtype="float32", device=0)
    env.set_trajectories(P)
    env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
    dev, st = env.device, torch.cuda.current_stream(env.device)
    shard = {"G": G, "env_slice": list(sl), "drones": E * D, "auto": {}, "forms": {}}
    for steps in CALLS:
        env.set_rollout_form(0)
        env.set_rollout_streams(0)
        shard["auto"][str(steps)] = {"form": env.rollout_form_for(steps), "streams": env.rollout_streams_for(steps)}
    for name, form, streams, chunk in FORMS:
        env.set_rollout_form(form, chunk)
        env.set_rollout_streams(streams)
        res = {}
        for steps in CALLS:
            env.rollout_geometric(0.0, steps, want_obs=True, obs_every_step=True)
            torch.cuda.synchronize(dev)
            ev, wl = [], []
            for r in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(dev)
                w0 = time.perf_counter()
                e0.record(st)
                env.rollout_geometric(0.01 * steps * (r + 1), steps, want_obs=True, obs_every_step=True)
                e1.record(st)
                torch.cuda.synchronize(dev)
                wl.append((time.perf_counter() - w0) * 1e6 / steps)
                ev.append(e0.elapsed_time(e1) * 1e3 / steps)
            assert env.last_rollout_form() == form and (form == 2 or env.last_rollout_streams() == streams), (name, env.last_rollout_form(), env.last_rollout_streams())
            res[str(steps)] = {"us_per_step_events": float(np.median(ev)), "us_per_step_events_min": float(min(ev)), "us_per_step_wall": float(np.median(wl))}
        sane = bool(torch.isfinite(env._obs).all().item())
        shard["forms"][name] = dict(res, state_sane=sane)
        print(G, E * D, name, {k: round(v["us_per_step_events"], 2) for k, v in res.items()}, {k: round(v["us_per_step_wall"], 2) for k, v in res.items()}, flush=True)
    rec["shards"].append(shard)
    env.close()
    del env
    torch.cuda.empty_cache()
# implied efficiency, best form per size, per call length
t1 = {c: min(f[str(c)]["us_per_step_events"] for f in rec["shards"][0]["forms"].values()) for c in CALLS}
for sh in rec["shards"]:
    best = {}
    for c in CALLS:
        name, t = min(((n, f[str(c)]["us_per_step_events"]) for n, f in sh["forms"].items()), key=lambda x: x[1])
        wall_name, tw = min(((n, f[str(c)]["us_per_step_wall"]) for n, f in sh["forms"].items()), key=lambda x: x[1])
        best[str(c)] = {"form": name, "us_per_step_events": t, "implied_efficiency_vs_one_gpu": t1[c] / (sh["G"] * t), "G_drone_steps_per_s_node": sh["G"] * sh["drones"] / t * 1e-3,
                        "best_by_wall": wall_name, "us_per_step_wall": tw}
    sh["best"] = best
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump(rec, open(out_path, "w"), indent=1)
print("wrote", out_path)
