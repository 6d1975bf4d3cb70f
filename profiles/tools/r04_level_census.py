"""C4 `level` (SURVEY 8d's spheres): active-set iterations of feasible against infeasible env-steps over the bench window (steps 20..220), step-by-step loop.
Run on the GPU box from the repo root:  python3 profiles/tools/r04_level_census.py [scene]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from multidronesim_amd.envs.CtrlAviary import CtrlAviary, DroneModel, Physics

scene = sys.argv[1] if len(sys.argv) > 1 else "level"
E, D = 16384, 16
env, tracker = bench.c4_make(CtrlAviary, DroneModel, Physics, E, D, 1000, "float32", 0)
c4_obs, c4_r = bench.c4_spheres(scene)
env.step(torch.zeros((E, D, 4), dtype=env.dtype, device=env.device))
t, dt = 0.0, env.CTRL_TIMESTEP
n0 = n1 = 0
i0 = i1 = 0
h1 = torch.zeros(64, dtype=torch.long)
for k in range(220):
    _, st = env.step_cbf_geometric(t, tracker, c4_obs, c4_r)
    t += dt
    if k >= 20:
        it = tracker.cbf.last_iterations()
        bad = st != 0
        n1 += int(bad.sum()); n0 += int((~bad).sum())
        i1 += int(it[bad].sum()); i0 += int(it[~bad].sum())
        h1 += torch.bincount(it[bad].clamp(max=63).long().cpu(), minlength=64)
print(f"{scene}: env-steps feasible {n0} (mean iterations {i0 / max(n0, 1):.3f}), infeasible {n1} (mean iterations {i1 / max(n1, 1):.3f}); share of all iterations spent on infeasible env-steps {i1 / max(i0 + i1, 1):.3f}")
print("iterations histogram of the infeasible env-steps:", {k: int(v) for k, v in enumerate(h1.tolist()) if v})
