"""SURVEY 8c G7 -- ORACLE-ONLY fixtures (no reference code involved): 1000-step rollouts of the restated Physics.DYN for
64 drones at 240 Hz under a fixed RPM sequence, float64, explicit Euler / RK4 / Euler + drag.  They pin the oracle against
itself between rounds (a change in oracle/np_oracle.py that moves these numbers must be deliberate) and give the f64 kernels a
long-horizon target that does not depend on the oracle's code being importable.

    python tests/golden/mint_oracle_rollouts.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import np_oracle as O  # noqa: E402

N, STEPS, CHECK = 64, 1000, (1, 100, 500, 1000)


def inputs():
    rng = np.random.default_rng(77)
    xyz = rng.uniform(-2, 2, size=(N, 3)) + np.array([0, 0, 3.0])
    rpy = rng.uniform(-0.3, 0.3, size=(N, 3))
    hover = np.sqrt(O.CF2P.M * O.CF2P.G / (4 * O.CF2P.KF))
    rpm = hover * (1 + 0.03 * rng.standard_normal((8, N, 4)))          # step k applies rpm[k % 8]
    rpm[3, :4] = 1.2 * O.CF2P.MAX_RPM                                   # a few clipped commands
    rpm[5, 4:8] = -50.0
    return xyz, rpy, rpm


def rollout(physics, integrator):
    xyz, rpy, rpm = inputs()
    ora = O.AviaryOracle(xyz, rpy, O.CF2P, 240, 240, physics=physics, integrator=integrator)
    out = {}
    for k in range(1, STEPS + 1):
        obs = ora.step(rpm[(k - 1) % 8])
        if k in CHECK:
            out[k] = obs.copy()
    return out


if __name__ == "__main__":
    xyz, rpy, rpm = inputs()
    data = dict(xyz=xyz, rpy=rpy, rpm=rpm, check=np.array(CHECK), numpy_version=np.array(np.__version__),
                note=np.array("oracle-only fixture (oracle/np_oracle.py AviaryOracle), not minted from the reference"))
    for name, (ph, integ) in dict(euler=("dyn", "euler"), rk4=("dyn", "rk4"), drag=("dyn_drag", "euler")).items():
        r = rollout(ph, integ)
        data[name] = np.stack([r[k] for k in CHECK])
        print(name, np.abs(data[name][-1]).max())
    np.savez_compressed(os.path.join(HERE, "dyn_rollouts_1000.npz"), **data)
