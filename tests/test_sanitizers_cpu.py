"""UndefinedBehaviorSanitizer run of the device arithmetic (the g++ build of csrc/mds_math.hpp, tests/emul):
GPU sanitizers are not available on the pool, so the per-drone math is exercised under UBSan on the CPU --
fused controller + physics (Euler, RK4, drag, substeps), saturating inputs included, and the CompareModels templates on degenerate inputs.  Runs in a subprocess."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import ctypes as C, sys, numpy as np
sys.path.insert(0, %(root)r)
from tests.emul import emul as E
E.SO = %(so)r
E._lib = None
lib = C.CDLL(E.SO)
for sfx in ("f32", "f64"):
    getattr(lib, "emul_create_" + sfx).restype = C.c_void_p
E._lib = lib
rng = np.random.default_rng(0)
for dt in ("f32", "f64"):
    for kw in (dict(), dict(integrator=1), dict(physics=1), dict(pyb_freq=240, ctrl_freq=48)):
        n = 64
        # (1) open loop with RPM far outside [0, MAX_RPM] from arbitrary attitudes (upside down included): clipping paths
        em = E.Emul(dt, num_envs=n, **kw)
        st = np.zeros((n, 13)); st[:, 0:3] = rng.normal(size=(n, 3)) * 3
        q = rng.normal(size=(n, 4)); st[:, 3:7] = q / np.linalg.norm(q, axis=1, keepdims=True)
        st[:, 7:10] = rng.normal(size=(n, 3)) * 4; st[:, 10:13] = rng.normal(size=(n, 3)) * 20
        em.set_state(st)
        for k in range(50):
            obs = em.step(rng.uniform(-1e4, 5e4, size=(n, 4)))
        assert np.isfinite(obs).all(), dt
        # (2) closed loop from large but recoverable errors (tilt clamp and motor clips active).  Twice these errors make the
        #     reference's own loop diverge (explicit Euler at 100 Hz: 1e7 rad/s in float64 too), so that is not tested.
        em = E.Emul(dt, num_envs=n, **kw)
        from scipy.spatial.transform import Rotation
        rng = np.random.default_rng(0)
        sc = 0.1 if kw.get("ctrl_freq") == 48 else 1.0       # at 48 Hz control the reference's gains (tuned for 100 Hz) diverge from large errors, in float64 too
        st = np.zeros((n, 13)); st[:, 0:3] = rng.normal(size=(n, 3)) * 1.5 * sc
        st[:, 3:7] = Rotation.from_euler("xyz", rng.uniform(-0.5, 0.5, size=(n, 3)) * sc).as_quat()
        st[:, 7:10] = rng.normal(size=(n, 3)) * sc; st[:, 10:13] = rng.normal(size=(n, 3)) * 1.5 * sc
        em.set_state(st)
        P = np.zeros((n, 7)); P[:, 0] = 1; P[:, 1] = rng.uniform(0.1, 2, n); P[:, 5] = rng.normal(size=n) * 0.3; P[:, 6] = rng.uniform(-7, 7, n) * sc
        em.set_lemniscate(P)
        t = 0.0
        for k in range(300):
            obs, act = em.step_geometric(t); t += 1.0 / em.cfg.ctrl_freq
        assert np.isfinite(obs).all(), dt
# (3) the CompareModels templates: arbitrary (also zero-norm and huge) quaternions, RPM far outside the clip range, rotation
#     matrices at the branch boundaries of the matrix -> quaternion conversion, angles far outside [-pi, pi]
for dt in ("f32", "f64"):
    em = E.Emul(dt)
    n = 256
    obs = rng.normal(size=(n, 20)) * 5
    obs[:8, 3:7] = 0.0
    obs[8:16, 3:7] *= 1e15
    obs[:, 16:20] = rng.uniform(-1e5, 1e5, size=(n, 4))
    A, B = rng.normal(size=(12, 12)), rng.normal(size=(12, 4))
    a, b, c = em.compare_models(obs, A, B, 0.26, 0.027, [1.05, 1.05, 2.05], 9.8)
    assert np.isfinite(c).all() and np.isfinite(a[16:]).all() and np.isfinite(b[16:]).all(), dt
    R = E.rpy_to_rot(rng.uniform(-50, 50, size=(64, 3)), dt)
    assert np.isfinite(R).all() and np.abs(np.linalg.det(R) - 1).max() < 1e-3, dt
    mats = np.concatenate([R, np.eye(3)[None], np.diag([1.0, -1, -1])[None], np.diag([-1.0, 1, -1])[None], np.diag([-1.0, -1, 1])[None],
                           np.zeros((1, 3, 3))])
    q = E.rot_to_quat(mats, dt)
    assert np.isfinite(q[:-1]).all() and np.abs(np.linalg.norm(q[:-1], axis=1) - 1).max() < 1e-5, dt
print("UBSAN_OK")
'''


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_device_math_under_ubsan(tmp_path):
    so = str(tmp_path / "libmds_emul_ubsan.so")
    src = os.path.join(ROOT, "tests", "emul", "mds_emul.cpp")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-mfma", "-ffp-contract=fast",
                           "-fsanitize=undefined", "-fno-sanitize-recover=all", "-o", so, src])
    env = dict(os.environ, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT, "so": so}], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "UBSAN_OK" in out.stdout, out.stderr[-3000:]
    assert "runtime error" not in out.stderr, out.stderr[-3000:]
