"""TEST TOOLING ONLY: builds and runs tests/emul/simt/simt_cbf.cpp -- the CBF kernels of multidronesim_amd/csrc compiled as host C++ against
the SIMT stand-in (one thread per lane) under AddressSanitizer + UndefinedBehaviorSanitizer.  The product never loads any of this."""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
CSRC = os.path.join(ROOT, "multidronesim_amd", "csrc")
SRC = os.path.join(HERE, "simt_cbf.cpp")
CLANG = os.environ.get("MDS_SIMT_CXX") or "/opt/rocm/lib/llvm/bin/clang++"
EXE = {1: os.path.join(HERE, "simt_filter"), 2: os.path.join(HERE, "simt_rollout"),          # (mode 3, the order-3 rollout, lives in simt_rollout)
       4: os.path.join(HERE, "simt_headline")}


def available():
    return os.path.exists(CLANG) or shutil.which("clang++") is not None


EXE_TSAN = os.path.join(HERE, "simt_rollout_tsan")
_COMMON = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread", "-I", HERE, "-Wno-unknown-attributes", "-Wno-ignored-attributes"]


def _jobs():
    """(executable, compile flags): three ASan + UBSan executables (the filter kernels / the persistent rollout kernels / the headline step and
    rollout kernels) and the ThreadSanitizer build of everything with two wavefronts per workgroup in the persistent kernels."""
    jobs = [(exe, ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", f"-DSIMT_ONLY_MODE={mode}"]) for mode, exe in EXE.items()]
    jobs.append((EXE_TSAN, ["-fsanitize=thread", "-DSIMT_NW=2"]))
    return jobs


def build(force=False, only=None):
    """All four executables compiled side by side (~2 minutes the first time on 4 cores); `only`: restrict to these executables."""
    cxx = CLANG if os.path.exists(CLANG) else shutil.which("clang++")
    deps = [SRC, os.path.abspath(__file__), os.path.join(HERE, "hip", "hip_runtime.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    procs = []
    for exe, flags in _jobs():
        if only is not None and exe not in only:
            continue
        if force or not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
            cmd = [cxx, *_COMMON, *flags, "-o", exe, SRC]
            procs.append((cmd, subprocess.Popen(cmd, cwd=HERE, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("simt build failed: " + " ".join(cmd) + "\n" + out[-4000:])
    return EXE


def build_tsan(force=False):
    """The kernels once more under ThreadSanitizer, two wavefronts per workgroup in the persistent kernels: the LDS hand-offs between lanes and
    between waves (records, bounds, tickets, solver scratch, observation staging) must all sit behind a wave or workgroup barrier -- pthread
    barriers here, which TSan understands; an unsynchronised pair of accesses is a race on the GPU too (or a missing wave-scope fence)."""
    build(force, only=(EXE_TSAN,))
    return EXE_TSAN


def _structs(D, E, cbf_fields, pyb_freq=100, ctrl_freq=100):
    from multidronesim_amd import _capi as capi
    lib = capi.load_library()
    cfg, gains, p = capi.MdsConfig(), capi.MdsGeometricGains(), capi.MdsCbfParams()
    capi.check(lib.mds_default_config(capi.MDS_CF2P, C.byref(cfg)), "mds_default_config")
    capi.check(lib.mds_default_geometric_gains(C.byref(gains)), "mds_default_geometric_gains")
    cfg.num_envs, cfg.num_drones, cfg.pyb_freq, cfg.ctrl_freq, cfg.dtype = E, D, pyb_freq, ctrl_freq, capi.MDS_F64
    for k, v in cbf_fields.items():
        if isinstance(v, (list, tuple, np.ndarray)):
            arr = getattr(p, k)
            for j, x in enumerate(v):
                arr[j] = float(x)
        else:
            setattr(p, k, v)
    return cfg, gains, p


DTYPE_CODE = {"float32": 0, "float64": 1, "float16": 2, "float32c": 3}


def run(mode, dtype, E, D, n_steps, cbf_fields, obstacles, arrays, timeout=900, tsan=False, nominal=0):
    """-> the bytes of out.bin.  obstacles: [n_obs, 4] (xyz, r).  arrays: the mode's float64 arrays, concatenated in order."""
    exe = build_tsan() if tsan else build()[2 if mode == 3 else mode]
    cfg, gains, p = _structs(D, E, cbf_fields)
    ob = np.zeros(64)
    ob[:np.asarray(obstacles).size] = np.asarray(obstacles, dtype=np.float64).reshape(-1)
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([mode, DTYPE_CODE[dtype], E, D, n_steps, nominal], dtype=np.int32).tobytes())
            f.write(bytes(cfg)); f.write(bytes(gains)); f.write(bytes(p)); f.write(ob.tobytes())
            for a in arrays:
                f.write(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes())
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
                   TSAN_OPTIONS="halt_on_error=1:exitcode=66:report_signal_unsafe=0")
        r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=timeout, env=env)
        if r.returncode != 0:
            raise RuntimeError(f"simt_cbf (mode {mode}, {dtype}, E={E}, D={D}) exited {r.returncode}:\n{r.stderr[-6000:]}")
        return open(fout, "rb").read(), r.stderr


def filter_(dtype, obs, xdes, unom, cbf_fields, obstacles, tsan=False):
    E, D = obs.shape[0], obs.shape[1]
    raw, err = run(1, dtype, E, D, 0, cbf_fields, obstacles, [obs, xdes, unom], tsan=tsan)
    n = E * D
    us = np.frombuffer(raw[:n * 4 * 8], dtype=np.float64).reshape(E, D, 4)
    st = np.frombuffer(raw[n * 32:n * 32 + 4 * E], dtype=np.int32)
    it = np.frombuffer(raw[n * 32 + 4 * E:n * 32 + 8 * E], dtype=np.int32)
    return us, st, it, err


def rollout(dtype, t0, P, state13, steps, cbf_fields, obstacles, tsan=False):
    E, D = P.shape[0], P.shape[1]
    raw, err = run(2, dtype, E, D, steps, cbf_fields, obstacles, [np.array([t0]), P, state13], tsan=tsan)
    n = E * D
    obs = np.frombuffer(raw[:n * 20 * 8], dtype=np.float64).reshape(E, D, 20)
    o = n * 160
    slog = np.frombuffer(raw[o:o + 4 * steps * E], dtype=np.int32).reshape(steps, E)
    it = np.frombuffer(raw[o + 4 * steps * E:o + 4 * steps * E + 4 * E], dtype=np.int32)
    return obs, slog, it, err


def rollout_o3(dtype, t0, K, P, state13, rpm_echo, steps, cbf_fields, obstacles, tsan=False):
    """k_cbf_rollout_o3: K [4,10] the LQR-yank-omega gain, rpm_echo [E,D,4] the current observation's clipped RPM."""
    E, D = P.shape[0], P.shape[1]
    raw, err = run(3, dtype, E, D, steps, cbf_fields, obstacles, [np.array([t0]), np.asarray(K).reshape(-1), P, state13, rpm_echo], tsan=tsan)
    n = E * D
    obs = np.frombuffer(raw[:n * 20 * 8], dtype=np.float64).reshape(E, D, 20)
    o = n * 160
    slog = np.frombuffer(raw[o:o + 4 * steps * E], dtype=np.int32).reshape(steps, E)
    it = np.frombuffer(raw[o + 4 * steps * E:o + 4 * steps * E + 4 * E], dtype=np.int32)
    return obs, slog, it, err


def headline(dtype, form, t0, P, state13, steps, actions=None, tsan=False):
    """Mode 4: form 0 k_step_geometric per step, 1 k_rollout_geometric (log ring), 2 k_step on an action table [3, E, D, 4], 3 k_rollout_step,
    4 k_rollout_geometric rewriting one [n, 20] array every step.
    -> (obs [E,D,20], state13 [E,D,13] world, action_out [E,D,4] (form 0), stderr)"""
    E, D = P.shape[0], P.shape[1]
    arrays = [np.array([t0]), P, state13] + ([actions] if form in (2, 3) else [])
    raw, err = run(4, dtype, E, D, steps, {"order": 2}, np.zeros((0, 4)), arrays, tsan=tsan, nominal=form)
    n = E * D
    obs = np.frombuffer(raw[:n * 160], dtype=np.float64).reshape(E, D, 20)
    st = np.frombuffer(raw[n * 160:n * 264], dtype=np.float64).reshape(E, D, 13)
    act = np.frombuffer(raw[n * 264:n * 296], dtype=np.float64).reshape(E, D, 4)
    return obs, st, act, err
