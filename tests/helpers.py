"""Shared input generators for the parity tests (same seeded inputs for oracle and HIP)."""
import numpy as np

from oracle import np_oracle as O


def c2_setup(E, D, seed=0, offset=5.0, phase="c2", omega=1.5, z0=0.5, yaw_rate=0.0):
    """SURVEY.md 8d generator: drones on an r=1 circle around a per-env centre U(-offset,offset)^2,
    one Lemniscate per drone.  phase "c2": -(pi/4)(d-1) (EnvGeometric.py:540); "c3": 2 pi d/(D+.25)
    (CBFTestOrd3.py:450)."""
    rng = np.random.default_rng(seed)
    cen = np.zeros((E, D, 3))
    cen[..., :2] = rng.uniform(-offset, offset, size=(E, 1, 2))
    cen[..., 2] = 0.5
    ang = 2 * np.pi * np.arange(D) / D
    xyz = cen.copy()
    xyz[..., 0] += np.sin(ang)
    xyz[..., 1] += np.cos(ang)
    xyz[..., 2] = z0
    P = np.zeros((E, D, 7))
    P[..., 0] = 1.0
    P[..., 1] = omega
    P[..., 2:5] = cen
    P[..., 5] = yaw_rate
    P[..., 6] = -(np.pi / 4) * (np.arange(D) - 1) if phase == "c2" else 2 * np.pi * np.arange(D) / (D + 0.25)
    return xyz, np.zeros((E, D, 3)), P


def oracle_closed_loop(xyz, rpy, P, steps, pyb_freq=100, ctrl_freq=100, physics="dyn", integrator="euler",
                       consts=O.CF2P, record_every=None):
    """Reference loop shape (simulations/EnvGeometric.py:431-473) on the oracle."""
    n = xyz.reshape(-1, 3).shape[0]
    Pf = P.reshape(-1, 7)
    ora = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), consts, pyb_freq, ctrl_freq, physics, integrator)
    obs = ora.step(np.zeros((n, 4)))
    t = 0.0
    out = [obs]
    for k in range(steps):
        pos, vel, acc, yaw, yd = O.lemniscate(t, Pf[:, 0], Pf[:, 1], Pf[:, 2:5], Pf[:, 5], Pf[:, 6])
        rpm = O.geometric_compute(obs, pos, vel, acc, yaw, yd, consts)
        obs = ora.step(rpm)
        t += ora.CTRL_TIMESTEP
        if record_every and (k + 1) % record_every == 0:
            out.append(obs)
    return obs, out


def open_loop_rpm(k, dt, ph, hover=O.CF2P.HOVER_RPM):
    """Deterministic near-hover RPM sequence: 3 % collective wobble + zero-mean differential
    terms, so an uncontrolled drone stays within a few metres over 1000 steps."""
    n = ph.shape[0]
    t = k * dt
    coll = 0.03 * np.sin(2 * np.pi * 1.3 * t + ph[:, 0:1])
    dx = 0.01 * np.cos(2 * np.pi * 5.0 * t) * ph[:, 1]
    dy = 0.01 * np.cos(2 * np.pi * 4.0 * t) * ph[:, 2]
    dz = 0.02 * np.cos(2 * np.pi * 3.0 * t) * ph[:, 3:4]
    s = np.ones((n, 4)) * (1 + coll)
    s[:, 1] += dx
    s[:, 3] -= dx
    s[:, 2] += dy
    s[:, 0] -= dy
    s += dz * np.array([-1, 1, -1, 1])
    return (hover * s).astype(np.float32).astype(np.float64)   # fp32-representable: identical inputs for both sides


def open_loop_setup(n, seed=1, tilt=0.02):
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1, 1, size=(n, 3)) + np.array([0, 0, 1.0])
    rpy = rng.uniform(-tilt, tilt, size=(n, 3))
    ph = np.concatenate([rng.uniform(0, 2 * np.pi, size=(n, 1)), rng.uniform(-1, 1, size=(n, 3))], axis=1)
    return xyz, rpy, ph


def oracle_cbf_closed_loop(xyz, rpy, P, steps, Kcbf, umax, safety_radius, zscale, x_obs, obs_r, pyb_freq=100, ctrl_freq=100,
                           consts=O.CF2P, nominal="geometric", order=2, Fmin=None, Fmax=None, first_rpm=0.0, physics="dyn", record_at=None):
    """simulations/CBFTest.py:303-350 on the oracle, per env: geometric nominal (return_omegas) ->
    u_hat = (force - M G, w_des), xdes = [0,0,yaw, vel, pos] -> ECBF QP (fallback to nominal) ->
    + M G -> ThrustOmega low level -> env.step.  order 3 / nominal "lqr_yank_omega": the loop of
    simulations/CBFTestOrd3.py:306-352 (yank - M G, xdes with G M in slot 3, YankOmega low level, nothing added back).
    Returns (obs [E,D,20], status history [steps,E]); with record_at = (m1, m2, ...) also {m: obs after m steps}."""
    E, D = xyz.shape[0], xyz.shape[1]
    n = E * D
    Pf = P.reshape(-1, 7)
    ora = O.AviaryOracle(xyz.reshape(-1, 3), rpy.reshape(-1, 3), consts, pyb_freq, ctrl_freq, physics=physics, drones_per_env=D)
    ll = O.YankOmegaOracle(n, consts) if order == 3 else O.ThrustOmegaOracle(n, consts)
    Klqr = O.lqr_omega_gain(consts) if nominal == "lqr_omega" else None
    Kyo = O.lqr_yank_omega_gain(consts, 1.0 / ctrl_freq) if nominal == "lqr_yank_omega" else None
    assert (order == 3) == (nominal == "lqr_yank_omega")
    obs = ora.step(np.full((n, 4), float(first_rpm)))     # the order-3 loop integrates thrust from the RPM echo: start it at hover
    t = 0.0
    hist = []
    rec = {}
    for k in range(steps):
        pos, vel, acc, yaw, yd = O.lemniscate(t, Pf[:, 0], Pf[:, 1], Pf[:, 2:5], Pf[:, 5], Pf[:, 6])
        if nominal == "lqr_omega":       # simulations/CBFTest.py:290-293, :339
            unom = O.lqr_omega_compute(obs, pos, vel, yaw, Klqr, consts)
            unom[:, 0] -= consts.M * consts.G
        elif nominal == "lqr_yank_omega":  # simulations/CBFTestOrd3.py:294-297, :341
            unom = O.lqr_yank_omega_compute(obs, pos, vel, yaw, Kyo, consts)
            unom[:, 0] -= consts.M * consts.G
        else:
            force, w_des, _ = O.geometric_compute(obs, pos, vel, acc, yaw, yd, consts, return_omegas=True)
            unom = np.concatenate([(force - consts.M * consts.G)[:, None], w_des], axis=1)
        if order == 3:
            xdes = np.concatenate([np.zeros((n, 2)), yaw[:, None], np.full((n, 1), consts.G * consts.M), vel, pos], axis=1)
            x = O.obs_to_lin_model(obs, 10, consts)
        else:
            xdes = np.concatenate([np.zeros((n, 2)), yaw[:, None], vel, pos], axis=1)
            x = O.obs_to_lin_model(obs, 9)
        usafe = np.zeros((n, 4))
        st = np.zeros(E, dtype=int)
        kw = dict(Fmin=Fmin, Fmax=Fmax) if order == 3 else {}
        for e in range(E):
            sl = slice(e * D, (e + 1) * D)
            usafe[sl], st[e] = O.cbf_filter(x[sl], xdes[sl], unom[sl], order, Kcbf, umax, safety_radius, zscale, consts,
                                            np.array(x_obs) if x_obs is not None else None, obs_r, **kw)
        hist.append(st)
        if order == 2:
            usafe[:, 0] += consts.M * consts.G
        rpm = ll.compute_low_level(usafe, obs, ora.CTRL_TIMESTEP)
        obs = ora.step(rpm)
        t += ora.CTRL_TIMESTEP
        if record_at is not None and k + 1 in record_at:
            rec[k + 1] = obs.reshape(E, D, 20).copy()
    if record_at is not None:
        return obs.reshape(E, D, 20), np.array(hist), rec
    return obs.reshape(E, D, 20), np.array(hist)


def oracle_threads():
    """Host threads for the plain-C oracle (oracle/c_oracle.c): this process's CPU share, at most 16 (a GPU box gives one GPU 16 cores)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def full_size_or_slice(E):
    """The full-size parity tests run the float64 C oracle on every unit of a BASELINE config (up to 0.5 G drone-steps): seconds on the
    16 host cores of a GPU box.  On a host with fewer than 8 cores they keep the first eighth of the envs instead of taking minutes
    (and say so): -> (envs to compare, note)."""
    if oracle_threads() >= 8:
        return E, ""
    return max(64, E // 8), f" [only {oracle_threads()} host cores: the first {max(64, E // 8)} of {E} envs]"
