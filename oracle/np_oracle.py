"""TEST INFRASTRUCTURE ONLY -- float64 NumPy restatement of the reference hot path.

Every function cites the reference lines it follows (paths relative to the
reference checkout).  ``[UPSTREAM]`` marks behaviour of gym-pybullet-drones /
pybullet, which are NOT in the reference tree and not installable here; those
functions follow the specification recorded in SURVEY.md section 3.4 and their
parity against real PyBullet is UNPINNED.  Everything restated from in-tree files
is pinned by ``tests/golden/*.npz`` (minted from the reference by
``tests/golden/mint_golden.py``).

All functions are batched over leading axes and compute in float64.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

# --------------------------------------------------------------------------------------
# drone constants  ([UPSTREAM] cf2p.urdf / cf2x.urdf; cross-checked against
# utils/graph_fedce.py:9 and model/dynamics.py:38-39)
# --------------------------------------------------------------------------------------


@dataclass
class DroneConsts:
    M: float = 0.027
    L: float = 0.0397
    KF: float = 3.16e-10
    KM: float = 7.94e-12
    J: tuple = (2.3951e-5, 2.3951e-5, 3.2347e-5)  # cf2p
    G: float = 9.8
    THRUST2WEIGHT: float = 2.25
    DRAG: tuple = (9.1785e-7, 9.1785e-7, 10.311e-7)
    MODEL: str = "cf2p"
    # [UPSTREAM] cf2x.urdf / cf2p.urdf <properties>: ground effect and downwash
    GND_EFF_COEFF: float = 11.36859
    PROP_RADIUS: float = 2.31348e-2
    DW_COEFF_1: float = 2267.18
    DW_COEFF_2: float = .16
    DW_COEFF_3: float = -.11
    GRAVITY: float = field(init=False)
    HOVER_RPM: float = field(init=False)
    MAX_RPM: float = field(init=False)
    MAX_THRUST: float = field(init=False)
    MAX_XY_TORQUE: float = field(init=False)
    MAX_Z_TORQUE: float = field(init=False)

    def __post_init__(self):
        # [UPSTREAM] BaseAviary.__init__
        self.GRAVITY = self.G * self.M
        self.HOVER_RPM = np.sqrt(self.GRAVITY / (4 * self.KF))
        self.MAX_RPM = np.sqrt((self.THRUST2WEIGHT * self.GRAVITY) / (4 * self.KF))
        self.MAX_THRUST = 4 * self.KF * self.MAX_RPM ** 2
        if self.MODEL == "cf2x":
            self.MAX_XY_TORQUE = (2 * self.L * self.KF * self.MAX_RPM ** 2) / np.sqrt(2)
        else:
            self.MAX_XY_TORQUE = self.L * self.KF * self.MAX_RPM ** 2
        self.MAX_Z_TORQUE = 2 * self.KM * self.MAX_RPM ** 2
        self.GND_EFF_H_CLIP = 0.25 * self.PROP_RADIUS * np.sqrt((15 * self.MAX_RPM ** 2 * self.KF * self.GND_EFF_COEFF) / self.MAX_THRUST)

    def prop_offsets(self):
        """Propeller link origins in the body frame ([UPSTREAM] cf2p.urdf: arms along +-x / +-y at L; cf2x.urdf: (+-0.028, +-0.028))."""
        if self.MODEL == "cf2x":
            return np.array([[0.028, -0.028], [-0.028, -0.028], [-0.028, 0.028], [0.028, 0.028]])
        return np.array([[self.L, 0.0], [0.0, self.L], [-self.L, 0.0], [0.0, -self.L]])


CF2P = DroneConsts()
CF2X = DroneConsts(J=(1.4e-5, 1.4e-5, 2.17e-5), MODEL="cf2x")

# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------


def cross(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def norm(a):
    return np.sqrt(np.sum(np.asarray(a, dtype=np.float64) ** 2, axis=-1))


def matvec(R, v):
    return np.einsum("...ij,...j->...i", R, v)


def matTvec(R, v):
    return np.einsum("...ji,...j->...i", R, v)


def quat_to_rotmat_scipy(q):
    """scipy ``Rotation.from_quat(q).as_matrix()`` (normalising, xyzw) as used by
    utils/model_conversions.py:110."""
    q = np.asarray(q, dtype=np.float64)
    q = q / norm(q)[..., None]
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    x2, y2, z2, w2 = x * x, y * y, z * z, w * w
    xy, zw, xz, yw, yz, xw = x * y, z * w, x * z, y * w, y * z, x * w
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = x2 - y2 - z2 + w2
    R[..., 1, 0] = 2 * (xy + zw)
    R[..., 2, 0] = 2 * (xz - yw)
    R[..., 0, 1] = 2 * (xy - zw)
    R[..., 1, 1] = -x2 + y2 - z2 + w2
    R[..., 2, 1] = 2 * (yz + xw)
    R[..., 0, 2] = 2 * (xz + yw)
    R[..., 1, 2] = 2 * (yz - xw)
    R[..., 2, 2] = -x2 - y2 + z2 + w2
    return R


def quat_to_rotmat_bullet(q):
    """[UPSTREAM] ``p.getMatrixFromQuaternion`` (btMatrix3x3::setRotation: s = 2/|q|^2)."""
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    d = x * x + y * y + z * z + w * w
    s = 2.0 / d
    xs, ys, zs = x * s, y * s, z * s
    wx, wy, wz = w * xs, w * ys, w * zs
    xx, xy, xz = x * xs, x * ys, x * zs
    yy, yz, zz = y * ys, y * zs, z * zs
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1.0 - (yy + zz)
    R[..., 0, 1] = xy - wz
    R[..., 0, 2] = xz + wy
    R[..., 1, 0] = xy + wz
    R[..., 1, 1] = 1.0 - (xx + zz)
    R[..., 1, 2] = yz - wx
    R[..., 2, 0] = xz - wy
    R[..., 2, 1] = yz + wx
    R[..., 2, 2] = 1.0 - (xx + yy)
    return R


def quat_from_euler_bullet(rpy):
    """[UPSTREAM] ``p.getQuaternionFromEuler`` (btQuaternion::setEulerZYX), xyzw."""
    rpy = np.asarray(rpy, dtype=np.float64)
    hr, hp, hy = rpy[..., 0] * 0.5, rpy[..., 1] * 0.5, rpy[..., 2] * 0.5
    cr, sr, cp, sp, cy, sy = np.cos(hr), np.sin(hr), np.cos(hp), np.sin(hp), np.cos(hy), np.sin(hy)
    return np.stack([sr * cp * cy - cr * sp * sy,
                     cr * sp * cy + sr * cp * sy,
                     cr * cp * sy - sr * sp * cy,
                     cr * cp * cy + sr * sp * sy], axis=-1)


def euler_from_quat_bullet(q):
    """[UPSTREAM] ``p.getEulerFromQuaternion``: ZYX with the
    +-0.99999 gimbal branches."""
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    sqx, sqy, sqz, squ = x * x, y * y, z * z, w * w
    sarg = -2.0 * (x * z - w * y)
    roll = np.arctan2(2 * (y * z + w * x), squ - sqx - sqy + sqz)
    pitch = np.arcsin(np.clip(sarg, -1.0, 1.0))
    yaw = np.arctan2(2 * (x * y + w * z), squ + sqx - sqy - sqz)
    lo = sarg <= -0.99999
    hi = sarg >= 0.99999
    roll = np.where(lo | hi, 0.0, roll)
    pitch = np.where(lo, -0.5 * np.pi, np.where(hi, 0.5 * np.pi, pitch))
    yaw = np.where(lo, 2 * np.arctan2(x, -y), np.where(hi, 2 * np.arctan2(-x, y), yaw))
    return np.stack([roll, pitch, yaw], axis=-1)


# --------------------------------------------------------------------------------------
# a1-a4: [UPSTREAM] BaseAviary.step / _dynamics / _integrateQ / _drag  
# --------------------------------------------------------------------------------------


def rotor_wrench(rpm, c: DroneConsts, extra_prop_force=None):
    """[UPSTREAM] _dynamics: total thrust (body z) and body torques from 4 RPM.  ``extra_prop_force`` [.,4] (ground effect)
    is added to the per-propeller thrusts before the mixing, not to the KM yaw torque."""
    rpm = np.asarray(rpm, dtype=np.float64)
    forces = rpm ** 2 * c.KF
    if extra_prop_force is not None:
        forces = forces + extra_prop_force
    thrust = forces[..., 0] + forces[..., 1] + forces[..., 2] + forces[..., 3]
    zt = rpm ** 2 * c.KM
    tz = -zt[..., 0] + zt[..., 1] - zt[..., 2] + zt[..., 3]
    if c.MODEL == "cf2x":
        tx = (forces[..., 0] + forces[..., 1] - forces[..., 2] - forces[..., 3]) * (c.L / np.sqrt(2))
        ty = (-forces[..., 0] + forces[..., 1] + forces[..., 2] - forces[..., 3]) * (c.L / np.sqrt(2))
    else:  # cf2p
        tx = (forces[..., 1] - forces[..., 3]) * c.L
        ty = (-forces[..., 0] + forces[..., 2]) * c.L
    return thrust, np.stack([tx, ty, tz], axis=-1)


def integrate_q(quat, omega, dt):
    """[UPSTREAM] BaseAviary._integrateQ: exact exponential for constant body rate."""
    quat = np.asarray(quat, dtype=np.float64)
    omega = np.asarray(omega, dtype=np.float64)
    on = norm(omega)
    p, q, r = omega[..., 0], omega[..., 1], omega[..., 2]
    still = np.isclose(on, 0.0)
    on_safe = np.where(still, 1.0, on)
    theta = on_safe * dt / 2
    ct = np.cos(theta)
    k = (2.0 / on_safe) * np.sin(theta) * 0.5
    x, y, z, w = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    # lambda_ = .5*[[0,r,-q,p],[-r,0,p,q],[q,-p,0,r],[-p,-q,-r,0]]
    nx = ct * x + k * (r * y - q * z + p * w)
    ny = ct * y + k * (-r * x + p * z + q * w)
    nz = ct * z + k * (q * x - p * y + r * w)
    nw = ct * w + k * (-p * x - q * y - r * z)
    out = np.stack([nx, ny, nz, nw], axis=-1)
    return np.where(still[..., None], quat, out)


def dyn_derivative(pos, quat, vel, rates, rpm, c: DroneConsts, drag_rpm=None):
    """Continuous-time derivative of the 13-state used by the RK4 integrator
    (same force/torque model as [UPSTREAM] _dynamics; qdot = Lambda(omega) q)."""
    R = quat_to_rotmat_bullet(quat)
    thrust, torques = rotor_wrench(rpm, c)
    force_w = R[..., :, 2] * thrust[..., None]
    force_w[..., 2] -= c.GRAVITY
    if drag_rpm is not None:
        force_w = force_w + drag_force_world(vel, drag_rpm, c)
    J = np.asarray(c.J)
    torques = torques - cross(rates, J * rates)
    rates_dot = torques / J
    acc = force_w / c.M
    p, q, r = rates[..., 0], rates[..., 1], rates[..., 2]
    x, y, z, w = quat[..., 0], quat[..., 1], quat[..., 2], quat[..., 3]
    qdot = 0.5 * np.stack([r * y - q * z + p * w,
                           -r * x + p * z + q * w,
                           q * x - p * y + r * w,
                           -p * x - q * y - r * z], axis=-1)
    return vel, qdot, acc, rates_dot


def drag_force_world(vel, rpm_prev, c: DroneConsts):
    """[UPSTREAM] BaseAviary._drag: body force R^T(-c * sum(2 pi rpm/60) * v_world)
    applied in the link frame, i.e. world force = -c (.) sum(2 pi rpm / 60) (.) v_world.
    ``rpm_prev`` is the PREVIOUS step's clipped action."""
    rpm_prev = np.asarray(rpm_prev, dtype=np.float64)
    s = np.sum(2 * np.pi * rpm_prev / 60, axis=-1)
    return -np.asarray(c.DRAG) * s[..., None] * np.asarray(vel, dtype=np.float64)


def ground_effect_forces(pos, quat, rpm, c: DroneConsts):
    """[UPSTREAM] BaseAviary._groundEffect: per-propeller extra thrust KF rpm^2 GND_EFF_COEFF (PROP_RADIUS / (4 h_k))^2 with the
    propeller heights h_k clipped from below at GND_EFF_H_CLIP, applied (in the propeller link frames, i.e. along body z) only
    while |roll|, |pitch| < pi/2.  Upstream hands these to Bullet (PYB_GND); here they enter the DYN wrench -- spec-level."""
    R = quat_to_rotmat_bullet(quat)
    off = c.prop_offsets()                                             # [4,2] body x, y
    h = pos[..., 2:3] + R[..., 2, 0:1] * off[:, 0] + R[..., 2, 1:2] * off[:, 1]
    h = np.clip(h, c.GND_EFF_H_CLIP, np.inf)
    g = np.asarray(rpm, dtype=np.float64) ** 2 * c.KF * c.GND_EFF_COEFF * (c.PROP_RADIUS / (4 * h)) ** 2
    rpy = euler_from_quat_bullet(quat)
    ok = (np.abs(rpy[..., 0]) < np.pi / 2) & (np.abs(rpy[..., 1]) < np.pi / 2)
    return np.where(ok[..., None], g, 0.0)


def downwash_force(pos, drones_per_env, c: DroneConsts):
    """[UPSTREAM] BaseAviary._downwash: body-z force on drone i from every drone j of the same env above it,
    -DW_COEFF_1 (PROP_RADIUS / (4 dz))^2 exp(-0.5 (dxy / (DW_COEFF_2 dz + DW_COEFF_3))^2), if dz > 0 and dxy < 10 m."""
    n = pos.shape[0]
    D = drones_per_env
    p = pos.reshape(n // D, D, 3)
    dz = p[:, None, :, 2] - p[:, :, None, 2]                           # [E, i, j] = z_j - z_i
    dxy = np.linalg.norm(p[:, None, :, 0:2] - p[:, :, None, 0:2], axis=-1)
    act = (dz > 0) & (dxy < 10)
    dzs = np.where(act, dz, 1.0)
    alpha = c.DW_COEFF_1 * (c.PROP_RADIUS / (4 * dzs)) ** 2
    beta = c.DW_COEFF_2 * dzs + c.DW_COEFF_3
    f = np.where(act, -alpha * np.exp(-0.5 * (dxy / beta) ** 2), 0.0)
    return f.sum(axis=2).reshape(n)


def dyn_step_euler(pos, quat, vel, rates, rpm, dt, c: DroneConsts, drag_rpm=None, wind=None, extra_prop_force=None, extra_body_z=None):
    """[UPSTREAM] BaseAviary._dynamics (Physics.DYN): explicit Euler on (v, omega),
    then p with the NEW v, q with the NEW omega.  ``drag_rpm`` (build extension
    DYN_DRAG) adds the _drag force computed from the previous clipped action."""
    R = quat_to_rotmat_bullet(quat)
    thrust, torques = rotor_wrench(rpm, c, extra_prop_force)
    if extra_body_z is not None:
        thrust = thrust + extra_body_z
    force_w = R[..., :, 2] * thrust[..., None]          # R @ [0,0,thrust]
    force_w[..., 2] -= c.GRAVITY
    if drag_rpm is not None:
        force_w = force_w + drag_force_world(vel, drag_rpm, c)
    if wind is not None:                                # EnvGeometric.py:463-467 world-frame external force
        force_w = force_w + np.asarray(wind, dtype=np.float64)
    J = np.asarray(c.J)
    torques = torques - cross(rates, J * rates)
    rates_dot = torques / J                             # J_INV @ torques (diagonal J)
    acc = force_w / c.M
    vel = vel + dt * acc
    rates = rates + dt * rates_dot
    pos = pos + dt * vel
    quat = integrate_q(quat, rates, dt)
    ang_v_world = matvec(R, rates)                      # resetBaseVelocity(..., R_old @ rpy_rates)
    return pos, quat, vel, rates, ang_v_world


def dyn_step_rk4(pos, quat, vel, rates, rpm, dt, c: DroneConsts, drag_rpm=None):
    """Classical RK4 on the 13-state (north_star integrator; not in upstream).
    Quaternion re-normalised after the step."""
    def f(s):
        return dyn_derivative(s[0], s[1], s[2], s[3], rpm, c, drag_rpm)

    s0 = (pos, quat, vel, rates)
    k1 = f(s0)
    k2 = f(tuple(a + 0.5 * dt * k for a, k in zip(s0, k1)))
    k3 = f(tuple(a + 0.5 * dt * k for a, k in zip(s0, k2)))
    k4 = f(tuple(a + dt * k for a, k in zip(s0, k3)))
    out = [a + (dt / 6.0) * (b1 + 2 * b2 + 2 * b3 + b4) for a, b1, b2, b3, b4 in zip(s0, k1, k2, k3, k4)]
    out[1] = out[1] / norm(out[1])[..., None]
    ang_v_world = matvec(quat_to_rotmat_bullet(out[1]), out[3])
    return out[0], out[1], out[2], out[3], ang_v_world


class AviaryOracle:
    """[UPSTREAM] CtrlAviary / BaseAviary state machine (step, obs packing), batched
    over n drones.  physics in {"dyn", "dyn_drag", "dyn_gnd", "dyn_dw", "dyn_gnd_drag_dw"}, integrator in {"euler", "rk4"}
    (the ground-effect / downwash modes: Euler only; ``drones_per_env`` groups the drones that see each other's downwash)."""

    def __init__(self, init_xyzs, init_rpys, consts: DroneConsts = CF2P, pyb_freq=240, ctrl_freq=240,
                 physics="dyn", integrator="euler", drones_per_env=None):
        if pyb_freq % ctrl_freq != 0:
            raise ValueError("pyb_freq must be a multiple of ctrl_freq")  # [UPSTREAM] BaseAviary.__init__
        self.c = consts
        self.PYB_FREQ, self.CTRL_FREQ = pyb_freq, ctrl_freq
        self.PYB_STEPS_PER_CTRL = pyb_freq // ctrl_freq
        self.PYB_TIMESTEP = 1.0 / pyb_freq
        self.CTRL_TIMESTEP = 1.0 / ctrl_freq
        self.physics, self.integrator = physics, integrator
        self.drones_per_env = drones_per_env
        self.wind = None
        self.init_xyzs = np.array(init_xyzs, dtype=np.float64).reshape(-1, 3)
        self.init_rpys = np.array(init_rpys, dtype=np.float64).reshape(-1, 3)
        self.reset()

    def reset(self):
        """[UPSTREAM] _housekeeping + _updateAndStoreKinematicInformation."""
        n = self.init_xyzs.shape[0]
        self.pos = self.init_xyzs.copy()
        self.quat = quat_from_euler_bullet(self.init_rpys)
        self.vel = np.zeros((n, 3))
        self.rates = np.zeros((n, 3))
        self.ang_v = np.zeros((n, 3))
        self.last_clipped_action = np.zeros((n, 4))
        return self.obs()

    def obs(self):
        """[UPSTREAM] _getDroneStateVector: pos3 | quat4 xyzw | rpy3 | vel3 | ang_v3 | last_clipped_action4."""
        return np.concatenate([self.pos, self.quat, euler_from_quat_bullet(self.quat), self.vel, self.ang_v,
                               self.last_clipped_action], axis=-1)

    def step(self, action):
        clipped = np.clip(np.asarray(action, dtype=np.float64).reshape(-1, 4), 0, self.c.MAX_RPM)
        stepf = dyn_step_euler if self.integrator == "euler" else dyn_step_rk4
        for _ in range(self.PYB_STEPS_PER_CTRL):
            drag_rpm = self.last_clipped_action if self.physics in ("dyn_drag", "dyn_gnd_drag_dw") else None
            kw = {"wind": self.wind} if (self.wind is not None and self.integrator == "euler") else {}
            if self.physics in ("dyn_gnd", "dyn_gnd_drag_dw"):       # kinematic info as of the start of the substep ([UPSTREAM] step())
                kw["extra_prop_force"] = ground_effect_forces(self.pos, self.quat, clipped, self.c)
            if self.physics in ("dyn_dw", "dyn_gnd_drag_dw"):
                kw["extra_body_z"] = downwash_force(self.pos, self.drones_per_env or self.pos.shape[0], self.c)
            self.pos, self.quat, self.vel, self.rates, self.ang_v = stepf(
                self.pos, self.quat, self.vel, self.rates, clipped, self.PYB_TIMESTEP, self.c, drag_rpm, **kw)
            self.last_clipped_action = clipped
        return self.obs()


# --------------------------------------------------------------------------------------
# a10: trajectories/Lemniscate.py:32-63
# --------------------------------------------------------------------------------------


def lemniscate(t, a, omega, center, yaw_rate, phase_shift):
    """Lemniscate.__call__ -> (pos, vel, acc, yaw, yaw_rate_out), batched over params."""
    a = np.asarray(a, dtype=np.float64)
    omega = np.asarray(omega, dtype=np.float64)
    center = np.asarray(center, dtype=np.float64)
    yaw_rate = np.asarray(yaw_rate, dtype=np.float64)
    th = t * omega + phase_shift
    s, c = np.sin(th), np.cos(th)
    z = np.zeros_like(th)
    pos = np.stack([center[..., 0] + (a * s * c) / (1 + s ** 2),
                    center[..., 1] + (a * c) / (1 + s ** 2),
                    center[..., 2] + z], axis=-1)
    vel = np.stack([-a * omega * (s ** 4 + s ** 2 + (s ** 2 - 1) * c ** 2) / (s ** 2 + 1) ** 2,
                    -a * omega * s * (s ** 2 + 2 * c ** 2 + 1) / (s ** 2 + 1) ** 2,
                    z], axis=-1)
    acc = np.stack([4 * a * omega ** 2 * np.sin(2 * th) * (3 * np.cos(2 * th) + 7) / (np.cos(2 * th) - 3) ** 3,
                    a * omega ** 2 * c * (44 * np.cos(2 * th) + np.cos(4 * th) - 21) / (np.cos(2 * th) - 3) ** 3,
                    z], axis=-1)
    yaw = np.pi * np.sin(yaw_rate * t)
    yaw_dot = np.pi * yaw_rate * np.cos(yaw_rate * t)
    return pos, vel, acc, yaw, yaw_dot


# --------------------------------------------------------------------------------------
# a9: utils/model_conversions.py:69-103
# --------------------------------------------------------------------------------------


def _mixer(c: DroneConsts):
    r = c.KM / c.KF
    return np.array([[1.0, 1.0, 1.0, 1.0],
                     [0.0, c.L, 0.0, -c.L],
                     [-c.L, 0.0, c.L, 0.0],
                     [-r, r, -r, r]])


def action_to_input(action, c: DroneConsts, cap_rpm=True):
    """model_conversions.py:69-83."""
    action = np.asarray(action, dtype=np.float64)
    if cap_rpm:
        action = np.clip(action, 0, c.MAX_RPM)
    return np.einsum("ij,...j->...i", _mixer(c), c.KF * action ** 2)


def input_to_action(u, c: DroneConsts):
    """model_conversions.py:85-103 (MAX_THRUST used as a per-motor clip, as in the reference)."""
    u = np.array(u, dtype=np.float64)
    u[..., 0] = np.clip(u[..., 0], 0, None)
    thrusts = np.einsum("ij,...j->...i", np.linalg.inv(_mixer(c)), u)
    thrusts = np.clip(thrusts, 9440.3 ** 2 * c.KF, c.MAX_THRUST)
    return np.sqrt(thrusts / c.KF)


# --------------------------------------------------------------------------------------
# a7, a8: utils/model_conversions.py:105-114, control/geometric.py:59-115
# --------------------------------------------------------------------------------------

GEO_GAINS = dict(Kp=(2.25, 2.25, 2.25), Kv=(3.5, 3.5, 3.5), KR=(125.0, 125.0, 125.0), Kw=(10.0, 10.0, 10.0),
                 g=9.81, max_tilt=40 * np.pi / 180)  # control/geometric.py:14-23


def obs_to_geo_model(obs):
    """model_conversions.py:105-114 -> (p, R, v_world, w)."""
    obs = np.asarray(obs, dtype=np.float64)
    return obs[..., 0:3], quat_to_rotmat_scipy(obs[..., 3:7]), obs[..., 10:13], obs[..., 13:16]


def geometric_compute(obs, p_des, v_des, a_des, yaw_des, yawrate_des, c: DroneConsts = CF2P, gains=None,
                      return_omegas=False):
    """GeometricControl.compute (control/geometric.py:59-115), quirks kept:
    g = 9.81 (:20); obs[13:16] used as body rate (:63); R_des.transpose(0,1) is a no-op on
    ndarrays so w_des_hat = R_des @ R_dot_des (:102); f_des_dot uses Kp and body v (:96)."""
    gn = dict(GEO_GAINS)
    if gains:
        gn.update(gains)
    Kp, Kv, KR, Kw = (np.asarray(gn[k], dtype=np.float64) for k in ("Kp", "Kv", "KR", "Kw"))
    g, max_tilt = gn["g"], gn["max_tilt"]
    m, J = c.M, np.asarray(c.J, dtype=np.float64)
    p, R, v_world, w = obs_to_geo_model(obs)
    p_des, v_des, a_des = (np.asarray(x, dtype=np.float64) for x in (p_des, v_des, a_des))
    yaw_des = np.asarray(yaw_des, dtype=np.float64)
    yawrate_des = np.asarray(yawrate_des, dtype=np.float64)
    e3 = np.array([0.0, 0.0, 1.0])

    v = matTvec(R, v_world)                                                     # :70
    RTvd = matTvec(R, v_des)
    f_b = (matTvec(R, m * g * e3 * np.ones_like(p)) - m * matTvec(R, Kp * (p - p_des))
           - m * Kv * (v - RTvd) + m * (matTvec(R, a_des) - cross(w, RTvd)))    # :73-74  (w_hat @ x = w x x)
    f_w = matvec(R, f_b)                                                        # :75
    fn = norm(f_w)
    tilt = np.arccos(f_w[..., 2] / fn)                                          # :79
    xy_mag = norm(f_w[..., :2])
    with np.errstate(divide="ignore", invalid="ignore"):
        scale = f_w[..., 2] * np.tan(max_tilt) / xy_mag                         # :81-83
    clamp = tilt > max_tilt
    f_w = f_w.copy()
    f_w[..., 0] = np.where(clamp, f_w[..., 0] * scale, f_w[..., 0])             # :84
    f_w[..., 1] = np.where(clamp, f_w[..., 1] * scale, f_w[..., 1])
    f_b = matTvec(R, f_w)                                                       # :85
    fn = norm(f_w)

    z = np.zeros_like(yaw_des)
    b1c = np.stack([np.cos(yaw_des), np.sin(yaw_des), z], axis=-1)              # :88
    b3d = f_w / fn[..., None]
    c1 = cross(b3d, b1c)
    b2d = c1 / norm(c1)[..., None]
    c2 = cross(b2d, b3d)
    b1d = c2 / norm(c2)[..., None]
    R_des = np.stack([b1d, b2d, b3d], axis=-1)                                  # columns (:92)

    b1c_dot = np.stack([-np.sin(yaw_des) * yawrate_des, np.cos(yaw_des) * yawrate_des, z], axis=-1)  # :95
    f_dot_w = m * matvec(R, Kp * (v - RTvd)) / fn[..., None]                    # :96
    b3d_dot = cross(cross(b3d, f_dot_w), b3d)                                   # :97
    inner = (cross(b1c_dot, b3d) + cross(b1c, b3d_dot)) / norm(cross(b1c, b3d))[..., None]
    b2d_dot = cross(cross(b2d, inner), b2d)                                     # :98-99
    b1d_dot = cross(b3d_dot, b2d) + cross(b3d, b2d_dot)                         # :100
    R_dot_des = np.stack([b1d_dot, b2d_dot, b3d_dot], axis=-1)
    W = np.einsum("...ij,...jk->...ik", R_des, R_dot_des)                       # :102 (no-op transpose)
    w_des = np.stack([W[..., 2, 1], W[..., 0, 2], W[..., 1, 0]], axis=-1)       # :103

    if return_omegas:
        force = np.sum(f_w * R[..., :, 2], axis=-1)                             # :107
        return force, w_des, R_des

    E = np.einsum("...ji,...jk->...ik", R_des, R) - np.einsum("...ji,...jk->...ik", R, R_des)
    vee = np.stack([-E[..., 1, 2], E[..., 0, 2], -E[..., 0, 1]], axis=-1)       # :36-44
    e_R = 0.5 * KR * vee                                                        # :109
    torque = J * (-e_R - Kw * (w - matTvec(R, matvec(R_des, w_des)))) - cross(w, J * w)  # :110-111
    u = np.concatenate([np.maximum(0.0, f_b[..., 2:3]), torque], axis=-1)       # :114
    return input_to_action(u, c)                                                # :115


# --------------------------------------------------------------------------------------
# f-1: control/low_level/thrust_omega_ctrl.py:81-132 (body-rate PID -> PWM -> RPM) and
# control/lqr/lqr_omega_controller.py:77-88 compute_low_level (world -> body rate)
# --------------------------------------------------------------------------------------


class ThrustOmegaOracle:
    """ThrustOmegaController (stateful: last_omega, integral_omega_e), batched over drones.
    Constants :39-60: P=17500, I=10, D=0, PWM2RPM 0.2685/4070.3, PWM in [20000, 65535],
    torque clip +-3200, CF2P / CF2X mixer."""

    P = np.array([17500., 17500., 17500.])
    I = np.array([10., 10., 10.])
    Dg = np.array([0., 0., 0.])
    SCALE, CONST, MIN_PWM, MAX_PWM = 0.2685, 4070.3, 20000, 65535
    MIX = {"cf2x": np.array([[-.5, -.5, -1], [-.5, .5, 1], [.5, .5, -1], [.5, -.5, 1]]),
           "cf2p": np.array([[0, -1, -1], [+1, 0, 1], [0, 1, -1], [-1, 0, 1]], dtype=np.float64)}

    def __init__(self, n, c: DroneConsts = CF2P):
        self.c = c
        self.last_omega = np.zeros((n, 3))
        self.integral = np.zeros((n, 3))

    def compute_from_input(self, u, dt, cur_omega_body):
        """computeControlFromInput (:81-98) + omega_PID (:103-132): u = [thrust, wx, wy, wz]."""
        u = np.array(u, dtype=np.float64)
        cur = np.asarray(cur_omega_body, dtype=np.float64)
        u0 = np.clip(u[..., 0], 0, None)
        pwm_thrust = np.clip((np.sqrt(u0 / (self.c.KF * 4)) - self.CONST) / self.SCALE, self.MIN_PWM, self.MAX_PWM)
        rate_e = -(cur - self.last_omega) / dt
        e = u[..., 1:4] - cur
        self.last_omega = cur.copy()
        self.integral = self.integral - e * dt                       # sic: minus (:117)
        self.integral = np.clip(self.integral, -1500., 1500.)
        self.integral[..., 0:2] = np.clip(self.integral[..., 0:2], -1., 1.)
        tq = np.clip(self.P * e + self.I * self.integral + self.Dg * rate_e, -3200, 3200)
        pwm = pwm_thrust[..., None] + np.einsum("ij,...j->...i", self.MIX[self.c.MODEL], tq)
        pwm = np.clip(pwm, self.MIN_PWM, self.MAX_PWM)
        return self.SCALE * pwm + self.CONST

    def compute_low_level(self, u, obs, dt):
        """LQROmegaController.compute_low_level (lqr_omega_controller.py:77-88): obs[13:16] is the
        world-frame rate, rotated into the body frame with R(obs quat)^T."""
        obs = np.asarray(obs, dtype=np.float64)
        R = quat_to_rotmat_scipy(obs[..., 3:7])
        return self.compute_from_input(u, dt, matTvec(R, obs[..., 13:16]))


# --------------------------------------------------------------------------------------
# f-2: [UPSTREAM] gym_pybullet_drones.control.DSLPIDControl as PIDEnv.py:124-134,166-169 uses it.
# NOT in the reference tree and not installable: restated from the published upstream source;
# parity UNPINNED (no golden vectors exist).
# --------------------------------------------------------------------------------------


class DSLPIDOracle:
    """Position PID -> desired thrust vector / attitude, attitude PID -> PWM -> RPM (stateful), batched."""

    SCALE, CONST, MIN_PWM, MAX_PWM = 0.2685, 4070.3, 20000, 65535

    def __init__(self, n, c: DroneConsts = CF2P, gain_scale=1.0):
        self.c = c
        g = gain_scale                                   # PIDEnv.py:128-133 halves every gain
        self.P_FOR, self.I_FOR, self.D_FOR = g * np.array([.4, .4, 1.25]), g * np.array([.05, .05, .05]), g * np.array([.2, .2, .5])
        self.P_TOR, self.I_TOR, self.D_TOR = g * np.array([70000., 70000., 60000.]), g * np.array([.0, .0, 500.]), g * np.array([20000., 20000., 12000.])
        self.last_rpy = np.zeros((n, 3))
        self.integral_pos_e = np.zeros((n, 3))
        self.integral_rpy_e = np.zeros((n, 3))

    def compute_from_state(self, dt, obs, target_pos, target_rpy):
        """computeControlFromState(control_timestep, state=obs[j], target_pos, target_rpy) -> rpm [n,4]."""
        from scipy.spatial.transform import Rotation
        obs = np.asarray(obs, dtype=np.float64)
        pos, quat, vel = obs[..., 0:3], obs[..., 3:7], obs[..., 10:13]
        target_pos = np.asarray(target_pos, dtype=np.float64)
        target_rpy = np.asarray(target_rpy, dtype=np.float64)
        R = quat_to_rotmat_bullet(quat)
        # _dslPIDPositionControl
        pos_e = target_pos - pos
        vel_e = -vel                                      # target_vel = 0
        self.integral_pos_e = np.clip(self.integral_pos_e + pos_e * dt, -2., 2.)
        self.integral_pos_e[..., 2] = np.clip(self.integral_pos_e[..., 2], -0.15, .15)
        tt = self.P_FOR * pos_e + self.I_FOR * self.integral_pos_e + self.D_FOR * vel_e + np.array([0, 0, self.c.GRAVITY])
        scalar = np.maximum(0., np.sum(tt * R[..., :, 2], axis=-1))
        thrust = (np.sqrt(scalar / (4 * self.c.KF)) - self.CONST) / self.SCALE
        z_ax = tt / norm(tt)[..., None]
        x_c = np.stack([np.cos(target_rpy[..., 2]), np.sin(target_rpy[..., 2]), np.zeros_like(target_rpy[..., 2])], axis=-1)
        y_ax = cross(z_ax, x_c)
        y_ax = y_ax / norm(y_ax)[..., None]
        x_ax = cross(y_ax, z_ax)
        Rt = np.stack([x_ax, y_ax, z_ax], axis=-1)
        target_euler = Rotation.from_matrix(Rt).as_euler("XYZ")
        # _dslPIDAttitudeControl (target_rpy_rates = 0)
        cur_rpy = euler_from_quat_bullet(quat)
        Rt2 = Rotation.from_euler("XYZ", target_euler).as_matrix()
        E = np.einsum("...ji,...jk->...ik", Rt2, R) - np.einsum("...ji,...jk->...ik", R, Rt2)
        rot_e = np.stack([E[..., 2, 1], E[..., 0, 2], E[..., 1, 0]], axis=-1)
        rates_e = -(cur_rpy - self.last_rpy) / dt
        self.last_rpy = cur_rpy.copy()
        self.integral_rpy_e = np.clip(self.integral_rpy_e - rot_e * dt, -1500., 1500.)
        self.integral_rpy_e[..., 0:2] = np.clip(self.integral_rpy_e[..., 0:2], -1., 1.)
        tq = np.clip(-self.P_TOR * rot_e + self.D_TOR * rates_e + self.I_TOR * self.integral_rpy_e, -3200, 3200)
        pwm = np.clip(thrust[..., None] + np.einsum("ij,...j->...i", ThrustOmegaOracle.MIX[self.c.MODEL], tq), self.MIN_PWM, self.MAX_PWM)
        return self.SCALE * pwm + self.CONST


# --------------------------------------------------------------------------------------
# f-3: control/lqr/lqr_omega_controller.py:12-57 (gain), :90-119 (compute, cap_u)
# --------------------------------------------------------------------------------------


def linear_omega_AB(c: DroneConsts = CF2P):
    """model/linear_omega.py:46-53."""
    A = np.zeros((9, 9))
    B = np.zeros((9, 4))
    A[6:, 3:6] = np.eye(3)
    A[3, 1] = c.G
    A[4, 0] = -c.G
    B[5, 0] = 1.0 / c.M
    B[:3, 1:] = np.eye(3)
    return A, B


def lqr_omega_gain(c: DroneConsts = CF2P):
    """LQROmegaController.__init__/compute_gain_matrix (:12-57): Bryson weights, continuous ARE."""
    import scipy.linalg as la
    R = np.diag([1 / c.MAX_THRUST ** 2, 1 / 0.1 ** 2, 1 / 0.1 ** 2, 1 / 0.1 ** 2])
    Q = np.diag([1 / (np.pi / 20) ** 2] * 2 + [1 / (np.pi / 40) ** 2] + [1 / 0.15 ** 2] * 3 + [1 / 0.05 ** 2] * 3)
    A, B = linear_omega_AB(c)
    P = la.solve_continuous_are(A, B, Q, R, e=None, s=None, balanced=True)
    return la.solve(R, B.T @ P)


def lqr_omega_compute(obs, pos_des, vel_des, yaw_des, K, c: DroneConsts = CF2P):
    """LQROmegaController.compute(obs, skip_low_level=True) (:90-119) -> u = [F, wx, wy, wz] after cap_u.
    R_eq^T R(rpy) = Rz(yaw - yaw_des) Ry Rx, so its 'xyz' euler angles are (roll, pitch, wrap(yaw - yaw_des))."""
    obs = np.asarray(obs, dtype=np.float64)
    x = obs_to_lin_model(obs, 9, c)
    yd = np.asarray(yaw_des, dtype=np.float64)
    e = x.copy()
    dy = x[..., 2] - yd
    e[..., 2] = np.arctan2(np.sin(dy), np.cos(dy))
    cy, sy = np.cos(yd), np.sin(yd)

    def rot_eqT(v):     # R_eq^T v, R_eq = Rz(yaw_des)
        return np.stack([cy * v[..., 0] + sy * v[..., 1], -sy * v[..., 0] + cy * v[..., 1], v[..., 2]], axis=-1)
    e[..., 6:9] = rot_eqT(x[..., 6:9] - np.asarray(pos_des, dtype=np.float64))
    e[..., 3:6] = rot_eqT(x[..., 3:6] - np.asarray(vel_des, dtype=np.float64))
    u = -np.einsum("ij,...j->...i", K, e)
    u[..., 0] += c.M * c.G
    u[..., 0] = np.clip(u[..., 0], 4 * (9440.3 ** 2 * c.KF), c.MAX_THRUST)
    return u


# --------------------------------------------------------------------------------------
# control/lqr/lqr_controller.py (the default 'lqr' controller of simulations/EnvGeometric.py:32,425-427) on
# model/linearized.py: 12-state x = [rpy, ang_v, vel, pos], u = [F, tau] -> input_to_action
# --------------------------------------------------------------------------------------


def linearized_AB(c: DroneConsts = CF2P, noisy=False):
    """model/linearized.py:50-75 (A, B) or its deliberately wrong (Ahat, Bhat): inertia and mass off by 0.75."""
    A = np.zeros((12, 12))
    B = np.zeros((12, 4))
    A[0:3, 3:6] = np.eye(3)
    A[9:, 6:9] = np.eye(3)
    A[6, 1] = c.G
    A[7, 0] = -c.G
    B[8, 0] = 1.0 / c.M
    B[3:6, 1:] = np.diag(1.0 / np.asarray(c.J))
    if noisy:
        B[3:6, 1:] *= 0.75
        B[8, 0] = 1.0 / (c.M * 0.75)
    return A, B


def lqr12_gain(c: DroneConsts = CF2P, noisy=False):
    """LQRController.__init__/compute_gain_matrix (lqr_controller.py:12-57): Bryson weights, continuous ARE."""
    import scipy.linalg as la
    R = np.diag([1 / c.MAX_THRUST ** 2, 1 / 0.001 ** 2, 1 / 0.001 ** 2, 1 / 0.001 ** 2])
    Q = np.diag([1 / (np.pi / 40) ** 2] * 3 + [1 / 0.25 ** 2] * 3 + [1 / 0.15 ** 2] * 3 + [1 / 0.05 ** 2] * 3)
    A, B = linearized_AB(c, noisy)
    P = la.solve_continuous_are(A, B, Q, R, e=None, s=None, balanced=True)
    return la.solve(R, B.T @ P)


def lqr12_compute(obs, pos_des, vel_des, yaw_des, omega_des, K, c: DroneConsts = CF2P):
    """LQRController.compute(obs) (:83-113) -> (action rpm [.,4], u [.,4]).  x[3:6] is the obs' world-frame rate."""
    obs = np.asarray(obs, dtype=np.float64)
    x = obs_to_lin_model(obs, 12, c)
    yd = np.asarray(yaw_des, dtype=np.float64)
    e = x.copy()
    dy = x[..., 2] - yd
    e[..., 2] = np.arctan2(np.sin(dy), np.cos(dy))
    cy, sy = np.cos(yd), np.sin(yd)

    def rot_eqT(v):
        return np.stack([cy * v[..., 0] + sy * v[..., 1], -sy * v[..., 0] + cy * v[..., 1], v[..., 2]], axis=-1)
    wd = np.zeros(x.shape[:-1] + (3,))
    wd[..., 2] = omega_des
    e[..., 9:12] = rot_eqT(x[..., 9:12] - np.asarray(pos_des, dtype=np.float64))
    e[..., 6:9] = rot_eqT(x[..., 6:9] - np.asarray(vel_des, dtype=np.float64))
    e[..., 3:6] = rot_eqT(x[..., 3:6] - wd)
    u = -np.einsum("ij,...j->...i", K, e)
    u[..., 0] += c.M * c.G
    action = input_to_action(u, c)
    u[..., 0] = np.clip(u[..., 0], 0, None)      # the reference's input_to_action clips u[0] IN PLACE (model_conversions.py:88): the returned u carries it
    return action, u


# --------------------------------------------------------------------------------------
# f-1/f-3, order-3 loop: control/lqr/lqr_YO_controller.py:14-58 (gain), :99-124 (compute),
# :85-97 (compute_low_level) and control/low_level/yank_omega_ctrl.py:39-55
# --------------------------------------------------------------------------------------


def linear_yank_omega_AB(c: DroneConsts = CF2P):
    """model/linear_yank_omega.py:45-51: x = [r,p,y,F,vx,vy,vz,x,y,z], u = [Y,wx,wy,wz]."""
    A = np.zeros((10, 10))
    B = np.zeros((10, 4))
    A[7:, 4:7] = np.eye(3)
    A[4, 1] = c.G
    A[5, 0] = -c.G
    A[6, 3] = 1.0 / c.M
    B[:3, 1:] = np.eye(3)
    B[3, 0] = 1.0
    return A, B


def lqr_yank_omega_gain(c: DroneConsts = CF2P, ctrl_timestep=0.01):
    """LQRYankOmegaController.__init__/compute_gain_matrix (:14-64): Bryson weights with
    max_yank = MAX_THRUST / CTRL_TIMESTEP / 200 and max_thrust = MAX_THRUST - M G."""
    import scipy.linalg as la
    max_yank = (c.MAX_THRUST / ctrl_timestep) / 200
    R = np.diag([1 / max_yank ** 2, 1 / 0.1 ** 2, 1 / 0.1 ** 2, 1 / 0.1 ** 2])
    Q = np.diag([1 / (np.pi / 20) ** 2] * 2 + [1 / (np.pi / 40) ** 2] + [1 / (c.MAX_THRUST - c.M * c.G) ** 2]
                + [1 / 0.15 ** 2] * 3 + [1 / 0.05 ** 2] * 3)
    A, B = linear_yank_omega_AB(c)
    P = la.solve_continuous_are(A, B, Q, R, e=None, s=None, balanced=True)
    return la.solve(R, B.T @ P)


def lqr_yank_omega_compute(obs, pos_des, vel_des, yaw_des, K, c: DroneConsts = CF2P):
    """LQRYankOmegaController.compute(obs, skip_low_level=True) (:99-124) -> u = [Y, wx, wy, wz] = -K e
    (no hover offset, cap_u is a no-op :126-128).  e[3] = calc_z_thrust(obs) - M G."""
    obs = np.asarray(obs, dtype=np.float64)
    x = obs_to_lin_model(obs, 10, c)
    yd = np.asarray(yaw_des, dtype=np.float64)
    e = x.copy()
    dy = x[..., 2] - yd
    e[..., 2] = np.arctan2(np.sin(dy), np.cos(dy))
    e[..., 3] = x[..., 3] - c.M * c.G
    cy, sy = np.cos(yd), np.sin(yd)

    def rot_eqT(v):
        return np.stack([cy * v[..., 0] + sy * v[..., 1], -sy * v[..., 0] + cy * v[..., 1], v[..., 2]], axis=-1)
    e[..., 7:10] = rot_eqT(x[..., 7:10] - np.asarray(pos_des, dtype=np.float64))
    e[..., 4:7] = rot_eqT(x[..., 4:7] - np.asarray(vel_des, dtype=np.float64))
    return -np.einsum("ij,...j->...i", K, e)


class YankOmegaOracle:
    """YankOmegaController (yank_omega_ctrl.py:39-55) through LQRYankOmegaController.compute_low_level
    (lqr_YO_controller.py:85-97): thrust_cmd = calc_z_thrust(obs) + yank * dt, then the ThrustOmega PID."""

    def __init__(self, n, c: DroneConsts = CF2P):
        self.c = c
        self.toc = ThrustOmegaOracle(n, c)

    def compute_low_level(self, u, obs, dt):
        obs = np.asarray(obs, dtype=np.float64)
        u = np.array(u, dtype=np.float64)
        cur_thrust = self.c.KF * np.sum(obs[..., 16:20] ** 2, axis=-1)      # utils.calc_z_thrust (model_conversions.py:137-143)
        u[..., 0] = cur_thrust + u[..., 0] * dt                              # yank2thrust (:49-53)
        return self.toc.compute_low_level(u, obs, dt)


# --------------------------------------------------------------------------------------
# a5: model/dynamics.py:83-106
# --------------------------------------------------------------------------------------


def quadrotor_dynamics(state18, u4, m=6.77, J=(1.05, 1.05, 2.05), g=9.81):
    """QuadrotorDynamics.dynamics -> 12 floats (x_dot=v, "R_dot"=w, v_dot, w_dot).
    Defaults are the Hummingbird constants (:24-28); after load_env_params the
    reference keeps the STALE Hummingbird J (:18) but takes m, g from the env."""
    s = np.asarray(state18, dtype=np.float64)
    u = np.asarray(u4, dtype=np.float64)
    R = s[..., 3:12].reshape(s.shape[:-1] + (3, 3))
    v, w = s[..., 12:15], s[..., 15:18]
    J = np.asarray(J, dtype=np.float64)
    v_dot = R[..., :, 2] * (u[..., 0:1] / m)
    v_dot = v_dot.copy()
    v_dot[..., 2] -= g
    w_dot = (u[..., 1:4] - cross(w, J * w)) / J
    return np.concatenate([v, w, v_dot, w_dot], axis=-1)


# --------------------------------------------------------------------------------------
# obs -> linear-model state  (utils/model_conversions.py:20-58, :137-143)
# --------------------------------------------------------------------------------------


def obs_to_lin_model(obs, dim, c: DroneConsts = CF2P):
    obs = np.asarray(obs, dtype=np.float64)
    rpy, vel, pos, angv = obs[..., 7:10], obs[..., 10:13], obs[..., 0:3], obs[..., 13:16]
    if dim == 12:
        return np.concatenate([rpy, angv, vel, pos], axis=-1)
    if dim == 9:
        return np.concatenate([rpy, vel, pos], axis=-1)
    if dim == 10:
        thrust = np.sum(c.KF * obs[..., 16:20] ** 2, axis=-1, keepdims=True)    # calc_z_thrust
        return np.concatenate([rpy, thrust, vel, pos], axis=-1)
    raise ValueError("Invalid dim for linear model")


# --------------------------------------------------------------------------------------
# the call site of a5 -- simulations/CompareModels.py:46-56 (per logged observation: the linear model's x_dot, the geometric
# model's x_dot in the linear model's layout, the linear state) and the helpers around it
# --------------------------------------------------------------------------------------


def rpy_to_rot(rpy):
    """utils/model_conversions.py:4-19: R = Rz(yaw) Ry(pitch) Rx(roll)."""
    rpy = np.asarray(rpy, dtype=np.float64)
    cr, sr = np.cos(rpy[..., 0]), np.sin(rpy[..., 0])
    cp, sp = np.cos(rpy[..., 1]), np.sin(rpy[..., 1])
    cy, sy = np.cos(rpy[..., 2]), np.sin(rpy[..., 2])
    R = np.empty(rpy.shape[:-1] + (3, 3))
    R[..., 0, 0], R[..., 0, 1], R[..., 0, 2] = cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr
    R[..., 1, 0], R[..., 1, 1], R[..., 1, 2] = sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr
    R[..., 2, 0], R[..., 2, 1], R[..., 2, 2] = -sp, cp * sr, cp * cr
    return R


def rotmat_to_quat_scipy(R):
    """scipy Rotation.from_matrix(R).as_quat() (xyzw) as utils/model_conversions.py:119 uses it: the branch with the largest of
    (R00, R11, R22, trace) -- scipy's `_rotation.pyx` from_matrix --, normalised; the sign is whatever that branch gives."""
    R = np.asarray(R, dtype=np.float64)
    lead = R.shape[:-2]
    Rf = R.reshape(-1, 3, 3)
    q = np.empty((Rf.shape[0], 4))
    dec = np.stack([Rf[:, 0, 0], Rf[:, 1, 1], Rf[:, 2, 2], Rf[:, 0, 0] + Rf[:, 1, 1] + Rf[:, 2, 2]], axis=1)
    ch = np.argmax(dec, axis=1)
    for n in range(Rf.shape[0]):
        m, c = Rf[n], ch[n]
        if c != 3:
            i, j, k = c, (c + 1) % 3, (c + 2) % 3
            q[n, i] = 1 - dec[n, 3] + 2 * m[i, i]
            q[n, j] = m[j, i] + m[i, j]
            q[n, k] = m[k, i] + m[i, k]
            q[n, 3] = m[k, j] - m[j, k]
        else:
            q[n, 0] = m[2, 1] - m[1, 2]
            q[n, 1] = m[0, 2] - m[2, 0]
            q[n, 2] = m[1, 0] - m[0, 1]
            q[n, 3] = 1 + dec[n, 3]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.reshape(lead + (4,))


def geo_model_to_obs(x18):
    """utils/model_conversions.py:116-122: [pos, R row-major, vel, ang_v] -> the first 16 observation values (rpy slots 7:10 left 0)."""
    x = np.asarray(x18, dtype=np.float64)
    obs = np.zeros(x.shape[:-1] + (16,))
    obs[..., 0:3] = x[..., 0:3]
    obs[..., 3:7] = rotmat_to_quat_scipy(x[..., 3:12].reshape(x.shape[:-1] + (3, 3)))
    obs[..., 10:13] = x[..., 12:15]
    obs[..., 13:16] = x[..., 15:18]
    return obs


def geo_x_dot_to_linear(geo_xdot):
    """utils/model_conversions.py:124-135: (v, w, v_dot, w_dot) -> the linear model's order (w, w_dot, v_dot, v)."""
    g = np.asarray(geo_xdot, dtype=np.float64)
    return np.concatenate([g[..., 3:6], g[..., 9:12], g[..., 6:9], g[..., 0:3]], axis=-1)


def linear_calc_xdot(x12, action, A, B, c: DroneConsts = CF2P):
    """model/linearized.py:92-104 LinearizedModel.calc_xdot: A (x - x_eq) + B (u - u_eq) with u = action_to_input(env, action),
    x_eq = (0, ..., 0, position of x), u_eq = (M G, 0, 0, 0)."""
    x = np.array(x12, dtype=np.float64)
    u = action_to_input(action, c)
    dx = x.copy()
    dx[..., 9:12] = 0.0
    du = u.copy()
    du[..., 0] -= c.M * c.G
    return np.einsum("ij,...j->...i", A, dx) + np.einsum("ij,...j->...i", B, du)


def compare_models(obs, A, B, c: DroneConsts = CF2P, dyn_m=None, dyn_J=(1.05, 1.05, 2.05), dyn_g=None):
    """The loop body of simulations/CompareModels.py:48-56 over observation rows: (x_dot_linear, x_dot_geometric, x_lin_obs).
    The geometric side is QuadrotorDynamics after load_env_params(env): m, g from the env, J the stale Hummingbird one
    (model/dynamics.py:8-20)."""
    obs = np.asarray(obs, dtype=np.float64)
    x_lin = obs_to_lin_model(obs, 12, c)
    xdot_lin = linear_calc_xdot(x_lin, obs[..., 16:20], A, B, c)
    p, R, v, w = obs_to_geo_model(obs)
    s18 = np.concatenate([p, R.reshape(R.shape[:-2] + (9,)), v, w], axis=-1)
    u = action_to_input(obs[..., 16:20], c)
    geo = quadrotor_dynamics(s18, u, c.M if dyn_m is None else dyn_m, dyn_J, c.G if dyn_g is None else dyn_g)
    return xdot_lin, geo_x_dot_to_linear(geo), x_lin


# --------------------------------------------------------------------------------------
# a11-a14: cbf/cbf.py rows in closed form (pinned against the dense
# reference construction by tests/golden/cbf_rows_*.npz)
# --------------------------------------------------------------------------------------


def place_poles_chain(poles):
    """cbf/cbf.py:115-124: Kcbf = place_poles(chain of integrators) = ascending
    coefficients of prod(s - p_i) without the leading 1."""
    co = np.poly(np.asarray(poles, dtype=np.float64))  # descending, leading 1
    return co[1:][::-1].copy()


def _cbf_pair_terms(xi, xj, xi_des, xj_des, order, Ds, zscale, Kcbf, m, g):
    """Closed form of custom_hdots (:135-178) + custom_control_affine_terms (:194-283) for
    the hover linearisations (model/linear_omega.py:46-53, linear_yank_omega.py:45-51).
    Returns (h_ij, Lg[4]) with G_ij[4i:4i+4] = -Lg, G_ij[4j:4j+4] = +Lg."""
    c4 = zscale ** 4
    e = xi[..., -3:] - xj[..., -3:]
    ex, ey, ez = e[..., 0], e[..., 1], e[..., 2]
    s = ex * ex + ey * ey
    h = s * s + (ez / zscale) ** 4 - Ds ** 4
    gx, gy, gz = 4 * ex * s, 4 * ey * s, 4 * ez ** 3 / c4
    Hxx, Hxy, Hyy, Hzz = 12 * ex * ex + 4 * ey * ey, 8 * ex * ey, 4 * ex * ex + 12 * ey * ey, 12 * ez * ez / c4
    hi = xi - xi_des
    hj = xj - xj_des
    d = hi - hj
    if order == 2:
        dr, dp = d[..., 0], d[..., 1]
        dvx, dvy, dvz = d[..., 3], d[..., 4], d[..., 5]
        dax, day, daz = g * dp, -g * dr, 0.0 * dr
        hdot = gx * dvx + gy * dvy + gz * dvz
        quad = Hxx * dvx * dvx + 2 * Hxy * dvx * dvy + Hyy * dvy * dvy + Hzz * dvz * dvz
        Lf2 = gx * dax + gy * day + gz * daz + quad
        hij = Kcbf[0] * h + Kcbf[1] * hdot + Lf2
        zero = 0.0 * gz
        Lg = np.stack([gz / m, zero, zero, zero], axis=-1)
        return hij, Lg
    if order == 3:
        dr, dp, dF = d[..., 0], d[..., 1], d[..., 3]
        dvx, dvy, dvz = d[..., 4], d[..., 5], d[..., 6]
        dax, day, daz = g * dp, -g * dr, dF / m
        hdot = gx * dvx + gy * dvy + gz * dvz
        # custom_hdots i == 2 uses hard-coded slots 6,7,8 = (vz, x, y) of the 10-state (:158-169):
        # dhdx picks (A xhat)[6,7,8] = (dF/m, dvx, dvy); d2h likewise.
        w0, w1, w2 = dF / m, dvx, dvy
        A2 = g * (gy * dp - gz * dr)       # dhdx@A@Axhat with dhde on slots 6,7,8: A[7,4]*(A xhat)[4] etc.
        hddot_ref = A2 + (Hxx * w0 * w0 + 2 * Hxy * w0 * w1 + Hyy * w1 * w1 + Hzz * w2 * w2)
        Hdv_da = (Hxx * dvx * dax + Hxy * (dvx * day + dvy * dax) + Hyy * dvy * day + Hzz * dvz * daz)
        T = (24 * ex * dvx ** 3 + 24 * ey * dvx ** 2 * dvy + 24 * ex * dvx * dvy ** 2 + 24 * ey * dvy ** 3
             + (24 * ez / c4) * dvz ** 3)
        Lf3 = 3 * Hdv_da + T
        hij = Kcbf[0] * h + Kcbf[1] * hdot + Kcbf[2] * hddot_ref + Lf3
        Lg = np.stack([gz / m, -g * gy, g * gx, 0.0 * gz], axis=-1)
        return hij, Lg
    raise ValueError("order must be 2 or 3")


def cbf_rows(x, xdes, order, Kcbf, umax, safety_radius, zscale, c: DroneConsts = CF2P, x_obs=None, obs_r=None,
             Fmin=None, Fmax=None):
    """CBF._build_ineq_const (cbf/cbf.py:308-367) for ONE env: x, xdes [N, xdim] ->
    (G [m, 4N], h [m]) in the reference's row order: pairs (i<j lexicographic) | +I | -I |
    (order 3: 2 force rows per agent) | obstacles agent-major."""
    x = np.asarray(x, dtype=np.float64)
    xdes = np.asarray(xdes, dtype=np.float64)
    N, xdim = x.shape
    Kcbf = np.asarray(Kcbf, dtype=np.float64)
    rows_G, rows_h = [], []
    for i in range(N - 1):
        for j in range(i + 1, N):
            hij, Lg = _cbf_pair_terms(x[i], x[j], xdes[i], xdes[j], order, 2 * safety_radius, zscale, Kcbf, c.M, c.G)
            Gij = np.zeros(4 * N)
            Gij[4 * i:4 * i + 4] = -Lg
            Gij[4 * j:4 * j + 4] = Lg
            rows_G.append(Gij)
            rows_h.append(hij)
    G = np.array(rows_G).reshape(-1, 4 * N)
    h = np.array(rows_h).reshape(-1)
    if umax is not None:                                                       # :400-412
        G = np.vstack([G, np.eye(4 * N), -np.eye(4 * N)])
        h = np.hstack([h, np.tile(np.asarray(umax, dtype=np.float64), 2 * N)])
    if order == 3:                                                             # :446-464 (quirk: column 4i+3)
        Fmin = -c.M * c.G if Fmin is None else Fmin
        Fmax = c.MAX_THRUST if Fmax is None else Fmax
        for i in range(N):
            Gi = np.zeros((2, 4 * N))
            Gi[0, 4 * i + 3] = 1
            Gi[1, 4 * i + 3] = -1
            hi = np.array([Kcbf[-1] * (Fmax - x[i, 3]), Kcbf[-1] * (x[i, 3] - Fmin)])
            G = np.vstack([G, Gi])
            h = np.hstack([h, hi])
    if x_obs is not None and obs_r is not None and len(obs_r) > 0:             # :369-398
        x_obs = np.asarray(x_obs, dtype=np.float64).reshape(len(obs_r), -1, 3)
        Go = np.zeros((N * len(obs_r), 4 * N))
        ho = np.zeros(N * len(obs_r))
        for i in range(N):
            for j, r in enumerate(obs_r):
                xo = np.zeros(xdim)
                xo[-3:] = x_obs[j][0]
                hij, Lg = _cbf_pair_terms(x[i], xo, xdes[i], xo, order, safety_radius + r, zscale, Kcbf, c.M, c.G)
                Go[i * len(obs_r) + j, 4 * i:4 * i + 4] = -Lg
                ho[i * len(obs_r) + j] = hij
        G = np.vstack([G, Go])
        h = np.hstack([h, ho])
    return G, h


# --------------------------------------------------------------------------------------
# a15: cbf/qptracker.py:86-114 -- min 1/2|u|^2 - uhat^T u  s.t. G u <= h
# cvxopt 1.3.2 (environment.yaml:52; interior point, coneqp) is absent here: QP *solution* parity
# is UNPINNED against cvxopt.
#  * Feasible rows: the problem is strictly convex, the minimiser is unique, and this exact
#    active-set solver is the oracle; cvxopt returns the same point up to its own tolerances
#    (feastol = abstol = 1e-7, reltol = 1e-6).
#  * Infeasible rows: (False, None) -> the caller keeps uhat (status 1).  This is a MODELLED
#    policy, NOT what the reference does: _rectify sets success = True as soon as
#    solvers.qp returns (qptracker.py:103-112) and falls back (:30-34) only when it RAISES.
#    cvxopt's coneqp does not certify QP infeasibility -- it returns status 'unknown' with its
#    last iterates (iteration limit, or a singular KKT system after iteration 0) and raises only
#    the rank ValueError that P = I excludes -- so on such an env the reference most likely
#    applies cvxopt's last iterate to all of its drones.  Parity on that branch: unpinned and
#    probably different.  oracle/cvxopt_qp.py restates coneqp from its published algorithm (unpinned too) to
#    make both statements measurable: tests/test_oracle_cvxopt_cpu.py, tests/tools/c4_cvxopt_probe.py.
# --------------------------------------------------------------------------------------


def qp_project(uhat, G, h, tol=1e-10, max_iter=None):
    """Goldfarb-Idnani style dual active-set for  min 1/2|u - uhat|^2  s.t. G u <= h.
    Returns (success, u, lam)."""
    uhat = np.asarray(uhat, dtype=np.float64).reshape(-1)
    G = np.asarray(G, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)
    m = G.shape[0]
    # rows scaled to unit norm (same feasible set, same minimiser; multipliers are returned for the
    # ORIGINAL rows): keeps the step-length logic well conditioned when |G_k| spans 1e-7..1e2
    rn = np.sqrt((G * G).sum(axis=1))
    if np.any((rn == 0) & (h < 0)):
        return False, None, None                          # 0 * u <= h with h < 0
    rs = np.where(rn > 0, rn, 1.0)
    G = G / rs[:, None]
    h = np.where(rn > 0, h / rs, np.inf)
    if max_iter is None:
        max_iter = 20 * (m + 10)
    u = uhat.copy()
    active = []
    lam = np.zeros(0)
    for _ in range(max_iter):
        viol = G @ u - h
        scale = np.maximum(1.0, np.where(np.isfinite(h), np.abs(h), 1.0))
        k = int(np.argmax(viol / scale))
        if viol[k] <= tol * scale[k]:
            full = np.zeros(m)
            full[active] = lam
            return True, u, full / rs
        # add constraint k: move along z (primal) / r (dual) as in Goldfarb-Idnani with H = I
        gk = G[k]
        lam_k = 0.0
        while True:
            if active:
                Na = G[active].T                      # n x q
                M = Na.T @ Na
                try:
                    r = np.linalg.solve(M, Na.T @ gk)
                except np.linalg.LinAlgError:
                    r = np.linalg.lstsq(M, Na.T @ gk, rcond=None)[0]
                z = gk - Na @ r
            else:
                r = np.zeros(0)
                z = gk.copy()
            zz = z @ z
            # dual step length
            t1, drop = np.inf, -1
            for idx in range(len(active)):
                if r[idx] > 1e-14:
                    cand = lam[idx] / r[idx]
                    if cand < t1:
                        t1, drop = cand, idx
            if zz > 1e-18 * max(1.0, gk @ gk):
                t2 = (gk @ u - h[k]) / zz
            else:
                t2 = np.inf
            t = min(t1, t2)
            if not np.isfinite(t):
                return False, None, None              # infeasible
            if t2 == np.inf:                          # dual step only
                lam = lam - t * r
                lam_k += t
                active.pop(drop)
                lam = np.delete(lam, drop)
                continue
            u = u - t * z
            lam = lam - t * r
            lam_k += t
            if t == t2:
                active.append(k)
                lam = np.append(lam, lam_k)
                break
            active.pop(drop)
            lam = np.delete(lam, drop)
    return False, None, None


def cbf_filter(x, xdes, u_nominal, order, Kcbf, umax, safety_radius, zscale, c: DroneConsts = CF2P, x_obs=None,
               obs_r=None, Fmin=None, Fmax=None):
    """DroneQPTracker.compute_control (qptracker.py:22-34) for ONE env, given linear-model
    states x [N, xdim].  Returns (u [N,4], status) with status 0 = solved, 1 = infeasible (modelled
    fallback to u_nominal -- see the note above qp_project: the reference falls back only when cvxopt raises)."""
    G, h = cbf_rows(x, xdes, order, Kcbf, umax, safety_radius, zscale, c, x_obs, obs_r, Fmin, Fmax)
    u_nominal = np.asarray(u_nominal, dtype=np.float64)
    ok, u, _ = qp_project(u_nominal.reshape(-1), G, h)
    if not ok:
        return u_nominal.copy(), 1
    return u.reshape(u_nominal.shape), 0
