/* TEST INFRASTRUCTURE ONLY -- plain C (float64) restatement of the reference's per-drone control loop, written from the reference's
 * files and from SURVEY.md section 3.4, independently of oracle/np_oracle.py (which it must agree with) and of the HIP kernels:
 *
 *     trajs[j](t) -> ctrl[j].compute(obs[j]) -> env.step(action)            simulations/EnvGeometric.py:434-469
 *
 * Every function cites the reference lines it follows (paths relative to the reference checkout).  [UPSTREAM] marks behaviour of
 * gym-pybullet-drones / pybullet, which are not in the reference tree: those follow SURVEY.md 3.4 and are pinned as np_oracle.py is
 * (tests/golden/dyn_wrench_accel.npz, attitude_flow.npz, euler_convention.npz; the update order and the observation packing stay
 * spec-level: "parity unpinned" for those pieces).
 *
 * Who may use it: tests/ (as a second checker), __graft_entry__.smoke() and bench.py's cpu_baseline leg (kind "port": the same loop on
 * the host cores, OpenMP over drones).  The product (multidronesim_amd/) never loads it.
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC -> oracle/libc_oracle.so; __graft_entry__.build() runs it). */
#include <math.h>
#include <string.h>
#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  double M, L, KF, KM, J[3], G, MAX_RPM, MAX_THRUST; /* [UPSTREAM] cf2p.urdf + BaseAviary.__init__ */
  double Kp[3], Kv[3], KR[3], Kw[3], g_ctrl, max_tilt; /* control/geometric.py:14-23 */
  int substeps;                                        /* PYB_FREQ // CTRL_FREQ */
  double pyb_dt, ctrl_dt;                              /* 1 / PYB_FREQ, 1 / CTRL_FREQ */
} co_consts;

/* one drone: pos3 | quat4 xyzw | vel3 | body rates3 | ang_v3 (world) | last clipped action 4 */
#define CO_STATE 20
#define CO_OBS 20

static void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static double norm3(const double a[3]) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static void matvec3(const double R[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
}
static void matTvec3(const double R[9], const double v[3], double o[3]) {
  for (int i = 0; i < 3; ++i) o[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2];
}

/* scipy Rotation.from_quat(q).as_matrix(): normalising, xyzw -- utils/model_conversions.py:110 */
static void quat_to_R_scipy(const double q0[4], double R[9]) {
  const double n = sqrt(q0[0] * q0[0] + q0[1] * q0[1] + q0[2] * q0[2] + q0[3] * q0[3]);
  const double x = q0[0] / n, y = q0[1] / n, z = q0[2] / n, w = q0[3] / n;
  R[0] = x * x - y * y - z * z + w * w; R[1] = 2 * (x * y - z * w);           R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);           R[4] = -x * x + y * y - z * z + w * w; R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);           R[7] = 2 * (y * z + x * w);           R[8] = -x * x - y * y + z * z + w * w;
}
/* [UPSTREAM] p.getMatrixFromQuaternion (btMatrix3x3::setRotation, s = 2 / |q|^2) */
static void quat_to_R_bullet(const double q[4], double R[9]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double s = 2.0 / (x * x + y * y + z * z + w * w);
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  R[0] = 1.0 - (yy + zz); R[1] = xy - wz;         R[2] = xz + wy;
  R[3] = xy + wz;         R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
  R[6] = xz - wy;         R[7] = yz + wx;         R[8] = 1.0 - (xx + yy);
}
/* [UPSTREAM] p.getQuaternionFromEuler (btQuaternion::setEulerZYX), xyzw */
static void quat_from_euler_bullet(const double rpy[3], double q[4]) {
  const double cr = cos(rpy[0] * 0.5), sr = sin(rpy[0] * 0.5), cp = cos(rpy[1] * 0.5), sp = sin(rpy[1] * 0.5), cy = cos(rpy[2] * 0.5),
               sy = sin(rpy[2] * 0.5);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}
/* [UPSTREAM] p.getEulerFromQuaternion: ZYX with the +-0.99999 gimbal branches */
static void euler_from_quat_bullet(const double q[4], double rpy[3]) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
  const double sarg = -2.0 * (x * z - w * y);
  if (sarg <= -0.99999) {
    rpy[0] = 0.0; rpy[1] = -0.5 * M_PI; rpy[2] = 2 * atan2(x, -y);
  } else if (sarg >= 0.99999) {
    rpy[0] = 0.0; rpy[1] = 0.5 * M_PI; rpy[2] = 2 * atan2(-x, y);
  } else {
    rpy[0] = atan2(2 * (y * z + w * x), squ - sqx - sqy + sqz);
    rpy[1] = asin(sarg);
    rpy[2] = atan2(2 * (x * y + w * z), squ + sqx - sqy - sqz);
  }
}

/* trajectories/Lemniscate.py:32-63: -> des[11] = pos3 vel3 acc3 yaw yaw_rate;  P[7] = a, omega, cx, cy, cz, yaw_rate, phase_shift */
static void lemniscate(double t, const double P[7], double des[11]) {
  const double a = P[0], om = P[1], th = t * om + P[6];
  const double s = sin(th), c = cos(th), s2 = s * s, c2t = cos(2 * th), d = (c2t - 3) * (c2t - 3) * (c2t - 3);
  des[0] = P[2] + (a * s * c) / (1 + s2);
  des[1] = P[3] + (a * c) / (1 + s2);
  des[2] = P[4];
  des[3] = -a * om * (s2 * s2 + s2 + (s2 - 1) * c * c) / ((s2 + 1) * (s2 + 1));
  des[4] = -a * om * s * (s2 + 2 * c * c + 1) / ((s2 + 1) * (s2 + 1));
  des[5] = 0.0;
  des[6] = 4 * a * om * om * sin(2 * th) * (3 * c2t + 7) / d;
  des[7] = a * om * om * c * (44 * c2t + cos(4 * th) - 21) / d;
  des[8] = 0.0;
  des[9] = M_PI * sin(P[5] * t);
  des[10] = M_PI * P[5] * cos(P[5] * t);
}

/* utils/model_conversions.py:85-103 input_to_action: the inverse of the "+"-frame mixer in closed form (the reference inverts the 4 x 4
 * matrix numerically every call), thrusts clipped to [9440.3^2 KF, MAX_THRUST], rpm = sqrt(T / KF) */
static void input_to_action(const co_consts* c, const double u_in[4], double rpm[4]) {
  const double r = c->KM / c->KF, L = c->L;
  const double u0 = u_in[0] < 0.0 ? 0.0 : u_in[0], u1 = u_in[1], u2 = u_in[2], u3 = u_in[3];
  /* rows of inv([[1,1,1,1],[0,L,0,-L],[-L,0,L,0],[-r,r,-r,r]]) */
  double T[4];
  T[0] = 0.25 * u0 - u2 / (2 * L) - u3 / (4 * r);
  T[1] = 0.25 * u0 + u1 / (2 * L) + u3 / (4 * r);
  T[2] = 0.25 * u0 + u2 / (2 * L) - u3 / (4 * r);
  T[3] = 0.25 * u0 - u1 / (2 * L) + u3 / (4 * r);
  const double lo = 9440.3 * 9440.3 * c->KF, hi = c->MAX_THRUST;
  for (int k = 0; k < 4; ++k) {
    const double t = T[k] < lo ? lo : (T[k] > hi ? hi : T[k]);
    rpm[k] = sqrt(t / c->KF);
  }
}

/* control/geometric.py:59-115 GeometricControl.compute with its quirks: g = 9.81 (:20); obs[13:16] (world angular velocity) used as
 * the body rate (:63); R_des.transpose(0, 1) is a no-op on an ndarray so w_des_hat = R_des @ R_dot_des (:102) */
static void geometric_compute_ex(const co_consts* c, const double obs[20], const double des[11], double rpm[4], double* omegas /* NULL | [4]: force, w_des (return_omegas=True, :105-107) */) {
  const double m = c->M, g = c->g_ctrl;
  double R[9];
  quat_to_R_scipy(obs + 3, R);                                   /* obs_to_geo_model, utils/model_conversions.py:105-114 */
  const double* p = obs;
  const double* w = obs + 13;
  const double *p_des = des, *v_des = des + 3, *a_des = des + 6;
  const double yaw = des[9], yaw_rate = des[10];
  double v[3], RTvd[3], RTa[3], wxv[3], e3g[3] = {0.0, 0.0, m * g}, RTe3[3], kpe[3], RTkpe[3];
  matTvec3(R, obs + 10, v);                                      /* :70 */
  matTvec3(R, v_des, RTvd);
  matTvec3(R, a_des, RTa);
  cross3(w, RTvd, wxv);                                          /* w_hat @ RT @ v_des */
  matTvec3(R, e3g, RTe3);
  for (int k = 0; k < 3; ++k) kpe[k] = c->Kp[k] * (p[k] - p_des[k]);
  matTvec3(R, kpe, RTkpe);
  double f_b[3], f_w[3];
  for (int k = 0; k < 3; ++k) f_b[k] = RTe3[k] - m * RTkpe[k] - m * c->Kv[k] * (v[k] - RTvd[k]) + m * (RTa[k] - wxv[k]);   /* :73-74 */
  matvec3(R, f_b, f_w);                                          /* :75 */
  const double tilt = acos(f_w[2] / norm3(f_w));                 /* :79 */
  if (tilt > c->max_tilt) {                                      /* :80-84 */
    const double xy_mag = sqrt(f_w[0] * f_w[0] + f_w[1] * f_w[1]);
    const double scale = f_w[2] * tan(c->max_tilt) / xy_mag;
    f_w[0] *= scale;
    f_w[1] *= scale;
  }
  matTvec3(R, f_w, f_b);                                         /* :85 */
  const double fn = norm3(f_w);
  const double b1c[3] = {cos(yaw), sin(yaw), 0.0};               /* :88 */
  double b3d[3], b2d[3], b1d[3], t3[3];
  for (int k = 0; k < 3; ++k) b3d[k] = f_w[k] / fn;
  cross3(b3d, b1c, t3);
  double n = norm3(t3);
  for (int k = 0; k < 3; ++k) b2d[k] = t3[k] / n;
  cross3(b2d, b3d, t3);
  n = norm3(t3);
  for (int k = 0; k < 3; ++k) b1d[k] = t3[k] / n;
  /* desired angular velocity (:95-103) */
  const double b1c_dot[3] = {-sin(yaw) * yaw_rate, cos(yaw) * yaw_rate, 0.0};
  double kpv[3], f_dot[3];
  for (int k = 0; k < 3; ++k) kpv[k] = c->Kp[k] * (v[k] - RTvd[k]);
  matvec3(R, kpv, f_dot);
  for (int k = 0; k < 3; ++k) f_dot[k] = m * f_dot[k] / fn;      /* :96 */
  double b3d_dot[3], b2d_dot[3], b1d_dot[3], u1[3], u2[3], inner[3];
  cross3(b3d, f_dot, u1);
  cross3(u1, b3d, b3d_dot);                                      /* :97 */
  cross3(b1c_dot, b3d, u1);
  cross3(b1c, b3d_dot, u2);
  cross3(b1c, b3d, t3);
  n = norm3(t3);
  for (int k = 0; k < 3; ++k) inner[k] = (u1[k] + u2[k]) / n;
  cross3(b2d, inner, u1);
  cross3(u1, b2d, b2d_dot);                                      /* :98-99 */
  cross3(b3d_dot, b2d, u1);
  cross3(b3d, b2d_dot, u2);
  for (int k = 0; k < 3; ++k) b1d_dot[k] = u1[k] + u2[k];        /* :100 */
  double Rd[9], Rdd[9], W[9];
  for (int k = 0; k < 3; ++k) {                                  /* columns b1d b2d b3d (:92, :101) */
    Rd[3 * k] = b1d[k]; Rd[3 * k + 1] = b2d[k]; Rd[3 * k + 2] = b3d[k];
    Rdd[3 * k] = b1d_dot[k]; Rdd[3 * k + 1] = b2d_dot[k]; Rdd[3 * k + 2] = b3d_dot[k];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) W[3 * i + j] = Rd[3 * i] * Rdd[j] + Rd[3 * i + 1] * Rdd[3 + j] + Rd[3 * i + 2] * Rdd[6 + j];   /* :102 (no transpose) */
  const double w_des[3] = {W[7], W[2], W[3]};                    /* :103 */
  if (omegas) {                                                  /* :105-107: force = f_des_world . (R e3) */
    omegas[0] = f_w[0] * R[2] + f_w[1] * R[5] + f_w[2] * R[8];
    omegas[1] = w_des[0]; omegas[2] = w_des[1]; omegas[3] = w_des[2];
    return;
  }
  /* attitude error and torque (:108-111) */
  double E[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0.0, b = 0.0;
      for (int k = 0; k < 3; ++k) {
        a += Rd[3 * k + i] * R[3 * k + j];                       /* R_des^T R */
        b += R[3 * k + i] * Rd[3 * k + j];                       /* R^T R_des */
      }
      E[3 * i + j] = a - b;
    }
  const double vee[3] = {-E[5], E[2], -E[1]};                    /* vee_map :36-44 */
  double Rdw[3], RTRdw[3], Jw[3], wJw[3], torque[3];
  matvec3(Rd, w_des, Rdw);
  matTvec3(R, Rdw, RTRdw);
  for (int k = 0; k < 3; ++k) Jw[k] = c->J[k] * w[k];
  cross3(w, Jw, wJw);
  for (int k = 0; k < 3; ++k) torque[k] = c->J[k] * (-0.5 * c->KR[k] * vee[k] - c->Kw[k] * (w[k] - RTRdw[k])) - wJw[k];
  const double u[4] = {f_b[2] > 0.0 ? f_b[2] : 0.0, torque[0], torque[1], torque[2]};   /* :114 */
  input_to_action(c, u, rpm);                                    /* :115 */
}

static void geometric_compute(const co_consts* c, const double obs[20], const double des[11], double rpm[4]) {
  geometric_compute_ex(c, obs, des, rpm, 0);
}

/* [UPSTREAM] BaseAviary._integrateQ: the exact exponential for a constant body rate over dt */
static void integrate_q(double q[4], const double om[3], double dt) {
  const double on = norm3(om);
  if (fabs(on) <= 1e-8) return;                                  /* np.isclose(norm, 0) */
  const double p = om[0], qq = om[1], r = om[2];
  const double th = on * dt / 2, ct = cos(th), k = sin(th) / on;
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  q[0] = ct * x + k * (r * y - qq * z + p * w);
  q[1] = ct * y + k * (-r * x + p * z + qq * w);
  q[2] = ct * z + k * (qq * x - p * y + r * w);
  q[3] = ct * w + k * (-p * x - qq * y - r * z);
}

/* [UPSTREAM] BaseAviary.step for one drone (Physics.DYN, explicit Euler): clip the RPM, PYB_FREQ // CTRL_FREQ substeps of _dynamics
 * (v and the body rates first; the position with the NEW velocity, the quaternion with the NEW rates; the world angular velocity
 * handed to Bullet is R_old @ rates), remember the clipped action */
static void aviary_step(const co_consts* c, double st[CO_STATE], const double action[4]) {
  double rpm[4];
  for (int k = 0; k < 4; ++k) rpm[k] = action[k] < 0.0 ? 0.0 : (action[k] > c->MAX_RPM ? c->MAX_RPM : action[k]);
  double *pos = st, *quat = st + 3, *vel = st + 7, *rates = st + 10, *angv = st + 13;
  const double dt = c->pyb_dt;
  for (int s = 0; s < c->substeps; ++s) {
    double R[9], f[4], zt[4];
    quat_to_R_bullet(quat, R);
    for (int k = 0; k < 4; ++k) {
      f[k] = rpm[k] * rpm[k] * c->KF;
      zt[k] = rpm[k] * rpm[k] * c->KM;
    }
    const double thrust = f[0] + f[1] + f[2] + f[3];
    double tq[3] = {(f[1] - f[3]) * c->L, (-f[0] + f[2]) * c->L, -zt[0] + zt[1] - zt[2] + zt[3]};   /* CF2P "+" frame */
    double Jw[3], wJw[3];
    for (int k = 0; k < 3; ++k) Jw[k] = c->J[k] * rates[k];
    cross3(rates, Jw, wJw);
    const double fw[3] = {R[2] * thrust, R[5] * thrust, R[8] * thrust - c->G * c->M};
    for (int k = 0; k < 3; ++k) {
      vel[k] += dt * (fw[k] / c->M);
      rates[k] += dt * ((tq[k] - wJw[k]) / c->J[k]);
    }
    for (int k = 0; k < 3; ++k) pos[k] += dt * vel[k];
    integrate_q(quat, rates, dt);
    matvec3(R, rates, angv);
  }
  for (int k = 0; k < 4; ++k) st[16 + k] = rpm[k];
}

/* [UPSTREAM] _getDroneStateVector: pos3 | quat4 | rpy3 | vel3 | ang_v3 | last clipped action 4 */
static void pack_obs(const double st[CO_STATE], double obs[CO_OBS]) {
  memcpy(obs, st, 7 * sizeof(double));
  euler_from_quat_bullet(st + 3, obs + 7);
  memcpy(obs + 10, st + 7, 3 * sizeof(double));
  memcpy(obs + 13, st + 13, 3 * sizeof(double));
  memcpy(obs + 16, st + 16, 4 * sizeof(double));
}

/* ---- exported entry points (ctypes) ------------------------------------------------------------------------------------------ */

int co_sizeof_consts(void) { return (int)sizeof(co_consts); }

/* [UPSTREAM] cf2p.urdf constants, BaseAviary.__init__ derived values, control/geometric.py:14-23 gains */
void co_default_consts(co_consts* c, int pyb_freq, int ctrl_freq) {
  c->M = 0.027; c->L = 0.0397; c->KF = 3.16e-10; c->KM = 7.94e-12;
  c->J[0] = 2.3951e-5; c->J[1] = 2.3951e-5; c->J[2] = 3.2347e-5;
  c->G = 9.8;
  c->MAX_RPM = sqrt(2.25 * c->G * c->M / (4 * c->KF));
  c->MAX_THRUST = 4 * c->KF * c->MAX_RPM * c->MAX_RPM;
  for (int k = 0; k < 3; ++k) { c->Kp[k] = 2.25; c->Kv[k] = 3.5; c->KR[k] = 125.0; c->Kw[k] = 10.0; }
  c->g_ctrl = 9.81;
  c->max_tilt = 40.0 * M_PI / 180.0;
  c->substeps = pyb_freq / ctrl_freq;
  c->pyb_dt = 1.0 / pyb_freq;
  c->ctrl_dt = 1.0 / ctrl_freq;
}

/* [UPSTREAM] _housekeeping: xyz [n,3], rpy [n,3] -> st [n,20] at rest */
void co_reset(int n, const double* xyz, const double* rpy, double* st) {
  for (int i = 0; i < n; ++i) {
    double* s = st + (size_t)i * CO_STATE;
    memset(s, 0, CO_STATE * sizeof(double));
    memcpy(s, xyz + 3 * i, 3 * sizeof(double));
    quat_from_euler_bullet(rpy + 3 * i, s + 3);
  }
}
void co_lemniscate(int n, double t, const double* P, double* des) {
  for (int i = 0; i < n; ++i) lemniscate(t, P + 7 * i, des + 11 * i);
}
void co_geometric_compute(const co_consts* c, int n, const double* obs, const double* des, double* rpm) {
  for (int i = 0; i < n; ++i) geometric_compute(c, obs + 20 * i, des + 11 * i, rpm + 4 * i);
}
/* env.step(action): st [n,20] in place, obs [n,20] out */
void co_step(const co_consts* c, int n, double* st, const double* action, double* obs) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) if (n >= 4096)
#endif
  for (int i = 0; i < n; ++i) {
    aviary_step(c, st + (size_t)i * CO_STATE, action + 4 * i);
    pack_obs(st + (size_t)i * CO_STATE, obs + (size_t)i * CO_OBS);
  }
}
/* The do_control loop (simulations/EnvGeometric.py:431-473) for n drones: env.step(zeros) first (:431) when first_zero_step, then
 * `steps` control steps from time t0 (t advances by CTRL_TIMESTEP).  The drones do not interact, so the loop runs
 * drone by drone (all steps of one drone, then the next), `threads` OpenMP threads over the drones (<= 0: the OpenMP default).
 * obs_out [n,20]: the last observation.  Returns the number of threads used. */
int co_geometric_loop(const co_consts* c, int n, int steps, double t0, int first_zero_step, const double* P, double* st, double* obs_out,
                      int threads) {
  int used = 1;
  const double dtc = c->ctrl_dt;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
  }
#pragma omp parallel for schedule(static)
#endif
  for (int i = 0; i < n; ++i) {
    double* s = st + (size_t)i * CO_STATE;
    double obs[CO_OBS], des[11], rpm[4];
    const double zero[4] = {0.0, 0.0, 0.0, 0.0};
    if (first_zero_step) aviary_step(c, s, zero);
    pack_obs(s, obs);
    double t = t0;
    for (int k = 0; k < steps; ++k) {
      lemniscate(t, P + 7 * i, des);
      geometric_compute(c, obs, des, rpm);
      aviary_step(c, s, rpm);
      pack_obs(s, obs);
      t += dtc;
    }
    memcpy(obs_out + (size_t)i * CO_OBS, obs, sizeof(obs));
  }
  return used;
}

/* ================================================================================================================================
 * The CBF-filtered loop (BASELINE config 4): simulations/CBFTest.py:303-350 for ONE env of D drones per call of co_cbf_env_step --
 * nominal GeometricControl(return_omegas) -> u_hat = (force - M G, w_des) (:339) -> DroneQPTracker.compute_control (cbf/qptracker.py:
 * 22-34) on x = obs_to_lin_model(obs, 9), xdes = [0, 0, yaw, vel, pos] (:332-336) -> + M G (:346) -> LQROmegaController.
 * compute_low_level -> ThrustOmegaController (control/lqr/lqr_omega_controller.py:77-88, control/low_level/thrust_omega_ctrl.py:81-132)
 * -> env.step.  Order-2 ECBF rows in the closed form of SURVEY.md 3.6 (cbf/cbf.py:135-303 for the hover linearisation of
 * model/linear_omega.py:46-53; pinned by tests/golden/cbf_rows_o2.npz), stacked in the reference's order (cbf/cbf.py:308-367: pairs
 * i < j | +I | -I | obstacles agent-major).  The QP min 1/2 |u - u_hat|^2 s.t. G u <= h is solved EXACTLY by a dual active-set
 * (Goldfarb-Idnani) iteration on the dense rows; an infeasible QP keeps u_hat with status 1 -- the MODELLED fallback of np_oracle.py /
 * DESIGN.md, not the reference's behaviour there (cvxopt returns an iterate).
 * ================================================================================================================================ */
#define CO_MAXD 16
#define CO_MAXOBS 8
#define CO_MAXN (4 * CO_MAXD)
#define CO_MAXM (CO_MAXD * (CO_MAXD - 1) / 2 + 10 * CO_MAXD + CO_MAXD * CO_MAXOBS)

typedef struct {
  double Kcbf[3], umax[4], safety_radius, zscale;       /* cbf/cbf.py:119-124, :566-572; DroneCBF(safety_radius, zscale); Kcbf[2]: order 3 only */
  int n_obs, order;                                     /* order 2 (simulations/CBFTest.py) or 3 (CBFTestOrd3.py) */
  double obs_xyz[CO_MAXOBS][3], obs_r[CO_MAXOBS];       /* x_obs_list[j][0], obs_r_list[j] (simulations/CBFTest.py:421-425) */
  double Fmin, Fmax;                                    /* order 3: thrust-state box, -M G and MAX_THRUST (cbf/cbf.py:564-565) */
} co_cbf;

int co_sizeof_cbf(void) { return (int)sizeof(co_cbf); }

/* one ECBF row between the 9-states xi, xj (rpy, vel, pos) with desired states xdi, xdj: h_ij and L_g (thrust column only for order 2) */
static void cbf_pair_o2(const double* xi, const double* xj, const double* xdi, const double* xdj, double Ds, double zscale, const double K[2],
                        double m, double g, double* hij, double* Lg0) {
  const double c4 = zscale * zscale * zscale * zscale;
  const double ex = xi[6] - xj[6], ey = xi[7] - xj[7], ez = xi[8] - xj[8];
  const double s = ex * ex + ey * ey;
  const double ezc = ez / zscale;
  const double h = s * s + ezc * ezc * ezc * ezc - Ds * Ds * Ds * Ds;
  const double gx = 4 * ex * s, gy = 4 * ey * s, gz = 4 * ez * ez * ez / c4;
  const double Hxx = 12 * ex * ex + 4 * ey * ey, Hxy = 8 * ex * ey, Hyy = 4 * ex * ex + 12 * ey * ey, Hzz = 12 * ez * ez / c4;
  double d[9];
  for (int k = 0; k < 9; ++k) d[k] = (xi[k] - xdi[k]) - (xj[k] - xdj[k]);
  const double dr = d[0], dp = d[1], dvx = d[3], dvy = d[4], dvz = d[5];
  const double dax = g * dp, day = -g * dr;
  const double hdot = gx * dvx + gy * dvy + gz * dvz;
  const double quad = Hxx * dvx * dvx + 2 * Hxy * dvx * dvy + Hyy * dvy * dvy + Hzz * dvz * dvz;
  const double Lf2 = gx * dax + gy * day + quad;
  *hij = K[0] * h + K[1] * hdot + Lf2;
  *Lg0 = gz / m;
}

/* CBF._build_ineq_const (cbf/cbf.py:308-367): x, xdes [D,9] -> G [m, 4D] row-major, h [m]; returns m */
static int cbf_rows_o2(const co_consts* c, const co_cbf* b, int D, const double* x, const double* xdes, double* G, double* h) {
  const int n = 4 * D;
  int m = 0;
  for (int i = 0; i < D - 1; ++i)
    for (int j = i + 1; j < D; ++j) {
      double hij, lg;
      cbf_pair_o2(x + 9 * i, x + 9 * j, xdes + 9 * i, xdes + 9 * j, 2 * b->safety_radius, b->zscale, b->Kcbf, c->M, c->G, &hij, &lg);
      double* row = G + (size_t)m * n;
      memset(row, 0, n * sizeof(double));
      row[4 * i] = -lg;
      row[4 * j] = lg;
      h[m++] = hij;
    }
  for (int sgn = 0; sgn < 2; ++sgn)                                  /* +I then -I, h = umax tiled (:400-412) */
    for (int k = 0; k < n; ++k) {
      double* row = G + (size_t)m * n;
      memset(row, 0, n * sizeof(double));
      row[k] = sgn ? -1.0 : 1.0;
      h[m++] = b->umax[k & 3];
    }
  for (int i = 0; i < D; ++i)                                          /* obstacles, agent-major (:369-398): obstacle state = position only, xj_des = xj */
    for (int j = 0; j < b->n_obs; ++j) {
      double xo[9] = {0, 0, 0, 0, 0, 0, b->obs_xyz[j][0], b->obs_xyz[j][1], b->obs_xyz[j][2]};
      double hij, lg;
      cbf_pair_o2(x + 9 * i, xo, xdes + 9 * i, xo, b->safety_radius + b->obs_r[j], b->zscale, b->Kcbf, c->M, c->G, &hij, &lg);
      double* row = G + (size_t)m * n;
      memset(row, 0, n * sizeof(double));
      row[4 * i] = -lg;
      h[m++] = hij;
    }
  return m;
}

/* solve M r = rhs (q x q, symmetric positive definite up to rounding) by Gaussian elimination with partial pivoting; returns 0 if singular */
static int solve_small(int q, double* M, double* rhs) {
  for (int k = 0; k < q; ++k) {
    int piv = k;
    for (int i = k + 1; i < q; ++i)
      if (fabs(M[i * q + k]) > fabs(M[piv * q + k])) piv = i;
    if (M[piv * q + k] == 0.0) return 0;
    if (piv != k) {
      for (int j = 0; j < q; ++j) { const double t = M[k * q + j]; M[k * q + j] = M[piv * q + j]; M[piv * q + j] = t; }
      const double t = rhs[k]; rhs[k] = rhs[piv]; rhs[piv] = t;
    }
    for (int i = k + 1; i < q; ++i) {
      const double f = M[i * q + k] / M[k * q + k];
      for (int j = k; j < q; ++j) M[i * q + j] -= f * M[k * q + j];
      rhs[i] -= f * rhs[k];
    }
  }
  for (int k = q - 1; k >= 0; --k) {
    double a = rhs[k];
    for (int j = k + 1; j < q; ++j) a -= M[k * q + j] * rhs[j];
    rhs[k] = a / M[k * q + k];
  }
  return 1;
}

/* min 1/2 |u - uhat|^2 s.t. G u <= h: Goldfarb-Idnani dual active set with H = I on unit-norm rows (same feasible set, same minimiser).
 * u [n] in: uhat, out: the minimiser.  Returns 1 solved, 0 infeasible (u untouched then).  *iters: constraint additions + drops. */
static int qp_project(int n, int m, const double* G0, const double* h0, double* u_io, int* iters) {
  static __thread double G[CO_MAXM * CO_MAXN], h[CO_MAXM], sc[CO_MAXM];
  double u[CO_MAXN], lam[CO_MAXN], r[CO_MAXN], z[CO_MAXN], M[CO_MAXN * CO_MAXN];
  int active[CO_MAXN], q = 0;
  const double tol = 1e-10;
  for (int k = 0; k < m; ++k) {
    double rn = 0.0;
    for (int j = 0; j < n; ++j) rn += G0[(size_t)k * n + j] * G0[(size_t)k * n + j];
    rn = sqrt(rn);
    if (rn == 0.0 && h0[k] < 0.0) return 0;                          /* 0 * u <= h with h < 0 */
    const double rs = rn > 0.0 ? rn : 1.0;
    for (int j = 0; j < n; ++j) G[(size_t)k * n + j] = G0[(size_t)k * n + j] / rs;
    h[k] = rn > 0.0 ? h0[k] / rs : INFINITY;
    sc[k] = isfinite(h[k]) ? (fabs(h[k]) > 1.0 ? fabs(h[k]) : 1.0) : 1.0;
  }
  memcpy(u, u_io, n * sizeof(double));
  const int max_iter = 20 * (m + 10);
  *iters = 0;
  for (int it = 0; it < max_iter; ++it) {
    int k = 0;
    double best = -INFINITY, viol_k = 0.0;
    for (int i = 0; i < m; ++i) {                                      /* most violated row, relative to max(1, |h|) */
      double v = -h[i];
      if (isfinite(h[i])) {
        for (int j = 0; j < n; ++j) v += G[(size_t)i * n + j] * u[j];
      } else {
        v = -INFINITY;
      }
      if (v / sc[i] > best) { best = v / sc[i]; k = i; viol_k = v; }
    }
    if (viol_k <= tol * sc[k]) {
      memcpy(u_io, u, n * sizeof(double));
      return 1;
    }
    const double* gk = G + (size_t)k * n;
    double lam_k = 0.0, gkgk = 0.0;
    for (int j = 0; j < n; ++j) gkgk += gk[j] * gk[j];
    for (;;) {                                                         /* bring row k in, dropping blocking rows on the way */
      ++*iters;
      if (q > 0) {
        for (int a = 0; a < q; ++a) {
          const double* ga = G + (size_t)active[a] * n;
          double d = 0.0;
          for (int j = 0; j < n; ++j) d += ga[j] * gk[j];
          r[a] = d;
          for (int b2 = 0; b2 < q; ++b2) {
            const double* gb = G + (size_t)active[b2] * n;
            double e = 0.0;
            for (int j = 0; j < n; ++j) e += ga[j] * gb[j];
            M[a * q + b2] = e;
          }
        }
        if (!solve_small(q, M, r)) return 0;
        for (int j = 0; j < n; ++j) {
          double a = gk[j];
          for (int b2 = 0; b2 < q; ++b2) a -= G[(size_t)active[b2] * n + j] * r[b2];
          z[j] = a;
        }
      } else {
        memcpy(z, gk, n * sizeof(double));
      }
      double zz = 0.0;
      for (int j = 0; j < n; ++j) zz += z[j] * z[j];
      double t1 = INFINITY, t2 = INFINITY;
      int drop = -1;
      for (int a = 0; a < q; ++a)
        if (r[a] > 1e-14) {
          const double cand = lam[a] / r[a];
          if (cand < t1) { t1 = cand; drop = a; }
        }
      if (zz > 1e-18 * (gkgk > 1.0 ? gkgk : 1.0)) {
        double v = -h[k];
        for (int j = 0; j < n; ++j) v += gk[j] * u[j];
        t2 = v / zz;
      }
      const double t = t1 < t2 ? t1 : t2;
      if (!isfinite(t)) return 0;                                      /* rows inconsistent: infeasible */
      if (isinf(t2)) {                                                 /* dual step only */
        for (int a = 0; a < q; ++a) lam[a] -= t * r[a];
        lam_k += t;
      } else {
        for (int j = 0; j < n; ++j) u[j] -= t * z[j];
        for (int a = 0; a < q; ++a) lam[a] -= t * r[a];
        lam_k += t;
        if (t == t2) {
          if (q >= n) return 0;
          active[q] = k;
          lam[q] = lam_k;
          ++q;
          break;
        }
      }
      for (int a = drop; a < q - 1; ++a) { active[a] = active[a + 1]; lam[a] = lam[a + 1]; }
      --q;
      if (*iters > max_iter) return 0;
    }
  }
  return 0;
}

/* control/low_level/thrust_omega_ctrl.py:81-132 through lqr_omega_controller.py:77-88: u = [thrust, w_des (body)], PID memory pid[6] =
 * last_omega3 | integral3 (P = 17500, I = 10, D = 0; PWM2RPM 0.2685 / 4070.3; PWM in [20000, 65535]; torque clip +-3200; CF2P mixer) */
static void thrust_omega_low_level(const co_consts* c, const double u[4], const double obs[20], double dt, double pid[6], double rpm[4]) {
  double R[9], cur[3];
  quat_to_R_scipy(obs + 3, R);
  matTvec3(R, obs + 13, cur);                                          /* world rate -> body frame (:82-86) */
  const double u0 = u[0] < 0.0 ? 0.0 : u[0];
  double pwm_thrust = (sqrt(u0 / (c->KF * 4)) - 4070.3) / 0.2685;
  pwm_thrust = pwm_thrust < 20000.0 ? 20000.0 : (pwm_thrust > 65535.0 ? 65535.0 : pwm_thrust);
  double tq[3];
  for (int k = 0; k < 3; ++k) {
    const double rate_e = -(cur[k] - pid[k]) / dt;
    const double e = u[1 + k] - cur[k];
    pid[k] = cur[k];
    double in = pid[3 + k] - e * dt;                                   /* sic: minus (:117) */
    in = in < -1500.0 ? -1500.0 : (in > 1500.0 ? 1500.0 : in);
    if (k < 2) in = in < -1.0 ? -1.0 : (in > 1.0 ? 1.0 : in);
    pid[3 + k] = in;
    const double t = 17500.0 * e + 10.0 * in + 0.0 * rate_e;
    tq[k] = t < -3200.0 ? -3200.0 : (t > 3200.0 ? 3200.0 : t);
  }
  static const double MIX[4][3] = {{0, -1, -1}, {1, 0, 1}, {0, 1, -1}, {-1, 0, 1}};
  for (int k = 0; k < 4; ++k) {
    double pwm = pwm_thrust + MIX[k][0] * tq[0] + MIX[k][1] * tq[1] + MIX[k][2] * tq[2];
    pwm = pwm < 20000.0 ? 20000.0 : (pwm > 65535.0 ? 65535.0 : pwm);
    rpm[k] = 0.2685 * pwm + 4070.3;
  }
}

/* One control step of one env: st [D,20], pid [D,6], obs [D,20] (current on entry, next on exit), P [D,7].  Returns the status
 * (0 solved, 1 infeasible: modelled fallback) and the solver's iteration count through *iters. */
/* control/lqr/lqr_omega_controller.py:90-119 LQROmegaController.compute(obs, skip_low_level=True): x = obs_to_lin_model(obs, 9); the error in
 * the goal frame as lqr12_compute; u = -K e (K [4,9]), u[0] += M G, cap_u: u[0] clipped to [4 * 9440.3^2 KF, MAX_THRUST] -- the nominal
 * controller simulations/CBFTest.py:290-293 builds by default (controllers[0] = 'lqr') */
static void lqr_omega_compute(const co_consts* c, const double K[36], const double obs[20], const double des[11], double u[4]) {
  const double yd = des[9], cy = cos(yd), sy = sin(yd);
  double e[9];
  e[0] = obs[7];
  e[1] = obs[8];
  const double dy = obs[9] - yd;
  e[2] = atan2(sin(dy), cos(dy));
  const double dv[3] = {obs[10] - des[3], obs[11] - des[4], obs[12] - des[5]};
  const double dp[3] = {obs[0] - des[0], obs[1] - des[1], obs[2] - des[2]};
  e[3] = cy * dv[0] + sy * dv[1]; e[4] = -sy * dv[0] + cy * dv[1]; e[5] = dv[2];
  e[6] = cy * dp[0] + sy * dp[1]; e[7] = -sy * dp[0] + cy * dp[1]; e[8] = dp[2];
  for (int r = 0; r < 4; ++r) {
    double a = 0.0;
    for (int k = 0; k < 9; ++k) a += K[9 * r + k] * e[k];
    u[r] = -a;
  }
  u[0] += c->M * c->G;
  const double lo = 4 * (9440.3 * 9440.3 * c->KF);
  u[0] = u[0] < lo ? lo : (u[0] > c->MAX_THRUST ? c->MAX_THRUST : u[0]);
}

static int cbf_env_step(const co_consts* c, const co_cbf* b, int D, double t, const double* P, double* st, double* pid, double* obs, int* iters,
                        const double* Klqr /* NULL: geometric nominal; [4,9]: LQR-omega nominal */) {
  double x[CO_MAXD * 9], xdes[CO_MAXD * 9], un[CO_MAXN], us[CO_MAXN];
  static __thread double G[CO_MAXM * CO_MAXN], h[CO_MAXM];
  for (int i = 0; i < D; ++i) {
    double des[11], om[4], rpm_unused[4];
    lemniscate(t, P + 7 * i, des);
    if (Klqr) lqr_omega_compute(c, Klqr, obs + 20 * i, des, om);
    else geometric_compute_ex(c, obs + 20 * i, des, rpm_unused, om);
    un[4 * i] = om[0] - c->M * c->G;                                   /* CBFTest.py:339 */
    un[4 * i + 1] = om[1]; un[4 * i + 2] = om[2]; un[4 * i + 3] = om[3];
    const double* o = obs + 20 * i;                                    /* obs_to_lin_model(obs, 9): rpy, vel, pos */
    double* xi = x + 9 * i;
    xi[0] = o[7]; xi[1] = o[8]; xi[2] = o[9]; xi[3] = o[10]; xi[4] = o[11]; xi[5] = o[12]; xi[6] = o[0]; xi[7] = o[1]; xi[8] = o[2];
    double* xd = xdes + 9 * i;                                         /* [0, 0, yaw, vel, pos] (:332-336) */
    xd[0] = 0.0; xd[1] = 0.0; xd[2] = des[9]; xd[3] = des[3]; xd[4] = des[4]; xd[5] = des[5]; xd[6] = des[0]; xd[7] = des[1]; xd[8] = des[2];
  }
  const int n = 4 * D, m = cbf_rows_o2(c, b, D, x, xdes, G, h);
  memcpy(us, un, n * sizeof(double));
  const int ok = qp_project(n, m, G, h, us, iters);
  if (!ok) memcpy(us, un, n * sizeof(double));
  for (int i = 0; i < D; ++i) {
    double u[4] = {us[4 * i] + c->M * c->G, us[4 * i + 1], us[4 * i + 2], us[4 * i + 3]}, rpm[4];   /* :346 */
    thrust_omega_low_level(c, u, obs + 20 * i, c->ctrl_dt, pid + 6 * i, rpm);
    aviary_step(c, st + (size_t)i * CO_STATE, rpm);
    pack_obs(st + (size_t)i * CO_STATE, obs + 20 * i);
  }
  return ok ? 0 : 1;
}

/* dense rows of one env for the tests: x, xdes [D,9] -> G [m,4D], h [m]; returns m */
int co_cbf_rows(const co_consts* c, const co_cbf* b, int D, const double* x, const double* xdes, double* G, double* h) {
  return cbf_rows_o2(c, b, D, x, xdes, G, h);
}
/* the QP alone: returns 1 solved / 0 infeasible */
int co_qp_project(int n, int m, const double* G, const double* h, double* u_io, int* iters) {
  if (n > CO_MAXN || m > CO_MAXM) return -1;
  return qp_project(n, m, G, h, u_io, iters);
}
/* The C4 loop for E envs of D drones: env.step(first action) is the caller's (co_step); st [E*D,20], pid [E*D,6] (zeros at the start),
 * P [E*D,7]; `steps` control steps from t0; status_log [steps,E] (may be NULL), obs_out [E*D,20]; Klqr NULL (geometric nominal) or the
 * [4,9] gain of the LQR-omega nominal.  OpenMP over the envs. */
int co_cbf_loop(const co_consts* c, const co_cbf* b, int E, int D, int steps, double t0, const double* P, double* st, double* pid, double* obs_out,
                int* status_log, long long* iter_total, int threads, const double* Klqr) {
  if (D > CO_MAXD || b->n_obs > CO_MAXOBS) return -1;
  int used = 1;
  long long total = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
  }
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
#endif
  for (int e = 0; e < E; ++e) {
    double obs[CO_MAXD * 20];
    for (int i = 0; i < D; ++i) pack_obs(st + (size_t)(e * D + i) * CO_STATE, obs + 20 * i);
    double t = t0;
    for (int k = 0; k < steps; ++k) {
      int it = 0;
      const int stt = cbf_env_step(c, b, D, t, P + (size_t)e * D * 7, st + (size_t)e * D * CO_STATE, pid + (size_t)e * D * 6, obs, &it, Klqr);
      if (status_log) status_log[(size_t)k * E + e] = stt;
      total += it;
      t += c->ctrl_dt;
    }
    memcpy(obs_out + (size_t)e * D * 20, obs, (size_t)D * 20 * sizeof(double));
  }
  if (iter_total) *iter_total = total;
  return used;
}

/* ================================================================================================================================
 * The scripts' DEFAULT controller (simulations/EnvGeometric.py:32, :425-427, :457): control/lqr/lqr_controller.py:73-113
 * LQRController.compute on model/linearized.py -- x = obs_to_lin_model(obs) (rpy, ang_v, vel, pos); the error in the goal frame
 * (euler of R_eq^T R = (roll, pitch, wrapped yaw - yaw_des) away from the gimbal; positions, velocities and rates rotated by R_eq^T,
 * :92-99); u = -K e, u[0] += M G (:111-112); input_to_action (:113) -- with the wind of :463-467 as a constant world-frame force in
 * env.step.  K [4,12] is the caller's (the host's continuous ARE, like the reference's constructor).
 * ================================================================================================================================ */
static void lqr12_compute(const co_consts* c, const double K[48], const double obs[20], const double des[11], double rpm[4]) {
  const double yd = des[9], cy = cos(yd), sy = sin(yd);
  double e[12];
  e[0] = obs[7];
  e[1] = obs[8];
  const double dy = obs[9] - yd;
  e[2] = atan2(sin(dy), cos(dy));
  const double dw[3] = {obs[13], obs[14], obs[15] - des[10]};
  const double dv[3] = {obs[10] - des[3], obs[11] - des[4], obs[12] - des[5]};
  const double dp[3] = {obs[0] - des[0], obs[1] - des[1], obs[2] - des[2]};
  const double* src[3] = {dw, dv, dp};
  for (int b = 0; b < 3; ++b) {
    e[3 + 3 * b] = cy * src[b][0] + sy * src[b][1];
    e[4 + 3 * b] = -sy * src[b][0] + cy * src[b][1];
    e[5 + 3 * b] = src[b][2];
  }
  double u[4];
  for (int r = 0; r < 4; ++r) {
    double a = 0.0;
    for (int k = 0; k < 12; ++k) a += K[12 * r + k] * e[k];
    u[r] = -a;
  }
  u[0] += c->M * c->G;
  input_to_action(c, u, rpm);
}

/* env.step with a constant world-frame external force (p.applyExternalForce(..., WORLD_FRAME) every control step, :463-467): the same
 * substep as aviary_step with wind / M added to the acceleration */
static void aviary_step_wind(const co_consts* c, double st[CO_STATE], const double action[4], const double wind[3]) {
  double rpm[4];
  for (int k = 0; k < 4; ++k) rpm[k] = action[k] < 0.0 ? 0.0 : (action[k] > c->MAX_RPM ? c->MAX_RPM : action[k]);
  double *pos = st, *quat = st + 3, *vel = st + 7, *rates = st + 10, *angv = st + 13;
  const double dt = c->pyb_dt;
  for (int s = 0; s < c->substeps; ++s) {
    double R[9], f[4], zt[4];
    quat_to_R_bullet(quat, R);
    for (int k = 0; k < 4; ++k) {
      f[k] = rpm[k] * rpm[k] * c->KF;
      zt[k] = rpm[k] * rpm[k] * c->KM;
    }
    const double thrust = f[0] + f[1] + f[2] + f[3];
    double tq[3] = {(f[1] - f[3]) * c->L, (-f[0] + f[2]) * c->L, -zt[0] + zt[1] - zt[2] + zt[3]};
    double Jw[3], wJw[3];
    for (int k = 0; k < 3; ++k) Jw[k] = c->J[k] * rates[k];
    cross3(rates, Jw, wJw);
    const double fw[3] = {R[2] * thrust + wind[0], R[5] * thrust + wind[1], R[8] * thrust - c->G * c->M + wind[2]};
    for (int k = 0; k < 3; ++k) {
      vel[k] += dt * (fw[k] / c->M);
      rates[k] += dt * ((tq[k] - wJw[k]) / c->J[k]);
    }
    for (int k = 0; k < 3; ++k) pos[k] += dt * vel[k];
    integrate_q(quat, rates, dt);
    matvec3(R, rates, angv);
  }
  for (int k = 0; k < 4; ++k) st[16 + k] = rpm[k];
}

void co_lqr12_compute(const co_consts* c, int n, const double* K, const double* obs, const double* des, double* rpm) {
  for (int i = 0; i < n; ++i) lqr12_compute(c, K, obs + 20 * i, des + 11 * i, rpm + 4 * i);
}
/* The do_control loop with controllers[0] = 'lqr' and the wind on from the first control step (EnvGeometric.py:431-473): as
 * co_geometric_loop.  wind [3] (newtons, world frame) or NULL. */
int co_lqr_loop(const co_consts* c, int n, int steps, double t0, int first_zero_step, const double* P, const double* K, const double* wind,
                double* st, double* obs_out, int threads) {
  int used = 1;
  const double w0[3] = {wind ? wind[0] : 0.0, wind ? wind[1] : 0.0, wind ? wind[2] : 0.0};
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
  }
#pragma omp parallel for schedule(static)
#endif
  for (int i = 0; i < n; ++i) {
    double* s = st + (size_t)i * CO_STATE;
    double obs[CO_OBS], des[11], rpm[4];
    const double zero[4] = {0.0, 0.0, 0.0, 0.0};
    if (first_zero_step) aviary_step(c, s, zero);                    /* :431, before the wind is on */
    pack_obs(s, obs);
    double t = t0;
    for (int k = 0; k < steps; ++k) {
      lemniscate(t, P + 7 * i, des);
      lqr12_compute(c, K, obs, des, rpm);
      aviary_step_wind(c, s, rpm, w0);
      pack_obs(s, obs);
      t += c->ctrl_dt;
    }
    memcpy(obs_out + (size_t)i * CO_OBS, obs, sizeof(obs));
  }
  return used;
}

/* ================================================================================================================================
 * The order-3 loop (simulations/CBFTestOrd3.py:306-352): LQRYankOmegaController nominal (control/lqr/lqr_YO_controller.py:99-124) on the
 * 10-state x = obs_to_lin_model(obs, 10) = [rpy, F = calc_z_thrust(obs), vel, pos]; u_hat = (yank - M G, w) (:341: the M G is subtracted
 * from the YANK and never added back, :350 -- kept); xdes = [0, 0, yaw, G M, vel, pos]; order-3 ECBF rows (cbf/cbf.py:135-283 with the
 * slot quirk of custom_hdots :158-169, closed form of SURVEY.md 3.6, pinned by cbf_rows_o3.npz) incl. the thrust-state box rows with
 * their column quirk (:446-464: column 4 i + 3); YankOmegaController low level (control/low_level/yank_omega_ctrl.py:39-55:
 * thrust = calc_z_thrust(obs) + yank * dt, then the ThrustOmega PID).
 * ================================================================================================================================ */
static void cbf_pair_o3(const double* xi, const double* xj, const double* xdi, const double* xdj, double Ds, double zscale, const double K[3],
                        double m, double g, double* hij, double Lg[4]) {
  const double c4 = zscale * zscale * zscale * zscale;
  const double ex = xi[7] - xj[7], ey = xi[8] - xj[8], ez = xi[9] - xj[9];
  const double s = ex * ex + ey * ey, ezc = ez / zscale;
  const double h = s * s + ezc * ezc * ezc * ezc - Ds * Ds * Ds * Ds;
  const double gx = 4 * ex * s, gy = 4 * ey * s, gz = 4 * ez * ez * ez / c4;
  const double Hxx = 12 * ex * ex + 4 * ey * ey, Hxy = 8 * ex * ey, Hyy = 4 * ex * ex + 12 * ey * ey, Hzz = 12 * ez * ez / c4;
  double d[10];
  for (int k = 0; k < 10; ++k) d[k] = (xi[k] - xdi[k]) - (xj[k] - xdj[k]);
  const double dr = d[0], dp = d[1], dF = d[3], dvx = d[4], dvy = d[5], dvz = d[6];
  const double dax = g * dp, day = -g * dr, daz = dF / m;
  const double hdot = gx * dvx + gy * dvy + gz * dvz;
  const double w0 = dF / m, w1 = dvx, w2 = dvy;                    /* custom_hdots i == 2: slots 6, 7, 8 of the 10-state (quirk kept) */
  const double hddot_ref = g * (gy * dp - gz * dr) + (Hxx * w0 * w0 + 2 * Hxy * w0 * w1 + Hyy * w1 * w1 + Hzz * w2 * w2);
  const double Hdv_da = Hxx * dvx * dax + Hxy * (dvx * day + dvy * dax) + Hyy * dvy * day + Hzz * dvz * daz;
  const double T3 = 24 * ex * dvx * dvx * dvx + 24 * ey * dvx * dvx * dvy + 24 * ex * dvx * dvy * dvy + 24 * ey * dvy * dvy * dvy +
                    (24 * ez / c4) * dvz * dvz * dvz;
  *hij = K[0] * h + K[1] * hdot + K[2] * hddot_ref + (3 * Hdv_da + T3);
  Lg[0] = gz / m; Lg[1] = -g * gy; Lg[2] = g * gx; Lg[3] = 0.0;
}

static int cbf_rows_o3(const co_consts* c, const co_cbf* b, int D, const double* x, const double* xdes, double* G, double* h) {
  const int n = 4 * D;
  int m = 0;
  for (int i = 0; i < D - 1; ++i)
    for (int j = i + 1; j < D; ++j) {
      double hij, lg[4];
      cbf_pair_o3(x + 10 * i, x + 10 * j, xdes + 10 * i, xdes + 10 * j, 2 * b->safety_radius, b->zscale, b->Kcbf, c->M, c->G, &hij, lg);
      double* row = G + (size_t)m * n;
      memset(row, 0, n * sizeof(double));
      for (int k = 0; k < 4; ++k) { row[4 * i + k] = -lg[k]; row[4 * j + k] = lg[k]; }
      h[m++] = hij;
    }
  for (int sgn = 0; sgn < 2; ++sgn)
    for (int k = 0; k < n; ++k) {
      double* row = G + (size_t)m * n;
      memset(row, 0, n * sizeof(double));
      row[k] = sgn ? -1.0 : 1.0;
      h[m++] = b->umax[k & 3];
    }
  for (int i = 0; i < D; ++i)                                          /* thrust-state box (:446-464), column 4 i + 3 (quirk) */
    for (int sgn = 0; sgn < 2; ++sgn) {
      double* row = G + (size_t)m * n;
      memset(row, 0, n * sizeof(double));
      row[4 * i + 3] = sgn ? -1.0 : 1.0;
      h[m++] = sgn ? b->Kcbf[2] * (x[10 * i + 3] - b->Fmin) : b->Kcbf[2] * (b->Fmax - x[10 * i + 3]);
    }
  for (int i = 0; i < D; ++i)
    for (int j = 0; j < b->n_obs; ++j) {
      double xo[10] = {0, 0, 0, 0, 0, 0, 0, b->obs_xyz[j][0], b->obs_xyz[j][1], b->obs_xyz[j][2]};
      double hij, lg[4];
      cbf_pair_o3(x + 10 * i, xo, xdes + 10 * i, xo, b->safety_radius + b->obs_r[j], b->zscale, b->Kcbf, c->M, c->G, &hij, lg);
      double* row = G + (size_t)m * n;
      memset(row, 0, n * sizeof(double));
      for (int k = 0; k < 4; ++k) row[4 * i + k] = -lg[k];
      h[m++] = hij;
    }
  return m;
}

static void lqr_yank_omega_compute(const co_consts* c, const double K[40], const double x[10], const double des[11], double u[4]) {
  const double yd = des[9], cy = cos(yd), sy = sin(yd);
  double e[10];
  e[0] = x[0];
  e[1] = x[1];
  const double dy = x[2] - yd;
  e[2] = atan2(sin(dy), cos(dy));
  e[3] = x[3] - c->M * c->G;
  const double dv[3] = {x[4] - des[3], x[5] - des[4], x[6] - des[5]};
  const double dp[3] = {x[7] - des[0], x[8] - des[1], x[9] - des[2]};
  e[4] = cy * dv[0] + sy * dv[1]; e[5] = -sy * dv[0] + cy * dv[1]; e[6] = dv[2];
  e[7] = cy * dp[0] + sy * dp[1]; e[8] = -sy * dp[0] + cy * dp[1]; e[9] = dp[2];
  for (int r = 0; r < 4; ++r) {
    double a = 0.0;
    for (int k = 0; k < 10; ++k) a += K[10 * r + k] * e[k];
    u[r] = -a;
  }
}

static int cbf3_env_step(const co_consts* c, const co_cbf* b, int D, double t, const double* P, double* st, double* pid, double* obs, int* iters,
                         const double* Kyo) {
  double x[CO_MAXD * 10], xdes[CO_MAXD * 10], un[CO_MAXN], us[CO_MAXN];
  static __thread double G[CO_MAXM * CO_MAXN], h[CO_MAXM];
  for (int i = 0; i < D; ++i) {
    double des[11];
    const double* o = obs + 20 * i;
    lemniscate(t, P + 7 * i, des);
    double* xi = x + 10 * i;                                           /* obs_to_lin_model(obs, 10): rpy, F = KF sum rpm^2, vel, pos */
    xi[0] = o[7]; xi[1] = o[8]; xi[2] = o[9];
    xi[3] = c->KF * (o[16] * o[16] + o[17] * o[17] + o[18] * o[18] + o[19] * o[19]);
    xi[4] = o[10]; xi[5] = o[11]; xi[6] = o[12]; xi[7] = o[0]; xi[8] = o[1]; xi[9] = o[2];
    lqr_yank_omega_compute(c, Kyo, xi, des, un + 4 * i);
    un[4 * i] -= c->M * c->G;                                          /* CBFTestOrd3.py:341 */
    double* xd = xdes + 10 * i;                                        /* [0, 0, yaw, G M, vel, pos] */
    xd[0] = 0.0; xd[1] = 0.0; xd[2] = des[9]; xd[3] = c->G * c->M; xd[4] = des[3]; xd[5] = des[4]; xd[6] = des[5]; xd[7] = des[0]; xd[8] = des[1];
    xd[9] = des[2];
  }
  const int n = 4 * D, m = cbf_rows_o3(c, b, D, x, xdes, G, h);
  memcpy(us, un, n * sizeof(double));
  const int ok = qp_project(n, m, G, h, us, iters);
  if (!ok) memcpy(us, un, n * sizeof(double));
  for (int i = 0; i < D; ++i) {
    const double thrust = x[10 * i + 3] + us[4 * i] * c->ctrl_dt;      /* yank2thrust (yank_omega_ctrl.py:49-53); nothing added back (:350) */
    const double u[4] = {thrust, us[4 * i + 1], us[4 * i + 2], us[4 * i + 3]};
    double rpm[4];
    thrust_omega_low_level(c, u, obs + 20 * i, c->ctrl_dt, pid + 6 * i, rpm);
    aviary_step(c, st + (size_t)i * CO_STATE, rpm);
    pack_obs(st + (size_t)i * CO_STATE, obs + 20 * i);
  }
  return ok ? 0 : 1;
}

int co_cbf_rows3(const co_consts* c, const co_cbf* b, int D, const double* x, const double* xdes, double* G, double* h) {
  return cbf_rows_o3(c, b, D, x, xdes, G, h);
}
/* as co_cbf_loop, order 3: Kyo [4,10] the yank-omega LQR gain; the caller has stepped the env once with hover RPM (the thrust state is
 * read from the observation's RPM echo) */
int co_cbf3_loop(const co_consts* c, const co_cbf* b, int E, int D, int steps, double t0, const double* P, double* st, double* pid, double* obs_out,
                 int* status_log, long long* iter_total, int threads, const double* Kyo) {
  if (D > CO_MAXD || b->n_obs > CO_MAXOBS || !Kyo) return -1;
  int used = 1;
  long long total = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
  {
#pragma omp single
    used = omp_get_num_threads();
  }
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
#endif
  for (int e = 0; e < E; ++e) {
    double obs[CO_MAXD * 20];
    for (int i = 0; i < D; ++i) pack_obs(st + (size_t)(e * D + i) * CO_STATE, obs + 20 * i);
    double t = t0;
    for (int k = 0; k < steps; ++k) {
      int it = 0;
      const int stt = cbf3_env_step(c, b, D, t, P + (size_t)e * D * 7, st + (size_t)e * D * CO_STATE, pid + (size_t)e * D * 6, obs, &it, Kyo);
      if (status_log) status_log[(size_t)k * E + e] = stt;
      total += it;
      t += c->ctrl_dt;
    }
    memcpy(obs_out + (size_t)e * D * 20, obs, (size_t)D * 20 * sizeof(double));
  }
  if (iter_total) *iter_total = total;
  return used;
}
