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
static void geometric_compute(const co_consts* c, const double obs[20], const double des[11], double rpm[4]) {
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
