// Per-drone arithmetic of the batched multi-drone step, templated on the compute type T
// (float for the fp32 / fp16-storage paths, double for the f64 verification path).
//
// Every function here runs inside the HIP kernels of mds_kernels.hip (one drone per
// lane).  The same header also compiles with plain g++ so that tests/emul can run the
// fp32 arithmetic on the CPU for precision studies and sanitizer runs -- that build is
// test tooling only and is never loaded by the multidronesim_amd package.
//
// Reference lines restated (paths relative to the reference checkout):
//   [UPSTREAM] gym_pybullet_drones BaseAviary._dynamics/_integrateQ/_drag and
//   pybullet getEulerFromQuaternion/getMatrixFromQuaternion  -- spec in SURVEY.md 3.4
//   trajectories/Lemniscate.py:32-63
//   utils/model_conversions.py:69-114
//   control/geometric.py:59-115
//   model/dynamics.py:83-106
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MDS_HD __host__ __device__ __forceinline__
#else
#define MDS_HD inline
#endif

namespace mds {

// ------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------
// fp32 reciprocal / square root / reciprocal square root: on the device the 1-ulp hardware
// instructions (v_rcp_f32, v_sqrt_f32, v_rsq_f32) -- the IEEE-exact expansions hipcc emits for
// `/` and sqrtf cost 12 / 18 VALU ops each and the fused kernel needs ~30 of them per drone.
// All operands here are far from the denormal range.  double keeps exact IEEE operations.
#if defined(__HIP_DEVICE_COMPILE__)
MDS_HD float m_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
MDS_HD float m_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
MDS_HD float m_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#else
MDS_HD float m_rcp(float x) { return 1.0f / x; }
MDS_HD float m_sqrt(float x) { return sqrtf(x); }
MDS_HD float m_rsqrt(float x) { return 1.0f / sqrtf(x); }
#endif
MDS_HD double m_rcp(double x) { return 1.0 / x; }
MDS_HD double m_sqrt(double x) { return sqrt(x); }
MDS_HD double m_rsqrt(double x) { return 1.0 / sqrt(x); }
MDS_HD float m_fma(float a, float b, float c) { return fmaf(a, b, c); }
MDS_HD double m_fma(double a, double b, double c) { return fma(a, b, c); }
MDS_HD float m_rint(float x) { return rintf(x); }
MDS_HD double m_rint(double x) { return rint(x); }
MDS_HD float m_exp(float x) { return expf(x); }
MDS_HD double m_exp(double x) { return exp(x); }
MDS_HD float m_abs(float x) { return fabsf(x); }
MDS_HD double m_abs(double x) { return fabs(x); }
MDS_HD double m_atan2(double y, double x) { return atan2(y, x); }
// fp32 atan2: octant reduction to a = min/max in [0,1], one near-minimax odd polynomial of
// degree 17 (max error 1e-7 rad incl. fp32 evaluation), ~25 VALU ops vs ocml's ~45.
MDS_HD float m_atan2(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  const float a = mx > 0.0f ? mn * m_rcp(mx) : 0.0f;
  const float z = a * a;
  float p = 2.4567161705e-03f;
  p = fmaf(p, z, -1.4401325081e-02f);
  p = fmaf(p, z, 3.9781171302e-02f);
  p = fmaf(p, z, -7.2348530072e-02f);
  p = fmaf(p, z, 1.0498944039e-01f);
  p = fmaf(p, z, -1.4161228666e-01f);
  p = fmaf(p, z, 1.9985906696e-01f);
  p = fmaf(p, z, -3.3332597024e-01f);
  p = fmaf(p, z, 9.9999988638e-01f);
  float r = p * a;
  r = ay > ax ? 1.57079632679489661923f - r : r;
  r = x < 0.0f ? 3.14159265358979323846f - r : r;
  return copysignf(r, y);
}
MDS_HD float m_asin(float x) { return asinf(x); }
MDS_HD double m_asin(double x) { return asin(x); }
template <typename T> MDS_HD T m_min(T a, T b) { return a < b ? a : b; }
template <typename T> MDS_HD T m_max(T a, T b) { return a > b ? a : b; }
template <typename T> MDS_HD T m_clamp(T x, T lo, T hi) { return m_min(m_max(x, lo), hi); }

MDS_HD void m_sincos(double x, double* s, double* c) {
  *s = sin(x);
  *c = cos(x);
}

// sin/cos of an argument ALREADY reduced to about [-pi, pi] (callers reduce phases in
// double): one more Cody-Waite step to [-pi/4, pi/4] and two short minimax polynomials.
// Max error < 1.5 ulp on the reduced range; ~25 VALU ops instead of ocml's generic path.
MDS_HD void m_sincos(float x, float* s, float* c) {
  const float k = rintf(x * 0.636619772367581343f);  // x * 2/pi
  // r = x - k*pi/2 with pi/2 split in three parts (exact for |k| <= 4)
  float r = fmaf(k, -1.57079601287841796875f, x);
  r = fmaf(k, -3.1391647326017846353352069854736328125e-7f, r);
  r = fmaf(k, -5.390302529957764765544e-15f, r);
  const float r2 = r * r;
  // sin(r) ~ r + r^3 * P(r^2), cos(r) ~ 1 - r^2/2 + r^4 * Q(r^2)
  float ps = fmaf(r2, 2.6083159809786593541503e-06f, -1.981069071916863322258e-04f);
  ps = fmaf(ps, r2, 8.333078585565090179443e-03f);
  ps = fmaf(ps, r2, -1.666665971279144287109e-01f);
  const float sr = fmaf(ps * r2, r, r);
  float pc = fmaf(r2, 2.443315711809948e-05f, -1.388731625493765e-03f);
  pc = fmaf(pc, r2, 4.166664568298827e-02f);
  const float cr = fmaf(pc * r2, r2, fmaf(r2, -0.5f, 1.0f));
  const int q = (int)k & 3;
  const float ss = (q & 1) ? cr : sr;
  const float cc = (q & 1) ? sr : cr;
  *s = (q & 2) ? -ss : ss;
  *c = ((q + 1) & 2) ? -cc : cc;
}

// phase = a*t + b reduced to [-pi, pi], formed in double so that a 30 s horizon does not
// cost the fp32 path 1e-6 rad of phase (t is passed as double through the C-ABI).
template <typename T> MDS_HD T reduced_phase(double t, T a, T b) {
  const double ph = fma(t, (double)a, (double)b);
  const double k = rint(ph * 0.15915494309189533577);  // 1/(2 pi)
  return (T)fma(k, -6.283185307179586476925, ph);
}
template <> MDS_HD double reduced_phase<double>(double t, double a, double b) { return t * a + b; }

template <typename T> struct V3 {
  T x, y, z;
};
template <typename T> MDS_HD V3<T> v3(T x, T y, T z) { return V3<T>{x, y, z}; }
template <typename T> MDS_HD V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> MDS_HD V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> MDS_HD V3<T> operator*(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> MDS_HD V3<T> hadamard(V3<T> a, V3<T> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename T> MDS_HD T dot(V3<T> a, V3<T> b) { return m_fma(a.x, b.x, m_fma(a.y, b.y, a.z * b.z)); }
template <typename T> MDS_HD V3<T> cross(V3<T> a, V3<T> b) {
  return {m_fma(a.y, b.z, -(a.z * b.y)), m_fma(a.z, b.x, -(a.x * b.z)), m_fma(a.x, b.y, -(a.y * b.x))};
}
template <typename T> MDS_HD T norm(V3<T> a) { return m_sqrt(dot(a, a)); }

// row-major 3x3
template <typename T> struct M3 {
  T m[9];
};
template <typename T> MDS_HD V3<T> mul(const M3<T>& R, V3<T> v) {
  return {m_fma(R.m[0], v.x, m_fma(R.m[1], v.y, R.m[2] * v.z)), m_fma(R.m[3], v.x, m_fma(R.m[4], v.y, R.m[5] * v.z)),
          m_fma(R.m[6], v.x, m_fma(R.m[7], v.y, R.m[8] * v.z))};
}
template <typename T> MDS_HD V3<T> mulT(const M3<T>& R, V3<T> v) {
  return {m_fma(R.m[0], v.x, m_fma(R.m[3], v.y, R.m[6] * v.z)), m_fma(R.m[1], v.x, m_fma(R.m[4], v.y, R.m[7] * v.z)),
          m_fma(R.m[2], v.x, m_fma(R.m[5], v.y, R.m[8] * v.z))};
}
template <typename T> MDS_HD V3<T> col(const M3<T>& R, int j) { return {R.m[j], R.m[3 + j], R.m[6 + j]}; }

// ------------------------------------------------------------------------------------
// constants handed to every kernel by value (host fills them in double, see mds_api)
// ------------------------------------------------------------------------------------
template <typename T> struct Consts {
  // [UPSTREAM] BaseAviary.__init__ / urdf
  T kf, km, arm, mass, inv_mass, gravity /* M*G */, max_rpm;
  T hover_rpm;    // sqrt(MG / 4KF) rounded to T
  T thrust_corr;  // 4*KF*hover_rpm^2 - M*G (rounding of hover_rpm, computed in double)
  T J[3], invJ[3], drag[3];
  T wind[3];      // constant external world force on every drone (EnvGeometric.py:34,463-467), N
  T dt;           // PYB_TIMESTEP
  int substeps;   // PYB_FREQ / CTRL_FREQ
  int cf2x;       // 1: X frame torques, 0: + frame (cf2p)
  int use_drag;   // physics == DYN_DRAG
  int rk4;        // integrator
  // control/geometric.py:14-23 and utils/model_conversions.py:85-103
  T kp[3], kv[3], kR[3], kw[3];
  T g_ctrl;                       // 9.81 (sic, env.G is 9.8)
  T cos_max_tilt, tan_max_tilt;   // 40 deg
  T min_motor_thrust;             // 9440.3^2 * KF
  T max_motor_thrust;             // env.MAX_THRUST used per motor (sic)
  T inv_2L, inv_4r;               // closed-form inverse of the "+" mixer, r = KM/KF
  T inv_kf;
};

template <typename T> struct State {
  V3<T> p;       // world position
  T q[4];        // xyzw
  V3<T> v;       // world velocity
  V3<T> w;       // body rates (upstream rpy_rates)
};

// [UPSTREAM] p.getMatrixFromQuaternion (btMatrix3x3::setRotation, s = 2/|q|^2).  Scale
// invariant, so it also equals scipy Rotation.from_quat(q).as_matrix() used by
// model_conversions.py:110.
template <typename T> MDS_HD M3<T> quat_to_rot(const T q[4]) {
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T d = m_fma(x, x, m_fma(y, y, m_fma(z, z, w * w)));
  const T s = T(2) * m_rcp(d);
  const T xs = x * s, ys = y * s, zs = z * s;
  const T wx = w * xs, wy = w * ys, wz = w * zs;
  const T xx = x * xs, xy = x * ys, xz = x * zs;
  const T yy = y * ys, yz = y * zs, zz = z * zs;
  M3<T> R;
  R.m[0] = T(1) - (yy + zz);
  R.m[1] = xy - wz;
  R.m[2] = xz + wy;
  R.m[3] = xy + wz;
  R.m[4] = T(1) - (xx + zz);
  R.m[5] = yz - wx;
  R.m[6] = xz - wy;
  R.m[7] = yz + wx;
  R.m[8] = T(1) - (xx + yy);
  return R;
}

// R(q) v for a (near-)unit quaternion without forming R: v + 2 q_w (q_v x v) + 2 q_v x (q_v x v),
// scaled by 1/|q|^2 like quat_to_rot.
template <typename T> MDS_HD V3<T> quat_rotate(const T q[4], V3<T> v) {
  const V3<T> qv = {q[0], q[1], q[2]};
  const T s = T(2) * m_rcp(m_fma(q[0], q[0], m_fma(q[1], q[1], m_fma(q[2], q[2], q[3] * q[3]))));
  const V3<T> t = s * cross(qv, v);
  const V3<T> u = cross(qv, t);
  return {m_fma(q[3], t.x, v.x) + u.x, m_fma(q[3], t.y, v.y) + u.y, m_fma(q[3], t.z, v.z) + u.z};
}

// [UPSTREAM] p.getQuaternionFromEuler (btQuaternion::setEulerZYX)
template <typename T> MDS_HD void quat_from_euler(T roll, T pitch, T yaw, T q[4]) {
  T sr, cr, sp, cp, sy, cy;
  m_sincos(roll * T(0.5), &sr, &cr);
  m_sincos(pitch * T(0.5), &sp, &cp);
  m_sincos(yaw * T(0.5), &sy, &cy);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}

// [UPSTREAM] p.getEulerFromQuaternion incl. the +-0.99999 gimbal branches
template <typename T> MDS_HD V3<T> euler_from_quat(const T q[4]) {
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T sqx = x * x, sqy = y * y, sqz = z * z, squ = w * w;
  const T sarg = T(-2) * (x * z - w * y);
  V3<T> rpy;
  if (sarg <= T(-0.99999)) {
    rpy = {T(0), T(-1.57079632679489661923), T(2) * m_atan2(x, -y)};
  } else if (sarg >= T(0.99999)) {
    rpy = {T(0), T(1.57079632679489661923), T(2) * m_atan2(-x, y)};
  } else {
    rpy.x = m_atan2(T(2) * (y * z + w * x), squ - sqx - sqy + sqz);
    rpy.y = m_asin(sarg);
    rpy.z = m_atan2(T(2) * (x * y + w * z), squ + sqx - sqy - sqz);
  }
  return rpy;
}

// [UPSTREAM] _dynamics: thrust (body z) and body torques from 4 clipped RPM.
// Same quantities, conditioned for fp32: differences of squares are formed as
// (a-b)(a+b) (a-b is exact for nearby RPM), and the thrust is returned as its EXCESS over
// the weight, T - M*G = KF*sum(rpm_i^2 - hover^2) + corr, so that near hover neither the
// torques nor the net vertical force lose digits to cancellation.
template <typename T> MDS_HD T dsq(T a, T b) { return (a - b) * (a + b); }
template <typename T> MDS_HD void rotor_wrench(const Consts<T>& c, const T rpm[4], T* thrust_excess, V3<T>* tau) {
  const T h = c.hover_rpm;
  *thrust_excess = m_fma(c.kf, (dsq(rpm[0], h) + dsq(rpm[1], h)) + (dsq(rpm[2], h) + dsq(rpm[3], h)), c.thrust_corr);
  tau->z = c.km * (dsq(rpm[1], rpm[0]) + dsq(rpm[3], rpm[2]));       // -z0 + z1 - z2 + z3
  if (c.cf2x) {
    const T l = c.arm * T(0.70710678118654752440) * c.kf;
    tau->x = l * (dsq(rpm[0], rpm[2]) + dsq(rpm[1], rpm[3]));        // (f0 + f1 - f2 - f3) L/sqrt2
    tau->y = l * (dsq(rpm[1], rpm[0]) + dsq(rpm[2], rpm[3]));        // (-f0 + f1 + f2 - f3) L/sqrt2
  } else {
    const T l = c.arm * c.kf;
    tau->x = l * dsq(rpm[1], rpm[3]);                                 // (f1 - f3) L
    tau->y = l * dsq(rpm[2], rpm[0]);                                 // (-f0 + f2) L
  }
}

// linear + angular acceleration of the rigid body (shared by Euler and RK4).
// force_world = R [0,0,T] - [0,0,MG] with T = MG + excess:  f_z = (R33 - 1) T + excess.
template <typename T, bool DRAG>
MDS_HD void body_accel(const Consts<T>& c, const T q[4], V3<T> vel, V3<T> w, T thrust_excess, V3<T> tau, const T drag_s,
                       V3<T>* acc, V3<T>* wdot) {
  const T x = q[0], y = q[1], z = q[2], ww = q[3];
  const T s = T(2) * m_rcp(m_fma(x, x, m_fma(y, y, m_fma(z, z, ww * ww))));
  const T thrust = c.gravity + thrust_excess;
  const T r02 = s * m_fma(x, z, ww * y), r12 = s * m_fma(y, z, -(ww * x)), r22m1 = -s * m_fma(x, x, y * y);
  V3<T> f = {m_fma(r02, thrust, c.wind[0]), m_fma(r12, thrust, c.wind[1]), m_fma(r22m1, thrust, thrust_excess) + c.wind[2]};
  if (DRAG) {  // [UPSTREAM] _drag: world force -c (.) sum(2 pi rpm_prev/60) (.) v_world
    f.x = m_fma(-c.drag[0] * drag_s, vel.x, f.x);
    f.y = m_fma(-c.drag[1] * drag_s, vel.y, f.y);
    f.z = m_fma(-c.drag[2] * drag_s, vel.z, f.z);
  }
  *acc = c.inv_mass * f;
  const V3<T> Jw = {c.J[0] * w.x, c.J[1] * w.y, c.J[2] * w.z};
  const V3<T> t = tau - cross(w, Jw);
  *wdot = {t.x * c.invJ[0], t.y * c.invJ[1], t.z * c.invJ[2]};
}

// [UPSTREAM] _integrateQ (exact exponential for a constant body rate); identity for
// |omega| <= 1e-8 (np.isclose default atol).  Re-normalised: Bullet stores unit quats.
template <typename T> MDS_HD void integrate_q(T q[4], V3<T> w, T dt) {
  const T w2 = dot(w, w);
  if (w2 <= T(1e-16)) return;                 // |omega| <= 1e-8
  const T inv_wn = m_rsqrt(w2);
  const T wn = w2 * inv_wn;
  T st, ct;
  m_sincos(wn * dt * T(0.5), &st, &ct);
  const T k = st * inv_wn;
  const T x = q[0], y = q[1], z = q[2], ww = q[3];
  const T p = w.x, qq = w.y, r = w.z;
  T nx = m_fma(ct, x, k * (r * y - qq * z + p * ww));
  T ny = m_fma(ct, y, k * (-r * x + p * z + qq * ww));
  T nz = m_fma(ct, z, k * (qq * x - p * y + r * ww));
  T nw = m_fma(ct, ww, k * (-p * x - qq * y - r * z));
  const T inv = m_rsqrt(m_fma(nx, nx, m_fma(ny, ny, m_fma(nz, nz, nw * nw))));
  q[0] = nx * inv;
  q[1] = ny * inv;
  q[2] = nz * inv;
  q[3] = nw * inv;
}

// [UPSTREAM] _dynamics, Physics.DYN: explicit Euler on (v, omega); p with NEW v, q with NEW omega
template <typename T, bool DRAG>
MDS_HD void step_euler_wrench(const Consts<T>& c, State<T>& s, T thrust, V3<T> tau, T drag_s) {
  V3<T> acc, wdot;
  body_accel<T, DRAG>(c, s.q, s.v, s.w, thrust, tau, drag_s, &acc, &wdot);
  s.v = {m_fma(c.dt, acc.x, s.v.x), m_fma(c.dt, acc.y, s.v.y), m_fma(c.dt, acc.z, s.v.z)};
  s.w = {m_fma(c.dt, wdot.x, s.w.x), m_fma(c.dt, wdot.y, s.w.y), m_fma(c.dt, wdot.z, s.w.z)};
  s.p = {m_fma(c.dt, s.v.x, s.p.x), m_fma(c.dt, s.v.y, s.p.y), m_fma(c.dt, s.v.z, s.p.z)};
  integrate_q(s.q, s.w, c.dt);
}
template <typename T, bool DRAG> MDS_HD void step_euler(const Consts<T>& c, State<T>& s, const T rpm[4], T drag_s) {
  T thrust;
  V3<T> tau;
  rotor_wrench(c, rpm, &thrust, &tau);
  step_euler_wrench<T, DRAG>(c, s, thrust, tau, drag_s);
}

// ------------------------------------------------------------------------------------
// Compensated state accumulation (storage dtype MDS_F32C: fp32 arithmetic, every one of the 13 state components carried as an
// fp32 value plus an fp32 residual).  The integrators only ever ADD small increments to the state; stored in plain fp32 each
// addition rounds at the state's magnitude (6e-8 relative) and an uncontrolled quadrotor integrates that four times (rates ->
// attitude -> velocity -> position: ~t^2.5).  Here the rounding error of every addition goes into the residual (two-sum) and is
// fed back into the next one, so what is left is the rounding of the increments themselves (~1e-7 of a step's change).
// Same update order and formulas as step_euler_wrench / step_rk4; the quaternion update is written in its additive form.
// ------------------------------------------------------------------------------------
template <typename T> struct Resid {
  V3<T> p;
  T q[4];
  V3<T> v, w;
};
template <typename T> MDS_HD void resid_zero(Resid<T>& r) {
  r.p = r.v = r.w = {T(0), T(0), T(0)};
  for (int i = 0; i < 4; ++i) r.q[i] = T(0);
}
// s + r += d: s gets the rounded sum, r what the rounding dropped (Knuth two-sum, no magnitude assumption)
template <typename T> MDS_HD void comp_add(T& s, T& r, T d) {
  const T y = d + r;
  const T t = s + y;
  const T bp = t - s;
  r = (s - (t - bp)) + (y - bp);
  s = t;
}
template <typename T> MDS_HD void comp_add3(V3<T>& s, V3<T>& r, V3<T> d) {
  comp_add(s.x, r.x, d.x);
  comp_add(s.y, r.y, d.y);
  comp_add(s.z, r.z, d.z);
}
// unit-norm pull in additive form: q += q (1 - |q|^2) / 2 (the first-order step of the re-normalisation integrate_q does;
// |q|^2 - 1 stays ~1e-7, so the second-order term is below 1e-14).  A radial error does not turn the attitude.
template <typename T> MDS_HD void comp_renorm(T q[4], T rq[4]) {
  T n2m1 = m_fma(q[0], q[0], m_fma(q[1], q[1], m_fma(q[2], q[2], m_fma(q[3], q[3], T(-1)))));
  n2m1 = m_fma(T(2), m_fma(q[0], rq[0], m_fma(q[1], rq[1], m_fma(q[2], rq[2], q[3] * rq[3]))), n2m1);
  const T eps = T(-0.5) * n2m1;
  for (int i = 0; i < 4; ++i) comp_add(q[i], rq[i], q[i] * eps);
}
// [UPSTREAM] _integrateQ as an increment: q' - q = (cos th - 1) q + (sin th / |w|) Lambda(w) q, cos th - 1 = -sin^2 th / (1 + cos th)
template <typename T> MDS_HD void integrate_q_comp(T q[4], T rq[4], V3<T> w, T dt) {
  const T w2 = dot(w, w);
  if (w2 <= T(1e-16)) return;                 // |omega| <= 1e-8
  const T inv_wn = m_rsqrt(w2);
  const T wn = w2 * inv_wn;
  T st, ct;
  m_sincos(wn * dt * T(0.5), &st, &ct);
  const T k = st * inv_wn;
  const T cm1 = -(st * st) * m_rcp(T(1) + ct);
  const T x = q[0], y = q[1], z = q[2], ww = q[3];
  const T p = w.x, qq = w.y, r = w.z;
  const T dx = m_fma(cm1, x, k * (r * y - qq * z + p * ww));
  const T dy = m_fma(cm1, y, k * (-r * x + p * z + qq * ww));
  const T dz = m_fma(cm1, z, k * (qq * x - p * y + r * ww));
  const T dw = m_fma(cm1, ww, k * (-p * x - qq * y - r * z));
  comp_add(q[0], rq[0], dx);
  comp_add(q[1], rq[1], dy);
  comp_add(q[2], rq[2], dz);
  comp_add(q[3], rq[3], dw);
  comp_renorm(q, rq);
}
template <typename T, bool DRAG>
MDS_HD void step_euler_wrench_comp(const Consts<T>& c, State<T>& s, Resid<T>& r, T thrust, V3<T> tau, T drag_s) {
  V3<T> acc, wdot;
  body_accel<T, DRAG>(c, s.q, s.v, s.w, thrust, tau, drag_s, &acc, &wdot);
  comp_add3(s.v, r.v, c.dt * acc);
  comp_add3(s.w, r.w, c.dt * wdot);
  comp_add3(s.p, r.p, V3<T>{m_fma(c.dt, s.v.x, c.dt * r.v.x), m_fma(c.dt, s.v.y, c.dt * r.v.y), m_fma(c.dt, s.v.z, c.dt * r.v.z)});
  integrate_q_comp(s.q, r.q, V3<T>{s.w.x + r.w.x, s.w.y + r.w.y, s.w.z + r.w.z}, c.dt);
}

// [UPSTREAM] BaseAviary._groundEffect / _downwash (the Bullet external forces of Physics.PYB_GND / PYB_DW / PYB_GND_DRAG_DW,
// urdf <properties> gnd_eff_coeff 11.36859, prop_radius 2.31348e-2, dw_coeff_1..3 2267.18, .16, -.11) as extra terms of the DYN
// wrench: per-propeller thrust KF rpm^2 c_g (r_p / (4 h_k))^2 through the same mixing as the rotor thrusts, and a body-z
// force -c_1 (r_p / (4 dz))^2 exp(-(dxy / (c_2 dz + c_3))^2 / 2) per drone of the same env above.  Spec-level, unpinned.
template <typename T> struct EnvFx {
  int gnd, dw;
  T gnd_coeff, prop_radius, h_clip, dw1, dw2, dw3;
  T prop_x[4], prop_y[4];          // propeller link origins, body frame
};
template <typename T>
MDS_HD void ground_effect(const Consts<T>& c, const EnvFx<T>& fx, const State<T>& s, T world_z, const T rpm[4], T* thrust_excess, V3<T>* tau) {
  const M3<T> R = quat_to_rot(s.q);
  const V3<T> rpy = euler_from_quat(s.q);
  if (!(m_abs(rpy.x) < T(1.57079632679489661923) && m_abs(rpy.y) < T(1.57079632679489661923))) return;
  T g[4];
  for (int k = 0; k < 4; ++k) {
    const T h = m_max(world_z + R.m[6] * fx.prop_x[k] + R.m[7] * fx.prop_y[k], fx.h_clip);
    const T ratio = fx.prop_radius / (T(4) * h);
    g[k] = rpm[k] * rpm[k] * c.kf * fx.gnd_coeff * ratio * ratio;
  }
  *thrust_excess += (g[0] + g[1]) + (g[2] + g[3]);
  if (c.cf2x) {
    const T l = c.arm * T(0.70710678118654752440);
    tau->x += l * ((g[0] + g[1]) - (g[2] + g[3]));
    tau->y += l * ((g[1] + g[2]) - (g[0] + g[3]));
  } else {
    tau->x += c.arm * (g[1] - g[3]);
    tau->y += c.arm * (g[2] - g[0]);
  }
}
template <typename T> MDS_HD T downwash_pair(const EnvFx<T>& fx, V3<T> me, V3<T> other) {
  const T dz = other.z - me.z, dx = other.x - me.x, dy = other.y - me.y;
  const T dxy = m_sqrt(m_fma(dx, dx, dy * dy));
  if (!(dz > T(0) && dxy < T(10))) return T(0);
  const T ratio = fx.prop_radius / (T(4) * dz), beta = m_fma(fx.dw2, dz, fx.dw3), u = dxy / beta;
  return -fx.dw1 * ratio * ratio * m_exp(T(-0.5) * u * u);
}

// classical RK4 on the 13-state (north_star integrator; qdot = 1/2 Lambda(omega) q)
template <typename T> struct Deriv {
  V3<T> dp;
  T dq[4];
  V3<T> dv, dw;
};
template <typename T, bool DRAG> MDS_HD Deriv<T> deriv13(const Consts<T>& c, const State<T>& s, T thrust, V3<T> tau, T drag_s) {
  Deriv<T> d;
  body_accel<T, DRAG>(c, s.q, s.v, s.w, thrust, tau, drag_s, &d.dv, &d.dw);
  d.dp = s.v;
  const T x = s.q[0], y = s.q[1], z = s.q[2], w = s.q[3], p = s.w.x, q = s.w.y, r = s.w.z;
  d.dq[0] = T(0.5) * (r * y - q * z + p * w);
  d.dq[1] = T(0.5) * (-r * x + p * z + q * w);
  d.dq[2] = T(0.5) * (q * x - p * y + r * w);
  d.dq[3] = T(0.5) * (-p * x - q * y - r * z);
  return d;
}
template <typename T> MDS_HD State<T> axpy13(const State<T>& s, T h, const Deriv<T>& d) {
  State<T> o;
  o.p = s.p + h * d.dp;
  for (int i = 0; i < 4; ++i) o.q[i] = m_fma(h, d.dq[i], s.q[i]);
  o.v = s.v + h * d.dv;
  o.w = s.w + h * d.dw;
  return o;
}
template <typename T, bool DRAG> MDS_HD void step_rk4(const Consts<T>& c, State<T>& s, const T rpm[4], T drag_s) {
  T thrust;
  V3<T> tau;
  rotor_wrench(c, rpm, &thrust, &tau);
  const Deriv<T> k1 = deriv13<T, DRAG>(c, s, thrust, tau, drag_s);
  const Deriv<T> k2 = deriv13<T, DRAG>(c, axpy13(s, T(0.5) * c.dt, k1), thrust, tau, drag_s);
  const Deriv<T> k3 = deriv13<T, DRAG>(c, axpy13(s, T(0.5) * c.dt, k2), thrust, tau, drag_s);
  const Deriv<T> k4 = deriv13<T, DRAG>(c, axpy13(s, c.dt, k3), thrust, tau, drag_s);
  const T h6 = c.dt / T(6);
  s.p = s.p + h6 * ((k1.dp + k4.dp) + T(2) * (k2.dp + k3.dp));
  s.v = s.v + h6 * ((k1.dv + k4.dv) + T(2) * (k2.dv + k3.dv));
  s.w = s.w + h6 * ((k1.dw + k4.dw) + T(2) * (k2.dw + k3.dw));
  T n2 = T(0);
  for (int i = 0; i < 4; ++i) {
    s.q[i] = m_fma(h6, (k1.dq[i] + k4.dq[i]) + T(2) * (k2.dq[i] + k3.dq[i]), s.q[i]);
    n2 = m_fma(s.q[i], s.q[i], n2);
  }
  const T inv = m_rsqrt(n2);
  for (int i = 0; i < 4; ++i) s.q[i] *= inv;
}

// [UPSTREAM] BaseAviary.step inner loop for one drone: clip, substeps, last_clipped_action.
// rpm_prev: previous control step's clipped action (only read when use_drag).
// RK4 / DRAG are compile-time so that the Euler/DYN kernels carry no RK4 register pressure.
template <typename T, bool RK4, bool DRAG>
MDS_HD void aviary_step(const Consts<T>& c, State<T>& s, const T action[4], T rpm_prev[4], T clipped[4]) {
  for (int i = 0; i < 4; ++i) clipped[i] = m_clamp(action[i], T(0), c.max_rpm);
  for (int k = 0; k < c.substeps; ++k) {
    T drag_s = T(0);
    if (DRAG) drag_s = T(0.10471975511965977462) * ((rpm_prev[0] + rpm_prev[1]) + (rpm_prev[2] + rpm_prev[3]));
    if (RK4) step_rk4<T, DRAG>(c, s, clipped, drag_s);
    else step_euler<T, DRAG>(c, s, clipped, drag_s);
    if (DRAG)
      for (int i = 0; i < 4; ++i) rpm_prev[i] = clipped[i];
  }
}

template <typename T, bool DRAG> MDS_HD void step_rk4_comp(const Consts<T>& c, State<T>& s, Resid<T>& r, const T rpm[4], T drag_s) {
  T thrust;
  V3<T> tau;
  rotor_wrench(c, rpm, &thrust, &tau);
  const Deriv<T> k1 = deriv13<T, DRAG>(c, s, thrust, tau, drag_s);
  const Deriv<T> k2 = deriv13<T, DRAG>(c, axpy13(s, T(0.5) * c.dt, k1), thrust, tau, drag_s);
  const Deriv<T> k3 = deriv13<T, DRAG>(c, axpy13(s, T(0.5) * c.dt, k2), thrust, tau, drag_s);
  const Deriv<T> k4 = deriv13<T, DRAG>(c, axpy13(s, c.dt, k3), thrust, tau, drag_s);
  const T h6 = c.dt / T(6);
  comp_add3(s.p, r.p, h6 * ((k1.dp + k4.dp) + T(2) * (k2.dp + k3.dp)));
  comp_add3(s.v, r.v, h6 * ((k1.dv + k4.dv) + T(2) * (k2.dv + k3.dv)));
  comp_add3(s.w, r.w, h6 * ((k1.dw + k4.dw) + T(2) * (k2.dw + k3.dw)));
  for (int i = 0; i < 4; ++i) comp_add(s.q[i], r.q[i], h6 * ((k1.dq[i] + k4.dq[i]) + T(2) * (k2.dq[i] + k3.dq[i])));
  // |q|^2 - 1 after one RK4 step is O(dt^4 |w|^4): the first-order pull is exact to rounding at these rates
  comp_renorm(s.q, r.q);
}
template <typename T, bool RK4, bool DRAG>
MDS_HD void aviary_step_comp(const Consts<T>& c, State<T>& s, Resid<T>& r, const T action[4], T rpm_prev[4], T clipped[4]) {
  for (int i = 0; i < 4; ++i) clipped[i] = m_clamp(action[i], T(0), c.max_rpm);
  for (int k = 0; k < c.substeps; ++k) {
    T drag_s = T(0);
    if (DRAG) drag_s = T(0.10471975511965977462) * ((rpm_prev[0] + rpm_prev[1]) + (rpm_prev[2] + rpm_prev[3]));
    if (RK4) {
      step_rk4_comp<T, DRAG>(c, s, r, clipped, drag_s);
    } else {
      T thrust;
      V3<T> tau;
      rotor_wrench(c, clipped, &thrust, &tau);
      step_euler_wrench_comp<T, DRAG>(c, s, r, thrust, tau, drag_s);
    }
    if (DRAG)
      for (int i = 0; i < 4; ++i) rpm_prev[i] = clipped[i];
  }
}

// [UPSTREAM] _getDroneStateVector: pos3 | quat4 xyzw | rpy3 | vel3 | ang_v3 (world) | last_clipped_action4.
// ang_v = R(q) w: upstream hands Bullet R(q_before) w, identical because Exp(w dt) w = w.
template <typename T> MDS_HD void pack_obs(const State<T>& s, V3<T> origin, const T rpm[4], T o[20]) {
  const V3<T> av = quat_rotate(s.q, s.w);
  const V3<T> rpy = euler_from_quat(s.q);
  o[0] = s.p.x + origin.x; o[1] = s.p.y + origin.y; o[2] = s.p.z + origin.z;
  o[3] = s.q[0]; o[4] = s.q[1]; o[5] = s.q[2]; o[6] = s.q[3];
  o[7] = rpy.x; o[8] = rpy.y; o[9] = rpy.z;
  o[10] = s.v.x; o[11] = s.v.y; o[12] = s.v.z;
  o[13] = av.x; o[14] = av.y; o[15] = av.z;
  o[16] = rpm[0]; o[17] = rpm[1]; o[18] = rpm[2]; o[19] = rpm[3];
}

// ------------------------------------------------------------------------------------
// trajectories/Lemniscate.py:32-63.  `centre` is NOT added here: the caller works in the
// drone's local frame (origin = trajectory centre) -- see DESIGN.md "local frame".
// ------------------------------------------------------------------------------------
template <typename T> struct Desired {
  V3<T> p, v, a;
  T yaw, yaw_rate;
};
template <typename T> struct LemniscateParams {
  T a, omega, cx, cy, cz, yaw_rate, phase_shift;
};
template <typename T> MDS_HD Desired<T> lemniscate_local(const LemniscateParams<T>& P, double t) {
  Desired<T> d;
  const T th = reduced_phase<T>(t, P.omega, P.phase_shift);
  T s, c;
  m_sincos(th, &s, &c);
  const T s2 = s * s, c2 = c * c;
  const T den = T(1) + s2;
  const T inv = m_rcp(den);
  const T inv2 = inv * inv;
  const T aw = P.a * P.omega;
  d.p = {P.a * s * c * inv, P.a * c * inv, T(0)};
  d.v = {-aw * (s2 * s2 + s2 + (s2 - T(1)) * c2) * inv2, -aw * s * (s2 + T(2) * c2 + T(1)) * inv2, T(0)};
  // sin 2th, cos 2th, cos 4th by double angle (the reference calls sin/cos on 2th, 4th)
  const T sin2 = T(2) * s * c, cos2 = c2 - s2, cos4 = T(1) - T(2) * sin2 * sin2;
  const T e = cos2 - T(3);
  const T inv3 = m_rcp(e * e * e);
  const T aw2 = aw * P.omega;
  d.a = {T(4) * aw2 * sin2 * (T(3) * cos2 + T(7)) * inv3, aw2 * c * (T(44) * cos2 + cos4 - T(21)) * inv3, T(0)};
  T sy = T(0), cy = T(1);
  if (P.yaw_rate != T(0)) {   // sin 0 = 0, cos 0 = 1 exactly: skipping is bitwise neutral, and wave-uniform in C2/C3
    const T ph = reduced_phase<T>(t, P.yaw_rate, T(0));
    m_sincos(ph, &sy, &cy);
  }
  d.yaw = T(3.14159265358979323846) * sy;
  d.yaw_rate = T(3.14159265358979323846) * P.yaw_rate * cy;
  return d;
}

// ------------------------------------------------------------------------------------
// utils/model_conversions.py:69-103 (CF2P "+" mixer)
// ------------------------------------------------------------------------------------
template <typename T> MDS_HD void input_to_action(const Consts<T>& c, const T u_in[4], T rpm[4]) {
  const T u0 = m_max(u_in[0], T(0));
  const T a = T(0.25) * u0, b1 = u_in[1] * c.inv_2L, b2 = u_in[2] * c.inv_2L, d = u_in[3] * c.inv_4r;
  T th[4] = {a - b2 - d, a + b1 + d, a + b2 - d, a - b1 + d};
  for (int i = 0; i < 4; ++i) {
    th[i] = m_clamp(th[i], c.min_motor_thrust, c.max_motor_thrust);
    rpm[i] = m_sqrt(th[i] * c.inv_kf);
  }
}
template <typename T> MDS_HD void action_to_input(const Consts<T>& c, const T action[4], int cap_rpm, T u[4]) {
  T r[4];
  for (int i = 0; i < 4; ++i) r[i] = cap_rpm ? m_clamp(action[i], T(0), c.max_rpm) : action[i];
  // The torques are differences of motor thrusts that nearly cancel near hover: KF (a^2 - b^2) is formed as KF (a - b)(a + b) -- the
  // difference of the RPMs is exact or rounded once, instead of the rounding of two ~0.07 N thrusts surviving in a ~1e-4 N difference
  // (fp32: 30x smaller error in tau / J downstream; the same identity the step kernel uses).  Mathematically the reference's mixer product.
  u[0] = c.kf * ((r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]));
  u[1] = (c.arm * c.kf) * ((r[1] - r[3]) * (r[1] + r[3]));
  u[2] = (c.arm * c.kf) * ((r[2] - r[0]) * (r[2] + r[0]));
  u[3] = c.km * ((r[1] - r[0]) * (r[1] + r[0]) + (r[3] - r[2]) * (r[3] + r[2]));
}

// ------------------------------------------------------------------------------------
// control/geometric.py:59-115 with its quirks (g = 9.81; obs[13:16] (world ang_v) used as
// the body rate; R_des.transpose(0,1) is a no-op so w_des_hat = R_des @ R_dot_des;
// f_des_dot uses Kp on the body-frame velocity error).
//   p_rel = p - p_des (formed by the caller so that the local frame cancels exactly)
// ------------------------------------------------------------------------------------
template <typename T> struct GeoAux {  // return_omegas=True outputs (:106-108)
  T force;
  V3<T> w_des;
  V3<T> b1d, b2d, b3d;  // columns of R_des
};
template <typename T>
MDS_HD void geometric_control(const Consts<T>& c, V3<T> p_rel, const M3<T>& R, V3<T> v_world, V3<T> w,
                              const Desired<T>& des, T u[4], GeoAux<T>* aux) {
  const V3<T> Kp = {c.kp[0], c.kp[1], c.kp[2]}, Kv = {c.kv[0], c.kv[1], c.kv[2]};
  const V3<T> v_b = mulT(R, v_world);                                   // :70
  const V3<T> RTvd = mulT(R, des.v);
  const V3<T> ev = v_b - RTvd;
  // f_des_body/m = R^T(g e3 - Kp ep + a_d) - Kv ev - w x R^T v_d      (:73-74)
  V3<T> tw = des.a - hadamard(Kp, p_rel);
  tw.z += c.g_ctrl;
  const V3<T> fb_m = mulT(R, tw) - hadamard(Kv, ev) - cross(w, RTvd);
  V3<T> f_w = mul(R, c.mass * fb_m);                                    // :75
  T f2 = dot(f_w, f_w);
  T inv_fn = m_rsqrt(f2);
  if (f_w.z * inv_fn < c.cos_max_tilt) {                                // tilt > 40 deg  (:79-80)
    const T scale = f_w.z * c.tan_max_tilt * m_rsqrt(m_fma(f_w.x, f_w.x, f_w.y * f_w.y));   // :81-83
    f_w.x *= scale;
    f_w.y *= scale;
    f2 = dot(f_w, f_w);
    inv_fn = m_rsqrt(f2);
  }
  const T fbz = dot(col(R, 2), f_w);                                    // (R^T f_w).z  (:85)
  T sy = T(0), cy = T(1);
  if (des.yaw != T(0)) m_sincos(des.yaw, &sy, &cy);                     // exact at 0, skipped when the whole wave has yaw 0
  const V3<T> b1c = {cy, sy, T(0)};                                     // :88
  const V3<T> b3d = inv_fn * f_w;
  const V3<T> c1 = cross(b3d, b1c);
  const T inv_n1 = m_rsqrt(dot(c1, c1));
  const V3<T> b2d = inv_n1 * c1;
  const V3<T> c2 = cross(b2d, b3d);
  const V3<T> b1d = m_rsqrt(dot(c2, c2)) * c2;                          // :91
  const V3<T> b1c_dot = {-sy * des.yaw_rate, cy * des.yaw_rate, T(0)};  // :95
  const V3<T> f_dot = (c.mass * inv_fn) * mul(R, hadamard(Kp, ev));     // :96
  const V3<T> b3d_dot = cross(cross(b3d, f_dot), b3d);                  // :97
  const V3<T> inner = inv_n1 * (cross(b1c_dot, b3d) + cross(b1c, b3d_dot));       // |b1c x b3d| = |b3d x b1c|
  const V3<T> b2d_dot = cross(cross(b2d, inner), b2d);                  // :98-99
  const V3<T> b1d_dot = cross(b3d_dot, b2d) + cross(b3d, b2d_dot);      // :100
  // W = R_des @ R_dot_des (no-op transpose, :102); w_des = (W21, W02, W10) (:103)
  const V3<T> w_des = {m_fma(b1d.z, b2d_dot.x, m_fma(b2d.z, b2d_dot.y, b3d.z * b2d_dot.z)),
                       m_fma(b1d.x, b3d_dot.x, m_fma(b2d.x, b3d_dot.y, b3d.x * b3d_dot.z)),
                       m_fma(b1d.y, b1d_dot.x, m_fma(b2d.y, b1d_dot.y, b3d.y * b1d_dot.z))};
  if (aux) {
    aux->force = fbz;                                                   // f_w^T (R e3)  (:107)
    aux->w_des = w_des;
    aux->b1d = b1d;
    aux->b2d = b2d;
    aux->b3d = b3d;
  }
  // e_R = 1/2 KR vee(R_des^T R - R^T R_des), vee(M) = (-M12, M02, -M01)  (:109, :36-44)
  const V3<T> r0 = col(R, 0), r1 = col(R, 1), r2 = col(R, 2);
  const T E12 = dot(b2d, r2) - dot(b3d, r1);
  const T E02 = dot(b1d, r2) - dot(b3d, r0);
  const T E01 = dot(b1d, r1) - dot(b2d, r0);
  const V3<T> eR = {T(-0.5) * c.kR[0] * E12, T(0.5) * c.kR[1] * E02, T(-0.5) * c.kR[2] * E01};
  const V3<T> Rdw = {m_fma(b1d.x, w_des.x, m_fma(b2d.x, w_des.y, b3d.x * w_des.z)),
                     m_fma(b1d.y, w_des.x, m_fma(b2d.y, w_des.y, b3d.y * w_des.z)),
                     m_fma(b1d.z, w_des.x, m_fma(b2d.z, w_des.y, b3d.z * w_des.z))};
  const V3<T> ew = w - mulT(R, Rdw);
  const V3<T> Jw = {c.J[0] * w.x, c.J[1] * w.y, c.J[2] * w.z};
  const V3<T> wxJw = cross(w, Jw);
  u[0] = m_max(T(0), fbz);                                              // :114
  u[1] = c.J[0] * (-eR.x - c.kw[0] * ew.x) - wxJw.x;                    // :110-111
  u[2] = c.J[1] * (-eR.y - c.kw[1] * ew.y) - wxJw.y;
  u[3] = c.J[2] * (-eR.z - c.kw[2] * ew.z) - wxJw.z;
}

// ------------------------------------------------------------------------------------
// control/low_level/thrust_omega_ctrl.py:81-132: body-rate PID -> PWM -> RPM
// (constants :39-60: P 17500, I 10, D 0, PWM2RPM 0.2685 / 4070.3, PWM in [20000, 65535],
// torque clip +-3200).  Stateful: last_omega and the (oddly signed, :117) integral.
// ------------------------------------------------------------------------------------
template <typename T> struct LowLevelState {
  V3<T> last_omega, integral;
};
template <typename T>
MDS_HD void thrust_omega_control(const Consts<T>& c, T ctrl_dt, const T u[4], V3<T> cur, LowLevelState<T>& s, T rpm[4]) {
  const T kP = T(17500), kI = T(10), kD = T(0);
  const T kScale = T(0.2685), kConst = T(4070.3), kMinPwm = T(20000), kMaxPwm = T(65535);
  const T u0 = m_max(u[0], T(0));                                                       // :91
  const T pwm_thrust = m_clamp((m_sqrt(u0 * c.inv_kf * T(0.25)) - kConst) / kScale, kMinPwm, kMaxPwm);   // :92-93
  const T inv_dt = T(1) / ctrl_dt;
  const V3<T> rate_e = {-(cur.x - s.last_omega.x) * inv_dt, -(cur.y - s.last_omega.y) * inv_dt, -(cur.z - s.last_omega.z) * inv_dt};
  const V3<T> e = {u[1] - cur.x, u[2] - cur.y, u[3] - cur.z};                           // :112
  s.last_omega = cur;
  s.integral = {m_clamp(m_clamp(s.integral.x - e.x * ctrl_dt, T(-1500), T(1500)), T(-1), T(1)),   // :117-119
                m_clamp(m_clamp(s.integral.y - e.y * ctrl_dt, T(-1500), T(1500)), T(-1), T(1)),
                m_clamp(s.integral.z - e.z * ctrl_dt, T(-1500), T(1500))};
  const T tx = m_clamp(kP * e.x + kI * s.integral.x + kD * rate_e.x, T(-3200), T(3200));  // :123-129
  const T ty = m_clamp(kP * e.y + kI * s.integral.y + kD * rate_e.y, T(-3200), T(3200));
  const T tz = m_clamp(kP * e.z + kI * s.integral.z + kD * rate_e.z, T(-3200), T(3200));
  T pwm[4];
  if (c.cf2x) {                                                                          // MIXER_MATRIX :46-59
    pwm[0] = pwm_thrust + (T(-0.5) * tx + T(-0.5) * ty - tz);
    pwm[1] = pwm_thrust + (T(-0.5) * tx + T(0.5) * ty + tz);
    pwm[2] = pwm_thrust + (T(0.5) * tx + T(0.5) * ty - tz);
    pwm[3] = pwm_thrust + (T(0.5) * tx + T(-0.5) * ty + tz);
  } else {
    pwm[0] = pwm_thrust + (-ty - tz);
    pwm[1] = pwm_thrust + (tx + tz);
    pwm[2] = pwm_thrust + (ty - tz);
    pwm[3] = pwm_thrust + (-tx + tz);
  }
  for (int i = 0; i < 4; ++i) rpm[i] = m_fma(kScale, m_clamp(pwm[i], kMinPwm, kMaxPwm), kConst);   // :130-132
}

// ------------------------------------------------------------------------------------
// control/lqr/lqr_omega_controller.py:90-119: u = -K e + [M G,0,0,0], cap_u.  x = [rpy, vel, pos]
// (obs_to_lin_model dim 9).  R_eq^T R(rpy) = Rz(yaw - yaw_des) Ry Rx, so the 'xyz' euler error
// is (roll, pitch, wrap(yaw - yaw_des)); position / velocity errors are rotated by Rz(yaw_des)^T.
// ------------------------------------------------------------------------------------
template <typename T> struct LqrGain {
  T k[4][9];
};
template <typename T>
MDS_HD void lqr_omega_control(const Consts<T>& c, const LqrGain<T>& K, V3<T> rpy, V3<T> vel, V3<T> pos, V3<T> pos_des, V3<T> vel_des,
                              T yaw_des, T u[4]) {
  T e[9];
  e[0] = rpy.x;
  e[1] = rpy.y;
  const T dy = rpy.z - yaw_des;
  e[2] = m_fma(T(-6.283185307179586476925), m_rint(dy * T(0.15915494309189533577)), dy);
  T sy, cy;
  m_sincos(reduced_phase<T>(0.0, T(0), yaw_des), &sy, &cy);
  const V3<T> dv = vel - vel_des, dp = pos - pos_des;
  e[3] = cy * dv.x + sy * dv.y; e[4] = -sy * dv.x + cy * dv.y; e[5] = dv.z;
  e[6] = cy * dp.x + sy * dp.y; e[7] = -sy * dp.x + cy * dp.y; e[8] = dp.z;
  for (int r = 0; r < 4; ++r) {
    T acc = T(0);
    for (int k = 0; k < 9; ++k) acc = m_fma(-K.k[r][k], e[k], acc);
    u[r] = acc;
  }
  u[0] += c.gravity;                                                                      // :111
  u[0] = m_clamp(u[0], T(4) * c.min_motor_thrust, c.max_motor_thrust);                    // cap_u :116-119
}

// ------------------------------------------------------------------------------------
// control/lqr/lqr_controller.py:83-113 on model/linearized.py -- the default 'lqr' controller of
// simulations/EnvGeometric.py (:32, :425-427).  x = [rpy, ang_v (the obs' WORLD-frame rate), vel, pos];
// e as in lqr_omega_control plus e[3:6] = R_eq^T (ang_v - [0,0,omega_des]); u = -K e + [M G,0,0,0], no cap;
// the mixer then clips u[0] at 0 in place (model_conversions.py:88), which the returned u shows.
// ------------------------------------------------------------------------------------
template <typename T> struct Lqr12Gain {
  T k[4][12];
};
template <typename T>
MDS_HD void lqr12_control(const Consts<T>& c, const Lqr12Gain<T>& K, V3<T> rpy, V3<T> angv_world, V3<T> vel, V3<T> pos_err, V3<T> vel_des,
                          T yaw_des, T omega_des, T u[4]) {
  T e[12];
  e[0] = rpy.x;
  e[1] = rpy.y;
  const T dy = rpy.z - yaw_des;
  e[2] = m_fma(T(-6.283185307179586476925), m_rint(dy * T(0.15915494309189533577)), dy);
  T sy, cy;
  m_sincos(reduced_phase<T>(0.0, T(0), yaw_des), &sy, &cy);
  const V3<T> dw = {angv_world.x, angv_world.y, angv_world.z - omega_des}, dv = vel - vel_des, dp = pos_err;
  e[3] = cy * dw.x + sy * dw.y; e[4] = -sy * dw.x + cy * dw.y; e[5] = dw.z;
  e[6] = cy * dv.x + sy * dv.y; e[7] = -sy * dv.x + cy * dv.y; e[8] = dv.z;
  e[9] = cy * dp.x + sy * dp.y; e[10] = -sy * dp.x + cy * dp.y; e[11] = dp.z;
  for (int r = 0; r < 4; ++r) {
    T acc = T(0);
    for (int k = 0; k < 12; ++k) acc = m_fma(-K.k[r][k], e[k], acc);
    u[r] = acc;
  }
  u[0] = m_max(u[0] + c.gravity, T(0));
}

// ------------------------------------------------------------------------------------
// control/lqr/lqr_YO_controller.py:99-124: u = [Y, wx, wy, wz] = -K e on the 10-state
// x = [rpy, F, vel, pos] (obs_to_lin_model dim 10); e[3] = calc_z_thrust(obs) - M G is taken in
// its excess form (rotor_wrench) so that near hover it keeps its digits in fp32.  No hover
// offset and no cap (cap_u is `pass`, :126-128).
// ------------------------------------------------------------------------------------
template <typename T> struct LqrYoGain {
  T k[4][10];
};
template <typename T> MDS_HD T thrust_excess_of(const Consts<T>& c, const T rpm[4]) {
  const T h = c.hover_rpm;
  return m_fma(c.kf, (dsq(rpm[0], h) + dsq(rpm[1], h)) + (dsq(rpm[2], h) + dsq(rpm[3], h)), c.thrust_corr);
}
template <typename T>
MDS_HD void lqr_yank_omega_control(const Consts<T>& c, const LqrYoGain<T>& K, V3<T> rpy, const T rpm[4], V3<T> vel, V3<T> pos,
                                   V3<T> pos_des, V3<T> vel_des, T yaw_des, T u[4]) {
  T e[10];
  e[0] = rpy.x;
  e[1] = rpy.y;
  const T dy = rpy.z - yaw_des;
  e[2] = m_fma(T(-6.283185307179586476925), m_rint(dy * T(0.15915494309189533577)), dy);
  e[3] = thrust_excess_of(c, rpm);
  T sy, cy;
  m_sincos(reduced_phase<T>(0.0, T(0), yaw_des), &sy, &cy);
  const V3<T> dv = vel - vel_des, dp = pos - pos_des;
  e[4] = cy * dv.x + sy * dv.y; e[5] = -sy * dv.x + cy * dv.y; e[6] = dv.z;
  e[7] = cy * dp.x + sy * dp.y; e[8] = -sy * dp.x + cy * dp.y; e[9] = dp.z;
  for (int r = 0; r < 4; ++r) {
    T acc = T(0);
    for (int k = 0; k < 10; ++k) acc = m_fma(-K.k[r][k], e[k], acc);
    u[r] = acc;
  }
}

// control/low_level/yank_omega_ctrl.py:39-55 under lqr_YO_controller.py:85-97:
// thrust_cmd = calc_z_thrust(obs) + yank * dt, then the ThrustOmega PID.
template <typename T>
MDS_HD void yank_omega_control(const Consts<T>& c, T ctrl_dt, const T u[4], const T rpm_obs[4], V3<T> cur, LowLevelState<T>& s,
                               T rpm[4]) {
  const T cur_thrust = c.gravity + thrust_excess_of(c, rpm_obs);
  const T ut[4] = {m_fma(u[0], ctrl_dt, cur_thrust), u[1], u[2], u[3]};
  thrust_omega_control(c, ctrl_dt, ut, cur, s, rpm);
}

// ------------------------------------------------------------------------------------
// [UPSTREAM] gym_pybullet_drones DSLPIDControl.computeControl as PIDEnv.py:166-169 calls it
// (computeControlFromState; target_vel = target_rpy_rates = 0).  Not in the reference tree:
// restated from the published upstream source, parity unpinned.  The intermediate
// as_euler('XYZ') / from_euler round trip of the target rotation is the identity and is skipped.
// ------------------------------------------------------------------------------------
template <typename T> struct DslPidGains {
  T Pf[3], If[3], Df[3], Pt[3], It[3], Dt[3];
};
template <typename T> struct DslPidState {
  V3<T> last_rpy, int_pos, int_rpy;
};
template <typename T>
MDS_HD void dslpid_control(const Consts<T>& c, const DslPidGains<T>& g, T ctrl_dt, V3<T> pos_e, const T q[4], V3<T> vel, T target_yaw,
                           DslPidState<T>& s, T rpm[4]) {
  const T kScale = T(0.2685), kConst = T(4070.3), kMinPwm = T(20000), kMaxPwm = T(65535);
  const M3<T> R = quat_to_rot(q);
  // _dslPIDPositionControl
  s.int_pos = {m_clamp(m_fma(pos_e.x, ctrl_dt, s.int_pos.x), T(-2), T(2)), m_clamp(m_fma(pos_e.y, ctrl_dt, s.int_pos.y), T(-2), T(2)),
               m_clamp(m_clamp(m_fma(pos_e.z, ctrl_dt, s.int_pos.z), T(-2), T(2)), T(-0.15), T(0.15))};
  const V3<T> tt = {g.Pf[0] * pos_e.x + g.If[0] * s.int_pos.x - g.Df[0] * vel.x, g.Pf[1] * pos_e.y + g.If[1] * s.int_pos.y - g.Df[1] * vel.y,
                    g.Pf[2] * pos_e.z + g.If[2] * s.int_pos.z - g.Df[2] * vel.z + c.gravity};
  const T scalar = m_max(T(0), dot(tt, col(R, 2)));
  const T thrust = (m_sqrt(scalar * c.inv_kf * T(0.25)) - kConst) / kScale;
  const V3<T> z_ax = m_rsqrt(dot(tt, tt)) * tt;
  T sy, cy;
  m_sincos(reduced_phase<T>(0.0, T(0), target_yaw), &sy, &cy);
  const V3<T> yc = cross(z_ax, V3<T>{cy, sy, T(0)});
  const V3<T> y_ax = m_rsqrt(dot(yc, yc)) * yc;
  const V3<T> x_ax = cross(y_ax, z_ax);
  // _dslPIDAttitudeControl
  const V3<T> cur_rpy = euler_from_quat(q);
  const V3<T> r0 = col(R, 0), r1 = col(R, 1), r2 = col(R, 2);
  const V3<T> rot_e = {dot(z_ax, r1) - dot(y_ax, r2), dot(x_ax, r2) - dot(z_ax, r0), dot(y_ax, r0) - dot(x_ax, r1)};
  const T inv_dt = T(1) / ctrl_dt;
  const V3<T> rates_e = {-(cur_rpy.x - s.last_rpy.x) * inv_dt, -(cur_rpy.y - s.last_rpy.y) * inv_dt, -(cur_rpy.z - s.last_rpy.z) * inv_dt};
  s.last_rpy = cur_rpy;
  s.int_rpy = {m_clamp(m_clamp(s.int_rpy.x - rot_e.x * ctrl_dt, T(-1500), T(1500)), T(-1), T(1)),
               m_clamp(m_clamp(s.int_rpy.y - rot_e.y * ctrl_dt, T(-1500), T(1500)), T(-1), T(1)),
               m_clamp(s.int_rpy.z - rot_e.z * ctrl_dt, T(-1500), T(1500))};
  const T tx = m_clamp(-g.Pt[0] * rot_e.x + g.Dt[0] * rates_e.x + g.It[0] * s.int_rpy.x, T(-3200), T(3200));
  const T ty = m_clamp(-g.Pt[1] * rot_e.y + g.Dt[1] * rates_e.y + g.It[1] * s.int_rpy.y, T(-3200), T(3200));
  const T tz = m_clamp(-g.Pt[2] * rot_e.z + g.Dt[2] * rates_e.z + g.It[2] * s.int_rpy.z, T(-3200), T(3200));
  T pwm[4];
  if (c.cf2x) {
    pwm[0] = thrust + (T(-0.5) * tx + T(-0.5) * ty - tz);
    pwm[1] = thrust + (T(-0.5) * tx + T(0.5) * ty + tz);
    pwm[2] = thrust + (T(0.5) * tx + T(0.5) * ty - tz);
    pwm[3] = thrust + (T(0.5) * tx + T(-0.5) * ty + tz);
  } else {
    pwm[0] = thrust + (-ty - tz);
    pwm[1] = thrust + (tx + tz);
    pwm[2] = thrust + (ty - tz);
    pwm[3] = thrust + (-tx + tz);
  }
  for (int i = 0; i < 4; ++i) rpm[i] = m_fma(kScale, m_clamp(pwm[i], kMinPwm, kMaxPwm), kConst);
}

// model/dynamics.py:83-106: (state18, u4) -> 12 floats (x_dot = v, "R_dot" = w, v_dot, w_dot)
template <typename T> MDS_HD void quadrotor_dynamics(const T s[18], const T u[4], T m, const T J[3], T g, T out[12]) {
  out[0] = s[12]; out[1] = s[13]; out[2] = s[14];
  out[3] = s[15]; out[4] = s[16]; out[5] = s[17];
  const T a = u[0] / m;
  out[6] = s[5] * a;   // R[:,2] = s[3+2], s[3+5], s[3+8]
  out[7] = s[8] * a;
  out[8] = s[11] * a - g;
  const V3<T> w = {s[15], s[16], s[17]};
  const V3<T> Jw = {J[0] * w.x, J[1] * w.y, J[2] * w.z};
  const V3<T> cx = cross(w, Jw);
  out[9] = (u[1] - cx.x) / J[0];
  out[10] = (u[2] - cx.y) / J[1];
  out[11] = (u[3] - cx.z) / J[2];
}

// ------------------------------------------------------------------------------------
// The call site of QuadrotorDynamics.dynamics: simulations/CompareModels.py:46-56 and the helpers it uses
// ------------------------------------------------------------------------------------

// utils/model_conversions.py:4-19 rpy_to_rot: R = Rz(yaw) Ry(pitch) Rx(roll), row-major (angles reduced by the caller for fp32)
template <typename T> MDS_HD M3<T> rpy_to_rot(T roll, T pitch, T yaw) {
  T sr, cr, sp, cp, sy, cy;
  m_sincos(roll, &sr, &cr);
  m_sincos(pitch, &sp, &cp);
  m_sincos(yaw, &sy, &cy);
  M3<T> R;
  R.m[0] = cy * cp; R.m[1] = m_fma(cy * sp, sr, -(sy * cr)); R.m[2] = m_fma(cy * sp, cr, sy * sr);
  R.m[3] = sy * cp; R.m[4] = m_fma(sy * sp, sr, cy * cr);    R.m[5] = m_fma(sy * sp, cr, -(cy * sr));
  R.m[6] = -sp;     R.m[7] = cp * sr;                        R.m[8] = cp * cr;
  return R;
}

// scipy's Rotation.from_matrix(R).as_quat() (xyzw) as geo_model_to_obs uses it (utils/model_conversions.py:119): the branch of
// the largest of (R00, R11, R22, trace), the first maximum winning, then normalised; the sign is whatever that branch gives
template <typename T> MDS_HD void rot_to_quat_scipy(const T m[9], T q[4]) {
  const T tr = (m[0] + m[4]) + m[8];
  int ch = 0;
  T best = m[0];
  if (m[4] > best) { best = m[4]; ch = 1; }
  if (m[8] > best) { best = m[8]; ch = 2; }
  if (tr > best) ch = 3;
  if (ch == 3) {
    q[0] = m[7] - m[5];
    q[1] = m[2] - m[6];
    q[2] = m[3] - m[1];
    q[3] = T(1) + tr;
  } else {
    // i = ch, j = i + 1, k = i + 2 (mod 3): q[i] = 1 - tr + 2 m[i][i], q[j] = m[j][i] + m[i][j], q[k] = m[k][i] + m[i][k], q[3] = m[k][j] - m[j][k]
    const int i = ch, j = (ch + 1) % 3, k = (ch + 2) % 3;
    const T qi = (T(1) - tr) + T(2) * m[4 * i], qj = m[3 * j + i] + m[3 * i + j], qk = m[3 * k + i] + m[3 * i + k];
    q[0] = i == 0 ? qi : (j == 0 ? qj : qk);
    q[1] = i == 1 ? qi : (j == 1 ? qj : qk);
    q[2] = i == 2 ? qi : (j == 2 ? qj : qk);
    q[3] = m[3 * k + j] - m[3 * j + k];
  }
  const T inv = m_rsqrt((q[0] * q[0] + q[1] * q[1]) + (q[2] * q[2] + q[3] * q[3]));
  for (int a = 0; a < 4; ++a) q[a] *= inv;
}

// model/linearized.py:92-104 LinearizedModel.calc_xdot: A (x - x_eq) + B (u - u_eq); x_eq = (0 .. 0, the position of x), u_eq = (M G, 0, 0, 0).
// A [12][12], B [12][4] are the caller's dense matrices (the true pair or Ahat / Bhat): uniform operands.
template <typename T> struct LinModel {
  T A[12][12], B[12][4];
  T ueq0;
};
template <typename T> MDS_HD void linear_xdot(const LinModel<T>& M, const T x[12], const T u[4], T out[12]) {
  T dx[12], du[4];
  for (int k = 0; k < 9; ++k) dx[k] = x[k];
  dx[9] = x[9] - x[9]; dx[10] = x[10] - x[10]; dx[11] = x[11] - x[11];           // x - x_eq (NaN / inf propagate as in the reference)
  du[0] = u[0] - M.ueq0; du[1] = u[1]; du[2] = u[2]; du[3] = u[3];
  for (int r = 0; r < 12; ++r) {
    T a = M.A[r][0] * dx[0];
    for (int k = 1; k < 12; ++k) a = m_fma(M.A[r][k], dx[k], a);
    T b = M.B[r][0] * du[0];
    for (int k = 1; k < 4; ++k) b = m_fma(M.B[r][k], du[k], b);
    out[r] = a + b;
  }
}

// CompareModels.py:48-56 for one observation row: x_lin = obs_to_lin_model(obs); xdot_lin = calc_xdot_from_obs(obs);
// xdot_geo = geo_x_dot_to_linear(dynamics(None, obs_to_geo_model(obs), action_to_input(env, obs[16:])))
template <typename T>
MDS_HD void compare_models_row(const Consts<T>& c, const LinModel<T>& M, const T o[20], T dm, const T dJ[3], T dg, T x_lin[12], T xdot_lin[12],
                               T xdot_geo[12]) {
  x_lin[0] = o[7]; x_lin[1] = o[8]; x_lin[2] = o[9];
  x_lin[3] = o[13]; x_lin[4] = o[14]; x_lin[5] = o[15];
  x_lin[6] = o[10]; x_lin[7] = o[11]; x_lin[8] = o[12];
  x_lin[9] = o[0]; x_lin[10] = o[1]; x_lin[11] = o[2];
  T u[4];
  action_to_input(c, o + 16, 1, u);
  linear_xdot(M, x_lin, u, xdot_lin);
  T s[18], g[12];
  const T q[4] = {o[3], o[4], o[5], o[6]};
  const M3<T> R = quat_to_rot(q);
  s[0] = o[0]; s[1] = o[1]; s[2] = o[2];
  for (int k = 0; k < 9; ++k) s[3 + k] = R.m[k];
  for (int k = 0; k < 6; ++k) s[12 + k] = o[10 + k];
  quadrotor_dynamics<T>(s, u, dm, dJ, dg, g);
  // geo_x_dot_to_linear (utils/model_conversions.py:124-135): (v, w, v_dot, w_dot) -> (w, w_dot, v_dot, v)
  xdot_geo[0] = g[3]; xdot_geo[1] = g[4]; xdot_geo[2] = g[5];
  xdot_geo[3] = g[9]; xdot_geo[4] = g[10]; xdot_geo[5] = g[11];
  xdot_geo[6] = g[6]; xdot_geo[7] = g[7]; xdot_geo[8] = g[8];
  xdot_geo[9] = g[0]; xdot_geo[10] = g[1]; xdot_geo[11] = g[2];
}

}  // namespace mds
