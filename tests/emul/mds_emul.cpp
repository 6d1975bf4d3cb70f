// TEST TOOLING ONLY.  Compiles the device arithmetic of multidronesim_amd/csrc/mds_math.hpp
// with g++ so the fp32 / f64 per-drone math can be exercised on the CPU (precision studies
// against the oracle, UBSan/ASan runs).  Never loaded by the multidronesim_amd package;
// the product path is the HIP library only.
#include <string.h>

#include "../../multidronesim_amd/csrc/mds_consts.hpp"

using namespace mds;

template <typename T> static void load_state(const double* st, State<T>& s, const double* org) {
  s.p = {(T)(st[0] - org[0]), (T)(st[1] - org[1]), (T)(st[2] - org[2])};
  for (int k = 0; k < 4; ++k) s.q[k] = (T)st[3 + k];
  s.v = {(T)st[7], (T)st[8], (T)st[9]};
  s.w = {(T)st[10], (T)st[11], (T)st[12]};
}

template <typename T> static void emul_aviary_step(const Consts<T>& c, State<T>& s, const T act[4], T prev[4], T clipped[4]) {
  if (c.rk4 && c.use_drag) aviary_step<T, true, true>(c, s, act, prev, clipped);
  else if (c.rk4) aviary_step<T, true, false>(c, s, act, prev, clipped);
  else if (c.use_drag) aviary_step<T, false, true>(c, s, act, prev, clipped);
  else aviary_step<T, false, false>(c, s, act, prev, clipped);
}

template <typename T> static void emul_aviary_step_comp(const Consts<T>& c, State<T>& s, Resid<T>& r, const T act[4], T prev[4], T clipped[4]) {
  if (c.rk4 && c.use_drag) aviary_step_comp<T, true, true>(c, s, r, act, prev, clipped);
  else if (c.rk4) aviary_step_comp<T, true, false>(c, s, r, act, prev, clipped);
  else if (c.use_drag) aviary_step_comp<T, false, true>(c, s, r, act, prev, clipped);
  else aviary_step_comp<T, false, false>(c, s, r, act, prev, clipped);
}

// Persistent emulated handle: state kept in T (like the device SoA), I/O in double.
template <typename T> struct Emul {
  Consts<T> c;
  int n;
  int comp = 0;          // 1: compensated accumulation (MDS_F32C)
  int comp_mask = 8;     // which residuals survive a step (1 p, 2 q, 4 v, 8 w).  8 = the device's MDS_F32C storage (body rates only,
                         // load_resid / store_resid in mds_kernels.hip); other masks: the study behind that choice
  Resid<T>* r;
  State<T>* s;
  T (*prev)[4];
  LemniscateParams<T>* P;
  double (*org)[3];
};

template <typename T> static void* emul_create(const mds_config* cfg, const mds_geometric_gains* g) {
  Emul<T>* e = new Emul<T>();
  fill_consts(*cfg, *g, e->c);
  e->n = cfg->num_envs * cfg->num_drones;
  e->s = new State<T>[e->n];
  e->r = new Resid<T>[e->n];
  for (int i = 0; i < e->n; ++i) resid_zero(e->r[i]);
  e->prev = new T[e->n][4];
  e->P = new LemniscateParams<T>[e->n];
  e->org = new double[e->n][3];
  memset(e->prev, 0, sizeof(T) * 4 * e->n);
  memset(e->org, 0, sizeof(double) * 3 * e->n);
  return e;
}

// residuals that are not stored are lost between control steps
template <typename T> static void emul_drop_unstored(Emul<T>* e, int i) {
  Resid<T>& r = e->r[i];
  if (!(e->comp_mask & 1)) r.p = {T(0), T(0), T(0)};
  if (!(e->comp_mask & 2)) r.q[0] = r.q[1] = r.q[2] = r.q[3] = T(0);
  if (!(e->comp_mask & 4)) r.v = {T(0), T(0), T(0)};
  if (!(e->comp_mask & 8)) r.w = {T(0), T(0), T(0)};
}
template <typename T> static void emul_set_state(void* h, const double* st) {
  Emul<T>* e = (Emul<T>*)h;
  for (int i = 0; i < e->n; ++i) {
    load_state<T>(st + 13 * i, e->s[i], e->org[i]);
    resid_zero(e->r[i]);
    if (e->comp) {          // residual = what the rounding to T dropped
      const double* o = st + 13 * i;
      State<T>& s = e->s[i];
      Resid<T>& r = e->r[i];
      r.p = {(T)((o[0] - e->org[i][0]) - (double)s.p.x), (T)((o[1] - e->org[i][1]) - (double)s.p.y), (T)((o[2] - e->org[i][2]) - (double)s.p.z)};
      for (int k = 0; k < 4; ++k) r.q[k] = (T)(o[3 + k] - (double)s.q[k]);
      r.v = {(T)(o[7] - (double)s.v.x), (T)(o[8] - (double)s.v.y), (T)(o[9] - (double)s.v.z)};
      r.w = {(T)(o[10] - (double)s.w.x), (T)(o[11] - (double)s.w.y), (T)(o[12] - (double)s.w.z)};
      emul_drop_unstored(e, i);
    }
  }
}
template <typename T> static void emul_get_state(void* h, double* st) {
  Emul<T>* e = (Emul<T>*)h;
  for (int i = 0; i < e->n; ++i) {
    const State<T>& s = e->s[i];
    double* o = st + 13 * i;
    o[0] = (double)s.p.x + e->org[i][0]; o[1] = (double)s.p.y + e->org[i][1]; o[2] = (double)s.p.z + e->org[i][2];
    for (int k = 0; k < 4; ++k) o[3 + k] = s.q[k];
    o[7] = s.v.x; o[8] = s.v.y; o[9] = s.v.z; o[10] = s.w.x; o[11] = s.w.y; o[12] = s.w.z;
    if (e->comp) {
      const Resid<T>& r = e->r[i];
      o[0] += r.p.x; o[1] += r.p.y; o[2] += r.p.z;
      for (int k = 0; k < 4; ++k) o[3 + k] += r.q[k];
      o[7] += r.v.x; o[8] += r.v.y; o[9] += r.v.z; o[10] += r.w.x; o[11] += r.w.y; o[12] += r.w.z;
    }
  }
}
template <typename T> static void emul_set_lem(void* h, const double* p) {
  Emul<T>* e = (Emul<T>*)h;
  double* st = new double[13 * (size_t)e->n];
  emul_get_state<T>(h, st);
  for (int i = 0; i < e->n; ++i) {
    const double* q = p + 7 * i;
    e->P[i] = {(T)q[0], (T)q[1], (T)q[2], (T)q[3], (T)q[4], (T)q[5], (T)q[6]};
    for (int k = 0; k < 3; ++k) e->org[i][k] = (double)(T)q[2 + k];
  }
  emul_set_state<T>(h, st);
  delete[] st;
}
template <typename T> static void emul_step(void* h, const double* action, double* obs) {
  Emul<T>* e = (Emul<T>*)h;
  for (int i = 0; i < e->n; ++i) {
    T act[4], clipped[4], o[20];
    for (int k = 0; k < 4; ++k) act[k] = (T)action[4 * i + k];
    if (e->comp) emul_aviary_step_comp(e->c, e->s[i], e->r[i], act, e->prev[i], clipped);
    else emul_aviary_step(e->c, e->s[i], act, e->prev[i], clipped);
    if (e->comp) emul_drop_unstored(e, i);
    if (obs) {
      pack_obs(e->s[i], V3<T>{(T)e->org[i][0], (T)e->org[i][1], (T)e->org[i][2]}, clipped, o);
      for (int k = 0; k < 20; ++k) obs[20 * i + k] = o[k];
    }
  }
}
template <typename T> static void emul_step_geo(void* h, double t, double* obs, double* act_out) {
  Emul<T>* e = (Emul<T>*)h;
  for (int i = 0; i < e->n; ++i) {
    T clipped[4], act[4], o[20], u[4];
    State<T>& s = e->s[i];
    const Desired<T> des = lemniscate_local(e->P[i], t);
    const M3<T> R = quat_to_rot(s.q);
    const V3<T> ang_v = mul(R, s.w);
    geometric_control<T>(e->c, s.p - des.p, R, s.v, ang_v, des, u, nullptr);
    input_to_action(e->c, u, act);
    if (e->comp) emul_aviary_step_comp(e->c, s, e->r[i], act, e->prev[i], clipped);
    else emul_aviary_step(e->c, s, act, e->prev[i], clipped);
    if (e->comp) emul_drop_unstored(e, i);
    if (act_out)
      for (int k = 0; k < 4; ++k) act_out[4 * i + k] = act[k];
    if (obs) {
      pack_obs(s, V3<T>{e->P[i].cx, e->P[i].cy, e->P[i].cz}, clipped, o);
      for (int k = 0; k < 20; ++k) obs[20 * i + k] = o[k];
    }
  }
}
template <typename T> static void emul_geo_compute(void* h, int n, const double* obs, const double* des, double* rpm, double* aux) {
  Emul<T>* e = (Emul<T>*)h;
  for (int i = 0; i < n; ++i) {
    const double* o = obs + 20 * i;
    const double* d = des + 11 * i;
    const T q[4] = {(T)o[3], (T)o[4], (T)o[5], (T)o[6]};
    const M3<T> R = quat_to_rot(q);
    Desired<T> D;
    D.p = {(T)d[0], (T)d[1], (T)d[2]};
    D.v = {(T)d[3], (T)d[4], (T)d[5]};
    D.a = {(T)d[6], (T)d[7], (T)d[8]};
    D.yaw = reduced_phase<T>(0.0, T(0), (T)d[9]);
    D.yaw_rate = (T)d[10];
    T u[4], act[4];
    GeoAux<T> A;
    geometric_control<T>(e->c, V3<T>{(T)o[0], (T)o[1], (T)o[2]} - D.p, R, V3<T>{(T)o[10], (T)o[11], (T)o[12]},
                         V3<T>{(T)o[13], (T)o[14], (T)o[15]}, D, u, &A);
    input_to_action(e->c, u, act);
    for (int k = 0; k < 4; ++k) rpm[4 * i + k] = act[k];
    if (aux) {
      double* a = aux + 13 * i;
      a[0] = A.force; a[1] = A.w_des.x; a[2] = A.w_des.y; a[3] = A.w_des.z;
      a[4] = A.b1d.x; a[5] = A.b2d.x; a[6] = A.b3d.x; a[7] = A.b1d.y; a[8] = A.b2d.y; a[9] = A.b3d.y;
      a[10] = A.b1d.z; a[11] = A.b2d.z; a[12] = A.b3d.z;
    }
  }
}
template <typename T> static void emul_lem(void* h, double t, double* des) {
  Emul<T>* e = (Emul<T>*)h;
  for (int i = 0; i < e->n; ++i) {
    const Desired<T> d = lemniscate_local(e->P[i], t);
    double* o = des + 11 * i;
    o[0] = (double)d.p.x + (double)e->P[i].cx; o[1] = (double)d.p.y + (double)e->P[i].cy; o[2] = (double)d.p.z + (double)e->P[i].cz;
    o[3] = d.v.x; o[4] = d.v.y; o[5] = d.v.z; o[6] = d.a.x; o[7] = d.a.y; o[8] = d.a.z; o[9] = d.yaw; o[10] = d.yaw_rate;
  }
}

// the call site of QuadrotorDynamics.dynamics (simulations/CompareModels.py:46-56) on the device templates
template <typename T>
static void emul_compare_models(void* h, int n, const double* obs, const double* A, const double* B, double ueq0, double dm, const double* dJ,
                                double dg, double* xdot_lin, double* xdot_geo, double* x_lin) {
  Emul<T>* e = (Emul<T>*)h;
  LinModel<T> M;
  for (int r = 0; r < 12; ++r) {
    for (int k = 0; k < 12; ++k) M.A[r][k] = (T)A[r * 12 + k];
    for (int k = 0; k < 4; ++k) M.B[r][k] = (T)B[r * 4 + k];
  }
  M.ueq0 = (T)ueq0;
  const T J[3] = {(T)dJ[0], (T)dJ[1], (T)dJ[2]};
  for (int i = 0; i < n; ++i) {
    T o[20], xl[12], xd[12], xg[12];
    for (int k = 0; k < 20; ++k) o[k] = (T)obs[20 * i + k];
    compare_models_row<T>(e->c, M, o, (T)dm, J, (T)dg, xl, xd, xg);
    for (int k = 0; k < 12; ++k) {
      x_lin[12 * i + k] = xl[k];
      xdot_lin[12 * i + k] = xd[k];
      xdot_geo[12 * i + k] = xg[k];
    }
  }
}
template <typename T> static void emul_rpy_to_rot(int n, const double* rpy, double* R) {
  for (int i = 0; i < n; ++i) {
    const M3<T> m = rpy_to_rot<T>(reduced_phase<T>(0.0, T(0), (T)rpy[3 * i]), reduced_phase<T>(0.0, T(0), (T)rpy[3 * i + 1]),
                                  reduced_phase<T>(0.0, T(0), (T)rpy[3 * i + 2]));
    for (int k = 0; k < 9; ++k) R[9 * i + k] = m.m[k];
  }
}
template <typename T> static void emul_rot_to_quat(int n, const double* R, double* q) {
  for (int i = 0; i < n; ++i) {
    T m[9], qq[4];
    for (int k = 0; k < 9; ++k) m[k] = (T)R[9 * i + k];
    rot_to_quat_scipy<T>(m, qq);
    for (int k = 0; k < 4; ++k) q[4 * i + k] = qq[k];
  }
}

extern "C" {
void* emul_create_f32(const mds_config* c, const mds_geometric_gains* g) { return emul_create<float>(c, g); }
void* emul_create_f64(const mds_config* c, const mds_geometric_gains* g) { return emul_create<double>(c, g); }
void emul_set_comp_f32(void* h, int on) { ((Emul<float>*)h)->comp = on; }
void emul_set_comp_mask_f32(void* h, int mask) { ((Emul<float>*)h)->comp_mask = mask; }
void emul_set_comp_f64(void* h, int on) { ((Emul<double>*)h)->comp = on; }
void emul_set_state_f32(void* h, const double* s) { emul_set_state<float>(h, s); }
void emul_set_state_f64(void* h, const double* s) { emul_set_state<double>(h, s); }
void emul_get_state_f32(void* h, double* s) { emul_get_state<float>(h, s); }
void emul_get_state_f64(void* h, double* s) { emul_get_state<double>(h, s); }
void emul_set_lem_f32(void* h, const double* p) { emul_set_lem<float>(h, p); }
void emul_set_lem_f64(void* h, const double* p) { emul_set_lem<double>(h, p); }
void emul_step_f32(void* h, const double* a, double* o) { emul_step<float>(h, a, o); }
void emul_step_f64(void* h, const double* a, double* o) { emul_step<double>(h, a, o); }
void emul_step_geo_f32(void* h, double t, double* o, double* a) { emul_step_geo<float>(h, t, o, a); }
void emul_step_geo_f64(void* h, double t, double* o, double* a) { emul_step_geo<double>(h, t, o, a); }
void emul_geo_compute_f32(void* h, int n, const double* o, const double* d, double* r, double* a) { emul_geo_compute<float>(h, n, o, d, r, a); }
void emul_geo_compute_f64(void* h, int n, const double* o, const double* d, double* r, double* a) { emul_geo_compute<double>(h, n, o, d, r, a); }
void emul_lem_f32(void* h, double t, double* d) { emul_lem<float>(h, t, d); }
void emul_lem_f64(void* h, double t, double* d) { emul_lem<double>(h, t, d); }
void emul_compare_models_f32(void* h, int n, const double* o, const double* A, const double* B, double u, double m, const double* J, double g,
                             double* a, double* b, double* c) { emul_compare_models<float>(h, n, o, A, B, u, m, J, g, a, b, c); }
void emul_compare_models_f64(void* h, int n, const double* o, const double* A, const double* B, double u, double m, const double* J, double g,
                             double* a, double* b, double* c) { emul_compare_models<double>(h, n, o, A, B, u, m, J, g, a, b, c); }
void emul_rpy_to_rot_f32(int n, const double* r, double* R) { emul_rpy_to_rot<float>(n, r, R); }
void emul_rpy_to_rot_f64(int n, const double* r, double* R) { emul_rpy_to_rot<double>(n, r, R); }
void emul_rot_to_quat_f32(int n, const double* R, double* q) { emul_rot_to_quat<float>(n, R, q); }
void emul_rot_to_quat_f64(int n, const double* R, double* q) { emul_rot_to_quat<double>(n, R, q); }
void emul_sincos_f32(int n, const float* x, float* s, float* c) {
  for (int i = 0; i < n; ++i) m_sincos(x[i], s + i, c + i);
}
void emul_default_config(int model, mds_config* cfg, mds_geometric_gains* g) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->num_envs = 1; cfg->num_drones = 1; cfg->dtype = MDS_F32; cfg->drone_model = model;
  cfg->pyb_freq = 100; cfg->ctrl_freq = 100;
  cfg->M = 0.027; cfg->L = 0.0397; cfg->KF = 3.16e-10; cfg->KM = 7.94e-12;
  if (model == MDS_CF2P) { cfg->J[0] = 2.3951e-5; cfg->J[1] = 2.3951e-5; cfg->J[2] = 3.2347e-5; }
  else { cfg->J[0] = 1.4e-5; cfg->J[1] = 1.4e-5; cfg->J[2] = 2.17e-5; }
  cfg->G = 9.8; cfg->thrust2weight = 2.25;
  cfg->drag_coeff[0] = 9.1785e-7; cfg->drag_coeff[1] = 9.1785e-7; cfg->drag_coeff[2] = 10.311e-7;
  for (int k = 0; k < 3; ++k) { g->Kp[k] = 2.25; g->Kv[k] = 3.5; g->KR[k] = 125.0; g->Kw[k] = 10.0; }
  g->g = 9.81; g->max_tilt_angle = 40.0 * M_PI / 180.0;
}
}
