// Exponential-CBF rows of cbf/cbf.py in closed form, templated on the compute type.
//
// The reference builds, per pair, dense 2*xdim matrices and a (2*xdim)^3 tensor
// (cbf/cbf.py:194-283).  For the two hover linearisations it is used with
// (model/linear_omega.py:46-53, model/linear_yank_omega.py:45-51) the result collapses to
// 3-vector algebra; tests/golden/cbf_rows_o{2,3}.npz, minted from the
// reference's own construction, pin this restatement (incl. its quirks: the hard-coded
// slots 6,7,8 in custom_hdots for order 3, :158-169, and the force-box rows written to the
// omega_z column, :446-464).
#pragma once
#include "mds_math.hpp"

namespace mds {

template <typename T> struct CbfParams {
  int order;      // 2: state [r,p,y,vx,vy,vz,x,y,z], inputs [F-mg, wx,wy,wz]; 3: [r,p,y,F,vx,vy,vz,x,y,z], [Y,wx,wy,wz]
  int n_obs;
  int num_drones;
  T k[3];         // Kcbf, ascending (cbf/cbf.py:119-124)
  T umax[4];      // cbf/cbf.py:566-572
  T Ds_pair;      // 2 * safety_radius (:291)
  T safety_radius;
  T zscale, inv_zscale, inv_c4;
  T c4x4, c4x12;  // 4 / zscale^4, 12 / zscale^4 (gradient and Hessian of (z / zscale)^4)
  int obs_magic;  // ceil(2^16 / n_obs): row index -> (agent, obstacle) without an integer division
  T inv_m, g;     // env.M, env.G (9.8) through the linear models
  T Fmin, Fmax;   // order 3 force box (:564-565)
};

// Order-2 row from the quantities it depends on: e = pos_i - pos_j (actual positions) and the differences of the tracking errors
// d = (x_i - xdes_i) - (x_j - xdes_j) in roll, pitch and velocity.  Returns h_row and the thrust coefficient L_g L_f h (the omega
// columns are zero with the omega linearisation).
//
// Takes its operands as the PAIRS the persistent rollout kernel keeps side by side in its LDS records -- (ex, ey),
// (d_pitch, -d_roll), (dvx, dvy), (ez, dvz) -- with every fused multiply-add written out (contraction off: every kernel that
// instantiates it rounds alike); 36 instructions.  A version of this body on a float ext_vector_type(2) (v_pk_mul_f32 /
// v_pk_fma_f32) was measured and dropped: on MI355X a packed fp32 instruction issues at half the rate of a scalar one
// (profiles/tools/ubench/valu_rate.hip: 0.96 vs 1.89 ns per instruction and SIMD at 8 waves), so packing buys no arithmetic
// throughput, and the rollout kernel ran 1 % slower with it.
// With s = ex^2 + ey^2 the Hessian of s^2 is Hxx = 4 s + 8 ex^2, Hyy = 4 s + 8 ey^2, 2 Hxy = 16 ex ey; the z terms use the
// host-side products c4x4 = 4 / zscale^4, c4x12 = 12 / zscale^4; nDs4 = -Ds^4 (cbf_neg_ds4).
#pragma clang fp contract(off)
template <typename T> struct Pair {
  T x, y;
};
template <typename T> MDS_HD Pair<T> operator-(Pair<T> a, Pair<T> b) { return {a.x - b.x, a.y - b.y}; }

template <typename T> MDS_HD T cbf_neg_ds4(T Ds) {
  const T Ds2 = Ds * Ds;
  return -(Ds2 * Ds2);
}

// N rows at once, statement by statement across the rows: N independent dependency chains side by side in program order (the
// persistent rollout kernel runs one wave per env and is bound by the latency of such chains, not by arithmetic throughput).
template <typename T, int N>
MDS_HD void cbf_row_o2_pairs(const CbfParams<T>& P, const Pair<T> (&exy)[N], const Pair<T> (&dpr)[N], const Pair<T> (&dvxy)[N],
                             const Pair<T> (&ezvz)[N], const T (&nDs4)[N], T (&h_row)[N], T (&Lg0)[N]) {
#define MDS_ROWS(stmt)          \
  _Pragma("unroll") for (int r = 0; r < N; ++r) { stmt; }
  T ex2[N], ey2[N], ez2[N], s[N], ezc[N], ezc2[N], h[N], s4[N], gx[N], gy[N], gz[N], Hxx[N], Hyy[N], Hxy2[N], Hzz[N], dax[N], day[N];
  T hdot[N], quad[N], Lf2[N], vxx[N], vxy[N], vyy[N], vzz[N];
  MDS_ROWS(ex2[r] = exy[r].x * exy[r].x)
  MDS_ROWS(ey2[r] = exy[r].y * exy[r].y)
  MDS_ROWS(ez2[r] = ezvz[r].x * ezvz[r].x)
  MDS_ROWS(s[r] = ex2[r] + ey2[r])
  MDS_ROWS(ezc[r] = ezvz[r].x * P.inv_zscale)
  MDS_ROWS(ezc2[r] = ezc[r] * ezc[r])
  MDS_ROWS(h[r] = m_fma(ezc2[r], ezc2[r], nDs4[r]))
  MDS_ROWS(h[r] = m_fma(s[r], s[r], h[r]))
  MDS_ROWS(s4[r] = T(4) * s[r])
  MDS_ROWS(gx[r] = exy[r].x * s4[r])
  MDS_ROWS(gy[r] = exy[r].y * s4[r])
  MDS_ROWS(gz[r] = (P.c4x4 * ez2[r]) * ezvz[r].x)
  MDS_ROWS(Hxx[r] = m_fma(T(8), ex2[r], s4[r]))
  MDS_ROWS(Hyy[r] = m_fma(T(8), ey2[r], s4[r]))
  MDS_ROWS(Hxy2[r] = (T(16) * exy[r].x) * exy[r].y)
  MDS_ROWS(Hzz[r] = P.c4x12 * ez2[r])
  MDS_ROWS(dax[r] = P.g * dpr[r].x)                            // g d_pitch
  MDS_ROWS(day[r] = P.g * dpr[r].y)                            // -g d_roll
  MDS_ROWS(hdot[r] = gz[r] * ezvz[r].y)
  MDS_ROWS(hdot[r] = m_fma(gy[r], dvxy[r].y, hdot[r]))
  MDS_ROWS(hdot[r] = m_fma(gx[r], dvxy[r].x, hdot[r]))
  MDS_ROWS(vxx[r] = dvxy[r].x * dvxy[r].x)
  MDS_ROWS(vxy[r] = dvxy[r].x * dvxy[r].y)
  MDS_ROWS(vyy[r] = dvxy[r].y * dvxy[r].y)
  MDS_ROWS(vzz[r] = ezvz[r].y * ezvz[r].y)
  MDS_ROWS(quad[r] = Hzz[r] * vzz[r])
  MDS_ROWS(quad[r] = m_fma(Hyy[r], vyy[r], quad[r]))
  MDS_ROWS(quad[r] = m_fma(Hxy2[r], vxy[r], quad[r]))
  MDS_ROWS(quad[r] = m_fma(Hxx[r], vxx[r], quad[r]))
  MDS_ROWS(Lf2[r] = m_fma(gy[r], day[r], quad[r]))
  MDS_ROWS(Lf2[r] = m_fma(gx[r], dax[r], Lf2[r]))
  MDS_ROWS(h_row[r] = m_fma(P.k[1], hdot[r], Lf2[r]))
  MDS_ROWS(h_row[r] = m_fma(P.k[0], h[r], h_row[r]))
  MDS_ROWS(Lg0[r] = gz[r] * P.inv_m)
#undef MDS_ROWS
}

template <typename T>
MDS_HD void cbf_row_o2_pairs(const CbfParams<T>& P, Pair<T> exy, Pair<T> dpr, Pair<T> dvxy, Pair<T> ezvz, T nDs4, T* h_row, T* Lg0) {
  const Pair<T> a[1] = {exy}, b[1] = {dpr}, c[1] = {dvxy}, d[1] = {ezvz};
  const T n[1] = {nDs4};
  T hr[1], lg[1];
  cbf_row_o2_pairs<T, 1>(P, a, b, c, d, n, hr, lg);
  *h_row = hr[0];
  *Lg0 = lg[0];
}

template <typename T>
MDS_HD void cbf_row_o2(const CbfParams<T>& P, T ex, T ey, T ez, T dr, T dp, T dvx, T dvy, T dvz, T Ds, T* h_row, T* Lg0) {
  cbf_row_o2_pairs<T>(P, Pair<T>{ex, ey}, Pair<T>{dp, -dr}, Pair<T>{dvx, dvy}, Pair<T>{ez, dvz}, cbf_neg_ds4(Ds), h_row, Lg0);
}
#pragma clang fp contract(fast)

// One ECBF row from e = pos_i - pos_j (ACTUAL positions) and d = (x_i - xdes_i) - (x_j - xdes_j), the difference of the two
// tracking errors in the model's state layout (d = x_i - xdes_i against an obstacle, cbf/cbf.py:380-392):
//   h_row = Kcbf . hdots + L_f^r h,   Lg[4] with G[4i:4i+4] = -Lg, G[4j:4j+4] = +Lg.
template <typename T, int ORDER>
MDS_HD void cbf_row_core(const CbfParams<T>& P, T ex, T ey, T ez, const T* d, T Ds, T* h_row, T Lg[4]) {
  if (ORDER == 2) {
    cbf_row_o2(P, ex, ey, ez, d[0], d[1], d[3], d[4], d[5], Ds, h_row, &Lg[0]);
    Lg[1] = T(0);
    Lg[2] = T(0);
    Lg[3] = T(0);
  } else {
    const T s = m_fma(ex, ex, ey * ey);
    const T ezc = ez * P.inv_zscale;
    const T ezc2 = ezc * ezc;
    const T Ds2 = Ds * Ds;
    const T h = m_fma(s, s, m_fma(ezc2, ezc2, -(Ds2 * Ds2)));
    const T gx = T(4) * ex * s, gy = T(4) * ey * s, gz = T(4) * ez * ez * ez * P.inv_c4;
    const T Hxx = T(12) * ex * ex + T(4) * ey * ey, Hxy = T(8) * ex * ey, Hyy = T(4) * ex * ex + T(12) * ey * ey,
            Hzz = T(12) * ez * ez * P.inv_c4;
    constexpr int o3 = ORDER == 2 ? 0 : 1;   // keeps the order-3 slots in range when instantiated for order 2
    const T dr = d[0], dp = d[1], dF = d[3], dvx = d[3 + o3], dvy = d[4 + o3], dvz = d[5 + o3];
    const T dax = P.g * dp, day = -P.g * dr, daz = dF * P.inv_m;
    const T hdot = m_fma(gx, dvx, m_fma(gy, dvy, gz * dvz));
    // custom_hdots i == 2 with slots 6,7,8 = (vz, x, y) of the 10-state (quirk kept)
    const T w0 = dF * P.inv_m, w1 = dvx, w2 = dvy;
    const T hddot_ref = P.g * (gy * dp - gz * dr) + (Hxx * w0 * w0 + T(2) * Hxy * w0 * w1 + Hyy * w1 * w1 + Hzz * w2 * w2);
    const T Hdv_da = Hxx * dvx * dax + Hxy * (dvx * day + dvy * dax) + Hyy * dvy * day + Hzz * dvz * daz;
    const T T3 = T(24) * ex * dvx * dvx * dvx + T(24) * ey * dvx * dvx * dvy + T(24) * ex * dvx * dvy * dvy +
                 T(24) * ey * dvy * dvy * dvy + (T(24) * ez * P.inv_c4) * dvz * dvz * dvz;
    const T Lf3 = T(3) * Hdv_da + T3;
    *h_row = P.k[0] * h + P.k[1] * hdot + P.k[2] * hddot_ref + Lf3;
    Lg[0] = gz * P.inv_m;
    Lg[1] = -P.g * gy;
    Lg[2] = P.g * gx;
    Lg[3] = T(0);
  }
}

// The same row between agent i (state xi, desired xdi) and agent / obstacle j from the states themselves; for an obstacle the
// caller passes xj = xdj = (0.., position) and obstacle = true.
template <typename T, int ORDER>
MDS_HD void cbf_pair_row(const CbfParams<T>& P, const T* xi, const T* xdi, const T* xj, const T* xdj, bool obstacle, T Ds,
                         T* h_row, T Lg[4]) {
  constexpr int xd = ORDER == 2 ? 9 : 10;
  const T ex = xi[xd - 3] - xj[xd - 3], ey = xi[xd - 2] - xj[xd - 2], ez = xi[xd - 1] - xj[xd - 1];
  T d[xd];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < xd; ++k) d[k] = (xi[k] - xdi[k]) - (obstacle ? T(0) : (xj[k] - xdj[k]));
  cbf_row_core<T, ORDER>(P, ex, ey, ez, d, Ds, h_row, Lg);
}

// utils/model_conversions.py:20-58 obs_to_lin_model(obs, dim = 9 | 10) for one drone
template <typename T> MDS_HD void obs_to_lin(const T* obs20, int order, T kf, T* x) {
  x[0] = obs20[7]; x[1] = obs20[8]; x[2] = obs20[9];
  int o = 3;
  if (order == 3) {  // calc_z_thrust (:137-143)
    x[3] = kf * (obs20[16] * obs20[16] + obs20[17] * obs20[17] + obs20[18] * obs20[18] + obs20[19] * obs20[19]);
    o = 4;
  }
  x[o] = obs20[10]; x[o + 1] = obs20[11]; x[o + 2] = obs20[12];
  x[o + 3] = obs20[0]; x[o + 4] = obs20[1]; x[o + 5] = obs20[2];
}

// row bookkeeping of CBF._build_ineq_const (cbf/cbf.py:308-367)
MDS_HD int cbf_num_pairs(int D) { return D * (D - 1) / 2; }
MDS_HD int cbf_num_rows(int D, int order, int n_obs) { return cbf_num_pairs(D) + 8 * D + (order == 3 ? 2 * D : 0) + D * n_obs; }

}  // namespace mds
