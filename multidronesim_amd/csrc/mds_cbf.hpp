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
// instantiates it rounds alike); 38 instructions.  A version of this body on a float ext_vector_type(2) (v_pk_mul_f32 /
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

template <typename T>
MDS_HD void cbf_row_o2_pairs(const CbfParams<T>& P, Pair<T> exy, Pair<T> dpr, Pair<T> dvxy, Pair<T> ezvz, T nDs4, T* h_row, T* Lg0) {
  const T ex = exy.x, ey = exy.y, ez = ezvz.x, dvx = dvxy.x, dvy = dvxy.y, dvz = ezvz.y;
  const T ex2 = ex * ex, ey2 = ey * ey, ez2 = ez * ez;
  const T s = ex2 + ey2;
  const T ezc = ez * P.inv_zscale;
  const T ezc2 = ezc * ezc;
  const T h = m_fma(s, s, m_fma(ezc2, ezc2, nDs4));
  const T s4 = T(4) * s;
  const T gx = ex * s4, gy = ey * s4, gz = (P.c4x4 * ez2) * ez;
  const T Hxx = m_fma(T(8), ex2, s4), Hyy = m_fma(T(8), ey2, s4), Hxy2 = (T(16) * ex) * ey, Hzz = P.c4x12 * ez2;
  const T dax = P.g * dpr.x, day = P.g * dpr.y;                 // g d_pitch, -g d_roll
  const T hdot = m_fma(gx, dvx, m_fma(gy, dvy, gz * dvz));
  const T quad = m_fma(Hxx, dvx * dvx, m_fma(Hxy2, dvx * dvy, m_fma(Hyy, dvy * dvy, Hzz * (dvz * dvz))));
  const T Lf2 = m_fma(gx, dax, m_fma(gy, day, quad));
  *h_row = m_fma(P.k[0], h, m_fma(P.k[1], hdot, Lf2));
  *Lg0 = gz * P.inv_m;
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
