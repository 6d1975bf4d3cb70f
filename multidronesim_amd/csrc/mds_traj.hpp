// Generic trajectory table: the reference's trajectories/ family (Lemniscate, CircleTrajectory,
// LineTrajectory, WaitTrajectory, CompoundTrajectory, RotateTrajectory) flattened on the host into
// per-drone lists of SEGMENTS and evaluated on the device in double (this is the general path;
// the all-Lemniscate case keeps its own fused fp32 kernel).
//
// segment = MDS_SEG_DIM doubles:
//   [0] kind (0 Lemniscate, 1 Circle, 2 Line, 3 Wait)   [1] t_start  [2] t_end   (cumulative, Compound)
//   [3..26] parameters (below)   [27..35] A row-major, [36..38] b : pos' = A pos + b, vel' = A vel
//   (RotateTrajectory.py:19-25 as an affine map; identity when not rotated)   [39] unused
// Reference lines: Lemniscate.py:32-63, Circle.py:24-45, LineTrajectory.py:4-15 and :71-104,
// CompoundTrajectory.py:26-40.
#pragma once
#include "mds_math.hpp"

namespace mds {

constexpr int kSegDim = 40;
constexpr int kSegParam = 3, kSegA = 27, kSegB = 36;

// local evaluation of one segment at its own time tl -> out[11] = pos3 vel3 acc3 yaw yaw_rate
MDS_HD void traj_segment_eval(const double* sg, double tl, double out[11]) {
  const double* p = sg + kSegParam;
  const int kind = (int)sg[0];
  for (int k = 0; k < 11; ++k) out[k] = 0.0;
  if (kind == 0) {          // Lemniscate: a, omega, cx, cy, cz, yaw_rate, phase_shift
    const double a = p[0], om = p[1], th = tl * om + p[6];
    const double s = sin(th), c = cos(th), s2 = s * s, c2 = c * c, den = 1.0 + s2;
    out[0] = p[2] + a * s * c / den;
    out[1] = p[3] + a * c / den;
    out[2] = p[4];
    out[3] = -a * om * (s2 * s2 + s2 + (s2 - 1.0) * c2) / (den * den);
    out[4] = -a * om * s * (s2 + 2.0 * c2 + 1.0) / (den * den);
    const double c2t = cos(2.0 * th), e = c2t - 3.0;
    out[6] = 4.0 * a * om * om * sin(2.0 * th) * (3.0 * c2t + 7.0) / (e * e * e);
    out[7] = a * om * om * c * (44.0 * c2t + cos(4.0 * th) - 21.0) / (e * e * e);
    out[9] = 3.14159265358979323846 * sin(p[5] * tl);
    out[10] = 3.14159265358979323846 * p[5] * cos(p[5] * tl);
  } else if (kind == 1) {   // Circle: r, v, cx, cy, cz, yaw_rate
    const double r = p[0], v = p[1], w = v / r, s = sin(w * tl), c = cos(w * tl);
    out[0] = p[2] + r * c;
    out[1] = p[3] + r * s;
    out[2] = p[4];
    out[3] = -v * s;
    out[4] = v * c;
    out[6] = -(v * v) / r * c;
    out[7] = -(v * v) / r * s;
    const double x = p[5] * tl - 3.14159265358979323846, twopi = 6.283185307179586476925;
    out[9] = (x - twopi * floor(x / twopi)) + 3.14159265358979323846;     // Python % (Circle.py:28)
    out[10] = p[5];
  } else if (kind == 3) {   // Wait: px, py, pz, yaw
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
    out[9] = p[3];
  } else {                  // Line: start3, end3, v0 3, vf 3, sign_init3, sign_end3, v_mid3, t_init, t_mid, total  (a_max = 1)
    const double ti = p[21], tm = p[22], total = p[23];
    if (tl > total) {
      for (int k = 0; k < 3; ++k) { out[k] = p[3 + k]; out[3 + k] = p[9 + k]; }
    } else if (tl < ti) {
      for (int k = 0; k < 3; ++k) {
        out[k] = p[k] + p[6 + k] * tl + 0.5 * p[12 + k] * tl * tl;
        out[3 + k] = p[6 + k] + p[12 + k] * tl;
        out[6 + k] = p[12 + k];
      }
    } else if (tl < tm + ti) {
      const double t2 = tl - ti;
      for (int k = 0; k < 3; ++k) {
        out[k] = p[k] + (p[6 + k] * ti + 0.5 * p[12 + k] * ti * ti) + p[18 + k] * t2;
        out[3 + k] = p[18 + k];
      }
    } else {
      const double t3 = tl - tm - ti;
      for (int k = 0; k < 3; ++k) {
        const double dpm = (p[6 + k] * ti + 0.5 * p[12 + k] * ti * ti) + p[18 + k] * tm;
        out[k] = p[k] + dpm + p[18 + k] * t3 + 0.5 * p[15 + k] * t3 * t3;
        out[3 + k] = p[18 + k] + p[15 + k] * t3;
        out[6 + k] = p[15 + k];
      }
    }
  }
  // RotateTrajectory as an affine map
  const double* A = sg + kSegA;
  const double* b = sg + kSegB;
  double r[9];
  for (int g = 0; g < 3; ++g)
    for (int k = 0; k < 3; ++k) r[3 * g + k] = A[3 * k] * out[3 * g] + A[3 * k + 1] * out[3 * g + 1] + A[3 * k + 2] * out[3 * g + 2];
  for (int k = 0; k < 3; ++k) {
    out[k] = r[k] + b[k];
    out[3 + k] = r[3 + k];
    out[6 + k] = r[6 + k];
  }
}

// CompoundTrajectory.__call__ (stateless form): past the end -> last piece at its own end time; else the
// first piece whose cumulative end time is >= t.  A single (non-compound) trajectory is evaluated at t.
MDS_HD void traj_eval(const double* segs, int first, int nseg, int compound, double t, double out[11]) {
  const double* last = segs + (size_t)(first + nseg - 1) * kSegDim;
  if (compound && t >= last[2]) {
    traj_segment_eval(last, last[2] - last[1], out);
    return;
  }
  int k = 0;
  if (compound)
    while (k < nseg - 1 && t > segs[(size_t)(first + k) * kSegDim + 2]) ++k;
  const double* sg = segs + (size_t)(first + k) * kSegDim;
  traj_segment_eval(sg, t - sg[1], out);
}

}  // namespace mds
