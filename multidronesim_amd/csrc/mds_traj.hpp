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

// Device storage is FIELD-MAJOR: field k of segment id lives at base[k * ns + id].  Row-major rows (320 B per segment) make
// every lane of a wave touch its own cache lines on every field read -- 64 L1 lookups per load instruction, and the step
// kernel became L1-tag bound (Line trajectories: 64 us per C3 step).  Field-major, lanes that share a table or own
// neighbouring ones read the same or adjacent words.
struct SegTable {
  const double* base;
  int ns;
};
struct SegRef {              // one segment (optionally shifted to a field offset); sg[k] reads field k
  const double* col;
  size_t ns;
  MDS_HD double operator[](int k) const { return col[(size_t)k * ns]; }
  MDS_HD SegRef operator+(int k) const { return SegRef{col + (size_t)k * ns, ns}; }
};
MDS_HD SegRef seg_ref(SegTable tb, int id) { return SegRef{tb.base + id, (size_t)tb.ns}; }

// per-drone entry of the table index (3 ints): piece k of the drone is segment first + k * stride.  Tables with the same
// number of pieces are stored piece-major in one block (stride = tables in the block), so that neighbouring drones on
// the same piece read neighbouring words.
struct TrajInfo {
  int first, nseg, compound, stride;
};
MDS_HD TrajInfo traj_info(const int* tinfo, int i) {
  const int a = tinfo[3 * i], b = tinfo[3 * i + 1], c = tinfo[3 * i + 2];
  return TrajInfo{a, b & 0xffff, b >> 16, c};
}
// kind field of a segment: kind | 8 when the affine map (RotateTrajectory) is not the identity
constexpr int kSegAffine = 8;

// local evaluation of one segment at its own time tl -> out[11] = pos3 vel3 acc3 yaw yaw_rate
MDS_HD void traj_segment_eval(SegRef sg, double tl, double out[11]) {
  const SegRef p = sg + kSegParam;
  const int kf = (int)sg[0], kind = kf & 7;
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
  if (!(kf & kSegAffine)) return;
  const SegRef A = sg + kSegA, b = sg + kSegB;
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
MDS_HD void traj_eval(SegTable segs, TrajInfo ti, double t, double out[11]) {
  const SegRef last = seg_ref(segs, ti.first + (ti.nseg - 1) * ti.stride);
  if (ti.compound && t >= last[2]) {
    traj_segment_eval(last, last[2] - last[1], out);
    return;
  }
  int k = 0;
  if (ti.compound)
    while (k < ti.nseg - 1 && t > seg_ref(segs, ti.first + k * ti.stride)[2]) ++k;
  const SegRef sg = seg_ref(segs, ti.first + k * ti.stride);
  traj_segment_eval(sg, t - sg[1], out);
}


// ---- the same evaluation for the step kernels: Desired<T> relative to the drone's local-frame origin ----
// T = double: the evaluation above, unchanged.  T = float: every phase (omega t + shift, v t / r, yaw_rate t) and every
// absolute position (centres, line end points, the affine offset, the origin) is formed in double, the periodic part is
// reduced to [-pi, pi] in double, and only then does fp32 take over (hardware-rate sincos, the small offsets around the
// centre, velocities and accelerations) -- the split lemniscate_local makes.  The all-double version spends 7 f64
// sin/cos per Lemniscate and 2 per Circle: 53 us per C3 step against 17.7 us for the Lemniscate planes.
MDS_HD float traj_red(double ph) {
  const double k = rint(ph * 0.15915494309189533577);
  return (float)fma(k, -6.283185307179586476925, ph);
}

template <typename T> struct TrajLocal;

template <> struct TrajLocal<double> {
  static MDS_HD Desired<double> eval(SegTable segs, TrajInfo ti, double t, V3<double> org) {
    double d[11];
    traj_eval(segs, ti, t, d);
    Desired<double> des;
    des.p = {d[0] - org.x, d[1] - org.y, d[2] - org.z};
    des.v = {d[3], d[4], d[5]};
    des.a = {d[6], d[7], d[8]};
    des.yaw = d[9];
    des.yaw_rate = d[10];
    return des;
  }
};

template <> struct TrajLocal<float> {
  static MDS_HD Desired<float> eval(SegTable segs, TrajInfo ti, double t, V3<float> org) {
    // piece selection: CompoundTrajectory.__call__ as in traj_eval
    SegRef sg = seg_ref(segs, ti.first + (ti.nseg - 1) * ti.stride);
    double tl;
    if (ti.compound && t >= sg[2]) {
      tl = sg[2] - sg[1];
    } else {
      int k = 0;
      if (ti.compound)
        while (k < ti.nseg - 1 && t > seg_ref(segs, ti.first + k * ti.stride)[2]) ++k;
      sg = seg_ref(segs, ti.first + k * ti.stride);
      tl = t - sg[1];
    }
    const SegRef p = sg + kSegParam;
    const int kf = (int)sg[0], kind = kf & 7;
    double base[3] = {p[2], p[3], p[4]};            // Lemniscate / Circle centre; overwritten by Line / Wait
    float dl[3] = {0.f, 0.f, 0.f}, v[3] = {0.f, 0.f, 0.f}, a[3] = {0.f, 0.f, 0.f};
    double yaw = 0.0;
    float yaw_f = 0.f, yaw_rate = 0.f;
    bool yaw_is_f = false;
    if (kind == 0) {
      const float A = (float)p[0], om = (float)p[1];
      float s, c;
      m_sincos(traj_red(fma(tl, p[1], p[6])), &s, &c);
      const float s2 = s * s, c2 = c * c, inv = m_rcp(1.f + s2), inv2 = inv * inv, aw = A * om;
      dl[0] = A * s * c * inv;
      dl[1] = A * c * inv;
      v[0] = -aw * (s2 * s2 + s2 + (s2 - 1.f) * c2) * inv2;
      v[1] = -aw * s * (s2 + 2.f * c2 + 1.f) * inv2;
      const float sin2 = 2.f * s * c, cos2 = c2 - s2, cos4 = 1.f - 2.f * sin2 * sin2, e = cos2 - 3.f;
      const float inv3 = m_rcp(e * e * e), aw2 = aw * om;
      a[0] = 4.f * aw2 * sin2 * (3.f * cos2 + 7.f) * inv3;
      a[1] = aw2 * c * (44.f * cos2 + cos4 - 21.f) * inv3;
      float sy, cy;
      m_sincos(traj_red(p[5] * tl), &sy, &cy);
      yaw_f = 3.14159265358979323846f * sy;
      yaw_rate = 3.14159265358979323846f * (float)p[5] * cy;
      yaw_is_f = true;
    } else if (kind == 1) {
      const float r = (float)p[0], vv = (float)p[1];
      float s, c;
      m_sincos(traj_red(p[1] / p[0] * tl), &s, &c);
      dl[0] = r * c;
      dl[1] = r * s;
      v[0] = -vv * s;
      v[1] = vv * c;
      const float cen = vv * vv * m_rcp(r);
      a[0] = -cen * c;
      a[1] = -cen * s;
      const double x = p[5] * tl - 3.14159265358979323846, twopi = 6.283185307179586476925;
      yaw = (x - twopi * floor(x / twopi)) + 3.14159265358979323846;
      yaw_rate = (float)p[5];
    } else if (kind == 3) {
      base[0] = p[0]; base[1] = p[1]; base[2] = p[2];
      yaw = p[3];
    } else {
      const double ti = p[21], tm = p[22], total = p[23];
      if (tl > total) {
        for (int k = 0; k < 3; ++k) { base[k] = p[3 + k]; v[k] = (float)p[9 + k]; }
      } else if (tl < ti) {
        for (int k = 0; k < 3; ++k) {
          base[k] = p[k] + p[6 + k] * tl + 0.5 * p[12 + k] * tl * tl;
          v[k] = (float)(p[6 + k] + p[12 + k] * tl);
          a[k] = (float)p[12 + k];
        }
      } else if (tl < tm + ti) {
        const double t2 = tl - ti;
        for (int k = 0; k < 3; ++k) {
          base[k] = p[k] + (p[6 + k] * ti + 0.5 * p[12 + k] * ti * ti) + p[18 + k] * t2;
          v[k] = (float)p[18 + k];
        }
      } else {
        const double t3 = tl - tm - ti;
        for (int k = 0; k < 3; ++k) {
          const double dpm = (p[6 + k] * ti + 0.5 * p[12 + k] * ti * ti) + p[18 + k] * tm;
          base[k] = p[k] + dpm + p[18 + k] * t3 + 0.5 * p[15 + k] * t3 * t3;
          v[k] = (float)(p[18 + k] + p[15 + k] * t3);
          a[k] = (float)p[15 + k];
        }
      }
    }
    // RotateTrajectory: pos' = A (base + dl) + b; the absolute part in double, the offsets in fp32
    const double orgd[3] = {(double)org.x, (double)org.y, (double)org.z};
    float pr[3], vr[3], ar[3];
    if (kf & kSegAffine) {
      const SegRef Ad = sg + kSegA, b = sg + kSegB;
      for (int k = 0; k < 3; ++k) {
        const float a0 = (float)Ad[3 * k], a1 = (float)Ad[3 * k + 1], a2 = (float)Ad[3 * k + 2];
        const double pb = Ad[3 * k] * base[0] + Ad[3 * k + 1] * base[1] + Ad[3 * k + 2] * base[2] + b[k] - orgd[k];
        pr[k] = (float)pb + (a0 * dl[0] + a1 * dl[1] + a2 * dl[2]);
        vr[k] = a0 * v[0] + a1 * v[1] + a2 * v[2];
        ar[k] = a0 * a[0] + a1 * a[1] + a2 * a[2];
      }
    } else {
      for (int k = 0; k < 3; ++k) {
        pr[k] = (float)(base[k] - orgd[k]) + dl[k];
        vr[k] = v[k];
        ar[k] = a[k];
      }
    }
    Desired<float> des;
    des.p = {pr[0], pr[1], pr[2]};
    des.v = {vr[0], vr[1], vr[2]};
    des.a = {ar[0], ar[1], ar[2]};
    des.yaw = yaw_is_f ? yaw_f : traj_red(yaw);
    des.yaw_rate = yaw_rate;
    return des;
  }
};

}  // namespace mds
