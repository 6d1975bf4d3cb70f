// Host-side set-up of the per-launch constant block (computed in double, rounded once).
// Shared by the C-ABI (mds_api.hip) and the test-only host emulation (tests/emul).
#pragma once
#include <math.h>

#include "../../include/mds.h"
#include "mds_math.hpp"
#include "mds_cbf.hpp"

namespace mds {

template <typename T> inline void fill_consts(const mds_config& cfg, const mds_geometric_gains& g, Consts<T>& c, const double* wind = nullptr) {
  for (int k = 0; k < 3; ++k) c.wind[k] = wind ? (T)wind[k] : T(0);
  c.kf = (T)cfg.KF;
  c.km = (T)cfg.KM;
  c.arm = (T)cfg.L;
  c.mass = (T)cfg.M;
  c.inv_mass = (T)(1.0 / cfg.M);
  c.gravity = (T)(cfg.G * cfg.M);
  c.max_rpm = (T)sqrt(cfg.thrust2weight * cfg.G * cfg.M / (4.0 * cfg.KF));
  c.hover_rpm = (T)sqrt(cfg.G * cfg.M / (4.0 * cfg.KF));
  c.thrust_corr = (T)(4.0 * (double)c.kf * (double)c.hover_rpm * (double)c.hover_rpm - (double)c.gravity);
  for (int k = 0; k < 3; ++k) {
    c.J[k] = (T)cfg.J[k];
    c.invJ[k] = (T)(1.0 / cfg.J[k]);
    c.drag[k] = (T)cfg.drag_coeff[k];
    c.kp[k] = (T)g.Kp[k];
    c.kv[k] = (T)g.Kv[k];
    c.kR[k] = (T)g.KR[k];
    c.kw[k] = (T)g.Kw[k];
  }
  c.dt = (T)(1.0 / cfg.pyb_freq);
  c.substeps = cfg.pyb_freq / cfg.ctrl_freq;
  c.cf2x = cfg.drone_model == MDS_CF2X;
  c.use_drag = cfg.physics == MDS_PHYSICS_DYN_DRAG || cfg.physics == MDS_PHYSICS_DYN_GND_DRAG_DW;
  c.rk4 = cfg.integrator == MDS_INTEGRATOR_RK4;
  c.g_ctrl = (T)g.g;
  c.cos_max_tilt = (T)cos(g.max_tilt_angle);
  c.tan_max_tilt = (T)tan(g.max_tilt_angle);
  const double max_rpm = sqrt(cfg.thrust2weight * cfg.G * cfg.M / (4.0 * cfg.KF));
  c.min_motor_thrust = (T)(9440.3 * 9440.3 * cfg.KF);   // utils/model_conversions.py:99
  c.max_motor_thrust = (T)(4.0 * cfg.KF * max_rpm * max_rpm);
  c.inv_2L = (T)(1.0 / (2.0 * cfg.L));
  c.inv_4r = (T)(cfg.KF / (4.0 * cfg.KM));
  c.inv_kf = (T)(1.0 / cfg.KF);
}

// [UPSTREAM] urdf <properties> (identical in cf2x.urdf and cf2p.urdf) and BaseAviary.__init__ GND_EFF_H_CLIP
template <typename T> inline void fill_envfx(const mds_config& cfg, EnvFx<T>& fx) {
  fx.gnd = cfg.physics == MDS_PHYSICS_DYN_GND || cfg.physics == MDS_PHYSICS_DYN_GND_DRAG_DW;
  fx.dw = cfg.physics == MDS_PHYSICS_DYN_DW || cfg.physics == MDS_PHYSICS_DYN_GND_DRAG_DW;
  const double gnd = 11.36859, rad = 2.31348e-2;
  const double max_rpm2 = cfg.thrust2weight * cfg.G * cfg.M / (4.0 * cfg.KF), max_thrust = 4.0 * cfg.KF * max_rpm2;
  fx.gnd_coeff = (T)gnd;
  fx.prop_radius = (T)rad;
  fx.h_clip = (T)(0.25 * rad * sqrt(15.0 * max_rpm2 * cfg.KF * gnd / max_thrust));
  fx.dw1 = (T)2267.18;
  fx.dw2 = (T)0.16;
  fx.dw3 = (T)-0.11;
  if (cfg.drone_model == MDS_CF2X) {
    const double x[4] = {0.028, -0.028, -0.028, 0.028}, y[4] = {-0.028, -0.028, 0.028, 0.028};
    for (int k = 0; k < 4; ++k) { fx.prop_x[k] = (T)x[k]; fx.prop_y[k] = (T)y[k]; }
  } else {
    const double x[4] = {cfg.L, 0.0, -cfg.L, 0.0}, y[4] = {0.0, cfg.L, 0.0, -cfg.L};
    for (int k = 0; k < 4; ++k) { fx.prop_x[k] = (T)x[k]; fx.prop_y[k] = (T)y[k]; }
  }
}

// the ECBF parameter block of the CBF kernels (shared with the test-only SIMT emulation)
template <typename T> inline void fill_cbf_params(const mds_config& cfg, const mds_cbf_params& p, CbfParams<T>& o) {
  o.order = p.order;
  o.n_obs = p.n_obs;
  o.num_drones = cfg.num_drones;
  for (int k = 0; k < 3; ++k) o.k[k] = (T)p.Kcbf[k];
  for (int k = 0; k < 4; ++k) o.umax[k] = (T)p.umax[k];
  o.Ds_pair = (T)(2.0 * p.safety_radius);
  o.safety_radius = (T)p.safety_radius;
  o.zscale = (T)p.zscale;
  o.inv_zscale = (T)(1.0 / p.zscale);
  o.obs_magic = p.n_obs > 0 ? (65536 + p.n_obs - 1) / p.n_obs : 0;
  o.inv_c4 = (T)(1.0 / (p.zscale * p.zscale * p.zscale * p.zscale));
  o.c4x4 = (T)(4.0 / (p.zscale * p.zscale * p.zscale * p.zscale));
  o.c4x12 = (T)(12.0 / (p.zscale * p.zscale * p.zscale * p.zscale));
  o.inv_m = (T)(1.0 / cfg.M);
  o.g = (T)cfg.G;
  o.Fmin = (T)p.Fmin;
  o.Fmax = (T)p.Fmax;
}

}  // namespace mds
