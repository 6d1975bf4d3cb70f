// TEST TOOLING ONLY.  The ECBF / QP kernels of multidronesim_amd/csrc -- k_cbf_filter_gi (one wavefront per env: rows, row table, dual
// active-set solver with its thin QR in LDS) and k_cbf_rollout (the persistent rollout kernel: row-slot table, per-drone bounds, ticket
// loop, per-wave LDS slices, observation staging) -- compiled from the .hip sources as HOST C++ against the SIMT stand-in of
// tests/emul/simt/hip/hip_runtime.h and run under AddressSanitizer + UndefinedBehaviorSanitizer: 64 lanes = 64 threads, LDS arrays as
// real arrays of the kernels' exact sizes.  tests/test_simt_sanitizers_cpu.py drives it with crowded scenes and checks the results
// against the plain-C oracle.  Never part of the product (the product path is the HIP library only).
//
//   simt_cbf <in.bin> <out.bin>
// in.bin : int32 mode (1 filter, 2 rollout, 3 order-3 rollout, 4 the headline step / rollout kernels), dtype (0 f32, 1 f64), E, D, n_steps, nominal; mds_config, mds_geometric_gains, mds_cbf_params
//          (raw structs); double obstacles[16*4]; then  filter: obs[n*20], xdes[n*xd], unom[n*4]   rollout: t0, P[n*7], state13[n*13] (world)
// out.bin: filter: double usafe[n*4], int32 status[E], iters[E]      rollout: double obs[n*20], int32 status_log[steps*E], iters[E]
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../../multidronesim_amd/csrc/mds_consts.hpp"
#include "../../../multidronesim_amd/csrc/mds_kernels.hip"
#include "../../../multidronesim_amd/csrc/mds_cbf_kernels.hip"

using namespace mds;

struct In {
  int mode, dtype, E, D, n_steps, nominal;
  mds_config cfg;
  mds_geometric_gains gains;
  mds_cbf_params cbf;
  double obstacles[16 * 4];
  std::vector<double> rest;
};

static bool read_in(const char* path, In& in) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  bool ok = fread(&in.mode, sizeof(int), 6, f) == 6 && fread(&in.cfg, sizeof(in.cfg), 1, f) == 1 && fread(&in.gains, sizeof(in.gains), 1, f) == 1 &&
            fread(&in.cbf, sizeof(in.cbf), 1, f) == 1 && fread(in.obstacles, sizeof(double), 64, f) == 64;
  if (ok) {
    double v;
    while (fread(&v, sizeof(double), 1, f) == 1) in.rest.push_back(v);
  }
  fclose(f);
  return ok;
}

template <typename T> static std::vector<T> to_t(const double* p, size_t n) {
  std::vector<T> v(n);
  for (size_t k = 0; k < n; ++k) v[k] = (T)p[k];
  return v;
}

template <typename T> static int run_filter(const In& in, FILE* out) {
  const int E = in.E, D = in.D, order = in.cbf.order, xd = order == 2 ? 9 : 10, nv = order == 2 ? 1 : 3, n = nv * D;
  const size_t nd = (size_t)E * D;
  if (in.rest.size() != nd * (20 + xd + 4)) return 3;
  CbfParams<T> P;
  fill_cbf_params(in.cfg, in.cbf, P);
  std::vector<int> pair;
  for (int i = 0; i < D - 1; ++i)
    for (int j = i + 1; j < D; ++j) pair.push_back(i | (j << 8));
  pair.push_back(0);
  std::vector<T> obst = to_t<T>(in.obstacles, 64), obs = to_t<T>(in.rest.data(), nd * 20), xdes = to_t<T>(in.rest.data() + nd * 20, nd * xd),
                 unom = to_t<T>(in.rest.data() + nd * (20 + xd), nd * 4), usafe(nd * 4, T(0));
  std::vector<int> status(E, -7), cost(E, -7);
  const int m = D * (D - 1) / 2 + D * in.cbf.n_obs + 2 * n, R = (m + 63) / 64;
  const int max_iter = in.cbf.max_iter > 0 ? in.cbf.max_iter : 64 * m;
  const double tol = in.cbf.tol > 0 ? in.cbf.tol : (sizeof(T) == 8 ? 1e-12 : 1e-6);
  if (n > 64 || R > 17) return 4;
#define GI(RR, NMAX, ORD)                                                                                                            \
  simt::launch((unsigned)E, 64, nullptr, [&]() {                                                                                     \
    k_cbf_filter_gi<T, T, RR, NMAX, ORD>(P, E, (T)in.cfg.KF, pair.data(), obst.data(), obs.data(), xdes.data(), unom.data(), usafe.data(), \
                                         status.data(), max_iter, (T)(tol * tol), nullptr, nullptr, cost.data());                     \
  })
#define GI_R(NMAX, ORD)            \
  do {                             \
    if (R <= 4) GI(4, NMAX, ORD);  \
    else if (R <= 8) GI(8, NMAX, ORD); \
    else GI(17, NMAX, ORD);        \
  } while (0)
  if (order == 2) {
    if (n <= 16) GI_R(16, 2);
    else GI_R(32, 2);
  } else {
    if (n <= 24) GI_R(24, 3);
    else if (n <= 48) GI_R(48, 3);
    else GI_R(63, 3);
  }
#undef GI_R
#undef GI
  std::vector<double> ud(usafe.begin(), usafe.end());
  fwrite(ud.data(), sizeof(double), ud.size(), out);
  fwrite(status.data(), sizeof(int), E, out);
  fwrite(cost.data(), sizeof(int), E, out);
  return 0;
}

template <typename T> static int run_rollout(const In& in, FILE* out) {
  const int E = in.E, D = in.D, n = E * D, steps = in.n_steps;
  if (in.cbf.order != 2 || D < 1 || D > 16 || in.rest.size() != (size_t)1 + (size_t)n * 20) return 3;
  const double t0 = in.rest[0];
  const double* Pd = in.rest.data() + 1;
  const double* st13 = Pd + (size_t)n * 7;
  const size_t ld = ((size_t)n + 255) / 256 * 256;
  RollArgs<T> ra;
  memset(&ra, 0, sizeof(ra));
  fill_consts(in.cfg, in.gains, ra.p.c);
  fill_cbf_params(in.cfg, in.cbf, ra.p.P);
  // the library's layouts: packed state planes (sidx), packed Lemniscate planes (lidx), local frame = the trajectory centre
  std::vector<T> state(13 * ld + 16, T(0)), lem(7 * ld + 16, T(0)), ll(6 * ld, T(0)), obst = to_t<T>(in.obstacles, 64), obs_last((size_t)n * 20, T(0)),
                 ring((size_t)3 * n * 20, T(0));
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 7; ++k) lem[lidx(k, i, ld)] = (T)Pd[(size_t)7 * i + k];
    const double* s = st13 + (size_t)13 * i;
    for (int k = 0; k < 13; ++k) state[sidx(k, i, ld)] = (T)(k < 3 ? s[k] - Pd[(size_t)7 * i + 2 + k] : s[k]);
  }
  std::vector<int> pair;
  for (int i = 0; i < D - 1; ++i)
    for (int j = i + 1; j < D; ++j) pair.push_back(i | (j << 8));
  pair.push_back(0);
  std::vector<int> status(E, -7), slog((size_t)steps * E, -7), cost(E, 0);
  const double tol = in.cbf.tol > 0 ? in.cbf.tol : (sizeof(T) == 8 ? 1e-12 : 1e-6);
  const int m2 = D * (D - 1) / 2 + 2 * D + D * in.cbf.n_obs;
  ra.Kp = nullptr; ra.n = n; ra.ld = ld; ra.E = E; ra.ctrl_dt = 1.0 / in.cfg.ctrl_freq;
  ra.state = state.data(); ra.state_lo = nullptr; ra.lem = lem.data(); ra.last_rpm = nullptr; ra.ll = ll.data(); ra.pair_ij = pair.data();
  ra.obstacles = obst.data(); ra.obs_log = ring.data(); ra.n_slots = 3; ra.obs_last = obs_last.data(); ra.status = status.data();
  ra.cost_io = cost.data(); ra.max_iter = in.cbf.max_iter > 0 ? in.cbf.max_iter : 64 * m2; ra.tol2 = (T)(tol * tol); ra.tol = (T)tol; ra.stamps = nullptr;
#ifndef SIMT_NW
#define SIMT_NW 1
#endif
  constexpr int NW = SIMT_NW;                                    // wavefronts per workgroup (1: 64 threads own 64 / Dp whole envs; the TSan build runs 2)
  const int Dp = D <= 4 ? 4 : (D <= 8 ? 8 : 16), envs_per_wg = 64 * NW / Dp, grid = (E + envs_per_wg - 1) / envs_per_wg;
  // launches of 7 steps (4 in runs of <= 5; a ragged last one), the observation ring of 3 slots carried across them -- as mds_rollout_cbf_geometric_fused does
  double t = t0;
  int slot = 0;
  const int per_launch = steps <= 5 ? 4 : 7;                   // (short runs still cross a launch boundary)
  for (int k0 = 0; k0 < steps; k0 += per_launch) {
    const int ks = steps - k0 < per_launch ? steps - k0 : per_launch;
    ra.t = t; ra.n_steps = ks; ra.slot = slot; ra.status_log = slog.data() + (size_t)k0 * E;
    if (D == Dp) simt::launch((unsigned)grid, 64 * NW, &ra, [&]() { k_cbf_rollout<T, 0, false, NW, false>(ra); });     // (as the library dispatches)
    else simt::launch((unsigned)grid, 64 * NW, &ra, [&]() { k_cbf_rollout<T, 0, false, NW, true>(ra); });
    for (int j = 0; j < ks; ++j) t += ra.ctrl_dt;
    slot = (slot + ks) % 3;
  }
  const int last = (steps - 1) % 3;
  std::vector<double> od((size_t)n * 20);
  for (size_t k = 0; k < od.size(); ++k) od[k] = (double)ring[(size_t)last * n * 20 + k];
  fwrite(od.data(), sizeof(double), od.size(), out);
  fwrite(slog.data(), sizeof(int), slog.size(), out);
  fwrite(cost.data(), sizeof(int), E, out);
  return 0;
}

// mode 3: k_cbf_rollout_o3 (one wavefront per env).  rest: t0, K[40] (the LQR-yank-omega gain), P[n*7], state13[n*13], rpm_echo[n*4]
template <typename T> static int run_rollout_o3(const In& in, FILE* out) {
  const int E = in.E, D = in.D, n = E * D, steps = in.n_steps;
  if (in.cbf.order != 3 || D < 1 || D > 16 || in.rest.size() != (size_t)1 + 40 + (size_t)n * 24) return 3;
  const double t0 = in.rest[0];
  const double* Kd = in.rest.data() + 1;
  const double* Pd = Kd + 40;
  const double* st13 = Pd + (size_t)n * 7;
  const double* echo = st13 + (size_t)n * 13;
  const size_t ld = ((size_t)n + 255) / 256 * 256;
  Consts<T> c;
  fill_consts(in.cfg, in.gains, c);
  CbfParams<T> P;
  fill_cbf_params(in.cfg, in.cbf, P);
  LqrYoGain<T> K;
  for (int r = 0; r < 4; ++r)
    for (int k = 0; k < 10; ++k) K.k[r][k] = (T)Kd[10 * r + k];
  std::vector<T> state(13 * ld + 16, T(0)), lem(7 * ld + 16, T(0)), ll(6 * ld, T(0)), rpm(4 * ld, T(0)), obst = to_t<T>(in.obstacles, 64), obs((size_t)n * 20, T(0)),
                 ring((size_t)3 * n * 20, T(0));
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 7; ++k) lem[lidx(k, i, ld)] = (T)Pd[(size_t)7 * i + k];
    const double* s = st13 + (size_t)13 * i;
    for (int k = 0; k < 13; ++k) state[sidx(k, i, ld)] = (T)(k < 3 ? s[k] - Pd[(size_t)7 * i + 2 + k] : s[k]);
    for (int k = 0; k < 4; ++k) obs[(size_t)20 * i + 16 + k] = (T)echo[(size_t)4 * i + k];       // the current observation's RPM echo (all the kernel reads of it)
  }
  std::vector<int> pair;
  for (int i = 0; i < D - 1; ++i)
    for (int j = i + 1; j < D; ++j) pair.push_back(i | (j << 8));
  pair.push_back(0);
  std::vector<int> status(E, -7), slog((size_t)steps * E, -7), cost(E, 0);
  const double tol = in.cbf.tol > 0 ? in.cbf.tol : (sizeof(T) == 8 ? 1e-12 : 1e-6);
  const int m3 = D * (D - 1) / 2 + D * in.cbf.n_obs + 6 * D, n3 = 3 * D, max_iter = in.cbf.max_iter > 0 ? in.cbf.max_iter : 64 * m3;
  const double dt = 1.0 / in.cfg.ctrl_freq;
  double t = t0;
  int slot = 0;
  for (int k0 = 0; k0 < steps; k0 += 5) {                          // launches of 5 steps, a 3-slot ring carried across them
    const int ks = steps - k0 < 5 ? steps - k0 : 5;
#define O3(RR, NM)                                                                                                                          \
  simt::launch((unsigned)E, 64, nullptr, [&]() {                                                                                            \
    k_cbf_rollout_o3<T, RR, NM>(c, P, K, E, ld, t, dt, ks, state.data(), lem.data(), rpm.data(), ll.data(), pair.data(), obst.data(), obs.data(), \
                                ring.data(), slot, 3, status.data(), slog.data() + (size_t)k0 * E, cost.data(), max_iter, (T)(tol * tol),   \
                                (T)(in.cfg.M * in.cfg.G));                                                                                  \
  })
    if (n3 <= 24 && m3 <= 256) O3(4, 24);
    else O3(8, 48);
#undef O3
    for (int j = 0; j < ks; ++j) t += dt;
    slot = (slot + ks) % 3;
  }
  std::vector<double> od(obs.begin(), obs.end());
  for (size_t k = 0; k < od.size(); ++k)
    if (od[k] != (double)ring[(size_t)((steps - 1) % 3) * n * 20 + k]) return 6;       // the ring's last slot is the observation returned
  fwrite(od.data(), sizeof(double), od.size(), out);
  fwrite(slog.data(), sizeof(int), slog.size(), out);
  fwrite(cost.data(), sizeof(int), E, out);
  return 0;
}

// mode 4: the headline kernels (BASELINE configs 1-3 and 5), 256-thread workgroups with a partial last one.  `nominal` picks the launch
// form: 0 k_step_geometric per control step in two half-shard launches (batch0), 1 k_rollout_geometric in launches of 7 steps (log ring
// of 7 slots + obs_last), 2 k_step per control step on an action table, 3 k_rollout_step in launches of 5 steps (3 action sets, 4-slot ring),
// 4 k_rollout_geometric with every step's rows rewritten in place (log stride 0: the default-policy-store path).
// dtype: 0 fp32, 1 fp64, 2 fp16 storage (fp32 arithmetic), 3 fp32 with compensated accumulation (forms 0 / 1).
// rest: t0, P[n*7], state13[n*13] (world), forms 2 / 3: actions[3*n*4].   out: double obs[n*20], state13[n*13] (world), act[n*4] (form 0)
template <typename T, typename S, bool COMP> static int run_headline(const In& in, FILE* out) {
  const int E = in.E, D = in.D, n = E * D, steps = in.n_steps, form = in.nominal, A = 3;
  const bool table = form == 2 || form == 3;
  if (steps < 1 || in.rest.size() != (size_t)1 + (size_t)n * 20 + (table ? (size_t)A * n * 4 : 0)) return 3;
  const double t0 = in.rest[0];
  const double* Pd = in.rest.data() + 1;
  const double* st13 = Pd + (size_t)n * 7;
  const double* actd = st13 + (size_t)n * 13;
  const size_t ld = ((size_t)n + 255) / 256 * 256;
  Consts<T> c;
  fill_consts(in.cfg, in.gains, c);
  // exact-size buffers: a store past drone n - 1 (the partial wave of the last workgroup) lands in a red zone
  std::vector<S> state(13 * ld, S(0)), lo(4 * ld, S(0)), obs((size_t)n * 20, S(0)), ring((size_t)7 * n * 20, S(0)), act_out((size_t)n * 4, S(0)), actions;
  std::vector<T> lem(7 * ld, T(0)), origin(3 * ld, T(0)), rpm(4 * ld, T(0));
  for (int i = 0; i < n; ++i) {
    for (int k = 0; k < 7; ++k) lem[lidx(k, i, ld)] = (T)Pd[(size_t)7 * i + k];
    for (int k = 0; k < 3; ++k) origin[k * ld + i] = (T)Pd[(size_t)7 * i + 2 + k];
    const double* s = st13 + (size_t)13 * i;
    for (int k = 0; k < 13; ++k) state[sidx(k, i, ld)] = (S)(T)(k < 3 ? s[k] - Pd[(size_t)7 * i + 2 + k] : s[k]);
  }
  if (table) {
    actions.resize((size_t)A * n * 4);
    for (size_t k = 0; k < actions.size(); ++k) actions[k] = (S)(T)actd[k];
  }
  S* lop = COMP ? lo.data() : nullptr;
  const unsigned grid = (unsigned)(ld / 256), g0 = (grid + 1) / 2;
  const double dt = 1.0 / in.cfg.ctrl_freq;
  double t = t0;
  int last_slot = 0;
  if (form == 0) {
    for (int k = 0; k < steps; ++k, t += dt) {
      simt::launch(g0, 256, nullptr, [&]() {
        k_step_geometric<T, S, true, true, false, false, COMP>(c, n, ld, t, state.data(), lem.data(), rpm.data(), obs.data(), act_out.data(), 0, lop);
      });
      if (grid > g0)
        simt::launch(grid - g0, 256, nullptr, [&]() {
          k_step_geometric<T, S, true, true, false, false, COMP>(c, n, ld, t, state.data(), lem.data(), rpm.data(), obs.data(), act_out.data(), (int)g0, lop);
        });
    }
  } else if (form == 1) {
    for (int k0 = 0; k0 < steps; k0 += 7) {
      const int ks = steps - k0 < 7 ? steps - k0 : 7;
      simt::launch(grid, 256, nullptr, [&]() {
        k_rollout_geometric<T, S, false, false, 0>(c, nullptr, n, ld, t, dt, ks, state.data(), lem.data(), rpm.data(), ring.data(), (size_t)n * 20, obs.data(),
                                                   nullptr, nullptr, lop);
      });
      for (int j = 0; j < ks; ++j) t += dt;
      last_slot = ks - 1;
    }
    for (size_t k = 0; k < obs.size(); ++k)
      if ((double)obs[k] != (double)ring[(size_t)last_slot * n * 20 + k]) return 6;       // obs_last = the log's last written slot
  } else if (form == 4) {                                      // mds_rollout_geometric's own use of the kernel: every step rewrites the same [n, 20] array
    for (int k0 = 0; k0 < steps; k0 += 7) {
      const int ks = steps - k0 < 7 ? steps - k0 : 7;
      simt::launch(grid, 256, nullptr, [&]() {
        k_rollout_geometric<T, S, false, false, 0>(c, nullptr, n, ld, t, dt, ks, state.data(), lem.data(), rpm.data(), obs.data(), 0, nullptr, nullptr, nullptr, lop);
      });
      for (int j = 0; j < ks; ++j) t += dt;
    }
  } else if (form == 2) {
    if (COMP) return 4;
    for (int k = 0; k < steps; ++k)
      simt::launch(grid, 256, nullptr, [&]() {
        k_step<T, S, true, false, false>(c, n, ld, state.data(), origin.data(), rpm.data(), actions.data() + (size_t)(k % A) * n * 4, obs.data(), 0);
      });
  } else {
    if (COMP) return 4;
    int a0 = 0, s0 = 0;
    ring.resize((size_t)4 * n * 20);
    for (int k0 = 0; k0 < steps; k0 += 5) {
      const int ks = steps - k0 < 5 ? steps - k0 : 5;
      simt::launch(grid, 256, nullptr, [&]() {
        k_rollout_step<T, S, false, false>(c, n, ld, state.data(), origin.data(), rpm.data(), actions.data(), a0, A, ring.data(), s0, 4, ks);
      });
      a0 = (a0 + ks) % A;
      s0 = (s0 + ks) % 4;
    }
    last_slot = (s0 + 3) % 4;
    for (size_t k = 0; k < obs.size(); ++k) obs[k] = ring[(size_t)last_slot * n * 20 + k];
  }
  std::vector<double> od(obs.begin(), obs.end()), sd((size_t)n * 13), ad(act_out.begin(), act_out.end());
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 13; ++k)
      sd[(size_t)13 * i + k] = (double)(T)state[sidx(k, i, ld)] + (COMP && k >= 10 ? (double)(T)lo[ridx(k, i)] : 0.0) + (k < 3 ? Pd[(size_t)7 * i + 2 + k] : 0.0);
  for (int i = 0; i < n; ++i)                                               // the clipped RPM the kernel kept = the observation's echo
    for (int k = 0; k < 4; ++k)
      if ((double)(S)rpm[k * ld + i] != od[(size_t)20 * i + 16 + k]) return 7;
  fwrite(od.data(), sizeof(double), od.size(), out);
  fwrite(sd.data(), sizeof(double), sd.size(), out);
  fwrite(ad.data(), sizeof(double), ad.size(), out);
  return 0;
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  In in;
  if (!read_in(argv[1], in)) return 2;
  FILE* out = fopen(argv[2], "wb");
  if (!out) return 2;
  int rc = 5;
#ifndef SIMT_ONLY_MODE
#define SIMT_ONLY_MODE 0        // 1 / 2 / 4: compile one group of modes only (tests/emul/simt/simt.py builds them side by side)
#endif
#if SIMT_ONLY_MODE == 0 || SIMT_ONLY_MODE == 1
  if (in.mode == 1) rc = in.dtype ? run_filter<double>(in, out) : run_filter<float>(in, out);
#endif
#if SIMT_ONLY_MODE == 0 || SIMT_ONLY_MODE == 2
  if (in.mode == 2) rc = in.dtype ? run_rollout<double>(in, out) : run_rollout<float>(in, out);
  if (in.mode == 3) rc = in.dtype ? run_rollout_o3<double>(in, out) : run_rollout_o3<float>(in, out);
#endif
#if SIMT_ONLY_MODE == 0 || SIMT_ONLY_MODE == 4
  if (in.mode == 4)
    rc = in.dtype == 0 ? run_headline<float, float, false>(in, out)
                       : (in.dtype == 1 ? run_headline<double, double, false>(in, out)
                                        : (in.dtype == 2 ? run_headline<float, half_t, false>(in, out) : run_headline<float, float, true>(in, out)));
#endif
  fclose(out);
  fprintf(stderr, "[simt] %ld wave collectives\n", simt::collectives.load());
  return rc;
}
