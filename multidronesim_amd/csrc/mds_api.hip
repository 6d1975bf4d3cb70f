// C-ABI of libmds (include/mds.h): handle management, constant set-up in double, dtype
// dispatch and kernel launches.  No torch types, no exceptions across the boundary.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <unordered_map>
#include <vector>

#include "../../include/mds.h"
#include "mds_consts.hpp"
#include "mds_kernels.hip"
#ifndef MDS_PART
#define MDS_PART 3
#endif
#if MDS_PART & 2
#include "mds_cbf_kernels.hip"      // (part 2 only: it defines a non-template kernel)
#else
#include "mds_cbf.hpp"
#endif

using namespace mds;

// The library builds as one translation unit (MDS_PART 3, the default: `hipcc -shared mds_api.hip`) or as two compiled side by
// side and linked (__graft_entry__.build(): -DMDS_PART=1 = handle management, the step / rollout / conversion entry points;
// -DMDS_PART=2 = the ECBF filter, the LQR / DSLPID / low-level controllers and the CBF rollouts) -- the device code of ~200 kernel
// instantiations compiles serially per unit, so two units halve the build and a change to the CBF kernels rebuilds one of them.
namespace mds_detail {
extern thread_local char g_err[512];            // last error text of the calling thread (mds_last_error): one copy for both parts
#if MDS_PART & 1
thread_local char g_err[512] = "";
#endif
}  // namespace mds_detail
using mds_detail::g_err;

namespace {


int fail_hip(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  return MDS_EHIP;
}
int fail(int code, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s", what);
  return code;
}

#define MDS_HIP(call)                          \
  do {                                         \
    hipError_t e__ = (call);                   \
    if (e__ != hipSuccess) return fail_hip(e__, #call); \
  } while (0)

size_t elem_size(int dtype) { return dtype == MDS_F64 ? 8 : (dtype == MDS_F16 ? 2 : 4); }   // MDS_F32C: fp32 buffers
size_t comp_size(int dtype) { return dtype == MDS_F64 ? 8 : 4; }
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Every entry point runs with the handle's device current and restores the caller's on return: a process may hold
// handles on several GPUs (one env per device, or a trajectory-evaluation handle next to an env), and neither the
// launches nor the set-up allocations may land on whatever device the calling thread happened to have selected.
struct DevGuard {
  int prev = -1;
  bool switched = false;
  hipError_t err = hipSuccess;
  explicit DevGuard(int dev) {
    if (dev < 0) return;
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != dev) {
      err = hipSetDevice(dev);
      switched = err == hipSuccess;
    }
  }
  ~DevGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DevGuard(const DevGuard&) = delete;
  DevGuard& operator=(const DevGuard&) = delete;
};
#define MDS_DEV(h) DevGuard dev_guard__((h) ? (h)->cfg.device : -1)

__global__ void k_noop() {}


}  // namespace

struct mds_handle {
  mds_config cfg;
  mds_geometric_gains gains;
  double wind[3];
  int n;
  size_t ld;           // plane stride (elements)
  void* state;         // S [13][ld]
  void* state_lo = nullptr;      // MDS_F32C: float [ld][4] = residuals of the three body rates + pad (load_resid in mds_kernels.hip)
  void* origin;        // T [3][ld]
  void* last_rpm;      // T [4][ld]
  void* lem;           // T [7][ld]
  double* scratch;     // double [n*20] device staging for host<->device set-up calls
  double* init_pose = nullptr;   // double [n*6]: xyz, rpy of the last mds_reset (episode resets on the device, mds_reset_async)
  bool has_traj;
  int traj_mode;        // 1: per-drone Lemniscate planes (fused fp32 fast path), 2: general segment tables
  double* segs;         // device [MDS_SEG_DIM, total]: field-major (mds_traj.hpp SegTable)
  int nseg_total = 0;
  int* tinfo;           // device [n, 3] = first segment, nseg | compound << 16, stride between pieces (mds_traj.hpp TrajInfo)
  Consts<float> cf;
  Consts<double> cd;
  // ECBF filter
  bool has_cbf;
  mds_cbf_params cbf;
  CbfParams<float> cbf_f;
  CbfParams<double> cbf_d;
  int* pair_ij;        // device [D(D-1)/2]
  void* obstacles;     // device T [n_obs,4]
  // slot 0: the whole batch; slots 1, 2: the two env halves of mds_rollout_cbf_geometric (each keeps its own cost classes)
  int* cbf_order;      // [3 slots][3,E] env ids by cost class (longest-first dispatch of the QP kernel)
  int* cbf_count;      // [3 slots][4]
  int cbf_calls_half[2] = {-1, -1};
  int* cbf_cost;       // [E] GI iterations of the last launch
  int cbf_calls;       // launches since the classes were rebuilt; -1: no classes yet
  int rollout_streams = 0;              // mds_set_rollout_streams: 0 auto, 1, 2
  // two-chain rollouts: chain 0 runs on the caller's stream, chain 1 on this internal stream; ev[0] forks it off the caller's
  // stream, ev[1] joins it back.  Created (and primed) by mds_create for shards that can split, else by mds_set_rollout_streams(h, 2).
  hipStream_t split_st = nullptr;
  hipEvent_t split_ev[2] = {nullptr, nullptr};
  int split_lds = 32768;                // LDS bytes a half-shard workgroup occupies in a two-chain rollout (MDS_TUNE_SPLIT_LDS, tuning only)
  int split_offset = 0;                 // 1: the fork event sits half way through chain 0's first step (MDS_TUNE_SPLIT_OFFSET, tuning only; the
                                        // fork's own latency already starts chain 1 about half a kernel late: 20-step calls 17.3 vs 17.6 us)
  int split_min_steps = 0;              // auto policy: calls shorter than this stay on one stream (MDS_TUNE_SPLIT_MIN_STEPS, tuning only)
  int last_rollout_streams = 0;         // what the last mds_rollout_* call did (mds_get_last_rollout_streams)
  int rollout_form = 0;                 // mds_set_rollout_form: 0 auto, 1 one launch per control step, 2 the whole-rollout kernel in chunks
  int rollout_chunk = 50;               // control steps per launch of form 2
  int last_rollout_form = 0;            // what the last mds_rollout_geometric did (mds_get_last_rollout_form)
  bool next_nom_ok[3] = {false, false, false};   // per env-range slot: the last low-level launch also left the nominal input of ...
  double next_nom_t[3] = {0.0, 0.0, 0.0};        // ... the control step at this time in the scratch (C rollout loops chain on it)
  bool cbf_hildreth = false;            // MDS_CBF_SOLVER=hildreth, read once by mds_cbf_configure
  bool cbf_q4 = false;                  // MDS_CBF_Q4=1 at configure time: the four-envs-per-wave QP kernel (k_cbf_filter_q4) where it applies; measured
                                        // no faster than one env per wave (see its header), so opt-in
  bool cbf_chain_nominal = true;        // MDS_CBF_CHAIN=0 at configure time: the C rollout loops launch the nominal kernel every step (A/B)
  int cbf_last_step_kernel = -1;        // what the most recent CBF-filtered step launched: 1 the one-launch kernel, 0 QP + low level
  bool cbf_step_persistent = false;     // mds_cbf_set_step_kernel(h, 2): a CBF-filtered step = one launch of the persistent rollout kernel where it applies
  bool cbf_fused = false;               // mds_cbf_set_step_kernel / MDS_CBF_FUSED=1 at configure time: the one-launch CBF step (k_cbf_step) where it applies; it wins only
                                        // on scenes whose QPs need no iterations (see the kernel's header), so the default is the three launches
  void* cbf_unom;      // S [n,4]  scratch of mds_step_cbf_geometric
  void* cbf_xdes;      // S [n,9]
  void* cbf_usafe;     // S [n,4]
  void* ll;            // T [6][ld]: ThrustOmega last_omega3 | integral3
  void* pid;           // T [9][ld]: DSLPID last_rpy3 | integral_pos_e3 | integral_rpy_e3
  DslPidGains<float> pid_f;
  DslPidGains<double> pid_d;
  bool has_lqr;
  int cbf_nominal;     // 0 geometric, 1 lqr-omega, 2 lqr-yank-omega (order 3)
  LqrGain<float> lqr_f;
  LqrGain<double> lqr_d;
  bool has_lqr_yo;
  LqrYoGain<float> lqr_yo_f;
  LqrYoGain<double> lqr_yo_d;
  bool has_lqr12;
  void* gain_dev[3];   // device copies of the gains for the whole-rollout kernels: 0 LQR-12, 1 LQR-omega, 2 LQR-yank-omega (written by the mds_set_*_gain calls)
  Lqr12Gain<float> lqr12_f;
  Lqr12Gain<double> lqr12_d;
  void* state_alt;     // second state buffer of the ground-effect / downwash step (double-buffered substeps)
  void* act_scratch = nullptr;   // S [n,4]: the controller's action, replayed by the later substeps of a ground-effect / downwash step
  bool envfx;          // physics has ground effect and / or downwash
  EnvFx<float> fx_f;
  EnvFx<double> fx_d;
  bool track_rpm;      // last_rpm planes maintained by every step kernel (DYN_DRAG, order-3 CBF, or cfg.track_last_rpm)
  bool rpm_stale;      // a step ran without tracking since the last reset
};

static inline bool has_drag(const mds_handle* h) {
  return h->cfg.physics == MDS_PHYSICS_DYN_DRAG || h->cfg.physics == MDS_PHYSICS_DYN_GND_DRAG_DW;
}

// The last clipped action lives in the obs a step call returns.  The SoA copy costs 16 B per drone-step
// and is kept only where something reads it back (the _drag term, calc_z_thrust of the yank path, mds_get_obs).
static inline void* rpm_track(mds_handle* h) {
  if (h->track_rpm) return h->last_rpm;
  h->rpm_stale = true;
  return nullptr;
}

// Shards at least this large step their two halves on two streams inside mds_rollout_geometric (0 = auto policy).
constexpr size_t kSplitMinDrones = size_t(1) << 18;
// Calls shorter than this stay on the caller's stream under the auto policy: a two-chain call costs a fork and a join (two
// cross-stream dependencies) and its halves start in lock step -- ~35 us more than n x its steady-state step.  C3, wall-clock us
// per control step (enqueue .. synchronize, median of 21 calls) by call length, one stream -> two chains: 8 steps 20.3 -> 20.8,
// 12 steps 19.0 -> 19.0, 16 steps 19.0 -> 18.1, 20 steps 18.7 -> 17.7, 32 steps 18.2 -> 16.9, 48 steps 18.0 -> 16.5; by HIP events
// 100 steps 17.4 -> 15.2, 2000 steps 17.4 -> 14.8 (profiles/r02_short_calls.log).
#ifndef MDS_SPLIT_MIN_STEPS
#define MDS_SPLIT_MIN_STEPS 16
#endif
constexpr int kSplitMinSteps = MDS_SPLIT_MIN_STEPS;

// The stream policy in one place (mds_set_rollout_streams; mds_rollout_streams_for reports it).  loop 0: the fused geometric /
// plain env.step loops (size sweep of DESIGN.md 4: below 2^18 drones the extra launches cost more than the overlap gains,
// between 2^18 and 2^19 it pays only once the chains have had ~1000 steps to drift out of phase); loop 1: the CBF loop
// (2^15 drones 36.6 vs 36.4 us, 2^16 41.0 vs 39.4, 2^17 52.9 vs 44.7, 2^18 71.7 vs 60.3).
static int rollout_streams_policy(const mds_handle* h, int loop, int n_steps) {
  if (n_steps < 2 || !h->split_st || h->envfx) return 1;     // ground effect / downwash: the substeps swap two state buffers, one chain
  if (h->rollout_streams) return h->rollout_streams;
  const size_t n = (size_t)h->n;
  const int min_steps = h->split_min_steps > 0 ? h->split_min_steps : kSplitMinSteps;
  if (loop == 1) return (n >= kSplitMinDrones / 4 && n_steps >= min_steps) ? 2 : 1;
  const bool big = n >= 2 * kSplitMinDrones ? n_steps >= min_steps : (n >= kSplitMinDrones && n_steps >= 1000);
  return big ? 2 : 1;
}

// The launch form of the fused geometric loop (mds_set_rollout_form; mds_rollout_form_for reports it).  1: one launch per control step
// (k_step_geometric; two chains on big shards, above).  2: the whole-rollout kernel in launches of `rollout_chunk` control steps
// (k_rollout_geometric: state in registers, every step's observation still written).  Auto: form 2 for every shard of kFusedMinDrones drones
// and more and calls of kFusedMinSteps steps and more -- it is the faster form at every size measured (profiles/r04_shard_sweep.json,
// r04_form_sweep.json): below 2^18 drones form 1 is launch-bound (a dependent launch costs ~4 us whatever it moves: C2, 16 384 drones, 3.9 us
// per step against 1.9; one eighth of config 3, 65 536 drones, 4.75 against 1.97), above it form 1 streams the 13-value state through HBM
// twice per step (212 B per drone-step at 0.8-0.93 of the roofline: config 3 15.4 us per step) where form 2 moves the observation row only
// (82.6 B: 7.5-9.0 us, VALU-bound).  float64 likewise (config 3: 32.7 -> 30.9 us per step, 4 M drones 330 -> 246; its whole-rollout kernel is
// arithmetic-bound -- software sin / cos / atan2 / asin, divisions -- and wins by less).  MDS_FUSED_MAX_DRONES (compile time) caps the window for A/B
// builds.  fp16 storage stays in form 1 (form 2 rounds the state to fp16 once per launch instead of once per step: not the same arithmetic).
#ifndef MDS_FUSED_MAX_DRONES
#define MDS_FUSED_MAX_DRONES (~size_t(0))
#endif
constexpr size_t kFusedMinDrones = size_t(1) << 13, kFusedMaxDrones = MDS_FUSED_MAX_DRONES;
constexpr int kFusedMinSteps = 8;
static int rollout_form_policy(const mds_handle* h, int n_steps) {
  if (h->envfx || n_steps < 1) return 1;          // ground effect / downwash: env-mates interact every substep, no state-in-registers form
  if (h->rollout_form) return h->rollout_form;
  if (h->cfg.dtype == MDS_F16) return 1;
  const size_t n = (size_t)h->n;
  const size_t nmax = kFusedMaxDrones;
  return (n >= kFusedMinDrones && n <= nmax && n_steps >= kFusedMinSteps) ? 2 : 1;
}

// Set-up path (mds_create / mds_set_rollout_streams): the internal stream and the two events of the two-chain rollouts.
// The new stream runs one empty kernel and is drained, so that its hardware queue exists and has dispatched before any
// rollout uses it -- the rollouts themselves never create or initialise anything (mds.h: hot-path calls only enqueue).
static int split_streams_ready(mds_handle* h) {
  if (h->split_st && h->split_ev[0] && h->split_ev[1]) return MDS_OK;
  if (!h->split_st) {
    // ROCm multiplexes a process's streams onto GPU_MAX_HW_QUEUES (4 by default) hardware queues; when the caller's stream and this one share
    // a queue the two chains serialise (C4 with 8 other active streams in the process: 49 -> 87 us per step).  A stream of another priority
    // level lives on queues of its own: MDS_SPLIT_STREAM_PRIORITY=high|low is for such applications (no measurable cost or gain otherwise).
    const char* pe = getenv("MDS_SPLIT_STREAM_PRIORITY");          // read per stream creation (per handle), as mds.h says
    const int prio_mode = !pe ? 0 : (pe[0] == 'h' || pe[0] == '1') ? 1 : (pe[0] == 'l' || pe[0] == '2') ? 2 : 0;
    if (prio_mode) {
      int lo = 0, hi = 0;
      MDS_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      MDS_HIP(hipStreamCreateWithPriority(&h->split_st, hipStreamNonBlocking, prio_mode == 1 ? hi : lo));
    } else {
      MDS_HIP(hipStreamCreateWithFlags(&h->split_st, hipStreamNonBlocking));
    }
    k_noop<<<1, 64, 0, h->split_st>>>();
    MDS_HIP(hipGetLastError());
  }
  for (int k = 0; k < 2; ++k)
    if (!h->split_ev[k]) MDS_HIP(hipEventCreateWithFlags(&h->split_ev[k], hipEventDisableTiming));
  MDS_HIP(hipEventRecord(h->split_ev[1], h->split_st));
  MDS_HIP(hipStreamSynchronize(h->split_st));
  return MDS_OK;
}

// Entry / exit of a two-chain rollout.  Chain 0 is the caller's stream itself: its launches wait for nothing, and the call
// costs two cross-stream dependencies in all (four, and 50 us per call, when both chains ran on internal streams).
// fork: the internal stream waits for everything enqueued on the caller's stream so far.
// join: the caller's stream waits for chain 1 -- ALWAYS run once fork has been attempted, also when a launch in between
// failed, so that the caller's stream keeps ordering whatever was enqueued (mds.h: the caller's stream orders the whole call).
static int split_fork(mds_handle* h, hipStream_t st) {
  MDS_HIP(hipEventRecord(h->split_ev[0], st));
  MDS_HIP(hipStreamWaitEvent(h->split_st, h->split_ev[0], 0));
  return MDS_OK;
}
static int split_join(mds_handle* h, hipStream_t st, int rc_body) {
  int rc = rc_body;
  hipError_t e = hipEventRecord(h->split_ev[1], h->split_st);
  if (e == hipSuccess) e = hipStreamWaitEvent(st, h->split_ev[1], 0);
  if (e != hipSuccess && rc == MDS_OK) rc = fail_hip(e, "two-chain rollout: join");
  return rc;
}

// dispatch on the handle dtype: F32 -> <float,float>, F64 -> <double,double>, F16 -> <float,half_t>
#define MDS_DISPATCH(h, EXPR)                                               \
  do {                                                                      \
    if ((h)->cfg.dtype == MDS_F32 || (h)->cfg.dtype == MDS_F32C) {           \
      typedef float T; typedef float S; const Consts<T>& C = (h)->cf; (void)C; EXPR; \
    } else if ((h)->cfg.dtype == MDS_F64) {                                 \
      typedef double T; typedef double S; const Consts<T>& C = (h)->cd; (void)C; EXPR; \
    } else {                                                                \
      typedef float T; typedef half_t S; const Consts<T>& C = (h)->cf; (void)C; EXPR; \
    }                                                                       \
  } while (0)

static inline dim3 grid_for(int n, int block) { return dim3((unsigned)((n + block - 1) / block)); }
static inline bool is_comp(const mds_handle* h) { return h->cfg.dtype == MDS_F32C; }
static inline bool is_f32(const mds_handle* h) { return h->cfg.dtype == MDS_F32 || h->cfg.dtype == MDS_F32C; }   // fp32 buffers
template <typename T> static void fill_cbf(const mds_handle* h, const mds_cbf_params& p, CbfParams<T>& o) { fill_cbf_params(h->cfg, p, o); }

template <typename T> static void fill_lin_model(const double* A, const double* B, double u_eq0, LinModel<T>& M) {
  for (int r = 0; r < 12; ++r) {
    for (int k = 0; k < 12; ++k) M.A[r][k] = (T)A[r * 12 + k];
    for (int k = 0; k < 4; ++k) M.B[r][k] = (T)B[r * 4 + k];
  }
  M.ueq0 = (T)u_eq0;
}

template <typename T, typename S>
static void launch_compare_models(const Consts<T>& C, int count, const void* obs, const double* A, const double* B, double u_eq0, double dyn_m,
                                  const double* dyn_J, double dyn_g, void* xdot_lin, void* xdot_geo, void* x_lin, hipStream_t st) {
  LinModel<T> M;
  fill_lin_model<T>(A, B, u_eq0, M);
  // MDS_TUNE_CMP_LDS: unused dynamic LDS per workgroup (tuning only: fewer resident workgroups per CU, so that the launch runs in rounds
  // whose load and store phases overlap instead of one round in lock step); out-of-range values are ignored
  static const size_t pad = [] {
    const char* v = getenv("MDS_TUNE_CMP_LDS");
    const long x = v ? atol(v) : 0;
    return (size_t)((x > 0 && x <= 120 * 1024) ? x : 0);
  }();
  k_compare_models<T, S><<<grid_for(count, kBlock), kBlock, pad, st>>>(C, M, count, (const S*)obs, (T)dyn_m, (T)dyn_J[0], (T)dyn_J[1], (T)dyn_J[2],
                                                                      (T)dyn_g, (S*)xdot_lin, (S*)xdot_geo, (S*)x_lin);
}
template <typename T, typename S>
static void launch_linear_xdot(const Consts<T>& C, int count, const void* x, const void* action, const double* A, const double* B, double u_eq0,
                               void* xdot, hipStream_t st) {
  LinModel<T> M;
  fill_lin_model<T>(A, B, u_eq0, M);
  k_linear_xdot<T, S><<<grid_for(count, 256), 256, 0, st>>>(C, M, count, (const S*)x, (const S*)action, (S*)xdot);
}

// set-up path: (re)write one gain struct to its device copy, synchronously
static int upload_gain(mds_handle* h, int slot, const void* f32, size_t nf, const void* f64, size_t nd) {
  if (!h->gain_dev[slot]) MDS_HIP(hipMalloc(&h->gain_dev[slot], nd));
  if (h->cfg.dtype == MDS_F64) MDS_HIP(hipMemcpy(h->gain_dev[slot], f64, nd, hipMemcpyHostToDevice));
  else MDS_HIP(hipMemcpy(h->gain_dev[slot], f32, nf, hipMemcpyHostToDevice));
  return MDS_OK;
}

extern "C" {

// (C linkage like everything in this block; not part of include/mds.h)
// helpers called across the two parts (defined once, in the part named)
struct EnvRange;
int step_env_plain(mds_handle* h, const void* action, void* obs, hipStream_t st, int first_substep, void* home = nullptr);              // part 1
int step_env_ctrl(mds_handle* h, int ctrl, double t, const void* u_in, double thrust_offset, void* obs, void* act, hipStream_t st);     // part 1
int rollout_fused(mds_handle* h, double t0, int n_steps, void* obs_log, void* obs_last, void* stream, int ctrl, const char* who);       // part 1
int step_nominal_lowlevel(mds_handle* h, double t, void* obs, int32_t* status, void* action, void* stream, bool with_filter,            // part 2
                          const char* who, const EnvRange* rgp = nullptr, bool have_nominal = false, bool want_next = false, double t_next = 0.0);

#if MDS_PART & 1
int mds_version(void) { return MDS_VERSION; }

const char* mds_strerror(int status) {
  switch (status) {
    case MDS_OK: return "ok";
    case MDS_EINVAL: return "invalid argument";
    case MDS_ENOMEM: return "out of device memory";
    case MDS_EHIP: return "HIP runtime error";
    case MDS_EALIGN: return "device pointer not 16-byte aligned";
    case MDS_ESTATE: return "call not valid in this handle state";
    case MDS_EUNSUPPORTED: return "unsupported combination";
    default: return "unknown status";
  }
}

const char* mds_last_error(void) { return g_err; }

int mds_default_config(int drone_model, mds_config* cfg) {
  if (!cfg || (drone_model != MDS_CF2X && drone_model != MDS_CF2P)) return fail(MDS_EINVAL, "mds_default_config");
  memset(cfg, 0, sizeof(*cfg));
  cfg->num_envs = 1;
  cfg->num_drones = 2;           // PIDEnv.py:29
  cfg->dtype = MDS_F32;
  cfg->physics = MDS_PHYSICS_DYN;
  cfg->integrator = MDS_INTEGRATOR_EULER;
  cfg->drone_model = drone_model;
  cfg->pyb_freq = 100;           // PIDEnv.py:24-25
  cfg->ctrl_freq = 100;
  cfg->M = 0.027;
  cfg->L = 0.0397;
  cfg->KF = 3.16e-10;
  cfg->KM = 7.94e-12;
  if (drone_model == MDS_CF2P) {
    cfg->J[0] = 2.3951e-5; cfg->J[1] = 2.3951e-5; cfg->J[2] = 3.2347e-5;
  } else {
    cfg->J[0] = 1.4e-5; cfg->J[1] = 1.4e-5; cfg->J[2] = 2.17e-5;
  }
  cfg->G = 9.8;
  cfg->thrust2weight = 2.25;
  cfg->drag_coeff[0] = 9.1785e-7; cfg->drag_coeff[1] = 9.1785e-7; cfg->drag_coeff[2] = 10.311e-7;
  return MDS_OK;
}

int mds_default_geometric_gains(mds_geometric_gains* g) {
  if (!g) return fail(MDS_EINVAL, "mds_default_geometric_gains");
  for (int k = 0; k < 3; ++k) {
    g->Kp[k] = 2.25; g->Kv[k] = 3.5; g->KR[k] = 125.0; g->Kw[k] = 10.0;   // control/geometric.py:14-17
  }
  g->g = 9.81;                                                             // :20
  g->max_tilt_angle = 40.0 * M_PI / 180.0;                                 // :23
  return MDS_OK;
}

int mds_create(const mds_config* cfg, mds_handle** out) {
  if (!cfg || !out) return fail(MDS_EINVAL, "mds_create: null argument");
  *out = nullptr;
  if (cfg->num_envs <= 0 || cfg->num_drones <= 0) return fail(MDS_EINVAL, "mds_create: num_envs/num_drones must be > 0");
  if ((long long)cfg->num_envs * cfg->num_drones > (1LL << 30)) return fail(MDS_EINVAL, "mds_create: too many drones");
  if (cfg->dtype < MDS_F32 || cfg->dtype > MDS_F32C) return fail(MDS_EINVAL, "mds_create: dtype");
  if (cfg->physics < MDS_PHYSICS_DYN || cfg->physics > MDS_PHYSICS_DYN_GND_DRAG_DW) return fail(MDS_EINVAL, "mds_create: physics");
  if (cfg->physics >= MDS_PHYSICS_DYN_GND && (cfg->integrator != MDS_INTEGRATOR_EULER || cfg->dtype == MDS_F16 || cfg->dtype == MDS_F32C))
    return fail(MDS_EINVAL, "mds_create: ground effect / downwash run with the explicit Euler integrator on f32 / f64 storage");
  if (cfg->integrator != MDS_INTEGRATOR_EULER && cfg->integrator != MDS_INTEGRATOR_RK4) return fail(MDS_EINVAL, "mds_create: integrator");
  if (cfg->drone_model != MDS_CF2X && cfg->drone_model != MDS_CF2P) return fail(MDS_EINVAL, "mds_create: drone_model");
  if (cfg->ctrl_freq <= 0 || cfg->pyb_freq <= 0 || cfg->pyb_freq % cfg->ctrl_freq != 0)
    return fail(MDS_EINVAL, "mds_create: pyb_freq must be a positive multiple of ctrl_freq");
  if (!(cfg->M > 0) || !(cfg->KF > 0) || !(cfg->KM > 0) || !(cfg->L > 0) || !(cfg->J[0] > 0) || !(cfg->J[1] > 0) || !(cfg->J[2] > 0))
    return fail(MDS_EINVAL, "mds_create: non-positive drone constant");
  DevGuard dev_guard__(cfg->device);        // the caller's current device is restored on return
  if (dev_guard__.err != hipSuccess) return fail_hip(dev_guard__.err, "mds_create: hipSetDevice(cfg->device)");
  mds_handle* h = new (std::nothrow) mds_handle();
  if (!h) return fail(MDS_ENOMEM, "mds_create: host allocation");
  h->cfg = *cfg;
  mds_default_geometric_gains(&h->gains);
  h->n = cfg->num_envs * cfg->num_drones;
  h->ld = ((size_t)h->n + 255) / 256 * 256;
  h->has_traj = false;
  h->traj_mode = 0;
  h->segs = nullptr;
  h->tinfo = nullptr;
  h->wind[0] = h->wind[1] = h->wind[2] = 0.0;
  fill_consts(h->cfg, h->gains, h->cf, h->wind);
  fill_consts(h->cfg, h->gains, h->cd, h->wind);
  const size_t es = elem_size(cfg->dtype), cs = comp_size(cfg->dtype);
  h->state = h->origin = h->last_rpm = h->lem = nullptr;
  h->scratch = nullptr;
  h->has_cbf = false;
  h->pair_ij = nullptr;
  h->obstacles = nullptr;
  h->cbf_unom = h->cbf_xdes = h->cbf_usafe = h->ll = nullptr;
  h->cbf_order = h->cbf_count = h->cbf_cost = nullptr;
  h->cbf_calls = -1;
  h->has_lqr = false;
  h->has_lqr_yo = false;
  h->has_lqr12 = false;
  h->gain_dev[0] = h->gain_dev[1] = h->gain_dev[2] = nullptr;
  h->cbf_nominal = 0;
  h->pid = nullptr;
  h->envfx = cfg->physics >= MDS_PHYSICS_DYN_GND;
  h->state_alt = nullptr;
  fill_envfx(h->cfg, h->fx_f);
  fill_envfx(h->cfg, h->fx_d);
  h->track_rpm = cfg->track_last_rpm != 0 || cfg->physics == MDS_PHYSICS_DYN_DRAG || cfg->physics == MDS_PHYSICS_DYN_GND_DRAG_DW;
  h->rpm_stale = false;
  {
    mds_dslpid_gains dg;
    mds_default_dslpid_gains(&dg);
    mds_set_dslpid_gains(h, &dg);
  }
  hipError_t e = hipMalloc(&h->state, 13 * h->ld * es);
  if (e == hipSuccess && is_comp(h)) e = hipMalloc(&h->state_lo, 4 * h->ld * es);
  if (e == hipSuccess && is_comp(h)) e = hipMemset(h->state_lo, 0, 4 * h->ld * es);
  if (e == hipSuccess && h->envfx) e = hipMalloc(&h->state_alt, 13 * h->ld * es);
  if (e == hipSuccess && h->envfx) e = hipMemset(h->state_alt, 0, 13 * h->ld * es);
  if (e == hipSuccess && h->envfx) e = hipMalloc(&h->act_scratch, (size_t)h->n * 4 * es);
  if (e == hipSuccess) e = hipMalloc(&h->origin, 3 * h->ld * cs);
  if (e == hipSuccess) e = hipMalloc(&h->last_rpm, 4 * h->ld * cs);
  if (e == hipSuccess) e = hipMalloc(&h->lem, 7 * h->ld * cs);
  if (e == hipSuccess) e = hipMalloc((void**)&h->scratch, (size_t)h->n * 20 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc(&h->ll, 6 * h->ld * cs);
  if (e == hipSuccess) e = hipMemset(h->ll, 0, 6 * h->ld * cs);
  if (e == hipSuccess) e = hipMalloc(&h->pid, 9 * h->ld * cs);
  if (e == hipSuccess) e = hipMemset(h->pid, 0, 9 * h->ld * cs);
  if (e == hipSuccess) e = hipMemset(h->state, 0, 13 * h->ld * es);
  if (e == hipSuccess) e = hipMemset(h->origin, 0, 3 * h->ld * cs);
  if (e == hipSuccess) e = hipMemset(h->last_rpm, 0, 4 * h->ld * cs);
  if (e == hipSuccess) e = hipMemset(h->lem, 0, 7 * h->ld * cs);
  if (e != hipSuccess) {
    mds_destroy(h);
    return e == hipErrorOutOfMemory ? fail(MDS_ENOMEM, "mds_create: hipMalloc") : fail_hip(e, "mds_create");
  }
  // identity attitude
  {
    const size_t nbytes = (size_t)h->n * 6 * sizeof(double);
    MDS_HIP(hipMemset(h->scratch, 0, nbytes));
    MDS_DISPATCH(h, (k_reset<T, S><<<grid_for(h->n, 256), 256, 0, 0>>>(h->n, h->ld, h->scratch, h->scratch + (size_t)3 * h->n,
                                                                         (const T*)h->origin, (S*)h->state, (T*)h->last_rpm, 0, (S*)h->state_lo)));
    MDS_HIP(hipGetLastError());
    MDS_HIP(hipDeviceSynchronize());
  }
  // shards large enough for the auto policy of the two-chain rollouts get their streams and events now (smallest auto
  // threshold: the CBF loop, 2^16 drones); smaller ones only if mds_set_rollout_streams(h, 2) asks for them
  // tuning variables: garbage or out-of-range values fall back to the defaults instead of turning into an opaque launch failure
  // (a negative LDS pad cast to size_t, or one above what a workgroup may own, fails every two-chain launch)
  if (const char* v = getenv("MDS_TUNE_SPLIT_OFFSET")) h->split_offset = atoi(v) == 1 ? 1 : 0;
  if (const char* v = getenv("MDS_TUNE_SPLIT_LDS")) {
    const long lds = strtol(v, nullptr, 10);
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device) != hipSuccess) lds_max = 65536;
    h->split_lds = (lds >= 0 && lds <= lds_max - (long)(kBlock * kObsDim * 8)) ? (int)lds : 32768;
  }
  if (const char* v = getenv("MDS_TUNE_SPLIT_MIN_STEPS")) {
    const long ms = strtol(v, nullptr, 10);
    h->split_min_steps = (ms >= 0 && ms <= 1000000) ? (int)ms : 0;
  }
  if ((size_t)h->n >= kSplitMinDrones / 4) {
    if (int rc = split_streams_ready(h)) {
      mds_destroy(h);
      return rc;
    }
  }
  *out = h;
  return MDS_OK;
}

int mds_destroy(mds_handle* h) {
  if (!h) return MDS_OK;
  MDS_DEV(h);
  if (h->state) (void)hipFree(h->state);
  if (h->state_lo) (void)hipFree(h->state_lo);
  if (h->state_alt) (void)hipFree(h->state_alt);
  if (h->act_scratch) (void)hipFree(h->act_scratch);
  if (h->origin) (void)hipFree(h->origin);
  if (h->last_rpm) (void)hipFree(h->last_rpm);
  if (h->lem) (void)hipFree(h->lem);
  if (h->scratch) (void)hipFree(h->scratch);
  if (h->init_pose) (void)hipFree(h->init_pose);
  if (h->pair_ij) (void)hipFree(h->pair_ij);
  if (h->obstacles) (void)hipFree(h->obstacles);
  if (h->cbf_order) (void)hipFree(h->cbf_order);
  if (h->cbf_count) (void)hipFree(h->cbf_count);
  if (h->cbf_cost) (void)hipFree(h->cbf_cost);
  for (int k = 0; k < 3; ++k)
    if (h->gain_dev[k]) (void)hipFree(h->gain_dev[k]);
  if (h->cbf_unom) (void)hipFree(h->cbf_unom);
  if (h->cbf_xdes) (void)hipFree(h->cbf_xdes);
  if (h->cbf_usafe) (void)hipFree(h->cbf_usafe);
  if (h->ll) (void)hipFree(h->ll);
  if (h->pid) (void)hipFree(h->pid);
  if (h->segs) (void)hipFree(h->segs);
  if (h->tinfo) (void)hipFree(h->tinfo);
  if (h->split_st) (void)hipStreamDestroy(h->split_st);
  for (int k = 0; k < 2; ++k)
    if (h->split_ev[k]) (void)hipEventDestroy(h->split_ev[k]);
  delete h;
  return MDS_OK;
}

int mds_get_derived(const mds_handle* h, double out[8]) {
  if (!h || !out) return fail(MDS_EINVAL, "mds_get_derived");
  const mds_config& c = h->cfg;
  const double grav = c.G * c.M;
  const double max_rpm = sqrt(c.thrust2weight * grav / (4 * c.KF));
  out[0] = grav;
  out[1] = sqrt(grav / (4 * c.KF));
  out[2] = max_rpm;
  out[3] = 4 * c.KF * max_rpm * max_rpm;
  out[4] = c.drone_model == MDS_CF2X ? (2 * c.L * c.KF * max_rpm * max_rpm) / sqrt(2.0) : c.L * c.KF * max_rpm * max_rpm;
  out[5] = 2 * c.KM * max_rpm * max_rpm;
  out[6] = 1.0 / c.ctrl_freq;
  out[7] = 1.0 / c.pyb_freq;
  return MDS_OK;
}

// drones [i0, i1) back to the poses of the last mds_reset (zero velocities, zero RPM echo); enqueue only
static int launch_reset_range(mds_handle* h, hipStream_t st, size_t i0, size_t i1) {
  const double* xyz = h->init_pose;
  const double* rpy = h->init_pose + (size_t)3 * h->n;
  MDS_DISPATCH(h, (k_reset<T, S><<<grid_for(i1 - i0, 256), 256, 0, st>>>((int)i1, h->ld, xyz, rpy, (const T*)h->origin, (S*)h->state,
                                                                          (T*)h->last_rpm, (int)i0, (S*)h->state_lo)));
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_reset(mds_handle* h, const double* xyz, const double* rpy, void* stream) {
  MDS_DEV(h);
  if (!h || !xyz || !rpy) return fail(MDS_EINVAL, "mds_reset: null argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t nb = (size_t)h->n * 3 * sizeof(double);
  if (!h->init_pose) MDS_HIP(hipMalloc((void**)&h->init_pose, 2 * nb));
  MDS_HIP(hipMemcpyAsync(h->init_pose, xyz, nb, hipMemcpyHostToDevice, st));
  MDS_HIP(hipMemcpyAsync(h->init_pose + (size_t)3 * h->n, rpy, nb, hipMemcpyHostToDevice, st));
  if (int rc = launch_reset_range(h, st, 0, h->n)) return rc;
  MDS_HIP(hipMemsetAsync(h->ll, 0, 6 * h->ld * comp_size(h->cfg.dtype), st));
  MDS_HIP(hipMemsetAsync(h->pid, 0, 9 * h->ld * comp_size(h->cfg.dtype), st));
  h->rpm_stale = false;
  MDS_HIP(hipStreamSynchronize(st));   // host buffers may be reused by the caller
  return MDS_OK;
}

int mds_reset_async(mds_handle* h, void* stream) {
  MDS_DEV(h);
  if (!h) return fail(MDS_EINVAL, "mds_reset_async: null handle");
  if (!h->init_pose) return fail(MDS_ESTATE, "mds_reset_async: no mds_reset yet");
  hipStream_t st = (hipStream_t)stream;
  if (int rc = launch_reset_range(h, st, 0, h->n)) return rc;
  MDS_HIP(hipMemsetAsync(h->ll, 0, 6 * h->ld * comp_size(h->cfg.dtype), st));
  MDS_HIP(hipMemsetAsync(h->pid, 0, 9 * h->ld * comp_size(h->cfg.dtype), st));
  h->rpm_stale = false;
  return MDS_OK;
}

int mds_state_ptrs(mds_handle* h, void* comp_dev[13], size_t stride_elems[13], void* origin_dev[3]) {
  MDS_DEV(h);
  if (!h || !comp_dev || !stride_elems) return fail(MDS_EINVAL, "mds_state_ptrs: null argument");
  const size_t es = elem_size(h->cfg.dtype), cs = comp_size(h->cfg.dtype);
  for (int k = 0; k < 13; ++k) {
    comp_dev[k] = (char*)h->state + sidx(k, 0, h->ld) * es;
    stride_elems[k] = k < 12 ? 4 : 1;
  }
  if (origin_dev)
    for (int k = 0; k < 3; ++k) origin_dev[k] = (char*)h->origin + (size_t)k * h->ld * cs;
  return MDS_OK;
}

int mds_get_state(mds_handle* h, double* out, void* stream) {
  MDS_DEV(h);
  if (!h || !out) return fail(MDS_EINVAL, "mds_get_state: null argument");
  hipStream_t st = (hipStream_t)stream;
  MDS_DISPATCH(h, (k_get_state<T, S><<<grid_for(h->n, 256), 256, 0, st>>>(h->n, h->ld, (const S*)h->state, (const T*)h->origin,
                                                                            h->scratch, (const S*)h->state_lo)));
  MDS_HIP(hipGetLastError());
  MDS_HIP(hipMemcpyAsync(out, h->scratch, (size_t)h->n * 13 * sizeof(double), hipMemcpyDeviceToHost, st));
  MDS_HIP(hipStreamSynchronize(st));
  return MDS_OK;
}

int mds_set_state(mds_handle* h, const double* in, void* stream) {
  MDS_DEV(h);
  if (!h || !in) return fail(MDS_EINVAL, "mds_set_state: null argument");
  hipStream_t st = (hipStream_t)stream;
  MDS_HIP(hipMemcpyAsync(h->scratch, in, (size_t)h->n * 13 * sizeof(double), hipMemcpyHostToDevice, st));
  MDS_DISPATCH(h, (k_set_state<T, S><<<grid_for(h->n, 256), 256, 0, st>>>(h->n, h->ld, h->scratch, (const T*)h->origin,
                                                                            (S*)h->state, (S*)h->state_lo)));
  MDS_HIP(hipGetLastError());
  MDS_HIP(hipStreamSynchronize(st));
  return MDS_OK;
}

int mds_set_origin(mds_handle* h, const double* origin, void* stream) {
  MDS_DEV(h);
  if (!h || !origin) return fail(MDS_EINVAL, "mds_set_origin: null argument");
  hipStream_t st = (hipStream_t)stream;
  MDS_HIP(hipMemcpyAsync(h->scratch, origin, (size_t)h->n * 3 * sizeof(double), hipMemcpyHostToDevice, st));
  MDS_DISPATCH(h, (k_set_origin<T, S><<<grid_for(h->n, 256), 256, 0, st>>>(h->n, h->ld, h->scratch, (T*)h->origin, (S*)h->state, (S*)h->state_lo)));
  MDS_HIP(hipGetLastError());
  MDS_HIP(hipStreamSynchronize(st));
  return MDS_OK;
}

int mds_get_obs(mds_handle* h, void* obs, void* stream) {
  MDS_DEV(h);
  if (!h || !obs) return fail(MDS_EINVAL, "mds_get_obs: null argument");
  if (!aligned16(obs)) return fail(MDS_EALIGN, "mds_get_obs: obs_dev");
  hipStream_t st = (hipStream_t)stream;
  MDS_DISPATCH(h, (k_get_obs<T, S><<<grid_for(h->n, kBlock), kBlock, 0, st>>>(h->n, h->ld, (const S*)h->state, (const T*)h->origin,
                                                                                (const T*)(h->rpm_stale ? nullptr : h->last_rpm), (S*)obs)));
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

// k_step over 256-drone batches [batch0, batch0 + nb) (nb == 0: the whole shard); half-shard launches of a two-stream rollout
// are padded to 32 KB LDS like the fused step's (launch_step_geometric)
static void launch_step_plain(mds_handle* h, const void* action, void* obs, hipStream_t st, unsigned batch0 = 0, unsigned nb = 0) {
  const dim3 grid(nb ? nb : (unsigned)((h->n + kBlock - 1) / kBlock));
  size_t pad = 0;
  if (nb) {
    const size_t static_lds = obs ? (size_t)kBlock * kObsDim * elem_size(h->cfg.dtype) : 16;
    pad = static_lds < (size_t)h->split_lds ? (size_t)h->split_lds - static_lds : 0;
  }
#define MDS_LAUNCH_STEP(HAS_OBS, RK4, DRAG)                                                                          \
  do {                                                                                                               \
    if (is_comp(h))                                                                                                  \
      k_step<float, float, HAS_OBS, RK4, DRAG, true><<<grid, kBlock, pad, st>>>(h->cf, h->n, h->ld, (float*)h->state, (const float*)h->origin, \
                                                                                (float*)rpm_track(h), (const float*)action, (float*)obs, (int)batch0, (float*)h->state_lo); \
    else                                                                                                             \
      MDS_DISPATCH(h, (k_step<T, S, HAS_OBS, RK4, DRAG><<<grid, kBlock, pad, st>>>(C, h->n, h->ld, (S*)h->state, (const T*)h->origin, \
                                                                                   (T*)rpm_track(h), (const S*)action, (S*)obs, (int)batch0))); \
  } while (0)
  const bool rk4 = h->cfg.integrator == MDS_INTEGRATOR_RK4, drag = has_drag(h);
#define MDS_STEP_OBS(HAS_OBS)                           \
  do {                                                  \
    if (rk4 && drag) MDS_LAUNCH_STEP(HAS_OBS, true, true);   \
    else if (rk4) MDS_LAUNCH_STEP(HAS_OBS, true, false);     \
    else if (drag) MDS_LAUNCH_STEP(HAS_OBS, false, true);    \
    else MDS_LAUNCH_STEP(HAS_OBS, false, false);             \
  } while (0)
  if (obs) MDS_STEP_OBS(true);
  else MDS_STEP_OBS(false);
#undef MDS_STEP_OBS
#undef MDS_LAUNCH_STEP
}

// [UPSTREAM] BaseAviary.step under ground effect / downwash: one launch per physics substep on the double-buffered state
// (k_step_env).  first_substep: substeps [first_substep, K) are run (a controller kernel has already done substep 0 and left its
// action in `action`).
// Stream capture and the double-buffered state.  Every substep reads one buffer, writes the other and flips the handle's pointers on
// the host; a captured call bakes the pointers of the moment into its graph, so a call with an odd number of substeps would replay
// A -> B every time and never advance.  Under capture such a call ends with a copy of the state back into the buffer it started
// from (`home`) and leaves the handle pointing there: every replay then starts and ends in the same buffer.  Eager calls flip only.
static bool stream_is_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
}
static int envfx_return_home(mds_handle* h, void* home, hipStream_t st) {
  if (h->state == home || !stream_is_capturing(st)) return MDS_OK;
  MDS_HIP(hipMemcpyAsync(home, h->state, 13 * h->ld * elem_size(h->cfg.dtype), hipMemcpyDeviceToDevice, st));
  h->state_alt = h->state;
  h->state = home;
  return MDS_OK;
}

int step_env_plain(mds_handle* h, const void* action, void* obs, hipStream_t st, int first_substep, void* home) {
  const dim3 grid = grid_for(h->n, kBlock);
  const int K = h->cfg.pyb_freq / h->cfg.ctrl_freq, D = h->cfg.num_drones;
  void* rpm = rpm_track(h);
  if (!home) home = h->state;
  for (int k = first_substep; k < K; ++k) {
    void* ob = k == K - 1 ? obs : nullptr;
    if (h->cfg.dtype == MDS_F64) {
      if (has_drag(h)) k_step_env<double, double, true><<<grid, kBlock, 0, st>>>(h->cd, h->fx_d, h->n, h->ld, D, (const double*)h->state, (double*)h->state_alt, (const double*)h->origin, (double*)rpm, (const double*)action, (double*)ob, k > 0, k == K - 1);
      else k_step_env<double, double, false><<<grid, kBlock, 0, st>>>(h->cd, h->fx_d, h->n, h->ld, D, (const double*)h->state, (double*)h->state_alt, (const double*)h->origin, (double*)rpm, (const double*)action, (double*)ob, k > 0, k == K - 1);
    } else {
      if (has_drag(h)) k_step_env<float, float, true><<<grid, kBlock, 0, st>>>(h->cf, h->fx_f, h->n, h->ld, D, (const float*)h->state, (float*)h->state_alt, (const float*)h->origin, (float*)rpm, (const float*)action, (float*)ob, k > 0, k == K - 1);
      else k_step_env<float, float, false><<<grid, kBlock, 0, st>>>(h->cf, h->fx_f, h->n, h->ld, D, (const float*)h->state, (float*)h->state_alt, (const float*)h->origin, (float*)rpm, (const float*)action, (float*)ob, k > 0, k == K - 1);
    }
    void* t = h->state; h->state = h->state_alt; h->state_alt = t;
  }
  MDS_HIP(hipGetLastError());
  return envfx_return_home(h, home, st);
}

// One control step of a controller path under ground effect / downwash: k_step_ctrl_env (trajectory sample or given input ->
// controller -> first substep), then the remaining substeps through k_step_env with the action it left in act_scratch.
// ctrl 0 GeometricControl, 1 LQRController (12-state), 2 ThrustOmega low level on u_in, 3 YankOmega low level on u_in.
int step_env_ctrl(mds_handle* h, int ctrl, double t, const void* u_in, double thrust_offset, void* obs, void* act, hipStream_t st) {
  const dim3 grid = grid_for(h->n, kBlock);
  const int K = h->cfg.pyb_freq / h->cfg.ctrl_freq, D = h->cfg.num_drones;
  void* rpm = rpm_track(h);
  void* abuf = K > 1 ? h->act_scratch : nullptr;
  const void* gain = ctrl == 1 ? h->gain_dev[0] : nullptr;
  const bool drag = has_drag(h);
#define MDS_CE(T, CC, FX, DRAG, CTRL)                                                                                              \
  k_step_ctrl_env<T, T, DRAG, CTRL><<<grid, kBlock, 0, st>>>(CC, FX, gain, h->n, h->ld, D, t, h->traj_mode, (const T*)h->state, (T*)h->state_alt, \
                                                             (const T*)h->origin, (const T*)h->lem, SegTable{h->segs, h->nseg_total}, h->tinfo, \
                                                             (T*)rpm, (T*)h->ll, (const T*)u_in, (T)(1.0 / h->cfg.ctrl_freq), (T)thrust_offset,  \
                                                             (T*)abuf, (T*)obs, (T*)act, K == 1)
#define MDS_CE_D(T, CC, FX, CTRL)            \
  do {                                       \
    if (drag) MDS_CE(T, CC, FX, true, CTRL); \
    else MDS_CE(T, CC, FX, false, CTRL);     \
  } while (0)
#define MDS_CE_C(T, CC, FX)                   \
  do {                                        \
    if (ctrl == 0) MDS_CE_D(T, CC, FX, 0);    \
    else if (ctrl == 1) MDS_CE_D(T, CC, FX, 1); \
    else if (ctrl == 2) MDS_CE_D(T, CC, FX, 2); \
    else MDS_CE_D(T, CC, FX, 3);              \
  } while (0)
  if (h->cfg.dtype == MDS_F64) MDS_CE_C(double, h->cd, h->fx_d);
  else MDS_CE_C(float, h->cf, h->fx_f);
#undef MDS_CE_C
#undef MDS_CE_D
#undef MDS_CE
  MDS_HIP(hipGetLastError());
  void* home = h->state;
  void* tmp = h->state; h->state = h->state_alt; h->state_alt = tmp;
  if (K > 1) return step_env_plain(h, h->act_scratch, obs, st, 1, home);
  return envfx_return_home(h, home, st);
}

int mds_step(mds_handle* h, const void* action, void* obs, void* stream) {
  MDS_DEV(h);
  if (!h || !action) return fail(MDS_EINVAL, "mds_step: null argument");
  if (!aligned16(action) || !aligned16(obs)) return fail(MDS_EALIGN, "mds_step: action_dev/obs_dev");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid = grid_for(h->n, kBlock);
  if (h->envfx) return step_env_plain(h, action, obs, st, 0);
  launch_step_plain(h, action, obs, st);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_set_lemniscate(mds_handle* h, const double* params, void* stream) {
  MDS_DEV(h);
  if (!h || !params) return fail(MDS_EINVAL, "mds_set_lemniscate: null argument");
  hipStream_t st = (hipStream_t)stream;
  const int n = h->n;
  // re-base the local frame onto the trajectory centres (host gathers them from params)
  double* centres = new (std::nothrow) double[(size_t)n * 3];
  if (!centres) return fail(MDS_ENOMEM, "mds_set_lemniscate: host allocation");
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) centres[(size_t)3 * i + k] = params[(size_t)7 * i + 2 + k];
  int rc = mds_set_origin(h, centres, stream);
  delete[] centres;
  if (rc != MDS_OK) return rc;
  MDS_HIP(hipMemcpyAsync(h->scratch, params, (size_t)n * 7 * sizeof(double), hipMemcpyHostToDevice, st));
  MDS_DISPATCH(h, (k_set_planes<T><<<grid_for(n, 256), 256, 0, st>>>(n, h->ld, 7, h->scratch, (T*)h->lem)));
  MDS_HIP(hipGetLastError());
  MDS_HIP(hipStreamSynchronize(st));
  h->has_traj = true;
  h->traj_mode = 1;
  return MDS_OK;
}

int mds_set_trajectory_segments(mds_handle* h, const double* segs, const int32_t* offsets, const int32_t* compound,
                                const double* anchor, int32_t total, void* stream) {
  MDS_DEV(h);
  if (!h || !segs || !offsets || !compound || !anchor) return fail(MDS_EINVAL, "mds_set_trajectory_segments: null argument");
  const int n = h->n;
  if (total <= 0 || offsets[0] != 0 || offsets[n] != total) return fail(MDS_EINVAL, "mds_set_trajectory_segments: offsets");
  for (int i = 0; i < n; ++i) {
    const int ns = offsets[i + 1] - offsets[i];
    if (ns < 1 || ns > 65535) return fail(MDS_EINVAL, "mds_set_trajectory_segments: every drone needs 1..65535 segments");
  }
  for (int k = 0; k < total; ++k) {
    const int kind = (int)segs[(size_t)k * MDS_SEG_DIM];
    if (kind < 0 || kind > 3) return fail(MDS_EINVAL, "mds_set_trajectory_segments: segment kind");
  }
  // Drones that follow identical tables (the same trajectory objects broadcast over every env) share one device copy:
  // the table then stays in L2 instead of costing up to 300 B of HBM reads per drone-step.  Unique tables with the same
  // number of pieces form a block stored piece-major (TrajInfo), and the image is field-major (SegTable).
  std::vector<int> ti((size_t)3 * n), uniq_of((size_t)n), usrc, uns;     // usrc/uns: first source row / piece count of a unique table
  std::vector<double> fm;
  int nu = 0;
  try {
    std::unordered_multimap<uint64_t, int> seen;        // hash of a drone's rows -> unique table
    for (int i = 0; i < n; ++i) {
      const int ns = offsets[i + 1] - offsets[i];
      const unsigned char* bytes = reinterpret_cast<const unsigned char*>(segs + (size_t)offsets[i] * MDS_SEG_DIM);
      const size_t nbytes = sizeof(double) * MDS_SEG_DIM * (size_t)ns;
      uint64_t hsh = 1469598103934665603ull ^ (uint64_t)ns;
      for (size_t w = 0; w < nbytes; w += 8) {
        uint64_t word;
        memcpy(&word, bytes + w, 8);
        hsh = (hsh ^ word) * 1099511628211ull;
        hsh ^= hsh >> 29;
      }
      int u = -1;
      auto range = seen.equal_range(hsh);
      for (auto it = range.first; it != range.second; ++it) {
        const int j = it->second;
        if (uns[j] == ns && memcmp(bytes, segs + (size_t)usrc[j] * MDS_SEG_DIM, nbytes) == 0) {
          u = j;
          break;
        }
      }
      if (u < 0) {
        u = (int)usrc.size();
        usrc.push_back(offsets[i]);
        uns.push_back(ns);
        seen.emplace(hsh, u);
      }
      uniq_of[i] = u;
    }
    // blocks by piece count, in order of first appearance
    const int nuniq = (int)usrc.size();
    std::unordered_map<int, int> block_of;              // piece count -> block
    std::vector<int> bcount, bns, rank((size_t)nuniq), blk((size_t)nuniq);
    for (int u = 0; u < nuniq; ++u) {
      auto it = block_of.find(uns[u]);
      if (it == block_of.end()) {
        it = block_of.emplace(uns[u], (int)bcount.size()).first;
        bcount.push_back(0);
        bns.push_back(uns[u]);
      }
      blk[u] = it->second;
      rank[u] = bcount[it->second]++;
    }
    std::vector<long long> bbase(bcount.size());
    long long acc = 0;
    for (size_t b = 0; b < bcount.size(); ++b) {
      bbase[b] = acc;
      acc += (long long)bcount[b] * bns[b];
    }
    if (acc > 0x7fffffffll) return fail(MDS_EINVAL, "mds_set_trajectory_segments: too many segments");
    nu = (int)acc;
    fm.resize((size_t)MDS_SEG_DIM * nu);
    for (int u = 0; u < nuniq; ++u) {
      const int stride = bcount[blk[u]];
      for (int k = 0; k < uns[u]; ++k) {
        const double* row = segs + (size_t)(usrc[u] + k) * MDS_SEG_DIM;
        const size_t id = (size_t)bbase[blk[u]] + rank[u] + (size_t)k * stride;
        for (int f = 0; f < MDS_SEG_DIM; ++f) fm[(size_t)f * nu + id] = row[f];
        // the affine map is skipped on the device when it is the identity (no RotateTrajectory above this piece)
        bool ident = true;
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c) ident = ident && row[27 + 3 * r + c] == (r == c ? 1.0 : 0.0);
          ident = ident && row[36 + r] == 0.0;
        }
        if (!ident) fm[id] = row[0] + 8.0;               // field 0 = kind | kSegAffine
      }
    }
    for (int i = 0; i < n; ++i) {
      const int u = uniq_of[i];
      ti[3 * i] = (int)bbase[blk[u]] + rank[u];
      ti[3 * i + 1] = uns[u] | ((compound[i] ? 1 : 0) << 16);
      ti[3 * i + 2] = bcount[blk[u]];
    }
  } catch (const std::bad_alloc&) {
    return fail(MDS_ENOMEM, "mds_set_trajectory_segments: host allocation");
  }
  int rc = mds_set_origin(h, anchor, stream);
  if (rc != MDS_OK) return rc;
  hipError_t e = hipSuccess;
  if (h->segs) (void)hipFree(h->segs);
  h->segs = nullptr;
  h->nseg_total = 0;
  if (!h->tinfo) e = hipMalloc((void**)&h->tinfo, sizeof(int) * 3 * n);
  if (e == hipSuccess) e = hipMalloc((void**)&h->segs, sizeof(double) * fm.size());
  if (e == hipSuccess) e = hipMemcpy(h->segs, fm.data(), sizeof(double) * fm.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(h->tinfo, ti.data(), sizeof(int) * 3 * n, hipMemcpyHostToDevice);
  if (e != hipSuccess) return fail_hip(e, "mds_set_trajectory_segments");
  h->nseg_total = nu;
  h->has_traj = true;
  h->traj_mode = 2;
  return MDS_OK;
}

int mds_traj_eval(mds_handle* h, double t, void* des, void* stream) {
  MDS_DEV(h);
  if (!h || !des) return fail(MDS_EINVAL, "mds_traj_eval: null argument");
  if (h->traj_mode == 1) return mds_lemniscate_eval(h, t, des, stream);
  if (h->traj_mode != 2) return fail(MDS_ESTATE, "mds_traj_eval: no trajectories set");
  hipStream_t st = (hipStream_t)stream;
  if (h->cfg.dtype == MDS_F64) k_traj_eval<double><<<grid_for(h->n, 256), 256, 0, st>>>(h->n, t, SegTable{h->segs, h->nseg_total}, h->tinfo, (double*)des);
  else if (is_f32(h)) k_traj_eval<float><<<grid_for(h->n, 256), 256, 0, st>>>(h->n, t, SegTable{h->segs, h->nseg_total}, h->tinfo, (float*)des);
  else k_traj_eval<half_t><<<grid_for(h->n, 256), 256, 0, st>>>(h->n, t, SegTable{h->segs, h->nseg_total}, h->tinfo, (half_t*)des);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_set_geometric_gains(mds_handle* h, const mds_geometric_gains* g) {
  if (!h || !g) return fail(MDS_EINVAL, "mds_set_geometric_gains: null argument");
  if (!(g->max_tilt_angle > 0) || !(g->max_tilt_angle < M_PI / 2)) return fail(MDS_EINVAL, "mds_set_geometric_gains: max_tilt_angle");
  h->gains = *g;
  fill_consts(h->cfg, h->gains, h->cf, h->wind);
  fill_consts(h->cfg, h->gains, h->cd, h->wind);
  return MDS_OK;
}

int mds_set_wind(mds_handle* h, const double force_world[3]) {
  if (!h || !force_world) return fail(MDS_EINVAL, "mds_set_wind: null argument");
  for (int k = 0; k < 3; ++k) h->wind[k] = force_world[k];
  fill_consts(h->cfg, h->gains, h->cf, h->wind);
  fill_consts(h->cfg, h->gains, h->cd, h->wind);
  return MDS_OK;
}

// batch0 / nb: the range of 256-drone batches this launch covers (nb == 0: all of them)
static int launch_step_geometric(mds_handle* h, double t, void* obs, void* act, hipStream_t st, unsigned batch0 = 0, unsigned nb = 0) {
  const unsigned nbatch = nb ? nb : (unsigned)((h->n + kBlock - 1) / kBlock);
  // Half-shard launches of the two-stream rollout carry unused dynamic LDS up to 32 KB per workgroup: 5 workgroups per CU
  // instead of 8.  A half then no longer fits the chip next to the other chain's half, so its workgroups enter as the other
  // chain's retire and the two chains interleave from the first step on (C3: 15.6 -> 15.0 us per step over 20 000 steps,
  // 16.5 -> 15.8 over 2000; on one stream the same padding costs 4 %, so full-shard launches do not get it).
  size_t pad = 0;
  if (nb) {
    const size_t static_lds = (h->traj_mode == 2 || obs) ? (size_t)kBlock * kObsDim * elem_size(h->cfg.dtype) : 16;
    pad = static_lds < (size_t)h->split_lds ? (size_t)h->split_lds - static_lds : 0;
  }
  if (h->traj_mode == 2) {      // general trajectories: segment tables
    const bool rk4_ = h->cfg.integrator == MDS_INTEGRATOR_RK4, drag_ = has_drag(h);
#define MDS_TRAJ(RK4, DRAG)                                                                                                   \
  do {                                                                                                                        \
    if (is_comp(h))                                                                                                           \
      k_step_traj<float, float, RK4, DRAG, true><<<dim3(nbatch), kBlock, pad, st>>>(h->cf, h->n, h->ld, t, (float*)h->state, (const float*)h->origin, \
                                                                                    SegTable{h->segs, h->nseg_total}, h->tinfo, (float*)rpm_track(h), (float*)obs, (float*)act, (int)batch0, (float*)h->state_lo); \
    else                                                                                                                      \
      MDS_DISPATCH(h, (k_step_traj<T, S, RK4, DRAG><<<dim3(nbatch), kBlock, pad, st>>>(C, h->n, h->ld, t, (S*)h->state, (const T*)h->origin, \
                                                                                   SegTable{h->segs, h->nseg_total}, h->tinfo, (T*)rpm_track(h), (S*)obs, (S*)act, (int)batch0))); \
  } while (0)
    if (rk4_ && drag_) MDS_TRAJ(true, true);
    else if (rk4_) MDS_TRAJ(true, false);
    else if (drag_) MDS_TRAJ(false, true);
    else MDS_TRAJ(false, false);
#undef MDS_TRAJ
    return MDS_OK;
  }
#define MDS_LAUNCH_GEO2(HAS_OBS, HAS_ACT, RK4, DRAG)                                                                           \
  do {                                                                                                                         \
    if (is_comp(h))                                                                                                            \
      k_step_geometric<float, float, HAS_OBS, HAS_ACT, RK4, DRAG, true><<<grid, kBlock, pad, st>>>(h->cf, h->n, h->ld, t, (float*)h->state, \
                                                                                                   (const float*)h->lem, (float*)rpm_track(h), \
                                                                                                   (float*)obs, (float*)act, (int)batch0, (float*)h->state_lo); \
    else                                                                                                                       \
      MDS_DISPATCH(h, (k_step_geometric<T, S, HAS_OBS, HAS_ACT, RK4, DRAG><<<grid, kBlock, pad, st>>>(C, h->n, h->ld, t, (S*)h->state, \
                                                                                                    (const T*)h->lem, (T*)rpm_track(h), \
                                                                                                    (S*)obs, (S*)act, (int)batch0))); \
  } while (0)
  const bool rk4 = h->cfg.integrator == MDS_INTEGRATOR_RK4, drag = has_drag(h);
  const dim3 grid(nbatch);
#define MDS_LAUNCH_GEO(HAS_OBS, HAS_ACT)                         \
  do {                                                           \
    if (rk4 && drag) MDS_LAUNCH_GEO2(HAS_OBS, HAS_ACT, true, true);   \
    else if (rk4) MDS_LAUNCH_GEO2(HAS_OBS, HAS_ACT, true, false);     \
    else if (drag) MDS_LAUNCH_GEO2(HAS_OBS, HAS_ACT, false, true);    \
    else MDS_LAUNCH_GEO2(HAS_OBS, HAS_ACT, false, false);             \
  } while (0)
  if (obs && act) MDS_LAUNCH_GEO(true, true);
  else if (obs) MDS_LAUNCH_GEO(true, false);
  else if (act) MDS_LAUNCH_GEO(false, true);
  else MDS_LAUNCH_GEO(false, false);
#undef MDS_LAUNCH_GEO2
#undef MDS_LAUNCH_GEO
  return MDS_OK;
}

int mds_step_geometric(mds_handle* h, double t, void* obs, void* act, void* stream) {
  MDS_DEV(h);
  if (!h) return fail(MDS_EINVAL, "mds_step_geometric: null handle");
  if (!h->has_traj) return fail(MDS_ESTATE, "mds_step_geometric: call mds_set_lemniscate first");
  if (!aligned16(obs) || !aligned16(act)) return fail(MDS_EALIGN, "mds_step_geometric: obs_dev/action_dev");
  if (h->envfx) return step_env_ctrl(h, 0, t, nullptr, 0.0, obs, act, (hipStream_t)stream);
  launch_step_geometric(h, t, obs, act, (hipStream_t)stream);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

// One launch of the whole-rollout kernel: n_steps control steps with the state in registers.  ctrl: 0 GeometricControl, 1 LQRController
// (12-state), 2 LQROmegaController + ThrustOmega, 3 LQRYankOmegaController + YankOmega.  Step k's observation goes to obs_log + k * log_stride
// elements (log_stride = n * 20: a [n_steps, n, 20] log; 0: every step overwrites the same [n, 20] buffer, what a step-by-step loop with one
// observation buffer does); obs_last (or NULL) receives the last step's.
static void launch_rollout_kernel(mds_handle* h, int ctrl, double t0, int n_steps, void* obs_log, size_t log_stride, void* obs_last, hipStream_t st) {
  const dim3 grid = grid_for(h->n, kBlock);
  const double dt = 1.0 / h->cfg.ctrl_freq;
  const bool rk4 = h->cfg.integrator == MDS_INTEGRATOR_RK4, drag = has_drag(h);
  // the gain (up to 48 values) is read from its device copy: passing it by value would not fit beside Consts in SGPRs
  const void* gain = ctrl >= 1 ? h->gain_dev[ctrl - 1] : nullptr;
#define MDS_ROLL(RK4, DRAG, CTRL)                                                                                                  \
  MDS_DISPATCH(h, (k_rollout_geometric<T, S, RK4, DRAG, CTRL><<<grid, kBlock, 0, st>>>(C, gain, h->n, h->ld, t0, dt,                       \
                                                                                       n_steps, (S*)h->state, (const T*)h->lem,       \
                                                                                       (T*)rpm_track(h), (S*)obs_log, log_stride, (S*)obs_last,   \
                                                                                       (T*)h->ll, (const S*)obs_last, (S*)h->state_lo)))
#define MDS_ROLLT(RK4, DRAG, CTRL)                                                                                                 \
  MDS_DISPATCH(h, (k_rollout_traj<T, S, RK4, DRAG, CTRL><<<grid, kBlock, 0, st>>>(C, gain, h->n, h->ld, t0, dt,                                 \
                                                                                  n_steps, (S*)h->state, (const T*)h->origin, SegTable{h->segs, h->nseg_total}, \
                                                                                  h->tinfo, (T*)rpm_track(h), (S*)obs_log, log_stride, (S*)obs_last, (S*)h->state_lo)))
#define MDS_ROLL_C(CTRL)                                                    \
  do {                                                                      \
    if (h->traj_mode == 2) {   /* general trajectories: segment tables */   \
      if (rk4 && drag) MDS_ROLLT(true, true, CTRL);                         \
      else if (rk4) MDS_ROLLT(true, false, CTRL);                           \
      else if (drag) MDS_ROLLT(false, true, CTRL);                          \
      else MDS_ROLLT(false, false, CTRL);                                   \
    } else if (rk4 && drag) MDS_ROLL(true, true, CTRL);                     \
    else if (rk4) MDS_ROLL(true, false, CTRL);                              \
    else if (drag) MDS_ROLL(false, true, CTRL);                             \
    else MDS_ROLL(false, false, CTRL);                                      \
  } while (0)
#define MDS_ROLL_LL(CTRL)                        \
  do {                                           \
    if (rk4 && drag) MDS_ROLL(true, true, CTRL); \
    else if (rk4) MDS_ROLL(true, false, CTRL);   \
    else if (drag) MDS_ROLL(false, true, CTRL);  \
    else MDS_ROLL(false, false, CTRL);           \
  } while (0)
  if (ctrl == 3) MDS_ROLL_LL(3);
  else if (ctrl == 2) MDS_ROLL_LL(2);
  else if (ctrl == 1) MDS_ROLL_C(1);
  else MDS_ROLL_C(0);
#undef MDS_ROLL_LL
#undef MDS_ROLL_C
#undef MDS_ROLLT
#undef MDS_ROLL
}

int mds_rollout_geometric(mds_handle* h, double t0, int n_steps, void* obs, int obs_every_step, void* stream) {
  MDS_DEV(h);
  if (!h || n_steps < 0) return fail(MDS_EINVAL, "mds_rollout_geometric");
  if (!h->has_traj) return fail(MDS_ESTATE, "mds_rollout_geometric: call mds_set_lemniscate first");
  if (!aligned16(obs)) return fail(MDS_EALIGN, "mds_rollout_geometric: obs_dev");
  const double dt = 1.0 / h->cfg.ctrl_freq;
  if (h->envfx) {          // ground effect / downwash: every substep is its own launch on the double-buffered state
    h->last_rollout_streams = 1;
    for (int k = 0; k < n_steps; ++k) {
      if (int rc = step_env_ctrl(h, 0, t0, nullptr, 0.0, (obs_every_step || k == n_steps - 1) ? obs : nullptr, nullptr, (hipStream_t)stream)) return rc;
      t0 += dt;
    }
    return MDS_OK;
  }
  if (rollout_form_policy(h, n_steps) == 2) {
    // launch-bound shard sizes: the same loop through the whole-rollout kernel, rollout_chunk control steps per launch; every step's
    // observation overwrites obs (obs_every_step) exactly as the per-step loop's launches do, or only the last step's is written
    h->last_rollout_streams = 1;
    h->last_rollout_form = 2;
    for (int k = 0; k < n_steps;) {
      const int chunk = n_steps - k < h->rollout_chunk ? n_steps - k : h->rollout_chunk;
      const bool last = k + chunk == n_steps;
      launch_rollout_kernel(h, 0, t0, chunk, obs_every_step ? obs : nullptr, 0, (!obs_every_step && last) ? obs : nullptr, (hipStream_t)stream);
      for (int j = 0; j < chunk; ++j) t0 += dt;        // t accumulates step by step, as inside the kernel and in the reference loop
      k += chunk;
    }
    MDS_HIP(hipGetLastError());
    return MDS_OK;
  }
  h->last_rollout_form = 1;
  const unsigned nbatch = (unsigned)((h->n + kBlock - 1) / kBlock);
  const int streams = rollout_streams_policy(h, 0, n_steps);
  if (streams == 2 && nbatch >= 2) {
    // Drones never read each other's rows in this kernel, so the two halves of the shard are two independent step
    // chains.  Run on two streams they drift out of phase: one half's load/store bursts fill the other's compute
    // phase (measured on C3: 17.5 -> 15.7-16.3 us per step).  The caller's stream orders both chains.
    hipStream_t st = (hipStream_t)stream, sb = h->split_st;
    h->last_rollout_streams = 2;
    const unsigned half = nbatch / 2;
    auto body = [&]() -> int {
      for (int k = 0; k < n_steps; ++k) {
        void* o = (obs_every_step || k == n_steps - 1) ? obs : nullptr;
        if (k == 0 && h->split_offset && half >= 2) {
          // phase offset: the fork event sits half way through chain 0's first step, so chain 1 starts half a kernel late
          // (started together the chains begin in lock step and need a few hundred steps to drift apart)
          launch_step_geometric(h, t0, o, nullptr, st, 0, half / 2);
          if (int rc = split_fork(h, st)) return rc;
          launch_step_geometric(h, t0, o, nullptr, st, half / 2, half - half / 2);
        } else {
          if (k == 0)
            if (int rc = split_fork(h, st)) return rc;
          launch_step_geometric(h, t0, o, nullptr, st, 0, half);
        }
        launch_step_geometric(h, t0, o, nullptr, sb, half, nbatch - half);
        t0 += dt;
      }
      MDS_HIP(hipGetLastError());
      return MDS_OK;
    };
    return split_join(h, st, body());
  }
  h->last_rollout_streams = 1;
  for (int k = 0; k < n_steps; ++k) {
    // t accumulates exactly like the reference loop (t += env.CTRL_TIMESTEP, EnvGeometric.py:473)
    launch_step_geometric(h, t0, (obs_every_step || k == n_steps - 1) ? obs : nullptr, nullptr, (hipStream_t)stream);
    t0 += dt;
  }
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_rollout_step(mds_handle* h, const void* actions, int n_action_sets, int first_step, int n_steps, void* obs_log, int log_slots,
                     int episode_len, void* stream) {
  MDS_DEV(h);
  if (!h || !actions || n_action_sets < 1 || first_step < 0 || n_steps < 0 || (obs_log && log_slots < 1) || episode_len < 0)
    return fail(MDS_EINVAL, "mds_rollout_step: arguments");
  if (episode_len > 0 && !h->init_pose) return fail(MDS_ESTATE, "mds_rollout_step: episode resets need an earlier mds_reset");
  const size_t es = elem_size(h->cfg.dtype), act_bytes = (size_t)h->n * 4 * es, obs_bytes = (size_t)h->n * kObsDim * es;
  if (!aligned16(actions) || !aligned16(obs_log) || (n_action_sets > 1 && act_bytes % 16) || (obs_log && log_slots > 1 && obs_bytes % 16))
    return fail(MDS_EALIGN, "mds_rollout_step: actions_dev/obs_log_dev (every action set and log slot must start 16-byte aligned)");
  hipStream_t st = (hipStream_t)stream;
  if (h->envfx) {          // ground effect / downwash: the env.step loop, one launch per substep
    h->last_rollout_streams = 1;
    for (int k = 0; k < n_steps; ++k) {
      const long long j = (long long)first_step + k;
      if (episode_len > 0 && j > 0 && j % episode_len == 0)
        if (int rc = launch_reset_range(h, st, 0, h->n)) return rc;
      const char* a = (const char*)actions + (size_t)(j % n_action_sets) * act_bytes;
      char* o = obs_log ? (char*)obs_log + (size_t)(j % log_slots) * obs_bytes : nullptr;
      if (int rc = step_env_plain(h, a, o, st, 0)) return rc;
    }
    return MDS_OK;
  }
  const unsigned nbatch = (unsigned)((h->n + kBlock - 1) / kBlock);
  const bool split = rollout_streams_policy(h, 0, n_steps) == 2 && nbatch >= 2;
  const unsigned half = nbatch / 2;
  if (split)         // the two halves of the shard as independent step chains, as in mds_rollout_geometric
    if (int rc = split_fork(h, st)) return rc;
  h->last_rollout_streams = split ? 2 : 1;
  auto body = [&]() -> int {
    for (int k = 0; k < n_steps; ++k) {
      const long long j = (long long)first_step + k;
      const char* a = (const char*)actions + (size_t)(j % n_action_sets) * act_bytes;
      char* o = obs_log ? (char*)obs_log + (size_t)(j % log_slots) * obs_bytes : nullptr;
      if (episode_len > 0 && j > 0 && j % episode_len == 0) {      // a new episode starts at step j: back to the initial poses
        const size_t mid = (size_t)half * kBlock;
        if (split) {
          if (int rc = launch_reset_range(h, st, 0, mid)) return rc;
          if (int rc = launch_reset_range(h, h->split_st, mid, h->n)) return rc;
        } else if (int rc = launch_reset_range(h, st, 0, h->n)) {
          return rc;
        }
      }
      if (split) {
        launch_step_plain(h, a, o, st, 0, half);
        launch_step_plain(h, a, o, h->split_st, half, nbatch - half);
      } else {
        launch_step_plain(h, a, o, st);
      }
    }
    MDS_HIP(hipGetLastError());
    return MDS_OK;
  };
  const int rc = body();
  return split ? split_join(h, st, rc) : rc;
}

int mds_rollout_step_fused(mds_handle* h, const void* actions, int n_action_sets, int first_step, int n_steps, void* obs_log,
                           int log_slots, int episode_len, int steps_per_launch, void* stream) {
  MDS_DEV(h);
  if (!h || !actions || n_action_sets < 1 || first_step < 0 || n_steps < 0 || (obs_log && log_slots < 1) || episode_len < 0 ||
      steps_per_launch < 1)
    return fail(MDS_EINVAL, "mds_rollout_step_fused: arguments");
  if (h->envfx)                 // env-mates interact every substep: no state-in-registers form; the step-by-step loop serves it
    return mds_rollout_step(h, actions, n_action_sets, first_step, n_steps, obs_log, log_slots, episode_len, stream);
  if (episode_len > 0 && !h->init_pose) return fail(MDS_ESTATE, "mds_rollout_step_fused: episode resets need an earlier mds_reset");
  const size_t es = elem_size(h->cfg.dtype), act_bytes = (size_t)h->n * 4 * es, obs_bytes = (size_t)h->n * kObsDim * es;
  if (!aligned16(actions) || !aligned16(obs_log) || (n_action_sets > 1 && act_bytes % 16) || (obs_log && log_slots > 1 && obs_bytes % 16))
    return fail(MDS_EALIGN, "mds_rollout_step_fused: actions_dev/obs_log_dev (every action set and log slot must start 16-byte aligned)");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid = grid_for(h->n, kBlock);
  const bool rk4 = h->cfg.integrator == MDS_INTEGRATOR_RK4, drag = has_drag(h);
  long long j = first_step;
  const long long j_end = (long long)first_step + n_steps;
  while (j < j_end) {
    if (episode_len > 0 && j > 0 && j % episode_len == 0)
      if (int rc = launch_reset_range(h, st, 0, h->n)) return rc;
    long long chunk = j_end - j < steps_per_launch ? j_end - j : steps_per_launch;
    if (episode_len > 0) {                                  // a launch never crosses an episode boundary
      const long long to_boundary = episode_len - j % episode_len;
      if (chunk > to_boundary) chunk = to_boundary;
    }
    const int a0 = (int)(j % n_action_sets), s0 = obs_log ? (int)(j % log_slots) : 0;
#define MDS_RS(RK4, DRAG)                                                                                                     \
  MDS_DISPATCH(h, (k_rollout_step<T, S, RK4, DRAG><<<grid, kBlock, 0, st>>>(C, h->n, h->ld, (S*)h->state, (const T*)h->origin,  \
                                                                            (T*)rpm_track(h), (const S*)actions, a0, n_action_sets, \
                                                                            (S*)obs_log, s0, obs_log ? log_slots : 1, (int)chunk, \
                                                                            (S*)h->state_lo)))
    if (rk4 && drag) MDS_RS(true, true);
    else if (rk4) MDS_RS(true, false);
    else if (drag) MDS_RS(false, true);
    else MDS_RS(false, false);
#undef MDS_RS
    j += chunk;
  }
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_set_rollout_streams(mds_handle* h, int n_streams) {
  if (!h || n_streams < 0 || n_streams > 2) return fail(MDS_EINVAL, "mds_set_rollout_streams: 0 (auto), 1 or 2");
  MDS_DEV(h);
  if (n_streams == 2)          // a set-up call: the only place besides mds_create that creates the internal streams
    if (int rc = split_streams_ready(h)) return rc;
  h->rollout_streams = n_streams;
  return MDS_OK;
}

int mds_rollout_streams_for(const mds_handle* h, int loop, int n_steps) {
  if (!h || loop < 0 || loop > 1 || n_steps < 0) return fail(MDS_EINVAL, "mds_rollout_streams_for");
  return rollout_streams_policy(h, loop, n_steps);
}

int mds_get_last_rollout_streams(const mds_handle* h) {
  if (!h) return fail(MDS_EINVAL, "mds_get_last_rollout_streams: null handle");
  return h->last_rollout_streams;
}

int mds_set_rollout_form(mds_handle* h, int form, int steps_per_launch) {
  if (!h || form < 0 || form > 2 || steps_per_launch < 0) return fail(MDS_EINVAL, "mds_set_rollout_form: form 0 (auto), 1 or 2; steps_per_launch >= 0 (0: keep)");
  h->rollout_form = form;
  if (steps_per_launch > 0) h->rollout_chunk = steps_per_launch;
  return MDS_OK;
}

int mds_rollout_form_for(const mds_handle* h, int n_steps) {
  if (!h || n_steps < 0) return fail(MDS_EINVAL, "mds_rollout_form_for");
  return rollout_form_policy(h, n_steps);
}

int mds_get_last_rollout_form(const mds_handle* h) {
  if (!h) return fail(MDS_EINVAL, "mds_get_last_rollout_form: null handle");
  return h->last_rollout_form;
}

// ctrl: 0 GeometricControl, 1 LQRController (12-state), 2 LQROmegaController + ThrustOmega, 3 LQRYankOmegaController + YankOmega
int rollout_fused(mds_handle* h, double t0, int n_steps, void* obs_log, void* obs_last, void* stream, int ctrl, const char* who) {
  if (!h || n_steps < 0) return fail(MDS_EINVAL, who);
  if (h->envfx) {
    // ground effect / downwash: env-mates interact every physics substep, so there is no state-in-registers form; the same loop
    // runs step by step (one launch per substep), each step's observation written straight into its slot of the log
    if (!h->has_traj) return fail(MDS_ESTATE, "mds_rollout_*_fused: call mds_set_lemniscate first");
    if (ctrl == 1 && !h->has_lqr12) return fail(MDS_ESTATE, "mds_rollout_lqr_fused: call mds_set_lqr_gain first");
    if (ctrl >= 2 && !obs_last) return fail(MDS_EINVAL, "mds_rollout_nominal_fused: obs_dev is required (in: current observation, out: last one)");
    if (!aligned16(obs_log) || !aligned16(obs_last)) return fail(MDS_EALIGN, "mds_rollout_*_fused: obs buffers");
    const size_t obs_bytes = (size_t)h->n * kObsDim * elem_size(h->cfg.dtype);
    const double dt = 1.0 / h->cfg.ctrl_freq;
    hipStream_t st = (hipStream_t)stream;
    for (int k = 0; k < n_steps; ++k) {
      char* slot = obs_log ? (char*)obs_log + (size_t)k * obs_bytes : nullptr;
      const bool last = k == n_steps - 1;
      if (ctrl <= 1) {
        void* o = slot ? (void*)slot : (last ? obs_last : nullptr);
        if (int rc = step_env_ctrl(h, ctrl, t0, nullptr, 0.0, o, nullptr, st)) return rc;
        if (slot && last && obs_last) MDS_HIP(hipMemcpyAsync(obs_last, slot, obs_bytes, hipMemcpyDeviceToDevice, st));
      } else {
        if (int rc = step_nominal_lowlevel(h, t0, obs_last, nullptr, nullptr, stream, false, who, nullptr, false, false, 0.0)) return rc;
        if (slot) MDS_HIP(hipMemcpyAsync(slot, obs_last, obs_bytes, hipMemcpyDeviceToDevice, st));
      }
      t0 += dt;
    }
    return MDS_OK;
  }
  if (!h->has_traj) return fail(MDS_ESTATE, "mds_rollout_*_fused: call mds_set_lemniscate first");
  const bool lqr = ctrl == 1;
  if (lqr && !h->has_lqr12) return fail(MDS_ESTATE, "mds_rollout_lqr_fused: call mds_set_lqr_gain first");
  if (ctrl >= 2) {
    if (h->traj_mode != 1) return fail(MDS_EUNSUPPORTED, "mds_rollout_nominal_fused: Lemniscate trajectories only (use mds_step_nominal)");
    if (h->cfg.dtype == MDS_F16) return fail(MDS_EUNSUPPORTED, "mds_rollout_nominal_fused: fp16 storage");
    if (!obs_last) return fail(MDS_EINVAL, "mds_rollout_nominal_fused: obs_dev is required (in: current observation, out: last one)");
  }
  if (!aligned16(obs_log) || !aligned16(obs_last)) return fail(MDS_EALIGN, "mds_rollout_*_fused: obs buffers");
  if (n_steps == 0) return MDS_OK;
  launch_rollout_kernel(h, ctrl, t0, n_steps, obs_log, (size_t)h->n * kObsDim, obs_last, (hipStream_t)stream);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_rollout_geometric_fused(mds_handle* h, double t0, int n_steps, void* obs_log, void* obs_last, void* stream) {
  MDS_DEV(h);
  return rollout_fused(h, t0, n_steps, obs_log, obs_last, stream, 0, "mds_rollout_geometric_fused");
}

int mds_rollout_lqr_fused(mds_handle* h, double t0, int n_steps, void* obs_log, void* obs_last, void* stream) {
  MDS_DEV(h);
  return rollout_fused(h, t0, n_steps, obs_log, obs_last, stream, 1, "mds_rollout_lqr_fused");
}

int mds_rollout_nominal_fused(mds_handle* h, double t0, int n_steps, void* obs_log, void* obs, void* stream) {
  MDS_DEV(h);
  if (!h) return fail(MDS_EINVAL, "mds_rollout_nominal_fused: null handle");
  if (h->cbf_nominal != 1 && h->cbf_nominal != 2)
    return fail(MDS_ESTATE, "mds_rollout_nominal_fused: select the LQR-omega (1) or LQR-yank-omega (2) controller with mds_cbf_set_nominal first");
  return rollout_fused(h, t0, n_steps, obs_log, obs, stream, h->cbf_nominal == 1 ? 2 : 3, "mds_rollout_nominal_fused");
}

int mds_lemniscate_eval(mds_handle* h, double t, void* des, void* stream) {
  MDS_DEV(h);
  if (!h || !des) return fail(MDS_EINVAL, "mds_lemniscate_eval: null argument");
  if (!h->has_traj || h->traj_mode != 1) return fail(MDS_ESTATE, "mds_lemniscate_eval: call mds_set_lemniscate first");
  MDS_DISPATCH(h, (k_lemniscate_eval<T, S><<<grid_for(h->n, 256), 256, 0, (hipStream_t)stream>>>(h->n, h->ld, t, (const T*)h->lem, (S*)des)));
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_geometric_compute(mds_handle* h, const void* obs, const void* des, void* rpm, void* aux, void* stream) {
  MDS_DEV(h);
  if (!h || !obs || !des || !rpm) return fail(MDS_EINVAL, "mds_geometric_compute: null argument");
  if (!aligned16(rpm)) return fail(MDS_EALIGN, "mds_geometric_compute: rpm_dev");
  MDS_DISPATCH(h, (k_geometric_compute<T, S><<<grid_for(h->n, 256), 256, 0, (hipStream_t)stream>>>(C, h->n, (const S*)obs, (const S*)des,
                                                                                                     (S*)rpm, (S*)aux)));
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_input_to_action(mds_handle* h, const void* u, void* rpm, void* stream) {
  MDS_DEV(h);
  if (!h || !u || !rpm) return fail(MDS_EINVAL, "mds_input_to_action: null argument");
  if (!aligned16(u) || !aligned16(rpm)) return fail(MDS_EALIGN, "mds_input_to_action");
  MDS_DISPATCH(h, (k_input_to_action<T, S><<<grid_for(h->n, 256), 256, 0, (hipStream_t)stream>>>(C, h->n, (const S*)u, (S*)rpm)));
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_obs_to_model(mds_handle* h, const void* obs, int dim, void* x, void* stream) {
  MDS_DEV(h);
  if (!h || !obs || !x) return fail(MDS_EINVAL, "mds_obs_to_model: null argument");
  if (dim != 9 && dim != 10 && dim != 12 && dim != 18) return fail(MDS_EINVAL, "mds_obs_to_model: dim must be 9, 10, 12 (linear models) or 18 (geometric model)");
  MDS_DISPATCH(h, (k_obs_to_model<T, S><<<grid_for(h->n, 256), 256, 0, (hipStream_t)stream>>>(C, h->n, dim, (const S*)obs, (S*)x)));
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_action_to_input(mds_handle* h, const void* rpm, int cap_rpm, void* u, void* stream) {
  MDS_DEV(h);
  if (!h || !u || !rpm) return fail(MDS_EINVAL, "mds_action_to_input: null argument");
  if (!aligned16(u) || !aligned16(rpm)) return fail(MDS_EALIGN, "mds_action_to_input");
  MDS_DISPATCH(h, (k_action_to_input<T, S><<<grid_for(h->n, 256), 256, 0, (hipStream_t)stream>>>(C, h->n, cap_rpm, (const S*)rpm, (S*)u)));
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_quadrotor_dynamics(int dtype, int count, const void* state, const void* u, double m, const double J[3], double g,
                           void* out, void* stream) {
  if (count < 0 || !state || !u || !out || !J) return fail(MDS_EINVAL, "mds_quadrotor_dynamics: bad argument");
  if (count == 0) return MDS_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid = grid_for(count, 256);
  if (dtype == MDS_F32)
    k_quadrotor_dynamics<float, float><<<grid, 256, 0, st>>>(count, (const float*)state, (const float*)u, (float)m, (float)J[0],
                                                              (float)J[1], (float)J[2], (float)g, (float*)out);
  else if (dtype == MDS_F64)
    k_quadrotor_dynamics<double, double><<<grid, 256, 0, st>>>(count, (const double*)state, (const double*)u, m, J[0], J[1], J[2], g,
                                                                (double*)out);
  else if (dtype == MDS_F16)
    k_quadrotor_dynamics<float, half_t><<<grid, 256, 0, st>>>(count, (const half_t*)state, (const half_t*)u, (float)m, (float)J[0],
                                                               (float)J[1], (float)J[2], (float)g, (half_t*)out);
  else
    return fail(MDS_EINVAL, "mds_quadrotor_dynamics: dtype");
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_compare_models(mds_handle* h, int count, const void* obs, const double* A, const double* B, double u_eq0, double dyn_m,
                       const double dyn_J[3], double dyn_g, void* xdot_lin, void* xdot_geo, void* x_lin, void* stream) {
  MDS_DEV(h);
  if (!h || count < 0 || !obs || !A || !B || !dyn_J) return fail(MDS_EINVAL, "mds_compare_models: bad argument");
  if (!xdot_lin && !xdot_geo && !x_lin) return fail(MDS_EINVAL, "mds_compare_models: no output requested");
  if (!aligned16(obs) || !aligned16(xdot_lin) || !aligned16(xdot_geo) || !aligned16(x_lin)) return fail(MDS_EALIGN, "mds_compare_models");
  if (count == 0) return MDS_OK;
  MDS_DISPATCH(h, (launch_compare_models<T, S>(C, count, obs, A, B, u_eq0, dyn_m, dyn_J, dyn_g, xdot_lin, xdot_geo, x_lin, (hipStream_t)stream)));
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_linear_xdot(mds_handle* h, int count, const void* x, const void* action, const double* A, const double* B, double u_eq0, void* xdot,
                    void* stream) {
  MDS_DEV(h);
  if (!h || count < 0 || !x || !action || !A || !B || !xdot) return fail(MDS_EINVAL, "mds_linear_xdot: bad argument");
  if (count == 0) return MDS_OK;
  MDS_DISPATCH(h, (launch_linear_xdot<T, S>(C, count, x, action, A, B, u_eq0, xdot, (hipStream_t)stream)));
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_rpy_to_rot(int dtype, int count, const void* rpy, void* R, void* stream) {
  if (count < 0 || !rpy || !R) return fail(MDS_EINVAL, "mds_rpy_to_rot: bad argument");
  if (count == 0) return MDS_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid = grid_for(count, 256);
  if (dtype == MDS_F32) k_rpy_to_rot<float, float><<<grid, 256, 0, st>>>(count, (const float*)rpy, (float*)R);
  else if (dtype == MDS_F64) k_rpy_to_rot<double, double><<<grid, 256, 0, st>>>(count, (const double*)rpy, (double*)R);
  else if (dtype == MDS_F16) k_rpy_to_rot<float, half_t><<<grid, 256, 0, st>>>(count, (const half_t*)rpy, (half_t*)R);
  else return fail(MDS_EINVAL, "mds_rpy_to_rot: dtype");
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_geo_model_to_obs(int dtype, int count, const void* x18, void* obs16, void* stream) {
  if (count < 0 || !x18 || !obs16) return fail(MDS_EINVAL, "mds_geo_model_to_obs: bad argument");
  if (count == 0) return MDS_OK;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid = grid_for(count, 256);
  if (dtype == MDS_F32) k_geo_model_to_obs<float, float><<<grid, 256, 0, st>>>(count, (const float*)x18, (float*)obs16);
  else if (dtype == MDS_F64) k_geo_model_to_obs<double, double><<<grid, 256, 0, st>>>(count, (const double*)x18, (double*)obs16);
  else if (dtype == MDS_F16) k_geo_model_to_obs<float, half_t><<<grid, 256, 0, st>>>(count, (const half_t*)x18, (half_t*)obs16);
  else return fail(MDS_EINVAL, "mds_geo_model_to_obs: dtype");
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

#endif  // MDS_PART & 1
#if MDS_PART & 2
int mds_cbf_configure(mds_handle* h, const mds_cbf_params* p, const double* obstacles) {
  MDS_DEV(h);
  if (!h || !p) return fail(MDS_EINVAL, "mds_cbf_configure: null argument");
  if (p->order != 2 && p->order != 3) return fail(MDS_EINVAL, "mds_cbf_configure: order must be 2 or 3");
  if (p->n_obs < 0 || p->n_obs > kCbfMaxObs) return fail(MDS_EINVAL, "mds_cbf_configure: n_obs out of range");
  if (p->n_obs > 0 && !obstacles) return fail(MDS_EINVAL, "mds_cbf_configure: obstacles_host is NULL");
  if (h->cfg.num_drones > kCbfMaxD) return fail(MDS_EUNSUPPORTED, "mds_cbf_configure: more than 32 drones per env");
  if (h->cfg.dtype == MDS_F16) return fail(MDS_EUNSUPPORTED, "mds_cbf_configure: fp16 storage");
  if (!(p->zscale > 0) || !(p->safety_radius > 0)) return fail(MDS_EINVAL, "mds_cbf_configure: zscale / safety_radius");
  const int D = h->cfg.num_drones, npairs = D * (D - 1) / 2;
  if (!h->pair_ij && npairs > 0) {
    int* host = new (std::nothrow) int[npairs];
    if (!host) return fail(MDS_ENOMEM, "mds_cbf_configure: host allocation");
    int r = 0;
    for (int i = 0; i < D - 1; ++i)
      for (int j = i + 1; j < D; ++j) host[r++] = i | (j << 8);
    hipError_t e = hipMalloc((void**)&h->pair_ij, sizeof(int) * npairs);
    if (e == hipSuccess) e = hipMemcpy(h->pair_ij, host, sizeof(int) * npairs, hipMemcpyHostToDevice);
    delete[] host;
    if (e != hipSuccess) return fail_hip(e, "mds_cbf_configure: pair table");
  }
  if (!h->obstacles) MDS_HIP(hipMalloc(&h->obstacles, sizeof(double) * 4 * kCbfMaxObs));
  if (p->n_obs > 0) {
    if (h->cfg.dtype == MDS_F64) {
      MDS_HIP(hipMemcpy(h->obstacles, obstacles, sizeof(double) * 4 * p->n_obs, hipMemcpyHostToDevice));
    } else {
      float tmp[4 * kCbfMaxObs];
      for (int k = 0; k < 4 * p->n_obs; ++k) tmp[k] = (float)obstacles[k];
      MDS_HIP(hipMemcpy(h->obstacles, tmp, sizeof(float) * 4 * p->n_obs, hipMemcpyHostToDevice));
    }
  }
  if (!h->cbf_order) MDS_HIP(hipMalloc((void**)&h->cbf_order, sizeof(int) * 9 * (size_t)h->cfg.num_envs));
  if (!h->cbf_count) MDS_HIP(hipMalloc((void**)&h->cbf_count, sizeof(int) * 12));
  if (!h->cbf_cost) {
    MDS_HIP(hipMalloc((void**)&h->cbf_cost, sizeof(int) * (size_t)h->cfg.num_envs));
    MDS_HIP(hipMemset(h->cbf_cost, 0, sizeof(int) * (size_t)h->cfg.num_envs));
  }
  h->cbf_calls = h->cbf_calls_half[0] = h->cbf_calls_half[1] = -1;   // a new problem: forget the cost classes
  {
    const char* solver = getenv("MDS_CBF_SOLVER");
    h->cbf_hildreth = solver && solver[0] == 'h';
    const char* q4e = getenv("MDS_CBF_Q4");
    h->cbf_q4 = q4e && q4e[0] == '1';
    const char* chain = getenv("MDS_CBF_CHAIN");
    h->cbf_chain_nominal = !(chain && chain[0] == '0');
    if (const char* fused = getenv("MDS_CBF_FUSED")) h->cbf_fused = fused[0] == '1';       // unset: what mds_cbf_set_step_kernel chose
  }
  h->cbf = *p;
  fill_cbf(h, *p, h->cbf_f);
  fill_cbf(h, *p, h->cbf_d);
  h->has_cbf = true;
  return MDS_OK;
}

int mds_cbf_num_rows(const mds_handle* h) {
  if (!h || !h->has_cbf) return fail(MDS_ESTATE, "mds_cbf_num_rows: call mds_cbf_configure first");
  return cbf_num_rows(h->cfg.num_drones, h->cbf.order, h->cbf.n_obs);
}

int mds_cbf_rows(mds_handle* h, const void* x, const void* xdes, void* G, void* hv, void* stream) {
  MDS_DEV(h);
  if (!h || !x || !xdes || !G || !hv) return fail(MDS_EINVAL, "mds_cbf_rows: null argument");
  if (!h->has_cbf) return fail(MDS_ESTATE, "mds_cbf_rows: call mds_cbf_configure first");
  hipStream_t st = (hipStream_t)stream;
  const int E = h->cfg.num_envs;
#define MDS_ROWS(T, CP, ORD)                                                                                            \
  k_cbf_rows<T, T, ORD><<<E, 256, 0, st>>>(CP, E, h->pair_ij, (const T*)h->obstacles, (const T*)x, (const T*)xdes, (T*)G, (T*)hv)
  if (h->cfg.dtype == MDS_F64) {
    if (h->cbf.order == 2) MDS_ROWS(double, h->cbf_d, 2);
    else MDS_ROWS(double, h->cbf_d, 3);
  } else {
    if (h->cbf.order == 2) MDS_ROWS(float, h->cbf_f, 2);
    else MDS_ROWS(float, h->cbf_f, 3);
  }
#undef MDS_ROWS
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

// envs [e0, e0 + ne) of the batch; slot selects the cost-class tables (0: whole batch, 1 / 2: halves of a two-stream rollout)
struct EnvRange {
  int e0, ne, slot;
};

static int cbf_filter_range(mds_handle* h, const void* obs_, const void* xdes_, const void* unom_, void* usafe_, int32_t* status_,
                            void* stream, EnvRange rg) {
  hipStream_t st = (hipStream_t)stream;
  const int E = rg.ne, D = h->cfg.num_drones, order = h->cbf.order;
  // an env's blocks are contiguous in every operand: a range is a pointer offset
  const size_t es = elem_size(h->cfg.dtype), d0 = (size_t)rg.e0 * D;
  const char* obs = (const char*)obs_ + d0 * kObsDim * es;
  const char* xdes = (const char*)xdes_ + d0 * (order == 2 ? 9 : 10) * es;
  const char* unom = (const char*)unom_ + d0 * 4 * es;
  char* usafe = (char*)usafe_ + d0 * 4 * es;
  int32_t* status = status_ + rg.e0;
  int& calls = rg.slot == 0 ? h->cbf_calls : h->cbf_calls_half[rg.slot - 1];
  int* const order_tab = h->cbf_order + (size_t)rg.slot * 3 * h->cfg.num_envs;
  int* const count_tab = h->cbf_count + 4 * rg.slot;
  int* const cost_tab = h->cbf_cost + rg.e0;
  const int nv = order == 2 ? 1 : 3, n = nv * D;
  const int m = D * (D - 1) / 2 + D * h->cbf.n_obs + 2 * n;          // rows of the coupled sub-problem
  const int R = (m + 63) / 64;
  const int max_iter = h->cbf.max_iter > 0 ? h->cbf.max_iter : 64 * m;
  const dim3 grid((unsigned)((E + 3) / 4));                // Hildreth kernel: 4 envs per workgroup
  const bool hildreth = h->cbf_hildreth && order == 2;     // MDS_CBF_SOLVER=hildreth at configure time: the coordinate-ascent kernel (A/B)
  if (n > 64) return fail(MDS_EUNSUPPORTED, "mds_cbf_filter: more than 64 coupled QP variables per env");
  if (R > 17) return fail(MDS_EUNSUPPORTED, "mds_cbf_filter: too many rows per env");
  // longest-first dispatch (see k_cbf_filter_gi): classes from the iteration counts of an earlier launch, rebuilt every 8th call
  const int* ord_in = calls < 0 ? nullptr : order_tab;
  const int* cnt_in = calls < 0 ? nullptr : count_tab;
  int* cost_out = hildreth ? nullptr : cost_tab;
  // one wavefront (= one env) per workgroup; NMAX bounds the QP variables (LDS footprint of Q, R ~ NMAX^2)
#define MDS_GI(T, CP, RR, NMAX, ORD, TOL)                                                                                   \
  k_cbf_filter_gi<T, T, RR, NMAX, ORD><<<dim3((unsigned)E), 64, 0, st>>>(                                                    \
      CP, E, (T)h->cfg.KF, h->pair_ij, (const T*)h->obstacles, (const T*)obs, (const T*)xdes, (const T*)unom, (T*)usafe,     \
      (int*)status, max_iter, (T)((TOL) * (TOL)), ord_in, cnt_in, cost_out)
#define MDS_GI_R(T, CP, NMAX, ORD, TOL)            \
  do {                                             \
    if (R <= 4) MDS_GI(T, CP, 4, NMAX, ORD, TOL);  \
    else if (R <= 8) MDS_GI(T, CP, 8, NMAX, ORD, TOL); \
    else MDS_GI(T, CP, 17, NMAX, ORD, TOL);        \
  } while (0)
#define MDS_GI_ALL(T, CP, TOL)                                       \
  do {                                                               \
    if (order == 2) {                                                \
      if (n <= 16) MDS_GI_R(T, CP, 16, 2, TOL);                      \
      else MDS_GI_R(T, CP, 32, 2, TOL);                              \
    } else {                                                         \
      if (n <= 24) MDS_GI_R(T, CP, 24, 3, TOL);                      \
      else if (n <= 48) MDS_GI_R(T, CP, 48, 3, TOL);                 \
      else MDS_GI_R(T, CP, 63, 3, TOL);                              \
    }                                                                \
  } while (0)
#define MDS_HILD(T, CP, RR, TOL)                                                                                          \
  k_cbf_filter_o2<T, T, RR><<<grid, 256, 0, st>>>(CP, E, h->pair_ij, (const T*)h->obstacles, (const T*)obs, (const T*)xdes, \
                                                  (const T*)unom, (T*)usafe, (int*)status, max_iter, (T)((TOL) * (TOL)))
#define MDS_HILD_R(T, CP, TOL)              \
  do {                                      \
    if (R <= 4) MDS_HILD(T, CP, 4, TOL);    \
    else if (R <= 8) MDS_HILD(T, CP, 8, TOL); \
    else MDS_HILD(T, CP, 17, TOL);          \
  } while (0)
  // order 2 with at most 16 thrust variables and 224 rows: four envs per wavefront, one per 16-lane row (k_cbf_filter_q4)
  const bool q4 = h->cbf_q4 && !hildreth && order == 2 && n <= 16 && m <= 224;
#define MDS_Q4(T, CP, RL, TOL)                                                                                                    \
  k_cbf_filter_q4<T, T, RL><<<dim3((unsigned)((E + 3) / 4)), 64, 0, st>>>(CP, E, h->pair_ij, (const T*)h->obstacles, (const T*)obs, \
                                                                          (const T*)xdes, (const T*)unom, (T*)usafe, (int*)status,  \
                                                                          max_iter, (T)((TOL) * (TOL)), cost_out)
#define MDS_Q4_R(T, CP, TOL)               \
  do {                                     \
    if (m <= 128) MDS_Q4(T, CP, 8, TOL);   \
    else MDS_Q4(T, CP, 14, TOL);           \
  } while (0)
  if (h->cfg.dtype == MDS_F64) {
    const double tol = h->cbf.tol > 0 ? h->cbf.tol : 1e-12;
    if (q4) MDS_Q4_R(double, h->cbf_d, tol);
    else if (hildreth) MDS_HILD_R(double, h->cbf_d, tol);
    else MDS_GI_ALL(double, h->cbf_d, tol);
  } else {
    const double tol = h->cbf.tol > 0 ? h->cbf.tol : 1e-6;
    if (q4) MDS_Q4_R(float, h->cbf_f, tol);
    else if (hildreth) MDS_HILD_R(float, h->cbf_f, tol);
    else MDS_GI_ALL(float, h->cbf_f, tol);
  }
#undef MDS_Q4_R
#undef MDS_Q4
#undef MDS_HILD_R
#undef MDS_HILD
#undef MDS_GI_ALL
#undef MDS_GI_R
#undef MDS_GI
  MDS_HIP(hipGetLastError());
  if (!hildreth && !q4 && E >= 1024) {                    // small batches have no tail to hide
    if (calls < 0 || calls >= 7) {
      k_cbf_order<<<1, kOrderThreads, 0, st>>>(E, cost_tab, order_tab, count_tab);
      MDS_HIP(hipGetLastError());
      calls = 0;
    } else {
      ++calls;
    }
  }
  return MDS_OK;
}

int mds_cbf_filter(mds_handle* h, const void* obs, const void* xdes, const void* unom, void* usafe, int32_t* status,
                   void* stream) {
  MDS_DEV(h);
  if (!h || !obs || !xdes || !unom || !usafe || !status) return fail(MDS_EINVAL, "mds_cbf_filter: null argument");
  if (!h->has_cbf) return fail(MDS_ESTATE, "mds_cbf_filter: call mds_cbf_configure first");
  return cbf_filter_range(h, obs, xdes, unom, usafe, status, stream, EnvRange{0, h->cfg.num_envs, 0});
}

int mds_cbf_last_iterations(mds_handle* h, int32_t* iters_dev, void* stream) {
  MDS_DEV(h);
  if (!h || !iters_dev) return fail(MDS_EINVAL, "mds_cbf_last_iterations: null argument");
  if (!h->has_cbf || !h->cbf_cost) return fail(MDS_ESTATE, "mds_cbf_last_iterations: call mds_cbf_configure first");
  MDS_HIP(hipMemcpyAsync(iters_dev, h->cbf_cost, sizeof(int) * (size_t)h->cfg.num_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MDS_OK;
}

int mds_default_dslpid_gains(mds_dslpid_gains* g) {
  if (!g) return fail(MDS_EINVAL, "mds_default_dslpid_gains");
  const double pf[3] = {.4, .4, 1.25}, ifo[3] = {.05, .05, .05}, df[3] = {.2, .2, .5};
  const double pt[3] = {70000., 70000., 60000.}, it[3] = {.0, .0, 500.}, dt[3] = {20000., 20000., 12000.};
  for (int k = 0; k < 3; ++k) {
    g->P_COEFF_FOR[k] = pf[k]; g->I_COEFF_FOR[k] = ifo[k]; g->D_COEFF_FOR[k] = df[k];
    g->P_COEFF_TOR[k] = pt[k]; g->I_COEFF_TOR[k] = it[k]; g->D_COEFF_TOR[k] = dt[k];
  }
  return MDS_OK;
}

int mds_set_dslpid_gains(mds_handle* h, const mds_dslpid_gains* g) {
  if (!h || !g) return fail(MDS_EINVAL, "mds_set_dslpid_gains: null argument");
  for (int k = 0; k < 3; ++k) {
    h->pid_d.Pf[k] = g->P_COEFF_FOR[k]; h->pid_d.If[k] = g->I_COEFF_FOR[k]; h->pid_d.Df[k] = g->D_COEFF_FOR[k];
    h->pid_d.Pt[k] = g->P_COEFF_TOR[k]; h->pid_d.It[k] = g->I_COEFF_TOR[k]; h->pid_d.Dt[k] = g->D_COEFF_TOR[k];
    h->pid_f.Pf[k] = (float)g->P_COEFF_FOR[k]; h->pid_f.If[k] = (float)g->I_COEFF_FOR[k]; h->pid_f.Df[k] = (float)g->D_COEFF_FOR[k];
    h->pid_f.Pt[k] = (float)g->P_COEFF_TOR[k]; h->pid_f.It[k] = (float)g->I_COEFF_TOR[k]; h->pid_f.Dt[k] = (float)g->D_COEFF_TOR[k];
  }
  return MDS_OK;
}

int mds_dslpid_reset(mds_handle* h, void* stream) {
  MDS_DEV(h);
  if (!h) return fail(MDS_EINVAL, "mds_dslpid_reset: null handle");
  MDS_HIP(hipMemsetAsync(h->pid, 0, 9 * h->ld * comp_size(h->cfg.dtype), (hipStream_t)stream));
  return MDS_OK;
}

// PID_GRID / PID_B0: the launch's batches (all of them, or one half of the shard in a two-chain rollout)
#define MDS_PID_LAUNCH(T, S, C, G, STEP, RK4, DRAG)                                                                           \
  k_dslpid<T, S, STEP, RK4, DRAG><<<PID_GRID, kBlock, 0, st>>>(C, G, h->n, h->ld, (T)(1.0 / h->cfg.ctrl_freq), (S*)h->state, \
                                                               (const T*)h->origin, (T*)rpm_track(h), (T*)h->pid,            \
                                                               (const S*)obs_in, (const S*)tpos, (const S*)trpy, (S*)obs,   \
                                                               (S*)act, PID_B0, (S*)h->state_lo)
#define MDS_PID_DTYPE(STEP, RK4, DRAG)                                                        \
  do {                                                                                        \
    if (h->cfg.dtype == MDS_F64) MDS_PID_LAUNCH(double, double, h->cd, h->pid_d, STEP, RK4, DRAG); \
    else if (is_f32(h)) MDS_PID_LAUNCH(float, float, h->cf, h->pid_f, STEP, RK4, DRAG); \
    else MDS_PID_LAUNCH(float, half_t, h->cf, h->pid_f, STEP, RK4, DRAG);                     \
  } while (0)

// one MultiDroneEnv.sim_step for the batches [batch0, batch0 + nb) (nb == 0: the whole shard)
static void launch_step_dslpid(mds_handle* h, const void* tpos, const void* trpy, void* obs, void* act, hipStream_t st, unsigned batch0 = 0,
                               unsigned nb = 0) {
  const dim3 PID_GRID = nb ? dim3(nb) : grid_for(h->n, kBlock);
  const int PID_B0 = (int)batch0;
  const void* obs_in = nullptr;
  const bool rk4 = h->cfg.integrator == MDS_INTEGRATOR_RK4, drag = has_drag(h);
  if (rk4 && drag) MDS_PID_DTYPE(true, true, true);
  else if (rk4) MDS_PID_DTYPE(true, true, false);
  else if (drag) MDS_PID_DTYPE(true, false, true);
  else MDS_PID_DTYPE(true, false, false);
}

// the same step under ground effect / downwash: the controller reads the handle's state and leaves the action in act_scratch (or the
// caller's buffer), then env.step runs it one substep per launch
static int step_env_dslpid(mds_handle* h, const void* tpos, const void* trpy, void* obs, void* act_out, hipStream_t st) {
  const dim3 PID_GRID = grid_for(h->n, kBlock);
  const int PID_B0 = 0;
  const void* obs_in = nullptr;
  void* act = act_out ? act_out : h->act_scratch;
  {
    void* obs = nullptr;
    MDS_PID_DTYPE(false, false, false);
  }
  return step_env_plain(h, act, obs, st, 0);
}

int mds_dslpid_compute(mds_handle* h, const void* obs_in, const void* tpos, const void* trpy, void* act, void* stream) {
  MDS_DEV(h);
  if (!h || !obs_in || !tpos || !trpy || !act) return fail(MDS_EINVAL, "mds_dslpid_compute: null argument");
  if (!aligned16(act)) return fail(MDS_EALIGN, "mds_dslpid_compute: rpm_dev");
  hipStream_t st = (hipStream_t)stream;
  void* obs = nullptr;
  const dim3 PID_GRID = grid_for(h->n, kBlock);
  const int PID_B0 = 0;
  MDS_PID_DTYPE(false, false, false);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_step_dslpid(mds_handle* h, const void* tpos, const void* trpy, void* obs, void* act, void* stream) {
  MDS_DEV(h);
  if (!h || !tpos || !trpy) return fail(MDS_EINVAL, "mds_step_dslpid: null argument");
  if (!aligned16(obs) || !aligned16(act)) return fail(MDS_EALIGN, "mds_step_dslpid: obs_dev/action_dev");
  if (h->envfx) return step_env_dslpid(h, tpos, trpy, obs, act, (hipStream_t)stream);
  launch_step_dslpid(h, tpos, trpy, obs, act, (hipStream_t)stream);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_rollout_dslpid(mds_handle* h, const void* tpos, const void* trpy, int n_target_sets, int first_step, int n_steps, void* obs,
                       int obs_every_step, void* stream) {
  MDS_DEV(h);
  if (!h || !tpos || !trpy || n_target_sets < 1 || first_step < 0 || n_steps < 0) return fail(MDS_EINVAL, "mds_rollout_dslpid: arguments");
  if (!aligned16(obs)) return fail(MDS_EALIGN, "mds_rollout_dslpid: obs_dev");
  hipStream_t st = (hipStream_t)stream;
  const size_t set_bytes = (size_t)h->n * 3 * elem_size(h->cfg.dtype);
  auto targets = [&](int k, const void** p, const void** r) {
    const size_t off = (size_t)(((long long)first_step + k) % n_target_sets) * set_bytes;
    *p = (const char*)tpos + off;
    *r = (const char*)trpy + off;
  };
  const void *p, *r;
  if (h->envfx) {
    h->last_rollout_streams = 1;
    for (int k = 0; k < n_steps; ++k) {
      targets(k, &p, &r);
      if (int rc = step_env_dslpid(h, p, r, (obs_every_step || k == n_steps - 1) ? obs : nullptr, nullptr, st)) return rc;
    }
    return MDS_OK;
  }
  const unsigned nbatch = (unsigned)((h->n + kBlock - 1) / kBlock);
  if (rollout_streams_policy(h, 0, n_steps) == 2 && nbatch >= 2) {   // the PID memory is per drone too: two independent chains
    hipStream_t sb = h->split_st;
    h->last_rollout_streams = 2;
    const unsigned half = nbatch / 2;
    auto body = [&]() -> int {
      for (int k = 0; k < n_steps; ++k) {
        void* o = (obs_every_step || k == n_steps - 1) ? obs : nullptr;
        targets(k, &p, &r);
        if (k == 0)
          if (int rc = split_fork(h, st)) return rc;
        launch_step_dslpid(h, p, r, o, nullptr, st, 0, half);
        launch_step_dslpid(h, p, r, o, nullptr, sb, half, nbatch - half);
      }
      MDS_HIP(hipGetLastError());
      return MDS_OK;
    };
    return split_join(h, st, body());
  }
  h->last_rollout_streams = 1;
  for (int k = 0; k < n_steps; ++k) {
    targets(k, &p, &r);
    launch_step_dslpid(h, p, r, (obs_every_step || k == n_steps - 1) ? obs : nullptr, nullptr, st);
  }
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}
#undef MDS_PID_DTYPE
#undef MDS_PID_LAUNCH

int mds_set_lqr_omega_gain(mds_handle* h, const double K[36]) {
  MDS_DEV(h);
  if (!h || !K) return fail(MDS_EINVAL, "mds_set_lqr_omega_gain: null argument");
  for (int r = 0; r < 4; ++r)
    for (int k = 0; k < 9; ++k) {
      h->lqr_d.k[r][k] = K[9 * r + k];
      h->lqr_f.k[r][k] = (float)K[9 * r + k];
    }
  h->has_lqr = true;
  return upload_gain(h, 1, &h->lqr_f, sizeof(h->lqr_f), &h->lqr_d, sizeof(h->lqr_d));
}

int mds_lqr_omega_compute(mds_handle* h, const void* obs, const void* des, void* u, void* stream) {
  MDS_DEV(h);
  if (!h || !obs || !des || !u) return fail(MDS_EINVAL, "mds_lqr_omega_compute: null argument");
  if (!h->has_lqr) return fail(MDS_ESTATE, "mds_lqr_omega_compute: call mds_set_lqr_omega_gain first");
  if (!aligned16(u)) return fail(MDS_EALIGN, "mds_lqr_omega_compute: u_dev");
  hipStream_t st = (hipStream_t)stream;
  if (h->cfg.dtype == MDS_F64)
    k_lqr_omega_compute<double, double><<<grid_for(h->n, 256), 256, 0, st>>>(h->cd, h->lqr_d, h->n, (const double*)obs, (const double*)des, (double*)u);
  else if (is_f32(h))
    k_lqr_omega_compute<float, float><<<grid_for(h->n, 256), 256, 0, st>>>(h->cf, h->lqr_f, h->n, (const float*)obs, (const float*)des, (float*)u);
  else
    k_lqr_omega_compute<float, half_t><<<grid_for(h->n, 256), 256, 0, st>>>(h->cf, h->lqr_f, h->n, (const half_t*)obs, (const half_t*)des, (half_t*)u);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_set_lqr_gain(mds_handle* h, const double K[48]) {
  MDS_DEV(h);
  if (!h || !K) return fail(MDS_EINVAL, "mds_set_lqr_gain: null argument");
  for (int r = 0; r < 4; ++r)
    for (int k = 0; k < 12; ++k) {
      h->lqr12_d.k[r][k] = K[12 * r + k];
      h->lqr12_f.k[r][k] = (float)K[12 * r + k];
    }
  h->has_lqr12 = true;
  return upload_gain(h, 0, &h->lqr12_f, sizeof(h->lqr12_f), &h->lqr12_d, sizeof(h->lqr12_d));
}

int mds_lqr_compute(mds_handle* h, const void* obs, const void* des, void* u, void* action, void* stream) {
  MDS_DEV(h);
  if (!h || !obs || !des || (!u && !action)) return fail(MDS_EINVAL, "mds_lqr_compute: null argument");
  if (!h->has_lqr12) return fail(MDS_ESTATE, "mds_lqr_compute: call mds_set_lqr_gain first");
  if (!aligned16(u) || !aligned16(action)) return fail(MDS_EALIGN, "mds_lqr_compute: u_dev/action_dev");
  hipStream_t st = (hipStream_t)stream;
  if (h->cfg.dtype == MDS_F64)
    k_lqr12_compute<double, double><<<grid_for(h->n, 256), 256, 0, st>>>(h->cd, h->lqr12_d, h->n, (const double*)obs, (const double*)des, (double*)u, (double*)action);
  else if (is_f32(h))
    k_lqr12_compute<float, float><<<grid_for(h->n, 256), 256, 0, st>>>(h->cf, h->lqr12_f, h->n, (const float*)obs, (const float*)des, (float*)u, (float*)action);
  else
    k_lqr12_compute<float, half_t><<<grid_for(h->n, 256), 256, 0, st>>>(h->cf, h->lqr12_f, h->n, (const half_t*)obs, (const half_t*)des, (half_t*)u, (half_t*)action);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_step_lqr(mds_handle* h, double t, void* obs, void* act, void* stream) {
  MDS_DEV(h);
  if (!h) return fail(MDS_EINVAL, "mds_step_lqr: null handle");
  if (!h->has_traj) return fail(MDS_ESTATE, "mds_step_lqr: call mds_set_lemniscate / mds_set_trajectory_segments first");
  if (!h->has_lqr12) return fail(MDS_ESTATE, "mds_step_lqr: call mds_set_lqr_gain first");
  if (!aligned16(obs) || !aligned16(act)) return fail(MDS_EALIGN, "mds_step_lqr: obs_dev/action_dev");
  if (h->envfx) return step_env_ctrl(h, 1, t, nullptr, 0.0, obs, act, (hipStream_t)stream);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid = grid_for(h->n, kBlock);
  const bool rk4 = h->cfg.integrator == MDS_INTEGRATOR_RK4, drag = has_drag(h);
#define MDS_LQR_T(T, S, C, K, RK4, DRAG)                                                                                          \
  k_step_lqr<T, S, RK4, DRAG><<<grid, kBlock, 0, st>>>(C, K, h->n, h->ld, t, h->traj_mode, (S*)h->state, (const T*)h->origin,     \
                                                       (const T*)h->lem, SegTable{h->segs, h->nseg_total}, h->tinfo, (T*)rpm_track(h), (S*)obs, (S*)act, \
                                                       (S*)h->state_lo)
#define MDS_LQR(RK4, DRAG)                                                                    \
  do {                                                                                        \
    if (h->cfg.dtype == MDS_F64) MDS_LQR_T(double, double, h->cd, h->lqr12_d, RK4, DRAG);     \
    else if (is_f32(h)) MDS_LQR_T(float, float, h->cf, h->lqr12_f, RK4, DRAG);  \
    else MDS_LQR_T(float, half_t, h->cf, h->lqr12_f, RK4, DRAG);                              \
  } while (0)
  if (rk4 && drag) MDS_LQR(true, true);
  else if (rk4) MDS_LQR(true, false);
  else if (drag) MDS_LQR(false, true);
  else MDS_LQR(false, false);
#undef MDS_LQR
#undef MDS_LQR_T
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_set_lqr_yank_omega_gain(mds_handle* h, const double K[40]) {
  MDS_DEV(h);
  if (!h || !K) return fail(MDS_EINVAL, "mds_set_lqr_yank_omega_gain: null argument");
  for (int r = 0; r < 4; ++r)
    for (int k = 0; k < 10; ++k) {
      h->lqr_yo_d.k[r][k] = K[10 * r + k];
      h->lqr_yo_f.k[r][k] = (float)K[10 * r + k];
    }
  h->has_lqr_yo = true;
  return upload_gain(h, 2, &h->lqr_yo_f, sizeof(h->lqr_yo_f), &h->lqr_yo_d, sizeof(h->lqr_yo_d));
}

int mds_lqr_yank_omega_compute(mds_handle* h, const void* obs, const void* des, void* u, void* stream) {
  MDS_DEV(h);
  if (!h || !obs || !des || !u) return fail(MDS_EINVAL, "mds_lqr_yank_omega_compute: null argument");
  if (!h->has_lqr_yo) return fail(MDS_ESTATE, "mds_lqr_yank_omega_compute: call mds_set_lqr_yank_omega_gain first");
  if (!aligned16(u)) return fail(MDS_EALIGN, "mds_lqr_yank_omega_compute: u_dev");
  hipStream_t st = (hipStream_t)stream;
  if (h->cfg.dtype == MDS_F64)
    k_lqr_yank_omega_compute<double, double><<<grid_for(h->n, 256), 256, 0, st>>>(h->cd, h->lqr_yo_d, h->n, (const double*)obs, (const double*)des, (double*)u);
  else if (is_f32(h))
    k_lqr_yank_omega_compute<float, float><<<grid_for(h->n, 256), 256, 0, st>>>(h->cf, h->lqr_yo_f, h->n, (const float*)obs, (const float*)des, (float*)u);
  else
    k_lqr_yank_omega_compute<float, half_t><<<grid_for(h->n, 256), 256, 0, st>>>(h->cf, h->lqr_yo_f, h->n, (const half_t*)obs, (const half_t*)des, (half_t*)u);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_cbf_set_nominal(mds_handle* h, int which) {
  if (!h || which < 0 || which > 2) return fail(MDS_EINVAL, "mds_cbf_set_nominal");
  if (which == 1 && !h->has_lqr) return fail(MDS_ESTATE, "mds_cbf_set_nominal: call mds_set_lqr_omega_gain first");
  if (which == 2 && !h->has_lqr_yo) return fail(MDS_ESTATE, "mds_cbf_set_nominal: call mds_set_lqr_yank_omega_gain first");
  h->cbf_nominal = which;
  return MDS_OK;
}

int mds_cbf_set_step_kernel(mds_handle* h, int one_launch) {
  if (!h || one_launch < 0 || one_launch > 2) return fail(MDS_EINVAL, "mds_cbf_set_step_kernel");
  h->cbf_fused = one_launch == 1;
  h->cbf_step_persistent = one_launch == 2;
  for (int k = 0; k < 3; ++k) h->next_nom_ok[k] = false;
  return MDS_OK;
}

int mds_cbf_last_step_kernel(const mds_handle* h) {
  if (!h) return fail(MDS_EINVAL, "mds_cbf_last_step_kernel: null handle");
  return h->cbf_last_step_kernel;
}

int mds_lowlevel_reset(mds_handle* h, void* stream) {
  MDS_DEV(h);
  if (!h) return fail(MDS_EINVAL, "mds_lowlevel_reset: null handle");
  MDS_HIP(hipMemsetAsync(h->ll, 0, 6 * h->ld * comp_size(h->cfg.dtype), (hipStream_t)stream));
  return MDS_OK;
}

static int launch_thrust_omega(mds_handle* h, const void* u, const void* src, int rates_given, int yank, void* rpm, hipStream_t st) {
  MDS_DISPATCH(h, (k_thrust_omega<T, S><<<grid_for(h->n, 256), 256, 0, st>>>(C, h->n, h->ld, (T)(1.0 / h->cfg.ctrl_freq), rates_given, yank,
                                                                             (T*)h->ll, (const S*)u, (const S*)src, (S*)rpm)));
  return MDS_OK;
}

int mds_yank_omega_compute(mds_handle* h, const void* u, const void* obs, void* rpm, void* stream) {
  MDS_DEV(h);
  if (!h || !u || !obs || !rpm) return fail(MDS_EINVAL, "mds_yank_omega_compute: null argument");
  if (!aligned16(u) || !aligned16(rpm)) return fail(MDS_EALIGN, "mds_yank_omega_compute");
  launch_thrust_omega(h, u, obs, 0, 1, rpm, (hipStream_t)stream);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_thrust_omega_compute(mds_handle* h, const void* u, const void* obs, void* rpm, void* stream) {
  MDS_DEV(h);
  if (!h || !u || !obs || !rpm) return fail(MDS_EINVAL, "mds_thrust_omega_compute: null argument");
  if (!aligned16(u) || !aligned16(rpm)) return fail(MDS_EALIGN, "mds_thrust_omega_compute");
  launch_thrust_omega(h, u, obs, 0, 0, rpm, (hipStream_t)stream);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

int mds_thrust_omega_from_rates(mds_handle* h, const void* u, const void* rates, void* rpm, void* stream) {
  MDS_DEV(h);
  if (!h || !u || !rates || !rpm) return fail(MDS_EINVAL, "mds_thrust_omega_from_rates: null argument");
  if (!aligned16(u) || !aligned16(rpm)) return fail(MDS_EALIGN, "mds_thrust_omega_from_rates");
  launch_thrust_omega(h, u, rates, 1, 0, rpm, (hipStream_t)stream);
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

// nominal controller -> [ECBF QP] -> low level -> env.step.  with_filter = false: the plain loops of
// simulations/EnvGeometricOmega.py / EnvGeometricYankOmega.py (ctrl[j].compute(obs[j]) = LQR + low level, :314 / :319).
// have_nominal: the previous step's low-level launch has already left this step's u_hat / xdes in the scratch (want_next of that call);
// want_next: this step's low-level launch also computes the nominal input of the step at t_next (C rollout loops; nominal 0 / 1 only).
int step_nominal_lowlevel(mds_handle* h, double t, void* obs, int32_t* status, void* action, void* stream, bool with_filter,
                          const char* who, const EnvRange* rgp, bool have_nominal, bool want_next, double t_next) {
  EnvRange rg = rgp ? *rgp : EnvRange{0, -1, 0};
  if (!h->has_traj || h->traj_mode != 1) return fail(MDS_ESTATE, "mds_step_cbf_geometric / mds_step_nominal: call mds_set_lemniscate first");
  if (!aligned16(obs) || !aligned16(action)) return fail(MDS_EALIGN, "mds_step_cbf_geometric / mds_step_nominal: obs_dev/action_dev");
  const bool yank = h->cbf_nominal == 2;
  (void)who;
  hipStream_t st = (hipStream_t)stream;
  const size_t es = elem_size(h->cfg.dtype);
  // One-launch form (k_cbf_step): order 2, whole envs per wave (D | 64, D <= 16), explicit Euler without drag, nominal 0 / 1, no
  // action output.  Everything else takes the three launches below.
  {
    const int D = h->cfg.num_drones;
    const int m2 = D * (D - 1) / 2 + D * h->cbf.n_obs + 2 * D;
    if (rg.ne < 0) rg.ne = h->cfg.num_envs;
    const size_t j0 = (size_t)rg.e0 * D, j1 = j0 + (size_t)rg.ne * D;
    if (with_filter && h->cbf_fused && !action && h->cbf.order == 2 && !h->cbf_hildreth && D <= 16 && 64 % D == 0 && m2 <= 512 && !h->envfx &&
        h->cfg.integrator == MDS_INTEGRATOR_EULER && !has_drag(h) && h->cbf_nominal <= 1 && j0 % 64 == 0 && h->cfg.dtype != MDS_F16) {
      const dim3 grid64((unsigned)((j1 - j0 + 63) / 64));
      const int batch0 = (int)(j0 / 64), max_iter = h->cbf.max_iter > 0 ? h->cbf.max_iter : 64 * m2;
      h->cbf_last_step_kernel = 1;
      const void* gain = h->cbf_nominal == 1 ? h->gain_dev[1] : nullptr;
      void* rpm = rpm_track(h);
#define MDS_CS(T, CC, CP, RR, NOM, COMP, TOL)                                                                                                \
  k_cbf_step<T, RR, NOM, COMP><<<grid64, 64, 0, st>>>(CC, CP, gain, (int)j1, h->ld, h->cfg.num_envs, t, (T)(1.0 / h->cfg.ctrl_freq), (T*)h->state, \
                                                      (T*)h->state_lo, (const T*)h->lem, (T*)rpm, (T*)h->ll, h->pair_ij, (const T*)h->obstacles,   \
                                                      (T*)obs, (int*)status, h->cbf_cost, max_iter, (T)((TOL) * (TOL)), batch0)
#define MDS_CS_N(T, CC, CP, RR, COMP, TOL)                         \
  do {                                                             \
    if (h->cbf_nominal == 1) MDS_CS(T, CC, CP, RR, 1, COMP, TOL);  \
    else MDS_CS(T, CC, CP, RR, 0, COMP, TOL);                      \
  } while (0)
#define MDS_CS_R(T, CC, CP, COMP, TOL)                 \
  do {                                                 \
    if (m2 <= 256) MDS_CS_N(T, CC, CP, 4, COMP, TOL);  \
    else MDS_CS_N(T, CC, CP, 8, COMP, TOL);            \
  } while (0)
      if (h->cfg.dtype == MDS_F64) MDS_CS_R(double, h->cd, h->cbf_d, false, (h->cbf.tol > 0 ? h->cbf.tol : 1e-12));
      else if (is_comp(h)) MDS_CS_R(float, h->cf, h->cbf_f, true, (h->cbf.tol > 0 ? h->cbf.tol : 1e-6));
      else MDS_CS_R(float, h->cf, h->cbf_f, false, (h->cbf.tol > 0 ? h->cbf.tol : 1e-6));
#undef MDS_CS_R
#undef MDS_CS_N
#undef MDS_CS
      MDS_HIP(hipGetLastError());
      return MDS_OK;
    }
  }
  if (with_filter) h->cbf_last_step_kernel = 0;
  if (!h->cbf_unom) {
    MDS_HIP(hipMalloc(&h->cbf_unom, (size_t)h->n * 4 * es));
    MDS_HIP(hipMalloc(&h->cbf_xdes, (size_t)h->n * 10 * es));
    MDS_HIP(hipMalloc(&h->cbf_usafe, (size_t)h->n * 4 * es));
  }
  // drones of the env range: whole 256-drone batches (the caller of a partial range guarantees the alignment)
  if (rg.ne < 0) rg.ne = h->cfg.num_envs;
  const size_t i0 = (size_t)rg.e0 * h->cfg.num_drones, i1 = i0 + (size_t)rg.ne * h->cfg.num_drones;
  const int batch0 = (int)(i0 / kBlock), n_end = (int)i1;
  const dim3 grid = grid_for(i1 - i0, kBlock);
  // the hover force is subtracted from the nominal input only on the way into the filter (CBFTest.py:339, CBFTestOrd3.py:341)
  const double hover_sub = with_filter ? h->cfg.M * h->cfg.G : 0.0;
  const bool skip_nominal = have_nominal && h->next_nom_ok[rg.slot] && h->next_nom_t[rg.slot] == t;
  h->next_nom_ok[rg.slot] = false;
  const bool rk4 = h->cfg.integrator == MDS_INTEGRATOR_RK4, drag = has_drag(h);
  const void* nom_K = h->cbf_nominal == 1 ? h->gain_dev[1] : nullptr;
  // k_lowlevel_step: (a) `only` mode = the nominal controller of this step (GeometricControl / LQR-omega: the ONE compiled copy of
  // cbf_nominal_of, see there); (b) low level + physics + observation, optionally followed by the nominal input of the next step
#define MDS_NX(T, S, ON, TT, ONLY) NextNominal<T, S>{(ON) ? (const T*)h->lem : nullptr, nom_K, (ON) ? (S*)h->cbf_unom : nullptr, (S*)h->cbf_xdes, TT, \
                                                     (T)hover_sub, h->cbf_nominal, ONLY}
#define MDS_LL(RK4, DRAG, YANK, ULL, OFFS, ON, TT, ONLY)                                                                         \
  do {                                                                                                                           \
    if (is_comp(h))                                                                                                              \
      k_lowlevel_step<float, float, RK4, DRAG, YANK, true><<<grid, kBlock, 0, st>>>(h->cf, n_end, h->ld, (float)(1.0 / h->cfg.ctrl_freq), \
                                                                                    (float)(OFFS), (float*)h->state, (const float*)h->origin, \
                                                                                    (float*)rpm_track(h), (float*)h->ll, (const float*)(ULL), \
                                                                                    (float*)obs, (float*)action, batch0, (float*)h->state_lo, \
                                                                                    MDS_NX(float, float, ON, TT, ONLY));         \
    else                                                                                                                         \
      MDS_DISPATCH(h, (k_lowlevel_step<T, S, RK4, DRAG, YANK><<<grid, kBlock, 0, st>>>(C, n_end, h->ld, (T)(1.0 / h->cfg.ctrl_freq),  \
                                                                                     (T)(OFFS), (S*)h->state, (const T*)h->origin, \
                                                                                     (T*)rpm_track(h), (T*)h->ll, (const S*)(ULL), \
                                                                                     (S*)obs, (S*)action, batch0, (S*)nullptr,    \
                                                                                     MDS_NX(T, S, ON, TT, ONLY))));               \
  } while (0)
#define MDS_LL_Y(YANK, ULL, OFFS, ON, TT, ONLY)                         \
  do {                                                                  \
    if (rk4 && drag) MDS_LL(true, true, YANK, ULL, OFFS, ON, TT, ONLY); \
    else if (rk4) MDS_LL(true, false, YANK, ULL, OFFS, ON, TT, ONLY);   \
    else if (drag) MDS_LL(false, true, YANK, ULL, OFFS, ON, TT, ONLY);  \
    else MDS_LL(false, false, YANK, ULL, OFFS, ON, TT, ONLY);           \
  } while (0)
  if (skip_nominal) {
    // u_hat / xdes of this step came out of the previous step's low-level launch
  } else if (h->cbf_nominal == 2) {
    if (h->cfg.dtype == MDS_F64)
      k_cbf_nominal_lqr_yo<double, double><<<grid, kBlock, 0, st>>>(h->cd, h->lqr_yo_d, n_end, h->ld, t, hover_sub, (const double*)h->state,
                                                                    (const double*)h->lem, (const double*)obs, (double*)h->cbf_unom,
                                                                    (double*)h->cbf_xdes, batch0);
    else
      k_cbf_nominal_lqr_yo<float, float><<<grid, kBlock, 0, st>>>(h->cf, h->lqr_yo_f, n_end, h->ld, t, (float)hover_sub, (const float*)h->state,
                                                                  (const float*)h->lem, (const float*)obs, (float*)h->cbf_unom,
                                                                  (float*)h->cbf_xdes, batch0);
  } else {
    MDS_LL_Y(false, nullptr, 0.0, true, t, 1);
  }
  MDS_HIP(hipGetLastError());
  const void* u_ll = h->cbf_unom;
  if (with_filter) {
    int rc = cbf_filter_range(h, obs, h->cbf_xdes, h->cbf_unom, h->cbf_usafe, status, stream, rg);
    if (rc != MDS_OK) return rc;
    u_ll = h->cbf_usafe;
  }
  // order 2: u_safe[0] += M G (CBFTest.py:346); order 3: the yank goes to the low level as it is (CBFTestOrd3.py:350)
  const double ll_offset = (with_filter && !yank) ? h->cfg.M * h->cfg.G : 0.0;
  if (h->envfx)            // ground effect / downwash: low level + first substep, then the remaining substeps (whole batch, one stream)
    return step_env_ctrl(h, yank ? 3 : 2, t, u_ll, ll_offset, obs, action, st);
  const bool next_on = want_next && !yank && h->cbf_nominal <= 1 && !action;
  h->next_nom_ok[rg.slot] = next_on;
  h->next_nom_t[rg.slot] = t_next;
  if (yank) MDS_LL_Y(true, u_ll, ll_offset, false, t_next, 0);
  else MDS_LL_Y(false, u_ll, ll_offset, next_on, t_next, 0);
#undef MDS_LL_Y
#undef MDS_LL
#undef MDS_NX
  MDS_HIP(hipGetLastError());
  return MDS_OK;
}

// what k_cbf_rollout covers (mds_rollout_cbf_geometric_fused; mds_cbf_set_step_kernel(h, 2))
static bool roll_fused_applies(const mds_handle* h) {
  const int D = h->cfg.num_drones;
  const int m2 = D * (D - 1) / 2 + (MDS_ROLL_BOUNDS ? 0 : D * h->cbf.n_obs) + 2 * D;      // (MDS_ROLL_BOUNDS: obstacle rows are per-drone bounds)
  return h->has_cbf && h->cbf.order == 2 && !h->cbf_hildreth && D >= 1 && D <= 16 && (MDS_ROLL_BOUNDS || 64 % D == 0) && m2 <= 256 && !h->envfx &&
         h->cfg.integrator == MDS_INTEGRATOR_EULER && !has_drag(h) && h->cbf_nominal <= 1 && h->cfg.dtype != MDS_F16 &&
         h->cfg.pyb_freq == h->cfg.ctrl_freq && h->n <= (1 << 27);
}

int mds_step_cbf_geometric(mds_handle* h, double t, void* obs, int32_t* status, void* action, void* stream) {
  MDS_DEV(h);
  if (!h || !obs || !status) return fail(MDS_EINVAL, "mds_step_cbf_geometric: null argument");
  if (!h->has_cbf) return fail(MDS_ESTATE, "mds_step_cbf_geometric: call mds_cbf_configure first");
  const bool ord3 = h->cbf.order == 3;
  // the order-3 rows act on (yank, w): only the yank-omega LQR produces that input (simulations/CBFTestOrd3.py:294-297;
  // GeometricControl.compute has no skip_low_level there), and it is meaningless for the order-2 rows
  if (ord3 != (h->cbf_nominal == 2))
    return fail(MDS_EUNSUPPORTED, "mds_step_cbf_geometric: order 3 needs (and order 2 excludes) the lqr-yank-omega nominal, mds_cbf_set_nominal(h, 2)");
  if (h->cfg.dtype == MDS_F16) return fail(MDS_EUNSUPPORTED, "mds_step_cbf_geometric: fp16 storage");
  if (h->cbf_step_persistent && !action && roll_fused_applies(h)) {   // mds_cbf_set_step_kernel(h, 2): one launch of k_cbf_rollout, one step
    for (int k = 0; k < 3; ++k) h->next_nom_ok[k] = false;
    return mds_rollout_cbf_geometric_fused(h, t, 1, 1, nullptr, 0, 0, obs, status, nullptr, stream);
  }
  return step_nominal_lowlevel(h, t, obs, status, action, stream, true, "mds_step_cbf_geometric");
}

int mds_rollout_cbf_geometric(mds_handle* h, double t0, int n_steps, void* obs, int32_t* status, void* stream) {
  MDS_DEV(h);
  if (!h || !obs || !status || n_steps < 0) return fail(MDS_EINVAL, "mds_rollout_cbf_geometric: null argument");
  if (h->cbf_step_persistent && h->has_cbf && h->cbf.order == 2 && h->cbf_nominal <= 1 && roll_fused_applies(h) && h->has_traj && h->traj_mode == 1) {
    for (int k = 0; k < 3; ++k) h->next_nom_ok[k] = false;
    h->last_rollout_streams = 1;
    return mds_rollout_cbf_geometric_fused(h, t0, n_steps, 25, nullptr, 0, 0, obs, status, nullptr, stream);
  }
  const double dt = 1.0 / h->cfg.ctrl_freq;
  // The env halves are independent step chains (a barrier row couples drones of one env only).  On two streams one half's
  // QP kernel (latency / ALU bound) runs beside the other half's nominal and low-level kernels (memory bound).  The split
  // must fall on a 256-drone batch boundary; each half keeps its own cost classes for the longest-first dispatch.
  const int E = h->cfg.num_envs, D = h->cfg.num_drones;
  int e_mid = 0;
  for (int e = E / 2; e > 0 && e >= E / 2 - 256; --e)
    if (((size_t)e * D) % kBlock == 0) {
      e_mid = e;
      break;
    }
  const int streams = rollout_streams_policy(h, 1, n_steps);
  if (streams == 2 && e_mid > 0) {
    if (int rc = mds_step_cbf_geometric(h, t0, obs, status, nullptr, stream)) return rc;   // validation, scratch; step 1 on the caller's stream
    hipStream_t st = (hipStream_t)stream;
    if (int rc = split_fork(h, st)) return rc;
    h->last_rollout_streams = 2;
    const EnvRange half[2] = {EnvRange{0, e_mid, 1}, EnvRange{e_mid, E - e_mid, 2}};
    auto body = [&]() -> int {
      double t = t0 + dt;
      for (int k = 1; k < n_steps; ++k) {
        // each chain's low-level launch of step k also computes the nominal input of its step k + 1 (next_nominal), so that
        // from its second step on a chain is two launches per step: QP, low level
        for (int s = 0; s < 2; ++s)
          if (int rc = step_nominal_lowlevel(h, t, obs, status, nullptr, s == 0 ? st : h->split_st, true, "mds_rollout_cbf_geometric", &half[s],
                                             k > 1, h->cbf_chain_nominal && k < n_steps - 1, t + dt))
            return rc;
        t += dt;
      }
      return MDS_OK;
    };
    return split_join(h, st, body());
  }
  h->last_rollout_streams = 1;
  for (int k = 0; k < n_steps; ++k) {
    if (k == 0) {
      if (int rc = mds_step_cbf_geometric(h, t0, obs, status, nullptr, stream)) return rc;   // validation; plain first step
    } else if (int rc = step_nominal_lowlevel(h, t0, obs, status, nullptr, stream, true, "mds_rollout_cbf_geometric", nullptr, k > 1,
                                              h->cbf_chain_nominal && k < n_steps - 1, t0 + dt)) {
      return rc;
    }
    t0 += dt;
  }
  return MDS_OK;
}

int mds_rollout_cbf_geometric_fused(mds_handle* h, double t0, int n_steps, int steps_per_launch, void* obs_log, int log_slots, int first_slot,
                                    void* obs, int32_t* status, int32_t* status_log, void* stream) {
  MDS_DEV(h);
  if (!h || !obs || !status || n_steps < 0 || steps_per_launch < 1 || log_slots < 0 || first_slot < 0 || (obs_log && first_slot >= log_slots) ||
      (obs_log && log_slots < 1))
    return fail(MDS_EINVAL, "mds_rollout_cbf_geometric_fused: null or out-of-range argument");
  if (!h->has_cbf) return fail(MDS_ESTATE, "mds_rollout_cbf_geometric_fused: call mds_cbf_configure first");
  if (!h->has_traj || h->traj_mode != 1) return fail(MDS_ESTATE, "mds_rollout_cbf_geometric_fused: call mds_set_lemniscate first");
  if (!aligned16(obs) || !aligned16(obs_log)) return fail(MDS_EALIGN, "mds_rollout_cbf_geometric_fused: obs buffers");
  const int D = h->cfg.num_drones;
  const int m2 = D * (D - 1) / 2 + D * h->cbf.n_obs + 2 * D;
  if (h->cbf.order == 3) {
    // the order-3 loop (simulations/CBFTestOrd3.py:306-352): k_cbf_rollout_o3, one wavefront per env, steps_per_launch steps per launch
    if (!(h->cbf_nominal == 2 && h->has_lqr_yo && D <= 16 && !h->envfx && !h->cbf_hildreth && h->cfg.integrator == MDS_INTEGRATOR_EULER && !has_drag(h) &&
          (h->cfg.dtype == MDS_F32 || h->cfg.dtype == MDS_F64) && h->cfg.pyb_freq == h->cfg.ctrl_freq))
      return fail(MDS_EUNSUPPORTED, "mds_rollout_cbf_geometric_fused (order 3): lqr-yank-omega nominal, D <= 16, explicit Euler at pyb_freq == ctrl_freq "
                                    "without drag / ground effect / downwash, f32 / f64");
    if (n_steps == 0) return MDS_OK;
    hipStream_t st3 = (hipStream_t)stream;
    const int m3 = D * (D - 1) / 2 + D * h->cbf.n_obs + 6 * D, n3 = 3 * D;
    const int max_iter3 = h->cbf.max_iter > 0 ? h->cbf.max_iter : 64 * m3;
    const double dt3 = 1.0 / h->cfg.ctrl_freq, hover_sub = h->cfg.M * h->cfg.G;      // the hover force is subtracted from the yank on the way into the filter (:341)
    void* rpm3 = rpm_track(h);
    h->cbf_last_step_kernel = 2;
    int slot3 = first_slot;
    double t3 = t0;
    for (int k0 = 0; k0 < n_steps; k0 += steps_per_launch) {
      const int ks = n_steps - k0 < steps_per_launch ? n_steps - k0 : steps_per_launch;
      int32_t* slog = status_log ? status_log + (size_t)k0 * h->cfg.num_envs : nullptr;
#define MDS_O3(T, CC, CP, KK, RR, NM, TOL)                                                                                                   \
  k_cbf_rollout_o3<T, RR, NM><<<dim3((unsigned)h->cfg.num_envs), 64, 0, st3>>>(CC, CP, KK, h->cfg.num_envs, h->ld, t3, dt3, ks, (T*)h->state,   \
                                                                             (const T*)h->lem, (T*)rpm3, (T*)h->ll, h->pair_ij,              \
                                                                             (const T*)h->obstacles, (T*)obs, (T*)obs_log, slot3,            \
                                                                             log_slots > 0 ? log_slots : 1, (int*)status, (int*)slog,        \
                                                                             h->cbf_cost, max_iter3, (T)((TOL) * (TOL)), (T)hover_sub)
#define MDS_O3_S(T, CC, CP, KK, TOL)                 \
  do {                                               \
    if (n3 <= 24 && m3 <= 256) MDS_O3(T, CC, CP, KK, 4, 24, TOL); \
    else MDS_O3(T, CC, CP, KK, 8, 48, TOL);          \
  } while (0)
      if (h->cfg.dtype == MDS_F64) MDS_O3_S(double, h->cd, h->cbf_d, h->lqr_yo_d, (h->cbf.tol > 0 ? h->cbf.tol : 1e-12));
      else MDS_O3_S(float, h->cf, h->cbf_f, h->lqr_yo_f, (h->cbf.tol > 0 ? h->cbf.tol : 1e-6));
#undef MDS_O3_S
#undef MDS_O3
      MDS_HIP(hipGetLastError());
      for (int j = 0; j < ks; ++j) t3 += dt3;
      if (obs_log) slot3 = (slot3 + ks) % log_slots;
    }
    return MDS_OK;
  }
  // what the persistent kernel covers (everything else: mds_rollout_cbf_geometric, one or two launches per step)
  if (!roll_fused_applies(h))
    return fail(MDS_EUNSUPPORTED, "mds_rollout_cbf_geometric_fused: order-2 CBF, D <= 16, explicit Euler at "
                                  "pyb_freq == ctrl_freq without drag / ground effect / downwash, geometric or LQR-omega nominal, f32 / f32c / f64, "
                                  "at most 2^27 drones (32-bit byte offsets into the per-drone planes)");
  if (n_steps == 0) return MDS_OK;
  hipStream_t st = (hipStream_t)stream;
  // wavefronts per workgroup: 8 in fp32 (two workgroups per CU at <= 128 VGPRs); 4 in double (one wavefront per SIMD: the double
  // instantiation needs more than the 256 registers two wavefronts per SIMD would leave it)
  constexpr int NWF = MDS_CBF_ROLL_NW, NWD = 4;
  const int nw = h->cfg.dtype == MDS_F64 ? NWD : NWF;
  // a workgroup owns 64 nw / Dp whole envs, Dp = the env width padded to 4, 8 or 16 lanes (any D <= 16)
  const int Dp = D <= 4 ? 4 : (D <= 8 ? 8 : 16), envs_per_wg = 64 * nw / Dp;
  const dim3 grid((unsigned)((h->cfg.num_envs + envs_per_wg - 1) / envs_per_wg));
  const int max_iter = h->cbf.max_iter > 0 ? h->cbf.max_iter : 64 * m2;
  const void* gain = h->cbf_nominal == 1 ? h->gain_dev[1] : nullptr;
  const double dt = 1.0 / h->cfg.ctrl_freq;
  void* rpm = rpm_track(h);
  const size_t es = elem_size(h->cfg.dtype);
  h->cbf_last_step_kernel = 2;
  // tuning aid: MDS_TUNE_ROLL_STAMPS=1 -> per-wave shader-clock ticks by part of the step, summed over this call, printed to stderr
  // (synchronises the stream: never set it in production)
  unsigned long long* stamps_dev = nullptr;
  const size_t n_stamp = (size_t)grid.x * nw * 10;
  if (const char* e = getenv("MDS_TUNE_ROLL_STAMPS"))
    if (e[0] == '1') {
      MDS_HIP(hipMalloc((void**)&stamps_dev, n_stamp * sizeof(unsigned long long) * ((n_steps + steps_per_launch - 1) / steps_per_launch)));
    }
  unsigned long long* const stamps_base = stamps_dev;
  // tuning aid: MDS_TUNE_ROLL_EXTRA_LDS=<bytes> of unused dynamic LDS per workgroup (fewer workgroups per CU: occupancy experiments)
  size_t extra_lds = 0;
  if (const char* e = getenv("MDS_TUNE_ROLL_EXTRA_LDS")) {
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || *end != '\0' || v < 0 || v > 65536) return fail(MDS_EINVAL, "MDS_TUNE_ROLL_EXTRA_LDS: expected 0..65536 bytes");
    extra_lds = (size_t)v;
  }
  int slot = first_slot;
  double t = t0;
  for (int k0 = 0; k0 < n_steps; k0 += steps_per_launch) {
    const int ks = n_steps - k0 < steps_per_launch ? n_steps - k0 : steps_per_launch;
    int32_t* slog = status_log ? status_log + (size_t)k0 * h->cfg.num_envs : nullptr;
#define MDS_CR(T, CC, CP, NOM, COMP, TOL)                                                                                                        \
  do {                                                                                                                                           \
    RollArgs<T> ra;                                                                                                                              \
    ra.p.c = CC; ra.p.P = CP; ra.Kp = gain; ra.n = (int)h->n; ra.ld = h->ld; ra.E = h->cfg.num_envs;             \
    ra.t = t; ra.ctrl_dt = dt; ra.n_steps = ks; ra.state = (T*)h->state; ra.state_lo = (T*)h->state_lo; ra.lem = (const T*)h->lem;                \
    ra.last_rpm = (T*)rpm; ra.ll = (T*)h->ll; ra.pair_ij = h->pair_ij; ra.obstacles = (const T*)h->obstacles; ra.obs_log = (T*)obs_log;           \
    ra.slot = slot; ra.n_slots = log_slots > 0 ? log_slots : 1; ra.obs_last = (T*)obs; ra.status = (int*)status; ra.status_log = (int*)slog;      \
    ra.cost_io = h->cbf_cost; ra.max_iter = max_iter; ra.tol2 = (T)((TOL) * (TOL)); ra.tol = (T)(TOL); ra.stamps = stamps_dev;                   \
    if (D == Dp) k_cbf_rollout<T, NOM, COMP, (sizeof(T) == 8 ? NWD : NWF), false><<<grid, 64 * nw, extra_lds, st>>>(ra);                          \
    else k_cbf_rollout<T, NOM, COMP, (sizeof(T) == 8 ? NWD : NWF), true><<<grid, 64 * nw, extra_lds, st>>>(ra);                                   \
  } while (0)
#define MDS_CR_N(T, CC, CP, COMP, TOL)                        \
  do {                                                        \
    if (h->cbf_nominal == 1) MDS_CR(T, CC, CP, 1, COMP, TOL); \
    else MDS_CR(T, CC, CP, 0, COMP, TOL);                     \
  } while (0)
    if (h->cfg.dtype == MDS_F64) MDS_CR_N(double, h->cd, h->cbf_d, false, (h->cbf.tol > 0 ? h->cbf.tol : 1e-12));
    else if (is_comp(h)) MDS_CR_N(float, h->cf, h->cbf_f, true, (h->cbf.tol > 0 ? h->cbf.tol : 1e-6));
    else MDS_CR_N(float, h->cf, h->cbf_f, false, (h->cbf.tol > 0 ? h->cbf.tol : 1e-6));
#undef MDS_CR_N
#undef MDS_CR
    MDS_HIP(hipGetLastError());
    // t advances on the host exactly as inside the kernel (one += per step), so that consecutive launches continue the same sequence
    for (int j = 0; j < ks; ++j) t += dt;
    if (obs_log) slot = (slot + ks) % log_slots;
    if (stamps_dev) stamps_dev += n_stamp;
  }
  if (stamps_base) {
    const int launches = (n_steps + steps_per_launch - 1) / steps_per_launch;
    std::vector<unsigned long long> hs(n_stamp * launches);
    MDS_HIP(hipStreamSynchronize(st));
    MDS_HIP(hipMemcpy(hs.data(), stamps_base, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(stamps_base);
    static const char* part[10] = {"wait before B", "stage B", "wait after B", "stage C", "obs rows", "stage A", " B: ticket", " B: rows", " B: bookkeeping", " B: scan+solve"};
    const size_t waves = (size_t)grid.x * nw;
    double tot_mean = 0;
    for (int p = 0; p < 10; ++p) {
      double sum = 0, mx = 0;
      for (size_t w = 0; w < waves; ++w) {
        double v = 0;
        for (int l = 0; l < launches; ++l) v += (double)hs[(size_t)l * n_stamp + w * 10 + p];
        sum += v;
        if (v > mx) mx = v;
      }
      if (p < 6) tot_mean += sum / waves / n_steps;
      fprintf(stderr, "[mds roll stamps] %-14s mean %9.0f  max %9.0f ticks per wave and step\n", part[p], sum / waves / n_steps, mx / n_steps);
    }
    fprintf(stderr, "[mds roll stamps] total mean %.0f ticks per wave and step, %d steps, %zu waves\n", tot_mean, n_steps, waves);
  }
  if (obs_log) {       // the kernel materialises each step's observation once, in its log slot: obs_dev gets a copy of the last one
    const size_t row = (size_t)h->n * kObsDim * es;
    const int last = (first_slot + n_steps - 1) % log_slots;
    MDS_HIP(hipMemcpyAsync(obs, (const char*)obs_log + (size_t)last * row, row, hipMemcpyDeviceToDevice, st));
  }
  return MDS_OK;
}

int mds_step_nominal(mds_handle* h, double t, void* obs, void* action, void* stream) {
  MDS_DEV(h);
  if (!h || !obs) return fail(MDS_EINVAL, "mds_step_nominal: null argument");
  if (h->cbf_nominal != 1 && h->cbf_nominal != 2)
    return fail(MDS_ESTATE, "mds_step_nominal: select the LQR-omega (1) or LQR-yank-omega (2) controller with mds_cbf_set_nominal first");
  if (h->cfg.dtype == MDS_F16) return fail(MDS_EUNSUPPORTED, "mds_step_nominal: fp16 storage");
  // no action wanted: the one-step instance of the whole-rollout kernel does LQR + low level + physics in one launch
  // (212 B per drone-step instead of 392 B over two launches: 34 / 48 us -> 18 us at C3)
  if (!action && h->has_traj && h->traj_mode == 1 && !h->envfx)
    return rollout_fused(h, t, 1, nullptr, obs, stream, h->cbf_nominal == 2 ? 3 : 2, "mds_step_nominal");
  return step_nominal_lowlevel(h, t, obs, nullptr, action, stream, false, "mds_step_nominal");
}

#endif  // MDS_PART & 2

}  // extern "C"
