/* libmds -- C-ABI of the MI355X-native batched multi-drone step.
 *
 * This is the drop-in boundary for the ONE hot path of JasonTStanley/MultiDroneSim: the
 * per-drone physics step the reference delegates to gym-pybullet-drones'
 * BaseAviary.step/_dynamics (PyBullet, CPU) plus the reference's per-drone trajectory
 * sampling, geometric controller, RPM mixer and ECBF safety filter.  The reference has no
 * FFI layer of its own (pure Python); each entry point below names the reference
 * interface it replaces (paths relative to the reference checkout; [UPSTREAM] =
 * gym-pybullet-drones, not in the tree -- see SURVEY.md 3.4).
 *
 * Conventions
 *   - n = num_envs * num_drones "drones"; drone d of env e has flat index e*num_drones + d.
 *   - "dev" pointers are device (HBM) pointers owned by the caller (e.g. PyTorch-ROCm
 *     tensors); their element type is the handle's storage dtype (mds_dtype): float for
 *     MDS_F32 and MDS_F32C, double for MDS_F64, IEEE half for MDS_F16 (fp16 storage, fp32 arithmetic).
 *   - "host" pointers are host double arrays (set-up / inspection only, never hot path).
 *   - `stream` is a hipStream_t (NULL = default stream).  Hot-path calls (mds_step*, mds_rollout*, the operators) only
 *     enqueue work on it and return; they never synchronise or copy, and they allocate only once: the first
 *     mds_step_cbf_geometric / mds_step_nominal of a handle creates its [n,18] scratch.  The internal streams and events of
 *     the two-chain rollouts are created and primed by mds_create (shards of 2^16 drones and more) or by
 *     mds_set_rollout_streams(h, 2), never by a rollout.  Set-up calls (mds_create, mds_reset,
 *     mds_set_*, mds_cbf_configure, mds_get/set_state) may allocate, copy from host memory and synchronise.
 *     Because they only enqueue, the mds_step* / mds_rollout* calls can be recorded by a stream capture on `stream`
 *     (hipStreamBeginCapture, torch.cuda.graph) and replayed as a hipGraph; a two-chain rollout forks from and joins back into
 *     the capturing stream through its event pair, which is the capture-legal pattern.  (Scalar arguments such as t are baked into
 *     the graph; replay buys nothing at these kernel sizes)  The ground-effect / downwash physics modes step a
 *     double-buffered state and flip the handle's two buffers on the host once per substep; a captured call bakes the buffers of
 *     the moment into the graph, so under capture a call with an odd number of substeps appends a device-to-device copy of the
 *     state back into the buffer it started from (every replay then starts and ends in the same buffer; eager calls only flip).
 *   - Every call taking a handle runs on the handle's device (mds_config.device) and leaves the calling thread's current
 *     HIP device as it found it; device pointers passed in must belong to that device.
 *   - Every call returns MDS_OK (0) or a negative mds_status; nothing throws or aborts
 *     across the ABI.  mds_strerror() names a status, mds_last_error() adds HIP detail.
 *   - A handle is not re-entrant (one simulation thread, as PIDEnv.py:99-103); distinct
 *     handles (one per GPU / process) are independent.
 *   - The library owns only the handle and its internal SoA state / parameter planes.
 */
#ifndef MDS_H
#define MDS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDS_VERSION 202 /* 0.2.2 */
#define MDS_OBS_DIM 20  /* [UPSTREAM] _getDroneStateVector */
#define MDS_ACT_DIM 4
#define MDS_STATE_DIM 13 /* pos3 | quat4 xyzw | vel3 (world) | body rates3 */
#define MDS_DES_DIM 11   /* pos3 | vel3 | acc3 | yaw | yaw_rate  (Lemniscate.__call__ 5-tuple, flattened) */
#define MDS_LEM_DIM 7    /* a | omega | centre3 | yaw_rate | phase_shift  (Lemniscate.__init__) */
#define MDS_GEO_AUX_DIM 13 /* force | w_des3 | R_des9 row-major  (GeometricControl.compute(return_omegas=True)) */

typedef enum mds_status {
  MDS_OK = 0,
  MDS_EINVAL = -1,      /* bad argument (null pointer, size, enum) */
  MDS_ENOMEM = -2,      /* hipMalloc failed */
  MDS_EHIP = -3,        /* a HIP runtime call failed; see mds_last_error() */
  MDS_EALIGN = -4,      /* a device pointer is not 16-byte aligned */
  MDS_ESTATE = -5,      /* call not valid in the handle's state (e.g. no trajectory set) */
  MDS_EUNSUPPORTED = -6 /* combination not built (e.g. order-3 CBF with fp16 storage) */
} mds_status;

/* MDS_F32C: fp32 buffers and fp32 arithmetic like MDS_F32, with compensated accumulation: inside a control step the integrators add
 * into (value, residual) pairs (two-sum), and between control steps the handle keeps the residuals of the three BODY RATES (one
 * 16-byte group per drone: +32 B per drone-step).  The rounding of the stored rate -- a random walk that turns the attitude and
 * tilts the thrust -- is what limits an uncontrolled fp32 quadrotor: open-loop 240 Hz flight holds 6e-6 instead of 1.4e-5 after 1000
 * steps (north_star's 1e-5).  (Round 2 kept all 13 residuals, +104 B, for 3e-6; residuals of the quaternion alone buy nothing:
 * tests/test_emul_device_math.py.)  Served by every step and rollout entry point of DYN / DYN_DRAG physics, Euler and RK4; the
 * ground-effect / downwash physics modes reject the dtype at mds_create. */
typedef enum mds_dtype { MDS_F32 = 0, MDS_F64 = 1, MDS_F16 = 2, MDS_F32C = 3 } mds_dtype;
/* DYN / DYN_DRAG: [UPSTREAM] Physics.DYN (+ _drag), every entry point.  DYN_GND / DYN_DW / DYN_GND_DRAG_DW add [UPSTREAM]
 * _groundEffect / _downwash (Physics.PYB_GND, PYB_DW, PYB_GND_DRAG_DW: Bullet external forces there, extra terms of the DYN wrench
 * here; spec-level): explicit Euler, f32 / f64.  Upstream refreshes every drone's kinematics between physics substeps and the
 * downwash couples the drones of an env, so these modes run ONE substep per launch on a double-buffered state: mds_step, and the
 * controller paths (mds_step_geometric, mds_step_lqr, mds_step_cbf_geometric, mds_step_nominal: controller + first substep in one
 * launch, the action replayed by the remaining substeps); the mds_rollout_* calls issue the same steps in a loop on the caller's
 * stream (no two-chain split, no state-in-registers form).  mds_step_dslpid / mds_rollout_dslpid run the controller from the state as
 * its own launch, then the substeps. */
typedef enum mds_physics {
  MDS_PHYSICS_DYN = 0, MDS_PHYSICS_DYN_DRAG = 1, MDS_PHYSICS_DYN_GND = 2, MDS_PHYSICS_DYN_DW = 3, MDS_PHYSICS_DYN_GND_DRAG_DW = 4
} mds_physics;
typedef enum mds_integrator { MDS_INTEGRATOR_EULER = 0, MDS_INTEGRATOR_RK4 = 1 } mds_integrator;
typedef enum mds_drone_model { MDS_CF2X = 0, MDS_CF2P = 1 } mds_drone_model;

/* Constructor arguments of [UPSTREAM] CtrlAviary as the reference passes them
 * (PIDEnv.py:106-116, simulations/EnvGeometric.py:89-100) plus the batch axis. */
typedef struct mds_config {
  int32_t num_envs;
  int32_t num_drones;   /* per env */
  int32_t dtype;        /* mds_dtype */
  int32_t physics;      /* mds_physics; [UPSTREAM] Physics.DYN (+ _drag) */
  int32_t integrator;   /* mds_integrator; EULER = [UPSTREAM] _dynamics semantics */
  int32_t drone_model;  /* mds_drone_model: selects the torque mixing of _dynamics */
  int32_t pyb_freq;     /* must be a multiple of ctrl_freq ([UPSTREAM] BaseAviary.__init__) */
  int32_t ctrl_freq;
  int32_t device;       /* HIP device ordinal */
  int32_t track_last_rpm; /* 1: every step also stores the last clipped action in the handle (16 B per drone-step) so
                           * that mds_get_obs can return obs[16:20] later.  0 (default): stored only where the path
                           * reads it back -- DYN_DRAG physics ([UPSTREAM] _drag) and the order-3 CBF / yank path
                           * (calc_z_thrust); otherwise the obs a step call returns is the only copy. */
  /* urdf constants ([UPSTREAM] _parseURDFParameters) */
  double M, L, KF, KM, J[3], G, thrust2weight, drag_coeff[3];
} mds_config;

/* control/geometric.py:14-23 */
typedef struct mds_geometric_gains {
  double Kp[3], Kv[3], KR[3], Kw[3];
  double g;              /* 9.81 in the reference (env.G is 9.8) */
  double max_tilt_angle; /* rad */
} mds_geometric_gains;

typedef struct mds_handle mds_handle;

int mds_version(void);
const char* mds_strerror(int status);
const char* mds_last_error(void);

/* Fills cfg with the CF2P / CF2X constants and the reference's defaults
 * (PIDEnv.py:18-29: pyb = ctrl = 100 Hz... callers override). */
int mds_default_config(int drone_model, mds_config* cfg);
int mds_default_geometric_gains(mds_geometric_gains* gains);

/* [UPSTREAM] CtrlAviary.__init__ / close().  State starts at the origin with identity
 * attitude; call mds_reset. */
int mds_create(const mds_config* cfg, mds_handle** out);
int mds_destroy(mds_handle* h);

/* Derived attributes the reference reads off `env` (SURVEY.md 3.4 census):
 * out[0..7] = GRAVITY(M*G), HOVER_RPM, MAX_RPM, MAX_THRUST, MAX_XY_TORQUE, MAX_Z_TORQUE,
 *             CTRL_TIMESTEP, PYB_TIMESTEP */
int mds_get_derived(const mds_handle* h, double out[8]);

/* [UPSTREAM] BaseAviary.reset()/_housekeeping: pose from initial_xyzs / initial_rpys
 * (host double [n,3] each), zero velocities, zero last action. */
int mds_reset(mds_handle* h, const double* xyz_host, const double* rpy_host, void* stream);

/* Zero-copy view of the library-owned state (SURVEY 8b `mds_state_ptrs`): comp_dev[k] points at component k of drone 0
 * (k = 0..2 position RELATIVE to the drone's local-frame origin, 3..6 quaternion xyzw, 7..9 velocity, 10..12 body rates) in
 * the handle's storage dtype, and component k of drone i lives stride_elems[k] * i elements further (the packed layout keeps
 * four components per 16-byte group: stride 4 for k < 12, 1 for k = 12).  origin_dev[k] (may be NULL): the origin planes,
 * stride 1, float for MDS_F32 / MDS_F16 handles and double for MDS_F64; world position = state position + origin.  The
 * pointers stay valid until mds_destroy, except that the ground-effect / downwash physics modes swap their two state
 * buffers every substep (ask again after each mds_step there).  MDS_F32C handles: these are the fp32 values; the rate residuals
 * stay private (mds_get_state returns value + residual for the body rates). */
int mds_state_ptrs(mds_handle* h, void* comp_dev[13], size_t stride_elems[13], void* origin_dev[3]);

/* Test / checkpoint access to the 13-float state in the WORLD frame (host double [n,13]).
 * Synchronises the stream. */
int mds_get_state(mds_handle* h, double* state_host, void* stream);
int mds_set_state(mds_handle* h, const double* state_host, void* stream);

/* Local-frame origin per drone (host double [n,3]); positions are stored relative to it so
 * that fp32 storage keeps ~1e-7 m resolution far from the world origin.  Re-bases the
 * stored state; observations are always world-frame.  mds_set_lemniscate sets it to the
 * trajectory centre. */
int mds_set_origin(mds_handle* h, const double* origin_host, void* stream);

/* [UPSTREAM] BaseAviary._computeObs(): obs_dev [n,20] from the current state.  obs[16:20]
 * is the last clipped action of the most recent mds_step* call (zeros after reset) when the handle
 * tracks it (mds_config.track_last_rpm); on a handle that does not, it is NaN once a step has run. */
int mds_get_obs(mds_handle* h, void* obs_dev, void* stream);

/* [UPSTREAM] BaseAviary.step(action) for every drone: clip RPM to [0, MAX_RPM],
 * pyb_freq/ctrl_freq physics substeps of _dynamics (+_drag), pack the 20-float obs.
 * Call sites replaced: simulations/EnvGeometric.py:469, PIDEnv.py:176.
 * action_dev [n,4] RPM, obs_dev [n,20] (may be NULL: state only). */
int mds_step(mds_handle* h, const void* action_dev, void* obs_dev, void* stream);

/* Constant external force on every drone, world frame, Newton -- the reference's wind:
 * p.applyExternalForce(DRONE_IDS[i], -1, [wind_force,0,0], WORLD_FRAME) every control step
 * (simulations/EnvGeometric.py:34,463-467; wind_force = 2.5e-4).  Added to the rigid-body force in
 * every physics substep.  Default zero. */
int mds_set_wind(mds_handle* h, const double force_world[3]);

/* General trajectories: the reference's trajectories/ family (Lemniscate, CircleTrajectory,
 * LineTrajectory, WaitTrajectory, CompoundTrajectory, RotateTrajectory) flattened by the caller into
 * per-drone segment lists (layout: multidronesim_amd/csrc/mds_traj.hpp; the Python classes build it).
 * segs_host double [total, MDS_SEG_DIM]; offsets_host int32 [n+1] (drone i owns segments
 * offsets[i] .. offsets[i+1]-1, at most 65535); compound_host int32 [n] (1: CompoundTrajectory lookup
 * with its past-the-end rule, 0: a single trajectory evaluated at t); anchor_host double [n,3]: local-frame
 * origin per drone (e.g. the first segment's centre / start).  Replaces mds_set_lemniscate's trajectories:
 * mds_step_geometric / mds_rollout_geometric then run the general kernel.  The device image is the library's own
 * (field-major; drones whose rows are bytewise identical share one copy; tables with equal piece counts piece-major),
 * so broadcasting the same D tables over every env costs D tables of device memory, not n. */
#define MDS_SEG_DIM 40
int mds_set_trajectory_segments(mds_handle* h, const double* segs_host, const int32_t* offsets_host,
                                const int32_t* compound_host, const double* anchor_host, int32_t total_segments, void* stream);
/* Trajectory.__call__(t) for every drone from the segment tables: des_dev [n,11] world frame */
int mds_traj_eval(mds_handle* h, double t, void* des_dev, void* stream);

/* trajectories/Lemniscate.py:14-30: one Lemniscate per drone, params_host double [n,7] =
 * (a, omega, centre_x, centre_y, centre_z, yaw_rate, phase_shift). */
int mds_set_lemniscate(mds_handle* h, const double* params_host, void* stream);
int mds_set_geometric_gains(mds_handle* h, const mds_geometric_gains* gains);

/* One fused control step of simulations/EnvGeometric.py:434-469 for every drone:
 * trajs[j](t) -> GeometricControl.compute(obs[j]) -> input_to_action -> env.step(action),
 * with no action / observation round trip through HBM.
 * obs_dev [n,20] or NULL; action_dev [n,4] (the unclipped controller RPM) or NULL. */
int mds_step_geometric(mds_handle* h, double t, void* obs_dev, void* action_dev, void* stream);

/* n_steps consecutive mds_step_geometric calls at t0, t0+dt, ... enqueued from C (dt =
 * CTRL_TIMESTEP, accumulated like the reference loop, EnvGeometric.py:473).  obs_dev (or
 * NULL) receives the observation of EVERY step (same buffer, overwritten) when
 * obs_every_step != 0, else only the last step's. */
int mds_rollout_geometric(mds_handle* h, double t0, int n_steps, void* obs_dev, int obs_every_step, void* stream);

/* The plain env.step loop with a replayed action table, issued from C: step j = first_step + k (k < n_steps) applies
 * actions_dev[j % n_action_sets] ([n_action_sets, n, 4] RPM) and writes its observation into obs_log_dev[j % log_slots]
 * ([log_slots, n, 20]; NULL: no observations) -- the reference's `obs = env.step(action)` + per-step log (the [T, D, 20]
 * array EnvGeometric.py:553 saves), for every env.  episode_len > 0: before every step j > 0 with j % episode_len == 0 the
 * drones go back to the poses of the last mds_reset (zero velocities), on the device -- an open-loop rollout under the
 * explicit-Euler model leaves every numeric range after ~1500 steps at 240 Hz.  Large shards run as two half-shard step
 * chains on two internal streams (mds_set_rollout_streams). */
int mds_rollout_step(mds_handle* h, const void* actions_dev, int n_action_sets, int first_step, int n_steps, void* obs_log_dev,
                     int log_slots, int episode_len, void* stream);
/* The same loop with steps_per_launch control steps per kernel launch (state in registers between them; a launch never
 * crosses an episode boundary): per drone-step only the action is read and the observation written.  Same arithmetic as
 * mds_rollout_step; with fp16 storage the state is rounded to fp16 once per launch instead of once per step. */
int mds_rollout_step_fused(mds_handle* h, const void* actions_dev, int n_action_sets, int first_step, int n_steps, void* obs_log_dev,
                           int log_slots, int episode_len, int steps_per_launch, void* stream);
/* mds_reset's effect again from the poses it was last given, enqueued on `stream` without any host copy or synchronisation */
int mds_reset_async(mds_handle* h, void* stream);

/* How mds_rollout_geometric / mds_rollout_step / mds_rollout_cbf_geometric issue their steps.  Drones never read each
 * other's rows in the fused step (and a barrier row couples drones of one env only), so the two halves of the shard are
 * independent step chains: on two internal streams one half's load/store bursts fill the other's compute phase, and the
 * gaps between dependent launches disappear (C3: 17.5 -> 15.0-15.7 us per step; half-shard launches also carry unused
 * LDS so that 5 instead of 8 workgroups share a CU and the chains interleave from the first step).
 * 0 = auto (geometric / plain step: two streams from 2^19 drones, from 2^18 for calls of 1000+ steps; CBF loop: from
 * 2^16 drones; calls of fewer than 16 steps always stay on the caller's stream -- a two-chain call costs ~35 us), 1 = the caller's stream only, 2 = always split (a set-up call then: it creates the internal streams of a
 * handle that mds_create gave none).  Results are bit-identical either way; the caller's stream orders the whole call
 * (events on entry and exit -- the exit events are recorded even when a launch in between failed), so the usual stream
 * semantics hold.  ROCm maps a process's streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues: in a process with more than
 * two or three other ACTIVE streams the internal stream may share a queue with the caller's and the chains then run one after the
 * other (results unchanged, time lost).  Raise GPU_MAX_HW_QUEUES, or set MDS_SPLIT_STREAM_PRIORITY=high (or low) before the handle's
 * streams are created: a stream of another priority level has hardware queues of its own.  The variable is read at every stream
 * creation (mds_create / the first mds_set_rollout_streams(h, 2) of a handle), so handles created after a change see it. */
int mds_set_rollout_streams(mds_handle* h, int n_streams);
/* What the most recent mds_rollout_geometric / mds_rollout_step / mds_rollout_dslpid / mds_rollout_cbf_geometric of this handle did: 1 = the
 * caller's stream only, 2 = two chains on the internal streams, 0 = no rollout yet. */
int mds_get_last_rollout_streams(const mds_handle* h);
/* What a call of n_steps would do under the current setting: loop 0 = mds_rollout_geometric / mds_rollout_step / mds_rollout_dslpid, loop 1 =
 * mds_rollout_cbf_geometric.  Returns 1 or 2.  (A caller that warms a path up asks this for the length it is going to time.) */
int mds_rollout_streams_for(const mds_handle* h, int loop, int n_steps);

/* The LAUNCH FORM of mds_rollout_geometric (the loop of simulations/EnvGeometric.py:434-469; SURVEY 8e: an env shard of config 3 on
 * G GPUs is E / G envs per GPU; below ~2^17 drones a dependent launch per control step costs more than the step moves, and above it a
 * launch per step streams the state through HBM twice per step where one launch per 50 steps keeps it in registers).
 * form 1 = one launch of the fused step kernel per control step (on two chains where mds_set_rollout_streams says so): bit-identical to
 * n_steps calls of mds_step_geometric.  form 2 = the whole-rollout kernel (mds_rollout_geometric_fused's) in launches of steps_per_launch
 * control steps (default 50): the state stays in registers between the steps of a launch, every step's observation is still written to
 * obs_dev when obs_every_step != 0; same arithmetic, results agree with form 1 to rounding (the two kernels contract FMAs differently;
 * each is parity-tested against the oracle).  form 0 = auto: form 2 for shards of 2^13 drones and more and calls of
 * 8 steps and more -- the faster form at every such size (measured: profiles/r04_shard_sweep.json, r04_form_sweep.json) --, form 1 otherwise
 * and always with fp16 storage or ground effect / downwash.  A caller that needs results bit-identical to mds_step_geometric calls, whatever the shard
 * size and call length, selects form 1.  steps_per_launch 0 keeps the current value. */
int mds_set_rollout_form(mds_handle* h, int form, int steps_per_launch);
/* 1 or 2: what an mds_rollout_geometric call of n_steps would do under the current setting; what the most recent one did (0: none yet). */
int mds_rollout_form_for(const mds_handle* h, int n_steps);
int mds_get_last_rollout_form(const mds_handle* h);

/* The same n_steps control steps in ONE kernel launch: state and trajectory parameters stay in
 * registers between steps; every step's observation is streamed to obs_log_dev [n_steps, n, 20]
 * (the reference's `observations.append(obs)` -> np.save, EnvGeometric.py:471,553) when it is not
 * NULL; obs_last_dev [n,20] (or NULL) receives the final observation.  Lemniscate planes or segment tables.  Same arithmetic as n_steps
 * calls of mds_step_geometric (results agree to rounding: the two kernels may contract FMAs differently). */
int mds_rollout_geometric_fused(mds_handle* h, double t0, int n_steps, void* obs_log_dev, void* obs_last_dev, void* stream);

/* ---- stand-alone per-drone operators (same arithmetic as the fused path) ---------------- */

/* trajectories/Lemniscate.py:32-63 `__call__(t)`: des_dev [n,11] world frame, using the
 * handle's trajectories. */
int mds_lemniscate_eval(mds_handle* h, double t, void* des_dev, void* stream);

/* control/geometric.py:59-115 `GeometricControl.compute(obs)` after
 * set_desired_trajectory(...): obs_dev [n,20], des_dev [n,11] -> rpm_dev [n,4]
 * (input_to_action applied, utils/model_conversions.py:85-103).  If aux_dev != NULL also
 * writes the return_omegas=True triple (force, w_des, R_des) as [n,13]. */
int mds_geometric_compute(mds_handle* h, const void* obs_dev, const void* des_dev, void* rpm_dev, void* aux_dev,
                          void* stream);

/* utils/model_conversions.py:85-103 / :69-83 */
int mds_input_to_action(mds_handle* h, const void* u_dev /*[n,4]*/, void* rpm_dev /*[n,4]*/, void* stream);
int mds_action_to_input(mds_handle* h, const void* rpm_dev /*[n,4]*/, int cap_rpm, void* u_dev /*[n,4]*/, void* stream);
/* utils/model_conversions.py:20-58 obs_to_lin_model(obs, dim = 9 | 10 | 12[, env]) and :105-114 obs_to_geo_model(obs) (dim = 18):
 * obs_dev [n,20] -> x_dev [n,dim].  dim 10 carries F = calc_z_thrust(env, obs) (:137-143); dim 18 = [pos, R row-major, vel, ang_v]. */
int mds_obs_to_model(mds_handle* h, const void* obs_dev, int dim, void* x_dev, void* stream);

/* model/dynamics.py:83-106 `QuadrotorDynamics.dynamics(t, state, u)`: state_dev [count,18]
 * (p, R row-major, v, w), u_dev [count,4] (thrust, torques) -> out_dev [count,12].
 * m, J, g are that class's own constants (Hummingbird defaults :24-28; J stays the stale
 * Hummingbird one after load_env_params, :18).  dtype = element type of the buffers. */
int mds_quadrotor_dynamics(int dtype, int count, const void* state_dev, const void* u_dev, double m, const double J[3],
                           double g, void* out_dev, void* stream);

/* The call site of QuadrotorDynamics.dynamics -- simulations/CompareModels.py:46-56, the loop body over a logged rollout, as ONE launch:
 * for each of `count` observation rows (obs_dev [count,20], e.g. the [T,D,20] history GeometricEnv.do_control leaves behind)
 *   x_lin_dev    [count,12] = obs_to_lin_model(obs)                                      (utils/model_conversions.py:20-58, dim 12)
 *   xdot_lin_dev [count,12] = LinearizedModel.calc_xdot_from_obs(obs) = A (x - x_eq) + B (u - u_eq), u = action_to_input(env, obs[16:20]),
 *                             x_eq = (0 .. 0, position), u_eq = (u_eq0, 0, 0, 0), u_eq0 = mass * g of the model     (model/linearized.py:83-104)
 *   xdot_geo_dev [count,12] = geo_x_dot_to_linear(QuadrotorDynamics.dynamics(None, obs_to_geo_model(obs), u))
 *                                                                    (utils/model_conversions.py:105-135, model/dynamics.py:83-106)
 * A_host [12,12], B_host [12,4] row-major doubles: the model's own matrices (A, B or Ahat, Bhat); dyn_m, dyn_J, dyn_g as
 * mds_quadrotor_dynamics (after load_env_params: the env's m and g, the stale Hummingbird J).  Any output may be NULL (not all three).
 * The handle supplies the env constants of action_to_input and the element type; count is independent of the handle's n. */
int mds_compare_models(mds_handle* h, int count, const void* obs_dev, const double* A_host, const double* B_host, double u_eq0, double dyn_m,
                       const double dyn_J[3], double dyn_g, void* xdot_lin_dev, void* xdot_geo_dev, void* x_lin_dev, void* stream);
/* model/linearized.py:92-104 `LinearizedModel.calc_xdot(x, action)` on states of the caller's own -- the right-hand side
 * roll_out_linear_system integrates (simulations/CompareModels.py:84-92): x_dev [count,12], action_dev [count,4] RPM -> xdot_dev [count,12]. */
int mds_linear_xdot(mds_handle* h, int count, const void* x_dev, const void* action_dev, const double* A_host, const double* B_host,
                    double u_eq0, void* xdot_dev, void* stream);
/* utils/model_conversions.py:4-19 `rpy_to_rot(rpy)`: rpy_dev [count,3] -> R_dev [count,9] row-major, R = Rz(yaw) Ry(pitch) Rx(roll). */
int mds_rpy_to_rot(int dtype, int count, const void* rpy_dev, void* R_dev, void* stream);
/* utils/model_conversions.py:116-122 `geo_model_to_obs(x)`: x18_dev [count,18] (p, R row-major, v, w) -> obs16_dev [count,16]
 * (p, quaternion xyzw as scipy's Rotation.from_matrix(R).as_quat() gives it, three zeros in the rpy slots, v, w). */
int mds_geo_model_to_obs(int dtype, int count, const void* x18_dev, void* obs16_dev, void* stream);

/* ---- ECBF safety filter (cbf/cbf.py, cbf/qptracker.py) ------------------------------------ */

/* DroneCBF.__init__ (cbf/cbf.py:545-580) after its own derivations: Kcbf = place_poles gains
 * (ascending), umax = [MAX_THRUST | Ymax, omega_max3], Fmin = -M*G, Fmax = MAX_THRUST.
 * order 2 = LinearizedOmegaModel (xdim 9), order 3 = LinearizedYankOmegaModel (xdim 10). */
typedef struct mds_cbf_params {
  int32_t order;     /* 2 or 3 */
  int32_t n_obs;     /* static sphere obstacles (x_obs_list / obs_r_list), <= 16 */
  int32_t max_iter;  /* QP iteration cap per env; 0 = default */
  int32_t reserved;
  double Kcbf[3];
  double umax[4];
  double safety_radius, zscale;
  double Fmin, Fmax;
  double tol;        /* convergence: largest remaining move of a thrust variable; 0 = default */
} mds_cbf_params;

/* obstacles_host: double [n_obs,4] = centre xyz, radius (NULL when n_obs == 0).  num_drones <= 32. */
int mds_cbf_configure(mds_handle* h, const mds_cbf_params* p, const double* obstacles_host);

/* rows of G u <= h per env: D(D-1)/2 + 8D (+2D for order 3) + D*n_obs (cbf/cbf.py:337-367) */
int mds_cbf_num_rows(const mds_handle* h);

/* CBF._build_ineq_const (cbf/cbf.py:308-367) for every env, dense, in the reference's row order:
 * x_dev, xdes_dev [n, xdim] (linear-model states, xdim = 9 | 10) -> G_dev [E, m, 4D], h_dev [E, m]. */
int mds_cbf_rows(mds_handle* h, const void* x_dev, const void* xdes_dev, void* G_dev, void* h_dev, void* stream);

/* DroneQPTracker.compute_control (cbf/qptracker.py:22-34) for every env:
 * obs_dev [n,20], xdes_dev [n,xdim], u_nominal_dev [n,4] (thrust already offset by -M*G,
 * simulations/CBFTest.py:339) -> u_safe_dev [n,4]; status_dev [E] int32: 0 = QP solved (the unique
 * minimiser), 1 = infeasible (MODELLED FALLBACK): the rows of that env admit no point (or the
 * iteration cap ran out) and its u_safe is u_nominal unchanged.
 * What status 1 is and is not: the reference takes its `return u_nominal` exit (qptracker.py:30-34)
 * only when cvxopt.solvers.qp RAISES (qptracker.py:103-112: success = True as soon as the call
 * returns).  cvxopt 1.3.2's coneqp does not certify infeasibility of a QP: on such rows it returns
 * status 'unknown' with its last iterates (iteration limit / singular KKT system) and raises only
 * the rank ValueError that P = I rules out -- so the reference most likely applies that last
 * iterate to all drones of the env.  cvxopt is not available to this build; "infeasible -> u_nominal"
 * is therefore this library's own, modelled, policy: parity with the reference on status-1 envs is
 * UNPINNED and probably different.  On status-0 envs the minimiser is unique, so any exact solver
 * agrees with cvxopt up to cvxopt's own tolerances (feastol / abstol 1e-7, reltol 1e-6).
 * Order 2: D coupled thrust variables, omega box-clipped.  Order 3: 3D coupled
 * (yank, wx, wy) variables (<= 63), omega_z clipped to its box / force-box interval. */
int mds_cbf_filter(mds_handle* h, const void* obs_dev, const void* xdes_dev, const void* u_nominal_dev, void* u_safe_dev,
                   int32_t* status_dev, void* stream);

/* Solver work of the most recent filter launch: iters_dev [E] int32 = active-set iterations (rows added or dropped) each env's QP
 * took; 0 = the nominal input already satisfied every row (or the env was declared infeasible while its rows were built).
 * A device-to-device copy enqueued on `stream`; for profiling the scene, not part of the control path. */
int mds_cbf_last_iterations(mds_handle* h, int32_t* iters_dev, void* stream);

/* ---- low-level body-rate controller (control/low_level/thrust_omega_ctrl.py) ---------------- */

/* ThrustOmegaController.reset(): zero last_omega and the integral of every drone (mds_reset does
 * this too). */
int mds_lowlevel_reset(mds_handle* h, void* stream);

/* LQROmegaController.compute_low_level(u, obs) (control/lqr/lqr_omega_controller.py:77-88) ->
 * ThrustOmegaController.computeControlFromInput (thrust_omega_ctrl.py:81-132) for every drone:
 * u_dev [n,4] = (thrust N, target body rates), obs_dev [n,20] (its world-frame rate is rotated to
 * the body frame) -> rpm_dev [n,4].  Stateful (PID memory lives in the handle). */
int mds_thrust_omega_compute(mds_handle* h, const void* u_dev, const void* obs_dev, void* rpm_dev, void* stream);
/* same with the current BODY rates given directly: rates_dev [n,3] (computeControlFromInput's own signature) */
int mds_thrust_omega_from_rates(mds_handle* h, const void* u_dev, const void* rates_dev, void* rpm_dev, void* stream);

/* ---- [UPSTREAM] DSLPIDControl as PIDEnv.py drives it (not in the reference tree: spec-level, unpinned) ---- */
typedef struct mds_dslpid_gains {
  double P_COEFF_FOR[3], I_COEFF_FOR[3], D_COEFF_FOR[3], P_COEFF_TOR[3], I_COEFF_TOR[3], D_COEFF_TOR[3];
} mds_dslpid_gains;
int mds_default_dslpid_gains(mds_dslpid_gains* g);           /* upstream defaults; PIDEnv.py:128-133 halves them */
int mds_set_dslpid_gains(mds_handle* h, const mds_dslpid_gains* g);
int mds_dslpid_reset(mds_handle* h, void* stream);           /* DSLPIDControl.reset(): zero the PID memory */
/* computeControlFromState(CTRL_TIMESTEP, state=obs[j], target_pos, target_rpy) for every drone (PIDEnv.py:166-169):
 * obs_dev [n,20], target_pos_dev [n,3], target_rpy_dev [n,3] -> rpm_dev [n,4].  Stateful. */
int mds_dslpid_compute(mds_handle* h, const void* obs_dev, const void* target_pos_dev, const void* target_rpy_dev, void* rpm_dev,
                       void* stream);
/* MultiDroneEnv.sim_step (PIDEnv.py:161-176): the same controller on the handle's own state, fused with
 * env.step(action).  obs_dev [n,20] / action_dev [n,4] optional. */
int mds_step_dslpid(mds_handle* h, const void* target_pos_dev, const void* target_rpy_dev, void* obs_dev, void* action_dev, void* stream);
/* n_steps of that sim_step as a C loop (PIDEnv.py:201-207, `for i in range(...): env.sim_step()`): step first_step + k uses target
 * set (first_step + k) % n_target_sets of target_pos_dev / target_rpy_dev [n_target_sets, n, 3] (1 = the fixed TARGET_POSITIONS of
 * PIDEnv; more = a waypoint table).  obs_dev [n,20] receives every step's observation (obs_every_step) or the last one only.
 * Bit-identical to n_steps calls of mds_step_dslpid; follows the stream policy of mds_set_rollout_streams (loop 0). */
int mds_rollout_dslpid(mds_handle* h, const void* target_pos_dev, const void* target_rpy_dev, int n_target_sets, int first_step,
                       int n_steps, void* obs_dev, int obs_every_step, void* stream);

/* LQROmegaController (control/lqr/lqr_omega_controller.py): K [4,9] row-major is the gain its
 * compute_gain_matrix() obtains from solve_continuous_are on the host (:53-57). */
int mds_set_lqr_omega_gain(mds_handle* h, const double K[36]);
/* LQROmegaController.compute(obs, skip_low_level=True) (:90-119): obs_dev [n,20], des_dev [n,11]
 * (pos, vel, -, yaw, - of set_desired_trajectory) -> u_dev [n,4] = (F, wx, wy, wz) after cap_u. */
int mds_lqr_omega_compute(mds_handle* h, const void* obs_dev, const void* des_dev, void* u_dev, void* stream);
/* LQRController (control/lqr/lqr_controller.py) on LinearizedModel (model/linearized.py) -- the default 'lqr' controller of
 * simulations/EnvGeometric.py (:32, :425-427).  K [4,12] row-major is the gain of its compute_gain_matrix() (:53-57; host ARE,
 * also with the Ahat/Bhat of use_noisy_model); state [rpy, ang_v, vel, pos], input [F, tau_x, tau_y, tau_z]. */
int mds_set_lqr_gain(mds_handle* h, const double K[48]);
/* LQRController.compute(obs) (:83-113): obs_dev [n,20], des_dev [n,11] (pos, vel, -, yaw, omega) -> u_dev [n,4] (F clipped at 0,
 * as the reference's in-place mixer leaves it, model_conversions.py:88) and action_dev [n,4] RPM; either output may be NULL. */
int mds_lqr_compute(mds_handle* h, const void* obs_dev, const void* des_dev, void* u_dev, void* action_dev, void* stream);
/* The do_control step of simulations/EnvGeometric.py:434-469 with that controller for every drone: trajectory sample ->
 * LQRController.compute -> env.step.  Trajectories as for mds_step_geometric (Lemniscate planes or segment tables). */
int mds_step_lqr(mds_handle* h, double t, void* obs_dev, void* action_dev, void* stream);
/* n_steps of that loop in ONE launch (Lemniscate trajectories), as mds_rollout_geometric_fused does for the geometric controller */
int mds_rollout_lqr_fused(mds_handle* h, double t0, int n_steps, void* obs_log_dev, void* obs_last_dev, void* stream);

/* LQRYankOmegaController (control/lqr/lqr_YO_controller.py): K [4,10] row-major from its
 * compute_gain_matrix() (:59-64), state [r,p,y,F,vx,vy,vz,x,y,z], input [yank, wx, wy, wz]. */
int mds_set_lqr_yank_omega_gain(mds_handle* h, const double K[40]);
/* LQRYankOmegaController.compute(obs, skip_low_level=True) (:99-124): obs_dev [n,20] (its columns 16:20 give the
 * thrust state through calc_z_thrust), des_dev [n,11] as mds_lqr_omega_compute -> u_dev [n,4] = -K e (no cap). */
int mds_lqr_yank_omega_compute(mds_handle* h, const void* obs_dev, const void* des_dev, void* u_dev, void* stream);
/* LQRYankOmegaController.compute_low_level(u, obs) (:85-97) -> YankOmegaController.computeControlFromInput
 * (control/low_level/yank_omega_ctrl.py:39-55): thrust = calc_z_thrust(obs) + yank * CTRL_TIMESTEP, then the
 * ThrustOmega PID of mds_thrust_omega_compute (same PID memory in the handle). */
int mds_yank_omega_compute(mds_handle* h, const void* u_dev, const void* obs_dev, void* rpm_dev, void* stream);
/* nominal controller of mds_step_cbf_geometric: 0 = GeometricControl(return_omegas), 1 = LQROmegaController,
 * 2 = LQRYankOmegaController (the only one valid with an order-3 CBF, and only with it) */
int mds_cbf_set_nominal(mds_handle* h, int which);
/* How mds_step_cbf_geometric / mds_rollout_cbf_geometric issue a control step.  0 (default): the QP of every env in its own
 * wavefront (k_cbf_filter_gi), then low level + physics + the next step's nominal input per drone (k_lowlevel_step) -- the faster
 * form when a sizeable share of the envs iterate (C4, SURVEY 8d scene: 49 us against 60 us per control step).  1: ONE launch per
 * step (k_cbf_step: nominal controller, the wavefront's 64 / D QPs, low level + physics) where it applies -- order 2, D a divisor
 * of 64 up to 16, Euler, DYN, geometric or LQR-omega nominal, f32 / f32c / f64, no action output; three launches otherwise --
 * the faster form when few envs iterate (obstacles far away: 29 us against 40 us).  Same QP, same statuses and iteration
 * counts; observations equal to rounding (the two forms contract FMAs differently), so pick one per handle: each form's rollout
 * is bitwise its own step-by-step loop.  MDS_CBF_FUSED=1 in the environment selects 1 at mds_cbf_configure.
 * 2: a step is ONE launch of the several-steps-per-launch kernel (k_cbf_rollout, see mds_rollout_cbf_geometric_fused) with one step,
 * and mds_rollout_cbf_geometric becomes mds_rollout_cbf_geometric_fused at 25 steps per launch, where that kernel applies and no
 * action output is asked for (form 0 otherwise) -- the fastest per-step form measured (C4: 27 us per control step against 38 us
 * for form 0; 21.6 us at 50 steps per launch); bitwise the fused rollout's results whatever the steps per launch. */
int mds_cbf_set_step_kernel(mds_handle* h, int one_launch);
/* What the most recent mds_step_cbf_geometric / mds_rollout_cbf_geometric[_fused] of this handle launched: 2 the several-steps-per-launch
 * kernel, 1 the one-launch kernel, 0 the QP launch + the low-level launch, -1 no CBF-filtered step yet (negative mds_status for a null handle is -1 as well: check the handle). */
int mds_cbf_last_step_kernel(const mds_handle* h);

/* One CBF-filtered control step for every env.
 * Order 2 (simulations/CBFTest.py:303-350): nominal (force - M G, w_des) from the geometric controller or the
 * LQR on the handle's trajectories -> ECBF QP -> + M G -> ThrustOmega low level -> env.step.
 * Order 3 (simulations/CBFTestOrd3.py:306-352): nominal (yank - M G, w) from the yank-omega LQR (the hover force is
 * subtracted from the yank there, :341, and not added back, :350 -- kept) -> ECBF QP on xdes =
 * [0,0,yaw, G M, vel, pos] -> YankOmega low level -> env.step.
 * obs_dev [n,20] holds the CURRENT observation on entry (as returned by the previous step: the order-3 path reads
 * its thrust state from columns 16:20) and the next one on return; status_dev [E] as mds_cbf_filter;
 * action_dev [n,4] (RPM) optional. */
int mds_step_cbf_geometric(mds_handle* h, double t, void* obs_dev, int32_t* status_dev, void* action_dev, void* stream);

/* n_steps of mds_step_cbf_geometric enqueued from C (t advances by 1/ctrl_freq per step, like the reference loop); status_dev
 * holds the last step's per-env status.  Large batches run as two env halves on two internal streams (mds_set_rollout_streams
 * policy: auto from 2^16 drones and 16 steps): a barrier couples drones of one env only, so the halves are independent step
 * chains and one half's QP kernel overlaps the other's memory-bound kernels.  Results are those of the step-by-step loop. */
int mds_rollout_cbf_geometric(mds_handle* h, double t0, int n_steps, void* obs_dev, int32_t* status_dev, void* stream);

/* n_steps of mds_step_cbf_geometric with steps_per_launch control steps PER LAUNCH (the persistent form of the loop
 * simulations/CBFTest.py:303-350: k_cbf_rollout).  A workgroup owns whole envs for the launch; state, u_hat, xdes and u_safe stay on the
 * chip between steps, the QPs of a workgroup's envs are handed out to its wavefronts heaviest first, and there is no chip-wide step
 * boundary inside a launch.  obs_log_dev: NULL, or a ring [log_slots, n, 20] -- step k of this call writes its observation (the
 * reference's observations.append(obs), CBFTest.py:351) into slot (first_slot + k) % log_slots; without a log only the last step's
 * observation is materialised.  obs_dev [n,20]: the last step's observation (out).  status_dev [E]: the last step's statuses as
 * mds_cbf_filter; status_log_dev: NULL or [n_steps, E], every step's.  Covers order 2, any D <= 16 (round 4: an env padded to 4, 8 or 16 lanes; obstacle and thrust-box rows folded into per-drone bounds),
 * explicit Euler at pyb_freq == ctrl_freq, DYN, geometric or LQR-omega nominal, f32 / f32c / f64, at most 2^27 drones per handle
 * (32-bit byte offsets into the per-drone planes); and ORDER 3 (round 4: the simulations/CBFTestOrd3.py:306-352 loop -- lqr-yank-omega nominal
 * selected with mds_cbf_set_nominal(h, 2), YankOmega low level -- one wavefront per env, D <= 16, f32 / f64; obs_dev is then IN as well: the current
 * observation, whose RPM echo starts the thrust state); MDS_EUNSUPPORTED otherwise (RK4, drag, ground effect / downwash, pyb_freq != ctrl_freq: use
 * mds_rollout_cbf_geometric).  Same feasible set, minimiser and statuses as the step-by-step loop
 * (the single-variable rows enter as two bounds per drone, so a dominated row never counts as an iteration); observations equal to rounding (the
 * kernels contract FMAs differently, as the one-launch step does: mds_cbf_set_step_kernel).  The call only enqueues (kernel launches and one
 * device-to-device copy of the last ring slot): it may be captured into a hipGraph. */
int mds_rollout_cbf_geometric_fused(mds_handle* h, double t0, int n_steps, int steps_per_launch, void* obs_log_dev, int log_slots,
                                    int first_slot, void* obs_dev, int32_t* status_dev, int32_t* status_log_dev, void* stream);

/* The same step without the filter: nominal LQR (mds_cbf_set_nominal 1 or 2) -> its low level -> env.step, i.e.
 * ctrl[j].compute(obs[j]) + env.step(action) of simulations/EnvGeometricOmega.py:314,327 (LQROmegaController +
 * ThrustOmegaController) and simulations/EnvGeometricYankOmega.py:319,332 (LQRYankOmegaController + YankOmegaController).
 * obs_dev as for mds_step_cbf_geometric.  With action_dev NULL and Lemniscate trajectories the call is one launch (the
 * one-step instance of the whole-rollout kernel); with the action wanted it is two (nominal, low level + step). */
int mds_step_nominal(mds_handle* h, double t, void* obs_dev, void* action_dev, void* stream);
/* n_steps of that loop in ONE launch (Lemniscate trajectories; the low level's PID memory stays in registers): obs_log_dev
 * [n_steps, n, 20] or NULL; obs_dev [n,20] holds the current observation on entry and the last one on return. */
int mds_rollout_nominal_fused(mds_handle* h, double t0, int n_steps, void* obs_log_dev, void* obs_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDS_H */
