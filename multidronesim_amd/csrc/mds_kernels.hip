// HIP kernels (gfx950 / CDNA4, wave64) of the batched multi-drone step.
//
// Data layout in HBM (DESIGN.md "Data layout"):
//   state   : packed struct-of-arrays.  The 13 components of drone i live in three 4-wide
//             groups (pos3|qx, qy qz qw|vx, vy vz|wx wy: element 4*i of group g) plus one scalar
//             plane (wz): lane i issues three 16-byte and one 4-byte access, every wave access
//             is one contiguous 1 KiB / 256-byte transaction (ld = n rounded up to 256).
//   origin  : 3 planes (compute type) -- local-frame origin per drone.
//   lem     : Lemniscate parameters, packed the same way: (a, omega, yaw_rate, phase) x4 |
//             (cx, cy) x2 | cz.
//   action  : caller's [n,4] array-of-structs: one 16-byte load per lane, contiguous.
//   obs     : caller's [n,20] array-of-structs.  A wave's 64 drones own one contiguous
//             5 KiB span of it; the 20 floats of each lane are staged through LDS and the
//             span is written with 16-byte-per-lane, fully coalesced stores.
// One drone per lane, 256 lanes per workgroup; no MFMA (no contraction wider than 4x4).
#include <hip/hip_runtime.h>

#include "mds_math.hpp"
#include "mds_traj.hpp"

namespace mds {

typedef _Float16 half_t;

#ifndef MDS_KBLOCK
#define MDS_KBLOCK 256
#endif
constexpr int kBlock = MDS_KBLOCK;
constexpr int kWave = 64;
// minimum waves per SIMD requested from the register allocator for k_step (2nd __launch_bounds__
// argument).  The fused kernel is left to the allocator: 78 VGPRs = 6 waves/SIMD without spills;
// forcing 7 or 8 spills and measured 18 % slower.
#ifndef MDS_STEP_MIN_WAVES
#define MDS_STEP_MIN_WAVES 4
#endif

template <typename S, typename T> __device__ __forceinline__ T ldp(const S* __restrict__ p, size_t i) { return (T)p[i]; }
template <typename S, typename T> __device__ __forceinline__ void stp(S* __restrict__ p, size_t i, T v) { p[i] = (S)v; }

// 4 consecutive storage elements (one drone's action / rpm / u row)
template <typename S, typename T> __device__ __forceinline__ void load4(const S* __restrict__ p, T out[4]) {
  struct alignas(4 * sizeof(S)) V {
    S v[4];
  };
  const V x = *reinterpret_cast<const V*>(p);
  for (int k = 0; k < 4; ++k) out[k] = (T)x.v[k];
}
template <typename S, typename T> __device__ __forceinline__ void store4(S* __restrict__ p, const T in[4]) {
  struct alignas(4 * sizeof(S)) V {
    S v[4];
  };
  V x;
  for (int k = 0; k < 4; ++k) x.v[k] = (S)in[k];
  *reinterpret_cast<V*>(p) = x;
}

// element index of state component k (0..12) of drone i in the packed layout
__host__ __device__ __forceinline__ size_t sidx(int k, size_t i, size_t ld) {
  return k < 12 ? (size_t)(k >> 2) * 4 * ld + 4 * i + (k & 3) : 12 * ld + i;
}
// element index of Lemniscate parameter k (a, omega, cx, cy, cz, yaw_rate, phase_shift) of drone i
__host__ __device__ __forceinline__ size_t lidx(int k, size_t i, size_t ld) {
  switch (k) {
    case 0: return 4 * i;
    case 1: return 4 * i + 1;
    case 5: return 4 * i + 2;
    case 6: return 4 * i + 3;
    case 2: return 4 * ld + 2 * i;
    case 3: return 4 * ld + 2 * i + 1;
    default: return 6 * ld + i;
  }
}

template <typename S, typename T> __device__ __forceinline__ void load_state(const S* __restrict__ st, size_t ld, size_t i, State<T>& s) {
  T a[4], b[4], c[4];
  load4<S, T>(st + 4 * i, a);
  load4<S, T>(st + 4 * ld + 4 * i, b);
  load4<S, T>(st + 8 * ld + 4 * i, c);
  s.p = {a[0], a[1], a[2]};
  s.q[0] = a[3]; s.q[1] = b[0]; s.q[2] = b[1]; s.q[3] = b[2];
  s.v = {b[3], c[0], c[1]};
  s.w = {c[2], c[3], ldp<S, T>(st + 12 * ld, i)};
}
template <typename S, typename T> __device__ __forceinline__ void store_state(S* __restrict__ st, size_t ld, size_t i, const State<T>& s) {
  const T a[4] = {s.p.x, s.p.y, s.p.z, s.q[0]}, b[4] = {s.q[1], s.q[2], s.q[3], s.v.x}, c[4] = {s.v.y, s.v.z, s.w.x, s.w.y};
  store4<S, T>(st + 4 * i, a);
  store4<S, T>(st + 4 * ld + 4 * i, b);
  store4<S, T>(st + 8 * ld + 4 * i, c);
  stp<S, T>(st + 12 * ld, i, s.w.z);
}

// the state as the next kernel will load it: every component through the storage type (a no-op unless S is narrower than T)
template <typename T, typename S> __device__ __forceinline__ State<T> state_as_stored(const State<T>& s) {
  State<T> o;
  o.p = {(T)(S)s.p.x, (T)(S)s.p.y, (T)(S)s.p.z};
  for (int k = 0; k < 4; ++k) o.q[k] = (T)(S)s.q[k];
  o.v = {(T)(S)s.v.x, (T)(S)s.v.y, (T)(S)s.v.z};
  o.w = {(T)(S)s.w.x, (T)(S)s.w.y, (T)(S)s.w.z};
  return o;
}

// Residual storage of the compensated dtype (MDS_F32C): the BODY RATES only -- one 4-wide group per drone, (r_wx, r_wy, r_wz, 0) at
// lo[4 i ..].  Round 2 kept a residual for each of the 13 components (+104 B per drone-step); measured on the device templates
// (tests/emul, 256 drones, open loop, 240 Hz x 1000 steps, max abs state error against the float64 oracle, five seeds): plain fp32
// 1.0-1.4e-5; residuals of q only 1.1e-5, p and q 9.8e-6, w only 4.6-6.2e-6, p and w 3.2-4.8e-6, all thirteen 2.5-3.3e-6.  The
// rounding of the stored RATE (6e-8 of ~1 rad/s per step, a random walk) is what tilts the thrust; the quaternion's own stored
// rounding is second order.  So the dtype keeps the three rate residuals (+32 B per drone-step: one 16-byte load and store) and
// runs the step's accumulations compensated in registers (all 13, across the substeps of a control step); what the other ten
// residuals hold at the end of the step is dropped.
__host__ __device__ __forceinline__ size_t ridx(int k, size_t i) { return 4 * i + (size_t)(k - 10); }   // k = 10, 11, 12 (w)
template <typename S, typename T> __device__ __forceinline__ void load_resid(const S* __restrict__ lo, size_t ld, size_t i, Resid<T>& r) {
  (void)ld;
  T a[4];
  load4<S, T>(lo + 4 * i, a);
  resid_zero(r);
  r.w = {a[0], a[1], a[2]};
}
template <typename S, typename T> __device__ __forceinline__ void store_resid(S* __restrict__ lo, size_t ld, size_t i, const Resid<T>& r) {
  (void)ld;
  const T a[4] = {r.w.x, r.w.y, r.w.z, T(0)};
  store4<S, T>(lo + 4 * i, a);
}
// one control step of the rigid body, plain or with compensated accumulation
template <typename T, bool RK4, bool DRAG, bool COMP>
__device__ __forceinline__ void aviary_step_any(const Consts<T>& c, State<T>& s, Resid<T>& r, const T action[4], T rpm_prev[4], T clipped[4]) {
  if (COMP) aviary_step_comp<T, RK4, DRAG>(c, s, r, action, rpm_prev, clipped);
  else aviary_step<T, RK4, DRAG>(c, s, action, rpm_prev, clipped);
}

// Observation packing: each lane owns one 20-element row; the wave's rows form one
// contiguous span of the caller's [n,20] array.  Rows go to LDS (ds_write_b128, conflict
// free at the 80-byte fp32 row stride), then lane l stores 16-byte chunk (it*64 + l).
constexpr int kObsDim = 20;

typedef unsigned int v4u_t __attribute__((ext_vector_type(4)));
// (KEEP is a template argument on purpose: as a run-time flag the two stores sit in the arms of one branch, the optimiser merges them into a single
// store and drops the non-temporal hint -- measured: form 1 15.6 -> 16.9 us per step, the 4 M-drone shard 135 -> 175, C5 8.4 -> 9.7)
template <bool KEEP> __device__ __forceinline__ void store_chunk(v4u_t v, v4u_t* __restrict__ dst) {
#if defined(MDS_TUNE_OBS_PLAIN_ALL)
  *dst = v;
#else
  if (KEEP) *dst = v;
  else __builtin_nontemporal_store(v, dst);          // write-once stream: non-temporal (measured +3..6 % on MI355X vs default-policy stores)
#endif
}
// KEEP: default-policy stores instead of non-temporal ones -- for a destination that is REWRITTEN every control step
// (the whole-rollout kernels with obs_every_step: the same [n, 20] array, 42 MB at config 3's size): the lines stay in the L2 / Infinity
// Cache between the steps of a launch instead of going out to HBM each time (measured, C3, 2000 steps: 10.5 -> 8.6 us per control step).
template <typename S, typename T, bool KEEP = false>
__device__ __forceinline__ void write_obs_rows(unsigned char* __restrict__ lds_block, S* __restrict__ obs, int n, int i,
                                               bool valid, const T o[kObsDim]) {
  constexpr int kRowBytes = kObsDim * (int)sizeof(S);           // 80 / 160 / 40
  constexpr int kUnit = (kRowBytes % 16 == 0) ? 16 : 8;          // widest aligned LDS store per row
#if defined(MDS_TUNE_OBS_DIRECT)   // tuning build: each lane stores its own row (strided 16-byte stores, no LDS)
  if (valid) {
    alignas(16) S row[kObsDim];
    for (int k = 0; k < kObsDim; ++k) row[k] = (S)o[k];
    unsigned char* dst = reinterpret_cast<unsigned char*>(obs) + (size_t)i * kRowBytes;
    for (int k = 0; k < kRowBytes / kUnit; ++k) {
      if (kUnit == 16) reinterpret_cast<uint4*>(dst)[k] = reinterpret_cast<const uint4*>(row)[k];
      else reinterpret_cast<uint2*>(dst)[k] = reinterpret_cast<const uint2*>(row)[k];
    }
  }
  return;
#endif
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  unsigned char* lds_wave = lds_block + wave * (kWave * kRowBytes);
  if (valid) {
    alignas(16) S row[kObsDim];
    for (int k = 0; k < kObsDim; ++k) row[k] = (S)o[k];
    unsigned char* dst = lds_wave + lane * kRowBytes;
    if (kUnit == 16) {
      for (int k = 0; k < kRowBytes / 16; ++k) reinterpret_cast<uint4*>(dst)[k] = reinterpret_cast<const uint4*>(row)[k];
    } else {
      for (int k = 0; k < kRowBytes / 8; ++k) reinterpret_cast<uint2*>(dst)[k] = reinterpret_cast<const uint2*>(row)[k];
    }
  }
  // each wave stages and drains its own LDS slice: the LDS unit executes one wave's
  // instructions in order, so a wave-scope fence (no s_barrier) orders write -> read
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int wave_base = i - lane;                                // first drone of this wave
  const int rows = min(kWave, n - wave_base);                    // <= 0 for fully invalid waves
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  constexpr int kIters = (kWave * kRowBytes + kWave * 16 - 1) / (kWave * 16);
  constexpr int kFull = (kWave * kRowBytes) / (kWave * 16);       // iterations in which all 64 lanes have a chunk of a full wave's rows
#if !defined(MDS_TUNE_OBS_NO_FAST)
  if (rows >= kWave) {
    // a full wave (every wave but the shard's last): all LDS reads in flight at once, then the stores -- no per-chunk bounds test, one
    // LDS round trip instead of kIters serial ones
    unsigned char* gdst = reinterpret_cast<unsigned char*>(obs) + (size_t)wave_base * kRowBytes;
    constexpr int kGroup = 5;                                    // chunks in flight per lane (20 VGPRs)
#pragma unroll
    for (int g = 0; g < kIters; g += kGroup) {
      v4u tmp[kGroup];
#pragma unroll
      for (int it = g; it < kIters && it < g + kGroup; ++it)
        if (it < kFull || lane * 16 + 16 <= kWave * kRowBytes - it * kWave * 16) tmp[it - g] = *reinterpret_cast<const v4u*>(lds_wave + (it * kWave + lane) * 16);
#pragma unroll
      for (int it = g; it < kIters && it < g + kGroup; ++it)
        if (it < kFull || lane * 16 + 16 <= kWave * kRowBytes - it * kWave * 16)
            store_chunk<KEEP>(tmp[it - g], reinterpret_cast<v4u*>(gdst + (it * kWave + lane) * 16));
    }
  } else
#endif
  if (rows > 0) {
    const int bytes = rows * kRowBytes;                          // multiple of 8; of 16 unless half with odd rows
    unsigned char* gdst = reinterpret_cast<unsigned char*>(obs) + (size_t)wave_base * kRowBytes;
    for (int it = 0; it < kIters; ++it) {
      const int off = (it * kWave + lane) * 16;
      if (off + 16 <= bytes) {
        // write-once stream: non-temporal (measured +3..6 % on MI355X vs default-policy stores)
        __builtin_nontemporal_store(*reinterpret_cast<const v4u*>(lds_wave + off), reinterpret_cast<v4u*>(gdst + off));
      } else if (off + 8 <= bytes) {                             // 8-byte tail (fp16 rows, odd row count)
        *reinterpret_cast<uint2*>(gdst + off) = *reinterpret_cast<const uint2*>(lds_wave + off);
      }
    }
  }
  // the slice is reused by the wave's next batch (persistent kernels): reads before the next writes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------
// [UPSTREAM] BaseAviary.step for every drone (a1-a4)
// ------------------------------------------------------------------------------------
template <typename T, typename S, bool HAS_OBS, bool RK4, bool DRAG, bool COMP = false>
__global__ __launch_bounds__(kBlock, (sizeof(T) == 4 && !RK4 && !COMP) ? MDS_STEP_MIN_WAVES : 1) void k_step(const Consts<T> c, const int n, const size_t ld, S* __restrict__ state,
                                                 const T* __restrict__ origin, T* __restrict__ last_rpm,
                                                 const S* __restrict__ action, S* __restrict__ obs, const int batch0,
                                                 S* __restrict__ state_lo = nullptr) {
  __shared__ __align__(16) unsigned char lds[HAS_OBS ? (kBlock * kObsDim * sizeof(S)) : 16];
  const int i = (batch0 + blockIdx.x) * kBlock + threadIdx.x;
  const bool valid = i < n;
  T o[kObsDim];
  if (valid) {
    State<T> s;
    Resid<T> r;
    load_state<S, T>(state, ld, i, s);
    if (COMP) load_resid<S, T>(state_lo, ld, i, r);
    T act[4], prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4];
    load4<S, T>(action + (size_t)i * 4, act);
    if (DRAG)
      for (int k = 0; k < 4; ++k) prev[k] = last_rpm[k * ld + i];
    aviary_step_any<T, RK4, DRAG, COMP>(c, s, r, act, prev, clipped);
    store_state<S, T>(state, ld, i, s);
    if (COMP) store_resid<S, T>(state_lo, ld, i, r);
    if (DRAG || last_rpm)
      for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = clipped[k];
    if (HAS_OBS) {
      const V3<T> org = {origin[i], origin[ld + i], origin[2 * ld + i]};
      pack_obs(s, org, clipped, o);
    }
  }
  if (HAS_OBS) write_obs_rows<S, T>(lds, obs, n, i, valid, o);
}

// n_steps of k_step in ONE launch with the state in registers: step k applies action set (a0 + k) % n_sets of the table and writes
// its observation into slot (s0 + k) % n_slots of the log ring.  Per drone-step only the action (4 values) is read and the
// observation (20 values) written; the 13-value state crosses HBM once per launch instead of twice per step.
template <typename T, typename S, bool RK4, bool DRAG>
__global__ __launch_bounds__(kBlock) void k_rollout_step(const Consts<T> c, const int n, const size_t ld, S* __restrict__ state,
                                                         const T* __restrict__ origin, T* __restrict__ last_rpm,
                                                         const S* __restrict__ actions, int a0, const int n_sets,
                                                         S* __restrict__ obs_log, int s0, const int n_slots, const int n_steps,
                                                         S* __restrict__ state_lo = nullptr) {
  __shared__ __align__(16) unsigned char lds[kBlock * kObsDim * sizeof(S)];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  State<T> s;
  Resid<T> r;
  resid_zero(r);
  if (valid && state_lo) load_resid<S, T>(state_lo, ld, i, r);
  V3<T> org = {T(0), T(0), T(0)};
  T prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4] = {T(0), T(0), T(0), T(0)};
  if (valid) {
    load_state<S, T>(state, ld, i, s);
    org = {origin[i], origin[ld + i], origin[2 * ld + i]};
    if (DRAG)
      for (int k = 0; k < 4; ++k) prev[k] = last_rpm[k * ld + i];
  }
  for (int k = 0; k < n_steps; ++k) {
    T o[kObsDim];
    if (valid) {
      T act[4];
      load4<S, T>(actions + ((size_t)a0 * n + i) * 4, act);
      if (state_lo) aviary_step_comp<T, RK4, DRAG>(c, s, r, act, prev, clipped);
      else aviary_step<T, RK4, DRAG>(c, s, act, prev, clipped);
      if (obs_log != nullptr) pack_obs(s, org, clipped, o);
    }
    if (obs_log != nullptr) write_obs_rows<S, T>(lds, obs_log + (size_t)s0 * n * kObsDim, n, i, valid, o);
    a0 = a0 + 1 == n_sets ? 0 : a0 + 1;
    s0 = s0 + 1 == n_slots ? 0 : s0 + 1;
  }
  if (valid) {
    store_state<S, T>(state, ld, i, s);
    if (state_lo) store_resid<S, T>(state_lo, ld, i, r);
    if ((DRAG || last_rpm) && n_steps > 0)
      for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = clipped[k];
  }
}

// [UPSTREAM] BaseAviary.step with ground effect and / or downwash: ONE physics substep per launch (upstream refreshes the
// kinematic information of every drone between substeps; the downwash on a drone depends on its env-mates' positions), reading
// state_in and writing state_out (double-buffered: env-mates are read while they are being updated).  Explicit Euler only.
// drag_from_action: substeps after the first take the _drag term from this step's clipped action (as aviary_step does).
// one explicit-Euler substep of drone i with the environment terms: rotor wrench of the clipped RPM, [UPSTREAM] _groundEffect on the
// drone's own propellers, _downwash from every env-mate above (positions read from state_in: the state BEFORE this substep),
// _drag from `prev`.  s holds drone i's state_in row on entry, the stepped state on return.
template <typename T, typename S, bool DRAG>
__device__ __forceinline__ void env_substep(const Consts<T>& c, const EnvFx<T>& fx, const int D, const size_t ld, const int i,
                                            const S* __restrict__ state_in, const T* __restrict__ origin, const V3<T> org,
                                            State<T>& s, const T clipped[4], const T prev[4]) {
  T drag_s = T(0);
  if (DRAG) drag_s = T(0.10471975511965977462) * ((prev[0] + prev[1]) + (prev[2] + prev[3]));
  T thrust;
  V3<T> tau;
  rotor_wrench(c, clipped, &thrust, &tau);
  if (fx.gnd) ground_effect(c, fx, s, s.p.z + org.z, clipped, &thrust, &tau);
  if (fx.dw) {
    const V3<T> me = {s.p.x + org.x, s.p.y + org.y, s.p.z + org.z};
    const int e0 = (i / D) * D;
    T f = T(0);
    for (int j = e0; j < e0 + D; ++j) {
      if (j == i) continue;
      T a4[4];
      load4<S, T>(state_in + 4 * (size_t)j, a4);                                // (px, py, pz, qx) of env-mate j, before this substep
      f += downwash_pair(fx, me, V3<T>{a4[0] + origin[j], a4[1] + origin[ld + j], a4[2] + origin[2 * ld + j]});
    }
    thrust += f;
  }
  step_euler_wrench<T, DRAG>(c, s, thrust, tau, drag_s);
}

template <typename T, typename S, bool DRAG>
__global__ __launch_bounds__(kBlock) void k_step_env(const Consts<T> c, const EnvFx<T> fx, const int n, const size_t ld, const int D,
                                                     const S* __restrict__ state_in, S* __restrict__ state_out,
                                                     const T* __restrict__ origin, T* __restrict__ last_rpm,
                                                     const S* __restrict__ action, S* __restrict__ obs, const int drag_from_action,
                                                     const int store_rpm) {
  __shared__ __align__(16) unsigned char lds[kBlock * kObsDim * sizeof(S)];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  T o[kObsDim];
  if (valid) {
    State<T> s;
    load_state<S, T>(state_in, ld, i, s);
    const V3<T> org = {origin[i], origin[ld + i], origin[2 * ld + i]};
    T act[4], clipped[4], prev[4] = {T(0), T(0), T(0), T(0)};
    load4<S, T>(action + (size_t)i * 4, act);
    for (int k = 0; k < 4; ++k) clipped[k] = m_clamp(act[k], T(0), c.max_rpm);
    if (DRAG)
      for (int k = 0; k < 4; ++k) prev[k] = drag_from_action ? clipped[k] : last_rpm[k * ld + i];
    env_substep<T, S, DRAG>(c, fx, D, ld, i, state_in, origin, org, s, clipped, prev);
    store_state<S, T>(state_out, ld, i, s);
    if (store_rpm && (DRAG || last_rpm))
      for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = clipped[k];
    if (obs) pack_obs(s, org, clipped, o);
  }
  if (obs) write_obs_rows<S, T>(lds, obs, n, i, valid, o);
}

// The controller kernels' first physics substep under ground effect / downwash: trajectory sample -> controller -> the substep
// above, reading state_in and writing state_out like k_step_env (env-mates are read while they are being updated).  CTRL 0:
// GeometricControl; 1: the 12-state LQRController (K read from its device copy); 2 / 3: a given input u_in [n,4] through the
// ThrustOmega / YankOmega low level (the CBF loops: u_safe + thrust_offset).  The unclipped action goes to act_buf when further
// substeps follow (k_step_env replays it); obs / action_out may be NULL.
template <typename T, typename S, bool DRAG, int CTRL>
__global__ __launch_bounds__(kBlock) void k_step_ctrl_env(const Consts<T> c, const EnvFx<T> fx, const void* __restrict__ Kp, const int n,
                                                          const size_t ld, const int D, const double t, const int traj_mode,
                                                          const S* __restrict__ state_in, S* __restrict__ state_out,
                                                          const T* __restrict__ origin, const T* __restrict__ lem, const SegTable segs,
                                                          const int* __restrict__ tinfo, T* __restrict__ last_rpm, T* __restrict__ ll,
                                                          const S* __restrict__ u_in, const T ctrl_dt, const T thrust_offset,
                                                          S* __restrict__ act_buf, S* __restrict__ obs, S* __restrict__ action_out,
                                                          const int store_rpm) {
  __shared__ __align__(16) unsigned char lds[kBlock * kObsDim * sizeof(S)];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  T o[kObsDim];
  if (valid) {
    State<T> s;
    load_state<S, T>(state_in, ld, i, s);
    const V3<T> org = {origin[i], origin[ld + i], origin[2 * ld + i]};
    T act[4], clipped[4], prev[4] = {T(0), T(0), T(0), T(0)};
    if (CTRL <= 1) {
      Desired<T> des;
      if (traj_mode == 1) {
        const LemniscateParams<T> P = {lem[lidx(0, i, ld)], lem[lidx(1, i, ld)], lem[lidx(2, i, ld)], lem[lidx(3, i, ld)], lem[lidx(4, i, ld)],
                                       lem[lidx(5, i, ld)], lem[lidx(6, i, ld)]};
        des = lemniscate_local(P, t);               // local frame: mds_set_lemniscate made the origin the trajectory centre
      } else {
        des = TrajLocal<T>::eval(segs, traj_info(tinfo, i), t, org);
      }
      T u[4];
      if (CTRL == 0) {
        const M3<T> R = quat_to_rot(s.q);
        geometric_control<T>(c, s.p - des.p, R, s.v, mul(R, s.w), des, u, nullptr);
      } else {
        lqr12_control<T>(c, *static_cast<const Lqr12Gain<T>*>(Kp), euler_from_quat(s.q), quat_rotate(s.q, s.w), s.v, s.p - des.p, des.v,
                         des.yaw, des.yaw_rate, u);
      }
      input_to_action(c, u, act);
    } else {
      T u[4];
      load4<S, T>(u_in + (size_t)i * 4, u);
      u[0] += thrust_offset;
      LowLevelState<T> L;
      L.last_omega = {ll[0 * ld + i], ll[1 * ld + i], ll[2 * ld + i]};
      L.integral = {ll[3 * ld + i], ll[4 * ld + i], ll[5 * ld + i]};
      if (CTRL == 3) {   // obs still holds the previous step's row here (read and later rewritten by the same wave)
        T rpm_prev[4];
        load4<S, T>(obs + (size_t)i * kObsDim + 16, rpm_prev);
        yank_omega_control(c, ctrl_dt, u, rpm_prev, s.w, L, act);
      } else {
        thrust_omega_control(c, ctrl_dt, u, s.w, L, act);
      }
      ll[0 * ld + i] = L.last_omega.x; ll[1 * ld + i] = L.last_omega.y; ll[2 * ld + i] = L.last_omega.z;
      ll[3 * ld + i] = L.integral.x; ll[4 * ld + i] = L.integral.y; ll[5 * ld + i] = L.integral.z;
    }
    for (int k = 0; k < 4; ++k) clipped[k] = m_clamp(act[k], T(0), c.max_rpm);
    if (DRAG)
      for (int k = 0; k < 4; ++k) prev[k] = last_rpm[k * ld + i];
    env_substep<T, S, DRAG>(c, fx, D, ld, i, state_in, origin, org, s, clipped, prev);
    store_state<S, T>(state_out, ld, i, s);
    if (store_rpm && (DRAG || last_rpm))
      for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = clipped[k];
    if (act_buf) store4<S, T>(act_buf + (size_t)i * 4, act);
    if (action_out) store4<S, T>(action_out + (size_t)i * 4, act);
    if (obs && store_rpm) pack_obs(s, org, clipped, o);
  }
  if (obs && store_rpm) write_obs_rows<S, T>(lds, obs, n, i, valid, o);
}

// ------------------------------------------------------------------------------------
// fused trajectory + geometric controller + mixer + physics step (a10, a7-a9, a1-a4)
// ------------------------------------------------------------------------------------
template <typename T> struct GeoIn {
  State<T> s;
  LemniscateParams<T> P;
  Resid<T> r;          // compensated storage only
};
template <typename T, typename S>
__device__ __forceinline__ void load_geo_in(const S* __restrict__ state, const T* __restrict__ lem, size_t ld, size_t i, GeoIn<T>& in) {
  load_state<S, T>(state, ld, i, in.s);
  T a[4];
  load4<T, T>(lem + 4 * i, a);
  struct alignas(2 * sizeof(T)) V2 {
    T v[2];
  };
  const V2 c = *reinterpret_cast<const V2*>(lem + 4 * ld + 2 * i);
  in.P.a = a[0];
  in.P.omega = a[1];
  in.P.yaw_rate = a[2];
  in.P.phase_shift = a[3];
  in.P.cx = c.v[0];
  in.P.cy = c.v[1];
  in.P.cz = lem[6 * ld + i];
}

// One batch row of the fused kernel: controller + physics on registers already loaded.
template <typename T, typename S, bool HAS_OBS, bool HAS_ACT, bool RK4, bool DRAG, bool COMP = false>
__device__ __forceinline__ void geo_process(const Consts<T>& c, const int n, const size_t ld, const double t, const int i,
                                            GeoIn<T>& in, S* __restrict__ state, T* __restrict__ last_rpm,
                                            S* __restrict__ obs, S* __restrict__ action_out, unsigned char* lds,
                                            S* __restrict__ state_lo = nullptr) {
  const bool valid = i < n;
  T o[kObsDim];
  if (valid) {
    State<T>& s = in.s;
    const LemniscateParams<T>& P = in.P;
    T prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4], act[4];
    if (DRAG)
      for (int k = 0; k < 4; ++k) prev[k] = last_rpm[k * ld + i];
#if defined(MDS_TUNE_NOCOMPUTE)   // timing-only build: memory traffic of the kernel without its arithmetic
    for (int k = 0; k < 4; ++k) act[k] = clipped[k] = P.a + P.omega + P.yaw_rate + P.phase_shift + T(k);
    s.p.x += T(1);
#else
    {
      const Desired<T> des = lemniscate_local(P, t);
      const M3<T> R = quat_to_rot(s.q);
      const V3<T> ang_v = mul(R, s.w);
      T u[4];
      geometric_control<T>(c, s.p - des.p, R, s.v, ang_v, des, u, nullptr);
      input_to_action(c, u, act);
    }
    aviary_step_any<T, RK4, DRAG, COMP>(c, s, in.r, act, prev, clipped);
#endif
    if (DRAG || last_rpm)
      for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = clipped[k];
    if (HAS_ACT) store4<S, T>(action_out + (size_t)i * 4, act);
#if defined(MDS_TUNE_NOCOMPUTE)
    if (HAS_OBS) {
      o[0] = s.p.x + P.cx; o[1] = s.p.y + P.cy; o[2] = s.p.z + P.cz; o[3] = s.q[0]; o[4] = s.q[1]; o[5] = s.q[2]; o[6] = s.q[3];
      o[7] = s.q[0]; o[8] = s.q[1]; o[9] = s.q[2]; o[10] = s.v.x; o[11] = s.v.y; o[12] = s.v.z; o[13] = s.w.x; o[14] = s.w.y; o[15] = s.w.z;
      for (int k = 0; k < 4; ++k) o[16 + k] = clipped[k];
    }
#else
    if (HAS_OBS) pack_obs(s, V3<T>{P.cx, P.cy, P.cz}, clipped, o);
#endif
  }
  // observation rows go out first (their LDS round trip must not sit behind the state stores)
  if (HAS_OBS) write_obs_rows<S, T>(lds, obs, n, i, valid, o);
  if (valid) {
    store_state<S, T>(state, ld, i, in.s);
    if (COMP) store_resid<S, T>(state_lo, ld, i, in.r);
  }
}

// One batch of 256 drones per workgroup, inputs loaded straight into registers; every dtype / integrator / physics
// combination is an instantiation of this kernel (the fp32 / Euler / DYN one is the bench's hot kernel).
#ifndef MDS_GEOSIMPLE_MIN_WAVES
#define MDS_GEOSIMPLE_MIN_WAVES 1
#endif
template <typename T, typename S, bool HAS_OBS, bool HAS_ACT, bool RK4, bool DRAG, bool COMP = false>
__global__ __launch_bounds__(kBlock, (sizeof(T) == 4 && !RK4) ? MDS_GEOSIMPLE_MIN_WAVES : 1) void k_step_geometric(const Consts<T> c, const int n, const size_t ld, const double t,
                                                           S* __restrict__ state, const T* __restrict__ lem,
                                                           T* __restrict__ last_rpm, S* __restrict__ obs,
                                                           S* __restrict__ action_out, const int batch0,
                                                           S* __restrict__ state_lo = nullptr) {
  __shared__ __align__(16) unsigned char lds[HAS_OBS ? (kBlock * kObsDim * sizeof(S)) : 16];
  // batch0: first 256-drone batch of this launch (a rollout may step the two halves of the shard on two streams)
  const int i = (batch0 + blockIdx.x) * kBlock + threadIdx.x;
  GeoIn<T> in;
  if (i < n) {
    load_geo_in<T, S>(state, lem, ld, i, in);
    if (COMP) load_resid<S, T>(state_lo, ld, i, in.r);
  }
  geo_process<T, S, HAS_OBS, HAS_ACT, RK4, DRAG, COMP>(c, n, ld, t, i, in, state, last_rpm, obs, action_out, lds, state_lo);
}

// ------------------------------------------------------------------------------------
// General-trajectory form of the fused step: the drone's desired state comes from its segment
// table (mds_traj.hpp TrajLocal: phases and absolute positions in double, the rest in T), relative to the
// drone's local-frame origin.  obs / action_out may be NULL.
// ------------------------------------------------------------------------------------
template <typename T, typename S, bool RK4, bool DRAG, bool COMP = false>
__global__ __launch_bounds__(kBlock) void k_step_traj(const Consts<T> c, const int n, const size_t ld, const double t,
                                                      S* __restrict__ state, const T* __restrict__ origin,
                                                      const SegTable segs, const int* __restrict__ tinfo,
                                                      T* __restrict__ last_rpm, S* __restrict__ obs, S* __restrict__ action_out,
                                                      const int batch0, S* __restrict__ state_lo = nullptr) {
  __shared__ __align__(16) unsigned char lds[kBlock * kObsDim * sizeof(S)];
  const int i = (batch0 + blockIdx.x) * kBlock + threadIdx.x;
  const bool valid = i < n;
  T o[kObsDim];
  State<T> s;
  Resid<T> r;
  if (valid) {
    load_state<S, T>(state, ld, i, s);
    if (COMP) load_resid<S, T>(state_lo, ld, i, r);
    const V3<T> org = {origin[i], origin[ld + i], origin[2 * ld + i]};
    const Desired<T> des = TrajLocal<T>::eval(segs, traj_info(tinfo, i), t, org);
    T prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4], act[4];
    if (DRAG)
      for (int k = 0; k < 4; ++k) prev[k] = last_rpm[k * ld + i];
    {
      const M3<T> R = quat_to_rot(s.q);
      const V3<T> ang_v = mul(R, s.w);
      T u[4];
      geometric_control<T>(c, s.p - des.p, R, s.v, ang_v, des, u, nullptr);
      input_to_action(c, u, act);
    }
    aviary_step_any<T, RK4, DRAG, COMP>(c, s, r, act, prev, clipped);
    if (DRAG || last_rpm)
      for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = clipped[k];
    if (action_out) store4<S, T>(action_out + (size_t)i * 4, act);
    if (obs) pack_obs(s, org, clipped, o);
  }
  if (obs) write_obs_rows<S, T>(lds, obs, n, i, valid, o);
  if (valid) {
    store_state<S, T>(state, ld, i, s);
    if (COMP) store_resid<S, T>(state_lo, ld, i, r);
  }
}

// The same fused step with the reference's 12-state LQR (control/lqr/lqr_controller.py) as the controller: the default
// ('lqr') branch of simulations/EnvGeometric.py do_control.  Desired state from the Lemniscate planes (traj_mode 1, local
// frame) or the segment tables (traj_mode 2).  obs / action_out may be NULL.
template <typename T, typename S, bool RK4, bool DRAG>
__global__ __launch_bounds__(kBlock) void k_step_lqr(const Consts<T> c, const Lqr12Gain<T> K, const int n, const size_t ld, const double t,
                                                     const int traj_mode, S* __restrict__ state, const T* __restrict__ origin,
                                                     const T* __restrict__ lem, const SegTable segs,
                                                     const int* __restrict__ tinfo, T* __restrict__ last_rpm, S* __restrict__ obs,
                                                     S* __restrict__ action_out, S* __restrict__ state_lo = nullptr) {
  __shared__ __align__(16) unsigned char lds[kBlock * kObsDim * sizeof(S)];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  T o[kObsDim];
  State<T> s;
  Resid<T> r;                  // state_lo != NULL (MDS_F32C handles, a uniform branch): compensated accumulation
  if (valid) {
    load_state<S, T>(state, ld, i, s);
    if (state_lo) load_resid<S, T>(state_lo, ld, i, r);
    const V3<T> org = {origin[i], origin[ld + i], origin[2 * ld + i]};
    Desired<T> des;
    if (traj_mode == 1) {
      GeoIn<T> in;
      load_geo_in<T, S>(state, lem, ld, i, in);
      des = lemniscate_local(in.P, t);
    } else {
      des = TrajLocal<T>::eval(segs, traj_info(tinfo, i), t, org);
    }
    T prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4], act[4], u[4];
    if (DRAG)
      for (int k = 0; k < 4; ++k) prev[k] = last_rpm[k * ld + i];
    lqr12_control<T>(c, K, euler_from_quat(s.q), quat_rotate(s.q, s.w), s.v, s.p - des.p, des.v, des.yaw, des.yaw_rate, u);
    input_to_action(c, u, act);
    if (state_lo) aviary_step_comp<T, RK4, DRAG>(c, s, r, act, prev, clipped);
    else aviary_step<T, RK4, DRAG>(c, s, act, prev, clipped);
    if (DRAG || last_rpm)
      for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = clipped[k];
    if (action_out) store4<S, T>(action_out + (size_t)i * 4, act);
    if (obs) pack_obs(s, org, clipped, o);
  }
  if (obs) write_obs_rows<S, T>(lds, obs, n, i, valid, o);
  if (valid) {
    store_state<S, T>(state, ld, i, s);
    if (state_lo) store_resid<S, T>(state_lo, ld, i, r);
  }
}

// LQRController.compute(obs) (lqr_controller.py:83-113): obs [n,20], des [n,11] (pos, vel, -, yaw, omega) -> u [n,4], action [n,4]
template <typename T, typename S>
__global__ void k_lqr12_compute(const Consts<T> c, const Lqr12Gain<T> K, const int n, const S* __restrict__ obs,
                                const S* __restrict__ des, S* __restrict__ u_out, S* __restrict__ act_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const S* o = obs + (size_t)i * 20;
  const S* d = des + (size_t)i * 11;
  T u[4], act[4];
  const V3<T> perr = {(T)o[0] - (T)d[0], (T)o[1] - (T)d[1], (T)o[2] - (T)d[2]};
  lqr12_control<T>(c, K, V3<T>{(T)o[7], (T)o[8], (T)o[9]}, V3<T>{(T)o[13], (T)o[14], (T)o[15]}, V3<T>{(T)o[10], (T)o[11], (T)o[12]}, perr,
                   V3<T>{(T)d[3], (T)d[4], (T)d[5]}, (T)d[9], (T)d[10], u);
  input_to_action(c, u, act);
  if (u_out) store4<S, T>(u_out + (size_t)i * 4, u);
  if (act_out) store4<S, T>(act_out + (size_t)i * 4, act);
}

// Trajectory.__call__(t) for every drone from the segment tables: des [n,11] world frame
template <typename S>
__global__ void k_traj_eval(const int n, const double t, const SegTable segs, const int* __restrict__ tinfo,
                            S* __restrict__ des) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d11[11];
  traj_eval(segs, traj_info(tinfo, i), t, d11);
  for (int k = 0; k < 11; ++k) des[(size_t)i * 11 + k] = (S)d11[k];
}

// ------------------------------------------------------------------------------------
// Whole-rollout form of the same loop (simulations/EnvGeometric.py:434-473): n_steps control
// steps in ONE launch.  State and trajectory parameters stay in registers between steps
// (no per-step 52+28+52 bytes of state traffic), each step's observation rows are streamed
// to obs_log[k] (the reference's `self.observations.append(obs)`, np.save'd as [T,D,20],
// EnvGeometric.py:471,553) through the same per-wave LDS transposition.  t advances in
// double exactly like the host loop (t += CTRL_TIMESTEP).
// ------------------------------------------------------------------------------------
// CTRL 0: GeometricControl; CTRL 1: the 12-state LQRController (K is only read then); CTRL 2 / 3: LQROmegaController +
// ThrustOmegaController / LQRYankOmegaController + YankOmegaController (Kp then points at an LqrGain / LqrYoGain; the low level's
// PID memory `ll` stays in registers for the whole rollout, the yank path's thrust state is the previous step's clipped RPM).
#ifndef MDS_RG_MIN_WAVES
#define MDS_RG_MIN_WAVES 1          // (A/B builds: waves per SIMD requested from the register allocator for the fp32 geometric instantiation)
#endif
template <typename T, typename S, bool RK4, bool DRAG, int CTRL = 0>
__global__ __launch_bounds__(kBlock, (sizeof(T) == 4 && !RK4 && CTRL == 0) ? MDS_RG_MIN_WAVES : 1) void k_rollout_geometric(const Consts<T> c, const void* __restrict__ Kp, const int n, const size_t ld, double t,
                                                              const double ctrl_dt, const int n_steps, S* __restrict__ state,
                                                              const T* __restrict__ lem, T* __restrict__ last_rpm,
                                                              S* __restrict__ obs_log, const size_t log_stride, S* __restrict__ obs_last,
                                                              T* __restrict__ ll = nullptr, const S* __restrict__ obs_prev = nullptr,
                                                              S* __restrict__ state_lo = nullptr) {
  __shared__ __align__(16) unsigned char lds[kBlock * kObsDim * sizeof(S)];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  GeoIn<T> in;
  resid_zero(in.r);            // state_lo != NULL (MDS_F32C handles, a uniform branch): compensated accumulation, all 13 residuals in registers
  if (valid && state_lo) load_resid<S, T>(state_lo, ld, i, in.r);
  T prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4] = {T(0), T(0), T(0), T(0)};
  LowLevelState<T> L;
  L.last_omega = L.integral = {T(0), T(0), T(0)};
  if (valid) {
    load_geo_in<T, S>(state, lem, ld, i, in);
    if (DRAG)
      for (int k = 0; k < 4; ++k) prev[k] = last_rpm[k * ld + i];
    if (CTRL >= 2) {
      L.last_omega = {ll[0 * ld + i], ll[1 * ld + i], ll[2 * ld + i]};
      L.integral = {ll[3 * ld + i], ll[4 * ld + i], ll[5 * ld + i]};
      if (CTRL == 3) load4<S, T>(obs_prev + (size_t)i * kObsDim + 16, clipped);      // calc_z_thrust(obs) of the first step
    }
  }
#if defined(MDS_TUNE_RG)           // timing-only builds: 1 = the arithmetic without the observation rows, 2 = the rows without the arithmetic
  T tune_acc = T(0);
#endif
#if defined(MDS_TUNE_RG_STAGGER)
  for (int r = (int)((blockIdx.x >> MDS_TUNE_RG_STAGGER_SHIFT) & 3); r > 0; --r) __builtin_amdgcn_s_sleep(MDS_TUNE_RG_STAGGER);
#endif
  for (int k = 0; k < n_steps; ++k) {
    T o[kObsDim];
    const bool want = obs_log != nullptr || (obs_last != nullptr && k == n_steps - 1);
#if defined(MDS_TUNE_RG) && MDS_TUNE_RG == 2
    if (valid) {
      for (int j = 0; j < kObsDim; ++j) o[j] = in.s.p.x + T(j) * (T)t;
    }
    if (false) {
      T act[4];
#else
    if (valid) {
      T act[4];
#endif
      {
        const Desired<T> des = lemniscate_local(in.P, t);
        T u[4];
        if (CTRL == 0) {
          const M3<T> R = quat_to_rot(in.s.q);
          const V3<T> ang_v = mul(R, in.s.w);
          geometric_control<T>(c, in.s.p - des.p, R, in.s.v, ang_v, des, u, nullptr);
        } else if (CTRL == 1) {
          lqr12_control<T>(c, *static_cast<const Lqr12Gain<T>*>(Kp), euler_from_quat(in.s.q), quat_rotate(in.s.q, in.s.w), in.s.v, in.s.p - des.p,
                           des.v, des.yaw, des.yaw_rate, u);
        } else if (CTRL == 2) {
          lqr_omega_control<T>(c, *static_cast<const LqrGain<T>*>(Kp), euler_from_quat(in.s.q), in.s.v, in.s.p, des.p, des.v, des.yaw, u);
        } else {
          lqr_yank_omega_control<T>(c, *static_cast<const LqrYoGain<T>*>(Kp), euler_from_quat(in.s.q), clipped, in.s.v, in.s.p, des.p, des.v,
                                    des.yaw, u);
        }
        if (CTRL <= 1) input_to_action(c, u, act);
        else if (CTRL == 2) thrust_omega_control(c, (T)ctrl_dt, u, in.s.w, L, act);
        else yank_omega_control(c, (T)ctrl_dt, u, clipped, in.s.w, L, act);
      }
      if (state_lo) aviary_step_comp<T, RK4, DRAG>(c, in.s, in.r, act, prev, clipped);
      else aviary_step<T, RK4, DRAG>(c, in.s, act, prev, clipped);
      if (want) pack_obs(in.s, V3<T>{in.P.cx, in.P.cy, in.P.cz}, clipped, o);
    }
#if defined(MDS_TUNE_RG) && MDS_TUNE_RG == 1
    if (valid && want)
      for (int j = 0; j < kObsDim; ++j) tune_acc += o[j];
#else
    if (obs_log != nullptr) {
      if (log_stride == 0) write_obs_rows<S, T, true>(lds, obs_log, n, i, valid, o);             // the same rows rewritten every step: keep them cached
      else write_obs_rows<S, T>(lds, obs_log + (size_t)k * log_stride, n, i, valid, o);
    }
    if (obs_last != nullptr && k == n_steps - 1) write_obs_rows<S, T>(lds, obs_last, n, i, valid, o);
#endif
    t += ctrl_dt;
  }
#if defined(MDS_TUNE_RG)
  if (valid && tune_acc == T(12345.678)) in.s.p.x += tune_acc;
#endif
  if (valid) {
    store_state<S, T>(state, ld, i, in.s);
    if (state_lo) store_resid<S, T>(state_lo, ld, i, in.r);
    if (DRAG || (last_rpm && n_steps > 0))
      for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = DRAG ? prev[k] : clipped[k];
    if (CTRL >= 2) {
      ll[0 * ld + i] = L.last_omega.x; ll[1 * ld + i] = L.last_omega.y; ll[2 * ld + i] = L.last_omega.z;
      ll[3 * ld + i] = L.integral.x; ll[4 * ld + i] = L.integral.y; ll[5 * ld + i] = L.integral.z;
    }
  }
}

// The same whole-rollout loop for general trajectories (segment tables, evaluated in double every step like k_step_traj).
template <typename T, typename S, bool RK4, bool DRAG, int CTRL>
__global__ __launch_bounds__(kBlock) void k_rollout_traj(const Consts<T> c, const void* __restrict__ Kp, const int n, const size_t ld, double t,
                                                         const double ctrl_dt, const int n_steps, S* __restrict__ state,
                                                         const T* __restrict__ origin, const SegTable segs,
                                                         const int* __restrict__ tinfo, T* __restrict__ last_rpm,
                                                         S* __restrict__ obs_log, const size_t log_stride, S* __restrict__ obs_last,
                                                         S* __restrict__ state_lo = nullptr) {
  __shared__ __align__(16) unsigned char lds[kBlock * kObsDim * sizeof(S)];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  State<T> s;
  Resid<T> r;
  resid_zero(r);
  if (valid && state_lo) load_resid<S, T>(state_lo, ld, i, r);
  V3<T> org = {T(0), T(0), T(0)};
  TrajInfo ti = {0, 1, 0, 0};
  T prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4] = {T(0), T(0), T(0), T(0)};
  if (valid) {
    load_state<S, T>(state, ld, i, s);
    org = {origin[i], origin[ld + i], origin[2 * ld + i]};
    ti = traj_info(tinfo, i);
    if (DRAG)
      for (int k = 0; k < 4; ++k) prev[k] = last_rpm[k * ld + i];
  }
  for (int k = 0; k < n_steps; ++k) {
    T o[kObsDim];
    const bool want = obs_log != nullptr || (obs_last != nullptr && k == n_steps - 1);
    if (valid) {
      const Desired<T> des = TrajLocal<T>::eval(segs, ti, t, org);
      T u[4], act[4];
      if (CTRL == 0) {
        const M3<T> R = quat_to_rot(s.q);
        geometric_control<T>(c, s.p - des.p, R, s.v, mul(R, s.w), des, u, nullptr);
      } else {
        lqr12_control<T>(c, *static_cast<const Lqr12Gain<T>*>(Kp), euler_from_quat(s.q), quat_rotate(s.q, s.w), s.v, s.p - des.p, des.v, des.yaw,
                         des.yaw_rate, u);
      }
      input_to_action(c, u, act);
      if (state_lo) aviary_step_comp<T, RK4, DRAG>(c, s, r, act, prev, clipped);
      else aviary_step<T, RK4, DRAG>(c, s, act, prev, clipped);
      if (want) pack_obs(s, org, clipped, o);
    }
    if (obs_log != nullptr) {
      if (log_stride == 0) write_obs_rows<S, T, true>(lds, obs_log, n, i, valid, o);             // the same rows rewritten every step: keep them cached
      else write_obs_rows<S, T>(lds, obs_log + (size_t)k * log_stride, n, i, valid, o);
    }
    if (obs_last != nullptr && k == n_steps - 1) write_obs_rows<S, T>(lds, obs_last, n, i, valid, o);
    t += ctrl_dt;
  }
  if (valid) {
    store_state<S, T>(state, ld, i, s);
    if (state_lo) store_resid<S, T>(state_lo, ld, i, r);
    if (DRAG || (last_rpm && n_steps > 0))
      for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = DRAG ? prev[k] : clipped[k];
  }
}

// ------------------------------------------------------------------------------------
// CBF-filtered control step (simulations/CBFTest.py:303-350) = three launches:
//   nominal          per drone : (k_lowlevel_step in its `only` mode, or the tail of the previous step's launch) trajs[j](t), GeometricControl.compute(return_omegas) ->
//                                u_hat = (force - M G, w_des), xdes = [0,0,yaw, vel, pos]   (:339-343)
//   k_cbf_filter_o2* per env   : DroneQPTracker.compute_control                             (:345)
//   k_lowlevel_step  per drone : u_safe[0] += M G (:346), ThrustOmegaController (:348),
//                                env.step(action) (:350)
// ------------------------------------------------------------------------------------
// Optional tail of k_lowlevel_step: the NEXT control step's nominal controller, on the state the kernel has just produced and still
// holds in registers (the C loop knows t_{k+1}): unom / xdes of step k+1 come out of step k's launch, one launch and one read of the
// state less per step.  kind 0: GeometricControl (return_omegas), 1: LQROmegaController (K points at its LqrGain).  unom == NULL: off.
template <typename T, typename S> struct NextNominal {
  const T* lem;
  const void* K;
  S* unom;
  S* xdes;
  double t;
  T hover_sub;
  int kind;
  int only;      // 1: the launch is ONLY this tail (the stand-alone nominal controller of the step-by-step loop): no low level, no physics
};

// The nominal controller of the CBF loops for drone i (simulations/CBFTest.py:303-343): trajectory sample, GeometricControl
// (return_omegas) or LQROmegaController, u_hat = (force - M G, w) and xdes = [0, 0, yaw, v_des, p_des] into the scratch.
// The step-by-step loop and the C rollout (which chains the nominal input of step k+1 onto the low-level launch of step k) must stay
// bitwise equal.  Inlined into two different kernels this body does NOT compile to the same arithmetic (1-ulp differences in 1 % of
// the RPMs after three steps, also with its inputs made opaque; a __noinline__ body is equal but costs 50 % of the C4 step), so it has
// exactly ONE call site: the tail of k_lowlevel_step, which the step-by-step loop launches in its `only` mode as its nominal kernel.
template <typename T, typename S>
__device__ __forceinline__ void cbf_nominal_of(const Consts<T>* cp, const NextNominal<T, S>* nxp, const State<T>* sp, const int i, const size_t ld) {
  const Consts<T>& c = *cp;
  const NextNominal<T, S>& nx = *nxp;
  const State<T>& sn = *sp;
  const LemniscateParams<T> P = {nx.lem[lidx(0, i, ld)], nx.lem[lidx(1, i, ld)], nx.lem[lidx(2, i, ld)], nx.lem[lidx(3, i, ld)],
                                 nx.lem[lidx(4, i, ld)], nx.lem[lidx(5, i, ld)], nx.lem[lidx(6, i, ld)]};
  const Desired<T> des = lemniscate_local(P, nx.t);
  T un[4];
  if (nx.kind == 0) {
    const M3<T> Rm = quat_to_rot(sn.q);
    const V3<T> ang_v = mul(Rm, sn.w);
    T uu[4];
    GeoAux<T> A;
    geometric_control<T>(c, sn.p - des.p, Rm, sn.v, ang_v, des, uu, &A);
    un[0] = A.force - nx.hover_sub;
    un[1] = A.w_des.x; un[2] = A.w_des.y; un[3] = A.w_des.z;
  } else {
    T uu[4];
    lqr_omega_control<T>(c, *static_cast<const LqrGain<T>*>(nx.K), euler_from_quat(sn.q), sn.v, sn.p, des.p, des.v, des.yaw, uu);
    un[0] = uu[0] - nx.hover_sub;
    un[1] = uu[1]; un[2] = uu[2]; un[3] = uu[3];
  }
  store4<S, T>(nx.unom + (size_t)i * 4, un);
  S* xd = nx.xdes + (size_t)i * 9;
  xd[0] = (S)0; xd[1] = (S)0; xd[2] = (S)des.yaw;
  xd[3] = (S)des.v.x; xd[4] = (S)des.v.y; xd[5] = (S)des.v.z;
  xd[6] = (S)(des.p.x + P.cx); xd[7] = (S)(des.p.y + P.cy); xd[8] = (S)(des.p.z + P.cz);
}

// LQROmegaController.compute(obs, skip_low_level=True): obs [n,20], des [n,11] -> u [n,4]
template <typename T, typename S>
__global__ void k_lqr_omega_compute(const Consts<T> c, const LqrGain<T> K, const int n, const S* __restrict__ obs,
                                    const S* __restrict__ des, S* __restrict__ u_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const S* o = obs + (size_t)i * 20;
  const S* d = des + (size_t)i * 11;
  T u[4];
  lqr_omega_control<T>(c, K, V3<T>{(T)o[7], (T)o[8], (T)o[9]}, V3<T>{(T)o[10], (T)o[11], (T)o[12]}, V3<T>{(T)o[0], (T)o[1], (T)o[2]},
                       V3<T>{(T)d[0], (T)d[1], (T)d[2]}, V3<T>{(T)d[3], (T)d[4], (T)d[5]}, (T)d[9], u);
  store4<S, T>(u_out + (size_t)i * 4, u);
}

// Order-3 loop (simulations/CBFTestOrd3.py:306-352): LQRYankOmegaController nominal.  The thrust state is
// calc_z_thrust(obs), i.e. the last clipped RPM in columns 16:20 of the obs the previous step returned.
//   u_hat = (yank - M G, w)  -- the hover force is subtracted from the YANK as well (:341, kept) and never added back (:350)
//   xdes  = [0, 0, yaw, G M, vel, pos]                                                              (:345-347)
template <typename T, typename S>
__global__ __launch_bounds__(kBlock) void k_cbf_nominal_lqr_yo(const Consts<T> c, const LqrYoGain<T> K, const int n, const size_t ld,
                                                               const double t, const T hover_sub, const S* __restrict__ state,
                                                               const T* __restrict__ lem, const S* __restrict__ obs_prev,
                                                               S* __restrict__ unom, S* __restrict__ xdes, const int batch0) {
  const int i = (batch0 + blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  GeoIn<T> in;
  load_geo_in<T, S>(state, lem, ld, i, in);
  const Desired<T> des = lemniscate_local(in.P, t);
  const V3<T> rpy = euler_from_quat(in.s.q);
  T rpm[4], u[4];
  load4<S, T>(obs_prev + (size_t)i * kObsDim + 16, rpm);
  lqr_yank_omega_control<T>(c, K, rpy, rpm, in.s.v, in.s.p, des.p, des.v, des.yaw, u);
  const T un[4] = {u[0] - hover_sub, u[1], u[2], u[3]};
  store4<S, T>(unom + (size_t)i * 4, un);
  S* xd = xdes + (size_t)i * 10;
  xd[0] = (S)0; xd[1] = (S)0; xd[2] = (S)des.yaw; xd[3] = (S)c.gravity;
  xd[4] = (S)des.v.x; xd[5] = (S)des.v.y; xd[6] = (S)des.v.z;
  xd[7] = (S)(des.p.x + in.P.cx); xd[8] = (S)(des.p.y + in.P.cy); xd[9] = (S)(des.p.z + in.P.cz);
}

// LQRYankOmegaController.compute(obs, skip_low_level=True): obs [n,20], des [n,11] -> u [n,4]
template <typename T, typename S>
__global__ void k_lqr_yank_omega_compute(const Consts<T> c, const LqrYoGain<T> K, const int n, const S* __restrict__ obs,
                                         const S* __restrict__ des, S* __restrict__ u_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const S* o = obs + (size_t)i * 20;
  const S* d = des + (size_t)i * 11;
  const T rpm[4] = {(T)o[16], (T)o[17], (T)o[18], (T)o[19]};
  T u[4];
  lqr_yank_omega_control<T>(c, K, V3<T>{(T)o[7], (T)o[8], (T)o[9]}, rpm, V3<T>{(T)o[10], (T)o[11], (T)o[12]},
                            V3<T>{(T)o[0], (T)o[1], (T)o[2]}, V3<T>{(T)d[0], (T)d[1], (T)d[2]}, V3<T>{(T)d[3], (T)d[4], (T)d[5]},
                            (T)d[9], u);
  store4<S, T>(u_out + (size_t)i * 4, u);
}

template <typename T, typename S, bool RK4, bool DRAG, bool YANK, bool COMP = false>
__global__ __launch_bounds__(kBlock) void k_lowlevel_step(const Consts<T> c, const int n, const size_t ld, const T ctrl_dt,
                                                          const T thrust_offset, S* __restrict__ state,
                                                          const T* __restrict__ origin, T* __restrict__ last_rpm,
                                                          T* __restrict__ ll, const S* __restrict__ u_in, S* __restrict__ obs,
                                                          S* __restrict__ action_out, const int batch0, S* __restrict__ state_lo = nullptr,
                                                          const NextNominal<T, S> nx = NextNominal<T, S>{nullptr, nullptr, nullptr, nullptr, 0.0, T(0), 0, 0}) {
  __shared__ __align__(16) unsigned char lds[kBlock * kObsDim * sizeof(S)];
  const int i = (batch0 + blockIdx.x) * kBlock + threadIdx.x;
  const bool valid = i < n;
  const bool do_step = !(!YANK && nx.only != 0);               // wave-uniform
  T o[kObsDim];
  State<T> s;
  Resid<T> r;
  if (valid) {
    load_state<S, T>(state, ld, i, s);
    if (do_step) {
      if (COMP) load_resid<S, T>(state_lo, ld, i, r);
      T u[4];
      load4<S, T>(u_in + (size_t)i * 4, u);
      u[0] += thrust_offset;
      LowLevelState<T> L;
      L.last_omega = {ll[0 * ld + i], ll[1 * ld + i], ll[2 * ld + i]};
      L.integral = {ll[3 * ld + i], ll[4 * ld + i], ll[5 * ld + i]};
      T act[4], prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4];
      // compute_low_level rotates obs[13:16] (= R w) back with R^T: the body rate is the state's w
      if (YANK) {   // obs still holds the previous step's row here: every row is read and later rewritten by the same wave
        T rpm_prev[4];
        load4<S, T>(obs + (size_t)i * kObsDim + 16, rpm_prev);
        yank_omega_control(c, ctrl_dt, u, rpm_prev, s.w, L, act);
      } else {
        thrust_omega_control(c, ctrl_dt, u, s.w, L, act);
      }
      ll[0 * ld + i] = L.last_omega.x; ll[1 * ld + i] = L.last_omega.y; ll[2 * ld + i] = L.last_omega.z;
      ll[3 * ld + i] = L.integral.x; ll[4 * ld + i] = L.integral.y; ll[5 * ld + i] = L.integral.z;
      if (DRAG)
        for (int k = 0; k < 4; ++k) prev[k] = last_rpm[k * ld + i];
      aviary_step_any<T, RK4, DRAG, COMP>(c, s, r, act, prev, clipped);
      if (DRAG || last_rpm)
        for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = clipped[k];
      if (action_out) store4<S, T>(action_out + (size_t)i * 4, act);
      pack_obs(s, V3<T>{origin[i], origin[ld + i], origin[2 * ld + i]}, clipped, o);
    }
  }
  if (do_step) write_obs_rows<S, T>(lds, obs, n, i, valid, o);
  if (valid) {
    if (do_step) {
      store_state<S, T>(state, ld, i, s);
      if (COMP) store_resid<S, T>(state_lo, ld, i, r);
    }
    if (!YANK && nx.unom != nullptr) {        // the nominal controller on the state just stored (only: on the state just loaded)
      const State<T> sn = state_as_stored<T, S>(s);
      cbf_nominal_of<T, S>(&c, &nx, &sn, i, ld);
    }
  }
}

// ThrustOmegaController.computeControlFromInput through LQROmegaController.compute_low_level
// (lqr_omega_controller.py:77-88): u [n,4] = (thrust, w_target), obs [n,20] -> rpm [n,4]; stateful.
template <typename T, typename S>
__global__ void k_thrust_omega(const Consts<T> c, const int n, const size_t ld, const T ctrl_dt, const int body_rates_given, const int yank,
                               T* __restrict__ ll, const S* __restrict__ u_in, const S* __restrict__ obs_or_rates,
                               S* __restrict__ rpm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T u[4], act[4];
  load4<S, T>(u_in + (size_t)i * 4, u);
  V3<T> cur;
  if (body_rates_given) {
    cur = {(T)obs_or_rates[(size_t)i * 3], (T)obs_or_rates[(size_t)i * 3 + 1], (T)obs_or_rates[(size_t)i * 3 + 2]};
  } else {
    const S* o = obs_or_rates + (size_t)i * 20;
    const T q[4] = {(T)o[3], (T)o[4], (T)o[5], (T)o[6]};
    cur = mulT(quat_to_rot(q), V3<T>{(T)o[13], (T)o[14], (T)o[15]});
  }
  LowLevelState<T> L;
  L.last_omega = {ll[0 * ld + i], ll[1 * ld + i], ll[2 * ld + i]};
  L.integral = {ll[3 * ld + i], ll[4 * ld + i], ll[5 * ld + i]};
  if (yank) {   // YankOmegaController: needs the obs (calc_z_thrust), never combined with body_rates_given
    const S* o = obs_or_rates + (size_t)i * 20;
    const T rpm_prev[4] = {(T)o[16], (T)o[17], (T)o[18], (T)o[19]};
    yank_omega_control(c, ctrl_dt, u, rpm_prev, cur, L, act);
  } else {
    thrust_omega_control(c, ctrl_dt, u, cur, L, act);
  }
  ll[0 * ld + i] = L.last_omega.x; ll[1 * ld + i] = L.last_omega.y; ll[2 * ld + i] = L.last_omega.z;
  ll[3 * ld + i] = L.integral.x; ll[4 * ld + i] = L.integral.y; ll[5 * ld + i] = L.integral.z;
  store4<S, T>(rpm + (size_t)i * 4, act);
}

// ------------------------------------------------------------------------------------
// PIDEnv.MultiDroneEnv.sim_step (PIDEnv.py:161-176) for every drone: [UPSTREAM] DSLPIDControl
// towards TARGET_POSITIONS / TARGET_RPYS, then env.step(action).  pid = 9 planes:
// last_rpy3 | integral_pos_e3 | integral_rpy_e3.  STEP = false: controller only, from an obs array (or the state, obs_in NULL).
// ------------------------------------------------------------------------------------
template <typename T, typename S, bool STEP, bool RK4, bool DRAG>
__global__ __launch_bounds__(kBlock) void k_dslpid(const Consts<T> c, const DslPidGains<T> g, const int n, const size_t ld, const T ctrl_dt,
                                                   S* __restrict__ state, const T* __restrict__ origin, T* __restrict__ last_rpm,
                                                   T* __restrict__ pid, const S* __restrict__ obs_in, const S* __restrict__ tpos,
                                                   const S* __restrict__ trpy, S* __restrict__ obs, S* __restrict__ action_out,
                                                   const int batch0, S* __restrict__ state_lo = nullptr) {
  __shared__ __align__(16) unsigned char lds[STEP ? (kBlock * kObsDim * sizeof(S)) : 16];
  // batch0: first 256-drone batch of this launch (mds_rollout_dslpid may step the two halves of the shard on two streams)
  const int i = (batch0 + blockIdx.x) * kBlock + threadIdx.x;
  const bool valid = i < n;
  T o[kObsDim];
  State<T> s;
  Resid<T> r;                   // STEP with state_lo != NULL (MDS_F32C handles, a uniform branch): compensated accumulation
  if (valid) {
    V3<T> org = {T(0), T(0), T(0)};
    if (STEP && state_lo) load_resid<S, T>(state_lo, ld, i, r);
    if (STEP || !obs_in) {      // controller only with obs_in == NULL: from the handle's own state (ground effect / downwash steps)
      load_state<S, T>(state, ld, i, s);
      org = {origin[i], origin[ld + i], origin[2 * ld + i]};
    } else {
      const S* ob = obs_in + (size_t)i * 20;
      s.p = {(T)ob[0], (T)ob[1], (T)ob[2]};
      for (int k = 0; k < 4; ++k) s.q[k] = (T)ob[3 + k];
      s.v = {(T)ob[10], (T)ob[11], (T)ob[12]};
      s.w = {T(0), T(0), T(0)};
    }
    DslPidState<T> P;
    P.last_rpy = {pid[0 * ld + i], pid[1 * ld + i], pid[2 * ld + i]};
    P.int_pos = {pid[3 * ld + i], pid[4 * ld + i], pid[5 * ld + i]};
    P.int_rpy = {pid[6 * ld + i], pid[7 * ld + i], pid[8 * ld + i]};
    const V3<T> pos_e = {((T)tpos[(size_t)i * 3] - org.x) - s.p.x, ((T)tpos[(size_t)i * 3 + 1] - org.y) - s.p.y,
                         ((T)tpos[(size_t)i * 3 + 2] - org.z) - s.p.z};
    T act[4], prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4];
    dslpid_control<T>(c, g, ctrl_dt, pos_e, s.q, s.v, (T)trpy[(size_t)i * 3 + 2], P, act);
    pid[0 * ld + i] = P.last_rpy.x; pid[1 * ld + i] = P.last_rpy.y; pid[2 * ld + i] = P.last_rpy.z;
    pid[3 * ld + i] = P.int_pos.x; pid[4 * ld + i] = P.int_pos.y; pid[5 * ld + i] = P.int_pos.z;
    pid[6 * ld + i] = P.int_rpy.x; pid[7 * ld + i] = P.int_rpy.y; pid[8 * ld + i] = P.int_rpy.z;
    if (action_out) store4<S, T>(action_out + (size_t)i * 4, act);
    if (STEP) {
      if (DRAG)
        for (int k = 0; k < 4; ++k) prev[k] = last_rpm[k * ld + i];
      if (state_lo) aviary_step_comp<T, RK4, DRAG>(c, s, r, act, prev, clipped);
      else aviary_step<T, RK4, DRAG>(c, s, act, prev, clipped);
      if (DRAG || last_rpm)
        for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = clipped[k];
      if (obs) pack_obs(s, org, clipped, o);
    }
  }
  if (STEP) {
    if (obs) write_obs_rows<S, T>(lds, obs, n, i, valid, o);
    if (valid) {
      store_state<S, T>(state, ld, i, s);
      if (state_lo) store_resid<S, T>(state_lo, ld, i, r);
    }
  }
}

// [UPSTREAM] _computeObs from the current state
template <typename T, typename S>
__global__ __launch_bounds__(kBlock) void k_get_obs(const int n, const size_t ld, const S* __restrict__ state,
                                                    const T* __restrict__ origin, const T* __restrict__ last_rpm,
                                                    S* __restrict__ obs) {
  __shared__ __align__(16) unsigned char lds[kBlock * kObsDim * sizeof(S)];
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = i < n;
  T o[kObsDim];
  if (valid) {
    State<T> s;
    load_state<S, T>(state, ld, i, s);
    // last_rpm == nullptr: the handle does not track the last clipped action and a step has run since the
    // last reset -- the columns are NaN rather than stale (include/mds.h, mds_config.track_last_rpm)
    const T kNaN = __builtin_nanf("");
    const T rpm[4] = {last_rpm ? last_rpm[i] : kNaN, last_rpm ? last_rpm[ld + i] : kNaN, last_rpm ? last_rpm[2 * ld + i] : kNaN,
                      last_rpm ? last_rpm[3 * ld + i] : kNaN};
    pack_obs(s, V3<T>{origin[i], origin[ld + i], origin[2 * ld + i]}, rpm, o);
  }
  write_obs_rows<S, T>(lds, obs, n, i, valid, o);
}

// ------------------------------------------------------------------------------------
// set-up kernels (double in, storage out; not on the hot path)
// ------------------------------------------------------------------------------------
// state_lo (compensated storage, else NULL): what the rounding to S dropped from the body rates (load_resid's layout)
template <typename T, typename S>
__global__ void k_reset(const int n, const size_t ld, const double* __restrict__ xyz, const double* __restrict__ rpy,
                        const T* __restrict__ origin, S* __restrict__ state, T* __restrict__ last_rpm, const int i0,
                        S* __restrict__ state_lo = nullptr) {
  const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double q[4];
  quat_from_euler<double>(rpy[3 * i], rpy[3 * i + 1], rpy[3 * i + 2], q);
  for (int k = 0; k < 13; ++k) {
    const double v = k < 3 ? xyz[3 * i + k] - (double)origin[k * ld + i] : (k < 7 ? q[k - 3] : 0.0);
    const S hi = (S)v;
    state[sidx(k, i, ld)] = hi;
    if (state_lo && k >= 10) state_lo[ridx(k, i)] = (S)(v - (double)hi);
  }
  if (state_lo) state_lo[4 * (size_t)i + 3] = (S)0;
  for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = T(0);
}

template <typename T, typename S>
__global__ void k_set_origin(const int n, const size_t ld, const double* __restrict__ new_origin, T* __restrict__ origin,
                             S* __restrict__ state, S* __restrict__ state_lo = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 3; ++k) {
    const double world = (double)state[sidx(k, i, ld)] + (double)origin[k * ld + i];      // (no position residual is stored)
    const T no = (T)new_origin[3 * i + k];
    origin[k * ld + i] = no;
    const double local = world - (double)no;
    state[sidx(k, i, ld)] = (S)local;
  }
  (void)state_lo;
}

template <typename T, typename S>
__global__ void k_get_state(const int n, const size_t ld, const S* __restrict__ state, const T* __restrict__ origin,
                            double* __restrict__ out, const S* __restrict__ state_lo = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 13; ++k) {
    double v = (double)state[sidx(k, i, ld)];
    if (state_lo && k >= 10) v += (double)state_lo[ridx(k, i)];
    if (k < 3) v += (double)origin[k * ld + i];
    out[13 * (size_t)i + k] = v;
  }
}

template <typename T, typename S>
__global__ void k_set_state(const int n, const size_t ld, const double* __restrict__ in, const T* __restrict__ origin,
                            S* __restrict__ state, S* __restrict__ state_lo = nullptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 13; ++k) {
    double v = in[13 * (size_t)i + k];
    if (k < 3) v -= (double)origin[k * ld + i];
    const S hi = (S)v;
    state[sidx(k, i, ld)] = hi;
    if (state_lo && k >= 10) state_lo[ridx(k, i)] = (S)(v - (double)hi);
  }
  if (state_lo) state_lo[4 * (size_t)i + 3] = (S)0;
}

template <typename T>
__global__ void k_set_planes(const int n, const size_t ld, const int dim, const double* __restrict__ in, T* __restrict__ planes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < dim; ++k) planes[dim == 7 ? lidx(k, i, ld) : (size_t)k * ld + i] = (T)in[(size_t)dim * i + k];
}

// ------------------------------------------------------------------------------------
// stand-alone operators (array-of-structs in/out; parity surface, not the fused path)
// ------------------------------------------------------------------------------------
template <typename T, typename S>
__global__ void k_lemniscate_eval(const int n, const size_t ld, const double t, const T* __restrict__ lem, S* __restrict__ des) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  LemniscateParams<T> P = {lem[lidx(0, i, ld)], lem[lidx(1, i, ld)], lem[lidx(2, i, ld)], lem[lidx(3, i, ld)], lem[lidx(4, i, ld)],
                           lem[lidx(5, i, ld)], lem[lidx(6, i, ld)]};
  const Desired<T> d = lemniscate_local(P, t);
  S* o = des + (size_t)i * 11;
  o[0] = (S)(d.p.x + P.cx); o[1] = (S)(d.p.y + P.cy); o[2] = (S)(d.p.z + P.cz);
  o[3] = (S)d.v.x; o[4] = (S)d.v.y; o[5] = (S)d.v.z;
  o[6] = (S)d.a.x; o[7] = (S)d.a.y; o[8] = (S)d.a.z;
  o[9] = (S)d.yaw; o[10] = (S)d.yaw_rate;
}

template <typename T, typename S>
__global__ void k_geometric_compute(const Consts<T> c, const int n, const S* __restrict__ obs, const S* __restrict__ des,
                                    S* __restrict__ rpm, S* __restrict__ aux) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const S* o = obs + (size_t)i * 20;
  const S* d = des + (size_t)i * 11;
  // utils/model_conversions.py:105-114 obs_to_geo_model
  const T q[4] = {(T)o[3], (T)o[4], (T)o[5], (T)o[6]};
  const M3<T> R = quat_to_rot(q);
  const V3<T> p = {(T)o[0], (T)o[1], (T)o[2]}, v = {(T)o[10], (T)o[11], (T)o[12]}, w = {(T)o[13], (T)o[14], (T)o[15]};
  Desired<T> D;
  D.p = {(T)d[0], (T)d[1], (T)d[2]};
  D.v = {(T)d[3], (T)d[4], (T)d[5]};
  D.a = {(T)d[6], (T)d[7], (T)d[8]};
  D.yaw = (T)d[9];
  D.yaw_rate = (T)d[10];
  // the stand-alone operator takes an arbitrary yaw: reduce it for the fp32 sincos
  D.yaw = reduced_phase<T>(0.0, T(0), D.yaw);
  T u[4], act[4];
  GeoAux<T> A;
  geometric_control<T>(c, p - D.p, R, v, w, D, u, &A);
  input_to_action(c, u, act);
  store4<S, T>(rpm + (size_t)i * 4, act);
  if (aux) {
    S* a = aux + (size_t)i * 13;
    a[0] = (S)A.force;
    a[1] = (S)A.w_des.x; a[2] = (S)A.w_des.y; a[3] = (S)A.w_des.z;
    a[4] = (S)A.b1d.x; a[5] = (S)A.b2d.x; a[6] = (S)A.b3d.x;     // R_des row-major, columns b1d b2d b3d
    a[7] = (S)A.b1d.y; a[8] = (S)A.b2d.y; a[9] = (S)A.b3d.y;
    a[10] = (S)A.b1d.z; a[11] = (S)A.b2d.z; a[12] = (S)A.b3d.z;
  }
}

// utils/model_conversions.py:105-114 obs_to_geo_model: x18 = [pos, R(quat) row-major (normalising, as scipy's Rotation), vel, ang_v];
// :20-58 obs_to_lin_model(obs, dim): [rpy, (ang_v | F |), vel, pos] for dim 12 / 10 / 9, F = calc_z_thrust (:137-143).
// Format adapters for callers written against those helpers; the fused kernels do the same in registers.
template <typename T, typename S>
__global__ void k_obs_to_model(const Consts<T> c, const int n, const int dim, const S* __restrict__ obs, S* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const S* o = obs + (size_t)i * kObsDim;
  S* out = x + (size_t)i * dim;
  if (dim == 18) {
    const T q[4] = {(T)o[3], (T)o[4], (T)o[5], (T)o[6]};
    const M3<T> R = quat_to_rot(q);
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
    for (int k = 0; k < 9; ++k) out[3 + k] = (S)R.m[k];                          // row-major
    for (int k = 0; k < 6; ++k) out[12 + k] = o[10 + k];
    return;
  }
  out[0] = o[7]; out[1] = o[8]; out[2] = o[9];
  int p = 3;
  if (dim == 12) {
    out[3] = o[13]; out[4] = o[14]; out[5] = o[15];
    p = 6;
  } else if (dim == 10) {
    const T r0 = (T)o[16], r1 = (T)o[17], r2 = (T)o[18], r3 = (T)o[19];
    out[3] = (S)(c.kf * (r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3));
    p = 4;
  }
  out[p] = o[10]; out[p + 1] = o[11]; out[p + 2] = o[12];
  out[p + 3] = o[0]; out[p + 4] = o[1]; out[p + 5] = o[2];
}

template <typename T, typename S>
__global__ void k_input_to_action(const Consts<T> c, const int n, const S* __restrict__ u, S* __restrict__ rpm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T ui[4], r[4];
  load4<S, T>(u + (size_t)i * 4, ui);
  input_to_action(c, ui, r);
  store4<S, T>(rpm + (size_t)i * 4, r);
}

template <typename T, typename S>
__global__ void k_action_to_input(const Consts<T> c, const int n, const int cap, const S* __restrict__ rpm, S* __restrict__ u) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T r[4], ui[4];
  load4<S, T>(rpm + (size_t)i * 4, r);
  action_to_input(c, r, cap, ui);
  store4<S, T>(u + (size_t)i * 4, ui);
}

template <typename T, typename S>
__global__ void k_quadrotor_dynamics(const int n, const S* __restrict__ state, const S* __restrict__ u, const T m, const T J0,
                                     const T J1, const T J2, const T g, S* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T s[18], ui[4], o[12];
  for (int k = 0; k < 18; ++k) s[k] = (T)state[(size_t)i * 18 + k];
  for (int k = 0; k < 4; ++k) ui[k] = (T)u[(size_t)i * 4 + k];
  const T J[3] = {J0, J1, J2};
  quadrotor_dynamics<T>(s, ui, m, J, g, o);
  for (int k = 0; k < 12; ++k) out[(size_t)i * 12 + k] = (S)o[k];
}

// ------------------------------------------------------------------------------------
// The call site of a5: simulations/CompareModels.py:46-56 over a logged rollout [count, 20] -> three [count, 12] arrays.
// Streaming, HBM-bound (fp32: 80 B read + 3 x 48 B written per row): the wave's 64 rows are one contiguous span of every array, so
// rows move through the wave's LDS slice in 16-byte chunks, coalesced on the memory side, one row per lane on the register side.
// ------------------------------------------------------------------------------------
template <typename S, typename T, int DIM>
__device__ __forceinline__ void read_rows(unsigned char* __restrict__ lds_wave, const S* __restrict__ src, int n, int wave_base, int lane,
                                          T out[DIM]) {
  constexpr int kRowBytes = DIM * (int)sizeof(S);
  constexpr int kUnit = (kRowBytes % 16 == 0) ? 16 : 8;
  static_assert(kRowBytes % 8 == 0, "rows are moved in 8- or 16-byte units");
  const int rows = min(kWave, n - wave_base);
  const int bytes = rows * kRowBytes;
  const unsigned char* g = reinterpret_cast<const unsigned char*>(src) + (size_t)wave_base * kRowBytes;
  constexpr int kIters = (kWave * kRowBytes + kWave * 16 - 1) / (kWave * 16);
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const int off = (it * kWave + lane) * 16;
    if (off + 16 <= bytes) *reinterpret_cast<v4u*>(lds_wave + off) = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(g + off));
    else if (off + 8 <= bytes) *reinterpret_cast<uint2*>(lds_wave + off) = *reinterpret_cast<const uint2*>(g + off);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  alignas(16) S row[DIM];
  const unsigned char* r = lds_wave + (lane < rows ? lane : 0) * kRowBytes;        // lanes past the end: row 0 (finite, never stored)
  if (kUnit == 16) {
#pragma unroll
    for (int k = 0; k < kRowBytes / 16; ++k) reinterpret_cast<uint4*>(row)[k] = reinterpret_cast<const uint4*>(r)[k];
  } else {
#pragma unroll
    for (int k = 0; k < kRowBytes / 8; ++k) reinterpret_cast<uint2*>(row)[k] = reinterpret_cast<const uint2*>(r)[k];
  }
#pragma unroll
  for (int k = 0; k < DIM; ++k) out[k] = (T)row[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                            // the slice is written again by write_rows
  __builtin_amdgcn_wave_barrier();
}

template <typename S, typename T, int DIM>
__device__ __forceinline__ void write_rows(unsigned char* __restrict__ lds_wave, S* __restrict__ dst, int n, int wave_base, int lane,
                                           const T v[DIM]) {
  constexpr int kRowBytes = DIM * (int)sizeof(S);
  constexpr int kUnit = (kRowBytes % 16 == 0) ? 16 : 8;
  static_assert(kRowBytes % 8 == 0, "rows are moved in 8- or 16-byte units");
  const int rows = min(kWave, n - wave_base);
  if (lane < rows) {
    alignas(16) S row[DIM];
#pragma unroll
    for (int k = 0; k < DIM; ++k) row[k] = (S)v[k];
    unsigned char* w = lds_wave + lane * kRowBytes;
    if (kUnit == 16) {
#pragma unroll
      for (int k = 0; k < kRowBytes / 16; ++k) reinterpret_cast<uint4*>(w)[k] = reinterpret_cast<const uint4*>(row)[k];
    } else {
#pragma unroll
      for (int k = 0; k < kRowBytes / 8; ++k) reinterpret_cast<uint2*>(w)[k] = reinterpret_cast<const uint2*>(row)[k];
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int bytes = rows * kRowBytes;
  unsigned char* g = reinterpret_cast<unsigned char*>(dst) + (size_t)wave_base * kRowBytes;
  constexpr int kIters = (kWave * kRowBytes + kWave * 16 - 1) / (kWave * 16);
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int it = 0; it < kIters; ++it) {
    const int off = (it * kWave + lane) * 16;
    if (off + 16 <= bytes) __builtin_nontemporal_store(*reinterpret_cast<const v4u*>(lds_wave + off), reinterpret_cast<v4u*>(g + off));
    else if (off + 8 <= bytes) *reinterpret_cast<uint2*>(g + off) = *reinterpret_cast<const uint2*>(lds_wave + off);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                            // reads before the slice's next writes
  __builtin_amdgcn_wave_barrier();
}

template <typename T, typename S>
__global__ __launch_bounds__(kBlock) void k_compare_models(const Consts<T> c, const LinModel<T> M, const int n, const S* __restrict__ obs, const T dm,
                                                           const T dJ0, const T dJ1, const T dJ2, const T dg, S* __restrict__ xdot_lin,
                                                           S* __restrict__ xdot_geo, S* __restrict__ x_lin) {
  __shared__ __align__(16) unsigned char lds[kBlock * kObsDim * sizeof(S)];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int wave_base = blockIdx.x * kBlock + wave * kWave;
  if (wave_base >= n) return;                                    // (wave-uniform; the staging uses wave-scope synchronisation only)
  unsigned char* lds_wave = lds + wave * (kWave * kObsDim * (int)sizeof(S));
  T o[kObsDim];
  read_rows<S, T, kObsDim>(lds_wave, obs, n, wave_base, lane, o);
  T xl[12], xd[12], xg[12];
  const T dJ[3] = {dJ0, dJ1, dJ2};
  compare_models_row<T>(c, M, o, dm, dJ, dg, xl, xd, xg);
  if (xdot_lin) write_rows<S, T, 12>(lds_wave, xdot_lin, n, wave_base, lane, xd);
  if (xdot_geo) write_rows<S, T, 12>(lds_wave, xdot_geo, n, wave_base, lane, xg);
  if (x_lin) write_rows<S, T, 12>(lds_wave, x_lin, n, wave_base, lane, xl);
}

// model/linearized.py:92-104 calc_xdot(x, action) on states of the caller's own (the right-hand side CompareModels.py:84-92 integrates)
template <typename T, typename S>
__global__ void k_linear_xdot(const Consts<T> c, const LinModel<T> M, const int n, const S* __restrict__ x, const S* __restrict__ action,
                              S* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T xi[12], a[4], u[4], o[12];
  for (int k = 0; k < 12; ++k) xi[k] = (T)x[(size_t)i * 12 + k];
  for (int k = 0; k < 4; ++k) a[k] = (T)action[(size_t)i * 4 + k];
  action_to_input(c, a, 1, u);
  linear_xdot(M, xi, u, o);
  for (int k = 0; k < 12; ++k) out[(size_t)i * 12 + k] = (S)o[k];
}

// utils/model_conversions.py:4-19 rpy_to_rot: [n,3] -> [n,9] row-major
template <typename T, typename S> __global__ void k_rpy_to_rot(const int n, const S* __restrict__ rpy, S* __restrict__ R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const M3<T> m = rpy_to_rot<T>(reduced_phase<T>(0.0, T(0), (T)rpy[(size_t)i * 3]), reduced_phase<T>(0.0, T(0), (T)rpy[(size_t)i * 3 + 1]),
                                reduced_phase<T>(0.0, T(0), (T)rpy[(size_t)i * 3 + 2]));
  for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = (S)m.m[k];
}

// utils/model_conversions.py:116-122 geo_model_to_obs: [n,18] (p, R row-major, v, w) -> [n,16] (p, quat xyzw, 0 0 0, v, w)
template <typename T, typename S> __global__ void k_geo_model_to_obs(const int n, const S* __restrict__ x, S* __restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const S* s = x + (size_t)i * 18;
  S* o = obs + (size_t)i * 16;
  T m[9], q[4];
  for (int k = 0; k < 9; ++k) m[k] = (T)s[3 + k];
  rot_to_quat_scipy<T>(m, q);
  o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
  for (int k = 0; k < 4; ++k) o[3 + k] = (S)q[k];
  o[7] = o[8] = o[9] = (S)T(0);
  for (int k = 0; k < 6; ++k) o[10 + k] = s[12 + k];
}

}  // namespace mds
