// ECBF safety filter kernels (a11-a15): constraint rows of cbf/cbf.py and the QP of
// cbf/qptracker.py:86-114, one wavefront per environment.
#include <hip/hip_runtime.h>

#include "mds_cbf.hpp"

namespace mds {

constexpr int kCbfMaxD = 32;      // drones per env supported by the wave-per-env kernels
constexpr int kCbfMaxObs = 16;
constexpr int kCbfHeavyIters = 6, kCbfMediumIters = 2;   // cost classes of the longest-first dispatch (GI iterations last step)

// pair_ij[r] = i | j << 8: (i, j) of pair row r in the reference's lexicographic order
// (cbf/cbf.py:342-346); built on the host by mds_cbf_configure.

// ------------------------------------------------------------------------------------
// Dense G, h exactly as CBF._build_ineq_const returns them (parity surface; one workgroup
// per env).  Row order: pairs | +I(4D) | -I(4D) | [order 3: 2 force rows per agent] | obstacles.
// ------------------------------------------------------------------------------------
template <typename T, typename S, int ORDER>
__global__ __launch_bounds__(256) void k_cbf_rows(const CbfParams<T> P, const int E, const int* __restrict__ pair_ij,
                                                  const T* __restrict__ obstacles, const S* __restrict__ x,
                                                  const S* __restrict__ xdes, S* __restrict__ G, S* __restrict__ h) {
  __shared__ T sx[kCbfMaxD][10], sxd[kCbfMaxD][10];
  const int env = blockIdx.x;
  constexpr int xd = ORDER == 2 ? 9 : 10;
  const int D = P.num_drones;
  const int npairs = cbf_num_pairs(D), m = cbf_num_rows(D, ORDER, P.n_obs), ncol = 4 * D;
  for (int k = threadIdx.x; k < D * xd; k += blockDim.x) {
    sx[k / xd][k % xd] = (T)x[(size_t)env * D * xd + k];
    sxd[k / xd][k % xd] = (T)xdes[(size_t)env * D * xd + k];
  }
  S* Ge = G + (size_t)env * m * ncol;
  S* he = h + (size_t)env * m;
  for (int k = threadIdx.x; k < m * ncol; k += blockDim.x) Ge[k] = (S)0;
  __syncthreads();
  const int box0 = npairs, force0 = npairs + 8 * D, obs0 = force0 + (ORDER == 3 ? 2 * D : 0);
  for (int r = threadIdx.x; r < m; r += blockDim.x) {
    T hr, Lg[4];
    if (r < npairs) {
      const int ij = pair_ij[r], i = ij & 255, j = ij >> 8;
      cbf_pair_row<T, ORDER>(P, sx[i], sxd[i], sx[j], sxd[j], false, P.Ds_pair, &hr, Lg);
      for (int k = 0; k < 4; ++k) {
        Ge[(size_t)r * ncol + 4 * i + k] = (S)(-Lg[k]);
        Ge[(size_t)r * ncol + 4 * j + k] = (S)Lg[k];
      }
      he[r] = (S)hr;
    } else if (r < force0) {                 // _build_umax_const (:400-412)
      const int q = r - box0, col = q % ncol;
      Ge[(size_t)r * ncol + col] = (S)(q < ncol ? 1 : -1);
      he[r] = (S)P.umax[col & 3];
    } else if (r < obs0) {                   // custom_force_bound_const (:446-464), column 4i+3 (sic)
      const int q = r - force0, i = q >> 1;
      Ge[(size_t)r * ncol + 4 * i + 3] = (S)((q & 1) ? -1 : 1);
      he[r] = (S)((q & 1) ? P.k[2] * (sx[i][3] - P.Fmin) : P.k[2] * (P.Fmax - sx[i][3]));
    } else {                                 // custom_build_obstacles_const (:369-398)
      const int q = r - obs0, i = q / P.n_obs, o = q % P.n_obs;
      T xo[xd];
#pragma unroll
      for (int k = 0; k < xd - 3; ++k) xo[k] = T(0);
      xo[xd - 3] = obstacles[4 * o];
      xo[xd - 2] = obstacles[4 * o + 1];
      xo[xd - 1] = obstacles[4 * o + 2];
      cbf_pair_row<T, ORDER>(P, sx[i], sxd[i], xo, xo, true, P.safety_radius + obstacles[4 * o + 3], &hr, Lg);
      for (int k = 0; k < 4; ++k) Ge[(size_t)r * ncol + 4 * i + k] = (S)(-Lg[k]);
      he[r] = (S)hr;
    }
  }
}

// ------------------------------------------------------------------------------------
// Order-2 filter: DroneQPTracker.compute_control (cbf/qptracker.py:22-34) for every env.
//
// With the omega linearisation only the thrust input reaches the barrier in two derivatives
// (LgLfh is non-zero in column 4i only), and P = I, so the QP of :86-114 separates exactly:
//   - omega components: box rows only  -> u = clip(u_hat, -umax, umax);
//   - thrust components: D-variable projection  min 1/2 |F - F_hat|^2  s.t.
//       -g_ij F_i + g_ij F_j <= h_ij (pairs), -g_io F_i <= h_io (obstacles), +-F_i <= umax_0.
// One wavefront per env.  Each lane keeps R rows (coefficients, multiplier) in registers; F
// lives in LDS.  Hildreth's dual coordinate ascent with Gauss-Southwell selection: every
// iteration all lanes evaluate their rows, a wave-wide arg-max picks the row whose multiplier
// update moves F the most, the owner applies it.  Converges to the unique minimiser when the
// rows are feasible.  Infeasibility is certified by weak duality: every update raises the dual
// value by score/2, and for a feasible problem the dual never exceeds the primal optimum, which
// the thrust box bounds by 1/2 sum_i (|F_hat_i| + umax_0)^2 -- once the accumulated dual value
// passes that bound the rows are infeasible and the env falls back to u_hat with status 1 (the
// reference falls back when cvxopt raises, qptracker.py:30-34).  The iteration cap is a backstop.
// ------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void wave_argmax(T& score, int& row) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const T os = __shfl_xor(score, off);
    const int orow = __shfl_xor(row, off);
    const bool take = (os > score) || (os == score && orow < row);
    score = take ? os : score;
    row = take ? orow : row;
  }
}

template <typename T, typename S, int R>
__global__ __launch_bounds__(256) void k_cbf_filter_o2(const CbfParams<T> P, const int E, const int* __restrict__ pair_ij,
                                                       const T* __restrict__ obstacles, const S* __restrict__ obs,
                                                       const S* __restrict__ xdes, const S* __restrict__ unom,
                                                       S* __restrict__ usafe, int* __restrict__ status, const int max_iter,
                                                       const T tol2) {
  __shared__ T sx[4][kCbfMaxD][9], sxd[4][kCbfMaxD][9], su[4][kCbfMaxD];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int env = blockIdx.x * 4 + wave;
  if (env >= E) return;                                    // wave-uniform
  const int D = P.num_drones;
  const size_t base = (size_t)env * D;
  for (int d = lane; d < D; d += 64) {
    T o[20];
    for (int k = 0; k < 20; ++k) o[k] = (T)obs[(base + d) * 20 + k];
    obs_to_lin<T>(o, 2, T(0), sx[wave][d]);
    for (int k = 0; k < 9; ++k) sxd[wave][d][k] = (T)xdes[(base + d) * 9 + k];
    su[wave][d] = (T)unom[(base + d) * 4];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int npairs = cbf_num_pairs(D), nobs_rows = D * P.n_obs, m = npairs + nobs_rows + 2 * D;
  T ci[R], cj[R], b[R], lam[R], n2[R], inv_n2[R];
  int ii[R], jj[R];
  bool bad = false;                                        // a row 0 * u <= h with h < 0: infeasible
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int r = lane + 64 * k;
    ci[k] = cj[k] = T(0);
    b[k] = T(0);
    ii[k] = jj[k] = 0;
    lam[k] = T(0);
    if (r < npairs) {
      const int ij = pair_ij[r];
      ii[k] = ij & 255;
      jj[k] = ij >> 8;
      T hr, Lg[4];
      cbf_pair_row<T, 2>(P, sx[wave][ii[k]], sxd[wave][ii[k]], sx[wave][jj[k]], sxd[wave][jj[k]], false, P.Ds_pair, &hr, Lg);
      ci[k] = -Lg[0];
      cj[k] = Lg[0];
      b[k] = hr;
    } else if (r < npairs + nobs_rows) {
      const int q = r - npairs, i = q / P.n_obs, o = q % P.n_obs;
      T xo[9] = {T(0), T(0), T(0), T(0), T(0), T(0), obstacles[4 * o], obstacles[4 * o + 1], obstacles[4 * o + 2]};
      T hr, Lg[4];
      cbf_pair_row<T, 2>(P, sx[wave][i], sxd[wave][i], xo, xo, true, P.safety_radius + obstacles[4 * o + 3], &hr, Lg);
      ii[k] = jj[k] = i;
      ci[k] = -Lg[0];
      b[k] = hr;
    } else if (r < m) {
      const int q = r - npairs - nobs_rows;
      ii[k] = jj[k] = q % D;
      ci[k] = q < D ? T(1) : T(-1);
      b[k] = P.umax[0];
    }
    n2[k] = m_fma(ci[k], ci[k], cj[k] * cj[k]);
    inv_n2[k] = n2[k] > T(0) ? T(1) / n2[k] : T(0);
    if (r < m && !(n2[k] > T(0)) && b[k] < T(0)) bad = true;
  }
  bool converged = false;
  const bool any_bad = __any(bad);
  // weak-duality bound on the optimum (wave-uniform): sum over drones of (|F_hat| + umax_0)^2
  T bound = T(0);
  for (int d = 0; d < D; ++d) {
    const T w = m_abs(su[wave][d]) + P.umax[0];
    bound = m_fma(w, w, bound);
  }
  bound *= T(1.0001);
  T dual2 = T(0);                                          // 2 x accumulated dual value
  int it = 0;
  for (; it < max_iter && !any_bad; ++it) {
    T best = T(0), best_dl = T(0);
    int best_k = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const T res = m_fma(ci[k], su[wave][ii[k]], m_fma(cj[k], su[wave][jj[k]], -b[k]));
      const T dl = m_max(-lam[k], res * inv_n2[k]);
      const T sc = dl * dl * n2[k];                        // squared length of the move of F
      if (sc > best) {
        best = sc;
        best_dl = dl;
        best_k = k;
      }
    }
    T wbest = best;
    int wrow = lane + 64 * best_k;
    wave_argmax(wbest, wrow);
    if (!(wbest > tol2)) {
      converged = true;
      break;
    }
    dual2 += wbest;
    if (dual2 > bound) break;                              // certified infeasible
    if (wrow == lane + 64 * best_k && best == wbest) {     // the owner applies its update
#pragma unroll
      for (int k = 0; k < R; ++k)
        if (k == best_k) {
          lam[k] += best_dl;
          const T ui = su[wave][ii[k]] - ci[k] * best_dl;
          if (jj[k] != ii[k]) su[wave][jj[k]] -= cj[k] * best_dl;
          su[wave][ii[k]] = ui;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  if (lane == 0) status[env] = converged ? 0 : 1;
  for (int d = lane; d < D; d += 64) {
    T u[4];
    for (int k = 0; k < 4; ++k) u[k] = (T)unom[(base + d) * 4 + k];
    if (converged) {
      u[0] = su[wave][d];
      for (int k = 1; k < 4; ++k) u[k] = m_clamp(u[k], -P.umax[k], P.umax[k]);
    }
    for (int k = 0; k < 4; ++k) usafe[(base + d) * 4 + k] = (S)u[k];
  }
}

// ------------------------------------------------------------------------------------
// Exact solver for the same thrust sub-problem: Goldfarb-Idnani dual active set with H = I,
// one wavefront per env.  The active normals N (n x q, q <= n = QP variables per env) are kept as a
// thin QR (Q: n x q orthonormal columns, R: q x q upper triangular) in the wave's LDS slice;
// adding a row appends a Gram-Schmidt column, dropping one re-triangularises with Givens
// rotations.  Rows are normalised to unit length so every threshold is a distance.  The number
// of iterations is of the order of the number of active rows (Hildreth's coordinate ascent
// needs 10^2..10^4 on crowded scenes); the result is the exact minimiser, the same the oracle's
// qp_project computes.
//
// The kernel is latency-bound per wave (a chain of small dependent steps), so:
//  * every global read of the env (obs, xdes, u_hat, pair table, obstacles) is issued before the
//    first wait;
//  * wave-wide reductions run on the VALU (DPP row all-reduce + 4 v_readlane), not through the
//    LDS crossbar (__shfl = ds_bpermute, ~12 dependent LDS round trips per arg-max);
//  * the selected row is fetched from an LDS copy of the rows by a uniform address instead of a
//    register-array select (which the compiler turns into scratch traffic);
//  * one wavefront per workgroup: a slow env does not pin the LDS / VGPRs of finished ones.
// ------------------------------------------------------------------------------------
#define MDS_WAVE_SYNC()                                   \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

namespace wv {
// DPP controls (gfx9): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141, row_mirror = 0x140.
// Must be called with all 64 lanes active (wave-uniform control flow): a disabled source lane leaves `old`.
template <int CTRL> __device__ __forceinline__ int mov(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL> __device__ __forceinline__ float mov(float v) {
  return __builtin_bit_cast(float, mov<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL> __device__ __forceinline__ double mov(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)mov<CTRL>((int)(unsigned)(u & 0xffffffffull)), hi = (unsigned)mov<CTRL>((int)(unsigned)(u >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int get(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float get(float v, int l) { return __builtin_bit_cast(float, get(__builtin_bit_cast(int, v), l)); }
__device__ __forceinline__ double get(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)get((int)(unsigned)(u & 0xffffffffull), l), hi = (unsigned)get((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
struct Max {
  template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return b > a ? b : a; }
};
struct Min {
  template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return b < a ? b : a; }
};
struct Add {
  template <typename T> __device__ __forceinline__ T operator()(T a, T b) const { return a + b; }
};
// all-reduce over the wave; the result is wave-uniform (it comes out of v_readlane)
template <typename T, typename Op> __device__ __forceinline__ T allreduce(T v, Op op) {
  v = op(v, mov<0xB1>(v));
  v = op(v, mov<0x4E>(v));
  v = op(v, mov<0x141>(v));
  v = op(v, mov<0x140>(v));
  return op(op(get(v, 0), get(v, 16)), op(get(v, 32), get(v, 48)));
}
// same when only lanes 0..15 hold non-neutral values: the first row's result is the wave's
template <typename T, typename Op> __device__ __forceinline__ T allreduce_row0(T v, Op op) {
  v = op(v, mov<0xB1>(v));
  v = op(v, mov<0x4E>(v));
  v = op(v, mov<0x141>(v));
  v = op(v, mov<0x140>(v));
  return get(v, 0);
}
template <bool ROW0, typename T, typename Op> __device__ __forceinline__ T allreduce_n(T v, Op op) {
  if (ROW0) return allreduce_row0(v, op);
  return allreduce(v, op);
}
}  // namespace wv

// 3-way partition of the envs by last step's solve cost: order[c*E + k] = k-th env of class c, count[c].
// One workgroup, coalesced strided passes (thread t owns envs t, t + 1024, ...).
__global__ __launch_bounds__(1024) void k_cbf_order(const int E, const int* __restrict__ cost, int* __restrict__ order,
                                                    int* __restrict__ count) {
  __shared__ int wtot[3][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int c[3] = {0, 0, 0};
#pragma unroll 8
  for (int e = tid; e < E; e += 1024) {
    const int it = cost[e];
    c[0] += it >= kCbfHeavyIters;
    c[1] += it >= kCbfMediumIters && it < kCbfHeavyIters;
    c[2] += it < kCbfMediumIters;
  }
  int off[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {                            // exclusive scan over the 1024 threads
    int v = c[k];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(v, d);
      if (lane >= d) v += o;
    }
    if (lane == 63) wtot[k][wave] = v;
    off[k] = v - c[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int base = 0, total = 0;
    for (int w = 0; w < 16; ++w) {
      if (w < wave) base += wtot[k][w];
      total += wtot[k][w];
    }
    off[k] += base;
    if (tid == 0) count[k] = total;
  }
#pragma unroll 8
  for (int e = tid; e < E; e += 1024) {
    const int it = cost[e], cls = it >= kCbfHeavyIters ? 0 : (it >= kCbfMediumIters ? 1 : 2);
    const int pos = cls == 0 ? off[0]++ : (cls == 1 ? off[1]++ : off[2]++);
    order[cls * E + pos] = e;
  }
}

template <typename T> struct GiEps;
template <> struct GiEps<float> {
  static constexpr float z = 1e-9f, r = 1e-6f, inf = 3.0e38f;
};
template <> struct GiEps<double> {
  static constexpr double z = 1e-16, r = 1e-12, inf = 1.0e300;
};

// one unit-norm row  sum_k ca[k] u[NV ia + k] + cb[k] u[NV ib + k] <= b,  ij = ia | ib << 8
template <typename T, int NV> struct CbfRow {
  T ca[NV], cb[NV], b;
  int ij;
};

// NV = QP variables per agent (order 2: thrust only -> 1; order 3: yank, wx, wy -> 3, wz is box-only),
// NMAX = compile-time bound on the number of QP variables n = NV * D (LDS footprint of Q, R ~ NMAX^2),
// R = rows per lane.  One wavefront (= one env) per workgroup.
template <typename T, typename S, int R, int NMAX, int ORDER>
__global__ __launch_bounds__(64) void k_cbf_filter_gi(const CbfParams<T> P, const int E, const T kf, const int* __restrict__ pair_ij,
                                                      const T* __restrict__ obstacles, const S* __restrict__ obs,
                                                      const S* __restrict__ xdes, const S* __restrict__ unom,
                                                      S* __restrict__ usafe, int* __restrict__ status, const int max_iter,
                                                      const T tol2, const int* __restrict__ order_in,
                                                      const int* __restrict__ count_in, int* __restrict__ cost_out) {
  constexpr int NV = ORDER == 2 ? 1 : 3;
  constexpr int XD = ORDER == 2 ? 9 : 10;
  constexpr bool PRE = NMAX * sizeof(T) <= 128;   // a lane's rows of Q and R fit in registers: one LDS round trip per step instead of 2q
  constexpr int kQS = PRE ? ((NMAX + 3) / 4 * 4 + 4) : NMAX + 1;   // LDS row stride: 16-byte rows (4 mod 16 dwords) / odd, both conflict-free column walks
  constexpr int DMAX = NMAX / NV;
  constexpr int NOBS_L = (DMAX * 20 + 63) / 64, NXD_L = (DMAX * XD + 63) / 64, NUN_L = (DMAX * 4 + 63) / 64;
  static_assert(sizeof(CbfRow<T, NV>) * R * 64 >= sizeof(S) * DMAX * 20, "raw obs staging aliases the row table");
  __shared__ T sx[DMAX * XD], sxd[DMAX * XD];
  __shared__ T su[NMAX], sd[NMAX], slam[NMAX], sdi[NMAX];
  __shared__ T sQ[NMAX][kQS], sR[NMAX][kQS];
  __shared__ int sact[NMAX];
  __shared__ T sob[kCbfMaxObs * 4];
  __shared__ T swz[2][DMAX];
  __shared__ __align__(16) CbfRow<T, NV> srow[R * 64];
  S* sraw = reinterpret_cast<S*>(srow);                     // the env's observation rows, staged before the rows are built
  const int lane = threadIdx.x;
  // Longest-first dispatch: the solve time of an env is ~ its number of active rows, which changes slowly from one
  // control step to the next.  Every wave records its iteration count; every few launches k_cbf_order bins the envs into
  // cost classes, and the launches walk the classes heaviest first, so the few long solves start at t = 0 instead of
  // forming the kernel's tail.  (One atomic per wave on a shared counter would serialise: 12 ns each, measured.)
  int env = blockIdx.x;
  if (order_in) {
    const int c0 = count_in[0], c1 = count_in[1];
    const int b = blockIdx.x;
    env = b < c0 ? order_in[b] : (b < c0 + c1 ? order_in[E + b - c0] : order_in[2 * E + b - c0 - c1]);
  }
  if (env >= E) return;
#if defined(MDS_TUNE_ITERS)
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
  const int D = P.num_drones, n = NV * D;
  const size_t base = (size_t)env * D;
  const int npairs = cbf_num_pairs(D), nobs_rows = D * P.n_obs, m = npairs + nobs_rows + 2 * n;

  // ---- every global read of this env, issued back to back ----
  // an env's D x 20 observation block, its D x xdim xdes block and D x 4 nominal block are contiguous
  S robs[NOBS_L], rxd[NXD_L], run[NUN_L];
  int rpair[R];
#pragma unroll
  for (int j = 0; j < NOBS_L; ++j) robs[j] = lane + 64 * j < D * 20 ? obs[base * 20 + lane + 64 * j] : (S)0;
#pragma unroll
  for (int j = 0; j < NXD_L; ++j) rxd[j] = lane + 64 * j < D * XD ? xdes[base * XD + lane + 64 * j] : (S)0;
#pragma unroll
  for (int j = 0; j < NUN_L; ++j) run[j] = lane + 64 * j < D * 4 ? unom[base * 4 + lane + 64 * j] : (S)0;
#pragma unroll
  for (int k = 0; k < R; ++k) rpair[k] = lane + 64 * k < npairs ? pair_ij[lane + 64 * k] : 0;
  const T rob = lane < 4 * P.n_obs ? obstacles[lane] : T(0);
#pragma unroll
  for (int j = 0; j < NOBS_L; ++j)
    if (lane + 64 * j < D * 20) sraw[lane + 64 * j] = robs[j];
#pragma unroll
  for (int j = 0; j < NXD_L; ++j)
    if (lane + 64 * j < D * XD) sxd[lane + 64 * j] = (T)rxd[j];
#pragma unroll
  for (int j = 0; j < NUN_L; ++j) {
    const int k = lane + 64 * j;
    if (k < D * 4 && (k & 3) < NV) su[NV * (k >> 2) + (k & 3)] = (T)run[j];
  }
  sob[lane] = rob;
  MDS_WAVE_SYNC();
  bool bad = false;
  T wz_lo = -P.umax[3], wz_hi = P.umax[3];
  if (lane < D) {                             // obs_to_lin_model(obs, dim = 9 | 10) (model_conversions.py:20-58)
    const S* o = &sraw[lane * 20];
    T* x = &sx[lane * XD];
    x[0] = (T)o[7]; x[1] = (T)o[8]; x[2] = (T)o[9];
    if (ORDER == 3) {
      const T r0 = (T)o[16], r1 = (T)o[17], r2 = (T)o[18], r3 = (T)o[19];
      const T F = kf * (r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);                             // calc_z_thrust (:137-143)
      x[3] = F;
      // order 3: the omega_z input only has box rows, +-umax_3 and the force box written to its column
      // (custom_force_bound_const, cbf/cbf.py:446-464, quirk kept): a 1-D interval per agent
      wz_hi = m_min(wz_hi, P.k[2] * (P.Fmax - F));
      wz_lo = m_max(wz_lo, -(P.k[2] * (F - P.Fmin)));
      if (wz_lo > wz_hi) bad = true;
    }
    x[XD - 6] = (T)o[10]; x[XD - 5] = (T)o[11]; x[XD - 4] = (T)o[12];
    x[XD - 3] = (T)o[0]; x[XD - 2] = (T)o[1]; x[XD - 1] = (T)o[2];
    swz[0][lane] = wz_lo;
    swz[1][lane] = wz_hi;
  }
  MDS_WAVE_SYNC();                            // sraw is dead from here on: srow may overwrite it

  // unit-norm rows, R per lane: registers for the violation scan, LDS copy for the broadcast fetch
  T ca[R][NV], cb[R][NV], b[R];
  int ia[R], ib[R];
  bool valid[R], act[R];
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int r = lane + 64 * k;
#pragma unroll
    for (int v = 0; v < NV; ++v) ca[k][v] = cb[k][v] = T(0);
    b[k] = T(0);
    ia[k] = ib[k] = 0;
    valid[k] = false;
    act[k] = false;
    if (r < npairs) {
      ia[k] = rpair[k] & 255;
      ib[k] = rpair[k] >> 8;
      T hr, Lg[4];
      cbf_pair_row<T, ORDER>(P, &sx[ia[k] * XD], &sxd[ia[k] * XD], &sx[ib[k] * XD], &sxd[ib[k] * XD], false, P.Ds_pair, &hr, Lg);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        ca[k][v] = -Lg[v];
        cb[k][v] = Lg[v];
      }
      b[k] = hr;
    } else if (r < npairs + nobs_rows) {
      const int q = r - npairs, i = (q * P.obs_magic) >> 16, o = q - i * P.n_obs;         // i = q / n_obs, exact for q < 4096
      T xo[XD];
#pragma unroll
      for (int v = 0; v < XD - 3; ++v) xo[v] = T(0);
      xo[XD - 3] = sob[4 * o];
      xo[XD - 2] = sob[4 * o + 1];
      xo[XD - 1] = sob[4 * o + 2];
      T hr, Lg[4];
      cbf_pair_row<T, ORDER>(P, &sx[i * XD], &sxd[i * XD], xo, xo, true, P.safety_radius + sob[4 * o + 3], &hr, Lg);
      ia[k] = ib[k] = i;
#pragma unroll
      for (int v = 0; v < NV; ++v) ca[k][v] = -Lg[v];
      b[k] = hr;
    } else if (r < m) {                                    // +-u_var <= umax (cbf/cbf.py:400-412)
      const int q = r - npairs - nobs_rows, var = q < n ? q : q - n;
      const int ag = var / NV, vv = var - ag * NV;
      ia[k] = ib[k] = ag;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (v == vv) {
          ca[k][v] = q < n ? T(1) : T(-1);
          b[k] = P.umax[v];
        }
    }
    if (r < m) {
      T n2 = T(0);
#pragma unroll
      for (int v = 0; v < NV; ++v) n2 = m_fma(ca[k][v], ca[k][v], m_fma(cb[k][v], cb[k][v], n2));
      if (n2 > T(0)) {
        const T inv = m_rsqrt(n2);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          ca[k][v] *= inv;
          cb[k][v] *= inv;
        }
        b[k] *= inv;
        valid[k] = true;
        // a barrier row that not even the corner of the input box satisfies makes the QP infeasible by itself: min over the box
        // of a.u is -sum |a_v| umax_v.  Typical case: an agent almost level with what it must avoid (tiny L_g, very negative h).
        // Exact (the box rows are rows of the same QP) and it spares the solver the 10-16 iterations it needs to find out.
        if (r < npairs + nobs_rows) {
          T reach = T(0);
#pragma unroll
          for (int v = 0; v < NV; ++v) reach = m_fma(m_abs(ca[k][v]) + m_abs(cb[k][v]), P.umax[v], reach);
          if (b[k] < -reach * (T(1) + T(sizeof(T) == 4 ? 1e-5 : 1e-10))) bad = true;
        }
      } else if (b[k] < T(0)) {
        bad = true;                                        // 0 * u <= h with h < 0
      }
      CbfRow<T, NV> w;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        w.ca[v] = ca[k][v];
        w.cb[v] = cb[k][v];
      }
      w.b = b[k];
      w.ij = ia[k] | (ib[k] << 8);
      srow[r] = w;
    }
  }
  MDS_WAVE_SYNC();
  bool converged = false;
  bool infeasible = __any(bad);
  int q = 0, it = 0;
  while (!infeasible && it < max_iter) {
    // ---- most violated row outside the active set (distance^2 to its half-space) ----
    T best = T(0);
    int best_k = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      T res = -b[k];
#pragma unroll
      for (int v = 0; v < NV; ++v) res = m_fma(ca[k][v], su[NV * ia[k] + v], m_fma(cb[k][v], su[NV * ib[k] + v], res));
      const T sc = (valid[k] && !act[k] && res > T(0)) ? res * res : T(0);
      if (sc > best) {
        best = sc;
        best_k = k;
      }
    }
    const T wbest = wv::allreduce(best, wv::Max());
    if (!(wbest > tol2)) {
      converged = true;
      break;
    }
    const int owner = (int)__builtin_ctzll(__ballot(best == wbest));                         // ties: lowest lane, then its lowest row
    const int kk = wv::get(best_k, owner), wrow = owner + 64 * kk;
    const CbfRow<T, NV> wr = srow[wrow];                                                   // uniform address: one broadcast read
    T wca[NV], wcb[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      wca[v] = wr.ca[v];
      wcb[v] = wr.cb[v];
    }
    const T wb = wr.b;
    const int wia = wr.ij & 255, wib = wr.ij >> 8;
    const bool two = wib != wia;
    T lam_new = T(0);
    // ---- bring that row into the active set, dropping blocking rows on the way ----
    while (true) {
      if (++it > max_iter) {
        infeasible = true;
        break;
      }
      // ---- everything this step reads from LDS, in one round trip ----
      T res = -wb;
#pragma unroll
      for (int v = 0; v < NV; ++v) res = m_fma(wca[v], su[NV * wia + v], m_fma(two ? wcb[v] : T(0), su[NV * wib + v], res));
      T dc = T(0);
      if (lane < q) {                                                                    // d = Q^T a
#pragma unroll
        for (int v = 0; v < NV; ++v) dc = m_fma(wca[v], sQ[NV * wia + v][lane], m_fma(two ? wcb[v] : T(0), sQ[NV * wib + v][lane], dc));
      }
      const int ln = lane < NMAX ? lane : NMAX - 1;                                      // lanes >= n hold no row: clamp the address only
      const T my_lam = slam[ln], my_di = sdi[ln], my_u = su[ln];
      T zv = T(0), rc = dc;
      if constexpr (PRE) {
        T qrow[NMAX], rrow[NMAX];
#pragma unroll
        for (int c = 0; c < NMAX; ++c) {
          qrow[c] = sQ[ln][c];
          rrow[c] = sR[ln][c];
        }
        if (lane < n) {                                                                  // z = a - Q d
          const int ag = lane / NV, vv = lane - ag * NV;
#pragma unroll
          for (int v = 0; v < NV; ++v)
            if (v == vv) zv = (ag == wia ? wca[v] : T(0)) + ((two && ag == wib) ? wcb[v] : T(0));
        }
#pragma unroll
        for (int c = 0; c < NMAX; ++c) {                                                 // one exit test per column; the compiler keeps it a loop
          if (c >= q) break;                                                             // with indexed VGPR reads of qrow (no scratch)
          zv = m_fma(-qrow[c], wv::get(dc, c), zv);                                      // lanes >= n: garbage, masked below
        }
        if (lane >= n) zv = T(0);
        // r = R^-1 d by back substitution on the row-scaled system (row l divided by its pivot, off the serial chain):
        // per step one v_readlane and one fma -- lane k's entry is final when step k reads it
#pragma unroll
        for (int c = 0; c < NMAX; ++c) rrow[c] *= my_di;
        rc *= my_di;
#pragma unroll
        for (int k = NMAX - 1; k >= 0; --k)
          if (k < q) {
            const T rk = wv::get(rc, k);
            rc = lane < k ? m_fma(-rrow[k], rk, rc) : rc;
          }
      } else {
        if (lane < n) sd[lane] = dc;
        MDS_WAVE_SYNC();
        if (lane < n) {
          const int ag = lane / NV, vv = lane - ag * NV;
#pragma unroll
          for (int v = 0; v < NV; ++v)
            if (v == vv) zv = (ag == wia ? wca[v] : T(0)) + ((two && ag == wib) ? wcb[v] : T(0));
          for (int c = 0; c < q; ++c) zv = m_fma(-sQ[lane][c], sd[c], zv);
        }
        for (int k = q - 1; k >= 0; --k) {
          const T rk = wv::get(rc, k) * sdi[k];
          if (lane == k) rc = rk;
          else if (lane < k) rc = m_fma(-sR[lane][k], rk, rc);
        }
      }
      constexpr bool ROW0 = NMAX <= 16;                                                  // z, r, lambda live in lanes 0..n-1 only
      const T zz = wv::allreduce_n<ROW0>(zv * zv, wv::Add());
      const T rmax = wv::allreduce_n<ROW0>(lane < q ? m_abs(rc) : T(0), wv::Max());
      T t1v = GiEps<T>::inf;
      if (lane < q && rc > GiEps<T>::r * rmax && rc > T(0)) t1v = m_max(my_lam, T(0)) * m_rcp(rc);
      const T t1 = wv::allreduce_n<ROW0>(t1v, wv::Min());
      const int drop = t1 < GiEps<T>::inf ? (int)__builtin_ctzll(__ballot(t1v == t1)) : 0;  // ties: lowest column
      const bool has_z = zz > GiEps<T>::z;
      const T t2 = has_z ? res * m_rcp(zz) : GiEps<T>::inf;
      const T t = m_min(t1, t2);
      if (!(t < GiEps<T>::inf)) {
        infeasible = true;                                                               // no step possible: rows inconsistent
        break;
      }
      const bool full = has_z && t2 <= t1;
      MDS_WAVE_SYNC();
      if (has_z && lane < n) su[lane] = m_fma(-t, zv, my_u);
      if (lane < q) slam[lane] = m_fma(-t, rc, my_lam);
      lam_new += t;
      if (!full) MDS_WAVE_SYNC();                                                        // the drop path reads slam / sact next; the add path only writes
      if (full) {                                                                        // add: N <- [N a]
        const T inz = m_rsqrt(zz), nz = zz * inz;
        if (lane < n) sQ[lane][q] = zv * inz;
        if (lane < q) sR[lane][q] = dc;
        if (lane == 0) {
          sR[q][q] = nz;
          sdi[q] = inz;
          slam[q] = lam_new;
          sact[q] = wrow;
        }
        if (lane == owner) {
#pragma unroll
          for (int k = 0; k < R; ++k)
            if (k == kk) act[k] = true;
        }
        ++q;
        MDS_WAVE_SYNC();
        break;
      }
      // ---- drop active column `drop` (its multiplier reached zero) ----
      const int drow = sact[drop];
      if (lane == (drow & 63)) {
#pragma unroll
        for (int k = 0; k < R; ++k)
          if (k == (drow >> 6)) act[k] = false;
      }
      T lnext = T(0);
      int anext = 0;
      if (lane >= drop && lane < q - 1) {
        lnext = slam[lane + 1];
        anext = sact[lane + 1];
      }
      MDS_WAVE_SYNC();
      if (lane >= drop && lane < q - 1) {
        slam[lane] = lnext;
        sact[lane] = anext;
      }
      if (lane < q)                                                                      // each lane shifts its own row of R
        for (int k = drop; k < q - 1; ++k) sR[lane][k] = sR[lane][k + 1];
      MDS_WAVE_SYNC();
      for (int l = drop; l < q - 1; ++l) {                                               // Givens on rows l, l+1
        const T a = sR[l][l], bb = sR[l + 1][l];
        const T rr = m_sqrt(m_fma(a, a, bb * bb));
        const T cs = rr > T(0) ? a / rr : T(1), sn = rr > T(0) ? bb / rr : T(0);
        MDS_WAVE_SYNC();
        if (lane >= l && lane < q - 1) {
          const T x = sR[l][lane], y = sR[l + 1][lane];
          sR[l][lane] = m_fma(cs, x, sn * y);
          sR[l + 1][lane] = m_fma(-sn, x, cs * y);
        }
        if (lane < n) {
          const T x = sQ[lane][l], y = sQ[lane][l + 1];
          sQ[lane][l] = m_fma(cs, x, sn * y);
          sQ[lane][l + 1] = m_fma(-sn, x, cs * y);
        }
        MDS_WAVE_SYNC();
      }
      --q;
      if (lane >= drop && lane < q) sdi[lane] = T(1) / sR[lane][lane];
      MDS_WAVE_SYNC();
    }
  }
  if (converged) {
    // final certificate: EVERY row (active ones included) holds at the returned point.  Guards the
    // near-dependent / infeasible corner where a step along a numerically tiny z is taken.
    T worst = T(0);
#pragma unroll
    for (int k = 0; k < R; ++k) {
      T res = -b[k];
#pragma unroll
      for (int v = 0; v < NV; ++v) res = m_fma(ca[k][v], su[NV * ia[k] + v], m_fma(cb[k][v], su[NV * ib[k] + v], res));
      if (valid[k]) worst = m_max(worst, res);
    }
    worst = wv::allreduce(worst, wv::Max());
    if (worst * worst > T(100) * tol2) converged = false;
  }
#if defined(MDS_TUNE_ITERS)   // tuning build: iteration count and final active-set size in the high bits of status
  {
    const int nbox = __popcll(__ballot(lane < q && sact[lane < NMAX ? lane : 0] >= npairs + nobs_rows));
    if (lane == 0) status[env] = (converged ? 0 : 1) | ((it & 0x7f) << 1) | ((q & 0x1f) << 8) | ((nbox & 0x1f) << 13) |
                                 ((int)m_min<unsigned long long>((__builtin_amdgcn_s_memtime() - t_start) >> 8, 0x1fffull) << 18);
  }
#else
  if (lane == 0) status[env] = converged ? 0 : 1;
#endif
  if (cost_out && lane == 0) cost_out[env] = it;
  MDS_WAVE_SYNC();
  // u_safe in the flat [D,4] layout of the nominal block that is still in registers
#pragma unroll
  for (int j = 0; j < NUN_L; ++j) {
    const int k = lane + 64 * j;
    if (k < D * 4) {
      const int d = k >> 2, c = k & 3;
      T u = (T)run[j];
      if (converged) {
        if (c < NV) {
          u = su[NV * d + c];
        } else if (ORDER == 2) {
          const T um = c == 1 ? P.umax[1] : (c == 2 ? P.umax[2] : P.umax[3]);
          u = m_clamp(u, -um, um);
        } else {
          u = m_clamp(u, swz[0][d], swz[1][d]);
        }
      }
      usafe[base * 4 + k] = (S)u;
    }
  }
}

}  // namespace mds
