// ECBF safety filter kernels (a11-a15): constraint rows of cbf/cbf.py and the QP of
// cbf/qptracker.py:86-114, one wavefront per environment.
#include <hip/hip_runtime.h>
#include <type_traits>

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
// passes that bound the rows are infeasible and the env keeps u_hat with status 1.  That exit is this
// library's MODELLED policy, not a restatement: the reference returns u_hat only when cvxopt.solvers.qp
// raises (qptracker.py:30-34, :103-112), and cvxopt's coneqp returns status 'unknown' + its last iterate
// on an infeasible QP instead of raising (include/mds.h, mds_cbf_filter).  The iteration cap is a backstop.
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

// keeps two loaded values live at this point (an empty asm the optimiser cannot look through): used to stop it from sinking LDS
// reads into the conditional code that consumes them
// MDS_KEEP_V / MDS_KEEP_S: an empty asm the optimiser cannot look through, on a vector / scalar register value.  In a host build (the
// SIMT emulation of tests/emul/simt: these kernels under AddressSanitizer on the CPU) they are no-ops.
#if defined(__HIP_DEVICE_COMPILE__)
#define MDS_KEEP_V(x) asm volatile("" : "+v"(x))
#define MDS_KEEP_S(x) asm volatile("" : "+s"(x))
#else
#define MDS_KEEP_V(x) ((void)0)
#define MDS_KEEP_S(x) ((void)0)
#endif
#define MDS_PIN2(a, b) \
  do {                 \
    MDS_KEEP_V(a);     \
    MDS_KEEP_V(b);     \
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
// Max / min of NON-NEGATIVE values (squared residuals, |r|, step lengths; never -0, never NaN): fp32 of that kind order like their
// bit patterns, so the reduction runs on v_max_i32 / v_min_i32 with the DPP read as the instruction's own operand (old = 0 with
// bound_ctrl, the form the compiler folds) -- two instructions per step instead of copy, DPP move, compare, select.  double keeps
// the generic path.
template <int CTRL> __device__ __forceinline__ int dpp0(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
struct IMax {
  __device__ __forceinline__ int operator()(int a, int b) const { return a > b ? a : b; }
};
struct IMin {
  __device__ __forceinline__ int operator()(int a, int b) const { return a < b ? a : b; }
};
template <bool ROW0, typename Op> __device__ __forceinline__ int allreduce_bits(int v, Op op) {
  v = op(v, dpp0<0xB1>(v));
  v = op(v, dpp0<0x4E>(v));
  v = op(v, dpp0<0x141>(v));
  v = op(v, dpp0<0x140>(v));
  if (ROW0) return get(v, 0);
  return op(op(get(v, 0), get(v, 16)), op(get(v, 32), get(v, 48)));
}
template <bool ROW0> __device__ __forceinline__ float max_nonneg(float v) {
  return __builtin_bit_cast(float, allreduce_bits<ROW0>(__builtin_bit_cast(int, v), IMax()));
}
template <bool ROW0> __device__ __forceinline__ float min_nonneg(float v) {
  return __builtin_bit_cast(float, allreduce_bits<ROW0>(__builtin_bit_cast(int, v), IMin()));
}
template <bool ROW0> __device__ __forceinline__ double max_nonneg(double v) { return allreduce_n<ROW0>(v, Max()); }
template <bool ROW0> __device__ __forceinline__ double min_nonneg(double v) { return allreduce_n<ROW0>(v, Min()); }

// f(c) for c = min(q, HI) - 1 down to LO with c a compile-time constant inside f (static register indices), q wave-uniform: a
// binary search on q (log2 tests) and then straight-line steps, instead of one guarded step -- scalar compare + branch -- per
// possible c.  f takes std::integral_constant-like tags.
template <int C> struct Ic {
  static constexpr int value = C;
};
template <int LO, int HI, typename F> __device__ __forceinline__ void desc_all(F&& f) {
  if constexpr (HI > LO) {
    f(Ic<HI - 1>{});
    desc_all<LO, HI - 1>(f);
  }
}
template <int LO, int HI, typename F> __device__ __forceinline__ void desc_upto(const int q, F&& f) {
  if constexpr (HI - LO == 1) {
    if (q > LO) f(Ic<LO>{});
  } else if constexpr (HI > LO) {
    constexpr int MID = (LO + HI) / 2;
    if (q > MID) {
      desc_upto<MID, HI>(q, f);
      desc_all<LO, MID>(f);
    } else {
      desc_upto<LO, MID>(q, f);
    }
  }
}
}  // namespace wv

// 3-way partition of the envs by last step's solve cost: order[c*E + k] = k-th env of class c, count[c].
// One workgroup, coalesced strided passes (thread t owns envs t, t + kOrderThreads, ...).
#ifndef MDS_ORDER_THREADS
#define MDS_ORDER_THREADS 256
#endif
constexpr int kOrderThreads = MDS_ORDER_THREADS;   // one small workgroup: it has to find room on a CU beside the other chain's QP waves
__global__ __launch_bounds__(kOrderThreads) void k_cbf_order(const int E, const int* __restrict__ cost, int* __restrict__ order,
                                                    int* __restrict__ count) {
  __shared__ int wtot[3][kOrderThreads / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int c[3] = {0, 0, 0};
#pragma unroll 8
  for (int e = tid; e < E; e += kOrderThreads) {
    const int it = cost[e];
    c[0] += it >= kCbfHeavyIters;
    c[1] += it >= kCbfMediumIters && it < kCbfHeavyIters;
    c[2] += it < kCbfMediumIters;
  }
  int off[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {                            // exclusive scan over the workgroup's threads
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
    for (int w = 0; w < kOrderThreads / 64; ++w) {
      if (w < wave) base += wtot[k][w];
      total += wtot[k][w];
    }
    off[k] += base;
    if (tid == 0) count[k] = total;
  }
#pragma unroll 8
  for (int e = tid; e < E; e += kOrderThreads) {
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

// The active-set iterations of k_cbf_filter_gi on rows already built: R unit-norm rows per lane in registers (ca, cb, b, ia, ib,
// valid), the QP point u in LDS (su, n = NV * D variables), the thin QR of the active normals and the multipliers in the wave's LDS
// scratch.  ROWTAB: the selected row is fetched from an LDS copy of the rows (srow) by a uniform address; without it, from the owning
// lane's registers (a select over its R rows + v_readlane) -- 16 bytes of LDS per row less.  Returns converged / iterations; su holds
// the minimiser when converged.  Must be called with all 64 lanes active.
// SMALLQ > 0 (thrust-only QPs, NV == 1): active sets of up to SMALLQ rows live in REGISTERS as wave-uniform values -- a row of this QP
// has at most two coefficients, so everything a step needs of the active set (d = N^T a, the Gram matrix N^T N of <= 3 x 3, r = its
// solve in closed form, the multipliers, the drop bookkeeping) is arithmetic on uniform operands that every lane runs alike: no Q, R
// or multipliers in LDS, no column reads, no back-substitution chain of v_readlane, one wave reduction (|z|^2) per step instead of
// three.  The step is the dual active-set step of the general path (solving the Gram system of the active normals directly, as the
// plain-C checker of the tests does); a set that needs row SMALLQ + 1 is written out once as the thin QR the general path continues on.  The census of the C4
// scenes: 0.76 tight pair rows per env-step on `under`, 2.6 iterations per env-step that iterates -- sets of 1-3 rows
// are what the solver meets.  Unit-norm rows of this QP have coefficients +-1 (bounds) or +-1/sqrt2 (pairs): the Gram matrix of an
// independent set is well conditioned (entries 0, +-1/2, +-1/sqrt2); should its determinant still come out tiny, the set is handed
// to the QR path before the step is taken.
#ifndef MDS_TUNE_HASZ
#define MDS_TUNE_HASZ 1
#endif
#ifndef MDS_GI_SMALLQ
#define MDS_GI_SMALLQ 0      // (measured on MI355X, round 4: 1 / 2 / 3 are 3 / 8 / 14 % SLOWER on C4 -- see DESIGN.md 4; kept for A/B)
#endif
template <typename T, int R, int NMAX, int NV, bool ROWTAB, int kQS, int SMALLQ = (NV == 1 ? MDS_GI_SMALLQ : 0)>
__device__ __forceinline__ void gi_solve(const int lane, const int n, const int max_iter, const T tol2, const bool infeasible0,
                                         const T (&ca)[R][NV], const T (&cb)[R][NV], const T (&b)[R], const int (&ia)[R], const int (&ib)[R],
                                         const bool (&valid)[R], bool (&act)[R], T* __restrict__ su, T* __restrict__ sd,
                                         T* __restrict__ slam, T* __restrict__ sdi, T (*__restrict__ sQ)[kQS], T (*__restrict__ sR)[kQS],
                                         int* __restrict__ sact, const CbfRow<T, NV>* __restrict__ srow, bool& converged_out, int& it_out,
                                         int& q_out) {
  constexpr bool PRE = NMAX * sizeof(T) <= 128;   // a lane's rows of Q and R fit in registers: one LDS round trip per step instead of 2q
  static_assert(kQS == (PRE ? ((NMAX + 3) / 4 * 4 + 4) : NMAX + 1), "LDS row stride of Q and R");
  static_assert(SMALLQ >= 0 && SMALLQ <= 3 && (SMALLQ == 0 || NV == 1), "register-resident active sets: thrust-only QPs, at most 3 rows");
  bool converged = false;
  bool infeasible = infeasible0;
  int q = 0, it = 0;
  // the register-resident active set (uniform): row k < q is  ra[k] u[ri[k]] + rb[k] u[rj[k]] <= .. (rb = 0, rj = ri on a one-variable
  // row), multiplier rl[k], row id rid[k] (lane + 64 slot of the owner); g..: its Gram matrix
  constexpr int QR = SMALLQ > 0 ? SMALLQ : 1;
  bool small = SMALLQ > 0;
  T ra[QR], rb[QR], rl[QR];
  int ri[QR], rj[QR], rid[QR];
  T g00 = T(1), g11 = T(1), g22 = T(1), g01 = T(0), g02 = T(0), g12 = T(0);
#pragma unroll
  for (int k = 0; k < QR; ++k) {
    ra[k] = rb[k] = rl[k] = T(0);
    ri[k] = rj[k] = rid[k] = 0;
  }
  while (!infeasible && it < max_iter) {
    // ---- most violated row outside the active set (distance^2 to its half-space) ----
    T best = T(0);
    int best_k = 0;

    T ua[R][NV], ub[R][NV];
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        ua[k][v] = su[NV * ia[k] + v];
        ub[k][v] = su[NV * ib[k] + v];
      }
#pragma unroll
    for (int k = 0; k < R; ++k)                            // every row's operands in ONE LDS round trip: left to itself the compiler sinks each
#pragma unroll
      for (int v = 0; v < NV; ++v) MDS_PIN2(ua[k][v], ub[k][v]);     // pair of reads under its row's `valid` test, R round trips in a row
#pragma unroll
    for (int k = 0; k < R; ++k) {
      T res = -b[k];
#pragma unroll
      for (int v = 0; v < NV; ++v) res = m_fma(ca[k][v], ua[k][v], m_fma(cb[k][v], ub[k][v], res));
      const T sc = (valid[k] && !act[k] && res > T(0)) ? res * res : T(0);
      if (sc > best) {
        best = sc;
        best_k = k;
      }
    }
    // No lane holds a violated row: converged.  The same decision as "wave maximum <= tol2" below, from one ballot instead of the DPP
    // reduction + four v_readlane + compare chain -- the exit most envs take most steps (the scan of an env that needs no iteration
    // cost as much as building its rows: 150 serial instructions, VALU -> scalar -> branch round trips).
    if (!__any(best > tol2)) {
      converged = true;
      break;
    }
    const T wbest = wv::max_nonneg<false>(best);
    if (!(wbest > tol2)) {
      converged = true;
      break;
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MDS_TUNE_NO_SETPRIO)
    // an env that iterates is on the launch's critical path (a serial chain of ~4 000-cycle steps, the envs that do not iterate
    // are throughput work): its wave takes the SIMD's issue slots first from here on
    __builtin_amdgcn_s_setprio(3);
#endif
    const int owner = (int)__builtin_ctzll(__ballot(best == wbest));                         // ties: lowest lane, then its lowest row
    const int kk = wv::get(best_k, owner), wrow = owner + 64 * kk;
    T wca[NV], wcb[NV], wb;
    int wia, wib;
    if constexpr (ROWTAB) {
      const CbfRow<T, NV> wr = srow[wrow];                                                 // uniform address: one broadcast read
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        wca[v] = wr.ca[v];
        wcb[v] = wr.cb[v];
      }
      wb = wr.b;
      wia = wr.ij & 255;
      wib = wr.ij >> 8;
    } else {                                                                               // no row table in LDS: the owner's registers, row kk
      T sb_ = b[0];
      int sia = ia[0], sib = ib[0];
#pragma unroll
      for (int k = 1; k < R; ++k) {
        sb_ = kk == k ? b[k] : sb_;
        sia = kk == k ? ia[k] : sia;
        sib = kk == k ? ib[k] : sib;
      }
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        T sa = ca[0][v], sc = cb[0][v];
#pragma unroll
        for (int k = 1; k < R; ++k) {
          sa = kk == k ? ca[k][v] : sa;
          sc = kk == k ? cb[k][v] : sc;
        }
        wca[v] = wv::get(sa, owner);
        wcb[v] = wv::get(sc, owner);
      }
      wb = wv::get(sb_, owner);
      wia = wv::get(sia, owner);
      wib = wv::get(sib, owner);
    }
    const bool two = wib != wia;
    T lam_new = T(0);
    // ---- bring that row into the active set, dropping blocking rows on the way ----
    while (true) {
      if (++it > max_iter) {
        infeasible = true;
        break;
      }
      constexpr bool ROW0 = NMAX <= 16;                                                  // z, r, lambda live in lanes 0..n-1 only
      // ---- everything this step reads from LDS, in one round trip ----
      T res = -wb;
#pragma unroll
      for (int v = 0; v < NV; ++v) res = m_fma(wca[v], su[NV * wia + v], m_fma(two ? wcb[v] : T(0), su[NV * wib + v], res));
      const int ln = lane < NMAX ? lane : NMAX - 1;                                      // lanes >= n hold no row: clamp the address only
      const int lu = lane < n ? lane : n - 1;                                            // (u: inside this env's n variables -- past them lie another env's, which another wave may be writing)
      if constexpr (SMALLQ > 0) {
        if (small) {
          const T my_u = su[lu];
          const T na = wca[0], nb = two ? wcb[0] : T(0);
          auto ncoef = [&](int v) { return (v == wia ? na : T(0)) + ((two && v == wib) ? nb : T(0)); };   // the new row's coefficient at variable v
          // d = N^T a, r = (N^T N)^-1 d
          T d[QR], r[QR];
#pragma unroll
          for (int k = 0; k < QR; ++k) {
            d[k] = k < q ? m_fma(ra[k], ncoef(ri[k]), (rj[k] != ri[k]) ? rb[k] * ncoef(rj[k]) : T(0)) : T(0);
            r[k] = T(0);
          }
          T det = T(1);
          if (q == 1) {
            det = g00;
            r[0] = d[0] * m_rcp(g00);
          } else if (QR >= 2 && q == 2) {
            det = m_fma(g00, g11, -(g01 * g01));
            const T id = m_rcp(det);
            r[0] = m_fma(g11, d[0], -(g01 * d[QR >= 2 ? 1 : 0])) * id;
            r[QR >= 2 ? 1 : 0] = m_fma(g00, d[QR >= 2 ? 1 : 0], -(g01 * d[0])) * id;
          } else if (QR >= 3 && q == 3) {
            constexpr int i1 = QR >= 3 ? 1 : 0, i2 = QR >= 3 ? 2 : 0;
            const T c00 = m_fma(g11, g22, -(g12 * g12)), c01 = m_fma(g02, g12, -(g01 * g22)), c02 = m_fma(g01, g12, -(g02 * g11));
            const T c11 = m_fma(g00, g22, -(g02 * g02)), c12 = m_fma(g01, g02, -(g00 * g12)), c22 = m_fma(g00, g11, -(g01 * g01));
            det = m_fma(g00, c00, m_fma(g01, c01, g02 * c02));
            const T id = m_rcp(det);
            r[0] = m_fma(c00, d[0], m_fma(c01, d[i1], c02 * d[i2])) * id;
            r[i1] = m_fma(c01, d[0], m_fma(c11, d[i1], c12 * d[i2])) * id;
            r[i2] = m_fma(c02, d[0], m_fma(c12, d[i1], c22 * d[i2])) * id;
          }
          // a set at capacity, or a Gram determinant that rounding could own: continue on the thin QR in LDS (below)
          if (q == SMALLQ || !(det > T(sizeof(T) == 4 ? 1e-3 : 1e-6))) {
            for (int j = 0; j < q; ++j) {                                                // Gram-Schmidt append of row j, as the general step's
              T ja = ra[0], jb = rb[0], jl = rl[0];
              int ji = ri[0], jj = rj[0], jid = rid[0];
#pragma unroll
              for (int k = 1; k < QR; ++k)
                if (k == j) { ja = ra[k]; jb = rb[k]; jl = rl[k]; ji = ri[k]; jj = rj[k]; jid = rid[k]; }
              const bool jtwo = jj != ji;
              T dcj = m_fma(ja, sQ[ji][ln], jtwo ? jb * sQ[jj][ln] : T(0));
              dcj = lane < j ? dcj : T(0);
              T zj = T(0);
              if (lane < n) {
                zj = (lane == ji ? ja : T(0)) + ((jtwo && lane == jj) ? jb : T(0));
                for (int c = 0; c < j; ++c) zj = m_fma(-sQ[lane][c], wv::get(dcj, c), zj);
              }
              const T zzj = wv::allreduce_n<ROW0>(zj * zj, wv::Add());
              const T inzj = m_rsqrt(zzj), nzj = zzj * inzj;
              MDS_WAVE_SYNC();
              if (lane < n) sQ[lane][j] = zj * inzj;
              if (lane < j) sR[lane][j] = dcj;
              if (lane == 0) {
                sR[j][j] = nzj;
                sdi[j] = inzj;
                slam[j] = jl;
                sact[j] = jid;
              }
              MDS_WAVE_SYNC();
            }
            small = false;
          } else {
            T zv = T(0);
            if (lane < n) {
              zv = ncoef(lane);
#pragma unroll
              for (int k = 0; k < QR; ++k)
                if (k < q) zv = m_fma(-r[k], (lane == ri[k] ? ra[k] : T(0)) + ((rj[k] != ri[k] && lane == rj[k]) ? rb[k] : T(0)), zv);
            }
            const T zz = wv::allreduce_n<ROW0>(zv * zv, wv::Add());
            T rmax = T(0);
#pragma unroll
            for (int k = 0; k < QR; ++k) rmax = k < q ? m_max(rmax, m_abs(r[k])) : rmax;
            T t1 = GiEps<T>::inf;
            int drop = 0;
#pragma unroll
            for (int k = 0; k < QR; ++k)
              if (k < q && r[k] > GiEps<T>::r * rmax && r[k] > T(0)) {
                const T cand = m_max(rl[k], T(0)) * m_rcp(r[k]);
                if (cand < t1) {                                                         // ties: lowest column
                  t1 = cand;
                  drop = k;
                }
              }
            // a set of n independent normals spans the space: whatever is left of z is rounding, and a 'full step' along it would add a
            // DEPENDENT row (q > n: past the thin QR's columns -- found by the host emulation under UBSan on an fp32 order-3 env that then
            // cycled to the iteration cap); only the dual step is possible there
            const bool has_z = zz > GiEps<T>::z && (MDS_TUNE_HASZ != 1 || q < n);
            const T t2 = has_z ? res * m_rcp(zz) : GiEps<T>::inf;
            const T t = m_min(t1, t2);
            if (!(t < GiEps<T>::inf)) {
              infeasible = true;                                                         // no step possible: rows inconsistent
              break;
            }
            const bool full = has_z && t2 <= t1;
            MDS_WAVE_SYNC();
            if (has_z && lane < n) su[lane] = m_fma(-t, zv, my_u);
#pragma unroll
            for (int k = 0; k < QR; ++k) rl[k] = k < q ? m_fma(-t, r[k], rl[k]) : rl[k];
            lam_new += t;
            if (full) {                                                                  // add: N <- [N a]
              const T aa = m_fma(na, na, nb * nb);
#pragma unroll
              for (int k = 0; k < QR; ++k)
                if (k == q) { ra[k] = na; rb[k] = nb; rl[k] = lam_new; ri[k] = wia; rj[k] = two ? wib : wia; rid[k] = wrow; }
              if (q == 0) g00 = aa;
              else if (q == 1) { g01 = d[0]; g11 = aa; }
              else { g02 = d[0]; g12 = d[QR >= 2 ? 1 : 0]; g22 = aa; }
              if (lane == owner) {
#pragma unroll
                for (int k = 0; k < R; ++k)
                  if (k == kk) act[k] = true;
              }
              ++q;
              MDS_WAVE_SYNC();
              break;
            }
            // ---- drop active row `drop` (its multiplier reached zero) ----
            int drow = rid[0];
#pragma unroll
            for (int k = 1; k < QR; ++k) drow = k == drop ? rid[k] : drow;
            if (lane == (drow & 63)) {
#pragma unroll
              for (int k = 0; k < R; ++k)
                if (k == (drow >> 6)) act[k] = false;
            }
            if (QR >= 3 && q == 3) {                                                     // the Gram matrix without row / column `drop`
              if (drop == 0) { g00 = g11; g01 = g12; g11 = g22; }
              else if (drop == 1) { g01 = g02; g11 = g22; }
            } else if (q == 2 && drop == 0) {
              g00 = g11;
            }
#pragma unroll
            for (int k = 0; k + 1 < QR; ++k)
              if (k >= drop) { ra[k] = ra[k + 1]; rb[k] = rb[k + 1]; rl[k] = rl[k + 1]; ri[k] = ri[k + 1]; rj[k] = rj[k + 1]; rid[k] = rid[k + 1]; }
            --q;
            MDS_WAVE_SYNC();
            continue;
          }
        }
        // (here: the set was just written out as a thin QR -- this step, same row and same count, runs on it below)
      }
#if !defined(MDS_TUNE_GI_NO_FIRST_STEP)
      if (q == 0) {
        // Empty active set (the first step of most solves, and the only one of more than half of them): the step runs along the
        // row's own normal -- no Q, R or multipliers to read, no blocking row, no back substitution.  The general step below with
        // q = 0, operation for operation (same sums in the same order: same bits), at a third of its instructions.
        const T my_u = su[lu];
        T zv = T(0);
        if (lane < n) {
          const int ag = lane / NV, vv = lane - ag * NV;
#pragma unroll
          for (int v = 0; v < NV; ++v)
            if (v == vv) zv = (ag == wia ? wca[v] : T(0)) + ((two && ag == wib) ? wcb[v] : T(0));
        }
        const T zz = wv::allreduce_n<ROW0>(zv * zv, wv::Add());
        const bool has_z = zz > GiEps<T>::z;
        const T t = has_z ? res * m_rcp(zz) : GiEps<T>::inf;
        if (!(t < GiEps<T>::inf)) {
          infeasible = true;
          break;
        }
        MDS_WAVE_SYNC();
        if (lane < n) su[lane] = m_fma(-t, zv, my_u);
        lam_new += t;
        const T inz = m_rsqrt(zz), nz = zz * inz;
        if (lane < n) sQ[lane][0] = zv * inz;
        if (lane == 0) {
          sR[0][0] = nz;
          sdi[0] = inz;
          slam[0] = lam_new;
          sact[0] = wrow;
        }
        if (lane == owner) {
#pragma unroll
          for (int k = 0; k < R; ++k)
            if (k == kk) act[k] = true;
        }
        q = 1;
        MDS_WAVE_SYNC();
        break;
      }
#endif
      T dc = T(0);
      {                                                                                  // d = Q^T a: read by every lane (clamped column) so that the
        T qa[NV], qb[NV];                                                                // reads join the step's one round trip, kept by lanes < q
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          qa[v] = sQ[NV * wia + v][ln];
          qb[v] = sQ[NV * wib + v][ln];
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) MDS_PIN2(qa[v], qb[v]);
#pragma unroll
        for (int v = 0; v < NV; ++v) dc = m_fma(wca[v], qa[v], m_fma(two ? wcb[v] : T(0), qb[v], dc));
        dc = lane < q ? dc : T(0);
      }
      const T my_lam = slam[ln], my_di = sdi[ln], my_u = su[lu];
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
        // columns q-1 .. 0, each a v_readlane + fma on a statically indexed register (lanes >= n: garbage, masked below)
        wv::desc_upto<0, NMAX>(q, [&](auto c) { zv = m_fma(-qrow[decltype(c)::value], wv::get(dc, decltype(c)::value), zv); });
        if (lane >= n) zv = T(0);
        // r = R^-1 d by back substitution on the row-scaled system (row l divided by its pivot, off the serial chain):
        // per step one v_readlane and one fma -- lane k's entry is final when step k reads it
        rc *= my_di;
        wv::desc_upto<0, NMAX>(q, [&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const T rk = wv::get(rc, k);
          rc = lane < k ? m_fma(-(rrow[k] * my_di), rk, rc) : rc;
        });
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
      const T zz = wv::allreduce_n<ROW0>(zv * zv, wv::Add());
      const T rmax = wv::max_nonneg<ROW0>(lane < q ? m_abs(rc) : T(0));
      T t1v = GiEps<T>::inf;
      if (lane < q && rc > GiEps<T>::r * rmax && rc > T(0)) t1v = m_max(my_lam, T(0)) * m_rcp(rc);
      const T t1 = wv::min_nonneg<ROW0>(t1v);
      const int drop = t1 < GiEps<T>::inf ? (int)__builtin_ctzll(__ballot(t1v == t1)) : 0;  // ties: lowest column
      // a set of n independent normals spans the space: whatever is left of z is rounding, and a 'full step' along it would add a
            // DEPENDENT row (q > n: past the thin QR's columns -- found by the host emulation under UBSan on an fp32 order-3 env that then
            // cycled to the iteration cap); only the dual step is possible there
            const bool has_z = zz > GiEps<T>::z && (MDS_TUNE_HASZ != 1 || q < n);
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
        if (MDS_TUNE_HASZ == 2 && q >= n) {
          infeasible = true;
          break;
        }
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
  if (converged && it > 0) {
    // final certificate: EVERY row (active ones included) holds at the returned point.  Guards the
    // near-dependent / infeasible corner where a step along a numerically tiny z is taken.  (No iteration: the scan that
    // declared convergence has just checked every row at the nominal point.)  Round 4 tried folding it into the scan that declares
    // convergence (a running maximum over all valid rows, one ballot at the exit: the same decision): 3 % SLOWER on every C4 scene --
    // every env pays for the maximum in every scan, only the envs that iterated ever ran the certificate.
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
  converged_out = converged;
  it_out = it;
  q_out = q;
}

// NV = QP variables per agent (order 2: thrust only -> 1; order 3: yank, wx, wy -> 3, wz is box-only),
// NMAX = compile-time bound on the number of QP variables n = NV * D (LDS footprint of Q, R ~ NMAX^2),
// R = rows per lane.  One wavefront (= one env) per workgroup.
#ifndef MDS_GI_ROWTAB
#define MDS_GI_ROWTAB 1
#endif
#ifndef MDS_GI_MINWAVES
#define MDS_GI_MINWAVES 1
#endif
// One env's QP on one wavefront (lane = threadIdx.x & 63, all 64 lanes active): its D x 20 observation block, D x xdim xdes block and
// D x 4 nominal block in, its D x 4 u_safe block, status and iteration count out.  The blocks may live in global memory (k_cbf_filter_gi)
// or in LDS (k_cbf_rollout_o3 stages them per control step); the LDS arrays of the solver are this function's own.
template <typename T, typename S, int R, int NMAX, int ORDER>
__device__ __forceinline__ void cbf_filter_env(const CbfParams<T>& P, const int lane, const T kf, const int* __restrict__ pair_ij,
                                               const T* __restrict__ obstacles, const S* obs, const S* xdes, const S* unom, S* usafe,
                                               int* status_env, const int max_iter, const T tol2, int* cost_env) {
  constexpr int NV = ORDER == 2 ? 1 : 3;
  constexpr int XD = ORDER == 2 ? 9 : 10;
  constexpr bool PRE = NMAX * sizeof(T) <= 128;   // a lane's rows of Q and R fit in registers: one LDS round trip per step instead of 2q
  constexpr int kQS = PRE ? ((NMAX + 3) / 4 * 4 + 4) : NMAX + 1;   // LDS row stride: 16-byte rows (4 mod 16 dwords) / odd, both conflict-free column walks
  constexpr int DMAX = NMAX / NV;
  constexpr int NOBS_L = (DMAX * 20 + 63) / 64, NXD_L = (DMAX * XD + 63) / 64, NUN_L = (DMAX * 4 + 63) / 64;
  static_assert(!MDS_GI_ROWTAB || sizeof(CbfRow<T, NV>) * R * 64 >= sizeof(S) * DMAX * 20, "raw obs staging aliases the row table");
  __shared__ T sx[DMAX * XD], sxd[DMAX * XD];
  __shared__ T su[NMAX], sd[NMAX], slam[NMAX], sdi[NMAX];
  __shared__ T sQ[NMAX][kQS], sR[NMAX][kQS];
  __shared__ int sact[NMAX];
  __shared__ T sob[kCbfMaxObs * 4];
  __shared__ T swz[2][DMAX];
#if MDS_GI_ROWTAB
  __shared__ __align__(16) CbfRow<T, NV> srow[R * 64];
#else   // tuning build: no LDS copy of the rows (the selected row comes from its owner's registers); only the observation staging remains
  __shared__ __align__(16) CbfRow<T, NV> srow[(sizeof(S) * DMAX * 20 + sizeof(CbfRow<T, NV>) - 1) / sizeof(CbfRow<T, NV>)];
#endif
  S* sraw = reinterpret_cast<S*>(srow);                     // the env's observation rows, staged before the rows are built
#if defined(MDS_TUNE_ITERS)
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
  const int D = P.num_drones, n = NV * D;
  constexpr size_t base = 0;                    // (the pointers are this env's blocks)
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
    // every row needs the agents' tracking errors x - xdes, never xdes itself: formed once per agent, in place
    T* xe = &sxd[lane * XD];
#pragma unroll
    for (int v = 0; v < XD - 3; ++v) xe[v] = x[v] - xe[v];
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
    if (r < npairs + nobs_rows) {
      // agent-agent and agent-obstacle rows through ONE instance of the row arithmetic (a wave whose lanes straddle the two
      // kinds would otherwise run it twice): only the operands are selected
      const bool ob = r >= npairs;
      int i = rpair[k] & 255, j = rpair[k] >> 8;
      const T* pj = &sx[j * XD + XD - 3];
      T Ds = P.Ds_pair;
      if (ob) {
        const int q = r - npairs;
        i = (q * P.obs_magic) >> 16;                       // q / n_obs, exact for q < 4096
        const int o = q - i * P.n_obs;
        j = i;
        pj = &sob[4 * o];
        Ds = P.safety_radius + sob[4 * o + 3];
      }
      const T* xi = &sx[i * XD];
      const T *ei = &sxd[i * XD], *ej = &sxd[j * XD];
      T d[XD], hr, Lg[4];
#pragma unroll
      for (int v = 0; v < XD - 3; ++v) d[v] = ei[v] - (ob ? T(0) : ej[v]);
      cbf_row_core<T, ORDER>(P, xi[XD - 3] - pj[0], xi[XD - 2] - pj[1], xi[XD - 1] - pj[2], d, Ds, &hr, Lg);
      ia[k] = i;
      ib[k] = j;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        ca[k][v] = -Lg[v];
        cb[k][v] = ob ? T(0) : Lg[v];
      }
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
#if MDS_GI_ROWTAB
      srow[r] = w;
#endif
    }
  }
  MDS_WAVE_SYNC();
  bool converged = false;
  int q = 0, it = 0;
  gi_solve<T, R, NMAX, NV, MDS_GI_ROWTAB != 0, kQS>(lane, n, max_iter, tol2, __any(bad), ca, cb, b, ia, ib, valid, act, su, sd, slam, sdi, sQ, sR, sact, srow,
                                      converged, it, q);
#if defined(MDS_TUNE_ITERS)   // tuning build: iteration count and final active-set size in the high bits of status
  {
    const int nbox = __popcll(__ballot(lane < q && sact[lane < NMAX ? lane : 0] >= npairs + nobs_rows));
    if (lane == 0) *status_env = (converged ? 0 : 1) | ((it & 0x7f) << 1) | ((q & 0x1f) << 8) | ((nbox & 0x1f) << 13) |
                                 ((int)m_min<unsigned long long>((__builtin_amdgcn_s_memtime() - t_start) >> 8, 0x1fffull) << 18);
  }
#else
  if (lane == 0) *status_env = converged ? 0 : 1;
#endif
  if (cost_env && lane == 0) *cost_env = it;
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

template <typename T, typename S, int R, int NMAX, int ORDER>
__global__ __launch_bounds__(64, MDS_GI_MINWAVES) void k_cbf_filter_gi(const CbfParams<T> P, const int E, const T kf, const int* __restrict__ pair_ij,
                                                      const T* __restrict__ obstacles, const S* __restrict__ obs,
                                                      const S* __restrict__ xdes, const S* __restrict__ unom,
                                                      S* __restrict__ usafe, int* __restrict__ status, const int max_iter,
                                                      const T tol2, const int* __restrict__ order_in,
                                                      const int* __restrict__ count_in, int* __restrict__ cost_out) {
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
#if !defined(MDS_TUNE_NO_SETPRIO)
    // the envs that iterated last time: on the critical path from their first instruction
    if (b < c0) __builtin_amdgcn_s_setprio(3);
    else if (b < c0 + c1) __builtin_amdgcn_s_setprio(2);
#endif
  }
  if (env >= E) return;
  constexpr int XD = ORDER == 2 ? 9 : 10;
  const size_t base = (size_t)env * P.num_drones;
  cbf_filter_env<T, S, R, NMAX, ORDER>(P, lane, kf, pair_ij, obstacles, obs + base * 20, xdes + base * XD, unom + base * 4, usafe + base * 4, status + env,
                                       max_iter, tol2, cost_out ? cost_out + env : nullptr);
}

// Row r of an order-2 env whose drones' world positions / tracking errors sit in LDS rows d0 .. d0 + D - 1 (k_cbf_step): the
// unit-norm row  ca u[ia] + cb u[ib] <= b, whether it exists (vld), and whether it makes the QP infeasible by itself (bad: a
// barrier row beyond the reach of the input box, or 0 * u <= h with h < 0).  Pair and obstacle rows are the same barrier with
// different second operands: the operands are selected and the ~100-op row body runs once per slot, whatever kinds it straddles.
template <typename T>
__device__ __forceinline__ void cbf_o2_slot(const CbfParams<T>& P, const T (*__restrict__ pos)[3], const T (*__restrict__ de)[5],
                                            const T* __restrict__ sob, const int d0, const int r, const int pair, const int npairs,
                                            const int nobs_rows, const int m, const int n, T& ca, T& cb, T& b, int& ia, int& ib, bool& vld,
                                            bool& bad) {
  ca = cb = b = T(0);
  ia = ib = 0;
  vld = false;
  const bool is_pair = r < npairs, is_obs = !is_pair && r < npairs + nobs_rows;
  if (is_pair || is_obs) {
    const int qo = r - npairs, ag = is_pair ? (pair & 255) : ((qo * P.obs_magic) >> 16);          // ag = q / n_obs, exact for q < 4096
    const int oo = is_pair ? 0 : qo - ag * P.n_obs, bg = is_pair ? (pair >> 8) : ag;
    const T* pi = pos[d0 + ag];
    const T* di = de[d0 + ag];
    const T* pj = is_pair ? pos[d0 + bg] : &sob[4 * oo];
    const T* dj = de[d0 + bg];
    const T z = T(0);
    T hr, lg;
    cbf_row_o2<T>(P, pi[0] - pj[0], pi[1] - pj[1], pi[2] - pj[2], di[0] - (is_pair ? dj[0] : z), di[1] - (is_pair ? dj[1] : z),
                  di[2] - (is_pair ? dj[2] : z), di[3] - (is_pair ? dj[3] : z), di[4] - (is_pair ? dj[4] : z),
                  is_pair ? P.Ds_pair : P.safety_radius + sob[4 * oo + 3], &hr, &lg);
    ia = ag;
    ib = bg;
    ca = -lg;
    cb = is_pair ? lg : T(0);
    b = hr;
  } else if (r < m) {                                          // +-u_var <= umax (cbf/cbf.py:400-412)
    const int q = r - npairs - nobs_rows, var = q < n ? q : q - n;
    ia = ib = var;
    ca = q < n ? T(1) : T(-1);
    b = P.umax[0];
  }
  if (r < m) {
    const T n2 = m_fma(ca, ca, cb * cb);
    if (n2 > T(0)) {
      const T inv = m_rsqrt(n2);
      ca *= inv;
      cb *= inv;
      b *= inv;
      vld = true;
      if (r < npairs + nobs_rows) {                            // beyond the reach of the input box: infeasible by itself (see k_cbf_filter_gi)
        const T reach = (m_abs(ca) + m_abs(cb)) * P.umax[0];
        if (b < -reach * (T(1) + T(sizeof(T) == 4 ? 1e-5 : 1e-10))) bad = true;
      }
    } else if (b < T(0)) {
      bad = true;                                              // 0 * u <= h with h < 0
    }
  }
}

// ------------------------------------------------------------------------------------
// One CBF-filtered control step of simulations/CBFTest.py:303-350 in ONE launch (order 2, D | 64, D <= 16):
//   stage A, one drone per lane : trajs[j](t), nominal controller (GeometricControl return_omegas, NOM 0, or LQROmegaController,
//                                 NOM 1) -> u_hat = (force - M G, w), and what the rows need of obs_to_lin_model(obs) - xdes: the
//                                 world position and the tracking errors in roll, pitch, velocity;
//   stage B, one env at a time  : the wave's 64 / D envs in turn -- rows built R per lane from the LDS copy of stage A's outputs,
//                                 most-violated-row scan, gi_solve (rows stay in registers: no row table);
//   stage C, one drone per lane : u_safe (nominal if the env's QP has no solution: the modelled fallback of mds_cbf_filter) + M G -> ThrustOmega low level
//                                 -> physics step -> observation row -> state.
// The three-launch path (nominal / k_cbf_filter_gi / k_lowlevel_step) moves u_hat, xdes, u_safe and the state through HBM
// twice and runs the per-drone stages at full lane use but the QP kernel one env per wave; here the per-drone stages keep every lane
// busy (a wave = 64 drones = 64 / D whole envs) and nothing but state, trajectory parameters and the observation crosses HBM.
// Same arithmetic as the three kernels (shared device functions); x is formed from the state exactly as pack_obs would hand it over.
// Measured at C4 (16 384 x 16, MI355X, profiles/r02_c4_fused.md): 32 us against 42 us per control step when no env needs an
// iteration, but 74-80 us against 56 us on SURVEY 8d's scene, where a third of the envs iterate: 4096 waves is 4 per SIMD, each a
// serial chain of four QPs (10.6 cycles per instruction per wave, VALU pipe 48 % busy), while the three-launch path gives the
// latency-bound QP 16 384 waves and a longest-first dispatch.  Hence opt-in: MDS_CBF_FUSED=1 at mds_cbf_configure time.
// ------------------------------------------------------------------------------------
template <typename T, int R, int NOM, bool COMP>
__global__ __launch_bounds__(64, (sizeof(T) == 4 && R == 4) ? 4 : 1) void k_cbf_step(const Consts<T> c, const CbfParams<T> P, const void* __restrict__ Kp, const int n_end,
                                                 const size_t ld, const int E, const double t, const T ctrl_dt, T* __restrict__ state,
                                                 T* __restrict__ state_lo, const T* __restrict__ lem, T* __restrict__ last_rpm,
                                                 T* __restrict__ ll, const int* __restrict__ pair_ij, const T* __restrict__ obstacles,
                                                 T* __restrict__ obs, int* __restrict__ status, int* __restrict__ cost_out,
                                                 const int max_iter, const T tol2, const int batch0) {
  constexpr int NMAX = 16, NV = 1;
  constexpr int kQS = (NMAX + 3) / 4 * 4 + 4;
  // LDS of the wave: [ solver scratch + stage A's per-drone outputs | aliased by the observation staging of stage C ] [ su | sconv ]
  struct Work {
    T pos[64][3], de[64][5];
    T sd[NMAX], slam[NMAX], sdi[NMAX];
    T sQ[NMAX][kQS], sR[NMAX][kQS];
    int sact[NMAX];
    T sob[kCbfMaxObs * 4];
  };
  constexpr size_t kObsBytes = (size_t)64 * kObsDim * sizeof(T);
  constexpr size_t kWork = sizeof(Work) > kObsBytes ? sizeof(Work) : kObsBytes;
  __shared__ __align__(16) unsigned char raw[(kWork + 15) / 16 * 16];
  __shared__ T su_all[64];
  __shared__ int sconv[64];
  Work& W = *reinterpret_cast<Work*>(raw);
  const int lane = threadIdx.x;
  const int D = P.num_drones, G = 64 / D;
  const int i = (batch0 + blockIdx.x) * 64 + lane;
  const bool valid = i < n_end;
  const int env0 = ((batch0 + blockIdx.x) * 64) / D;

  // ---- stage A: nominal controller of drone i ----
  GeoIn<T> in;
  T un[4] = {T(0), T(0), T(0), T(0)};
  int rpair[R];
#pragma unroll
  for (int k = 0; k < R; ++k) rpair[k] = lane + 64 * k < cbf_num_pairs(D) ? pair_ij[lane + 64 * k] : 0;
  W.sob[lane] = lane < 4 * P.n_obs ? obstacles[lane] : T(0);
  if (valid) {
    load_geo_in<T, T>(state, lem, ld, i, in);
    const Desired<T> des = lemniscate_local(in.P, t);
    const V3<T> rpy = euler_from_quat(in.s.q);
    if (NOM == 0) {
      const M3<T> Rm = quat_to_rot(in.s.q);
      const V3<T> ang_v = mul(Rm, in.s.w);
      T u[4];
      GeoAux<T> A;
      geometric_control<T>(c, in.s.p - des.p, Rm, in.s.v, ang_v, des, u, &A);
      un[0] = A.force - c.gravity;
      un[1] = A.w_des.x; un[2] = A.w_des.y; un[3] = A.w_des.z;
    } else {
      T u[4];
      lqr_omega_control<T>(c, *static_cast<const LqrGain<T>*>(Kp), rpy, in.s.v, in.s.p, des.p, des.v, des.yaw, u);
      un[0] = u[0] - c.gravity;                                                    // CBFTest.py:339
      un[1] = u[1]; un[2] = u[2]; un[3] = u[3];
    }
    // obs_to_lin_model(obs, 9) - xdes with obs = pack_obs(state): rpy, world velocity, world position; xdes = [0, 0, yaw, v_des, p_des]
    W.pos[lane][0] = in.s.p.x + in.P.cx;
    W.pos[lane][1] = in.s.p.y + in.P.cy;
    W.pos[lane][2] = in.s.p.z + in.P.cz;
    W.de[lane][0] = rpy.x - T(0);
    W.de[lane][1] = rpy.y - T(0);
    W.de[lane][2] = in.s.v.x - des.v.x;
    W.de[lane][3] = in.s.v.y - des.v.y;
    W.de[lane][4] = in.s.v.z - des.v.z;
    su_all[lane] = un[0];
  }
  MDS_WAVE_SYNC();

  // ---- stage B: the wave's envs, one after the other ----
  const int n = D;                                             // QP variables per env (order 2: the thrusts)
  const int npairs = cbf_num_pairs(D), nobs_rows = D * P.n_obs, m = npairs + nobs_rows + 2 * n;
  for (int g = 0; g < G; ++g) {
    const int env = env0 + g;
    if (env >= E) break;                                       // wave-uniform
    const int d0 = g * D;                                      // first lane / LDS row of this env's drones
    T ca[R][NV], cb[R][NV], b[R];
    int ia[R], ib[R];
    bool vld[R], act[R];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      act[k] = false;
      cbf_o2_slot<T>(P, W.pos, W.de, W.sob, d0, lane + 64 * k, rpair[k], npairs, nobs_rows, m, n, ca[k][0], cb[k][0], b[k], ia[k], ib[k], vld[k], bad);
    }
    bool converged = false;
    int it = 0, q = 0;
    gi_solve<T, R, NMAX, NV, false, kQS>(lane, n, max_iter, tol2, __any(bad), ca, cb, b, ia, ib, vld, act, &su_all[d0], W.sd, W.slam, W.sdi, W.sQ,
                                         W.sR, W.sact, nullptr, converged, it, q);
    if (lane == 0) {
      status[env] = converged ? 0 : 1;
      if (cost_out) cost_out[env] = it;
    }
    if (lane >= d0 && lane < d0 + D) sconv[lane] = converged ? 1 : 0;
    MDS_WAVE_SYNC();
  }

  // ---- stage C: low level + physics of drone i ----
  // The state and the trajectory centre are read again here rather than held in 20 registers across stage B (146 -> 4 waves per
  // SIMD); the pointers go through an empty asm so that the compiler does not keep the first copies alive instead.
  T o[kObsDim];
  T safe = T(0);
  int conv = 0;
  if (valid) {
    safe = su_all[lane];
    conv = sconv[lane];
  }
  MDS_WAVE_SYNC();                                             // raw is the observation staging from here on
  const T* state2 = state;
  const T* lem2 = lem;
  MDS_KEEP_S(state2); MDS_KEEP_S(lem2);
  GeoIn<T> in2;
  if (valid) {
    load_geo_in<T, T>(state2, lem2, ld, i, in2);
    if (COMP) load_resid<T, T>(state_lo, ld, i, in2.r);
    T u[4];
    u[0] = (conv ? safe : un[0]) + c.gravity;                                              // CBFTest.py:346
    u[1] = conv ? m_clamp(un[1], -P.umax[1], P.umax[1]) : un[1];
    u[2] = conv ? m_clamp(un[2], -P.umax[2], P.umax[2]) : un[2];
    u[3] = conv ? m_clamp(un[3], -P.umax[3], P.umax[3]) : un[3];
    LowLevelState<T> L;
    L.last_omega = {ll[0 * ld + i], ll[1 * ld + i], ll[2 * ld + i]};
    L.integral = {ll[3 * ld + i], ll[4 * ld + i], ll[5 * ld + i]};
    T act4[4], prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4];
    thrust_omega_control(c, ctrl_dt, u, in2.s.w, L, act4);
    ll[0 * ld + i] = L.last_omega.x; ll[1 * ld + i] = L.last_omega.y; ll[2 * ld + i] = L.last_omega.z;
    ll[3 * ld + i] = L.integral.x; ll[4 * ld + i] = L.integral.y; ll[5 * ld + i] = L.integral.z;
    aviary_step_any<T, false, false, COMP>(c, in2.s, in2.r, act4, prev, clipped);
    if (last_rpm)
      for (int k = 0; k < 4; ++k) last_rpm[k * ld + i] = clipped[k];
    pack_obs(in2.s, V3<T>{in2.P.cx, in2.P.cy, in2.P.cz}, clipped, o);
  }
  write_obs_rows<T, T>(raw, obs, n_end, i, valid, o);
  if (valid) {
    store_state<T, T>(state, ld, i, in2.s);
    if (COMP) store_resid<T, T>(state_lo, ld, i, in2.r);
  }
}

// ------------------------------------------------------------------------------------
// n_steps CBF-filtered control steps of simulations/CBFTest.py:303-350 in ONE launch (order 2, any D <= 16): the persistent form of
// k_cbf_step.  A workgroup of NW wavefronts owns 64 NW drones = 64 NW / D whole envs for the whole launch and walks them through
//   stage A, one drone per lane : trajs[j](t), nominal controller -> u_hat; the drone's record (world position, tracking errors in
//                                 roll, pitch, velocity: what the rows need of obs_to_lin_model(obs) - xdes) into LDS;
//   stage B, one env per wave   : the workgroup's envs handed out through an LDS ticket counter, heaviest (by their last step's
//                                 iteration count) first -- rows built R per lane, violation scan, gi_solve;
//   stage C, one drone per lane : u_safe + M G -> ThrustOmega low level -> physics step -> observation row (every step into
//                                 slot (slot0 + k) % n_slots of the log ring, or only the last one) -> stage A of the next step,
// with a workgroup barrier before and after stage B only.  Stage C of step k and stage A of step k + 1 are one straight piece of
// per-lane code with the state in registers (rotation matrix and Euler angles of the new state are formed once); across stage B the
// state waits in LDS, u_hat in registers; the trajectory parameters and the low level's PID memory are re-read from their global
// planes each step (L2 / Infinity-Cache hits: only this lane touches them).
// LDS: every wave has a 5 KiB slice = [its 64 drones' records | its solver scratch], which is also its observation staging in stage
// C -- a wave's staging overwrites nothing another wave still needs once stage B is over, so stage C needs no barrier of its own.
// Row slots: row r = lane + 64 k of every env has the same kind, agents and obstacle for a given (lane, k): a 16-byte entry of a
// table built once per launch replaces the per-env index decoding; an agent's record is two 16-byte LDS reads, an obstacle is a
// record with zero tracking errors, so pair and obstacle rows run one branch-free body; box rows are constants.
// What this buys over one launch per step: no chip-wide step boundary (workgroups drift apart: the launch ends with the slowest
// WORKGROUP's sum over all steps instead of every step ending with its slowest wave); an env that needs 10 iterations delays one
// wave of its workgroup while the other waves take the remaining envs (k_cbf_step: the three other envs of its wave wait);
// u_hat, xdes, u_safe and the state never cross HBM between steps.  Arithmetic: the device functions of the other CBF kernels
// (cbf_row_o2, the normalisation and reach test of cbf_o2_slot, gi_solve, the controller / physics templates).
// t advances in double exactly like the host loop (t += CTRL_TIMESTEP, CBFTest.py:352).
// ------------------------------------------------------------------------------------
#ifndef MDS_CBF_ROLL_NW
#define MDS_CBF_ROLL_NW 8
#endif
#ifndef MDS_ROLL_STAMPS
#define MDS_ROLL_STAMPS 0       // 1: per-stage shader-clock stamps compiled in (profiles/tools/r03_stamps.sh builds such a library)
#endif
#ifndef MDS_TUNE_ROLL_PRIO_C
#define MDS_TUNE_ROLL_PRIO_C 2     // issue priority of stages C and A (row build 0, scan / bookkeeping 1, an iterating solve 3); 0..3 measure within 2 % of each other
#endif
#ifndef MDS_TUNE_ROLL_SKIP
#define MDS_TUNE_ROLL_SKIP 0   // tuning aid (cost breakdown of stage B): 1 no row polynomial, 2 no normalisation, 4 no scan / solve, 8 no rows
#endif
#ifndef MDS_ROLL_F64_WAVES
#define MDS_ROLL_F64_WAVES 2      // waves per SIMD the float64 instantiation is compiled for (2: <= 256 VGPRs, two 4-wave workgroups per CU run side by side)
#endif
#ifndef MDS_ROLL_OBS_CHUNK
#define MDS_ROLL_OBS_CHUNK 2
#endif
#ifndef MDS_ROLL_BOUNDS
#define MDS_ROLL_BOUNDS 1      // 1: obstacle and thrust-box rows folded into per-drone bounds in the drone-per-lane stage (round 4); 0: round 3's row layout (A/B)
#endif
struct RollSlot {       // row r = lane + 64 k of any env: what it is, as the row build consumes it (built once per launch by roll_slot_of)
  int a0;               // barrier rows: byte offset of agent i's record half 0 from the env's first record (swizzle applied); else 0
  int b0;               // pair rows: the same for agent j; other rows: byte offset of an obstacle record (obstacle o, or the first) in the
                        // workgroup's LDS block
  int ds;               // byte offset in the LDS block of -Ds^4: pair distance 2 safety_radius, or safety_radius + r_o
  int kind;             // 0 none, 1 pair, 2 obstacle, 3 box row +u <= umax, 4 box row -u <= umax;  | agent i << 8 | agent j << 16 | pair << 24
};

// rec: record stride in bytes; swz: agents 8..15 keep their record halves swapped (16-byte halves, D = 16 only: see k_cbf_rollout);
// sob / dso: byte offsets of the obstacle records and of the -Ds^4 table in the LDS block; wT: sizeof(T)
template <typename T>
__device__ __forceinline__ RollSlot roll_slot_of(const CbfParams<T>& P, const int* __restrict__ pair_ij, const int r, const int rec, const bool swz,
                                                 const int sob, const int dso, const bool bounds = false) {
  // bounds: the single-variable rows (obstacles, thrust box) are per-drone bounds made in stage A; the QP sees the pair rows, then
  // D rows +u_var <= hi_var (kind 3) and D rows -u_var <= -lo_var (kind 4)
  const int D = P.num_drones, npairs = cbf_num_pairs(D), nobs_rows = bounds ? 0 : D * P.n_obs, m = npairs + nobs_rows + 2 * D;
  auto half0 = [&](int ag) { return ag * rec + ((swz && ((ag >> 3) & 1)) ? 16 : 0); };
  RollSlot sl = {0, sob, dso, 0};
  if (r < npairs) {
    const int ij = pair_ij[r], ia = ij & 255, ib = ij >> 8;
    sl = {half0(ia), half0(ib), dso, 1 | (ia << 8) | (ib << 16) | (1 << 24)};
  } else if (r < npairs + nobs_rows) {
    const int q = r - npairs, ag = (q * P.obs_magic) >> 16, oo = q - ag * P.n_obs;       // as cbf_o2_slot
    sl = {half0(ag), sob + oo * rec, dso + (1 + oo) * (int)sizeof(T), 2 | (ag << 8) | (ag << 16)};
  } else if (r < m) {
    const int q = r - npairs - nobs_rows, var = q < D ? q : q - D;
    sl = {0, sob, dso, (q < D ? 3 : 4) | (var << 8) | (var << 16)};
  }
  return sl;
}

// Kernel arguments are loop invariants of a kernel that never leaves its step loop: left alone, the compiler hoists every VGPR copy
// a VALU instruction with two scalar operands needs (and every address it can form) out of the loop, keeps them live across all
// three stages and spills them; held in SGPRs across all stages (65 dwords of constants beside 14 pointers) they overflowed the
// 106 SGPRs into VGPR lanes and every use paid a v_readlane (133 in stage C alone).  k_cbf_rollout therefore reads its arguments
// -- the constants included (RollArgs::p) -- through the kernarg segment pointer, made fresh per stage: scalar loads of the fields
// a stage uses, where it uses them.
template <typename T> struct RollParams {
  Consts<T> c;
  CbfParams<T> P;
};
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define MDS_CONST_AS __attribute__((address_space(4)))
#else
#define MDS_CONST_AS            // (host build of the SIMT emulation: a plain pointer)
#endif
// a by-value copy of a struct behind a constant-address-space pointer, word by word (the words that are used become scalar loads,
// the others disappear)
template <typename V> __device__ __forceinline__ V load_const(const V MDS_CONST_AS* p) {
  static_assert(sizeof(V) % 4 == 0, "word-sized struct");
  constexpr int N = (int)(sizeof(V) / 4);
  const unsigned MDS_CONST_AS* w = reinterpret_cast<const unsigned MDS_CONST_AS*>(p);
  unsigned tmp[N];
#pragma unroll
  for (int k = 0; k < N; ++k) tmp[k] = w[k];
  V out;
  __builtin_memcpy(&out, tmp, sizeof(V));
  return out;
}

// The arguments of k_cbf_rollout, passed as ONE struct by value: the kernel never touches the parameter itself but reads the fields
// through the kernarg segment pointer, made fresh per stage (fresh_args) -- scalar loads where a field is used, instead of 40 SGPRs
// of loop invariants that the step loop kept spilling to VGPR lanes (a v_readlane, often with wait states, per use).
template <typename T> struct RollArgs {
  RollParams<T> p;                // drone constants, controller gains, CBF parameters
  const void* Kp;
  int n;
  size_t ld;
  int E;
  double t, ctrl_dt;
  int n_steps;
  T *state, *state_lo;
  const T* lem;
  T *last_rpm, *ll;
  const int* pair_ij;
  const T* obstacles;
  T* obs_log;
  int slot, n_slots;
  T* obs_last;
  int *status, *status_log, *cost_io;
  int max_iter;
  T tol2;
  unsigned long long* stamps;
  T tol;                          // sqrt(tol2): the distance by which a drone's bound interval may be empty before its env is infeasible
};
template <typename T> __device__ __forceinline__ const RollArgs<T> MDS_CONST_AS* fresh_args() {
  const RollArgs<T> MDS_CONST_AS* p = (const RollArgs<T> MDS_CONST_AS*)__builtin_amdgcn_kernarg_segment_ptr();
  MDS_KEEP_S(p);
  return p;
}

// element idx of a per-lane plane behind a UNIFORM base pointer, addressed by a 32-bit byte offset: the access compiles to the
// scalar-base + 32-bit-VGPR-offset form, one VGPR of address shared by every plane of the stage (64-bit per-lane addresses, one
// pair per plane, are what the step loop of k_cbf_rollout spilled).  idx * sizeof(U) < 2^32: planes of at most 2^28 doubles.
template <typename U> __device__ __forceinline__ U* lane_ptr(U* uniform_base, unsigned idx) {
  using B = std::conditional_t<std::is_const_v<U>, const unsigned char, unsigned char>;
  return reinterpret_cast<U*>(reinterpret_cast<B*>(uniform_base) + idx * (unsigned)sizeof(U));
}

// PAD: D is not 4, 8 or 16 -- an env is padded to the next of those widths (false: the padded and the real index coincide, and the
// index arithmetic, the staging row and its guard fold away: the any-D form costs the C4 shape 2-3 %, measured).
template <typename T, int NOM, bool COMP, int NW, bool PAD = true>
__global__ __launch_bounds__(64 * NW, sizeof(T) == 4 ? 4 : MDS_ROLL_F64_WAVES) void k_cbf_rollout(const RollArgs<T>) {

  constexpr int NT = 64 * NW;
  constexpr bool kBounds = MDS_ROLL_BOUNDS != 0;
  constexpr int R = kBounds ? 3 : 4, NMAX = 16, NV = 1;            // rows per lane: 120 pair rows + 32 bound rows <= 192 (round 3: 216 rows)
  constexpr int kQS = (NMAX + 3) / 4 * 4 + 4;
  constexpr int GBMAX = NT / 4;                                // envs per workgroup (D >= 4)
  // One drone's record: two 16-byte halves (px py e_pitch -e_roll | e_vx e_vy pz e_vz) -- the operand pairs of cbf_row_o2_pairs side
  // by side --, read by the row slots as two ds_read_b128 per agent.  At the plain 8-word stride agents a and a + 8 start in the same
  // bank, and a 16-lane group of such a read that walks 16 consecutive agents would take two passes: with D = 16, agents 8..15 keep
  // their halves in swapped order (half h of agent a sits at 16 (h ^ (a >> 3)) bytes), so that group reads 16 x 4 different banks;
  // envs of 4 or 8 drones span at most 64 banks as they stand.  (First version: 8 scalar fields at 8 words -- 124 M bank-conflict
  // cycles against 81 M LDS-instruction cycles per launch; then a 9-word stride, conflict-free but 16 ds_read2_b32 per row slot: the
  // LDS pipe was ~45 % busy and every LDS access of the kernel queued behind them.)
  constexpr int kRec = 8 * (int)sizeof(T);
  constexpr bool kSwz = sizeof(T) == 4;
  struct alignas(16) V4 {
    T v[4];
  };
  struct Scratch {                                             // one wave's active-set solver
    T sd[NMAX], slam[NMAX], sdi[NMAX];
    T sQ[NMAX][kQS], sR[NMAX][kQS];
    int sact[NMAX];
  };
  struct alignas(16) Slice {                                   // one wave: stage A -> B data, aliased by its stage C observation staging
    T rec[64][8];
    Scratch sc;
  };
  constexpr int kObsWave = 64 * kObsDim * (int)sizeof(T);      // write_obs_rows' slice per wave
  static_assert(sizeof(Slice) <= (size_t)kObsWave, "a wave's records + solver scratch must fit in its observation staging slice");
  static_assert(kObsWave % (16 * kRec) == 0, "an env's first record sits at a multiple of D * kRec bytes (pair rows OR their offsets in)");
  // the LDS block the row build addresses by byte offset: [NW wave slices | obstacle records | -Ds^4 table]
  constexpr int kSobOff = NW * kObsWave, kDsOff = kSobOff + kCbfMaxObs * kRec, kBlock = kDsOff + (kCbfMaxObs + 1 + 3) / 4 * 4 * (int)sizeof(T);
  __shared__ __align__(128) unsigned char raw[kBlock];
  T(*const sobrec)[8] = reinterpret_cast<T(*)[8]>(raw + kSobOff);     // obstacles as records with zero tracking errors (halves in plain order)
  T* const sDs = reinterpret_cast<T*>(raw + kDsOff);                  // -Ds^4: [0] pairs, [1 + o] obstacle o
  __shared__ T st[14][NT];                                     // the state and u_hat[0] across stage B (lane-contiguous planes: conflict-free)
  __shared__ T su_all[NT];                                     // thrust variable of every drone: u_hat[0] in, QP minimiser out
  // Per-drone bounds of the thrust variable (kBounds): every obstacle row (cbf/cbf.py:369-398) and the thrust box (:400-412) of order 2
  // constrain u[4 i] of drone i alone and depend on drone i's own state only, so stage A -- one drone per lane, state in registers --
  // evaluates them and folds them into  u_i <= sbnd[0][i]  and  -u_i <= sbnd[1][i];  the QP of stage B sees the pair rows and these two
  // rows per drone.  An obstacle row beyond the reach of the box, or an empty interval, writes -inf to both: the env is infeasible
  // before any row of it is built.
  __shared__ T sbnd[kBounds ? 2 : 1][kBounds ? NT : 1];
  __shared__ __align__(16) RollSlot stab[R][64];               // row slot table
  __shared__ int sconv[GBMAX], scost[GBMAX], sorder[GBMAX];    // per env of the workgroup: QP solved; iterations of its last solve; hand-out order
  __shared__ int sticket;
  // tuning aid (a build with -DMDS_ROLL_STAMPS=1, stamps != NULL, MDS_TUNE_ROLL_STAMPS=1 on the host): shader-clock ticks every wave
  // spent in each part of the step, summed over the launch's steps -> stamps[(workgroup * NW + wave) * 10 + part]: 0 wait for stage
  // B, 1 stage B, 2 wait after stage B, 3 stage C, 4 observation rows, 5 stage A; within B: 6 ticket, 7 rows, 8 bookkeeping, 9 scan +
  // solve.  Compiled out of the product: the two running stamps are loop-carried 64-bit values that the step loop otherwise spills.
#if MDS_ROLL_STAMPS
  constexpr bool kStamps = true;
#else
  constexpr bool kStamps = false;
#endif
  const RollArgs<T> MDS_CONST_AS* a0 = fresh_args<T>();        // (the arguments: read through the kernarg segment where they are used)
  const int n_steps = a0->n_steps;
  double t = a0->t;
  int slot = a0->slot;
  unsigned long long* const stamps = kStamps ? a0->stamps : nullptr;
  __shared__ unsigned long long sst[kStamps ? NW : 1][10];
  unsigned long long tk0 = 0;
  unsigned long long tk1 = 0;
  auto stamp_b = [&](int part) {
    if (kStamps && stamps != nullptr) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      if ((threadIdx.x & 63) == 0) sst[threadIdx.x >> 6][part] += now - tk1;
      tk1 = now;
    }
  };
  auto stamp = [&](int part) {
    if (kStamps && stamps != nullptr) {
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      if ((threadIdx.x & 63) == 0) sst[threadIdx.x >> 6][part] += now - tk0;
      tk0 = now;
    }
  };
  if (kStamps && stamps != nullptr && threadIdx.x < NW * 10) sst[threadIdx.x / 10][threadIdx.x % 10] = 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Any D <= 16 (round 4; the reference's own scripts run 2 and 7 drones: simulations/CBFTest.py:31, CBFTestOrd3.py:31): an env occupies
  // Dp = 4, 8 or 16 consecutive lanes (the next power of two, at least 4), the lanes past its D drones idle.  Everything in LDS is indexed
  // by the padded lane (records, state planes, bounds, su); global planes and observation rows by the drone's real index i = env D + d.
  const int D = a0->p.P.num_drones;
  const int log2P = D <= 4 ? 2 : (D <= 8 ? 3 : 4), Dp = 1 << log2P, GB = NT >> log2P;      // padded env width; envs per workgroup
  const int env0 = blockIdx.x * GB;
  const int i = PAD ? (env0 + (tid >> log2P)) * D + (tid & (Dp - 1)) : (int)(blockIdx.x * NT) + tid;
  const bool valid = PAD ? ((tid & (Dp - 1)) < D && env0 + (tid >> log2P) < a0->E) : i < a0->n;
  const int nenv = min(GB, a0->E - env0);                          // envs this workgroup really owns (>= 1: the grid covers E)
  auto slice_of = [&](int w) -> Slice& { return *reinterpret_cast<Slice*>(raw + (size_t)w * kObsWave); };

  // ---- once per launch ----
  {
    const CbfParams<T> P = load_const(&a0->p.P);
    const T* obstacles = a0->obstacles;
  for (int r = tid; r < R * 64; r += NT)                       // (a loop: NW = 1 or 2 -- the host emulation of the tests -- has fewer threads than slots)
    stab[r >> 6][r & 63] = roll_slot_of<T>(P, a0->pair_ij, r, kRec, kSwz && P.num_drones > 8, kSobOff, kDsOff, kBounds);
  if (tid < kCbfMaxObs) {
    const bool on = tid < P.n_obs;
    for (int k = 0; k < 8; ++k) sobrec[tid][k] = T(0);
    if (on) {
      sobrec[tid][0] = obstacles[4 * tid];
      sobrec[tid][1] = obstacles[4 * tid + 1];
      sobrec[tid][6] = obstacles[4 * tid + 2];
    }
    sDs[1 + tid] = cbf_neg_ds4(on ? P.safety_radius + obstacles[4 * tid + 3] : T(1));
  }
  if (tid == 0) sDs[0] = cbf_neg_ds4(P.Ds_pair);
  }
  if (tid < GB) scost[tid] = (tid < nenv && a0->cost_io) ? a0->cost_io[env0 + tid] : 0;
  State<T> s;
  s.p = s.v = s.w = {T(0), T(0), T(0)};
  s.q[0] = s.q[1] = s.q[2] = T(0);
  s.q[3] = T(1);
  if (valid) load_state<T, T>(a0->state, a0->ld, i, s);
  // The obstacle records and -Ds^4 written above are read by the FIRST stage A below (the per-drone bounds), by every wave: a workgroup
  // barrier per launch.  (Found by the host SIMT emulation of tests/emul/simt: without it the first control step of a launch raced --
  // round 3 read these tables in stage B only, behind the step's own barrier.)
  if (kBounds) __syncthreads();
  T un1 = T(0), un2 = T(0), un3 = T(0);                        // u_hat[1..3] of this step (stage A -> stage C, in registers across stage B; [0]: st[13])

  // stage A of drone i on the state in registers at time ta: u_hat, the record, the stash of the state
  auto stage_a = [&](const RollArgs<T> MDS_CONST_AS* a, const Consts<T>& c, const LemniscateParams<T>& Pl, const double ta, const int tq) {
    const int lq = tq & 63, wq = tq >> 6;
    const Desired<T> des = lemniscate_local(Pl, ta);
    const V3<T> rpy = euler_from_quat(s.q);
    T un0;
    if (NOM == 0) {
      const M3<T> Rm = quat_to_rot(s.q);
      const V3<T> ang_v = mul(Rm, s.w);
      T u[4];
      GeoAux<T> A;
      geometric_control<T>(c, s.p - des.p, Rm, s.v, ang_v, des, u, &A);
      un0 = A.force - c.gravity;
      un1 = A.w_des.x; un2 = A.w_des.y; un3 = A.w_des.z;
    } else {
      T u[4];
      lqr_omega_control<T>(c, *static_cast<const LqrGain<T>*>(a->Kp), rpy, s.v, s.p, des.p, des.v, des.yaw, u);
      un0 = u[0] - c.gravity;                                                      // CBFTest.py:339
      un1 = u[1]; un2 = u[2]; un3 = u[3];
    }
    // obs_to_lin_model(obs, 9) - xdes with obs = pack_obs(state): world position, roll, pitch, world velocity; xdes = [0, 0, yaw, v_des, p_des]
    T* rc = slice_of(wq).rec[lq];
    const int sw = kSwz ? ((lq >> 3) & (Dp >> 4)) : 0;           // Dp = 16, agents 8..15: halves swapped
    *reinterpret_cast<V4*>(rc + 4 * sw) = V4{{s.p.x + Pl.cx, s.p.y + Pl.cy, rpy.y - T(0), -(rpy.x - T(0))}};
    *reinterpret_cast<V4*>(rc + 4 * (sw ^ 1)) = V4{{s.v.x - des.v.x, s.v.y - des.v.y, s.p.z + Pl.cz, s.v.z - des.v.z}};
    if constexpr (kBounds) {
      // the drone's single-variable rows -> its bounds.  Same operands as round 3's obstacle row slots (an obstacle is a record with
      // zero tracking errors: x - 0 is exact), same row polynomial (contraction off), four obstacles side by side.
      const CbfParams<T> Pb = load_const(&a->p.P);
      const T wx = s.p.x + Pl.cx, wy = s.p.y + Pl.cy, wz = s.p.z + Pl.cz, evx = s.v.x - des.v.x, evy = s.v.y - des.v.y, evz = s.v.z - des.v.z;
      T hi = Pb.umax[0], nlo = Pb.umax[0];
      bool badl = false;
      constexpr int kOb = MDS_ROLL_OBS_CHUNK;                      // obstacle rows evaluated side by side (4: one scratch reload per step -- 128 VGPRs; 2: none)
      for (int o0 = 0; o0 < Pb.n_obs; o0 += kOb) {                   // (uniform)
        Pair<T> exy[kOb], dpr[kOb], dvxy[kOb], ezvz[kOb];
        T nds4[kOb], hr[kOb], lg[kOb];
#pragma unroll
        for (int j = 0; j < kOb; ++j) {
          const int o = o0 + j < kCbfMaxObs ? o0 + j : kCbfMaxObs - 1;
          exy[j] = Pair<T>{wx - sobrec[o][0], wy - sobrec[o][1]};
          dpr[j] = Pair<T>{rpy.y, -rpy.x};
          dvxy[j] = Pair<T>{evx, evy};
          ezvz[j] = Pair<T>{wz - sobrec[o][6], evz};
          nds4[j] = sDs[1 + o];
        }
        cbf_row_o2_pairs<T, kOb>(Pb, exy, dpr, dvxy, ezvz, nds4, hr, lg);
#pragma unroll
        for (int j = 0; j < kOb; ++j) {
          const bool on = o0 + j < Pb.n_obs;                       // (uniform)
          const T aa = m_abs(lg[j]);                               // the row: -lg u <= hr
          // beyond the reach of the box (round 3's test on the normalised row, b < -|a| umax): also the row 0 u <= hr < 0
          badl = badl | (on & (hr[j] < -(aa * Pb.umax[0]) * (T(1) + T(sizeof(T) == 4 ? 1e-5 : 1e-10))));
          const T bnd = hr[j] * m_rcp(aa);
          hi = (on && lg[j] < T(0)) ? m_min(hi, bnd) : hi;         //  |lg| u <= hr
          nlo = (on && lg[j] > T(0)) ? m_min(nlo, bnd) : nlo;      // -|lg| u <= hr
        }
      }
      sbnd[0][tq] = badl ? -GiEps<T>::inf : hi;
      sbnd[1][tq] = badl ? -GiEps<T>::inf : nlo;
    }
    su_all[tq] = un0;
    st[13][tq] = un0;                                              // (su_all holds the QP's answer after stage B)
    st[0][tq] = s.p.x; st[1][tq] = s.p.y; st[2][tq] = s.p.z;
    st[3][tq] = s.q[0]; st[4][tq] = s.q[1]; st[5][tq] = s.q[2]; st[6][tq] = s.q[3];
    st[7][tq] = s.v.x; st[8][tq] = s.v.y; st[9][tq] = s.v.z;
    st[10][tq] = s.w.x; st[11][tq] = s.w.y; st[12][tq] = s.w.z;
  };
  auto load_params = [&](const RollArgs<T> MDS_CONST_AS* a, unsigned iu) {
    const T* lem = a->lem;
    const size_t ld = a->ld;
    LemniscateParams<T> Pl;
    struct alignas(2 * sizeof(T)) V2 {
      T v[2];
    };
    const V4 a4 = *lane_ptr(reinterpret_cast<const V4*>(lem), iu);
    const V2 c2 = *lane_ptr(reinterpret_cast<const V2*>(lem + 4 * ld), iu);
    Pl.a = a4.v[0]; Pl.omega = a4.v[1]; Pl.yaw_rate = a4.v[2]; Pl.phase_shift = a4.v[3];
    Pl.cx = c2.v[0]; Pl.cy = c2.v[1]; Pl.cz = *lane_ptr(lem + 6 * ld, iu);
    return Pl;
  };
  if (valid && n_steps > 0) {
    const Consts<T> c1 = load_const(&a0->p.c);
    stage_a(a0, c1, load_params(a0, (unsigned)i), t, tid);
  }
  const int nbs = (cbf_num_pairs(D) + (kBounds ? 0 : D * a0->p.P.n_obs) + 63) >> 6;   // row slots that hold barrier rows

  for (int k = 0; k < n_steps; ++k) {
    if (wave == 0) {
      // hand-out order of this step's QPs: envs that iterated last step first (their solves are the long ones), stable within a class
      if (GB <= 64) {
        const bool has = lane < nenv;
        const bool heavy = has && scost[has ? lane : 0] >= kCbfMediumIters;
        const unsigned long long mh = __ballot(heavy), ml = __ballot(has && !heavy);
        auto below = [](unsigned long long m) {                    // set bits of m in the lanes below this one (v_mbcnt: no per-lane mask to hold)
          return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        };
        const int pos = heavy ? below(mh) : __popcll(mh) + below(ml);
        if (has) sorder[pos] = lane;
      } else {
        for (int e = lane; e < nenv; e += 64) sorder[e] = e;
      }
      if (lane == 0) sticket = 0;
    }
    if (kStamps && stamps != nullptr && k == 0) tk0 = __builtin_amdgcn_s_memtime();
#if !(defined(MDS_TUNE_ROLL_NO_BAR) && (MDS_TUNE_ROLL_NO_BAR & 1))   // tuning aid: WRONG results (a race), only to bound what the barrier costs
    __syncthreads();
#endif
    stamp(0);

    // ---- stage B: the workgroup's envs, one per wave at a time ----
    {
      Scratch& S = slice_of(wave).sc;
      if (kStamps && stamps != nullptr) tk1 = __builtin_amdgcn_s_memtime();
#if !defined(MDS_TUNE_ROLL_STATIC)
      // tickets are drawn one env ahead: the LDS atomic of the next env's ticket is in flight while this env's rows are built
      int tk_next = 0;
      if (lane == 0) tk_next = atomicAdd(&sticket, 1);
      while (true) {
        const int tk = __builtin_amdgcn_readfirstlane(tk_next);
        if (tk >= nenv) break;                                     // wave-uniform
        if (lane == 0) tk_next = atomicAdd(&sticket, 1);
#else
      // A/B: the hand-out order dealt round-robin over the waves instead of the ticket counter (no LDS atomic per env).  Measured
      // slower on every scene (MI355X, C4, us per control step, tickets -> static: far 29.5 -> 31.5, under 29.8 -> 31.9, level
      // 41.2 -> 51.1): waves of a workgroup do not run at the same pace even when their envs cost the same, and the ticket absorbs it.
      for (int tk = wave; tk < nenv; tk += NW) {
#endif
        const int el = sorder[tk];                                 // env of the workgroup (uniform)
        stamp_b(6);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MDS_TUNE_NO_SETPRIO)
        __builtin_amdgcn_s_setprio(0);
#endif
        // the CBF parameters of the row build: scalar loads per env (held across the solver they were spilled to VGPR lanes)
        const RollArgs<T> MDS_CONST_AS* ab = fresh_args<T>();
        const CbfParams<T> P = load_const(&ab->p.P);
        const int max_iter = ab->max_iter;                         // (requested with them: the solver starts right after the scan)
        const T tol2 = ab->tol2;
        const int d0 = el << log2P;                                // the env's first (padded) lane
        const unsigned ebase = (unsigned)(d0 >> 6) * kObsWave + (unsigned)(d0 & 63) * kRec;   // the env's first record in the LDS block
        int tl = lane;
        MDS_KEEP_V(tl);                               // re-read per env: 4 LDS reads instead of 16 registers held across the stages
        T ca[R][NV], cb[R][NV], b[R];
        int ia[R], ib[R];
        bool vld[R], act[R];
        bool bad = false;
        // kBounds: a drone whose single-variable rows leave no thrust value (stage A wrote -inf / -inf, or lo > hi) makes the env
        // infeasible before any row is built -- the same decision as round 3's reach test while the rows were built, plus rows of one
        // drone that contradict each other (round 3: found by the solver, "no step possible")
        bool skip = false;
        if constexpr (kBounds) {
          const int lc = tl < D ? tl : D - 1;
          const T h_l = sbnd[0][d0 + lc], n_l = sbnd[1][d0 + lc];
          skip = __any(h_l + n_l < -ab->tol);
        }
        // Rows r = lane + 64 k.  Slots below NBS hold barrier rows -- all of them in slots below NBS - 1, beside the first box rows in
        // slot NBS - 1: ONE straight-line body per slot, every lane runs the barrier arithmetic (lanes of other kinds on a harmless
        // record pair, their result replaced by a select), all slots' LDS reads issued together (one round trip per env).
        // The normalisation without its branch: n2 == 0 leaves the row unscaled (inv = 1) and turns the reach test into b < 0, exactly
        // the two cases of cbf_o2_slot.
        auto build_rows = [&](auto nbc) {
        constexpr int NBS = decltype(nbc)::value;                  // compile time: the slots' bodies share one basic block
        constexpr int kHalf = 4 * (int)sizeof(T);
        RollSlot sl[R];
#pragma unroll
        for (int r = 0; r < R; ++r) sl[r] = stab[r][tl];
        T bndv[R];                                                 // kBounds: the right-hand side of a bound row (kind 3: hi, kind 4: -lo of its drone)
#pragma unroll
        for (int r = 0; r < R; ++r) bndv[r] = P.umax[0];
        if constexpr (kBounds) {
#pragma unroll
          for (int r = NBS - 1; r < R; ++r)
            bndv[r] = (&sbnd[0][0])[(((sl[r].kind & 255) == 4) ? NT : 0) + d0 + ((sl[r].kind >> 8) & 255)];
        }
        using P2 = Pair<T>;
        P2 oa[NBS][4], ob[NBS][4];                                 // operand pairs of agent i / agent j or obstacle: (px py) (e_pitch -e_roll) (e_vx e_vy) (pz e_vz)
        T nds4[NBS];
        int pm[NBS];
#pragma unroll
        for (int r = 0; r < NBS; ++r) {
          pm[r] = (sl[r].kind << 7) >> 31;                         // all ones on a pair row
          const unsigned pa = ebase + (unsigned)sl[r].a0, pb = ((unsigned)pm[r] & ebase) | (unsigned)sl[r].b0;
          const V4 a0 = *reinterpret_cast<const V4*>(raw + pa), a1 = *reinterpret_cast<const V4*>(raw + (kSwz ? (pa ^ 16u) : pa + kHalf));
          const V4 b0 = *reinterpret_cast<const V4*>(raw + pb), b1 = *reinterpret_cast<const V4*>(raw + (kSwz ? (pb ^ 16u) : pb + kHalf));
          oa[r][0] = P2{a0.v[0], a0.v[1]}; oa[r][1] = P2{a0.v[2], a0.v[3]}; oa[r][2] = P2{a1.v[0], a1.v[1]}; oa[r][3] = P2{a1.v[2], a1.v[3]};
          ob[r][0] = P2{b0.v[0], b0.v[1]}; ob[r][1] = P2{b0.v[2], b0.v[3]}; ob[r][2] = P2{b1.v[0], b1.v[1]}; ob[r][3] = P2{b1.v[2], b1.v[3]};
          nds4[r] = *reinterpret_cast<const T*>(raw + sl[r].ds);
        }
#pragma unroll
        for (int r = 0; r < NBS; ++r)                              // every slot's operands in ONE LDS round trip
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            MDS_KEEP_V(oa[r][c].x);
            MDS_KEEP_V(oa[r][c].y);
            MDS_KEEP_V(ob[r][c].x);
            MDS_KEEP_V(ob[r][c].y);
          }
        // the barrier arithmetic of all NBS slots, statement by statement across the slots: NBS independent chains side by side
        Pair<T> exy[NBS], dpr[NBS], dvxy[NBS], ezvz[NBS];
        T hr[NBS], lg[NBS];
#pragma unroll
        for (int r = 0; r < NBS; ++r) {
          exy[r] = oa[r][0] - ob[r][0];
          dpr[r] = oa[r][1] - ob[r][1];
          dvxy[r] = oa[r][2] - ob[r][2];
          ezvz[r] = oa[r][3] - ob[r][3];
        }
#if (MDS_TUNE_ROLL_SKIP & 1)   // tuning aid: the row polynomial replaced by a sum of its operands (rows always satisfied)
        for (int r = 0; r < NBS; ++r) {
          hr[r] = T(100) + nds4[r] + exy[r].x + exy[r].y + dpr[r].x + dpr[r].y + dvxy[r].x + dvxy[r].y + ezvz[r].x + ezvz[r].y;
          lg[r] = T(1);
        }
#else
        cbf_row_o2_pairs<T, NBS>(P, exy, dpr, dvxy, ezvz, nds4, hr, lg);
#endif
        T cak[NBS], cbk[NBS], bk[NBS], n2[NBS], inv[NBS], reach[NBS];
        bool pos[NBS], unreachable[NBS];
#define MDS_SLOTS(stmt)         \
  _Pragma("unroll") for (int r = 0; r < NBS; ++r) { stmt; }
        MDS_SLOTS(cak[r] = -lg[r])
        if constexpr (sizeof(T) == 4) {
          MDS_SLOTS(cbk[r] = __builtin_bit_cast(float, __builtin_bit_cast(int, lg[r]) & pm[r]))     // pair rows: +lg at agent j
        } else {
          MDS_SLOTS(cbk[r] = pm[r] ? lg[r] : T(0))
        }
#if (MDS_TUNE_ROLL_SKIP & 2)   // tuning aid: no normalisation
        MDS_SLOTS(pos[r] = true; inv[r] = T(1); n2[r] = T(1))
#else
        MDS_SLOTS(n2[r] = m_fma(cak[r], cak[r], cbk[r] * cbk[r]))
        MDS_SLOTS(pos[r] = n2[r] > T(0))
        MDS_SLOTS(inv[r] = pos[r] ? m_rsqrt(n2[r]) : T(1))
#endif
        MDS_SLOTS(cak[r] *= inv[r])
        MDS_SLOTS(cbk[r] *= inv[r])
        MDS_SLOTS(bk[r] = hr[r] * inv[r])
        MDS_SLOTS(reach[r] = (m_abs(cak[r]) + m_abs(cbk[r])) * P.umax[0])
        MDS_SLOTS(unreachable[r] = bk[r] < -reach[r] * (T(1) + T(sizeof(T) == 4 ? 1e-5 : 1e-10)))
#undef MDS_SLOTS
#pragma unroll
        for (int r = 0; r < R; ++r) {
          act[r] = false;
          const int kind = sl[r].kind & 255;
          ia[r] = (sl[r].kind >> 8) & 255;
          ib[r] = (sl[r].kind >> 16) & 255;
          if (r < NBS - 1) {                                       // barrier rows only
            bad = bad | unreachable[r];                            // (bitwise: no short-circuit branch)
            ca[r][0] = cak[r];
            cb[r][0] = cbk[r];
            b[r] = bk[r];
            vld[r] = pos[r];
          } else if (r < NBS) {
            const bool bar = kind == 1 || kind == 2;
            bad = bad | (bar & unreachable[r]);
            // +-u_var <= umax (cbf/cbf.py:400-412): unit norm as it stands
            const T box_c = kind == 3 ? T(1) : (kind == 4 ? T(-1) : T(0)), box_b = kind >= 3 ? bndv[r] : T(0);
            ca[r][0] = bar ? cak[r] : box_c;
            cb[r][0] = bar ? cbk[r] : T(0);
            b[r] = bar ? bk[r] : box_b;
            vld[r] = (bar & pos[r]) | (kind >= 3);
          } else {
            ca[r][0] = kind == 3 ? T(1) : (kind == 4 ? T(-1) : T(0));
            cb[r][0] = T(0);
            b[r] = kind >= 3 ? bndv[r] : T(0);
            vld[r] = kind >= 3;
          }
        }
        };
#if (MDS_TUNE_ROLL_SKIP & 8)   // tuning aid: no rows at all
        for (int r = 0; r < R; ++r) { ca[r][0] = cb[r][0] = b[r] = T(0); ia[r] = ib[r] = 0; vld[r] = act[r] = false; }
        if (false)
#endif
        if (!skip) {
          if constexpr (kBounds) {
            if (__builtin_amdgcn_readfirstlane(nbs) <= 1) build_rows(wv::Ic<1>{});     // (wave-uniform; a scalar branch)
            else build_rows(wv::Ic<2>{});
          } else {
            switch (__builtin_amdgcn_readfirstlane(nbs)) {
              case 1: build_rows(wv::Ic<1>{}); break;
              case 2: build_rows(wv::Ic<2>{}); break;
              case 3: build_rows(wv::Ic<3>{}); break;
              default: build_rows(wv::Ic<(R < 4 ? R : 4)>{}); break;
            }
          }
        }
        stamp_b(7);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MDS_TUNE_NO_SETPRIO)
        // the scan, the bookkeeping and the next ticket are short serial chains (VALU -> scalar -> branch, LDS round trips): at equal
        // priority they queue behind the other waves' dense row arithmetic on every instruction; they go first, the row build yields
        __builtin_amdgcn_s_setprio(1);
#endif
        bool converged = false;
        int it = 0, q = 0;
#if (MDS_TUNE_ROLL_SKIP & 4)   // tuning aid: no scan, no solve (the rows are still built: their sum decides "converged")
        {
          T acc = T(0);
          for (int r = 0; r < R; ++r) acc += ca[r][0] + cb[r][0] + b[r] + T(ia[r] + ib[r]) + (vld[r] ? T(1) : T(0));
          converged = !__any(bad) && __any(acc > T(-1e30));
        }
        if (false)
#endif
        if (!skip)
        gi_solve<T, R, NMAX, NV, false, kQS>(lane, D, max_iter, tol2, __any(bad), ca, cb, b, ia, ib, vld, act, &su_all[d0], S.sd, S.slam, S.sdi,
                                             S.sQ, S.sR, S.sact, nullptr, converged, it, q);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MDS_TUNE_NO_SETPRIO)
        __builtin_amdgcn_s_setprio(1);                             // (gi_solve raised it to 3 if the env iterated)
#endif
        stamp_b(9);
        if (lane == 0) {                                           // (status_log / status / cost_io: written from these after the barrier)
          sconv[el] = converged ? 1 : 0;
          scost[el] = it;
        }
        MDS_WAVE_SYNC();
        stamp_b(8);
      }
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MDS_TUNE_NO_SETPRIO)
    __builtin_amdgcn_s_setprio(MDS_TUNE_ROLL_PRIO_C);
#endif
    stamp(1);
#if !(defined(MDS_TUNE_ROLL_NO_BAR) && (MDS_TUNE_ROLL_NO_BAR & 2))   // tuning aid: WRONG results (a race), only to bound what the barrier costs
    __syncthreads();
#endif
    stamp(2);

    // ---- stage C of this step, then stage A of the next: one drone per lane, state in registers ----
    const RollArgs<T> MDS_CONST_AS* ac = fresh_args<T>();
    T* const obs_log = ac->obs_log;
    const double ctrl_dt = ac->ctrl_dt;
    {                                                              // the step's statuses, one coalesced store per workgroup
      int ts = tid;
      MDS_KEEP_V(ts);                                 // (LDS addresses formed here, not held across the loop)
      if (ts < nenv) {
        if (ac->status_log) ac->status_log[(size_t)k * ac->E + env0 + ts] = sconv[ts] ? 0 : 1;
        if (k == n_steps - 1) {
          ac->status[env0 + ts] = sconv[ts] ? 0 : 1;
          if (ac->cost_io) ac->cost_io[env0 + ts] = scost[ts];
        }
      }
    }
    const bool want = obs_log != nullptr || k == n_steps - 1;
    const bool more = k + 1 < n_steps;
    {                                                              // (uniform: kept in SGPRs -- as a VGPR pair the step loop spilled it)
      const unsigned long long tb = __builtin_bit_cast(unsigned long long, t + ctrl_dt);
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)tb), hi = __builtin_amdgcn_readfirstlane((unsigned)(tb >> 32));
      t = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    }
    unsigned eg0 = blockIdx.x * GB;
    MDS_KEEP_S(eg0);                                  // the drone index is formed here, not held -- spilled -- across the loop
    int tq = tid;
    MDS_KEEP_V(tq);
    unsigned wg0 = blockIdx.x * NT;
    MDS_KEEP_S(wg0);
    const unsigned iu = PAD ? (eg0 + ((unsigned)tq >> log2P)) * (unsigned)D + ((unsigned)tq & (unsigned)(Dp - 1)) : wg0 + (unsigned)tq;
    const Consts<T> c = load_const(&ac->p.c);
    const CbfParams<T> P = load_const(&ac->p.P);
    constexpr int kRowBytes = kObsDim * (int)sizeof(T);
    const int lq = tq & 63;
    unsigned char* lds_wave = raw + (tq >> 6) * kObsWave;          // the wave's staging slice (its records / solver scratch of stage B: done with)
    // the next step's trajectory parameters: requested here, consumed by stage A below (their L2 latency under this stage's arithmetic)
    // (unconditional, from a clamped index: nothing to merge, so nothing waits for the loads before stage A)
    const unsigned il = min(iu, (unsigned)ac->n - 1u);
    const LemniscateParams<T> Pl_next = load_params(ac, il);
    LowLevelState<T> L;                                            // the low level's PID memory of this drone: requested with them
    {
      const size_t ld = ac->ld;
      const T* const ll = ac->ll;
      L.last_omega = {*lane_ptr(ll + 0 * ld, il), *lane_ptr(ll + 1 * ld, il), *lane_ptr(ll + 2 * ld, il)};
      L.integral = {*lane_ptr(ll + 3 * ld, il), *lane_ptr(ll + 4 * ld, il), *lane_ptr(ll + 5 * ld, il)};
    }
    {
      // Every lane runs the stage (a lane past the last drone on a clamped index, its stores skipped): no divergent region for the
      // compiler to sink the loads above into.
      const int conv = sconv[tq >> log2P];
      const T safe = su_all[tq];
      T u[4];
      u[0] = (conv ? safe : st[13][tq]) + c.gravity;                                                // CBFTest.py:346
      u[1] = conv ? m_clamp(un1, -P.umax[1], P.umax[1]) : un1;
      u[2] = conv ? m_clamp(un2, -P.umax[2], P.umax[2]) : un2;
      u[3] = conv ? m_clamp(un3, -P.umax[3], P.umax[3]) : un3;
      s.p = {st[0][tq], st[1][tq], st[2][tq]};
      s.q[0] = st[3][tq]; s.q[1] = st[4][tq]; s.q[2] = st[5][tq]; s.q[3] = st[6][tq];
      s.v = {st[7][tq], st[8][tq], st[9][tq]};
      s.w = {st[10][tq], st[11][tq], st[12][tq]};
      const size_t ld = ac->ld;
      T* const ll = ac->ll;
      Resid<T> rs;
      if (COMP) load_resid<T, T>(ac->state_lo, ld, il, rs);
      // per-lane planes through a uniform base + a 32-bit byte offset (lane_ptr): one VGPR of address for all of them
      T act4[4], prev[4] = {T(0), T(0), T(0), T(0)}, clipped[4];
      thrust_omega_control(c, (T)ctrl_dt, u, s.w, L, act4);
      if (valid) {
        *lane_ptr(ll + 0 * ld, iu) = L.last_omega.x; *lane_ptr(ll + 1 * ld, iu) = L.last_omega.y; *lane_ptr(ll + 2 * ld, iu) = L.last_omega.z;
        *lane_ptr(ll + 3 * ld, iu) = L.integral.x; *lane_ptr(ll + 4 * ld, iu) = L.integral.y; *lane_ptr(ll + 5 * ld, iu) = L.integral.z;
      }
      aviary_step_any<T, false, false, COMP>(c, s, rs, act4, prev, clipped);
      if (COMP && valid) store_resid<T, T>(ac->state_lo, ld, iu, rs);
      if (valid && !more && ac->last_rpm)
        for (int j = 0; j < 4; ++j) ac->last_rpm[j * ld + iu] = clipped[j];
      if (want) {
        alignas(16) T o[kObsDim];
        pack_obs(s, V3<T>{Pl_next.cx, Pl_next.cy, Pl_next.cz}, clipped, o);                  // (the trajectory centre: this drone's, whatever the step)
        // the row goes to the wave's staging slice at once (20 values held across the rest of the stage were spilled to scratch)
        typedef unsigned int v4u __attribute__((ext_vector_type(4)));
        // (row of the wave's staging slice: its envs' drones back to back -- the padded lanes of an env hold no row)
        const int crow = PAD ? (lq >> log2P) * D + (lq & (Dp - 1)) : lq;
        if (!PAD || (lq & (Dp - 1)) < D) {
#pragma unroll
          for (int j = 0; j < kRowBytes / 16; ++j) reinterpret_cast<v4u*>(lds_wave + crow * kRowBytes)[j] = reinterpret_cast<const v4u*>(o)[j];
        }
      }
      if (valid && !more) store_state<T, T>(ac->state, ld, iu, s);
    }
    stamp(3);
    if (want) {
      // write_obs_rows with this stage's fresh indices: the wave's staging slice -> 16-byte coalesced non-temporal stores
      const int n = ac->n;
      T* dst = obs_log != nullptr ? obs_log + (size_t)slot * n * kObsDim : ac->obs_last;      // (log AND obs_last: the host copies the last slot)
      MDS_WAVE_SYNC();
      const int ew = __builtin_amdgcn_readfirstlane((int)eg0 + ((tq >> 6) << (6 - log2P)));  // the wave's first env (uniform)
      const int wave_base = PAD ? ew * D : __builtin_amdgcn_readfirstlane((int)iu - lq);   // its first observation row: scalar base + 32-bit lane offsets
      const int rows = PAD ? (min(ew + (64 >> log2P), ac->E) - ew) * D : min(64, n - wave_base);   // (<= 0: no env of the shard in this wave)
      if (rows > 0) {
        const int bytes = rows * kRowBytes;
        unsigned char* gdst = reinterpret_cast<unsigned char*>(dst) + (size_t)wave_base * kRowBytes;
#pragma unroll
        for (int it = 0; it < kRowBytes / 16; ++it) {
          const unsigned off = (unsigned)(it * 64 + lq) * 16u;
          if ((int)off + 16 <= bytes) {
            typedef unsigned int v4u __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(*reinterpret_cast<const v4u*>(lds_wave + off), reinterpret_cast<v4u*>(gdst + off));
          }
        }
      }
      MDS_WAVE_SYNC();                                             // the slice is stage A's record / scratch space next
    }
    stamp(4);
    slot = slot + 1 == ac->n_slots ? 0 : slot + 1;
    if (valid && more) {                                           // (the wave's own staging slice is drained: write_obs_rows ends with a wave sync)
      int ta2 = tid;
      MDS_KEEP_V(ta2);
      const RollArgs<T> MDS_CONST_AS* aa = fresh_args<T>();
      stage_a(aa, c, Pl_next, t, ta2);
    }
    stamp(5);
  }
  if (kStamps && stamps != nullptr) {
    __syncthreads();
    if (threadIdx.x < NW * 10) stamps[(size_t)blockIdx.x * NW * 10 + threadIdx.x] = sst[threadIdx.x / 10][threadIdx.x % 10];
  }
}

// ------------------------------------------------------------------------------------
// The order-3 loop of simulations/CBFTestOrd3.py:306-352 in launches of n_steps control steps -- the persistent form for the
// reference's own order-3 shape (7 drones, one env: CBFTestOrd3.py:31, :452).  One wavefront per env, its drones on lanes 0 .. D - 1
// with the state, the RPM echo (the thrust state is calc_z_thrust of it) and the low level's memory in registers for the whole launch.
// Per step: every drone's current observation row, its nominal LQR-yank-omega input (yank - M G, w: :341) and xdes = [0, 0, yaw, M G,
// v_des, p_des] (:345-347) go to LDS blocks; cbf_filter_env -- the body of the step-by-step filter kernel itself -- solves the env's
// 3 D-variable QP on them; the drone lanes run the YankOmega low level on u_safe (the yank as it is: :350) and the physics step, and
// write the observation row (slot (slot0 + k) % n_slots of the log ring, or only the last).  obs_io [n, 20]: in, the current observation
// (its RPM echo starts the thrust state); out, the last step's.  t advances in double like the host loop.  Not tuned -- D of 64 lanes work
// in the per-drone stages: the reference's order-3 scenes are a handful of drones -- but no launch, no u_hat / xdes / u_safe round
// trip through HBM per control step.
// ------------------------------------------------------------------------------------
// The filter body is CALLED here, not inlined (MDS_TUNE_O3_CALL 1).  Inlined into this kernel's step loop, the float64 instantiation
// (390 registers incl. 134 AGPRs) came out of the compiler wrong as soon as gi_solve was touched: with the q == n guard -- or with an
// equivalent guard placed elsewhere -- step 0 was exact and every env's status wrong from step 1 on, while the same sources pass on the
// host emulation under ASan / UBSan / TSan / MSan, the fp32 instantiation passes, and the step-by-step kernel (the same body, inlined
// into a kernel without a loop around it) passes.  As a call the solver is compiled once per (T, R, NMAX) on its own registers; this
// kernel is not a tuned path.
#ifndef MDS_TUNE_O3_CALL
#define MDS_TUNE_O3_CALL 1
#endif
template <typename T, int R, int NMAX>
__device__ __noinline__ void cbf_filter_env_o3_call(const CbfParams<T>* P, int lane, T kf, const int* pair_ij, const T* obstacles, const T* obs, const T* xdes,
                                                    const T* unom, T* usafe, int* status_env, int max_iter, T tol2, int* cost_env) {
  cbf_filter_env<T, T, R, NMAX, 3>(*P, lane, kf, pair_ij, obstacles, obs, xdes, unom, usafe, status_env, max_iter, tol2, cost_env);
}
template <typename T, int R, int NMAX>
__global__ __launch_bounds__(64) void k_cbf_rollout_o3(const Consts<T> c, const CbfParams<T> P, const LqrYoGain<T> K, const int E, const size_t ld, double t,
                                                       const double ctrl_dt, const int n_steps, T* __restrict__ state, const T* __restrict__ lem,
                                                       T* __restrict__ last_rpm, T* __restrict__ ll, const int* __restrict__ pair_ij,
                                                       const T* __restrict__ obstacles, T* __restrict__ obs_io, T* __restrict__ obs_log, int slot,
                                                       const int n_slots, int* __restrict__ status, int* __restrict__ status_log,
                                                       int* __restrict__ cost, const int max_iter, const T tol2, const T hover_sub) {
  constexpr int DMAX = NMAX / 3;
  __shared__ __align__(16) T sobs[DMAX * kObsDim], sxdes[DMAX * 10], sunom[DMAX * 4], susafe[DMAX * 4];
  __shared__ int sres[2];                                          // status, iterations of this step's solve
  const int lane = threadIdx.x, env = blockIdx.x, D = P.num_drones;
  const bool drone = lane < D;
  const size_t i = (size_t)env * D + (drone ? lane : 0);           // (the other lanes: a clamped index, nothing stored)
  const size_t n = (size_t)E * D;
  GeoIn<T> in;
  load_geo_in<T, T>(state, lem, ld, i, in);
  LowLevelState<T> L;
  L.last_omega = {ll[0 * ld + i], ll[1 * ld + i], ll[2 * ld + i]};
  L.integral = {ll[3 * ld + i], ll[4 * ld + i], ll[5 * ld + i]};
  T clipped[4];
  load4<T, T>(obs_io + i * kObsDim + 16, clipped);
  const V3<T> centre = {in.P.cx, in.P.cy, in.P.cz};
  for (int k = 0; k < n_steps; ++k) {
    if (drone) {
      const Desired<T> des = lemniscate_local(in.P, t);
      T u[4], o[kObsDim];
      lqr_yank_omega_control<T>(c, K, euler_from_quat(in.s.q), clipped, in.s.v, in.s.p, des.p, des.v, des.yaw, u);
      sunom[4 * lane] = u[0] - hover_sub; sunom[4 * lane + 1] = u[1]; sunom[4 * lane + 2] = u[2]; sunom[4 * lane + 3] = u[3];
      T* xd = &sxdes[10 * lane];
      xd[0] = T(0); xd[1] = T(0); xd[2] = des.yaw; xd[3] = c.gravity;
      xd[4] = des.v.x; xd[5] = des.v.y; xd[6] = des.v.z;
      xd[7] = des.p.x + in.P.cx; xd[8] = des.p.y + in.P.cy; xd[9] = des.p.z + in.P.cz;
      pack_obs(in.s, centre, clipped, o);                          // the observation the previous step returned
#pragma unroll
      for (int j = 0; j < kObsDim; ++j) sobs[kObsDim * lane + j] = o[j];
    }
    MDS_WAVE_SYNC();
#if MDS_TUNE_O3_CALL
    cbf_filter_env_o3_call<T, R, NMAX>(&P, lane, c.kf, pair_ij, obstacles, sobs, sxdes, sunom, susafe, &sres[0], max_iter, tol2, &sres[1]);
#else
    cbf_filter_env<T, T, R, NMAX, 3>(P, lane, c.kf, pair_ij, obstacles, sobs, sxdes, sunom, susafe, &sres[0], max_iter, tol2, &sres[1]);
#endif
    MDS_WAVE_SYNC();
    if (lane == 0) {
      if (status_log) status_log[(size_t)k * E + env] = sres[0];
      if (k == n_steps - 1) {
        status[env] = sres[0];
        if (cost) cost[env] = sres[1];
      }
    }
    if (drone) {
      const T u[4] = {susafe[4 * lane], susafe[4 * lane + 1], susafe[4 * lane + 2], susafe[4 * lane + 3]};
      T act[4], prev[4] = {T(0), T(0), T(0), T(0)}, cl2[4];
      Resid<T> rs;
      resid_zero(rs);
      yank_omega_control(c, (T)ctrl_dt, u, clipped, in.s.w, L, act);
      aviary_step_any<T, false, false, false>(c, in.s, rs, act, prev, cl2);
#pragma unroll
      for (int j = 0; j < 4; ++j) clipped[j] = cl2[j];
      const bool last = k == n_steps - 1;
      if (obs_log != nullptr || last) {
        T o[kObsDim];
        pack_obs(in.s, centre, clipped, o);
        if (obs_log != nullptr) {
          T* dst = obs_log + ((size_t)slot * n + i) * kObsDim;
#pragma unroll
          for (int j = 0; j < kObsDim; j += 4) store4<T, T>(dst + j, o + j);
        }
        if (last) {
          T* dst = obs_io + i * kObsDim;
#pragma unroll
          for (int j = 0; j < kObsDim; j += 4) store4<T, T>(dst + j, o + j);
        }
      }
    }
    MDS_WAVE_SYNC();                                               // the LDS blocks are rewritten by the next step
    t += ctrl_dt;
    slot = slot + 1 == n_slots ? 0 : slot + 1;
  }
  if (drone) {
    store_state<T, T>(state, ld, i, in.s);
    ll[0 * ld + i] = L.last_omega.x; ll[1 * ld + i] = L.last_omega.y; ll[2 * ld + i] = L.last_omega.z;
    ll[3 * ld + i] = L.integral.x; ll[4 * ld + i] = L.integral.y; ll[5 * ld + i] = L.integral.z;
    if (last_rpm != nullptr && n_steps > 0)
      for (int j = 0; j < 4; ++j) last_rpm[j * ld + i] = clipped[j];
  }
}

// ------------------------------------------------------------------------------------
// Order-2 filter, FOUR envs per wavefront: each 16-lane DPP row of the wave owns one env (D <= 16 thrust variables, RL rows of
// G u <= h per lane).  Same Goldfarb-Idnani active-set iteration as gi_solve, with every "wave-uniform" quantity of that version
// (violation maximum, selected row, step lengths, active-set size q, iteration count) uniform per ROW instead: reductions are the
// four DPP steps of a row all-reduce (no v_readlane), a lane's value is handed to its row with ds_bpermute (__shfl, width 16),
// first-set-lane comes from the row's 16 bits of a ballot.  The four envs run in lock step through a two-phase loop (scan for the
// most violated row | one add-or-drop step); a row that is done idles under a predicate.  Control flow stays wave-uniform (DPP
// needs every lane), per-row state is a predicate on stores.
// Why: the one-env-per-wave kernel is latency-bound -- 11.5 cycles per instruction per wave, an env with nothing to do still costs
// 15 k cycles, the vectors of the iteration are 16 long in a 64-lane wave (profiles/r02_pmc_c4_*).  Four envs per wave quarter the
// waves, fill the lanes of every step of the iteration and amortise the latency of the global reads over four envs.
// Measured at C4 (profiles/r02_c4_fused.md): exact (every parity test passes with it: statuses equal to the oracle's at every step),
// but no faster -- per full-batch launch 31.4 us against 29.6 us when no env iterates, 71 us against 56 us on SURVEY 8d's scene.
// 14 rows per lane cost 168-187 VGPRs (3 waves per SIMD at best, Q and R left in LDS), a wave lives as long as the slowest of its
// four envs and pays scan + step in every pass.  Opt-in: MDS_CBF_Q4=1 at mds_cbf_configure time.
// ------------------------------------------------------------------------------------
namespace r16 {
template <typename T, typename Op> __device__ __forceinline__ T allreduce(T v, Op op) {   // over each 16-lane row; all 64 lanes active
  v = op(v, wv::mov<0xB1>(v));
  v = op(v, wv::mov<0x4E>(v));
  v = op(v, wv::mov<0x141>(v));
  v = op(v, wv::mov<0x140>(v));
  return v;
}
template <typename T> __device__ __forceinline__ T get(T v, int idx) { return __shfl(v, idx, 16); }   // lane idx of the caller's row
__device__ __forceinline__ int first(bool pred, int lane) {       // lowest lane of the caller's row with pred, 0 if none
  const unsigned long long m = __ballot(pred);
  const unsigned f = (unsigned)(m >> (lane & 48)) & 0xffffu;
  return f ? __ffs(f) - 1 : 0;
}
}  // namespace r16

template <typename T, typename S, int RL>
__global__ __launch_bounds__(64, sizeof(T) == 4 ? 3 : 1) void k_cbf_filter_q4(const CbfParams<T> P, const int E, const int* __restrict__ pair_ij,
                                                      const T* __restrict__ obstacles, const S* __restrict__ obs,
                                                      const S* __restrict__ xdes, const S* __restrict__ unom, S* __restrict__ usafe,
                                                      int* __restrict__ status, const int max_iter, const T tol2,
                                                      int* __restrict__ cost_out) {
  constexpr int NMAX = 16, kQS = 20;
  struct EnvLds {
    T pos[NMAX][3], de[NMAX][5];
    T u[NMAX], lam[NMAX], di[NMAX];
    T Q[NMAX][kQS], R[NMAX][kQS];
    int act[NMAX];
  };
  __shared__ __align__(16) EnvLds L4[4];
  __shared__ T sob[kCbfMaxObs * 4];
  const int lane = threadIdx.x, rl = lane & 15, r4 = lane >> 4;
  const int env = blockIdx.x * 4 + r4;
  const bool live = env < E;                                   // uniform per row
  const int D = P.num_drones, n = D;
  const int npairs = cbf_num_pairs(D), nobs_rows = D * P.n_obs, m = npairs + nobs_rows + 2 * n;
  EnvLds& L = L4[r4];

  // ---- inputs of the row's env: drone rl reads its own observation / xdes / nominal rows ----
  int rpair[RL];
#pragma unroll
  for (int k = 0; k < RL; ++k) rpair[k] = rl + 16 * k < npairs ? pair_ij[rl + 16 * k] : 0;
  sob[lane] = lane < 4 * P.n_obs ? obstacles[lane] : T(0);
  T un[4] = {T(0), T(0), T(0), T(0)};
  if (live && rl < D) {
    const size_t i = (size_t)env * D + rl;
    const S* o = obs + i * 20;
    const S* xd = xdes + i * 9;
    load4<S, T>(unom + i * 4, un);
    T o0[4], o1[4], o2[4], o3[4];
    load4<S, T>(o, o0);                                        // pos3, q0
    load4<S, T>(o + 4, o1);                                    // q1..3, roll
    load4<S, T>(o + 8, o2);                                    // pitch, yaw, vx, vy
    load4<S, T>(o + 12, o3);                                   // vz, ...
    L.pos[rl][0] = o0[0]; L.pos[rl][1] = o0[1]; L.pos[rl][2] = o0[2];
    L.de[rl][0] = o1[3] - (T)xd[0];                            // obs_to_lin_model(obs, 9) - xdes: roll, pitch, velocity
    L.de[rl][1] = o2[0] - (T)xd[1];
    L.de[rl][2] = o2[2] - (T)xd[3];
    L.de[rl][3] = o2[3] - (T)xd[4];
    L.de[rl][4] = o3[0] - (T)xd[5];
    L.u[rl] = un[0];
  }
  MDS_WAVE_SYNC();

  // ---- rows: RL per lane, row index r = rl + 16 k ----
  // a row is (ca, b, ia | ib << 8): a unit-norm pair row has cb = -ca on its second agent, every other row has one agent (cb = 0)
  T ca[RL], b[RL];
  int iab[RL];
  unsigned vmask = 0, amask = 0;                               // bit k: row k of this lane exists / is in the active set
  bool bad = false;
#pragma unroll
  for (int k = 0; k < RL; ++k) {
    bool vld;
    T cbk;
    int iak, ibk;
    cbf_o2_slot<T>(P, L.pos, L.de, sob, 0, rl + 16 * k, rpair[k], npairs, nobs_rows, m, n, ca[k], cbk, b[k], iak, ibk, vld, bad);
    iab[k] = iak | (ibk << 8);
    vmask |= vld ? 1u << k : 0u;
  }
  const bool env_bad = r16::allreduce(bad ? 1 : 0, wv::Max()) != 0;

  // ---- per-row solver state (uniform within a 16-lane row) ----
  int phase = (!live || env_bad) ? 2 : 0;                      // 0 scan, 1 step, 2 done
  bool converged = false;
  int q = 0, it = 0;
  T wca = T(0), wcb = T(0), wb = T(0), lam_new = T(0);
  int wia = 0, wib = 0, wown = 0, wkk = 0;
  for (int guard = 0; guard < max_iter + 2; ++guard) {
    if (!__any(phase != 2)) break;
    // ================= scan: most violated row outside the active set =================
    {
      T best = T(0);
      int best_k = 0;
#pragma unroll
      for (int k = 0; k < RL; ++k) {
        const int iak = iab[k] & 255, ibk = iab[k] >> 8;
        const T res = m_fma(ca[k], L.u[iak] - (ibk != iak ? L.u[ibk] : T(0)), -b[k]);
        const T sc = (((vmask & ~amask) >> k) & 1u) && res > T(0) ? res * res : T(0);
        if (sc > best) {
          best = sc;
          best_k = k;
        }
      }
      const T wbest = r16::allreduce(best, wv::Max());
      const bool scan = phase == 0;
      const bool conv_now = scan && !(wbest > tol2);
      const bool sel = scan && !conv_now;
      const int owner = r16::first(best == wbest, lane);
      const int kk = r16::get(best_k, owner);
      T sa = ca[0], sb_ = b[0];
      int siab = iab[0];
#pragma unroll
      for (int k = 1; k < RL; ++k) {
        sa = kk == k ? ca[k] : sa;
        sb_ = kk == k ? b[k] : sb_;
        siab = kk == k ? iab[k] : siab;
      }
      const T nca = r16::get(sa, owner), nb = r16::get(sb_, owner);
      const int niab = r16::get(siab, owner);
      const int nia = niab & 255, nib = niab >> 8;
      const T ncb = nib != nia ? -nca : T(0);
      if (conv_now) {
        converged = true;
        phase = 2;
      }
      if (sel) {
        wca = nca; wcb = ncb; wb = nb; wia = nia; wib = nib; wown = owner; wkk = kk;
        lam_new = T(0);
        phase = 1;
      }
    }
    // ================= one add-or-drop step for the rows in phase 1 =================
    const bool step = phase == 1;
    const bool over = step && it + 1 > max_iter;
    if (step) ++it;
    if (over) phase = 2;                                        // not converged: falls back
    const bool stepping = step && !over;
    const bool two = wib != wia;
    const T wcb2 = two ? wcb : T(0);
    const T res = m_fma(wca, L.u[wia], m_fma(wcb2, L.u[wib], -wb));
    const T dc = rl < q ? m_fma(wca, L.Q[wia][rl], wcb2 * L.Q[wib][rl]) : T(0);                   // d = Q^T a
    const T my_lam = L.lam[rl], my_di = L.di[rl], my_u = L.u[rl];
    T zv = rl < n ? ((rl == wia ? wca : T(0)) + ((two && rl == wib) ? wcb : T(0))) : T(0);      // z = a - Q d
    const int qmax = wv::allreduce(stepping ? q : 0, wv::Max());                                  // wave-uniform loop bound
#pragma unroll
    for (int c = 0; c < NMAX; ++c) {
      if (c >= qmax) break;
      const T dcc = r16::get(dc, c);
      zv = c < q ? m_fma(-L.Q[rl][c], dcc, zv) : zv;
    }
    if (rl >= n) zv = T(0);
    T rc = dc * my_di;                                                                            // r = R^-1 d on the row-scaled system
#pragma unroll
    for (int k = NMAX - 1; k >= 0; --k) {
      if (k >= qmax) continue;
      const T rk = r16::get(rc, k);
      rc = (k < q && rl < k) ? m_fma(-(L.R[rl][k] * my_di), rk, rc) : rc;
    }
    const T zz = r16::allreduce(zv * zv, wv::Add());
    const T rmax = r16::allreduce(rl < q ? m_abs(rc) : T(0), wv::Max());
    T t1v = GiEps<T>::inf;
    if (rl < q && rc > GiEps<T>::r * rmax && rc > T(0)) t1v = m_max(my_lam, T(0)) * m_rcp(rc);
    const T t1 = r16::allreduce(t1v, wv::Min());
    const int drop = t1 < GiEps<T>::inf ? r16::first(t1v == t1, lane) : 0;
    const bool has_z = zz > GiEps<T>::z && (MDS_TUNE_HASZ != 1 || q < n);                                                 // (q == n: only the dual step, as in gi_solve)
    const T t2 = has_z ? res * m_rcp(zz) : GiEps<T>::inf;
    const T t = m_min(t1, t2);
    const bool nostep = stepping && !(t < GiEps<T>::inf);                                         // rows inconsistent: falls back
    if (nostep) phase = 2;
    const bool doing = stepping && !nostep;
    const bool full = has_z && t2 <= t1;
    MDS_WAVE_SYNC();
    if (doing && has_z && rl < n) L.u[rl] = m_fma(-t, zv, my_u);
    if (doing && rl < q) L.lam[rl] = m_fma(-t, rc, my_lam);
    if (doing) lam_new += t;
    MDS_WAVE_SYNC();
    const bool adding = doing && full, dropping = doing && !full;
    if (adding) {                                                                                 // N <- [N a]
      const T inz = m_rsqrt(zz), nz = zz * inz;
      if (rl < n) L.Q[rl][q] = zv * inz;
      if (rl < q) L.R[rl][q] = dc;
      if (rl == 0) {
        L.R[q][q] = nz;
        L.di[q] = inz;
        L.lam[q] = lam_new;
        L.act[q] = wown + 16 * wkk;
      }
      if (rl == wown) amask |= 1u << wkk;
      ++q;
      phase = 0;
    }
    MDS_WAVE_SYNC();
    if (__any(dropping)) {                                                                        // drop active column `drop`
      const int dcol = dropping ? drop : 0;
      const int drow = L.act[dcol];
      if (dropping && rl == (drow & 15)) amask &= ~(1u << (drow >> 4));
      T lnext = T(0);
      int anext = 0;
      const bool shift = dropping && rl >= dcol && rl < q - 1;
      if (shift) {
        lnext = L.lam[rl + 1];
        anext = L.act[rl + 1];
      }
      MDS_WAVE_SYNC();
      if (shift) {
        L.lam[rl] = lnext;
        L.act[rl] = anext;
      }
      if (dropping && rl < q)                                                                     // each lane shifts its own row of R
        for (int k = dcol; k < q - 1; ++k) L.R[rl][k] = L.R[rl][k + 1];
      MDS_WAVE_SYNC();
      const int lmax = wv::allreduce(dropping ? q - 1 : 0, wv::Max());
      for (int l = 0; l < lmax; ++l) {                                                            // Givens on rows l, l+1 (rows with dcol <= l < q-1)
        const bool lact = dropping && l >= dcol && l < q - 1;
        const T a = L.R[l][l], bb = L.R[l + 1 < NMAX ? l + 1 : l][l];
        const T rr = m_sqrt(m_fma(a, a, bb * bb));
        const T cs = rr > T(0) ? a / rr : T(1), sn = rr > T(0) ? bb / rr : T(0);
        MDS_WAVE_SYNC();
        if (lact && rl >= l && rl < q - 1) {
          const T x = L.R[l][rl], y = L.R[l + 1][rl];
          L.R[l][rl] = m_fma(cs, x, sn * y);
          L.R[l + 1][rl] = m_fma(-sn, x, cs * y);
        }
        if (lact && rl < n) {
          const T x = L.Q[rl][l], y = L.Q[rl][l + 1];
          L.Q[rl][l] = m_fma(cs, x, sn * y);
          L.Q[rl][l + 1] = m_fma(-sn, x, cs * y);
        }
        MDS_WAVE_SYNC();
      }
      if (dropping) --q;
      if (dropping && rl >= dcol && rl < q) L.di[rl] = T(1) / L.R[rl][rl];
      MDS_WAVE_SYNC();
    }
  }
  if (phase != 2) converged = false;                                                              // guard ran out
  // final certificate for rows that iterated: EVERY row holds at the returned point (see gi_solve)
  {
    T worst = T(0);
#pragma unroll
    for (int k = 0; k < RL; ++k) {
      const int iak = iab[k] & 255, ibk = iab[k] >> 8;
      const T res = m_fma(ca[k], L.u[iak] - (ibk != iak ? L.u[ibk] : T(0)), -b[k]);
      if ((vmask >> k) & 1u) worst = m_max(worst, res);
    }
    worst = r16::allreduce(worst, wv::Max());
    if (converged && it > 0 && worst * worst > T(100) * tol2) converged = false;
  }
  if (live && rl == 0) {
    status[env] = converged ? 0 : 1;
    if (cost_out) cost_out[env] = it;
  }
  if (live && rl < D) {
    T u[4] = {un[0], un[1], un[2], un[3]};
    if (converged) {
      u[0] = L.u[rl];
      for (int k = 1; k < 4; ++k) u[k] = m_clamp(un[k], -P.umax[k], P.umax[k]);
    }
    store4<S, T>(usafe + ((size_t)env * D + rl) * 4, u);
  }
}

}  // namespace mds
