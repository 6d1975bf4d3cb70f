// ECBF safety filter kernels (a11-a15): constraint rows of cbf/cbf.py and the QP of
// cbf/qptracker.py:86-114, one wavefront per environment.
#include <hip/hip_runtime.h>

#include "mds_cbf.hpp"

namespace mds {

constexpr int kCbfMaxD = 32;      // drones per env supported by the wave-per-env kernels
constexpr int kCbfMaxObs = 16;

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
// one wavefront per env.  The active normals N (n x q, q <= n = drones per env) are kept as a
// thin QR (Q: n x q orthonormal columns, R: q x q upper triangular) in the wave's LDS slice;
// adding a row appends a Gram-Schmidt column, dropping one re-triangularises with Givens
// rotations.  Rows are normalised to unit length so every threshold is a distance.  The number
// of iterations is of the order of the number of active rows (Hildreth's coordinate ascent
// needs 10^2..10^4 on crowded scenes); the result is the exact minimiser, the same the oracle's
// qp_project computes.
// ------------------------------------------------------------------------------------
#define MDS_WAVE_SYNC()                                   \
  do {                                                    \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                      \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
template <typename T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = m_max(v, __shfl_xor(v, off));
  return v;
}
template <typename T> __device__ __forceinline__ void wave_argmin(T& val, int& idx) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const T ov = __shfl_xor(val, off);
    const int oi = __shfl_xor(idx, off);
    const bool take = (ov < val) || (ov == val && oi < idx);
    val = take ? ov : val;
    idx = take ? oi : idx;
  }
}
template <typename T, int R> __device__ __forceinline__ T pick(const T (&a)[R], int k) {
  T v = a[0];
#pragma unroll
  for (int q = 1; q < R; ++q) v = (q == k) ? a[q] : v;
  return v;
}
template <typename T> struct GiEps;
template <> struct GiEps<float> {
  static constexpr float z = 1e-9f, r = 1e-6f, inf = 3.0e38f;
};
template <> struct GiEps<double> {
  static constexpr double z = 1e-16, r = 1e-12, inf = 1.0e300;
};

// NV = QP variables per agent (order 2: thrust only -> 1; order 3: yank, wx, wy -> 3, wz is box-only),
// NMAX = compile-time bound on the number of QP variables n = NV * D (LDS footprint of Q, R ~ NMAX^2).
// WPB = wavefronts (= envs) per workgroup, chosen so that the LDS slices fit 160 KiB
template <typename T, typename S, int R, int NMAX, int ORDER, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_cbf_filter_gi(const CbfParams<T> P, const int E, const T kf, const int* __restrict__ pair_ij,
                                                       const T* __restrict__ obstacles, const S* __restrict__ obs,
                                                       const S* __restrict__ xdes, const S* __restrict__ unom,
                                                       S* __restrict__ usafe, int* __restrict__ status, const int max_iter,
                                                       const T tol2) {
  constexpr int NV = ORDER == 2 ? 1 : 3;
  constexpr int XD = ORDER == 2 ? 9 : 10;
  constexpr int kQS = NMAX + 1;     // padded LDS row stride (conflict-free column walks)
  constexpr int DMAX = NMAX / NV;
  __shared__ T sx[WPB][DMAX][XD], sxd[WPB][DMAX][XD];
  __shared__ T su_[WPB][NMAX], sd_[WPB][NMAX], slam_[WPB][NMAX];
  __shared__ T sQ_[WPB][NMAX][kQS], sR_[WPB][NMAX][kQS];
  __shared__ int sact_[WPB][NMAX];
  __shared__ __align__(16) S sraw[WPB][DMAX * 20];            // the env's observation rows, loaded coalesced
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int env = blockIdx.x * WPB + wave;
  if (env >= E) return;                                    // wave-uniform
  T* su = su_[wave];
  T* sd = sd_[wave];
  T* slam = slam_[wave];
  T(*sQ)[kQS] = sQ_[wave];
  T(*sR)[kQS] = sR_[wave];
  int* sact = sact_[wave];
  const int D = P.num_drones, n = NV * D;
  const size_t base = (size_t)env * D;
  // an env's D x 20 observation block, its D x xdim xdes block and D x 4 nominal block are contiguous
  for (int k = lane; k < D * 20; k += 64) sraw[wave][k] = obs[base * 20 + k];
  for (int k = lane; k < D * XD; k += 64) sxd[wave][k / XD][k % XD] = (T)xdes[base * XD + k];
  MDS_WAVE_SYNC();
  bool bad = false;
  for (int d = lane; d < D; d += 64) {       // obs_to_lin_model(obs, dim = 9 | 10) (model_conversions.py:20-58)
    const S* o = &sraw[wave][d * 20];
    T* x = sx[wave][d];
    x[0] = (T)o[7]; x[1] = (T)o[8]; x[2] = (T)o[9];
    if (ORDER == 3) {
      const T r0 = (T)o[16], r1 = (T)o[17], r2 = (T)o[18], r3 = (T)o[19];
      x[3] = kf * (r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);                                  // calc_z_thrust (:137-143)
    }
    x[XD - 6] = (T)o[10]; x[XD - 5] = (T)o[11]; x[XD - 4] = (T)o[12];
    x[XD - 3] = (T)o[0]; x[XD - 2] = (T)o[1]; x[XD - 1] = (T)o[2];
    for (int k = 0; k < NV; ++k) su[NV * d + k] = (T)unom[(base + d) * 4 + k];
  }
  MDS_WAVE_SYNC();

  const int npairs = cbf_num_pairs(D), nobs_rows = D * P.n_obs, m = npairs + nobs_rows + 2 * n;
  // unit-norm rows:  sum_k ca[k] u[NV ia + k] + cb[k] u[NV ib + k] <= b
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
      const int ij = pair_ij[r];
      ia[k] = ij & 255;
      ib[k] = ij >> 8;
      T hr, Lg[4];
      cbf_pair_row<T, ORDER>(P, sx[wave][ia[k]], sxd[wave][ia[k]], sx[wave][ib[k]], sxd[wave][ib[k]], false, P.Ds_pair, &hr, Lg);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        ca[k][v] = -Lg[v];
        cb[k][v] = Lg[v];
      }
      b[k] = hr;
    } else if (r < npairs + nobs_rows) {
      const int q = r - npairs, i = q / P.n_obs, o = q % P.n_obs;
      T xo[XD];
#pragma unroll
      for (int v = 0; v < XD - 3; ++v) xo[v] = T(0);
      xo[XD - 3] = obstacles[4 * o];
      xo[XD - 2] = obstacles[4 * o + 1];
      xo[XD - 1] = obstacles[4 * o + 2];
      T hr, Lg[4];
      cbf_pair_row<T, ORDER>(P, sx[wave][i], sxd[wave][i], xo, xo, true, P.safety_radius + obstacles[4 * o + 3], &hr, Lg);
      ia[k] = ib[k] = i;
#pragma unroll
      for (int v = 0; v < NV; ++v) ca[k][v] = -Lg[v];
      b[k] = hr;
    } else if (r < m) {                                    // +-u_var <= umax (cbf/cbf.py:400-412)
      const int q = r - npairs - nobs_rows, var = q % n;
      ia[k] = ib[k] = var / NV;
#pragma unroll
      for (int v = 0; v < NV; ++v)
        if (v == var % NV) ca[k][v] = q < n ? T(1) : T(-1);
      b[k] = P.umax[var % NV];
    }
    if (r < m) {
      T n2 = T(0);
#pragma unroll
      for (int v = 0; v < NV; ++v) n2 = m_fma(ca[k][v], ca[k][v], m_fma(cb[k][v], cb[k][v], n2));
      if (n2 > T(0)) {
        const T inv = T(1) / m_sqrt(n2);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          ca[k][v] *= inv;
          cb[k][v] *= inv;
        }
        b[k] *= inv;
        valid[k] = true;
      } else if (b[k] < T(0)) {
        bad = true;                                        // 0 * u <= h with h < 0
      }
    }
  }
  // order 3: the omega_z input only has box rows, +-umax_3 and the force box written to its column
  // (custom_force_bound_const, cbf/cbf.py:446-464, quirk kept): a 1-D interval per agent
  T wz_lo = -P.umax[3], wz_hi = P.umax[3];
  if (ORDER == 3 && lane < D) {
    const T F = sx[wave][lane][3];
    wz_hi = m_min(wz_hi, P.k[2] * (P.Fmax - F));
    wz_lo = m_max(wz_lo, -(P.k[2] * (F - P.Fmin)));
    if (wz_lo > wz_hi) bad = true;
  }
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
    T wbest = best;
    int wrow = lane + 64 * best_k;
    wave_argmax(wbest, wrow);
    if (!(wbest > tol2)) {
      converged = true;
      break;
    }
    const int owner = wrow & 63, kk = wrow >> 6;
    T wca[NV], wcb[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      T tca = ca[0][v], tcb = cb[0][v];
#pragma unroll
      for (int k = 1; k < R; ++k) {
        tca = (k == kk) ? ca[k][v] : tca;
        tcb = (k == kk) ? cb[k][v] : tcb;
      }
      wca[v] = __shfl(tca, owner);
      wcb[v] = __shfl(tcb, owner);
    }
    const T wb = __shfl(pick<T, R>(b, kk), owner);
    const int wia = __shfl(pick<int, R>(ia, kk), owner), wib = __shfl(pick<int, R>(ib, kk), owner);
    const bool two = wib != wia;
    T lam_new = T(0);
    // ---- bring that row into the active set, dropping blocking rows on the way ----
    while (true) {
      if (++it > max_iter) {
        infeasible = true;
        break;
      }
      T res = -wb;
#pragma unroll
      for (int v = 0; v < NV; ++v) res = m_fma(wca[v], su[NV * wia + v], m_fma(two ? wcb[v] : T(0), su[NV * wib + v], res));
      T dc = T(0);
      if (lane < q) {                                                                    // d = Q^T a
#pragma unroll
        for (int v = 0; v < NV; ++v) dc = m_fma(wca[v], sQ[NV * wia + v][lane], m_fma(two ? wcb[v] : T(0), sQ[NV * wib + v][lane], dc));
      }
      if (lane < n) sd[lane] = dc;
      MDS_WAVE_SYNC();
      T zv = T(0);                                                                       // z = a - Q d
      if (lane < n) {
        const int ag = lane / NV, vv = lane - ag * NV;
#pragma unroll
        for (int v = 0; v < NV; ++v)
          if (v == vv) zv = (ag == wia ? wca[v] : T(0)) + ((two && ag == wib) ? wcb[v] : T(0));
        for (int c = 0; c < q; ++c) zv = m_fma(-sQ[lane][c], sd[c], zv);
      }
      const T zz = wave_sum(zv * zv);
      T rc = dc;                                                                         // r = R^-1 d
      for (int k = q - 1; k >= 0; --k) {
        const T rk = __shfl(rc, k) / sR[k][k];
        if (lane == k) rc = rk;
        else if (lane < k) rc = m_fma(-sR[lane][k], rk, rc);
      }
      const T rmax = wave_max(lane < q ? m_abs(rc) : T(0));
      T t1 = GiEps<T>::inf;
      int drop = lane;
      if (lane < q && rc > GiEps<T>::r * rmax && rc > T(0)) t1 = m_max(slam[lane], T(0)) / rc;
      wave_argmin(t1, drop);
      const bool has_z = zz > GiEps<T>::z;
      const T t2 = has_z ? res / zz : GiEps<T>::inf;
      const T t = m_min(t1, t2);
      if (!(t < GiEps<T>::inf)) {
        infeasible = true;                                                               // no step possible: rows inconsistent
        break;
      }
      const bool full = has_z && t2 <= t1;
      MDS_WAVE_SYNC();
      if (has_z && lane < n) su[lane] = m_fma(-t, zv, su[lane]);
      if (lane < q) slam[lane] = m_fma(-t, rc, slam[lane]);
      lam_new += t;
      MDS_WAVE_SYNC();
      if (full) {                                                                        // add: N <- [N a]
        const T nz = m_sqrt(zz);
        if (lane < n) sQ[lane][q] = zv / nz;
        if (lane < q) sR[lane][q] = dc;
        if (lane == 0) {
          sR[q][q] = nz;
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
    worst = wave_max(worst);
    if (worst * worst > T(100) * tol2) converged = false;
  }
  if (lane == 0) status[env] = converged ? 0 : 1;
  MDS_WAVE_SYNC();
  for (int d = lane; d < D; d += 64) {
    T u[4];
    for (int k = 0; k < 4; ++k) u[k] = (T)unom[(base + d) * 4 + k];
    if (converged) {
      for (int k = 0; k < NV; ++k) u[k] = su[NV * d + k];
      if (ORDER == 2) {
        for (int k = 1; k < 4; ++k) u[k] = m_clamp(u[k], -P.umax[k], P.umax[k]);
      } else {
        u[3] = m_clamp(u[3], wz_lo, wz_hi);
      }
    }
    for (int k = 0; k < 4; ++k) usafe[(base + d) * 4 + k] = (S)u[k];
  }
}

}  // namespace mds
