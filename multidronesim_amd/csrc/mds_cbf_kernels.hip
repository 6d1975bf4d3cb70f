// ECBF safety filter kernels (a11-a15): constraint rows of cbf/cbf.py and the QP of
// cbf/qptracker.py:86-114, one wavefront per environment.
#include <hip/hip_runtime.h>

#include "mds_cbf.hpp"

namespace mds {

constexpr int kCbfMaxD = 32;      // drones per env supported by the wave-per-env kernels
constexpr int kCbfMaxObs = 16;

// (i, j) of pair row r in the reference's lexicographic order (cbf/cbf.py:342-346); built on the host
struct CbfTables {
  const int* pair_ij;   // [D(D-1)/2] packed i | j << 8
};

// ------------------------------------------------------------------------------------
// Dense G, h exactly as CBF._build_ineq_const returns them (parity surface; one workgroup
// per env).  Row order: pairs | +I(4D) | -I(4D) | [order 3: 2 force rows per agent] | obstacles.
// ------------------------------------------------------------------------------------
template <typename T, typename S>
__global__ __launch_bounds__(256) void k_cbf_rows(const CbfParams<T> P, const int E, const int* __restrict__ pair_ij,
                                                  const T* __restrict__ obstacles, const S* __restrict__ x,
                                                  const S* __restrict__ xdes, S* __restrict__ G, S* __restrict__ h) {
  __shared__ T sx[kCbfMaxD][10], sxd[kCbfMaxD][10];
  const int env = blockIdx.x;
  const int D = P.num_drones, xd = P.order == 2 ? 9 : 10;
  const int npairs = cbf_num_pairs(D), m = cbf_num_rows(D, P.order, P.n_obs), ncol = 4 * D;
  for (int k = threadIdx.x; k < D * xd; k += blockDim.x) {
    sx[k / xd][k % xd] = (T)x[(size_t)env * D * xd + k];
    sxd[k / xd][k % xd] = (T)xdes[(size_t)env * D * xd + k];
  }
  S* Ge = G + (size_t)env * m * ncol;
  S* he = h + (size_t)env * m;
  for (int k = threadIdx.x; k < m * ncol; k += blockDim.x) Ge[k] = (S)0;
  __syncthreads();
  const int box0 = npairs, force0 = npairs + 8 * D, obs0 = force0 + (P.order == 3 ? 2 * D : 0);
  for (int r = threadIdx.x; r < m; r += blockDim.x) {
    T hr, Lg[4];
    if (r < npairs) {
      const int ij = pair_ij[r], i = ij & 255, j = ij >> 8;
      cbf_pair_row(P, sx[i], sxd[i], sx[j], sxd[j], false, P.Ds_pair, &hr, Lg);
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
      T xo[10];
      for (int k = 0; k < 10; ++k) xo[k] = T(0);
      xo[xd - 3] = obstacles[4 * o];
      xo[xd - 2] = obstacles[4 * o + 1];
      xo[xd - 1] = obstacles[4 * o + 2];
      cbf_pair_row(P, sx[i], sxd[i], xo, xo, true, P.safety_radius + obstacles[4 * o + 3], &hr, Lg);
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
// rows are feasible; otherwise the iteration cap trips and the env falls back to u_hat with
// status 1 (the reference falls back when cvxopt raises, qptracker.py:30-34).
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
      cbf_pair_row(P, sx[wave][ii[k]], sxd[wave][ii[k]], sx[wave][jj[k]], sxd[wave][jj[k]], false, P.Ds_pair, &hr, Lg);
      ci[k] = -Lg[0];
      cj[k] = Lg[0];
      b[k] = hr;
    } else if (r < npairs + nobs_rows) {
      const int q = r - npairs, i = q / P.n_obs, o = q % P.n_obs;
      T xo[9] = {T(0), T(0), T(0), T(0), T(0), T(0), obstacles[4 * o], obstacles[4 * o + 1], obstacles[4 * o + 2]};
      T hr, Lg[4];
      cbf_pair_row(P, sx[wave][i], sxd[wave][i], xo, xo, true, P.safety_radius + obstacles[4 * o + 3], &hr, Lg);
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

}  // namespace mds
