// TEST TOOLING ONLY -- a stand-in for <hip/hip_runtime.h> that lets the CBF kernels of multidronesim_amd/csrc (the .hip sources as they
// stand) compile as HOST C++ and run one workgroup at a time on CPU threads: one std::thread per lane, 64 lanes per wavefront, every wave
// intrinsic the kernels use (ballot / any, v_readlane, the four DPP permutations, mbcnt, wave and workgroup barriers, the LDS atomic) as an
// exchange through a per-wave buffer between two pthread barriers.  `__shared__` becomes a function-local static (one workgroup runs at
// a time), so AddressSanitizer sees every LDS array with its real size and red zones, and UBSan every index expression -- the only
// memory-safety check this code can get (there is no GPU sanitizer on the pool).  Never part of the product.
#pragma once
#include <pthread.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <cmath>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static
#define __align__(n) __attribute__((aligned(n)))

namespace simt {
struct Dim {
  unsigned x = 0, y = 0, z = 0;
};
struct Wave {
  pthread_barrier_t bar;
  uint64_t xch[64];
};
struct Block {
  pthread_barrier_t bar;
  std::vector<Wave> waves;
};
struct Ctx {
  Dim tid, bid, bdim, gdim;
  int lane = 0;
  Wave* wave = nullptr;
  Block* block = nullptr;
  const void* kernarg = nullptr;
};
inline thread_local Ctx ctx;
inline std::atomic<long> collectives{0};      // (statistics only)

// every lane publishes `mine`, then reads what f makes of the 64 published values; two barriers per collective
template <typename F> inline uint64_t collective(uint64_t mine, F&& f) {
  Wave& w = *ctx.wave;
  w.xch[ctx.lane] = mine;
  pthread_barrier_wait(&w.bar);
  const uint64_t r = f(w.xch);
  pthread_barrier_wait(&w.bar);
  if (ctx.lane == 0) collectives.fetch_add(1, std::memory_order_relaxed);
  return r;
}
inline int dpp_src(int lane, int ctrl) {
  switch (ctrl) {
    case 0xB1: return (lane & ~3) | ((lane & 3) ^ 1);            // quad_perm [1,0,3,2]
    case 0x4E: return (lane & ~3) | ((lane & 3) ^ 2);            // quad_perm [2,3,0,1]
    case 0x141: return (lane & ~7) | (7 - (lane & 7));           // row_half_mirror
    case 0x140: return (lane & ~15) | (15 - (lane & 15));        // row_mirror
    default: __builtin_trap();
  }
}
// one workgroup at a time, `nthreads` lanes each on its own thread
template <typename K> inline void launch(unsigned grid, unsigned nthreads, const void* kernarg, K&& kernel) {
  for (unsigned b = 0; b < grid; ++b) {
    Block blk;
    const unsigned nw = (nthreads + 63) / 64;
    blk.waves.resize(nw);
    pthread_barrier_init(&blk.bar, nullptr, nthreads);
    for (unsigned w = 0; w < nw; ++w) pthread_barrier_init(&blk.waves[w].bar, nullptr, (nthreads - 64 * w) < 64 ? (nthreads - 64 * w) : 64);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nthreads; ++t)
      th.emplace_back([&, t]() {
        ctx.tid.x = t;
        ctx.bid.x = b;
        ctx.bdim.x = nthreads;
        ctx.gdim.x = grid;
        ctx.lane = (int)(t & 63);
        ctx.wave = &blk.waves[t >> 6];
        ctx.block = &blk;
        ctx.kernarg = kernarg;
        kernel();
      });
    for (auto& t : th) t.join();
    for (unsigned w = 0; w < nw; ++w) pthread_barrier_destroy(&blk.waves[w].bar);
    pthread_barrier_destroy(&blk.bar);
  }
}
}  // namespace simt

#define threadIdx (simt::ctx.tid)
#define blockIdx (simt::ctx.bid)
#define blockDim (simt::ctx.bdim)
#define gridDim (simt::ctx.gdim)

inline void __syncthreads() { pthread_barrier_wait(&simt::ctx.block->bar); }
inline unsigned long long __ballot(bool p) {
  return simt::collective(p ? 1u : 0u, [](const uint64_t* x) {
    uint64_t m = 0;
    for (int l = 0; l < 64; ++l) m |= (x[l] & 1u) << l;
    return m;
  });
}
inline bool __any(bool p) { return __ballot(p) != 0; }
inline bool __all(bool p) { return __ballot(!p) == 0; }
inline int __builtin_amdgcn_readlane(int v, int l) {
  return (int)(unsigned)simt::collective((unsigned)v, [l](const uint64_t* x) { return x[l & 63]; });
}
inline int __builtin_amdgcn_readfirstlane(int v) { return __builtin_amdgcn_readlane(v, 0); }
inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 0); }
inline int __builtin_amdgcn_update_dpp(int /*old*/, int v, int ctrl, int /*row_mask*/, int /*bank_mask*/, bool /*bound_ctrl*/) {
  const int src = simt::dpp_src(simt::ctx.lane, ctrl);            // all 64 lanes active: `old` and bound_ctrl never matter
  return (int)(unsigned)simt::collective((unsigned)v, [src](const uint64_t* x) { return x[src]; });
}
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned add) {
  const int l = simt::ctx.lane;
  return add + (unsigned)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u)));
}
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned add) {
  const int l = simt::ctx.lane;
  return add + (l > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (l - 32)) - 1u)) : 0u);
}
inline void __builtin_amdgcn_wave_barrier() { pthread_barrier_wait(&simt::ctx.wave->bar); }
#define __builtin_amdgcn_fence(...) ((void)0)
inline void __builtin_amdgcn_s_setprio(int) {}
inline unsigned long long __builtin_amdgcn_s_memtime() { return 0; }
inline void* __builtin_amdgcn_kernarg_segment_ptr() { return const_cast<void*>(simt::ctx.kernarg); }
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
// (the older kernels' shuffles: declared so that their templates parse; the emulation does not instantiate them)
template <typename T> inline T __shfl(T v, int src, int width = 64) {
  uint64_t u = 0;
  memcpy(&u, &v, sizeof(T));
  const int l = simt::ctx.lane, s = (l & ~(width - 1)) | (src & (width - 1));
  u = simt::collective(u, [s](const uint64_t* x) { return x[s]; });
  memcpy(&v, &u, sizeof(T));
  return v;
}
template <typename T> inline T __shfl_xor(T v, int m, int width = 64) { return __shfl(v, (simt::ctx.lane ^ m) & (width - 1), width); }
template <typename T> inline T __shfl_up(T v, int d, int width = 64) {
  const int l = simt::ctx.lane & (width - 1);
  return __shfl(v, l >= d ? l - d : l, width);
}
// HIP's vector types and integer min / max, as far as the kernel sources use them
struct alignas(16) uint4 {
  unsigned x, y, z, w;
};
struct alignas(8) uint2 {
  unsigned x, y;
};
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline size_t min(size_t a, size_t b) { return a < b ? a : b; }
inline size_t max(size_t a, size_t b) { return a > b ? a : b; }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
