// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for scalar and packed fp32, dependent and independent chains,
// at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP 256
template <int MODE> __global__ void k(float* out, unsigned long long* cyc, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float m = 1.0000001f, c = 1e-9f;
  const f2 pm = {m, m}, pc = {c, c};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (MODE == 0) {  // 8 independent scalar fma chains
        a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
        a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
      } else if (MODE == 1) {  // one dependent scalar chain
        a0 = fmaf(a0, m, c); a0 = fmaf(a0, m, c); a0 = fmaf(a0, m, c); a0 = fmaf(a0, m, c);
        a0 = fmaf(a0, m, c); a0 = fmaf(a0, m, c); a0 = fmaf(a0, m, c); a0 = fmaf(a0, m, c);
      } else if (MODE == 2) {  // 4 independent packed fma chains (8 instructions: two rounds)
        p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
        p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
        p0 = __builtin_elementwise_fma(p0, pm, pc); p1 = __builtin_elementwise_fma(p1, pm, pc);
        p2 = __builtin_elementwise_fma(p2, pm, pc); p3 = __builtin_elementwise_fma(p3, pm, pc);
      } else if (MODE == 3) {  // one dependent packed chain
        for (int q = 0; q < 8; ++q) p0 = __builtin_elementwise_fma(p0, pm, pc);
      } else if (MODE == 4) {  // packed mul, 4 independent chains
        p0 = p0 * pm; p1 = p1 * pm; p2 = p2 * pm; p3 = p3 * pm; p0 = p0 * pm; p1 = p1 * pm; p2 = p2 * pm; p3 = p3 * pm;
      } else if (MODE == 5) {  // two dependent scalar chains interleaved
        a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c);
        a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  const int block = 64 * 4 * (waves_per_simd > 4 ? 4 : waves_per_simd), cus = 256 * (waves_per_simd > 4 ? waves_per_simd / 4 : 1);   // workgroups of <= 16 waves, 4 SIMDs per CU
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * cus * block); hipMalloc(&cyc, 8 * cus);
  const int iters = 200;
  k<MODE><<<cus, block>>>(out, cyc, iters, 1.0f); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<MODE><<<cus, block>>>(out, cyc, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(cus); hipMemcpy(h.data(), cyc, 8 * cus, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= cus;
  const double instr = (double)iters * REP;   // per wave
  printf("%-34s waves/SIMD %d: %.0f memtime ticks, %.3f ms -> %.2f ticks / instr / wave, %.2f ns * SIMD per instr\n", name, waves_per_simd, mean, ms,
         mean / instr, ms * 1e6 / (instr * waves_per_simd));
  hipFree(out); hipFree(cyc);
}
int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("scalar fma, 8 independent", w);
    run<1>("scalar fma, dependent", w);
    run<5>("scalar fma, 2 dependent chains", w);
    run<2>("packed fma, 4 independent", w);
    run<3>("packed fma, dependent", w);
    run<4>("packed mul, 4 independent", w);
  }
  return 0;
}
