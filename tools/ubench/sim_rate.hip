// Diagnostic microbenchmark: what bounds the issue rate of cloud_sim_kernel's inner loop on gfx950?
// The loop is, per pair (row i uniform, column j per lane) and feature f:  d = z_i[f] - z_j[f];  acc = fma(d, d, acc)
// with z_i in SGPRs and z_j in VGPRs.  Variants isolate the accumulation chain, the SGPR operand, the occupancy.
//   build:  hipcc --offload-arch=gfx950 -O3 -o /tmp/sim_rate tools/ubench/sim_rate.hip     run: /tmp/sim_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: one chain, SGPR row operand (the kernel's form)      F features held per lane
// MODE 1: two chains (even / odd f)
// MODE 2: four chains
// MODE 3: row operand in VGPRs (uniform values)
// MODE 4: fma only (acc = fma(s, v, acc)), one chain
// MODE 5: sub only (independent results)
// MODE 6: one chain, but the subtraction batched: 16 subs, then 16 fmas
template <int MODE, int F, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const float* __restrict__ z, float* out, int rows) {
  float zj[F];
  const int lane = threadIdx.x;
#pragma unroll
  for (int f = 0; f < F; ++f) zj[f] = z[(lane * 67 + f) & 4095];
  float srow[F];
#pragma unroll
  for (int f = 0; f < F; ++f) {
    float v = z[(blockIdx.x + f) & 4095];
    if constexpr (MODE != 3) v = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
    else asm volatile("" : "+v"(v));
    srow[f] = v;
  }
  float tot = 0.f;
#pragma unroll 1
  for (int i = 0; i < rows; ++i) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if constexpr (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int f = 0; f < F; ++f) { const float d = srow[f] - zj[f]; a0 = __builtin_fmaf(d, d, a0); }
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int f = 0; f < F; f += 2) {
        const float d = srow[f] - zj[f]; a0 = __builtin_fmaf(d, d, a0);
        const float e = srow[f + 1] - zj[f + 1]; a1 = __builtin_fmaf(e, e, a1);
      }
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int f = 0; f < F; f += 4) {
        const float d = srow[f] - zj[f]; a0 = __builtin_fmaf(d, d, a0);
        const float e = srow[f + 1] - zj[f + 1]; a1 = __builtin_fmaf(e, e, a1);
        const float g = srow[f + 2] - zj[f + 2]; a2 = __builtin_fmaf(g, g, a2);
        const float h = srow[f + 3] - zj[f + 3]; a3 = __builtin_fmaf(h, h, a3);
      }
    } else if constexpr (MODE == 4) {
#pragma unroll
      for (int f = 0; f < F; ++f) a0 = __builtin_fmaf(srow[f], zj[f], a0);
    } else if constexpr (MODE == 5) {
      float d[8];
#pragma unroll
      for (int f = 0; f < F; f += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { d[u] = srow[f + u] - zj[f + u]; asm volatile("" : "+v"(d[u])); }
      }
      a0 = d[0];
    } else if constexpr (MODE == 6) {
#pragma unroll
      for (int f = 0; f < F; f += 16) {
        float d[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { d[u] = srow[f + u] - zj[f + u]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) asm volatile("" : "+v"(d[u]));
#pragma unroll
        for (int u = 0; u < 16; ++u) a0 = __builtin_fmaf(d[u], d[u], a0);
      }
    }
    tot += (a0 + a1) + (a2 + a3);
    // keep the row operand "fresh" so that nothing is hoisted out of the row loop
#pragma unroll
    for (int f = 0; f < F; f += 16) {
      if constexpr (MODE != 3) asm volatile("" : "+s"(srow[f]));
      else asm volatile("" : "+v"(srow[f]));
    }
#pragma unroll
    for (int f = 0; f < F; ++f) asm volatile("" : "+v"(zj[f]));
  }
  out[blockIdx.x * THREADS + threadIdx.x] = tot;
}

template <int MODE, int F, int THREADS>
int run(const char* name, const float* z, float* out) {
  const int rows = 2048, grid = 1024;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<MODE, F, THREADS>), dim3(grid), dim3(THREADS), 0, 0, z, out, rows);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  int occ = 0;
  CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<MODE, F, THREADS>, THREADS, 0));
  const double ninstr = (MODE == 4 || MODE == 5 ? 1.0 : 2.0) * F * rows;         // per wave
  const double waves = (double)grid * THREADS / 64;
  const double per_simd = ninstr * waves / 1024.0;                                // wave instructions per SIMD
  const double cyc = best * 1e-3 * 2.4e9;
  printf("%-44s F=%2d threads=%4d WG/CU=%d (waves/SIMD %4.1f): %8.1f us -> %.2f cycles per instruction per SIMD (at 2.4 GHz)\n",
         name, F, THREADS, occ, occ * THREADS / 256.0, best * 1e3, cyc / per_simd);
  return 0;
}

int main() {
  float *z, *out;
  CHECK(hipMalloc(&z, 4096 * 4));
  CHECK(hipMalloc(&out, 1024 * 1024 * 4));
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 37) % 101) * 0.01f;
  CHECK(hipMemcpy(z, h, sizeof(h), hipMemcpyHostToDevice));
  run<0, 64, 1024>("one chain, SGPR row (the kernel's form)", z, out);
  run<1, 64, 1024>("two chains", z, out);
  run<2, 64, 1024>("four chains", z, out);
  run<6, 64, 1024>("one chain, 16 subs then 16 fmas", z, out);
  run<4, 64, 1024>("fma only (SGPR x VGPR), one chain", z, out);
  run<5, 64, 1024>("sub only, independent", z, out);
  run<0, 32, 1024>("one chain, SGPR row", z, out);
  run<3, 32, 1024>("one chain, row operand in VGPRs", z, out);
  run<0, 64, 512>("one chain, SGPR row", z, out);
  run<0, 64, 256>("one chain, SGPR row", z, out);
  run<2, 64, 256>("four chains", z, out);
  run<0, 32, 256>("one chain, SGPR row", z, out);
  run<2, 32, 256>("four chains", z, out);
  return 0;
}
