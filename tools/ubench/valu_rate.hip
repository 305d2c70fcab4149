// Diagnostic microbenchmark: VALU issue rate on gfx950 for v_fma_f32 vs v_pk_fma_f32 vs MFMA, at 1/2/4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, int iters) {
  float a[8];
  f32x2 p[8];
  f32x4 m[4];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; p[i] = (f32x2){a[i], a[i] + 1.f}; }
  for (int i = 0; i < 4; ++i) m[i] = (f32x4){a[i], a[i], a[i], a[i]};
  const float x = 0.999f + threadIdx.x * 1e-9f, y = 1e-3f;
  __syncthreads();
  const unsigned long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"((f32x2){x, x}), "v"((f32x2){y, y}));
    } else if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, m[i], 0, 0, 0);
    } else if (MODE == 3) {   // v_add_f32 / v_sub
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
    } else if (MODE == 4) {   // v_add_u32 (int)
      int b[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) b[i] = __float_as_int(a[i]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(b[i]) : "v"(it));
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __int_as_float(b[i]);
    }
  }
  const unsigned long long c1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  for (int i = 0; i < 4; ++i) s += m[i][0] + m[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = c1 - c0;
}

int main() {
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 1024 * 1024 * 4));
  CHECK(hipMalloc(&cyc, 1024 * 16 * 8));
  const int iters = 200;
  const char* names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mfma_f32_16x16x4_f32", "v_sub_f32", "v_add_u32"};
  for (int mode = 0; mode < 5; ++mode)
    for (int threads : {64, 256, 512, 1024}) {
      for (int rep = 0; rep < 2; ++rep) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        CHECK(hipDeviceSynchronize());
      }
      unsigned long long h[16];
      CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
      const double per = (double)h[0] / (iters * 32.0);
      const int wps = threads >= 256 ? threads / 256 : 1;
      printf("%-24s threads=%4d (%d wave/SIMD): %.2f cycles per instruction per wave -> %.2f cycles per instruction per SIMD\n",
             names[mode], threads, wps, per, per / wps);
    }
  return 0;
}
