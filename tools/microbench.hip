// Calibration microbenchmarks (diagnostic only): what do the primitive operations of the
// per-graph kernels cost on this machine, measured in-kernel with the 100 MHz wall clock?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(1024) void probe(const float* __restrict__ buf, int n_elems, float* out,
                                              unsigned long long* stamps, int mode) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  unsigned long long t[12];
  int k = 0;
  const unsigned long long c0 = clock64();
  float acc = 0.f;
  t[k++] = wall_clock64();
  // 1. one dependent global load per thread (distinct lines), then barrier
  const int stride = 64;  // floats
  int idx = (blockIdx.x * nthr + tid) * stride % n_elems;
  float v = buf[idx];
  acc += v;
  __syncthreads();
  t[k++] = wall_clock64();
  // 2. second load depending on the first (pointer chase-ish)
  idx = (idx + (int)(v * 0.f) + 4096 * 17) % n_elems;
  v = buf[idx];
  acc += v;
  __syncthreads();
  t[k++] = wall_clock64();
  // 3. eight independent loads then barrier
  float r[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) r[u] = buf[(idx + (u + 1) * 8192 + tid * 16) % n_elems];
#pragma unroll
  for (int u = 0; u < 8; ++u) acc += r[u];
  __syncthreads();
  t[k++] = wall_clock64();
  // 4. ten bare barriers
#pragma unroll
  for (int u = 0; u < 10; ++u) __syncthreads();
  t[k++] = wall_clock64();
  // 5. LDS write + barrier + 64 dependent LDS reads
  lds[tid] = acc;
  __syncthreads();
  float s = 0.f;
  int j = tid;
  for (int u = 0; u < 64; ++u) { s += lds[j]; j = (j + 33) & (nthr - 1); }
  acc += s;
  __syncthreads();
  t[k++] = wall_clock64();
  // 6. 16 x (sqrtf + expf) dependent chain
  float e = fabsf(acc) * 1e-3f + 0.5f;
  for (int u = 0; u < 16; ++u) e = expf(-sqrtf(e)) + 0.25f;
  acc += e;
  __syncthreads();
  t[k++] = wall_clock64();
  // 7. 300 dependent fma
  float f = acc;
  for (int u = 0; u < 300; ++u) f = fmaf(f, 0.999f, 0.001f);
  acc += f;
  __syncthreads();
  t[k++] = wall_clock64();
  // 8. global store + barrier
  out[blockIdx.x * nthr + tid] = acc;
  __syncthreads();
  t[k++] = wall_clock64();
  const unsigned long long c1 = clock64();
  if (tid == 0) {
    for (int i = 0; i < k; ++i) stamps[blockIdx.x * 12 + i] = t[i];
    stamps[blockIdx.x * 12 + 10] = c1 - c0;                 // shader-clock ticks over the kernel
    stamps[blockIdx.x * 12 + 11] = t[k - 1] - t[0];         // 100 MHz ticks over the same span
  }
}

int main() {
  const int n = 64 << 20;  // 256 MB
  float *buf, *out;
  unsigned long long* st;
  CHECK(hipMalloc(&buf, n * sizeof(float)));
  CHECK(hipMemset(buf, 0, n * sizeof(float)));
  CHECK(hipMalloc(&out, 1024 * 1024 * sizeof(float)));
  CHECK(hipMalloc(&st, 1024 * 12 * sizeof(unsigned long long)));
  const char* names[] = {"1 global load + barrier", "dependent 2nd load + barrier", "8 indep loads + barrier",
                         "10 bare barriers", "lds wr + barrier + 64 dep lds reads + barrier", "16x(sqrt+exp) + barrier",
                         "300 dep fma + barrier", "global store + barrier"};
  for (int cfg = 0; cfg < 8; ++cfg) {
    const int blocks = (cfg & 1) ? 256 : 1;
    const int threads = (cfg & 2) ? 1024 : 256;
    const int burst = (cfg & 4) ? 20000 : 3;   // back-to-back launches before the measured one (clock ramp)
    for (int rep = 0; rep < burst; ++rep) {
      hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 4096 * 4, 0, buf, n, out, st, 0);
      if (burst <= 3) CHECK(hipDeviceSynchronize());
    }
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 12);
    CHECK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    printf("blocks=%d threads=%d burst=%d  shader clock over the kernel: %.0f MHz (median over blocks, us):\n", blocks, threads, burst,
           100.0 * (double)h[10] / (double)h[11]);
    for (int p = 0; p < 8; ++p) {
      std::vector<double> d;
      for (int b = 0; b < blocks; ++b) d.push_back((double)(h[b * 12 + p + 1] - h[b * 12 + p]) / 100.0);
      std::sort(d.begin(), d.end());
      printf("   %-48s %6.2f (max %6.2f)\n", names[p], d[d.size() / 2], d.back());
    }
  }
  return 0;
}
