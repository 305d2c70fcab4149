// Device-side building blocks shared by the gfx950 kernels.
//   * wave = 64 lanes; workgroups are 256 threads (LDS-tile mode) or 1024 (global-tile mode)
//   * f32-input MFMA (v_mfma_f32_16x16x4_f32): bit-for-bit a k-ordered fmaf chain, so the
//     feature contractions keep exact fp32 semantics
//   * all reductions use a fixed order (no atomics): results are reproducible run to run
#ifndef AGCN_DEVICE_H_
#define AGCN_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace agcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Exact x / d for x * d < 2^32 (d >= 1) by one mul-hi.
struct FastDiv {
  uint32_t m, d;
  __host__ __device__ explicit FastDiv(uint32_t d_) : m(0), d(d_) {
    if (d_ > 1) m = (uint32_t)(((1ull << 32) + d_ - 1) / d_);
  }
  __device__ __forceinline__ uint32_t div(uint32_t x) const { return d == 1 ? x : __umulhi(x, m); }
};

__device__ __forceinline__ int round4(int v) { return (v + 3) & ~3; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;  // lane 0 holds the total
}

// Deterministic block-wide sum, result broadcast to every thread.
// red: >= 32 floats of LDS.  Contains barriers: call from uniform control flow.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    const int nw = blockDim.x >> 6;
    for (int i = 0; i < nw; ++i) t += red[i];
    red[31] = t;
  }
  __syncthreads();
  return red[31];
}

// relu(x) - alpha * relu(-x)   (models/operators/model_operatos.py:534-558, tensor alpha)
__device__ __forceinline__ float lrelu(float x, float alpha) {
  return fmaxf(x, 0.f) - alpha * fmaxf(-x, 0.f);
}
// TF relu gradient: 1 where the argument is > 0  =>  [x > 0] + alpha [x < 0]
__device__ __forceinline__ float lrelu_grad(float x, float alpha) {
  return x > 0.f ? 1.f : (x < 0.f ? alpha : 0.f);
}

// TF-0.12 clip_by_norm(t, 1) / clip_by_average_norm(t, 1):
//   scale = min(rsqrt(sum t^2) [* numel], 1)
__device__ __forceinline__ float clip_scale(float ss, float q) {
  const float inv = 1.0f / sqrtf(ss);
  return fminf(inv * q, 1.0f);
}

// One 16 x (16*NT) accumulator strip: acc[j] += A[row0.., :Kd] . B[:Kd, col0 + 16 j ..].
//   la(i, k) -> A[i][k],  lb(k, c) -> B[k][c]; both must return 0 outside their matrix.
// Lane l holds A[row0 + (l & 15)][k0 + (l >> 4)] and B[k0 + (l >> 4)][col + (l & 15)];
// result element (row0 + 4 * (l >> 4) + i, col0 + 16 j + (l & 15)) = acc[j][i].
template <int NT, class LA, class LB>
__device__ __forceinline__ void mfma_acc(f32x4 (&acc)[NT], int row0, int col0, int Kd, LA la, LB lb) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  for (int k0 = 0; k0 < Kd; k0 += 4) {
    const float av = la(row0 + r, k0 + q);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float bv = lb(k0 + q, col0 + 16 * j + r);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
    }
  }
}

// Workgroup-cooperative tiling of an [M x N] result: 16 x (16*NT) strips are dealt
// round-robin to the waves.  body(acc, row0, col0) accumulates, st(i, c, v) stores one
// element (and must bounds-check).
template <int NT, class BODY, class ST>
__device__ __forceinline__ void wg_tiles(int M, int N, BODY body, ST st) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int tm = (M + 15) >> 4, tn = (N + 16 * NT - 1) / (16 * NT);
  for (int t = wave; t < tm * tn; t += nw) {
    const int in = t / tm, im = t - in * tm;
    const int row0 = im * 16, col0 = in * 16 * NT;
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    body(acc, row0, col0);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) st(row0 + q * 4 + i, col0 + 16 * j + r, acc[j][i]);
  }
}

template <int NT, class LA, class LB, class ST>
__device__ __forceinline__ void wg_gemm(int M, int N, int Kd, LA la, LB lb, ST st) {
  wg_tiles<NT>(M, N, [&](f32x4 (&acc)[NT], int row0, int col0) { mfma_acc<NT>(acc, row0, col0, Kd, la, lb); }, st);
}

// Enumerates the 2x2 blocks (bi <= bj) of the upper triangle of an nb2 x nb2 block grid
// without idle iterations: rows r and nb2-1-r are folded into one strip of nb2+1 entries.
struct TriFold {
  int nb2, H, W;
  __device__ explicit TriFold(int nb2_) : nb2(nb2_), H((nb2_ + 1) >> 1), W(nb2_ + 1) {}
  __device__ __forceinline__ int count() const { return H * W; }
  __device__ __forceinline__ bool get(int t, const FastDiv& dW, int& bi, int& bj) const {
    const int r = (int)dW.div((uint32_t)t), c = t - r * W;
    if (c < nb2 - r) { bi = r; bj = r + c; return true; }
    const int r2 = nb2 - 1 - r;
    if (r2 == r) return false;  // middle row of an odd grid is covered by the first branch
    bi = r2; bj = r2 + (c - (nb2 - r));
    return true;
  }
};

}  // namespace agcn
#endif  // AGCN_DEVICE_H_
