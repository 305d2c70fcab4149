// Parameter gradients that reduce over every node of the batch:
//   dweight[f*K + k, c] = sum_r X_k[r, f] Gp[r, c]      (graphconv.py:245-247 under TF autodiff)
//   dbias[c]            = sum_r Gp[r, c]
//   dM_L[f, c]          = sum_r X[r, f] dZ[r, c]        (paper mode only)
//   dalpha, dbeta       = sum over graphs of the per-graph partials
// with Gp = dY . act'(Y).  Two launches: per-chunk partial products on the matrix cores
// (A^T B with the row index as the MFMA k dimension), then a fixed-order reduction -- no
// atomics, so the result is reproducible bit for bit.
#include "agcn_device.h"
#include "sgc_kernels.h"

namespace agcn {

// grid: (items, problems).  problem p < K: dweight plane k = p; p == K: dbias; p == K + 1: dM_L.
// One wave per (row chunk, 16 x 32 output strip).
__global__ __launch_bounds__(256) void param_grad_partial_kernel(ParamGradArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int p = blockIdx.y;
  const int K = a.K, Fin = a.Fin, Fout = a.Fout;
  int Fa, Fb;       // output is [Fa x Fb]
  int64_t out_base; // offset inside one chunk's partial block
  if (p < K) { Fa = Fin; Fb = Fout; out_base = 0; }
  else if (p == K) { Fa = 1; Fb = Fout; out_base = (int64_t)K * Fin * Fout; }
  else { Fa = Fin; Fb = Fin; out_base = (int64_t)K * Fin * Fout + Fout; }
  const int tf = (Fa + 15) >> 4, tc = (Fb + 31) >> 5;
  const int items = tf * tc * a.n_chunks;
  const int item = blockIdx.x * 4 + wave;
  if (item >= items) return;
  const int chunk = item / (tf * tc);
  const int rem = item - chunk * (tf * tc);
  const int itc = rem / tf, itf = rem - itc * tf;
  const int f0 = itf * 16, c0 = itc * 32;
  const int r_begin = chunk * a.chunk_rows;
  const int r_end = min(a.rows, r_begin + a.chunk_rows);
  const bool relu = a.activation == AGCN_ACT_RELU;

  const float* A;  // [rows, lda], indexed [row][f]
  int lda;
  if (p == 0 || p == K + 1) { A = a.X; lda = a.ldx; }
  else if (p < K) { A = a.sv_X + (int64_t)(p - 1) * a.rows * Fin; lda = Fin; }
  else { A = nullptr; lda = 0; }
  const bool bias_prob = (p == K);
  const bool dm_prob = (p == K + 1);

  f32x4 acc[2];
  acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
  acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int f = f0 + r;
  for (int rr = r_begin; rr < r_end; rr += 4) {
    const int row = rr + q;
    const bool rv = row < r_end;
    float av = 0.f;
    if (rv && f < Fa) av = bias_prob ? 1.0f : A[(int64_t)row * lda + f];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + 16 * j + r;
      float bv = 0.f;
      if (rv && c < Fb) {
        if (dm_prob) {
          bv = a.dZ[(int64_t)row * Fin + c];
        } else {
          const float g = a.dY[(int64_t)row * a.ldy + c];
          bv = (!relu || a.Y[(int64_t)row * a.ldy + c] > 0.f) ? g : 0.f;
        }
      }
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j], 0, 0, 0);
    }
  }
  float* out = a.part + (int64_t)chunk * a.part_stride + out_base;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ff = f0 + q * 4 + i, c = c0 + 16 * j + r;
      if (ff < Fa && c < Fb) {
        // dweight rows are interleaved f*K + k (graphconv.py:243-244)
        const int64_t o = (p < K) ? ((int64_t)(ff * K + p) * Fout + c) : ((int64_t)ff * Fb + c);
        out[o] = acc[j][i];
      }
    }
}

__global__ __launch_bounds__(256) void param_grad_reduce_kernel(ParamGradArgs a) {
  const int64_t nW = (int64_t)a.K * a.Fin * a.Fout, nB = a.Fout, nM = a.paper ? (int64_t)a.Fin * a.Fin : 0;
  const int64_t total = nW + nB + nM;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) {
    float s = 0.f;
    for (int c = 0; c < a.n_chunks; ++c) s += a.part[(int64_t)c * a.part_stride + e];
    if (e < nW) a.dweight[e] = s;
    else if (e < nW + nB) a.dbias[e - nW] = s;
    else a.dM_L[e - nW - nB] = s;
  }
  // scalars: one wave each, fixed order
  if (blockIdx.x == 0) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 2) {
      float* dst = wave == 0 ? a.dalpha : a.dbeta;
      if (dst != nullptr) {
        float s = 0.f;
        for (int b = lane; b < a.B; b += 64) s += a.pg[2 * b + wave];
        s = wave_sum(s);
        if (lane == 0) dst[0] = s;
      }
    }
  }
}

int launch_param_grads(const ParamGradArgs& a, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int K = a.K;
  const int tfW = (a.Fin + 15) / 16, tcW = (a.Fout + 31) / 32, tcM = (a.Fin + 31) / 32;
  int items = tfW * tcW;
  if (a.paper) items = items > tfW * tcM ? items : tfW * tcM;
  items *= a.n_chunks;
  const int nprob = K + 1 + (a.paper ? 1 : 0);
  hipLaunchKernelGGL(param_grad_partial_kernel, dim3((items + 3) / 4, nprob), dim3(256), 0, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "param_grad_partial launch: %s", hipGetErrorString(e));
  const int64_t total = (int64_t)K * a.Fin * a.Fout + a.Fout + (a.paper ? (int64_t)a.Fin * a.Fin : 0);
  hipLaunchKernelGGL(param_grad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "param_grad_reduce launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

}  // namespace agcn
