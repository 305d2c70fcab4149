// Parameter gradients that reduce over every node of the batch:
//   dweight[j, c] = sum_r C[r, j] Gp[r, c],  C[r, f*K + k] = X_k[r, f]   (graphconv.py:243-247 under autodiff)
//   dbias[c]      = sum_r Gp[r, c]
//   dM_L[f, c]    = sum_r X[r, f] dZ[r, c]                                (paper mode only)
//   dalpha, dbeta = sum over graphs of the per-graph partials
// with Gp = dY . act'(Y) (written by the RG_FILTER_BWD launch).  Two launches: per-chunk partial
// products on the matrix cores (A^T B with the row index as the MFMA k dimension), then a
// fixed-order reduction -- no atomics, so the result is reproducible bit for bit.
#include "agcn_device.h"
#include "sgc_kernels.h"

namespace agcn {

namespace {
constexpr int PG_UNROLL = 4;   // MFMA k-steps (of 4 rows) whose operand loads are issued together
}

// grid: (items / 4, problems).  problem 0: dweight; 1: dbias; 2: dM_L.
// One wave per (row chunk, 16 x 32 output strip).
__global__ __launch_bounds__(256) void param_grad_partial_kernel(ParamGradArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int p = blockIdx.y;
  const int K = a.K, Fin = a.Fin, Fout = a.Fout;
  int Fa, Fb;       // output is [Fa x Fb]
  int64_t out_base; // offset inside one chunk's partial block
  if (p == 0) { Fa = Fin * K; Fb = Fout; out_base = 0; }
  else if (p == 1) { Fa = 1; Fb = Fout; out_base = (int64_t)K * Fin * Fout; }
  else { Fa = Fin; Fb = Fin; out_base = (int64_t)K * Fin * Fout + Fout; }
  const int tf = (Fa + 15) >> 4, tc = (Fb + 31) >> 5;
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (blockIdx.x % 8 share one, the
  // grid's x extent is a multiple of 8), and every output tile of one ROW CHUNK is computed on the same
  // XCD, so the chunk's rows of X / Gp / dZ are fetched into ONE L2 instead of up to eight.
  const int W = a.wgs_per_chunk;                       // workgroups (of 4 wave items) per chunk
  const int g = blockIdx.x;
  const int chunk = (g / (8 * W)) * 8 + (g & 7);
  const int rem = ((g >> 3) % W) * 4 + wave;           // tile index inside the chunk
  if (chunk >= a.n_chunks || rem >= tf * tc) return;
  const int itc = rem / tf, itf = rem - itc * tf;
  const int f0 = itf * 16, c0 = itc * 32;
  const int r_begin = chunk * a.chunk_rows;
  const int r_end = min(a.rows, r_begin + a.chunk_rows);
  const int last_row = a.rows - 1;

  // A operand: column j of C / X / ones.  Addresses are clamped so that loads are unconditional
  // (they can all be issued before the first MFMA); out-of-range lanes are zeroed afterwards.
  const int j = min(f0 + r, Fa - 1);
  const bool jv = (f0 + r) < Fa;
  const float* Ap;
  int64_t lda;
  if (p == 0) {
    const int f = j / K, kx = j - f * K;
    if (kx == 0) { Ap = a.X + f; lda = a.ldx; }
    else { Ap = a.sv_X + (int64_t)(kx - 1) * a.rows * Fin + f; lda = Fin; }
  } else if (p == 2) { Ap = a.X + j; lda = a.ldx; }
  else { Ap = a.X; lda = 0; }
  const float* Bp = (p == 2) ? a.dZ : a.Gp;
  const int ldb = (p == 2) ? Fin : Fout;
  const int cA = min(c0 + r, Fb - 1), cB = min(c0 + 16 + r, Fb - 1);
  const bool cAv = (c0 + r) < Fb, cBv = (c0 + 16 + r) < Fb;

  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  for (int rr = r_begin; rr < r_end; rr += 4 * PG_UNROLL) {
    float av[PG_UNROLL], b0[PG_UNROLL], b1[PG_UNROLL];
#pragma unroll
    for (int u = 0; u < PG_UNROLL; ++u) {
      const int row = min(rr + 4 * u + q, last_row);
      av[u] = (p == 1) ? 1.0f : Ap[(int64_t)row * lda];
      b0[u] = Bp[(int64_t)row * ldb + cA];
      b1[u] = Bp[(int64_t)row * ldb + cB];
    }
#pragma unroll
    for (int u = 0; u < PG_UNROLL; ++u) {
      const bool rv = (rr + 4 * u + q) < r_end;
      const float x = (rv && jv) ? av[u] : 0.f;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, (rv && cAv) ? b0[u] : 0.f, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, (rv && cBv) ? b1[u] : 0.f, acc1, 0, 0, 0);
    }
  }
  float* out = a.part + (int64_t)chunk * a.part_stride + out_base;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ff = f0 + q * 4 + i;
    if (ff < Fa) {
      if (c0 + r < Fb) out[(int64_t)ff * Fb + c0 + r] = acc0[i];
      if (c0 + 16 + r < Fb) out[(int64_t)ff * Fb + c0 + 16 + r] = acc1[i];
    }
  }
}

__global__ __launch_bounds__(256) void param_grad_reduce_kernel(ParamGradArgs a) {
  const int64_t nW = (int64_t)a.K * a.Fin * a.Fout, nB = a.Fout, nM = a.paper ? (int64_t)a.Fin * a.Fin : 0;
  const int64_t total = nW + nB + nM;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) {
    float s = 0.f;
#pragma unroll 8
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
      if (dst != nullptr && a.pg != nullptr) {
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
  const int tfW = (a.Fin * a.K + 15) / 16, tcW = (a.Fout + 31) / 32;
  const int tfM = (a.Fin + 15) / 16, tcM = (a.Fin + 31) / 32;
  int items = tfW * tcW;
  if (a.paper && tfM * tcM > items) items = tfM * tcM;
  ParamGradArgs aa = a;
  aa.wgs_per_chunk = (items + 3) / 4;
  const int gx = 8 * aa.wgs_per_chunk * ((a.n_chunks + 7) / 8);
  const int nprob = a.paper ? 3 : 2;
  prof_begin(PK_PARAM_PARTIAL, stream);
  hipLaunchKernelGGL(param_grad_partial_kernel, dim3(gx, nprob), dim3(256), 0, s, aa);
  prof_end(stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "param_grad_partial launch: %s", hipGetErrorString(e));
  const int64_t total = (int64_t)a.K * a.Fin * a.Fout + a.Fout + (a.paper ? (int64_t)a.Fin * a.Fin : 0);
  prof_begin(PK_PARAM_REDUCE, stream);
  hipLaunchKernelGGL(param_grad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  prof_end(stream);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "param_grad_reduce launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

}  // namespace agcn
