// Forward of SGC_LL / SGC_LL_Reslap for a whole ragged batch: one workgroup per graph,
// every stage of the reference's per-graph pipeline fused into one launch.
//
// Stage map (reference: models/layers/graphconv.py, twin graphconv_reslap.py):
//   P1  Z = X . M_L                               graphconv.py:164        (MFMA f32)
//   P2  S_ij = exp(-||Z_i - Z_j||), S_ii = 0      graphconv.py:168-180    (LDS tile, direct differences)
//       d_j = sum_i S_ij + eps, s = d^-1/2        graphconv.py:195-197
//   P3  R = I (reference) | I - s S s (paper)     graphconv.py:198-200
//       clip, leaky-ReLU(alpha), + L_orig (+ beta L_prev, second clip + leaky for Reslap)
//                                                 graphconv.py:212-216 / graphconv_reslap.py:185-195
//   P5  X_k = L X_{k-1} (Chebyshev recurrence)    graphconv.py:222-236    (MFMA f32)
//   P6  Y = act([X_0|X_1|..] . W + b), pad rows 0 graphconv.py:238-251,117-123 (MFMA f32)
//
// The n x n tile and the n x Fin image of Z live in LDS (LDS_TILE = true, 256 threads) or,
// for graphs too large for 160 KiB, in global scratch (LDS_TILE = false, 1024 threads).
#include "agcn_device.h"
#include "sgc_kernels.h"

#include <float.h>

namespace agcn {

template <bool LDS_TILE>
__global__ __launch_bounds__(LDS_TILE ? 256 : 1024) void sgc_forward_kernel(SgcArgs a, int first) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int b = a.order[first + blockIdx.x];
  const int n = a.n[b];
  const int row0 = a.row_off[b];
  const int nrows_owned = a.row_off[b + 1] - row0;
  const int64_t toff = a.tile_off[b];
  const int Fin = a.Fin, Fout = a.Fout, K = a.K;
  const int ldt = n | 1, ldb = Fin | 1;
  const int ldx = a.ldx, ldy = a.ldy;
  const bool reslap = a.variant == AGCN_SGC_LL_RESLAP;
  const bool paper = a.mode == AGCN_MODE_PAPER;

  float* red = smem;         // 32 floats
  float* s_vec = smem + 32;  // n floats
  float* tile;
  float* buf;
  if (LDS_TILE) {
    tile = s_vec + round4(n);
    buf = tile + round4(n * ldt);
  } else {
    tile = a.ws_tile + a.wtile_off[b];
    buf = a.ws_buf + (int64_t)row0 * ldb;
  }
  const float* Xg = a.X + (int64_t)row0 * ldx;
  const FastDiv divn((uint32_t)n);
  const int nn = n * n;

  // ---- P1: Z = X . M_L -> buf ------------------------------------------------------------
  {
    const float* M = a.M_L;
    wg_gemm<2>(
        n, Fin, Fin,
        [&](int i, int k) { return (i < n && k < Fin) ? Xg[(int64_t)i * ldx + k] : 0.f; },
        [&](int k, int c) { return (k < Fin && c < Fin) ? M[k * Fin + c] : 0.f; },
        [&](int i, int c, float v) { if (i < n && c < Fin) buf[i * ldb + c] = v; });
  }
  __syncthreads();

  // ---- P2: similarity tile, 2x2 register blocks over the upper triangle ---------------------
  {
    const int nb2 = (n + 1) >> 1;
    const TriFold tri(nb2);
    const FastDiv dW((uint32_t)tri.W);
    for (int t = tid; t < tri.count(); t += nthr) {
      int bi, bj;
      if (!tri.get(t, dW, bi, bj)) continue;
      const int i0 = 2 * bi, j0 = 2 * bj;
      const int i1 = min(i0 + 1, n - 1), j1 = min(j0 + 1, n - 1);
      const float* zi0 = buf + i0 * ldb;
      const float* zi1 = buf + i1 * ldb;
      const float* zj0 = buf + j0 * ldb;
      const float* zj1 = buf + j1 * ldb;
      float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
      for (int f = 0; f < Fin; ++f) {
        const float u0 = zi0[f], u1 = zi1[f], v0 = zj0[f], v1 = zj1[f];
        float d;
        d = u0 - v0; a00 = fmaf(d, d, a00);
        d = u0 - v1; a01 = fmaf(d, d, a01);
        d = u1 - v0; a10 = fmaf(d, d, a10);
        d = u1 - v1; a11 = fmaf(d, d, a11);
      }
      const float acc[2][2] = {{a00, a01}, {a10, a11}};
#pragma unroll
      for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj) {
          const int i = i0 + di, j = j0 + dj;
          if (i < n && j < n) {
            // S_ii = 0 (graphconv.py:171-173); exp(-1 * norm) otherwise (:176-177)
            const float s = (i == j) ? 0.f : expf(-sqrtf(acc[di][dj]));
            tile[i * ldt + j] = s;
            tile[j * ldt + i] = s;  // bit-symmetric by construction
          }
        }
    }
  }
  __syncthreads();

  // ---- P2b: degree (column sums in row order, like numpy's sum(axis=0)), s = d^-1/2 ----------
  for (int j = tid; j < n; j += nthr) {
    float d = 0.f;
    for (int i = 0; i < n; ++i) d += tile[i * ldt + j];
    d += FLT_TRUE_MIN;  // np.spacing(np.array(0, float32)), graphconv.py:196
    const float s = 1.0f / sqrtf(d);
    s_vec[j] = s;
    a.sv_s[row0 + j] = s;
  }
  // the (<= 3) alignment floats between consecutive tiles are kept at 0
  {
    const int npadt = (int)(a.tile_off[b + 1] - toff) - nn;
    if (tid < npadt) {
      a.res_W[toff + nn + tid] = 0.f;
      a.res_L[toff + nn + tid] = 0.f;
      if (a.L_sum) a.L_sum[toff + nn + tid] = 0.f;
    }
  }
  // res_W output (graphconv.py:215)
  for (int idx = tid; idx < nn; idx += nthr) {
    const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
    a.res_W[toff + idx] = tile[i * ldt + j];
  }
  __syncthreads();

  // ---- P3: residual Laplacian, clip, leaky-ReLU, combine -----------------------------------
  const float alpha = a.alpha[0];
  float ssR;
  {
    float part = 0.f;
    for (int idx = tid; idx < nn; idx += nthr) {
      const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
      float R = (i == j) ? 1.f : 0.f;  // reference: I - D*W*D elementwise == I (graphconv.py:198-200)
      if (paper) R = R - (s_vec[i] * tile[i * ldt + j]) * s_vec[j];
      tile[i * ldt + j] = R;
      part += R * R;
    }
    ssR = block_sum(part, red);
  }
  // clip_by_average_norm (SGC_LL, graphconv.py:212) / clip_by_norm (Reslap, graphconv_reslap.py:185)
  const float scR = clip_scale(ssR, reslap ? 1.0f : (float)nn);
  const bool l0_dense = a.L0_dense != nullptr;
  const bool l0_csr = a.L0_rowptr != nullptr;
  const float* L0d = nullptr;
  int l0ld = n;
  if (l0_dense) {
    if (a.L0_ld > 0) { L0d = a.L0_dense + (int64_t)b * a.L0_ld * a.L0_ld; l0ld = a.L0_ld; }
    else { L0d = a.L0_dense + toff; }
  }
  const bool has_prev = reslap && a.L_prev != nullptr;
  const float beta = has_prev ? a.beta[0] : 0.f;
  // pass A: R2 = lrelu(R * scR) -> res_L; tile = R2 (+ dense L0)
  for (int idx = tid; idx < nn; idx += nthr) {
    const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
    const float R1 = tile[i * ldt + j] * scR;
    const float R2 = lrelu(R1, alpha);
    a.res_L[toff + idx] = R2;
    float v = R2;
    if (l0_dense) v = v + L0d[(int64_t)i * l0ld + j];
    tile[i * ldt + j] = v;
  }
  if (l0_csr) {
    __syncthreads();
    const int32_t* rp = a.L0_rowptr + row0;
    for (int i = tid; i < n; i += nthr) {
      const int p0 = rp[i], p1 = rp[i + 1];
      for (int p = p0; p < p1; ++p) tile[i * ldt + a.L0_col[p]] += a.L0_val[p];
    }
  }
  float ssL = 0.f, scL = 1.f;
  if (reslap) {
    __syncthreads();
    // pass B: + beta * L_prev (graphconv_reslap.py:190); keep L_pre; second clip + leaky (:194-195)
    float part = 0.f;
    for (int idx = tid; idx < nn; idx += nthr) {
      const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
      float v = tile[i * ldt + j];
      if (has_prev) v = v + a.L_prev[toff + idx] * beta;
      tile[i * ldt + j] = v;
      a.sv_Lpre[toff + idx] = v;
      part += v * v;
    }
    ssL = block_sum(part, red);
    scL = clip_scale(ssL, 1.0f);
    for (int idx = tid; idx < nn; idx += nthr) {
      const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
      const float La = lrelu(tile[i * ldt + j] * scL, alpha);
      tile[i * ldt + j] = La;
      a.L_sum[toff + idx] = La;
    }
  } else if (a.L_sum != nullptr) {
    __syncthreads();
    for (int idx = tid; idx < nn; idx += nthr) {
      const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
      a.L_sum[toff + idx] = tile[i * ldt + j];
    }
  }
  if (tid == 0) {
    float* sc = a.sv_scal + 4 * b;
    sc[0] = ssR; sc[1] = scR; sc[2] = ssL; sc[3] = scL;
  }
  __syncthreads();

  // ---- P5: Chebyshev terms X_k = L X_{k-1} (k = 1), 2 L X_{k-1} - X_{k-2} (k >= 2) -----------
  const int64_t svx_stride = (int64_t)a.rows * Fin;
  for (int k = 1; k < K; ++k) {
    const float* Xp = (k == 1) ? Xg : a.sv_X + (k - 2) * svx_stride + (int64_t)row0 * Fin;
    const int ldp = (k == 1) ? ldx : Fin;
    const float* Xpp = nullptr;
    int ldpp = Fin;
    if (k == 2) { Xpp = Xg; ldpp = ldx; }
    else if (k > 2) { Xpp = a.sv_X + (k - 3) * svx_stride + (int64_t)row0 * Fin; }
    float* Xo = a.sv_X + (k - 1) * svx_stride + (int64_t)row0 * Fin;
    wg_gemm<2>(
        n, Fin, n,
        [&](int i, int kk) { return (i < n && kk < n) ? tile[i * ldt + kk] : 0.f; },
        [&](int kk, int c) { return (kk < n && c < Fin) ? Xp[(int64_t)kk * ldp + c] : 0.f; },
        [&](int i, int c, float v) {
          if (i < n && c < Fin) {
            if (Xpp) v = 2.0f * v - Xpp[(int64_t)i * ldpp + c];
            Xo[(int64_t)i * Fin + c] = v;
          }
        });
    __syncthreads();  // X_k (global) is read back by this workgroup
  }

  // ---- P6: Y = act(sum_k X_k . W[f*K + k, :] + b) ------------------------------------------
  {
    const float* Wt = a.weight;
    const float* bias = a.bias;
    const bool relu = a.activation == AGCN_ACT_RELU;
    float* Yg = a.Y + (int64_t)row0 * ldy;
    wg_tiles<2>(
        n, Fout,
        [&](f32x4 (&acc)[2], int r0, int c0) {
          for (int k = 0; k < K; ++k) {
            const float* Xk = (k == 0) ? Xg : a.sv_X + (k - 1) * svx_stride + (int64_t)row0 * Fin;
            const int ldk = (k == 0) ? ldx : Fin;
            mfma_acc<2>(
                acc, r0, c0, Fin,
                [&](int i, int f) { return (i < n && f < Fin) ? Xk[(int64_t)i * ldk + f] : 0.f; },
                [&](int f, int c) { return (f < Fin && c < Fout) ? Wt[(int64_t)(f * K + k) * Fout + c] : 0.f; });
          }
        },
        [&](int i, int c, float v) {
          if (i < n && c < Fout) {
            v = v + bias[c];
            if (relu) v = fmaxf(v, 0.f);
            Yg[(int64_t)i * ldy + c] = v;
          }
        });
    // pad rows are exactly 0 (tf.pad, graphconv.py:249-250; relu(0) = 0)
    const int npad = (nrows_owned - n) * Fout;
    const FastDiv divF((uint32_t)Fout);
    for (int idx = tid; idx < npad; idx += nthr) {
      const int i = (int)divF.div((uint32_t)idx), c = idx - i * Fout;
      Yg[(int64_t)(n + i) * ldy + c] = 0.f;
    }
  }
}

size_t sgc_lds_bytes(int32_t n, int32_t Fin) {
  auto r4 = [](int64_t v) { return (v + 3) & ~int64_t(3); };
  const int64_t ldt = n | 1, ldb = Fin | 1;
  return (size_t)(4 * (32 + r4(n) + r4((int64_t)n * ldt) + r4((int64_t)n * ldb)));
}

int launch_sgc_forward(const SgcArgs& a, int32_t first, int32_t count, bool lds_tile, size_t lds_bytes,
                       void* stream) {
  if (count <= 0) return AGCN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e;
  if (lds_tile) {
    if (lds_bytes > 64 * 1024) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sgc_forward_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
      if (e != hipSuccess) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(fwd): %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(sgc_forward_kernel<true>, dim3(count), dim3(256), lds_bytes, s, a, first);
  } else {
    hipLaunchKernelGGL(sgc_forward_kernel<false>, dim3(count), dim3(1024), lds_bytes, s, a, first);
  }
  e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "sgc_forward launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

}  // namespace agcn
