// Graph-local part of the SGC_LL / SGC_LL_Reslap forward: one workgroup per graph (largest
// first), everything that couples the nodes of ONE graph.  The feature contractions that do
// not (Z = X.M_L before, Y = [X_k].W + b after) run as packed-row GEMMs (rows_gemm.hip).
//
// Stage map (reference: models/layers/graphconv.py, twin graphconv_reslap.py):
//   P2  S_ij = exp(-||Z_i - Z_j||), S_ii = 0      graphconv.py:168-180   LDS image of Z^T, 2x2
//                                                 register blocks over the upper triangle
//       d_j = sum_i S_ij + eps, s = d^-1/2        graphconv.py:195-197   fixed-order column sums
//   P3  R = I (reference) | I - s S s (paper)     graphconv.py:198-200
//       clip, leaky-ReLU(alpha), + L_orig (+ beta L_prev, second clip + leaky for Reslap)
//                                                 graphconv.py:212-216 / graphconv_reslap.py:185-195
//   P5  X_k = L X_{k-1} (Chebyshev recurrence)    graphconv.py:222-236   MFMA f32, both operands in LDS
//
// The n x n tile and the feature images live in LDS (LDS_TILE = true) or, for graphs whose
// image exceeds 160 KiB, in global scratch (LDS_TILE = false, 1024 threads): same code.
#include "agcn_device.h"
#include "sgc_kernels.h"

#include <float.h>

namespace agcn {

template <bool LDS_TILE>
__global__ __launch_bounds__(1024) void sgc_forward_kernel(SgcArgs a, int first) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int b = a.order[first + blockIdx.x];
  const int n = a.n[b];
  const int row0 = a.row_off[b];
  const int64_t toff = a.tile_off[b];
  const int Fin = a.Fin, K = a.K;
  const int ldt = ld_frag(n);     // tile and Z^T image
  const int ldbx = ld_b(Fin);     // row-major X image (B operand)
  const int ldx = a.ldx;
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
    buf = a.ws_buf + ((int64_t)row0 + 16 * (int64_t)b) * scratch_fs(Fin);
  }
  const float* Xg = a.X + (int64_t)row0 * ldx;
  const FastDiv divn((uint32_t)n);
  const FastDiv divF((uint32_t)Fin);
  const int nn = n * n;

  AGCN_STAMP(1, 0);
  // ---- round 0: every global load whose address is known now is issued here, so that the kernel
  // pays ONE memory latency up front instead of one per phase: Z rows (for the Z^T image), the X rows
  // of the first Chebyshev product (kept in registers until the image buffer is free), the CSR row
  // pointers of L_orig.
  constexpr int UX = 8;
  const int n4 = round4(n);
  const int ximg = n4 * Fin;                    // elements of the X image (rows n..n4 are zero)
  const bool x_pref = (K > 1) && (ximg <= UX * nthr);
  float xreg[UX];
  int csr_p0 = 0, csr_p1 = 0;
  {
    const float* Zg = a.sv_Z + (int64_t)row0 * Fin;
    const int ztot = n * Fin;
    float zreg[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) zreg[u] = Zg[min(tid + u * nthr, ztot - 1)];
    if (x_pref) {
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int idx = min(tid + u * nthr, ximg - 1);
        const int i = (int)divF.div((uint32_t)idx), f = idx - i * Fin;
        xreg[u] = Xg[(int64_t)min(i, n - 1) * ldx + f];
      }
    }
    if (a.L0_rowptr != nullptr && tid < n) {
      csr_p0 = a.L0_rowptr[row0 + tid];
      csr_p1 = a.L0_rowptr[row0 + tid + 1];
    }
    // Z^T image: Zt[f][i] = Z[row0 + i][f]
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + u * nthr;
      if (idx < ztot) {
        const int i = (int)divF.div((uint32_t)idx), f = idx - i * Fin;
        buf[f * ldt + i] = zreg[u];
      }
    }
    for (int base = 8 * nthr; base < ztot; base += 8 * nthr) {      // graphs with n*Fin > 8*nthr
#pragma unroll
      for (int u = 0; u < 8; ++u) zreg[u] = Zg[min(base + tid + u * nthr, ztot - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + tid + u * nthr;
        if (idx < ztot) {
          const int i = (int)divF.div((uint32_t)idx), f = idx - i * Fin;
          buf[f * ldt + i] = zreg[u];
        }
      }
    }
  }
  // CSR entries of this thread's row (first 4; longer rows fall back to a loop in P3)
  int csr_c[4] = {0, 0, 0, 0};
  float csr_v[4] = {0.f, 0.f, 0.f, 0.f};
  if (a.L0_rowptr != nullptr && tid < n && csr_p1 > csr_p0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = min(csr_p0 + u, csr_p1 - 1);
      csr_c[u] = a.L0_col[pp];
      csr_v[u] = a.L0_val[pp];
    }
  }
  __syncthreads();

  AGCN_STAMP(1, 1);
  // ---- P2: similarity tile, 2x2 register blocks over the upper triangle ---------------------------
  {
    const int nb2 = (n + 1) >> 1;
    const TriFold tri(nb2);
    const FastDiv dW((uint32_t)tri.W);
    // small graphs have fewer 2x2 blocks than threads: split the feature loop of a block over
    // `fs` adjacent lanes (fs = 1, 2, 4, 8 or 16) and combine the partial sums with shuffles
    int fs_log = 0;
    while (fs_log < 4 && (tri.count() << (fs_log + 1)) <= nthr) ++fs_log;
    const int fs = 1 << fs_log;
    const int part = tid & (fs - 1);
    for (int tb = 0; tb < tri.count() * fs; tb += nthr) {   // uniform trip count: shuffles below
      const int t = (tb + tid) >> fs_log;
      int bi = 0, bj = 0;
      const bool live = t < tri.count() && tri.get(t, dW, bi, bj);
      const int i0 = 2 * bi, j0 = 2 * bj;
      const float* zi = buf + i0;
      const float* zj = buf + j0;
      float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;
      if (live) {
#pragma unroll 5
        for (int f = part; f < Fin; f += fs) {
          const float2 u = *reinterpret_cast<const float2*>(zi + f * ldt);
          const float2 v = *reinterpret_cast<const float2*>(zj + f * ldt);
          float d;
          d = u.x - v.x; a00 = fmaf(d, d, a00);
          d = u.x - v.y; a01 = fmaf(d, d, a01);
          d = u.y - v.x; a10 = fmaf(d, d, a10);
          d = u.y - v.y; a11 = fmaf(d, d, a11);
        }
      }
      for (int o = 1; o < fs; o <<= 1) {
        a00 += __shfl_xor(a00, o, 64);
        a01 += __shfl_xor(a01, o, 64);
        a10 += __shfl_xor(a10, o, 64);
        a11 += __shfl_xor(a11, o, 64);
      }
      if (live && part == 0) {
        const float acc[2][2] = {{a00, a01}, {a10, a11}};
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
          for (int dj = 0; dj < 2; ++dj) {
            const int i = i0 + di, j = j0 + dj;
            if (i < n && j < n) {
              // S_ii = 0 (graphconv.py:171-173); exp(-1 * norm) otherwise (:176-177)
              const float sv = (i == j) ? 0.f : expf(-sqrtf(acc[di][dj]));
              tile[i * ldt + j] = sv;
              tile[j * ldt + i] = sv;  // bit-symmetric by construction
            }
          }
      }
    }
  }
  __syncthreads();

  AGCN_STAMP(1, 2);
  // ---- P2b: degree (column sums in row order, like numpy's sum(axis=0)), s = d^-1/2 ----------------
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

  AGCN_STAMP(1, 3);
  // ---- P3: residual Laplacian, clip, leaky-ReLU, combine ---------------------------------------------
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
    if (tid < n) {     // one thread per row: entries fetched in round 0, the rest (rare) in a loop
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (csr_p0 + u < csr_p1) tile[tid * ldt + csr_c[u]] += csr_v[u];
      for (int p = csr_p0 + 4; p < csr_p1; ++p) tile[tid * ldt + a.L0_col[p]] += a.L0_val[p];
    }
    for (int i = tid + nthr; i < n; i += nthr) {   // n > nthr cannot happen in LDS mode; global-tile mode only
      const int p0 = a.L0_rowptr[row0 + i], p1 = a.L0_rowptr[row0 + i + 1];
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

  AGCN_STAMP(1, 4);
  // ---- P5: Chebyshev terms X_k = L X_{k-1} (k = 1), 2 L X_{k-1} - X_{k-2} (k >= 2) ---------------------
  const int64_t svx_stride = (int64_t)a.rows * Fin;
  for (int k = 1; k < K; ++k) {
    const float* Xp = (k == 1) ? Xg : a.sv_X + (k - 2) * svx_stride + (int64_t)row0 * Fin;
    const int ldp = (k == 1) ? ldx : Fin;
    const float* Xpp = nullptr;
    int ldpp = Fin;
    if (k == 2) { Xpp = Xg; ldpp = ldx; }
    else if (k > 2) { Xpp = a.sv_X + (k - 3) * svx_stride + (int64_t)row0 * Fin; }
    float* Xo = a.sv_X + (k - 1) * svx_stride + (int64_t)row0 * Fin;
    __syncthreads();   // tile complete / previous X_k visible; the image below overwrites Z^T
    // row-major image of X_{k-1}: Xs[node][f], rows n..n4 zeroed (MFMA k padding)
    if (k == 1 && x_pref) {
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int idx = tid + u * nthr;
        if (idx < ximg) {
          const int i = (int)divF.div((uint32_t)idx), f = idx - i * Fin;
          buf[i * ldbx + f] = (i < n) ? xreg[u] : 0.f;
        }
      }
    } else {
      bulk_for<8>(
          n4 * Fin,
          [&](int idx) {
            const int i = (int)divF.div((uint32_t)idx), f = idx - i * Fin;
            return Xp[(int64_t)min(i, n - 1) * ldp + f];
          },
          [&](int idx, float v) {
            const int i = (int)divF.div((uint32_t)idx), f = idx - i * Fin;
            buf[i * ldbx + f] = (i < n) ? v : 0.f;
          });
    }
    __syncthreads();
    wg_gemm<2>(
        n, Fin, n,
        [&](int i, int kk) { return (i < n && kk < n) ? tile[i * ldt + kk] : 0.f; },
        [&](int kk, int c) { return (c < Fin) ? buf[kk * ldbx + c] : 0.f; },
        [&](int i, int c, float v) {
          if (i < n && c < Fin) {
            if (Xpp) v = 2.0f * v - Xpp[(int64_t)i * ldpp + c];
            Xo[(int64_t)i * Fin + c] = v;
          }
        });
  }
  AGCN_STAMP(1, 5);
}

namespace {
inline int64_t r4(int64_t v) { return (v + 3) & ~int64_t(3); }
inline int64_t imax(int64_t x, int64_t y) { return x > y ? x : y; }
}  // namespace

size_t sgc_lds_bytes(int32_t n, int32_t Fin) {
  const int64_t ldt = ld_frag(n), ldbx = ld_b(Fin);
  const int64_t buf = imax((int64_t)Fin * ldt, r4(n) * ldbx);
  return (size_t)(4 * (32 + r4(n) + r4((int64_t)n * ldt) + r4(buf)));
}

size_t graph_scratch_floats(int32_t rows, int32_t B, int32_t Fin) {
  return (size_t)graph_scratch_floats_dev(rows, B, Fin);
}

int launch_sgc_forward(const SgcArgs& a, int32_t first, int32_t count, bool lds_tile, size_t lds_bytes,
                       void* stream) {
  if (count <= 0) return AGCN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e;
  prof_begin(PK_GRAPH_FWD, stream);
  if (lds_tile) {
    if (lds_bytes > 64 * 1024) {
      e = static_cast<hipError_t>(ensure_dynamic_lds(reinterpret_cast<const void*>(&sgc_forward_kernel<true>), lds_bytes));
      if (e != hipSuccess) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(fwd): %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(sgc_forward_kernel<true>, dim3(count), dim3(graph_threads(count)), lds_bytes, s, a, first);
  } else {
    hipLaunchKernelGGL(sgc_forward_kernel<false>, dim3(count), dim3(1024), lds_bytes, s, a, first);
  }
  prof_end(stream);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "sgc_forward launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

}  // namespace agcn
