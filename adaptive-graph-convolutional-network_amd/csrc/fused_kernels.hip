// Fused per-graph kernels: the whole SGC_LL / SGC_LL_Reslap layer (K = 2) forward in ONE launch and backward in
// ONE launch, one 1024-thread workgroup per graph, the graph's working set staged in LDS once.
//
// At Tox21 sizes a launch costs more than the arithmetic inside it (each kernel sits on a ~5-9 us floor of launch +
// dependent memory round trips), so the split pipeline (rows_gemm -> graph kernel -> rows_gemm) is latency-bound by
// its kernel COUNT.  When the working set of a graph fits the 160 KiB of LDS all stages run back to back:
//
//   F0  bulk loads: X rows, CSR rows of L_orig (+ M_L, W images in the LDS-image flavour)            (one round)
//   F1  Z = X . M_L                      graphconv.py:164        MFMA -> Z^T image + reserve
//   F2  S = exp(-||Z_i - Z_j||), degree  graphconv.py:168-197    2x2 blocks, feature-split lanes
//   F3  R = I | I - s S s, (Reslap: clip_by_norm,) leaky-ReLU, + L_orig (CSR scatter or dense rows),
//       (Reslap: + beta L_prev, second clip + leaky-ReLU)        graphconv.py:198-216, graphconv_reslap.py:184-198
//   F4  X_1 = L X_0                      graphconv.py:231        MFMA, operands in LDS
//   F5  Y = act(X_0 W_0 + X_1 W_1 + b)   graphconv.py:238-251    MFMA
//
// Template flags: GB (forward) / GBW, GBM (backward) -- the parameter (W, M_L) is read straight from global memory
// (L2-resident) inside the MFMA k-loops instead of being staged as an LDS image: no image => wide layers (Fin = 128)
// and graphs up to 96 nodes fit, and for 16-byte aligned rows it is also the faster choice; RESLAP -- the
// SGC_LL_Reslap combine and its adjoints.  agcn_api.hip (fused_split) picks the flavour per batch.
//
// Same arithmetic (and the same fixed reduction orders) as sgc_forward.hip / sgc_backward.hip + rows_gemm.hip.
#include "agcn_device.h"
#include "sgc_kernels.h"

#include <float.h>
#include <cstdlib>

namespace agcn {

namespace {
struct FusedDims {
  int n4, F4, F16, ldt, ldxa, ldm, ldw, KF4;   // n4 / F4 are the 16-padded node / feature counts
  int64_t o_svec, o_tile, o_ximg, o_zx, o_mimg, o_wimg, total;   // float offsets
};

// One 16 x 16 output tile out of LDS images, four MFMA k-steps per iteration with the eight operand
// reads issued first.  kdim must be a multiple of 16 and the images zero-padded up to it.
//   A image: [row][k] (ld lda);  B image row-major: [k][col] (ld ldb)
__device__ __forceinline__ void tile_gemm_rm(f32x4& acc, const float* A, int lda, int row0, const float* B,
                                             int ldb, int col0, int kdim) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const float* Ap = A + (row0 + r) * lda + q;
  const float* Bp = B + q * ldb + col0 + r;
  for (int k0 = 0; k0 < kdim; k0 += 16) {
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = Ap[k0 + 4 * u];
      bv[u] = Bp[(k0 + 4 * u) * ldb];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  }
}

// B given as a transposed image Bt[col][k] (ld ldb): B[k][col] = Bt[col][k]
__device__ __forceinline__ void tile_gemm_tb(f32x4& acc, const float* A, int lda, int row0, const float* Bt,
                                             int ldb, int col0, int kdim) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const float* Ap = A + (row0 + r) * lda + q;
  const float* Bp = Bt + (col0 + r) * ldb + q;
#pragma unroll 1
  for (int k0 = 0; k0 < kdim; k0 += 16) {
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = Ap[k0 + 4 * u];
      bv[u] = Bp[k0 + 4 * u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  }
}

// A given as a transposed image At[k][row] (ld lda): A[row][k] = At[k][row]
__device__ __forceinline__ void tile_gemm_ta(f32x4& acc, const float* At, int lda, int row0, const float* B,
                                             int ldb, int col0, int kdim) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const float* Ap = At + q * lda + row0 + r;
  const float* Bp = B + q * ldb + col0 + r;
#pragma unroll 1
  for (int k0 = 0; k0 < kdim; k0 += 16) {
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = Ap[(k0 + 4 * u) * lda];
      bv[u] = Bp[(k0 + 4 * u) * ldb];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  }
}

struct FusedBwdDims {
  int n4, F4, O4, ldt, ldxa, ldg;
  int64_t o_svec, o_dd, o_rs, o_colp, o_tile, o_dx, o_g1, o_r3, o_wt, o_x0, o_m, total;
};

AGCN_HD inline FusedBwdDims fused_bwd_dims(int n, int Fin, int Fout, int K, bool gbw = false, bool gbm = false) {
  FusedBwdDims d;
  d.n4 = (n + 15) & ~15;
  d.F4 = (Fin + 15) & ~15;
  d.O4 = (Fout + 15) & ~15;
  d.ldt = ld_frag(d.n4);
  d.ldxa = ld_frag(d.F4);
  d.ldg = ld_frag(d.O4);
  auto r4 = [](int64_t v) { return (v + 3) & ~int64_t(3); };
  int64_t o = 32;
  d.o_svec = o; o += d.n4;
  d.o_dd = o;   o += d.n4;
  d.o_rs = o;   o += d.n4;
  d.o_colp = o; o += 16 * (int64_t)d.n4;
  d.o_tile = o; o += r4((int64_t)d.n4 * d.ldt);
  d.o_dx = o;   o += r4((int64_t)d.n4 * d.ldxa);
  d.o_g1 = o;   o += r4((int64_t)d.n4 * d.ldxa);
  d.o_r3 = o;
  const int64_t gp = r4((int64_t)d.n4 * d.ldg), wt = gbw ? 0 : r4((int64_t)K * d.F4 * d.ldg);
  const int64_t x0 = r4((int64_t)d.n4 * d.ldxa), m = gbm ? 0 : r4((int64_t)d.F4 * d.ldxa);
  d.o_wt = o + gp;
  d.o_x0 = o;
  d.o_m = o + x0;
  o += (gp + wt > x0 + m) ? gp + wt : x0 + m;
  d.total = o;
  return d;
}

AGCN_HD inline FusedDims fused_dims(int n, int Fin, int Fout, int K, bool gb = false) {
  FusedDims d;
  d.n4 = (n + 15) & ~15;
  d.F4 = (Fin + 15) & ~15;
  d.F16 = d.F4;
  d.ldt = ld_frag(d.n4);
  d.ldxa = ld_frag(d.F16);
  d.ldm = ld_b(Fin);
  d.ldw = ld_b(Fout);
  d.KF4 = K * d.F4;
  auto r4 = [](int64_t v) { return (v + 3) & ~int64_t(3); };
  int64_t o = 32;
  d.o_svec = o; o += r4(d.n4);
  d.o_tile = o; o += r4((int64_t)d.n4 * d.ldt);
  d.o_ximg = o; o += r4((int64_t)d.n4 * d.ldxa);
  const int64_t zt = (int64_t)Fin * d.ldt, x1 = (int64_t)d.n4 * d.ldxa;
  d.o_zx = o; o += r4(zt > x1 ? zt : x1);
  d.o_mimg = o; if (!gb) o += r4((int64_t)d.F4 * d.ldm);   // gb: M_L and W are read from global memory
  d.o_wimg = o; if (!gb) o += r4((int64_t)d.KF4 * d.ldw);
  d.total = o;
  return d;
}
}  // namespace

template <bool GB, bool RESLAP>
__global__ __launch_bounds__(1024) void fused_forward_kernel(SgcArgs a, int first) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  // one 32-byte record instead of the dependent order -> n / row_off / tile_off loads
  const f32x4* gdp = reinterpret_cast<const f32x4*>(a.graph_desc + first + blockIdx.x);
  const f32x4 gd0 = gdp[0], gd1 = gdp[1];
  const int b = __float_as_int(gd0[0]);
  const int n = __float_as_int(gd0[1]);
  const int row0 = __float_as_int(gd0[2]);
  const int64_t toff = (int64_t)(((uint64_t)(uint32_t)__float_as_int(gd1[1]) << 32) | (uint32_t)__float_as_int(gd1[0]));
  const int64_t tend = (int64_t)(((uint64_t)(uint32_t)__float_as_int(gd1[3]) << 32) | (uint32_t)__float_as_int(gd1[2]));
  const int Fin = a.Fin, Fout = a.Fout, K = a.K;   // K == 2
  const FusedDims d = fused_dims(n, Fin, Fout, K, GB);
  const int n4 = d.n4, F4 = d.F4, ldt = d.ldt, ldxa = d.ldxa, ldm = d.ldm, ldw = d.ldw;
  const bool paper = a.mode == AGCN_MODE_PAPER;
  const bool relu = a.activation == AGCN_ACT_RELU;
  float* s_vec = smem + d.o_svec;
  float* tile = smem + d.o_tile;
  float* Ximg = smem + d.o_ximg;     // [n4][ldxa] X_0, rows >= n and columns >= Fin zero
  float* Zt = smem + d.o_zx;         // [Fin][ldt] Z^T; later
  float* X1img = smem + d.o_zx;      // [n4][ldxa] X_1
  float* Mimg = smem + d.o_mimg;     // [F4][ldm]  M_L, rows >= Fin zero
  float* Wimg = smem + d.o_wimg;     // [K*F4][ldw] W regrouped: row k*F4 + f = W[f*K + k, :]
  const float* Xg = a.X + (int64_t)row0 * a.ldx;
  const FastDiv divn((uint32_t)n);
  const int nn = n * n;

  AGCN_STAMP(1, 0);
  // ---- F0: one bulk round for everything the graph needs from memory -------------------------------
  int csr_p0 = 0, csr_p1 = 0;
  if (a.L0_rowptr && tid < n) {
    csr_p0 = a.L0_rowptr[row0 + tid];
    csr_p1 = a.L0_rowptr[row0 + tid + 1];
  }
  // dense original Laplacian (the reference's feed, graph_topology.py:92-98, or packed tiles): added row by row
  const int l0ld = a.L0_ld > 0 ? a.L0_ld : n;
  const float* L0d = !a.L0_dense ? nullptr
                                 : (a.L0_ld > 0 ? a.L0_dense + (int64_t)b * a.L0_ld * a.L0_ld : a.L0_dense + toff);
  int csr_c[4] = {0, 0, 0, 0};
  float csr_v[4] = {0.f, 0.f, 0.f, 0.f};
  bool csr_loaded = false;
  const int cpF = (F4 + 63) >> 6, cpO = (Fout + 63) >> 6;   // column passes of 64 lanes
  auto ident = [](int r) { return r; };
  // image row k*F4 + f  <-  W row f*K + k   (graphconv.py:243-244 interleaving undone once, here)
  auto wmap = [&](int jr) { const int k = jr >= F4 ? 1 : 0; const int f = jr - k * F4; return f < Fin ? f * K + k : -1; };
  const int lane = tid & 63, wave = tid >> 6;
  if (GB) {
    // parameters stay in global memory: only the graph's own rows are staged
    if (cpF <= 2) stage_matrix<2>(Ximg, ldxa, Xg, a.ldx, n, Fin, n4, F4, ident);
    else stage_matrix<4>(Ximg, ldxa, Xg, a.ldx, n, Fin, n4, F4, ident);
  } else if (nthr == 1024 && n4 <= 80 && F4 <= 128 && Fout <= 64 && K * F4 <= 160 && ldw >= 64) {
    // common shapes: ALL loads of X, M_L and W are issued before the first LDS store (one round trip):
    // wave w takes rows w, w+16, ..; lanes run along the columns (2 passes for the features)
    float xr[5][2], mr[8][2], wr[10];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const float* rp = Xg + (int64_t)min(wave + 16 * u, n - 1) * a.ldx;
      xr[u][0] = rp[min(lane, Fin - 1)];
      xr[u][1] = rp[min(lane + 64, Fin - 1)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float* rp = a.M_L + (int64_t)min(wave + 16 * u, Fin - 1) * Fin;
      mr[u][0] = rp[min(lane, Fin - 1)];
      mr[u][1] = rp[min(lane + 64, Fin - 1)];
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int jr = min(wave + 16 * u, K * F4 - 1);
      wr[u] = a.weight[(int64_t)max(wmap(jr), 0) * Fout + min(lane, Fout - 1)];
    }
    // CSR entries of this thread's row: their addresses depend on the row pointers, which were the FIRST
    // loads issued -- requesting them here, before the LDS stores below, hides this dependent round trip
    if (tid < n && csr_p1 > csr_p0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = min(csr_p0 + u, csr_p1 - 1);
        csr_c[u] = a.L0_col[pp];
        csr_v[u] = a.L0_val[pp];
      }
    }
    csr_loaded = true;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = wave + 16 * u;
      if (i < n4) {
        if (lane < F4) Ximg[i * ldxa + lane] = (i < n && lane < Fin) ? xr[u][0] : 0.f;
        if (lane + 64 < F4) Ximg[i * ldxa + lane + 64] = (i < n && lane + 64 < Fin) ? xr[u][1] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int k = wave + 16 * u;
      if (k < F4) {
        if (lane < Fin) Mimg[k * ldm + lane] = (k < Fin) ? mr[u][0] : 0.f;
        if (lane + 64 < Fin) Mimg[k * ldm + lane + 64] = (k < Fin) ? mr[u][1] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 10; ++u) {
      const int jr = wave + 16 * u;
      if (jr < K * F4 && lane < Fout) Wimg[jr * ldw + lane] = (wmap(jr) >= 0) ? wr[u] : 0.f;
    }
  } else {
    if (cpF <= 2) {
      stage_matrix<2>(Ximg, ldxa, Xg, a.ldx, n, Fin, n4, F4, ident);
      stage_matrix<2>(Mimg, ldm, a.M_L, Fin, Fin, Fin, F4, Fin, ident);
    } else {
      stage_matrix<4>(Ximg, ldxa, Xg, a.ldx, n, Fin, n4, F4, ident);
      stage_matrix<4>(Mimg, ldm, a.M_L, Fin, Fin, Fin, F4, Fin, ident);
    }
    if (cpO <= 1) stage_matrix<1>(Wimg, ldw, a.weight, Fout, K * F4, Fout, K * F4, Fout, wmap);
    else stage_matrix<4>(Wimg, ldw, a.weight, Fout, K * F4, Fout, K * F4, Fout, wmap);
  }
  if (!csr_loaded && tid < n && csr_p1 > csr_p0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = min(csr_p0 + u, csr_p1 - 1);
      csr_c[u] = a.L0_col[pp];
      csr_v[u] = a.L0_val[pp];
    }
  }
  __syncthreads();

  AGCN_STAMP(1, 1);
  // ---- F1: Z = X . M_L -> Z^T image + reserve ------------------------------------------------------------
  {
    float* Zg = a.sv_Z + (int64_t)row0 * Fin;
    const int wv = tid >> 6, nw = nthr >> 6, r = tid & 15, q = (tid & 63) >> 4;
    const int tm = n4 >> 4, tn = F4 >> 4;
    for (int t = wv; t < tm * tn; t += nw) {
      const int in = t / tm, im = t - in * tm;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (GB) tile_gemm_gb(acc, Ximg, ldxa, 16 * im, a.M_L, Fin, Fin, 16 * in, Fin, F4);
      else tile_gemm_rm(acc, Ximg, ldxa, 16 * im, Mimg, ldm, 16 * in, F4);
      const int c = 16 * in + r;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int i = 16 * im + 4 * q + i4;
        if (i < n && c < Fin) {
          Zt[c * ldt + i] = acc[i4];
          Zg[(int64_t)i * Fin + c] = acc[i4];
        }
      }
    }
  }
  __syncthreads();

  AGCN_STAMP(1, 2);
  // ---- F2: similarity tile (2x2 blocks, upper triangle, feature loop split over adjacent lanes) -----
  {
    const int nb2 = (n + 1) >> 1;
    const TriFold tri(nb2);
    const FastDiv dW((uint32_t)tri.W);
    int fs_log = 0;
    while (fs_log < 4 && (tri.count() << (fs_log + 1)) <= nthr) ++fs_log;
    const int fs = 1 << fs_log;
    const int part = tid & (fs - 1);
    for (int tb = 0; tb < tri.count() * fs; tb += nthr) {
      const int t = (tb + tid) >> fs_log;
      int bi = 0, bj = 0;
      const bool live = t < tri.count() && tri.get(t, dW, bi, bj);
      const int i0 = 2 * bi, j0 = 2 * bj;
      const float* zi = Zt + i0;
      const float* zj = Zt + j0;
      f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
      if (live) {
#pragma unroll 5
        for (int f = part; f < Fin; f += fs) {
          // packed fp32: two pairs per instruction (same per-element fma chain as the scalar form)
          const f32x2 u = *reinterpret_cast<const f32x2*>(zi + f * ldt);
          const f32x2 v = *reinterpret_cast<const f32x2*>(zj + f * ldt);
          const f32x2 d0 = (f32x2){u.x, u.x} - v, d1 = (f32x2){u.y, u.y} - v;
          acc0 = __builtin_elementwise_fma(d0, d0, acc0);
          acc1 = __builtin_elementwise_fma(d1, d1, acc1);
        }
      }
      float a00 = acc0.x, a01 = acc0.y, a10 = acc1.x, a11 = acc1.y;
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
              const float r = sqrtf(acc[di][dj]);
              const float sv = (i == j) ? 0.f : expf(-r);   // graphconv.py:171-177
              tile[i * ldt + j] = sv;
              tile[j * ldt + i] = sv;   // bit-symmetric by construction
              if (paper) {              // r_ij for the backward (c_ij = (dr_ij + dr_ji) / r_ij)
                a.sv_r[toff + (int64_t)i * n + j] = r;
                a.sv_r[toff + (int64_t)j * n + i] = r;
              }
            }
          }
      }
    }
  }
  __syncthreads();
  AGCN_STAMP(1, 3);
  // degree (column sums in row order), s = d^-1/2   graphconv.py:195-197
  for (int j = tid; j < n; j += nthr) {
    float dsum = 0.f;
    for (int i = 0; i < n; ++i) dsum += tile[i * ldt + j];
    dsum += FLT_TRUE_MIN;
    const float s = 1.0f / sqrtf(dsum);
    s_vec[j] = s;
    a.sv_s[row0 + j] = s;
  }
  {
    const int npadt = (int)(tend - toff) - nn;
    if (tid < npadt) {
      a.res_W[toff + nn + tid] = 0.f;
      a.res_L[toff + nn + tid] = 0.f;
      if (a.L_sum) a.L_sum[toff + nn + tid] = 0.f;
      if (paper) a.sv_r[toff + nn + tid] = 0.f;
    }
    if (!RESLAP && tid == 0) {   // clip_by_average_norm never clips for SGC_LL: scale exactly 1 (see slab_kernels.hip)
      float* sc = a.sv_scal + 4 * b;
      sc[0] = 1.f; sc[1] = 1.f; sc[2] = 0.f; sc[3] = 1.f;
    }
  }
  __syncthreads();

  AGCN_STAMP(1, 4);
  // ---- F3: res_W out; R, (clip,) leaky-ReLU, res_L out; L = lrelu(R) + L_orig (+ Reslap combine) ------------
  float* red = smem;   // the 32 floats in front of s_vec: block reductions
  float ssR = 1.f, scR = 1.f, ssL = 0.f, scL = 1.f;
  {
    const float alpha = a.alpha[0];
    if (RESLAP) {
      // clip_by_norm(R, 1) needs ||R||_F of the whole tile first (graphconv_reslap.py:185)
      float part = 0.f;
      for (int idx = tid; idx < nn; idx += nthr) {
        const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
        const float S = tile[i * ldt + j];
        a.res_W[toff + idx] = S;
        float R = (i == j) ? 1.f : 0.f;
        if (paper) R = R - (s_vec[i] * S) * s_vec[j];
        tile[i * ldt + j] = R;
        part += R * R;
      }
      ssR = block_sum(part, red);
      scR = clip_scale(ssR, 1.0f);
      for (int idx = tid; idx < nn; idx += nthr) {
        const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
        const float R2 = lrelu(tile[i * ldt + j] * scR, alpha);
        a.res_L[toff + idx] = R2;
        tile[i * ldt + j] = R2;
      }
    } else {
      for (int idx = tid; idx < nn; idx += nthr) {
        const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
        const float S = tile[i * ldt + j];
        a.res_W[toff + idx] = S;
        float R = (i == j) ? 1.f : 0.f;
        if (paper) R = R - (s_vec[i] * S) * s_vec[j];
        const float R2 = lrelu(R, alpha);   // (R * 1.0f: clip scale exactly 1)
        a.res_L[toff + idx] = R2;
        tile[i * ldt + j] = R2;
      }
    }
    // the MFMA loops read the 16-padded tile: zero columns n..n4 of the real rows and rows n..n4
    for (int idx = tid; idx < n * (n4 - n); idx += nthr) {
      const int w = n4 - n, i = idx / w, j = n + idx - i * w;
      tile[i * ldt + j] = 0.f;
    }
    for (int idx = tid; idx < (n4 - n) * n4; idx += nthr) {
      const int i = n + idx / n4, j = idx - (idx / n4) * n4;
      tile[i * ldt + j] = 0.f;
    }
    __syncthreads();
    if (L0d) {
      for (int i = wave; i < n; i += nthr >> 6)
        for (int j = lane; j < n; j += 64) tile[i * ldt + j] += L0d[(int64_t)i * l0ld + j];
    } else if (tid < n) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (csr_p0 + u < csr_p1) tile[tid * ldt + csr_c[u]] += csr_v[u];
      for (int p = csr_p0 + 4; p < csr_p1; ++p) tile[tid * ldt + a.L0_col[p]] += a.L0_val[p];
    }
    __syncthreads();
    if (RESLAP) {
      // + beta * L_prev (graphconv_reslap.py:188-190); keep L_pre for the backward; second clip + leaky (:194-198)
      const bool has_prev = a.L_prev != nullptr;
      const float beta = has_prev ? a.beta[0] : 0.f;
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
      if (tid == 0) {   // norms / scales for the backward projections
        float* sc = a.sv_scal + 4 * b;
        sc[0] = ssR; sc[1] = scR; sc[2] = ssL; sc[3] = scL;
      }
      __syncthreads();
    } else if (a.L_sum) {
      for (int idx = tid; idx < nn; idx += nthr) {
        const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
        a.L_sum[toff + idx] = tile[i * ldt + j];
      }
    }
  }

  AGCN_STAMP(1, 5);
  // ---- F4: X_1 = L X_0 -> X_1 image (over Z^T) + reserve -----------------------------------------------
  {
    float* Xo = a.sv_X + (int64_t)row0 * Fin;
    const int wv = tid >> 6, nw = nthr >> 6, r = tid & 15, q = (tid & 63) >> 4;
    const int tm = n4 >> 4, tn = F4 >> 4;
    for (int t = wv; t < tm * tn; t += nw) {
      const int in = t / tm, im = t - in * tm;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      tile_gemm_rm(acc, tile, ldt, 16 * im, Ximg, ldxa, 16 * in, n4);
      const int c = 16 * in + r;
      // every cell of the 16-padded image is written: rows >= n and columns >= Fin come out as 0
      // (zero rows of L / zero columns of the X image)
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int i = 16 * im + 4 * q + i4;
        X1img[i * ldxa + c] = acc[i4];
        if (i < n && c < Fin) Xo[(int64_t)i * Fin + c] = acc[i4];
      }
    }
  }
  __syncthreads();

  AGCN_STAMP(1, 6);
  // ---- F5: Y = act(X_0 W_0 + X_1 W_1 + b) -----------------------------------------------------------------
  {
    float* Yg = a.Y + (int64_t)row0 * a.ldy;
    const int wv = tid >> 6, nw = nthr >> 6, r = tid & 15, q = (tid & 63) >> 4;
    const int tm = n4 >> 4, tn = (Fout + 15) >> 4;
    for (int t = wv; t < tm * tn; t += nw) {
      const int in = t / tm, im = t - in * tm;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (GB) {   // W row f*K + k (graphconv.py:243-244): consecutive f are K rows apart
        tile_gemm_gb(acc, Ximg, ldxa, 16 * im, a.weight, K * Fout, Fin, 16 * in, Fout, F4);
        tile_gemm_gb(acc, X1img, ldxa, 16 * im, a.weight + Fout, K * Fout, Fin, 16 * in, Fout, F4);
      } else {
        tile_gemm_rm(acc, Ximg, ldxa, 16 * im, Wimg, ldw, 16 * in, F4);
        tile_gemm_rm(acc, X1img, ldxa, 16 * im, Wimg + F4 * ldw, ldw, 16 * in, F4);
      }
      const int c = 16 * in + r;
      if (c < Fout) {
        const float bc = a.bias[c];
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const int i = 16 * im + 4 * q + i4;
          if (i < n) {
            float v = acc[i4] + bc;
            if (relu) v = fmaxf(v, 0.f);
            Yg[(int64_t)i * a.ldy + c] = v;
          }
        }
      }
    }
  }
  AGCN_STAMP(1, 7);
}

// ------------------------------------------------------------------------------------------------
// Fused backward (SGC_LL, K = 2): everything local to a graph in one launch.
//   B0  Gp = dY . act'(Y);  [g_0|g_1] = Gp . W^T            adjoint of graphconv.py:247
//   B1  dX = g_0 + L^T g_1                                   adjoint of graphconv.py:231
//   B2  (paper) dL = g_1 X_0^T -> dalpha, dA, dd, dr, c -> dZ = (diag(c 1) - c) Z;  dX += dZ M_L^T
// The parameter gradients that reduce over the whole batch (dweight, dbias, dM_L) are formed by
// param_grads.hip from Gp / dZ written here.
// GBW / GBM: W (for [g_0|g_1] = Gp . W^T) / M_L (for dX += dZ . M_L^T) read from global memory instead of an LDS image
template <bool GBW, bool GBM, bool RESLAP>
__global__ __launch_bounds__(1024) void fused_backward_kernel(SgcArgs a, int first) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nw = nthr >> 6;
  const int r = lane & 15, q = lane >> 4;
  // one 32-byte record instead of the dependent order -> n / row_off / tile_off loads
  const f32x4* gdp = reinterpret_cast<const f32x4*>(a.graph_desc + first + blockIdx.x);
  const f32x4 gd0 = gdp[0], gd1 = gdp[1];
  const int b = __float_as_int(gd0[0]);
  const int n = __float_as_int(gd0[1]);
  const int row0 = __float_as_int(gd0[2]);
  const int64_t toff = (int64_t)(((uint64_t)(uint32_t)__float_as_int(gd1[1]) << 32) | (uint32_t)__float_as_int(gd1[0]));
  const int64_t tend = (int64_t)(((uint64_t)(uint32_t)__float_as_int(gd1[3]) << 32) | (uint32_t)__float_as_int(gd1[2]));
  const int Fin = a.Fin, Fout = a.Fout, K = a.K;   // K == 2
  const FusedBwdDims d = fused_bwd_dims(n, Fin, Fout, K, GBW, GBM);
  const int n4 = d.n4, F4 = d.F4, O4 = d.O4, ldt = d.ldt, ldxa = d.ldxa, ldg = d.ldg;
  const bool paper = a.mode == AGCN_MODE_PAPER;
  const bool relu = a.activation == AGCN_ACT_RELU;
  float* red = smem;
  float* s_vec = smem + d.o_svec;
  float* ddv = smem + d.o_dd;
  float* rsv = smem + d.o_rs;
  float* colp = smem + d.o_colp;     // [16][n4] per-wave column partials
  float* tile = smem + d.o_tile;     // L -> dL -> dr -> c
  float* dXimg = smem + d.o_dx;      // [n4][ldxa] dX accumulator
  float* G1img = smem + d.o_g1;      // [n4][ldxa] g_1; later Z (row-major)
  float* GPimg = smem + d.o_r3;      // [n4][ldg]  Gp                 (B0)
  float* WTimg = smem + d.o_wt;      // [K*F4][ldg] W regrouped        (B0)
  float* X0img = smem + d.o_x0;      // [n4][ldxa] X_0; later dZ      (B2)
  float* Mimg = smem + d.o_m;        // [F4][ldxa] M_L row-major      (B2)
  const float* Xg = a.X + (int64_t)row0 * a.ldx;
  const float* Yg = a.Y + (int64_t)row0 * a.ldy;
  const float* dYg = a.dY + (int64_t)row0 * a.ldy;
  const float alpha = a.alpha[0];

  if (blockIdx.x == 0) {   // arrival counters of the parameter-gradient kernel that follows this one
    for (int e = tid; e < a.zero_count; e += nthr) a.zero_ptr[e] = 0;
  }
  // global-B with 16-byte loads along the contiguous k of W^T / M_L^T rows (A images stored k-permuted)
  const bool vecW = GBW, vecM = GBM;
  AGCN_STAMP(4, 0);
  // ---- B0: loads: dY, Y rows -> Gp image (+ global copy); W -> regrouped image; CSR rows; s --------
  int csr_p0 = 0, csr_p1 = 0;
  if (tid < n) {
    if (a.L0_rowptr) {
      csr_p0 = a.L0_rowptr[row0 + tid];
      csr_p1 = a.L0_rowptr[row0 + tid + 1];
    }
    s_vec[tid] = paper ? a.sv_s[row0 + tid] : 0.f;
  }
  const int l0ld = a.L0_ld > 0 ? a.L0_ld : n;
  const float* L0d = !a.L0_dense ? nullptr
                                 : (a.L0_ld > 0 ? a.L0_dense + (int64_t)b * a.L0_ld * a.L0_ld : a.L0_dense + toff);
  // L rows (wave = row, lanes = columns, 2 column passes): requested now, stored after the staging.
  // SGC_LL: L = res_L + L_orig is rebuilt; Reslap: the combined, twice clipped Laplacian is the saved 4th output
  const float* Lsrc = RESLAP ? a.L_sum : a.res_L;
  const bool lpref = (nw == 16 && n4 <= 128);
  float lreg[8][2];
  if (lpref) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(wave + 16 * u, n - 1);
      lreg[u][0] = Lsrc[toff + (int64_t)i * n + min(lane, n - 1)];
      lreg[u][1] = Lsrc[toff + (int64_t)i * n + min(lane + 64, n - 1)];
    }
  }
  {
    // rows i = wave, wave + 16, ..; lanes along the Fout columns (passes of 64)
    for (int i0 = wave; i0 < n4; i0 += 4 * nw) {
      float g[4], y[4];
      for (int cp = 0; cp < O4; cp += 64) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = min(i0 + u * nw, n - 1), c = min(cp + lane, Fout - 1);
          g[u] = dYg[(int64_t)i * a.ldy + c];
          y[u] = relu ? Yg[(int64_t)i * a.ldy + c] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * nw, c = cp + lane;
          if (i < n4 && c < O4) {
            const float v = (i < n && c < Fout && y[u] > 0.f) ? g[u] : 0.f;
            GPimg[i * ldg + (vecW ? perm16(c) : c)] = v;
            if (i < n && c < Fout) a.ws_Gp[(int64_t)(row0 + i) * Fout + c] = v;
          }
        }
      }
    }
    // W image row k*F4 + f <- W row f*K + k, columns = Fout (the GEMM's k dimension)
    auto wmap = [&](int jr) { const int k = jr >= F4 ? 1 : 0; const int f = jr - k * F4; return f < Fin ? f * K + k : -1; };
    if (!GBW) {
      if (O4 <= 64) stage_matrix<1>(WTimg, ldg, a.weight, Fout, K * F4, Fout, K * F4, O4, wmap);
      else stage_matrix<4>(WTimg, ldg, a.weight, Fout, K * F4, Fout, K * F4, O4, wmap);
    }
  }
  int csr_c[4] = {0, 0, 0, 0};
  float csr_v[4] = {0.f, 0.f, 0.f, 0.f};
  if (tid < n && csr_p1 > csr_p0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int pp = min(csr_p0 + u, csr_p1 - 1);
      csr_c[u] = a.L0_col[pp];
      csr_v[u] = a.L0_val[pp];
    }
  }
  AGCN_STAMP(4, 1);
  // tile = L = res_L + L_orig (the tile's 16-padding is zeroed: it is the MFMA k range).  The res_L
  // loads were requested at the top of the kernel (lreg), together with the dY / Y / W loads.
  if (lpref) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = wave + nw * u;
      if (i < n4) {
        if (lane < n4) tile[i * ldt + lane] = (i < n && lane < n) ? lreg[u][0] : 0.f;
        if (lane + 64 < n4) tile[i * ldt + lane + 64] = (i < n && lane + 64 < n) ? lreg[u][1] : 0.f;
      }
    }
  } else {
    for (int i = wave; i < n4; i += nw) {
      for (int j = lane; j < n4; j += 64) {
        float v = 0.f;
        if (i < n && j < n) v = Lsrc[toff + (int64_t)i * n + j];
        tile[i * ldt + j] = v;
      }
    }
  }
  __syncthreads();
  if (RESLAP) {
    // (L_sum already contains L_orig and beta * L_prev)
  } else if (L0d) {
    for (int i = wave; i < n; i += nw)
      for (int j = lane; j < n; j += 64) tile[i * ldt + j] += L0d[(int64_t)i * l0ld + j];
  } else if (tid < n) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (csr_p0 + u < csr_p1) tile[tid * ldt + csr_c[u]] += csr_v[u];
    for (int p = csr_p0 + 4; p < csr_p1; ++p) tile[tid * ldt + a.L0_col[p]] += a.L0_val[p];
  }
  AGCN_STAMP(4, 2);
  // [g_0 | g_1] = Gp . W^T : output columns jr = k*F4 + f
  {
    const int tm = n4 >> 4, tn = (K * F4) >> 4;
    for (int t = wave; t < tm * tn; t += nw) {
      const int in = t / tm, im = t - in * tm;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int jr = 16 * in + r;
      float* dst = (jr < F4) ? dXimg : G1img;
      const int f = (jr < F4) ? jr : jr - F4;
      if (GBW) {   // B[c][jr] = W[f*K + k][c]: the lane's W row, k along the Fout columns
        const int kk = (jr < F4) ? 0 : 1;
        const float* wrow = a.weight + (int64_t)(min(f, Fin - 1) * K + kk) * Fout;
        tile_gemm_gbt4(acc, GPimg, ldg, 16 * im, wrow, Fout, O4);
        if (f >= Fin) acc = (f32x4){0.f, 0.f, 0.f, 0.f};   // image columns Fin..F4 are the zero padding
      } else {
        tile_gemm_tb(acc, GPimg, ldg, 16 * im, WTimg, ldg, 16 * in, O4);
      }
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) dst[(16 * im + 4 * q + i4) * ldxa + f] = acc[i4];
    }
  }
  __syncthreads();

  AGCN_STAMP(4, 3);
  // paper mode: the X_0, M_L and Z rows needed by B2 are requested NOW so that their round trip overlaps the
  // matrix work of B1; they are written to LDS after the barrier that retires the B0 images.
  const bool xpref = paper && nw == 16 && n4 <= 80 && F4 <= 128;
  float x0r[5][2], m0r[8][2], z0r[5][2];
  if (xpref) {
    const float* Zg0 = a.sv_Z + (int64_t)row0 * Fin;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int i = min(wave + 16 * u, n - 1);
      x0r[u][0] = Xg[(int64_t)i * a.ldx + min(lane, Fin - 1)];
      x0r[u][1] = Xg[(int64_t)i * a.ldx + min(lane + 64, Fin - 1)];
      z0r[u][0] = Zg0[(int64_t)i * Fin + min(lane, Fin - 1)];
      z0r[u][1] = Zg0[(int64_t)i * Fin + min(lane + 64, Fin - 1)];
    }
    if (!GBM) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = min(wave + 16 * u, Fin - 1);
        m0r[u][0] = a.M_L[(int64_t)k * Fin + min(lane, Fin - 1)];
        m0r[u][1] = a.M_L[(int64_t)k * Fin + min(lane + 64, Fin - 1)];
      }
    }
  }
  // ---- B1: dX += L^T g_1 -----------------------------------------------------------------------------------
  {
    const int tm = n4 >> 4, tn = F4 >> 4;
    for (int t = wave; t < tm * tn; t += nw) {
      const int in = t / tm, im = t - in * tm;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      tile_gemm_ta(acc, tile, ldt, 16 * im, G1img, ldxa, 16 * in, n4);
      const int f = 16 * in + r;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) dXimg[(16 * im + 4 * q + i4) * ldxa + f] += acc[i4];
    }
  }
  float dalpha_part = 0.f, dbeta_part = 0.f;
  if (RESLAP && a.dL_prev) {   // alignment floats behind the tile
    const int npadt = (int)(tend - toff) - n * n;
    if (tid < npadt) a.dL_prev[toff + n * n + tid] = 0.f;
  }
  if (paper || RESLAP) {
    __syncthreads();   // GPimg / WTimg and the L tile are dead
    AGCN_STAMP(4, 4);
    // ---- B2: stage X_0 and M_L (row-major) over the dead B0 images -----------------------------------
    if (xpref) {
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int i = wave + 16 * u;
        if (i < n4) {
          if (lane < F4) X0img[i * ldxa + lane] = (i < n && lane < Fin) ? x0r[u][0] : 0.f;
          if (lane + 64 < F4) X0img[i * ldxa + lane + 64] = (i < n && lane + 64 < Fin) ? x0r[u][1] : 0.f;
        }
      }
      if (!GBM) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = wave + 16 * u;
          if (k < F4) {
            if (lane < F4) Mimg[k * ldxa + lane] = (k < Fin && lane < Fin) ? m0r[u][0] : 0.f;
            if (lane + 64 < F4) Mimg[k * ldxa + lane + 64] = (k < Fin && lane + 64 < Fin) ? m0r[u][1] : 0.f;
          }
        }
      }
    } else {
      auto ident = [](int rr) { return rr; };
      if (F4 <= 128) {
        stage_matrix<2>(X0img, ldxa, Xg, a.ldx, n, Fin, n4, F4, ident);
        if (!GBM && paper) stage_matrix<2>(Mimg, ldxa, a.M_L, Fin, Fin, Fin, F4, F4, ident);
      } else {
        stage_matrix<4>(X0img, ldxa, Xg, a.ldx, n, Fin, n4, F4, ident);
        if (!GBM && paper) stage_matrix<4>(Mimg, ldxa, a.M_L, Fin, Fin, Fin, F4, F4, ident);
      }
    }
    __syncthreads();
    AGCN_STAMP(4, 5);
    // dL = g_1 X_0^T -> tile
    {
      const int tm = n4 >> 4;
      for (int t = wave; t < tm * tm; t += nw) {
        const int in = t / tm, im = t - in * tm;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        tile_gemm_tb(acc, G1img, ldxa, 16 * im, X0img, ldxa, 16 * in, F4);
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) tile[(16 * im + 4 * q + i4) * ldt + 16 * in + r] = acc[i4];
      }
    }
    __syncthreads();
    AGCN_STAMP(4, 6);
    const FastDiv divn((uint32_t)n);
    const int nn = n * n;
    if (a.dL_sum) {   // upstream gradient of the combined Laplacian (4th output; the next Reslap layer's L_prev)
      for (int idx = tid; idx < nn; idx += nthr) {
        const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
        tile[i * ldt + j] += a.dL_sum[toff + idx];
      }
      __syncthreads();
    }
    if (RESLAP) {
      // adjoint of  La = lrelu(cn(Lpre)),  Lpre = R2 + L_orig + beta L_prev   (graphconv_reslap.py:188-198)
      const float* sc = a.sv_scal + 4 * b;
      const bool has_prev = a.L_prev != nullptr && a.dL_prev != nullptr;
      {
        const float ssL = sc[2], scL = sc[3];
        const float invL = 1.0f / sqrtf(ssL);
        const bool clipped = invL <= 1.0f;   // TF MinimumGrad: ties take the norm branch
        float dot_part = 0.f;
        for (int idx = tid; idx < nn; idx += nthr) {
          const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
          const float Lpre = a.sv_Lpre[toff + idx];
          const float Lc = Lpre * scL;
          const float dLa = tile[i * ldt + j];
          dalpha_part += dLa * fminf(Lc, 0.f);
          const float dLc = dLa * lrelu_grad(Lc, alpha);
          tile[i * ldt + j] = dLc;
          dot_part += dLc * Lpre;
        }
        float dot = 0.f;
        if (clipped) dot = block_sum(dot_part, red);
        const float beta = has_prev ? a.beta[0] : 0.f;
        const float k3 = dot * invL * invL * invL;
        for (int idx = tid; idx < nn; idx += nthr) {
          const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
          float dv = tile[i * ldt + j];
          if (clipped) dv = dv * invL - a.sv_Lpre[toff + idx] * k3;
          tile[i * ldt + j] = dv;   // dLpre == dR2
          if (has_prev) {
            dbeta_part += dv * a.L_prev[toff + idx];
            a.dL_prev[toff + idx] = dv * beta;
          }
        }
      }
      if (paper) {
        // adjoint of  R2 = lrelu(cn(R)),  R = I - s S s  (clip_by_norm: q = 1): the tile becomes dR
        const float ssR = sc[0], scR = sc[1];
        const float invR = 1.0f / sqrtf(ssR);
        const bool clippedR = invR <= 1.0f;
        const float* Sg = a.res_W + toff;
        float dot_part = 0.f;
        for (int idx = tid; idx < nn; idx += nthr) {
          const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
          float R = (i == j) ? 1.f : 0.f;
          R = R - (s_vec[i] * Sg[idx]) * s_vec[j];
          const float R1 = R * scR;
          const float dR2 = tile[i * ldt + j];
          dalpha_part += dR2 * fminf(R1, 0.f);
          const float dR1 = dR2 * lrelu_grad(R1, alpha);
          tile[i * ldt + j] = dR1;
          dot_part += dR1 * R;
        }
        float dotR = 0.f;
        if (clippedR) dotR = block_sum(dot_part, red);
        const float k3 = dotR * invR * invR * invR;
        if (clippedR) {
          for (int idx = tid; idx < nn; idx += nthr) {
            const int i = (int)divn.div((uint32_t)idx), j = idx - i * n;
            float R = (i == j) ? 1.f : 0.f;
            R = R - (s_vec[i] * Sg[idx]) * s_vec[j];
            tile[i * ldt + j] = tile[i * ldt + j] * invR - R * k3;
          }
        }
      }
      __syncthreads();
    }
    if (paper) {
    // Z rows -> G1img region (row-major B operand), g_1 is dead
    if (xpref) {
#pragma unroll
      for (int u = 0; u < 5; ++u) {
        const int i = wave + 16 * u;
        if (i < n4) {
          if (lane < F4) G1img[i * ldxa + lane] = (i < n && lane < Fin) ? z0r[u][0] : 0.f;
          if (lane + 64 < F4) G1img[i * ldxa + lane + 64] = (i < n && lane + 64 < Fin) ? z0r[u][1] : 0.f;
        }
      }
    } else {
      auto ident = [](int rr) { return rr; };
      const float* Zg = a.sv_Z + (int64_t)row0 * Fin;
      if (F4 <= 128) stage_matrix<2>(G1img, ldxa, Zg, Fin, n, Fin, n4, F4, ident);
      else stage_matrix<4>(G1img, ldxa, Zg, Fin, n, Fin, n4, F4, ident);
    }
    const float* S = a.res_W + toff;
    // pass 1: dalpha, P = dA . S, row sums  sum_j P_ij s_j  and per-wave column partials  sum_i P_ij s_i
    for (int j = lane; j < n4; j += 64) colp[wave * n4 + j] = 0.f;
    for (int i = wave; i < n; i += nw) {
      float rsum = 0.f;
      const float si = s_vec[i];
      for (int j = lane; j < n; j += 64) {
        const float Sv = S[(int64_t)i * n + j];
        float R = (i == j) ? 1.f : 0.f;
        R = R - (si * Sv) * s_vec[j];                  // R1 == R (clip scale 1)
        const float dL = tile[i * ldt + j];
        if (!RESLAP) dalpha_part += dL * fminf(R, 0.f);
        const float P = -(RESLAP ? dL : dL * lrelu_grad(R, alpha)) * Sv;   // dA_ij S_ij (Reslap: the tile is dR already)
        rsum += P * s_vec[j];
        colp[wave * n4 + j] += P * si;
      }
      rsum = wave_sum(rsum);
      if (lane == 0) rsv[i] = rsum;
    }
    __syncthreads();
    for (int k = tid; k < n; k += nthr) {
      float cs = 0.f;
      for (int w = 0; w < nw; ++w) cs += colp[w * n4 + k];
      const float sk = s_vec[k];
      ddv[k] = degree_grad(sk, rsv[k] + cs);   // dd_k = -1/2 s_k^3 ds_k
    }
    __syncthreads();
    AGCN_STAMP(4, 7);
    // pass 2: dr_ij = -S_ij (dA_ij s_i s_j + dd_j), in place
    for (int i = wave; i < n; i += nw) {
      const float si = s_vec[i];
      for (int j = lane; j < n; j += 64) {
        const float Sv = S[(int64_t)i * n + j];
        float R = (i == j) ? 1.f : 0.f;
        R = R - (si * Sv) * s_vec[j];
        const float dA = -(RESLAP ? tile[i * ldt + j] : tile[i * ldt + j] * lrelu_grad(R, alpha));
        tile[i * ldt + j] = (Sv == 0.f) ? 0.f : -Sv * (dA * si * s_vec[j] + ddv[j]);   // S == 0: no inf * 0
      }
    }
    __syncthreads();
    AGCN_STAMP(4, 8);
    // pass 3: c_ij = (dr_ij + dr_ji) / r_ij for i < j (0 where r_ij == 0), mirrored; diagonal 0
    for (int i = wave; i < n; i += nw) {
      for (int j = lane; j < n; j += 64) {
        if (j > i) {
          const float rr = a.sv_r[toff + (int64_t)i * n + j];
          const float num = tile[i * ldt + j] + tile[j * ldt + i];
          const float cij = rr > 0.f ? num / rr : 0.f;
          tile[i * ldt + j] = cij;
          tile[j * ldt + i] = cij;
        } else if (j == i) {
          tile[i * ldt + i] = 0.f;
        }
      }
    }
    __syncthreads();
    for (int i = wave; i < n; i += nw) {
      float rsum = 0.f;
      for (int j = lane; j < n; j += 64) rsum += tile[i * ldt + j];
      rsum = wave_sum(rsum);
      if (lane == 0) rsv[i] = rsum;
    }
    __syncthreads();
    AGCN_STAMP(4, 9);
    // dZ = rowsum(c) Z - c Z -> X0img region (A operand) + global
    {
      float* dZg = a.ws_dZ + (int64_t)row0 * Fin;
      const int tm = n4 >> 4, tn = F4 >> 4;
      for (int t = wave; t < tm * tn; t += nw) {
        const int in = t / tm, im = t - in * tm;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        tile_gemm_rm(acc, tile, ldt, 16 * im, G1img, ldxa, 16 * in, n4);
        const int f = 16 * in + r;
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const int i = 16 * im + 4 * q + i4;
          const float v = (i < n) ? rsv[i] * G1img[i * ldxa + f] - acc[i4] : 0.f;
          X0img[i * ldxa + (vecM ? perm16(f) : f)] = v;
          if (i < n && f < Fin) dZg[(int64_t)i * Fin + f] = v;
        }
      }
    }
    __syncthreads();
    AGCN_STAMP(4, 10);
    // dX += dZ M_L^T   (B[kk][f] = M[f][kk]: transposed read of the row-major M image)
    {
      const int tm = n4 >> 4, tn = F4 >> 4;
      for (int t = wave; t < tm * tn; t += nw) {
        const int in = t / tm, im = t - in * tm;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int f = 16 * in + r;
        if (GBM) {   // the lane's M_L row f, k along its columns
          const float* mrow = a.M_L + (int64_t)min(f, Fin - 1) * Fin;
          tile_gemm_gbt4(acc, X0img, ldxa, 16 * im, mrow, Fin, F4);
          if (f >= Fin) acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        } else {
          tile_gemm_tb(acc, X0img, ldxa, 16 * im, Mimg, ldxa, 16 * in, F4);
        }
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) dXimg[(16 * im + 4 * q + i4) * ldxa + f] += acc[i4];
      }
    }
    }   // paper
  }
  __syncthreads();
  AGCN_STAMP(4, 11);
  // ---- B3: dX rows out ---------------------------------------------------------------------------------------
  {
    float* dXg = a.dX + (int64_t)row0 * a.ldx;
    for (int i = wave; i < n; i += nw)
      for (int f = lane; f < Fin; f += 64) dXg[(int64_t)i * a.ldx + f] = dXimg[i * ldxa + f];
  }
  const float da = block_sum(dalpha_part, red);
  const float db = RESLAP ? block_sum(dbeta_part, red) : 0.f;
  if (tid == 0) {
    a.ws_pg[2 * b + 0] = da;
    a.ws_pg[2 * b + 1] = db;
  }
}

// diagnostic: AGCN_FUSED_THREADS overrides the workgroup size of the fused kernels (multiple of 64)
static int fused_threads() {
  static const int t = [] {
    const char* e = std::getenv("AGCN_FUSED_THREADS");
    int v = e ? std::atoi(e) : 1024;
    if (v < 64 || v > 1024 || (v & 63)) v = 1024;
    return v;
  }();
  return t;
}

size_t fused_lds_bytes(int32_t n, int32_t Fin, int32_t Fout, int32_t K) {
  return sizeof(float) * (size_t)fused_dims(n, Fin, Fout, K).total;
}
size_t fused_lds_bytes_gb(int32_t n, int32_t Fin, int32_t Fout, int32_t K) {
  return sizeof(float) * (size_t)fused_dims(n, Fin, Fout, K, true).total;
}

size_t fused_bwd_lds_bytes(int32_t n, int32_t Fin, int32_t Fout, int32_t K) {
  return sizeof(float) * (size_t)fused_bwd_dims(n, Fin, Fout, K).total;
}
size_t fused_bwd_lds_bytes_gb(int32_t n, int32_t Fin, int32_t Fout, int32_t K, bool gbw, bool gbm) {
  return sizeof(float) * (size_t)fused_bwd_dims(n, Fin, Fout, K, gbw, gbm).total;
}

template <bool GBW, bool GBM, bool RESLAP>
static int launch_fused_backward_t(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, hipStream_t s) {
  if (lds_bytes > 64 * 1024) {
    hipError_t e = static_cast<hipError_t>(
        ensure_dynamic_lds(reinterpret_cast<const void*>(&fused_backward_kernel<GBW, GBM, RESLAP>), lds_bytes));
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(fused_bwd): %s", hipGetErrorString(e));
  }
  prof_begin(PK_FUSED_BWD, s);
  hipLaunchKernelGGL((fused_backward_kernel<GBW, GBM, RESLAP>), dim3(count), dim3(fused_threads()), lds_bytes, s, a, first);
  prof_end(s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "fused_backward launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

template <bool RESLAP>
static int launch_fused_backward_v(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, bool gbw, bool gbm,
                                   hipStream_t s) {
  if (gbw) return gbm ? launch_fused_backward_t<true, true, RESLAP>(a, first, count, lds_bytes, s)
                      : launch_fused_backward_t<true, false, RESLAP>(a, first, count, lds_bytes, s);
  return gbm ? launch_fused_backward_t<false, true, RESLAP>(a, first, count, lds_bytes, s)
             : launch_fused_backward_t<false, false, RESLAP>(a, first, count, lds_bytes, s);
}

int launch_fused_backward(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, bool global_w,
                          bool global_m, void* stream) {
  if (count <= 0) return AGCN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.variant == AGCN_SGC_LL_RESLAP) return launch_fused_backward_v<true>(a, first, count, lds_bytes, global_w, global_m, s);
  return launch_fused_backward_v<false>(a, first, count, lds_bytes, global_w, global_m, s);
}

template <bool GB, bool RESLAP>
static int launch_fused_forward_t(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, hipStream_t s) {
  if (lds_bytes > 64 * 1024) {
    hipError_t e = static_cast<hipError_t>(ensure_dynamic_lds(reinterpret_cast<const void*>(&fused_forward_kernel<GB, RESLAP>), lds_bytes));
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(fused_fwd): %s", hipGetErrorString(e));
  }
  prof_begin(PK_FUSED_FWD, s);
  hipLaunchKernelGGL((fused_forward_kernel<GB, RESLAP>), dim3(count), dim3(fused_threads()), lds_bytes, s, a, first);
  prof_end(s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "fused_forward launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

int launch_fused_forward(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, bool global_b,
                         void* stream) {
  if (count <= 0) return AGCN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool reslap = a.variant == AGCN_SGC_LL_RESLAP;
  if (reslap) return global_b ? launch_fused_forward_t<true, true>(a, first, count, lds_bytes, s)
                              : launch_fused_forward_t<false, true>(a, first, count, lds_bytes, s);
  return global_b ? launch_fused_forward_t<true, false>(a, first, count, lds_bytes, s)
                  : launch_fused_forward_t<false, false>(a, first, count, lds_bytes, s);
}

}  // namespace agcn
