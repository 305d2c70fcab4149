// Feature contractions over the PACKED row space of the whole batch (every graph's nodes at
// once): Out[rows, N] = epilogue(A[rows, Kd] . B[Kd, N]) on the f32 matrix cores.
// Rows are uniform work, so these launches are load-balanced however ragged the graphs are.
//
//   RG_PROJECT      Z  = X . M_L                                   graphconv.py:164
//   RG_FILTER       Y  = act([X_0|X_1|..] . W + b), pad rows = 0   graphconv.py:238-251,117-123
//                        (A column j = f*K + k is X_k[:, f]: the interleaved concat is never built)
//   RG_FILTER_BWD   [g_0|g_1|..] = (dY . act'(Y)) . W^T            adjoint of the above; also emits Gp
//   RG_PROJECT_BWD  dX += dZ . M_L^T                               adjoint of RG_PROJECT (paper mode)
//   RG_DENSE        Y  = act([s *] A . W [+ b] [+ Add])            DenseMol (dense_layer.py:33-50, bias also on pad
//                                                                  rows), BlockEnd (blockend.py:67-86, x + x_res . W) and
//                                                                  one term of DenseBlockEnd (densenet_block.py:113-123)
//
// Workgroup = 1024 threads on 64 rows x up to 192 columns: all 16 waves stage the operands (the
// whole K extent, or the largest chunk that fits LDS, with 8 independent loads in flight per
// thread -- the kernel is latency-bound, not bandwidth-bound), then wave (rg, cg) owns row group rg
// (16 rows) and the column tiles cg, cg+4, cg+8.  A image ld = 2|14 mod 16, B image ld = 16 mod 32:
// conflict-free fragment reads.
// v_mfma_f32_16x16x4_f32 keeps exact fp32 fmaf-chain semantics.
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "agcn_device.h"
#include "sgc_kernels.h"

namespace agcn {

namespace {
constexpr int BM = 64;
constexpr int NTHR = 1024;
constexpr int LDS_BUDGET_FLOATS = 36 * 1024;   // 144 KiB
}  // namespace

template <int MODE, int NTW>
__global__ __launch_bounds__(NTHR) void rows_gemm_kernel(RowsGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int nwaves = blockDim.x >> 6, nthr = blockDim.x;   // 4, 8 or 16 waves
  const int ncg = nwaves >> 2;                              // column groups (4 row groups always)
  const int rg = wave & 3, cg = wave >> 2;
  const int r = lane & 15, q = lane >> 4;
  const int row_base = blockIdx.x * BM;
  const int ntb = a.nt;                // column tiles per workgroup
  const int col_base = blockIdx.y * (16 * ntb);
  const int Kd = a.Kd, N = a.N;
  const int bn = min(16 * ntb, N - col_base);
  const int KC = a.kc;                 // K chunk, multiple of 4
  const int ldA = ld_frag(KC);
  const int ldB = ld_b(16 * ntb);
  float* As = smem;                    // [BM][ldA]
  float* Bs = smem + BM * ldA;         // [KC][ldB]
  const int K = a.K, Fin = a.Fin, Fout = a.Fout;
  const bool relu = a.activation == AGCN_ACT_RELU;
  const int last_row = a.rows - 1;
  const FastDiv dK((uint32_t)K);
  const FastDiv dRS((uint32_t)(a.row_stride > 0 ? a.row_stride : 1));

  // row validity (padded feeds only: rows >= n[b] of a graph are padding), computed once per row
  float* rowok = Bs + KC * ldB;        // [BM] 1.0f / 0.0f
  if (tid < BM) {
    const int row = row_base + tid;
    bool ok = row < a.rows;
    if (ok && a.row_stride > 0) {
      const int g = (int)dRS.div((uint32_t)row);
      ok = row - g * a.row_stride < a.n[g];
    }
    if (a.keep_pad && row < a.rows) ok = true;
    if (ok && a.big_n > 0) ok = a.n[a.row_graph[row]] > a.big_n;   // mixed dispatch: big graphs only
    rowok[tid] = ok ? 1.f : 0.f;
  }
  if (MODE == RG_FILTER_BWD && blockIdx.x == 0 && blockIdx.y == 0) {
    for (int e = tid; e < a.zero_count; e += nthr) a.zero_ptr[e] = 0;
  }
  __syncthreads();
  if (a.big_n > 0) {   // nothing to do for a block without rows of big graphs
    float any = 0.f;
    for (int i = 0; i < BM; ++i) any += rowok[i];
    if (any == 0.f) return;
  }

  AGCN_STAMP(MODE == RG_PROJECT ? 0 : MODE == RG_FILTER ? 2 : MODE == RG_FILTER_BWD ? 3 : MODE == RG_DENSE ? 6 : 5, 0);
  f32x4 acc[NTW];
#pragma unroll
  for (int j = 0; j < NTW; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // this wave's column tiles: cg, cg + ncg, cg + 2 ncg, .. (those below ntb)
  int my_nt = 0;
#pragma unroll
  for (int j = 0; j < NTW; ++j) my_nt += (cg + ncg * j < ntb) ? 1 : 0;

  // Staging layout: wave `wave` (0..15) x lane; lanes run along the source-contiguous index, waves
  // along the other one.  No divisions, 32-bit offsets from per-row base pointers, and every load of
  // a round is issued before the first LDS store (the kernel is latency-bound).
  for (int k0 = 0; k0 < Kd; k0 += KC) {
    const int kc = min(KC, Kd - k0);
    // ---- B[KC x bn] loads (flat index, 4 per thread per round; rows kc..KC are zero) ---------------------
    // source orientation: PROJECT/FILTER  B[k][c] = Bm[k * ldbm + col_base + c]   (lanes along c)
    //                     *_BWD           B[k][c] = Bm[(col_base + c) * ldbm + k] (lanes along k)
    const bool transposed = (MODE == RG_FILTER_BWD || MODE == RG_PROJECT_BWD);
    const int ldbm = (MODE == RG_PROJECT) ? Fin : (MODE == RG_FILTER) ? Fout : (MODE == RG_FILTER_BWD) ? Fout
                     : (MODE == RG_DENSE) ? N : Fin;
    const int inner = transposed ? KC : bn;            // contiguous index extent
    const FastDiv dInner((uint32_t)inner);
    const int btotal = KC * bn;
    auto b_load = [&](int e) -> float {
      e = min(e, btotal - 1);
      const int o = (int)dInner.div((uint32_t)e), in = e - o * inner;
      const int kk = transposed ? in : o, c = transposed ? o : in;
      const int k = min(k0 + kk, Kd - 1);
      return transposed ? a.Bm[(int64_t)(col_base + c) * ldbm + k] : a.Bm[(int64_t)k * ldbm + col_base + c];
    };
    auto b_store = [&](int e, float v) {
      if (e < btotal) {
        const int o = (int)dInner.div((uint32_t)e), in = e - o * inner;
        const int kk = transposed ? in : o, c = transposed ? o : in;
        Bs[kk * ldB + c] = (kk < kc) ? v : 0.f;
      }
    };
    // ---- A[BM x KC]: 4 rows per wave, up to 3 column passes of 64 lanes.  The first round of B loads
    // is issued together with the A loads so that both latencies overlap. ---------------------------------
    float tb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) tb[u] = b_load(tid + nthr * u);
    for (int ipass = 0; ipass < BM; ipass += 4 * nwaves)
    for (int c0 = 0; c0 < KC; c0 += 192) {
      float t[4][3];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int i = min(wave + nwaves * rr + ipass, BM - 1);
        const int row = min(row_base + i, last_row);
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          const int kk = c0 + lane + 64 * cc;
          const int k = min(k0 + kk, Kd - 1);
          if (MODE == RG_FILTER) {
            const int f = (int)dK.div((uint32_t)k), kx = k - f * K;
            const float* p0 = a.A0 + (int64_t)row * a.lda0 + f;
            const float* p1 = a.A1 + (int64_t)(kx > 0 ? kx - 1 : 0) * a.a1_plane + (int64_t)row * Fin + f;
            t[rr][cc] = *(kx == 0 ? p0 : p1);
          } else if (MODE == RG_FILTER_BWD) {
            const float g = a.A0[(int64_t)row * a.lda0 + k];
            const float y = relu ? a.Yact[(int64_t)row * a.ldy + k] : 1.f;
            t[rr][cc] = y > 0.f ? g : 0.f;
          } else {
            t[rr][cc] = a.A0[(int64_t)row * a.lda0 + k];
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int i = wave + nwaves * rr + ipass;
        if (i >= BM) continue;
        const float ok = rowok[i];
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
          const int kk = c0 + lane + 64 * cc;
          if (kk < KC) {
            const float v = (kk < kc) ? t[rr][cc] * ok : 0.f;
            if (MODE == RG_FILTER_BWD) {
              const int row = row_base + i;
              if (kk < kc && row < a.rows && blockIdx.y == 0 && (a.big_n == 0 || ok != 0.f))
                a.Gp[(int64_t)row * Fout + k0 + kk] = v;   // masked upstream gradient, 0 on pad rows
            }
            As[i * ldA + kk] = v;
          }
        }
      }
    }
    AGCN_STAMP(MODE == RG_PROJECT ? 0 : MODE == RG_FILTER ? 2 : MODE == RG_FILTER_BWD ? 3 : MODE == RG_DENSE ? 6 : 5, 1);
#pragma unroll
    for (int u = 0; u < 4; ++u) b_store(tid + nthr * u, tb[u]);
    for (int base = 4 * nthr; base < btotal; base += 4 * nthr) {
#pragma unroll
      for (int u = 0; u < 4; ++u) tb[u] = b_load(base + tid + nthr * u);
#pragma unroll
      for (int u = 0; u < 4; ++u) b_store(base + tid + nthr * u, tb[u]);
    }
    AGCN_STAMP(MODE == RG_PROJECT ? 0 : MODE == RG_FILTER ? 2 : MODE == RG_FILTER_BWD ? 3 : MODE == RG_DENSE ? 6 : 5, 2);
    __syncthreads();
    AGCN_STAMP(MODE == RG_PROJECT ? 0 : MODE == RG_FILTER ? 2 : MODE == RG_FILTER_BWD ? 3 : MODE == RG_DENSE ? 6 : 5, 3);
    if (my_nt > 0) {
      const float* Aw = As + (rg * 16 + r) * ldA + q;
      const float* Bw = Bs + q * ldB + 16 * cg + r;
      // KC is a multiple of 16: four MFMA k-steps per iteration, operands fetched first
      for (int kk = 0; kk < KC; kk += 16) {
        float av[4], bv[4][NTW];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[u] = Aw[kk + 4 * u];
#pragma unroll
          for (int j = 0; j < NTW; ++j) bv[u][j] = (j < my_nt) ? Bw[(kk + 4 * u) * ldB + 16 * ncg * j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < NTW; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][j], acc[j], 0, 0, 0);
      }
    }
    if (k0 + KC < Kd) __syncthreads();
  }

  AGCN_STAMP(MODE == RG_PROJECT ? 0 : MODE == RG_FILTER ? 2 : MODE == RG_FILTER_BWD ? 3 : MODE == RG_DENSE ? 6 : 5, 4);
  // ---- epilogue -----------------------------------------------------------------------------------
#pragma unroll
  for (int j = 0; j < NTW; ++j) {
    if (j >= my_nt) continue;
    const int c = col_base + 16 * (cg + ncg * j) + r;
    if (c >= N) continue;
    const float bias_c = (MODE == RG_FILTER || (MODE == RG_DENSE && a.bias)) ? a.bias[c] : 0.f;
    const float scale = ((MODE == RG_DENSE || MODE == RG_FILTER_BWD) && a.scale) ? a.scale[0] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row_base + rg * 16 + q * 4 + i;
      if (row >= a.rows) continue;
      if (a.big_n > 0 && rowok[rg * 16 + q * 4 + i] == 0.f) continue;   // row owned by another kernel
      float v = acc[j][i];
      if (MODE == RG_PROJECT) {
        a.Out0[(int64_t)row * a.ldo0 + c] = v;
      } else if (MODE == RG_FILTER) {
        v = v + bias_c;
        if (relu) v = fmaxf(v, 0.f);
        v = v * rowok[rg * 16 + q * 4 + i];   // tf.pad rows (graphconv.py:249-250)
        a.Out0[(int64_t)row * a.ldo0 + c] = v;
      } else if (MODE == RG_DENSE) {
        if (a.scale) v = v * scale;            // (x_res . W) * beta, then + x  (densenet_block.py:116,122)
        if (a.bias) v = v + bias_c;
        if (a.Add) v = v + a.Add[(int64_t)row * a.ldadd + c];
        if (relu) v = fmaxf(v, 0.f);
        v = v * rowok[rg * 16 + q * 4 + i];
        a.Out0[(int64_t)row * a.ldo0 + c] = v;
      } else if (MODE == RG_FILTER_BWD) {
        if (a.scale) v = v * scale;
        const int f = (int)dK.div((uint32_t)c), kx = c - f * K;   // column f*K + k: adjoint of X_k[:, f]
        if (kx == 0) a.Out0[(int64_t)row * a.ldo0 + f] = v;
        else a.Out1[(int64_t)(kx - 1) * a.o1_plane + (int64_t)row * Fin + f] = v;
      } else {
        a.Out0[(int64_t)row * a.ldo0 + c] += v;
      }
    }
  }
  if (MODE == RG_FILTER_BWD && a.ztag_count16 > 0) {
    // granules of the parameter-gradient kernel behind this launch: 16-byte agent-scope stores (spec_kernels.h: zero_tags)
    f32x4* zp = reinterpret_cast<f32x4*>(a.ztag_ptr);
    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int step = (int)(gridDim.x * gridDim.y * blockDim.x);
    for (int e = (int)((blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x); e < a.ztag_count16; e += step)
      asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(zp + e), "v"(z) : "memory");
  }
  AGCN_STAMP(MODE == RG_PROJECT ? 0 : MODE == RG_FILTER ? 2 : MODE == RG_FILTER_BWD ? 3 : MODE == RG_DENSE ? 6 : 5, 5);
}

template <int MODE, int NTW>
static int launch_ntw(RowsGemmArgs a, int gy, int nt, int threads, hipStream_t s) {
  a.nt = nt;
  // largest K chunk whose A and B images fit the LDS budget
  // (multiple of 16: the MFMA loop consumes four k-steps per iteration; at most 192: one lane pass
  // of the staging covers it)
  int kc = (a.Kd + 15) & ~15;
  if (kc > 192) kc = 192;
  while (kc > 32 && (int64_t)BM * ld_frag(kc) + (int64_t)kc * ld_b(16 * nt) + BM > LDS_BUDGET_FLOATS)
    kc = ((kc / 2) + 15) & ~15;
  a.kc = kc;
  const size_t lds = sizeof(float) * ((size_t)BM * ld_frag(kc) + (size_t)kc * ld_b(16 * nt) + BM);
  static const int kProfId[5] = {PK_PROJECT, PK_FILTER, PK_FILTER_BWD, PK_PROJECT_BWD, PK_FILTER};
  if (lds > 64 * 1024) {
    hipError_t e = static_cast<hipError_t>(ensure_dynamic_lds(reinterpret_cast<const void*>(&rows_gemm_kernel<MODE, NTW>), lds));
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(rows_gemm): %s", hipGetErrorString(e));
  }
  const dim3 grid((a.rows + BM - 1) / BM, gy);
  prof_begin(kProfId[MODE], s);
  hipLaunchKernelGGL((rows_gemm_kernel<MODE, NTW>), grid, dim3(threads), lds, s, a);
  prof_end(s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "rows_gemm launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

template <int MODE>
static int launch_mode(const RowsGemmArgs& a, hipStream_t s) {
  // 16 waves stage the operands fastest, but only 4 * (column tiles) of them have an output tile: a product of many
  // rows with one or two column tiles (the point-cloud layers: N = K * Fin = 6, Fin = 3) runs as more, smaller workgroups
  const int nt_all = (a.N + 15) / 16;
  static const int thr_env = [] { const char* e = std::getenv("AGCN_RG_THREADS"); return e ? std::atoi(e) : 0; }();
  int threads = NTHR;
  if (a.rows >= 16384 && nt_all <= 2) threads = 512;
  if (thr_env == 256 || thr_env == 512 || thr_env == 1024) threads = thr_env;
  const int ncg = threads / 256;
  // split N into gy column blocks: at most 3 tiles per wave, and enough blocks that the grid
  // covers the chip (few rows => narrow blocks; the A image is then re-staged from L2 by each)
  const int nt_total = (a.N + 15) / 16;
  const int row_blocks = (a.rows + BM - 1) / BM;
  int want_gy = (128 + row_blocks - 1) / row_blocks;   // measured: ~150 workgroups of 1024 threads is the sweet spot
  if (want_gy < 1) want_gy = 1;
  if (want_gy > nt_total) want_gy = nt_total;
  int nt = (nt_total + want_gy - 1) / want_gy;
  if (nt > 3 * ncg) nt = 3 * ncg;
  const int gy = (nt_total + nt - 1) / nt;
  const int ntw = (nt + ncg - 1) / ncg;
  if (ntw <= 1) return launch_ntw<MODE, 1>(a, gy, nt, threads, s);
  if (ntw == 2) return launch_ntw<MODE, 2>(a, gy, nt, threads, s);
  return launch_ntw<MODE, 3>(a, gy, nt, threads, s);
}

// ------------------------------------------------------------------------------------------------
// Narrow products of MANY rows (the point-cloud first layer: Fin = 3, K * Fin = 6 columns, 32 x 1024 rows).  The
// 64-row MFMA workgroups above spend their time staging a [64 x Fout] image for one sixteen-wide column tile of which
// six columns are real; these two kernels stream the rows instead (one pass over dY / Y / Gp resp. X at full width).

// RG_FILTER_BWD with N = K * Fin <= 8, packed rows: sixteen lanes per row (four rows per wave instruction), each lane
// four consecutive k of a 64-wide pass (float4 loads of dY and Y, float4 store of Gp), the W^T rows in registers,
// partial dot products reduced over the sixteen lanes by a fixed xor tree.
template <int NC, int PASSES>
__global__ __launch_bounds__(256) void rows_filter_bwd_narrow_kernel(RowsGemmArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int l16 = lane & 15, sub = lane >> 4;
  const int Fout = a.Fout, K = a.K, Fin = a.Fin;
  const bool relu = a.activation == AGCN_ACT_RELU;
  if (blockIdx.x == 0)
    for (int e = tid; e < a.zero_count; e += 256) a.zero_ptr[e] = 0;
  f32x4 w[PASSES][NC];
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    const int k = 64 * p + 4 * l16;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      w[p][c] = (k < Fout && c < a.N) ? *reinterpret_cast<const f32x4*>(a.Bm + (int64_t)c * Fout + k)
                                      : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float scale = a.scale ? a.scale[0] : 1.f;
  const int rows_per_wg = 16, wave = tid >> 6;
  for (int base = blockIdx.x * rows_per_wg; base < a.rows; base += gridDim.x * rows_per_wg) {
    const int row = base + 4 * wave + sub;
    const int rc = min(row, a.rows - 1);
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const int k = 64 * p + 4 * l16;
      if (k < Fout) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.A0 + (int64_t)rc * a.lda0 + k);
        f32x4 v = g;
        if (relu) {
          const f32x4 y = *reinterpret_cast<const f32x4*>(a.Yact + (int64_t)rc * a.ldy + k);
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = y[u] > 0.f ? g[u] : 0.f;
        }
        if (row < a.rows) *reinterpret_cast<f32x4*>(a.Gp + (int64_t)row * Fout + k) = v;   // masked upstream gradient
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[c] = __builtin_fmaf(v[u], w[p][c][u], acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 64);
    // lane c of the sixteen writes column c = f * K + k: the adjoint of X_k[:, f]
    float mine = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) mine = (l16 == c) ? acc[c] : mine;
    if (row < a.rows && l16 < a.N) {
      const int f = l16 / K, kx = l16 - f * K;
      const float v = mine * scale;
      if (kx == 0) a.Out0[(int64_t)row * a.ldo0 + f] = v;
      else a.Out1[(int64_t)(kx - 1) * a.o1_plane + (int64_t)row * Fin + f] = v;
    }
  }
}

// RG_PROJECT with Fin <= 4, packed rows: one thread per row, M_L in registers
template <int FIN>
__global__ __launch_bounds__(256) void rows_project_narrow_kernel(RowsGemmArgs a) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  float m[FIN][FIN];
#pragma unroll
  for (int k = 0; k < FIN; ++k)
#pragma unroll
    for (int c = 0; c < FIN; ++c) m[k][c] = a.Bm[k * FIN + c];
  if (row >= a.rows) return;
  float x[FIN];
#pragma unroll
  for (int k = 0; k < FIN; ++k) x[k] = a.A0[(int64_t)row * a.lda0 + k];
#pragma unroll
  for (int c = 0; c < FIN; ++c) {
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < FIN; ++k) z = __builtin_fmaf(x[k], m[k][c], z);   // the MFMA's k-ascending fmaf chain
    a.Out0[(int64_t)row * a.ldo0 + c] = z;
  }
}

static bool narrow_env() {
  static const int off = [] { const char* e = std::getenv("AGCN_ROWS_NARROW"); return (e && e[0] == '0') ? 1 : 0; }();
  return !off;
}
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <class Kern>
static int launch_narrow(Kern kernel, const RowsGemmArgs& a, int grid, int prof_id, hipStream_t s) {
  prof_begin(prof_id, s);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, a);
  prof_end(s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "rows_gemm (narrow) launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

// returns 1 if the product was launched by a narrow kernel, 0 if the general kernel must take it, < 0 on error
static int try_narrow(const RowsGemmArgs& a, int mode, hipStream_t s) {
  if (!narrow_env() || a.row_stride != 0 || a.big_n != 0 || a.rows < 1024) return 0;
  if (mode == RG_PROJECT && a.Kd == a.N && a.N == a.Fin && a.Fin <= 4) {
    const int grid = (a.rows + 255) / 256;
    int rc;
    switch (a.Fin) {
      case 1: rc = launch_narrow(rows_project_narrow_kernel<1>, a, grid, PK_PROJECT, s); break;
      case 2: rc = launch_narrow(rows_project_narrow_kernel<2>, a, grid, PK_PROJECT, s); break;
      case 3: rc = launch_narrow(rows_project_narrow_kernel<3>, a, grid, PK_PROJECT, s); break;
      default: rc = launch_narrow(rows_project_narrow_kernel<4>, a, grid, PK_PROJECT, s); break;
    }
    return rc == AGCN_OK ? 1 : rc;
  }
  if (mode == RG_FILTER_BWD && a.N <= 8 && a.Kd == a.Fout && (a.Fout & 3) == 0 && a.Fout <= 256 && (a.lda0 & 3) == 0 &&
      (a.ldy & 3) == 0 && aligned16(a.A0) && aligned16(a.Yact) && aligned16(a.Gp) && aligned16(a.Bm) && a.Gp != nullptr) {
    const int grid = std::min((a.rows + 15) / 16, 2048);
    const int passes = (a.Fout + 63) / 64;
    int rc;
    if (a.N <= 4) {
      rc = passes == 1 ? launch_narrow(rows_filter_bwd_narrow_kernel<4, 1>, a, grid, PK_FILTER_BWD, s)
         : passes == 2 ? launch_narrow(rows_filter_bwd_narrow_kernel<4, 2>, a, grid, PK_FILTER_BWD, s)
                       : launch_narrow(rows_filter_bwd_narrow_kernel<4, 4>, a, grid, PK_FILTER_BWD, s);
    } else {
      rc = passes == 1 ? launch_narrow(rows_filter_bwd_narrow_kernel<8, 1>, a, grid, PK_FILTER_BWD, s)
         : passes == 2 ? launch_narrow(rows_filter_bwd_narrow_kernel<8, 2>, a, grid, PK_FILTER_BWD, s)
                       : launch_narrow(rows_filter_bwd_narrow_kernel<8, 4>, a, grid, PK_FILTER_BWD, s);
    }
    return rc == AGCN_OK ? 1 : rc;
  }
  return 0;
}

int launch_rows_gemm(const RowsGemmArgs& a, int mode, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    const int nr = try_narrow(a, mode, s);
    if (nr != 0) return nr > 0 ? AGCN_OK : nr;
  }
  switch (mode) {
    case RG_PROJECT: return launch_mode<RG_PROJECT>(a, s);
    case RG_FILTER: return launch_mode<RG_FILTER>(a, s);
    case RG_FILTER_BWD: return launch_mode<RG_FILTER_BWD>(a, s);
    case RG_PROJECT_BWD: return launch_mode<RG_PROJECT_BWD>(a, s);
    case RG_DENSE: return launch_mode<RG_DENSE>(a, s);
  }
  return fail(AGCN_EINVAL, "rows_gemm: bad mode");
}

}  // namespace agcn
