// Packed-row products of the 128-INPUT-FEATURE point-cloud layers (the third and fourth convolution of the reference's
// point-cloud network, models/networks/Point_AGCN.py:66-83; l_n_filters = [64, 128, 128, 64], utils/hyper_parameters.py:8-24):
//   Z = X M_L                                   (graphconv.py:164)
//   Y = act(X_0 W_0 + X_1 W_1 + b)              (graphconv.py:238-251, column f*K + k of the basis <-> row f*K + k of W)
//   Gp = dY act'(Y), g_0 = Gp W_0^T -> dX, g_1 = Gp W_1^T        (the adjoint of the filter)
//   dX += dZ M_L^T                              (paper mode: the adjoint of the projection)
//   dweight / dbias / dM_L row-chunk partials   (A^T B with the ROW index as the MFMA k dimension)
// at tens of thousands of rows (32 clouds x 1024 points).  Until round 4 these ran on the general rows_gemm / one-pass kernels,
// built for a few thousand rows: 29 + 69 + 100 + 40 + 93 = 331 us of the 1.08 ms SGC_LL 128 -> 128 step
// (profiles/r04a_modelnet128_*), 20-30 % of the f32 matrix rate and 1-1.5 TB/s.  The kernels here follow the 64-feature ones of
// cloud_kernels.hip (wave-private / pair-shared A images in LDS read 16 bytes wide, the small B matrix row-major straight from
// L1 / L2 as B[k][4r .. 4r + 3] a 16-k block ahead, lane r owns the output columns 4r .. 4r + 3: 16-byte stores) with one
// change for occupancy: a wave owns 16 rows x 64 OUTPUT COLUMNS, so a 128-wide product is split over two waves that share the
// A image (4 waves per SIMD at 32768 rows instead of 2; 16 accumulator registers per product instead of 32).
// fp32-input MFMA throughout (v_mfma_f32_16x16x4_f32): bit-for-bit an fmaf chain per output element.
// Measured (32768 rows, 128 -> 128, tools/bench_shapes.py --prof): projection 29 -> 19, filter 69 -> 31, filter adjoint 100 -> 42
// (+ two 2.5 us transposes), projection adjoint 40 -> 21, parameter gradients 93 -> 51 + 10 us: 331 -> 181 us.  What is left is
// phase lock, not latency (fragments a block ahead instead of a step ahead: 20.6 -> 19.4 us): 32768 rows are ONE round of
// workgroups, every wave loads (16.8 MB per array: ~4 us at the rate HBM sustains), multiplies (6.8 us of f32 matrix work per
// 128 x 128 product and SIMD) and stores at the same time as every other.  Next step: two row tiles per wave, software-pipelined.
#include "agcn_device.h"
#include "sgc_kernels.h"

#include <cstdlib>

namespace agcn {

namespace {
constexpr int FW = 128;        // input features these kernels are compiled for
constexpr int LDA = FW + 8;    // leading dimension of a 128-wide LDS image (floats): 136 mod 64 = 8 -> conflict-free b128 rows
constexpr int WG_ROWS = 32;    // rows per 256-thread workgroup: 2 row groups of 16 x 2 column halves of 64

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// acc[e] (rows 4q + i, column 4r + e of a 64-column block) += A[16 x KD] . B[KD x 64]
//   A: LDS image [16][lda], read as "row r, columns 16 blk + 4q .. + 3" (one ds_read_b128 per 4 MFMA steps)
//   B: global, row-major, B[k * ldb + 4r .. 4r + 3], the four fragments of a 16-k block requested a block ahead
//   MFMA step j of block blk pairs A[r][16 blk + 4q + j] with B[16 blk + 4q + j][4r + e] (the k index permuted inside 16-blocks).
template <int KD>
__device__ __forceinline__ void rows_mma(f32x4 (&acc)[4], const float* __restrict__ As, int lda, const float* __restrict__ B, int ldb,
                                         int r, int q) {
  // B fragments a whole 16-k block (16 MFMAs = 512 cycles of this wave's matrix work) ahead of their use: with the fragment
  // of step s + 1 requested in front of step s (128 cycles) every step waited for most of an L2 round trip (project 20.6 us)
  constexpr int NBLK = KD / 16;
  const float* Bq = B + (int64_t)(4 * q) * ldb + 4 * r;
  f32x4 bc[4], bn[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bc[j] = *reinterpret_cast<const f32x4*>(Bq + (int64_t)j * ldb);
  f32x4 av = *reinterpret_cast<const f32x4*>(As + r * lda + 4 * q);
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    f32x4 av_next = av;
    if (blk + 1 < NBLK) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bn[j] = *reinterpret_cast<const f32x4*>(Bq + (int64_t)(16 * (blk + 1) + j) * ldb);
      av_next = *reinterpret_cast<const f32x4*>(As + r * lda + 16 * (blk + 1) + 4 * q);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = mfma4(av[j], bc[j][e], acc[e]);
    __builtin_amdgcn_sched_barrier(0);
    if (blk + 1 < NBLK) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bc[j] = bn[j];
      av = av_next;
    }
  }
}

struct Rows8 { f32x4 v[4]; };
// the two waves of a row group load its 16 rows x 128 floats between them: wave half h takes the f32x4 groups
// idx = lane + 64 m + 256 h (row = idx / 32, group = idx % 32): two rows x 512 contiguous bytes per instruction
__device__ __forceinline__ Rows8 load_rows(const float* __restrict__ src, int64_t row0, int ld, int lane, int h) {
  Rows8 g;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int idx = lane + 64 * m + 256 * h;
    g.v[m] = *reinterpret_cast<const f32x4*>(src + (row0 + (idx >> 5)) * ld + 4 * (idx & 31));
  }
  return g;
}
__device__ __forceinline__ void store_rows(float* As, const Rows8& g, int lane, int h) {
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int idx = lane + 64 * m + 256 * h;
    *reinterpret_cast<f32x4*>(As + (idx >> 5) * LDA + 4 * (idx & 31)) = g.v[m];
  }
}

// Z = X . M_L
__global__ __launch_bounds__(256) void rw_project_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ M,
                                                         float* __restrict__ Z) {
  __shared__ __attribute__((aligned(16))) float As_all[2][16 * LDA];
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = wave >> 1, h = wave & 1;
  const int64_t row0 = (int64_t)blockIdx.x * WG_ROWS + 16 * grp;
  float* As = As_all[grp];
  store_rows(As, load_rows(X, row0, ldx, lane, h), lane, h);
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  rows_mma<FW>(acc, As, LDA, M + 64 * h, FW, r, q);
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<f32x4*>(Z + (row0 + 4 * q + i) * FW + 64 * h + 4 * r) = (f32x4){acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
}

// Y = act(X_0 W_0 + X_1 W_1 + b): row f * 2 + k of W belongs to feature f of Chebyshev term k (B row stride 2 Fout, term k at
// + k Fout).  Fout = 128: wave half h takes the output columns 64 h ..; Fout = 64: half 1 only helps with the staging.
template <int FOUT>
__global__ __launch_bounds__(256) void rw_filter_kernel(const float* __restrict__ X0, int ldx, const float* __restrict__ X1,
                                                        const float* __restrict__ W, const float* __restrict__ bias, int relu,
                                                        float* __restrict__ Y, int ldy) {
  __shared__ __attribute__((aligned(16))) float As_all[2][2][16 * LDA];     // [row group][Chebyshev term]
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = wave >> 1, h = wave & 1;
  const int64_t row0 = (int64_t)blockIdx.x * WG_ROWS + 16 * grp;
  const Rows8 g0 = load_rows(X0, row0, ldx, lane, h);
  const Rows8 g1 = load_rows(X1, row0, FW, lane, h);
  store_rows(As_all[grp][0], g0, lane, h);
  store_rows(As_all[grp][1], g1, lane, h);
  __syncthreads();
  if (FOUT == 64 && h == 1) return;
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  rows_mma<FW>(acc, As_all[grp][0], LDA, W + 64 * h, 2 * FOUT, r, q);
  rows_mma<FW>(acc, As_all[grp][1], LDA, W + FOUT + 64 * h, 2 * FOUT, r, q);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + 64 * h + 4 * r);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4 y = (f32x4){acc[0][i] + bv.x, acc[1][i] + bv.y, acc[2][i] + bv.z, acc[3][i] + bv.w};
    if (relu) y = (f32x4){fmaxf(y.x, 0.f), fmaxf(y.y, 0.f), fmaxf(y.z, 0.f), fmaxf(y.w, 0.f)};
    *reinterpret_cast<f32x4*>(Y + (row0 + 4 * q + i) * ldy + 64 * h + 4 * r) = y;
  }
}

// WT[k][c][f] = W[f * K + k][c]  (row-major in the contraction index c of the filter adjoint); with K = 1 and W = M_L: M_L^T
__global__ __launch_bounds__(256) void rw_transpose_kernel(const float* __restrict__ W, float* __restrict__ WT, int Fout, int K) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < K * Fout * FW) {
    const int f = e % FW, kc = e / FW, c = kc % Fout, k = kc / Fout;
    WT[e] = W[(int64_t)(f * K + k) * Fout + c];
  }
}

// Gp = dY . act'(Y) -> workspace;  g_0 = Gp . W_0^T -> dX,  g_1 = Gp . W_1^T -> workspace.  Wave half h: the input-feature
// columns 64 h .. 64 h + 63 of both products (the A image serves both).
template <int FOUT>
__global__ __launch_bounds__(256) void rw_filter_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y, int ldy,
                                                            int relu, const float* __restrict__ WT, float* __restrict__ Gp,
                                                            float* __restrict__ dX, int ldx, float* __restrict__ g1) {
  constexpr int LDG = FOUT + 8;
  __shared__ __attribute__((aligned(16))) float As_all[2][16 * LDG];
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = wave >> 1, h = wave & 1;
  const int64_t row0 = (int64_t)blockIdx.x * WG_ROWS + 16 * grp;
  float* As = As_all[grp];
  constexpr int F4 = FOUT / 4, PER = 16 * F4 / 128;          // f32x4 groups per row; per lane of the wave pair
#pragma unroll
  for (int m = 0; m < PER; ++m) {
    const int idx = lane + 64 * m + 64 * PER * h;
    const int row = idx / F4, c4 = idx - row * F4;
    const f32x4 d = *reinterpret_cast<const f32x4*>(dY + (row0 + row) * ldy + 4 * c4);
    const f32x4 y = *reinterpret_cast<const f32x4*>(Y + (row0 + row) * ldy + 4 * c4);
    f32x4 g = d;
    if (relu) g = (f32x4){y.x > 0.f ? d.x : 0.f, y.y > 0.f ? d.y : 0.f, y.z > 0.f ? d.z : 0.f, y.w > 0.f ? d.w : 0.f};
    *reinterpret_cast<f32x4*>(Gp + (row0 + row) * FOUT + 4 * c4) = g;
    *reinterpret_cast<f32x4*>(As + row * LDG + 4 * c4) = g;
  }
  __syncthreads();
#pragma unroll 1
  for (int k = 0; k < 2; ++k) {
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    rows_mma<FOUT>(acc, As, LDG, WT + (int64_t)k * FOUT * FW + 64 * h, FW, r, q);
    float* dst = k == 0 ? dX : g1;
    const int ld = k == 0 ? ldx : FW;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(dst + (row0 + 4 * q + i) * ld + 64 * h + 4 * r) = (f32x4){acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
  }
}

// dX += dZ . M_L^T   (MT[g][f] = M_L[f][g]: rw_transpose_kernel)
__global__ __launch_bounds__(256) void rw_project_bwd_kernel(const float* __restrict__ dZ, const float* __restrict__ MT,
                                                             float* __restrict__ dX, int ldx) {
  __shared__ __attribute__((aligned(16))) float As_all[2][16 * LDA];
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int grp = wave >> 1, h = wave & 1;
  const int64_t row0 = (int64_t)blockIdx.x * WG_ROWS + 16 * grp;
  float* As = As_all[grp];
  store_rows(As, load_rows(dZ, row0, FW, lane, h), lane, h);
  f32x4 old[4];                                    // the dX values this wave adds to: in flight behind the product
#pragma unroll
  for (int i = 0; i < 4; ++i) old[i] = *reinterpret_cast<const f32x4*>(dX + (row0 + 4 * q + i) * ldx + 64 * h + 4 * r);
  __syncthreads();
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  rows_mma<FW>(acc, As, LDA, MT + 64 * h, FW, r, q);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 v = (f32x4){old[i].x + acc[0][i], old[i].y + acc[1][i], old[i].z + acc[2][i], old[i].w + acc[3][i]};
    *reinterpret_cast<f32x4*>(dX + (row0 + 4 * q + i) * ldx + 64 * h + 4 * r) = v;
  }
}

// Row-chunk partials of the parameter gradients, the layout of cloud_param_partial_kernel (cloud_kernels.hip) widened to 128
// input features: workgroup (chunk, y = 2 fh + kx, z) takes the 64 input features 64 fh .. of Chebyshev term kx against the
// 64-column slice z of Gp (kx = 0, fh = 0: also that slice of the bias) -- or, the extra slices z = Fout / 64 + gh in paper mode,
// X[:, 64 fh ..] against dZ[:, 64 gh ..] (dM_L; kx = 0 only).  Partials: [chunk][dweight | dbias | dM_L], summed in fixed order by
// cloud_param_reduce_kernel.
template <int FOUT>
__global__ __launch_bounds__(256) void rw_param_partial_kernel(ParamGradArgs a) {
  constexpr int PR = 64;   // rows per piece
  __shared__ __attribute__((aligned(16))) float Xs[PR * 64], Gps[PR * 64];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NH = FOUT / 64;
  const int chunk = blockIdx.x, kx = blockIdx.y & 1, fh = blockIdx.y >> 1, z = blockIdx.z;
  const int r_begin = chunk * a.chunk_rows, r_end = min(a.rows, r_begin + a.chunk_rows);
  const bool with_m = z >= NH;          // the extra z slices: X^T dZ instead of X_k^T Gp (only against X_0)
  if (with_m && kx != 0) return;
  const int gh = z - NH, h = z;
  const bool with_b = kx == 0 && fh == 0 && !with_m;
  f32x4 accW[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) accW[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};
  struct Piece { f32x4 xa[4], gp[4]; };
  const float* Asrc = (kx == 0 ? a.X : a.sv_X) + 64 * fh;
  const int lda = kx == 0 ? a.ldx : FW;
  const float* Bsrc = with_m ? a.dZ + 64 * gh : a.Gp + 64 * h;
  const int ldb = with_m ? FW : FOUT;
  auto load_piece = [&](int p0) {        // rows p0 .. p0 + 63 (clamped to the batch; rows are multiples of 64)
    Piece g;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int id = tid + 256 * m;
      const int64_t row = min(p0 + (id >> 4), a.rows - 1);
      g.xa[m] = *reinterpret_cast<const f32x4*>(Asrc + row * lda + 4 * (id & 15));
      g.gp[m] = *reinterpret_cast<const f32x4*>(Bsrc + row * ldb + 4 * (id & 15));
    }
    return g;
  };
  auto store_piece = [&](const Piece& g) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int id = tid + 256 * m;
      *reinterpret_cast<f32x4*>(Xs + 4 * id) = g.xa[m];
      *reinterpret_cast<f32x4*>(Gps + 4 * id) = g.gp[m];
    }
  };
  Piece pre = load_piece(r_begin);
#pragma unroll 1
  for (int p0 = r_begin; p0 < r_end; p0 += PR) {
    __syncthreads();
    store_piece(pre);
    __syncthreads();
    if (p0 + PR < r_end) pre = load_piece(p0 + PR);
    const int nk = min(PR, r_end - p0) >> 2;       // k-steps of 4 rows in this piece
#pragma unroll 4
    for (int u = 0; u < nk; ++u) {
      const int row = 4 * u + q;
      const f32x4 xa = *reinterpret_cast<const f32x4*>(Xs + row * 64 + 4 * r);
      const float xw = wave == 0 ? xa.x : wave == 1 ? xa.y : wave == 2 ? xa.z : xa.w;
      const f32x4 gp = *reinterpret_cast<const f32x4*>(Gps + row * 64 + 4 * r);
#pragma unroll
      for (int e = 0; e < 4; ++e) accW[e] = mfma4(xw, gp[e], accW[e]);
      if (with_b && wave == 0) bsum += gp;
    }
  }
  float* out = a.part + (int64_t)chunk * a.part_stride;
  // accumulator element i of tile (A tile w, B tile e'): f = 64 fh + 4 (4q + i) + w, c = 64 h + 4r + e'
  if (!with_m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = 64 * fh + 16 * q + 4 * i + wave;
      *reinterpret_cast<f32x4*>(out + (int64_t)(f * 2 + kx) * FOUT + 64 * h + 4 * r) = (f32x4){accW[0][i], accW[1][i], accW[2][i], accW[3][i]};
    }
  }
  if (with_b && wave == 0) {
    f32x4 t = bsum;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      t.x += __shfl_xor(t.x, o, 64); t.y += __shfl_xor(t.y, o, 64); t.z += __shfl_xor(t.z, o, 64); t.w += __shfl_xor(t.w, o, 64);
    }
    if (q == 0) *reinterpret_cast<f32x4*>(out + (int64_t)2 * FW * FOUT + 64 * h + 4 * r) = t;
  }
  if (with_m) {
    float* outM = out + (int64_t)2 * FW * FOUT + FOUT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = 64 * fh + 16 * q + 4 * i + wave;
      *reinterpret_cast<f32x4*>(outM + (int64_t)f * FW + 64 * gh + 4 * r) = (f32x4){accW[0][i], accW[1][i], accW[2][i], accW[3][i]};
    }
  }
}

template <class K, class... A>
int launch_rw(K kernel, dim3 grid, int prof_id, const char* what, void* stream, A... args) {
  if (grid.x == 0) return AGCN_OK;
  prof_begin(prof_id, stream);
  hipLaunchKernelGGL(kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), args...);
  prof_end(stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s launch: %s", what, hipGetErrorString(e));
  return AGCN_OK;
}
bool rw_env() {
  static const int off = [] { const char* e = std::getenv("AGCN_ROWS_WIDE"); return (e && e[0] == '0') ? 1 : 0; }();
  return !off;
}
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
}  // namespace

// 128 input features, Fout 64 or 128, packed rows in multiples of 32, 16-byte aligned rows (AGCN_ROWS_WIDE=0: the general
// rows_gemm kernels instead, for A/B measurements)
bool rows_wide_ok(int rows, int Fin, int Fout, int K, int ldx, int ldy) {
  return rw_env() && Fin == FW && K == 2 && (Fout == 64 || Fout == 128) && (rows % WG_ROWS) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0;
}

int launch_rows_wide_project(const float* X, int ldx, const float* M, float* Z, int rows, void* stream) {
  if (!al16(X) || !al16(M) || !al16(Z)) return fail(AGCN_EINVAL, "rows_wide_project: unaligned buffer");
  return launch_rw(rw_project_kernel, dim3((unsigned)(rows / WG_ROWS)), PK_PROJECT, "rw_project", stream, X, ldx, M, Z);
}

int launch_rows_wide_filter(const float* X0, int ldx, const float* X1, const float* W, const float* bias, int relu, float* Y, int ldy,
                            int Fout, int rows, void* stream) {
  if (!al16(X0) || !al16(X1) || !al16(W) || !al16(bias) || !al16(Y)) return fail(AGCN_EINVAL, "rows_wide_filter: unaligned buffer");
  const dim3 grid((unsigned)(rows / WG_ROWS));
  return Fout == 128 ? launch_rw(rw_filter_kernel<128>, grid, PK_FILTER, "rw_filter", stream, X0, ldx, X1, W, bias, relu, Y, ldy)
                     : launch_rw(rw_filter_kernel<64>, grid, PK_FILTER, "rw_filter", stream, X0, ldx, X1, W, bias, relu, Y, ldy);
}

// WT: 2 * Fout * 128 floats of workspace, MT (paper mode, may be nullptr): 128 * 128 floats -- both (re)built here
int launch_rows_wide_filter_bwd(const float* dY, const float* Y, int ldy, int relu, const float* W, const float* M, float* WT, float* MT,
                                float* Gp, float* dX, int ldx, float* g1, int Fout, int rows, void* stream) {
  if (!al16(dY) || !al16(Y) || !al16(WT) || !al16(Gp) || !al16(dX) || !al16(g1)) return fail(AGCN_EINVAL, "rows_wide_filter_bwd: unaligned buffer");
  int rc = launch_rw(rw_transpose_kernel, dim3((unsigned)((2 * Fout * FW + 255) / 256)), PK_FILTER_BWD, "rw_transpose", stream, W, WT, Fout, 2);
  if (rc != AGCN_OK) return rc;
  if (MT != nullptr) {
    rc = launch_rw(rw_transpose_kernel, dim3((unsigned)((FW * FW + 255) / 256)), PK_FILTER_BWD, "rw_transpose", stream, M, MT, FW, 1);
    if (rc != AGCN_OK) return rc;
  }
  const dim3 grid((unsigned)(rows / WG_ROWS));
  return Fout == 128 ? launch_rw(rw_filter_bwd_kernel<128>, grid, PK_FILTER_BWD, "rw_filter_bwd", stream, dY, Y, ldy, relu, (const float*)WT, Gp, dX, ldx, g1)
                     : launch_rw(rw_filter_bwd_kernel<64>, grid, PK_FILTER_BWD, "rw_filter_bwd", stream, dY, Y, ldy, relu, (const float*)WT, Gp, dX, ldx, g1);
}

int launch_rows_wide_project_bwd(const float* dZ, const float* MT, float* dX, int ldx, int rows, void* stream) {
  if (!al16(dZ) || !al16(MT) || !al16(dX)) return fail(AGCN_EINVAL, "rows_wide_project_bwd: unaligned buffer");
  return launch_rw(rw_project_bwd_kernel, dim3((unsigned)(rows / WG_ROWS)), PK_PROJECT_BWD, "rw_project_bwd", stream, dZ, MT, dX, ldx);
}

// the 128-feature flavour of launch_cloud_param_partials (cloud_kernels.hip): the caller runs the fixed-order reduce behind it
int launch_rows_wide_param_partials(const ParamGradArgs& a, void* stream) {
  const unsigned zm = a.paper ? 2u : 0u;   // the X^T dZ slices (two column halves of dM_L)
  return a.Fout == 128 ? launch_rw(rw_param_partial_kernel<128>, dim3((unsigned)a.n_chunks, 4, 2 + zm), PK_PARAM_PARTIAL, "rw_param_partial", stream, a)
                       : launch_rw(rw_param_partial_kernel<64>, dim3((unsigned)a.n_chunks, 4, 1 + zm), PK_PARAM_PARTIAL, "rw_param_partial", stream, a);
}

}  // namespace agcn
