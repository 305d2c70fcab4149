// Device-side building blocks shared by the gfx950 kernels.
//   * wave = 64 lanes; workgroups are 256 threads (LDS-tile mode) or 1024 (global-tile mode)
//   * f32-input MFMA (v_mfma_f32_16x16x4_f32): bit-for-bit a k-ordered fmaf chain, so the
//     feature contractions keep exact fp32 semantics
//   * all reductions use a fixed order (no atomics): results are reproducible run to run
#ifndef AGCN_DEVICE_H_
#define AGCN_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "agcn_internal.h"

// Diagnostic build only (-DAGCN_STAMPS): per-workgroup 100 MHz wall-clock stamps at phase
// boundaries, written to a buffer that nothing else reads.  No stamp exists in the product build.
#ifdef AGCN_STAMPS
extern __device__ unsigned long long g_agcn_stamps[];
#define AGCN_STAMP(kernel_slot, i)                                                              \
  do {                                                                                          \
    if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.y == 0)                               \
      g_agcn_stamps[((kernel_slot) * 1024 + blockIdx.x) * 16 + (i)] = wall_clock64();           \
  } while (0)
// ... the same from thread `t` of the block (e.g. the first lane of the last wave: how far behind wave 0 it runs)
#define AGCN_STAMP_T(kernel_slot, i, t)                                                         \
  do {                                                                                          \
    if ((int)threadIdx.x == (t) && blockIdx.x < 1024 && blockIdx.y == 0)                        \
      g_agcn_stamps[((kernel_slot) * 1024 + blockIdx.x) * 16 + (i)] = wall_clock64();           \
  } while (0)
#else
#define AGCN_STAMP(kernel_slot, i) do {} while (0)
#define AGCN_STAMP_T(kernel_slot, i, t) do {} while (0)
#endif

namespace agcn {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));   // packed fp32 math (v_pk_add_f32 / v_pk_fma_f32)

// Exact x / d for x * d < 2^32 (d >= 1) by one mul-hi.
struct FastDiv {
  uint32_t m, d;
  __host__ __device__ explicit FastDiv(uint32_t d_) : m(0), d(d_) {
    if (d_ > 1) m = (uint32_t)(((1ull << 32) + d_ - 1) / d_);
  }
  __device__ __forceinline__ uint32_t div(uint32_t x) const { return d == 1 ? x : __umulhi(x, m); }
};

__host__ __device__ __forceinline__ int round4(int v) { return (v + 3) & ~3; }


// x += v on an LDS cell as ONE ds_add_f32 without a return value: a handful of them issue back to back, where `cell += v` is a read, a
// wait for it and a write per term (the CSR scatters of L_orig: six dependent LDS round trips per row).  Same single fp32 addition
// per term, terms of one lane applied in program order: the same bits.
__device__ __forceinline__ void lds_add(float* cell, float v) {
  __hip_atomic_fetch_add(cell, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the value of another lane by a DPP control word (0xB1: lane ^ 1, 0x4E: lane ^ 2 inside quads; 0x141: the mirrored half row
// (lane 7 - l of every 8); 0x140: the mirrored row (15 - l of every 16)): a register move, not an LDS permute.  EXEC all ones.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// Sum over the 64 lanes: DPP row operations (lane ^ 1, lane ^ 2, the mirrored half row, the mirrored row: every lane then holds its
// 16-lane row's sum, the same bits in all of them) and four lane reads; EVERY lane returns the total.  Fixed order: bit-reproducible.
// (Until round 6 a __shfl_down tree: six dependent ds_bpermute round trips, ~0.15 us per sum on a phase's critical path.)
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
  return (r0 + r1) + (r2 + r3);
}

// Sum over the B per-graph (or per-slab) scalar partials pg[2 b + which] by ONE wave, in a fixed order: 16 loads in flight per
// round (a plain strided loop waits out one memory round trip per 64 entries: ~16 us at 2048 slabs).
__device__ __forceinline__ float pg_wave_sum(const float* pg, int B, int which, int lane) {
  float t = 0.f;
  for (int b0 = 0; b0 < B; b0 += 64 * 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int b = b0 + 64 * u + lane;
      v[u] = b < B ? pg[2 * b + which] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) t += v[u];
  }
  return wave_sum(t);
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

// Bulk copy with U independent loads in flight per thread: at these sizes every kernel is bound by
// memory LATENCY, so loops are written as "issue all the loads, then consume".
//   ld(idx) -> value; must be safe for every idx in [0, total) and is called UNCONDITIONALLY with a
//   clamped index (so the compiler can issue the U loads back to back);  st(idx, v) consumes it.
template <int U, class LD, class ST>
__device__ __forceinline__ void bulk_for(int total, LD ld, ST st) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (total <= 0) return;
  for (int base = tid; base < total; base += U * nthr) {
    float t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) t[u] = ld(min(base + u * nthr, total - 1));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = base + u * nthr;
      if (idx < total) st(idx, t[u]);
    }
  }
}

// 2-D variant over a [nrows x ncols] index space, lanes along the column index.  The (row, col)
// pair is advanced incrementally (no division per element).
//   ld(i, c) -> value (called unconditionally with an in-range pair);  st(i, c, v).
template <int U, class LD, class ST>
__device__ __forceinline__ void bulk_for2d(int nrows, int ncols, LD ld, ST st) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int total = nrows * ncols;
  if (total <= 0) return;
  const FastDiv dc((uint32_t)ncols);
  const int di = (int)dc.div((uint32_t)nthr), dcn = nthr - di * ncols;   // step of one stride
  int i = (int)dc.div((uint32_t)tid), c = tid - i * ncols;
  for (int base = tid; base < total; base += U * nthr) {
    float t[U];
    int ii[U], cc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ii[u] = i; cc[u] = c;
      const bool in = i < nrows;
      t[u] = ld(in ? i : nrows - 1, in ? c : 0);
      c += dcn; i += di;
      if (c >= ncols) { c -= ncols; ++i; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (ii[u] < nrows) st(ii[u], cc[u], t[u]);
  }
}

// Copy a row-major global matrix into an LDS image with (almost) no index arithmetic: wave w takes
// rows w, w + nwaves, ..; lanes run along the columns in passes of 64.  Up to 4 rows x CP column
// passes of loads are in flight per thread.  Image rows nrows..prow and columns ncols..pcol are
// zero-filled.  map_row(r) gives the SOURCE row of image row r (identity for plain copies), or a
// negative value for an image row that must be zero.
template <int CP, class MAP>
__device__ __forceinline__ void stage_matrix(float* img, int ld, const float* src, int64_t src_ld, int nrows,
                                             int ncols, int prow, int pcol, MAP map_row) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int r0 = wave; r0 < prow; r0 += 4 * nw) {
    float t[4][CP];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = min(r0 + u * nw, nrows - 1);
      const float* rp = src + (int64_t)max(map_row(r), 0) * src_ld;
#pragma unroll
      for (int cp = 0; cp < CP; ++cp) t[u][cp] = rp[min(lane + 64 * cp, ncols - 1)];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = r0 + u * nw;
      if (r < prow) {
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
          const int c = lane + 64 * cp;
          if (c < pcol) img[r * ld + c] = (r < nrows && c < ncols && map_row(r) >= 0) ? t[u][cp] : 0.f;
        }
      }
    }
  }
}

// tile[i][col] += val for the CSR rows of one graph (one thread per row; entries of a row are
// fetched four at a time so their latencies overlap).
__device__ __forceinline__ void csr_scatter_add(float* tile, int ldt, int n, const int32_t* rp,
                                                const int32_t* col, const float* val) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int p0 = rp[i], p1 = rp[i + 1];
    for (int p = p0; p < p1; p += 4) {
      int c[4];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pp = min(p + u, p1 - 1);
        c[u] = col[pp];
        v[u] = val[pp];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p + u < p1) tile[i * ldt + c[u]] += v[u];
    }
  }
}

// relu(x) - alpha * relu(-x)   (models/operators/model_operatos.py:534-558, tensor alpha)
__device__ __forceinline__ float lrelu(float x, float alpha) {
  return fmaxf(x, 0.f) - alpha * fmaxf(-x, 0.f);
}
// TF relu gradient: 1 where the argument is > 0  =>  [x > 0] + alpha [x < 0]
__device__ __forceinline__ float lrelu_grad(float x, float alpha) {
  return x > 0.f ? 1.f : (x < 0.f ? alpha : 0.f);
}

// dd_k = -1/2 s_k^3 ds_k (s = d^-1/2).  A node whose similarities all underflow has d_k = FLT_TRUE_MIN, s_k ~ 2.7e22 and
// s_k^3 = inf while ds_k = 0: inf * 0 would poison dM_L of the whole batch with NaN.  Such a node is cut out of the
// degree gradient instead (dd_k = 0 whenever s_k^3 is not finite or ds_k == 0); the oracle applies the same rule.
__device__ __forceinline__ float degree_grad(float sk, float dsk) {
  const float s3 = sk * sk * sk;
  return (dsk == 0.f || !(fabsf(s3) <= 3.402823466e+38f)) ? 0.f : -0.5f * s3 * dsk;
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
#pragma unroll 4
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

// B operand straight from global memory (L2-resident parameters), no LDS image:
//   B[k][c] = Bg[k * krow + c] for k < kvalid, c < cvalid; addresses are clamped, the A image's zero padding
//   (columns >= kvalid) cancels the clamped rows and out-of-range columns are never stored.
// The loads of the block two ahead are issued before the MFMAs of the current 16-k block.
__device__ __forceinline__ void tile_gemm_gbx(f32x4& acc, const float* A, int lda, int row0, const float* Bp,
                                              int kstride, int klast, int kdim) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const float* Ap = A + (row0 + r) * lda + q;
  float b0[4], b1[4];
  auto ldb = [&](int k0, float (&dst)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) dst[u] = Bp[(int64_t)min(k0 + 4 * u + q, klast) * kstride];
  };
  auto step = [&](int k0, const float (&bv)[4]) {
    float av[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) av[u] = Ap[k0 + 4 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  };
  float b2[4];
  ldb(0, b0);
  if (16 < kdim) ldb(16, b1);
  // the loads of the block TWO ahead are issued before the MFMAs of the current one
#pragma unroll 1
  for (int k0 = 0; k0 < kdim; k0 += 48) {
    if (k0 + 32 < kdim) ldb(k0 + 32, b2);
    step(k0, b0);
    if (k0 + 16 < kdim) {
      if (k0 + 48 < kdim) ldb(k0 + 48, b0);
      step(k0 + 16, b1);
    }
    if (k0 + 32 < kdim) {
      if (k0 + 64 < kdim) ldb(k0 + 64, b1);
      step(k0 + 32, b2);
    }
  }
}
// row-major B in global memory: B[k][c] = Bg[k * krow + c]
__device__ __forceinline__ void tile_gemm_gb(f32x4& acc, const float* A, int lda, int row0, const float* Bg,
                                             int krow, int kvalid, int col0, int cvalid, int kdim) {
  const int r = threadIdx.x & 15;
  tile_gemm_gbx(acc, A, lda, row0, Bg + min(col0 + r, cvalid - 1), krow, kvalid - 1, kdim);
}
// Transposed B with k contiguous in memory, 16-byte loads: lane (r, q) fetches Brow[k0 + 4q .. 4q + 3] in one
// instruction (the 16 rows of a wave-load are 64 contiguous bytes each: every fetched line is fully used; the
// dword form above touches the same 16 lines four times).  MFMA step j therefore pairs B[k0 + 4q + j] with the A
// element at image position k0 + 4j + q: the A image must be stored k-permuted inside every 16-block (perm16).
__device__ __forceinline__ int perm16(int c) { return (c & ~15) | ((c & 3) << 2) | ((c >> 2) & 3); }
__device__ __forceinline__ void tile_gemm_gbt4(f32x4& acc, const float* A, int lda, int row0, const float* Brow,
                                               int kvalid, int kdim) {
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const float* Ap = A + (row0 + r) * lda + q;
  const float* Bq = Brow + 4 * q;
  const int klast = kvalid - 1;
  // full 16-blocks of 16-byte aligned rows: one 16-byte load; otherwise (tail block, rows of e.g. 75 floats):
  // clamped dword loads of the same elements (the A image's zero padding cancels the clamped ones)
  const bool al = (reinterpret_cast<uintptr_t>(Brow) & 15) == 0;   // wave-uniform when the row stride is 16-byte
  auto ldb = [&](int k0, float (&dst)[4]) {
    if (al && k0 + 16 <= kvalid) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(Bq + k0);
#pragma unroll
      for (int u = 0; u < 4; ++u) dst[u] = t[u];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) dst[u] = Brow[min(k0 + 4 * q + u, klast)];
    }
  };
  auto step = [&](int k0, const float (&bv)[4]) {
    float av[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) av[u] = Ap[k0 + 4 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  };
  float b0[4], b1[4];
  float b2[4];
  ldb(0, b0);
  if (16 < kdim) ldb(16, b1);
  // the loads of the block TWO ahead are issued before the MFMAs of the current one
#pragma unroll 1
  for (int k0 = 0; k0 < kdim; k0 += 48) {
    if (k0 + 32 < kdim) ldb(k0 + 32, b2);
    step(k0, b0);
    if (k0 + 16 < kdim) {
      if (k0 + 48 < kdim) ldb(k0 + 48, b0);
      step(k0 + 16, b1);
    }
    if (k0 + 32 < kdim) {
      if (k0 + 64 < kdim) ldb(k0 + 64, b1);
      step(k0 + 32, b2);
    }
  }
}


}  // namespace agcn
#endif  // AGCN_DEVICE_H_
