// Packed-row products at point-cloud sizes (tens of thousands of rows) on the BF16 matrix pipe at fp32 accuracy (bf16_split.h:
// every fp32 operand split exactly into three bf16 pieces, six v_mfma_f32_16x16x32_bf16 per 32 k):
//   Z = X M_L                                   (graphconv.py:164)
//   Y = act(X_0 W_0 + X_1 W_1 + b)              (graphconv.py:238-251, column f*K + k of the basis <-> row f*K + k of W)
//   Gp = dY act'(Y), g_0 = Gp W_0^T -> dX, g_1 = Gp W_1^T        (the adjoint of the filter)
//   dX += dZ M_L^T                              (paper mode: the adjoint of the projection)
//   dweight / dbias / dM_L row-chunk partials   (A^T B with the ROW index as the MFMA k dimension)
// for the layers of the reference's point-cloud network (models/networks/Point_AGCN.py:66-83: 64- and 128-feature layers).
// They replace the fp32-MFMA kernels of rows_wide.hip / cloud_kernels.hip (kept behind AGCN_ROWS_SPLIT=0 for A/B runs), which
// ran at 20-30 % of the f32 matrix rate: one round of workgroups, every wave loading, multiplying (6.8 us of f32 matrix work
// per 128 x 128 product and SIMD) and storing in lock step.  Here the matrix work of such a product is 2.6 us.
//
// Layout of a product C[rows x N] = A[rows x K] . B[K x N]:
//  * B (a parameter matrix, K x N <= 256 x 256) is split ONCE per call by rs_presplit_kernel into an image in MFMA operand
//    order: for k-block kb (32 k) and column tile ct, piece p, lane (r, q): the eight bf16 of B[32 kb + {4q .. 4q + 3, 16 + 4q .. 16 + 4q + 3}][col(ct, r)],
//    col(ct, r) = 64 (ct / 4) + 4r + ct % 4 -- so that a lane's four tiles of a 64-column group are four CONSECUTIVE output
//    columns (16-byte stores) -- 1 KB per (kb, ct, p), read with one coalesced 16-byte load per lane.  Transposes (W^T, M_L^T)
//    are strides of the split, not launches of their own.
//  * a workgroup (4 waves) takes 128 rows x NT column tiles: the image of its column block (KB x NT x 3 KB, at most 96 KB)
//    is copied into LDS once; a wave owns 32 rows (two row tiles), loads its A rows straight from global memory in operand
//    order (row r, 8 consecutive k: 32 bytes per lane, every 128-byte line of a row used whole by the wave's four k-blocks),
//    splits them in registers, and runs 12 MFMAs per (kb, ct) against B pieces read from
//    LDS (3 x ds_read_b128) -- 64 B / clk of LDS traffic per CU.
// Results differ from the fp32-MFMA kernels by fp32 summation order only; bit-reproducible run to run.
#include "agcn_device.h"
#include "bf16_split.h"
#include "sgc_kernels.h"

#include <cstdlib>

namespace agcn {

namespace {

constexpr int WG_ROWS = 128;     // rows per workgroup: 4 waves x 2 row tiles of 16
constexpr int TILE16 = 192;      // u32x4 per (k-block, column tile): 3 pieces x 64 lanes

// B[k][c] = src[k ks + c cs], K x N; its tiles (kb, ct) land at img + dst + (kb nt_dst + ct_off + ct) x 192 u32x4: several jobs
// can fill column ranges (ct_off, a multiple of 4) or k ranges (dst) of ONE image [KB][nt_dst]
struct SplitJob { const float* src; int32_t ks, cs, K, N; uint32_t dst; int32_t nt_dst, ct_off; };
struct SplitJobs { SplitJob j[4]; };

__global__ __launch_bounds__(256) void rs_presplit_kernel(SplitJobs jobs, u32x4* __restrict__ img) {
  const SplitJob jb = jobs.j[blockIdx.y];
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int nt = jb.N >> 4;
  if (t >= (jb.K >> 5) * nt * 64) return;
  const int lane = t & 63, tile = t >> 6, ct = tile % nt, kb = tile / nt;
  const int r = lane & 15, q = lane >> 4;
  const int col = 64 * (ct >> 2) + 4 * r + (ct & 3);
  // the lane's eight k of a k-block: 4q .. 4q + 3 and 16 + 4q .. 16 + 4q + 3 (rs_rows_kernel: the A rows are then loaded as two
  // instructions of 64 CONTIGUOUS bytes per row; the order of the contraction index is free as long as A and B agree)
  const float* p = jb.src + (int64_t)(32 * kb + 4 * q) * jb.ks + (int64_t)col * jb.cs;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)(j + (j >= 4 ? 12 : 0)) * jb.ks];
  const Pieces P = split3x8((f32x4){v[0], v[1], v[2], v[3]}, (f32x4){v[4], v[5], v[6], v[7]});
  u32x4* dst = img + jb.dst + (size_t)(kb * jb.nt_dst + jb.ct_off + ct) * TILE16 + lane;
  dst[0] = P.h; dst[64] = P.m; dst[128] = P.l;
}

enum { RS_PLAIN = 0, RS_FILTER = 1, RS_FILTER_BWD = 2 };

struct RsArgs {
  const float* A0; int32_t lda0;      // rows of term 0
  const float* A1; int32_t lda1;      // rows of term 1 (RS_FILTER: X_1), k-blocks kb >= kb_term
  int32_t kb_term;                    // k-blocks per term
  const float* Yact; int32_t ldy; int32_t relu;   // RS_FILTER_BWD: A = A0 (dY) where Yact > 0 (relu) ; RS_FILTER: relu on the output
  float* Gp; int32_t ldgp;            // RS_FILTER_BWD: the masked A rows, stored for the parameter-gradient kernels
  const float* bias;                  // RS_FILTER
  float* Out0; int32_t ldo0;          // output columns [0, n_split)
  float* Out1; int32_t ldo1;          // output columns [n_split, N) (RS_FILTER_BWD: g_1)
  int32_t n_split;
  int32_t accumulate;                 // RS_PLAIN: Out0 += product (dX += dZ M_L^T)
  const float* scale;                 // RS_FILTER / RS_FILTER_BWD: optional device scalar on the product (DenseBlockEnd's beta)
  const float* Add; int32_t ldadd;    // RS_FILTER: optional addend [rows][ldadd] (the dense layers' running sum / residual)
  int32_t* zero_ptr; int32_t zero_count;   // RS_FILTER_BWD: arrival counters of the parameter-gradient kernel behind it, zeroed by workgroup 0
  int32_t rows;
  const u32x4* Bimg; int32_t nt_total;   // image of the whole B: [KB][nt_total][3][64] u32x4
};

// KB: 32-k blocks of the whole contraction; NT: column tiles of a column block (4 | 8: 64 | 128 columns); NCB: column blocks
// of the product, taken one after the other by the same workgroup (its A rows stay in registers).
// The B image streams through LDS in CHUNKS of 16 tiles (48 KB: KBC = 16 / NT k-blocks of one column block), double buffered:
// chunk c + 1 travels global -> registers while chunk c feeds the matrix cores, one barrier per chunk.  (With the whole image
// of a column block resident -- 96 KB -- its copy was 2 us of every launch in front of the first product, and a product of
// K = 256 or N = 256 needed two workgroups per row block, i.e. two rounds of workgroups.)
template <int KB, int NT, int NCB, int MODE>
__global__ __launch_bounds__(256) void rs_rows_kernel(RsArgs a) {
  constexpr int KBC = (16 / NT) < KB ? (16 / NT) : KB;     // k-blocks per chunk
  constexpr int KC = KB / KBC;                             // chunks per column block
  constexpr int NCH = NCB * KC;
  constexpr int CHUNK = KBC * NT * TILE16;                 // u32x4 per chunk (at most 16 tiles: 48 KB)
  constexpr int PER = CHUNK / 256;                         // u32x4 per thread and chunk
  constexpr int RUN = NT * TILE16;                         // u32x4 of one k-block of a column block
  static_assert(KB % KBC == 0, "whole chunks");
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_lds[];
  u32x4* Bs = reinterpret_cast<u32x4*>(rs_lds);            // [2][KBC][NT][3][64]
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * WG_ROWS + 32 * wave;
  const int64_t last = (int64_t)a.rows - 1;

  // this wave's A rows of k-block kb, in operand order (row r of the tile, k = 32 kb + 4q .. + 3 and 32 kb + 16 + 4q .. + 3)
  struct ARaw { f32x4 lo[2], hi[2]; };
  auto load_a = [&](int kb) {
    ARaw g;
    const bool t1 = MODE == RS_FILTER && kb >= a.kb_term;
    const float* base = t1 ? a.A1 : a.A0;
    const int ld = t1 ? a.lda1 : a.lda0;
    const int kk = 32 * (t1 ? kb - a.kb_term : kb) + 4 * q;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int64_t row = min(row0 + 16 * rt + r, last);
      const float* p = base + row * ld + kk;
      g.lo[rt] = *reinterpret_cast<const f32x4*>(p);
      g.hi[rt] = *reinterpret_cast<const f32x4*>(p + 16);
      if constexpr (MODE == RS_FILTER_BWD) {
        if (a.relu) {
          const float* py = a.Yact + row * a.ldy + kk;
          const f32x4 y0 = *reinterpret_cast<const f32x4*>(py), y1 = *reinterpret_cast<const f32x4*>(py + 16);
          f32x4& d0 = g.lo[rt];
          f32x4& d1 = g.hi[rt];
          d0 = (f32x4){y0.x > 0.f ? d0.x : 0.f, y0.y > 0.f ? d0.y : 0.f, y0.z > 0.f ? d0.z : 0.f, y0.w > 0.f ? d0.w : 0.f};
          d1 = (f32x4){y1.x > 0.f ? d1.x : 0.f, y1.y > 0.f ? d1.y : 0.f, y1.z > 0.f ? d1.z : 0.f, y1.w > 0.f ? d1.w : 0.f};
        }
      }
    }
    return g;
  };
  // chunk c of the image: k-blocks KBC (c % KC) .. of column block c / KC; a k-block of a column block is RUN contiguous u32x4
  u32x4 pre[PER];
  auto load_chunk = [&](int c) {
    const int cb = c / KC, kb0 = KBC * (c - cb * KC);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int e = tid + 256 * i;                         // element of the chunk: [kbl][RUN]
      const int kbl = e / RUN, off = e - kbl * RUN;
      pre[i] = a.Bimg[((size_t)(kb0 + kbl) * a.nt_total + (size_t)cb * NT) * TILE16 + off];
    }
  };
  // ALL of the wave's A rows are requested up front: one memory round trip (a row's k-blocks are different 128-byte lines;
  // requested one k-block ahead of the products, every k-block waited for most of a round trip: 17 us for the 128 x 128
  // projection, no faster than the fp32-MFMA kernel)
  ARaw raw[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) raw[kb] = load_a(kb);
  load_chunk(0);

  if constexpr (MODE == RS_FILTER_BWD) {   // the masked gradient rows for the parameter-gradient kernels
    if (blockIdx.x == 0 && a.zero_ptr != nullptr)
      for (int i = tid; i < a.zero_count; i += 256) a.zero_ptr[i] = 0;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int64_t row = row0 + 16 * rt + r;
        if (row <= last) {
          float* pg = a.Gp + row * a.ldgp + 32 * kb + 4 * q;
          *reinterpret_cast<f32x4*>(pg) = raw[kb].lo[rt];
          *reinterpret_cast<f32x4*>(pg + 16) = raw[kb].hi[rt];
        }
      }
  }

  f32x4 acc[2][NT];
  f32x4 old[2][NT / 4][4];
  const float sc = (MODE != RS_PLAIN && a.scale != nullptr) ? a.scale[0] : 1.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int cb = c / KC, kc = c - cb * KC;
    u32x4* buf = Bs + (c & 1) * CHUNK;
#pragma unroll
    for (int i = 0; i < PER; ++i) buf[tid + 256 * i] = pre[i];
    __syncthreads();        // chunk c is complete -- and every wave is done with chunk c - 1, whose buffer chunk c + 1 will take
    if (c + 1 < NCH) load_chunk(c + 1);
    if (kc == 0) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (MODE == RS_PLAIN) {
        if (a.accumulate) {
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int g = 0; g < NT / 4; ++g)
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int64_t row = min(row0 + 16 * rt + 4 * q + i, last);
                old[rt][g][i] = *reinterpret_cast<const f32x4*>(a.Out0 + row * a.ldo0 + cb * (16 * NT) + 64 * g + 4 * r);
              }
        }
      }
    }
#pragma unroll
    for (int kbl = 0; kbl < KBC; ++kbl) {
      const int kb = KBC * kc + kbl;
      const Pieces A0 = split3x8(raw[kb].lo[0], raw[kb].hi[0]);
      const Pieces A1 = split3x8(raw[kb].lo[1], raw[kb].hi[1]);
      const u32x4* Bk = buf + kbl * RUN + lane;
#pragma unroll
      for (int ct = 0; ct < NT; ct += 2) {
        Pieces B0, B1;
        B0.h = Bk[(ct * 3 + 0) * 64]; B0.m = Bk[(ct * 3 + 1) * 64]; B0.l = Bk[(ct * 3 + 2) * 64];
        B1.h = Bk[(ct * 3 + 3) * 64]; B1.m = Bk[(ct * 3 + 4) * 64]; B1.l = Bk[(ct * 3 + 5) * 64];
        // four independent accumulation chains, small terms first (bf16_split.h: mfma_x6)
#define RS_STEP(PA, PB)                                            \
  acc[0][ct] = mfma_bf(A0.PA, B0.PB, acc[0][ct]);                  \
  acc[1][ct] = mfma_bf(A1.PA, B0.PB, acc[1][ct]);                  \
  acc[0][ct + 1] = mfma_bf(A0.PA, B1.PB, acc[0][ct + 1]);          \
  acc[1][ct + 1] = mfma_bf(A1.PA, B1.PB, acc[1][ct + 1]);
        RS_STEP(l, h) RS_STEP(h, l) RS_STEP(m, m) RS_STEP(m, h) RS_STEP(h, m) RS_STEP(h, h)
#undef RS_STEP
      }
    }
    if (kc + 1 < KC) continue;

    // acc[rt][4g + e][i] = C[row0 + 16 rt + 4q + i][16 NT cb + 64 g + 4r + e]
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int g = 0; g < NT / 4; ++g) {
        const int col = cb * (16 * NT) + 64 * g + 4 * r;
        f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE == RS_FILTER) { if (a.bias != nullptr) bv = *reinterpret_cast<const f32x4*>(a.bias + col); }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int64_t row = row0 + 16 * rt + 4 * q + i;
          f32x4 v = (f32x4){acc[rt][4 * g][i], acc[rt][4 * g + 1][i], acc[rt][4 * g + 2][i], acc[rt][4 * g + 3][i]};
          if (row > last) continue;
          if constexpr (MODE == RS_FILTER) {
            if (a.scale != nullptr) v = (f32x4){sc * v.x, sc * v.y, sc * v.z, sc * v.w};
            v = (f32x4){v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w};
            if (a.Add != nullptr) {
              const f32x4 ad = *reinterpret_cast<const f32x4*>(a.Add + row * a.ldadd + col);
              v = (f32x4){v.x + ad.x, v.y + ad.y, v.z + ad.z, v.w + ad.w};
            }
            if (a.relu) v = (f32x4){fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
            *reinterpret_cast<f32x4*>(a.Out0 + row * a.ldo0 + col) = v;
          } else if constexpr (MODE == RS_FILTER_BWD) {
            if (a.scale != nullptr) v = (f32x4){sc * v.x, sc * v.y, sc * v.z, sc * v.w};
            if (col < a.n_split) *reinterpret_cast<f32x4*>(a.Out0 + row * a.ldo0 + col) = v;
            else *reinterpret_cast<f32x4*>(a.Out1 + row * a.ldo1 + (col - a.n_split)) = v;
          } else {
            if (a.accumulate) {
              const f32x4 o = old[rt][g][i];
              v = (f32x4){o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w};
            }
            *reinterpret_cast<f32x4*>(a.Out0 + row * a.ldo0 + col) = v;
          }
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Row-chunk partials of the parameter gradients: dweight[f * 2 + t][c] = sum_rows X_t[row][f] Gp[row][c], dbias[c] = sum_rows
// Gp[row][c], dM_L[f][g] = sum_rows X[row][f] dZ[row][g] -- A^T B with the ROW index as the MFMA k dimension.  Workgroup
// (chunk, y): y = 0 / 1: X_0 / X_1 against Gp (y = 0 also the bias), y = 2 (paper mode): X_0 against dZ -- the whole [FM x NB]
// output block in the accumulators of its four waves (see rs_param_body for the waves' shares).
// A 32-row piece of both operands is loaded coalesced (16 bytes per lane), split into three bf16 pieces in registers and kept
// as row-major bf16 images in LDS (double buffered: piece j + 1 travels while piece j is multiplied); BOTH operands are read
// transposed (ds_read_b64_tr_b16: k = the image row).  Partials [chunk][dweight | dbias | dM_L] as cloud_param_partial_kernel
// writes them; cloud_param_reduce_kernel adds them in chunk order.
template <int LD>
__device__ __forceinline__ u32x4 tr8_ld(const uint16_t* p) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * LD));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  return (u32x4){l2.x, l2.y, h2.x, h2.y};
}
template <int LD>
__device__ __forceinline__ Pieces tr_pieces(const uint16_t* img, int off) {     // planes h, m, l: 32 rows x LD each
  Pieces b;
  b.h = tr8_ld<LD>(img + off);
  b.m = tr8_ld<LD>(img + 32 * LD + off);
  b.l = tr8_ld<LD>(img + 64 * LD + off);
  return b;
}
constexpr int ld_img(int cols) { return cols == 64 ? 72 : cols + 16; }    // conflict-free transposed reads (144 / 288 bytes per row)
constexpr size_t param_lds_bytes(int FM, int NB) { return (size_t)2 * 2 * 3 * 32 * (ld_img(FM) + ld_img(NB)); }

// EIGHT waves per workgroup (two per SIMD): with four, a SIMD's one wave issued ~870 instructions per 32-row piece -- 96 MFMAs,
// the splits, the transposed reads -- at the one-wave rate of 5.5 cycles each: 2.9 us per piece, 37.5 us for 128 -> 128 where the
// matrix work is 8 us.  The product is taken TRANSPOSED (MFMA m = column c of Gp / dZ, n = feature f): a lane's four accumulator
// elements are four consecutive columns of one output row (16-byte stores of the partials).  Wave (wm, wn): the m-tiles NB / 64 wm ..,
// the n-tiles of feature half wn.  Two pieces are on their way from memory at any time (register sets R0, R1); LDS is double
// buffered (one barrier per piece).  (Splitting piece j + 1 in between the MFMA groups of piece j, into the other buffer: 29.1
// against 26.8-28.3 us -- the in-order wave does not overlap them any better than back to back.)
constexpr int PW = 8, PT = 64 * PW;      // waves / threads per workgroup of the parameter-gradient kernel
template <int FM, int NB>
__device__ __forceinline__ void rs_param_body(const ParamGradArgs& a, const float* __restrict__ A, int lda, const float* __restrict__ Bm,
                                              int ldb, float* __restrict__ out, int out_ld, int out_col0, float* __restrict__ out_bias,
                                              uint16_t* lds) {
  constexpr int LDA = ld_img(FM), LDB = ld_img(NB);
  constexpr int BUF = 3 * 32 * (LDA + LDB);                // bf16 elements per buffer: X planes, then Gp / dZ planes
  constexpr int MW = NB / 64, NW2 = FM / 32;               // m-tiles (columns c) per wave, n-tiles (features f) per wave
  constexpr int PA = FM / 64, PB = NB / 64;                // f32x4 per thread and piece
  constexpr int NPUT = PA + PB;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int chunk = blockIdx.x;
  const int r_begin = chunk * a.chunk_rows, r_end = min(a.rows, r_begin + a.chunk_rows);
  const int nkb = max(r_end - r_begin, 0) >> 5;            // (the launcher guarantees whole 32-row pieces)

  struct Regs { f32x4 v[NPUT]; };                          // [0, PA): X groups, [PA, NPUT): Gp / dZ groups
  const float* src[NPUT];
  int64_t step_elems[NPUT];
  int off[NPUT];                                           // bf16 offsets of the thread's groups inside the h plane of their image
#pragma unroll
  for (int m = 0; m < PA; ++m) {
    const int idx = tid + PT * m, row = idx / (FM / 4), c4 = idx - row * (FM / 4);
    src[m] = A + (int64_t)(r_begin + row) * lda + 4 * c4;
    step_elems[m] = (int64_t)32 * lda;
    off[m] = row * LDA + 4 * c4;
  }
#pragma unroll
  for (int m = 0; m < PB; ++m) {
    const int idx = tid + PT * m, row = idx / (NB / 4), c4 = idx - row * (NB / 4);
    src[PA + m] = Bm + (int64_t)(r_begin + row) * ldb + 4 * c4;
    step_elems[PA + m] = (int64_t)32 * ldb;
    off[PA + m] = 3 * 32 * LDA + row * LDB + 4 * c4;
  }
  auto load_piece = [&](Regs& R, int j) {
#pragma unroll
    for (int m = 0; m < NPUT; ++m) R.v[m] = *reinterpret_cast<const f32x4*>(src[m] + j * step_elems[m]);
  };
  f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};               // this thread's columns 4 (tid % (NB / 4)) .. + 3 of Gp, over its rows
  auto put = [&](const Regs& R, uint16_t* buf, int m) {
    const f32x4 v = R.v[m];
    const int ld = m < PA ? LDA : LDB;
    uint32_t h0, m0, l0, h1, m1, l1;
    split3(v.x, v.y, h0, m0, l0);
    split3(v.z, v.w, h1, m1, l1);
    uint16_t* p = buf + off[m];
    *reinterpret_cast<u32x2*>(p) = (u32x2){h0, h1};
    *reinterpret_cast<u32x2*>(p + 32 * ld) = (u32x2){m0, m1};
    *reinterpret_cast<u32x2*>(p + 64 * ld) = (u32x2){l0, l1};
    if (m >= PA && out_bias != nullptr) bsum = (f32x4){bsum.x + v.x, bsum.y + v.y, bsum.z + v.z, bsum.w + v.w};
  };

  f32x4 acc[MW][NW2];
#pragma unroll
  for (int mw = 0; mw < MW; ++mw)
#pragma unroll
    for (int nt = 0; nt < NW2; ++nt) acc[mw][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int toff_m = 3 * 32 * LDA + (8 * q + (r >> 2)) * LDB + 4 * (r & 3) + 16 * (wm * MW);   // A operand: columns of Gp / dZ
  const int toff_n = (8 * q + (r >> 2)) * LDA + 4 * (r & 3) + 16 * (wn * NW2);                  // B operand: features of X
  // piece j: split into LDS buffer j & 1, one barrier, piece j + 2 requested into the freed registers, then the products
  auto step = [&](Regs& R, int j) {
    uint16_t* buf = lds + (j & 1) * BUF;
#pragma unroll
    for (int m = 0; m < NPUT; ++m) put(R, buf, m);
    __syncthreads();        // piece j is complete -- and every wave is done with piece j - 1, whose buffer piece j + 1 will take
    if (j + 2 < nkb) load_piece(R, j + 2);
    Pieces Af[MW];
#pragma unroll
    for (int mw = 0; mw < MW; ++mw) Af[mw] = tr_pieces<LDB>(buf, toff_m + 16 * mw);
#pragma unroll
    for (int nt = 0; nt < NW2; nt += 2) {
      const Pieces B0 = tr_pieces<LDA>(buf, toff_n + 16 * nt), B1 = tr_pieces<LDA>(buf, toff_n + 16 * (nt + 1));
      // independent accumulation chains side by side, small terms first (bf16_split.h: mfma_x6)
#define RS_PSTEP(PA_, PB_)                                                                   \
  _Pragma("unroll") for (int mw = 0; mw < MW; ++mw) {                                        \
    acc[mw][nt] = mfma_bf(Af[mw].PA_, B0.PB_, acc[mw][nt]);                                  \
    acc[mw][nt + 1] = mfma_bf(Af[mw].PA_, B1.PB_, acc[mw][nt + 1]);                          \
  }
      RS_PSTEP(l, h) RS_PSTEP(h, l) RS_PSTEP(m, m) RS_PSTEP(m, h) RS_PSTEP(h, m) RS_PSTEP(h, h)
#undef RS_PSTEP
    }
  };
  Regs R0, R1;
  if (nkb > 0) load_piece(R0, 0);
  if (nkb > 1) load_piece(R1, 1);
#pragma unroll 1
  for (int j = 0; j < nkb; j += 2) {
    step(R0, j);
    if (j + 1 < nkb) step(R1, j + 1);
  }

  // acc[mw][nt][i] = out[f = 16 (wn NW2 + nt) + r][c = 16 (wm MW + mw) + 4q + i]
#pragma unroll
  for (int mw = 0; mw < MW; ++mw)
#pragma unroll
    for (int nt = 0; nt < NW2; ++nt) {
      const int f = 16 * (wn * NW2 + nt) + r;
      *reinterpret_cast<f32x4*>(out + (int64_t)f * out_ld + out_col0 + 16 * (wm * MW + mw) + 4 * q) = acc[mw][nt];
    }
  if (out_bias != nullptr) {    // column sums of Gp: the PT / (NB / 4) threads that share a column group, added in thread order
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
    *reinterpret_cast<f32x4*>(red + 4 * tid) = bsum;
    __syncthreads();
    if (tid < NB / 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(red + 4 * tid);
      for (int k = 1; k < PT / (NB / 4); ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(red + 4 * (tid + k * (NB / 4)));
        t = (f32x4){t.x + v.x, t.y + v.y, t.z + v.z, t.w + v.w};
      }
      *reinterpret_cast<f32x4*>(out_bias + 4 * tid) = t;
    }
  }
}

template <int FIN, int FOUT>
__global__ __launch_bounds__(PT) void rs_param_kernel(ParamGradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char rs_plds[];
  uint16_t* lds = reinterpret_cast<uint16_t*>(rs_plds);
  float* out = a.part + (int64_t)blockIdx.x * a.part_stride;
  const int y = blockIdx.y;
  // dweight row f * 2 + t: leading dimension 2 FOUT, term t at column offset t FOUT
  if (y == 0) rs_param_body<FIN, FOUT>(a, a.X, a.ldx, a.Gp, FOUT, out, 2 * FOUT, 0, out + (int64_t)2 * FIN * FOUT, lds);
  else if (y == 1) rs_param_body<FIN, FOUT>(a, a.sv_X, FIN, a.Gp, FOUT, out, 2 * FOUT, FOUT, nullptr, lds);
  else rs_param_body<FIN, FIN>(a, a.X, a.ldx, a.dZ, FIN, out + (int64_t)2 * FIN * FOUT + FOUT, FIN, 0, nullptr, lds);
}

bool rs_env() { return knob_rows_split() != 0; }
bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int KB, int NT, int NCB, int MODE>
int launch_rs(const RsArgs& a, int prof_id, const char* what, void* stream) {
  if (a.rows <= 0) return AGCN_OK;
  constexpr int KBC = (16 / NT) < KB ? (16 / NT) : KB;
  const size_t lds = (size_t)2 * KBC * NT * TILE16 * 16;   // two chunks (of at most 16 tiles: 48 KB each)
  const void* fn = reinterpret_cast<const void*>(&rs_rows_kernel<KB, NT, NCB, MODE>);
  if (int e = ensure_dynamic_lds(fn, lds)) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString((hipError_t)e));
  prof_begin(prof_id, stream);
  hipLaunchKernelGGL((rs_rows_kernel<KB, NT, NCB, MODE>), dim3((unsigned)((a.rows + WG_ROWS - 1) / WG_ROWS)), dim3(256), lds,
                     static_cast<hipStream_t>(stream), a);
  prof_end(stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s launch: %s", what, hipGetErrorString(e));
  return AGCN_OK;
}

int launch_presplit(const SplitJobs& jobs, int n_jobs, u32x4* img, void* stream) {
  int most = 0;
  for (int i = 0; i < n_jobs; ++i) most = std::max(most, (jobs.j[i].K >> 5) * (jobs.j[i].N >> 4) * 64);
  prof_begin(PK_ROWS_PRESPLIT, stream);
  hipLaunchKernelGGL(rs_presplit_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)n_jobs), dim3(256), 0, static_cast<hipStream_t>(stream), jobs, img);
  prof_end(stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "rs_presplit launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

// u32x4 units of the image of a K x N matrix
constexpr size_t img_units(int K, int N) { return (size_t)(K >> 5) * (N >> 4) * TILE16; }

}  // namespace

// 64 or 128 input features, Fout 64 or 128, K = 2, 16-byte aligned rows (AGCN_ROWS_SPLIT=0: the fp32-MFMA kernels instead)
bool rows_split_ok(int rows, int Fin, int Fout, int K, int ldx, int ldy) {
  return rs_env() && (Fin == 64 || Fin == 128) && K == 2 && (Fout == 64 || Fout == 128) && rows >= 1024 && (ldx & 3) == 0 && (ldy & 3) == 0;
}

// bytes of the parameter images a forward / backward call builds in its workspace: [M_L | W_0 ; W_1] and [W^T | M_L^T]
size_t rows_split_fwd_image_bytes(int Fin, int Fout) { return 16 * (img_units(Fin, Fin) + img_units(2 * Fin, Fout)); }
size_t rows_split_bwd_image_bytes(int Fin, int Fout) { return 16 * (img_units(Fout, 2 * Fin) + img_units(Fin, Fin)); }

// forward images: M_L (K = Fin, N = Fin) and the filter [W_0 ; W_1] (K = 2 Fin, term-major: k-block t Fin / 32 + kb; N = Fout)
int launch_rows_split_fwd_images(const float* M, const float* W, int Fin, int Fout, void* img, void* stream) {
  if (!al16(img)) return fail(AGCN_EINVAL, "rows_split: unaligned image workspace");
  SplitJobs jobs;
  const uint32_t w0 = (uint32_t)img_units(Fin, Fin);
  jobs.j[0] = SplitJob{M, Fin, 1, Fin, Fin, 0u, Fin / 16, 0};
  jobs.j[1] = SplitJob{W, 2 * Fout, 1, Fin, Fout, w0, Fout / 16, 0};                                           // term 0: rows f * 2
  jobs.j[2] = SplitJob{W + Fout, 2 * Fout, 1, Fin, Fout, w0 + (uint32_t)img_units(Fin, Fout), Fout / 16, 0};   // term 1: rows f * 2 + 1
  return launch_presplit(jobs, 3, static_cast<u32x4*>(img), stream);
}

// backward images: [W_0^T | W_1^T] (k = output feature c, column t Fin + f <-> W[f * 2 + t][c]: ONE image of N = 2 Fin columns)
// and M_L^T (k = g, column f <-> M_L[f][g])
int launch_rows_split_bwd_images(const float* M, const float* W, int Fin, int Fout, bool with_m, void* img, void* stream) {
  if (!al16(img)) return fail(AGCN_EINVAL, "rows_split: unaligned image workspace");
  SplitJobs jobs;
  jobs.j[0] = SplitJob{W, 1, 2 * Fout, Fout, Fin, 0u, 2 * Fin / 16, 0};
  jobs.j[1] = SplitJob{W + Fout, 1, 2 * Fout, Fout, Fin, 0u, 2 * Fin / 16, Fin / 16};
  int n = 2;
  if (with_m) jobs.j[n++] = SplitJob{M, 1, Fin, Fin, Fin, (uint32_t)img_units(Fout, 2 * Fin), Fin / 16, 0};
  return launch_presplit(jobs, n, static_cast<u32x4*>(img), stream);
}

int launch_rows_split_project(const float* X, int ldx, float* Z, int Fin, int rows, const void* img, void* stream) {
  if (!al16(X) || !al16(Z)) return fail(AGCN_EINVAL, "rows_split_project: unaligned buffer");
  RsArgs a{};
  a.A0 = X; a.lda0 = ldx; a.Out0 = Z; a.ldo0 = Fin; a.rows = rows;
  a.Bimg = static_cast<const u32x4*>(img); a.nt_total = Fin / 16;
  return Fin == 128 ? launch_rs<4, 8, 1, RS_PLAIN>(a, PK_PROJECT, "rs_project", stream)
                    : launch_rs<2, 4, 1, RS_PLAIN>(a, PK_PROJECT, "rs_project", stream);
}

int launch_rows_split_filter(const float* X0, int ldx, const float* X1, int ldx1, const float* bias, int relu, float* Y, int ldy, int Fin,
                             int Fout, int rows, const void* img, void* stream) {
  if (!al16(X0) || !al16(X1) || !al16(bias) || !al16(Y)) return fail(AGCN_EINVAL, "rows_split_filter: unaligned buffer");
  RsArgs a{};
  a.A0 = X0; a.lda0 = ldx; a.A1 = X1; a.lda1 = ldx1; a.kb_term = Fin / 32; a.relu = relu; a.bias = bias;
  a.Out0 = Y; a.ldo0 = ldy; a.rows = rows;
  a.Bimg = static_cast<const u32x4*>(img) + img_units(Fin, Fin); a.nt_total = Fout / 16;
  if (Fin == 128) return Fout == 128 ? launch_rs<8, 8, 1, RS_FILTER>(a, PK_FILTER, "rs_filter", stream)
                                       : launch_rs<8, 4, 1, RS_FILTER>(a, PK_FILTER, "rs_filter", stream);
  return Fout == 128 ? launch_rs<4, 8, 1, RS_FILTER>(a, PK_FILTER, "rs_filter", stream)
                     : launch_rs<4, 4, 1, RS_FILTER>(a, PK_FILTER, "rs_filter", stream);
}

// Gp = dY . act'(Y) -> Gp;  g_0 = Gp . W_0^T -> dX,  g_1 = Gp . W_1^T -> g1
int launch_rows_split_filter_bwd(const float* dY, const float* Y, int ldy, int relu, float* Gp, float* dX, int ldx, float* g1, int Fin,
                                 int Fout, int rows, const void* img, void* stream) {
  if (!al16(dY) || !al16(Y) || !al16(Gp) || !al16(dX) || !al16(g1)) return fail(AGCN_EINVAL, "rows_split_filter_bwd: unaligned buffer");
  RsArgs a{};
  a.A0 = dY; a.lda0 = ldy; a.Yact = Y; a.ldy = ldy; a.relu = relu; a.Gp = Gp; a.ldgp = Fout;
  a.Out0 = dX; a.ldo0 = ldx; a.Out1 = g1; a.ldo1 = Fin; a.n_split = Fin; a.rows = rows;
  a.Bimg = static_cast<const u32x4*>(img); a.nt_total = 2 * Fin / 16;
  // N = 2 Fin columns in blocks of 128: one block (Fin = 64) or two, taken one after the other by the same workgroup
  if (Fin == 128)
    return Fout == 128 ? launch_rs<4, 8, 2, RS_FILTER_BWD>(a, PK_FILTER_BWD, "rs_filter_bwd", stream)
                       : launch_rs<2, 8, 2, RS_FILTER_BWD>(a, PK_FILTER_BWD, "rs_filter_bwd", stream);
  return Fout == 128 ? launch_rs<4, 8, 1, RS_FILTER_BWD>(a, PK_FILTER_BWD, "rs_filter_bwd", stream)
                     : launch_rs<2, 8, 1, RS_FILTER_BWD>(a, PK_FILTER_BWD, "rs_filter_bwd", stream);
}

// dX += dZ . M_L^T
int launch_rows_split_project_bwd(const float* dZ, float* dX, int ldx, int Fin, int Fout, int rows, const void* img, void* stream) {
  if (!al16(dZ) || !al16(dX)) return fail(AGCN_EINVAL, "rows_split_project_bwd: unaligned buffer");
  RsArgs a{};
  a.A0 = dZ; a.lda0 = Fin; a.Out0 = dX; a.ldo0 = ldx; a.accumulate = 1; a.rows = rows;
  a.Bimg = static_cast<const u32x4*>(img) + img_units(Fout, 2 * Fin); a.nt_total = Fin / 16;
  return Fin == 128 ? launch_rs<4, 8, 1, RS_PLAIN>(a, PK_PROJECT_BWD, "rs_project_bwd", stream)
                    : launch_rs<2, 4, 1, RS_PLAIN>(a, PK_PROJECT_BWD, "rs_project_bwd", stream);
}

// ---- the dense layers next to the convolutions (DenseMol, BlockEnd, the terms of DenseBlockEnd: rows_ops.hip) at point-cloud row counts:
// Y = act([scale] A W [+ bias] [+ Add]) and the adjoint dX = [scale] (dY act'(Y)) W^T with Gp = dY act'(Y) stored for the parameter
// gradients.  One parameter image per call (W, or W^T as strides of the split) in the caller's workspace.
// (ldadd: the addend's row stride, or 0 when the call has no addend -- an unused stride must not cost the fast path)
bool rows_split_dense_ok(int rows, int Fin, int Fout, int ldx, int ldy, int ldadd) {
  auto w = [](int f) { return f == 64 || f == 128 || f == 256; };
  // AGCN_ROWS_SPLIT_DENSE_MIN (include/agcn.h, environment section): fewest packed rows at which the dense layers take these
  // kernels; a workgroup owns 128 rows, so anything below 128 (or unparsable) is clamped to 128
  static const int min_rows = [] {
    const char* e = std::getenv("AGCN_ROWS_SPLIT_DENSE_MIN");
    if (!e) return 8192;
    char* end = nullptr;
    const long v = std::strtol(e, &end, 10);
    if (end == e || v < 128) return 128;
    return v > (1 << 30) ? (1 << 30) : (int)v;
  }();
  return rs_env() && rows >= min_rows && w(Fin) && w(Fout) && (ldx & 3) == 0 && (ldy & 3) == 0 && (ldadd & 3) == 0;
}
size_t rows_split_dense_image_bytes(int Fin, int Fout) { return 16 * img_units(Fin, Fout); }

namespace {
template <int MODE>
int launch_rs_kn(const RsArgs& a, int K, int N, int prof_id, const char* what, void* stream) {
#define RS_KN(KB_, NT_, NCB_) return launch_rs<KB_, NT_, NCB_, MODE>(a, prof_id, what, stream)
  if (K == 64) { if (N == 64) RS_KN(2, 4, 1); if (N == 128) RS_KN(2, 8, 1); if (N == 256) RS_KN(2, 8, 2); }
  if (K == 128) { if (N == 64) RS_KN(4, 4, 1); if (N == 128) RS_KN(4, 8, 1); if (N == 256) RS_KN(4, 8, 2); }
  if (K == 256) { if (N == 64) RS_KN(8, 4, 1); if (N == 128) RS_KN(8, 8, 1); if (N == 256) RS_KN(8, 8, 2); }
#undef RS_KN
  return fail(AGCN_EINVAL, "%s: unsupported shape %d x %d", what, K, N);
}
}  // namespace

int launch_rows_split_dense_fwd(const float* A, int lda, const float* W, const float* scale, const float* bias, const float* Add, int ldadd,
                                int relu, float* Y, int ldy, int Fin, int Fout, int rows, void* img, void* stream) {
  if (!al16(A) || !al16(Y) || !al16(img) || (bias && !al16(bias)) || (Add && !al16(Add)))
    return fail(AGCN_EINVAL, "rows_split_dense_fwd: unaligned buffer");
  SplitJobs jobs;
  jobs.j[0] = SplitJob{W, Fout, 1, Fin, Fout, 0u, Fout / 16, 0};
  int rc = launch_presplit(jobs, 1, static_cast<u32x4*>(img), stream);
  if (rc != AGCN_OK) return rc;
  RsArgs a{};
  a.A0 = A; a.lda0 = lda; a.kb_term = Fin / 32; a.relu = relu; a.bias = bias; a.scale = scale; a.Add = Add; a.ldadd = ldadd;
  a.Out0 = Y; a.ldo0 = ldy; a.rows = rows;
  a.Bimg = static_cast<const u32x4*>(img); a.nt_total = Fout / 16;
  return launch_rs_kn<RS_FILTER>(a, Fin, Fout, PK_FILTER, "rs_dense_fwd", stream);
}

int launch_rows_split_dense_bwd(const float* dY, const float* Y, int ldy, int relu, const float* W, const float* scale, float* Gp, float* dX,
                                int ldx, int Fin, int Fout, int rows, void* img, int32_t* zero_ptr, int32_t zero_count, void* stream) {
  if (!al16(dY) || !al16(Y) || !al16(Gp) || !al16(dX) || !al16(img)) return fail(AGCN_EINVAL, "rows_split_dense_bwd: unaligned buffer");
  SplitJobs jobs;
  jobs.j[0] = SplitJob{W, 1, Fout, Fout, Fin, 0u, Fin / 16, 0};       // B[k = c][column f] = W[f][c]
  int rc = launch_presplit(jobs, 1, static_cast<u32x4*>(img), stream);
  if (rc != AGCN_OK) return rc;
  RsArgs a{};
  a.A0 = dY; a.lda0 = ldy; a.Yact = Y; a.ldy = ldy; a.relu = relu; a.Gp = Gp; a.ldgp = Fout; a.scale = scale;
  a.Out0 = dX; a.ldo0 = ldx; a.Out1 = dX; a.ldo1 = ldx; a.n_split = Fin; a.rows = rows;
  a.zero_ptr = zero_ptr; a.zero_count = zero_count;
  a.Bimg = static_cast<const u32x4*>(img); a.nt_total = Fin / 16;
  return launch_rs_kn<RS_FILTER_BWD>(a, Fout, Fin, PK_FILTER_BWD, "rs_dense_bwd", stream);
}

// Row-chunk partials of dweight / dbias / dM_L (the caller runs cloud_param_reduce_kernel behind it).  Sets a.n_chunks /
// a.chunk_rows: about one workgroup per CU over the (chunk, output block) grid, chunks in multiples of 32 rows, never more chunks
// than the caller's partial buffer holds (a.n_chunks on entry).
int launch_rows_split_param_partials(ParamGradArgs& a, void* stream) {
  if (!al16(a.X) || !al16(a.sv_X) || !al16(a.Gp) || (a.paper && !al16(a.dZ)) || !al16(a.part) || (a.part_stride & 3))
    return fail(AGCN_EINVAL, "rows_split_param_partials: unaligned buffer");
  if (a.rows & 31) return fail(AGCN_EINVAL, "rows_split_param_partials: rows must come in multiples of 32");
  const int ny = a.paper ? 3 : 2;
  const int cus = current_device_info().n_cus > 0 ? current_device_info().n_cus : 256;
  int want = std::max(1, std::min(a.n_chunks, cus / ny));
  a.chunk_rows = ((a.rows + want - 1) / want + 31) / 32 * 32;
  a.n_chunks = (a.rows + a.chunk_rows - 1) / a.chunk_rows;
  const dim3 grid((unsigned)a.n_chunks, (unsigned)ny);
#define RS_PARAM_LAUNCH(FI, FO)                                                                                              \
  {                                                                                                                          \
    const size_t lds = std::max(param_lds_bytes(FI, FO), param_lds_bytes(FI, FI));                                           \
    const void* fn = reinterpret_cast<const void*>(&rs_param_kernel<FI, FO>);                                                \
    if (int e = ensure_dynamic_lds(fn, lds)) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(rs_param): %s", hipGetErrorString((hipError_t)e)); \
    prof_begin(PK_PARAM_PARTIAL, stream);                                                                                    \
    hipLaunchKernelGGL((rs_param_kernel<FI, FO>), grid, dim3(PT), lds, static_cast<hipStream_t>(stream), a);                \
    prof_end(stream);                                                                                                        \
  }
  if (a.Fin == 128 && a.Fout == 128) RS_PARAM_LAUNCH(128, 128)
  else if (a.Fin == 128 && a.Fout == 64) RS_PARAM_LAUNCH(128, 64)
  else if (a.Fin == 64 && a.Fout == 128) RS_PARAM_LAUNCH(64, 128)
  else if (a.Fin == 64 && a.Fout == 64) RS_PARAM_LAUNCH(64, 64)
  else return fail(AGCN_EINVAL, "rows_split_param_partials: unsupported layer shape");
#undef RS_PARAM_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "rs_param launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

}  // namespace agcn
