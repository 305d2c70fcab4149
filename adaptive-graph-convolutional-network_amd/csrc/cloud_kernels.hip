// Point-cloud ("cloud") kernels: SGC_LL and SGC_LL_Reslap, K = 2, 64 or 128 input features, DENSE symmetric L_orig, every
// graph a multiple of 64 nodes (ModelNet-shaped batches, n = 1024: BASELINE config 3, the "MFMA A.X path"; the layers of
// models/networks/Point_AGCN.py).  Same math as slab_kernels.hip (graphconv.py:163-236, graphconv_reslap.py:136-230 and their
// TF autodiff) -- the n x n work of a graph cut into row strips walked in 64-column chunks -- rebuilt around what the general
// slab kernels measured as their limits at this size (profiles/r03a_modelnet_baseline_*: bwd1 358 us at 23 % matrix-pipe
// utilisation and 700 MB of traffic):
//
//  * 64-row strips, one wave per 16 rows.  The chunk's 64 rows of X_0 / g_1 / Z (the B operands of every product) are
//    staged in LDS ONCE per workgroup and chunk with fully coalesced 16-byte loads and serve four waves (the 16-row
//    strips fetched them from L2 once per strip, as 16-rows-by-64-bytes gathers for the transposed products).
//  * every n x n element is touched in the MFMA accumulator layout (lane (r, q) <-> rows 4q..4q+3, column r of a 16 x 16
//    tile): tiles are loaded / stored as 4 rows x 64 bytes per instruction, the element-wise math runs on the
//    accumulator registers of the products it belongs to, and only L (the A operand of the next product) goes through a
//    WAVE-PRIVATE LDS strip -- no workgroup barrier besides the two around the staging.
//  * all fragment reads are 16 bytes wide: images have a leading dimension of 72 floats (conflict-free ds_read_b128 for
//    "row r, columns 16 blk + 4q .. + 3"), the k index of every product is permuted inside 16-blocks accordingly, and
//    the row-major products take their B operand as B[k][4r .. 4r + 3], which makes lane r own the output columns
//    4r .. 4r + 3 (a 16-byte store).
//  * the backward keeps ONE n x n tile between its two kernels instead of two: only Q_ij = dA_ij + dA_ji is ever used
//    (ds_i = sum_j Q_ij S_ij s_j, c_ij = -S_ij (Q_ij s_i s_j + dd_i + dd_j) / r_ij, dalpha = 1/2 sum P_ij min(R_ij, 0)
//    with P = dL + dL^T: R and S are symmetric), and P is ONE product of depth 2 Fin: [g_1 | X_0][I] . [X_0 | g_1][J]^T.
//  * the similarity is computed ONCE per unordered pair (tasks of 32 rows x 64 columns of the blocks on / above the diagonal,
//    one wave each): a lane keeps its column's Z row in registers and reads the row operand through the SCALAR cache
//    (uniform address -> s_load): two vector instructions per pair and feature; both orientations of a block leave as
//    16-byte stores through wave-private LDS patches; partial degrees per (row block piece, column), added by a tiny kernel.
//  * the backward products of paper mode run on the BF16 matrix pipe at fp32 accuracy: operands split exactly into three bf16
//    pieces, six v_mfma_f32_16x16x32_bf16 per 32 k (cloud_bwd1_paper_kernel, cloud_rbwd_kernel).
//  * SGC_LL_Reslap: its two clip_by_norm stages become extra strip passes (graph-wide sums as per-strip partials), and
//    because the clip adjoints are linear in their inner products ONE backward pass collects every graph-wide sum.
//  * 128 input features: the strip kernels run once per 64-column half (SgcArgs::foff / fstride / half_acc); what couples the
//    halves is linear in the per-half products, so the second run adds to the first one's outputs.
// Reductions are per-lane sequential + fixed shuffle trees: bit-reproducible.  No floating-point atomics.
#include "agcn_device.h"
#include "bf16_split.h"
#include "sgc_kernels.h"

#include <float.h>

#include <cstdlib>

namespace agcn {

namespace {
constexpr int FIN = 64;     // feature width these kernels are compiled for
constexpr int CS = 64;      // rows per workgroup (4 waves x 16)
constexpr int CC = 64;      // columns per chunk
constexpr int LD = 72;      // leading dimension of every LDS image (floats): 72 mod 64 = 8 -> conflict-free b128 rows
constexpr float NLOG2E = -1.44269504088896341f;

typedef float f32x16 __attribute__((ext_vector_type(16)));
// sixteen consecutive floats at a wave-uniform address -> SGPRs (scalar cache); SWAIT2 = s_waitcnt lgkmcnt(0), tied to the
// registers it guards so that no use of them is scheduled above it
#define SLOAD16(dst, ptr, off) asm volatile("s_load_dwordx16 %0, %1, " off : "=s"(dst) : "s"(ptr))
// the same, ordered BEFORE every later use of `keep` (a fake read-modify-write of it): the arithmetic on the buffer in
// use cannot be scheduled in front of the request for the other buffer
#define SLOAD16_BEFORE(dst, ptr, off, keep) asm volatile("s_load_dwordx16 %0, %2, " off : "=&s"(dst), "+s"(keep) : "s"(ptr))
// ... and ordered AFTER the arithmetic that produced `dep` (a vector register)
#define SWAIT2(x, y, dep) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(x), "+s"(y), "+v"(dep))

struct WaveCtx { int b, i0, n, row0; int64_t toff; };
// wave `slab` of the plan's 16-row slab table (uniform per wave: everything lands in scalar registers)
__device__ __forceinline__ WaveCtx wave_ctx(const SgcArgs& a, int slab) {
  WaveCtx c;
  c.b = a.slab_graph[slab];
  c.i0 = a.slab_row[slab];
  c.n = a.n[c.b];
  c.row0 = a.row_off[c.b];
  c.toff = a.tile_off[c.b];
  return c;
}

// XCD-aware order of the 64-row strips: workgroups are dealt round-robin over the 8 XCDs (blockIdx.x % 8 share one L2), the
// strip table is sorted by graph, so XCD x takes the x-th EIGHTH of the table -- the strips of one cloud run on one XCD and
// its rows of X_0 / g_1 / Z / s (re-read by every strip of the cloud) come out of ONE L2 instead of being fetched into all
// eight (placement is a speed matter only).  Returns -1 for the padding workgroups of the rounded-up grid.
#ifndef AGCN_CLOUD_XCD
#define AGCN_CLOUD_XCD 1
#endif
__device__ __forceinline__ int xcd_strip(int n_strips) {
  const int per = (n_strips + 7) >> 3, g = (int)blockIdx.x;
  const int s = AGCN_CLOUD_XCD ? (g & 7) * per + (g >> 3) : g;
  return s < n_strips ? s : -1;
}
__host__ inline unsigned xcd_grid(int n_strips) { return (unsigned)(8 * ((n_strips + 7) >> 3)); }

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// sum over the 16 lanes that share q (= the 16 columns of a tile row), result in each of them
// (DPP: lane ^ 1, lane ^ 2 inside the quad, then the half row mirrored, then the row mirrored -- no LDS round trips)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2, 3, 0, 1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// acc[e] (rows 4q + i, column 4r + e) += A[16 x 64] . B[64 x 64]
//   A: wave-private image [16][LD], k = image column;  B: image [k][c] (the chunk's rows), both read 16 bytes wide.
//   MFMA step j of block blk pairs A[r][16 blk + 4q + j] with B[16 blk + 4q + j][4r + e].
__device__ __forceinline__ void strip_mma(f32x4 (&acc)[4], const float* A, const float* B, int r, int q) {
  // the fragment reads run ONE 4-MFMA step ahead of their use (the compiler, left alone, puts each ds_read right in front
  // of its MFMAs behind an lgkmcnt(0): ~100 cycles of LDS latency exposed per 128 cycles of matrix work)
  f32x4 av = *reinterpret_cast<const f32x4*>(A + r * LD + 4 * q);
  f32x4 bv = *reinterpret_cast<const f32x4*>(B + (4 * q) * LD + 4 * r);
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    f32x4 av_next = av;
    if (blk < 3) av_next = *reinterpret_cast<const f32x4*>(A + r * LD + 16 * (blk + 1) + 4 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int step = 4 * blk + j;
      f32x4 bv_next = bv;
      if (step < 15) {
        const int nb = (step + 1) >> 2, nj = (step + 1) & 3;
        bv_next = *reinterpret_cast<const f32x4*>(B + (16 * nb + 4 * q + nj) * LD + 4 * r);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = mfma4(av[j], bv[e], acc[e]);
      __builtin_amdgcn_sched_barrier(0);
      bv = bv_next;
    }
    av = av_next;
  }
}

// The chunk's 64 rows of a [rows][ld] matrix -> registers (four coalesced 16-byte loads per thread: a wave-load covers
// 4 rows x 256 bytes) -> LDS image [64][LD].
struct StageRegs { f32x4 v[4]; };
__device__ __forceinline__ StageRegs stage_load(const float* src, int ld, int c0) {
  const int tid = threadIdx.x;
  StageRegs s;
#pragma unroll
  for (int m = 0; m < 4; ++m)
    s.v[m] = *reinterpret_cast<const f32x4*>(src + (int64_t)(c0 + (tid >> 4) + 16 * m) * ld + 4 * (tid & 15));
  return s;
}
// Chunk row rho lands in image row 16 (rho % 4) + rho / 4: tile t of the accumulator layout then stands for the chunk
// columns 4r + t (lane r), i.e. a lane's four tiles are four CONSECUTIVE columns of the n x n matrices -- every tile
// access to global memory is one 16-byte load / store per row -- while the images are still indexed 16 t + r.
__device__ __forceinline__ void stage_store(float* img, const StageRegs& s) {
  const int tid = threadIdx.x, rho = tid >> 4;
#pragma unroll
  for (int m = 0; m < 4; ++m)
    *reinterpret_cast<f32x4*>(img + (16 * (rho & 3) + (rho >> 2) + 4 * m) * LD + 4 * (tid & 15)) = s.v[m];
}

__device__ __forceinline__ const float* dense_l0(const SgcArgs& a, const WaveCtx& c, int* ld) {
  *ld = a.L0_ld > 0 ? a.L0_ld : c.n;
  return a.L0_ld > 0 ? a.L0_dense + (int64_t)c.b * a.L0_ld * a.L0_ld : a.L0_dense + c.toff;
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// forward 1: S = exp(-||Z_i - Z_j||) (graphconv.py:163-180) -> res_W, r -> reserve (paper mode), partial degrees.
// S and r are symmetric: each unordered pair is computed ONCE.  A task = RT rows (i) of row block I against the 64 columns (j)
// of a column block J >= I, run by ONE wave: a lane keeps its column's Z row in 64 registers, the rows come through the scalar
// cache into SGPRs (uniform address: s_load_dwordx16), half a row at a time into one of two buffers -- the half after the one
// being consumed is requested right behind the wait for the current one (scalar loads return out of order, so the only usable
// wait is lgkmcnt(0)) and has the 64 vector instructions of a half row to arrive -- two vector instructions per pair and
// feature, four accumulation chains.  The tasks of a graph are numbered in the order (I, piece h of RT rows, J): the four waves
// of a workgroup take four consecutive ids, i.e. mostly the same rows (one scalar-cache fill serves them).
// Stores are 16 bytes per lane in both orientations: four rows of the block are turned through a wave-private LDS patch [4][64]
// (row lane / 16, columns 4 (lane % 16) ..); the MIRRORED block (off-diagonal tasks) leaves through patches [64 columns j][16
// rows i] (the lane's value of row i at [lane][i], 16-byte groups swizzled by (lane / 2) % 4: 4-way conflicts on the scalar
// writes, none on the 16-byte reads) as 64-byte row segments, 16 rows x 64 bytes per store instruction.
// Degrees: a task gives the column sums of its RT rows to partial slot (I, h) and -- the row sums = the column sums of the
// mirrored block -- to partial slot (J, 0); cloud_degree_kernel adds a node's slots in fixed order.
// Measured (32 clouds of 1024 points, paper mode; tools/ubench/sim_rate.hip has the loop in isolation: 2.3 cycles per
// instruction with four chains at 4 waves per SIMD; the kernel runs at ~1.97 GHz, SQ_BUSY_CYCLES / duration):
//   full square, the compiler's schedule of the scalar loads (8 instructions of cover)              175 us
//   full square, explicit double buffering, 16 waves of a row block per workgroup                  100-104 us
//   each pair once, a lane's four rows as one 16-byte piece of the mirrored row (64 scattered pieces per instruction)  264 us
//   each pair once, mirrored block through the [64][16] patches, 64-row tasks, workgroup = (strip, 4 column blocks)   88-90 us
//   32-row tasks from task lists (below)                                                            75 us: 47 us of vector issue
//     at the kernel's clock and the ubench's rate, 55-57 measured without stores / patches / sqrt + exp, +10 stores, +2, +3
// NF: features (64, or 128: the lane keeps 128 registers of its column's row and every row takes two rounds of the
// double-buffered 64-float loads -- 2 waves per SIMD)
template <bool PAPER, int RT, int NF = 64>
__global__ __launch_bounds__(256, NF == 64 ? 4 : 2) void cloud_sim_kernel(SgcArgs a, float* dpart, int t_quota, int nquads, int qstride) {
  constexpr int WSZ = PAPER ? 2560 : 1280;      // floats of LDS per wave: [4][64] S (, r) + [64][16] S^T (, r^T)
  constexpr int NS = CS / RT;                   // partial-degree slots per row block
  extern __shared__ __attribute__((aligned(16))) float trbuf[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* nS = trbuf + wave * WSZ;                // [4][64]
  float* nR = nS + 256;
  float* tS = nS + (PAPER ? 512 : 256);          // [64][16]
  float* tR = tS + 1024;
  const int fz = 4 * ((lane >> 1) & 3);
  // PERSISTENT waves (4 workgroups per CU), static task lists: the ids (graph, block pair t, piece h), a quota of t_quota block
  // pairs per graph (smaller graphs leave ids empty), are cut into eight contiguous ranges, one per XCD (workgroups go round
  // robin over the XCDs: a cloud's Z rows and tiles stay in one L2), and the waves of an XCD take a range's ids in turn.
  // (Task queues with an atomic cursor per XCD: 470-580 us -- thousands of same-address atomics, one round trip each.)
  const int64_t total = (int64_t)a.B * t_quota * NS;
  const int per_q = (int)((total + 7) >> 3);
  const int xcd = (int)(blockIdx.x & 7);
  const int wx = 4 * (int)(gridDim.x >> 3);      // waves per XCD
#pragma unroll 1
  for (int lid0 = 4 * (int)(blockIdx.x >> 3) + wave; lid0 < 4 * nquads; lid0 += wx) {
  // (quads of ids -- the four waves of a workgroup: same rows -- in a strided order: workgroups that run side by side work on
  //  different row blocks.  In id order they all write the same 256-byte column segment of rows 4 KB apart -- the mirrored
  //  blocks (J, I) of one I -- i.e. into the same memory channels: 492 us.)
  const int lid = 4 * (int)(((int64_t)(lid0 >> 2) * qstride) % nquads) + (lid0 & 3);
  const int64_t id = (int64_t)xcd * per_q + lid;
  if (lid >= per_q || id >= total) continue;
  // (wave-uniform by construction; readfirstlane tells the compiler, which then keeps the pointers below in SGPRs -- the
  //  s_load operands must be: with the row pointer in VGPRs the kernel assembled, gave the right results and took 492 us)
  const int b = __builtin_amdgcn_readfirstlane((int)(id / (t_quota * NS)));
  const int rem = __builtin_amdgcn_readfirstlane((int)(id - (int64_t)b * (t_quota * NS)));
  const int n = __builtin_amdgcn_readfirstlane(a.n[b]), row0 = __builtin_amdgcn_readfirstlane(a.row_off[b]);
  const int ncb = n >> 6;
  if (rem >= ((ncb * (ncb + 1)) >> 1) * NS) continue;
  // ids of a graph in the order (I, h, J >= I): neighbouring waves share their rows (the scalar cache serves them once)
  int I = 0, J, h;
  {
    int left = rem, len = ncb;                   // (<= 25 steps of scalar arithmetic)
    while (left >= len * NS) { left -= len * NS; --len; ++I; }
    h = left / len;
    J = I + (left - h * len);
  }
  const int rb0 = 64 * I;
  const bool diag = J == I;
  const int r0 = rb0 + RT * h;                   // first row of the task
  const int64_t toff_v = a.tile_off[b];
  const int64_t toff = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(toff_v >> 32)) << 32) |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)toff_v));
  const float* Zg = a.sv_Z + (int64_t)row0 * NF;
  float zj[NF];
  {
    const f32x4* zp = reinterpret_cast<const f32x4*>(Zg + (int64_t)(64 * J + lane) * NF);
#pragma unroll
    for (int f4 = 0; f4 < NF / 4; ++f4) {
      const f32x4 t = zp[f4];
      zj[4 * f4 + 0] = t.x; zj[4 * f4 + 1] = t.y; zj[4 * f4 + 2] = t.z; zj[4 * f4 + 3] = t.w;
    }
  }
  const int dlane = diag ? lane - RT * h : -1;   // S_ii = 0 (graphconv.py:171-173): on the diagonal block only
  float cs = 0.f;
  float* dpI = dpart + (toff >> 6) * NS + (int64_t)(I * NS + h) * n;    // partial degrees [n / RT][n] per graph
  float* dpJ = dpart + (toff >> 6) * NS + (int64_t)(J * NS) * n;
  const float* zrow = Zg + (int64_t)r0 * NF;
  f32x16 A0, A1, B0, B1;
  SLOAD16(A0, zrow, "0x0");
  SLOAD16(A1, zrow, "0x40");
#pragma unroll 1
  for (int g16 = 0; g16 < RT; g16 += 16) {
#pragma unroll 1
    for (int i4 = 0; i4 < 16; i4 += 4) {
#pragma unroll 1
      for (int u = 0; u < 4; ++u) {
        const int ii = g16 + i4 + u;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
#define AGCN_SIM16(SRC, BASE)                                                                                       \
  _Pragma("unroll") for (int f = 0; f < 16; f += 4) {                                                             \
    const float d0 = SRC[f] - zj[BASE + f], d1 = SRC[f + 1] - zj[BASE + f + 1];                                     \
    const float d2 = SRC[f + 2] - zj[BASE + f + 2], d3 = SRC[f + 3] - zj[BASE + f + 3];                             \
    c0 = __builtin_fmaf(d0, d0, c0); c1 = __builtin_fmaf(d1, d1, c1);                                               \
    c2 = __builtin_fmaf(d2, d2, c2); c3 = __builtin_fmaf(d3, d3, c3);                                               \
  }
#pragma unroll
        for (int hh = 0; hh < NF / 64; ++hh) {   // 64 floats of the row per round (rows are contiguous: + 64 floats each time)
          SWAIT2(A0, A1, cs);
          SLOAD16_BEFORE(B0, zrow, "0x80", A0);
          SLOAD16_BEFORE(B1, zrow, "0xc0", A1);
          AGCN_SIM16(A0, 64 * hh)
          AGCN_SIM16(A1, 64 * hh + 16)
          SWAIT2(B0, B1, c3);
          zrow += (ii + 1 < RT || hh + 1 < NF / 64) ? 64 : 0;   // (the last request re-reads the last 64 floats: never used)
          SLOAD16_BEFORE(A0, zrow, "0x0", B0);
          SLOAD16_BEFORE(A1, zrow, "0x40", B1);
          AGCN_SIM16(B0, 64 * hh + 32)
          AGCN_SIM16(B1, 64 * hh + 48)
        }
#undef AGCN_SIM16
        // (each pair is computed once: S and r are symmetric to the bit)
        const float rr = __builtin_amdgcn_sqrtf((c0 + c1) + (c2 + c3));
        const float S = (ii == dlane) ? 0.f : __builtin_amdgcn_exp2f(rr * NLOG2E);   // exp(-1 * norm) (:176-177)
        const int tcol = 16 * lane + ((i4 + u) ^ fz);
        nS[u * 64 + lane] = S;
        tS[tcol] = S;
        if (PAPER) { nR[u * 64 + lane] = rr; tR[tcol] = rr; }
        cs += S;
      }
      __builtin_amdgcn_wave_barrier();
      {
        const int64_t o = toff + (int64_t)(r0 + g16 + i4 + (lane >> 4)) * n + 64 * J + 4 * (lane & 15);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(nS + 4 * lane);
        *reinterpret_cast<f32x4*>(a.res_W + o) = sv;
        if (PAPER) __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(nR + 4 * lane), reinterpret_cast<f32x4*>(a.sv_r + o));
        if (!diag) {   // row sums of this block = column sums of the mirrored one
          const float rs = row16_sum((sv.x + sv.y) + (sv.z + sv.w));
          if ((lane & 15) == 0) dpJ[r0 + g16 + i4 + (lane >> 4)] = rs;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (!diag) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int j = (lane >> 2) + 16 * m;
        const int src = 16 * j + 4 * ((lane & 3) ^ ((j >> 1) & 3));
        const int64_t o = toff + (int64_t)(64 * J + j) * n + r0 + g16 + 4 * (lane & 3);
        *reinterpret_cast<f32x4*>(a.res_W + o) = *reinterpret_cast<const f32x4*>(tS + src);
        if (PAPER) __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(tR + src), reinterpret_cast<f32x4*>(a.sv_r + o));
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  SWAIT2(A0, A1, cs);                          // (nothing of the row requests is left in flight)
  dpI[64 * J + lane] = cs;
  }
}

// forward 1 at 128 features in TWO PASSES of 64 features per 16-row group, at the 64-feature kernel's register budget (four waves
// per SIMD instead of the two of cloud_sim_kernel<.., 128>, whose lane keeps all 128 floats of its column's row): pass 0 leaves the
// per-pair partial sums of squares over features 0..63 in the wave's LDS slots the pass-1 results will overwrite anyway (tS: the
// mirrored-block patch, slot (lane, row % 16)), pass 1 adds features 64..127 and finishes the pair as before.  The lane's column row
// is fetched twice per group (64 floats per pass: L1 / L2 hits), the task's rows twice (scalar cache).  Same tasks, slots and
// stores as cloud_sim_kernel; the sum of squares of a pair is (first half) + (second half), each half four chains as before.
template <bool PAPER, int RT>
__global__ __launch_bounds__(256, 4) void cloud_sim2_kernel(SgcArgs a, float* dpart, int t_quota, int nquads, int qstride) {
  constexpr int NF = 128;
  constexpr int WSZ = PAPER ? 2560 : 1280;      // floats of LDS per wave: [4][64] S (, r) + [64][16] S^T (, r^T)
  constexpr int NS = CS / RT;                   // partial-degree slots per row block
  extern __shared__ __attribute__((aligned(16))) float trbuf[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  float* nS = trbuf + wave * WSZ;                // [4][64]
  float* nR = nS + 256;
  float* tS = nS + (PAPER ? 512 : 256);          // [64][16]
  float* tR = tS + 1024;
  const int fz = 4 * ((lane >> 1) & 3);
  const int64_t total = (int64_t)a.B * t_quota * NS;
  const int per_q = (int)((total + 7) >> 3);
  const int xcd = (int)(blockIdx.x & 7);
  const int wx = 4 * (int)(gridDim.x >> 3);      // waves per XCD
#pragma unroll 1
  for (int lid0 = 4 * (int)(blockIdx.x >> 3) + wave; lid0 < 4 * nquads; lid0 += wx) {
  const int lid = 4 * (int)(((int64_t)(lid0 >> 2) * qstride) % nquads) + (lid0 & 3);
  const int64_t id = (int64_t)xcd * per_q + lid;
  if (lid >= per_q || id >= total) continue;
  const int b = __builtin_amdgcn_readfirstlane((int)(id / (t_quota * NS)));
  const int rem = __builtin_amdgcn_readfirstlane((int)(id - (int64_t)b * (t_quota * NS)));
  const int n = __builtin_amdgcn_readfirstlane(a.n[b]), row0 = __builtin_amdgcn_readfirstlane(a.row_off[b]);
  const int ncb = n >> 6;
  if (rem >= ((ncb * (ncb + 1)) >> 1) * NS) continue;
  int I = 0, J, h;
  {
    int left = rem, len = ncb;
    while (left >= len * NS) { left -= len * NS; --len; ++I; }
    h = left / len;
    J = I + (left - h * len);
  }
  const int rb0 = 64 * I;
  const bool diag = J == I;
  const int r0 = rb0 + RT * h;                   // first row of the task
  const int64_t toff_v = a.tile_off[b];
  const int64_t toff = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(toff_v >> 32)) << 32) |
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)toff_v));
  const float* Zg = a.sv_Z + (int64_t)row0 * NF;
  const f32x4* zp = reinterpret_cast<const f32x4*>(Zg + (int64_t)(64 * J + lane) * NF);
  const int dlane = diag ? lane - RT * h : -1;   // S_ii = 0 (graphconv.py:171-173): on the diagonal block only
  float cs = 0.f;
  float* dpI = dpart + (toff >> 6) * NS + (int64_t)(I * NS + h) * n;    // partial degrees [n / RT][n] per graph
  float* dpJ = dpart + (toff >> 6) * NS + (int64_t)(J * NS) * n;
#pragma unroll 1
  for (int g16 = 0; g16 < RT; g16 += 16) {
    float zj[64];
    f32x16 A0, A1, B0, B1;
#define AGCN_SIM16(SRC, BASE)                                                                                       \
  _Pragma("unroll") for (int f = 0; f < 16; f += 4) {                                                             \
    const float d0 = SRC[f] - zj[BASE + f], d1 = SRC[f + 1] - zj[BASE + f + 1];                                     \
    const float d2 = SRC[f + 2] - zj[BASE + f + 2], d3 = SRC[f + 3] - zj[BASE + f + 3];                             \
    c0 = __builtin_fmaf(d0, d0, c0); c1 = __builtin_fmaf(d1, d1, c1);                                               \
    c2 = __builtin_fmaf(d2, d2, c2); c3 = __builtin_fmaf(d3, d3, c3);                                               \
  }
    // one row's 64 floats of this pass in two double-buffered rounds (as cloud_sim_kernel); rows are 128 floats apart
#define AGCN_SIM_ROW(LAST)                                                                                          \
    SWAIT2(A0, A1, cs);                                                                                             \
    SLOAD16_BEFORE(B0, zrow, "0x80", A0);                                                                           \
    SLOAD16_BEFORE(B1, zrow, "0xc0", A1);                                                                           \
    AGCN_SIM16(A0, 0)                                                                                               \
    AGCN_SIM16(A1, 16)                                                                                              \
    SWAIT2(B0, B1, c3);                                                                                             \
    zrow += (LAST) ? 0 : NF;                     /* (the last request re-reads the last row: never used) */        \
    SLOAD16_BEFORE(A0, zrow, "0x0", B0);                                                                            \
    SLOAD16_BEFORE(A1, zrow, "0x40", B1);                                                                           \
    AGCN_SIM16(B0, 32)                                                                                              \
    AGCN_SIM16(B1, 48)
    // ---- pass 0: features 0 .. 63 -> partial sums into the tS slots
    {
#pragma unroll
      for (int f4 = 0; f4 < 16; ++f4) {
        const f32x4 t = zp[f4];
        zj[4 * f4 + 0] = t.x; zj[4 * f4 + 1] = t.y; zj[4 * f4 + 2] = t.z; zj[4 * f4 + 3] = t.w;
      }
      const float* zrow = Zg + (int64_t)(r0 + g16) * NF;
      SLOAD16(A0, zrow, "0x0");
      SLOAD16(A1, zrow, "0x40");
#pragma unroll 1
      for (int rr16 = 0; rr16 < 16; ++rr16) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        AGCN_SIM_ROW(rr16 == 15)
        tS[16 * lane + (rr16 ^ fz)] = (c0 + c1) + (c2 + c3);
      }
      SWAIT2(A0, A1, cs);
    }
    // ---- pass 1: features 64 .. 127, the pair finished
    {
#pragma unroll
      for (int f4 = 0; f4 < 16; ++f4) {
        const f32x4 t = zp[16 + f4];
        zj[4 * f4 + 0] = t.x; zj[4 * f4 + 1] = t.y; zj[4 * f4 + 2] = t.z; zj[4 * f4 + 3] = t.w;
      }
      const float* zrow = Zg + (int64_t)(r0 + g16) * NF + 64;
      SLOAD16(A0, zrow, "0x0");
      SLOAD16(A1, zrow, "0x40");
#pragma unroll 1
      for (int i4 = 0; i4 < 16; i4 += 4) {
#pragma unroll 1
        for (int u = 0; u < 4; ++u) {
          const int ii = g16 + i4 + u;
          float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
          AGCN_SIM_ROW(i4 + u == 15)
          const int tcol = 16 * lane + ((i4 + u) ^ fz);
          const float rr = __builtin_amdgcn_sqrtf(tS[tcol] + ((c0 + c1) + (c2 + c3)));
          const float S = (ii == dlane) ? 0.f : __builtin_amdgcn_exp2f(rr * NLOG2E);   // exp(-1 * norm) (:176-177)
          nS[u * 64 + lane] = S;
          tS[tcol] = S;
          if (PAPER) { nR[u * 64 + lane] = rr; tR[tcol] = rr; }
          cs += S;
        }
        __builtin_amdgcn_wave_barrier();
        {
          const int64_t o = toff + (int64_t)(r0 + g16 + i4 + (lane >> 4)) * n + 64 * J + 4 * (lane & 15);
          const f32x4 sv = *reinterpret_cast<const f32x4*>(nS + 4 * lane);
          *reinterpret_cast<f32x4*>(a.res_W + o) = sv;
          if (PAPER) __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(nR + 4 * lane), reinterpret_cast<f32x4*>(a.sv_r + o));
          if (!diag) {   // row sums of this block = column sums of the mirrored one
            const float rs = row16_sum((sv.x + sv.y) + (sv.z + sv.w));
            if ((lane & 15) == 0) dpJ[r0 + g16 + i4 + (lane >> 4)] = rs;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      SWAIT2(A0, A1, cs);
    }
#undef AGCN_SIM_ROW
#undef AGCN_SIM16
    if (!diag) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int j = (lane >> 2) + 16 * m;
        const int src = 16 * j + 4 * ((lane & 3) ^ ((j >> 1) & 3));
        const int64_t o = toff + (int64_t)(64 * J + j) * n + r0 + g16 + 4 * (lane & 3);
        *reinterpret_cast<f32x4*>(a.res_W + o) = *reinterpret_cast<const f32x4*>(tS + src);
        if (PAPER) __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(tR + src), reinterpret_cast<f32x4*>(a.sv_r + o));
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  dpI[64 * J + lane] = cs;
  }
}

// forward 2: d_j = sum of the node's partial column sums in slot order + np.spacing(0) (graphconv.py:195-196), s = d^-1/2 (:197).
// Slot (rb, h) of column j: h-th RT-row piece of row block rb for rb <= j / 64 (computed blocks), for rb > j / 64 the mirrored
// block's sum in slot (rb, 0) alone.  One 64-thread workgroup per 64-row block.
__global__ __launch_bounds__(64) void cloud_degree_kernel(SgcArgs a, const float* dpart, int ns) {
  const int slab0 = 4 * (int)blockIdx.x;
  const int b = a.slab_graph[slab0], r0 = a.slab_row[slab0];
  const int n = a.n[b], row0 = a.row_off[b];
  const int64_t toff = a.tile_off[b];
  const int j = r0 + (int)threadIdx.x;
  const float* dp = dpart + (toff >> 6) * ns + j;
  float d = 0.f;
  const int nslot = (n >> 6) * ns, own = ((r0 >> 6) + 1) * ns;   // slots below `own`: all ns pieces; above: piece 0 only
  for (int s0 = 0; s0 < nslot; s0 += 8) {   // eight loads in flight, added in slot order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int sl = s0 + u;
      const bool live = sl < nslot && (sl < own || sl % ns == 0);
      v[u] = live ? dp[(int64_t)sl * n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) d += v[u];
  }
  a.sv_s[row0 + j] = 1.0f / sqrtf(d + FLT_TRUE_MIN);
  if (r0 == 0 && threadIdx.x == 0) {   // clip_by_average_norm(R, 1) never clips for SGC_LL: scale exactly 1
    float* sc = a.sv_scal + 4 * b;
    sc[0] = 1.f; sc[1] = 1.f; sc[2] = 0.f; sc[3] = 1.f;
  }
  // W^T for the backward's filter adjoint (WT[k][c][f] = W[f * K + k][c]) -> reserve, a slice per workgroup
  if (a.Fin == FIN)
  for (int e = (int)blockIdx.x * 64 + (int)threadIdx.x; e < 2 * a.Fout * FIN; e += (int)gridDim.x * 64) {
    const int f = e % FIN, kc = e / FIN, c = kc % a.Fout, k = kc / a.Fout;
    a.sv_WT[e] = a.weight[(int64_t)(f * 2 + k) * a.Fout + c];
  }
}

// ------------------------------------------------------------------------------------------------
// SGC_LL_Reslap at point-cloud sizes (graphconv_reslap.py:184-198).  Its two clip_by_norm stages need graph-wide sums of
// squares before a single element can leave, so the Laplacian takes three strip kernels instead of one:
//   cloud_rnorm_kernel   ||R||_F^2 per strip (paper mode; R = I in reference mode: n)
//   cloud_rlap1_kernel   R2 = lrelu(clip(R)) -> res_L;  L_pre = R2 + L_orig + beta L_prev -> reserve;  ||L_pre||_F^2 per strip
//   cloud_lap_kernel<.., RES>   La = lrelu(clip(L_pre)) -> L_sum, X_1 = La X_0, Y
// The per-strip partials of a graph are added by every workgroup that needs the sum, lane k taking strip k, in one fixed
// butterfly order (graphs have at most 25 strips).
__device__ __forceinline__ float graph_sum(const float* part, int count, int lane) {
  return wave_sum_all(lane < count ? part[lane] : 0.f);
}
__device__ __forceinline__ float block4_sum(float v, float* red, int wave, int lane) {
  v = wave_sum_all(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void cloud_rnorm_kernel(SgcArgs a, float* rpart) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = xcd_strip(a.n_slabs >> 2);
  if (strip < 0) return;
  const WaveCtx c = wave_ctx(a, 4 * strip + wave);
  const int n = c.n;
  const float* Wt = a.res_W + c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r;
  const float* sjp = a.sv_s + c.row0 + 4 * r;
  float si[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) si[i] = a.sv_s[c.row0 + c.i0 + 4 * q + i];
  float part = 0.f;
#pragma unroll 2
  for (int c0 = 0; c0 < n; c0 += CC) {
    const f32x4 sj = *reinterpret_cast<const f32x4*>(sjp + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 S = *reinterpret_cast<const f32x4*>(Wt + (int64_t)i * n + c0);
      const int row = c.i0 + 4 * q + i;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float R = ((row == c0 + 4 * r + t) ? 1.f : 0.f) - (si[i] * S[t]) * sj[t];   // cloud_lap's instruction
        part = __builtin_fmaf(R, R, part);
      }
    }
  }
  const float tot = block4_sum(part, red, wave, lane);
  if (tid == 0) rpart[strip] = tot;
}

template <bool PAPER>
__global__ __launch_bounds__(256) void cloud_rlap1_kernel(SgcArgs a, const float* rpart, float* lpart) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = xcd_strip(a.n_slabs >> 2);
  if (strip < 0) return;
  const WaveCtx c = wave_ctx(a, 4 * strip + wave);
  const int n = c.n;
  const int strip0 = strip - (a.slab_row[4 * strip] >> 6);
  const float ssR = PAPER ? graph_sum(rpart + strip0, n >> 6, lane) : (float)n;
  const float scR = clip_scale(ssR, 1.0f);
  const float alpha = a.alpha[0];
  const bool has_prev = a.L_prev != nullptr;
  const float beta = has_prev ? a.beta[0] : 0.f;
  int l0ld;
  const float* L0d = dense_l0(a, c, &l0ld);
  const int64_t o0 = c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r;
  const float* Wt = a.res_W + o0;
  const float* Pt = has_prev ? a.L_prev + o0 : nullptr;
  float* RLt = a.res_L + o0;
  float* Lpt = a.sv_Lpre + o0;
  const float* L0t = L0d + (int64_t)(c.i0 + 4 * q) * l0ld + 4 * r;
  const float* sjp = a.sv_s + c.row0 + 4 * r;
  float si[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) si[i] = PAPER ? a.sv_s[c.row0 + c.i0 + 4 * q + i] : 0.f;
  struct Tiles { f32x4 s[4], l0[4], lp[4], sj; };
  auto load_tiles = [&](int c0) {
    Tiles g;
    g.sj = PAPER ? *reinterpret_cast<const f32x4*>(sjp + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g.s[i] = PAPER ? *reinterpret_cast<const f32x4*>(Wt + (int64_t)i * n + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
      g.l0[i] = *reinterpret_cast<const f32x4*>(L0t + (int64_t)i * l0ld + c0);
      g.lp[i] = has_prev ? *reinterpret_cast<const f32x4*>(Pt + (int64_t)i * n + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    return g;
  };
  float part = 0.f;
  Tiles cur = load_tiles(0);
#pragma unroll 1
  for (int c0 = 0; c0 < n; c0 += CC) {
    const Tiles nxt = (c0 + CC < n) ? load_tiles(c0 + CC) : cur;      // the next chunk's tiles in flight
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = c.i0 + 4 * q + i;
      f32x4 R2, Lp;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float R = (row == c0 + 4 * r + t) ? 1.f : 0.f;
        if (PAPER) R = R - (si[i] * cur.s[i][t]) * cur.sj[t];
        R2[t] = lrelu(R * scR, alpha);                                  // reslap:185-186
        float v = R2[t] + cur.l0[i][t];                                // :193
        if (has_prev) v = v + cur.lp[i][t] * beta;                     // :190
        Lp[t] = v;
        part = __builtin_fmaf(v, v, part);
      }
      __builtin_nontemporal_store(R2, reinterpret_cast<f32x4*>(RLt + (int64_t)i * n + c0));
      *reinterpret_cast<f32x4*>(Lpt + (int64_t)i * n + c0) = Lp;
    }
    cur = nxt;
  }
  const float tot = block4_sum(part, red, wave, lane);
  if (tid == 0) lpart[strip] = tot;
}

// ------------------------------------------------------------------------------------------------
// Tile convention of the three strip kernels below: a wave owns 16 rows; of a 64-column chunk lane (r, q) holds, for
// its rows 4q + i (i = 0..3), the four columns c0 + 4r + t (t = 0..3) as ONE f32x4 -- component t is the lane's element
// of accumulator tile t, whose image index (column of the L / c strips, row of the staged chunk images) is 16 t + r.

// forward 3: R = I - s S s (paper) | I (reference), res_L = lrelu(R), L = res_L + L_orig, X_1[I] = L[I, :] X_0
// (graphconv.py:198-231) and the filter product Y[I] = act(X_0[I] W_0 + X_1[I] W_1 + b) (:238-251, 117-123).
// RES: second Laplacian stage of SGC_LL_Reslap (graphconv_reslap.py:194-195): the tiles are L_pre (reserve, written by
// cloud_rlap1_kernel), La = lrelu(clip_by_norm(L_pre, 1)) -> L_sum (the 4th output) and the L of the Chebyshev product.
// NH: 64-column halves of the feature matrices handled by ONE run (1: 64-feature layers -- and the filter product behind the
// strip loop --; 2: 128-feature layers, foff = 0: the n x n tiles are read and written ONCE for both halves -- run once per half
// (rounds 3 and 4a) every tile was read twice: 83 + 73 us and 277 + 143 MB of reads for X_1 = L X_0, profiles/r04_modelnet128_*).
template <bool PAPER, bool RES = false, int NH = 1>
__global__ __launch_bounds__(256) void cloud_lap_kernel(SgcArgs a, const float* rpart, const float* lpart) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = xcd_strip(a.n_slabs >> 2);
  if (strip < 0) return;
  const WaveCtx c = wave_ctx(a, 4 * strip + wave);
  const int n = c.n;
  float* Bs = smem;                             // [NH][64][LD]  the chunk's rows of X_0 (one image per feature half)
  float* Ls = smem + NH * CC * LD + wave * (16 * LD);   // [16][LD]  this wave's strip of L
  const float alpha = a.alpha[0];
  const float* Xg = a.X + (int64_t)c.row0 * a.ldx + a.foff;
  int l0ld;
  const float* L0d = dense_l0(a, c, &l0ld);
  const float* Wt = a.res_W + c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r;
  float* RLt = a.res_L + c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r;
  float* LSt = a.L_sum ? a.L_sum + c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r : nullptr;
  const float* L0t = L0d + (int64_t)(c.i0 + 4 * q) * l0ld + 4 * r;
  const float* sjp = a.sv_s + c.row0 + 4 * r;
  float si[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) si[i] = (PAPER && !RES) ? a.sv_s[c.row0 + c.i0 + 4 * q + i] : 0.f;

  // Reslap: the two clip scales from the strips' partial sums of squares (every workgroup of a graph adds the same values in
  // the same order: the same bits)
  float ssR = 1.f, scR = 1.f, ssL = 1.f, scL = 1.f;
  if constexpr (RES) {
    const int strip0 = strip - (a.slab_row[4 * strip] >> 6), ncb = n >> 6;
    ssR = PAPER ? graph_sum(rpart + strip0, ncb, lane) : (float)n;
    ssL = graph_sum(lpart + strip0, ncb, lane);
    scR = clip_scale(ssR, 1.0f);
    scL = clip_scale(ssL, 1.0f);
    if (c.i0 == 0 && tid == 0) {   // norms / scales for the backward projections
      float* sc = a.sv_scal + 4 * c.b;
      sc[0] = ssR; sc[1] = scR; sc[2] = ssL; sc[3] = scL;
    }
  }
  const float* Lpt = RES ? a.sv_Lpre + c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r : nullptr;
  struct Tiles { f32x4 s[4], l0[4], sj; };   // (RES: s = the L_pre tiles, nothing else)
  auto load_tiles = [&](int c0) {
    Tiles g;
    g.sj = (PAPER && !RES) ? *reinterpret_cast<const f32x4*>(sjp + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (RES) {
        g.s[i] = *reinterpret_cast<const f32x4*>(Lpt + (int64_t)i * n + c0);
        g.l0[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      } else {
        g.s[i] = PAPER ? *reinterpret_cast<const f32x4*>(Wt + (int64_t)i * n + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
        g.l0[i] = *reinterpret_cast<const f32x4*>(L0t + (int64_t)i * l0ld + c0);
      }
    }
    return g;
  };

  f32x4 acc[NH][4];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[h][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  StageRegs pre[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) pre[h] = stage_load(Xg + 64 * h, a.ldx, 0);
  Tiles cur = load_tiles(0);
#pragma unroll 1
  for (int c0 = 0; c0 < n; c0 += CC) {
    __syncthreads();                  // every wave is done with the previous chunk's rows
#pragma unroll
    for (int h = 0; h < NH; ++h) stage_store(Bs + h * CC * LD, pre[h]);
    __syncthreads();
    if (c0 + CC < n) {                // the next chunk's rows in flight behind this chunk's arithmetic
#pragma unroll
      for (int h = 0; h < NH; ++h) pre[h] = stage_load(Xg + 64 * h, a.ldx, c0 + CC);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = c.i0 + 4 * q + i;
      f32x4 Lv, Ltv;
      if constexpr (RES) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          Ltv[t] = lrelu(cur.s[i][t] * scL, alpha);                    // La = lrelu(clip_by_norm(L_pre, 1))  (reslap:194-195)
          Ls[(4 * q + i) * LD + 16 * t + r] = Ltv[t];
        }
        __builtin_nontemporal_store(Ltv, reinterpret_cast<f32x4*>(LSt + (int64_t)i * n + c0));
      } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float R = (row == c0 + 4 * r + t) ? 1.f : 0.f;                // reference: I - D*W*D elementwise == I
        if (PAPER) R = R - (si[i] * cur.s[i][t]) * cur.sj[t];          // paper: I - s S s
        Lv[t] = lrelu(R, alpha);                                      // (R * 1.0f: the clip scale is exactly 1)
        Ltv[t] = Lv[t] + cur.l0[i][t];
        Ls[(4 * q + i) * LD + 16 * t + r] = Ltv[t];
      }
      __builtin_nontemporal_store(Lv, reinterpret_cast<f32x4*>(RLt + (int64_t)i * n + c0));
      if (LSt) *reinterpret_cast<f32x4*>(LSt + (int64_t)i * n + c0) = Ltv;
      }
    }
    // the next chunk's tiles: requested once this chunk's are consumed (into the same registers), in flight behind the product
    if (c0 + CC < n) cur = load_tiles(c0 + CC);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < NH; ++h) strip_mma(acc[h], Ls, Bs + h * CC * LD, r, q);
  }
  // X_1[I] = L[I, :] X_0 -> reserve; and, with X_0[I], the A images of the filter product
  __syncthreads();                    // the chunk image is dead: its space holds the X_0[I] images now
  float* X0s = smem + wave * (16 * LD);   // (inside Bs: 4 x 16 rows)
  float* Xo = a.sv_X + (int64_t)(c.row0 + c.i0) * a.fstride + a.foff;
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = (f32x4){acc[h][0][i], acc[h][1][i], acc[h][2][i], acc[h][3][i]};
      *reinterpret_cast<f32x4*>(Xo + (int64_t)(4 * q + i) * a.fstride + 64 * h + 4 * r) = v;
      if (NH == 1) *reinterpret_cast<f32x4*>(Ls + (4 * q + i) * LD + 4 * r) = v;
    }
  if (NH != 1 || a.fstride != FIN) return;   // 128-feature layer: the filter product is a packed-row launch
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int row = q + 4 * m;
    *reinterpret_cast<f32x4*>(X0s + row * LD + 4 * r) =
        *reinterpret_cast<const f32x4*>(Xg + (int64_t)(c.i0 + row) * a.ldx + 4 * r);
  }
  __builtin_amdgcn_wave_barrier();
  {
    const int Fout = a.Fout, K = a.K;
    const bool relu = a.activation == AGCN_ACT_RELU;
    float* Yg = a.Y + (int64_t)(c.row0 + c.i0) * a.ldy;
    for (int h = 0; 64 * h < Fout; ++h) {
      const int col = 64 * h + 4 * r;
      const bool cok = col < Fout;                       // (Fout is a multiple of 4: a 16-byte group is whole or absent)
      const int cc = cok ? col : Fout - 4;
      f32x4 y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const float* A = kk == 0 ? X0s : Ls;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(A + r * LD + 16 * blk + 4 * q);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int f = 16 * blk + 4 * q + j;          // weight row f * K + k (graphconv.py:243: column f*K + k)
            const f32x4 bv = *reinterpret_cast<const f32x4*>(a.weight + (int64_t)(f * K + kk) * Fout + cc);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = mfma4(av[j], bv[e], y[e]);
          }
        }
      }
      if (cok) {
        const f32x4 bc = *reinterpret_cast<const f32x4*>(a.bias + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 v = (f32x4){y[0][i] + bc.x, y[1][i] + bc.y, y[2][i] + bc.z, y[3][i] + bc.w};
          if (relu) v = (f32x4){fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
          *reinterpret_cast<f32x4*>(Yg + (int64_t)(4 * q + i) * a.ldy + col) = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward 1, reference mode (R = I: no gradient reaches the similarity): dX[I] += L[I, :] g_1 with L = I + L_orig (symmetric).
// (Paper mode: cloud_bwd1_paper_kernel below.  Its predecessor on the f32 MFMA instruction -- P = [g_1 | X_0][I] . [X_0 | g_1][J]^T
// as two chains of 16 v_mfma_f32_16x16x4_f32 per tile, 6144 matrix cycles per chunk and wave -- ran at 143-154 us.)
__global__ __launch_bounds__(256, 2) void cloud_bwd1_ref_kernel(SgcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = xcd_strip(a.n_slabs >> 2);
  if (strip < 0) return;
  const int slab = 4 * strip + wave;
  const WaveCtx c = wave_ctx(a, slab);
  const int n = c.n;
  float* Gs = smem;                                  // [64][LD]  g_1 rows of the chunk
  float* Ls = smem + CC * LD + wave * (16 * LD);     // [16][LD] this wave's strip of L
  const float* g1 = a.ws_g + (int64_t)c.row0 * a.fstride + a.foff;   // adjoint of X_1 (K = 2: plane 0)
  int l0ld;
  const float* L0d = dense_l0(a, c, &l0ld);
  const float* L0t = L0d + (int64_t)(c.i0 + 4 * q) * l0ld + 4 * r;
  struct Tiles { f32x4 l0[4]; };
  auto load_tiles = [&](int c0) {
    Tiles g;
#pragma unroll
    for (int i = 0; i < 4; ++i) g.l0[i] = *reinterpret_cast<const f32x4*>(L0t + (int64_t)i * l0ld + c0);
    return g;
  };
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  StageRegs preG = stage_load(g1, a.fstride, 0);
  Tiles cur = load_tiles(0);
#pragma unroll 1
  for (int c0 = 0; c0 < n; c0 += CC) {
    __syncthreads();
    stage_store(Gs, preG);
    __syncthreads();
    if (c0 + CC < n) preG = stage_load(g1, a.fstride, c0 + CC);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = c.i0 + 4 * q + i;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        Ls[(4 * q + i) * LD + 16 * t + r] = ((row == c0 + 4 * r + t) ? 1.f : 0.f) + cur.l0[i][t];   // lrelu(I) = I
    }
    if (c0 + CC < n) cur = load_tiles(c0 + CC);
    __builtin_amdgcn_wave_barrier();
    strip_mma(acc, Ls, Gs, r, q);     // dX[I] += L[I, chunk] g_1[chunk]
  }
  // dX rows of the strip (g_0 already there from the packed filter adjoint)
  float* dXg = a.dX + (int64_t)(c.row0 + c.i0) * a.ldx + a.foff;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4* p = reinterpret_cast<f32x4*>(dXg + (int64_t)(4 * q + i) * a.ldx + 4 * r);
    f32x4 v = *p;
    v.x += acc[0][i]; v.y += acc[1][i]; v.z += acc[2][i]; v.w += acc[3][i];
    *p = v;
  }
  if (lane == 0) {
    a.ws_pg[2 * (a.pg_slab_base + slab) + 0] = 0.f;
    a.ws_pg[2 * (a.pg_slab_base + slab) + 1] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// backward 1, paper mode, on the BF16 matrix pipe at fp32 accuracy ("bf16 x 6"): every fp32 operand is split EXACTLY into
// three bf16 pieces x = h + m + l (8 + 8 + 8 significant bits: h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); the
// differences are exact in fp32), and x y is taken as l h' + h l' + m m' + m h' + h m' + h h' (small terms first; the dropped
// m l', l m', l l' are below 2^-24 |x y|): every bf16 x bf16 product is exact in the fp32 accumulator, so the error is that of
// the fp32 additions alone -- on random data 0.2 x the error of the fmaf chain the f32 MFMA stands for (numpy emulation against
// float64; the parity tests hold the same bounds as before).  Six v_mfma_f32_16x16x32_bf16 (16 cycles each) cover 32 k: 96
// cycles instead of the 256 of eight v_mfma_f32_16x16x4_f32; the matrix time of a chunk goes 6144 -> 2304 cycles per wave.
//   P[I, chunk]   = [g_1 | X_0][I] . [X_0 | g_1][chunk]^T   A: the wave's rows, split once; B: rows of the chunk's images, 16-byte reads
//   dX[I]        += L[I, chunk] . g_1[chunk]                 A: the L strip (fp32 patch -> split per chunk); B: the SAME g_1 images read
//                                                            transposed (ds_read_b64_tr_b16: k = the image row; tools/ubench/tr_read_check.hip)
// The chunk's rows are staged as six bf16 images [64][72] (h / m / l of X_0 and g_1; image row 16 (rho % 4) + rho / 4 as
// everywhere in this file): 54 KB + the four L patches = 72 KB of LDS, two workgroups per CU.  B pieces are requested one
// 6-MFMA block ahead.  History: on the f32 MFMA instruction 143-154 us; P alone on bf16 x 6 with 8-wave workgroups (110 KB:
// ONE workgroup per CU, nothing to run behind its barriers) 211 us; this kernel 121 us.
namespace {
constexpr int LDH = 72;                          // bf16 elements per image row (144 bytes)
constexpr int PLANE_H = 64 * LDH;                // bf16 elements per plane
// B pieces [k = 8q .. 8q + 7 of the block][column r]: rows of an image (the column index is the image ROW: 16-byte reads) ...
__device__ __forceinline__ Pieces rows_B(const uint16_t* img, int off) {
  Pieces b;
  b.h = *reinterpret_cast<const u32x4*>(img + off);
  b.m = *reinterpret_cast<const u32x4*>(img + PLANE_H + off);
  b.l = *reinterpret_cast<const u32x4*>(img + 2 * PLANE_H + off);
  return b;
}
// ... or columns of it (k = the image row): two transposed reads of a 4 x 16 block each; `off` = this lane's address inside the
// first block (row 8q + r / 4, columns 4 (r % 4) ..), the second block 4 rows below
__device__ __forceinline__ u32x4 tr8(const uint16_t* p) {
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p + 4 * LDH));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  return (u32x4){l2.x, l2.y, h2.x, h2.y};
}
__device__ __forceinline__ Pieces cols_B(const uint16_t* img, int off) {
  Pieces b;
  b.h = tr8(img + off);
  b.m = tr8(img + PLANE_H + off);
  b.l = tr8(img + 2 * PLANE_H + off);
  return b;
}
}  // namespace

// FS: row stride of the packed feature arrays (= Fin: 64, or 128 with the kernel run once per 64-column half); ACC: second half
// (Q, ds and dalpha are linear in the per-half products: this run adds to the first one's)
// PART (128-feature layers, both 64-column feature halves per run; the recipe of the SGC_LL_Reslap passes below):
//   0  everything, one feature half per run (64-feature layers; the 128-feature layers until round 5: two runs, every tile read
//      twice + the read-modify-write of Q: 124 + 149 us)
//   1  P = g_1[I] X_0[J]^T over all 128 features (four k-blocks; X_0 images of both halves in the six planes) -> a.ws_P; no tile is read
//   2  everything else, element-wise from that tile (P_sym = P + P^T: P_ji read as 16-byte pieces along i), the L g_1 product over
//      both halves (g_1 images of both halves, two accumulator sets): r, L_orig, P, P^T in, Q out, one run
template <int FS, bool ACC, int PART = 0>
__global__ __launch_bounds__(256, 2) void cloud_bwd1_paper_kernel(SgcArgs a) {
  static_assert(PART == 0 || (FS == 128 && !ACC), "by parts: 128 features, both halves per run");
  constexpr bool DO_P = PART != 2;        // the P products are formed here
  constexpr bool DO_REST = PART != 1;     // tiles, element-wise work, the L g_1 product
  constexpr int NHX = PART == 2 ? 2 : 1;  // feature halves of the L g_1 product
  constexpr int NKB = PART == 1 ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = xcd_strip(a.n_slabs >> 2);
  if (strip < 0) return;
  const int slab = 4 * strip + wave;
  const WaveCtx c = wave_ctx(a, slab);
  const int n = c.n;
  uint16_t* PX = reinterpret_cast<uint16_t*>(smem);   // [3][64][LDH]  bf16 pieces of the chunk's X_0 rows  (part 2: g_1, second half)
  uint16_t* PG = PX + 3 * PLANE_H;                     // [3][64][LDH]  ... of its g_1 rows                (part 1: X_0, second half)
  float* Ls = smem + (6 * PLANE_H) / 2 + wave * (16 * LD);   // [16][LD] this wave's strip of L (fp32)
  const float alpha = a.alpha[0];
  const float* Xg = a.X + (int64_t)c.row0 * a.ldx + a.foff;
  const float* g1 = a.ws_g + (int64_t)c.row0 * FS + a.foff;   // adjoint of X_1 (K = 2: plane 0)
  int l0ld;
  const float* L0d = dense_l0(a, c, &l0ld);
  const int64_t o0 = c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r;
  const float* Rt = a.sv_r + o0;
  float* Qt = a.ws_dA + o0;
  const float* L0t = L0d + (int64_t)(c.i0 + 4 * q) * l0ld + 4 * r;
  const float* sjp = a.sv_s + c.row0 + 4 * r;
  float* Pw = PART == 1 ? a.ws_P + o0 : nullptr;
  const float* Ppt = PART == 2 ? a.ws_P + o0 : nullptr;
  const float* PpT = PART == 2 ? a.ws_P + c.toff + (int64_t)(4 * r) * n + c.i0 + 4 * q : nullptr;      // + (c0 + t) n: [t] over i
  float si[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) si[i] = a.sv_s[c.row0 + c.i0 + 4 * q + i];
  // A pieces of the wave's own rows: lane (r, q) holds [row r][32 kb + 8q .. + 7]
  Pieces Ag[NKB], Ax[2];
  if constexpr (DO_P) {
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      const float* gp = g1 + (int64_t)(c.i0 + r) * FS + 32 * kb + 8 * q;
      Ag[kb] = split3x8(*reinterpret_cast<const f32x4*>(gp), *reinterpret_cast<const f32x4*>(gp + 4));
      if constexpr (PART == 0) {
        const float* xp = Xg + (int64_t)(c.i0 + r) * a.ldx + 32 * kb + 8 * q;
        Ax[kb] = split3x8(*reinterpret_cast<const f32x4*>(xp), *reinterpret_cast<const f32x4*>(xp + 4));
      }
    }
  }
  struct Tiles { f32x4 rv[4], l0[4], sj; };
  auto load_tiles = [&](int c0) {
    Tiles g;
    g.sj = *reinterpret_cast<const f32x4*>(sjp + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g.rv[i] = *reinterpret_cast<const f32x4*>(Rt + (int64_t)i * n + c0);
      g.l0[i] = *reinterpret_cast<const f32x4*>(L0t + (int64_t)i * l0ld + c0);
    }
    return g;
  };
  // part 2: P_ij ([i][t]) and P_ji ([t][i]) of a chunk, requested at the top of the chunk (in flight behind the element-wise stage)
  struct PTiles { f32x4 pp[4], pt[4]; };
  auto load_p = [&](int c0) {
    PTiles g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g.pp[i] = PART == 2 ? *reinterpret_cast<const f32x4*>(Ppt + (int64_t)i * n + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
      g.pt[i] = PART == 2 ? *reinterpret_cast<const f32x4*>(PpT + (int64_t)(c0 + i) * n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    return g;
  };
  // staging: thread = (chunk row rho = tid / 4, features 16 (tid % 4) .. + 15) of both matrices.  What the planes hold: PX <- X_0,
  // PG <- g_1 of this run's half; part 1: X_0's two halves; part 2: g_1's second / first half.
  const int rho = tid >> 2, sg = tid & 3, irow = 16 * (rho & 3) + (rho >> 2);
  struct Stage { f32x4 g[4], x[4]; };
  auto stage_load16 = [&](int c0) {
    Stage st;
    const float* x0 = Xg + (int64_t)(c0 + rho) * a.ldx + 16 * sg;
    const float* g0 = g1 + (int64_t)(c0 + rho) * FS + 16 * sg;
    const float* xp = PART == 2 ? g0 + 64 : x0;
    const float* gp = PART == 1 ? x0 + 64 : g0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      st.g[m] = *reinterpret_cast<const f32x4*>(gp + 4 * m);
      st.x[m] = *reinterpret_cast<const f32x4*>(xp + 4 * m);
    }
    return st;
  };
  auto stage_store16 = [&](const Stage& st) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const Pieces pg = split3x8(st.g[2 * hf], st.g[2 * hf + 1]), px = split3x8(st.x[2 * hf], st.x[2 * hf + 1]);
      const int o = irow * LDH + 16 * sg + 8 * hf;
      *reinterpret_cast<u32x4*>(PG + o) = pg.h; *reinterpret_cast<u32x4*>(PG + PLANE_H + o) = pg.m; *reinterpret_cast<u32x4*>(PG + 2 * PLANE_H + o) = pg.l;
      *reinterpret_cast<u32x4*>(PX + o) = px.h; *reinterpret_cast<u32x4*>(PX + PLANE_H + o) = px.m; *reinterpret_cast<u32x4*>(PX + 2 * PLANE_H + o) = px.l;
    }
  };

  f32x4 acc[NHX][4];     // dX[I]: tile e of half h <-> features 64 h + 16 e + r
#pragma unroll
  for (int h = 0; h < NHX; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[h][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float ds[4] = {0.f, 0.f, 0.f, 0.f};
  float dalpha_part = 0.f;
  Stage pre = stage_load16(0);
  Tiles cur;
  if constexpr (DO_REST) cur = load_tiles(0);
  const int offB = r * LDH + 8 * q;                                // rows_B: image row 16 t + r, k 32 kb + 8q ..
  const int offT = (8 * q + (r >> 2)) * LDH + 4 * (r & 3);         // cols_B: row 32 kb + 8q + r / 4, columns 16 e + 4 (r % 4) ..
#pragma unroll 1
  for (int c0 = 0; c0 < n; c0 += CC) {
    __syncthreads();
    stage_store16(pre);
    __syncthreads();
    if (c0 + CC < n) pre = stage_load16(c0 + CC);
    f32x4 Sv[4];          // [i][t]: S_ij
    const PTiles pt_ = load_p(c0);
    f32x4 sj = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (DO_REST) {
      sj = cur.sj;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = c.i0 + 4 * q + i;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool diag = row == c0 + 4 * r + t;
          Sv[i][t] = diag ? 0.f : __builtin_amdgcn_exp2f(cur.rv[i][t] * NLOG2E);          // cloud_sim's instruction
          const float R = (diag ? 1.f : 0.f) - (si[i] * Sv[i][t]) * sj[t];                  // cloud_lap's instruction
          Ls[(4 * q + i) * LD + 16 * t + r] = lrelu(R, alpha) + cur.l0[i][t];
        }
      }
      if (c0 + CC < n) cur = load_tiles(c0 + CC);
      __builtin_amdgcn_wave_barrier();
    }
    // P[I, tile] = g_1[I] . X_0[tile]^T + X_0[I] . g_1[tile]^T (two accumulation chains), then the element-wise epilogue.
    // Blocks (t, kb, which product) in a row, the next block's B pieces requested before this block's MFMAs.
    f32x4 Qv[4];         // [i][t]  (part 1: P)
    Pieces Bn;
    if constexpr (PART == 0) Bn = rows_B(PX, offB);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 pa = (f32x4){0.f, 0.f, 0.f, 0.f}, pb = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (PART == 0) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          Pieces Bc = Bn;
          Bn = rows_B(PG, offB + 16 * t * LDH + 32 * kb);
          __builtin_amdgcn_sched_barrier(0);
          pa = mfma_x6(Ag[kb], Bc, pa);
          Bc = Bn;
          if (kb == 0) Bn = rows_B(PX, offB + 16 * t * LDH + 32);
          else if (t < 3) Bn = rows_B(PX, offB + 16 * (t + 1) * LDH);
          __builtin_amdgcn_sched_barrier(0);
          pb = mfma_x6(Ax[kb], Bc, pb);
        }
      } else if constexpr (PART == 1) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) pa = mfma_x6(Ag[kb], rows_B(kb < 2 ? PX : PG, offB + 16 * t * LDH + 32 * (kb & 1)), pa);
#pragma unroll
        for (int i = 0; i < 4; ++i) Qv[i][t] = pa[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { pa[i] = pt_.pp[i][t]; pb[i] = pt_.pt[t][i]; }
      }
      if constexpr (DO_REST) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = c.i0 + 4 * q + i;
          const bool diag = row == c0 + 4 * r + t;
          const float R = (diag ? 1.f : 0.f) - (si[i] * Sv[i][t]) * sj[t];
          const float P = pa[i] + pb[i];
          dalpha_part = __builtin_fmaf(P, fminf(R, 0.f), dalpha_part);
          const float Q = -(P * lrelu_grad(R, alpha));            // dA_ij + dA_ji, dA = -dR
          Qv[i][t] = Q;
          ds[i] = __builtin_fmaf(Q, Sv[i][t] * sj[t], ds[i]);      // ds_i = sum_j (dA_ij + dA_ji) S_ij s_j
        }
      }
    }
    if constexpr (PART == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Pw + (int64_t)i * n + c0) = Qv[i];
      continue;
    }
    // dX[I] += L[I, chunk] g_1[chunk]: A = the L patch (k = the patch column = the image row of the chunk's g_1), split here
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float* lp = Ls + r * LD + 32 * kb + 8 * q;
      const Pieces Al = split3x8(*reinterpret_cast<const f32x4*>(lp), *reinterpret_cast<const f32x4*>(lp + 4));
#pragma unroll
      for (int h = 0; h < NHX; ++h) {
        const uint16_t* img = h == 0 ? PG : PX;
        Pieces Tn = cols_B(img, offT + 32 * kb * LDH);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const Pieces Tc = Tn;
          if (e < 3) Tn = cols_B(img, offT + 32 * kb * LDH + 16 * (e + 1));
          __builtin_amdgcn_sched_barrier(0);
          acc[h][e] = mfma_x6(Al, Tc, acc[h][e]);
        }
      }
    }
    if (ACC) {    // second feature half of a 128-feature layer: Q is linear in P
#pragma unroll
      for (int i = 0; i < 4; ++i) Qv[i] += *reinterpret_cast<const f32x4*>(Qt + (int64_t)i * n + c0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Qt + (int64_t)i * n + c0) = Qv[i];
  }
  if constexpr (PART == 1) return;
  // dX rows of the strip (g_0 already there from the packed filter adjoint): lane (r, q) holds rows 4q + i, features 16 e + r
  float* dXg = a.dX + (int64_t)(c.row0 + c.i0) * a.ldx + a.foff;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int h = 0; h < NHX; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) dXg[(int64_t)(4 * q + i) * a.ldx + 64 * h + 16 * e + r] += acc[h][e][i];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float t = row16_sum(ds[i]);
    if (r == 0) {
      float* dd = a.ws_dd + c.row0 + c.i0 + 4 * q + i;
      if (ACC) t += *dd;                                      // (the first half left the raw ds)
      *dd = a.half_last ? degree_grad(si[i], t) : t;                 // dd_k = -1/2 s_k^3 ds_k
    }
  }
  const float da = wave_sum_all(dalpha_part);
  if (lane == 0) {
    float* pg = a.ws_pg + 2 * (a.pg_slab_base + slab);
    pg[0] = (ACC ? pg[0] : 0.f) + 0.5f * da;
    pg[1] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// SGC_LL_Reslap backward at point-cloud sizes.  With D = g_1 X_0^T + dL_sum (the gradient arriving on La; NOT symmetric) the
// chain of graphconv_reslap.py:184-198 reads, element by element but for two graph-wide inner products,
//   dLc   = D lrelu'(Lc)                       Lc = L_pre scL
//   dLpre = aL dLc - bL L_pre                  clip adjoint: aL = scL, bL = <dLc, L_pre> / ||L_pre||^3 if the clip is active
//   dL_prev = beta dLpre, dbeta = <dLpre, L_prev>, dalpha = <D, min(Lc, 0)> + <dLpre, min(R1, 0)>      R1 = R scR
//   dR1   = dLpre lrelu'(R1),  dR = aR dR1 - bR R        aR = scR, bR = <dR1, R> / ||R||^3 if active
// Everything behind dLc is LINEAR in the two inner products, so ONE pass over the pairs (pass A) collects eight sums per
// graph from which bL, bR, dalpha and dbeta follow, and the passes behind it are purely element-wise:
//   A  sums (per-strip partials; one product: g_1[I] X_0[J]^T)
//   B  (paper) Q = -(dR + dR^T) -> workspace, ds -> dd: as backward 1 of SGC_LL with D_ij + D_ji = P_ij + dL_sum_ij + dL_sum_ji
//      (R, L_pre, L_prev symmetric; the transposed dL_sum tile is read as 16-byte pieces along i)
//   C  dL_prev -> out, dX[I] += La[I, :] g_1 (La symmetric: cloud_path), the graph's dalpha / dbeta -> ws_pg
//   then cloud_bwd2_kernel as it is (Q, r, dd -> dZ).
// All products on the bf16 pipe at fp32 accuracy (cloud_bwd1_paper_kernel's helpers).
// Per-strip partial sums: [strip][8] = <dLc,Lpre>, <dLc,Lprev>, <Lpre,Lprev>, <dLc,min(R1,0)>, <Lpre,min(R1,0)>, <dLc,mR R>,
// <Lpre,mR R>, <D,min(Lc,0)>.
struct ResScal { float aL, bL, aR, bR, dalpha, dbeta; };
__device__ __forceinline__ ResScal res_scalars(const SgcArgs& a, const float* spart, int strip0, int ncb, int b, int lane) {
  float sm[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) sm[k] = wave_sum_all(lane < ncb ? spart[(int64_t)(strip0 + lane) * 8 + k] : 0.f);
  const float* sc = a.sv_scal + 4 * b;
  const float ssR = sc[0], scR = sc[1], ssL = sc[2], scL = sc[3];
  const float invL = 1.0f / sqrtf(ssL), invR = 1.0f / sqrtf(ssR);
  ResScal z;
  z.aL = scL;
  z.bL = invL <= 1.0f ? sm[0] * invL * invL * invL : 0.f;          // (the oracle's clip_backward)
  z.dbeta = z.aL * sm[1] - z.bL * sm[2];
  z.dalpha = sm[7] + (z.aL * sm[3] - z.bL * sm[4]);
  const float ipR = z.aL * sm[5] - z.bL * sm[6];
  z.aR = scR;
  z.bR = invR <= 1.0f ? ipR * invR * invR * invR : 0.f;
  return z;
}

// BOTH (pass A of a 128-feature layer): both 64-column halves of the feature matrices in ONE run -- the only product of the pass,
// g_1[I] X_0[J]^T, runs over all 128 features (four k-blocks; the second half's X_0 images take the LDS planes the pass has no g_1
// images for), so L_pre / r / L_prev / dL_sum are read once instead of once per half (2 x 420 MB, 110 + 99 us before).
// Pass C of a 128-feature layer in one run per PART: its two jobs share nothing but the L_pre tile, and together they would need the
// g_1 AND X_0 images of both halves (12 planes: one workgroup per CU), so with BOTH they are two launches -- CPART 1: dX[I] += La[I, :] g_1
// over all 128 features (g_1 images of both halves, two accumulator sets; + the graph's dalpha / dbeta), CPART 2 (only with a previous
// Laplacian): dL_prev from D = g_1 X_0^T over four k-blocks, as pass A.  4 tile passes instead of 6, no read-modify-write of dL_prev.
// FROMP (pass B; with BOTH): the product P = g_1 X_0^T is not formed here but READ from the tile part 2 of pass C wrote (a.ws_P,
// all 128 features; that part runs first then) -- and read again transposed for X_0 g_1^T = P^T: the pass is element-wise, no images,
// no matrix instructions, one run.  (Pass B with the products: twice 256 registers' worth of operand pieces for both halves, or one
// run per half = every tile read twice + the read-modify-write of Q: 2 x 135 us.)
template <int PASS, bool PAPER, int FS = 64, bool ACC = false, bool BOTH = false, int CPART = 0, bool FROMP = false>   // PASS 0 = A, 1 = B, 2 = C; FS / ACC: see cloud_bwd1_paper_kernel
__global__ __launch_bounds__(256, 2) void cloud_rbwd_kernel(SgcArgs a, float* spart) {
  static_assert(!BOTH || (FS == 128 && !ACC && (PASS == 0 || FROMP || (PASS == 2 && CPART != 0))), "both halves in one run: pass A, pass C by parts, passes from P");
  static_assert(CPART == 0 || (PASS == 2 && BOTH), "pass C by parts goes with BOTH");
  static_assert(!FROMP || (BOTH && PASS == 1), "from the P tile: pass B");
  constexpr bool DO_L = PASS == 2 && CPART != 2;      // the La g_1 product, dX, the graph's dalpha / dbeta
  constexpr bool DO_D = PASS != 2 || CPART != 1;      // the D = g_1 X_0^T product and what hangs on it
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[4 * 8];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = xcd_strip(a.n_slabs >> 2);
  if (strip < 0) return;
  const int slab = 4 * strip + wave;
  const WaveCtx c = wave_ctx(a, slab);
  const int n = c.n;
  const bool has_prev = a.L_prev != nullptr, has_dls = a.dL_sum != nullptr && !ACC;
  constexpr bool NEED_X = !FROMP;                     // X_0 images: the B operand of g_1[I] X_0[J]^T
  constexpr bool NEED_G = !FROMP && (PASS != 0 || BOTH);   // g_1 images: X_0[I] g_1[J]^T (B), La g_1 (C); BOTH: the second half's rows
  // pass C without a previous Laplacian has no dL_prev to write: only the La g_1 product is left
  const bool want_D = DO_D && (PASS != 2 || has_prev || (CPART == 2 && a.ws_P != nullptr));
  uint16_t* PX = reinterpret_cast<uint16_t*>(smem);   // [3][64][LDH]
  uint16_t* PG = PX + 3 * PLANE_H;
  float* Ls = smem + (6 * PLANE_H) / 2 + wave * (16 * LD);
  const float alpha = a.alpha[0];
  const float* Xg = a.X + (int64_t)c.row0 * a.ldx + a.foff;
  const float* g1 = a.ws_g + (int64_t)c.row0 * FS + a.foff;
  const float* scal = a.sv_scal + 4 * c.b;
  const float scR = scal[1], scL = scal[3];
  const int64_t o0 = c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r;
  const float* Rt = a.sv_r + o0;
  const float* Lpt = a.sv_Lpre + o0;
  const float* Pvt = has_prev ? a.L_prev + o0 : nullptr;
  const float* Dst = has_dls ? a.dL_sum + o0 : nullptr;
  const float* DsT = has_dls ? a.dL_sum + c.toff + (int64_t)(4 * r) * n + c.i0 + 4 * q : nullptr;   // + (c0 + t) n: [t] over i
  float* Qt = a.ws_dA + o0;
  float* DPt = has_prev ? a.dL_prev + o0 : nullptr;
  const float* sjp = a.sv_s + c.row0 + 4 * r;
  float si[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) si[i] = PAPER ? a.sv_s[c.row0 + c.i0 + 4 * q + i] : 0.f;
  ResScal z;
  z.aL = z.bL = z.aR = z.bR = z.dalpha = z.dbeta = 0.f;
  if constexpr (PASS != 0) z = res_scalars(a, spart, strip - (a.slab_row[4 * strip] >> 6), n >> 6, c.b, lane);
  const float beta = has_prev ? a.beta[0] : 0.f;
  // A pieces of the wave's own rows
  constexpr int NKB = BOTH ? 4 : 2;
  Pieces Ag[NKB], Ax[2];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const float* gp = g1 + (int64_t)(c.i0 + r) * FS + 32 * kb + 8 * q;
    Ag[kb] = split3x8(*reinterpret_cast<const f32x4*>(gp), *reinterpret_cast<const f32x4*>(gp + 4));
    if constexpr (PASS == 1) {
      const float* xp = Xg + (int64_t)(c.i0 + r) * a.ldx + 32 * kb + 8 * q;
      Ax[kb] = split3x8(*reinterpret_cast<const f32x4*>(xp), *reinterpret_cast<const f32x4*>(xp + 4));
    }
  }
  const int rho = tid >> 2, sg = tid & 3, irow = 16 * (rho & 3) + (rho >> 2);
  struct Stage { f32x4 g[4], x[4]; };
  auto stage_load16 = [&](int c0) {
    Stage st;
    // what the PX / PG planes hold: X_0 / g_1 of this run's half; BOTH, pass A and part 2 of C: X_0's two halves; part 1 of C:
    // g_1's second / first half
    const float* x0 = Xg + (int64_t)(c0 + rho) * a.ldx + 16 * sg;
    const float* g0 = g1 + (int64_t)(c0 + rho) * FS + 16 * sg;
    const float* xp = (BOTH && CPART == 1) ? g0 + 64 : x0;
    const float* gp = !BOTH ? g0 : (CPART == 1 ? g0 : x0 + 64);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (NEED_G) st.g[m] = *reinterpret_cast<const f32x4*>(gp + 4 * m);
      if (NEED_X) st.x[m] = *reinterpret_cast<const f32x4*>(xp + 4 * m);
    }
    return st;
  };
  auto stage_store16 = [&](const Stage& st) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int o = irow * LDH + 16 * sg + 8 * hf;
      if (NEED_G) {
        const Pieces pg = split3x8(st.g[2 * hf], st.g[2 * hf + 1]);
        *reinterpret_cast<u32x4*>(PG + o) = pg.h; *reinterpret_cast<u32x4*>(PG + PLANE_H + o) = pg.m; *reinterpret_cast<u32x4*>(PG + 2 * PLANE_H + o) = pg.l;
      }
      if (NEED_X) {
        const Pieces px = split3x8(st.x[2 * hf], st.x[2 * hf + 1]);
        *reinterpret_cast<u32x4*>(PX + o) = px.h; *reinterpret_cast<u32x4*>(PX + PLANE_H + o) = px.m; *reinterpret_cast<u32x4*>(PX + 2 * PLANE_H + o) = px.l;
      }
    }
  };
  // the tiles an element needs BEFORE the products (prefetched a chunk ahead into the same registers) ...
  struct Early { f32x4 rv[4], lp[4], sj; };
  auto load_early = [&](int c0) {
    Early g;
    g.sj = PAPER ? *reinterpret_cast<const f32x4*>(sjp + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g.rv[i] = (PAPER && PASS != 2) ? *reinterpret_cast<const f32x4*>(Rt + (int64_t)i * n + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
      g.lp[i] = *reinterpret_cast<const f32x4*>(Lpt + (int64_t)i * n + c0);
    }
    return g;
  };
  // ... and the ones only the epilogue needs (requested at the top of the chunk, in flight behind the products)
  // part 2 of pass C (both halves) leaves P = g_1 X_0^T for the element-wise pass B behind it.  (Pass A forms the same product
  // but has no register to spare: with the sixteen of a held-back tile it spilled 44, storing four bytes at a time 23 -- 115 -> 199 us.)
  float* Pw = (PASS == 2 && CPART == 2 && !FROMP && a.ws_P != nullptr) ? a.ws_P + o0 : nullptr;
  const float* Ppt = FROMP ? a.ws_P + o0 : nullptr;
  const float* PpT = FROMP ? a.ws_P + c.toff + (int64_t)(4 * r) * n + c.i0 + 4 * q : nullptr;      // + (c0 + t) n: [t] over i
  struct Late { f32x4 pv[4], ds[4], dt[4], pp[4], pt[4]; };
  auto load_late = [&](int c0) {
    Late g;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g.pp[i] = FROMP ? *reinterpret_cast<const f32x4*>(Ppt + (int64_t)i * n + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
      g.pt[i] = (FROMP && PASS == 1) ? *reinterpret_cast<const f32x4*>(PpT + (int64_t)(c0 + i) * n) : (f32x4){0.f, 0.f, 0.f, 0.f};
      g.pv[i] = (PASS == 0 && has_prev) ? *reinterpret_cast<const f32x4*>(Pvt + (int64_t)i * n + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
      g.ds[i] = (has_dls && (PASS != 2 || has_prev)) ? *reinterpret_cast<const f32x4*>(Dst + (int64_t)i * n + c0) : (f32x4){0.f, 0.f, 0.f, 0.f};
      g.dt[i] = (PASS == 1 && has_dls) ? *reinterpret_cast<const f32x4*>(DsT + (int64_t)(c0 + i) * n) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    return g;
  };
  constexpr int NHC = BOTH ? 2 : 1;
  f32x4 acc[NHC][4];     // pass C: dX[I], tile e of half h <-> features 64 h + 16 e + r
#pragma unroll
  for (int h = 0; h < NHC; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[h][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dsum[4] = {0.f, 0.f, 0.f, 0.f};
  float sm[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  Stage pre = stage_load16(0);
  Early cur = load_early(0);
  const int offB = r * LDH + 8 * q;
  const int offT = (8 * q + (r >> 2)) * LDH + 4 * (r & 3);
#pragma unroll 1
  for (int c0 = 0; c0 < n; c0 += CC) {
    if constexpr (!FROMP) {
      __syncthreads();
      stage_store16(pre);
      __syncthreads();
      if (c0 + CC < n) pre = stage_load16(c0 + CC);
    }
    const Late late = load_late(c0);
    // element-wise, before the products: S, R from r; Lc, La from L_pre
    f32x4 Sv[4], Lp[4];
    const f32x4 sj = cur.sj;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = c.i0 + 4 * q + i;
      Lp[i] = cur.lp[i];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const bool diag = row == c0 + 4 * r + t;
        Sv[i][t] = (!PAPER || PASS == 2 || diag) ? 0.f : __builtin_amdgcn_exp2f(cur.rv[i][t] * NLOG2E);
        if constexpr (DO_L) Ls[(4 * q + i) * LD + 16 * t + r] = lrelu(Lp[i][t] * scL, alpha);      // La
      }
    }
    if (c0 + CC < n) cur = load_early(c0 + CC);
    if constexpr (DO_L) __builtin_amdgcn_wave_barrier();
    // L product of pass C first (its A operand, the patch, is the only consumer of La)
    if constexpr (DO_L) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        const float* lp = Ls + r * LD + 32 * kb + 8 * q;
        const Pieces Al = split3x8(*reinterpret_cast<const f32x4*>(lp), *reinterpret_cast<const f32x4*>(lp + 4));
#pragma unroll
        for (int h = 0; h < NHC; ++h)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[h][e] = mfma_x6(Al, cols_B(h == 0 ? PG : PX, offT + 32 * kb * LDH + 16 * e), acc[h][e]);
      }
    }
    if constexpr (!DO_D) continue;
    // tile by tile: the product(s), then the element-wise part behind them
    f32x4 outv[4];
    f32x4 pout[(PASS == 2 && CPART == 2 && !FROMP) ? 4 : 1];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 pa = (f32x4){0.f, 0.f, 0.f, 0.f}, pb = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (FROMP) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { pa[i] = late.pp[i][t]; pb[i] = late.pt[t][i]; }
      } else if (want_D) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
          pa = mfma_x6(Ag[kb], rows_B(kb < 2 ? PX : PG, offB + 16 * t * LDH + 32 * (kb & 1)), pa);
          if constexpr (PASS == 1) pb = mfma_x6(Ax[kb], rows_B(PG, offB + 16 * t * LDH + 32 * kb), pb);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = c.i0 + 4 * q + i;
        const bool diag = row == c0 + 4 * r + t;
        const float lpre = Lp[i][t];
        const float Lc = lpre * scL;
        const float mL = lrelu_grad(Lc, alpha);
        float R = diag ? 1.f : 0.f;
        if (PAPER && PASS != 2) R = R - (si[i] * Sv[i][t]) * sj[t];
        const float R1 = R * scR;
        const float mR = lrelu_grad(R1, alpha);
        // (second feature half of a 128-feature layer, ACC: only the terms linear in this half's products; dL_sum and
        //  everything that does not depend on D went in with the first half)
        if constexpr (PASS == 0) {
          const float D = pa[i] + late.ds[i][t];
          const float dLc = D * mL;
          const float lprev = late.pv[i][t];
          const float nR = fminf(R1, 0.f), wR = mR * R;
          sm[0] = __builtin_fmaf(dLc, lpre, sm[0]);
          sm[1] = __builtin_fmaf(dLc, lprev, sm[1]);
          sm[3] = __builtin_fmaf(dLc, nR, sm[3]);
          sm[5] = __builtin_fmaf(dLc, wR, sm[5]);
          sm[7] = __builtin_fmaf(D, fminf(Lc, 0.f), sm[7]);
          if (!ACC) {
            sm[2] = __builtin_fmaf(lpre, lprev, sm[2]);
            sm[4] = __builtin_fmaf(lpre, nR, sm[4]);
            sm[6] = __builtin_fmaf(lpre, wR, sm[6]);
          }
        } else if constexpr (PASS == 1) {
          const float Dsym = (pa[i] + pb[i]) + (late.ds[i][t] + late.dt[t][i]);
          const float dLp2 = ACC ? (z.aL * mL) * Dsym : (z.aL * mL) * Dsym - 2.f * (z.bL * lpre);
          const float Q = ACC ? -(z.aR * (dLp2 * mR)) : -(z.aR * (dLp2 * mR) - 2.f * (z.bR * R));   // dA_ij + dA_ji, dA = -dR
          outv[i][t] = Q;
          dsum[i] = __builtin_fmaf(Q, Sv[i][t] * sj[t], dsum[i]);         // ds_i = sum_j (dA_ij + dA_ji) S_ij s_j
        } else {
          if constexpr (CPART == 2 && !FROMP) pout[i][t] = pa[i];
          const float D = pa[i] + late.ds[i][t];
          outv[i][t] = ACC ? beta * ((z.aL * mL) * D) : beta * ((z.aL * mL) * D - z.bL * lpre);   // dL_prev = beta dLpre
        }
      }
    }
    if constexpr (PASS == 1) {
      if (ACC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) outv[i] += *reinterpret_cast<const f32x4*>(Qt + (int64_t)i * n + c0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Qt + (int64_t)i * n + c0) = outv[i];
    }
    if constexpr (PASS == 2 && CPART == 2 && !FROMP) {
      if (Pw != nullptr) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Pw + (int64_t)i * n + c0) = pout[i];
      }
    }
    if constexpr (PASS == 2 && DO_D) {
      if (has_prev) {
        if (ACC) {
#pragma unroll
          for (int i = 0; i < 4; ++i) outv[i] += *reinterpret_cast<const f32x4*>(DPt + (int64_t)i * n + c0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_nontemporal_store(outv[i], reinterpret_cast<f32x4*>(DPt + (int64_t)i * n + c0));
      }
    }
  }
  if constexpr (PASS == 0) {
    // the strip's eight partial sums: lanes, then the four waves, in a fixed order
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float v = wave_sum_all(sm[k]);
      if (lane == 0) red[wave * 8 + k] = v;
    }
    __syncthreads();
    if (tid < 8) {
      float* sp = spart + (int64_t)strip * 8 + tid;
      *sp = (ACC ? *sp : 0.f) + ((red[tid] + red[8 + tid]) + (red[16 + tid] + red[24 + tid]));
    }
  }
  if constexpr (PASS == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = row16_sum(dsum[i]);
      if (r == 0) {
        float* dd = a.ws_dd + c.row0 + c.i0 + 4 * q + i;
        if (ACC) t += *dd;                                    // (the first half left the raw ds)
        *dd = a.half_last ? degree_grad(si[i], t) : t;               // dd_k = -1/2 s_k^3 ds_k
      }
    }
  }
  if constexpr (DO_L) {
    float* dXg = a.dX + (int64_t)(c.row0 + c.i0) * a.ldx + a.foff;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int h = 0; h < NHC; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) dXg[(int64_t)(4 * q + i) * a.ldx + 64 * h + 16 * e + r] += acc[h][e][i];
    if (lane == 0) {   // the graph's dalpha / dbeta once, at its first slab
      const bool first = c.i0 == 0;
      a.ws_pg[2 * (a.pg_slab_base + slab) + 0] = first ? z.dalpha : 0.f;
      a.ws_pg[2 * (a.pg_slab_base + slab) + 1] = (first && has_prev) ? z.dbeta : 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward 2 (paper mode): c_ij = -S_ij (Q_ij s_i s_j + dd_i + dd_j) / r_ij (0 where r_ij = 0 or S_ij = 0),
// dZ[I] = rowsum(c) Z[I] - c Z, dX[I] += dZ[I] M_L^T.
// NH: feature halves per run, as cloud_lap_kernel (2: the Q and r tiles are read once for both halves of dZ = rowsum(c) Z - c Z).
template <int NH = 1>
__global__ __launch_bounds__(256) void cloud_bwd2_kernel(SgcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = xcd_strip(a.n_slabs >> 2);
  if (strip < 0) return;
  const WaveCtx c = wave_ctx(a, 4 * strip + wave);
  const int n = c.n;
  float* Zs = smem;                                  // [NH][64][LD]  Z rows of the chunk
  float* Cs = smem + NH * CC * LD + wave * (16 * LD);   // [16][LD]  this wave's strip of c
  const float* Zg = a.sv_Z + (int64_t)c.row0 * a.fstride + a.foff;
  const float* Rt = a.sv_r + c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r;
  const float* Qt = a.ws_dA + c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r;
  const float* sjp = a.sv_s + c.row0 + 4 * r;
  const float* ddp = a.ws_dd + c.row0 + 4 * r;
  float si[4], ddi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    si[i] = a.sv_s[c.row0 + c.i0 + 4 * q + i];
    ddi[i] = a.ws_dd[c.row0 + c.i0 + 4 * q + i];
  }
  struct Tiles { f32x4 rv[4], qv[4], sj, ddj; };
  auto load_tiles = [&](int c0) {
    Tiles g;
    g.sj = *reinterpret_cast<const f32x4*>(sjp + c0);
    g.ddj = *reinterpret_cast<const f32x4*>(ddp + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g.rv[i] = *reinterpret_cast<const f32x4*>(Rt + (int64_t)i * n + c0);
      g.qv[i] = *reinterpret_cast<const f32x4*>(Qt + (int64_t)i * n + c0);
    }
    return g;
  };
  f32x4 acc[NH][4];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[h][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  StageRegs pre[NH];
#pragma unroll
  for (int h = 0; h < NH; ++h) pre[h] = stage_load(Zg + 64 * h, a.fstride, 0);
  Tiles cur = load_tiles(0);
#pragma unroll 1
  for (int c0 = 0; c0 < n; c0 += CC) {
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NH; ++h) stage_store(Zs + h * CC * LD, pre[h]);
    __syncthreads();
    if (c0 + CC < n) {
#pragma unroll
      for (int h = 0; h < NH; ++h) pre[h] = stage_load(Zg + 64 * h, a.fstride, c0 + CC);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = c.i0 + 4 * q + i;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float rr = cur.rv[i][t];
        const float S = __builtin_amdgcn_exp2f(rr * NLOG2E);      // cloud_sim's instruction
        // dS_ij = dA_ij s_i s_j + dd_j, dr_ij = -S_ij dS_ij; the pair (i, j) contributes dr_ij + dr_ji
        const float dr = -(S * __builtin_fmaf(cur.qv[i][t] * si[i], cur.sj[t], ddi[i] + cur.ddj[t]));
        // r == 0 (coincident rows): the term is defined as 0; S == 0 (underflow): dr = 0, never inf * 0
        const bool ok = row != c0 + 4 * r + t && rr > 0.f && S != 0.f;
        const float cij = ok ? dr * __builtin_amdgcn_rcpf(rr) : 0.f;
        cs[i] += cij;
        Cs[(4 * q + i) * LD + 16 * t + r] = cij;
      }
    }
    if (c0 + CC < n) cur = load_tiles(c0 + CC);   // (into the registers just consumed; in flight behind the product)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < NH; ++h) strip_mma(acc[h], Cs, Zs + h * CC * LD, r, q);
  }
  // dZ_i = rowsum(c)_i Z_i - (c Z)_i -> workspace (for dM_L) and the A image of the product below
  float* dZg = a.ws_dZ + (int64_t)(c.row0 + c.i0) * a.fstride + a.foff;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float csum = row16_sum(cs[i]);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const f32x4 z = *reinterpret_cast<const f32x4*>(Zg + (int64_t)(c.i0 + 4 * q + i) * a.fstride + 64 * h + 4 * r);
      const f32x4 v = (f32x4){csum * z.x - acc[h][0][i], csum * z.y - acc[h][1][i], csum * z.z - acc[h][2][i], csum * z.w - acc[h][3][i]};
      *reinterpret_cast<f32x4*>(dZg + (int64_t)(4 * q + i) * a.fstride + 64 * h + 4 * r) = v;
      if (NH == 1) *reinterpret_cast<f32x4*>(Cs + (4 * q + i) * LD + 4 * r) = v;
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (NH != 1 || a.fstride != FIN) return;   // 128-feature layer: dX += dZ M_L^T is a packed-row launch behind it
  // dX[I] += dZ[I] M_L^T   (adjoint of graphconv.py:164): B[k][f] = M_L[f][k], the lane's M_L rows read along k; tile t
  // stands for the output columns 4r + t, so a lane's four results of a row are one 16-byte read-modify-write
  float* dXg = a.dX + (int64_t)(c.row0 + c.i0) * a.ldx + a.foff;
  f32x4 d[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(Cs + r * LD + 16 * blk + 4 * q);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.M_L + (int64_t)(4 * r + t) * FIN + 16 * blk + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) d[t] = mfma4(av[j], bv[j], d[t]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4* p = reinterpret_cast<f32x4*>(dXg + (int64_t)(4 * q + i) * a.ldx + 4 * r);
    f32x4 v = *p;
    v.x += d[0][i]; v.y += d[1][i]; v.z += d[2][i]; v.w += d[3][i];
    *p = v;
  }
}

// backward 2 with the product c Z on the BF16 matrix pipe at fp32 accuracy (bf16_split.h; the construction of the L g_1 product of
// cloud_bwd1_paper_kernel): the chunk's Z rows are staged as three bf16 piece images per feature half and read TRANSPOSED
// (ds_read_b64_tr_b16: k = the image row = the chunk column of c), the wave's strip of c goes through its fp32 LDS patch and is
// split per chunk.  48 MFMAs of 16 cycles per chunk and feature half instead of 64 of 32 (the f32 kernel's matrix pipe was busy
// 34 % / 50 % of the time at 64 / 128 features: profiles/r05_modelnet*_counters.json).  Accumulator tile e <-> features 16 e + r.
// Used for the 128-feature layers only (launch_bwd2).
template <int NH = 1>
__global__ __launch_bounds__(256, 2) void cloud_bwd2_bf_kernel(SgcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = xcd_strip(a.n_slabs >> 2);
  if (strip < 0) return;
  const WaveCtx c = wave_ctx(a, 4 * strip + wave);
  const int n = c.n;
  uint16_t* PZ = reinterpret_cast<uint16_t*>(smem);               // [NH][3][64][LDH]  bf16 pieces of the chunk's Z rows
  float* Cs = smem + (NH * 3 * PLANE_H) / 2 + wave * (16 * LD);   // [16][LD]  this wave's strip of c (fp32)
  const float* Zg = a.sv_Z + (int64_t)c.row0 * a.fstride + a.foff;
  const float* Rt = a.sv_r + c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r;
  const float* Qt = a.ws_dA + c.toff + (int64_t)(c.i0 + 4 * q) * n + 4 * r;
  const float* sjp = a.sv_s + c.row0 + 4 * r;
  const float* ddp = a.ws_dd + c.row0 + 4 * r;
  float si[4], ddi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    si[i] = a.sv_s[c.row0 + c.i0 + 4 * q + i];
    ddi[i] = a.ws_dd[c.row0 + c.i0 + 4 * q + i];
  }
  struct Tiles { f32x4 rv[4], qv[4], sj, ddj; };
  auto load_tiles = [&](int c0) {
    Tiles g;
    g.sj = *reinterpret_cast<const f32x4*>(sjp + c0);
    g.ddj = *reinterpret_cast<const f32x4*>(ddp + c0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      g.rv[i] = *reinterpret_cast<const f32x4*>(Rt + (int64_t)i * n + c0);
      g.qv[i] = *reinterpret_cast<const f32x4*>(Qt + (int64_t)i * n + c0);
    }
    return g;
  };
  // staging: thread = (chunk row rho = tid / 4, features 16 (tid % 4) .. + 15 of each half); image row 16 (rho % 4) + rho / 4
  const int rho = tid >> 2, sg = tid & 3, irow = 16 * (rho & 3) + (rho >> 2);
  struct Stage { f32x4 z[NH][4]; };
  auto stage_load16 = [&](int c0) {
    Stage st;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const float* zp = Zg + (int64_t)(c0 + rho) * a.fstride + 64 * h + 16 * sg;
#pragma unroll
      for (int m = 0; m < 4; ++m) st.z[h][m] = *reinterpret_cast<const f32x4*>(zp + 4 * m);
    }
    return st;
  };
  auto stage_store16 = [&](const Stage& st) {
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const Pieces pz = split3x8(st.z[h][2 * hf], st.z[h][2 * hf + 1]);
        uint16_t* o = PZ + h * 3 * PLANE_H + irow * LDH + 16 * sg + 8 * hf;
        *reinterpret_cast<u32x4*>(o) = pz.h; *reinterpret_cast<u32x4*>(o + PLANE_H) = pz.m; *reinterpret_cast<u32x4*>(o + 2 * PLANE_H) = pz.l;
      }
  };
  f32x4 acc[NH][4];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[h][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  Stage pre = stage_load16(0);
  Tiles cur = load_tiles(0);
  const int offT = (8 * q + (r >> 2)) * LDH + 4 * (r & 3);         // cols_B: row 32 kb + 8q + r / 4, columns 16 e + 4 (r % 4) ..
#pragma unroll 1
  for (int c0 = 0; c0 < n; c0 += CC) {
    __syncthreads();
    stage_store16(pre);
    __syncthreads();
    if (c0 + CC < n) pre = stage_load16(c0 + CC);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = c.i0 + 4 * q + i;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float rr = cur.rv[i][t];
        const float S = __builtin_amdgcn_exp2f(rr * NLOG2E);      // cloud_sim's instruction
        // dS_ij = dA_ij s_i s_j + dd_j, dr_ij = -S_ij dS_ij; the pair (i, j) contributes dr_ij + dr_ji
        const float dr = -(S * __builtin_fmaf(cur.qv[i][t] * si[i], cur.sj[t], ddi[i] + cur.ddj[t]));
        // r == 0 (coincident rows): the term is defined as 0; S == 0 (underflow): dr = 0, never inf * 0
        const bool ok = row != c0 + 4 * r + t && rr > 0.f && S != 0.f;
        const float cij = ok ? dr * __builtin_amdgcn_rcpf(rr) : 0.f;
        cs[i] += cij;
        Cs[(4 * q + i) * LD + 16 * t + r] = cij;
      }
    }
    if (c0 + CC < n) cur = load_tiles(c0 + CC);   // (into the registers just consumed; in flight behind the product)
    __builtin_amdgcn_wave_barrier();
    // (c Z)[I] += c[I, chunk] Z[chunk]: A = the c patch (k = the patch column = the image row of the chunk's Z), split here
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float* lp = Cs + r * LD + 32 * kb + 8 * q;
      const Pieces Ac = split3x8(*reinterpret_cast<const f32x4*>(lp), *reinterpret_cast<const f32x4*>(lp + 4));
#pragma unroll
      for (int h = 0; h < NH; ++h) {
        const uint16_t* img = PZ + h * 3 * PLANE_H;
        Pieces Tn = cols_B(img, offT + 32 * kb * LDH);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const Pieces Tc = Tn;
          if (e < 3) Tn = cols_B(img, offT + 32 * kb * LDH + 16 * (e + 1));
          __builtin_amdgcn_sched_barrier(0);
          acc[h][e] = mfma_x6(Ac, Tc, acc[h][e]);
        }
      }
    }
  }
  // dZ_i = rowsum(c)_i Z_i - (c Z)_i -> workspace (for dM_L) and the A image of the product below
  float* dZg = a.ws_dZ + (int64_t)(c.row0 + c.i0) * a.fstride + a.foff;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float csum = row16_sum(cs[i]);
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int f = 64 * h + 16 * e + r;
        const float z = Zg[(int64_t)(c.i0 + 4 * q + i) * a.fstride + f];
        const float v = csum * z - acc[h][e][i];
        dZg[(int64_t)(4 * q + i) * a.fstride + f] = v;
        if (NH == 1) Cs[(4 * q + i) * LD + 16 * e + r] = v;
      }
  }
  __builtin_amdgcn_wave_barrier();
  if (NH != 1 || a.fstride != FIN) return;   // 128-feature layer: dX += dZ M_L^T is a packed-row launch behind it
  // dX[I] += dZ[I] M_L^T   (adjoint of graphconv.py:164), as in cloud_bwd2_kernel
  float* dXg = a.dX + (int64_t)(c.row0 + c.i0) * a.ldx + a.foff;
  f32x4 d[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    d[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(Cs + r * LD + 16 * blk + 4 * q);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.M_L + (int64_t)(4 * r + t) * FIN + 16 * blk + 4 * q);
#pragma unroll
      for (int j = 0; j < 4; ++j) d[t] = mfma4(av[j], bv[j], d[t]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x4* p = reinterpret_cast<f32x4*>(dXg + (int64_t)(4 * q + i) * a.ldx + 4 * r);
    f32x4 v = *p;
    v.x += d[0][i]; v.y += d[1][i]; v.z += d[2][i]; v.w += d[3][i];
    *p = v;
  }
}

// ------------------------------------------------------------------------------------------------
// The packed-row products of the layer at point-cloud sizes (tens of thousands of rows).  The general rows_gemm /
// param_grad kernels are built for a few thousand rows (1024-thread workgroups that stage whole operands, one round of
// loads): at 32768 rows they ran at 22 TFLOP/s (projection 16 us, filter adjoint 49 us, parameter gradients 44 us of a
// 550 us step).  Here a wave owns 16 rows, its A rows sit in a wave-private LDS image read 16 bytes wide, and the small B
// matrix (M_L, W^T: L1 / L2 resident) comes row-major straight from memory as B[k][4r .. 4r + 3] (lane r owns the output
// columns 4r .. 4r + 3: 16-byte stores).

// Z = X . M_L   (graphconv.py:164)
__global__ __launch_bounds__(256) void cloud_project_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ M,
                                                            float* __restrict__ Z) {
  __shared__ __attribute__((aligned(16))) float As_all[4][16 * LD];
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t row0 = (int64_t)blockIdx.x * CS + 16 * wave;
  float* As = As_all[wave];
#pragma unroll
  for (int m = 0; m < 4; ++m)
    *reinterpret_cast<f32x4*>(As + (q + 4 * m) * LD + 4 * r) = *reinterpret_cast<const f32x4*>(X + (row0 + q + 4 * m) * ldx + 4 * r);
  __builtin_amdgcn_wave_barrier();
  f32x4 acc[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int blk = 0; blk < 4; ++blk) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(As + r * LD + 16 * blk + 4 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(M + (16 * blk + 4 * q + j) * FIN + 4 * r);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = mfma4(av[j], bv[e], acc[e]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    *reinterpret_cast<f32x4*>(Z + (row0 + 4 * q + i) * FIN + 4 * r) = (f32x4){acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
}

// WT[k][c][f] = W[f * K + k][c]  (the filter adjoint's B operand, row-major in its contraction index c)
__global__ __launch_bounds__(256) void cloud_prep_kernel(const float* __restrict__ W, float* __restrict__ WT, int Fout, int K) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < K * Fout * FIN) {
    const int f = e % FIN, kc = e / FIN, c = kc % Fout, k = kc / Fout;
    WT[e] = W[(int64_t)(f * K + k) * Fout + c];
  }
}

// Gp = dY . act'(Y) -> workspace;  g_0 = Gp . W_0^T -> dX,  g_1 = Gp . W_1^T -> workspace   (adjoint of graphconv.py:238-251)
__global__ __launch_bounds__(256) void cloud_filter_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Y,
                                                               int ldy, int Fout, int relu, const float* __restrict__ WT,
                                                               float* __restrict__ Gp, float* __restrict__ dX, int ldx,
                                                               float* __restrict__ g1) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lda = Fout + 8;
  float* As = smem + wave * (16 * lda);
  const int64_t row0 = (int64_t)blockIdx.x * CS + 16 * wave;
  const int f4 = Fout >> 2;                                  // 16-byte groups per row
  for (int idx = lane; idx < 16 * f4; idx += 64) {
    const int row = idx / f4, c4 = idx - row * f4;
    const f32x4 d = *reinterpret_cast<const f32x4*>(dY + (row0 + row) * ldy + 4 * c4);
    const f32x4 y = *reinterpret_cast<const f32x4*>(Y + (row0 + row) * ldy + 4 * c4);
    f32x4 g = d;
    if (relu) g = (f32x4){y.x > 0.f ? d.x : 0.f, y.y > 0.f ? d.y : 0.f, y.z > 0.f ? d.z : 0.f, y.w > 0.f ? d.w : 0.f};
    *reinterpret_cast<f32x4*>(Gp + (row0 + row) * Fout + 4 * c4) = g;
    *reinterpret_cast<f32x4*>(As + row * lda + 4 * c4) = g;
  }
  __builtin_amdgcn_wave_barrier();
  // both products at once (the A fragment serves g_0 and g_1), the W^T fragments one 8-MFMA step ahead of their use
  f32x4 acc[2][4];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[k][e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float* B0 = WT + 4 * r + (int64_t)(4 * q) * FIN;
  const float* B1 = B0 + (int64_t)Fout * FIN;
  f32x4 b0 = *reinterpret_cast<const f32x4*>(B0), b1 = *reinterpret_cast<const f32x4*>(B1);
  const int nsteps = Fout >> 2;            // (cb, j) steps of 4 contraction indices per lane group
#pragma unroll 1
  for (int cb = 0; cb < Fout; cb += 16) {
    const f32x4 av = *reinterpret_cast<const f32x4*>(As + r * lda + cb + 4 * q);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int nxt = min((cb >> 2) + j + 1, nsteps - 1);            // next step: row 16 (nxt / 4) + 4 q + nxt % 4 of W^T
      const int64_t o = (int64_t)(16 * (nxt >> 2) + (nxt & 3)) * FIN;
      const f32x4 n0 = *reinterpret_cast<const f32x4*>(B0 + o), n1 = *reinterpret_cast<const f32x4*>(B1 + o);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][e] = mfma4(av[j], b0[e], acc[0][e]);
        acc[1][e] = mfma4(av[j], b1[e], acc[1][e]);
      }
      __builtin_amdgcn_sched_barrier(0);
      b0 = n0; b1 = n1;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<f32x4*>(dX + (row0 + 4 * q + i) * ldx + 4 * r) = (f32x4){acc[0][0][i], acc[0][1][i], acc[0][2][i], acc[0][3][i]};
    *reinterpret_cast<f32x4*>(g1 + (row0 + 4 * q + i) * FIN + 4 * r) = (f32x4){acc[1][0][i], acc[1][1][i], acc[1][2][i], acc[1][3][i]};
  }
}

// Row-chunk partials of the parameter gradients (param_grads.hip's layout: [chunk][dweight | dbias | dM_L]):
//   dweight[f*K + k][c] = sum_rows X_k[row][f] Gp[row][c], dbias[c] = sum_rows Gp[row][c], dM_L[f][g] = sum_rows X[row][f] dZ[row][g]
// -- A^T B with the row index as the MFMA k dimension.  Chunks of 256 rows (AGCN_CLOUD_PG_ROWS), and the OUTPUT split over
// workgroups: workgroup (chunk, k, z) takes the A tiles of X_k against the 64-column slice z of Gp (k = 0: also that slice of the
// bias) -- or, the extra slice z = Fout / 64 in paper mode, X against dZ (dM_L); wave w of it the A tile e = w (f = 4r + e).
// A chunk is walked in pieces of 64 rows staged in LDS with coalesced 16-byte loads, the next piece's loads in flight in
// registers behind the current piece's MFMAs; both operands are 16-byte LDS reads of "row 4u + q" (B tile e' <-> c = 64 z + 4r + e').
// History (32768 rows, 64 -> 128; partial + reduce): operands straight into the MFMA registers, 256 chunks: 46 + 25 us;
// LDS-staged 32-row pieces, one workgroup per (chunk, k): 25 + 19; this layout: 25 + 8 -- 11 us of the reduce had been ONE
// wave's serial loop over the 2048 slabs' dalpha partials (pg_wave_sum now), not the partials; 512-row chunks: 33 + 7.
template <int FOUT>
__global__ __launch_bounds__(256) void cloud_param_partial_kernel(ParamGradArgs a) {
  constexpr int PR = 64;   // rows per piece
  __shared__ __attribute__((aligned(16))) float Xs[PR * FIN], Gps[PR * 64];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NH = FOUT / 64;
  const int chunk = blockIdx.x, kx = blockIdx.y, h = blockIdx.z;
  const int r_begin = chunk * a.chunk_rows, r_end = min(a.rows, r_begin + a.chunk_rows);
  const bool with_m = h == NH;          // the extra z slice: X^T dZ instead of X_k^T Gp (only against X_0)
  if (with_m && kx != 0) return;
  const bool with_b = kx == 0 && !with_m;
  f32x4 accW[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) accW[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};
  struct Piece { f32x4 xa[4], gp[4]; };
  const float* Asrc = kx == 0 ? a.X : a.sv_X;
  const int lda = kx == 0 ? a.ldx : FIN;
  auto load_piece = [&](int p0) {        // rows p0 .. p0 + 63 (clamped to the batch; rows are multiples of 64)
    Piece g;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int id = tid + 256 * m;
      const int64_t row = min(p0 + (id >> 4), a.rows - 1);
      g.xa[m] = *reinterpret_cast<const f32x4*>(Asrc + row * lda + 4 * (id & 15));
      g.gp[m] = with_m ? *reinterpret_cast<const f32x4*>(a.dZ + row * FIN + 4 * (id & 15))
                       : *reinterpret_cast<const f32x4*>(a.Gp + row * FOUT + 64 * h + 4 * (id & 15));
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
      const f32x4 xa = *reinterpret_cast<const f32x4*>(Xs + row * FIN + 4 * r);
      const float xw = wave == 0 ? xa.x : wave == 1 ? xa.y : wave == 2 ? xa.z : xa.w;
      const f32x4 gp = *reinterpret_cast<const f32x4*>(Gps + row * 64 + 4 * r);
#pragma unroll
      for (int e = 0; e < 4; ++e) accW[e] = mfma4(xw, gp[e], accW[e]);
      if (with_b && wave == 0) bsum += gp;
    }
  }
  float* out = a.part + (int64_t)chunk * a.part_stride;
  // accumulator element i of tile (A tile w, B tile e'): f = 4 (4q + i) + w, c = 64 h + 4r + e'
  if (!with_m) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = 16 * q + 4 * i + wave;
      *reinterpret_cast<f32x4*>(out + (int64_t)(f * 2 + kx) * FOUT + 64 * h + 4 * r) = (f32x4){accW[0][i], accW[1][i], accW[2][i], accW[3][i]};
    }
  }
  if (with_b && wave == 0) {
    f32x4 t = bsum;
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {
      t.x += __shfl_xor(t.x, o, 64); t.y += __shfl_xor(t.y, o, 64); t.z += __shfl_xor(t.z, o, 64); t.w += __shfl_xor(t.w, o, 64);
    }
    if (q == 0) *reinterpret_cast<f32x4*>(out + (int64_t)2 * FIN * FOUT + 64 * h + 4 * r) = t;
  }
  if (with_m) {
    float* outM = out + (int64_t)2 * FIN * FOUT + FOUT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = 16 * q + 4 * i + wave;
      *reinterpret_cast<f32x4*>(outM + (int64_t)f * FIN + 4 * r) = (f32x4){accW[0][i], accW[1][i], accW[2][i], accW[3][i]};
    }
  }
}

// Fixed-order sum of the chunk partials: element e of [dweight | dbias | dM_L] = sum over EIGHT ranges of the chunks (one range
// per wave, all of its loads in flight at once -- up to 32 -- and added in chunk order), the ranges added in order; and the
// scalars (dalpha, dbeta: the per-slab partials in index order).  64 elements per 512-thread workgroup.
__global__ __launch_bounds__(512) void cloud_param_reduce_kernel(ParamGradArgs a) {
  __shared__ float red[8][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nW = (int64_t)a.K * a.Fin * a.Fout, nB = a.Fout, nM = a.paper ? (int64_t)a.Fin * a.Fin : 0;
  const int64_t total = nW + nB + nM;
  const int64_t e = (int64_t)blockIdx.x * 64 + lane;
  const int per = (a.n_chunks + 7) >> 3, c_begin = wave * per, c_end = min(a.n_chunks, c_begin + per);
  float s = 0.f;
  if (e < total && c_begin < c_end) {
    const float* p = a.part + e;
    for (int c = c_begin; c < c_end; c += 32) {
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = p[(int64_t)min(c + u, c_end - 1) * a.part_stride];
#pragma unroll
      for (int u = 0; u < 32; ++u) s += (c + u < c_end) ? v[u] : 0.f;
    }
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && e < total) {
    float t = red[0][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) t += red[w][lane];
    if (e < nW) a.dweight[e] = t;
    else if (e < nW + nB) a.dbias[e - nW] = t;
    else a.dM_L[e - nW - nB] = t;
  }
  if (blockIdx.x == 0 && (wave == 2 || wave == 3)) {   // scalars: one wave each, fixed order
    float* dst = wave == 2 ? a.dalpha : a.dbeta;
    if (dst != nullptr && a.pg != nullptr) {
      const float t = pg_wave_sum(a.pg, a.B, wave - 2, lane);
      if (lane == 0) dst[0] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
namespace {
bool cloud_env() {
  static const int off = [] { const char* e = std::getenv("AGCN_CLOUD"); return (e && e[0] == '0') ? 1 : 0; }();
  return !off;
}
template <class K, class... A>
int launch_cloud(K kernel, dim3 grid, int threads, size_t lds, int prof_id, const char* what, void* stream, A... args) {
  if (grid.x == 0) return AGCN_OK;
  prof_begin(prof_id, stream);
  hipLaunchKernelGGL(kernel, grid, dim3(threads), lds, static_cast<hipStream_t>(stream), args...);
  prof_end(stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s launch: %s", what, hipGetErrorString(e));
  return AGCN_OK;
}
}  // namespace

// Which slab batches these kernels take: the forward and the backward of a call pair must agree (same answer from the
// same arguments).  `quad` = every graph a multiple of 64 nodes.
bool cloud_path(const SgcArgs& a, bool quad) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  // SGC_LL_Reslap: the previous layer's Laplacian must be symmetric like L_orig (it is, whenever it is a Laplacian this
  // path or the reference produced from symmetric inputs): La is then symmetric and La^T g_1 is taken as La g_1
  const bool res_ok = a.variant != AGCN_SGC_LL_RESLAP ||
                      (a.sv_Lpre != nullptr && a.L_sum != nullptr && al16(a.sv_Lpre) && al16(a.L_prev) && al16(a.L_sum));
  return cloud_env() && quad && (a.Fin == FIN || a.Fin == 2 * FIN) && a.K == 2 && (a.variant == AGCN_SGC_LL || a.variant == AGCN_SGC_LL_RESLAP) && res_ok &&
         a.L0_dense != nullptr && (a.ldx & 3) == 0 && (a.ldy & 3) == 0 && (a.Fout & 15) == 0 && al16(a.X) && al16(a.weight) &&
         al16(a.bias) && al16(a.M_L) && (a.n_slabs & 3) == 0;
}
constexpr int SIM_RT = 32;   // rows per task of cloud_sim_kernel: 64 / SIM_RT partial-degree slots per row block (16: 72 + 9.5 us
                             // similarity + degree kernel, 32: 75 + 7.7, 64: 84 + 6.5)
// forward workspace: the partial degrees [n / SIM_RT][n] per graph, then (Reslap) two partial sums of squares per 64-row strip
size_t cloud_dpart_floats(int64_t tile_elems) { return (size_t)(tile_elems / SIM_RT + 64) + 2 * (size_t)(tile_elems / 4096 + 16); }

int launch_cloud_forward(const SgcArgs& a, float* dpart, int32_t max_n, int64_t tile_elems, void* stream) {
  const bool paper = a.mode == AGCN_MODE_PAPER;
  const unsigned strips = (unsigned)(a.n_slabs / 4);
  // 128-feature layers (the third / fourth convolution of the point-cloud network): Z = X M_L and the filter product are the
  // caller's packed-row launches; the similarity takes all 128 features, the Laplacian kernel runs once per 64-column half of X_0
  const bool wide = a.Fin == 2 * FIN;
  if (!wide) {
    int rc0 = launch_cloud(cloud_project_kernel, dim3(strips), 256, 0, PK_PROJECT, "cloud_project", stream, a.X, a.ldx, a.M_L, a.sv_Z);
    if (rc0 != AGCN_OK) return rc0;
  }
  const int ncb = std::max(1, (int)max_n / CS);   // column blocks of the largest graph
  const int n_cu = std::max(8, current_device_info().n_cus);   // (of the device this call runs on)
  const int ns = CS / SIM_RT;
  int rc;
  {
    const int t_quota = ncb * (ncb + 1) / 2;
    const int64_t tasks = (int64_t)a.B * t_quota * ns;
    const int per_q = (int)((tasks + 7) / 8), nquads = (per_q + 3) / 4;
    int qstride = 1;                               // a stride coprime with the number of quads, about a third of it
    for (int c = std::max(1, nquads / 3) | 1; c < nquads; c += 2) {
      int x = c, y = nquads;
      while (y) { const int tq = x % y; x = y; y = tq; }
      if (x == 1) { qstride = c; break; }
    }
    // one task per wave up to 12 workgroups per CU (the hardware's workgroup scheduler balances better than a static split
    // over resident waves: 75 against 79 us), beyond that the waves walk their lists
    const unsigned wgs = (unsigned)std::max<int64_t>(8, std::min<int64_t>((int64_t)12 * n_cu, (tasks + 3) / 4 + 7) / 8 * 8);
    const size_t lds = sizeof(float) * 4 * (paper ? 2560 : 1280);
    // 128 features: two passes of 64 at four waves per SIMD (AGCN_CLOUD_SIM2=0: all 128 in registers at two waves, for A/B runs)
    static const int sim2 = [] { const char* e = std::getenv("AGCN_CLOUD_SIM2"); return (e && e[0] == '0') ? 0 : 1; }();
    if (wide && sim2)
      rc = paper ? launch_cloud(cloud_sim2_kernel<true, SIM_RT>, dim3(wgs), 256, lds, PK_CLOUD_SIM, "cloud_sim", stream, a, dpart, t_quota, nquads, qstride)
                 : launch_cloud(cloud_sim2_kernel<false, SIM_RT>, dim3(wgs), 256, lds, PK_CLOUD_SIM, "cloud_sim", stream, a, dpart, t_quota, nquads, qstride);
    else if (wide)
      rc = paper ? launch_cloud(cloud_sim_kernel<true, SIM_RT, 128>, dim3(wgs), 256, lds, PK_CLOUD_SIM, "cloud_sim", stream, a, dpart, t_quota, nquads, qstride)
                 : launch_cloud(cloud_sim_kernel<false, SIM_RT, 128>, dim3(wgs), 256, lds, PK_CLOUD_SIM, "cloud_sim", stream, a, dpart, t_quota, nquads, qstride);
    else
    rc = paper ? launch_cloud(cloud_sim_kernel<true, SIM_RT>, dim3(wgs), 256, lds, PK_CLOUD_SIM, "cloud_sim", stream, a, dpart, t_quota, nquads, qstride)
               : launch_cloud(cloud_sim_kernel<false, SIM_RT>, dim3(wgs), 256, lds, PK_CLOUD_SIM, "cloud_sim", stream, a, dpart, t_quota, nquads, qstride);
  }
  if (rc != AGCN_OK) return rc;
  rc = launch_cloud(cloud_degree_kernel, dim3(strips), 64, 0, PK_CLOUD_DEG, "cloud_degree", stream, a, (const float*)dpart, ns);
  if (rc != AGCN_OK) return rc;
  const size_t lds = sizeof(float) * ((wide ? 2 : 1) * CC * LD + 4 * 16 * LD);
  const dim3 sgrid(xcd_grid((int)strips));
  float* rpart = dpart + (size_t)(tile_elems / SIM_RT + 64);
  float* lpart = rpart + (size_t)(tile_elems / 4096 + 16);
  if (a.variant == AGCN_SGC_LL_RESLAP) {
    if (paper) {
      rc = launch_cloud(cloud_rnorm_kernel, sgrid, 256, 0, PK_CLOUD_RNORM, "cloud_rnorm", stream, a, rpart);
      if (rc != AGCN_OK) return rc;
    }
    rc = paper ? launch_cloud(cloud_rlap1_kernel<true>, sgrid, 256, 0, PK_CLOUD_RLAP1, "cloud_rlap1", stream, a, (const float*)rpart, lpart)
               : launch_cloud(cloud_rlap1_kernel<false>, sgrid, 256, 0, PK_CLOUD_RLAP1, "cloud_rlap1", stream, a, (const float*)rpart, lpart);
    if (rc != AGCN_OK) return rc;
  }
  const bool res = a.variant == AGCN_SGC_LL_RESLAP;
  if (wide) {      // both feature halves in one run (NH = 2): the n x n tiles are read and written once
    SgcArgs ah = a;
    ah.foff = 0;
    if (res)
      rc = paper ? launch_cloud(cloud_lap_kernel<true, true, 2>, sgrid, 256, lds, PK_CLOUD_LAP, "cloud_lap", stream, ah, (const float*)rpart, (const float*)lpart)
                 : launch_cloud(cloud_lap_kernel<false, true, 2>, sgrid, 256, lds, PK_CLOUD_LAP, "cloud_lap", stream, ah, (const float*)rpart, (const float*)lpart);
    else
      rc = paper ? launch_cloud(cloud_lap_kernel<true, false, 2>, sgrid, 256, lds, PK_CLOUD_LAP, "cloud_lap", stream, ah, (const float*)nullptr, (const float*)nullptr)
                 : launch_cloud(cloud_lap_kernel<false, false, 2>, sgrid, 256, lds, PK_CLOUD_LAP, "cloud_lap", stream, ah, (const float*)nullptr, (const float*)nullptr);
    return rc;
  }
  {
    SgcArgs ah = a;
    ah.foff = 0;
    if (res)
      rc = paper ? launch_cloud(cloud_lap_kernel<true, true>, sgrid, 256, lds, PK_CLOUD_LAP, "cloud_lap", stream, ah, (const float*)rpart, (const float*)lpart)
                 : launch_cloud(cloud_lap_kernel<false, true>, sgrid, 256, lds, PK_CLOUD_LAP, "cloud_lap", stream, ah, (const float*)rpart, (const float*)lpart);
    else
      rc = paper ? launch_cloud(cloud_lap_kernel<true>, sgrid, 256, lds, PK_CLOUD_LAP, "cloud_lap", stream, ah, (const float*)nullptr, (const float*)nullptr)
                 : launch_cloud(cloud_lap_kernel<false>, sgrid, 256, lds, PK_CLOUD_LAP, "cloud_lap", stream, ah, (const float*)nullptr, (const float*)nullptr);
    if (rc != AGCN_OK) return rc;
  }
  return AGCN_OK;
}

// The row-chunk partials of the parameter gradients (the caller runs param_grads.hip's reduce kernel behind it).
bool cloud_param_grads_ok(const ParamGradArgs& a) {
  const bool wide = a.Fin == 2 * FIN && rows_wide_ok(a.rows, a.Fin, a.Fout, a.K, a.ldx, a.Fout);    // (rows_wide.hip)
  return cloud_env() && (a.Fin == FIN || wide) && a.K == 2 && (a.Fout == 64 || a.Fout == 128) && (a.rows & 63) == 0 && (a.ldx & 3) == 0 &&
         (a.chunk_rows & 3) == 0;
}
int launch_cloud_param_partials(const ParamGradArgs& a0, void* stream) {
  // few, long chunks (<= the caller's chunk count: its partial buffer is sized for that many)
  ParamGradArgs a = a0;
  static const int chunk_target = [] { const char* e = std::getenv("AGCN_CLOUD_PG_ROWS"); return e ? std::max(64, std::atoi(e)) : 256; }();
  a.n_chunks = std::max(1, std::min(a0.n_chunks, (a0.rows + chunk_target - 1) / chunk_target));
  a.chunk_rows = ((a0.rows + a.n_chunks - 1) / a.n_chunks + 63) / 64 * 64;
  const unsigned zm = a.paper ? 1u : 0u;   // the X^T dZ slice
  const bool split = (a0.rows & 31) == 0 && rows_split_ok(a0.rows, a0.Fin, a0.Fout, a0.K, a0.ldx, a0.Fout);
  if (split) a = a0;                        // (rows_split.hip chooses its own chunks, at most the caller's count)
  int rc = split ? launch_rows_split_param_partials(a, stream) : a.Fin == 2 * FIN ? launch_rows_wide_param_partials(a, stream) : a.Fout == 128 ? launch_cloud(cloud_param_partial_kernel<128>, dim3((unsigned)a.n_chunks, 2, 2 + zm), 256, 0, PK_PARAM_PARTIAL,
                                      "cloud_param_partial", stream, a)
                       : launch_cloud(cloud_param_partial_kernel<64>, dim3((unsigned)a.n_chunks, 2, 1 + zm), 256, 0, PK_PARAM_PARTIAL,
                                      "cloud_param_partial", stream, a);
  if (rc != AGCN_OK) return rc;
  const int64_t total = (int64_t)a.K * a.Fin * a.Fout + a.Fout + (a.paper ? (int64_t)a.Fin * a.Fin : 0);
  return launch_cloud(cloud_param_reduce_kernel, dim3((unsigned)((total + 63) / 64)), 512, 0, PK_PARAM_REDUCE, "cloud_param_reduce",
                      stream, a);
}

// backward 2: the product c Z on the bf16 matrix pipe at fp32 accuracy for the 128-feature layers (32 clouds x 1024 points:
// 114 -> 105 us; at 64 features the piece images and the splits cost more than the shorter matrix work gains: 84 -> 96 us, so those
// keep the fp32-MFMA kernel).  AGCN_CLOUD_BWD2_BF = 0: never, 2: always, for A/B runs.
static int launch_bwd2(const SgcArgs& ah, bool wide, dim3 grid, void* stream) {
  static const int bf = [] { const char* e = std::getenv("AGCN_CLOUD_BWD2_BF"); return e ? std::atoi(e) : 1; }();
  const int nh = wide ? 2 : 1;
  if (bf == 2 || (bf == 1 && wide)) {
    const size_t lds = sizeof(float) * (4 * 16 * LD) + sizeof(uint16_t) * 3 * PLANE_H * nh;
    return wide ? launch_cloud(cloud_bwd2_bf_kernel<2>, grid, 256, lds, PK_CLOUD_BWD2, "cloud_bwd2", stream, ah)
                : launch_cloud(cloud_bwd2_bf_kernel<1>, grid, 256, lds, PK_CLOUD_BWD2, "cloud_bwd2", stream, ah);
  }
  const size_t lds = sizeof(float) * (nh * CC * LD + 4 * 16 * LD);
  return wide ? launch_cloud(cloud_bwd2_kernel<2>, grid, 256, lds, PK_CLOUD_BWD2, "cloud_bwd2", stream, ah)
              : launch_cloud(cloud_bwd2_kernel<1>, grid, 256, lds, PK_CLOUD_BWD2, "cloud_bwd2", stream, ah);
}

// The packed filter adjoint of a 64-feature layer: Gp = dY act'(Y), g_0 = Gp W_0^T -> dX, g_1 = Gp W_1^T -> workspace.  On the bf16
// matrix pipe at fp32 accuracy (rows_split.hip: one launch builds the image of [W_0^T | W_1^T]) where the caller gave the image
// workspace, else cloud_filter_bwd_kernel behind the transposed copy of W (`WT`, or the one the cloud forward left in the reserve).
static int packed_filter_bwd(const SgcArgs& a, float* WT, unsigned strips, void* stream) {
  if (a.rs_img != nullptr && rows_split_ok(a.rows, a.Fin, a.Fout, a.K, a.ldx, a.ldy)) {
    int rc = launch_rows_split_bwd_images(a.M_L, a.weight, a.Fin, a.Fout, false, a.rs_img, stream);
    if (rc != AGCN_OK) return rc;
    return launch_rows_split_filter_bwd(a.dY, a.Y, a.ldy, a.activation == AGCN_ACT_RELU ? 1 : 0, a.ws_Gp, a.dX, a.ldx, a.ws_g, a.Fin, a.Fout,
                                        a.rows, a.rs_img, stream);
  }
  int rc = AGCN_OK;
  if (WT != nullptr)
    rc = launch_cloud(cloud_prep_kernel, dim3((unsigned)((2 * a.Fout * FIN + 255) / 256)), 256, 0, PK_FILTER_BWD, "cloud_prep", stream,
                      a.weight, WT, a.Fout, 2);
  else
    WT = a.sv_WT;
  if (rc != AGCN_OK) return rc;
  return launch_cloud(cloud_filter_bwd_kernel, dim3(strips), 256, sizeof(float) * 4 * 16 * (a.Fout + 8), PK_FILTER_BWD, "cloud_filter_bwd",
                      stream, a.dY, (const float*)a.Y, a.ldy, a.Fout, a.activation == AGCN_ACT_RELU ? 1 : 0, (const float*)WT, a.ws_Gp, a.dX,
                      a.ldx, a.ws_g);
}

// Gp, g_0 -> dX, g_1 (the packed filter adjoint), then the strip kernels.  `WT`: K * Fout * 64 floats of workspace for W^T, or
// nullptr when the forward of this call pair was the cloud forward (which left W^T in the reserve: a.sv_WT).
int launch_cloud_backward(const SgcArgs& a, float* WT, void* stream) {
  const bool paper = a.mode == AGCN_MODE_PAPER;
  const bool wide = a.Fin == 2 * FIN;      // (128 features: the packed filter adjoint was the caller's launch; two halves)
  const unsigned strips = (unsigned)(a.n_slabs / 4);
  if (a.dL_sum != nullptr) return fail(AGCN_EINVAL, "cloud kernels: a gradient of L_sum is not supported");
  if (!wide) {
    int rc0 = packed_filter_bwd(a, WT, strips, stream);
    if (rc0 != AGCN_OK) return rc0;
  }
  const dim3 grid(xcd_grid((int)strips));
  const size_t lds_f32 = sizeof(float) * (CC * LD + 4 * 16 * LD);
  const size_t lds_bf = sizeof(float) * (4 * 16 * LD) + sizeof(uint16_t) * 6 * PLANE_H;
  int rc = AGCN_OK;
  // 128 features, paper mode, a P tile in the workspace: backward 1 by parts (AGCN_CLOUD_BWD1_PARTS=0: once per half, for A/B runs)
  static const int parts = [] { const char* e = std::getenv("AGCN_CLOUD_BWD1_PARTS"); return (e && e[0] == '0') ? 0 : 1; }();
  const bool by_parts = wide && paper && parts && a.ws_P != nullptr;
  for (int half = 0; half < (wide ? 2 : 1) && rc == AGCN_OK; ++half) {
    SgcArgs ah = a;
    ah.foff = 64 * half; ah.half_acc = half; ah.half_last = (half == (wide ? 1 : 0)) ? 1 : 0;
    if (!paper)
      rc = launch_cloud(cloud_bwd1_ref_kernel, grid, 256, lds_f32, PK_CLOUD_BWD1, "cloud_bwd1", stream, ah);
    else if (!wide)
      rc = launch_cloud(cloud_bwd1_paper_kernel<64, false>, grid, 256, lds_bf, PK_CLOUD_BWD1, "cloud_bwd1", stream, ah);
    else if (by_parts) {     // both feature halves per run: P = g_1 X_0^T -> a.ws_P, then everything else from that tile
      if (half != 0) break;
      ah.half_last = 1;
      rc = launch_cloud(cloud_bwd1_paper_kernel<128, false, 1>, grid, 256, lds_bf, PK_CLOUD_BWD1, "cloud_bwd1", stream, ah);
      if (rc == AGCN_OK) rc = launch_cloud(cloud_bwd1_paper_kernel<128, false, 2>, grid, 256, lds_bf, PK_CLOUD_BWD1, "cloud_bwd1", stream, ah);
    } else
      rc = half == 0 ? launch_cloud(cloud_bwd1_paper_kernel<128, false>, grid, 256, lds_bf, PK_CLOUD_BWD1, "cloud_bwd1", stream, ah)
                     : launch_cloud(cloud_bwd1_paper_kernel<128, true>, grid, 256, lds_bf, PK_CLOUD_BWD1, "cloud_bwd1", stream, ah);
  }
  if (rc != AGCN_OK || !paper) return rc;
  {   // (128 features: both halves in one run, the Q and r tiles read once)
    SgcArgs ah = a;
    ah.foff = 0;
    rc = launch_bwd2(ah, wide, grid, stream);
  }
  return rc;
}

// SGC_LL_Reslap: the packed filter adjoint, then passes A, (B, backward 2 in paper mode,) C.  spart: 8 floats per 64-row strip.
int launch_cloud_backward_reslap(const SgcArgs& a, float* WT, float* spart, void* stream) {
  const bool paper = a.mode == AGCN_MODE_PAPER;
  const bool wide = a.Fin == 2 * FIN;      // (128 features: the packed filter adjoint was the caller's launch; two halves per pass)
  const unsigned strips = (unsigned)(a.n_slabs / 4);
  int rc = AGCN_OK;
  if (!wide) {
    rc = packed_filter_bwd(a, WT, strips, stream);
    if (rc != AGCN_OK) return rc;
  }
  const dim3 grid(xcd_grid((int)strips));
  const size_t lds = sizeof(float) * (4 * 16 * LD) + sizeof(uint16_t) * 6 * PLANE_H;
  const int nh = wide ? 2 : 1;
  auto args = [&](int half) {
    SgcArgs ah = a;
    ah.foff = 64 * half; ah.half_acc = half; ah.half_last = (half == nh - 1) ? 1 : 0;
    return ah;
  };
#define AGCN_RBWD(PASS, PAPER, ID, NAME)                                                                                            \
  for (int half = 0; half < nh && rc == AGCN_OK; ++half) {                                                                         \
    const SgcArgs ah = args(half);                                                                                                  \
    rc = !wide ? launch_cloud(cloud_rbwd_kernel<PASS, PAPER, 64, false>, grid, 256, lds, ID, NAME, stream, ah, spart)              \
       : half == 0 ? launch_cloud(cloud_rbwd_kernel<PASS, PAPER, 128, false>, grid, 256, lds, ID, NAME, stream, ah, spart)         \
                   : launch_cloud(cloud_rbwd_kernel<PASS, PAPER, 128, true>, grid, 256, lds, ID, NAME, stream, ah, spart);          \
  }
  // pass A of a 128-feature layer: both feature halves in one run (AGCN_CLOUD_RBWD_BOTH=0: once per half, for A/B runs)
  static const int both = [] { const char* e = std::getenv("AGCN_CLOUD_RBWD_BOTH"); return (e && e[0] == '0') ? 0 : 1; }();
  if (wide && both) {
    SgcArgs ah = args(0);
    ah.half_last = 1;
    rc = paper ? launch_cloud(cloud_rbwd_kernel<0, true, 128, false, true>, grid, 256, lds, PK_CLOUD_RBWD_A, "cloud_rbwdA", stream, ah, spart)
               : launch_cloud(cloud_rbwd_kernel<0, false, 128, false, true>, grid, 256, lds, PK_CLOUD_RBWD_A, "cloud_rbwdA", stream, ah, spart);
  } else if (paper) { AGCN_RBWD(0, true, PK_CLOUD_RBWD_A, "cloud_rbwdA") } else { AGCN_RBWD(0, false, PK_CLOUD_RBWD_A, "cloud_rbwdA") }
  if (rc != AGCN_OK) return rc;
  // 128 features, paper mode: part 2 of pass C runs FIRST and leaves P = g_1 X_0^T (all 128 features) for pass B, which is then
  // element-wise (P, P^T from the tile): one run instead of one per half
  const bool from_p = wide && both && paper && a.ws_P != nullptr;
  auto c_part2 = [&]() {
    SgcArgs ah = args(0);
    ah.half_last = 1;
    if (!from_p) ah.ws_P = nullptr;      // (nobody reads P then: reference mode has no pass B)
    return paper ? launch_cloud(cloud_rbwd_kernel<2, true, 128, false, true, 2>, grid, 256, lds, PK_CLOUD_RBWD_C, "cloud_rbwdC", stream, ah, spart)
                 : launch_cloud(cloud_rbwd_kernel<2, false, 128, false, true, 2>, grid, 256, lds, PK_CLOUD_RBWD_C, "cloud_rbwdC", stream, ah, spart);
  };
  if (from_p) {
    rc = c_part2();
    if (rc != AGCN_OK) return rc;
    SgcArgs ah = args(0);
    ah.half_last = 1;
    rc = launch_cloud(cloud_rbwd_kernel<1, true, 128, false, true, 0, true>, grid, 256, 0, PK_CLOUD_RBWD_B, "cloud_rbwdB", stream, ah, spart);
  } else if (paper) { AGCN_RBWD(1, true, PK_CLOUD_RBWD_B, "cloud_rbwdB") }
  if (rc != AGCN_OK) return rc;
  if (wide && both) {   // pass C by parts, both feature halves per run: dX += La g_1, and (with a previous Laplacian) dL_prev
    SgcArgs ah = args(0);
    ah.half_last = 1;
    rc = paper ? launch_cloud(cloud_rbwd_kernel<2, true, 128, false, true, 1>, grid, 256, lds, PK_CLOUD_RBWD_C, "cloud_rbwdC", stream, ah, spart)
               : launch_cloud(cloud_rbwd_kernel<2, false, 128, false, true, 1>, grid, 256, lds, PK_CLOUD_RBWD_C, "cloud_rbwdC", stream, ah, spart);
    if (rc == AGCN_OK && !from_p && a.L_prev != nullptr) rc = c_part2();
  } else if (paper) { AGCN_RBWD(2, true, PK_CLOUD_RBWD_C, "cloud_rbwdC") } else { AGCN_RBWD(2, false, PK_CLOUD_RBWD_C, "cloud_rbwdC") }
#undef AGCN_RBWD
  if (rc != AGCN_OK || !paper) return rc;
  {
    SgcArgs ah = a;
    ah.foff = 0;
    rc = launch_bwd2(ah, wide, grid, stream);
  }
  return rc;
}

}  // namespace agcn
