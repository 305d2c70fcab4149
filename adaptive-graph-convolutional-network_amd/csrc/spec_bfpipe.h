// Feature contractions of the per-graph kernels on the BF16 matrix pipe at fp32 accuracy (the construction of bf16_split.h:
// x = h + m + l exactly, six v_mfma_f32_16x16x32_bf16 per 32 k instead of eight v_mfma_f32_16x16x4_f32 per 32 k -- and THREE when
// one operand is bf16 data already: AgcnLayerDesc.feat_dtype = bf16, whose node features / masked upstream gradients are single
// pieces).
//
// Why it pays in kernels that are "latency-bound": on the CU that holds the batch's largest graph the MFMA phases ARE bound by the
// matrix pipe -- 16 waves share 4 SIMDs, a v_mfma_f32_16x16x4_f32 occupies its SIMD for 32 cycles and a tile's k-steps form ONE
// dependent accumulator chain (40 cycles per link): the filter phase of a 64-node graph is 16 tiles x 38 links = 2.0 us of pure
// matrix time per SIMD (measured 2.2 us).  v_mfma_f32_16x16x32_bf16 issues every 16 cycles, also on a single chain.
//
// Layouts (LDS):
//  * the PARAMETER operand (B[k][c]) as an image in MFMA operand order, built by the workgroup itself in its first phase (the
//    matrix is read once, coalesced, split in registers): tile (kb, ct) = 3 pieces x 64 lanes x 16 bytes; lane (r, q) holds
//    B[32 kb + 8 q + j][16 ct + r], j = 0 .. 7.  A wave reads a piece with ONE ds_read_b128 per lane, consecutive lanes
//    consecutive 16-byte words: conflict-free.
//  * the ROW operand (A[i][k]: node features and the like) as NP planes (h, m, l; one plane for bf16 data) of bf16 rows with a
//    row stride of BFP_S(KP) dwords == 8 (mod 16): lane (r, q) reads row r, k = 32 kb + 8 q .. + 7 with one ds_read_b128,
//    conflict-free under gfx950's 4 x 16-lane service groups of that instruction.
// Any assignment of the 32 k of a block to (q, j) is a valid contraction order as long as A and B agree; this one makes both
// reads 16 contiguous bytes.
#ifndef AGCN_SPEC_BFPIPE_H_
#define AGCN_SPEC_BFPIPE_H_

#include "bf16_split.h"

namespace agcn {
namespace spec {

// k extent of the images: the feature count rounded up to whole 32-k blocks
AGCN_HD constexpr int bfp_kp(int F) { return (F + 31) & ~31; }
// row stride of a piece plane in DWORDS (two bf16 each): >= KP / 2, == 8 (mod 16)
AGCN_HD constexpr int bfp_stride(int KP) { return ((KP / 2 - 8 + 15) / 16) * 16 + 8; }
// dwords of the operand-order image of a K x N parameter matrix (KP / 32 k-blocks, N / 16 column tiles, 3 pieces x 64 lanes x 4 dwords)
AGCN_HD constexpr int bfp_image_dwords(int KP, int NP16) { return (KP / 32) * (NP16 / 16) * 3 * 64 * 4; }

#if defined(__HIPCC__)
// Builds the image of B[k][c] = src[k * ks + c * cs] (k < K, c < N; zero beyond) for k-blocks [0, KP / 32) and column tiles
// [0, NP16 / 16): thread t < tiles * 64 fills the three pieces of (tile t / 64, lane t % 64).  Loads are clamped, selection
// happens afterwards (no load under a per-lane branch).  Returns after the LDS stores have been issued (caller's barrier).
// (called by EVERY thread, t clamped to the last image thread: a load under a per-lane branch would be waited for on the spot)
template <int K, int N, int KP, int NP16>
__device__ __forceinline__ void bfp_load_param(float (&v)[8], const float* src, int ks, int cs, int t) {
  constexpr int NT = NP16 / 16;
  t = min(t, (KP / 32) * NT * 64 - 1);
  const int lane = t & 63, tile = t >> 6, ct = tile % NT, kb = tile / NT;
  const int r = lane & 15, q = lane >> 4;
  const int c = 16 * ct + r, k0 = 32 * kb + 8 * q;
  const float* p = src + (int64_t)min(c, N - 1) * cs;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = p[(int64_t)min(k0 + j, K - 1) * ks];
}
template <int K, int N, int KP, int NP16>
__device__ __forceinline__ void bfp_store_param(u32x4* img, const float (&v)[8], int t) {
  constexpr int NT = NP16 / 16;
  const int lane = t & 63, tile = t >> 6, ct = tile % NT, kb = tile / NT;
  const int r = lane & 15, q = lane >> 4;
  const int c = 16 * ct + r, k0 = 32 * kb + 8 * q;
  float w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) w[j] = (c < N && k0 + j < K) ? v[j] : 0.f;
  const Pieces P = split3x8((f32x4){w[0], w[1], w[2], w[3]}, (f32x4){w[4], w[5], w[6], w[7]});
  u32x4* dst = img + (size_t)tile * 192 + lane;
  dst[0] = P.h; dst[64] = P.m; dst[128] = P.l;
}

// one fp32 value -> its NP bf16 pieces at element `e` of the planes (plane p at planes + p * plane_halfs, 16-bit elements)
template <int NP>
__device__ __forceinline__ void bfp_store_elem(uint16_t* planes, int plane_halfs, int e, float x) {
  const __bf16 h = (__bf16)x;
  const uint16_t hb = __builtin_bit_cast(uint16_t, h);
  planes[e] = hb;
  if constexpr (NP == 3) {
    const float r1 = x - __uint_as_float((uint32_t)hb << 16);
    const __bf16 m = (__bf16)r1;
    const uint16_t mb = __builtin_bit_cast(uint16_t, m);
    planes[plane_halfs + e] = mb;
    const float r2 = r1 - __uint_as_float((uint32_t)mb << 16);
    const __bf16 l = (__bf16)r2;
    planes[2 * plane_halfs + e] = __builtin_bit_cast(uint16_t, l);
  }
}

// two values of adjacent columns (c even, c + 1) -> one dword per plane at dword index `e`
template <int NP>
__device__ __forceinline__ void bfp_store_pair(uint32_t* planes, int plane_dwords, int e, float x0, float x1) {
  if constexpr (NP == 3) {
    uint32_t h, m, l;
    split3(x0, x1, h, m, l);
    planes[e] = h;
    planes[plane_dwords + e] = m;
    planes[2 * plane_dwords + e] = l;
  } else {
    const bf16x2 hv = (bf16x2){(__bf16)x0, (__bf16)x1};
    planes[e] = __builtin_bit_cast(uint32_t, hv);
  }
}
// A lane that holds columns `lane` (v0) and `lane + 64` (v1; CP2 = false: only v0) of one row pairs up with its neighbour: even
// lanes get the two consecutive columns (lane, lane + 1), odd lanes (64 + lane - 1, 64 + lane) -- ONE 4-byte LDS store per plane
// instead of two 2-byte ones.  Returns the (even) column of the pair in `col`, -1 when this lane has nothing to store.
// the neighbour lane's value (lane ^ 1) by a DPP quad permutation [1, 0, 3, 2]: a register move, not an LDS round trip
__device__ __forceinline__ float bfp_swap1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
template <bool CP2>
__device__ __forceinline__ void bfp_pair_columns(float v0, float v1, int lane, float& lo, float& hi, int& col) {
  const bool odd = (lane & 1) != 0;
  if constexpr (CP2) {
    const float recv = bfp_swap1(odd ? v0 : v1);
    lo = odd ? recv : v0;
    hi = odd ? v1 : recv;
    col = odd ? 63 + lane : lane;
  } else {
    const float recv = bfp_swap1(v0);
    lo = v0; hi = recv;
    col = odd ? -1 : lane;
  }
}

// acc += A[16 rows x 32 k of block kb] . B[tile (kb, ct)]: arow = plane 0 + (row of this lane) * S dwords; bt = image + tile * 192
// NP = 3: both operands three pieces (six products, small terms first); NP = 1: A is one piece (three products)
template <int NP>
__device__ __forceinline__ f32x4 bfp_mma(f32x4 acc, const uint32_t* arow, int plane_dwords, int kb, int q, const u32x4* bt, int lane) {
  const u32x4 Bh = bt[lane], Bm = bt[64 + lane], Bl = bt[128 + lane];
  const u32x4 Ah = *reinterpret_cast<const u32x4*>(arow + 16 * kb + 4 * q);
  if constexpr (NP == 3) {
    const u32x4 Am = *reinterpret_cast<const u32x4*>(arow + plane_dwords + 16 * kb + 4 * q);
    const u32x4 Al = *reinterpret_cast<const u32x4*>(arow + 2 * plane_dwords + 16 * kb + 4 * q);
    acc = mfma_bf(Al, Bh, acc);
    acc = mfma_bf(Ah, Bl, acc);
    acc = mfma_bf(Am, Bm, acc);
    acc = mfma_bf(Am, Bh, acc);
    acc = mfma_bf(Ah, Bm, acc);
    acc = mfma_bf(Ah, Bh, acc);
  } else {
    acc = mfma_bf(Ah, Bl, acc);
    acc = mfma_bf(Ah, Bm, acc);
    acc = mfma_bf(Ah, Bh, acc);
  }
  return acc;
}
// the same against B pieces held in REGISTERS (bp[0 | 1 | 2] = h, m, l: a global image tile fetched ahead); np (1 | 3) is a
// launch argument: a uniform branch
__device__ __forceinline__ f32x4 bfp_mma_regs(f32x4 acc, const uint32_t* arow, int plane_dwords, int np, int kb, int q, const u32x4 (&bp)[3]) {
  const u32x4 Ah = *reinterpret_cast<const u32x4*>(arow + 16 * kb + 4 * q);
  if (np == 3) {
    const u32x4 Am = *reinterpret_cast<const u32x4*>(arow + plane_dwords + 16 * kb + 4 * q);
    const u32x4 Al = *reinterpret_cast<const u32x4*>(arow + 2 * plane_dwords + 16 * kb + 4 * q);
    acc = mfma_bf(Al, bp[0], acc);
    acc = mfma_bf(Ah, bp[2], acc);
    acc = mfma_bf(Am, bp[1], acc);
    acc = mfma_bf(Am, bp[0], acc);
    acc = mfma_bf(Ah, bp[1], acc);
    acc = mfma_bf(Ah, bp[0], acc);
  } else {
    acc = mfma_bf(Ah, bp[2], acc);
    acc = mfma_bf(Ah, bp[1], acc);
    acc = mfma_bf(Ah, bp[0], acc);
  }
  return acc;
}
#endif

}  // namespace spec
}  // namespace agcn
#endif  // AGCN_SPEC_BFPIPE_H_
