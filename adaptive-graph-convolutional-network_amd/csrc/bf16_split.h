// fp32 products on the BF16 matrix pipe at fp32 accuracy ("bf16 x 6"), shared by cloud_kernels.hip and rows_split.hip.
// Every fp32 operand is split EXACTLY into three bf16 pieces x = h + m + l (8 + 8 + 8 significant bits: h = bf16(x),
// m = bf16(x - h), l = bf16(x - h - m); the differences are exact in fp32), and x y is taken as
// l h' + h l' + m m' + m h' + h m' + h h' (small terms first; the dropped m l', l m', l l' are below 2^-24 |x y|): every
// bf16 x bf16 product is exact in the fp32 accumulator, so the error is that of the fp32 additions alone -- on random data 0.2 x
// the error of the fmaf chain the f32 MFMA stands for (tests/test_bf16_split.py: numpy emulation against float64).
// Six v_mfma_f32_16x16x32_bf16 (16 cycles each) cover 32 k: 96 cycles instead of the 256 of eight v_mfma_f32_16x16x4_f32.
#ifndef AGCN_BF16_SPLIT_H_
#define AGCN_BF16_SPLIT_H_

#include "agcn_device.h"

namespace agcn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// two floats -> their three bf16 pieces, packed (low half = x0)
// Non-finite and near-maximum inputs (documented in include/agcn.h, agcn_set_rows_split): h = bf16(x) rounds to infinity for |x| within
// half a bf16 ulp of the bf16 maximum, x - h is then -inf (NaN for x = inf), and the products of that row come out NaN where the
// fp32-MFMA kernels would propagate the infinity.  Guarding it (r = isfinite(h) ? x - h : 0) costs two vector instructions per pair
// on the critical path of every split; finite results are identical either way, so the kernels do not pay for it.
__device__ __forceinline__ void split3(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
  const bf16x2 hv = (bf16x2){(__bf16)x0, (__bf16)x1};
  h = __builtin_bit_cast(uint32_t, hv);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  const bf16x2 mv = (bf16x2){(__bf16)r0, (__bf16)r1};
  m = __builtin_bit_cast(uint32_t, mv);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  const bf16x2 lv = (bf16x2){(__bf16)s0, (__bf16)s1};
  l = __builtin_bit_cast(uint32_t, lv);
}
struct Pieces { u32x4 h, m, l; };                // eight consecutive k of one row / column
__device__ __forceinline__ Pieces split3x8(const f32x4& lo, const f32x4& hi) {
  uint32_t h[4], m[4], l[4];
  split3(lo.x, lo.y, h[0], m[0], l[0]);
  split3(lo.z, lo.w, h[1], m[1], l[1]);
  split3(hi.x, hi.y, h[2], m[2], l[2]);
  split3(hi.z, hi.w, h[3], m[3], l[3]);
  Pieces p;
  p.h = (u32x4){h[0], h[1], h[2], h[3]}; p.m = (u32x4){m[0], m[1], m[2], m[3]}; p.l = (u32x4){l[0], l[1], l[2], l[3]};
  return p;
}
__device__ __forceinline__ f32x4 mfma_bf(const u32x4& a, const u32x4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_x6(const Pieces& A, const Pieces& B, f32x4 acc) {
  acc = mfma_bf(A.l, B.h, acc);
  acc = mfma_bf(A.h, B.l, acc);
  acc = mfma_bf(A.m, B.m, acc);
  acc = mfma_bf(A.m, B.h, acc);
  acc = mfma_bf(A.h, B.m, acc);
  acc = mfma_bf(A.h, B.h, acc);
  return acc;
}

}  // namespace agcn
#endif  // AGCN_BF16_SPLIT_H_
