// Shared pieces of the shape-specialised fused kernels (spec_forward.hip / spec_backward.hip).
//
// Why these exist (measured, profiles/README.md r02): the per-graph kernels are bound by instruction ISSUE, not by
// memory or the matrix cores -- a SIMD retires about one wave-instruction per cycle whatever the number of resident
// waves, and with fully runtime-parameterised shapes most of those instructions are index arithmetic and the precise
// sqrtf / expf sequences.  So the layer shapes that matter (SimpleAGCN's 75->64, 64->128, 128->128, 128->64, and
// 64->64) get kernels in which the feature widths are template constants (strides and k-loops fold into immediates
// and unrolled MFMA chains), the parameter fragments of the MFMA B operand are streamed from L2 into registers one
// tile ahead, the pairwise-distance loop uses packed fp32 math (a v_pk_fma_f32 costs one issue slot like a
// v_fma_f32), and sqrt / exp use the hardware instructions directly.
#ifndef AGCN_SPEC_KERNELS_H_
#define AGCN_SPEC_KERNELS_H_

#include <type_traits>
#include "agcn_device.h"
#include "sgc_kernels.h"

// Store policy of the bulk outputs (diagnostic A/B switch, -DAGCN_NT=<bits>): bit 0 res_W / res_L, bit 1 the reserve
// (r, Z, X_1), bit 2 Y, bit 3 dX -- nontemporal (streaming) stores leave no dirty lines for the end-of-kernel write-back
#ifndef AGCN_NT
#define AGCN_NT 0
#endif
// -DAGCN_WT=<bits>: the same classes as agent-scope (write-through) stores -- they leave the XCD's L2 as they are issued instead of
// at the end-of-kernel write-back, and the next kernel finds them in memory.  Default 8: the backward's dX and its hand-off to
// the parameter-gradient kernel (Gp, dZ) -- measured on the headline, A/B/A/B on one box: backward 18.9 -> 18.3 us, the
// parameter-gradient kernel behind it 9.55 -> 9.2 us, step 43.4-43.6 -> 42.4-42.6 us.  The other classes do not pay: Y (4) costs
// the forward 0.35 us and gains the backward nothing (12: 43.1 us), res_W / res_L (1) and the reserve (2) are neutral to slightly
// negative (3: 43.6, 9: 42.7, 10: 42.7-42.8, 7: 45.1, 15: 43.6-43.8 us).
#ifndef AGCN_WT
#define AGCN_WT 8
#endif
#define AGCN_ST(bit, ptr, val)                                          \
  do {                                                                  \
    if ((AGCN_WT) & (bit)) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   \
    else if ((AGCN_NT) & (bit)) __builtin_nontemporal_store((val), (ptr));   \
    else *(ptr) = (val);                                                \
  } while (0)

namespace agcn {
namespace spec {

// One wave issues at most one vector instruction every ~5.5 cycles on gfx950 (tools/ubench/valu_rate.hip: v_fma_f32,
// v_pk_fma_f32, v_sub_f32 and v_add_u32 alike, independent operands), and a SIMD only approaches its issue rate
// (~1.4 cycles per instruction) with four resident waves.  With one graph per CU (256-graph batches) latency is
// (instructions of the busiest wave) x 5.5 cycles, so a graph is spread over the largest workgroup there is: 16
// waves.  With many graphs per CU (batches of 1536 graphs and up) the small graphs (n <= 48: all but one or two in a
// hundred) take 8-wave workgroups instead, two or three of which fit a CU at once, so that one graph's waits
// (barriers, LDS and memory round trips: ~60 % of a wave's life) are filled by another graph's instructions; the
// few large graphs run in a launch of their own on a side stream, beside them (agcn_api.hip).
constexpr int NW_LARGE = 16, NW_SMALL = 8;
// Waves per SIMD the 8-wave flavour's register allocation is held to.  Two of its waves sit on each SIMD per workgroup,
// so 6 (<= 80 VGPRs; the backward spills 16-48 bytes per lane) lets THREE workgroups share a CU instead of two.
// Measured, 1536 / 2048 / 4096 graphs: 4 (93 VGPRs, no spill): 8.8 / 10.0 / 11.9 M graphs/s; 6: 9.2 / 10.6 / 13.0 M;
// 7: 8.9 / 10.3 / 12.6 M; 8 (64 VGPRs, 90-1050 bytes spilled): 8.9 / 10.2 / 12.4 M.
#ifndef AGCN_SMALL_OCC
#define AGCN_SMALL_OCC 6
#endif
#ifndef AGCN_SMALL_N
#define AGCN_SMALL_N 48
#endif
constexpr int SMALL_N = AGCN_SMALL_N;   // largest graph the 8-wave flavour takes (<= 48: six rows per wave)

template <int FIN>
struct Feat {
  static constexpr int F4 = (FIN + 15) & ~15;   // 16-padded feature count (MFMA k / n extent)
  static constexpr int LDX = F4 + 2;            // row stride of the node-feature images: A-fragment reads conflict-free
  static constexpr int NCB = F4 / 16;           // 16-column blocks over the features
  static constexpr int NKS = (FIN + 3) / 4;     // MFMA k-steps over the features
  static constexpr int CP = (F4 + 63) / 64;     // passes of 64 lanes over a feature row
};

// LDS plan of the forward kernel for a graph of n nodes (floats; every offset a multiple of 4).
struct FwdLds {
  int n4, ldt, ldz;
  int o_s, o_tile, o_x, o_z, total;
};
AGCN_HD inline FwdLds fwd_lds(int n, int F4) {
  FwdLds d;
  d.n4 = (n + 15) & ~15;
  d.ldt = d.n4 + 2;    // n x n tile read as an MFMA A operand: ld mod 16 == 2
  d.ldz = d.n4 + 4;    // Z^T rows: 16-byte aligned for the b128 accumulator stores
  const int ldx = F4 + 2;
  d.o_s = 32;
  d.o_tile = d.o_s + d.n4;
  d.o_x = d.o_tile + d.n4 * d.ldt;
  d.o_z = d.o_x + d.n4 * ldx;
  const int zt = F4 * d.ldz, x1 = d.n4 * ldx;
  d.total = d.o_z + (zt > x1 ? zt : x1);
  return d;
}

// LDS plan of the backward kernel.
struct BwdLds {
  int n4, ldt;
  int o_s, o_dd, o_rs, o_colp, o_tile, o_rt, o_g1, o_u, total;
};
// gnp = 3: the Gp image as THREE bf16 piece planes of [n4][bfp_stride(Fout)] dwords (spec_bfpipe.h: fp32 features, phase G on the
// bf16 pipe); 0 | 1: the fp32 image, or one plane in its place (smaller)
AGCN_HD inline BwdLds bwd_lds(int n, int F4, int Fout, int NW, int gnp = 0) {
  BwdLds d;
  d.n4 = (n + 15) & ~15;
  d.ldt = d.n4 + 2;
  const int ldx = F4 + 2, ldg = Fout + 2;
  d.o_s = 32;
  d.o_dd = d.o_s + d.n4;
  d.o_rs = d.o_dd + d.n4;
  d.o_colp = d.o_rs + d.n4;                 // [NW][n4] per-wave column partials
  d.o_tile = d.o_colp + NW * d.n4;          // L -> dL -> c
  d.o_rt = d.o_tile + d.n4 * d.ldt;         // r_ij
  d.o_g1 = d.o_rt + d.n4 * d.ldt;           // g_1 image, later the Z image
  d.o_u = d.o_g1 + d.n4 * ldx;              // Gp image, later the X_0 image, later the dZ image
  const int gp = gnp == 3 ? 3 * d.n4 * (((Fout / 2 - 8 + 15) / 16) * 16 + 8) : d.n4 * ldg, x = d.n4 * ldx;
  d.total = d.o_u + (gp > x ? gp : x);
  return d;
}

#if defined(__HIPCC__)
// exp(-r) and sqrt on the hardware transcendental instructions (1 ulp each; exp(-r) = exp2(-r log2 e) carries a
// relative error of about r * 6e-8 from the rounded product -- far inside the 1e-5 bar).  The
// precise library sequences are ~45 instructions per pair, which is what the similarity phase was made of.
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_exp_neg(float r) { return __builtin_amdgcn_exp2f(r * -1.44269504088896341f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// bf16 feature storage (AgcnLayerDesc.feat_dtype = AGCN_DTYPE_BF16: X, Y, dY, dX are bf16 matrices with the same row strides in
// ELEMENTS): widened exactly on load, rounded to nearest even on store (NaN stays NaN); everything in between is fp32
__device__ __forceinline__ float bf16_load(const float* base, int64_t idx) {
  return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[idx] << 16);
}
// The load and the widening SEPARATELY: a load loop that widens on the spot (`x << 16` right behind each 2-byte load) makes the
// compiler wait for every load before it issues the next one wherever the loads sit in blocks of their own (the wave-uniform
// `u < rpw` branches): a dependent memory round trip per row -- the bf16 forward / backward were 2.3 us slower than the fp32 ones
// for that alone.  bf16_load_raw leaves the 16 bits in the low half (no arithmetic), bf16_widen runs behind the whole loop.
// (Backward kernels only: the forward's X loads sit in ONE block and are hoisted as they are -- split the same way its bf16
// flavour got 3.4 us SLOWER.)
__device__ __forceinline__ float bf16_load_raw(const float* base, int64_t idx) {
  return __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(base)[idx]);
}
__device__ __forceinline__ float bf16_widen(float raw) { return __uint_as_float(__float_as_uint(raw) << 16); }
__device__ __forceinline__ void bf16_store(float* base, int64_t idx, float v) {
  uint32_t u = __float_as_uint(v);
  u = ((u & 0x7fffffffu) > 0x7f800000u) ? (u | 0x00400000u) : u + 0x7fffu + ((u >> 16) & 1u);
  reinterpret_cast<uint16_t*>(base)[idx] = (uint16_t)(u >> 16);
}
// the two flavours of a load / store loop: `body(std::true_type / false_type)`, chosen by ONE wave-uniform branch around the
// whole loop (a dtype test per load would put every load into a branch region of its own, and a load in one is waited for
// before the next is issued)
template <class F>
__device__ __forceinline__ void by_dtype(bool bf, F body) {
  if (bf) body(std::true_type{}); else body(std::false_type{});
}

// LDS-only workgroup barrier: global stores issued before it need not have completed.
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }

// wave-uniform value -> SGPR
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

struct GraphRec { int b, n, row0; uint32_t toff, npad; };
// one 32-byte record per schedule position (plan.cpp); tile offsets fit 32 bits on this path (checked by the launcher)
// The record comes through the SCALAR cache (one s_load_dwordx8 at a wave-uniform address; the plan is read-only while a step
// runs).  Every other load of the workgroup depends on it: as a vector load (what the compiler emits for the plain C++ form: it
// cannot prove the record is not stored to) it was a whole vector-memory round trip in front of the first round of loads -- the
// first phase of the one-workgroup bodies took 3.2 us against 1.5 us in the members, which read their tables with scalar loads.
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ GraphRec load_graph(const AgcnGraphDesc* gd) {
  // (the address is wave-uniform but not always provably so -- the paired workgroups pick their graph by threadIdx.x >> 9 --:
  //  made scalar explicitly; a vector register here would assemble and run several times slower)
  const uint64_t ga = reinterpret_cast<uint64_t>(gd);
  const uint64_t gs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ga >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ga);
  i32x8 rec;
  asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rec) : "s"(gs) : "memory");
  GraphRec g;
  g.b = rec[0];
  g.n = rec[1];
  g.row0 = rec[2];
  g.toff = (uint32_t)rec[4];
  g.npad = (uint32_t)rec[6] - g.toff - (uint32_t)(g.n * g.n);
  return g;
}

// Enumerates the 2x2 blocks (bi <= bj) of the upper triangle of an nb2 x nb2 block grid without idle iterations
// (rows r and nb2-1-r folded into one strip of nb2+1 entries, as TriFold in agcn_device.h), with the strip division
// done by a 19-bit magic number from the hardware reciprocal: exact for t < 2^12 and strip widths < 2^7
// (n <= 128 nodes), and no 64-bit division in the kernel prologue.
struct TriFold19 {
  int nb2, W, cnt;
  uint32_t m;
  __device__ explicit TriFold19(int nb2_) : nb2(nb2_), W(nb2_ + 1), cnt(((nb2_ + 1) >> 1) * (nb2_ + 1)) {
    m = (uint32_t)(524288.0f * __builtin_amdgcn_rcpf((float)W)) + 1u;
  }
  __device__ __forceinline__ bool get(int t, int& bi, int& bj) const {
    const int rr = (int)(((uint32_t)t * m) >> 19), c = t - rr * W;
    if (c < nb2 - rr) { bi = rr; bj = rr + c; return true; }
    const int r2 = nb2 - 1 - rr;
    if (r2 == rr) return false;   // middle row of an odd grid is covered by the first branch
    bi = r2; bj = r2 + (c - (nb2 - rr));
    return true;
  }
};

// deterministic sum over the NW waves of one body (block_sum of agcn_device.h for a body that may be half a workgroup);
// the member workgroups' counterpart: slab record `sl` of the plan (AgcnSlabDesc, one s_load_dwordx8; before: slab_graph / slab_row, then
// n / row_off / tile_off of that graph -- two dependent round trips, 0.7 us between a member's first instruction and its first load)
struct SlabRec { int b, i0, n, row0; uint32_t toff, npad; };
__device__ __forceinline__ SlabRec load_slab(const AgcnSlabDesc* sd) {
  const uint64_t ga = reinterpret_cast<uint64_t>(sd);
  const uint64_t gs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ga >> 32)) << 32) |
                      (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ga);
  i32x8 rec;
  asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rec) : "s"(gs) : "memory");
  SlabRec g;
  g.b = rec[0];
  g.i0 = rec[1];
  g.n = rec[2];
  g.row0 = rec[3];
  g.toff = (uint32_t)rec[4];
  g.npad = (uint32_t)rec[6] - g.toff - (uint32_t)(g.n * g.n);
  return g;
}

// acc += sum_u A[4 u] b[u] over NKS k-steps of v_mfma_f32_16x16x4_f32, A an LDS image row (this lane's k column: stride 4 floats): ALL the
// A operands are read before the first matrix instruction.  Left to itself the compiler reads a pair into ONE register pair, waits for
// it, issues two instructions and reads the next pair -- an LDS round trip per two links: ~1750 cycles for a 19-link chain that takes
// 19 x 40 when its operands are there (tools/stamps.py: 1.36 us for the one F1 tile of a wave).  The scheduling barrier keeps the reads in
// front of the instructions (an empty asm with a memory clobber did not: the reads were sunk back to their uses).  Same instructions in the same order per accumulator: same bits.
// (Chains of more than 20 links -- 128 input features -- keep the plain loop: 32 more live registers would spill.)
template <int NKS>
__device__ __forceinline__ f32x4 mfma_chain(const float* Ap, const float (&b)[NKS], f32x4 acc) {
  if constexpr (NKS <= 20) {
    float av[NKS];
#pragma unroll
    for (int u = 0; u < NKS; ++u) av[u] = Ap[4 * u];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NKS; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], b[u], acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int u = 0; u < NKS; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ap[4 * u], b[u], acc, 0, 0, 0);
  }
  return acc;
}
// ... both operands LDS image rows (B: this lane's column, `bs` floats between k rows 4 u and 4 u + 4)
template <int NKS>
__device__ __forceinline__ f32x4 mfma_chain2(const float* Ap, const float* Bp, int bs, f32x4 acc) {
  if constexpr (NKS <= 20) {
    float av[NKS], bv[NKS];
#pragma unroll
    for (int u = 0; u < NKS; ++u) { av[u] = Ap[4 * u]; bv[u] = Bp[u * bs]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NKS; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int u = 0; u < NKS; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ap[4 * u], Bp[u * bs], acc, 0, 0, 0);
  }
  return acc;
}

// red: 32 floats of the body's LDS; contains barriers: uniform control flow only
template <int NW>
__device__ __forceinline__ float body_sum(float v, float* red, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    red[31] = t;
  }
  __syncthreads();
  return red[31];
}

// the granules of the parameter-gradient kernel that follows this launch (param_grad_tagged_kernel): every workgroup clears a
// slice, behind everything else it does.  AGENT-SCOPE stores like every other access to these words: a plain store leaves the
// (cleared) line in this XCD's L2, where the reducer's polls of the next kernel kept finding it for ~100 us (measured: the
// step 47 -> 143 us) -- lines that only ever see agent-scope accesses are not kept anywhere.
__device__ __forceinline__ void zero_tags(const SgcArgs& a) {
  if (a.ztag_count16 > 0) {
    f32x4* p = reinterpret_cast<f32x4*>(a.ztag_ptr);
    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int step = (int)(gridDim.x * blockDim.x);
    // (16-byte agent-scope stores: two granules each; as 8-byte atomic stores the slice cost the backward kernel 0.6 us)
    for (int e = (int)(blockIdx.x * blockDim.x + threadIdx.x); e < a.ztag_count16; e += step)
      asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p + e), "v"(z) : "memory");
  }
}

// The backward's products against the transposed parameter copies (g = Gp W^T, dX += dZ M_L^T) walk k in blocks of 16 with the
// k index PERMUTED inside a block: MFMA step j of block b pairs A[row r][16 b + 4 q + j] with B[k = 16 b + 4 q + j][column] -- so
// that a lane's A values of four steps are 16 consecutive bytes of the LDS image (two 8-byte reads: the images' row strides are
// 2 mod 4 floats) and its B values ONE 16-byte load of the k-blocked copy [k / 4][f][k % 4] the forward left in the reserve.
// (Row-major copies, one dword per lane and k-step: sixteen loads of 4 x 64 bytes per lane and 16 k-steps -- the first round of
// loads of the backward kernel was 1024 cache lines of parameter fragments per workgroup.)
//   kb_frag: the B fragment of block `kblk` for output column `f` (clamped by the caller); base = copy of one Chebyshev term
__device__ __forceinline__ f32x4 kb_frag(const float* base, int kblk, int q, int f, int fin) {
  return *reinterpret_cast<const f32x4*>(base + ((int64_t)(4 * kblk + q) * fin + f) * 4);
}
//   kb_mma: acc += A[16 x 16 of block kblk] . B;  arow = image row of this lane (image + row * ld), 8-byte aligned at 16 kblk + 4 q
__device__ __forceinline__ f32x4 kb_mma(f32x4 acc, const float* arow, int kblk, int q, const f32x4& b) {
  const f32x2 a0 = *reinterpret_cast<const f32x2*>(arow + 16 * kblk + 4 * q);
  const f32x2 a1 = *reinterpret_cast<const f32x2*>(arow + 16 * kblk + 4 * q + 2);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b[3], acc, 0, 0, 0);
  return acc;
}

// contiguous, balanced share of T tiles for wave w of NW
template <int NW>
__device__ __forceinline__ void tile_range(int T, int w, int& t0, int& t1) {
  const int base = T / NW, rem = T - base * NW;
  t0 = w * base + min(w, rem);
  t1 = t0 + base + (w < rem ? 1 : 0);
}
#endif

}  // namespace spec

// host-side entry points (agcn_api.hip)
bool spec_supported(int32_t Fin, int32_t Fout, int32_t K, int32_t variant);
// nw = waves per workgroup: spec::NW_LARGE (16) or spec::NW_SMALL (8, graphs of at most spec::SMALL_N nodes)
size_t spec_fwd_lds_bytes(int32_t n, int32_t Fin, int32_t Fout, int32_t nw);
size_t spec_bwd_lds_bytes(int32_t n, int32_t Fin, int32_t Fout, int32_t nw, int32_t gnp = 0);
size_t spec_coop_lds_bytes(int32_t n, int32_t Fin, int32_t Fout, int32_t gnp = 0);
size_t spec_coop_fwd_lds_bytes(int32_t n, int32_t Fin);
// ... with Z = X M_L on the bf16 pipe (spec_bfpipe.h): np piece planes of X (1: bf16 features, 3: fp32) + the M_L image
size_t spec_coop_fwd_lds_bytes_bfp(int32_t n, int32_t Fin, int32_t np);
int launch_spec_forward(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, int32_t nw, void* stream);
int launch_spec_backward(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, int32_t nw, void* stream);

}  // namespace agcn
#endif  // AGCN_SPEC_KERNELS_H_
