// Shape-specialised fused forward of SGC_LL (K = 2): the whole layer for one graph in one 1024-thread workgroup.
// Same math, same outputs and the same reserve layout as fused_forward_kernel (fused_kernels.hip), so either
// backward kernel can follow it.  See spec_kernels.h for why it exists.
//
//   F0  X rows -> LDS image (zero padded); CSR row pointers / entries requested; first M_L fragment requested
//   F1  Z = X . M_L                      graphconv.py:164        MFMA, B fragments streamed from L2 -> Z^T image
//   F2  S = exp(-||Z_i - Z_j||)          graphconv.py:168-180    2x2 register blocks over the upper triangle
//   F2b d_j = sum_i S_ij (row order), s = d^-1/2                 graphconv.py:195-197
//   F3  res_W, res_L out; L = lrelu(I | I - s S s) + L_orig      graphconv.py:198-216   (in place over S)
//   F4  X_1 = L X_0                      graphconv.py:231        MFMA, both operands in LDS
//   F5  Y = act(X_0 W_0 + X_1 W_1 + b)   graphconv.py:238-251    MFMA, W fragments streamed from L2
//   F6  Y, X_1 (reserve) out
//
// Where the global stores sit matters: gfx950 counts loads and stores in ONE in-order counter (vmcnt), so a load
// issued after a store cannot be waited for without also waiting for that store's acknowledgement from L2 (~1 us).
// The reserve and the n x n outputs are stored in F1 .. F3, at least a phase before the next load (the W fragments,
// requested at the start of F4) is waited for; the outputs of F4 / F5 (X_1, Y) are stored after the last W fragment
// has been requested (the Y accumulators wait in registers).  Fragments are requested one phase ahead at most: held
// longer, the register allocator moves them around and has to wait for them on the spot (measured).
#include "spec_backward_coop.h"
#include "spec_kernels.h"

#include <float.h>

namespace agcn {
namespace spec {

constexpr int MAXT = 3;   // most MFMA tiles of one phase a wave may own (accumulators are held in registers)

// The body for ONE graph (schedule position gpos), run by NT = 64 * NW threads.  DUAL: the workgroup has 1024 threads
// and runs TWO such bodies side by side, threads 0-511 and 512-1023 each on a graph of their own with their own half of
// the LDS (smem_ + dual_stride); the barriers are workgroup-wide, which is fine because both halves run the same
// sequence of them (every barrier of the body is in control flow that depends on the launch arguments only).
// RESLAP: SGC_LL_Reslap (graphconv_reslap.py:184-198): clip_by_norm on R and on the combined Laplacian (two graph-wide
// Frobenius norms), + beta L_prev, the pre-clip Laplacian kept for the backward, L_sum = the 4th output.
// RPWC > 0 (Reslap): the body for graphs whose waves own at most RPWC rows of the tile (n4 <= RPWC NW): the unrolled per-row loops over
// this thread's L_prev elements then have RPWC iterations instead of 6 | 8 (spec_backward.hip: the dead ones are issued all the same)
template <int FIN, int FOUT, int NW, bool DUAL, bool RESLAP = false, int RPWC = 0>
__device__ __forceinline__ void spec_forward_body(const SgcArgs& a, const int gpos, float* smem_) {
  constexpr int NT = 64 * NW;
  using FT = Feat<FIN>;
  constexpr int F4 = FT::F4, LDX = FT::LDX, NCB = FT::NCB, NKS = FT::NKS, CP = FT::CP;
  constexpr int NOB = FOUT / 16;
  const int tid = DUAL ? (int)(threadIdx.x & (NT - 1)) : (int)threadIdx.x;
  float* smem = smem_ + ((DUAL && (int)threadIdx.x >= NT) ? a.dual_stride : 0);
  const int lane = tid & 63, wave = uni(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  AGCN_STAMP(1, 9);   // (workgroup start; sorts first in tools/stamps.py)
  const GraphRec g = load_graph(a.graph_desc + gpos);
  const int n = g.n, row0 = g.row0;
  const uint32_t toff = g.toff;
  const FwdLds d = fwd_lds(n, F4);
  const int n4 = d.n4, ldt = d.ldt, ldz = d.ldz, ntm = n4 >> 4;
  const bool paper = a.mode == AGCN_MODE_PAPER;
  const bool relu = a.activation == AGCN_ACT_RELU;
  float* s_vec = smem + d.o_s;
  float* tile = smem + d.o_tile;   // [n4][ldt] S, then L = lrelu(R) + L_orig with the 16-padding zeroed
  float* Ximg = smem + d.o_x;      // [n4][LDX] X_0; rows >= n and columns >= FIN zero
  float* Zt = smem + d.o_z;        // [FIN][ldz] Z^T; later
  float* X1img = smem + d.o_z;     // [n4][LDX] X_1
  const float* Xg = a.X + (int64_t)row0 * a.ldx;

  AGCN_STAMP(1, 0);
  // ---- F0 ------------------------------------------------------------------------------------------------
  // first M_L fragment of this wave's F1 tiles: requested before anything else (it is consumed last)
  int t1a, t1b;
  tile_range<NW>(ntm * NCB, wave, t1a, t1b);
  float bA[NKS], bB[NKS];
  const float* mq = a.M_L + q * FIN;
  auto load_m = [&](float (&bf)[NKS], int t) {
    const int cb = t % NCB;
    const float* p = mq + min(cb * 16 + r, FIN - 1);
#pragma unroll
    for (int u = 0; u < NKS; ++u) {
      // row 4u + q of M_L; only the last k-step can run past FIN - 1 (the A image is zero there)
      if (4 * u + 3 < FIN) bf[u] = p[4 * u * FIN];
      else bf[u] = p[(min(4 * u + q, FIN - 1) - q) * FIN];
    }
  };
  if (t1a < t1b) load_m(bA, t1a);

  int csr_p0 = 0, csr_p1 = 0;
  if (a.L0_rowptr && tid < n) {
    csr_p0 = a.L0_rowptr[row0 + tid];
    csr_p1 = a.L0_rowptr[row0 + tid + 1];
  }
  const float alpha = a.alpha[0];
  // X rows: wave w takes rows w, w + 16, ..; lanes run along the features
  for (int i0 = wave; i0 < n4; i0 += 4 * NW) {
    float xr[4][CP];
    by_dtype(a.feat_bf16 != 0, [&](auto BF) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t ro = (int64_t)min(i0 + NW * u, n - 1) * a.ldx;
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
          const int c = min(lane + 64 * cp, FIN - 1);
          xr[u][cp] = 0.f;
          // (a UNIFORM bound: rows beyond the image are not requested at all -- a two-tile graph on sixteen waves has two of these
          //  four rows, and the clamped requests of the other two were half of the phase's bytes through the CU's address path)
          if (i0 + NW * u < n4) {
            if constexpr (decltype(BF)::value) xr[u][cp] = bf16_load(a.X, (int64_t)row0 * a.ldx + ro + c);
            else xr[u][cp] = Xg[ro + c];
          }
        }
      }
    });
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + NW * u;
      if (i < n4) {
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
          const int c = lane + 64 * cp;
          if (c < F4) Ximg[i * LDX + c] = (i < n && c < FIN) ? xr[u][cp] : 0.f;
        }
      }
    }
  }
  // Reslap: this thread's elements of L_prev (rows wave + NW u, columns lane + 64 cp), requested with the first round of
  // loads: in F3 they would sit behind the res_W / res_L stores (one in-order counter for loads and stores)
  constexpr int RPWF = RPWC > 0 ? RPWC : (NW == NW_SMALL ? 6 : 8);
  float lprv[RESLAP ? RPWF : 1][2];
  if constexpr (RESLAP) {
#pragma unroll
    for (int u = 0; u < RPWF; ++u)
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        lprv[u][cp] = 0.f;
        if (a.L_prev != nullptr && wave + NW * u < n4 && 64 * cp < n)
          lprv[u][cp] = a.L_prev[toff + (uint32_t)(min(wave + NW * u, n - 1) * n + min(lane + 64 * cp, n - 1))];
      }
  }
  // (a molecule row has at most 5 entries: the diagonal + 4 bonds; longer rows take the dependent tail loop)
  int csr_c[6] = {0, 0, 0, 0, 0, 0};
  float csr_v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (tid < n && csr_p1 > csr_p0) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int pp = min(csr_p0 + u, csr_p1 - 1);
      csr_c[u] = a.L0_col[pp];
      csr_v[u] = a.L0_val[pp];
    }
  }
  lds_barrier();

  AGCN_STAMP(1, 1);
  // ---- F1: Z = X . M_L -> Z^T image + reserve ----------------------------------------------------------------
  {
    float* Zg = a.sv_Z + (int64_t)row0 * FIN;
    auto tile_f1 = [&](const float (&bf)[NKS], int t) {
      const int rt = t / NCB, cb = t - rt * NCB;
      const float* Ap = Ximg + (rt * 16 + r) * LDX + q;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = mfma_chain<NKS>(Ap, bf, acc);
      const int c = cb * 16 + r, i0 = rt * 16 + 4 * q;
      if (c < FIN) {
        *reinterpret_cast<f32x4*>(Zt + c * ldz + i0) = acc;   // rows >= n come out 0 (zero rows of the X image)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4)
          if (i0 + i4 < n) AGCN_ST(2, &Zg[(i0 + i4) * FIN + c], acc[i4]);
      }
    };
    AGCN_STAMP_T(1, 13, 64 * 9);      // (wave 9: the last wave with an F1 tile of a two-tile graph)
    for (int t = t1a; t < t1b; t += 2) {
      if (t + 1 < t1b) load_m(bB, t + 1);
      tile_f1(bA, t);
      if (t + 1 < t1b) {
        if (t + 2 < t1b) load_m(bA, t + 2);
        tile_f1(bB, t + 1);
      }
    }
  }
  // W fragments of this wave's F5 tiles.  (ROOMY: 16-wave launches) the tiles go to the LAST waves -- away from wave 0, which sums the
  // degree's columns alone -- and their first term T_0 = X_0 W_0, which depends on nothing but the X image, runs beside that DEGREE
  // phase (fifteen waves wait there); F5 then only adds X_1 W_1 onto the same accumulators (the same instruction sequence per
  // accumulator, split in time: the same bits).  (First placed beside the pair loop F2: the one wave with both a share of the pairs
  // and a tile was then the last one at F2's barrier.)  F5 is bound by the fp32 matrix pipe (two-tile graph: 8 tiles x 38 chained MFMAs = 1.0 us of
  // pipe time): half of it moves into a phase that does not use the pipe.  (The standalone 8-wave kernels live within 80 registers:
  // there both terms stay in F5.)
  constexpr bool ROOMY = NW == NW_LARGE || DUAL;
  constexpr bool EARLYF = ROOMY && FIN <= 80 && !RESLAP;     // (Reslap holds 16 registers of L_prev across F0 .. F3: it would spill)
  int t5a, t5b;
  tile_range<NW>(ntm * NOB, EARLYF ? NW - 1 - wave : wave, t5a, t5b);
  const float* wq = a.weight + q * 2 * FOUT + r;
  auto load_w = [&](float (&bf)[NKS], int t, int h) {
    const int ob = t % NOB;
    const float* p = wq + h * FOUT + ob * 16;   // W row f*K + h (graphconv.py:243-244), f = 4u + q
#pragma unroll
    for (int u = 0; u < NKS; ++u) {
      if (4 * u + 3 < FIN) bf[u] = p[4 * u * 2 * FOUT];
      else bf[u] = p[(min(4 * u + q, FIN - 1) - q) * 2 * FOUT];
    }
  };
  f32x4 yacc[MAXT];
  float ybias[MAXT];
  float bW1[ROOMY ? NKS : 1];
  AGCN_STAMP_T(1, 14, 64 * 9);
  if constexpr (EARLYF) {
    if (t5a < t5b) load_w(bA, t5a, 0);      // (its round trip overlaps the barrier and the pair loop)
  }
  AGCN_STAMP_T(1, 15, 64 * 9);
  lds_barrier();

  AGCN_STAMP(1, 2);
  // ---- F2: similarity (2x2 blocks, upper triangle, feature loop split over adjacent lanes) -----------------
  {
    const int nb2 = (n + 1) >> 1;
    const TriFold19 tri(nb2);
    const int cnt = tri.cnt;
    int fs_log = 0;
    while (fs_log < 4 && (cnt << (fs_log + 1)) <= NT) ++fs_log;
    const int fs = 1 << fs_log;
    const int part = tid & (fs - 1);
    const int step = fs >= 4 ? 4 : fs;        // results of a block handled per lane group pass
    // (a wave whose threads all lie beyond the loop's cnt * fs work items skips it -- a uniform branch: walking through its index
    //  arithmetic and reductions at the lowest issue priority took the idle waves 1.3 us, and they carry the T_0 work below)
    if (64 * wave < cnt * fs)
    for (int tb = 0; tb < cnt * fs; tb += NT) {
      const int t = (tb + tid) >> fs_log;
      int bi = 0, bj = 0;
      const bool live = t < cnt && tri.get(t, bi, bj);
      const int i0 = 2 * bi, j0 = 2 * bj;
      const float* zi = Zt + i0;
      const float* zj = Zt + j0;
      // packed fp32: a v_pk_add_f32 / v_pk_fma_f32 takes one issue slot like its scalar form and does two pairs
      f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
      if (live) {
        // five features' operands requested before the first is used (a run-time stride keeps the compiler from hoisting the reads
        // over the fma chain by itself: it waited out one LDS round trip per feature), one running pointer per operand
        constexpr int DEPTH = (NW == NW_LARGE || DUAL) ? 5 : 3;     // (the standalone 8-wave kernels live within 80 registers)
        const int fstep = fs * ldz;
        const float* pi = zi + part * ldz;
        const float* pj = zj + part * ldz;
        int f = part;
        for (; f + (DEPTH - 1) * fs < FIN; f += DEPTH * fs) {
          f32x2 u[DEPTH], v[DEPTH];
#pragma unroll
          for (int k = 0; k < DEPTH; ++k) {
            u[k] = *reinterpret_cast<const f32x2*>(pi + k * fstep);
            v[k] = *reinterpret_cast<const f32x2*>(pj + k * fstep);
          }
          pi += DEPTH * fstep; pj += DEPTH * fstep;
#pragma unroll
          for (int k = 0; k < DEPTH; ++k) {
            const f32x2 d0 = (f32x2){u[k].x, u[k].x} - v[k], d1 = (f32x2){u[k].y, u[k].y} - v[k];
            acc0 = __builtin_elementwise_fma(d0, d0, acc0);
            acc1 = __builtin_elementwise_fma(d1, d1, acc1);
          }
        }
        {   // the last (up to DEPTH - 1) features of this lane: requested together as well (addresses clamped, sums under the test)
          f32x2 u[DEPTH - 1], v[DEPTH - 1];
#pragma unroll
          for (int k = 0; k < DEPTH - 1; ++k) {
            const bool on = f + k * fs < FIN;
            u[k] = *reinterpret_cast<const f32x2*>(on ? pi + k * fstep : zi);
            v[k] = *reinterpret_cast<const f32x2*>(on ? pj + k * fstep : zi);
          }
#pragma unroll
          for (int k = 0; k < DEPTH - 1; ++k) {
            if (f + k * fs < FIN) {
              const f32x2 d0 = (f32x2){u[k].x, u[k].x} - v[k], d1 = (f32x2){u[k].y, u[k].y} - v[k];
              acc0 = __builtin_elementwise_fma(d0, d0, acc0);
              acc1 = __builtin_elementwise_fma(d1, d1, acc1);
            }
          }
        }
      }
      AGCN_STAMP(1, 10);
      float a00 = acc0.x, a01 = acc0.y, a10 = acc1.x, a11 = acc1.y;
      // sum over the fs lanes of a block (fs = 1 .. 16, a power of two): lane ^ 1, lane ^ 2 as DPP quad permutations, then the
      // mirrored half row and the mirrored row -- after the first two steps every lane of a quad holds the same bits, so the
      // mirrored partner stands for lane ^ 4 / lane ^ 8: the same sums, bit for bit, as the __shfl_xor butterfly this replaces
      // (a run-time xor distance made each of its steps an LDS permute: 0.46 us for eight of them)
      if (fs >= 2) { a00 += dpp_mov<0xB1>(a00); a01 += dpp_mov<0xB1>(a01); a10 += dpp_mov<0xB1>(a10); a11 += dpp_mov<0xB1>(a11); }
      if (fs >= 4) { a00 += dpp_mov<0x4E>(a00); a01 += dpp_mov<0x4E>(a01); a10 += dpp_mov<0x4E>(a10); a11 += dpp_mov<0x4E>(a11); }
      if (fs >= 8) { a00 += dpp_mov<0x141>(a00); a01 += dpp_mov<0x141>(a01); a10 += dpp_mov<0x141>(a10); a11 += dpp_mov<0x141>(a11); }
      if (fs >= 16) { a00 += dpp_mov<0x140>(a00); a01 += dpp_mov<0x140>(a01); a10 += dpp_mov<0x140>(a10); a11 += dpp_mov<0x140>(a11); }
      AGCN_STAMP(1, 11);
      // every lane of the group now holds the four sums; lane `part` finishes element e0 + part, so that the
      // sqrt / exp / store sequence is issued once per `step` elements instead of once per element
      for (int e0 = 0; e0 < 4; e0 += step) {
        const int e = e0 + (part & (step - 1));
        const float v2 = (e & 2) ? ((e & 1) ? a11 : a10) : ((e & 1) ? a01 : a00);
        const int i = i0 + (e >> 1), j = j0 + (e & 1);
        if (live && part < step && i < n && j < n) {
          const float rr = fast_sqrt(v2);
          const float sv = (i == j) ? 0.f : fast_exp_neg(rr);   // graphconv.py:171-177
          tile[i * ldt + j] = sv;
          tile[j * ldt + i] = sv;   // bit-symmetric by construction
          if (paper && i < j) {   // the backward reads r_ij for i < j only ...
            AGCN_ST(2, &a.sv_r[toff + (uint32_t)(i * n + j)], rr);
            // ... unless the graph's backward is spread over several workgroups, each of which reads whole rows
            if (n > 16) AGCN_ST(2, &a.sv_r[toff + (uint32_t)(j * n + i)], rr);
          }
        }
      }
    }
    AGCN_STAMP(1, 12);
  }
  lds_barrier();
  AGCN_STAMP(1, 3);
  // (beside the DEGREE phase, where one wave sums the columns and fifteen wait: beside the pair loop F2 the wave that has both a share
  //  of the pairs and a tile was the last one at F2's barrier by 1.1 us)
  if constexpr (EARLYF) {   // T_0 = X_0 W_0 of this wave's F5 tiles (two fragment buffers: the next tile's is in flight)
#pragma unroll
    for (int sl = 0; sl < MAXT; ++sl) {
      const int t = t5a + sl;
      if (t < t5b) {
        const int rt = t / NOB, ob = t - rt * NOB;
        ybias[sl] = a.bias[ob * 16 + r];
        const float* A0 = Ximg + (rt * 16 + r) * LDX + q;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if ((sl & 1) == 0) {
          if (t + 1 < t5b) load_w(bB, t + 1, 0);
          acc = mfma_chain<NKS>(A0, bA, acc);
        } else {
          if (t + 1 < t5b) load_w(bA, t + 1, 0);
          acc = mfma_chain<NKS>(A0, bB, acc);
        }
        yacc[sl] = acc;
      }
    }
    if (t5a < t5b) load_w(bW1, t5a, 1);     // first W_1 fragment: consumed in F5
  }

  // degree: column sums in row order (numpy's axis-0 order), s = d^-1/2   graphconv.py:195-197
  if (tid < n) {
    float dsum = 0.f;
    const float* col = tile + tid;
    int i = 0;
    for (; i + 8 <= n; i += 8) {      // (eight reads in flight; the adds stay in row order)
      float c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] = col[(i + k) * ldt];
#pragma unroll
      for (int k = 0; k < 8; ++k) dsum += c[k];
    }
    {
      float c[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k] = col[min(i + k, n - 1) * ldt];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i + k < n) dsum += c[k];
    }
    dsum += FLT_TRUE_MIN;
    const float sd = 1.0f / sqrtf(dsum);
    s_vec[tid] = sd;
    a.sv_s[row0 + tid] = sd;
  }
  if (tid < (int)g.npad) {   // alignment floats behind the tile
    a.res_W[toff + (uint32_t)(n * n) + tid] = 0.f;
    a.res_L[toff + (uint32_t)(n * n) + tid] = 0.f;
    if (a.L_sum) a.L_sum[toff + (uint32_t)(n * n) + tid] = 0.f;
  }
  if (!RESLAP && tid == 0) {   // clip_by_average_norm never clips for SGC_LL: scale exactly 1 (see slab_kernels.hip)
    float* sc = a.sv_scal + 4 * g.b;
    sc[0] = 1.f; sc[1] = 1.f; sc[2] = 0.f; sc[3] = 1.f;
  }
  lds_barrier();

  AGCN_STAMP(1, 4);
  // ---- F3: res_W out; R, leaky-ReLU, res_L out; tile <- lrelu(R) (+ dense L_orig), 16-padding zeroed --------
  {
    const int l0ld = a.L0_ld > 0 ? a.L0_ld : n;
    const float* L0d = !a.L0_dense ? nullptr
                                   : (a.L0_ld > 0 ? a.L0_dense + (int64_t)g.b * a.L0_ld * a.L0_ld : a.L0_dense + toff);
    float scR = 1.f, ssR = 1.f;
    if constexpr (RESLAP) {
      // clip_by_norm(R, 1) needs ||R||_F of the whole tile first (graphconv_reslap.py:185): res_W out, R -> tile
      float part = 0.f;
      for (int j = lane; j < n4; j += 64) {
        const bool jv = j < n;
        const float sj = (paper && jv) ? s_vec[j] : 0.f;
        for (int i = wave; i < n4; i += NW) {
          float R = 0.f;
          if (jv && i < n) {
            const float S = tile[i * ldt + j];
            R = (i == j) ? 1.f : 0.f;
            if (paper) R = R - (s_vec[i] * S) * sj;
            AGCN_ST(1, &a.res_W[toff + (uint32_t)(i * n + j)], S);
            part += R * R;
          }
          tile[i * ldt + j] = R;
        }
      }
      ssR = body_sum<NW>(part, smem, tid);   // (a lane re-reads only the tile elements it wrote itself)
      scR = clip_scale(ssR, 1.0f);
    }
    if constexpr (RESLAP) {
      // R2 = lrelu(cn(R)) -> res_L; L_pre = R2 + L_orig + beta L_prev (graphconv_reslap.py:188-190) -> tile, and the sum
      // of its squares on the way: the dense L_orig is added here, the CSR entries (a handful per row) by the scatter
      // below, which corrects the sum of squares by 2 v l0 + l0^2 per entry instead of costing another sweep of the tile
      const bool has_prev = a.L_prev != nullptr;
      const float beta = has_prev ? a.beta[0] : 0.f;
      float part = 0.f;
#pragma unroll
      for (int u = 0; u < RPWF; ++u)
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const int i = wave + NW * u, j = lane + 64 * cp;
          if (i < n4 && j < n4) {
            float v = 0.f;
            if (i < n && j < n) {
              const float R2 = lrelu(tile[i * ldt + j] * scR, alpha);
              AGCN_ST(1, &a.res_L[toff + (uint32_t)(i * n + j)], R2);
              v = R2;
              if (L0d) v += L0d[(int64_t)i * l0ld + j];
              if (has_prev) v = v + lprv[u][cp] * beta;
              part += v * v;
            }
            tile[i * ldt + j] = v;
          }
        }
      lds_barrier();
      if (a.L0_rowptr && tid < n) {
        auto add = [&](int c, float l0) {
          const float v = tile[tid * ldt + c];
          part += (2.f * v + l0) * l0;          // (v + l0)^2 - v^2
          tile[tid * ldt + c] = v + l0;
        };
#pragma unroll
        for (int u = 0; u < 6; ++u)
          if (csr_p0 + u < csr_p1) add(csr_c[u], csr_v[u]);
        for (int p = csr_p0 + 6; p < csr_p1; ++p) add(a.L0_col[p], a.L0_val[p]);
      }
      const float ssL = body_sum<NW>(part, smem, tid);
      const float scL = clip_scale(ssL, 1.0f);
      // second clip + leaky-ReLU (:194-198): L_pre out (reserve), La -> tile and the 4th output
#pragma unroll
      for (int u = 0; u < RPWF; ++u)
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const int i = wave + NW * u, j = lane + 64 * cp;
          if (i < n && j < n) {
            const float v = tile[i * ldt + j];
            const float La = lrelu(v * scL, alpha);
            const uint32_t o = toff + (uint32_t)(i * n + j);
            a.sv_Lpre[o] = v;
            a.L_sum[o] = La;
            tile[i * ldt + j] = La;
          }
        }
      if (tid == 0) {   // norms / scales for the backward projections
        float* sc = a.sv_scal + 4 * g.b;
        sc[0] = ssR; sc[1] = scR; sc[2] = ssL; sc[3] = scL;
      }
      lds_barrier();
    } else {
      for (int j = lane; j < n4; j += 64) {
        const bool jv = j < n;
        const float sj = (paper && jv) ? s_vec[j] : 0.f;
        for (int i = wave; i < n4; i += NW) {
          float out = 0.f;
          if (jv && i < n) {
            const float l0 = L0d ? L0d[(int64_t)i * l0ld + j] : 0.f;
            const float S = tile[i * ldt + j];
            float R = (i == j) ? 1.f : 0.f;
            if (paper) R = R - (s_vec[i] * S) * sj;
            const float R2 = lrelu(R, alpha);   // (clip_by_average_norm never clips for SGC_LL: scale exactly 1)
            const uint32_t o = toff + (uint32_t)(i * n + j);
            AGCN_ST(1, &a.res_W[o], S);
            AGCN_ST(1, &a.res_L[o], R2);
            out = R2 + l0;
          }
          tile[i * ldt + j] = out;
        }
      }
      lds_barrier();
      if (a.L0_rowptr && tid < n) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
          if (csr_p0 + u < csr_p1) lds_add(&tile[tid * ldt + csr_c[u]], csr_v[u]);
        for (int p = csr_p0 + 6; p < csr_p1; ++p) lds_add(&tile[tid * ldt + a.L0_col[p]], a.L0_val[p]);
      }
      lds_barrier();
      if (a.L_sum) {   // SGC_LL with the combined Laplacian requested
        for (int i = wave; i < n; i += NW)
          for (int j = lane; j < n; j += 64) a.L_sum[toff + (uint32_t)(i * n + j)] = tile[i * ldt + j];
      }
    }
  }

  AGCN_STAMP(1, 5);
  // ---- F4: X_1 = L X_0 -> X_1 image (over Z^T) -------------------------------------------------------------
  // first W fragments of F5 (both terms): their round trip overlaps F4.  (Before: the W_1 fragment was requested at the start of F5
  // and waited for on the spot -- one L2 round trip of the phase's 1.4 us.)
  // (the standalone 8-wave kernels live within 80 registers: there the second fragment is requested in F5 as before)
  if constexpr (!EARLYF) {
    if (t5a < t5b) {
      load_w(bA, t5a, 0);
      if constexpr (ROOMY) load_w(bW1, t5a, 1);
    }
  }
  {
    int t4a, t4b;
    tile_range<NW>(ntm * NCB, wave, t4a, t4b);
    for (int t = t4a; t < t4b; ++t) {
      const int rt = t / NCB, cb = t - rt * NCB;
      const float* Ap = tile + (rt * 16 + r) * ldt + q;
      const float* Bp = Ximg + q * LDX + cb * 16 + r;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int k0 = 0; k0 < n4; k0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[u] = Ap[k0 + 4 * u];
          bv[u] = Bp[(k0 + 4 * u) * LDX];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      }
      const int c = cb * 16 + r, i0 = rt * 16 + 4 * q;
      // every cell of the 16-padded image is written: rows >= n and columns >= FIN come out as 0
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) X1img[(i0 + i4) * LDX + c] = acc[i4];
    }
  }
  lds_barrier();

  AGCN_STAMP(1, 6);
  // ---- F5: Y = act(X_0 W_0 + X_1 W_1 + b); accumulators held until every W fragment has been requested -----
#pragma unroll
  for (int sl = 0; sl < MAXT; ++sl) {
    const int t = t5a + sl;
    if (t < t5b) {
      const int rt = t / NOB, ob = t - rt * NOB;
      if (ROOMY && sl == 0) {
#pragma unroll
        for (int u = 0; u < NKS; ++u) bB[u] = bW1[ROOMY ? u : 0];
      } else {
        load_w(bB, t, 1);
      }
      const float* A1 = X1img + (rt * 16 + r) * LDX + q;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (EARLYF) {
        acc = yacc[sl];
      } else {
        ybias[sl] = a.bias[ob * 16 + r];
        const float* A0 = Ximg + (rt * 16 + r) * LDX + q;
        acc = mfma_chain<NKS>(A0, bA, acc);
        if (t + 1 < t5b) load_w(bA, t + 1, 0);
      }
      acc = mfma_chain<NKS>(A1, bB, acc);
      yacc[sl] = acc;
    }
  }
  AGCN_STAMP(1, 7);
  // ---- F6: Y and X_1 out -------------------------------------------------------------------------------
  {
    float* Yg = a.Y + (int64_t)row0 * a.ldy;
#pragma unroll
    for (int sl = 0; sl < MAXT; ++sl) {
      const int t = t5a + sl;
      if (t < t5b) {
        const int rt = t / NOB, ob = t - rt * NOB;
        const int c = ob * 16 + r, i0 = rt * 16 + 4 * q;
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          if (i0 + i4 < n) {
            float v = yacc[sl][i4] + ybias[sl];
            if (relu) v = fmaxf(v, 0.f);
            if (a.feat_bf16) bf16_store(a.Y, (int64_t)(row0 + i0 + i4) * a.ldy + c, v);      // (stores only: no load behind them)
            else AGCN_ST(4, &Yg[(int64_t)(i0 + i4) * a.ldy + c], v);
          }
        }
      }
    }
    // X_1 rows (reserve: the parameter gradients read them)
    float* Xo = a.sv_X + (int64_t)row0 * FIN;
    for (int i = wave; i < n; i += NW)
#pragma unroll
      for (int cp = 0; cp < CP; ++cp) {
        const int c = lane + 64 * cp;
        if (c < FIN) AGCN_ST(2, &Xo[i * FIN + c], X1img[i * LDX + c]);
      }
    if (a.feat_bf16) {     // the widened X rows for the backward and the parameter gradients
      float* X0o = a.sv_X0 + (int64_t)row0 * FIN;
      for (int i = wave; i < n; i += NW)
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
          const int c = lane + 64 * cp;
          if (c < FIN) X0o[i * FIN + c] = Ximg[i * LDX + c];
        }
    }
  }
  // The blocks of the last (smallest) graphs of the batch -- positions tp_first .. B-1, at most 32 of them -- also leave
  // transposed, k-blocked copies of the parameters in the reserve for the backward kernel (its B operands W^T and M_L^T then
  // load as 16 bytes per lane for four k-steps, 8 cache lines per load: 16 lines per four k-steps row-major, 64 from W itself).  Stores only, behind every load of the block; in
  // chunks of 512 elements dealt round-robin to the sharing blocks (whatever their thread count).  (One block doing
  // all of it was the tail of the launch at Fin = 128: 48 k strided loads, 42 instead of 25 us.)
  if (gpos >= a.tp_first) {
    const int nsh = a.B - a.tp_first, me = gpos - a.tp_first;
    if (tid < 512) {
      // k-BLOCKED: [k / 4][f][k % 4] (k = the backward's MFMA k index: an input feature of M_L resp. an output feature of W),
      // zero beyond the last k -- a lane's B values of FOUR k-steps are one 16-byte load (spec_kernels.h: kb_frag)
      for (int e = me * 512 + tid; e < F4 * FIN; e += nsh * 512) {
        const int kg = e / (4 * FIN), rem = e - kg * (4 * FIN);
        const int f = rem >> 2, k = 4 * kg + (rem & 3);
        a.sv_MT[e] = k < FIN ? a.M_L[f * FIN + k] : 0.f;
      }
      for (int e = me * 512 + tid; e < 2 * FOUT * FIN; e += nsh * 512) {
        const int kk = e / (FOUT * FIN), rem = e - kk * (FOUT * FIN);
        const int cg = rem / (4 * FIN), rem2 = rem - cg * (4 * FIN);
        const int f = rem2 >> 2, c = 4 * cg + (rem2 & 3);
        a.sv_WT[e] = a.weight[(f * 2 + kk) * FOUT + c];
      }
      // bf16 features: W^T once more as a THREE-PIECE image in MFMA operand order for the backward's filter adjoint on the bf16
      // pipe (spec_bfpipe.h): tile (kk, kb, ct), lane (r, q) holds B[k = 32 kb + 8 q + j][f = 16 ct + r] = W[f K + kk][k], j = 0 .. 7
      // -- eight consecutive floats of one row of W (two 16-byte loads); zero rows for f >= FIN
      if (a.sv_WTI != nullptr) {
        if constexpr ((FOUT & 31) == 0) {
          constexpr int KBW = FOUT / 32;
          u32x4* img = static_cast<u32x4*>(a.sv_WTI);
          for (int e = me * 512 + tid; e < 2 * KBW * NCB * 64; e += nsh * 512) {
            const int ln = e & 63, tile = e >> 6, ct = tile % NCB, kb = (tile / NCB) % KBW, kk = tile / (NCB * KBW);
            const int f = 16 * ct + (ln & 15), k0 = 32 * kb + 8 * (ln >> 4);
            const float* src = a.weight + (size_t)(min(f, FIN - 1) * 2 + kk) * FOUT + k0;
            f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
            if (f >= FIN) { lo = (f32x4){0.f, 0.f, 0.f, 0.f}; hi = lo; }
            const Pieces P = split3x8(lo, hi);
            u32x4* dst = img + (size_t)tile * 192 + ln;
            dst[0] = P.h; dst[64] = P.m; dst[128] = P.l;
          }
        }
      }
    }
  }
  AGCN_STAMP(1, 8);
}

// grid: [members of the cooperative graphs][one workgroup per graph: plain_count of them][two small graphs per workgroup]
// FULL: the launch may carry members of cooperative graphs and workgroups of two small graphs (16-wave flavour only; a
// variant of its own so that launches without them keep the plain kernel's register allocation).
// 8-wave flavour: held to AGCN_SMALL_OCC waves per SIMD where that does not spill much (feature widths up to 80).
// FULL == 2: as 1, with the WIDE members (graphs of up to COOP_MAX_N_WIDE nodes; spec_backward.hip).
template <int FIN, int FOUT, int NW, int FULL, bool RESLAP = false>
__global__ __launch_bounds__(64 * NW, NW == NW_SMALL ? (FIN > 80 ? 4 : AGCN_SMALL_OCC) : 4) void spec_forward_kernel(SgcArgs a, int first) {
  constexpr int MAXN = FULL == 2 ? COOP_MAX_N_WIDE : COOP_MAX_N;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if constexpr (RESLAP) {   // grid: [members of the cooperative graphs][one workgroup per graph: plain_count][two small graphs per workgroup]
    int blk = (int)blockIdx.x;
    if constexpr (NW == NW_LARGE) {
      if (blk < a.coop_slabs) {
        {
          const SlabRec srec = load_slab(a.slab_desc + blk);
          if (MAXN == COOP_MAX_N && srec.n <= COOP_SMALL_N) coop_forward_member<FIN, FOUT, true, MAXN == COOP_MAX_N ? COOP_SMALL_N : MAXN>(a, blk, smem, srec);     // (uniform)
          else coop_forward_member<FIN, FOUT, true, MAXN>(a, blk, smem, srec);
        }
        return;
      }
      blk -= a.coop_slabs;
      if (blk >= a.plain_count) {
        // (paired graphs have at most DUAL_MAX_N = 16 nodes: two tile rows per wave of their eight)
        spec_forward_body<FIN, FOUT, NW_SMALL, true, true, 2>(a, first + a.plain_count + 2 * (blk - a.plain_count) + (int)(threadIdx.x >> 9), smem);
        return;
      }
    }
    if constexpr (NW == NW_LARGE) {
      const int npk = load_graph(a.graph_desc + first + blk).n;      // (uniform; the body reads the record again: scalar cache)
      if (npk <= 2 * NW_LARGE) { spec_forward_body<FIN, FOUT, NW_LARGE, false, true, 2>(a, first + blk, smem); return; }
      if (npk <= 4 * NW_LARGE) { spec_forward_body<FIN, FOUT, NW_LARGE, false, true, 4>(a, first + blk, smem); return; }
    }
    spec_forward_body<FIN, FOUT, NW, false, true>(a, first + blk, smem);
  } else if constexpr (FULL) {
    // grid: [members of the cooperative graphs][one workgroup per graph: plain_count][two small graphs per workgroup]
    int blk = (int)blockIdx.x;
    if (blk < a.coop_slabs) {   // (spec_backward_coop.h)
      {
          const SlabRec srec = load_slab(a.slab_desc + blk);
          if (MAXN == COOP_MAX_N && srec.n <= COOP_SMALL_N) coop_forward_member<FIN, FOUT, false, MAXN == COOP_MAX_N ? COOP_SMALL_N : MAXN>(a, blk, smem, srec);     // (uniform)
          else coop_forward_member<FIN, FOUT, false, MAXN>(a, blk, smem, srec);
        }
      return;
    }
    blk -= a.coop_slabs;
    if (blk >= a.plain_count) {
      spec_forward_body<FIN, FOUT, NW_SMALL, true>(a, first + a.plain_count + 2 * (blk - a.plain_count) + (int)(threadIdx.x >> 9), smem);
      return;
    }
    spec_forward_body<FIN, FOUT, NW_LARGE, false>(a, first + blk, smem);
  } else {
    spec_forward_body<FIN, FOUT, NW, false>(a, first + (int)blockIdx.x, smem);
  }
}

}  // namespace spec

bool spec_supported(int32_t Fin, int32_t Fout, int32_t K, int32_t variant) {
  if (K != 2 || (variant != AGCN_SGC_LL && variant != AGCN_SGC_LL_RESLAP)) return false;
  return (Fin == 75 && Fout == 64) || (Fin == 64 && Fout == 128) || (Fin == 128 && Fout == 128) ||
         (Fin == 128 && Fout == 64) || (Fin == 64 && Fout == 64);
}

// LDS bytes of the forward kernel for a largest graph of n nodes; SIZE_MAX when a wave would own more than MAXT
// tiles of an MFMA phase (their accumulators are held in registers until the final store phase)
size_t spec_fwd_lds_bytes(int32_t n, int32_t Fin, int32_t Fout, int32_t nw) {
  const int F4 = (Fin + 15) & ~15, ntm = (n + 15) / 16;
  if (ntm * (F4 / 16) > spec::MAXT * nw || ntm * (Fout / 16) > spec::MAXT * nw) return ~(size_t)0;
  if (nw == spec::NW_SMALL && n > spec::SMALL_N) return ~(size_t)0;
  return sizeof(float) * (size_t)spec::fwd_lds(n, F4).total;
}

// LDS bytes of a member of the cooperative forward for a graph of n nodes
size_t spec_coop_fwd_lds_bytes(int32_t n, int32_t Fin) {
  return sizeof(float) * (size_t)spec::coop_fwd_lds(n, (Fin + 15) & ~15).total;
}

size_t spec_coop_fwd_lds_bytes_bfp(int32_t n, int32_t Fin, int32_t np) {
  return sizeof(float) * (size_t)spec::coop_fwd_lds(n, (Fin + 15) & ~15, np).total;
}

template <int FIN, int FOUT, int NW>
static int launch_spec_forward_t(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, hipStream_t s) {
  if (a.variant == AGCN_SGC_LL_RESLAP) {
    if ((a.dual_pairs > 0 || a.coop_slabs > 0) && NW != spec::NW_LARGE) return fail(AGCN_EINVAL, "spec_forward: members / paired graphs need the 16-wave launch");
    SgcArgs aa = a;
    aa.plain_count = count;   // one-graph workgroups; behind them a.dual_pairs workgroups of two small graphs each
    aa.lds_bytes = (int32_t)lds_bytes;
    auto kern = &spec::spec_forward_kernel<FIN, FOUT, NW, 0, true>;
    if constexpr (NW == spec::NW_LARGE) {
      if (a.coop_wide && a.coop_slabs > 0) kern = &spec::spec_forward_kernel<FIN, FOUT, NW, 2, true>;
    }
    if (lds_bytes > 64 * 1024) {
      hipError_t e = static_cast<hipError_t>(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes));
      if (e != hipSuccess) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(spec_fwd): %s", hipGetErrorString(e));
    }
    prof_begin(PK_SPEC_FWD, s);
    hipLaunchKernelGGL(kern, dim3((unsigned)(count + (NW == spec::NW_LARGE ? a.coop_slabs + a.dual_pairs : 0))), dim3(64 * NW), lds_bytes, s, aa, first);
    prof_end(s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "spec_forward launch: %s", hipGetErrorString(e));
    return AGCN_OK;
  }
  const bool full = NW == spec::NW_LARGE && (a.coop_slabs > 0 || a.dual_pairs > 0);
  auto kern = &spec::spec_forward_kernel<FIN, FOUT, NW, 0>;
  if constexpr (NW == spec::NW_LARGE) {
    if (full) kern = (a.coop_wide && a.coop_slabs > 0) ? &spec::spec_forward_kernel<FIN, FOUT, NW, 2> : &spec::spec_forward_kernel<FIN, FOUT, NW, 1>;
  }
  if (lds_bytes > 64 * 1024) {
    hipError_t e = static_cast<hipError_t>(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes));
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(spec_fwd): %s", hipGetErrorString(e));
  }
  prof_begin(PK_SPEC_FWD, s);
  const unsigned members = NW == spec::NW_LARGE ? (unsigned)(a.coop_slabs + a.dual_pairs) : 0u;
  SgcArgs aa = a;
  aa.plain_count = count;   // one-graph workgroups; behind them a.dual_pairs workgroups of two small graphs each
  aa.lds_bytes = (int32_t)lds_bytes;
  // count = plain workgroups + workgroups of two small graphs each (a.plain_count is set by the caller for the 16-wave
  // launch; the 8-wave launch has one graph per workgroup throughout)
  hipLaunchKernelGGL(kern, dim3(members + count), dim3(64 * NW), lds_bytes, s, aa, first);
  prof_end(s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "spec_forward launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

template <int FIN, int FOUT>
static int launch_spec_forward_w(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, int32_t nw, hipStream_t s) {
  if (nw == spec::NW_SMALL) return launch_spec_forward_t<FIN, FOUT, spec::NW_SMALL>(a, first, count, lds_bytes, s);
  return launch_spec_forward_t<FIN, FOUT, spec::NW_LARGE>(a, first, count, lds_bytes, s);
}

int launch_spec_forward(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, int32_t nw, void* stream) {
  if (count <= 0 && a.coop_slabs <= 0 && a.dual_pairs <= 0) return AGCN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.Fin == 75 && a.Fout == 64) return launch_spec_forward_w<75, 64>(a, first, count, lds_bytes, nw, s);
  if (a.Fin == 64 && a.Fout == 128) return launch_spec_forward_w<64, 128>(a, first, count, lds_bytes, nw, s);
  if (a.Fin == 128 && a.Fout == 128) return launch_spec_forward_w<128, 128>(a, first, count, lds_bytes, nw, s);
  if (a.Fin == 128 && a.Fout == 64) return launch_spec_forward_w<128, 64>(a, first, count, lds_bytes, nw, s);
  if (a.Fin == 64 && a.Fout == 64) return launch_spec_forward_w<64, 64>(a, first, count, lds_bytes, nw, s);
  return fail(AGCN_EINVAL, "spec_forward: unsupported layer shape %d -> %d", a.Fin, a.Fout);
}

}  // namespace agcn
