// Shape-specialised fused backward of SGC_LL (K = 2): everything local to one graph in one 1024-thread workgroup.
// Same math and the same hand-off buffers (ws_Gp, ws_dZ, ws_pg -> param_grads.hip) as fused_backward_kernel
// (fused_kernels.hip); see spec_kernels.h / spec_forward.hip for the design rules.
//
//   B0  one bulk round of loads: dY, Y rows -> Gp = dY . act'(Y) image (A operand);  r tile
//       (upper triangle, mirrored into LDS);  s;  CSR rows of L_orig;  first W fragment
//   B0b L = lrelu(I | I - s S s) + L_orig rebuilt in LDS (S = exp2(-r log2 e): the forward's own instruction, so the
//       same bits as res_W -- neither res_W nor res_L is read back)
//   G   [g_0 | g_1] = Gp . W^T            adjoint of graphconv.py:247     MFMA, W^T streamed from the forward's transposed copy;
//       g_0 stays in the accumulators of the wave that will also own that dX tile, g_1 -> LDS image
//   B1  dX = g_0 + L^T g_1                adjoint of graphconv.py:231     MFMA (dX accumulators stay in registers)
//   B2  (paper) dL = g_1 X_0^T -> dalpha, dA, ds -> dd;  c_ij = (dr_ij + dr_ji) / r_ij in one pass over the pairs;
//       dZ = (diag(c 1) - c) Z;  dX += dZ M_L^T
//   B3  every global store: dX, and the hand-off to the parameter-gradient kernel (Gp, dZ, dalpha)
// No global store precedes the last global load (one in-order vmcnt for loads and stores on gfx950).
#include "spec_backward_coop.h"
#include "spec_tail.h"
#include "spec_kernels.h"

#include <float.h>

namespace agcn {
namespace spec {

constexpr int MAXTB = 3;   // most MFMA tiles of one phase a wave may own

// The body for ONE graph; DUAL as in spec_forward.hip.  RESLAP: SGC_LL_Reslap (graphconv_reslap.py:184-198 under autodiff):
// L is the saved, twice clipped L_sum; between dL = g_1 X_0^T and the pair passes sit the adjoints of the two
// clip_by_norm + leaky-ReLU stages (two graph-wide inner products), dbeta and dL_prev.
// RPWC > 0: the body for graphs whose waves own at most RPWC rows of a node-feature matrix (n4 <= RPWC NW) -- the unrolled
// per-row loops of the load / store phases then have RPWC iterations instead of 6 | 8: a two-tile graph on sixteen waves owns TWO
// rows per wave, and the dead iterations (predicated off, but issued) were most of the instructions of its first phase.
template <int FIN, int FOUT, int NW, bool DUAL, bool RESLAP, int RPWC = 0>
__device__ __forceinline__ void spec_backward_body(const SgcArgs& a, const int gpos, float* smem_, const GraphRec& g) {
  constexpr int NT = 64 * NW;
  using FT = Feat<FIN>;
  constexpr int F4 = FT::F4, LDX = FT::LDX, NCB = FT::NCB, NKS = FT::NKS, CP = FT::CP;
  constexpr int LDG = FOUT + 2, CPO = (FOUT + 63) / 64;
  constexpr int RPW = RPWC > 0 ? RPWC : (NW == NW_SMALL ? 6 : 8);   // rows of a node-feature matrix one wave may own (n4 <= 48 | 128)
  const int tid = DUAL ? (int)(threadIdx.x & (NT - 1)) : (int)threadIdx.x;
  float* smem = smem_ + ((DUAL && (int)threadIdx.x >= NT) ? a.dual_stride : 0);
  const int lane = tid & 63, wave = uni(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int n = g.n, row0 = g.row0;
  const uint32_t toff = g.toff;
  // phase G on the bf16 pipe (spec_bfpipe.h; a.g_np: 1 = bf16 features, Gp is one piece; 3 = fp32 features, three piece planes)
  // -- FOUT = 64: 48 registers of image pieces in flight; not in the 8-wave launch of its own (batches of >= 1024 graphs), which
  // is held to 80 registers
  const int gnp = (FOUT == 64 && (NW == NW_LARGE || DUAL)) ? a.g_np : 0;      // (launch argument: uniform)
  const BwdLds d = bwd_lds(n, F4, FOUT, NW, gnp);
  const int n4 = d.n4, ldt = d.ldt, ntm = n4 >> 4, rpw = n4 / NW;
  const bool paper = a.mode == AGCN_MODE_PAPER;
  const bool relu = a.activation == AGCN_ACT_RELU;
  const bool gbf = gnp != 0;
  constexpr int SGP = bfp_stride(FOUT);                 // row stride of the Gp plane, dwords
  constexpr int KBW = FOUT / 32;
  float* red = smem;
  float* s_vec = smem + d.o_s;
  float* ddv = smem + d.o_dd;
  float* rsv = smem + d.o_rs;
  float* colp = smem + d.o_colp;     // [NW][n4] per-wave column partials
  float* tile = smem + d.o_tile;     // L -> dL -> c
  float* Rt = smem + d.o_rt;         // r_ij, symmetric, diagonal 0 (paper)
  float* G1img = smem + d.o_g1;      // [n4][LDX] g_1                                 (G .. dL)
  float* Zimg = smem + d.o_g1;       // [n4][LDX] Z rows                              (dZ)
  float* GPimg = smem + d.o_u;       // [n4][LDG] Gp                                  (B0 .. G)
  float* X0img = smem + d.o_u;       // [n4][LDX] X_0                                 (B1 .. dL)
  float* DZimg = smem + d.o_u;       // [n4][LDX] dZ                                  (dZ .. end)
  const float* Xg = a.X + (int64_t)row0 * a.ldx;
  const float* Yg = a.Y + (int64_t)row0 * a.ldy;
  const float* dYg = a.dY + (int64_t)row0 * a.ldy;
  const float alpha = a.alpha[0];
  // parameter gradients as trailing workgroups of this launch (spec_tail.h): dZ and the dalpha partial leave with agent-scope
  // stores, Gp is not handed off at all (the dweight tiles rebuild it from dY and Y), a flag says when this graph is done
  const bool tail = !RESLAP && a.tail != nullptr;
  const uint32_t tgen = tail ? tail_load(a.tail + TAIL_GEN) : 0u;
  if (tail && threadIdx.x == 0) tail_started(a, tgen);

  AGCN_STAMP(4, 0);
  // ---- B0 ------------------------------------------------------------------------------------------------
  // W^T fragments of phase G from the transposed copy the forward left in the reserve: WT[kk][c][f] = W[f*K + kk][c],
  // so that B[k = c][col = f] is row-major (lane (r, q) reads row 4u + q, column cb * 16 + r: four 64-byte runs per
  // load).  Tiles (row tile, 16-column block) are dealt to the waves once, for every phase that produces a [n x Fin]
  // matrix: a wave keeps its dX tiles in registers.
  // Fragments are streamed in chunks of at most 16 / 20 k-steps through two register buffers (the next chunk is in
  // flight while the current one feeds the matrix cores), whatever the layer width.
  constexpr int NKBW = FOUT / 16;          // 16-k blocks over the output features (spec_kernels.h: kb_frag / kb_mma)
  constexpr int CHW = 4, NCHW = NKBW / CHW; // blocks per chunk (16 k-steps: four 16-byte loads per lane)
  int t1a, t1b;
  tile_range<NW>(ntm * NCB, wave, t1a, t1b);
  f32x4 wb0[CHW], wb1[CHW];
  auto load_wc = [&](f32x4 (&wf)[CHW], int t, int kk, int ch) {
    const float* base = a.sv_WT + kk * (FOUT * FIN);
    const int f = min((t % NCB) * 16 + r, FIN - 1);
#pragma unroll
    for (int u = 0; u < CHW; ++u) wf[u] = kb_frag(base, CHW * ch + u, q, f, FIN);
  };
  // One memory round trip for the whole phase: every independent load is requested before the first LDS store (loads
  // return in order, so a store that waits for its data also waits for every older load), the row pointers first --
  // the only dependent loads, the CSR entries, follow the bulk requests and ride the same round trip.
  int csr_p0 = 0, csr_p1 = 0;
  if (!RESLAP && a.L0_rowptr && tid < n) {
    csr_p0 = a.L0_rowptr[row0 + tid];
    csr_p1 = a.L0_rowptr[row0 + tid + 1];
  }
  // dY, Y rows: wave w takes rows w, w + NW, ..
  float gy[RPW][CPO], yy[RPW][CPO];
  by_dtype(a.feat_bf16 != 0, [&](auto BF) {
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int i = min(wave + NW * u, n - 1);
#pragma unroll
      for (int cp = 0; cp < CPO; ++cp) {
        const int c = min(lane + 64 * cp, FOUT - 1);
        gy[u][cp] = 0.f;
        yy[u][cp] = 0.f;
        // (uniform bound + clamped addresses, never a per-lane branch: a load under one is waited for on the spot)
        if (u < rpw) {
          if constexpr (decltype(BF)::value) {
            gy[u][cp] = bf16_load_raw(a.dY, (int64_t)(row0 + i) * a.ldy + c);
            yy[u][cp] = bf16_load_raw(a.Y, (int64_t)(row0 + i) * a.ldy + c);
          } else {
            gy[u][cp] = dYg[(int64_t)i * a.ldy + c];
            yy[u][cp] = Yg[(int64_t)i * a.ldy + c];
          }
        }
      }
    }
  });
  u32x4 bpa[KBW][3], bpb[KBW][3];          // (gbf) the three pieces of the KBW k-blocks of one (tile, Chebyshev term), two buffers
  auto load_bp = [&](u32x4 (&bp)[KBW][3], int t, int kk) {
    const u32x4* img = static_cast<const u32x4*>(a.sv_WTI);
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) {
      const u32x4* bt = img + (size_t)((kk * KBW + kb) * NCB + t % NCB) * 192 + lane;
      bp[kb][0] = bt[0]; bp[kb][1] = bt[64]; bp[kb][2] = bt[128];
    }
  };
  // r tile rows (the forward stored r_ij for i < j)
  float rr[RPW][2];
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int i = wave + NW * u;
#pragma unroll
    for (int cp = 0; cp < 2; ++cp) {
      const int j = lane + 64 * cp;
      rr[u][cp] = 0.f;
      if (paper && u < rpw && 64 * cp < n)     // (uniform bounds, clamped addresses)
        rr[u][cp] = a.sv_r[toff + (uint32_t)(min(i, n - 1) * n + min(j, n - 1))];
    }
  }
  // Reslap: the combined Laplacian is the forward's 4th output (it already holds L_orig and beta L_prev)
  float lsr[RESLAP ? RPW : 1][2];
  if constexpr (RESLAP) {
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int i = wave + NW * u;
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        const int j = lane + 64 * cp;
        lsr[u][cp] = 0.f;
        if (u < rpw && 64 * cp < n) lsr[u][cp] = a.L_sum[toff + (uint32_t)(min(i, n - 1) * n + min(j, n - 1))];
      }
    }
  }
  const float s_own = a.sv_s[row0 + min(tid, n - 1)];
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
  // first parameter fragments of phase G: requested LAST (loads return in order: in front of the r / s / CSR requests their 4-6 KB per
  // wave held up everything this phase stores to LDS; they are consumed two barriers further on).  (Requested behind B0's barrier
  // instead -- B0 is bound by the CU's vector-memory path, its last wave issues its loads 2 us into the phase -- B0 ends 0.4 us
  // earlier and B0b / G pay it back: kernel 16.95 -> 17.12 us, measured.)
  if (t1a < t1b) {
    if (gbf) load_bp(bpa, t1a, 0);
    else load_wc(wb0, t1a, 0, 0);
  }
  AGCN_STAMP_T(4, 10, 64 * NW - 64);
  if (a.feat_bf16) {     // (widened HERE, behind every load of the round: spec_kernels.h, bf16_load_raw)
#pragma unroll
    for (int u = 0; u < RPW; ++u)
#pragma unroll
      for (int cp = 0; cp < CPO; ++cp) {
        gy[u][cp] = bf16_widen(gy[u][cp]);
        yy[u][cp] = bf16_widen(yy[u][cp]);
      }
  }
  uint32_t* GPpl = reinterpret_cast<uint32_t*>(GPimg);
  const int gplane = n4 * SGP;
  // Gp = dY . act'(Y) -> image (A operand of phase G); kept in registers for the hand-off store at the end
  // bf16 features + the W^T piece image of the forward (a.sv_WTI): Gp is bf16 data, ONE piece -- phase G runs on the bf16 pipe
  // (spec_bfpipe.h) and the image is a plane of bf16 rows (in the place of the fp32 image: smaller)
  float gpv[RPW][CPO];
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int i = wave + NW * u;
#pragma unroll
    for (int cp = 0; cp < CPO; ++cp) {
      const int c = lane + 64 * cp;
      const float v = (i < n && c < FOUT && (!relu || yy[u][cp] > 0.f)) ? gy[u][cp] : 0.f;
      gpv[u][cp] = v;
      if (!gbf && i < n4 && c < FOUT) GPimg[i * LDG + c] = v;
    }
    if (gbf) {   // (uniform) two adjacent columns per lane, one 4-byte store
      float lo, hi;
      int col;
      bfp_pair_columns<(CPO > 1)>(gpv[u][0], gpv[u][CPO > 1 ? 1 : 0], lane, lo, hi, col);
      if (i < n4 && col >= 0 && col < FOUT) {
        if (gnp == 3) bfp_store_pair<3>(GPpl, gplane, i * SGP + (col >> 1), lo, hi);
        else bfp_store_pair<1>(GPpl, 0, i * SGP + (col >> 1), lo, hi);
      }
    }
  }
  AGCN_STAMP_T(4, 11, 64 * NW - 64);
  // r tile: mirrored, diagonal 0
  if (paper) {
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int i = wave + NW * u;
#pragma unroll
      for (int cp = 0; cp < 2; ++cp) {
        const int j = lane + 64 * cp;
        if (i < n && j < n) {
          if (j > i) {
            Rt[i * ldt + j] = rr[u][cp];
            Rt[j * ldt + i] = rr[u][cp];
          } else if (j == i) {
            Rt[i * ldt + i] = 0.f;
          }
        }
      }
    }
  }
  if (tid < n) s_vec[tid] = paper ? s_own : 0.f;
  AGCN_STAMP_T(4, 12, 64 * NW - 64);
  lds_barrier();

  AGCN_STAMP(4, 1);
  // ---- B0b: L = lrelu(R) (+ L_orig) rebuilt from r and s; the tile's 16-padding zeroed ---------------------
  if constexpr (RESLAP) {
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int i = wave + NW * u;
      if (i < n4) {
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
          const int j = lane + 64 * cp;
          if (j < n4) tile[i * ldt + j] = (i < n && j < n) ? lsr[u][cp] : 0.f;
        }
      }
    }
  } else {
    const int l0ld = a.L0_ld > 0 ? a.L0_ld : n;
    const float* L0d = !a.L0_dense ? nullptr
                                   : (a.L0_ld > 0 ? a.L0_dense + (int64_t)g.b * a.L0_ld * a.L0_ld : a.L0_dense + toff);
    for (int j = lane; j < n4; j += 64) {
      const bool jv = j < n;
      const float sj = (paper && jv) ? s_vec[j] : 0.f;
      for (int i = wave; i < n4; i += NW) {
        float out = 0.f;
        if (jv && i < n) {
          float R = (i == j) ? 1.f : 0.f;
          if (paper) {
            const float S = (i == j) ? 0.f : fast_exp_neg(Rt[i * ldt + j]);
            R = R - (s_vec[i] * S) * sj;
          }
          out = lrelu(R, alpha);
          if (L0d) out += L0d[(int64_t)i * l0ld + j];
        }
        tile[i * ldt + j] = out;
      }
    }
  }
  lds_barrier();
  if (!RESLAP && a.L0_rowptr && tid < n) {
#pragma unroll
    for (int u = 0; u < 6; ++u)
      if (csr_p0 + u < csr_p1) lds_add(&tile[tid * ldt + csr_c[u]], csr_v[u]);
    for (int p = csr_p0 + 6; p < csr_p1; ++p) lds_add(&tile[tid * ldt + a.L0_col[p]], a.L0_val[p]);
  }

  AGCN_STAMP(4, 2);
  // ---- G: g_0 = Gp . W_0^T -> registers, g_1 = Gp . W_1^T -> image ------------------------------------------------
  f32x4 dxacc[MAXTB];
  {
    auto mma_wc = [&](f32x4 acc, const f32x4 (&wf)[CHW], int t, int ch) {
      const float* Ap = GPimg + ((t / NCB) * 16 + r) * LDG;
#pragma unroll
      for (int u = 0; u < CHW; ++u) acc = kb_mma(acc, Ap, CHW * ch + u, q, wf[u]);
      return acc;
    };
    if (gbf) {   // (uniform) bf16 pipe: three products per 32 k against the image's pieces (16-byte loads, one (tile, term) ahead)
      auto mma_bp = [&](f32x4 acc, const u32x4 (&bp)[KBW][3], const uint32_t* arow) {
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) acc = bfp_mma_regs(acc, arow, gplane, gnp, kb, q, bp[kb]);
        return acc;
      };
#pragma unroll
      for (int sl = 0; sl < MAXTB; ++sl) {
        const int t = t1a + sl;
        if (t < t1b) {
          const int rt = t / NCB, cb = t - rt * NCB;
          const uint32_t* arow = GPpl + (rt * 16 + r) * SGP;
          // step 2 sl (term 0) reads buffer a, step 2 sl + 1 (term 1) buffer b; each requests the next step's pieces first
          load_bp(bpb, t, 1);
          const f32x4 a0 = mma_bp((f32x4){0.f, 0.f, 0.f, 0.f}, bpa, arow);
          if (t + 1 < t1b) load_bp(bpa, t + 1, 0);
          const f32x4 a1 = mma_bp((f32x4){0.f, 0.f, 0.f, 0.f}, bpb, arow);
          dxacc[sl] = a0;
          const int f = cb * 16 + r;
          float* dst = G1img + (rt * 16 + 4 * q) * LDX + f;
          const bool fv = f < FIN;
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) dst[i4 * LDX] = fv ? a1[i4] : 0.f;
        }
      }
    } else
#pragma unroll
    for (int sl = 0; sl < MAXTB; ++sl) {
      const int t = t1a + sl;
      if (t < t1b) {
        f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int st = 0; st < 2 * NCHW; ++st) {
          const int kk = st / NCHW, ch = st - kk * NCHW;
          const bool odd = ((sl * 2 * NCHW + st) & 1) != 0;      // compile-time: which buffer holds this chunk
          // request the next chunk (same tile, or the first one of the next tile) into the other buffer
          if (st + 1 < 2 * NCHW) {
            if (odd) load_wc(wb0, t, (st + 1) / NCHW, (st + 1) % NCHW);
            else load_wc(wb1, t, (st + 1) / NCHW, (st + 1) % NCHW);
          } else if (t + 1 < t1b) {
            if (odd) load_wc(wb0, t + 1, 0, 0);
            else load_wc(wb1, t + 1, 0, 0);
          }
          acc[kk] = odd ? mma_wc(acc[kk], wb1, t, ch) : mma_wc(acc[kk], wb0, t, ch);
        }
        dxacc[sl] = acc[0];
        const int rt = t / NCB, f = (t - rt * NCB) * 16 + r;
        float* dst = G1img + (rt * 16 + 4 * q) * LDX + f;
        const bool fv = f < FIN;             // image columns FIN .. F4 are zero padding (the W row was clamped)
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) dst[i4 * LDX] = fv ? acc[1][i4] : 0.f;
      }
    }
  }
  // (paper) the X_0 rows B2 needs are requested HERE, behind phase G's products (its fragment registers are free again) and in front
  // of its closing barrier: the round trip overlaps the barrier and B1's matrix work (requested at the start of B1 it outlasted
  // that work by most of a microsecond)
  float xzr[RPW][CP];
  if (paper || RESLAP) {
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const float* rp = Xg + (int64_t)min(wave + NW * u, n - 1) * a.ldx;
#pragma unroll
      for (int cp = 0; cp < CP; ++cp) {
        xzr[u][cp] = 0.f;
        if (u < rpw) xzr[u][cp] = rp[min(lane + 64 * cp, FIN - 1)];
      }
    }
  }
  lds_barrier();

  AGCN_STAMP(4, 3);
  // ---- B1: dX += L^T g_1 --------------------------------------------------------------------------------------
#pragma unroll
  for (int sl = 0; sl < MAXTB; ++sl) {
    const int t = t1a + sl;
    if (t < t1b) {
      const int rt = t / NCB, cb = t - rt * NCB;
      const float* Ap = tile + q * ldt + rt * 16 + r;          // A[i][k] = L[k][i]
      const float* Bp = G1img + q * LDX + cb * 16 + r;
      f32x4 acc = dxacc[sl];
      for (int k0 = 0; k0 < n4; k0 += 16) {
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          av[u] = Ap[(k0 + 4 * u) * ldt];
          bv[u] = Bp[(k0 + 4 * u) * LDX];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      }
      dxacc[sl] = acc;
    }
  }
  float dalpha_part = 0.f, dbeta_part = 0.f;
  f32x4 zacc[MAXTB];   // dZ tiles, for the hand-off store
  if (paper || RESLAP) {
    // X_0 rows -> image over the (dead) Gp image
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int i = wave + NW * u;
      if (i < n4) {
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
          const int c = lane + 64 * cp;
          if (c < F4) X0img[i * LDX + c] = (i < n && c < FIN) ? xzr[u][cp] : 0.f;
        }
      }
    }
    lds_barrier();   // L is dead (every wave is past B1)
    AGCN_STAMP(4, 4);
    // ---- B2: dL = g_1 X_0^T -> tile; the Z rows of the dZ product are requested meanwhile ---------------------
    {
      const float* Zg = a.sv_Z + (int64_t)row0 * FIN;
#pragma unroll
      for (int u = 0; u < RPW; ++u) {
        const int i = min(wave + NW * u, n - 1);
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
          xzr[u][cp] = 0.f;
          if (paper && u < rpw) xzr[u][cp] = Zg[i * FIN + min(lane + 64 * cp, FIN - 1)];
        }
      }
      int ta, tb;
      tile_range<NW>(ntm * ntm, wave, ta, tb);
      for (int t = ta; t < tb; ++t) {
        const int rt = t / ntm, ct = t - rt * ntm;
        const float* Ap = G1img + (rt * 16 + r) * LDX + q;
        const float* Bp = X0img + (ct * 16 + r) * LDX + q;   // B[k][j] = X_0[j][k]
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc = mfma_chain2<NKS>(Ap, Bp, 4, acc);
        const int j = ct * 16 + r, i0 = rt * 16 + 4 * q;
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          float v = acc[i4];
          if (a.dL_sum && i0 + i4 < n && j < n) v += a.dL_sum[toff + (uint32_t)((i0 + i4) * n + j)];
          tile[(i0 + i4) * ldt + j] = v;
        }
      }
    }
    for (int j = lane; j < n4; j += 64) colp[wave * n4 + j] = 0.f;
    lds_barrier();
    AGCN_STAMP(4, 5);
    if constexpr (RESLAP) {
      // adjoint of  La = lrelu(cn(Lpre)),  Lpre = R2 + L_orig + beta L_prev   (graphconv_reslap.py:188-198); the tile
      // holds dLa and becomes dLpre == dR2 (reference mode ends here), then -- paper -- dR through the first clip + leaky-ReLU
      const float* sc = a.sv_scal + 4 * g.b;
      const bool has_prev = a.L_prev != nullptr && a.dL_prev != nullptr;
      if (a.dL_prev && tid < (int)g.npad) a.dL_prev[toff + (uint32_t)(n * n) + tid] = 0.f;   // alignment floats behind the tile
      {
        const float ssL = sc[2], scL = sc[3];
        const float invL = 1.0f / sqrtf(ssL);
        const bool clipped = invL <= 1.0f;   // TF MinimumGrad: ties take the norm branch
        // this thread's elements of L_pre and L_prev (rows wave + NW u, columns lane + 64 cp): ONE round of loads, kept in
        // registers for the second pass -- which then only stores (dL_prev): a load behind a store waits for the store
        float lpre[RPW][2], lprv[RPW][2];
#pragma unroll
        for (int u = 0; u < RPW; ++u)
#pragma unroll
          for (int cp = 0; cp < 2; ++cp) {
            const uint32_t o = toff + (uint32_t)(min(wave + NW * u, n - 1) * n + min(lane + 64 * cp, n - 1));
            lpre[u][cp] = 0.f;
            lprv[u][cp] = 0.f;
            if (u < rpw && 64 * cp < n) {
              lpre[u][cp] = a.sv_Lpre[o];
              if (has_prev) lprv[u][cp] = a.L_prev[o];
            }
          }
        float dot_part = 0.f;
#pragma unroll
        for (int u = 0; u < RPW; ++u)
#pragma unroll
          for (int cp = 0; cp < 2; ++cp) {
            const int i = wave + NW * u, j = lane + 64 * cp;
            if (i < n && j < n) {
              const float Lpre = lpre[u][cp];
              const float Lc = Lpre * scL;
              const float dLa = tile[i * ldt + j];
              dalpha_part += dLa * fminf(Lc, 0.f);
              const float dLc = dLa * lrelu_grad(Lc, alpha);
              tile[i * ldt + j] = dLc;
              dot_part += dLc * Lpre;
            }
          }
        const float dot = body_sum<NW>(dot_part, red, tid);   // (each lane re-reads only its own tile elements below)
        const float beta = has_prev ? a.beta[0] : 0.f;
        const float k3 = clipped ? dot * invL * invL * invL : 0.f;
        // second pass: dLpre == dR2 (-> dbeta, dL_prev) and, in the same sweep (paper), the adjoint of
        // R2 = lrelu(cn(R)), R = I - s S s (clip_by_norm, q = 1): the tile becomes dR1, then dR
        const float scR = sc[1];
        float dotR_part = 0.f;
#pragma unroll
        for (int u = 0; u < RPW; ++u)
#pragma unroll
          for (int cp = 0; cp < 2; ++cp) {
            const int i = wave + NW * u, j = lane + 64 * cp;
            if (i < n && j < n) {
              float dv = tile[i * ldt + j];
              if (clipped) dv = dv * invL - lpre[u][cp] * k3;
              if (has_prev) {
                dbeta_part += dv * lprv[u][cp];
                a.dL_prev[toff + (uint32_t)(i * n + j)] = dv * beta;
              }
              if (paper) {
                const float Sv = (i == j) ? 0.f : fast_exp_neg(Rt[i * ldt + j]);
                const float R = ((i == j) ? 1.f : 0.f) - (s_vec[i] * Sv) * s_vec[j];
                const float R1 = R * scR;
                dalpha_part += dv * fminf(R1, 0.f);
                dv = dv * lrelu_grad(R1, alpha);          // dR1
                dotR_part += dv * R;
              }
              tile[i * ldt + j] = dv;
            }
          }
        if (paper) {
          const float ssR = sc[0];
          const float invR = 1.0f / sqrtf(ssR);
          const bool clippedR = invR <= 1.0f;
          const float dotR = body_sum<NW>(dotR_part, red, tid);
          if (clippedR) {
            const float k3R = dotR * invR * invR * invR;
#pragma unroll
            for (int u = 0; u < RPW; ++u)
#pragma unroll
              for (int cp = 0; cp < 2; ++cp) {
                const int i = wave + NW * u, j = lane + 64 * cp;
                if (i < n && j < n) {
                  const float Sv = (i == j) ? 0.f : fast_exp_neg(Rt[i * ldt + j]);
                  const float R = ((i == j) ? 1.f : 0.f) - (s_vec[i] * Sv) * s_vec[j];
                  tile[i * ldt + j] = tile[i * ldt + j] * invR - R * k3R;
                }
              }
          }
        }
      }
      lds_barrier();
    }
    if (paper) {
    // Z rows -> image over g_1 (dead: every wave is past dL); read from the dZ product on, behind later barriers
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int i = wave + NW * u;
      if (i < n4) {
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
          const int c = lane + 64 * cp;
          if (c < F4) Zimg[i * LDX + c] = (i < n && c < FIN) ? xzr[u][cp] : 0.f;
        }
      }
    }
    // pass 1: dalpha, P = dA . S, row sums  sum_j P_ij s_j  and per-wave column partials  sum_i P_ij s_i
    for (int i = wave; i < n; i += NW) {
      float rsum = 0.f;
      const float si = s_vec[i];
      for (int j = lane; j < n; j += 64) {
        const float Sv = (i == j) ? 0.f : fast_exp_neg(Rt[i * ldt + j]);
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
    lds_barrier();
    if (tid < n) {
      float cs = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) cs += colp[w * n4 + tid];
      ddv[tid] = degree_grad(s_vec[tid], rsv[tid] + cs);   // dd_k = -1/2 s_k^3 ds_k
    }
    lds_barrier();
    AGCN_STAMP(4, 6);
    // pass 2: c_ij = (dr_ij + dr_ji) / r_ij for i < j, dr_ij = -S_ij (dA_ij s_i s_j + dd_j) (0 where S or r is 0),
    // mirrored in place: each unordered pair is read and written by exactly one lane
    for (int i = wave; i < n; i += NW) {
      const float si = s_vec[i], ddi = ddv[i];
      for (int j = lane; j < n; j += 64) {
        if (j > i) {
          const float rr = Rt[i * ldt + j];
          const float Sv = fast_exp_neg(rr);
          const float sj = s_vec[j];
          const float Rij = -((si * Sv) * sj), Rji = -((sj * Sv) * si);
          const float dAij = -(RESLAP ? tile[i * ldt + j] : tile[i * ldt + j] * lrelu_grad(Rij, alpha));
          const float dAji = -(RESLAP ? tile[j * ldt + i] : tile[j * ldt + i] * lrelu_grad(Rji, alpha));
          float cij = 0.f;
          if (Sv != 0.f && rr > 0.f) {
            const float dr_ij = -Sv * (dAij * si * sj + ddv[j]);
            const float dr_ji = -Sv * (dAji * sj * si + ddi);
            cij = (dr_ij + dr_ji) * fast_rcp(rr);
          }
          tile[i * ldt + j] = cij;
          tile[j * ldt + i] = cij;
        } else if (j == i) {
          tile[i * ldt + i] = 0.f;
        }
      }
    }
    lds_barrier();
    for (int i = wave; i < n; i += NW) {
      float rsum = 0.f;
      for (int j = lane; j < n; j += 64) rsum += tile[i * ldt + j];
      rsum = wave_sum(rsum);
      if (lane == 0) rsv[i] = rsum;
    }
    lds_barrier();
    AGCN_STAMP(4, 7);
    // ---- dZ = rowsum(c) Z - c Z -> image over X_0 (A operand of the next product) ------------------------------
#pragma unroll
    for (int sl = 0; sl < MAXTB; ++sl) {
      const int t = t1a + sl;
      if (t < t1b) {
        const int rt = t / NCB, cb = t - rt * NCB;
        const float* Ap = tile + (rt * 16 + r) * ldt + q;
        const float* Bp = Zimg + q * LDX + cb * 16 + r;
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
        const int f = cb * 16 + r, i0 = rt * 16 + 4 * q;
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const int i = i0 + i4;
          const float v = (i < n) ? rsv[i] * Zimg[i * LDX + f] - acc[i4] : 0.f;
          DZimg[i * LDX + f] = v;
          zacc[sl][i4] = v;
        }
      }
    }
    }   // paper
  }
  // M_L^T fragments of the last product from the transposed copy: B[k][f] = MT[k][f] row-major
  constexpr int NKBM = F4 / 16;            // 16-k blocks over the input features (the copy is zero beyond FIN)
  constexpr int CHM = 5, NCHM = (NKBM + CHM - 1) / CHM;
  f32x4 mb0[CHM], mb1[CHM];
  auto load_mc = [&](f32x4 (&bf)[CHM], int t, int ch) {
    const int f = min((t % NCB) * 16 + r, FIN - 1);
#pragma unroll
    for (int uu = 0; uu < CHM; ++uu) {
      const int u = ch * CHM + uu;       // (compile-time after unrolling)
      if (u < NKBM) bf[uu] = kb_frag(a.sv_MT, u, q, f, FIN);
    }
  };
  if (paper && t1a < t1b) load_mc(mb0, t1a, 0);
  lds_barrier();

  AGCN_STAMP(4, 8);
  // ---- dX += dZ M_L^T, then B3: every global store -------------------------------------------------------------
  {
    float* dXg = a.dX + (int64_t)row0 * a.lddx;
    auto mma_mc = [&](f32x4 acc, const f32x4 (&bf)[CHM], int t, int ch) {
      const float* Ap = DZimg + ((t / NCB) * 16 + r) * LDX;
#pragma unroll
      for (int uu = 0; uu < CHM; ++uu) {
        const int u = ch * CHM + uu;
        if (u < NKBM) acc = kb_mma(acc, Ap, u, q, bf[uu]);
      }
      return acc;
    };
    f32x4 xacc[MAXTB];
#pragma unroll
    for (int sl = 0; sl < MAXTB; ++sl) {
      const int t = t1a + sl;
      if (t < t1b) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (paper) {
#pragma unroll
          for (int ch = 0; ch < NCHM; ++ch) {
            const bool odd = ((sl * NCHM + ch) & 1) != 0;
            if (ch + 1 < NCHM) {
              if (odd) load_mc(mb0, t, ch + 1);
              else load_mc(mb1, t, ch + 1);
            } else if (t + 1 < t1b) {
              if (odd) load_mc(mb0, t + 1, 0);
              else load_mc(mb1, t + 1, 0);
            }
            acc = odd ? mma_mc(acc, mb1, t, ch) : mma_mc(acc, mb0, t, ch);
          }
        }
        xacc[sl] = acc;
      }
    }
    // (no load is issued past this point)
    AGCN_STAMP(4, 13);
#pragma unroll
    for (int sl = 0; sl < MAXTB; ++sl) {
      const int t = t1a + sl;
      if (t < t1b) {
        const int rt = t / NCB, cb = t - rt * NCB;
        const int f = cb * 16 + r, i0 = rt * 16 + 4 * q;
        if (f < FIN) {
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) {
            const int i = i0 + i4;
            if (i < n) {
              if (a.feat_bf16) bf16_store(a.dX, (int64_t)(row0 + i) * a.lddx + f, dxacc[sl][i4] + xacc[sl][i4]);
              else AGCN_ST(8, &dXg[(int64_t)i * a.lddx + f], dxacc[sl][i4] + xacc[sl][i4]);
              if (paper) {
                if (tail) tail_store_f(a.ws_dZ + (int64_t)(row0 + i) * FIN + f, zacc[sl][i4]);
                else AGCN_ST(8, &a.ws_dZ[(int64_t)(row0 + i) * FIN + f], zacc[sl][i4]);
              }
            }
          }
        }
      }
    }
    if (!tail) {
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int i = wave + NW * u;
#pragma unroll
      for (int cp = 0; cp < CPO; ++cp) {
        const int c = lane + 64 * cp;
        if (i < n && c < FOUT) AGCN_ST(8, &a.ws_Gp[(int64_t)(row0 + i) * FOUT + c], gpv[u][cp]);
      }
    }
    }
  }
  AGCN_STAMP(4, 14);
  const float da = body_sum<NW>(dalpha_part, red, tid);
  const float db = RESLAP ? body_sum<NW>(dbeta_part, red, tid) : 0.f;
  AGCN_STAMP(4, 15);
  if (tid == 0) {
    if (tail) {
      tail_store_f(a.ws_pg + 2 * g.b + 0, da);
    } else {
      a.ws_pg[2 * g.b + 0] = da;
      a.ws_pg[2 * g.b + 1] = db;
    }
  }
  if (tail) {
    tail_publish(a, a.coop_slabs + (gpos - a.tail_first), tgen, tid == 0, [] { __syncthreads(); });
    AGCN_STAMP(4, 9);
    return;
  }
  zero_tags(a);
  if (blockIdx.x == 0) {   // arrival counters of the parameter-gradient kernel that follows this one
    for (int e = tid; e < a.zero_count; e += NT) a.zero_ptr[e] = 0;
  }
  AGCN_STAMP(4, 9);
}

// FULL: the launch may carry members of cooperative graphs and workgroups of two small graphs (16-wave flavour only; a
// variant of its own so that launches without them keep the plain kernel's register allocation).
// 8-wave flavour: held to AGCN_SMALL_OCC waves per SIMD where that does not spill much (feature widths up to 80).
// FULL == 2: as 1, with the WIDE members (graphs of up to COOP_MAX_N_WIDE nodes: 12 rows of an image per wave, three 64-column
// passes per strip row -- more registers, so a variant of its own, launched only when a batch has such a graph).
template <int FIN, int FOUT, int NW, int FULL, bool RESLAP = false>
__global__ __launch_bounds__(64 * NW, NW == NW_SMALL ? (FIN > 80 ? 4 : AGCN_SMALL_OCC) : 4) void spec_backward_kernel(SgcArgs a, int first) {
  constexpr int MAXN = FULL == 2 ? COOP_MAX_N_WIDE : COOP_MAX_N;
  // paired small graphs have at most DUAL_MAX_N nodes: n4 = 16, two rows of a node-feature matrix per wave of their eight
  static_assert(DUAL_MAX_N <= 16, "the paired bodies' row loops are sized for one 16-row tile");
  constexpr int DUAL_RPW = 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if constexpr (RESLAP) {   // grid: [members of the cooperative graphs][one workgroup per graph: plain_count][two small graphs per workgroup]
    int blk = (int)blockIdx.x;
    if constexpr (NW == NW_LARGE) {
      if (blk < a.coop_slabs) {
        {
          const SlabRec srec = load_slab(a.slab_desc + blk);
          if (MAXN == COOP_MAX_N && srec.n <= COOP_SMALL_N) coop_backward_member<FIN, FOUT, true, MAXN == COOP_MAX_N ? COOP_SMALL_N : MAXN>(a, blk, smem, srec);     // (uniform)
          else coop_backward_member<FIN, FOUT, true, MAXN>(a, blk, smem, srec);
        }
        return;
      }
      blk -= a.coop_slabs;
      if (blk >= a.plain_count) {
        const int gp = first + a.plain_count + 2 * (blk - a.plain_count) + (int)(threadIdx.x >> 9);
        spec_backward_body<FIN, FOUT, NW_SMALL, true, true, DUAL_RPW>(a, gp, smem, load_graph(a.graph_desc + gp));
        return;
      }
    }
    {
      const GraphRec g = load_graph(a.graph_desc + first + blk);
      if (NW == NW_LARGE && g.n <= 2 * NW) spec_backward_body<FIN, FOUT, NW, false, true, NW == NW_LARGE ? 2 : 0>(a, first + blk, smem, g);
      else if (NW == NW_LARGE && g.n <= 4 * NW) spec_backward_body<FIN, FOUT, NW, false, true, NW == NW_LARGE ? 4 : 0>(a, first + blk, smem, g);
      else spec_backward_body<FIN, FOUT, NW, false, true>(a, first + blk, smem, g);
    }
  } else if constexpr (FULL) {
    // grid: [members of the cooperative graphs][one workgroup per graph: plain_count][two small graphs per workgroup]
    //       [parameter-gradient tiles: tail_tiles (spec_tail.h)]
    int blk = (int)blockIdx.x;
    if (a.tail_tiles > 0 && blk >= (int)gridDim.x - a.tail_tiles) {
      tail_param_grads<FIN, FOUT>(a, blk - ((int)gridDim.x - a.tail_tiles), smem);
      return;
    }
    if (blk < a.coop_slabs) {   // (spec_backward_coop.h)
      {
          const SlabRec srec = load_slab(a.slab_desc + blk);
          if (MAXN == COOP_MAX_N && srec.n <= COOP_SMALL_N) coop_backward_member<FIN, FOUT, false, MAXN == COOP_MAX_N ? COOP_SMALL_N : MAXN>(a, blk, smem, srec);     // (uniform)
          else coop_backward_member<FIN, FOUT, false, MAXN>(a, blk, smem, srec);
        }
      return;
    }
    blk -= a.coop_slabs;
    if (blk >= a.plain_count) {
      const int gp = first + a.plain_count + 2 * (blk - a.plain_count) + (int)(threadIdx.x >> 9);
      spec_backward_body<FIN, FOUT, NW_SMALL, true, false, DUAL_RPW>(a, gp, smem, load_graph(a.graph_desc + gp));
      return;
    }
    {
      const GraphRec g = load_graph(a.graph_desc + first + blk);
      if (g.n <= 2 * NW_LARGE) spec_backward_body<FIN, FOUT, NW_LARGE, false, false, 2>(a, first + blk, smem, g);     // (uniform)
      else if (g.n <= 4 * NW_LARGE) spec_backward_body<FIN, FOUT, NW_LARGE, false, false, 4>(a, first + blk, smem, g);
      else spec_backward_body<FIN, FOUT, NW_LARGE, false, false>(a, first + blk, smem, g);
    }
  } else {
    if constexpr (NW == NW_LARGE) {
      if (a.tail_tiles > 0 && (int)blockIdx.x >= (int)gridDim.x - a.tail_tiles) {
        tail_param_grads<FIN, FOUT>(a, (int)blockIdx.x - ((int)gridDim.x - a.tail_tiles), smem);
        return;
      }
    }
    {
      const GraphRec g = load_graph(a.graph_desc + first + (int)blockIdx.x);
      // (the launches without members or pairs -- batches of 1024 graphs and up on eight waves, or sixteen -- by the rows a wave owns)
      if (g.n <= 2 * NW) spec_backward_body<FIN, FOUT, NW, false, false, 2>(a, first + (int)blockIdx.x, smem, g);     // (uniform)
      else if (g.n <= 4 * NW) spec_backward_body<FIN, FOUT, NW, false, false, 4>(a, first + (int)blockIdx.x, smem, g);
      else spec_backward_body<FIN, FOUT, NW, false, false>(a, first + (int)blockIdx.x, smem, g);
    }
  }
}

}  // namespace spec

// LDS bytes of the backward kernel for a largest graph of n nodes; SIZE_MAX when a wave would own more tiles (or
// node-feature rows) than its register arrays hold
size_t spec_bwd_lds_bytes(int32_t n, int32_t Fin, int32_t Fout, int32_t nw, int32_t gnp) {
  const int F4 = (Fin + 15) & ~15, ntm = (n + 15) / 16;
  const int lim = spec::MAXTB * nw;
  if (ntm * (F4 / 16) > lim || ntm * ntm > lim || ntm * 16 > (nw == spec::NW_SMALL ? 6 : 8) * nw) return ~(size_t)0;
  if (nw == spec::NW_SMALL && n > spec::SMALL_N) return ~(size_t)0;
  return sizeof(float) * (size_t)spec::bwd_lds(n, F4, Fout, nw, gnp).total;
}

// LDS bytes of a member of the cooperative backward for a graph of n nodes
size_t spec_coop_lds_bytes(int32_t n, int32_t Fin, int32_t Fout, int32_t gnp) {
  return sizeof(float) * (size_t)spec::coop_lds(n, (Fin + 15) & ~15, Fout, gnp).total;
}

template <int FIN, int FOUT, int NW>
static int launch_spec_backward_t(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, hipStream_t s) {
  if (a.variant == AGCN_SGC_LL_RESLAP) {
    if ((a.dual_pairs > 0 || a.coop_slabs > 0) && NW != spec::NW_LARGE) return fail(AGCN_EINVAL, "spec_backward: members / paired graphs need the 16-wave launch");
    SgcArgs aa = a;
    aa.plain_count = count;   // one-graph workgroups; behind them a.dual_pairs workgroups of two small graphs each
    auto kern = &spec::spec_backward_kernel<FIN, FOUT, NW, 0, true>;
    if constexpr (NW == spec::NW_LARGE) {
      if (a.coop_wide && a.coop_slabs > 0) kern = &spec::spec_backward_kernel<FIN, FOUT, NW, 2, true>;
    }
    if (lds_bytes > 64 * 1024) {
      hipError_t e = static_cast<hipError_t>(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes));
      if (e != hipSuccess) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(spec_bwd): %s", hipGetErrorString(e));
    }
    prof_begin(PK_SPEC_BWD, s);
    hipLaunchKernelGGL(kern, dim3((unsigned)(count + (NW == spec::NW_LARGE ? a.coop_slabs + a.dual_pairs : 0))), dim3(64 * NW), lds_bytes, s, aa, first);
    prof_end(s);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "spec_backward launch: %s", hipGetErrorString(e));
    return AGCN_OK;
  }
  const bool full = NW == spec::NW_LARGE && (a.coop_slabs > 0 || a.dual_pairs > 0);
  auto kern = &spec::spec_backward_kernel<FIN, FOUT, NW, 0>;
  if constexpr (NW == spec::NW_LARGE) {
    if (full) kern = (a.coop_wide && a.coop_slabs > 0) ? &spec::spec_backward_kernel<FIN, FOUT, NW, 2> : &spec::spec_backward_kernel<FIN, FOUT, NW, 1>;
  }
  if (lds_bytes > 64 * 1024) {
    hipError_t e = static_cast<hipError_t>(ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds_bytes));
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "hipFuncSetAttribute(spec_bwd): %s", hipGetErrorString(e));
  }
  prof_begin(PK_SPEC_BWD, s);
  const unsigned members = NW == spec::NW_LARGE ? (unsigned)(a.coop_slabs + a.dual_pairs) : 0u;
  SgcArgs aa = a;
  aa.plain_count = count;   // one-graph workgroups; behind them a.dual_pairs workgroups of two small graphs each
  const unsigned tiles = (NW == spec::NW_LARGE && aa.tail != nullptr) ? (unsigned)aa.tail_tiles : 0u;
  if (tiles == 0) { aa.tail = nullptr; aa.tail_tiles = 0; }
  hipLaunchKernelGGL(kern, dim3(members + count + tiles), dim3(64 * NW), lds_bytes, s, aa, first);
  prof_end(s);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "spec_backward launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

template <int FIN, int FOUT>
static int launch_spec_backward_w(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, int32_t nw, hipStream_t s) {
  if (nw == spec::NW_SMALL) return launch_spec_backward_t<FIN, FOUT, spec::NW_SMALL>(a, first, count, lds_bytes, s);
  return launch_spec_backward_t<FIN, FOUT, spec::NW_LARGE>(a, first, count, lds_bytes, s);
}

int launch_spec_backward(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, int32_t nw, void* stream) {
  if (count <= 0 && a.coop_slabs <= 0 && a.dual_pairs <= 0) return AGCN_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.Fin == 75 && a.Fout == 64) return launch_spec_backward_w<75, 64>(a, first, count, lds_bytes, nw, s);
  if (a.Fin == 64 && a.Fout == 128) return launch_spec_backward_w<64, 128>(a, first, count, lds_bytes, nw, s);
  if (a.Fin == 128 && a.Fout == 128) return launch_spec_backward_w<128, 128>(a, first, count, lds_bytes, nw, s);
  if (a.Fin == 128 && a.Fout == 64) return launch_spec_backward_w<128, 64>(a, first, count, lds_bytes, nw, s);
  if (a.Fin == 64 && a.Fout == 64) return launch_spec_backward_w<64, 64>(a, first, count, lds_bytes, nw, s);
  return fail(AGCN_EINVAL, "spec_backward: unsupported layer shape %d -> %d", a.Fin, a.Fout);
}

}  // namespace agcn
