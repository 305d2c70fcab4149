// A large graph's backward spread over SEVERAL workgroups of the specialised backward launch (spec_backward.hip).
//
// With one workgroup per graph a 256-graph step is as long as the dependent chain of its largest graph on ONE CU
// (n = 59: 26 us, while the median graph is done after 14 us and its CU idles).  Here every 16-row tile of a large
// graph (n >= COOP_MIN_N) gets a workgroup of its own -- a "member", one per slab of the plan's slab table, so the
// table's (graph, first row) pairs are the member list -- which computes everything that belongs to its 16 rows:
//   dX[I], dZ[I], the rows I of Gp, and its share of dalpha.
// What a member needs from the other rows it recomputes (g_1 = Gp W_1^T for all rows: the rows of dY / Y are read by
// every member) -- except ONE vector that depends on whole columns of the tile: dd_k = -1/2 s_k^3 ds_k (the degree
// gradient), needed for every k by the pair pass.  The members exchange it through global memory: each publishes its
// 16 values as (value, epoch + 1) 64-bit words with agent-scope stores and polls the others' words until their flag
// matches (one store-to-load latency, no separate flag).  The words live in a fixed-size HEADER at the start of the
// reserve (COOP_HEADER_WORDS 64-bit words: per direction one slot per member row and one epoch word per member, keyed
// by the graph's first slab).  The header is zero when a reserve is first used -- the caller's only obligation -- and
// only ever holds (value, epoch) words afterwards: a graph's epoch word is read by every member when it starts and
// bumped by member 0 when it is done, so words of earlier launches, whatever batch they belonged to, never match.
// Members are the FIRST workgroups of the launch (dispatched before any other), at most COOP_MAX_MEMBERS of them, so
// all members of a graph are resident while they wait for each other; the waits are bounded all the same (10 ms of wall
// time), and a wait that runs out is reported: status word + NaN, see coop_wait_word.
// The forward is spread the same way (coop_forward_member): a member computes Z for all rows, the rows I of the
// similarity tile and their degrees -- and the members exchange s = d^-1/2.
#ifndef AGCN_SPEC_BACKWARD_COOP_H_
#define AGCN_SPEC_BACKWARD_COOP_H_

#include <float.h>

#include "spec_kernels.h"
#include "spec_bfpipe.h"

namespace agcn {
namespace spec {

constexpr int COOP_MIN_N = 33;         // graphs of at least three 16-row tiles (while the launch fits the chip)
constexpr int COOP_MIN_N_FULL = 65;    // ... of at least five, when every extra workgroup has to wait for a CU
// At most this many members per launch: all members of a graph must be resident together.  In ONE launch they are (first
// workgroups of the grid, fewer than the CUs); 120 keeps that true when TWO launches with members run side by side on
// different streams (epoch lanes): 240 members still fit 256 CUs.  (Beyond that a wait can run out: bounded, loud.)
constexpr int COOP_MAX_MEMBERS = 120;
constexpr int DUAL_MAX_N = 16;         // graphs that may be packed two to a workgroup (8 waves each)
// Largest graph the members can take: a member holds [n x F] images and 16-row strips, not n x n tiles, so it reaches
// beyond the 96 nodes of the one-workgroup kernels (whose backward keeps two n x n tiles) as far as its LDS plan fits
// (spec_coop_lds_bytes): lanes cover MAXN columns in MAXN / 64 passes, a wave owns MAXN / 16 rows of an image (register arrays),
// the epoch table has MAXN / 16 entries -- the member code is templated on MAXN:
constexpr int COOP_MAX_N = 128;
// ... and a variant for graphs of up to 64 nodes (one column pass, four rows of an image per wave): the unrolled per-row / per-pass loops
// of the 128-node code run eight / two iterations whatever the graph -- dead ones are predicated off but issued (the headline batch's
// members all have 33-59 nodes)
constexpr int COOP_SMALL_N = 64;
// ... and the WIDE members (a kernel variant of their own, launched only for a batch with such a graph: spec_backward.hip):
// three 64-column passes, 12 rows of an image per wave -- HIV-shaped batches: 5.6 % have a molecule beyond 128 atoms, none beyond 192
constexpr int COOP_MAX_N_WIDE = 192;
// exchange header at the start of the reserve, per direction (0 backward, 1 forward): one slot per member row, one
// epoch word per member (a graph uses the rows / the epoch of its first slab)
constexpr int COOP_DIR_WORDS = COOP_MAX_MEMBERS * 16 + COOP_MAX_MEMBERS;
constexpr int COOP_HEADER_WORDS = 2 * COOP_DIR_WORDS;
// A member's wait for a sibling's word is bounded in WALL time (the 100 MHz constant clock): 10 ms, three orders of
// magnitude beyond a step.  A wait that runs out counts itself in the STATUS word behind the exchange words (sticky: the
// library never clears it; agcn_sgc_ll_reserve_status reads it) and poisons the value it was waiting for with NaN: a
// timed-out step is wrong loudly, in the gradients AND in the return code of the next status check, never silently.
constexpr int COOP_STATUS_WORD = COOP_HEADER_WORDS;     // 64-bit word index inside the reserve header
}  // namespace spec
}  // namespace agcn
#include "spec_tail_proto.h"
namespace agcn {
namespace spec {
// SGC_LL_Reslap members also exchange graph-wide SUMS (the two clip norms of the forward, the two clip inner products of the
// backward): one 64-bit (value, epoch) word per member, direction and kind, behind the tail words of the header
constexpr int COOP_SCAL_WORD = (TAIL_OFF_FLOATS + TAIL_WORDS + 1) / 2;     // 64-bit word index inside the reserve header
constexpr int COOP_SCAL_WORDS = 4 * COOP_MAX_MEMBERS;                      // [direction][kind][member]
constexpr unsigned COOP_TIMEOUT_TICKS = 1000000u;   // 10 ms at 100 MHz: the default of agcn_set_coop_timeout_ticks (SgcArgs::coop_ticks)
__device__ __forceinline__ uint64_t coop_wait_word(uint64_t* status, uint32_t* dev_status, uint32_t ticks, const uint64_t* word, uint32_t want, bool* ok) {
  const unsigned long long t0 = wall_clock64();
  uint64_t w;
  for (;;) {
    w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(w >> 32) == want) { *ok = true; return w; }
    if (wall_clock64() - t0 >= (unsigned long long)ticks) break;
    __builtin_amdgcn_s_sleep(2);
  }
  __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(status), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (dev_status != nullptr) __hip_atomic_fetch_add(dev_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  *ok = false;
  return w;
}

#if defined(__HIPCC__)
// Sum of one scalar over ALL members of a graph: this member's `part` is summed over the workgroup, published as a (value,
// epoch) word in its slot (slots[me]), the siblings' words are polled, and every member adds the ntm values in member order --
// the same bits in every member.  wants[m]: the epoch tag member m publishes under.  Contains barriers (uniform control flow).
__device__ __forceinline__ float coop_graph_sum(const SgcArgs& a, float part, float* red, float* gs, uint64_t* slots, int me, int ntm,
                                                const uint32_t* wants, int tid) {
  const float mine = block_sum(part, red);
  if (tid == 0) {      // published before anyone polls (a divergent branch could put a spinning sibling poll in front of it)
    __hip_atomic_store(slots + me, ((uint64_t)wants[me] << 32) | (uint64_t)__float_as_uint(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gs[me] = mine;
  }
  if (tid < ntm && tid != me) {
    bool got;
    const uint64_t w = coop_wait_word(a.sv_xch + COOP_STATUS_WORD, a.dev_status, a.coop_ticks, slots + tid, wants[tid], &got);
    gs[tid] = got ? __uint_as_float((uint32_t)w) : __builtin_nanf("");
  }
  __syncthreads();
  float tot = 0.f;
  for (int m = 0; m < ntm; ++m) tot += gs[m];
  __syncthreads();
  return tot;
}
#endif

struct CoopLds {
  int n4, ldt;
  int o_s, o_dd, o_rs, o_lt, o_r, o_dlr, o_dlc, o_g0, o_dz, o_g1, o_u, total;
};
// Regions with disjoint lifetimes share their floats (a barrier lies between the last read of the one and the first write of the
// other): the dL COLUMN strip (written in M4) takes the place of the L^T strip (last read in M3); and where a strip row is at
// least as long as an image row (n4 >= F4: the large graphs, the only ones whose plan is tight) the g_0 rows (M2 -> M3) live in
// the dL row strip (written from M4 on) and the dZ rows (M7 -> M8) in the L^T / dL column strip (last read in M6).  That is what
// lets a 128-node graph fit at 128 input features (159 KB), 176 nodes at 75 -> 64 and 160 at 64 -> 128.
AGCN_HD inline CoopLds coop_lds(int n, int F4, int Fout, int gnp = 0) {
  CoopLds d;
  d.n4 = (n + 15) & ~15;
  d.ldt = d.n4 + 2;
  const int ldx = F4 + 2, ldg = Fout + 2;
  d.o_s = 32;
  d.o_dd = d.o_s + d.n4;
  d.o_rs = d.o_dd + d.n4;
  d.o_lt = d.o_rs + 16;                  // [16][ldt]  L^T strip: LT[il][k] = L[k][i0 + il]
  d.o_r = d.o_lt + 16 * d.ldt;           // [16][ldt]  r rows of the strip
  d.o_dlr = d.o_r + 16 * d.ldt;          // [16][ldt]  dL[i0 + il][j]   -> c strip
  d.o_dlc = d.o_lt;                      // [16][ldt]  dL[j][i0 + il]   (in place of the L^T strip)
  int next = d.o_dlr + 16 * d.ldt;
  if (d.n4 >= F4) {
    d.o_g0 = d.o_dlr;                    // [16][ldx]  g_0 rows of the strip (until M3)
    d.o_dz = d.o_lt;                     // [16][ldx]  dZ rows of the strip (from M7 on)
  } else {
    d.o_g0 = d.o_dz = next;              // [16][ldx]  g_0 rows of the strip, later the dZ rows
    next += 16 * ldx;
  }
  d.o_g1 = next;                         // [n4][ldx]  g_1 (all rows), later Z (all rows)
  d.o_u = d.o_g1 + d.n4 * ldx;           // [n4][ldg]  Gp (all rows; gnp = 3: three bf16 piece planes), later [n4][ldx] X_0 (all rows)
  const int gp = gnp == 3 ? 3 * d.n4 * bfp_stride(Fout) : d.n4 * ldg, x = d.n4 * ldx;
  d.total = d.o_u + (gp > x ? gp : x);
  return d;
}

#if defined(__HIPCC__)
__device__ __forceinline__ uint64_t coop_pack(float v, uint32_t flag) {
  return ((uint64_t)flag << 32) | (uint64_t)__float_as_uint(v);
}

// One member: slab `sl` of the plan's slab table.  1024 threads.
// RESLAP (SGC_LL_Reslap, paper mode): L is the forward's 4th output La (its COLUMN strip is loaded: La need not be symmetric);
// between dLa = g_1 X_0^T + dL_sum and the pair passes sit the adjoints of the two clip + leaky-ReLU stages -- element-wise but
// for two graph-wide inner products, which the members exchange as sums (coop_graph_sum).  Every element-wise quantity is
// evaluated for (i, j) AND (j, i) by the thread that owns (row i of the strip, column j): the row strips of L_pre / L_prev /
// dL_sum arrive as coalesced loads, the column strips as gathers, all in that thread's registers.
template <int FIN, int FOUT, bool RESLAP = false, int MAXN = COOP_MAX_N>
__device__ __forceinline__ void coop_backward_member(const SgcArgs& a, const int sl, float* smem, const SlabRec& srec) {
  __shared__ float coop_gs[MAXN / 16];
  constexpr int NCP = MAXN / 64;               // 64-column passes of a strip row (lanes cover MAXN columns)
  constexpr int NW = 16, NT = 1024;
  using FT = Feat<FIN>;
  constexpr int F4 = FT::F4, LDX = FT::LDX, NCB = FT::NCB, NKS = FT::NKS, CP = FT::CP;
  constexpr int LDG = FOUT + 2, CPO = (FOUT + 63) / 64, NKW = FOUT / 4;
  constexpr int RPW = MAXN / 16;               // rows of an [n x F] matrix one wave may own (n4 <= MAXN)
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = uni(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int b = srec.b, i0 = srec.i0, n = srec.n, row0 = srec.row0;
  const uint32_t toff = srec.toff;
  const CoopLds d = coop_lds(n, F4, FOUT, a.g_np);
  const int n4 = d.n4, ldt = d.ldt, ntm = n4 >> 4, rpw = n4 / NW;
  const int me = i0 >> 4;                      // member index == this strip's row tile
  const bool relu = a.activation == AGCN_ACT_RELU;
  // REFERENCE mode (R = I: no gradient reaches the similarity tile): the members exist for the graphs no single workgroup can hold
  // (n > 96); the pair passes, the dd exchange and dZ are skipped -- SGC_LL: dX[I] = g_0[I] + L^T[I, :] g_1 and the Gp rows;
  // SGC_LL_Reslap: also the clip adjoints towards dL_prev / dbeta / dalpha (one exchanged inner product)
  const bool paper = a.mode == AGCN_MODE_PAPER;
  const bool tail = !RESLAP && a.tail != nullptr;         // (spec_tail.h)
  const uint32_t tgen = tail ? tail_load(a.tail + TAIL_GEN) : 0u;
  if (tail && tid == 0) tail_started(a, tgen);
  float* red = smem;
  float* s_vec = smem + d.o_s;
  float* ddv = smem + d.o_dd;
  float* rsv = smem + d.o_rs;
  float* LT = smem + d.o_lt;
  float* RS = smem + d.o_r;
  float* DLr = smem + d.o_dlr;
  float* DLc = smem + d.o_dlc;
  float* G0img = smem + d.o_g0;
  float* DZimg = smem + d.o_dz;
  float* G1img = smem + d.o_g1;
  float* Zimg = smem + d.o_g1;
  float* GPimg = smem + d.o_u;
  float* X0img = smem + d.o_u;
  const float* Xg = a.X + (int64_t)row0 * a.ldx;
  const float* Yg = a.Y + (int64_t)row0 * a.ldy;
  const float* dYg = a.dY + (int64_t)row0 * a.ldy;
  const float alpha = a.alpha[0];
  uint64_t* xch = a.sv_xch + (sl - me) * 16;                        // this graph's slots (direction 0): row i at xch + i
  uint64_t* xep = a.sv_xch + COOP_MAX_MEMBERS * 16 + (sl - me);     // ... and its members' epoch words
  uint64_t* xsc = a.sv_xch + COOP_SCAL_WORD + (sl - me);            // scalar sums, direction 0, kind 0; kind 1 COOP_MAX_MEMBERS on
  // Every member has an epoch word of its own (keyed by ITS slab index, like its 16 slots: a slot is only ever written
  // under one epoch counter, whatever graphs of whatever batches use it over time).  A member reads the epochs of all
  // members of its graph before it publishes anything and bumps its own when it is done -- by then it holds every
  // sibling's words, so every sibling has read the epochs.
  uint32_t* wants = reinterpret_cast<uint32_t*>(smem + 16);   // [<= 16] in the (otherwise unused) tail of the `red` area
  // (requested by the first wave under a UNIFORM branch and stored behind the first round of loads: as `wants[tid] = load + 1`
  //  under `tid < ntm` the first wave sat out a whole agent-scope round trip -- 0.7 us -- before it issued any of its loads, and the
  //  first barrier waited for it)
  uint64_t epw = 0;
  if (wave == 0) epw = __hip_atomic_load(xep + min(lane, ntm - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  AGCN_STAMP(5, 0);
  // ---- M0: one round of loads ---------------------------------------------------------------------------------
  int csr_p0 = 0, csr_p1 = 0;
  if (!RESLAP && a.L0_rowptr && tid < n) {
    csr_p0 = a.L0_rowptr[row0 + tid];
    csr_p1 = a.L0_rowptr[row0 + tid + 1];
  }
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
  // r rows of the strip: wave il owns row i0 + il (the forward stored both triangles for these graphs)
  float rr[NCP];
  {
    const int i = min(i0 + wave, n - 1);
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp) {
      rr[cp] = 0.f;
      if (paper && 64 * cp < n) rr[cp] = a.sv_r[toff + (uint32_t)(i * n + min(lane + 64 * cp, n - 1))];
    }
  }
  // Reslap: the COLUMN strip of La (wave il = column i0 + il, lanes run over the rows k): LT[il][k] = La[k][i0 + il]
  float lac[NCP] = {};
  if constexpr (RESLAP) {
    const int i = min(i0 + wave, n - 1);
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp)
      if (64 * cp < n) lac[cp] = a.L_sum[toff + (uint32_t)(min(lane + 64 * cp, n - 1) * n + i)];
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
  if (a.feat_bf16) {     // (widened HERE, behind every load of the round: spec_kernels.h, bf16_load_raw)
#pragma unroll
    for (int u = 0; u < RPW; ++u)
#pragma unroll
      for (int cp = 0; cp < CPO; ++cp) {
        gy[u][cp] = bf16_widen(gy[u][cp]);
        yy[u][cp] = bf16_widen(yy[u][cp]);
      }
  }
  float gpown[CPO];   // Gp row i0 + wave (this wave loaded it: rows wave, wave + 16, ..), for the hand-off store
#pragma unroll
  for (int cp = 0; cp < CPO; ++cp) gpown[cp] = 0.f;
  // bf16 features + the W^T piece image of the forward: Gp is ONE bf16 piece, phase M2 runs on the bf16 pipe (spec_bfpipe.h) and
  // the image is a plane of bf16 rows in the place of the fp32 one
  const int gnp = FOUT == 64 ? a.g_np : 0;              // (launch argument: uniform) 0: fp32 instruction; 1 | 3 piece planes
  const bool gbf = gnp != 0;
  constexpr int SGP = bfp_stride(FOUT), KBW = FOUT / 32;
  uint32_t* GPpl = reinterpret_cast<uint32_t*>(GPimg);
  const int gplane = n4 * SGP;
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const int i = wave + NW * u;
    float vv[CPO];
#pragma unroll
    for (int cp = 0; cp < CPO; ++cp) {
      const int c = lane + 64 * cp;
      const float v = (i < n && c < FOUT && (!relu || yy[u][cp] > 0.f)) ? gy[u][cp] : 0.f;
      vv[cp] = v;
      if (u == me) gpown[cp] = v;
      if (!gbf && i < n4 && c < FOUT) GPimg[i * LDG + c] = v;
    }
    if (gbf) {
      float lo, hi;
      int col;
      bfp_pair_columns<(CPO > 1)>(vv[0], vv[CPO > 1 ? 1 : 0], lane, lo, hi, col);
      if (i < n4 && col >= 0 && col < FOUT) {
        if (gnp == 3) bfp_store_pair<3>(GPpl, gplane, i * SGP + (col >> 1), lo, hi);
        else bfp_store_pair<1>(GPpl, 0, i * SGP + (col >> 1), lo, hi);
      }
    }
  }
  {
    const int i = i0 + wave;
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp) {
      const int j = lane + 64 * cp;
      if (j < n4) RS[wave * ldt + j] = (i < n && j < n && j != i) ? rr[cp] : 0.f;
    }
  }
  if (tid < n4) s_vec[tid] = tid < n ? s_own : 0.f;
  if (tid < ntm) wants[tid] = (uint32_t)epw + 1u;
  lds_barrier();
  AGCN_STAMP(5, 1);

  // ---- M1: L^T strip  LT[il][k] = lrelu(R_k,i) + L_orig[k][i],  R_ki = delta - (s_k S) s_i  (the forward's operand order)
  if constexpr (RESLAP) {
    const int i = i0 + wave;
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp) {
      const int k = lane + 64 * cp;
      if (k < n4) LT[wave * ldt + k] = (i < n && k < n) ? lac[cp] : 0.f;
    }
  }
  if constexpr (!RESLAP)
  {
    const int l0ld = a.L0_ld > 0 ? a.L0_ld : n;
    const float* L0d = !a.L0_dense ? nullptr
                                   : (a.L0_ld > 0 ? a.L0_dense + (int64_t)b * a.L0_ld * a.L0_ld : a.L0_dense + toff);
    const int i = i0 + wave;
    const float si = s_vec[min(i, n4 - 1)];
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp) {
      const int k = lane + 64 * cp;
      if (k < n4) {
        float out = 0.f;
        if (i < n && k < n) {
          const float S = (i == k || !paper) ? 0.f : fast_exp_neg(RS[wave * ldt + k]);
          float R = (i == k) ? 1.f : 0.f;
          if (paper) R = R - (s_vec[k] * S) * si;
          out = lrelu(R, alpha);
          if (L0d) out += L0d[(int64_t)k * l0ld + i];
        }
        LT[wave * ldt + k] = out;
      }
    }
  }
  lds_barrier();
  if (!RESLAP && a.L0_rowptr && tid < n) {      // CSR row tid: its entries in the columns of the strip
#pragma unroll
    for (int u = 0; u < 6; ++u)
      if (csr_p0 + u < csr_p1 && csr_c[u] >= i0 && csr_c[u] < i0 + 16) lds_add(&LT[(csr_c[u] - i0) * ldt + tid], csr_v[u]);
    for (int p = csr_p0 + 6; p < csr_p1; ++p) {
      const int c = a.L0_col[p];
      if (c >= i0 && c < i0 + 16) lds_add(&LT[(c - i0) * ldt + tid], a.L0_val[p]);
    }
  }

  AGCN_STAMP(5, 2);
  // ---- M2: g_1 = Gp W_1^T for ALL rows -> image;  g_0 = Gp[I] W_0^T -> image -------------------------------------
  {
    constexpr int CHW = 4, NCHW = NKW / (4 * CHW);   // 16-k blocks per chunk (spec_kernels.h: kb_frag / kb_mma)
    int ta, tb;
    tile_range<NW>((ntm + 1) * NCB, wave, ta, tb);
    if (gbf) {   // (uniform) three products per 32 k against the image's pieces, the next tile's pieces requested first
      const u32x4* img = static_cast<const u32x4*>(a.sv_WTI);
      u32x4 bpa[KBW][3], bpb[KBW][3];
      auto load_bp = [&](u32x4 (&bp)[KBW][3], int t) {
        const bool own = t >= ntm * NCB;
        const int cb = own ? t - ntm * NCB : t % NCB, kk = own ? 0 : 1;
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) {
          const u32x4* bt = img + (size_t)((kk * KBW + kb) * NCB + cb) * 192 + lane;
          bp[kb][0] = bt[0]; bp[kb][1] = bt[64]; bp[kb][2] = bt[128];
        }
      };
      auto tile = [&](const u32x4 (&bp)[KBW][3], int t) {
        const bool own = t >= ntm * NCB;
        const int rt = own ? me : t / NCB, cb = own ? t - ntm * NCB : t - rt * NCB;
        const uint32_t* arow = GPpl + (rt * 16 + r) * SGP;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) acc = bfp_mma_regs(acc, arow, gplane, gnp, kb, q, bp[kb]);
        const int f = cb * 16 + r;
        const bool fv = f < FIN;
        float* dst = own ? G0img + (4 * q) * LDX + f : G1img + (rt * 16 + 4 * q) * LDX + f;
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) dst[i4 * LDX] = fv ? acc[i4] : 0.f;
      };
      if (ta < tb) load_bp(bpa, ta);
      for (int t = ta; t < tb; t += 2) {
        if (t + 1 < tb) load_bp(bpb, t + 1);
        tile(bpa, t);
        if (t + 1 < tb) {
          if (t + 2 < tb) load_bp(bpa, t + 2);
          tile(bpb, t + 1);
        }
      }
    } else
    for (int t = ta; t < tb; ++t) {
      const bool own = t >= ntm * NCB;            // a g_0 tile of the strip
      const int rt = own ? me : t / NCB, cb = own ? t - ntm * NCB : t - rt * NCB;
      const int kk = own ? 0 : 1;
      const float* Ap = GPimg + (rt * 16 + r) * LDG;
      const float* wbase = a.sv_WT + kk * (FOUT * FIN);
      const int fcl = min(cb * 16 + r, FIN - 1);
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ch = 0; ch < NCHW; ++ch) {
        f32x4 wf[CHW];
#pragma unroll
        for (int u = 0; u < CHW; ++u) wf[u] = kb_frag(wbase, CHW * ch + u, q, fcl, FIN);
#pragma unroll
        for (int u = 0; u < CHW; ++u) acc = kb_mma(acc, Ap, CHW * ch + u, q, wf[u]);
      }
      const int f = cb * 16 + r;
      const bool fv = f < FIN;
      float* dst = own ? G0img + (4 * q) * LDX + f : G1img + (rt * 16 + 4 * q) * LDX + f;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) dst[i4 * LDX] = fv ? acc[i4] : 0.f;
    }
  }
  lds_barrier();
  AGCN_STAMP(5, 3);

  // ---- M3: dX[I] = g_0[I] + L^T[I, :] g_1 (waves < NCB keep their tile);  X_0 rows requested meanwhile ------------
  float xzr[RPW][CP];
#pragma unroll
  for (int u = 0; u < RPW; ++u) {
    const float* rp = Xg + (int64_t)min(wave + NW * u, n - 1) * a.ldx;
#pragma unroll
    for (int cp = 0; cp < CP; ++cp) {
      xzr[u][cp] = 0.f;
      if (u < rpw) xzr[u][cp] = rp[min(lane + 64 * cp, FIN - 1)];
    }
  }
  f32x4 dxacc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (wave < NCB) {
    const int cb = wave;
    const float* Ap = LT + r * ldt + q;
    const float* Bp = G1img + q * LDX + cb * 16 + r;
    const int f = cb * 16 + r;
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) dxacc[i4] = G0img[(4 * q + i4) * LDX + f];
    for (int k0 = 0; k0 < n4; k0 += 16) {
      float av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = Ap[k0 + 4 * u];
        bv[u] = Bp[(k0 + 4 * u) * LDX];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) dxacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], dxacc, 0, 0, 0);
    }
  }
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
  lds_barrier();
  AGCN_STAMP(5, 4);

  // ---- M4: dL row strip (g_1[I] X_0^T) and column strip ((g_1 X_0[I]^T)^T = X_0[I] g_1^T);  Z rows requested --------
  float lpr[NCP] = {}, lpc[NCP] = {}, lpv[NCP] = {}, dlsr[NCP] = {}, dlsc[NCP] = {};
  {
    const float* Zg = a.sv_Z + (int64_t)row0 * FIN;
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int i = min(wave + NW * u, n - 1);
#pragma unroll
      for (int cp = 0; cp < CP; ++cp) {
        xzr[u][cp] = 0.f;
        if (u < rpw) xzr[u][cp] = Zg[i * FIN + min(lane + 64 * cp, FIN - 1)];
      }
    }
    if constexpr (RESLAP) {      // the strips of the clip adjoints (consumed in M5), requested with the Z rows
      const bool has_prev = a.L_prev != nullptr && a.dL_prev != nullptr;
      const int i = min(i0 + wave, n - 1);
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp) {
        if (64 * cp < n) {
          const int j = min(lane + 64 * cp, n - 1);
          const uint32_t oij = toff + (uint32_t)(i * n + j), oji = toff + (uint32_t)(j * n + i);
          lpr[cp] = a.sv_Lpre[oij];
          lpc[cp] = a.sv_Lpre[oji];
          if (has_prev) lpv[cp] = a.L_prev[oij];
          if (a.dL_sum) { dlsr[cp] = a.dL_sum[oij]; dlsc[cp] = a.dL_sum[oji]; }
        }
      }
    }
    int ta, tb;
    tile_range<NW>(2 * ntm, wave, ta, tb);
    if (!paper && !RESLAP) tb = ta;      // (SGC_LL, reference mode: no gradient reaches the tile)
    for (int t = ta; t < tb; ++t) {
      const bool col = t >= ntm;
      const int ct = col ? t - ntm : t;
      const float* Ap = (col ? X0img : G1img) + (me * 16 + r) * LDX + q;
      const float* Bp = (col ? G1img : X0img) + (ct * 16 + r) * LDX + q;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = mfma_chain2<NKS>(Ap, Bp, 4, acc);
      float* dst = (col ? DLc : DLr) + (4 * q) * ldt + ct * 16 + r;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) dst[i4 * ldt] = acc[i4];
    }
  }
  lds_barrier();
  // Z rows -> image over g_1 (dead)
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

  AGCN_STAMP(5, 5);
  // ---- M5: pass 1 over the strip's rows (wave il = row i0 + il): dalpha, ds_i -> dd_i, published to the other members
  float dalpha_part = 0.f, dbeta_part = 0.f;
  float dlpk[NCP] = {};
  if constexpr (RESLAP) {
    // adjoints of  La = lrelu(cn(Lpre)),  Lpre = R2 + L_orig + beta L_prev,  R2 = lrelu(cn(R))   (graphconv_reslap.py:184-198),
    // for (i, j) and (j, i); the strips DLr / DLc hold g_1 X_0^T and become dR_ij / dR_ji
    const float* sc = a.sv_scal + 4 * b;
    const float ssR = sc[0], scR = sc[1], ssL = sc[2], scL = sc[3];
    const float invL = 1.0f / sqrtf(ssL), invR = 1.0f / sqrtf(ssR);
    const bool clipL = invL <= 1.0f, clipR = invR <= 1.0f;   // TF MinimumGrad: ties take the norm branch
    const bool has_prev = a.L_prev != nullptr && a.dL_prev != nullptr;
    const float beta = has_prev ? a.beta[0] : 0.f;
    const int i = i0 + wave;
    const bool iv = i < n;
    const float si = s_vec[min(i, n4 - 1)];
    float dij[NCP] = {}, dji[NCP] = {};     // dLc, then dLpre, dR1, dR
    float dot_part = 0.f;
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp) {
      const int j = lane + 64 * cp;
      if (iv && j < n) {
        const float dLa_ij = DLr[wave * ldt + j] + dlsr[cp], dLa_ji = DLc[wave * ldt + j] + dlsc[cp];
        const float Lc_ij = lpr[cp] * scL, Lc_ji = lpc[cp] * scL;
        dalpha_part += dLa_ij * fminf(Lc_ij, 0.f);
        dij[cp] = dLa_ij * lrelu_grad(Lc_ij, alpha);
        dji[cp] = dLa_ji * lrelu_grad(Lc_ji, alpha);
        dot_part += dij[cp] * lpr[cp];
      }
    }
    const float dot = coop_graph_sum(a, dot_part, red, coop_gs, xsc, me, ntm, wants, tid);
    const float k3 = clipL ? dot * invL * invL * invL : 0.f;
    float Rij[NCP] = {}, Rji[NCP] = {};
    float dotR_part = 0.f;
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp) {
      const int j = lane + 64 * cp;
      if (iv && j < n) {
        if (clipL) {
          dij[cp] = dij[cp] * invL - lpr[cp] * k3;
          dji[cp] = dji[cp] * invL - lpc[cp] * k3;
        }
        if (has_prev) {
          dbeta_part += dij[cp] * lpv[cp];
          dlpk[cp] = dij[cp] * beta;      // dL_prev[i][j]: stored with everything else at the end (no store in front of a load)
        }
        const float sj = s_vec[j];
        const float Sv = (i == j || !paper) ? 0.f : fast_exp_neg(RS[wave * ldt + j]);
        const float dlt = (i == j) ? 1.f : 0.f;
        Rij[cp] = paper ? dlt - (si * Sv) * sj : dlt;
        Rji[cp] = paper ? dlt - (sj * Sv) * si : dlt;
        const float R1ij = Rij[cp] * scR, R1ji = Rji[cp] * scR;
        dalpha_part += dij[cp] * fminf(R1ij, 0.f);
        dij[cp] = dij[cp] * lrelu_grad(R1ij, alpha);          // dR1
        dji[cp] = dji[cp] * lrelu_grad(R1ji, alpha);
        dotR_part += dij[cp] * Rij[cp];
      }
    }
    const float dotR = coop_graph_sum(a, dotR_part, red, coop_gs, xsc + COOP_MAX_MEMBERS, me, ntm, wants, tid);
    const float k3R = clipR ? dotR * invR * invR * invR : 0.f;
    float dsum = 0.f;
#pragma unroll
    for (int cp = 0; cp < NCP; ++cp) {
      const int j = lane + 64 * cp;
      if (j < n4) {
        float dRij = 0.f, dRji = 0.f;
        if (iv && j < n) {
          dRij = clipR ? dij[cp] * invR - Rij[cp] * k3R : dij[cp];
          dRji = clipR ? dji[cp] * invR - Rji[cp] * k3R : dji[cp];
          const float sj = s_vec[j];
          const float Sv = (i == j || !paper) ? 0.f : fast_exp_neg(RS[wave * ldt + j]);
          dsum += (-dRij * Sv) * sj + (-dRji * Sv) * sj;         // P_ij s_j + P_ji s_j, P = dA . S, dA = -dR
        }
        DLr[wave * ldt + j] = dRij;
        DLc[wave * ldt + j] = dRji;
      }
    }
    dsum = wave_sum(dsum);
    if (paper && lane == 0 && iv) {
      const float ddk = degree_grad(s_vec[i], dsum);   // dd_k = -1/2 s_k^3 ds_k
      ddv[i] = ddk;
      __hip_atomic_store(xch + i, coop_pack(ddk, wants[me]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (paper)
  {
    const int i = i0 + wave;
    float dsum = 0.f;
    if (i < n) {   // (uniform)
      const float si = s_vec[i];
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp) {
        const int j = lane + 64 * cp;
        if (j < n) {
          const float sj = s_vec[j];
          const float Sv = (i == j) ? 0.f : fast_exp_neg(RS[wave * ldt + j]);
          const float dlt = (i == j) ? 1.f : 0.f;
          const float Rij = dlt - (si * Sv) * sj, Rji = dlt - (sj * Sv) * si;
          const float dLij = DLr[wave * ldt + j], dLji = DLc[wave * ldt + j];
          dalpha_part += dLij * fminf(Rij, 0.f);
          const float Pij = -(dLij * lrelu_grad(Rij, alpha)) * Sv;   // dA_ij S_ij
          const float Pji = -(dLji * lrelu_grad(Rji, alpha)) * Sv;   // dA_ji S_ji
          dsum += Pij * sj + Pji * sj;
        }
      }
    }
    dsum = wave_sum(dsum);
    if (lane == 0 && i < n) {
      const float ddk = degree_grad(s_vec[i], dsum);   // dd_k = -1/2 s_k^3 ds_k
      ddv[i] = ddk;
      __hip_atomic_store(xch + i, coop_pack(ddk, wants[me]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // the other members' dd: poll their words until the flag is this backward's
  if (paper && tid < n && (tid < i0 || tid >= i0 + 16)) {
    bool got;
    const uint64_t w = coop_wait_word(a.sv_xch + COOP_STATUS_WORD, a.dev_status, a.coop_ticks, xch + tid, wants[tid >> 4], &got);
    ddv[tid] = got ? __uint_as_float((uint32_t)w) : __builtin_nanf("");   // (a wait that ran out is loud: NaN + status word)
  }
  lds_barrier();
  AGCN_STAMP(5, 6);

  // ---- M6: pass 2: c_ij = (dr_ij + dr_ji) / r_ij over the strip's rows, in place over the dL row strip; row sums -----
  if (paper) {   // (uniform: launch argument)
    const int i = i0 + wave;
    float rsum = 0.f;
    if (i < n) {
      const float si = s_vec[i], ddi = ddv[i];
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp) {
        const int j = lane + 64 * cp;
        if (j < n4) {
          float cij = 0.f;
          if (j < n && j != i) {
            const float rv = RS[wave * ldt + j];
            const float Sv = fast_exp_neg(rv);
            const float sj = s_vec[j];
            const float Rij = -((si * Sv) * sj), Rji = -((sj * Sv) * si);
            // (Reslap: the strips hold dR already)
            const float dAij = -(RESLAP ? DLr[wave * ldt + j] : DLr[wave * ldt + j] * lrelu_grad(Rij, alpha));
            const float dAji = -(RESLAP ? DLc[wave * ldt + j] : DLc[wave * ldt + j] * lrelu_grad(Rji, alpha));
            if (Sv != 0.f && rv > 0.f) {
              const float dr_ij = -Sv * (dAij * si * sj + ddv[j]);
              const float dr_ji = -Sv * (dAji * sj * si + ddi);
              cij = (dr_ij + dr_ji) * fast_rcp(rv);
            }
          }
          DLr[wave * ldt + j] = cij;
          rsum += cij;
        }
      }
    } else {
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp)
        if (lane + 64 * cp < n4) DLr[wave * ldt + lane + 64 * cp] = 0.f;
    }
    rsum = wave_sum(rsum);
    if (lane == 0) rsv[wave] = rsum;
  }
  lds_barrier();
  AGCN_STAMP(5, 7);

  // ---- M7: dZ[I] = rowsum(c) Z[I] - c[I, :] Z  -> image + registers ------------------------------------------------
  f32x4 zacc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (paper && wave < NCB) {
    const int cb = wave;
    const float* Ap = DLr + r * ldt + q;
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
    const int f = cb * 16 + r;
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int il = 4 * q + i4;
      const float v = (i0 + il < n) ? rsv[il] * Zimg[(i0 + il) * LDX + f] - acc[i4] : 0.f;
      DZimg[il * LDX + f] = v;
      zacc[i4] = v;
    }
  }
  lds_barrier();
  AGCN_STAMP(5, 8);

  // ---- M8: dX[I] += dZ[I] M_L^T, then every global store ------------------------------------------------------------
  if (wave < NCB) {
    const int cb = wave;
    const float* Ap = DZimg + r * LDX;
    const int fcl = min(cb * 16 + r, FIN - 1);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NKBM = F4 / 16, CHM = 4;       // 16-k blocks over the input features (the copy is zero beyond FIN)
    if (paper) {
#pragma unroll
    for (int u0 = 0; u0 < NKBM; u0 += CHM) {
      f32x4 bf[CHM];
#pragma unroll
      for (int uu = 0; uu < CHM; ++uu)
        if (u0 + uu < NKBM) bf[uu] = kb_frag(a.sv_MT, u0 + uu, q, fcl, FIN);
#pragma unroll
      for (int uu = 0; uu < CHM; ++uu)
        if (u0 + uu < NKBM) acc = kb_mma(acc, Ap, u0 + uu, q, bf[uu]);
    }
    }
    const int f = cb * 16 + r;
    if (f < FIN) {
      float* dXg = a.dX + (int64_t)row0 * a.lddx;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        const int i = i0 + 4 * q + i4;
        if (i < n) {
          if (a.feat_bf16) bf16_store(a.dX, (int64_t)(row0 + i) * a.lddx + f, dxacc[i4] + acc[i4]);
          else AGCN_ST(8, &dXg[(int64_t)i * a.lddx + f], dxacc[i4] + acc[i4]);
          if (tail) tail_store_f(a.ws_dZ + (int64_t)(row0 + i) * FIN + f, zacc[i4]);
          else if (paper) AGCN_ST(8, &a.ws_dZ[(int64_t)(row0 + i) * FIN + f], zacc[i4]);
        }
      }
    }
  }
  if (!tail && i0 + wave < n) {
#pragma unroll
    for (int cp = 0; cp < CPO; ++cp) {
      const int c = lane + 64 * cp;
      if (c < FOUT) AGCN_ST(8, &a.ws_Gp[(int64_t)(row0 + i0 + wave) * FOUT + c], gpown[cp]);
    }
  }
  if constexpr (RESLAP) {
    if (a.L_prev != nullptr && a.dL_prev != nullptr) {
      const int i = i0 + wave;
      if (i < n) {
#pragma unroll
        for (int cp = 0; cp < NCP; ++cp)
          if (lane + 64 * cp < n) a.dL_prev[toff + (uint32_t)(i * n + lane + 64 * cp)] = dlpk[cp];
      }
      if (i0 == 0 && tid < (int)srec.npad)
        a.dL_prev[toff + (uint32_t)(n * n) + tid] = 0.f;   // alignment floats behind the tile
    }
  }
  const float da = block_sum(dalpha_part, red);
  const float db = RESLAP ? block_sum(dbeta_part, red) : 0.f;
  if (tid == 0) {
    if (tail) {
      tail_store_f(a.ws_pg + 2 * (a.B + sl), da);
      if (i0 == 0) tail_store_f(a.ws_pg + 2 * b, 0.f);
    } else {
      a.ws_pg[2 * (a.B + sl) + 0] = da;
      a.ws_pg[2 * (a.B + sl) + 1] = db;
      if (i0 == 0) {
        a.ws_pg[2 * b + 0] = 0.f;
        a.ws_pg[2 * b + 1] = 0.f;
      }
    }
    // this member holds every sibling's words by now: its next launch gets a new epoch
    __hip_atomic_store(xep + me, (uint64_t)wants[me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  AGCN_STAMP(5, 9);
  if (tail) {      // (spec_tail.h: this slab's dZ rows are out -> its flag; every workgroup counts itself out)
    tail_publish(a, sl, tgen, tid == 0, [] { __syncthreads(); });
    return;
  }
  zero_tags(a);
  if (sl == 0) {   // arrival counters of the parameter-gradient kernel that follows
    for (int e = tid; e < a.zero_count; e += NT) a.zero_ptr[e] = 0;
  }
}

// ---- forward -------------------------------------------------------------------------------------------------------
#if defined(__HIPCC__)
// Squared distances of the row pair (zr[0], zr[1]) against this lane's column(s) over FIN features, Z^T rows LDZ floats apart
// (LDZ = 0: the run-time stride ldz).  TWO: columns (zc[0], zc[1]) -> accA = row 0, accB = row 1; else one column -> .x of each.
template <int FIN, int LDZ, bool TWO>
__device__ __forceinline__ void coop_pair_dist(const float* zr, const float* zc, f32x2& accA, f32x2& accB, int ldz = 0) {
  const int st = LDZ > 0 ? LDZ : ldz;
  if constexpr (TWO) {
#pragma unroll 5
    for (int f = 0; f < FIN; ++f) {
      const f32x2 u = *reinterpret_cast<const f32x2*>(zr + f * st);
      const f32x2 v = *reinterpret_cast<const f32x2*>(zc + f * st);
      const f32x2 dA = (f32x2){u.x, u.x} - v, dB = (f32x2){u.y, u.y} - v;
      accA = __builtin_elementwise_fma(dA, dA, accA);
      accB = __builtin_elementwise_fma(dB, dB, accB);
    }
  } else {
    f32x2 acc = {0.f, 0.f};                               // (row 0, row 1) against the one column
#pragma unroll 5
    for (int f = 0; f < FIN; ++f) {
      const f32x2 u = *reinterpret_cast<const f32x2*>(zr + f * st);
      const float v = zc[f * st];
      const f32x2 dd = u - (f32x2){v, v};
      acc = __builtin_elementwise_fma(dd, dd, acc);
    }
    accA.x = acc.x; accB.x = acc.y;
  }
}
#endif

struct CoopFwdLds {
  int n4, ldt, ldz;
  int o_s, o_st, o_x, o_z, o_x1, o_ml, total;
  int o_xp, o_mi, xp_plane;   // np > 0 (spec_bfpipe.h): the bf16 piece planes of X [np][n4][S] and the operand-order image of M_L (dwords)
};
// np = 0: Z = X M_L on the fp32 matrix instruction (parameter fragments from global memory); np = 1 | 3: on the bf16 pipe
// (spec_bfpipe.h), X as np piece planes (1: bf16 features, 3: fp32 features split exactly) + the M_L image, both in LDS
// row stride of the staged M_L: == 16 (mod 64) floats, so that the four k rows of a fragment read fall into four different
// 16-bank groups (lane (r, q) reads row 4 u + q, column 16 cb + r)
AGCN_HD constexpr int coop_ml_stride(int F4) { return ((F4 + 47) / 64) * 64 + 16; }
AGCN_HD inline CoopFwdLds coop_fwd_lds(int n, int F4, int np = 0) {
  CoopFwdLds d;
  d.n4 = (n + 15) & ~15;
  d.ldt = d.n4 + 2;
  d.ldz = d.n4 + 4;
  const int ldx = F4 + 2;
  d.o_s = 32;
  d.o_st = d.o_s + d.n4;              // [16][ldt]  S rows of the strip, then L rows
  d.o_x = d.o_st + 16 * d.ldt;        // [n4][ldx]  X_0 (all rows)
  d.o_z = d.o_x + d.n4 * ldx;         // [F4][ldz]  Z^T (all rows)
  d.o_x1 = d.o_z + F4 * d.ldz;        // [16][ldx]  X_1 rows of the strip
  d.total = d.o_x1 + 16 * ldx;
  d.o_xp = d.o_mi = d.xp_plane = 0;
  // M_L itself, staged ONCE per workgroup ([F4][coop_ml_stride(F4)], zero beyond the last feature): the fragments of Z = X M_L are
  // LDS reads then -- from global memory every row tile re-read the whole matrix with 64-byte pieces (n / 16 x 24 KB through the
  // CU's address path), a fragment round trip in front of every tile.  Layers of up to 80 input features, graphs of up to
  // COOP_MAX_N nodes (beyond, the plan is tight), fp32 matrix path only.
  d.o_ml = 0;
  if (np == 0 && F4 <= 80 && n <= COOP_MAX_N) {
    d.o_ml = d.total;
    d.total += F4 * coop_ml_stride(F4);
  }
  if (np > 0) {
    const int KP = bfp_kp(F4);
    d.total = (d.total + 3) & ~3;       // (16-byte aligned: the planes and the image are read with ds_read_b128)
    d.xp_plane = d.n4 * bfp_stride(KP);
    d.o_xp = d.total;
    d.o_mi = d.o_xp + np * d.xp_plane;
    d.total = d.o_mi + bfp_image_dwords(KP, F4);
  }
  return d;
}

// One forward member: slab `sl`.  Same math as spec_forward_kernel, restricted to the 16 rows of the strip:
//   Z (all rows, recomputed by every member) -> S[I, :], r[I, :], d[I] -> s[I] published, s of the other rows polled
//   -> res_L[I, :], L[I, :] -> X_1[I] = L[I, :] X_0 -> Y[I].
// The distance sums run over the features in index order for both (i, j) and (j, i), so S stays bit-symmetric across
// members; degrees are wave sums (not the serial column sums of the one-workgroup kernel: last-ulp differences).
// RESLAP (SGC_LL_Reslap, graphconv_reslap.py:184-198): the two clip_by_norm stages need ||R||_F and ||L_pre||_F of the WHOLE graph:
// two more exchanges (coop_graph_sum), the same sums in the same order in every member.
template <int FIN, int FOUT, bool RESLAP = false, int MAXN = COOP_MAX_N>
__device__ __forceinline__ void coop_forward_member(const SgcArgs& a, const int sl, float* smem, const SlabRec& srec) {
  __shared__ float coop_gs[MAXN / 16];
  constexpr int NCP = MAXN / 64;               // 64-column passes of a strip row (lanes cover MAXN columns)
  constexpr int NW = 16;
  using FT = Feat<FIN>;
  constexpr int F4 = FT::F4, LDX = FT::LDX, NCB = FT::NCB, NKS = FT::NKS, CP = FT::CP;
  constexpr int NOB = FOUT / 16;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = uni(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  AGCN_STAMP(0, 14);
  const int b = srec.b, i0 = srec.i0, n = srec.n, row0 = srec.row0;
  const uint32_t toff = srec.toff;
  // Z = X M_L on the bf16 pipe (spec_bfpipe.h) where this graph's planes + the M_L image fit the launch's LDS
  constexpr int KP = bfp_kp(F4), SXP = bfp_stride(KP);
  int np = a.bfp_fwd ? (a.feat_bf16 ? 1 : 3) : 0;
  if (np > 0 && (size_t)coop_fwd_lds(n, F4, np).total * sizeof(float) > (size_t)a.lds_bytes) np = 0;
  np = uni(np);
  const CoopFwdLds d = coop_fwd_lds(n, F4, np);
  const int n4 = d.n4, ldt = d.ldt, ldz = d.ldz, ntm = n4 >> 4;
  const int me = i0 >> 4;
  const bool relu = a.activation == AGCN_ACT_RELU;
  const bool paper = a.mode == AGCN_MODE_PAPER;   // (reference mode: R = I, nothing to exchange)
  uint16_t* XPh = reinterpret_cast<uint16_t*>(smem + d.o_xp);
  u32x4* MI = reinterpret_cast<u32x4*>(smem + d.o_mi);
  float* s_vec = smem + d.o_s;
  float* St = smem + d.o_st;
  float* Ximg = smem + d.o_x;
  float* Zt = smem + d.o_z;
  float* X1s = smem + d.o_x1;
  const float* Xg = a.X + (int64_t)row0 * a.ldx;
  const float alpha = a.alpha[0];
  uint64_t* xch = a.sv_xch + COOP_DIR_WORDS + (sl - me) * 16;                     // direction 1
  uint64_t* xep = a.sv_xch + COOP_DIR_WORDS + COOP_MAX_MEMBERS * 16 + (sl - me);
  uint32_t* wants = reinterpret_cast<uint32_t*>(smem + 16);   // [<= 16], see coop_backward_member
  uint64_t epw = 0;      // (stored behind the first round of loads, see coop_backward_member)
  if ((paper || RESLAP) && wave == 0) epw = __hip_atomic_load(xep + min(lane, ntm - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint64_t* xsc = a.sv_xch + COOP_SCAL_WORD + 2 * COOP_MAX_MEMBERS + (sl - me);      // direction 1, kind 0; kind 1 COOP_MAX_MEMBERS on

  AGCN_STAMP(0, 0);
  // ---- P0: X rows (all), CSR rows of the strip ----------------------------------------------------------------------
  int csr_p0 = 0, csr_p1 = 0;
  if (a.L0_rowptr && tid < 16 && i0 + tid < n) {
    csr_p0 = a.L0_rowptr[row0 + i0 + tid];
    csr_p1 = a.L0_rowptr[row0 + i0 + tid + 1];
  }
  // M_L for the LDS copy (coop_fwd_lds): coalesced rows, requested with the first round of loads
  float* MLs = smem + d.o_ml;
  constexpr int SML = coop_ml_stride(F4);
  const bool ml_staged = d.o_ml > 0;      // (uniform)
  constexpr int PER = (F4 * F4 + 64 * NW - 1) / (64 * NW);
  float mv[PER];
  if (ml_staged) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int idx = min(tid + 64 * NW * e, F4 * F4 - 1);
      const int k = idx / F4, c = idx - k * F4;
      mv[e] = a.M_L[min(k, FIN - 1) * FIN + min(c, FIN - 1)];
    }
  }
  // (bf16 pipe) this thread's eight values of the M_L image, requested with the first round of loads
  constexpr int MI_THREADS = (KP / 32) * (F4 / 16) * 64;
  float mlv[8];
  if (np > 0) bfp_load_param<FIN, FIN, KP, F4>(mlv, a.M_L, FIN, 1, tid);      // (uniform branch; every thread loads, clamped)
  for (int ib = wave; ib < n4; ib += 4 * NW) {
    float xr[4][CP];
    by_dtype(a.feat_bf16 != 0, [&](auto BF) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t ro = (int64_t)min(ib + NW * u, n - 1) * a.ldx;
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
          const int c = min(lane + 64 * cp, FIN - 1);
          xr[u][cp] = 0.f;
          if (ib + NW * u < n4) {      // (uniform: rows beyond the image are not requested)
            if constexpr (decltype(BF)::value) xr[u][cp] = bf16_load(a.X, (int64_t)row0 * a.ldx + ro + c);
            else xr[u][cp] = Xg[ro + c];
          }
        }
      }
    });
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = ib + NW * u;
      if (i < n4) {
#pragma unroll
        for (int cp = 0; cp < CP; ++cp) {
          const int c = lane + 64 * cp;
          if (c < F4) Ximg[i * LDX + c] = (i < n && c < FIN) ? xr[u][cp] : 0.f;
        }
        if (np > 0) {   // (uniform) the same rows as bf16 piece planes, zero beyond the last feature / the last node
          const float v0 = (i < n && lane < FIN) ? xr[u][0] : 0.f;
          const float v1 = (CP > 1 && i < n && lane + 64 < FIN) ? xr[u][CP > 1 ? 1 : 0] : 0.f;
          float lo, hi;
          int col;
          bfp_pair_columns<(CP > 1)>(v0, v1, lane, lo, hi, col);
          if (col >= 0 && col < KP) {
            uint32_t* XPw = reinterpret_cast<uint32_t*>(XPh);
            if (np == 1) bfp_store_pair<1>(XPw, d.xp_plane, i * SXP + (col >> 1), lo, hi);
            else bfp_store_pair<3>(XPw, d.xp_plane, i * SXP + (col >> 1), lo, hi);
          }
        }
      }
    }
  }
  if (np > 0 && tid < MI_THREADS) bfp_store_param<FIN, FIN, KP, F4>(MI, mlv, tid);
  // M_L -> LDS (zero beyond the last feature)
  if (ml_staged) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int idx = tid + 64 * NW * e;
      const int k = idx / F4, c = idx - k * F4;
      if (idx < F4 * F4) MLs[k * SML + c] = (k < FIN && c < FIN) ? mv[e] : 0.f;
    }
  }
  // Reslap: the strip's rows of L_prev (wave il = row i0 + il), requested with the first round of loads
  float lprv[NCP] = {};
  if constexpr (RESLAP) {
    if (a.L_prev != nullptr) {
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp)
        if (64 * cp < n) lprv[cp] = a.L_prev[toff + (uint32_t)(min(i0 + wave, n - 1) * n + min(lane + 64 * cp, n - 1))];
    }
  }
  int csr_c[6] = {0, 0, 0, 0, 0, 0};
  float csr_v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (tid < 16 && csr_p1 > csr_p0) {
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int pp = min(csr_p0 + u, csr_p1 - 1);
      csr_c[u] = a.L0_col[pp];
      csr_v[u] = a.L0_val[pp];
    }
  }
  if ((paper || RESLAP) && tid < ntm) wants[tid] = (uint32_t)epw + 1u;
  lds_barrier();

  AGCN_STAMP(0, 1);
  // ---- P1: Z = X M_L for all rows -> Z^T image; the strip's rows also go to the reserve -----------------------------
  {
    float* Zg = a.sv_Z + (int64_t)row0 * FIN;
    const float* mq = a.M_L + q * FIN;
    int ta, tb;
    tile_range<NW>(ntm * NCB, wave, ta, tb);
    for (int t = ta; t < tb; ++t) {
      const int rt = t / NCB, cb = t - rt * NCB;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (np > 0) {   // (uniform) bf16 pipe: A pieces from the planes, B pieces from the image, both 16-byte LDS reads
        const uint32_t* arow = reinterpret_cast<const uint32_t*>(XPh) + (rt * 16 + r) * SXP;
#pragma unroll
        for (int kb = 0; kb < KP / 32; ++kb) {
          const u32x4* bt = MI + (size_t)(kb * NCB + cb) * 192;
          acc = np == 1 ? bfp_mma<1>(acc, arow, d.xp_plane, kb, q, bt, lane) : bfp_mma<3>(acc, arow, d.xp_plane, kb, q, bt, lane);
        }
      } else if (ml_staged) {   // (uniform) fragments from the staged matrix: rows 4 u + q (zero from row FIN on, as the A image's columns)
      const float* p = MLs + q * SML + cb * 16 + r;
      const float* Ap = Ximg + (rt * 16 + r) * LDX + q;
      acc = mfma_chain2<NKS>(Ap, p, 4 * SML, acc);
      } else {
      const float* p = mq + min(cb * 16 + r, FIN - 1);
      float bf[NKS];
#pragma unroll
      for (int u = 0; u < NKS; ++u) bf[u] = (4 * u + 3 < FIN) ? p[4 * u * FIN] : p[(min(4 * u + q, FIN - 1) - q) * FIN];
      const float* Ap = Ximg + (rt * 16 + r) * LDX + q;
      acc = mfma_chain<NKS>(Ap, bf, acc);
      }
      const int c = cb * 16 + r, ib = rt * 16 + 4 * q;
      if (c < FIN) {
        *reinterpret_cast<f32x4*>(Zt + c * ldz + ib) = acc;
        if (rt == me) {
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4)
            if (ib + i4 < n) AGCN_ST(2, &Zg[(ib + i4) * FIN + c], acc[i4]);
        }
      }
    }
  }
  lds_barrier();

  AGCN_STAMP(0, 2);
  // ---- P2: S[I, :], r[I, :], degree of the strip's rows (wave il = row i0 + il) -> s published -----------------------
  float rkeep[NCP] = {}, skeep[NCP] = {}, sd_keep = 0.f;
  // members of up to 128 nodes: TWO rows per wave (waves 0 .. 7: rows ia, ia + 1) against all columns, two columns per lane beyond
  // 64 nodes: the row pair's Z values come as ONE 8-byte broadcast read, the columns' as one read for both rows -- two LDS reads and
  // four packed instructions per feature for 256 pairs, with the Z^T row stride a compile-time constant per 16-node size class
  // (immediate offsets: no address arithmetic in the loop).  Before: one row per wave, three 4-byte reads (half the LDS rate),
  // four scalar instructions and two address adds per feature for 128 pairs.  Per pair still ONE fma chain over the features in
  // index order: S stays bit-symmetric across members.  The two degrees are summed with DPP row operations + four lane reads
  // (a __shfl_down tree is six dependent LDS permutes per sum) and inverted by lanes 0 and 1 side by side.
  constexpr bool ROWPAIR = NCP <= 2;
  f32x2 rk2[2] = {{0.f, 0.f}, {0.f, 0.f}}, sk2[2] = {{0.f, 0.f}, {0.f, 0.f}};      // [row of the pair](columns c0, c0 + 1 | c0)
  float sdk = 0.f;                                            // lane 0 | 1: s of row ia | ia + 1
  const bool two = n > 64;                                    // (uniform) two columns per lane: 2 lane, 2 lane + 1
  const int ia = i0 + 2 * wave, c0 = two ? 2 * lane : lane;
  if constexpr (ROWPAIR) {
    if (wave < 8) {
      float ds[2] = {0.f, 0.f};
      if (ia < n) {   // (uniform)
        f32x2 accA = {0.f, 0.f}, accB = {0.f, 0.f};           // row ia, row ia + 1 against columns (c0, c0 + 1)  [one column: .x]
        const float* zr = Zt + ia;                            // (8-byte aligned: ia, ldz, the region's offset are even)
        const float* zc = Zt + (two ? min(c0, n4 - 2) : min(c0, n4 - 1));
        switch (n4) {   // (uniform)
          case 48: coop_pair_dist<FIN, 52, false>(zr, zc, accA, accB); break;
          case 64: coop_pair_dist<FIN, 68, false>(zr, zc, accA, accB); break;
          case 80: coop_pair_dist<FIN, 84, true>(zr, zc, accA, accB); break;
          case 96: coop_pair_dist<FIN, 100, true>(zr, zc, accA, accB); break;
          case 112: coop_pair_dist<FIN, 116, true>(zr, zc, accA, accB); break;
          case 128: coop_pair_dist<FIN, 132, true>(zr, zc, accA, accB); break;
          default:   // (no other size class reaches a member kernel of up to 128 nodes; kept for safety: run-time stride)
            if (two) coop_pair_dist<FIN, 0, true>(zr, zc, accA, accB, ldz);
            else coop_pair_dist<FIN, 0, false>(zr, zc, accA, accB, ldz);
        }
        AGCN_STAMP(0, 10);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int i = ia + rr;
          const f32x2 acc = rr ? accB : accA;
          float sv[2] = {0.f, 0.f}, rv[2] = {0.f, 0.f};
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const int j = c0 + cc;
            if (i < n && j < n && (cc == 0 || two)) {
              rv[cc] = (i == j) ? 0.f : fast_sqrt(cc ? acc.y : acc.x);
              sv[cc] = (i == j) ? 0.f : fast_exp_neg(rv[cc]);
            }
          }
          rk2[rr] = (f32x2){rv[0], rv[1]};
          sk2[rr] = (f32x2){sv[0], sv[1]};
          ds[rr] = sv[0] + sv[1];
        }
      }
      // S rows of the pair -> strip image (zero beyond the last node / the last row)
      if (two) {
        if (c0 < n4) {
          *reinterpret_cast<f32x2*>(St + (2 * wave) * ldt + c0) = sk2[0];
          *reinterpret_cast<f32x2*>(St + (2 * wave + 1) * ldt + c0) = sk2[1];
        }
      } else if (c0 < n4) {
        St[(2 * wave) * ldt + c0] = sk2[0].x;
        St[(2 * wave + 1) * ldt + c0] = sk2[1].x;
      }
      AGCN_STAMP(0, 12);
      const float da = wave_sum(ds[0]), db = wave_sum(ds[1]);      // (every lane holds both)
      AGCN_STAMP(0, 13);
      if (paper && lane < 2 && ia + lane < n) {
        // (published before any other store of this phase: the siblings' polls -- and this workgroup's own, which queue
        // behind its stores in the one in-order memory counter -- then see it a store round trip earlier)
        sdk = 1.0f / sqrtf((lane ? db : da) + FLT_TRUE_MIN);
        s_vec[ia + lane] = sdk;
        __hip_atomic_store(xch + ia + lane, coop_pack(sdk, wants[me]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  } else
  {
    const int i = i0 + wave;
    float dsum = 0.f;
    if (i < n) {   // (uniform)
      float acc[NCP] = {};
      const int j0 = min(lane, n - 1), j1 = min(lane + 64, n - 1);
      const float* zi = Zt + i;
      if constexpr (NCP > 2) {   // (wide members: n > 128)
        int jc[NCP];
#pragma unroll
        for (int cp = 0; cp < NCP; ++cp) jc[cp] = min(lane + 64 * cp, n - 1);
#pragma unroll 4
        for (int f = 0; f < FIN; ++f) {
          const float zv = zi[f * ldz];
#pragma unroll
          for (int cp = 0; cp < NCP; ++cp) {
            const float dv = zv - Zt[f * ldz + jc[cp]];
            acc[cp] = fmaf(dv, dv, acc[cp]);
          }
        }
      } else if (n > 64) {   // (uniform)
#pragma unroll 4
        for (int f = 0; f < FIN; ++f) {
          const float zv = zi[f * ldz];
          const float d0 = zv - Zt[f * ldz + j0], d1 = zv - Zt[f * ldz + j1];
          acc[0] = fmaf(d0, d0, acc[0]);
          acc[1] = fmaf(d1, d1, acc[1]);
        }
      } else {
#pragma unroll 5
        for (int f = 0; f < FIN; ++f) {
          const float d0 = zi[f * ldz] - Zt[f * ldz + j0];
          acc[0] = fmaf(d0, d0, acc[0]);
        }
      }
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp) {
        const int j = lane + 64 * cp;
        if (j < n4) {
          float sv = 0.f;
          if (j < n) {
            rkeep[cp] = (i == j) ? 0.f : fast_sqrt(acc[cp]);
            sv = (i == j) ? 0.f : fast_exp_neg(rkeep[cp]);
          }
          skeep[cp] = sv;
          St[wave * ldt + j] = sv;
          dsum += sv;
        }
      }
    } else {
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp)
        if (lane + 64 * cp < n4) St[wave * ldt + lane + 64 * cp] = 0.f;
    }
    dsum = wave_sum(dsum);
    if (paper && lane == 0 && i < n) {
      const float sd = 1.0f / sqrtf(dsum + FLT_TRUE_MIN);
      s_vec[i] = sd;
      sd_keep = sd;
      // (published before any other store of this phase: the siblings' polls -- and this workgroup's own, which queue
      // behind its stores in the one in-order memory counter -- then see it a store round trip earlier)
      __hip_atomic_store(xch + i, coop_pack(sd, wants[me]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  AGCN_STAMP(0, 7);
  if (tid < n4 && (!paper || tid >= n)) s_vec[tid] = 0.f;
  // While the first waves (tid < n: at most three) wait for the siblings' words -- one hand-off latency, ~2.5 us -- the LAST NOB
  // waves, which never poll, do the part of the filter that depends on nothing but this strip's X rows: they request both terms'
  // W fragments and run T_0 = X_0[I] W_0; P5 then only adds X_1[I] W_1 onto the same accumulator (the same instruction sequence as
  // before, split in time: the same bits).  Before: the fragments' round trip and 19 of the 38 chained MFMAs sat in P5.
  // (layers of up to 80 input features: at 128 the 32 fragment registers held across two phases spill)
  constexpr bool EARLY = FIN <= 80;
  constexpr int P5W0 = NW - NOB;          // first wave of the filter tiles (tile ob = wave - P5W0)
  float w1[NKS];
  f32x4 yacc = (f32x4){0.f, 0.f, 0.f, 0.f};
  float ybias = 0.f;
  if (EARLY && wave >= P5W0) {
    const int ob = wave - P5W0;
    const float* wq = a.weight + q * 2 * FOUT + r + ob * 16;
    float w0[NKS];
#pragma unroll
    for (int u = 0; u < NKS; ++u) {
      const int off = (4 * u + 3 < FIN) ? 4 * u * 2 * FOUT : (min(4 * u + q, FIN - 1) - q) * 2 * FOUT;
      w0[u] = wq[off];
      w1[u] = wq[off + FOUT];
    }
    ybias = a.bias[ob * 16 + r];
    const float* A0 = Ximg + (me * 16 + r) * LDX + q;
    yacc = mfma_chain<NKS>(A0, w0, yacc);
  }
  if (paper && tid < n && (tid < i0 || tid >= i0 + 16)) {
    bool got;
    const uint64_t w = coop_wait_word(a.sv_xch + COOP_STATUS_WORD, a.dev_status, a.coop_ticks, xch + tid, wants[tid >> 4], &got);
    s_vec[tid] = got ? __uint_as_float((uint32_t)w) : __builtin_nanf("");
  }
  AGCN_STAMP(0, 8);
  if constexpr (ROWPAIR) {   // the strip's rows of res_W and r, and s: out behind the exchange
    if (wave < 8) {
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int i = ia + rr;
        if (i < n) {
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const int j = c0 + cc;
            if (j < n && (cc == 0 || two)) {
              const uint32_t o = toff + (uint32_t)(i * n + j);
              AGCN_ST(1, &a.res_W[o], cc ? sk2[rr].y : sk2[rr].x);
              if (paper) AGCN_ST(2, &a.sv_r[o], cc ? rk2[rr].y : rk2[rr].x);
            }
          }
        }
      }
      if (lane < 2 && ia + lane < n) a.sv_s[row0 + ia + lane] = paper ? sdk : 0.f;
    }
  } else {
    const int i = i0 + wave;
    if (i < n) {
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp) {
        const int j = lane + 64 * cp;
        if (j < n) {
          const uint32_t o = toff + (uint32_t)(i * n + j);
          AGCN_ST(1, &a.res_W[o], skeep[cp]);
          if (paper) AGCN_ST(2, &a.sv_r[o], rkeep[cp]);
        }
      }
      if (lane == 0) a.sv_s[row0 + i] = paper ? sd_keep : 0.f;
    }
  }
  AGCN_STAMP(0, 9);
  lds_barrier();

  AGCN_STAMP(0, 3);
  // ---- P3: res_L[I, :], L[I, :] = lrelu(R) + L_orig ----------------------------------------------------------------
  float ssR = 1.f, scR = 1.f, ssL = 0.f, scL = 1.f;
  {
    const int l0ld = a.L0_ld > 0 ? a.L0_ld : n;
    const float* L0d = !a.L0_dense ? nullptr
                                   : (a.L0_ld > 0 ? a.L0_dense + (int64_t)b * a.L0_ld * a.L0_ld : a.L0_dense + toff);
    const int i = i0 + wave;
    if constexpr (RESLAP) {
      // R rows of the strip (kept in registers), ||R||_F^2 of the whole graph -> scR   (graphconv_reslap.py:185)
      float Rk[NCP] = {};
      float part = 0.f;
      if (i < n) {
        const float si = s_vec[i];
#pragma unroll
        for (int cp = 0; cp < NCP; ++cp) {
          const int j = lane + 64 * cp;
          if (j < n) {
            float R = (i == j) ? 1.f : 0.f;
            if (paper) R = R - (si * St[wave * ldt + j]) * s_vec[j];
            Rk[cp] = R;
            part += R * R;
          }
        }
      }
      ssR = paper ? coop_graph_sum(a, part, smem, coop_gs, xsc, me, ntm, wants, tid) : (float)n;   // (reference mode: R = I)
      scR = clip_scale(ssR, 1.0f);
      // R2 = lrelu(cn(R)) -> res_L; L_pre = R2 + L_orig + beta L_prev (:188-190) -> strip image (dense L_orig here, CSR below)
      const bool has_prev = a.L_prev != nullptr;
      const float beta = has_prev ? a.beta[0] : 0.f;
      if (i < n) {
#pragma unroll
        for (int cp = 0; cp < NCP; ++cp) {
          const int j = lane + 64 * cp;
          if (j < n) {
            const float R2 = lrelu(Rk[cp] * scR, alpha);
            AGCN_ST(1, &a.res_L[toff + (uint32_t)(i * n + j)], R2);
            float v = R2;
            if (L0d) v += L0d[(int64_t)i * l0ld + j];
            if (has_prev) v = v + lprv[cp] * beta;
            St[wave * ldt + j] = v;
          }
        }
      }
      lds_barrier();
      if (a.L0_rowptr && tid < 16 && i0 + tid < n) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
          if (csr_p0 + u < csr_p1) lds_add(&St[tid * ldt + csr_c[u]], csr_v[u]);
        for (int p = csr_p0 + 6; p < csr_p1; ++p) lds_add(&St[tid * ldt + a.L0_col[p]], a.L0_val[p]);
      }
      lds_barrier();
      // ||L_pre||_F^2 of the whole graph -> scL; second clip + leaky-ReLU (:194-198): L_pre out (reserve), La -> strip + 4th output
      float part2 = 0.f;
      float vk[NCP] = {};
      if (i < n) {
#pragma unroll
        for (int cp = 0; cp < NCP; ++cp) {
          const int j = lane + 64 * cp;
          if (j < n) {
            vk[cp] = St[wave * ldt + j];
            part2 += vk[cp] * vk[cp];
          }
        }
      }
      ssL = coop_graph_sum(a, part2, smem, coop_gs, xsc + COOP_MAX_MEMBERS, me, ntm, wants, tid);
      scL = clip_scale(ssL, 1.0f);
      if (i < n) {
#pragma unroll
        for (int cp = 0; cp < NCP; ++cp) {
          const int j = lane + 64 * cp;
          if (j < n) {
            const float La = lrelu(vk[cp] * scL, alpha);
            const uint32_t o = toff + (uint32_t)(i * n + j);
            AGCN_ST(2, &a.sv_Lpre[o], vk[cp]);
            AGCN_ST(1, &a.L_sum[o], La);
            St[wave * ldt + j] = La;
          }
        }
      }
      lds_barrier();
    } else {
    if (i < n) {
      const float si = s_vec[i];
#pragma unroll
      for (int cp = 0; cp < NCP; ++cp) {
        const int j = lane + 64 * cp;
        if (j < n) {
          const float S = St[wave * ldt + j];
          float R = (i == j) ? 1.f : 0.f;
          if (paper) R = R - (si * S) * s_vec[j];
          const float R2 = lrelu(R, alpha);
          AGCN_ST(1, &a.res_L[toff + (uint32_t)(i * n + j)], R2);
          St[wave * ldt + j] = R2 + (L0d ? L0d[(int64_t)i * l0ld + j] : 0.f);
        }
      }
    }
    }
  }
  if constexpr (!RESLAP) {
  lds_barrier();
  if (a.L0_rowptr && tid < 16 && i0 + tid < n) {
#pragma unroll
    for (int u = 0; u < 6; ++u)
      if (csr_p0 + u < csr_p1) lds_add(&St[tid * ldt + csr_c[u]], csr_v[u]);
    for (int p = csr_p0 + 6; p < csr_p1; ++p) lds_add(&St[tid * ldt + a.L0_col[p]], a.L0_val[p]);
  }
  lds_barrier();
  if (a.L_sum) {
    const int i = i0 + wave;
    if (i < n)
      for (int j = lane; j < n; j += 64) a.L_sum[toff + (uint32_t)(i * n + j)] = St[wave * ldt + j];
  }
  }

  AGCN_STAMP(0, 4);
  // ---- P4: X_1[I] = L[I, :] X_0 ------------------------------------------------------------------------------------
  if (wave < NCB) {
    const int cb = wave;
    const float* Ap = St + r * ldt + q;
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
    const int c = cb * 16 + r;
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) X1s[(4 * q + i4) * LDX + c] = acc[i4];
  }
  lds_barrier();

  AGCN_STAMP(0, 5);
  // ---- P5: Y[I] = act(X_0[I] W_0 + X_1[I] W_1 + b), then the stores -------------------------------------------------
  if (wave >= P5W0) {      // (EARLY: T_0 and the W_1 fragments come from the exchange's waiting time, above)
    const int ob = wave - P5W0;
    float bias = ybias;
    const float* A1 = X1s + r * LDX + q;
    f32x4 acc = yacc;
    if constexpr (!EARLY) {
      const float* wq = a.weight + q * 2 * FOUT + r + ob * 16;
      float w0[NKS];
#pragma unroll
      for (int u = 0; u < NKS; ++u) {
        const int off = (4 * u + 3 < FIN) ? 4 * u * 2 * FOUT : (min(4 * u + q, FIN - 1) - q) * 2 * FOUT;
        w0[u] = wq[off];
        w1[u] = wq[off + FOUT];
      }
      bias = a.bias[ob * 16 + r];
      const float* A0 = Ximg + (me * 16 + r) * LDX + q;
      acc = mfma_chain<NKS>(A0, w0, acc);
    }
    acc = mfma_chain<NKS>(A1, w1, acc);
    float* Yg = a.Y + (int64_t)row0 * a.ldy;
    const int c = ob * 16 + r;
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const int i = i0 + 4 * q + i4;
      if (i < n) {
        float v = acc[i4] + bias;
        if (relu) v = fmaxf(v, 0.f);
        if (a.feat_bf16) bf16_store(a.Y, (int64_t)(row0 + i) * a.ldy + c, v);
        else AGCN_ST(4, &Yg[(int64_t)i * a.ldy + c], v);
      }
    }
  }
  if (i0 + wave < n) {   // X_1 rows of the strip (reserve: the parameter gradients read them)
    float* Xo = a.sv_X + (int64_t)(row0 + i0 + wave) * FIN;
#pragma unroll
    for (int cp = 0; cp < CP; ++cp) {
      const int c = lane + 64 * cp;
      if (c < FIN) AGCN_ST(2, &Xo[c], X1s[wave * LDX + c]);
    }
    if (a.feat_bf16) {   // ... and, bf16 features, the widened X rows of the strip (spec_kernels.h)
      float* X0o = a.sv_X0 + (int64_t)(row0 + i0 + wave) * FIN;
#pragma unroll
      for (int cp = 0; cp < CP; ++cp) {
        const int c = lane + 64 * cp;
        if (c < FIN) X0o[c] = Ximg[(i0 + wave) * LDX + c];
      }
    }
  }
  if (i0 == 0) {
    const uint32_t npad = srec.npad;
    if (tid < (int)npad) {   // alignment floats behind the tile
      a.res_W[toff + (uint32_t)(n * n) + tid] = 0.f;
      a.res_L[toff + (uint32_t)(n * n) + tid] = 0.f;
      if (a.L_sum) a.L_sum[toff + (uint32_t)(n * n) + tid] = 0.f;
    }
    if (tid == 0) {
      float* sc = a.sv_scal + 4 * b;
      if constexpr (RESLAP) { sc[0] = ssR; sc[1] = scR; sc[2] = ssL; sc[3] = scL; }
      else { sc[0] = 1.f; sc[1] = 1.f; sc[2] = 0.f; sc[3] = 1.f; }
    }
  }
  if ((paper || RESLAP) && tid == 0) __hip_atomic_store(xep + me, (uint64_t)wants[me], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  AGCN_STAMP(0, 6);
}
#endif

}  // namespace spec
}  // namespace agcn

#endif  // AGCN_SPEC_BACKWARD_COOP_H_
