// Parameter gradients as TRAILING WORKGROUPS of the specialised backward launch (spec_backward.hip).
//
// Until round 4 the parameter gradients of the headline layer were a kernel of their own behind the backward
// (param_grads.hip: row splits, a partial tile written through to memory, an arrival counter, the last arriver's sum --
// 10.2 us, a chain of four memory round trips, for 0.14 GFLOP).  Here they ride at the END OF THE BACKWARD GRID:
//
//   grid = [members][one workgroup per graph][pairs of small graphs][TAIL: one workgroup per 16 x 16 output tile]
//
// A 256-graph launch fills the chip with one round of workgroups, and its smallest graphs are done after ~10 of the launch's
// ~18 us: tail workgroups are dispatched LAST, so they start on the CUs the small graphs free, and by the time a tail workgroup
// runs every producer workgroup has been dispatched (workgroups start in index order) -- waiting for them cannot deadlock.
//   * dweight / dbias tiles need nothing the backward computes (X_0, X_1, dY, Y are final before it starts): they walk all
//     rows straight away, in the shadow of the large graphs.
//   * dM_L tiles need dZ, which the producers write.  Every producer UNIT (a plain graph, one graph of a pair, one member slab)
//     stores its dZ rows with agent-scope stores and then publishes flag[unit] = generation + 1.
// The units are walked in ONE FIXED ORDER that follows the expected completion order -- smallest graph first (the schedule
// backwards), the member slabs last -- and the walk's 4-row k-steps are cut into S contiguous SPLITS; a tail workgroup is
// (split, 16 x 32 tile), 16 waves, k-step g of the split -> wave g mod 16: about ten k-steps per wave, ONE round of loads in
// flight.  (The first version had no splits: one workgroup per tile walked all ~1250 k-steps, 78 dependent k-steps per wave at
// a memory round trip per eight -- the backward went from 18.4 + 10.2 to 80.6 us.)  A split waits for the flags of ITS units
// only, so the early splits are done while the large graphs still run; split S - 1 of a tile -- the largest graphs, which
// finish last anyway -- is the tile's REDUCER: it first fetches the other splits' partial tiles (handed over with agent-scope
// stores + a flag per (split, tile)), then waits for its own producers, and adds partials in split order and its own sum last:
// the order is fixed by the plan, whatever the timing.  Bit-reproducible run to run and replay to replay.
// Generation protocol (the flags must not be left over from the previous launch / HIP-graph replay on the same reserve):
// word TAIL_GEN is read by every workgroup when it starts, and each says so with a plain agent-scope store into a word of its own
// (TAIL_STARTED + blockIdx); one designated tail workgroup bumps TAIL_GEN once it has seen all of those -- i.e. after every
// workgroup of this launch has read it.  (No read-modify-write atomics: an exit counter on one address serialised ~600
// device-scope atomics, +40 us.)  All of it lives in the reserve header (zero when a reserve is first used: include/agcn.h).
#ifndef AGCN_SPEC_TAIL_H_
#define AGCN_SPEC_TAIL_H_

#include "spec_backward_coop.h"

namespace agcn {
namespace spec {

#if defined(__HIPCC__)
// One tail workgroup = (row split s, 16 x 32 output tile); 16 waves.  Grid order: split-major (the early splits -- the smallest
// graphs -- first).  Split S - 1 of a tile is its REDUCER: it adds the other splits' partial tiles in split order and its own
// accumulators last, and writes the result.
template <int FIN, int FOUT>
__device__ __forceinline__ void tail_param_grads(const SgcArgs& a, const int wg, float* smem) {
  constexpr int K = 2, NWV = 16, TE = TAIL_TE;
  constexpr int tfW = (FIN * K + 15) / 16, tcW = (FOUT + 31) / 32, tfM = (FIN + 15) / 16, tcM = (FIN + 31) / 32;
  constexpr int NTW = tfW * tcW;
  static_assert(tfM * tcM + NTW <= TAIL_MAX_TILES, "tail tiles");
  const int tid = (int)threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = uni(tid >> 6);
  const uint32_t gen = tail_load(a.tail + TAIL_GEN);
  if (tid == 0) tail_started(a, gen);
  if (a.tail_debug & 1) {
    if (wg == (a.tail_splits - 1) * a.tail_tiles_per_split && wave == 0) tail_bump(a, gen, lane);
    return;
  }
  const int tiles = a.tail_tiles_per_split, S = a.tail_splits;
  const int split = wg / tiles, tile = wg - split * tiles;
  const bool isW = tile < NTW;
  const int rem = isW ? tile : tile - NTW;
  const int tf = isW ? tfW : tfM;
  const int itc = rem / tf, itf = rem - itc * tf;
  const int f0 = itf * 16, c0 = itc * 32;
  const int Fa = isW ? FIN * K : FIN, Fb = isW ? FOUT : FIN;
  // A operand: column j = f0 + r of [X_0 | X_1] interleaved (j = f * K + k) or of X;  B operand: columns c0 + r, c0 + 16 + r
  const int j = min(f0 + r, Fa - 1);
  const bool jv = (f0 + r) < Fa;
  const float* Ap;
  int64_t lda;
  if (isW) {
    const int f = j >> 1;
    if ((j & 1) == 0) { Ap = a.X + f; lda = a.ldx; } else { Ap = a.sv_X + f; lda = FIN; }
  } else { Ap = a.X + j; lda = a.ldx; }
  const int cA = min(c0 + r, Fb - 1), cB = min(c0 + 16 + r, Fb - 1);
  const bool cAv = (c0 + r) < Fb, cBv = (c0 + 16 + r) < Fb;
  const bool relu = a.activation == AGCN_ACT_RELU;
  const bool with_bias = isW && itf == 0;
  const bool reducer = split == S - 1;
  const unsigned long long t_start = wall_clock64();
  bool timed_out = false;
  float* wsum = smem;                          // [16][512]
  float* bs = smem + NWV * TE;                 // [16][32]
  int* any_timeout = reinterpret_cast<int*>(smem + NWV * TE + NWV * 32);

  // ---- the reducer fetches the other splits' partial tiles first: their producers are the smaller graphs -----------------
  float part[TAIL_MAX_SPLITS - 1];     // thread tid: element tid of the partial tile (tid < 512) or bias partial tid - 512
#pragma unroll
  for (int x = 0; x < TAIL_MAX_SPLITS - 1; ++x) part[x] = 0.f;
  if (reducer && S > 1 && !(a.tail_debug & 2)) {
    const uint32_t* f2 = a.tail + TAIL_FLAGS2 + tile;      // flag of (split x, tile): f2[x * tiles]
    for (;;) {
      const uint32_t fl = (lane < S - 1) ? tail_load(f2 + lane * tiles) : gen + 1u;
      if (__ballot(fl == gen + 1u) == ~0ull) break;
      if (wall_clock64() - t_start >= (unsigned long long)a.coop_ticks) { timed_out = true; break; }
      __builtin_amdgcn_s_sleep(4);
    }
    if (!timed_out && tid < TE + 32) {
#pragma unroll
      for (int x = 0; x < TAIL_MAX_SPLITS - 1; ++x)
        if (x < S - 1) {
          const float* pp = a.tail_part + ((int64_t)x * tiles + tile) * TAIL_PSTRIDE;
          if (tid < TE || with_bias) part[x] = tail_load_f(pp + tid);
        }
    }
  }

  // ---- the walk: units 0 .. G-1 = the graphs of schedule positions first + G - 1 down to first (smallest first), then the
  // member slabs; lane l holds the units l, l + 64, ... (TAIL_NWIN windows) and the inclusive prefix of their k-steps ----------
  const int G = a.tail_graphs, P = G + a.coop_slabs;
  int urow[TAIL_NWIN], un[TAIL_NWIN], pe[TAIL_NWIN];
  int carry = 0;
#pragma unroll
  for (int w = 0; w < TAIL_NWIN; ++w) {
    const int u = 64 * w + lane;
    urow[w] = 0; un[w] = 0;
    if (u < G) {
      const int sp = a.tail_first + (G - 1 - u);
      const AgcnGraphDesc* gd = a.graph_desc + sp;
      un[w] = gd->n; urow[w] = gd->row0;
    } else if (u < P) {
      const int sl = u - G;
      const int b = a.slab_graph[sl], i0 = a.slab_row[sl];
      urow[w] = a.row_off[b] + i0;
      un[w] = min(16, a.n[b] - i0);
    }
  }
#pragma unroll
  for (int w = 0; w < TAIL_NWIN; ++w) {
    int v = (un[w] + 3) >> 2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(v, o, 64);
      if (lane >= o) v += t;
    }
    pe[w] = v + carry;
    carry = __shfl(pe[w], 63, 64);
  }
  const int T = carry;                                   // k-steps of the whole walk (== a.tail_ksteps)
  const int K0 = (int)(((int64_t)T * split) / S), K1 = (int)(((int64_t)T * (split + 1)) / S);
  // unit of walk k-step g: the number of units whose inclusive prefix is <= g
  auto unit_of = [&](int g) {
    int cnt = 0;
#pragma unroll
    for (int w = 0; w < TAIL_NWIN; ++w) cnt += __builtin_popcountll(__ballot(pe[w] <= g));
    return cnt;
  };
  // ---- wait for this split's producers (dM_L tiles only) -------------------------------------------------------------------
  if (!isW && K1 > K0 && !timed_out && !(a.tail_debug & 2)) {
    const int u0 = unit_of(K0), u1 = unit_of(K1 - 1);     // units u0 .. u1
    for (;;) {
      bool ok = true;
#pragma unroll
      for (int w = 0; w < TAIL_NWIN; ++w) {
        const int u = 64 * w + lane;
        // flag slot of unit u: graphs behind the member slabs, in schedule order (the walk runs the schedule backwards)
        const int slot = u < G ? a.coop_slabs + (G - 1 - u) : u - G;
        if (u >= u0 && u <= u1) ok = ok && (tail_load(a.tail + TAIL_FLAGS + slot) == gen + 1u);
      }
      if (__ballot(ok) == ~0ull) break;
      if (wall_clock64() - t_start >= (unsigned long long)a.coop_ticks) { timed_out = true; break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  // ---- this wave's k-steps of the split: K0 + wave, + 16, ...; TAIL_KR per round, all loads of a round in flight --------------
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  float bsum0 = 0.f, bsum1 = 0.f;
  if (!timed_out && !(a.tail_debug & 4)) {
#pragma unroll 1
    for (int g0 = K0 + wave; g0 < K1; g0 += 16 * TAIL_KR) {
      float av[TAIL_KR], b0[TAIL_KR], b1[TAIL_KR], y0[TAIL_KR], y1[TAIL_KR];
      bool rv[TAIL_KR];
#pragma unroll
      for (int x = 0; x < TAIL_KR; ++x) {
        const int g = g0 + 16 * x;
        const bool in = g < K1;
        const int gc = in ? g : g0;
        const int ui = unit_of(gc);
        int ur = 0, nn = 0, pend = 0;
#pragma unroll
        for (int w = 0; w < TAIL_NWIN; ++w)
          if ((ui >> 6) == w) { ur = __shfl(urow[w], ui & 63, 64); nn = __shfl(un[w], ui & 63, 64); pend = __shfl(pe[w], ui & 63, 64); }
        const int jstep = gc - (pend - ((nn + 3) >> 2));          // k-step inside the unit
        const int row = ur + 4 * jstep + q;
        rv[x] = in && (4 * jstep + q) < nn;
        const int rowc = min(row, a.rows - 1);
        av[x] = Ap[(int64_t)rowc * lda];
        if (isW) {
          b0[x] = a.dY[(int64_t)rowc * a.ldy + cA]; b1[x] = a.dY[(int64_t)rowc * a.ldy + cB];
          y0[x] = a.Y[(int64_t)rowc * a.ldy + cA]; y1[x] = a.Y[(int64_t)rowc * a.ldy + cB];
        } else {
          b0[x] = tail_load_f(a.ws_dZ + (int64_t)rowc * FIN + cA); b1[x] = tail_load_f(a.ws_dZ + (int64_t)rowc * FIN + cB);
          y0[x] = y1[x] = 1.f;
        }
      }
#pragma unroll
      for (int x = 0; x < TAIL_KR; ++x) {
        const float xa = (rv[x] && jv) ? av[x] : 0.f;
        float xb0 = (rv[x] && cAv) ? b0[x] : 0.f, xb1 = (rv[x] && cBv) ? b1[x] : 0.f;
        if (isW && relu) { if (!(y0[x] > 0.f)) xb0 = 0.f; if (!(y1[x] > 0.f)) xb1 = 0.f; }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, xb0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, xb1, acc1, 0, 0, 0);
        bsum0 += xb0; bsum1 += xb1;
      }
    }
  }
  // ---- the split's tile: the 16 waves' accumulators in wave order ------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    wsum[wave * TE + (4 * q + i) * 32 + r] = acc0[i];
    wsum[wave * TE + (4 * q + i) * 32 + 16 + r] = acc1[i];
  }
  if (with_bias) {
    float u0 = bsum0, u1 = bsum1;
    u0 += __shfl_xor(u0, 16, 64); u0 += __shfl_xor(u0, 32, 64);
    u1 += __shfl_xor(u1, 16, 64); u1 += __shfl_xor(u1, 32, 64);
    if (q == 0) { bs[wave * 32 + r] = u0; bs[wave * 32 + 16 + r] = u1; }
  }
  if (tid == 0) *any_timeout = 0;
  __syncthreads();
  if (timed_out && lane == 0) *any_timeout = 1;
  __syncthreads();
  const bool bad = *any_timeout != 0;
  float v = 0.f;
  if (tid < TE) {
#pragma unroll
    for (int w = 0; w < NWV; ++w) v += wsum[w * TE + tid];
  } else if (tid < TE + 32 && with_bias) {
#pragma unroll
    for (int w = 0; w < NWV; ++w) v += bs[w * 32 + (tid - TE)];
  }
  if (!reducer) {
    // hand-off: the partial tile (+ bias partials) with agent-scope stores, then the flag of (split, tile)
    float* pp = a.tail_part + ((int64_t)split * tiles + tile) * TAIL_PSTRIDE;
    if (tid < TE || (tid < TE + 32 && with_bias)) tail_store_f(pp + tid, bad ? __builtin_nanf("") : v);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) tail_store(a.tail + TAIL_FLAGS2 + split * tiles + tile, gen + 1u);
  } else {
    float t = 0.f;
#pragma unroll
    for (int x = 0; x < TAIL_MAX_SPLITS - 1; ++x)
      if (x < S - 1) t += part[x];
    t += v;
    if (bad) t = __builtin_nanf("");
    if (tid < TE) {
      const int ff = f0 + (tid >> 5), cc = c0 + (tid & 31);
      if (ff < Fa && cc < Fb) {
        if (isW) a.g_dweight[(int64_t)ff * FOUT + cc] = t;
        else a.g_dM[(int64_t)ff * FIN + cc] = t;
      }
    } else if (tid < TE + 32 && with_bias) {
      const int cc = c0 + (tid - TE);
      if (cc < FOUT) a.g_dbias[cc] = t;
    }
    // dalpha: the per-unit partials, by the reducer of the LAST tile (a dM_L tile: every split of it has waited for its
    // producers, and this workgroup has seen every split's flag), one wave, index order
    if (tile == tiles - 1 && wave == 0 && a.g_dalpha != nullptr) {
      const int npg = a.B + a.coop_slabs;
      float tsum = 0.f;
      for (int b0i = 0; b0i < npg; b0i += 64 * 8) {
        float pv[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const int b = b0i + 64 * x + lane;
          pv[x] = b < npg ? tail_load_f(a.ws_pg + 2 * b) : 0.f;
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) tsum += pv[x];
      }
      tsum = wave_sum(tsum);
      if (lane == 0) a.g_dalpha[0] = bad ? __builtin_nanf("") : tsum;
    }
  }
  if (bad && tid == 0) {
    __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(a.sv_xch + COOP_STATUS_WORD), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.dev_status != nullptr) __hip_atomic_fetch_add(a.dev_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the generation word moves on once every workgroup of the launch has read it: by the reducer of tile 0 (a dweight tile)
  if (reducer && tile == 0 && wave == 0) tail_bump(a, gen, lane);
}
#endif  // __HIPCC__

}  // namespace spec
}  // namespace agcn

#endif  // AGCN_SPEC_TAIL_H_
