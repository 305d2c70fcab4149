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
//     stores its dZ rows with agent-scope stores and then publishes flag[unit] = generation + 1.  A tile walks the units in ONE
//     FIXED ORDER chosen to follow the expected completion order -- smallest graph first (the schedule backwards), the member
//     slabs last -- polling the flags 64 at a time and consuming the ready prefix, so that when the last producer finishes only
//     its own rows are left: flag round trip + one load round trip + the 16-wave sum.
// No row split, so no cross-workgroup hand-off: a tile's 16 waves share the 4-row k-steps of the walk round-robin (k-step g of
// the walk -> wave g mod 16: fixed by the plan, whatever the timing), add their accumulators in wave order in LDS and write the
// result.  Bit-reproducible run to run and replay to replay.
// Generation protocol (the flags must not be left over from the previous launch / HIP-graph replay on the same reserve):
// word TAIL_GEN is read by every workgroup when it starts; every workgroup counts itself in TAIL_DONE when it ends, and the
// last one resets the counter and bumps TAIL_GEN -- i.e. after every workgroup of this launch has read it.  All of it lives in
// the reserve header (zero when a reserve is first used: include/agcn.h).
#ifndef AGCN_SPEC_TAIL_H_
#define AGCN_SPEC_TAIL_H_

#include "spec_backward_coop.h"

namespace agcn {
namespace spec {

#if defined(__HIPCC__)
// One tail workgroup = one 16 x 16 output tile; 16 waves.
template <int FIN, int FOUT>
__device__ __forceinline__ void tail_param_grads(const SgcArgs& a, const int tile, float* smem) {
  constexpr int K = 2, NWV = 16;
  constexpr int tfW = (FIN * K + TAIL_T - 1) / TAIL_T, tcW = (FOUT + TAIL_T - 1) / TAIL_T, tfM = (FIN + TAIL_T - 1) / TAIL_T;
  const int tid = (int)threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
  const int wave = uni(tid >> 6);
  const uint32_t gen = tail_load(a.tail + TAIL_GEN);
  const bool isW = tile < tfW * tcW;
  const int rem = isW ? tile : tile - tfW * tcW;
  const int tf = isW ? tfW : tfM;
  const int itc = rem / tf, itf = rem - itc * tf;
  const int f0 = itf * TAIL_T, c0 = itc * TAIL_T;
  const int Fa = isW ? FIN * K : FIN, Fb = isW ? FOUT : FIN;
  // A operand: column j = f0 + r of [X_0 | X_1] interleaved (j = f * K + k) or of X;  B operand: column c0 + r of Gp or dZ
  const int j = min(f0 + r, Fa - 1);
  const bool jv = (f0 + r) < Fa;
  const float* Ap;
  int64_t lda;
  if (isW) {
    const int f = j >> 1;
    if ((j & 1) == 0) { Ap = a.X + f; lda = a.ldx; } else { Ap = a.sv_X + f; lda = FIN; }
  } else { Ap = a.X + j; lda = a.ldx; }
  const int c = min(c0 + r, Fb - 1);
  const bool cv = (c0 + r) < Fb;
  const bool relu = a.activation == AGCN_ACT_RELU;
  const bool with_bias = isW && itf == 0;
  const bool last_tile = tile == (int)tail_tiles(FIN, FOUT, K, a.mode == AGCN_MODE_PAPER) - 1;

  // the walk: positions 0 .. G-1 = the graphs of schedule positions first + G - 1 down to first (smallest first), then the
  // member slabs 0 .. coop_slabs - 1
  const int G = a.tail_graphs, P = G + a.coop_slabs;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  int64_t kbase = 0;                 // k-steps of the walk in front of the current window
  bool timed_out = false;
  const unsigned long long t_start = wall_clock64();
#pragma unroll 1
  for (int pos = 0; pos < P; pos += 64) {
    // lane i: unit pos + i
    const int u = pos + lane;
    int urow = 0, un = 0, slot = 0;
    if (u < G) {
      const int sp = a.tail_first + (G - 1 - u);
      const AgcnGraphDesc* gd = a.graph_desc + sp;
      un = gd->n; urow = gd->row0;
      slot = a.coop_slabs + (sp - a.tail_first);
    } else if (u < P) {
      const int sl = u - G;
      const int b = a.slab_graph[sl], i0 = a.slab_row[sl];
      urow = a.row_off[b] + i0;
      un = min(16, a.n[b] - i0);
      slot = sl;
    }
    const int ks = (un + 3) >> 2;
    // inclusive prefix of the k-steps over the window's lanes
    int pe = ks;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(pe, o, 64);
      if (lane >= o) pe += v;
    }
    const int pb = pe - ks;
    const int Tw = __shfl(pe, 63, 64);
    // this wave's k-steps of the window: t with (kbase + t) % 16 == wave
    int t = (int)(((int64_t)wave - kbase) & 15);
    int nready = isW ? 64 : 0;
    for (;;) {
      if (!isW) {
        const uint32_t fl = (u < P) ? tail_load(a.tail + TAIL_FLAGS + slot) : gen + 1u;
        const unsigned long long ok = __ballot(fl == gen + 1u);
        const int lead = (ok == ~0ull) ? 64 : __builtin_ctzll(~ok);
        if (lead == nready) {
          if (wall_clock64() - t_start >= (unsigned long long)a.coop_ticks + 1000000ull) { timed_out = true; break; }
          __builtin_amdgcn_s_sleep(4);
          continue;
        }
        nready = lead;
      }
      const int Tr = nready == 64 ? Tw : __shfl(pb, nready, 64);   // k-steps of the ready prefix
      // eight k-steps per round: all loads requested, then the products
#pragma unroll 1
      while (t < Tr) {
        float av[8], bv[8], yv[8];
        bool rv[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const int tt = t + 16 * x;
          const bool in = tt < Tr;
          const int tc = in ? tt : t;
          const int ui = __builtin_popcountll(__ballot(pe <= tc));       // unit of k-step tc (window-local)
          const int row = __shfl(urow, ui, 64) + 4 * (tc - __shfl(pb, ui, 64)) + q;
          const bool valid = in && (4 * (tc - __shfl(pb, ui, 64)) + q) < __shfl(un, ui, 64);
          const int rowc = min(row, a.rows - 1);
          rv[x] = valid;
          av[x] = Ap[(int64_t)rowc * lda];
          if (isW) {
            bv[x] = a.dY[(int64_t)rowc * a.ldy + c];
            yv[x] = a.Y[(int64_t)rowc * a.ldy + c];
          } else {
            bv[x] = tail_load_f(a.ws_dZ + (int64_t)rowc * FIN + c);
            yv[x] = 1.f;
          }
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const float xa = (rv[x] && jv) ? av[x] : 0.f;
          float xb = (rv[x] && cv) ? bv[x] : 0.f;
          if (isW && relu && !(yv[x] > 0.f)) xb = 0.f;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, xb, acc, 0, 0, 0);
          bsum += xb;
        }
        t += 16 * min(8, (Tr - t + 15) >> 4);      // (only the k-steps that were inside the ready prefix)
      }
      if (nready == 64) break;
    }
    if (timed_out) break;
    kbase += Tw;
  }
  // ---- the tile: sum of the 16 waves' accumulators in wave order ---------------------------------------------------------
  float* wsum = smem;                         // [16][256]
  float* bs = smem + NWV * 256;               // [16][16]
#pragma unroll
  for (int i = 0; i < 4; ++i) wsum[wave * 256 + (4 * q + i) * 16 + r] = acc[i];
  if (with_bias) {
    float b = bsum;                           // column c0 + r: the four row groups q of the lane's k-steps
    b += __shfl_xor(b, 16, 64);
    b += __shfl_xor(b, 32, 64);
    if (q == 0) bs[wave * 16 + r] = b;
  }
  int* any_timeout = reinterpret_cast<int*>(smem + NWV * 256 + 256);
  if (tid == 0) *any_timeout = 0;
  __syncthreads();
  if (timed_out && lane == 0) *any_timeout = 1;
  __syncthreads();
  const bool bad = *any_timeout != 0;
  if (tid < 256) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) v += wsum[w * 256 + tid];
    if (bad) v = __builtin_nanf("");
    const int ff = f0 + (tid >> 4), cc = c0 + (tid & 15);
    if (ff < Fa && cc < Fb) {
      if (isW) a.g_dweight[(int64_t)ff * FOUT + cc] = v;
      else a.g_dM[(int64_t)ff * FIN + cc] = v;
    }
  } else if (tid < 272 && with_bias) {
    const int cc = tid - 256;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) v += bs[w * 16 + cc];
    if (bad) v = __builtin_nanf("");
    if (c0 + cc < FOUT) a.g_dbias[c0 + cc] = v;
  }
  if (bad && tid == 0) {
    __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(a.sv_xch + COOP_STATUS_WORD), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.dev_status != nullptr) __hip_atomic_fetch_add(a.dev_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // dalpha: the per-unit partials (agent-scope stores of the producers, in front of their flags), by the LAST tile -- a dM_L
  // tile in paper mode, which has seen every flag by now --, one wave, index order
  if (last_tile && wave == 0 && a.g_dalpha != nullptr) {
    const int npg = a.B + a.coop_slabs;
    float tsum = 0.f;
    for (int b0 = 0; b0 < npg; b0 += 64 * 8) {
      float v[8];
#pragma unroll
      for (int x = 0; x < 8; ++x) {
        const int b = b0 + 64 * x + lane;
        v[x] = b < npg ? tail_load_f(a.ws_pg + 2 * b) : 0.f;
      }
#pragma unroll
      for (int x = 0; x < 8; ++x) tsum += v[x];
    }
    tsum = wave_sum(tsum);
    if (lane == 0) a.g_dalpha[0] = bad ? __builtin_nanf("") : tsum;
  }
  if (tid == 0) tail_exit(a, gen);
}
#endif  // __HIPCC__

}  // namespace spec
}  // namespace agcn

#endif  // AGCN_SPEC_TAIL_H_
