// C ABI glue for the SGC_LL / SGC_LL_Reslap layer (include/agcn.h): argument validation,
// reserve/workspace carving, LDS-vs-global tile scheduling and kernel launches.
// No allocation, no synchronisation, everything on the caller's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <unordered_map>

#include "sgc_kernels.h"
#include "spec_backward_coop.h"
#include "spec_tail.h"
#include "spec_kernels.h"

namespace agcn {
namespace {
constexpr int MAX_DEVICES = 64;
std::atomic<int> g_cus_of[MAX_DEVICES];
std::atomic<uint32_t*> g_status_of[MAX_DEVICES];
}  // namespace
DeviceInfo current_device_info() {
  DeviceInfo d{256, nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < MAX_DEVICES) {
    int v = g_cus_of[dev].load(std::memory_order_relaxed);
    if (v == 0) {
      if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
      (void)hipGetLastError();
      g_cus_of[dev].store(v, std::memory_order_relaxed);
    }
    d.n_cus = v;
    d.status_word = g_status_of[dev].load(std::memory_order_relaxed);
  }
  return d;
}

namespace {

struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* p) : base(static_cast<char*>(p)), off(0) {}
  float* take(size_t floats) {
    float* r = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += align256(floats * sizeof(float));
    return r;
  }
};

int check_desc(const AgcnLayerDesc* d, const char* who) {
  if (!d) return fail(AGCN_EINVAL, "%s: null layer descriptor", who);
  if (d->Fin < 1 || d->Fin > 1024 || d->Fout < 1 || d->Fout > 1024)
    return fail(AGCN_EINVAL, "%s: Fin/Fout must be in [1, 1024] (got %d, %d)", who, d->Fin, d->Fout);
  if (d->K < 1 || d->K > 64) return fail(AGCN_EINVAL, "%s: K must be in [1, 64] (got %d)", who, d->K);
  if (d->mode != AGCN_MODE_REFERENCE && d->mode != AGCN_MODE_PAPER)
    return fail(AGCN_EINVAL, "%s: unknown laplacian mode %d", who, d->mode);
  if (d->variant != AGCN_SGC_LL && d->variant != AGCN_SGC_LL_RESLAP)
    return fail(AGCN_EINVAL, "%s: unknown layer variant %d", who, d->variant);
  if (d->activation != AGCN_ACT_LINEAR && d->activation != AGCN_ACT_RELU)
    return fail(AGCN_EINVAL, "%s: unsupported activation %d (linear or relu)", who, d->activation);
  if (d->ldx < d->Fin || d->ldy < d->Fout) return fail(AGCN_EINVAL, "%s: ldx/ldy smaller than Fin/Fout", who);
  if (d->feat_dtype != AGCN_DTYPE_F32 && d->feat_dtype != AGCN_DTYPE_BF16)
    return fail(AGCN_EINVAL, "%s: unknown feature dtype %d", who, d->feat_dtype);
  return AGCN_OK;
}

// Graphs are scheduled largest first; the leading ones whose on-chip image exceeds the LDS
// limit run with their tile in global scratch.
int count_global(const PlanView& host, int32_t Fin, bool bwd) {
  const size_t limit = (size_t)agcn_get_lds_limit();
  int g = 0;
  for (int i = 0; i < host.info.B; ++i) {
    const int32_t nb = host.n[host.order[i]];
    const size_t need = bwd ? sgc_lds_bytes_bwd(nb, Fin) : sgc_lds_bytes(nb, Fin);
    if (need > limit) ++g; else break;
  }
  return g;
}

// Slab-parallel kernels (slab_kernels.hip): SGC_LL, K = 2, packed rows, symmetric L_orig (CSR or dense).
bool use_slabs(const AgcnLayerDesc* d, const AgcnGraphBatch* g, const AgcnPlanInfo& pi) {
  if (knob_slab_mode() == 0) return false;
  return d->variant == AGCN_SGC_LL && d->K == 2 && pi.rows == pi.N1 &&
         (g->L0_rowptr != nullptr || g->L0_dense != nullptr) && g->L0_symmetric == 1 && slab_path_fits(d->Fin);
}

// Fused per-graph kernels (fused_kernels.hip): SGC_LL and SGC_LL_Reslap, K = 2, packed rows, CSR or dense L_orig, for the graphs
// whose working set fits LDS.  Graphs are scheduled largest first, so the ones that do NOT fit are a
// prefix of plan.order (and their slabs a prefix of the slab table): with `mixed` dispatch those few
// graphs run on the slab-parallel kernels on a side stream, concurrently with the fused kernels.
struct FusedSplit {
  int mode;        // 0: fused path not used; 1: every graph fused; 2: mixed
  int n_big;       // graphs (prefix of order) that do not fit
  int n_big_slabs; // their slabs (prefix of the slab table)
  int thr;         // largest fitting graph size: a graph is "big" iff n > thr
  int big_pg;      // mixed: 1 = the big graphs run on the per-graph LDS-tile kernels (sgc_forward/backward.hip),
                   //        0 = on the slab-parallel kernels
  // 1 = the fused forward / backward kernel reads M_L / W straight from global memory (no LDS images of them):
  // wide layers (Fin = 128) and graphs up to ~100 nodes fit this way
  int gb_fwd, gb_bwd_w, gb_bwd_m;   // backward: W and M_L separately
  // shape-specialised kernels (spec_*.hip) instead of the general fused ones, per direction
  int spec_fwd, spec_bwd;
  int spec_bwd_first;   // schedule position of the first graph the specialised backward kernel takes
  int spec_small_first; // schedule position of the first graph of the 8-wave flavour (== B: no such launch)
  int coop_forced;      // leading graphs that only the cooperative members can take (96 < n <= 128)
};

// The specialised kernels cover the whole batch or nothing (one launch per direction).
void pick_spec(FusedSplit* f, const AgcnLayerDesc* d, const PlanView& host, size_t lim) {
  const AgcnPlanInfo& pi = host.info;
  f->spec_fwd = f->spec_bwd = f->spec_bwd_first = 0;
  f->spec_small_first = pi.B;
  if (f->mode != 1 || !spec_supported(d->Fin, d->Fout, d->K, d->variant) || pi.tile_elems >= (1ll << 31)) return;
  const int m = knob_spec_mode();
  if ((m & 1) && spec_fwd_lds_bytes(pi.max_n, d->Fin, d->Fout, spec::NW_LARGE) <= lim) f->spec_fwd = 1;
  if ((m & 2) && f->spec_fwd) {   // (the specialised forward wrote the transposed parameter copies the backward reads)
    int first = 0;
    while (first < pi.B && spec_bwd_lds_bytes(host.n[host.order[first]], d->Fin, d->Fout, spec::NW_LARGE) > lim) ++first;
    // worth a second launch only when the specialised kernel takes (nearly) the whole batch
    if (first < pi.B && first <= pi.B / 8) { f->spec_bwd = 1; f->spec_bwd_first = first; }
  }
  // Many graphs per CU: the small graphs (a suffix of the largest-first schedule) go to the 8-wave flavour, whose
  // workgroups share a CU (spec_kernels.h).  With about one graph per CU a second launch only adds a boundary.
  // (measured, Tox21-shaped batches, graphs/s with one launch | two launches side by side:
  //  512: 5.7 | 5.1 M;  768: 6.6 | 6.0 M;  1024: 7.2 | 7.4 M;  1280: 7.5 | 8.5 M;  2048: 8.0 | 10.6 M;  4096: 7.3 | 13.0 M)
  static const int min_b = [] { const char* e = std::getenv("AGCN_SMALL_MIN_B"); return e ? std::atoi(e) : 1024; }();
  if ((f->spec_fwd || f->spec_bwd) && pi.B >= min_b && knob_spec_small() != 0) {
    // the 8-wave launch's LDS per workgroup is that of its largest graph and its registers allow three workgroups per
    // CU: a lower cut keeps three of them resident (measured, 2048 graphs: cut at 40 / 32 / 24 nodes: 9.9 / 10.5 /
    // 10.6 M graphs/s; 4096 graphs: 12.7 / 13.0 / 12.2 M)
    static const int cut_force = [] { const char* e = std::getenv("AGCN_SMALL_CUT"); return e ? std::atoi(e) : 0; }();
    const int cut = cut_force > 0 ? std::min(cut_force, spec::SMALL_N) : (pi.B < 3072 ? 24 : 32);
    int sf = pi.B;
    while (sf > 0 && host.n[host.order[sf - 1]] <= cut) --sf;
    if (pi.B - sf >= std::min(1024, min_b / 2)) f->spec_small_first = sf;
  }
}
// A batch whose few largest graphs exceed the specialised kernels' LDS plan (n > 96: HIV-shaped batches have one or two
// such molecules in 40 % of the batches): those graphs take the slab-parallel pipeline on the side stream, everything
// else the specialised kernels beside it.
void mixed_spec(FusedSplit* f, const AgcnLayerDesc* d, const AgcnGraphBatch* g, const PlanView& host, size_t lim) {
  const AgcnPlanInfo& pi = host.info;
  // opt-in (agcn_set_fused_mode(3)): measured SLOWER than the whole batch on the slab path -- the side chain's six
  // launches each wait for CUs the specialised workgroups hold (HIV-shaped epoch: 1.58 M graphs/s mixed, 1.78 M all-slab)
  // SGC_LL_Reslap has no slab flavour: without this its whole batch would leave for the general per-graph pipeline (HIV-shaped,
  // one 158-atom molecule: 445 us per step against 73) -- so its oversized molecules take the per-graph kernels on the side
  // stream BY DEFAULT (big_pg; AGCN_RESLAP_MIXED=0 disables)
  static const int res_mixed = [] { const char* e = std::getenv("AGCN_RESLAP_MIXED"); return e ? std::atoi(e) : 1; }();
  const bool reslap = d->variant == AGCN_SGC_LL_RESLAP;
  if (knob_fused_mode() < ((reslap && res_mixed) ? 1 : 3) || (knob_spec_mode() & 3) != 3) return;
  if (!spec_supported(d->Fin, d->Fout, d->K, d->variant) || pi.tile_elems >= (1ll << 31)) return;
  if (pi.rows != pi.N1 || (g->L0_rowptr != nullptr) == (g->L0_dense != nullptr) || (!reslap && !use_slabs(d, g, pi))) return;
  if (reslap && g->L0_dense != nullptr) {   // (point clouds -- every graph a multiple of 64 nodes, dense L_orig -- stay with the cloud kernels)
    bool all64 = true;
    for (int i = 0; i < pi.B && all64; ++i) all64 = (host.n[i] & 63) == 0;
    if (all64) return;
  }
  int nb = 0, ns = 0;
  while (nb < pi.B) {
    const int n = host.n[host.order[nb]];
    if (std::max(spec_fwd_lds_bytes(n, d->Fin, d->Fout, spec::NW_LARGE),
                 spec_bwd_lds_bytes(n, d->Fin, d->Fout, spec::NW_LARGE)) <= lim) break;
    ns += (n + AGCN_SLAB_ROWS - 1) / AGCN_SLAB_ROWS;
    ++nb;
  }
  if (nb == 0 || nb == pi.B || nb > std::max(2, pi.B / 16)) return;
  FusedSplit m{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  m.mode = 2; m.n_big = nb; m.n_big_slabs = reslap ? 0 : ns; m.thr = host.n[host.order[nb]]; m.big_pg = reslap ? 1 : 0;
  m.spec_fwd = m.spec_bwd = 1; m.spec_bwd_first = nb; m.spec_small_first = pi.B;
  *f = m;
}
// Cooperative kernels (spec_backward_coop.h): how many of the largest graphs (a prefix of the schedule) are spread over
// one workgroup per 16-row tile, and how many members (slabs) that makes.  The same answer for the forward and the
// backward of a batch.
void coop_plan(const AgcnLayerDesc* d, const PlanView& host, const FusedSplit& fs, bool backward, int* graphs,
               int* slabs, int* dual_graphs) {
  *graphs = *slabs = *dual_graphs = 0;
  const int B = host.info.B;
  // (reference mode: the backward has no n x n work to spread -- members there only for the graphs no single workgroup can
  //  hold, fs.coop_forced; the forward still has the similarity tile)
  // SGC_LL_Reslap: members in both directions too (they exchange the graph-wide clip norms / inner products as well)
  const bool reslap = d->variant == AGCN_SGC_LL_RESLAP;
  if (d->variant != AGCN_SGC_LL && !reslap) return;
  static const int res_coop = [] { const char* e = std::getenv("AGCN_RESLAP_COOP"); return e ? std::atoi(e) : 3; }();   // bit 0 forward, bit 1 backward
  if (reslap && !((res_coop >> (backward ? 1 : 0)) & 1)) return;
  if (fs.mode != 1 || !fs.spec_fwd || !fs.spec_bwd || fs.spec_bwd_first != 0 || fs.spec_small_first < B || (knob_spec_mode() & 4) == 0)
    return;
  if (d->mode != AGCN_MODE_PAPER && backward) {
    // reference mode: members only for the graphs no single workgroup can hold (coop_reach: the fs.coop_forced leading ones)
    int s = 0;
    for (int k = 0; k < fs.coop_forced && k < B - 1; ++k) s += (host.n[host.order[k]] + AGCN_SLAB_ROWS - 1) / AGCN_SLAB_ROWS;
    if (fs.coop_forced > 0 && fs.coop_forced < B && s <= std::min((int)spec::COOP_MAX_MEMBERS, (int)knob_coop_members())) {
      *graphs = fs.coop_forced; *slabs = s;
    }
    return;
  }
  // Members are whole-CU workgroups like every other one.  While the launch still fits the chip (graphs + extra
  // members <= CUs) every graph of three tiles and up is spread out (B = 128: backward 27.2 -> 17.8 us).  Beyond
  // that an extra workgroup starts only when another one has finished -- 12 us even for the smallest graph -- and
  // runs 12 us itself, so only graphs whose own chain is much longer than that 24 us pay: five tiles and up
  // (the n = 95 graph of an 8-GPU shard: backward 46.2 -> 26.3 us; B = 256 with n <= 59: nothing to gain, measured).
  static const int min_force = [] { const char* e = std::getenv("AGCN_COOP_MIN_N"); return e ? std::atoi(e) : 0; }();
  // CU count of the CURRENT device (the thresholds below rest on "members fit the chip")
  const int n_cus = current_device_info().n_cus;
  const int member_budget = std::min((int)spec::COOP_MAX_MEMBERS, (int)knob_coop_members());
  // (with two and more rounds of workgroups per CU the tail graph matters less than the extra workgroups cost:
  //  512 graphs: 6.0 M graphs/s without, 5.7 M with members for its one 70-node graph)
  static const int dual_on = [] { const char* e = std::getenv("AGCN_DUAL"); return e ? std::atoi(e) : 1; }();
  // (never beyond DUAL_MAX_N: both halves of a paired workgroup must run the same barrier sequence, spec_forward.hip)
  static const int dual_max = [] {
    const char* e = std::getenv("AGCN_DUAL_MAX_N");
    return std::max(1, std::min(e ? std::atoi(e) : (int)spec::DUAL_MAX_N, (int)spec::DUAL_MAX_N));
  }();
  int tiny = 0;   // graphs that may be packed two to a workgroup: a suffix of the schedule
  while (tiny < B - 1 && host.n[host.order[B - 1 - tiny]] <= dual_max) ++tiny;
  if (B > n_cus + n_cus / 2) {
    // no members -- but from two rounds of workgroups per CU on, two small graphs per workgroup shorten the queue
    // (512 / 768 / 1000 graphs: 6.0 -> 6.2, 6.9 -> 7.1, 7.2 -> 7.7 M graphs/s; 400 graphs: 5.3 -> 5.0 M, not used)
    if (dual_on && B >= 2 * n_cus) *dual_graphs = (tiny / 2) * 2;
    // ... except for the graphs only members can take (fs.coop_forced, coop_reach): the members are the first workgroups of
    // the launch, so they are resident together however long the queue behind them is
    if (fs.coop_forced > 0 && fs.coop_forced < B - *dual_graphs) {
      int s = 0;
      for (int k = 0; k < fs.coop_forced; ++k) s += (host.n[host.order[k]] + AGCN_SLAB_ROWS - 1) / AGCN_SLAB_ROWS;
      if (s <= member_budget) { *graphs = fs.coop_forced; *slabs = s; }
    }
    return;
  }
  const int base_min = min_force >= 17 ? min_force : spec::COOP_MIN_N;
  // The threshold: the smallest of base_min, base_min + 8, .. , COOP_MIN_N_FULL at which the launch still fits the chip -- after
  // room has been made by packing the SMALLEST graphs (at most DUAL_MAX_N nodes: a suffix of the schedule) two to a workgroup
  // (8 waves each, spec_forward.hip): a pair frees one CU.  (HIV-shaped batches have ~60 graphs of three tiles and up: the
  // three- and four-tile ones cannot all be spread, the five-tile ones can once pairs make room: 57.3 -> 56.3 us per step
  // against going straight to the last threshold without pairs; workgroups beyond the chip cost more than they save: 64 us.)
  int min_n = std::max(base_min, spec::COOP_MIN_N_FULL), pairs = 0;
  for (int t = base_min; t <= std::max(base_min, spec::COOP_MIN_N_FULL); t = (t < spec::COOP_MIN_N_FULL && t + 8 > spec::COOP_MIN_N_FULL) ? spec::COOP_MIN_N_FULL : t + 8) {
    int extra = 0;
    for (int k = 0; k < B; ++k) {
      const int n = host.n[host.order[k]];
      if (n < t) break;
      extra += (n + AGCN_SLAB_ROWS - 1) / AGCN_SLAB_ROWS - 1;
    }
    const int pr = (dual_on && B + extra > n_cus) ? std::min(tiny / 2, B + extra - n_cus) : 0;
    if (B + extra - pr <= n_cus) { min_n = t; pairs = pr; break; }
  }
  // (graphs only the members can take -- fs.coop_forced leading ones, n > 96 -- are above either threshold)
  *dual_graphs = 2 * pairs;
  int g = 0, s = 0;
  while (g < B - 1 - 2 * pairs) {   // (never every graph: the last one of the schedule writes the transposed parameters)
    const int n = host.n[host.order[g]];
    const int sl = (n + AGCN_SLAB_ROWS - 1) / AGCN_SLAB_ROWS;
    if (n < min_n || s + sl > member_budget) break;
    s += sl;
    ++g;
  }
  *graphs = g; *slabs = s;
}
// Batches whose few largest graphs are beyond the one-workgroup kernels (n > 96) but within the members' reach
// (n <= COOP_MAX_N and their LDS plan fits): the specialised kernels take the batch with those graphs spread over
// members -- instead of the whole batch on the slab kernels (HIV-shaped epoch: 36 % of the batches).  One 16-wave launch.
// (Reference mode: the backward spreads exactly these graphs -- its plain workgroups have no n x n work worth spreading.)
void coop_reach(FusedSplit* f, const AgcnLayerDesc* d, const AgcnGraphBatch* g, const PlanView& host, size_t lim) {
  const AgcnPlanInfo& pi = host.info;
  if (knob_fused_mode() == 0 || (knob_spec_mode() & 7) != 7 ||
      (d->variant != AGCN_SGC_LL && d->variant != AGCN_SGC_LL_RESLAP)) return;
  if (!spec_supported(d->Fin, d->Fout, d->K, d->variant) || pi.tile_elems >= (1ll << 31)) return;
  if (pi.rows != pi.N1 || (g->L0_rowptr != nullptr) == (g->L0_dense != nullptr) || pi.B >= 1024 || pi.B < 2) return;
  int k = 0;
  while (k < pi.B - 1) {
    const int n = host.n[host.order[k]];
    if (std::max(spec_fwd_lds_bytes(n, d->Fin, d->Fout, spec::NW_LARGE),
                 spec_bwd_lds_bytes(n, d->Fin, d->Fout, spec::NW_LARGE)) <= lim) break;
    if (n > spec::COOP_MAX_N_WIDE || spec_coop_lds_bytes(n, d->Fin, d->Fout) > lim || spec_coop_fwd_lds_bytes(n, d->Fin) > lim)
      return;
    ++k;
  }
  if (k == 0 || k >= pi.B - 1) return;
  FusedSplit m{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  m.mode = 1; m.thr = host.n[host.order[k]];
  m.spec_fwd = m.spec_bwd = 1; m.spec_bwd_first = 0; m.spec_small_first = pi.B; m.coop_forced = k;
  int cg = 0, cs = 0, dg = 0;
  coop_plan(d, host, m, true, &cg, &cs, &dg);
  if (cg < k) return;   // (the planner would not spread all of them: member budget, batch too large)
  coop_plan(d, host, m, false, &cg, &cs, &dg);
  if (cg < k) return;
  *f = m;
}
FusedSplit fused_split_general(const AgcnLayerDesc* d, const AgcnGraphBatch* g, const PlanView& host);
FusedSplit fused_split(const AgcnLayerDesc* d, const AgcnGraphBatch* g, const PlanView& host) {
  FusedSplit f = fused_split_general(d, g, host);
  size_t lim = (size_t)agcn_get_lds_limit();
  if (lim > 160 * 1024) lim = 160 * 1024;
  pick_spec(&f, d, host, lim);
  if (!(f.mode == 1 && f.spec_fwd && f.spec_bwd && f.spec_bwd_first == 0)) coop_reach(&f, d, g, host, lim);
  if (!(f.mode == 1 && f.spec_fwd && f.spec_bwd)) mixed_spec(&f, d, g, host, lim);
  return f;
}
FusedSplit fused_split_general(const AgcnLayerDesc* d, const AgcnGraphBatch* g, const PlanView& host) {
  FusedSplit f{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const AgcnPlanInfo& pi = host.info;
  if (knob_fused_mode() == 0) return f;
  if (!(d->K == 2 && pi.rows == pi.N1 && (g->L0_rowptr != nullptr) != (g->L0_dense != nullptr))) return f;
  size_t lim = (size_t)agcn_get_lds_limit();
  if (lim > 160 * 1024) lim = 160 * 1024;
  auto need = [&](int n) {
    return std::max(fused_lds_bytes(n, d->Fin, d->Fout, d->K), fused_bwd_lds_bytes(n, d->Fin, d->Fout, d->K));
  };
  auto count_big = [&](size_t limit, int* slabs) {
    int nb = 0, ns = 0;
    while (nb < pi.B && need(host.n[host.order[nb]]) > limit) {
      ns += (host.n[host.order[nb]] + AGCN_SLAB_ROWS - 1) / AGCN_SLAB_ROWS;
      ++nb;
    }
    *slabs = ns;
    return nb;
  };
  const int force_gb = knob_fused_gb();
  int ns = 0;
  int nb = count_big(lim, &ns);
  const bool gb_fits = std::max(fused_lds_bytes_gb(pi.max_n, d->Fin, d->Fout, d->K),
                                fused_bwd_lds_bytes_gb(pi.max_n, d->Fin, d->Fout, d->K, true, true)) <= lim;
  if (nb == 0) {
    // Both flavours fit.  Measured on MI355X (256 Tox21-shaped graphs): reading the parameters from global
    // memory is faster in the forward kernel (75->64: 23.5 vs 24.7 us; 64->128: 27.5 vs 29.9 us) and, when the
    // rows of the parameter are 16-byte aligned (vector loads along k), in the backward kernel too -- decided per
    // parameter: W rows have Fout floats, M_L rows Fin floats (64->128: 35.7 vs 36.4 us with both from global
    // memory; 75->64: W from global memory, M_L, rows of 75 floats, as an LDS image).
    f.mode = 1; f.thr = pi.max_n;
    if (gb_fits && force_gb != 0) {
      f.gb_fwd = 1;
      f.gb_bwd_w = (force_gb == 1 || (d->Fout & 3) == 0) ? 1 : 0;
      f.gb_bwd_m = (force_gb == 1 || (d->Fin & 3) == 0) ? 1 : 0;
      if (fused_bwd_lds_bytes_gb(pi.max_n, d->Fin, d->Fout, d->K, f.gb_bwd_w != 0, f.gb_bwd_m != 0) > lim)
        f.gb_bwd_w = f.gb_bwd_m = 1;
    }
    return f;
  }
  // the parameter images are what does not fit: read M_L / W from global memory instead (AGCN_FUSED_GB=0 disables)
  if (gb_fits && force_gb != 0) {
    f.mode = 1; f.thr = pi.max_n; f.gb_fwd = f.gb_bwd_w = f.gb_bwd_m = 1;
    return f;
  }
  // Some graph does not fit.  Measured on MI355X (Tox21-shaped, 256 graphs per GPU, one n = 95 graph):
  // all-slab 117 us/step, mixed 144 us/step -- the side chain of four dependent launches is longer than
  // the fused kernel it runs beside, so mixed dispatch is opt-in (agcn_set_fused_mode(2)) and the whole
  // batch takes the slab path by default.
  if (knob_fused_mode() < 2) return f;
  if (nb == pi.B) return f;
  // Big graphs whose tile still fits LDS run one workgroup each on the per-graph kernels (3 + 3 launches on the
  // side stream, any free CU); no co-residency with the fused workgroups is needed.
  if (knob_fused_mode() == 2) {
    const int maxn = host.n[host.order[0]];
    if (std::max(sgc_lds_bytes(maxn, d->Fin), sgc_lds_bytes_bwd(maxn, d->Fin)) <= lim) {
      f.mode = 2; f.n_big = nb; f.n_big_slabs = 0; f.thr = host.n[host.order[nb]]; f.big_pg = 1;
      return f;
    }
  }
  if (!use_slabs(d, g, pi)) return f;   // the big graphs have no slab path
  // Mixed dispatch only pays if the side-stream kernels can be CO-RESIDENT with the fused workgroups
  // (a fused workgroup is 16 waves + up to 150 KiB of LDS, i.e. a whole CU).  So the fused part is limited
  // to graphs whose image leaves ~48 KiB of LDS (one slab workgroup) free on every CU; everything larger
  // goes to the slab-parallel side chain, which is spread over all CUs.
  const size_t co_lim = 112 * 1024;
  if (lim > co_lim) nb = count_big(co_lim, &ns);
  if (nb == pi.B) return f;
  f.mode = 2; f.n_big = nb; f.n_big_slabs = ns; f.thr = host.n[host.order[nb]];
  return f;
}

// Side stream + fork / join events of the mixed dispatch: one set per DEVICE (created on first use with that device
// current), and a lock that a mixed-dispatch call holds while it enqueues -- two threads driving the same device
// would otherwise interleave their fork / join records on the shared events.
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  int ok = 0;
  std::mutex mu;
};
SideStream& side_stream() {
  static std::mutex table_mu;
  static std::unordered_map<int, std::unique_ptr<SideStream>> table;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(table_mu);
  std::unique_ptr<SideStream>& p = table[dev];
  if (!p) p.reset(new SideStream());
  SideStream& ss = *p;
  if (!ss.ok) {
    if (hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) == hipSuccess &&
        hipEventCreateWithFlags(&ss.fork, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess) ss.ok = 1;
  }
  return ss;
}

struct ReserveLayout { size_t Z, s, scal, X, X0, Lpre, r, MT, WT, WTI, xch, total; };
ReserveLayout reserve_layout(const AgcnLayerDesc* d, const AgcnPlanInfo& pi) {
  Carver c(nullptr);
  ReserveLayout r;
  // exchange header of the cooperative kernels (spec_backward_coop.h): fixed size, at offset 0 whatever the batch, zero
  // when the reserve is first used (include/agcn.h) and self-maintaining afterwards
  // + the words of the trailing parameter-gradient workgroups (spec_tail.h): generation, exit counter, producer flags
  // + the scalar exchange words of the SGC_LL_Reslap members (behind the tail words)
  r.xch = c.off; c.take((size_t)2 * (spec::COOP_SCAL_WORD + spec::COOP_SCAL_WORDS));
  static_assert(sizeof(float) * (spec::TAIL_OFF_FLOATS + spec::TAIL_WORDS) <= AGCN_RESERVE_HEADER_BYTES, "reserve header");
  static_assert(sizeof(uint64_t) * (spec::COOP_SCAL_WORD + spec::COOP_SCAL_WORDS) <= AGCN_RESERVE_HEADER_BYTES, "reserve header");
  r.Z = c.off; c.take((size_t)pi.rows * d->Fin);
  r.s = c.off; c.take((size_t)pi.rows);
  r.scal = c.off; c.take((size_t)4 * pi.B);
  r.X = c.off; c.take((size_t)(d->K - 1) * pi.rows * d->Fin);
  // bf16 feature storage: the widened fp32 copy of X the forward leaves for the backward and the parameter gradients
  r.X0 = c.off; if (d->feat_dtype == AGCN_DTYPE_BF16) c.take((size_t)pi.rows * d->Fin);
  r.Lpre = c.off; if (d->variant == AGCN_SGC_LL_RESLAP) c.take((size_t)pi.tile_elems);
  // r_ij tiles (fused and slab paths, paper mode): reserved whenever one of them COULD be taken
  r.r = c.off;
  if (d->K == 2 && d->mode == AGCN_MODE_PAPER) c.take((size_t)pi.tile_elems);
  // transposed parameter copies (specialised kernels)
  r.MT = c.off; c.take((size_t)((d->Fin + 15) & ~15) * d->Fin);      // k-blocked: [F4 / 4][Fin][4], zero beyond Fin
  r.WT = c.off; c.take((size_t)d->K * d->Fout * d->Fin);
  // bf16 features (spec_bfpipe.h): W^T as a three-piece image in MFMA operand order for the backward's filter adjoint
  r.WTI = c.off;
  if (d->K == 2 && d->Fout == 64)
    c.take((size_t)d->K * (d->Fout / 32) * (((d->Fin + 15) & ~15) / 16) * 768);
  r.total = c.off;
  return r;
}

// Row chunks of the parameter-gradient partial products: 64-row chunks (16 MFMA k-steps) give
// a few thousand wave-sized work items at Tox21 batch sizes; capped so the partial buffer stays small.
int n_chunks_for(int32_t rows) {
  int c = (rows + 63) / 64;
  if (c < 1) c = 1;
  if (c > 256) c = 256;
  return c;
}

void fill_rows(RowsGemmArgs& g, const AgcnLayerDesc* d, const PlanView& host, const PlanView& dev) {
  std::memset(&g, 0, sizeof(g));
  g.rows = host.info.rows; g.Fin = d->Fin; g.Fout = d->Fout; g.K = d->K; g.activation = d->activation;
  // padded rows: every graph owns the same number of rows
  g.row_stride = (host.info.rows != host.info.N1) ? host.info.rows / host.info.B : 0;
  g.n = dev.n;
}

// every graph a multiple of 64 rows: four consecutive slabs always belong to one graph (slab_kernels.hip, Q = 4)
static int slab_quad_ok(const PlanView& host) {
  for (int b = 0; b < host.info.B; ++b)
    if (host.n[b] % 64 != 0) return 0;
  return 1;
}

void fill_common(SgcArgs& a, const AgcnLayerDesc* d, const AgcnGraphBatch* g, const AgcnParams* p,
                 const AgcnOutputs* o, const PlanView& dev) {
  std::memset(&a, 0, sizeof(a));
  a.n = dev.n; a.row_off = dev.row_off; a.tile_off = dev.tile_off; a.order = dev.order;
  a.graph_desc = dev.graph_desc;
  a.wtile_off = dev.wtile_off;
  a.B = dev.info.B; a.rows = dev.info.rows;
  a.Fin = d->Fin; a.Fout = d->Fout; a.K = d->K; a.mode = d->mode; a.variant = d->variant;
  a.activation = d->activation; a.ldx = d->ldx; a.ldy = d->ldy;
  a.X = g->X; a.L0_dense = g->L0_dense; a.L0_ld = g->L0_ld; a.L0_rowptr = g->L0_rowptr;
  a.L0_col = g->L0_col; a.L0_val = g->L0_val;
  a.L_prev = d->variant == AGCN_SGC_LL_RESLAP ? g->L_prev : nullptr;
  a.weight = p->weight; a.bias = p->bias; a.M_L = p->M_L; a.alpha = p->alpha; a.beta = p->beta;
  a.Y = o->Y; a.res_L = o->res_L; a.res_W = o->res_W; a.L_sum = o->L_sum;
  a.foff = 0; a.fstride = d->Fin; a.half_acc = 0; a.half_last = 1;
  a.feat_bf16 = d->feat_dtype == AGCN_DTYPE_BF16 ? 1 : 0; a.lddx = d->ldx;
}

int check_batch(const AgcnLayerDesc* d, const AgcnGraphBatch* g, const AgcnParams* p, const char* who) {
  if (!g || !p) return fail(AGCN_EINVAL, "%s: null batch/params", who);
  if (!g->plan_host || !g->plan_dev || !g->X) return fail(AGCN_EINVAL, "%s: null plan or X", who);
  if (g->L0_dense && g->L0_rowptr) return fail(AGCN_EINVAL, "%s: give the original Laplacian dense OR CSR, not both", who);
  if (g->L0_rowptr && (!g->L0_col || !g->L0_val)) return fail(AGCN_EINVAL, "%s: incomplete CSR Laplacian", who);
  if (g->L0_ld < 0) return fail(AGCN_EINVAL, "%s: negative L0_ld", who);
  if (!p->weight || !p->bias || !p->M_L || !p->alpha) return fail(AGCN_EINVAL, "%s: null parameter", who);
  if (d->variant == AGCN_SGC_LL_RESLAP && g->L_prev && !p->beta)
    return fail(AGCN_EINVAL, "%s: L_prev given without beta", who);
  return AGCN_OK;
}

}  // namespace
}  // namespace agcn

using namespace agcn;

// Which kernels a forward / backward call on this batch would launch (host logic only: nothing is launched, no device
// pointer is dereferenced).  For tests and diagnostics.
static bool cloud_shape(const AgcnLayerDesc* d);
static bool split_images(const AgcnLayerDesc* d);
static bool cloud_candidate(const AgcnLayerDesc* d, const AgcnGraphBatch* g, const AgcnPlanInfo& pi);

// bf16 feature storage is served when every graph of the batch takes the specialised kernels in both directions (the general
// fused / slab / per-graph kernels read and write fp32 rows only)
static bool bf16_served(const AgcnLayerDesc* d, const PlanView& host, const FusedSplit& fs) {
  (void)host;
  return fs.mode == 1 && fs.spec_fwd && fs.spec_bwd && d->variant == AGCN_SGC_LL && d->K == 2 && fs.spec_bwd_first == 0;
}

// Parameter gradients as trailing workgroups of the specialised backward launch (spec_tail.h): SGC_LL, paper mode, the
// whole batch in ONE 16-wave launch (no general fused kernel in front, no 8-wave launch beside it), units within the header.
// agcn_set_spec_tail(0) / AGCN_SPEC_TAIL=0 keeps the separate parameter-gradient kernel (A/B measurements).
static bool spec_tail_ok(const AgcnLayerDesc* d, const PlanView& host, const FusedSplit& fs, int coop_graphs, int coop_slabs) {
  if (!knob_spec_tail() || fs.mode != 1 || !fs.spec_bwd || d->variant != AGCN_SGC_LL || d->mode != AGCN_MODE_PAPER || d->K != 2) return false;
  const int B = host.info.B;
  if (fs.spec_bwd_first != 0 || std::max(fs.spec_bwd_first, fs.spec_small_first) < B) return false;
  return coop_slabs + (B - coop_graphs) <= spec::TAIL_MAX_UNITS &&
         spec::tail_tiles(d->Fin, d->Fout, d->K, true) <= spec::TAIL_MAX_TILES;
}
// 4-row k-steps of the tail's walk: the graphs that are units of their own + the member slabs
static int spec_tail_ksteps(const PlanView& host, int coop_graphs, int coop_slabs) {
  int T = 0;
  for (int sp = coop_graphs; sp < host.info.B; ++sp) T += (host.n[host.order[sp]] + 3) / 4;
  for (int sl = 0; sl < coop_slabs; ++sl) T += (std::min(16, host.n[host.slab_graph[sl]] - host.slab_row[sl]) + 3) / 4;
  return T;
}

extern "C" int agcn_sgc_ll_dispatch_info(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch, int32_t* out, int32_t n_out) {
  KnobScope knobs(desc ? desc->flags : 0u);
  const char* who = "agcn_sgc_ll_dispatch_info";
  int rc = check_desc(desc, who);
  if (rc != AGCN_OK) return rc;
  if (!batch || !batch->plan_host || !out || n_out < AGCN_DISPATCH_INFO_INTS) return fail(AGCN_EINVAL, "%s: bad argument", who);
  PlanView host;
  rc = plan_view(batch->plan_host, batch->plan_host, &host);
  if (rc != AGCN_OK) return rc;
  const FusedSplit fs = fused_split(desc, batch, host);
  int cg = 0, cs = 0, dg = 0, cgb = 0, csb = 0, dgb = 0;
  coop_plan(desc, host, fs, false, &cg, &cs, &dg);
  coop_plan(desc, host, fs, true, &cgb, &csb, &dgb);
  int path = fs.mode == 1 ? ((fs.spec_fwd ? 1 : 0) | (fs.spec_bwd ? 2 : 0) | 4)      // fused (+ specialised bits)
                          : (fs.mode == 2 ? 8 : (use_slabs(desc, batch, host.info) ? 16 : 32));
  // 64: the slab batch has the shape the point-cloud kernels take (pointer alignment is checked at the call)
  if ((path == 16 || path == 32) && cloud_shape(desc) && cloud_candidate(desc, batch, host.info) &&
      batch->L0_dense != nullptr && slab_quad_ok(host) && (desc->ldx & 3) == 0 && (desc->Fout & 15) == 0 && (desc->ldy & 3) == 0)
    path |= 64;
  if (spec_tail_ok(desc, host, fs, cgb, csb)) path |= 128;     // the backward launch carries the parameter gradients
  if (bf16_served(desc, host, fs)) path |= 256;                // feat_dtype = AGCN_DTYPE_BF16 is served on this batch
  const int32_t v[AGCN_DISPATCH_INFO_INTS] = {path, fs.spec_small_first, cg, cs, dg, cgb, csb, dgb};
  for (int i = 0; i < AGCN_DISPATCH_INFO_INTS; ++i) out[i] = v[i];
  return AGCN_OK;
}

extern "C" int agcn_sgc_ll_reserve_status(const void* reserve, size_t reserve_bytes, void* hip_stream, int32_t* timeouts) {
  const char* who = "agcn_sgc_ll_reserve_status";
  if (!reserve || !timeouts || reserve_bytes < sizeof(uint64_t) * ((size_t)spec::COOP_STATUS_WORD + 1)) return fail(AGCN_EINVAL, "%s: bad argument", who);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  uint32_t v = 0;
  const char* word = static_cast<const char*>(reserve) + sizeof(uint64_t) * (size_t)spec::COOP_STATUS_WORD;
  hipError_t e = hipMemcpyAsync(&v, word, sizeof(v), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: %s", who, hipGetErrorString(e));
  *timeouts = (int32_t)v;
  if (v != 0)
    return fail(AGCN_EDEVICE, "%s: %u wait(s) between the workgroups that share a large graph ran out (members not co-resident: "
                "too many concurrent launches for the chip, see agcn_set_coop_members); the results of those steps are NaN",
                who, v);
  return AGCN_OK;
}

extern "C" int agcn_set_device_status_word(int32_t device, void* word) {
  if (device < 0 || device >= MAX_DEVICES) return fail(AGCN_EINVAL, "agcn_set_device_status_word: device %d out of range", device);
  if (word != nullptr && (reinterpret_cast<uintptr_t>(word) & 3) != 0) return fail(AGCN_EINVAL, "agcn_set_device_status_word: unaligned word");
  g_status_of[device].store(static_cast<uint32_t*>(word), std::memory_order_relaxed);
  return AGCN_OK;
}

extern "C" int agcn_device_status(int32_t device, void* hip_stream, int32_t* timeouts) {
  const char* who = "agcn_device_status";
  if (device < 0 || device >= MAX_DEVICES || !timeouts) return fail(AGCN_EINVAL, "%s: bad argument", who);
  *timeouts = 0;
  const uint32_t* word = g_status_of[device].load(std::memory_order_relaxed);
  if (word == nullptr) return AGCN_OK;       // nothing registered: nothing to report
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  uint32_t v = 0;
  hipError_t e = hipMemcpyAsync(&v, word, sizeof(v), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: %s", who, hipGetErrorString(e));
  *timeouts = (int32_t)v;
  if (v != 0)
    return fail(AGCN_EDEVICE, "%s: %u wait(s) between the workgroups that share a large graph ran out on device %d since the "
                "status word was last zeroed (members not co-resident, see agcn_set_coop_members); the results of those steps "
                "are NaN", who, v, device);
  return AGCN_OK;
}

extern "C" size_t agcn_sgc_ll_reserve_bytes(const AgcnLayerDesc* desc, const void* plan_host) {
  KnobScope knobs(desc ? desc->flags : 0u);
  AgcnPlanInfo pi;
  if (check_desc(desc, "agcn_sgc_ll_reserve_bytes") != AGCN_OK || agcn_plan_info(plan_host, &pi) != AGCN_OK) return 0;
  return reserve_layout(desc, pi).total + 256;
}

// layer shapes the point-cloud kernels (cloud_kernels.hip) are compiled for: their forward needs partial-degree scratch
static bool cloud_shape(const AgcnLayerDesc* d) {
  return (d->variant == AGCN_SGC_LL || d->variant == AGCN_SGC_LL_RESLAP) && d->K == 2 && (d->Fin == 64 || d->Fin == 128);
}
// layer shapes whose packed-row products may run on the split-bf16 kernels (rows_split.hip): room for their parameter images
static bool split_images(const AgcnLayerDesc* d) { return cloud_shape(d) && (d->Fout == 64 || d->Fout == 128); }
// Batches the point-cloud kernels (cloud_kernels.hip) may take: what the slab kernels take (SGC_LL), and SGC_LL_Reslap with a
// dense symmetric L_orig on packed rows (cloud_path() checks the rest: multiples of 64 nodes, alignment, Fin = 64)
static bool cloud_candidate(const AgcnLayerDesc* d, const AgcnGraphBatch* g, const AgcnPlanInfo& pi) {
  if (d->variant == AGCN_SGC_LL && use_slabs(d, g, pi)) return true;
  // (the general slab kernels' LDS plan may not fit a 128-feature layer: the point-cloud kernels do not depend on it)
  if (d->variant == AGCN_SGC_LL)
    return knob_slab_mode() != 0 && d->K == 2 && pi.rows == pi.N1 && g->L0_dense != nullptr && g->L0_rowptr == nullptr &&
           g->L0_symmetric == 1;
  // (the Reslap kernels add a graph's per-strip partial sums with one lane per 64-row strip: graphs of up to 4096 nodes)
  return knob_slab_mode() != 0 && d->variant == AGCN_SGC_LL_RESLAP && d->K == 2 && pi.rows == pi.N1 && pi.max_n <= 4096 &&
         g->L0_dense != nullptr && g->L0_rowptr == nullptr && g->L0_symmetric == 1 &&
         (g->L_prev == nullptr || g->L_prev_symmetric == 1);
}

static size_t global_scratch_floats(const AgcnLayerDesc* d, const PlanView& host, bool bwd, Carver* c,
                                    float** tile, float** buf) {
  const int ng = count_global(host, d->Fin, bwd);
  *tile = *buf = nullptr;
  if (ng > 0) {
    *tile = c->take((size_t)host.wtile_off[host.info.B]);
    // one image region per graph forward, two (g_k and X_{k-1} / Z^T) backward
    *buf = c->take((bwd ? 2 : 1) * graph_scratch_floats(host.info.rows, host.info.B, d->Fin));
  }
  return ng;
}

extern "C" size_t agcn_sgc_ll_forward_workspace_bytes(const AgcnLayerDesc* desc, const void* plan_host) {
  KnobScope knobs(desc ? desc->flags : 0u);
  PlanView host;
  if (check_desc(desc, "agcn_sgc_ll_forward_workspace_bytes") != AGCN_OK ||
      plan_view(plan_host, plan_host, &host) != AGCN_OK) return 0;
  Carver c(nullptr);
  float *t, *b;
  global_scratch_floats(desc, host, false, &c, &t, &b);
  if (cloud_shape(desc)) c.take(cloud_dpart_floats(host.info.tile_elems));
  if (split_images(desc)) c.take(rows_split_fwd_image_bytes(desc->Fin, desc->Fout) / 4);
  return c.off + 256;
}

struct BwdWs { float *g, *dZ, *Gp, *pg, *part, *tile, *buf, *dA, *dAt, *dd, *WT, *MT, *spart, *simg, *P; int n_chunks; int64_t part_stride; int n_glb; size_t part_floats; };

static size_t carve_bwd(const AgcnLayerDesc* d, const PlanView& host, void* ws, BwdWs* w) {
  Carver c(ws);
  const AgcnPlanInfo& pi = host.info;
  const bool paper = d->mode == AGCN_MODE_PAPER;
  w->g = c.take((size_t)(d->K - 1) * pi.rows * d->Fin);
  w->dZ = paper ? c.take((size_t)pi.rows * d->Fin) : nullptr;
  w->Gp = c.take((size_t)pi.rows * d->Fout);
  w->pg = c.take((size_t)2 * ((size_t)pi.n_slabs + pi.B));
  w->dA = w->dAt = w->dd = nullptr;
  if (d->variant == AGCN_SGC_LL && d->K == 2 && paper) {   // slab path scratch
    w->dA = c.take((size_t)pi.tile_elems);
    w->dAt = c.take((size_t)pi.tile_elems);
    w->dd = c.take((size_t)pi.rows);
  }
  w->spart = nullptr;
  if (d->variant == AGCN_SGC_LL_RESLAP && cloud_shape(d)) {   // point-cloud kernels, Reslap: Q tile + dd (paper), strip partials
    if (paper) {
      w->dA = c.take((size_t)pi.tile_elems);
      w->dd = c.take((size_t)pi.rows);
    }
    w->spart = c.take((size_t)8 * ((size_t)pi.n_slabs / 4 + 8));
  }
  // ... 128 features: the P = g_1 X_0^T tile for the element-wise passes behind its writer (Reslap: pass B; SGC_LL: backward 1 by parts)
  // (paper mode only: in reference mode no pass reads it -- the launchers pass ws_P = nullptr there)
  w->P = (cloud_shape(d) && d->Fin == 128 && paper) ? c.take((size_t)pi.tile_elems) : nullptr;
  w->n_chunks = n_chunks_for(pi.rows);
  w->part_stride = (int64_t)d->K * d->Fin * d->Fout + d->Fout + (paper ? (int64_t)d->Fin * d->Fin : 0);
  w->part_stride = (w->part_stride + 3) / 4 * 4;
  w->part_floats = std::max((size_t)w->n_chunks * w->part_stride,
                            param_grad_onepass_floats(pi.rows, d->Fin, d->Fout, d->K, paper ? 1 : 0));
  w->part = c.take(w->part_floats);
  w->n_glb = (int)global_scratch_floats(d, host, true, &c, &w->tile, &w->buf);
  w->WT = cloud_shape(d) ? c.take((size_t)d->K * d->Fout * d->Fin) : nullptr;   // cloud kernels: W^T for the filter adjoint
  w->MT = (cloud_shape(d) && d->Fin == 128 && paper) ? c.take((size_t)d->Fin * d->Fin) : nullptr;   // rows_wide.hip: M_L^T
  w->simg = split_images(d) ? c.take(rows_split_bwd_image_bytes(d->Fin, d->Fout) / 4) : nullptr;    // rows_split.hip: images of W^T, M_L^T
  return c.off;
}

extern "C" size_t agcn_sgc_ll_backward_workspace_bytes(const AgcnLayerDesc* desc, const void* plan_host) {
  KnobScope knobs(desc ? desc->flags : 0u);
  PlanView host;
  if (check_desc(desc, "agcn_sgc_ll_backward_workspace_bytes") != AGCN_OK ||
      plan_view(plan_host, plan_host, &host) != AGCN_OK) return 0;
  BwdWs w;
  return carve_bwd(desc, host, nullptr, &w) + 256;
}

extern "C" int agcn_sgc_ll_forward(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch,
                                   const AgcnParams* params, const AgcnOutputs* out, void* reserve,
                                   size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                                   void* hip_stream) {
  KnobScope knobs(desc ? desc->flags : 0u);
  const char* who = "agcn_sgc_ll_forward";
  int rc = check_desc(desc, who);
  if (rc != AGCN_OK) return rc;
  rc = check_batch(desc, batch, params, who);
  if (rc != AGCN_OK) return rc;
  if (!out || !out->Y || !out->res_L || !out->res_W) return fail(AGCN_EINVAL, "%s: null output", who);
  const bool reslap = desc->variant == AGCN_SGC_LL_RESLAP;
  if (reslap && !out->L_sum) return fail(AGCN_EINVAL, "%s: SGC_LL_Reslap needs the L_sum output", who);
  PlanView host, dev;
  rc = plan_view(batch->plan_host, batch->plan_host, &host);
  if (rc != AGCN_OK) return rc;
  plan_view(batch->plan_host, batch->plan_dev, &dev);
  const ReserveLayout rl = reserve_layout(desc, host.info);
  if (!reserve || reserve_bytes < rl.total) return fail(AGCN_ENOSPACE, "%s: reserve too small (%zu < %zu)", who, reserve_bytes, rl.total);

  SgcArgs a;
  fill_common(a, desc, batch, params, out, dev);
  char* rb = static_cast<char*>(reserve);
  a.sv_Z = reinterpret_cast<float*>(rb + rl.Z);
  a.sv_s = reinterpret_cast<float*>(rb + rl.s);
  a.sv_scal = reinterpret_cast<float*>(rb + rl.scal);
  a.sv_X = reinterpret_cast<float*>(rb + rl.X);
  a.sv_X0 = reinterpret_cast<float*>(rb + rl.X0);
  a.sv_Lpre = reslap ? reinterpret_cast<float*>(rb + rl.Lpre) : nullptr;
  a.sv_MT = reinterpret_cast<float*>(rb + rl.MT);
  a.sv_WT = reinterpret_cast<float*>(rb + rl.WT);
  a.sv_WTI = (desc->K == 2 && desc->Fout == 64 && (knob_bfpipe() & (a.feat_bf16 ? 4 : 8)) &&
              (reinterpret_cast<uintptr_t>(params->weight) & 15) == 0) ? static_cast<void*>(rb + rl.WTI) : nullptr;
  a.sv_xch = reinterpret_cast<uint64_t*>(rb + rl.xch);
  a.dev_status = current_device_info().status_word;
  a.coop_ticks = agcn_get_coop_timeout_ticks();
  a.coop_slabs = 0; a.plain_count = 0; a.dual_pairs = 0; a.dual_stride = 0;
  a.tp_first = host.info.B - 1;
  if (a.feat_bf16) {   // (host logic: which kernels this batch takes)
    const FusedSplit fsb = fused_split(desc, batch, host);
    if (!bf16_served(desc, host, fsb))
      return fail(AGCN_EINVAL, "%s: bf16 features are read / written by the specialised kernels only and this batch does not take "
                  "them (agcn_sgc_ll_dispatch_info bit 256); convert with agcn_cast_rows", who);
  }

  Carver c(workspace);
  const int n_glb = (int)global_scratch_floats(desc, host, false, &c, &a.ws_tile, &a.ws_buf);
  float* dpart = cloud_shape(desc) ? c.take(cloud_dpart_floats(host.info.tile_elems)) : nullptr;
  float* simg = split_images(desc) ? c.take(rows_split_fwd_image_bytes(desc->Fin, desc->Fout) / 4) : nullptr;
  if (c.off > workspace_bytes || ((n_glb > 0 || dpart) && !workspace))
    return fail(AGCN_ENOSPACE, "%s: workspace too small (%zu < %zu)", who, workspace_bytes, c.off);

  a.slab_graph = dev.slab_graph; a.slab_row = dev.slab_row; a.slab_desc = dev.slab_desc; a.n_slabs = host.info.n_slabs;
    a.slab_quad = slab_quad_ok(host);
  a.sv_r = reinterpret_cast<float*>(rb + rl.r);
  const FusedSplit fs = fused_split(desc, batch, host);
  if (fs.mode == 1) {
    // small graphs: the whole layer forward in one launch, everything staged in LDS once
    if (fs.spec_fwd) {
      // Two bins: the few large graphs (16-wave workgroups, one per CU, as long as the largest graph takes) and the
      // many small ones (8-wave workgroups sharing CUs).  Run one after the other the first launch leaves most of the
      // chip idle behind its longest graph, so it goes to a side stream and the two launches overlap (they touch
      // disjoint graphs; the transposed parameter copies are written by the block of the last graph only).
      const int B = host.info.B, ss = fs.spec_small_first;
      const bool two = ss > 0 && ss < B, conc = two && knob_spec_small() >= 2;
      hipStream_t main_s = static_cast<hipStream_t>(hip_stream);
      SideStream* side = nullptr;
      std::unique_lock<std::mutex> side_lock;
      if (conc) {
        side = &side_stream();
        if (!side->ok) return fail(AGCN_EDEVICE, "%s: cannot create the side stream", who);
        side_lock = std::unique_lock<std::mutex>(side->mu);
        if (hipEventRecord(side->fork, main_s) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess)
          return fail(AGCN_EDEVICE, "%s: stream fork failed", who);
      }
      {
        int cg = 0, cs = 0, dg = 0;
        coop_plan(desc, host, fs, false, &cg, &cs, &dg);
        a.coop_slabs = cs;
        a.coop_wide = (cs > 0 && host.n[host.order[0]] > spec::COOP_MAX_N) ? 1 : 0;
        a.tp_first = std::max(cg, B - 32);   // (cg < B: the last graph of the schedule is never spread over members)
        size_t lds = spec_fwd_lds_bytes(host.n[host.order[cg]], desc->Fin, desc->Fout, spec::NW_LARGE);
        if (cs > 0) lds = std::max(lds, spec_coop_fwd_lds_bytes(host.n[host.order[0]], desc->Fin));
        // bf16 pipe (spec_bfpipe.h): room for the piece planes + parameter image of the largest member graph that fits with them
        // (a member decides for itself against the launch's LDS size: larger graphs keep the fp32 matrix instruction)
        // (off by default, errors.cpp knob_bfpipe: the phase gains what the image build and the larger launch LDS cost)
        a.bfp_fwd = (knob_bfpipe() & (a.feat_bf16 ? 1 : 2)) != 0 ? 1 : 0;
        if (a.bfp_fwd && cs > 0) {
          const size_t lim = (size_t)agcn_get_lds_limit();
          for (int g = 0; g < cg; ++g) {
            const size_t want = spec_coop_fwd_lds_bytes_bfp(host.n[host.order[g]], desc->Fin, a.feat_bf16 ? 1 : 3);
            if (want <= lim) { lds = std::max(lds, want); break; }     // (largest first)
          }
        }
        if (dg > 0) {   // the dg smallest graphs, two to a workgroup
          const size_t half = spec_fwd_lds_bytes(host.n[host.order[B - dg]], desc->Fin, desc->Fout, spec::NW_SMALL);
          a.dual_pairs = dg / 2;
          a.dual_stride = (int32_t)(half / sizeof(float));
          lds = std::max(lds, 2 * half);
        }
        rc = launch_spec_forward(a, cg, ss - cg - dg, lds, spec::NW_LARGE, conc ? side->s : main_s);
      }
      if (rc != AGCN_OK) return rc;
      if (ss < B)
        rc = launch_spec_forward(a, ss, B - ss,
                                 spec_fwd_lds_bytes(host.n[host.order[ss]], desc->Fin, desc->Fout, spec::NW_SMALL),
                                 spec::NW_SMALL, main_s);
      if (rc != AGCN_OK) return rc;
      if (conc && (hipEventRecord(side->join, side->s) != hipSuccess || hipStreamWaitEvent(main_s, side->join, 0) != hipSuccess))
        return fail(AGCN_EDEVICE, "%s: stream join failed", who);
      return AGCN_OK;
    }
    if (fs.gb_fwd)
      return launch_fused_forward(a, 0, host.info.B, fused_lds_bytes_gb(fs.thr, desc->Fin, desc->Fout, desc->K), true,
                                  hip_stream);
    return launch_fused_forward(a, 0, host.info.B, fused_lds_bytes(fs.thr, desc->Fin, desc->Fout, desc->K), false,
                                hip_stream);
  }
  if (fs.mode == 2) {
    // mixed: the few graphs too large for the fused kernel go through the slab-parallel pipeline on a side
    // stream (row-restricted packed GEMMs + the prefix of the slab table), the rest through the fused kernel
    SideStream& ss = side_stream();
    if (!ss.ok) return fail(AGCN_EDEVICE, "%s: cannot create the side stream", who);
    std::lock_guard<std::mutex> side_lock(ss.mu);
    hipStream_t main_s = static_cast<hipStream_t>(hip_stream);
    if (hipEventRecord(ss.fork, main_s) != hipSuccess || hipStreamWaitEvent(ss.s, ss.fork, 0) != hipSuccess)
      return fail(AGCN_EDEVICE, "%s: stream fork failed", who);
    RowsGemmArgs g;
    fill_rows(g, desc, host, dev);
    g.big_n = fs.thr; g.row_graph = dev.row_graph;
    g.Kd = desc->Fin; g.N = desc->Fin;
    g.A0 = batch->X; g.lda0 = desc->ldx; g.Bm = params->M_L; g.Out0 = a.sv_Z; g.ldo0 = desc->Fin;
    rc = launch_rows_gemm(g, RG_PROJECT, ss.s);
    if (rc != AGCN_OK) return rc;
    if (fs.big_pg) {
      // (the leading ones whose n x n tile does not fit LDS: the flavour with the tile in workspace)
      const int ng = std::min(n_glb, fs.n_big);
      if (ng > 0) rc = launch_sgc_forward(a, 0, ng, false, 4 * (32 + (size_t)((host.info.max_n + 3) & ~3)), ss.s);
      if (rc == AGCN_OK && ng < fs.n_big)
        rc = launch_sgc_forward(a, ng, fs.n_big - ng, true, sgc_lds_bytes(host.n[host.order[ng]], desc->Fin), ss.s);
    } else {
      SgcArgs ab = a;
      ab.n_slabs = fs.n_big_slabs;
      ab.slab_quad = 0;
      rc = launch_slab_forward(ab, ss.s);
    }
    if (rc != AGCN_OK) return rc;
    if (fs.big_pg) {   // (the slab kernels apply the filter themselves)
      fill_rows(g, desc, host, dev);
      g.big_n = fs.thr; g.row_graph = dev.row_graph;
      g.Kd = desc->Fin * desc->K; g.N = desc->Fout;
      g.A0 = batch->X; g.lda0 = desc->ldx; g.A1 = a.sv_X; g.a1_plane = (int64_t)host.info.rows * desc->Fin;
      g.Bm = params->weight; g.bias = params->bias; g.Out0 = out->Y; g.ldo0 = desc->ldy;
      rc = launch_rows_gemm(g, RG_FILTER, ss.s);
      if (rc != AGCN_OK) return rc;
    }
    if (fs.spec_fwd) a.tp_first = std::max(fs.n_big, host.info.B - 32);
    if (fs.spec_fwd)
      rc = launch_spec_forward(a, fs.n_big, host.info.B - fs.n_big,
                               spec_fwd_lds_bytes(fs.thr, desc->Fin, desc->Fout, spec::NW_LARGE), spec::NW_LARGE, main_s);
    else
      rc = launch_fused_forward(a, fs.n_big, host.info.B - fs.n_big,
                                fused_lds_bytes(fs.thr, desc->Fin, desc->Fout, desc->K), false, hip_stream);
    if (rc != AGCN_OK) return rc;
    if (hipEventRecord(ss.join, ss.s) != hipSuccess || hipStreamWaitEvent(main_s, ss.join, 0) != hipSuccess)
      return fail(AGCN_EDEVICE, "%s: stream join failed", who);
    return AGCN_OK;
  }
  // point clouds (cloud_kernels.hip): projection, similarity, degree, Laplacian + Chebyshev + filter in four launches
  if (cloud_candidate(desc, batch, host.info) && dpart && cloud_path(a, a.slab_quad != 0) &&
      (reinterpret_cast<uintptr_t>(out->Y) & 15) == 0) {
    if (desc->Fin != 128) return launch_cloud_forward(a, dpart, host.info.max_n, host.info.tile_elems, hip_stream);
    // 128 input features: Z = X M_L and the filter product as packed-row launches around the strip kernels
    const bool rw = rows_wide_ok(host.info.rows, desc->Fin, desc->Fout, desc->K, desc->ldx, desc->ldy);
    const bool rsp = simg && rows_split_ok(host.info.rows, desc->Fin, desc->Fout, desc->K, desc->ldx, desc->ldy);
    RowsGemmArgs g;
    if (rsp) {     // (rows_split.hip: bf16 matrix pipe at fp32 accuracy; one launch builds the images of M_L and W)
      rc = launch_rows_split_fwd_images(params->M_L, params->weight, desc->Fin, desc->Fout, simg, hip_stream);
      if (rc != AGCN_OK) return rc;
      rc = launch_rows_split_project(batch->X, desc->ldx, a.sv_Z, desc->Fin, host.info.rows, simg, hip_stream);
    } else if (rw) {
      rc = launch_rows_wide_project(batch->X, desc->ldx, params->M_L, a.sv_Z, host.info.rows, hip_stream);
    } else {
      fill_rows(g, desc, host, dev);
      g.Kd = desc->Fin; g.N = desc->Fin;
      g.A0 = batch->X; g.lda0 = desc->ldx; g.Bm = params->M_L; g.Out0 = a.sv_Z; g.ldo0 = desc->Fin;
      rc = launch_rows_gemm(g, RG_PROJECT, hip_stream);
    }
    if (rc != AGCN_OK) return rc;
    rc = launch_cloud_forward(a, dpart, host.info.max_n, host.info.tile_elems, hip_stream);
    if (rc != AGCN_OK) return rc;
    if (rsp)
      return launch_rows_split_filter(batch->X, desc->ldx, a.sv_X, desc->Fin, params->bias, desc->activation == AGCN_ACT_RELU ? 1 : 0,
                                      out->Y, desc->ldy, desc->Fin, desc->Fout, host.info.rows, simg, hip_stream);
    if (rw)
      return launch_rows_wide_filter(batch->X, desc->ldx, a.sv_X, params->weight, params->bias, desc->activation == AGCN_ACT_RELU ? 1 : 0,
                                     out->Y, desc->ldy, desc->Fout, host.info.rows, hip_stream);
    fill_rows(g, desc, host, dev);
    g.Kd = desc->Fin * desc->K; g.N = desc->Fout;
    g.A0 = batch->X; g.lda0 = desc->ldx; g.A1 = a.sv_X; g.a1_plane = (int64_t)host.info.rows * desc->Fin;
    g.Bm = params->weight; g.bias = params->bias; g.Out0 = out->Y; g.ldo0 = desc->ldy;
    return launch_rows_gemm(g, RG_FILTER, hip_stream);
  }
  // 1. Z = X . M_L over the packed rows (graphconv.py:164)
  RowsGemmArgs g;
  fill_rows(g, desc, host, dev);
  g.Kd = desc->Fin; g.N = desc->Fin;
  g.A0 = batch->X; g.lda0 = desc->ldx; g.Bm = params->M_L; g.Out0 = a.sv_Z; g.ldo0 = desc->Fin;
  rc = launch_rows_gemm(g, RG_PROJECT, hip_stream);
  if (rc != AGCN_OK) return rc;

  // 2. graph-local stages (similarity tile, learned Laplacian, Chebyshev terms)
  const int B = host.info.B;
  a.slab_graph = dev.slab_graph; a.slab_row = dev.slab_row; a.slab_desc = dev.slab_desc; a.n_slabs = host.info.n_slabs;
    a.slab_quad = slab_quad_ok(host);
  a.sv_r = reinterpret_cast<float*>(rb + rl.r);
  if (use_slabs(desc, batch, host.info)) {
    // slab_lap also applies the filter to its rows (Y = act([X_0|X_1] . W + b)): no packed FILTER launch
    return launch_slab_forward(a, hip_stream);
  } else {
  if (n_glb > 0) {
    const size_t lds = 4 * (32 + (size_t)((host.info.max_n + 3) & ~3));
    rc = launch_sgc_forward(a, 0, n_glb, false, lds, hip_stream);
    if (rc != AGCN_OK) return rc;
  }
  if (n_glb < B) {
    const size_t lds = sgc_lds_bytes(host.n[host.order[n_glb]], desc->Fin);
    rc = launch_sgc_forward(a, n_glb, B - n_glb, true, lds, hip_stream);
    if (rc != AGCN_OK) return rc;
  }
  }

  // 3. Y = act([X_0|X_1|..] . W + b), pad rows 0 (graphconv.py:238-251,117-123)
  fill_rows(g, desc, host, dev);
  g.Kd = desc->Fin * desc->K; g.N = desc->Fout;
  g.A0 = batch->X; g.lda0 = desc->ldx; g.A1 = a.sv_X; g.a1_plane = (int64_t)host.info.rows * desc->Fin;
  g.Bm = params->weight; g.bias = params->bias; g.Out0 = out->Y; g.ldo0 = desc->ldy;
  return launch_rows_gemm(g, RG_FILTER, hip_stream);
}

// stages: bit 0 = per-graph adjoints (dX, dL_prev, dZ, per-graph dalpha/dbeta partials),
//         bit 1 = parameter gradients (dweight, dbias, dM_L, dalpha, dbeta)
static int backward_impl(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch,
                         const AgcnParams* params, const AgcnOutputs* fwd, const float* dY,
                         const float* dL_sum, const AgcnGrads* grads, const void* reserve,
                         size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                         void* hip_stream, int stages, const char* who, void* param_stream = nullptr, void* fork_event = nullptr) {
  int rc = check_desc(desc, who);
  if (rc != AGCN_OK) return rc;
  rc = check_batch(desc, batch, params, who);
  if (rc != AGCN_OK) return rc;
  if (!fwd || !fwd->Y || !fwd->res_L || !fwd->res_W || !dY) return fail(AGCN_EINVAL, "%s: null forward tensor / dY", who);
  const bool reslap = desc->variant == AGCN_SGC_LL_RESLAP;
  const bool paper = desc->mode == AGCN_MODE_PAPER;
  if (reslap && !fwd->L_sum) return fail(AGCN_EINVAL, "%s: SGC_LL_Reslap needs L_sum", who);
  if (!grads || !grads->dX || !grads->dweight || !grads->dbias || !grads->dalpha)
    return fail(AGCN_EINVAL, "%s: null gradient output", who);
  if (paper && !grads->dM_L) return fail(AGCN_EINVAL, "%s: paper mode needs dM_L", who);
  const bool has_prev = reslap && batch->L_prev != nullptr;
  if (has_prev && (!grads->dbeta || !grads->dL_prev)) return fail(AGCN_EINVAL, "%s: L_prev given but dbeta/dL_prev null", who);

  PlanView host, dev;
  rc = plan_view(batch->plan_host, batch->plan_host, &host);
  if (rc != AGCN_OK) return rc;
  plan_view(batch->plan_host, batch->plan_dev, &dev);
  const ReserveLayout rl = reserve_layout(desc, host.info);
  if (!reserve || reserve_bytes < rl.total) return fail(AGCN_ENOSPACE, "%s: reserve too small", who);
  BwdWs w;
  const size_t need = carve_bwd(desc, host, workspace, &w);
  if (!workspace || workspace_bytes < need) return fail(AGCN_ENOSPACE, "%s: workspace too small (%zu < %zu)", who, workspace_bytes, need);

  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  SgcArgs a;
  fill_common(a, desc, batch, params, fwd, dev);
  char* rb = const_cast<char*>(static_cast<const char*>(reserve));
  a.sv_Z = reinterpret_cast<float*>(rb + rl.Z);
  a.sv_s = reinterpret_cast<float*>(rb + rl.s);
  a.sv_scal = reinterpret_cast<float*>(rb + rl.scal);
  a.sv_X = reinterpret_cast<float*>(rb + rl.X);
  a.sv_Lpre = reslap ? reinterpret_cast<float*>(rb + rl.Lpre) : nullptr;
  a.sv_MT = reinterpret_cast<float*>(rb + rl.MT);
  a.sv_WT = reinterpret_cast<float*>(rb + rl.WT);
  a.sv_WTI = (desc->K == 2 && desc->Fout == 64 && (knob_bfpipe() & (a.feat_bf16 ? 4 : 8)) &&
              (reinterpret_cast<uintptr_t>(params->weight) & 15) == 0) ? static_cast<void*>(rb + rl.WTI) : nullptr;
  a.sv_xch = reinterpret_cast<uint64_t*>(rb + rl.xch);
  a.dev_status = current_device_info().status_word;
  a.coop_ticks = agcn_get_coop_timeout_ticks();
  a.coop_slabs = 0; a.plain_count = 0; a.dual_pairs = 0; a.dual_stride = 0;
  a.tp_first = host.info.B - 1;
  a.dY = dY; a.dL_sum = dL_sum; a.dX = grads->dX; a.dL_prev = has_prev ? grads->dL_prev : nullptr;
  if (a.feat_bf16) {
    const FusedSplit fsb = fused_split(desc, batch, host);
    if (!bf16_served(desc, host, fsb))
      return fail(AGCN_EINVAL, "%s: bf16 features are read / written by the specialised kernels only (agcn_sgc_ll_dispatch_info)", who);
    // X as the forward widened it (the reserve's fp32 copy): the backward and the parameter gradients read fp32 rows
    a.sv_X0 = reinterpret_cast<float*>(rb + rl.X0);
    a.X = a.sv_X0; a.ldx = desc->Fin;
  }
  a.ws_g = w.g; a.ws_dZ = w.dZ; a.ws_pg = w.pg; a.ws_tile = w.tile; a.ws_buf = w.buf;

  // Reslap with K == 1 and no incoming dL_sum: L never reaches Y, so dL_prev == 0
  const bool need_dL = (reslap || paper) && (desc->K > 1 || dL_sum != nullptr);
  if ((stages & 1) && has_prev && !need_dL) {
    hipError_t e = hipMemsetAsync(grads->dL_prev, 0, (size_t)host.info.tile_elems * sizeof(float), s);
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: hipMemsetAsync: %s", who, hipGetErrorString(e));
  }

  const int B = host.info.B;
  const bool slabs = use_slabs(desc, batch, host.info);
  const FusedSplit fs = fused_split(desc, batch, host);
  const bool fused = fs.mode != 0;
  bool cloud_bwd = false, cloud_res = false;
  bool filter_bwd_zeroed = true;    // the general packed filter adjoint zeroes the one-pass kernel's arrival counters
  bool tail_done = false;           // the specialised backward launch carried the parameter gradients (spec_tail.h)
  // SGC_LL_Reslap on the point-cloud kernels: only behind the cloud forward (the per-graph forward leaves no r tiles); the
  // same answer in both stages of a split call
  // SGC_LL: either backward works behind either forward (both forwards leave Z, X_1, s and the r tiles)
  if (!reslap && !fused && w.WT && (w.dA || !paper) && cloud_shape(desc) && cloud_candidate(desc, batch, host.info)) {
    SgcArgs t = a;
    t.n_slabs = host.info.n_slabs;
    cloud_bwd = cloud_path(t, slab_quad_ok(host) != 0) && dL_sum == nullptr && (reinterpret_cast<uintptr_t>(grads->dX) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(dY) & 15) == 0;
  }
  if (reslap && !fused && w.WT && w.spart && cloud_candidate(desc, batch, host.info)) {
    SgcArgs t = a;
    t.n_slabs = host.info.n_slabs;
    // the forward's own predicate (forward_impl: cloud_candidate, cloud_shape -> workspace, cloud_path, Y aligned) ...
    const bool fwd_cloud = cloud_shape(desc) && cloud_path(t, slab_quad_ok(host) != 0) && (reinterpret_cast<uintptr_t>(fwd->Y) & 15) == 0;
    // ... and what the backward passes need on top of it.  A point-cloud forward leaves L_pre, the scales and the r tiles the
    // way ITS backward reads them; nothing tests the per-graph backward behind it, so a call that cannot take the
    // point-cloud backward is refused instead of silently pairing the two.
    cloud_res = fwd_cloud && (reinterpret_cast<uintptr_t>(grads->dX) & 15) == 0 && (reinterpret_cast<uintptr_t>(dY) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(dL_sum) & 15) == 0 &&
                (!has_prev || (reinterpret_cast<uintptr_t>(grads->dL_prev) & 15) == 0);
    if (fwd_cloud && !cloud_res)
      return fail(AGCN_EINVAL, "%s: this batch took the point-cloud forward kernels; their backward needs 16-byte aligned dX, dY, "
                  "dL_sum and dL_prev", who);
  }
  // arrival counters of the one-pass parameter-gradient kernel: zeroed by the first kernel of this call
  int32_t n_counters = 0;
  int32_t* counters = param_grad_counters(w.part, w.part_floats, host.info.rows, desc->Fin, desc->Fout, desc->K,
                                          paper ? 1 : 0, &n_counters);
  a.zero_ptr = counters; a.zero_count = n_counters;
  // ... or, where every graph of the call takes the specialised backward kernel (all of whose workgroups clear a slice), the
  // granules of the tagged hand-off (param_grad_tagged_kernel); a whole backward call only
  bool pg_tagged = false;
  if (stages == 3 && fused && fs.mode == 1 && fs.spec_bwd && fs.spec_bwd_first == 0) {
    int32_t n16 = 0;
    float* tags = param_grad_tag_region(w.part, w.part_floats, host.info.rows, desc->Fin, desc->Fout, desc->K, paper ? 1 : 0, &n16);
    if (tags != nullptr) {
      pg_tagged = true;
      a.ztag_ptr = tags; a.ztag_count16 = n16;
      a.zero_ptr = nullptr; a.zero_count = 0;
    }
  }
  // cooperative backward (spec_backward_coop.h)
  int coop_graphs = 0, coop_slabs = 0, dual_graphs = 0;
  coop_plan(desc, host, fs, true, &coop_graphs, &coop_slabs, &dual_graphs);
  if ((stages & 1) && fused) {
    a.slab_graph = dev.slab_graph; a.slab_row = dev.slab_row; a.slab_desc = dev.slab_desc; a.n_slabs = host.info.n_slabs;
    a.slab_quad = slab_quad_ok(host);
    a.sv_r = reinterpret_cast<float*>(rb + rl.r);
    a.ws_Gp = w.Gp;
    a.ws_dA = w.dA; a.ws_dAt = w.dAt; a.ws_dd = w.dd;
    SideStream* ss = nullptr;
    std::unique_lock<std::mutex> side_lock;
    hipStream_t main_s = static_cast<hipStream_t>(hip_stream);
    if (fs.mode == 2) {
      ss = &side_stream();
      if (!ss->ok) return fail(AGCN_EDEVICE, "%s: cannot create the side stream", who);
      side_lock = std::unique_lock<std::mutex>(ss->mu);
      if (hipEventRecord(ss->fork, main_s) != hipSuccess || hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess)
        return fail(AGCN_EDEVICE, "%s: stream fork failed", who);
      RowsGemmArgs g;
      fill_rows(g, desc, host, dev);
      g.big_n = fs.thr; g.row_graph = dev.row_graph;
      g.Kd = desc->Fout; g.N = desc->Fin * desc->K;
      g.A0 = dY; g.lda0 = desc->ldy; g.Yact = fwd->Y; g.ldy = desc->ldy; g.Bm = params->weight;
      g.Out0 = grads->dX; g.ldo0 = desc->ldx; g.Out1 = w.g; g.o1_plane = (int64_t)host.info.rows * desc->Fin;
      g.Gp = w.Gp;
      rc = launch_rows_gemm(g, RG_FILTER_BWD, ss->s);
      if (rc != AGCN_OK) return rc;
      if (fs.big_pg) {
        const int ng = std::min(w.n_glb, fs.n_big);
        if (ng > 0) rc = launch_sgc_backward(a, 0, ng, false, 4 * (32 + 2 * (size_t)((host.info.max_n + 3) & ~3)), ss->s);
        if (rc == AGCN_OK && ng < fs.n_big)
          rc = launch_sgc_backward(a, ng, fs.n_big - ng, true, sgc_lds_bytes_bwd(host.n[host.order[ng]], desc->Fin), ss->s);
      } else {
        SgcArgs ab = a;
        ab.n_slabs = fs.n_big_slabs;
        ab.pg_slab_base = B;
        ab.slab_quad = 0;
        rc = launch_slab_backward(ab, ss->s);
      }
      if (rc != AGCN_OK) return rc;
      if (paper && fs.big_pg) {   // (slab_bwd2 adds dZ . M_L^T to its rows itself)
        fill_rows(g, desc, host, dev);
        g.big_n = fs.thr; g.row_graph = dev.row_graph;
        g.Kd = desc->Fin; g.N = desc->Fin;
        g.A0 = w.dZ; g.lda0 = desc->Fin; g.Bm = params->M_L; g.Out0 = grads->dX; g.ldo0 = desc->ldx;
        rc = launch_rows_gemm(g, RG_PROJECT_BWD, ss->s);
        if (rc != AGCN_OK) return rc;
      }
    }
    if (fs.spec_bwd) {
      // graphs are scheduled largest first: the leading fs.spec_bwd_first ones are too large for the specialised
      // kernel's LDS plan and take the general fused kernel, the rest the specialised one (disjoint graphs)
      const int sf = fs.spec_bwd_first;
      if (sf > 0 && fs.mode == 1)
        rc = launch_fused_backward(a, 0, sf,
                                   fused_bwd_lds_bytes_gb(fs.thr, desc->Fin, desc->Fout, desc->K, fs.gb_bwd_w != 0,
                                                          fs.gb_bwd_m != 0),
                                   fs.gb_bwd_w != 0, fs.gb_bwd_m != 0, hip_stream);
      if (rc != AGCN_OK) return rc;
      const int ss = std::max(sf, fs.spec_small_first);
      const bool two = ss > sf && ss < B, conc = two && sf == 0 && knob_spec_small() >= 2 && fs.mode == 1;
      // one launch for every graph: the largest graphs are spread over one workgroup per 16-row tile (members: the
      // leading slabs of the slab table, which is in the same largest-first order), spec_backward_coop.h
      const int first_plain = coop_graphs;
      a.coop_slabs = coop_slabs;
      a.coop_wide = (coop_slabs > 0 && host.n[host.order[0]] > spec::COOP_MAX_N) ? 1 : 0;
      // parameter gradients as trailing workgroups of this launch (spec_tail.h): a whole backward call only
      tail_done = stages == 3 && !conc && !two && spec_tail_ok(desc, host, fs, coop_graphs, coop_slabs) && grads->dweight &&
                  grads->dbias && grads->dM_L && grads->dalpha;
      if (tail_done) {
        const int tiles = spec::tail_tiles(desc->Fin, desc->Fout, desc->K, true);
        const int S = spec::tail_splits(spec_tail_ksteps(host, coop_graphs, coop_slabs));
        tail_done = (size_t)(S - 1) * tiles * spec::TAIL_PSTRIDE <= w.part_floats &&     // (the partial tiles live in `part`)
                    coop_slabs + (B - coop_graphs) + S * tiles <= spec::TAIL_MAX_WGS;
        a.tail_splits = S; a.tail_tiles_per_split = tiles; a.tail_tiles = S * tiles;
        a.tail_part = w.part;
        static const int dbg = [] { const char* e = std::getenv("AGCN_TAIL_DEBUG"); return e ? std::atoi(e) : 0; }();
        a.tail_debug = dbg;
      }
      if (tail_done) {
        a.ztag_ptr = nullptr; a.ztag_count16 = 0; pg_tagged = false;
        a.tail = reinterpret_cast<uint32_t*>(rb + rl.xch) + spec::TAIL_OFF_FLOATS;
        a.tail_first = first_plain;
        a.tail_graphs = B - first_plain;
        a.g_dweight = grads->dweight; a.g_dbias = grads->dbias; a.g_dM = grads->dM_L; a.g_dalpha = grads->dalpha;
      }
      SideStream* side = nullptr;
      std::unique_lock<std::mutex> side_lock2;
      if (conc) {   // large graphs beside the small ones, as in the forward
        side = &side_stream();
        if (!side->ok) return fail(AGCN_EDEVICE, "%s: cannot create the side stream", who);
        side_lock2 = std::unique_lock<std::mutex>(side->mu);
        if (hipEventRecord(side->fork, main_s) != hipSuccess || hipStreamWaitEvent(side->s, side->fork, 0) != hipSuccess)
          return fail(AGCN_EDEVICE, "%s: stream fork failed", who);
      }
      {
        const int f0 = std::max(sf, first_plain);
        // phase G on the bf16 pipe (spec_bfpipe.h) where the forward left the W^T piece image: bf16 features one piece plane in
        // the place of the fp32 image; fp32 features THREE planes -- more LDS: only when every body of this launch still fits
        size_t lds = 0;
        for (int gnp = a.sv_WTI ? (a.feat_bf16 ? 1 : 3) : 0;; gnp = 0) {
          lds = f0 < B ? spec_bwd_lds_bytes(host.n[host.order[f0]], desc->Fin, desc->Fout, spec::NW_LARGE, gnp) : 0;
          if (coop_slabs > 0) lds = std::max(lds, spec_coop_lds_bytes(host.n[host.order[0]], desc->Fin, desc->Fout, gnp));
          if (dual_graphs > 0) {
            const size_t half = spec_bwd_lds_bytes(host.n[host.order[B - dual_graphs]], desc->Fin, desc->Fout, spec::NW_SMALL, gnp);
            a.dual_pairs = dual_graphs / 2;
            a.dual_stride = (int32_t)(half / sizeof(float));
            lds = std::max(lds, 2 * half);
          }
          a.g_np = gnp;
          if (gnp == 0 || lds <= (size_t)agcn_get_lds_limit()) break;
        }
        rc = launch_spec_backward(a, f0, ss - f0 - dual_graphs, lds, spec::NW_LARGE, conc ? side->s : main_s);
      }
      if (rc != AGCN_OK) return rc;
      if (ss < B)
        rc = launch_spec_backward(a, ss, B - ss,
                                  spec_bwd_lds_bytes(host.n[host.order[ss]], desc->Fin, desc->Fout, spec::NW_SMALL),
                                  spec::NW_SMALL, main_s);
      if (rc != AGCN_OK) return rc;
      if (conc && (hipEventRecord(side->join, side->s) != hipSuccess || hipStreamWaitEvent(main_s, side->join, 0) != hipSuccess))
        return fail(AGCN_EDEVICE, "%s: stream join failed", who);
    } else {
    rc = launch_fused_backward(a, fs.n_big, B - fs.n_big,
                               fused_bwd_lds_bytes_gb(fs.thr, desc->Fin, desc->Fout, desc->K, fs.gb_bwd_w != 0,
                                                      fs.gb_bwd_m != 0),
                               fs.gb_bwd_w != 0, fs.gb_bwd_m != 0, hip_stream);
    if (rc != AGCN_OK) return rc;
    }
    if (ss) {
      if (hipEventRecord(ss->join, ss->s) != hipSuccess || hipStreamWaitEvent(main_s, ss->join, 0) != hipSuccess)
        return fail(AGCN_EDEVICE, "%s: stream join failed", who);
    }
  } else if (stages & 1) {
    a.slab_graph = dev.slab_graph; a.slab_row = dev.slab_row; a.slab_desc = dev.slab_desc; a.n_slabs = host.info.n_slabs;
    a.slab_quad = slab_quad_ok(host);
    a.sv_r = reinterpret_cast<float*>(rb + rl.r);
    a.ws_dA = w.dA; a.ws_dAt = w.dAt; a.ws_dd = w.dd; a.ws_Gp = w.Gp;
    a.rs_img = w.simg;
    a.ws_P = w.P;
    // point clouds: either backward works behind either forward (both forwards leave Z, X_1, s and the r tiles)
    RowsGemmArgs g;
    if (cloud_res || cloud_bwd) {
      const bool wide = desc->Fin == 128;
      const bool rw = wide && w.WT != nullptr && (w.MT != nullptr || !paper) &&
                      rows_wide_ok(host.info.rows, desc->Fin, desc->Fout, desc->K, desc->ldx, desc->ldy);
      const bool rsp = rw && w.simg && rows_split_ok(host.info.rows, desc->Fin, desc->Fout, desc->K, desc->ldx, desc->ldy);
      if (rsp) {     // (rows_split.hip: one launch builds the images of W^T and M_L^T -- no transposes)
        rc = launch_rows_split_bwd_images(params->M_L, params->weight, desc->Fin, desc->Fout, paper, w.simg, hip_stream);
        if (rc != AGCN_OK) return rc;
        rc = launch_rows_split_filter_bwd(dY, fwd->Y, desc->ldy, desc->activation == AGCN_ACT_RELU ? 1 : 0, w.Gp, grads->dX, desc->ldx,
                                          w.g, desc->Fin, desc->Fout, host.info.rows, w.simg, hip_stream);
        if (rc != AGCN_OK) return rc;
        filter_bwd_zeroed = false;
      } else if (rw) {      // (rows_wide.hip; the parameter gradients then take the row-chunk partials: no arrival counters to zero)
        rc = launch_rows_wide_filter_bwd(dY, fwd->Y, desc->ldy, desc->activation == AGCN_ACT_RELU ? 1 : 0, params->weight, params->M_L,
                                         w.WT, w.MT, w.Gp, grads->dX, desc->ldx, w.g, desc->Fout, host.info.rows, hip_stream);
        if (rc != AGCN_OK) return rc;
        filter_bwd_zeroed = false;
      } else if (wide) {    // [g_0 | g_1] = (dY . act'(Y)) . W^T over the packed rows; g_0 -> dX, g_1 -> workspace
        fill_rows(g, desc, host, dev);
        g.Kd = desc->Fout; g.N = desc->Fin * desc->K;
        g.A0 = dY; g.lda0 = desc->ldy; g.Yact = fwd->Y; g.ldy = desc->ldy; g.Bm = params->weight;
        g.Out0 = grads->dX; g.ldo0 = desc->ldx; g.Out1 = w.g; g.o1_plane = (int64_t)host.info.rows * desc->Fin;
        g.Gp = w.Gp; g.zero_ptr = counters; g.zero_count = n_counters;
        rc = launch_rows_gemm(g, RG_FILTER_BWD, hip_stream);
        if (rc != AGCN_OK) return rc;
      }
      if (cloud_res) {
        a.pg_slab_base = 0;
        rc = launch_cloud_backward_reslap(a, nullptr, w.spart, hip_stream);
      } else {
        // the same predicate as the forward's: the cloud forward left W^T in the reserve (else it is built here, in workspace)
        const bool fwd_cloud = (reinterpret_cast<uintptr_t>(fwd->Y) & 15) == 0;
        rc = launch_cloud_backward(a, fwd_cloud ? nullptr : w.WT, hip_stream);
      }
      if (rc != AGCN_OK) return rc;
      if (rsp && paper) {
        rc = launch_rows_split_project_bwd(w.dZ, grads->dX, desc->ldx, desc->Fin, desc->Fout, host.info.rows, w.simg, hip_stream);
        if (rc != AGCN_OK) return rc;
      } else if (rw && paper) {
        rc = launch_rows_wide_project_bwd(w.dZ, w.MT, grads->dX, desc->ldx, host.info.rows, hip_stream);
        if (rc != AGCN_OK) return rc;
      } else if (wide && paper) {   // dX += dZ . M_L^T
        fill_rows(g, desc, host, dev);
        g.Kd = desc->Fin; g.N = desc->Fin;
        g.A0 = w.dZ; g.lda0 = desc->Fin; g.Bm = params->M_L; g.Out0 = grads->dX; g.ldo0 = desc->ldx;
        rc = launch_rows_gemm(g, RG_PROJECT_BWD, hip_stream);
        if (rc != AGCN_OK) return rc;
      }
    } else {
    // 1. [g_0|g_1|..] = (dY . act'(Y)) . W^T over the packed rows; g_0 -> dX, g_k -> workspace
    fill_rows(g, desc, host, dev);
    g.Kd = desc->Fout; g.N = desc->Fin * desc->K;
    g.A0 = dY; g.lda0 = desc->ldy; g.Yact = fwd->Y; g.ldy = desc->ldy; g.Bm = params->weight;
    g.Out0 = grads->dX; g.ldo0 = desc->ldx; g.Out1 = w.g; g.o1_plane = (int64_t)host.info.rows * desc->Fin;
    g.Gp = w.Gp; g.zero_ptr = counters; g.zero_count = n_counters;
    rc = launch_rows_gemm(g, RG_FILTER_BWD, hip_stream);
    if (rc != AGCN_OK) return rc;
    // 2. graph-local adjoints
    if (slabs) {
      rc = launch_slab_backward(a, hip_stream, host.info.max_n);
      if (rc != AGCN_OK) return rc;
    } else {
    if (w.n_glb > 0) {
      const size_t lds = 4 * (32 + 2 * (size_t)((host.info.max_n + 3) & ~3));
      rc = launch_sgc_backward(a, 0, w.n_glb, false, lds, hip_stream);
      if (rc != AGCN_OK) return rc;
    }
    if (w.n_glb < B) {
      const size_t lds = sgc_lds_bytes_bwd(host.n[host.order[w.n_glb]], desc->Fin);
      rc = launch_sgc_backward(a, w.n_glb, B - w.n_glb, true, lds, hip_stream);
      if (rc != AGCN_OK) return rc;
    }
    }
    // 3. paper mode: dX += dZ . M_L^T  (the slab path does it inside slab_bwd2)
    if (paper && !slabs) {
      fill_rows(g, desc, host, dev);
      g.Kd = desc->Fin; g.N = desc->Fin;
      g.A0 = w.dZ; g.lda0 = desc->Fin; g.Bm = params->M_L; g.Out0 = grads->dX; g.ldo0 = desc->ldx;
      rc = launch_rows_gemm(g, RG_PROJECT_BWD, hip_stream);
      if (rc != AGCN_OK) return rc;
    }
    }
  }
  if (!(stages & 2) || tail_done) return AGCN_OK;
  // agcn_sgc_ll_backward_forked: the parameter gradients go to `param_stream`, behind everything this call has enqueued on
  // `hip_stream` so far (one event record + one wait: graph edges under stream capture) -- the caller's stream is free for the
  // next layer's backward while they run; the caller joins the two streams once, at the end of its step
  if (param_stream != nullptr && param_stream != hip_stream) {
    hipEvent_t ev = static_cast<hipEvent_t>(fork_event);
    if (ev == nullptr) return fail(AGCN_EINVAL, "%s: param_stream without a fork event", who);
    if (hipEventRecord(ev, static_cast<hipStream_t>(hip_stream)) != hipSuccess ||
        hipStreamWaitEvent(static_cast<hipStream_t>(param_stream), ev, 0) != hipSuccess)
      return fail(AGCN_EDEVICE, "%s: stream fork failed", who);
    hip_stream = param_stream;
  }

  ParamGradArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.rows = host.info.rows; pa.Fin = desc->Fin; pa.Fout = desc->Fout; pa.K = desc->K;
  pa.ldx = a.ldx; pa.paper = paper ? 1 : 0;
  pa.n_chunks = w.n_chunks;
  pa.chunk_rows = ((host.info.rows + w.n_chunks - 1) / w.n_chunks + 15) / 16 * 16;
  pa.B = fused ? B + fs.n_big_slabs + coop_slabs : ((slabs || cloud_res || cloud_bwd) ? host.info.n_slabs : B); pa.part_stride = w.part_stride;
  pa.X = a.X; pa.sv_X = a.sv_X; pa.Gp = w.Gp; pa.dZ = w.dZ; pa.part = w.part; pa.pg = w.pg;
  pa.part_floats = w.part_floats;
  pa.counters_zeroed = ((stages & 1) && !pg_tagged && ((!cloud_bwd && !cloud_res) || (desc->Fin == 128 && filter_bwd_zeroed))) ? 1 : 0;
  pa.tags_zeroed = pg_tagged ? 1 : 0;
  pa.dev_status = a.dev_status; pa.ticks = a.coop_ticks;
  pa.dweight = grads->dweight; pa.dbias = grads->dbias; pa.dM_L = paper ? grads->dM_L : nullptr;
  pa.dalpha = grads->dalpha; pa.dbeta = has_prev ? grads->dbeta : nullptr;
  // point-cloud sizes: row-chunk partials with 16-byte operand loads + the fixed-order reduce (cloud_kernels.hip)
  static const int cloud_pg = [] { const char* e = std::getenv("AGCN_CLOUD_PG"); return e ? std::atoi(e) : 1; }();
  if (cloud_pg && !fused && (slabs || cloud_res || cloud_bwd) && w.WT && cloud_param_grads_ok(pa) && (size_t)pa.n_chunks * pa.part_stride <= w.part_floats &&
      (reinterpret_cast<uintptr_t>(batch->X) & 15) == 0) {
    return launch_cloud_param_partials(pa, hip_stream);   // (partials + the fixed-order reduce)
  }
  return launch_param_grads(pa, hip_stream);
}

extern "C" int agcn_sgc_ll_backward(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch,
                                    const AgcnParams* params, const AgcnOutputs* fwd, const float* dY,
                                    const float* dL_sum, const AgcnGrads* grads, const void* reserve,
                                    size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                                    void* hip_stream) {
  KnobScope knobs(desc ? desc->flags : 0u);
  return backward_impl(desc, batch, params, fwd, dY, dL_sum, grads, reserve, reserve_bytes, workspace,
                       workspace_bytes, hip_stream, 3, "agcn_sgc_ll_backward");
}

extern "C" int agcn_sgc_ll_backward_forked(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch,
                                           const AgcnParams* params, const AgcnOutputs* fwd, const float* dY,
                                           const float* dL_sum, const AgcnGrads* grads, const void* reserve,
                                           size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                                           void* hip_stream, void* param_stream, void* fork_event) {
  KnobScope knobs(desc ? desc->flags : 0u);
  return backward_impl(desc, batch, params, fwd, dY, dL_sum, grads, reserve, reserve_bytes, workspace,
                       workspace_bytes, hip_stream, 3, "agcn_sgc_ll_backward_forked", param_stream, fork_event);
}

extern "C" int agcn_sgc_ll_backward_nodes(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch,
                                          const AgcnParams* params, const AgcnOutputs* fwd, const float* dY,
                                          const float* dL_sum, const AgcnGrads* grads, const void* reserve,
                                          size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                                          void* hip_stream) {
  KnobScope knobs(desc ? desc->flags : 0u);
  return backward_impl(desc, batch, params, fwd, dY, dL_sum, grads, reserve, reserve_bytes, workspace,
                       workspace_bytes, hip_stream, 1, "agcn_sgc_ll_backward_nodes");
}

extern "C" int agcn_sgc_ll_backward_params(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch,
                                           const AgcnParams* params, const AgcnOutputs* fwd, const float* dY,
                                           const float* dL_sum, const AgcnGrads* grads, const void* reserve,
                                           size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                                           void* hip_stream) {
  KnobScope knobs(desc ? desc->flags : 0u);
  return backward_impl(desc, batch, params, fwd, dY, dL_sum, grads, reserve, reserve_bytes, workspace,
                       workspace_bytes, hip_stream, 2, "agcn_sgc_ll_backward_params");
}
