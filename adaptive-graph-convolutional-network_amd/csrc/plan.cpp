// Host-side plan builder: the integer bookkeeping of one ragged batch.
// Replaces the per-graph slice tensors of the reference feed
// (models/tf_modules/graph_topology.py:84-135: data_slice rows [n,-1],
// lap_slice rows [n,n]) by offsets into packed (or padded) device arrays plus
// a largest-first schedule.  Pure C++, no HIP calls: testable without a GPU.
#include "agcn_internal.h"

#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

namespace {
constexpr uint64_t kMagic = 0x4147434E504C4E31ull;  // "AGCNPLN1"

inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

struct Layout {
  int64_t off_n, off_row_off, off_tile_off, off_order, off_wtile_off, off_slab_graph, off_slab_row, off_row_graph,
      off_graph_desc, off_slab_desc, total;
};

// max_slabs: upper bound used for sizing (sum of ceil(n/16) <= N1/16 + B, but N1 is unknown when
// only B is given; the blob reserves room for B + B * AGCN_MAX_NODES / 16 ... far too much).  The
// slab tables are therefore sized from the ACTUAL node counts: agcn_plan_bytes(B) returns the size
// for the worst case of the plan WITHOUT slab tables plus a fixed per-graph allowance, and
// agcn_plan_build fails with AGCN_ENOSPACE if the batch needs more (see agcn_plan_bytes_for).
Layout layout_for(int32_t B, int64_t n_slabs, int64_t rows) {
  Layout l;
  int64_t o = align_up((int64_t)sizeof(uint64_t) + (int64_t)sizeof(AgcnPlanInfo), 16);
  l.off_n = o;           o = align_up(o + 4ll * B, 16);
  l.off_row_off = o;     o = align_up(o + 4ll * (B + 1), 16);
  l.off_tile_off = o;    o = align_up(o + 8ll * (B + 1), 16);
  l.off_order = o;       o = align_up(o + 4ll * B, 16);
  l.off_wtile_off = o;   o = align_up(o + 8ll * (B + 1), 16);
  l.off_slab_graph = o;  o = align_up(o + 4ll * n_slabs, 16);
  l.off_slab_row = o;    o = align_up(o + 4ll * n_slabs, 16);
  l.off_row_graph = o;   o = align_up(o + 4ll * rows, 16);
  l.off_graph_desc = o;  o = align_up(o + (int64_t)sizeof(AgcnGraphDesc) * B, 16);
  l.off_slab_desc = o;   o = align_up(o + (int64_t)sizeof(AgcnSlabDesc) * n_slabs, 16);
  l.total = o;
  return l;
}

int64_t slabs_of(int32_t n) { return (n + AGCN_SLAB_ROWS - 1) / AGCN_SLAB_ROWS; }
}  // namespace

extern "C" size_t agcn_plan_bytes(int32_t B) {
  // worst case: every graph has AGCN_MAX_NODES nodes
  if (B < 0) return 0;
  return (size_t)layout_for(B, (int64_t)B * slabs_of(AGCN_MAX_NODES), (int64_t)B * AGCN_MAX_NODES).total;
}

extern "C" size_t agcn_plan_bytes_for(const int32_t* n, int32_t B, int32_t row_stride) {
  if (B < 0 || !n) return 0;
  int64_t ns = 0, rows = 0;
  for (int32_t b = 0; b < B; ++b) {
    const int32_t nb = n[b] > 0 ? n[b] : 1;
    ns += slabs_of(nb);
    rows += row_stride > 0 ? row_stride : nb;
  }
  return (size_t)layout_for(B, ns, rows).total;
}

extern "C" int agcn_plan_build(const int32_t* n, int32_t B, int32_t row_stride, void* plan_host,
                               size_t plan_bytes) {
  if (!n || !plan_host || B < 1 || row_stride < 0) return agcn::fail(AGCN_EINVAL, "agcn_plan_build: bad argument");
  int64_t n_slabs = 0;
  for (int32_t b = 0; b < B; ++b) {
    if (n[b] < 1) return agcn::fail(AGCN_EINVAL, "agcn_plan_build: graph with < 1 node");
    if (n[b] > AGCN_MAX_NODES) return agcn::fail(AGCN_EINVAL, "agcn_plan_build: graph larger than AGCN_MAX_NODES");
    n_slabs += slabs_of(n[b]);
  }
  int64_t rows_total = 0;
  for (int32_t b = 0; b < B; ++b) rows_total += row_stride > 0 ? row_stride : n[b];
  const Layout l = layout_for(B, n_slabs, rows_total);
  if (plan_bytes < (size_t)l.total) return agcn::fail(AGCN_ENOSPACE, "agcn_plan_build: plan buffer too small");
  char* base = static_cast<char*>(plan_host);
  std::memset(base, 0, (size_t)l.total);
  int32_t* pn = reinterpret_cast<int32_t*>(base + l.off_n);
  int32_t* row_off = reinterpret_cast<int32_t*>(base + l.off_row_off);
  int64_t* tile_off = reinterpret_cast<int64_t*>(base + l.off_tile_off);
  int32_t* order = reinterpret_cast<int32_t*>(base + l.off_order);
  int64_t* wtile_off = reinterpret_cast<int64_t*>(base + l.off_wtile_off);
  int32_t* slab_graph = reinterpret_cast<int32_t*>(base + l.off_slab_graph);
  int32_t* slab_row = reinterpret_cast<int32_t*>(base + l.off_slab_row);
  int32_t* row_graph = reinterpret_cast<int32_t*>(base + l.off_row_graph);

  int64_t N1 = 0, N2 = 0, rows = 0, tiles = 0, wtiles = 0;
  int32_t max_n = 0;
  for (int32_t b = 0; b < B; ++b) {
    const int32_t nb = n[b];
    if (nb < 1) return agcn::fail(AGCN_EINVAL, "agcn_plan_build: graph with < 1 node");
    if (nb > AGCN_MAX_NODES) return agcn::fail(AGCN_EINVAL, "agcn_plan_build: graph larger than AGCN_MAX_NODES");
    if (row_stride > 0 && nb > row_stride)
      return agcn::fail(AGCN_EINVAL, "agcn_plan_build: graph larger than row_stride (max_atom)");
    pn[b] = nb;
    row_off[b] = (int32_t)rows;
    for (int64_t rr = rows; rr < rows + (row_stride > 0 ? row_stride : nb); ++rr) row_graph[rr] = b;
    tile_off[b] = tiles;
    wtile_off[b] = wtiles;
    rows += row_stride > 0 ? row_stride : nb;
    N1 += nb;
    N2 += (int64_t)nb * nb;
    tiles += align_up((int64_t)nb * nb, 4);
    wtiles += align_up((int64_t)((nb + 3) & ~3) * agcn::ld_frag((nb + 3) & ~3), 4);
    max_n = std::max(max_n, nb);
    if (rows > 0x7fffffffll / 2) return agcn::fail(AGCN_EINVAL, "agcn_plan_build: batch too large for 32-bit rows");
  }
  row_off[B] = (int32_t)rows;
  tile_off[B] = tiles;
  wtile_off[B] = wtiles;
  std::iota(order, order + B, 0);
  // largest graph first (longest-processing-time order); stable => deterministic
  std::stable_sort(order, order + B, [&](int32_t a, int32_t b) { return pn[a] > pn[b]; });
  // slabs in the same largest-first order (a slab's cost grows with the width of its graph)
  {
    int64_t s = 0;
    for (int32_t i = 0; i < B; ++i) {
      const int32_t g = order[i];
      for (int32_t r = 0; r < pn[g]; r += AGCN_SLAB_ROWS) { slab_graph[s] = g; slab_row[s] = r; ++s; }
    }
  }

  // one 32-byte record per schedule position: everything a per-graph workgroup needs to find its graph in ONE load
  {
    AgcnGraphDesc* gd = reinterpret_cast<AgcnGraphDesc*>(base + l.off_graph_desc);
    for (int32_t i = 0; i < B; ++i) {
      const int32_t g = order[i];
      gd[i].graph = g; gd[i].n = pn[g]; gd[i].row0 = row_off[g]; gd[i].reserved = 0;
      gd[i].tile_off = tile_off[g]; gd[i].tile_end = tile_off[g + 1];
    }
  }

  // ... and one per slab, for the workgroups that own a single slab of a graph
  {
    AgcnSlabDesc* sd = reinterpret_cast<AgcnSlabDesc*>(base + l.off_slab_desc);
    for (int64_t s = 0; s < n_slabs; ++s) {
      const int32_t g = slab_graph[s];
      sd[s].graph = g; sd[s].row = slab_row[s]; sd[s].n = pn[g]; sd[s].row0 = row_off[g];
      sd[s].tile_off = tile_off[g]; sd[s].tile_end = tile_off[g + 1];
    }
  }

  AgcnPlanInfo info;
  std::memset(&info, 0, sizeof(info));
  info.B = B;
  info.N1 = (int32_t)N1;
  info.rows = (int32_t)rows;
  info.max_n = max_n;
  info.N2 = N2;
  info.tile_elems = tiles;
  info.plan_bytes = l.total;
  info.off_n = l.off_n;
  info.off_row_off = l.off_row_off;
  info.off_tile_off = l.off_tile_off;
  info.off_order = l.off_order;
  info.off_wtile_off = l.off_wtile_off;
  info.n_slabs = (int32_t)n_slabs;
  info.off_slab_graph = l.off_slab_graph;
  info.off_slab_row = l.off_slab_row;
  info.off_row_graph = l.off_row_graph;
  info.off_graph_desc = l.off_graph_desc;
  info.off_slab_desc = l.off_slab_desc;
  std::memcpy(base, &kMagic, sizeof(kMagic));
  std::memcpy(base + sizeof(kMagic), &info, sizeof(info));
  return AGCN_OK;
}

extern "C" int agcn_plan_info(const void* plan_host, AgcnPlanInfo* info) {
  if (!plan_host || !info) return agcn::fail(AGCN_EINVAL, "agcn_plan_info: null argument");
  uint64_t magic;
  std::memcpy(&magic, plan_host, sizeof(magic));
  if (magic != kMagic) return agcn::fail(AGCN_EPLAN, "agcn_plan_info: bad magic (not a plan blob)");
  std::memcpy(info, static_cast<const char*>(plan_host) + sizeof(kMagic), sizeof(*info));
  if (info->B < 1 || info->n_slabs < info->B || info->plan_bytes != layout_for(info->B, info->n_slabs, info->rows).total)
    return agcn::fail(AGCN_EPLAN, "agcn_plan_info: corrupt header");
  return AGCN_OK;
}
