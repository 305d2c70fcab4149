// Host-side packed-batch ingest: the step BEFORE the hot path.
// The reference rebuilds its feed every step in Python (models/tf_modules/graph_topology.py:84-135:
// every graph zero-padded to max_atom, every CSR Laplacian densified, ~28 MB of mostly zeros per
// 256-graph batch) from graph objects handed out by DiskDataset.iterbatches
// (utils/dataset/diskdataset.py:239-295).  Here the dataset is laid out ONCE as a "graph store"
// (concatenated node features + one CSR over all rows, float32 values, per-graph symmetry flags) and
// a batch is a gather of contiguous segments into caller-owned (pinned) staging buffers: memcpy-bound,
// no per-graph Python, no densification.  Pure C++, no HIP calls: testable without a GPU.
#include "agcn_internal.h"

#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

namespace {
int check_store(const AgcnGraphStore* st, const char* who) {
  using agcn::fail;
  if (!st) return fail(AGCN_EINVAL, "%s: null store", who);
  if (st->n_graphs < 1 || st->F < 1) return fail(AGCN_EINVAL, "%s: empty store", who);
  if (!st->n || !st->row_off || !st->X || !st->rowptr || !st->col || !st->val)
    return fail(AGCN_EINVAL, "%s: null store array", who);
  return AGCN_OK;
}
}  // namespace

extern "C" int agcn_store_batch_sizes(const AgcnGraphStore* st, const int64_t* ids, int32_t B, int64_t* rows,
                                      int64_t* nnz, int32_t* max_n) {
  const char* who = "agcn_store_batch_sizes";
  int rc = check_store(st, who);
  if (rc != AGCN_OK) return rc;
  if (!ids || B < 1 || !rows || !nnz) return agcn::fail(AGCN_EINVAL, "%s: bad arguments", who);
  int64_t r = 0, z = 0;
  int32_t mx = 0;
  for (int32_t b = 0; b < B; ++b) {
    const int64_t g = ids[b];
    if (g < 0 || g >= st->n_graphs) return agcn::fail(AGCN_EINVAL, "%s: graph id %lld out of range", who, (long long)g);
    const int64_t r0 = st->row_off[g], r1 = st->row_off[g + 1];
    r += r1 - r0;
    z += st->rowptr[r1] - st->rowptr[r0];
    mx = std::max(mx, st->n[g]);
  }
  *rows = r; *nnz = z;
  if (max_n) *max_n = mx;
  return AGCN_OK;
}

extern "C" int agcn_store_pack_batch(const AgcnGraphStore* st, const int64_t* ids, int32_t B, int32_t* n_out,
                                     float* X_out, int32_t ldx_out, int32_t* rowptr_out, int32_t* col_out,
                                     float* val_out, int32_t* symmetric_out, int32_t n_threads) {
  const char* who = "agcn_store_pack_batch";
  int rc = check_store(st, who);
  if (rc != AGCN_OK) return rc;
  if (!ids || B < 1 || !n_out || !X_out || !rowptr_out || !col_out || !val_out)
    return agcn::fail(AGCN_EINVAL, "%s: bad arguments", who);
  if (ldx_out < st->F) return agcn::fail(AGCN_EINVAL, "%s: ldx_out smaller than F", who);
  // pass 1: destination offsets of every graph
  std::vector<int64_t> drow(B + 1), dnnz(B + 1);
  drow[0] = dnnz[0] = 0;
  int sym = 1;
  for (int32_t b = 0; b < B; ++b) {
    const int64_t g = ids[b];
    if (g < 0 || g >= st->n_graphs) return agcn::fail(AGCN_EINVAL, "%s: graph id %lld out of range", who, (long long)g);
    const int64_t r0 = st->row_off[g], r1 = st->row_off[g + 1];
    drow[b + 1] = drow[b] + (r1 - r0);
    dnnz[b + 1] = dnnz[b] + (st->rowptr[r1] - st->rowptr[r0]);
    n_out[b] = st->n[g];
    if (st->symmetric && !st->symmetric[g]) sym = 0;
  }
  if (dnnz[B] > INT32_MAX || drow[B] > INT32_MAX) return agcn::fail(AGCN_EINVAL, "%s: batch too large for int32 offsets", who);
  if (symmetric_out) *symmetric_out = st->symmetric ? sym : 0;
  rowptr_out[0] = 0;
  // pass 2: gather the segments (graphs are independent: split over threads for large batches)
  auto work = [&](int32_t b0, int32_t b1) {
    const int32_t F = st->F;
    for (int32_t b = b0; b < b1; ++b) {
      const int64_t g = ids[b];
      const int64_t r0 = st->row_off[g], nr = st->row_off[g + 1] - r0;
      const int64_t z0 = st->rowptr[r0], nz = st->rowptr[r0 + nr] - z0;
      if (ldx_out == F) {
        std::memcpy(X_out + drow[b] * F, st->X + r0 * F, sizeof(float) * (size_t)(nr * F));
      } else {
        for (int64_t i = 0; i < nr; ++i)
          std::memcpy(X_out + (drow[b] + i) * ldx_out, st->X + (r0 + i) * F, sizeof(float) * (size_t)F);
      }
      const int64_t shift = dnnz[b] - z0;
      for (int64_t i = 1; i <= nr; ++i) rowptr_out[drow[b] + i] = (int32_t)(st->rowptr[r0 + i] + shift);
      std::memcpy(col_out + dnnz[b], st->col + z0, sizeof(int32_t) * (size_t)nz);
      std::memcpy(val_out + dnnz[b], st->val + z0, sizeof(float) * (size_t)nz);
    }
  };
  int nt = n_threads > 0 ? n_threads : 1;
  if (nt > B) nt = B;
  if (nt <= 1) {
    work(0, B);
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, (int32_t)((int64_t)B * t / nt), (int32_t)((int64_t)B * (t + 1) / nt));
    for (auto& x : th) x.join();
  }
  return AGCN_OK;
}
