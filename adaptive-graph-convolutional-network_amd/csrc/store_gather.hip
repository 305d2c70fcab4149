// Batches out of a graph store that is RESIDENT in HBM (include/agcn.h, "graph store resident in HBM").
// The reference re-reads its dataset shards from disk and rebuilds the feed on the host every step
// (utils/dataset/diskdataset.py:239-295 -> models/tf_modules/graph_topology.py:84-135); with 288 GB of HBM the
// whole dataset lives on the device, and a batch is one small table upload + one gather launch.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <vector>

#include "agcn_internal.h"

namespace agcn {
namespace {

struct GatherRec {      // one per graph of the batch, behind the plan in the stage blob
  int64_t src_row;      // first row of the graph in the store
  int64_t src_nnz;      // rowptr[src_row]
  int32_t dst_nnz;      // first CSR entry of the graph in the batch
  int32_t pad;
};

struct GatherArgs {
  const int32_t* n;         // plan arrays and gather table: in the DEVICE blob, or -- zero-copy staging -- in the pinned
  const int32_t* row_off;   // host blob, read over the host link by the gather workgroups themselves
  const GatherRec* rec;
  const uint2* blob_src;    // zero-copy staging: the pinned host blob (device-visible address) ...
  uint2* blob_dst;          // ... copied into the device blob by this launch (a slice per workgroup); else null
  int32_t blob_vec;         // 8-byte words of the blob (plan 16-byte aligned + 24-byte records: a multiple of 8 bytes)
  const void* X;            // float or bf16 (x_bf16) rows of the store
  const int64_t* rowptr;
  const int32_t* col;
  const float* val;
  void* X_out;              // float rows, or bf16 rows (out_bf16)
  int32_t* rowptr_out;
  int32_t* col_out;
  float* val_out;
  int32_t F, ldx_out, x_bf16, out_bf16;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float v) {      // round to nearest even, NaN stays NaN (as spec_kernels.h: bf16_store)
  uint32_t u = __float_as_uint(v);
  u = ((u & 0x7fffffffu) > 0x7f800000u) ? (u | 0x00400000u) : u + 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// one workgroup per graph: contiguous segments in, contiguous segments out
__global__ __launch_bounds__(256) void store_gather_kernel(GatherArgs a) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = a.n[b], r0 = a.row_off[b];
  const GatherRec rc = a.rec[b];
  const int F = a.F;
  if (a.blob_dst) {   // this workgroup's slice of the plan blob: host -> device (the layer kernels read the device copy)
    const int per = (a.blob_vec + (int)gridDim.x - 1) / (int)gridDim.x;
    for (int i = b * per + tid; i < min((b + 1) * per, a.blob_vec); i += 256) a.blob_dst[i] = a.blob_src[i];
  }
  // (features kept in bf16 at rest are widened here: the layer computes in fp32, include/agcn.h agcn_cast_rows)
  const float* xs = static_cast<const float*>(a.X) + rc.src_row * F;
  const uint16_t* xh = static_cast<const uint16_t*>(a.X) + rc.src_row * F;
  const bool h = a.x_bf16 != 0;   // (uniform)
  if (a.out_bf16) {               // (uniform) the staged rows stay bf16: the layer's kernels read them themselves (feat_dtype = bf16)
    uint16_t* xo = static_cast<uint16_t*>(a.X_out);
    if (a.ldx_out == F) {
      uint16_t* xd = xo + (int64_t)r0 * F;
      for (int i = tid; i < n * F; i += 256) xd[i] = h ? xh[i] : f32_to_bf16(xs[i]);
    } else {
      for (int i = tid; i < n * F; i += 256) {
        const int r = i / F, f = i - r * F;
        xo[(int64_t)(r0 + r) * a.ldx_out + f] = h ? xh[i] : f32_to_bf16(xs[i]);
      }
    }
  } else if (a.ldx_out == F) {
    float* xd = static_cast<float*>(a.X_out) + (int64_t)r0 * F;
    for (int i = tid; i < n * F; i += 256) xd[i] = h ? bf16_to_f32(xh[i]) : xs[i];
  } else {
    float* xo = static_cast<float*>(a.X_out);
    for (int i = tid; i < n * F; i += 256) {
      const int r = i / F, f = i - r * F;
      xo[(int64_t)(r0 + r) * a.ldx_out + f] = h ? bf16_to_f32(xh[i]) : xs[i];
    }
  }
  const int64_t shift = (int64_t)rc.dst_nnz - rc.src_nnz;
  // entry r0 + n is the next graph's first (same value from both) or the batch's total
  for (int i = tid; i <= n; i += 256) a.rowptr_out[r0 + i] = (int32_t)(a.rowptr[rc.src_row + i] + shift);
  const int nz = (int)(a.rowptr[rc.src_row + n] - rc.src_nnz);
  for (int i = tid; i < nz; i += 256) {
    a.col_out[rc.dst_nnz + i] = a.col[rc.src_nnz + i];
    a.val_out[rc.dst_nnz + i] = a.val[rc.src_nnz + i];
  }
}

inline size_t align16(size_t v) { return (v + 15) / 16 * 16; }

}  // namespace
}  // namespace agcn

extern "C" size_t agcn_store_stage_bytes(int32_t B, int64_t rows) {
  if (B < 1 || rows < B) return 0;
  // plan of B graphs over `rows` packed rows: slabs <= rows / 16 + B; sized through the plan's own layout function
  // with the worst split of the rows (every graph one slab more than its share)
  std::vector<int32_t> n((size_t)B, (int32_t)(rows / B));
  n[0] += (int32_t)(rows - (rows / B) * B);
  size_t plan = agcn_plan_bytes_for(n.data(), B, 0);
  // slab tables: 8 bytes (slab_graph, slab_row) + one AgcnSlabDesc per slab; uneven graphs have at most B more slabs than the even split
  plan += (8 + sizeof(AgcnSlabDesc)) * (size_t)B + 64;
  return agcn::align16(plan) + sizeof(agcn::GatherRec) * (size_t)B;
}

extern "C" int agcn_store_stage_batch(const AgcnResidentStore* st, const int64_t* ids, int32_t B, void* stage_host,
                                      void* stage_dev, size_t stage_cap, void* X_out, int32_t ldx_out,
                                      int64_t rows_cap, int32_t* rowptr_out, int32_t* col_out, float* val_out,
                                      int64_t nnz_cap, AgcnStagedBatch* info, void* hip_stream) {
  using agcn::fail;
  const char* who = "agcn_store_stage_batch";
  if (!st || !ids || B < 1 || !stage_host || !stage_dev || !X_out || !rowptr_out || !col_out || !val_out || !info)
    return fail(AGCN_EINVAL, "%s: null argument", who);
  const AgcnGraphStore& h = st->host;
  if (h.n_graphs < 1 || h.F < 1 || !h.n || !h.row_off || !h.rowptr)
    return fail(AGCN_EINVAL, "%s: incomplete host index", who);
  if (!st->X_dev || !st->rowptr_dev || !st->col_dev || !st->val_dev)
    return fail(AGCN_EINVAL, "%s: store is not resident (null device array)", who);
  if (ldx_out < h.F) return fail(AGCN_EINVAL, "%s: ldx_out smaller than F", who);
  static thread_local std::vector<int32_t> n;
  n.resize((size_t)B);
  int64_t rows = 0, nnz = 0;
  int32_t mx = 0, sym = h.symmetric ? 1 : 0;
  for (int32_t b = 0; b < B; ++b) {
    const int64_t g = ids[b];
    if (g < 0 || g >= h.n_graphs) return fail(AGCN_EINVAL, "%s: graph id %lld out of range", who, (long long)g);
    n[b] = h.n[g];
    rows += h.n[g];
    nnz += h.rowptr[h.row_off[g + 1]] - h.rowptr[h.row_off[g]];
    mx = n[b] > mx ? n[b] : mx;
    if (h.symmetric && !h.symmetric[g]) sym = 0;
  }
  if (rows > rows_cap || nnz > nnz_cap)
    return fail(AGCN_ENOSPACE, "%s: batch (%lld rows, %lld entries) exceeds the output buffers (%lld, %lld)", who,
                (long long)rows, (long long)nnz, (long long)rows_cap, (long long)nnz_cap);
  if (nnz > INT32_MAX) return fail(AGCN_EINVAL, "%s: batch too large for int32 offsets", who);
  const size_t pbytes = agcn_plan_bytes_for(n.data(), B, 0);
  const size_t tab_off = agcn::align16(pbytes);
  const size_t total = tab_off + sizeof(agcn::GatherRec) * (size_t)B;
  if (total > stage_cap) return fail(AGCN_ENOSPACE, "%s: stage blob needs %zu bytes, %zu given", who, total, stage_cap);
  int rc = agcn_plan_build(n.data(), B, 0, stage_host, pbytes);
  if (rc != AGCN_OK) return rc;
  agcn::GatherRec* rec = reinterpret_cast<agcn::GatherRec*>(static_cast<char*>(stage_host) + tab_off);
  int64_t z = 0;
  for (int32_t b = 0; b < B; ++b) {
    const int64_t g = ids[b], r0 = h.row_off[g];
    rec[b].src_row = r0;
    rec[b].src_nnz = h.rowptr[r0];
    rec[b].dst_nnz = (int32_t)z;
    rec[b].pad = 0;
    z += h.rowptr[h.row_off[g + 1]] - h.rowptr[r0];
  }
  rc = agcn_plan_info(stage_host, &info->plan);
  if (rc != AGCN_OK) return rc;
  info->rows = rows; info->nnz = nnz; info->max_n = mx; info->symmetric = sym; info->stage_bytes = (int64_t)total;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  // Zero-copy staging: when the pinned blob is visible to the device (hipHostMalloc'ed / registered memory: torch's pinned
  // tensors are), the gather launch reads plan + table over the host link itself and copies the blob into the device
  // buffer on the way -- ONE launch per batch instead of a copy-engine transfer (whose hand-over to the compute queue cost
  // ~25 us of stream time per batch) plus a launch.  Otherwise: the asynchronous copy as before.
  static const int zc_env = [] { const char* v = std::getenv("AGCN_STAGE_ZEROCOPY"); return v ? std::atoi(v) : 1; }();
  const char* hd = nullptr;
  if (zc_env) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, stage_host) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer != nullptr)
      hd = static_cast<const char*>(at.devicePointer);
    else
      (void)hipGetLastError();   // (plain pageable memory: not an error, the copy path takes it)
  }
  hipError_t e = hipSuccess;
  if (!hd) {
    e = hipMemcpyAsync(stage_dev, stage_host, total, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: blob upload: %s", who, hipGetErrorString(e));
  }
  agcn::GatherArgs a;
  const char* d = hd ? hd : static_cast<const char*>(stage_dev);
  a.blob_src = reinterpret_cast<const uint2*>(hd);
  a.blob_dst = hd ? static_cast<uint2*>(stage_dev) : nullptr;
  a.blob_vec = (int32_t)(total / 8);
  a.n = reinterpret_cast<const int32_t*>(d + info->plan.off_n);
  a.row_off = reinterpret_cast<const int32_t*>(d + info->plan.off_row_off);
  a.rec = reinterpret_cast<const agcn::GatherRec*>(d + tab_off);
  a.X = st->X_dev; a.rowptr = st->rowptr_dev; a.col = st->col_dev; a.val = st->val_dev;
  a.X_out = X_out; a.rowptr_out = rowptr_out; a.col_out = col_out; a.val_out = val_out;
  a.F = h.F; a.ldx_out = ldx_out; a.x_bf16 = st->x_bf16 != 0 ? 1 : 0; a.out_bf16 = st->stage_bf16 != 0 ? 1 : 0;
  hipLaunchKernelGGL(agcn::store_gather_kernel, dim3((unsigned)B), dim3(256), 0, s, a);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: gather launch: %s", who, hipGetErrorString(e));
  return AGCN_OK;
}
