// GraphGatherMol (models/layers/graphgather.py:40-77): per graph, sum the real rows of the
// node-feature matrix and apply the activation -> [B, F].  One workgroup per graph; lanes run
// across features (coalesced row reads), rows are added in index order (deterministic).
#include "agcn_device.h"
#include "agcn_internal.h"

namespace agcn {

__global__ __launch_bounds__(256) void graph_gather_fwd_kernel(const int32_t* n, const int32_t* row_off,
                                                               const float* X, int F, int ldx, int act,
                                                               float* out) {
  const int b = blockIdx.x;
  const int nb = n[b];
  const float* Xg = X + (int64_t)row_off[b] * ldx;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    float s = 0.f;
    // tf.reduce_sum(x, 0), graphgather.py:74: rows added in index order, sixteen loads in flight at a time (one dependent load
    // per row was 15 us for 256 molecules x 256 features: a memory round trip per row)
    for (int i0 = 0; i0 < nb; i0 += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = Xg[(int64_t)min(i0 + u, nb - 1) * ldx + f];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (i0 + u < nb) s += v[u];
    }
    if (act == AGCN_ACT_TANH) s = tanhf(s);
    else if (act == AGCN_ACT_RELU) s = fmaxf(s, 0.f);
    out[(int64_t)b * F + f] = s;
  }
}

// Large graphs (point clouds: 1024 nodes): one workgroup per (graph, 64 features), its sixteen waves take contiguous row chunks
// (eight rows in flight per lane), the chunk sums are added in chunk order by wave 0 -- a fixed order for a given graph size.
// (One 256-thread workgroup per graph walking 1024 rows serially: 237 us for 32 clouds x 256 features, 33 MB; this: ~15 us.)
__global__ __launch_bounds__(1024) void graph_gather_fwd_large_kernel(const int32_t* n, const int32_t* row_off,
                                                                      const float* X, int F, int ldx, int act,
                                                                      float* out) {
  __shared__ float part[16][64];
  const int b = blockIdx.x, f = blockIdx.y * 64 + (threadIdx.x & 63), wave = threadIdx.x >> 6;
  const int nb = n[b];
  const int chunk = (nb + 15) / 16, i0 = wave * chunk, i1 = min(nb, i0 + chunk);
  const float* Xg = X + (int64_t)row_off[b] * ldx + min(f, F - 1);
  float s = 0.f;
  for (int i = i0; i < i1; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = Xg[(int64_t)min(i + u, nb - 1) * ldx];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i + u < i1) s += v[u];
  }
  part[wave][threadIdx.x & 63] = s;
  __syncthreads();
  if (wave == 0 && f < F) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w][threadIdx.x];
    if (act == AGCN_ACT_TANH) t = tanhf(t);
    else if (act == AGCN_ACT_RELU) t = fmaxf(t, 0.f);
    out[(int64_t)b * F + f] = t;
  }
}

// ... and its adjoint: one workgroup per (graph, 64 owned rows), lanes across the features
__global__ __launch_bounds__(256) void graph_gather_bwd_large_kernel(const int32_t* n, const int32_t* row_off,
                                                                     const float* out, const float* dOut, int F,
                                                                     int ldx, int act, float* dX) {
  const int b = blockIdx.x;
  const int nb = n[b];
  const int r0 = row_off[b], owned = row_off[b + 1] - r0;
  const int i0 = blockIdx.y * 64, i1 = min(owned, i0 + 64);
  if (i0 >= owned) return;
  float* dXg = dX + (int64_t)r0 * ldx;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    const float o = out[(int64_t)b * F + f];
    float g = dOut[(int64_t)b * F + f];
    if (act == AGCN_ACT_TANH) g = g * (1.0f - o * o);
    else if (act == AGCN_ACT_RELU) g = o > 0.f ? g : 0.f;
    for (int i = i0; i < i1; ++i) dXg[(int64_t)i * ldx + f] = i < nb ? g : 0.f;
  }
}

__global__ __launch_bounds__(256) void graph_gather_bwd_kernel(const int32_t* n, const int32_t* row_off,
                                                               const float* out, const float* dOut, int F,
                                                               int ldx, int act, float* dX) {
  const int b = blockIdx.x;
  const int nb = n[b];
  const int r0 = row_off[b], owned = row_off[b + 1] - r0;
  float* dXg = dX + (int64_t)r0 * ldx;
  for (int f = threadIdx.x; f < F; f += blockDim.x) {
    const float o = out[(int64_t)b * F + f];
    float g = dOut[(int64_t)b * F + f];
    if (act == AGCN_ACT_TANH) g = g * (1.0f - o * o);
    else if (act == AGCN_ACT_RELU) g = o > 0.f ? g : 0.f;
    for (int i = 0; i < owned; ++i) dXg[(int64_t)i * ldx + f] = i < nb ? g : 0.f;
  }
}

}  // namespace agcn

using namespace agcn;

extern "C" int agcn_graph_gather_forward(const void* plan_host, const void* plan_dev, const float* X,
                                         int32_t F, int32_t ldx, int32_t activation, float* out,
                                         void* hip_stream) {
  PlanView pv;
  int rc = plan_view(plan_host, plan_dev, &pv);
  if (rc != AGCN_OK) return rc;
  if (!X || !out || F < 1 || ldx < F) return fail(AGCN_EINVAL, "agcn_graph_gather_forward: bad argument");
  if (activation < AGCN_ACT_LINEAR || activation > AGCN_ACT_TANH)
    return fail(AGCN_EINVAL, "agcn_graph_gather_forward: unsupported activation %d", activation);
  // graphs of hundreds of nodes (point clouds): rows split over the waves of one workgroup per (graph, 64 features); molecules:
  // one workgroup per graph, rows in index order
  if (pv.info.max_n >= 256)
    hipLaunchKernelGGL(graph_gather_fwd_large_kernel, dim3(pv.info.B, (F + 63) / 64), dim3(1024), 0, static_cast<hipStream_t>(hip_stream),
                       pv.n, pv.row_off, X, F, ldx, activation, out);
  else
  hipLaunchKernelGGL(graph_gather_fwd_kernel, dim3(pv.info.B), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     pv.n, pv.row_off, X, F, ldx, activation, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "graph_gather_fwd launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

extern "C" int agcn_graph_gather_backward(const void* plan_host, const void* plan_dev, const float* out,
                                          const float* dOut, int32_t F, int32_t ldx, int32_t activation,
                                          float* dX, void* hip_stream) {
  PlanView pv;
  int rc = plan_view(plan_host, plan_dev, &pv);
  if (rc != AGCN_OK) return rc;
  if (!out || !dOut || !dX || F < 1 || ldx < F) return fail(AGCN_EINVAL, "agcn_graph_gather_backward: bad argument");
  if (activation < AGCN_ACT_LINEAR || activation > AGCN_ACT_TANH)
    return fail(AGCN_EINVAL, "agcn_graph_gather_backward: unsupported activation %d", activation);
  if (pv.info.max_n >= 256) {
    const int owned_max = pv.info.rows == pv.info.N1 ? pv.info.max_n : pv.info.rows / pv.info.B;   // packed | padded rows
    hipLaunchKernelGGL(graph_gather_bwd_large_kernel, dim3(pv.info.B, (owned_max + 63) / 64), dim3(256), 0,
                       static_cast<hipStream_t>(hip_stream), pv.n, pv.row_off, out, dOut, F, ldx, activation, dX);
  } else
  hipLaunchKernelGGL(graph_gather_bwd_kernel, dim3(pv.info.B), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     pv.n, pv.row_off, out, dOut, F, ldx, activation, dX);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "graph_gather_bwd launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}
