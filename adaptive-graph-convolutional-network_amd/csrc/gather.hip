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
    for (int i = 0; i < nb; ++i) s += Xg[(int64_t)i * ldx + f];   // tf.reduce_sum(x, 0), graphgather.py:74
    if (act == AGCN_ACT_TANH) s = tanhf(s);
    else if (act == AGCN_ACT_RELU) s = fmaxf(s, 0.f);
    out[(int64_t)b * F + f] = s;
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
  hipLaunchKernelGGL(graph_gather_bwd_kernel, dim3(pv.info.B), dim3(256), 0, static_cast<hipStream_t>(hip_stream),
                     pv.n, pv.row_off, out, dOut, F, ldx, activation, dX);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "graph_gather_bwd launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}
