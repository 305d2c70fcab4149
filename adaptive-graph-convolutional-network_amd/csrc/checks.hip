// Debug / first-call checks of what a caller VOUCHES for (include/agcn.h: AgcnGraphBatch.L0_symmetric, L_prev_symmetric).
// The point-cloud kernels take La^T g_1 as La g_1 on the strength of those flags: a false vouch gives a silently wrong dX.
// agcn_tiles_asymmetry counts, per call, the element pairs (i < j) of packed per-graph tiles whose two sides differ -- bit
// for bit, which is what the kernels' own symmetric outputs satisfy (res_W, res_L, L_sum of symmetric inputs).
#include "agcn_device.h"
#include "agcn_internal.h"

namespace agcn {

// one workgroup per (row i, graph b): lanes run over the columns j > i; row reads coalesced, column reads strided (a check,
// not a hot path).  NaN on either side counts as a difference.
__global__ __launch_bounds__(256) void tiles_asymmetry_kernel(const int32_t* n, const int64_t* tile_off, const float* tiles,
                                                              int32_t ld, int32_t* count) {
  const int b = blockIdx.y, i = blockIdx.x;
  const int nb = n[b];
  if (i >= nb) return;
  const int l = ld > 0 ? ld : nb;
  const float* T = tiles + (ld > 0 ? (int64_t)b * ld * ld : tile_off[b]);
  int bad = 0;
  for (int j = i + 1 + threadIdx.x; j < nb; j += 256) {
    const float u = T[(int64_t)i * l + j], v = T[(int64_t)j * l + i];
    bad += (u == v) ? 0 : 1;
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(count, bad);      // (an integer count: order does not matter)
}

}  // namespace agcn

using namespace agcn;

extern "C" int agcn_tiles_asymmetry(const void* plan_host, const void* plan_dev, const float* tiles, int32_t ld,
                                    int32_t* count_dev, void* hip_stream) {
  PlanView pv, host;
  int rc = plan_view(plan_host, plan_dev, &pv);
  if (rc != AGCN_OK) return rc;
  plan_view(plan_host, plan_host, &host);
  if (!tiles || !count_dev || ld < 0) return fail(AGCN_EINVAL, "agcn_tiles_asymmetry: bad argument");
  if (host.info.B == 0 || host.info.max_n < 2) return AGCN_OK;
  hipLaunchKernelGGL(tiles_asymmetry_kernel, dim3((unsigned)host.info.max_n, (unsigned)host.info.B), dim3(256), 0,
                     static_cast<hipStream_t>(hip_stream), pv.n, pv.tile_off, tiles, ld, count_dev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "tiles_asymmetry launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}
