// Boundary conversions of node-feature matrices: storage dtype (bf16 <-> f32) and row layout (the reference's
// padded feed <-> packed rows).
// BASELINE config "bf16 features + fp32 Laplacian mixed precision": node features / activations (and their
// gradients) may be STORED in bf16 between layers; every kernel of the path computes in fp32 on the tiles
// and on the feature contractions, so the conversion happens once per layer input / output here.
// f32 -> bf16 is round-to-nearest-even (the hardware v_cvt_pk_bf16_f32; NaN stays NaN); bf16 -> f32 is exact.
#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include "agcn_internal.h"

namespace agcn {
namespace {
template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_rows_kernel(const S* src, D* dst, int64_t rows, int32_t cols,
                                                        int32_t lds, int32_t ldd) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / cols;
    const int32_t c = (int32_t)(e - r * cols);
    dst[r * ldd + c] = static_cast<D>(static_cast<float>(src[r * lds + c]));
  }
}
// Row layout conversion between the reference's padded feed ([B][stride][F], graph b's rows at b * stride,
// graph_topology.py:84-91) and the packed rows of a plan.  One workgroup per graph; lanes along the features.
//   PACK:   dst[row_off[b] + i][:] = src[b * stride + i][:]                      i < n[b]
//   UNPACK: dst[b * stride + i][:] = i < n[b] ? src[row_off[b] + i][:] : fill     (fill = 0 or a row vector)
template <bool PACK>
__global__ __launch_bounds__(256) void rows_layout_kernel(const int32_t* n, const int32_t* row_off, const float* src,
                                                          float* dst, int32_t stride, int32_t F, int32_t lds,
                                                          int32_t ldd, const float* fill) {
  const int b = blockIdx.x, nb = n[b], r0 = row_off[b];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nrows = PACK ? nb : stride;
  for (int i = wave; i < nrows; i += 4) {
    const int64_t ps = (int64_t)b * stride + i, pk = (int64_t)r0 + i;
    for (int f = lane; f < F; f += 64) {
      if (PACK) dst[pk * ldd + f] = src[ps * lds + f];
      else dst[ps * ldd + f] = (i < nb) ? src[pk * lds + f] : (fill ? fill[f] : 0.f);
    }
  }
}

int rows_layout(bool pack, const void* plan_host, const void* plan_dev, const float* src, float* dst, int32_t stride,
                int32_t F, int32_t lds, int32_t ldd, const float* fill, void* hip_stream, const char* who) {
  PlanView host, dev;
  int rc = plan_view(plan_host, plan_host, &host);
  if (rc != AGCN_OK) return rc;
  plan_view(plan_host, plan_dev, &dev);
  if (host.info.rows != host.info.N1) return fail(AGCN_EINVAL, "%s: needs a packed plan", who);
  if (!src || !dst) return fail(AGCN_EINVAL, "%s: null tensor", who);
  if (F < 1 || lds < F || ldd < F || stride < host.info.max_n)
    return fail(AGCN_EINVAL, "%s: bad F / leading dimensions / stride (%d < largest graph %d)", who, stride, host.info.max_n);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  if (pack) hipLaunchKernelGGL((rows_layout_kernel<true>), dim3(host.info.B), dim3(256), 0, s, dev.n, dev.row_off, src, dst,
                               stride, F, lds, ldd, fill);
  else hipLaunchKernelGGL((rows_layout_kernel<false>), dim3(host.info.B), dim3(256), 0, s, dev.n, dev.row_off, src, dst,
                          stride, F, lds, ldd, fill);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: %s", who, hipGetErrorString(e));
  return AGCN_OK;
}
}  // namespace
}  // namespace agcn

using namespace agcn;

extern "C" int agcn_rows_pack(const void* plan_host, const void* plan_dev, const float* padded, int32_t stride_rows,
                              int32_t F, int32_t ld_padded, float* packed, int32_t ld_packed, void* hip_stream) {
  return rows_layout(true, plan_host, plan_dev, padded, packed, stride_rows, F, ld_padded, ld_packed, nullptr,
                     hip_stream, "agcn_rows_pack");
}

extern "C" int agcn_rows_unpack(const void* plan_host, const void* plan_dev, const float* packed, int32_t F,
                                int32_t ld_packed, float* padded, int32_t stride_rows, int32_t ld_padded,
                                const float* pad_fill, void* hip_stream) {
  return rows_layout(false, plan_host, plan_dev, packed, padded, stride_rows, F, ld_packed, ld_padded, pad_fill,
                     hip_stream, "agcn_rows_unpack");
}

extern "C" int agcn_cast_rows(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t rows,
                              int32_t cols, int32_t ld_src, int32_t ld_dst, void* hip_stream) {
  const char* who = "agcn_cast_rows";
  if (!src || !dst) return fail(AGCN_EINVAL, "%s: null tensor", who);
  if (rows < 0 || cols < 1 || ld_src < cols || ld_dst < cols) return fail(AGCN_EINVAL, "%s: bad shape", who);
  if (rows == 0) return AGCN_OK;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  const int64_t total = rows * cols;
  const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, 4096);
  if (src_dtype == AGCN_DTYPE_BF16 && dst_dtype == AGCN_DTYPE_F32) {
    hipLaunchKernelGGL((cast_rows_kernel<__hip_bfloat16, float>), dim3(grid), dim3(256), 0, s,
                       static_cast<const __hip_bfloat16*>(src), static_cast<float*>(dst), rows, cols, ld_src, ld_dst);
  } else if (src_dtype == AGCN_DTYPE_F32 && dst_dtype == AGCN_DTYPE_BF16) {
    hipLaunchKernelGGL((cast_rows_kernel<float, __hip_bfloat16>), dim3(grid), dim3(256), 0, s,
                       static_cast<const float*>(src), static_cast<__hip_bfloat16*>(dst), rows, cols, ld_src, ld_dst);
  } else {
    return fail(AGCN_EINVAL, "%s: unsupported conversion %d -> %d (f32 <-> bf16)", who, src_dtype, dst_dtype);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: %s", who, hipGetErrorString(e));
  return AGCN_OK;
}
