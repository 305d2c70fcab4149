// Storage-dtype conversion of node-feature matrices at the layer boundary: bf16 <-> f32.
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
}  // namespace
}  // namespace agcn

using namespace agcn;

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
