// (placeholder until the specialised backward lands: never selected)
#include "spec_kernels.h"
namespace agcn {
size_t spec_bwd_lds_bytes(int32_t, int32_t, int32_t) { return ~(size_t)0; }
int launch_spec_backward(const SgcArgs&, int32_t, int32_t, size_t, void*) { return fail(AGCN_EINVAL, "spec_backward: not built"); }
}
