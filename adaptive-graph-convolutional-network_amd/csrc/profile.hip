// Optional per-kernel timing with HIP events recorded on the launch stream, around each kernel
// of the path (include/agcn.h: agcn_profile_*).  Off by default; when off the launch path does
// not touch it.  Debug facility: single-threaded use only.
#include <hip/hip_runtime.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "sgc_kernels.h"

namespace agcn {
namespace {
struct Rec { int id; hipEvent_t a, b; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
const char* const kNames[PK_COUNT] = {
    "rows_gemm_kernel<PROJECT>", "sgc_forward_kernel", "rows_gemm_kernel<FILTER>",
    "rows_gemm_kernel<FILTER_BWD>", "sgc_backward_kernel", "rows_gemm_kernel<PROJECT_BWD>",
    "param_grad_partial_kernel", "param_grad_reduce_kernel", "slab_sim_kernel", "slab_lap_kernel",
    "slab_bwd1_kernel", "slab_bwd2_kernel", "fused_forward_kernel", "fused_backward_kernel",
    "param_grad_onepass_kernel", "spec_forward_kernel", "spec_backward_kernel",
    "cloud_sim_kernel", "cloud_degree_kernel", "cloud_lap_kernel", "cloud_bwd1_kernel", "cloud_bwd2_kernel",
    "cloud_rnorm_kernel", "cloud_rlap1_kernel", "cloud_rbwd_kernel<A>", "cloud_rbwd_kernel<B>", "cloud_rbwd_kernel<C>", "param_grad_tagged_kernel", "rs_presplit_kernel"};

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

int ensure_dynamic_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::unordered_map<uint64_t, size_t> cap;   // (device, kernel) -> cap already set
  if (bytes <= 64 * 1024) return 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(mu);
  size_t& have = cap[reinterpret_cast<uint64_t>(kernel) ^ ((uint64_t)(dev + 1) << 56)];
  if (have >= bytes) return 0;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return (int)e;
}

bool prof_on() { return g_on; }

void prof_begin(int id, void* stream) {
  if (!g_on) return;
  Rec r{id, get_event(), get_event()};
  (void)hipEventRecord(r.a, static_cast<hipStream_t>(stream));
  g_recs.push_back(r);
}

void prof_end(void* stream) {
  if (!g_on || g_recs.empty()) return;
  (void)hipEventRecord(g_recs.back().b, static_cast<hipStream_t>(stream));
}
}  // namespace agcn

using namespace agcn;

extern "C" void agcn_profile_enable(int on) { g_on = on != 0; }

extern "C" int32_t agcn_profile_kernels(void) { return PK_COUNT; }

extern "C" const char* agcn_profile_kernel_name(int32_t id) {
  return (id >= 0 && id < PK_COUNT) ? kNames[id] : "";
}

// Waits for the recorded events, returns per-kernel totals since the last call and clears them.
extern "C" int agcn_profile_collect(double* total_ms /* [PK_COUNT] */, int64_t* calls /* [PK_COUNT] */) {
  if (!total_ms || !calls) return fail(AGCN_EINVAL, "agcn_profile_collect: null argument");
  for (int i = 0; i < PK_COUNT; ++i) { total_ms[i] = 0.0; calls[i] = 0; }
  for (Rec& r : g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return fail(AGCN_EDEVICE, "agcn_profile_collect: event sync failed");
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    total_ms[r.id] += ms;
    calls[r.id] += 1;
    g_pool.push_back(r.a);
    g_pool.push_back(r.b);
  }
  g_recs.clear();
  return AGCN_OK;
}

#ifdef AGCN_STAMPS
__device__ unsigned long long g_agcn_stamps[8 * 1024 * 16];
extern "C" int agcn_debug_read_stamps(unsigned long long* out, size_t count, int clear) {
  (void)hipDeviceSynchronize();
  if (count > 8 * 1024 * 16) count = 8 * 1024 * 16;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_agcn_stamps), count * sizeof(unsigned long long)) != hipSuccess)
    return -3;
  if (clear) {
    static unsigned long long zeros[8 * 1024 * 16];
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_agcn_stamps), zeros, sizeof(zeros));
  }
  return 0;
}
#endif
