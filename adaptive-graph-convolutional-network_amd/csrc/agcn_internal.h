// Internal declarations shared by the host-side translation units.
#ifndef AGCN_INTERNAL_H_
#define AGCN_INTERNAL_H_

#include "../../include/agcn.h"

#include <cstdint>
#include <cstring>

// n^3 must stay below 2^32 for the exact magic-number division used by the kernels.
#define AGCN_MAX_NODES 1600

namespace agcn {

int fail(int code, const char* fmt, ...);

// Leading dimension of an on-chip [rows x cols] float image: odd, so that a
// column walk (stride ld) touches 32 different LDS banks.
inline int32_t odd_ld(int32_t cols) { return cols | 1; }

// Read-only view of a plan blob (host or device copy).
struct PlanView {
  AgcnPlanInfo info;
  const int32_t* n;
  const int32_t* row_off;
  const int64_t* tile_off;
  const int32_t* order;
  const int64_t* wtile_off;
};

// Validates the HOST blob and returns views whose pointers point into `base`
// (pass the device blob as `base` to get device pointers).
inline int plan_view(const void* plan_host, const void* base, PlanView* v) {
  int rc = agcn_plan_info(plan_host, &v->info);
  if (rc != AGCN_OK) return rc;
  const char* p = static_cast<const char*>(base);
  v->n = reinterpret_cast<const int32_t*>(p + v->info.off_n);
  v->row_off = reinterpret_cast<const int32_t*>(p + v->info.off_row_off);
  v->tile_off = reinterpret_cast<const int64_t*>(p + v->info.off_tile_off);
  v->order = reinterpret_cast<const int32_t*>(p + v->info.off_order);
  v->wtile_off = reinterpret_cast<const int64_t*>(p + v->info.off_wtile_off);
  return AGCN_OK;
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace agcn

#endif  // AGCN_INTERNAL_H_
