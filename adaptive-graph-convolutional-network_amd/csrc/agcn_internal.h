// Internal declarations shared by the host-side translation units.
#ifndef AGCN_INTERNAL_H_
#define AGCN_INTERNAL_H_

#include "../../include/agcn.h"

#include <cstdint>
#include <cstring>

// n^3 must stay below 2^32 for the exact magic-number division used by the kernels.
#define AGCN_MAX_NODES 1600

#if defined(__HIPCC__) || defined(__HIP__)
#define AGCN_HD __host__ __device__
#else
#define AGCN_HD
#endif

namespace agcn {

int fail(int code, const char* fmt, ...);

// Dispatch knobs as THIS call sees them: the per-call flags of its descriptor (AgcnLayerDesc.flags / AgcnDenseDesc.flags) where a
// field is set, the process default (environment / agcn_set_*) otherwise.  A KnobScope lives on the stack of every entry point
// that takes a descriptor; the override is thread-local, so concurrent calls with different flags do not see each other.
struct KnobScope {
  explicit KnobScope(uint32_t flags);
  ~KnobScope();
  KnobScope(const KnobScope&) = delete;
  KnobScope& operator=(const KnobScope&) = delete;
 private:
  uint32_t prev_;
};
int32_t knob_spec_mode();
int32_t knob_fused_mode();
int32_t knob_pg_onepass();
int32_t knob_spec_small();
int32_t knob_rows_split();
int32_t knob_slab_mode();
int32_t knob_fused_gb();
int32_t knob_spec_tail();
int32_t knob_coop_members();
int32_t knob_bfpipe();

// Leading dimensions of on-chip images, chosen so that MFMA fragment reads are LDS
// bank-conflict free:
//  * an image read as the A operand (lane (r, q) reads [row0 + r][k0 + q]) -- or, transposed,
//    as the B operand -- wants ld mod 16 in {2, 14}: r*ld + q hits 32 different banks per
//    half-wave;
//  * a row-major image read as the B operand (lane (r, q) reads [k0 + q][c0 + r]) wants
//    ld = 16 mod 32.
AGCN_HD inline int32_t ld_frag(int32_t cols) {
  int32_t v = cols < 2 ? 2 : cols;
  while ((v & 15) != 2 && (v & 15) != 14) ++v;
  return v;
}
AGCN_HD inline int32_t ld_b(int32_t cols) { return cols <= 16 ? 16 : ((cols - 16 + 31) / 32) * 32 + 16; }
// row stride (floats) of the per-graph image scratch used in global-tile mode
AGCN_HD inline int32_t scratch_fs(int32_t Fin) { return (Fin + 35) & ~3; }
// floats of one per-graph image scratch region for the whole batch (global-tile mode)
AGCN_HD inline int64_t graph_scratch_floats_dev(int32_t rows, int32_t B, int32_t Fin) {
  return ((int64_t)rows + 16 * (int64_t)B) * (int64_t)scratch_fs(Fin);
}

// Read-only view of a plan blob (host or device copy).
struct PlanView {
  AgcnPlanInfo info;
  const int32_t* n;
  const int32_t* row_off;
  const int64_t* tile_off;
  const int32_t* order;
  const int64_t* wtile_off;
  const int32_t* slab_graph;
  const int32_t* slab_row;
  const int32_t* row_graph;
  const AgcnGraphDesc* graph_desc;
  const AgcnSlabDesc* slab_desc;
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
  v->slab_graph = reinterpret_cast<const int32_t*>(p + v->info.off_slab_graph);
  v->slab_row = reinterpret_cast<const int32_t*>(p + v->info.off_slab_row);
  v->row_graph = reinterpret_cast<const int32_t*>(p + v->info.off_row_graph);
  v->graph_desc = reinterpret_cast<const AgcnGraphDesc*>(p + v->info.off_graph_desc);
  v->slab_desc = reinterpret_cast<const AgcnSlabDesc*>(p + v->info.off_slab_desc);
  return AGCN_OK;
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

// Per-device facts of the CURRENT HIP device (agcn_api.hip; cached per device id, lock-free): its CU count -- the grids and
// the cooperative-member budget are sized for the device a call runs on, not the one the process saw first -- and the
// caller-registered persistent status word (agcn_set_device_status_word), or nullptr.
struct DeviceInfo { int n_cus; uint32_t* status_word; };
DeviceInfo current_device_info();

}  // namespace agcn

#endif  // AGCN_INTERNAL_H_
