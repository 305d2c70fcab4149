// Error reporting + small global knobs of the C ABI (include/agcn.h).
#include "agcn_internal.h"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

namespace {
thread_local char g_err[512] = "";
std::atomic<int32_t> g_lds_limit{160 * 1024};
std::atomic<int32_t> g_slab_mode{[] {
  const char* e = std::getenv("AGCN_SLAB_MODE");   // 0 disables the slab-parallel kernels
  return (e && e[0] == '0') ? 0 : 1;
}()};
}  // namespace

namespace agcn {
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace agcn

extern "C" const char* agcn_last_error(void) { return g_err; }
extern "C" int32_t agcn_version(void) { return AGCN_VERSION; }
extern "C" int32_t agcn_get_lds_limit(void) { return g_lds_limit.load(); }
std::atomic<int32_t> g_fused_mode{[] {
  // 0: never fused; 1: fused when every graph fits; 2: + mixed (big graphs on the per-graph kernels when their
  // tile fits LDS, else on the slab kernels); 3: mixed, big graphs always on the slab kernels
  const char* e = std::getenv("AGCN_FUSED_MODE");
  return (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : 1;
}()};
extern "C" int32_t agcn_get_fused_mode(void) { return g_fused_mode.load(); }
extern "C" void agcn_set_fused_mode(int32_t mode) { g_fused_mode.store(mode < 0 ? 0 : (mode > 3 ? 3 : mode)); }
// -1 (default): the library picks the flavour of the fused kernels per direction; 0: always LDS parameter images;
// 1: always parameters from global memory (when the working set fits)
std::atomic<int32_t> g_fused_gb{[] {
  const char* e = std::getenv("AGCN_FUSED_GB");
  return (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : -1;
}()};
extern "C" int32_t agcn_get_fused_global_b(void) { return g_fused_gb.load(); }
extern "C" void agcn_set_fused_global_b(int32_t mode) { g_fused_gb.store(mode < 0 ? -1 : (mode > 0 ? 1 : 0)); }
// 2 (default): one-pass parameter-gradient kernel, the row splits' partial tiles handed to the tile's last split as tagged
// 8-byte granules where the preceding kernel of the call can clear them (param_grad_tagged_kernel; else as 1);
// 1: one-pass kernel with arrival counters (the last workgroup to arrive adds the splits); 0: partial + reduce launches
std::atomic<int32_t> g_pg_onepass{[] {
  const char* e = std::getenv("AGCN_PG_ONEPASS");
  return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2;
}()};
extern "C" int32_t agcn_get_param_grad_onepass(void) { return g_pg_onepass.load(); }
extern "C" void agcn_set_param_grad_onepass(int32_t on) { g_pg_onepass.store(on < 0 ? 0 : (on > 2 ? 2 : on)); }
std::atomic<int32_t> g_spec_mode{[] {
  // bit 0: specialised forward, bit 1: specialised backward, bit 2: large graphs' backward spread over several workgroups
  const char* e = std::getenv("AGCN_SPEC");
  return (e && e[0] >= '0' && e[0] <= '7') ? e[0] - '0' : 7;
}()};
extern "C" int32_t agcn_get_spec_mode(void) { return g_spec_mode.load(); }
std::atomic<int32_t> g_spec_small{[] {
  // 0: never launch the 8-wave flavour for the small graphs; 1: after the large graphs' launch; 2: beside it
  const char* e = std::getenv("AGCN_SPEC_SMALL");
  return (e && e[0] >= '0' && e[0] <= '2') ? e[0] - '0' : 2;
}()};
extern "C" int32_t agcn_get_spec_small(void) { return g_spec_small.load(); }
std::atomic<int32_t> g_coop_members{[] {
  const char* e = std::getenv("AGCN_COOP_MEMBERS");
  const int v = e ? std::atoi(e) : 120;
  return v < 0 ? 0 : (v > 120 ? 120 : v);
}()};
extern "C" int32_t agcn_get_coop_members(void) { return g_coop_members.load(); }
extern "C" void agcn_set_coop_members(int32_t m) { g_coop_members.store(m < 0 ? 0 : (m > 120 ? 120 : m)); }
std::atomic<uint32_t> g_coop_ticks{1000000u};   // 10 ms of the 100 MHz constant clock
extern "C" uint32_t agcn_get_coop_timeout_ticks(void) { return g_coop_ticks.load(); }
extern "C" void agcn_set_coop_timeout_ticks(uint32_t ticks) { g_coop_ticks.store(ticks); }
std::atomic<int32_t> g_spec_tail{[] {
  // 1: parameter gradients in trailing workgroups of the backward launch (spec_tail.h).  OFF by default: correct and
  // bit-reproducible, but measured SLOWER (first version, one workgroup per tile walking every row: backward 18.4 + 10.2 -> 80.6 us
  // -- ~78 dependent k-steps per wave at a memory round trip per eight: profiles/EXPERIMENTS.md)
  const char* e = std::getenv("AGCN_SPEC_TAIL");
  return (e && e[0] == '1') ? 1 : 0;
}()};
// 1 (default; env AGCN_ROWS_SPLIT=0 turns it off): the packed-row products of point-cloud sized batches run on the bf16 matrix pipe
// at fp32 accuracy (rows_split.hip); 0: the fp32-MFMA kernels of rows_wide.hip / cloud_kernels.hip
std::atomic<int32_t> g_rows_split{[] {
  const char* e = std::getenv("AGCN_ROWS_SPLIT");
  return (e && e[0] == '0') ? 0 : 1;
}()};
extern "C" int32_t agcn_get_rows_split(void) { return g_rows_split.load(); }
extern "C" void agcn_set_rows_split(int32_t on) { g_rows_split.store(on ? 1 : 0); }
extern "C" int32_t agcn_get_spec_tail(void) { return g_spec_tail.load(); }
extern "C" void agcn_set_spec_tail(int32_t on) { g_spec_tail.store(on ? 1 : 0); }
extern "C" void agcn_set_spec_small(int32_t on) { g_spec_small.store(on < 0 ? 0 : (on > 2 ? 2 : on)); }
extern "C" void agcn_set_spec_mode(int32_t mode) { g_spec_mode.store(mode < 0 ? 0 : (mode > 7 ? 7 : mode)); }
extern "C" int32_t agcn_get_slab_mode(void) { return g_slab_mode.load(); }
extern "C" void agcn_set_slab_mode(int32_t mode) { g_slab_mode.store(mode ? 1 : 0); }
extern "C" void agcn_set_lds_limit(int32_t bytes) {
  if (bytes < 0) bytes = 0;
  if (bytes > 160 * 1024) bytes = 160 * 1024;
  g_lds_limit.store(bytes);
}

namespace agcn {
namespace {
thread_local uint32_t t_flags = 0;   // the flags of the call this thread is inside (0: none / nothing overridden)
inline int32_t field(int shift, uint32_t mask) { return (int32_t)((t_flags >> shift) & mask); }
}  // namespace
KnobScope::KnobScope(uint32_t flags) : prev_(t_flags) { t_flags = flags; }
KnobScope::~KnobScope() { t_flags = prev_; }
static inline int32_t clampi(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }
int32_t knob_spec_mode() { const int32_t f = field(0, 15u); return f ? clampi(f - 1, 0, 7) : g_spec_mode.load(); }
int32_t knob_fused_mode() { const int32_t f = field(4, 7u); return f ? clampi(f - 1, 0, 3) : g_fused_mode.load(); }
int32_t knob_pg_onepass() { const int32_t f = field(7, 3u); return f ? clampi(f - 1, 0, 2) : g_pg_onepass.load(); }
int32_t knob_spec_small() { const int32_t f = field(9, 3u); return f ? clampi(f - 1, 0, 2) : g_spec_small.load(); }
int32_t knob_rows_split() { const int32_t f = field(11, 3u); return f ? clampi(f - 1, 0, 1) : g_rows_split.load(); }
int32_t knob_slab_mode() { const int32_t f = field(13, 3u); return f ? clampi(f - 1, 0, 1) : g_slab_mode.load(); }
int32_t knob_fused_gb() { const int32_t f = field(15, 3u); return f ? clampi(f - 2, -1, 1) : g_fused_gb.load(); }
int32_t knob_spec_tail() { const int32_t f = field(17, 3u); return f ? clampi(f - 1, 0, 1) : g_spec_tail.load(); }
// AGCN_BFPIPE (environment, read once; A/B measurements): bit mask of the per-graph kernels' feature contractions that run on the
// bf16 matrix pipe at fp32 accuracy (csrc/spec_bfpipe.h).
//   bit 2 (value 4, DEFAULT): the filter adjoint [g_0 | g_1] = Gp W^T of the backward for calls with bf16 features -- Gp is one
//          bf16 piece, W^T a three-piece image in MFMA operand order that the forward's last blocks leave in the reserve: three
//          v_mfma_f32_16x16x32_bf16 per 32 k.  Measured (HIV-shaped 256-molecule step, profiles/EXPERIMENTS.md round 6): the phase
//          2.04 -> 0.49 us in the one-workgroup bodies, 2.98 -> 1.48 us in the members; the bf16 step 69.9 -> 67.5 us.
//   bit 3 (value 8, DEFAULT): the same phase for calls with fp32 features: Gp split exactly into three bf16 piece planes in LDS (six
//          instructions per 32 k), where the launch's bodies still fit their LDS with the planes.
//   bit 0 (value 1): Z = X M_L of the member workgroups' forward for calls with bf16 features, parameter image built in LDS by each
//          member.  Off: the phase itself goes 3.5 -> 1.8 us, but the image costs 1.0-1.5 us of LDS stores per member and the
//          launch's larger dynamic LDS (up to 140 KB for every workgroup) keeps two small graphs from sharing a CU: step + 1 us.
//   bit 1 (value 2): the same for fp32 features (three piece planes): level with the fp32 instruction.  Off.
int32_t knob_bfpipe() {
  static const int32_t v = [] { const char* e = std::getenv("AGCN_BFPIPE"); return e ? (int32_t)std::atoi(e) : 12; }();
  return v;
}
int32_t knob_coop_members() { const int32_t f = field(24, 255u); return f ? clampi(f - 1, 0, 120) : g_coop_members.load(); }

// One workgroup per graph: with few graphs per CU, 16 waves per graph hide the memory latency of
// the bulk loads and spread the MFMA tiles; with many graphs per CU the CU is full anyway.
int graph_threads(int32_t n_graphs) {
  if (n_graphs <= 2 * 256) return 1024;
  if (n_graphs <= 4 * 256) return 512;
  return 256;
}
}  // namespace agcn
