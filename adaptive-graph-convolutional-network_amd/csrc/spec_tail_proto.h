// Constants and the producer / exit side of the trailing parameter-gradient workgroups (spec_tail.h has the description and the
// consumer).  Included by spec_backward_coop.h behind its own header constants.
#ifndef AGCN_SPEC_TAIL_PROTO_H_
#define AGCN_SPEC_TAIL_PROTO_H_

namespace agcn {
namespace spec {

constexpr int TAIL_OFF_FLOATS = 2 * COOP_HEADER_WORDS + 64;   // behind the exchange words and the status word
constexpr int TAIL_GEN = 0, TAIL_DONE = 1, TAIL_FLAGS = 8;    // uint32 word indices inside the tail region
constexpr int TAIL_NWIN = 6;                                  // the walk is held in registers: 6 windows of 64 units
constexpr int TAIL_MAX_UNITS = 64 * TAIL_NWIN;
constexpr int TAIL_MAX_SPLITS = 12;
constexpr int TAIL_MAX_TILES = 128;                           // (128 -> 128: 16 * 4 + 8 * 4 = 96)
constexpr int TAIL_FLAGS2 = TAIL_FLAGS + TAIL_MAX_UNITS;      // flags of the partial tiles, [split][tile]
constexpr int TAIL_STARTED = TAIL_FLAGS2 + TAIL_MAX_SPLITS * TAIL_MAX_TILES;   // one word per workgroup of the launch
constexpr int TAIL_MAX_WGS = 4096;
constexpr int TAIL_WORDS = TAIL_STARTED + TAIL_MAX_WGS;
constexpr int TAIL_TE = 512;                                  // output tile: 16 x 32
constexpr int TAIL_PSTRIDE = TAIL_TE + 32;                    // floats per partial tile (+ the bias partials of its columns)
constexpr int TAIL_KR = 8;                                    // k-steps per wave and round of loads

// output tiles of dweight (16 rows of the interleaved [f * K + k] index x 32 columns) and dM_L
AGCN_HD inline int tail_tiles(int Fin, int Fout, int K, bool paper) {
  return ((Fin * K + 15) / 16) * ((Fout + 31) / 32) + (paper ? ((Fin + 15) / 16) * ((Fin + 31) / 32) : 0);
}
// row splits for a walk of T k-steps: about TAIL_KR k-steps per wave (16 waves per workgroup)
AGCN_HD inline int tail_splits(int T) {
  int s = (T + 16 * TAIL_KR - 1) / (16 * TAIL_KR);
  return s < 1 ? 1 : (s > TAIL_MAX_SPLITS ? TAIL_MAX_SPLITS : s);
}

#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t tail_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tail_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tail_store_f(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float tail_load_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// A producer unit is done: every thread of the unit has issued its agent-scope stores (dZ rows, the dalpha partial).  Each
// storing wave drains them, the unit's threads meet, one thread publishes.  `bar`: a barrier every thread of the WORKGROUP
// reaches (the bodies' barriers are workgroup-wide also in the paired flavour).
template <class Barrier>
__device__ __forceinline__ void tail_publish(const SgcArgs& a, int slot, uint32_t gen, bool leader, Barrier bar) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bar();
  if (leader) tail_store(a.tail + TAIL_FLAGS + slot, gen + 1u);
}

// Every workgroup of the launch, as soon as it has read the generation word (one thread): "I have read it".  Plain agent-scope
// stores -- an exit COUNTER on one address (first version) serialised ~600 device-scope atomics: +40 us on the launch.
__device__ __forceinline__ void tail_started(const SgcArgs& a, uint32_t gen) {
  tail_store(a.tail + TAIL_STARTED + blockIdx.x, gen + 1u);
}
// ONE designated workgroup (a dweight reducer: done early, off the critical path), at its end (one wave): when every workgroup
// of the launch has read the generation word it may change -- the next launch's flags are told apart by it.
__device__ __forceinline__ void tail_bump(const SgcArgs& a, uint32_t gen, int lane) {
  const int n = (int)gridDim.x;
  const unsigned long long t0 = wall_clock64();
  for (;;) {
    bool ok = true;
    for (int i = lane; i < n; i += 64) ok = ok && (tail_load(a.tail + TAIL_STARTED + i) == gen + 1u);
    if (__ballot(ok) == ~0ull) break;
    if (wall_clock64() - t0 >= 100000000ull) break;     // (1 s: a launch that never got all its workgroups started)
    __builtin_amdgcn_s_sleep(8);
  }
  if (lane == 0) tail_store(a.tail + TAIL_GEN, gen + 1u);
}
#endif  // __HIPCC__

}  // namespace spec
}  // namespace agcn

#endif  // AGCN_SPEC_TAIL_PROTO_H_
