// Diagnostic microbenchmark: latency of a (value, epoch) word hand-off between two workgroups on gfx950, by the cache-scope bits of the
// store and of the polling load, for a pair on the SAME XCD (blocks b, b + 8: workgroups are dealt round-robin over the 8 XCDs) and on
// DIFFERENT XCDs (blocks b, b + 1).  Ping-pong of `rounds` exchanges, 100 MHz wall clock; one way = total / (2 rounds).
//   hipcc --offload-arch=gfx950 -O3 -o handoff handoff.hip && ./handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int S> __device__ __forceinline__ void store_w(uint64_t* p, uint64_t v) {
  if (S == 0) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");
  if (S == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
  if (S == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  if (S == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int L> __device__ __forceinline__ uint64_t load_w(const uint64_t* p) {
  uint64_t v;
  if (L == 0) asm volatile("global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (L == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// blocks A and B play; the others leave at once.  words[0]: A -> B, words[16]: B -> A (separate lines).
template <int S, int L>
__global__ void pingpong(uint64_t* words, int A, int B, int rounds, uint32_t base, unsigned long long* out, uint32_t* xcc) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) xcc[b] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
  if ((b != A && b != B) || threadIdx.x != 0) return;
  uint64_t* mine = words + (b == A ? 0 : 16);
  const uint64_t* theirs = words + (b == A ? 16 : 0);
  const unsigned long long t0 = wall_clock64();
  unsigned long long spins = 0;
  for (int k = 1; k <= rounds; ++k) {
    const uint32_t ep = base + k;
    if (b == A) store_w<S>(mine, ((uint64_t)ep << 32) | k);
    for (;;) {
      const uint64_t w = load_w<L>(theirs);
      if ((uint32_t)(w >> 32) == ep) break;
      ++spins;
      if (wall_clock64() - t0 > 100000000ull) { out[2] = 1; return; }      // 1 s: give up (a scope that never sees the word)
    }
    if (b == B) store_w<S>(mine, ((uint64_t)ep << 32) | k);
  }
  if (b == A) { out[0] = wall_clock64() - t0; out[1] = spins; }
}

template <int S, int L>
static int run(const char* name, uint64_t* words, unsigned long long* out, uint32_t* xcc, int A, int B, uint32_t& base) {
  const int rounds = 200;
  CHECK(hipMemset(out, 0, 3 * sizeof(unsigned long long)));
  hipLaunchKernelGGL((pingpong<S, L>), dim3(16), dim3(64), 0, 0, words, A, B, rounds, base, out, xcc);
  CHECK(hipDeviceSynchronize());
  base += rounds + 7;
  unsigned long long h[3];
  uint32_t x[16];
  CHECK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(x, xcc, sizeof(x), hipMemcpyDeviceToHost));
  if (h[2]) printf("%-34s blocks %2d,%2d (xcc %u,%u): NEVER SEEN (gave up after 1 s)\n", name, A, B, x[A] & 15, x[B] & 15);
  else printf("%-34s blocks %2d,%2d (xcc %u,%u): one way %.2f us  (%.1f polls per wait)\n", name, A, B, x[A] & 15, x[B] & 15,
              h[0] / 100.0 / (2.0 * rounds), (double)h[1] / rounds);
  return 0;
}

int main() {
  uint64_t* words;
  unsigned long long* out;
  uint32_t* xcc;
  CHECK(hipMalloc(&words, 4096));
  CHECK(hipMemset(words, 0, 4096));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMalloc(&xcc, 64));
  uint32_t base = 1000;
  for (int rep = 0; rep < 2; ++rep) {
    for (int pair = 0; pair < 2; ++pair) {
      const int A = 0, B = pair == 0 ? 8 : 1;
      printf("-- %s XCD\n", pair == 0 ? "same" : "other");
      run<2, 2>("store sc1      / load sc1", words, out, xcc, A, B, base);
      run<3, 3>("store sc0 sc1  / load sc0 sc1", words, out, xcc, A, B, base);
      run<2, 1>("store sc1      / load sc0", words, out, xcc, A, B, base);
      run<1, 1>("store sc0      / load sc0", words, out, xcc, A, B, base);
      run<0, 1>("store (none)   / load sc0", words, out, xcc, A, B, base);
      run<0, 0>("store (none)   / load (none)", words, out, xcc, A, B, base);
    }
  }
  return 0;
}
