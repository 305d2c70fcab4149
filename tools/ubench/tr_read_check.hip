// Diagnostic: semantics of ds_read_b64_tr_b16 on gfx950 as used for the B operand of v_mfma_f32_16x16x32_bf16 out of a
// row-major [k][column] 16-bit LDS image: lane (q = lane / 16, r = lane % 16) supplies the address of (row 8q + r / 4,
// columns 4 (r % 4) ..) and must receive column r of rows 8q .. 8q + 3.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/tr_read_check tools/ubench/tr_read_check.hip && /tmp/tr_read_check
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = 72;
__global__ void k(unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[64 * LDH];
  for (int i = threadIdx.x; i < 64 * LDH; i += 64) img[i] = (unsigned short)((i / LDH) * 100 + (i % LDH));
  __syncthreads();
  const int lane = threadIdx.x, q = lane >> 4, r = lane & 15;
  const unsigned short* p = img + (8 * q + (r >> 2)) * LDH + 4 * (r & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
  out[lane] = __builtin_bit_cast(unsigned long long, v);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned long long h[64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    const int q = lane >> 4, r = lane & 15;
    for (int e = 0; e < 4; ++e) {
      const int got = (int)((h[lane] >> (16 * e)) & 0xffff), want = (8 * q + e) * 100 + r;
      if (got != want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", lane, e, got, want); ++bad; }
    }
  }
  printf("tr_read_check: %s (%d mismatches)\n", bad ? "MISMATCH" : "ok", bad);
  return bad != 0;
}
