// Development harness for csrc/rows_split.hip: every packed-row product of a point-cloud layer at 32768 rows, timed as 50
// back-to-back launches (HIP events; the queue stays full, so the average is the kernel's duration) and checked against a
// float64 host evaluation on sampled rows.  Includes the kernel source itself (internal launchers are not part of the C ABI).
//   build:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -I include \
//                 -o tools/ubench/rows_split_bench tools/ubench/rows_split_bench.hip
//   run:    tools/ubench/rows_split_bench [Fin] [Fout] [rows]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../adaptive-graph-convolutional-network_amd/csrc/rows_split.hip"

extern "C" int32_t agcn_get_rows_split(void) { return 1; }
namespace agcn {
int fail(int code, const char* fmt, ...) {
  va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr);
  return code;
}
int ensure_dynamic_lds(const void* kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  return (int)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
DeviceInfo current_device_info() { return DeviceInfo{256, nullptr}; }
bool prof_on() { return false; }
void prof_begin(int, void*) {}
void prof_end(void*) {}
}  // namespace agcn

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

using namespace agcn;

static std::vector<float> rnd(size_t n, unsigned seed, float scale = 1.f) {
  std::mt19937 g(seed);
  std::normal_distribution<float> d(0.f, scale);
  std::vector<float> v(n);
  for (auto& x : v) x = d(g);
  return v;
}
template <class T> static T* up(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}
static std::vector<float> down(const float* d, size_t n) {
  std::vector<float> h(n);
  (void)hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost);
  return h;
}
template <class F> static float time_us(F f, int reps = 50) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) f();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / reps;
}
// max over sampled rows of |got - want| / (row scale)
struct Err { double worst = 0; void add(double got, double want, double scale) { worst = std::max(worst, std::fabs(got - want) / scale); } };

int main(int argc, char** argv) {
  const int Fin = argc > 1 ? atoi(argv[1]) : 128, Fout = argc > 2 ? atoi(argv[2]) : 128, rows = argc > 3 ? atoi(argv[3]) : 32768;
  printf("rows_split_bench: rows %d, Fin %d, Fout %d\n", rows, Fin, Fout);
  const auto X = rnd((size_t)rows * Fin, 1), X1 = rnd((size_t)rows * Fin, 2), M = rnd((size_t)Fin * Fin, 3, 0.1f);
  const auto W = rnd((size_t)2 * Fin * Fout, 4, 0.1f), bias = rnd(Fout, 5), dY = rnd((size_t)rows * Fout, 6), dZ = rnd((size_t)rows * Fin, 7);
  float *dX_ = up(X), *dX1 = up(X1), *dM = up(M), *dW = up(W), *dB = up(bias), *ddY = up(dY), *ddZ = up(dZ);
  float *Z, *Y, *Gp, *gX, *g1;
  CHECK(hipMalloc(&Z, (size_t)rows * Fin * 4)); CHECK(hipMalloc(&Y, (size_t)rows * Fout * 4)); CHECK(hipMalloc(&Gp, (size_t)rows * Fout * 4));
  CHECK(hipMalloc(&gX, (size_t)rows * Fin * 4)); CHECK(hipMalloc(&g1, (size_t)rows * Fin * 4));
  void *imgF, *imgB;
  CHECK(hipMalloc(&imgF, rows_split_fwd_image_bytes(Fin, Fout))); CHECK(hipMalloc(&imgB, rows_split_bwd_image_bytes(Fin, Fout)));
  if (!dX_ || !dX1 || !dM || !dW || !dB || !ddY || !ddZ) return 1;

  int rc = launch_rows_split_fwd_images(dM, dW, Fin, Fout, imgF, nullptr);
  rc |= launch_rows_split_bwd_images(dM, dW, Fin, Fout, true, imgB, nullptr);
  rc |= launch_rows_split_project(dX_, Fin, Z, Fin, rows, imgF, nullptr);
  rc |= launch_rows_split_filter(dX_, Fin, dX1, Fin, dB, 1, Y, Fout, Fin, Fout, rows, imgF, nullptr);
  rc |= launch_rows_split_filter_bwd(ddY, Y, Fout, 1, Gp, gX, Fin, g1, Fin, Fout, rows, imgB, nullptr);
  CHECK(hipDeviceSynchronize());
  if (rc) { printf("launch failed\n"); return 1; }
  const auto hZ = down(Z, (size_t)rows * Fin), hY = down(Y, (size_t)rows * Fout), hGp = down(Gp, (size_t)rows * Fout);
  const auto hg0 = down(gX, (size_t)rows * Fin), hg1 = down(g1, (size_t)rows * Fin);
  rc = launch_rows_split_project_bwd(ddZ, gX, Fin, Fin, Fout, rows, imgB, nullptr);
  CHECK(hipDeviceSynchronize());
  const auto hdX = down(gX, (size_t)rows * Fin);

  Err eZ, eY, eG, e0, e1, eD;
  for (int s = 0; s < 97; ++s) {
    const int i = (int)(((int64_t)s * 1000003) % rows);
    const int ii = s == 0 ? rows - 1 : s == 1 ? 0 : i;
    double sx = 0;
    for (int k = 0; k < Fin; ++k) sx = std::max(sx, (double)std::fabs(X[(size_t)ii * Fin + k]));
    for (int c = 0; c < Fin; ++c) {
      double z = 0;
      for (int k = 0; k < Fin; ++k) z += (double)X[(size_t)ii * Fin + k] * M[(size_t)k * Fin + c];
      eZ.add(hZ[(size_t)ii * Fin + c], z, sx);
    }
    std::vector<double> gp(Fout);
    for (int c = 0; c < Fout; ++c) {
      double y = bias[c];
      for (int f = 0; f < Fin; ++f)
        y += (double)X[(size_t)ii * Fin + f] * W[(size_t)(2 * f) * Fout + c] + (double)X1[(size_t)ii * Fin + f] * W[(size_t)(2 * f + 1) * Fout + c];
      y = std::max(y, 0.0);
      eY.add(hY[(size_t)ii * Fout + c], y, 4 * sx);
      gp[c] = hY[(size_t)ii * Fout + c] > 0.f ? dY[(size_t)ii * Fout + c] : 0.f;
      eG.add(hGp[(size_t)ii * Fout + c], gp[c], 1.0);
    }
    for (int f = 0; f < Fin; ++f) {
      double a0 = 0, a1 = 0, dz = 0;
      for (int c = 0; c < Fout; ++c) { a0 += gp[c] * W[(size_t)(2 * f) * Fout + c]; a1 += gp[c] * W[(size_t)(2 * f + 1) * Fout + c]; }
      for (int g = 0; g < Fin; ++g) dz += (double)dZ[(size_t)ii * Fin + g] * M[(size_t)f * Fin + g];
      e0.add(hg0[(size_t)ii * Fin + f], a0, 4.0);
      e1.add(hg1[(size_t)ii * Fin + f], a1, 4.0);
      eD.add(hdX[(size_t)ii * Fin + f], (double)hg0[(size_t)ii * Fin + f] + dz, 4.0);
    }
  }
  printf("  max scaled error: Z %.2e  Y %.2e  Gp %.2e  g0 %.2e  g1 %.2e  dX+= %.2e   (fp32 eps 6e-8)\n", eZ.worst, eY.worst, eG.worst, e0.worst, e1.worst, eD.worst);
  const bool ok = eZ.worst < 2e-6 && eY.worst < 2e-6 && eG.worst == 0 && e0.worst < 2e-6 && e1.worst < 2e-6 && eD.worst < 2e-6;
  printf("  %s\n", ok ? "numerics OK" : "NUMERICS MISMATCH");

  const double mb = 1e-6;
  auto line = [&](const char* name, float us, double bytes) { printf("  %-28s %7.2f us   %6.1f MB  %5.2f TB/s\n", name, us, bytes * mb, bytes / us * 1e-6); };
  line("presplit fwd images", time_us([&] { launch_rows_split_fwd_images(dM, dW, Fin, Fout, imgF, nullptr); }), 0);
  line("presplit bwd images", time_us([&] { launch_rows_split_bwd_images(dM, dW, Fin, Fout, true, imgB, nullptr); }), 0);
  line("project", time_us([&] { launch_rows_split_project(dX_, Fin, Z, Fin, rows, imgF, nullptr); }), 8.0 * rows * Fin);
  line("filter", time_us([&] { launch_rows_split_filter(dX_, Fin, dX1, Fin, dB, 1, Y, Fout, Fin, Fout, rows, imgF, nullptr); }), 4.0 * rows * (2 * Fin + Fout));
  line("filter_bwd", time_us([&] { launch_rows_split_filter_bwd(ddY, Y, Fout, 1, Gp, gX, Fin, g1, Fin, Fout, rows, imgB, nullptr); }), 4.0 * rows * (3 * Fout + 2 * Fin));
  line("project_bwd", time_us([&] { launch_rows_split_project_bwd(ddZ, gX, Fin, Fin, Fout, rows, imgB, nullptr); }), 12.0 * rows * Fin);
  // parameter-gradient partials: X_0 = X, X_1 = X1, Gp as stored above, dZ
  bool okp = true;
  {
    ParamGradArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.rows = rows; pa.Fin = Fin; pa.Fout = Fout; pa.K = 2; pa.ldx = Fin; pa.paper = 1;
    pa.n_chunks = 256;
    pa.part_stride = ((int64_t)2 * Fin * Fout + Fout + (int64_t)Fin * Fin + 3) / 4 * 4;
    pa.X = dX_; pa.sv_X = dX1; pa.Gp = Gp; pa.dZ = ddZ;
    float* part;
    CHECK(hipMalloc(&part, (size_t)256 * pa.part_stride * 4));
    CHECK(hipMemset(part, 0xff, (size_t)256 * pa.part_stride * 4));
    pa.part = part;
    ParamGradArgs p1 = pa;
    if (launch_rows_split_param_partials(p1, nullptr) != 0) { printf("param launch failed\n"); return 1; }
    CHECK(hipDeviceSynchronize());
    printf("  param partials: %d chunks of %d rows\n", p1.n_chunks, p1.chunk_rows);
    const auto hp = down(part, (size_t)p1.n_chunks * pa.part_stride);
    auto total = [&](int64_t e) { double t = 0; for (int c = 0; c < p1.n_chunks; ++c) t += hp[(size_t)c * pa.part_stride + e]; return t; };
    Err eW, eB, eM;
    for (int sidx = 0; sidx < 48; ++sidx) {
      const int f = (sidx * 37) % Fin, c = (sidx * 53) % Fout, g = (sidx * 29) % Fin, t = sidx & 1;
      double w = 0, b = 0, m = 0, sw = 0, sm = 0;
      const float* Xt = t ? X1.data() : X.data();
      for (int i = 0; i < rows; ++i) {
        const double gp = hGp[(size_t)i * Fout + c];
        w += (double)Xt[(size_t)i * Fin + f] * gp; sw += std::fabs((double)Xt[(size_t)i * Fin + f] * gp);
        b += gp;
        m += (double)X[(size_t)i * Fin + f] * dZ[(size_t)i * Fin + g]; sm += std::fabs((double)X[(size_t)i * Fin + f] * dZ[(size_t)i * Fin + g]);
      }
      eW.add(total((int64_t)(f * 2 + t) * Fout + c), w, sw);
      eB.add(total((int64_t)2 * Fin * Fout + c), b, sw);
      eM.add(total((int64_t)2 * Fin * Fout + Fout + (int64_t)f * Fin + g), m, sm);
    }
    printf("  max scaled error (sum of |terms|): dweight %.2e  dbias %.2e  dM_L %.2e\n", eW.worst, eB.worst, eM.worst);
    okp = eW.worst < 1e-6 && eB.worst < 1e-6 && eM.worst < 1e-6;
    printf("  %s\n", okp ? "param numerics OK" : "PARAM NUMERICS MISMATCH");
    line("param partials", time_us([&] { ParamGradArgs p2 = pa; launch_rows_split_param_partials(p2, nullptr); }), 4.0 * rows * (3.0 * Fin + Fout + Fin + Fout));
  }
  return ok && okp ? 0 : 2;
}
