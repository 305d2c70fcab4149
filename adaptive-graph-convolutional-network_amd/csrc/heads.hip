// Classifier tail of every Tox21-style network: T two-class task heads on the per-graph feature vector and
// the weighted sigmoid cross-entropy training loss.
//   logits_t = feat . W[t] + b[t]                 model_operatos.py:710-759 (fully_connected_layer: xw_plus_b),
//                                                 :792-864 (multitask_logits: one head per task)
//   loss = (1 / batch_size) sum_t sum_{b,c} w[b,t] sce(logits_t[b,c], one_hot(y[b,t], 2)[c])
//                                                 multitask_classifier.py:41-44, :187-209
// B x F x 2T is a few hundred kFLOP: plain FMA loops out of LDS, one launch (+ one tiny fixed-order
// reduction) each way.  No atomics: bit-reproducible.
#include <hip/hip_runtime.h>

#include "agcn_device.h"
#include "sgc_kernels.h"

namespace agcn {
namespace {
constexpr int HR = 8;   // feature rows per workgroup

struct HeadArgs {
  int32_t B, F, T, ldf;
  float inv_batch;
  const float *feat, *W, *bias, *labels, *weights, *logits_in, *dloss;
  float *logits, *loss, *part, *dlog, *dfeat, *dW, *dbias;
  int32_t n_blocks;
};

__device__ __forceinline__ float sce(float x, float z) {
  // tf.nn.sigmoid_cross_entropy_with_logits: max(x, 0) - x z + log(1 + exp(-|x|))
  return fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
}

// grid: ceil(B / HR) workgroups of 256 threads; thread (r = tid / 32, j = tid % 32) owns logit column
// j = 2 t + c of row r.
__global__ __launch_bounds__(256) void head_forward_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ float red[32];
  const int tid = threadIdx.x, F = a.F, J = 2 * a.T;
  float* Fs = smem;              // [HR][F]
  const int b0 = blockIdx.x * HR;
  for (int e = tid; e < HR * F; e += 256) {
    const int r = e / F, f = e - r * F;
    Fs[e] = (b0 + r < a.B) ? a.feat[(int64_t)(b0 + r) * a.ldf + f] : 0.f;
  }
  __syncthreads();
  const int r = tid >> 5, b = b0 + r;
  float term = 0.f;
  for (int j = tid & 31; j < J; j += 32) {
    const int t = j >> 1, c = j & 1;
    const float* w = a.W + (int64_t)t * F * 2 + c;
    float acc = 0.f;
    for (int f = 0; f < F; ++f) acc = fmaf(Fs[r * F + f], w[2 * f], acc);
    const float x = acc + a.bias[j];
    if (b < a.B) {
      a.logits[((int64_t)t * a.B + b) * 2 + c] = x;
      const float z = ((int)a.labels[(int64_t)b * a.T + t] == c) ? 1.f : 0.f;   // tf.one_hot(to_int32(y), 2)
      term += a.weights[(int64_t)b * a.T + t] * sce(x, z);
    }
  }
  term = block_sum(term, red);
  if (tid == 0) a.part[blockIdx.x] = term;
}

// loss = inv_batch * sum of the per-workgroup partials, in index order
__global__ __launch_bounds__(64) void head_loss_kernel(HeadArgs a) {
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < a.n_blocks; ++i) s += a.part[i];
    a.loss[0] = s * a.inv_batch;      // tf.div(total_loss, batch_size)
  }
}

// dlogit[b][j] = dloss * inv_batch * w[b,t] * (sigmoid(x) - z);  dfeat[b, f] = sum_j dlogit[b][j] W[t][f][c]
__global__ __launch_bounds__(256) void head_backward_rows_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, F = a.F, J = 2 * a.T;
  float* Ds = smem;              // [HR][J]
  const int b0 = blockIdx.x * HR;
  const float g = a.dloss[0] * a.inv_batch;
  for (int e = tid; e < HR * J; e += 256) {
    const int r = e / J, j = e - r * J, b = b0 + r;
    float d = 0.f;
    if (b < a.B) {
      const int t = j >> 1, c = j & 1;
      const float x = a.logits_in[((int64_t)t * a.B + b) * 2 + c];
      const float z = ((int)a.labels[(int64_t)b * a.T + t] == c) ? 1.f : 0.f;
      d = g * a.weights[(int64_t)b * a.T + t] * (1.f / (1.f + expf(-x)) - z);
      a.dlog[(int64_t)b * J + j] = d;
    }
    Ds[e] = d;
  }
  __syncthreads();
  for (int e = tid; e < HR * F; e += 256) {
    const int r = e / F, f = e - r * F, b = b0 + r;
    if (b >= a.B) continue;
    float acc = 0.f;
    for (int j = 0; j < J; ++j) acc = fmaf(Ds[r * J + j], a.W[((int64_t)(j >> 1) * F + f) * 2 + (j & 1)], acc);
    a.dfeat[(int64_t)b * a.ldf + f] = acc;
  }
}

// dW[t][f][c] = sum_b feat[b, f] dlogit[b][2t + c] (b ascending);  dbias[t][c] = sum_b dlogit[b][2t + c].
// One thread per output element; element index e = (t * (F + 1) + f) * 2 + c with f == F for the bias.
__global__ __launch_bounds__(256) void head_backward_params_kernel(HeadArgs a) {
  const int F = a.F, J = 2 * a.T;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.T * (F + 1) * 2) return;
  const int c = e & 1, tf = e >> 1, t = tf / (F + 1), f = tf - t * (F + 1);
  const int j = 2 * t + c;
  float acc = 0.f;
  if (f < F) {
    for (int b = 0; b < a.B; ++b) acc = fmaf(a.feat[(int64_t)b * a.ldf + f], a.dlog[(int64_t)b * J + j], acc);
    a.dW[((int64_t)t * F + f) * 2 + c] = acc;
  } else {
    for (int b = 0; b < a.B; ++b) acc += a.dlog[(int64_t)b * J + j];
    a.dbias[j] = acc;
  }
}

int check_head(int32_t B, int32_t F, int32_t T, int32_t ldf, const char* who) {
  if (B < 1 || F < 1 || T < 1 || F > 8192 || T > 4096) return fail(AGCN_EINVAL, "%s: bad B/F/T (%d, %d, %d)", who, B, F, T);
  if (ldf < F) return fail(AGCN_EINVAL, "%s: ldf smaller than F", who);
  if ((size_t)HR * F * sizeof(float) > 64 * 1024) return fail(AGCN_EINVAL, "%s: F too large for the row image", who);
  return AGCN_OK;
}

struct HeadWs { float *part, *dlog; };
size_t carve_head(int32_t B, int32_t T, void* ws, HeadWs* w) {
  char* base = static_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t floats) { float* r = base ? reinterpret_cast<float*>(base + off) : nullptr;
                                   off += align256(floats * sizeof(float)); return r; };
  w->part = take((size_t)(B + HR - 1) / HR);
  w->dlog = take((size_t)B * 2 * T);
  return off;
}
}  // namespace
}  // namespace agcn

using namespace agcn;

extern "C" size_t agcn_multitask_ce_workspace_bytes(int32_t B, int32_t F, int32_t T) {
  if (check_head(B, F, T, F, "agcn_multitask_ce_workspace_bytes") != AGCN_OK) return 0;
  HeadWs w;
  return carve_head(B, T, nullptr, &w) + 256;
}

extern "C" int agcn_multitask_ce_forward(int32_t B, int32_t F, int32_t T, const float* feat, int32_t ldf,
                                         const float* W, const float* bias, const float* labels,
                                         const float* weights, float inv_batch, float* logits, float* loss,
                                         void* workspace, size_t workspace_bytes, void* hip_stream) {
  const char* who = "agcn_multitask_ce_forward";
  int rc = check_head(B, F, T, ldf, who);
  if (rc != AGCN_OK) return rc;
  if (!feat || !W || !bias || !labels || !weights || !logits || !loss) return fail(AGCN_EINVAL, "%s: null tensor", who);
  HeadWs w;
  const size_t need = carve_head(B, T, workspace, &w);
  if (!workspace || workspace_bytes < need) return fail(AGCN_ENOSPACE, "%s: workspace too small (%zu < %zu)", who, workspace_bytes, need);
  HeadArgs a;
  std::memset(&a, 0, sizeof(a));
  a.B = B; a.F = F; a.T = T; a.ldf = ldf; a.inv_batch = inv_batch;
  a.feat = feat; a.W = W; a.bias = bias; a.labels = labels; a.weights = weights;
  a.logits = logits; a.loss = loss; a.part = w.part;
  a.n_blocks = (B + HR - 1) / HR;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(head_forward_kernel, dim3(a.n_blocks), dim3(256), sizeof(float) * HR * F, s, a);
  hipLaunchKernelGGL(head_loss_kernel, dim3(1), dim3(64), 0, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: %s", who, hipGetErrorString(e));
  return AGCN_OK;
}

extern "C" int agcn_multitask_ce_backward(int32_t B, int32_t F, int32_t T, const float* feat, int32_t ldf,
                                          const float* W, const float* logits, const float* labels,
                                          const float* weights, float inv_batch, const float* dloss, float* dfeat,
                                          float* dW, float* dbias, void* workspace, size_t workspace_bytes,
                                          void* hip_stream) {
  const char* who = "agcn_multitask_ce_backward";
  int rc = check_head(B, F, T, ldf, who);
  if (rc != AGCN_OK) return rc;
  if (!feat || !W || !logits || !labels || !weights || !dloss || !dfeat || !dW || !dbias)
    return fail(AGCN_EINVAL, "%s: null tensor", who);
  HeadWs w;
  const size_t need = carve_head(B, T, workspace, &w);
  if (!workspace || workspace_bytes < need) return fail(AGCN_ENOSPACE, "%s: workspace too small (%zu < %zu)", who, workspace_bytes, need);
  HeadArgs a;
  std::memset(&a, 0, sizeof(a));
  a.B = B; a.F = F; a.T = T; a.ldf = ldf; a.inv_batch = inv_batch;
  a.feat = feat; a.W = W; a.labels = labels; a.weights = weights; a.logits_in = logits; a.dloss = dloss;
  a.dlog = w.dlog; a.dfeat = dfeat; a.dW = dW; a.dbias = dbias;
  a.n_blocks = (B + HR - 1) / HR;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  hipLaunchKernelGGL(head_backward_rows_kernel, dim3(a.n_blocks), dim3(256), sizeof(float) * HR * 2 * T, s, a);
  const int total = T * (F + 1) * 2;
  hipLaunchKernelGGL(head_backward_params_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: %s", who, hipGetErrorString(e));
  return AGCN_OK;
}
