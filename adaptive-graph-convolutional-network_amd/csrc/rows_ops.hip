// The layers that sit next to the SGC layers in every reference network (SURVEY.md 8(f) "next" rows):
//   rows_dense  : Y = act(A . W [+ b] [+ Add]) over the packed row space -- DenseMol
//                 (models/layers/dense_layer.py:33-50: no activation, bias also lands on pad rows) and
//                 BlockEnd (models/layers/blockend.py:67-86: x + x_res . W, activation, pad rows 0)
//   graph_pool  : GraphPoolMol (models/layers/graphpool.py:55-110): per node, max over the rows of X
//                 selected by the non-zeros of the node's Laplacian row (self included via L_ii != 0)
// Built on the packed GEMM (rows_gemm.hip) and the parameter-gradient kernels (param_grads.hip).
#include <hip/hip_runtime.h>

#include "agcn_device.h"
#include "sgc_kernels.h"

namespace agcn {
namespace {

int check_dense(const AgcnDenseDesc* d, const char* who) {
  if (!d) return fail(AGCN_EINVAL, "%s: null descriptor", who);
  if (d->Fin < 1 || d->Fin > 1024 || d->Fout < 1 || d->Fout > 1024)
    return fail(AGCN_EINVAL, "%s: Fin/Fout must be in [1, 1024]", who);
  if (d->activation != AGCN_ACT_LINEAR && d->activation != AGCN_ACT_RELU)
    return fail(AGCN_EINVAL, "%s: unsupported activation %d (linear or relu)", who, d->activation);
  if (d->ldx < d->Fin || d->ldy < d->Fout) return fail(AGCN_EINVAL, "%s: ldx/ldy smaller than Fin/Fout", who);
  return AGCN_OK;
}

void fill_rows_dense(RowsGemmArgs& g, const AgcnDenseDesc* d, const PlanView& host, const PlanView& dev) {
  std::memset(&g, 0, sizeof(g));
  g.rows = host.info.rows; g.Fin = d->Fin; g.Fout = d->Fout; g.K = 1; g.activation = d->activation;
  g.row_stride = (host.info.rows != host.info.N1) ? host.info.rows / host.info.B : 0;
  g.n = dev.n;
  g.keep_pad = d->keep_pad_rows ? 1 : 0;
}

struct DenseWs { float *Gp, *part, *dbias, *img; int n_chunks; int64_t part_stride; size_t part_floats; };
// packed rows (no pad rows) at point-cloud row counts: the products may run on the split-bf16 row kernels (rows_split.hip)
// (the addend's stride is tested by the forward itself, and only when the call HAS an addend)
bool dense_split(const AgcnDenseDesc* d, const AgcnPlanInfo& pi) {
  return pi.rows == pi.N1 && rows_split_dense_ok(pi.rows, d->Fin, d->Fout, d->ldx, d->ldy, 0);
}
size_t carve_dense(const AgcnDenseDesc* d, const AgcnPlanInfo& pi, void* ws, bool need_gp, DenseWs* w) {
  char* base = static_cast<char*>(ws);
  size_t off = 0;
  auto take = [&](size_t floats) { float* r = base ? reinterpret_cast<float*>(base + off) : nullptr;
                                   off += align256(floats * sizeof(float)); return r; };
  w->Gp = need_gp ? take((size_t)pi.rows * d->Fout) : nullptr;
  int c = (pi.rows + 63) / 64;
  if (c < 1) c = 1;
  if (c > 256) c = 256;
  w->n_chunks = c;
  w->part_stride = ((int64_t)d->Fin * d->Fout + d->Fout + 3) / 4 * 4;
  w->part_floats = std::max((size_t)c * w->part_stride, param_grad_onepass_floats(pi.rows, d->Fin, d->Fout, 1, 0));
  w->part = take(w->part_floats);
  w->dbias = take((size_t)d->Fout);
  w->img = dense_split(d, pi) ? take(rows_split_dense_image_bytes(d->Fin, d->Fout) / 4) : nullptr;
  return off;
}

// ds = sum_e W[e] U[e] (per-thread sequential partial + fixed tree), then U <- s U in place.
__global__ __launch_bounds__(1024) void dense_scale_grad_kernel(const float* W, const float* scale, float* U,
                                                                float* dscale, int total) {
  __shared__ float red[32];
  const float s = scale[0];
  float acc = 0.f;
  for (int e = threadIdx.x; e < total; e += 1024) acc += W[e] * U[e];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) dscale[0] = acc;
  for (int e = threadIdx.x; e < total; e += 1024) U[e] = s * U[e];
}

// ---- GraphPoolMol ---------------------------------------------------------------------------------------
struct PoolArgs {
  const int32_t *n, *row_off;
  const float* X;
  int32_t F, ldx;
  const float* L0_dense; int32_t L0_ld;
  const int64_t* tile_off;
  const int32_t* L0_rowptr; const int32_t* L0_col; const float* L0_val;
  float* out;
  int32_t* argmax;
  const float* dOut;
  float* dX;
};

// one workgroup per graph; wave = node, lanes = features
__global__ __launch_bounds__(256) void graph_pool_fwd_kernel(PoolArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int nb = a.n[b], r0 = a.row_off[b], owned = a.row_off[b + 1] - r0;
  const float* Xg = a.X + (int64_t)r0 * a.ldx;
  for (int i = wave; i < owned; i += nw) {
    for (int f = lane; f < a.F; f += 64) {
      float best = 0.f;
      int arg = -1;
      if (i < nb) {
        if (a.L0_rowptr) {
          const int p0 = a.L0_rowptr[r0 + i], p1 = a.L0_rowptr[r0 + i + 1];
          for (int p = p0; p < p1; ++p) {
            if (a.L0_val[p] != 0.f) {          // np.nonzero(l), graphpool.py:93
              const int j = a.L0_col[p];
              const float v = Xg[(int64_t)j * a.ldx + f];
              if (arg < 0 || v > best) { best = v; arg = j; }
            }
          }
        } else {
          const float* Lr = a.L0_ld > 0 ? a.L0_dense + (int64_t)b * a.L0_ld * a.L0_ld + (int64_t)i * a.L0_ld
                                        : a.L0_dense + a.tile_off[b] + (int64_t)i * nb;
          for (int j = 0; j < nb; ++j) {
            if (Lr[j] != 0.f) {
              const float v = Xg[(int64_t)j * a.ldx + f];
              if (arg < 0 || v > best) { best = v; arg = j; }
            }
          }
        }
        if (arg < 0) { best = Xg[(int64_t)i * a.ldx + f]; arg = i; }   // empty row: pooled_x = x[i] (:98)
      }
      a.out[(int64_t)(r0 + i) * a.F + f] = best;                       // pad rows: 0 (tf.pad, :108)
      if (a.argmax) a.argmax[(int64_t)(r0 + i) * a.F + f] = arg;
    }
  }
}

// dX[j][f] = sum over i in N(j) u {j} with argmax[i][f] == j of dOut[i][f]   (needs a symmetric pattern)
__global__ __launch_bounds__(256) void graph_pool_bwd_kernel(PoolArgs a) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int nb = a.n[b], r0 = a.row_off[b], owned = a.row_off[b + 1] - r0;
  for (int j = wave; j < owned; j += nw) {
    for (int f = lane; f < a.F; f += 64) {
      float s = 0.f;
      if (j < nb) {
        if (a.argmax[(int64_t)(r0 + j) * a.F + f] == j) s += a.dOut[(int64_t)(r0 + j) * a.F + f];
        if (a.L0_rowptr) {
          const int p0 = a.L0_rowptr[r0 + j], p1 = a.L0_rowptr[r0 + j + 1];
          for (int p = p0; p < p1; ++p) {
            const int i = a.L0_col[p];
            if (i != j && a.L0_val[p] != 0.f && a.argmax[(int64_t)(r0 + i) * a.F + f] == j)
              s += a.dOut[(int64_t)(r0 + i) * a.F + f];
          }
        } else {
          const float* Lr = a.L0_ld > 0 ? a.L0_dense + (int64_t)b * a.L0_ld * a.L0_ld + (int64_t)j * a.L0_ld
                                        : a.L0_dense + a.tile_off[b] + (int64_t)j * nb;
          for (int i = 0; i < nb; ++i)
            if (i != j && Lr[i] != 0.f && a.argmax[(int64_t)(r0 + i) * a.F + f] == j)
              s += a.dOut[(int64_t)(r0 + i) * a.F + f];
        }
      }
      a.dX[(int64_t)(r0 + j) * a.ldx + f] = s;
    }
  }
}

}  // namespace
}  // namespace agcn

using namespace agcn;

extern "C" int agcn_rows_dense_forward(const AgcnDenseDesc* desc, const void* plan_host, const void* plan_dev,
                                       const float* X, const float* W, const float* bias, const float* Add,
                                       float* Y, void* hip_stream) {
  return agcn_rows_dense_scaled_forward(desc, plan_host, plan_dev, X, W, nullptr, bias, Add, Y, hip_stream);
}

extern "C" int agcn_rows_dense_scaled_forward(const AgcnDenseDesc* desc, const void* plan_host,
                                              const void* plan_dev, const float* X, const float* W,
                                              const float* scale, const float* bias, const float* Add, float* Y,
                                              void* hip_stream) {
  return agcn_rows_dense_scaled_forward_ws(desc, plan_host, plan_dev, X, W, scale, bias, Add, Y, nullptr, 0, hip_stream);
}

extern "C" size_t agcn_rows_dense_forward_workspace_bytes(const AgcnDenseDesc* desc, const void* plan_host) {
  KnobScope knobs(desc ? desc->flags : 0u);
  AgcnPlanInfo pi;
  if (check_dense(desc, "agcn_rows_dense_forward_workspace_bytes") != AGCN_OK || agcn_plan_info(plan_host, &pi) != AGCN_OK) return 0;
  return dense_split(desc, pi) ? rows_split_dense_image_bytes(desc->Fin, desc->Fout) + 256 : 0;
}

extern "C" int agcn_rows_dense_scaled_forward_ws(const AgcnDenseDesc* desc, const void* plan_host,
                                                 const void* plan_dev, const float* X, const float* W,
                                                 const float* scale, const float* bias, const float* Add, float* Y,
                                                 void* workspace, size_t workspace_bytes, void* hip_stream) {
  KnobScope knobs(desc ? desc->flags : 0u);
  const char* who = "agcn_rows_dense_forward";
  int rc = check_dense(desc, who);
  if (rc != AGCN_OK) return rc;
  if (!X || !W || !Y) return fail(AGCN_EINVAL, "%s: null tensor", who);
  if (Add && desc->ldadd < desc->Fout) return fail(AGCN_EINVAL, "%s: ldadd smaller than Fout", who);
  PlanView host, dev;
  rc = plan_view(plan_host, plan_host, &host);
  if (rc != AGCN_OK) return rc;
  plan_view(plan_host, plan_dev, &dev);
  // point-cloud row counts, packed rows, a workspace for the parameter image: the bf16 matrix pipe at fp32 accuracy (rows_split.hip)
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (workspace != nullptr && dense_split(desc, host.info) && workspace_bytes >= rows_split_dense_image_bytes(desc->Fin, desc->Fout) &&
      al16(X) && al16(Y) && al16(workspace) && (!bias || al16(bias)) && (!Add || (al16(Add) && (desc->ldadd & 3) == 0)))
    return launch_rows_split_dense_fwd(X, desc->ldx, W, scale, bias, Add, desc->ldadd, desc->activation == AGCN_ACT_RELU ? 1 : 0, Y, desc->ldy,
                                       desc->Fin, desc->Fout, host.info.rows, workspace, hip_stream);
  RowsGemmArgs g;
  fill_rows_dense(g, desc, host, dev);
  g.Kd = desc->Fin; g.N = desc->Fout;
  g.A0 = X; g.lda0 = desc->ldx; g.Bm = W; g.bias = bias; g.Add = Add; g.ldadd = desc->ldadd; g.scale = scale;
  g.Out0 = Y; g.ldo0 = desc->ldy;
  return launch_rows_gemm(g, RG_DENSE, hip_stream);
}

extern "C" size_t agcn_rows_dense_backward_workspace_bytes(const AgcnDenseDesc* desc, const void* plan_host) {
  KnobScope knobs(desc ? desc->flags : 0u);
  AgcnPlanInfo pi;
  if (check_dense(desc, "agcn_rows_dense_backward_workspace_bytes") != AGCN_OK ||
      agcn_plan_info(plan_host, &pi) != AGCN_OK) return 0;
  DenseWs w;
  return carve_dense(desc, pi, nullptr, true, &w) + 256;
}

extern "C" int agcn_rows_dense_backward(const AgcnDenseDesc* desc, const void* plan_host, const void* plan_dev,
                                        const float* X, const float* W, const float* Y, const float* dY,
                                        float* dX, float* dAdd, float* dW, float* dbias, void* workspace,
                                        size_t workspace_bytes, void* hip_stream) {
  return agcn_rows_dense_scaled_backward(desc, plan_host, plan_dev, X, W, nullptr, Y, dY, dX, dAdd, dW, dbias,
                                         nullptr, workspace, workspace_bytes, hip_stream);
}

// dX = s Gp . W^T, dW = s A^T Gp, dbias = colsum(Gp), dAdd = Gp, ds = <W, A^T Gp>, with Gp = dY . act'(Y)
// and s = scale[0] (1 when scale is NULL).
extern "C" int agcn_rows_dense_scaled_backward(const AgcnDenseDesc* desc, const void* plan_host,
                                               const void* plan_dev, const float* X, const float* W,
                                               const float* scale, const float* Y, const float* dY, float* dX,
                                               float* dAdd, float* dW, float* dbias, float* dscale,
                                               void* workspace, size_t workspace_bytes, void* hip_stream) {
  KnobScope knobs(desc ? desc->flags : 0u);
  const char* who = "agcn_rows_dense_backward";
  if ((scale == nullptr) != (dscale == nullptr))
    return fail(AGCN_EINVAL, "%s: scale and dscale go together", who);
  int rc = check_dense(desc, who);
  if (rc != AGCN_OK) return rc;
  if (!X || !W || !Y || !dY || !dX || !dW) return fail(AGCN_EINVAL, "%s: null tensor", who);
  PlanView host, dev;
  rc = plan_view(plan_host, plan_host, &host);
  if (rc != AGCN_OK) return rc;
  plan_view(plan_host, plan_dev, &dev);
  DenseWs w;
  const size_t need = carve_dense(desc, host.info, workspace, dAdd == nullptr, &w);
  if (!workspace || workspace_bytes < need) return fail(AGCN_ENOSPACE, "%s: workspace too small (%zu < %zu)", who, workspace_bytes, need);
  float* Gp = dAdd ? dAdd : w.Gp;   // dAdd must be contiguous [rows, Fout]

  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const bool split = w.img != nullptr && al16(dY) && al16(Y) && al16(Gp) && al16(dX);
  RowsGemmArgs g;
  fill_rows_dense(g, desc, host, dev);
  g.Kd = desc->Fout; g.N = desc->Fin;
  g.A0 = dY; g.lda0 = desc->ldy; g.Yact = Y; g.ldy = desc->ldy; g.Bm = W;
  g.Out0 = dX; g.ldo0 = desc->ldx; g.Gp = Gp; g.scale = scale;
  // the hand-off of the parameter-gradient kernel behind it: tagged granules cleared by this launch's workgroups (general
  // kernel only: the narrow kernels of few-column products keep the arrival counters), else arrival counters
  int32_t n16 = 0;
  float* tags = (desc->Fin > 8 && !split) ? param_grad_tag_region(w.part, w.part_floats, host.info.rows, desc->Fin, desc->Fout, 1, 0, &n16) : nullptr;
  if (tags != nullptr) { g.ztag_ptr = tags; g.ztag_count16 = n16; }
  else g.zero_ptr = param_grad_counters(w.part, w.part_floats, host.info.rows, desc->Fin, desc->Fout, 1, 0, &g.zero_count);
  if (split)       // (rows_split.hip: Gp and dX = [scale] Gp W^T; its first workgroup zeroes the parameter-gradient kernel's counters)
    rc = launch_rows_split_dense_bwd(dY, Y, desc->ldy, desc->activation == AGCN_ACT_RELU ? 1 : 0, W, scale, Gp, dX, desc->ldx, desc->Fin,
                                     desc->Fout, host.info.rows, w.img, g.zero_ptr, g.zero_count, hip_stream);
  else
    rc = launch_rows_gemm(g, RG_FILTER_BWD, hip_stream);
  if (rc != AGCN_OK) return rc;

  ParamGradArgs pa;
  std::memset(&pa, 0, sizeof(pa));
  pa.rows = host.info.rows; pa.Fin = desc->Fin; pa.Fout = desc->Fout; pa.K = 1; pa.ldx = desc->ldx; pa.paper = 0;
  pa.n_chunks = w.n_chunks;
  pa.chunk_rows = ((host.info.rows + w.n_chunks - 1) / w.n_chunks + 15) / 16 * 16;
  pa.B = 0; pa.part_stride = w.part_stride;
  pa.X = X; pa.Gp = Gp; pa.part = w.part; pa.part_floats = w.part_floats;
  pa.counters_zeroed = tags ? 0 : 1; pa.tags_zeroed = tags ? 1 : 0;
  pa.dev_status = current_device_info().status_word; pa.ticks = agcn_get_coop_timeout_ticks();
  pa.dweight = dW; pa.dbias = dbias ? dbias : w.dbias;
  rc = launch_param_grads(pa, hip_stream);
  if (rc != AGCN_OK || !scale) return rc;
  // dW holds U = A^T Gp: ds = <W, U>, then dW = s U  (one workgroup, fixed order)
  hipLaunchKernelGGL(dense_scale_grad_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(hip_stream), W, scale,
                     dW, dscale, desc->Fin * desc->Fout);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: %s", who, hipGetErrorString(e));
  return AGCN_OK;
}

static int pool_args(PoolArgs& a, const void* plan_host, const void* plan_dev, const AgcnGraphBatch* lap,
                     int32_t F, int32_t ldx, const char* who, PlanView* host) {
  PlanView dev;
  int rc = plan_view(plan_host, plan_host, host);
  if (rc != AGCN_OK) return rc;
  plan_view(plan_host, plan_dev, &dev);
  if (!lap || (!lap->L0_dense && !lap->L0_rowptr)) return fail(AGCN_EINVAL, "%s: GraphPoolMol needs the Laplacian", who);
  if (lap->L0_rowptr && (!lap->L0_col || !lap->L0_val)) return fail(AGCN_EINVAL, "%s: incomplete CSR Laplacian", who);
  if (F < 1 || ldx < F) return fail(AGCN_EINVAL, "%s: bad F/ldx", who);
  std::memset(&a, 0, sizeof(a));
  a.n = dev.n; a.row_off = dev.row_off; a.tile_off = dev.tile_off;
  a.F = F; a.ldx = ldx;
  a.L0_dense = lap->L0_dense; a.L0_ld = lap->L0_ld;
  a.L0_rowptr = lap->L0_rowptr; a.L0_col = lap->L0_col; a.L0_val = lap->L0_val;
  return AGCN_OK;
}

// out [rows, F] contiguous; argmax [rows, F] int32 or NULL.  `lap`: only the L0_* members are read.
extern "C" int agcn_graph_pool_forward(const void* plan_host, const void* plan_dev, const AgcnGraphBatch* lap,
                                       const float* X, int32_t F, int32_t ldx, float* out, int32_t* argmax,
                                       void* hip_stream) {
  const char* who = "agcn_graph_pool_forward";
  PoolArgs a;
  PlanView host;
  int rc = pool_args(a, plan_host, plan_dev, lap, F, ldx, who, &host);
  if (rc != AGCN_OK) return rc;
  if (!X || !out) return fail(AGCN_EINVAL, "%s: null tensor", who);
  a.X = X; a.out = out; a.argmax = argmax;
  hipLaunchKernelGGL(graph_pool_fwd_kernel, dim3(host.info.B), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: %s", who, hipGetErrorString(e));
  return AGCN_OK;
}

// Arg-max routing.  NOT what the reference does (its py_func passes no gradient at all); offered as the
// differentiable extension.  Needs a symmetric sparsity pattern (L0_symmetric == 1).
extern "C" int agcn_graph_pool_backward(const void* plan_host, const void* plan_dev, const AgcnGraphBatch* lap,
                                        const int32_t* argmax, const float* dOut, int32_t F, int32_t ldx,
                                        float* dX, void* hip_stream) {
  const char* who = "agcn_graph_pool_backward";
  PoolArgs a;
  PlanView host;
  int rc = pool_args(a, plan_host, plan_dev, lap, F, ldx, who, &host);
  if (rc != AGCN_OK) return rc;
  if (!argmax || !dOut || !dX) return fail(AGCN_EINVAL, "%s: null tensor", who);
  if (lap->L0_symmetric != 1) return fail(AGCN_EINVAL, "%s: needs a symmetric Laplacian pattern (L0_symmetric)", who);
  a.argmax = const_cast<int32_t*>(argmax); a.dOut = dOut; a.dX = dX;
  hipLaunchKernelGGL(graph_pool_bwd_kernel, dim3(host.info.B), dim3(256), 0, static_cast<hipStream_t>(hip_stream), a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "%s: %s", who, hipGetErrorString(e));
  return AGCN_OK;
}
