// One batch of an epoch in ONE C-ABI call (include/agcn.h: agcn_epoch_step): stage the batch `ids` out of the HBM-resident
// store (agcn_store_stage_batch: plan + gather table into a pinned blob, one copy, one gather launch), then the layer's
// forward and backward on the staged batch -- everything the Python epoch loop did per batch around three C calls (plan
// objects, size queries, ctypes structs: 44 us of host time per batch, which bounded two batches in flight).
// Host logic only: every launch goes through the entry points a caller would use one by one.
#include "agcn_internal.h"

#include <hip/hip_runtime.h>

using namespace agcn;

extern "C" int agcn_epoch_step(const AgcnResidentStore* store, const int64_t* ids, int32_t B, const AgcnLayerDesc* desc,
                               const AgcnParams* params, const AgcnEpochBuffers* buf, int32_t restage, AgcnStagedBatch* info,
                               AgcnEpochNeeds* needs, void* slot_event, void* hip_stream) {
  const char* who = "agcn_epoch_step";
  if (!store || !ids || !desc || !params || !buf || !info || !needs || B < 1) return fail(AGCN_EINVAL, "%s: bad argument", who);
  if (desc->variant != AGCN_SGC_LL) return fail(AGCN_EINVAL, "%s: drives SGC_LL layers", who);
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  hipEvent_t ev = static_cast<hipEvent_t>(slot_event);
  // `needs` is an OUTPUT of this call only: an AGCN_ENOSPACE that leaves it zero is not a "grow the reserve" answer
  needs->reserve_bytes = 0; needs->workspace_bytes = 0;
  if (restage) {
    // a staging failure must not leave the previous batch's description behind for a retry with restage = 0
    info->plan.rows = -1; info->plan.tile_elems = -1;
    // the copy that last read this pinned blob must have executed before the blob is rewritten
    if (ev != nullptr && hipEventSynchronize(ev) != hipSuccess) return fail(AGCN_EDEVICE, "%s: hipEventSynchronize failed", who);
    int rc = agcn_store_stage_batch(store, ids, B, buf->stage_host, buf->stage_dev, buf->stage_cap, buf->X, desc->Fin,
                                    buf->rows_cap, buf->rowptr, buf->col, buf->val, buf->nnz_cap, info, hip_stream);
    if (rc != AGCN_OK) return rc;
    if (ev != nullptr && hipEventRecord(ev, s) != hipSuccess) return fail(AGCN_EDEVICE, "%s: hipEventRecord failed", who);
  }
  if (info->plan.rows < 0) return fail(AGCN_EINVAL, "%s: restage = 0 without a staged batch", who);
  if (info->plan.rows > buf->rows_cap || info->plan.tile_elems > buf->tiles_cap)
    return fail(AGCN_ENOSPACE, "%s: batch (rows %d, tile floats %lld) exceeds the output capacity (%lld, %lld)", who,
                info->plan.rows, (long long)info->plan.tile_elems, (long long)buf->rows_cap, (long long)buf->tiles_cap);
  AgcnLayerDesc d = *desc;
  d.ldx = desc->Fin; d.ldy = desc->Fout;
  const bool bf16 = desc->feat_dtype == AGCN_DTYPE_BF16;
  if (bf16 != (store->stage_bf16 != 0))
    return fail(AGCN_EINVAL, "%s: desc->feat_dtype and store->stage_bf16 disagree (bf16 layer calls need bf16 staging and vice versa)", who);
  AgcnGraphBatch gb;
  std::memset(&gb, 0, sizeof(gb));
  gb.plan_host = buf->stage_host; gb.plan_dev = buf->stage_dev;
  gb.X = buf->X;
  gb.L0_symmetric = info->symmetric;
  gb.L0_rowptr = buf->rowptr; gb.L0_col = buf->col; gb.L0_val = buf->val;
  // bf16 rows are read and written by the specialised kernels themselves; a batch they do not serve (a molecule beyond the members'
  // reach: dispatch bit 256 unset) runs the fp32 kernels between four conversions, on scratch behind the layer's own workspace
  bool convert = false;
  if (bf16) {
    int32_t di[AGCN_DISPATCH_INFO_INTS];
    int rc = agcn_sgc_ll_dispatch_info(&d, &gb, di, AGCN_DISPATCH_INFO_INTS);
    if (rc != AGCN_OK) return rc;
    convert = (di[0] & 256) == 0;
    if (convert) d.feat_dtype = AGCN_DTYPE_F32;
  }
  needs->reserve_bytes = agcn_sgc_ll_reserve_bytes(&d, buf->stage_host);
  const size_t wf = agcn_sgc_ll_forward_workspace_bytes(&d, buf->stage_host), wb = agcn_sgc_ll_backward_workspace_bytes(&d, buf->stage_host);
  const size_t wlayer = align256(wf > wb ? wf : wb);
  const size_t rows = (size_t)info->plan.rows;
  const size_t sx = align256(rows * desc->Fin * sizeof(float)), sy = align256(rows * desc->Fout * sizeof(float));
  needs->workspace_bytes = wlayer + (convert ? 2 * sx + 2 * sy : 0);
  if (needs->reserve_bytes == 0) return fail(AGCN_EINVAL, "%s: staged plan rejected", who);
  if (needs->reserve_bytes > buf->reserve_bytes || needs->workspace_bytes > buf->workspace_bytes)
    return AGCN_ENOSPACE;        // (the caller grows the buffers -- zero-filled reserve -- and calls again with restage = 0)
  AgcnOutputs out;
  std::memset(&out, 0, sizeof(out));
  out.Y = buf->Y; out.res_L = buf->res_L; out.res_W = buf->res_W;
  AgcnGrads gr;
  std::memset(&gr, 0, sizeof(gr));
  gr.dX = buf->dX; gr.dweight = buf->dweight; gr.dbias = buf->dbias; gr.dM_L = buf->dM_L; gr.dalpha = buf->dalpha;
  const float* dY = buf->dY;
  char* scratch = static_cast<char*>(buf->workspace) + wlayer;
  float *X32 = nullptr, *Y32 = nullptr, *dY32 = nullptr, *dX32 = nullptr;
  if (convert) {
    X32 = reinterpret_cast<float*>(scratch); dX32 = reinterpret_cast<float*>(scratch + sx);
    Y32 = reinterpret_cast<float*>(scratch + 2 * sx); dY32 = reinterpret_cast<float*>(scratch + 2 * sx + sy);
    int rc = agcn_cast_rows(buf->X, AGCN_DTYPE_BF16, X32, AGCN_DTYPE_F32, (int64_t)rows, desc->Fin, desc->Fin, desc->Fin, hip_stream);
    if (rc == AGCN_OK) rc = agcn_cast_rows(buf->dY, AGCN_DTYPE_BF16, dY32, AGCN_DTYPE_F32, (int64_t)rows, desc->Fout, desc->Fout, desc->Fout, hip_stream);
    if (rc != AGCN_OK) return rc;
    gb.X = X32; out.Y = Y32; gr.dX = dX32; dY = dY32;
  }
  int rc = agcn_sgc_ll_forward(&d, &gb, params, &out, buf->reserve, needs->reserve_bytes, buf->workspace, wlayer, hip_stream);
  if (rc != AGCN_OK) return rc;
  rc = agcn_sgc_ll_backward(&d, &gb, params, &out, dY, nullptr, &gr, buf->reserve, needs->reserve_bytes, buf->workspace, wlayer, hip_stream);
  if (rc != AGCN_OK || !convert) return rc;
  rc = agcn_cast_rows(Y32, AGCN_DTYPE_F32, buf->Y, AGCN_DTYPE_BF16, (int64_t)rows, desc->Fout, desc->Fout, desc->Fout, hip_stream);
  if (rc == AGCN_OK) rc = agcn_cast_rows(dX32, AGCN_DTYPE_F32, buf->dX, AGCN_DTYPE_BF16, (int64_t)rows, desc->Fin, desc->Fin, desc->Fin, hip_stream);
  return rc;
}
