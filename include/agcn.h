/*
 * agcn.h -- C ABI of the MI355X-native AGCN adaptive graph-convolution path.
 *
 * One call = one layer over a whole ragged batch of graphs (not one call per
 * graph).  Plain pointers and sizes only; the caller (PyTorch host code, via
 * ctypes) owns every buffer including workspace and reserve space; the
 * library allocates no device memory, keeps no per-call state between calls,
 * never synchronises, and enqueues all work on the stream it is given.
 *
 * Threading.  Calls are re-entrant across threads (one stream per thread, disjoint
 * buffers; tests/test_gpu_parity.py runs two threads concurrently, each with its own
 * dispatch flags); the error string is thread-local.  Which of the equivalent kernel
 * paths serves a call is a PER-CALL argument (AgcnLayerDesc.flags / AgcnDenseDesc.flags,
 * ABI 106); a field left 0 takes the process default, which comes from the AGCN_*
 * environment variables and -- for diagnostics and A/B measurements -- the agcn_set_*
 * functions of the last section.  What else is process-wide, by design: a per-(device,
 * kernel) cache of the dynamic-LDS caps already raised, and -- only with the opt-in mixed
 * dispatch, fused mode 2 | 3 -- one internal side stream + event pair per device;
 * a mixed-dispatch call holds that device's lock while it enqueues.  The launch
 * profiler (agcn_profile_*) is a single-threaded debug facility.
 *
 * What each entry point replaces in the reference (paths relative to the
 * reference repository root):
 *
 *   agcn_sgc_ll_forward    models/layers/graphconv.py:85-252  (SGC_LL.call + specgraph_LL,
 *                          incl. the tf.py_func body :163-211) and
 *                          models/layers/graphconv_reslap.py:45-230 (SGC_LL_Reslap)
 *   agcn_sgc_ll_backward   the TF autodiff of the above, driven by
 *                          models/tf_modules/multitask_classifier.py:237
 *   agcn_graph_gather_*    models/layers/graphgather.py:40-77 (GraphGatherMol)
 *   agcn_plan_*            the per-graph slice bookkeeping of
 *                          models/tf_modules/graph_topology.py:84-135 (data_slice / lap_slice)
 *
 * All functions return 0 on success or a negative AGCN_E* code;
 * agcn_last_error() returns a thread-local description of the last failure.
 */
#ifndef AGCN_H_
#define AGCN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the functions declared here are its whole dynamic symbol table
 * (tests/test_abi.py compares `nm -D` with this header). */
#if defined(__GNUC__)
#define AGCN_API __attribute__((visibility("default")))
#else
#define AGCN_API
#endif

#define AGCN_VERSION 107 /* 0.1.7: per-slab plan records (AgcnSlabDesc, AgcnPlanInfo.off_slab_desc); 106: per-call dispatch flags (AgcnLayerDesc.flags, AgcnDenseDesc.flags), hidden visibility */

enum {
  AGCN_OK = 0,
  AGCN_EINVAL = -1,    /* bad argument / unsupported combination */
  AGCN_ENOSPACE = -2,  /* workspace / reserve / plan buffer too small */
  AGCN_EDEVICE = -3,   /* HIP launch or runtime error */
  AGCN_EPLAN = -4      /* corrupt or mismatching plan blob */
};

/* laplacian_mode: how the residual Laplacian is formed from the similarity W
 * (graphconv.py:195-200).  REFERENCE reproduces the repository bit for bit:
 * `I - D * W * D` with numpy's ELEMENTWISE `*`, i.e. res_L == I, and no
 * gradient to M_L (py_func).  PAPER is the matrix product I - D^-1/2 W D^-1/2
 * the comment at graphconv.py:194 intends, differentiable w.r.t. M_L and X. */
enum { AGCN_MODE_REFERENCE = 0, AGCN_MODE_PAPER = 1 };
/* layer variant: SGC_LL (graphconv.py) or SGC_LL_Reslap (graphconv_reslap.py) */
enum { AGCN_SGC_LL = 0, AGCN_SGC_LL_RESLAP = 1 };
/* activations.get(): the networks only use these */
enum { AGCN_ACT_LINEAR = 0, AGCN_ACT_RELU = 1, AGCN_ACT_TANH = 2 };

/* ------------------------------------------------------------------------
 * Plan: the integer bookkeeping of one ragged batch (graph sizes, row and
 * tile offsets, size-sorted schedule).  Built on the host, in caller memory;
 * the caller uploads the same bytes to the device and passes both copies.
 *
 * Layout of the node-feature matrices ("rows"): graph b owns rows
 * [row_off[b], row_off[b+1]) of X / Y / dX / dY, of which the first n[b] are
 * real nodes; the rest (if any) are padding.  Packed batches have
 * row_off = prefix sums of n; the reference's padded feed
 * (graph_topology.py:84-90) has row_off[b] = b * max_atom.
 * Layout of the per-graph square tiles (res_L, res_W, L_sum, L_prev, dense
 * L_orig): graph b owns n[b]*n[b] floats, row-major with leading dimension
 * n[b], starting at tile_off[b] (each tile start is 16-byte aligned).
 */
typedef struct AgcnPlanInfo {
  int32_t B;          /* graphs */
  int32_t N1;         /* sum of n[b] (real nodes) */
  int32_t rows;       /* row_off[B] (== N1 when packed) */
  int32_t max_n;      /* largest graph */
  int64_t N2;         /* sum of n[b]^2 */
  int64_t tile_elems; /* tile_off[B]: floats to allocate for one tile array */
  int64_t plan_bytes; /* size of the blob */
  /* byte offsets of the arrays inside the blob */
  int64_t off_n;        /* int32 [B]   */
  int64_t off_row_off;  /* int32 [B+1] */
  int64_t off_tile_off; /* int64 [B+1] */
  int64_t off_order;    /* int32 [B]   graph ids, largest first */
  int64_t off_wtile_off; /* int64 [B+1] scratch-tile offsets (global-memory tile mode) */
  /* slabs: strips of AGCN_SLAB_ROWS consecutive rows of one graph's n x n tile; the unit of work
   * of the slab-parallel kernels (largest graphs first) */
  int32_t n_slabs;
  int32_t reserved1;
  int64_t off_slab_graph; /* int32 [n_slabs] graph id      */
  int64_t off_slab_row;   /* int32 [n_slabs] first row inside the graph */
  int64_t off_row_graph;  /* int32 [rows]    owning graph of every row (pad rows included) */
  int64_t off_graph_desc; /* AgcnGraphDesc [B], in schedule order (largest graph first) */
  int64_t off_slab_desc;  /* AgcnSlabDesc [n_slabs], in slab order (ABI 107) */
} AgcnPlanInfo;

/* Per-graph record of the plan, indexed by schedule position: a per-graph workgroup finds its graph with one
 * 32-byte load instead of the dependent order -> n / row_off / tile_off chain. */
typedef struct AgcnGraphDesc {
  int32_t graph;     /* graph id (= order[position]) */
  int32_t n;         /* nodes */
  int32_t row0;      /* row_off[graph] */
  int32_t reserved;
  int64_t tile_off;  /* tile_off[graph] */
  int64_t tile_end;  /* tile_off[graph + 1] */
} AgcnGraphDesc;

/* Per-slab record of the plan (ABI 107): a workgroup that owns ONE slab of a graph spread over several workgroups finds the slab and
 * its graph with one 32-byte load instead of the dependent slab_graph / slab_row -> n / row_off / tile_off chain (two memory round
 * trips in front of every other load of the workgroup: 0.6 us of a 11 us kernel). */
typedef struct AgcnSlabDesc {
  int32_t graph;     /* graph id (= slab_graph[slab]) */
  int32_t row;       /* first row of the slab inside the graph (= slab_row[slab]) */
  int32_t n;         /* nodes of the graph */
  int32_t row0;      /* row_off[graph] */
  int64_t tile_off;  /* tile_off[graph] */
  int64_t tile_end;  /* tile_off[graph + 1] */
} AgcnSlabDesc;

#define AGCN_SLAB_ROWS 16

/* Bytes needed for a plan: worst case for B graphs / exact for the given node counts. */
AGCN_API size_t agcn_plan_bytes(int32_t B);
AGCN_API size_t agcn_plan_bytes_for(const int32_t* n, int32_t B, int32_t row_stride);

/* Build a plan from host arrays.  n[B] = node counts (data_slice[:,0] /
 * lap_slice[:,0], graph_topology.py:90,98).  row_stride = 0 -> packed rows,
 * otherwise every graph owns row_stride rows (the reference's max_atom).
 * Fails with AGCN_EINVAL if any n[b] < 1, n[b] > row_stride (padded mode) or
 * the batch does not fit 32-bit row indexing. */
AGCN_API int agcn_plan_build(const int32_t* n, int32_t B, int32_t row_stride, void* plan_host, size_t plan_bytes);

/* Read the header back (validates the magic). */
AGCN_API int agcn_plan_info(const void* plan_host, AgcnPlanInfo* info);

/* ------------------------------------------------------------------------ */
typedef struct AgcnLayerDesc {
  int32_t Fin;        /* n_atom_feature  (graphconv.py:58) */
  int32_t Fout;       /* nb_filter       (graphconv.py:57) */
  int32_t K;          /* Chebyshev order (graphconv.py:61); any K >= 1 */
  int32_t mode;       /* AGCN_MODE_* */
  int32_t variant;    /* AGCN_SGC_LL / AGCN_SGC_LL_RESLAP */
  int32_t activation; /* AGCN_ACT_LINEAR or AGCN_ACT_RELU (graphconv.py:53,117-123) */
  int32_t ldx;        /* row stride of X and dX in floats (>= Fin) */
  int32_t ldy;        /* row stride of Y and dY in floats (>= Fout) */
  int32_t feat_dtype; /* AGCN_DTYPE_F32 (0), or AGCN_DTYPE_BF16: X, Y, dY, dX are bf16 matrices (row strides in ELEMENTS) --
                       * "bf16 features + fp32 Laplacian": widened exactly on load, rounded to nearest even on store, every
                       * tile, product and parameter gradient in fp32.  Served by the specialised kernels only (dispatch bits
                       * 1 | 2 of agcn_sgc_ll_dispatch_info); other batches: AGCN_EINVAL (convert with agcn_cast_rows) */
  uint32_t flags;     /* per-CALL dispatch overrides (ABI 106), 0 = the process defaults (AGCN_* environment / agcn_set_*): which of
                       * the equivalent kernel paths serves THIS call, for this thread only -- AGCN_FLAG_* below.  The size queries
                       * (reserve / workspace bytes, dispatch_info) and the forward / backward calls of one step take the same flags */
} AgcnLayerDesc;

/* AgcnLayerDesc.flags / AgcnDenseDesc.flags: each field holds (value + 1), 0 = "not overridden"; values as the setter of the
 * same name documents them (last section of this header). */
#define AGCN_FLAG_SPEC_MODE(v)     (((uint32_t)(v) + 1u) & 15u)            /* agcn_set_spec_mode: 0..7 (bit 2 off = no member workgroups) */
#define AGCN_FLAG_FUSED_MODE(v)    ((((uint32_t)(v) + 1u) & 7u) << 4)      /* agcn_set_fused_mode: 0..3 */
#define AGCN_FLAG_PG_ONEPASS(v)    ((((uint32_t)(v) + 1u) & 3u) << 7)      /* agcn_set_param_grad_onepass: 0..2 */
#define AGCN_FLAG_SPEC_SMALL(v)    ((((uint32_t)(v) + 1u) & 3u) << 9)      /* agcn_set_spec_small: 0..2 */
#define AGCN_FLAG_ROWS_SPLIT(v)    ((((uint32_t)(v) + 1u) & 3u) << 11)     /* agcn_set_rows_split: 0..1 */
#define AGCN_FLAG_SLAB_MODE(v)     ((((uint32_t)(v) + 1u) & 3u) << 13)     /* agcn_set_slab_mode: 0..1 */
#define AGCN_FLAG_FUSED_GB(v)      ((((uint32_t)((v) + 2)) & 3u) << 15)    /* agcn_set_fused_global_b: -1..1 */
#define AGCN_FLAG_SPEC_TAIL(v)     ((((uint32_t)(v) + 1u) & 3u) << 17)     /* agcn_set_spec_tail: 0..1 */
#define AGCN_FLAG_COOP_MEMBERS(v)  ((((uint32_t)(v) + 1u) & 255u) << 24)   /* agcn_set_coop_members: 0..120 */

/* Device pointers unless stated otherwise. */
typedef struct AgcnGraphBatch {
  const void* plan_host;     /* HOST copy of the plan blob */
  const void* plan_dev;      /* DEVICE copy of the same bytes */
  const float* X;            /* [rows, ldx]  node_features */
  /* original Laplacian (graph_structure.py:85-130, cast to f32 at the feed):
   * either dense or CSR, the other NULL.
   * dense: L0_ld == 0 -> packed tiles at tile_off[b] with ld n[b];
   *        L0_ld  > 0 -> padded feed, graph b at b*L0_ld*L0_ld with ld L0_ld
   *                      (graph_topology.py:92-98).
   * CSR:   rowptr[rows+1] over the row space of X (pad rows are empty),
   *        col = column index INSIDE the graph, val f32. */
  const float* L0_dense;
  int32_t L0_ld;
  int32_t L0_symmetric;      /* 1 = caller guarantees L_orig == L_orig^T (graph_structure.py:106 asserts it);
                              * enables the slab-parallel kernels, which read L_orig^T rows as L_orig rows */
  const int32_t* L0_rowptr;
  const int32_t* L0_col;
  const float* L0_val;
  const float* L_prev;       /* Reslap only: res_lap tiles (graphconv_reslap.py:68,188-190) or NULL */
  int32_t L_prev_symmetric;  /* 1 = caller guarantees L_prev == L_prev^T for every graph (true for the L_sum of a previous
                              * layer whose own inputs were symmetric); with L0_symmetric it makes the summed Laplacian
                              * symmetric, which the point-cloud kernels (graphs of hundreds of nodes and more) rely on --
                              * without it such batches take the general per-graph path.  dL_sum / dL_prev are never
                              * assumed symmetric */
  int32_t reserved0;
} AgcnGraphBatch;

typedef struct AgcnParams { /* graphconv.py:75-83, graphconv_reslap.py:30-43 */
  const float* weight;       /* [Fin*K, Fout], row index f*K + k */
  const float* bias;         /* [Fout] */
  const float* M_L;          /* [Fin, Fin] */
  const float* alpha;        /* [1] */
  const float* beta;         /* [1] (Reslap; may be NULL when L_prev is NULL) */
} AgcnParams;

typedef struct AgcnOutputs { /* graphconv.py:125,252; graphconv_reslap.py:89,230 */
  float* Y;                  /* [rows, ldy] activated nodes; pad rows are written as 0 */
  float* res_L;              /* tiles: residual Laplacian after clip + leaky-ReLU */
  float* res_W;              /* tiles: similarity matrix exp(-||(x_i-x_j) M_L||) */
  float* L_sum;              /* tiles: Reslap's 4th output; for SGC_LL optional (may be NULL) */
} AgcnOutputs;

typedef struct AgcnGrads {
  float* dX;                 /* [rows, ldx] */
  float* dweight;            /* [Fin*K, Fout] */
  float* dbias;              /* [Fout] */
  float* dM_L;               /* [Fin, Fin]; written only in AGCN_MODE_PAPER (else untouched; may be NULL) */
  float* dalpha;             /* [1] */
  float* dbeta;              /* [1]; written only for Reslap with L_prev (may be NULL otherwise) */
  float* dL_prev;            /* tiles; written only for Reslap with L_prev (may be NULL otherwise) */
} AgcnGrads;

/* Reserve space: intermediates the forward keeps for the backward (opaque
 * layout).  Workspace: scratch for one call.
 * The first AGCN_RESERVE_HEADER_BYTES of a reserve buffer must be ZERO when the buffer is first handed to the
 * library (hipMemset once after allocating it); the library keeps that header consistent from then on, also when
 * the buffer is reused for other batches.  (It holds the words through which the workgroups that share one large
 * graph exchange a vector inside a launch; csrc/spec_backward_coop.h.) */
#define AGCN_RESERVE_HEADER_BYTES 65536
AGCN_API size_t agcn_sgc_ll_reserve_bytes(const AgcnLayerDesc* desc, const void* plan_host);
/* Device-side status of the calls that used `reserve`.  The workgroups that share one large graph wait for each other's
 * words inside a launch; such a wait is bounded (10 ms of wall time) and can only run out when the members are not
 * co-resident -- more concurrent launches with members than the chip holds (see agcn_set_coop_members), or other work
 * keeping the CUs.  A wait that ran out poisons its step with NaN AND counts itself in a sticky word of the header.
 * This call copies that word to the host on `hip_stream` and WAITS for it (the one entry point that synchronises: call it
 * where the host synchronises anyway -- reading a loss, the end of an epoch); *timeouts = the count.  Returns AGCN_OK
 * when it is 0, AGCN_EDEVICE otherwise (every result computed with this reserve since the header was last zeroed is
 * suspect; zero the header to re-arm). */
AGCN_API int agcn_sgc_ll_reserve_status(const void* reserve, size_t reserve_bytes, void* hip_stream, int32_t* timeouts);
/* The same count, kept where it outlives the reserve.  A caller whose reserves are short-lived (an autograd graph frees
 * them as soon as backward has run) registers ONE 4-byte device word per device -- caller-owned, zeroed by the caller, alive
 * until it is unregistered with NULL --; every wait that runs out in a later call ON THAT DEVICE counts itself there too.
 * agcn_device_status copies the word to the host on `hip_stream` and waits for it, like agcn_sgc_ll_reserve_status:
 * AGCN_OK when 0 (or when nothing is registered), AGCN_EDEVICE otherwise.  The library never clears the word. */
AGCN_API int agcn_set_device_status_word(int32_t device, void* word);
AGCN_API int agcn_device_status(int32_t device, void* hip_stream, int32_t* timeouts);
AGCN_API size_t agcn_sgc_ll_forward_workspace_bytes(const AgcnLayerDesc* desc, const void* plan_host);
AGCN_API size_t agcn_sgc_ll_backward_workspace_bytes(const AgcnLayerDesc* desc, const void* plan_host);

/* Which kernels a forward / backward call on this batch would launch: host logic only (nothing is launched, device
 * pointers of `batch` are only tested for NULL).  out[0] = path: 4 fused one-launch-per-direction kernels (| 1 / | 2:
 * the specialised forward / backward; | 128: a whole-backward call computes the parameter gradients in trailing workgroups
 * of the backward launch instead of a kernel of their own), 8 mixed, 16 slab kernels, 32 per-graph kernels, | 64 point-cloud
 * kernels; out[1] = schedule position of
 * the first graph of the 8-wave launch (== B: none); out[2..4] = forward: graphs spread over member workgroups,
 * members, graphs packed two to a workgroup; out[5..7] = the same for the backward. */
#define AGCN_DISPATCH_INFO_INTS 8
AGCN_API int agcn_sgc_ll_dispatch_info(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch, int32_t* out, int32_t n_out);

AGCN_API int agcn_sgc_ll_forward(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch, const AgcnParams* params,
                        const AgcnOutputs* out, void* reserve, size_t reserve_bytes,
                        void* workspace, size_t workspace_bytes, void* hip_stream);

/* `fwd` are the tensors the forward produced (read only here).  dY [rows, ldy]
 * is the gradient w.r.t. Y; dL_sum (tiles, nullable) is the gradient arriving
 * on the L_sum output from a later Reslap layer's dL_prev. */
AGCN_API int agcn_sgc_ll_backward(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch, const AgcnParams* params,
                         const AgcnOutputs* fwd, const float* dY, const float* dL_sum,
                         const AgcnGrads* grads, const void* reserve, size_t reserve_bytes,
                         void* workspace, size_t workspace_bytes, void* hip_stream);

/* The two stages of agcn_sgc_ll_backward as separate calls (same arguments, same workspace):
 * _nodes  = everything local to a graph: dX, dL_prev, and the per-graph partials in workspace;
 * _params = the reductions over the batch: dweight, dbias, dM_L, dalpha, dbeta.
 * _params must follow _nodes on the same stream with the same workspace. */
/* The whole backward with its PARAMETER GRADIENTS ON A SECOND STREAM (ABI 106).  dX (and dL_prev) are complete on `hip_stream` when the
 * call's launches there are; dweight / dbias / dM_L / dalpha / dbeta are computed on `param_stream`, behind the per-graph kernels of
 * this call: the library records `fork_event` (a hipEvent_t the caller owns) on hip_stream and makes param_stream wait for it -- graph
 * edges under stream capture, no host synchronisation.  In a multi-layer step the caller's stream goes straight on to the previous
 * layer's backward while this layer's parameter gradients run beside it; the caller JOINS the streams once, at the end of the step
 * (record on param_stream, wait on hip_stream), and keeps reserve and workspace untouched until then.  Same kernels, same bits as
 * agcn_sgc_ll_backward. */
AGCN_API int agcn_sgc_ll_backward_forked(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch, const AgcnParams* params,
                                const AgcnOutputs* fwd, const float* dY, const float* dL_sum, const AgcnGrads* grads,
                                const void* reserve, size_t reserve_bytes, void* workspace, size_t workspace_bytes,
                                void* hip_stream, void* param_stream, void* fork_event);
AGCN_API int agcn_sgc_ll_backward_nodes(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch, const AgcnParams* params,
                               const AgcnOutputs* fwd, const float* dY, const float* dL_sum,
                               const AgcnGrads* grads, const void* reserve, size_t reserve_bytes,
                               void* workspace, size_t workspace_bytes, void* hip_stream);
AGCN_API int agcn_sgc_ll_backward_params(const AgcnLayerDesc* desc, const AgcnGraphBatch* batch, const AgcnParams* params,
                                const AgcnOutputs* fwd, const float* dY, const float* dL_sum,
                                const AgcnGrads* grads, const void* reserve, size_t reserve_bytes,
                                void* workspace, size_t workspace_bytes, void* hip_stream);

/* GraphGatherMol (graphgather.py:40-77): out[b, :] = act(sum of graph b's real rows). */
AGCN_API int agcn_graph_gather_forward(const void* plan_host, const void* plan_dev, const float* X, int32_t F,
                              int32_t ldx, int32_t activation, float* out /* [B, F] */, void* hip_stream);
AGCN_API int agcn_graph_gather_backward(const void* plan_host, const void* plan_dev, const float* out,
                               const float* dOut, int32_t F, int32_t ldx, int32_t activation,
                               float* dX /* [rows, ldx], pad rows zeroed */, void* hip_stream);

/* Check of a symmetry vouch (AgcnGraphBatch.L0_symmetric for a dense L_orig, L_prev_symmetric): ADDS to *count_dev (a
 * caller-owned, caller-zeroed device int32) the number of element pairs i < j of the per-graph tiles whose two sides differ
 * bit for bit.  `tiles`: packed tiles (ld = 0: graph b at tile_off[b], leading dimension n_b) or the reference's padded feed
 * (ld = max_atom: graph b at b * ld * ld).  Asynchronous on `hip_stream`; reading the count is the caller's synchronisation.
 * The point-cloud kernels take La^T g_1 as La g_1 on the strength of those flags: a false vouch is a silently wrong dX, so the
 * Python layers run this once per layer object (and on every call under AGCN_DEBUG_CHECKS=1) and raise on a non-zero count. */
AGCN_API int agcn_tiles_asymmetry(const void* plan_host, const void* plan_dev, const float* tiles, int32_t ld,
                         int32_t* count_dev, void* hip_stream);

/* ---- "next" rows: the layers around the SGC stack --------------------------------
 * rows_dense: Y = act(A . W [+ bias] [+ Add]) over the row space of a batch.
 *   DenseMol  (models/layers/dense_layer.py:33-50): activation LINEAR (the layer stores one but never
 *             applies it), bias, keep_pad_rows = 1 (the bias also lands on pad rows), Add = NULL.
 *   BlockEnd  (models/layers/blockend.py:67-86): Y = act(x + x_res . W): A = x_res, Add = x, no bias,
 *             keep_pad_rows = 0 (tf.pad -> pad rows 0).
 * Backward: Gp = dY . act'(Y); dX = Gp . W^T; dW = A^T Gp; dbias = colsum(Gp) (nullable); dAdd = Gp
 * (nullable; contiguous [rows, Fout]). */
typedef struct AgcnDenseDesc {
  int32_t Fin, Fout;
  int32_t activation;    /* AGCN_ACT_LINEAR | AGCN_ACT_RELU */
  int32_t keep_pad_rows; /* 1: pad rows are ordinary rows */
  int32_t ldx, ldy, ldadd;
  uint32_t flags;        /* per-call dispatch overrides (AGCN_FLAG_ROWS_SPLIT, AGCN_FLAG_PG_ONEPASS), 0 = process defaults */
} AgcnDenseDesc;
AGCN_API int agcn_rows_dense_forward(const AgcnDenseDesc* desc, const void* plan_host, const void* plan_dev,
                            const float* X, const float* W, const float* bias, const float* Add, float* Y,
                            void* hip_stream);
AGCN_API size_t agcn_rows_dense_backward_workspace_bytes(const AgcnDenseDesc* desc, const void* plan_host);
AGCN_API int agcn_rows_dense_backward(const AgcnDenseDesc* desc, const void* plan_host, const void* plan_dev,
                             const float* X, const float* W, const float* Y, const float* dY, float* dX,
                             float* dAdd, float* dW, float* dbias, void* workspace, size_t workspace_bytes,
                             void* hip_stream);

/* The same with a DEVICE scalar on the product: Y = act(scale[0] * (A . W) [+ bias] [+ Add]).
 *   DenseBlockEnd (models/layers/densenet_block.py:99-131) is a chain of these, one per in-block activation
 *   (scale = beta_inblock, :113-117) and per preceding block output (scale = beta_outblock, :119-123), each
 *   adding onto the previous sum (Add), the last one carrying the activation (:92) and zeroing pad rows (:126).
 * Backward additionally returns dscale[0] = <W, A^T Gp>; dX and dW carry the factor scale[0].
 * scale == NULL (and dscale == NULL) is exactly agcn_rows_dense_forward / _backward. */
AGCN_API int agcn_rows_dense_scaled_forward(const AgcnDenseDesc* desc, const void* plan_host, const void* plan_dev,
                                   const float* X, const float* W, const float* scale, const float* bias,
                                   const float* Add, float* Y, void* hip_stream);
/* ... and with a WORKSPACE for the forward (ABI 105): at point-cloud row counts (>= 8192 packed rows without pad rows, 64 / 128 / 256
 * features) the product runs on the bf16 matrix pipe at fp32 accuracy (csrc/rows_split.hip), which needs room for the parameter image
 * it builds per call.  agcn_rows_dense_forward_workspace_bytes returns 0 when the batch / layer does not take that path (then
 * workspace may be NULL and the call is exactly agcn_rows_dense_scaled_forward).  The backward's workspace covers its own image. */
AGCN_API size_t agcn_rows_dense_forward_workspace_bytes(const AgcnDenseDesc* desc, const void* plan_host);
AGCN_API int agcn_rows_dense_scaled_forward_ws(const AgcnDenseDesc* desc, const void* plan_host, const void* plan_dev,
                                      const float* X, const float* W, const float* scale, const float* bias,
                                      const float* Add, float* Y, void* workspace, size_t workspace_bytes, void* hip_stream);
AGCN_API int agcn_rows_dense_scaled_backward(const AgcnDenseDesc* desc, const void* plan_host, const void* plan_dev,
                                    const float* X, const float* W, const float* scale, const float* Y,
                                    const float* dY, float* dX, float* dAdd, float* dW, float* dbias,
                                    float* dscale, void* workspace, size_t workspace_bytes, void* hip_stream);

/* Classifier tail: T task heads + the weighted sigmoid cross-entropy training loss.
 *   logits[t][b][c] = feat[b, :] . W[t][:, c] + bias[t][c], c in {0, 1}
 *       (models/operators/model_operatos.py:710-759 fully_connected_layer, :792-864 multitask_logits / logits)
 *   loss = inv_batch * sum_t sum_b sum_c w[b][t] * sce(logits[t][b][c], one_hot(int(y[b][t]), 2)[c])
 *       sce(x, z) = max(x, 0) - x z + log(1 + exp(-|x|))   (tf.nn.sigmoid_cross_entropy_with_logits)
 *       (models/tf_modules/multitask_classifier.py:41-44 'cross_entropy' loss_fn, :187-209 add_training_loss;
 *        inv_batch = 1 / batch_size, :207-208)
 * feat [B, ldf] (F used), W [T][F][2], bias [T][2], labels / weights [B][T] fp32 (the reference's placeholders),
 * logits [T][B][2], loss [1].  workspace: agcn_multitask_ce_workspace_bytes(B, F, T).
 * Backward: dloss is a DEVICE scalar (the upstream gradient of the loss); writes dfeat [B, ldf] (columns < F),
 * dW [T][F][2], dbias [T][2].  Reductions run in a fixed order (no atomics). */
AGCN_API size_t agcn_multitask_ce_workspace_bytes(int32_t B, int32_t F, int32_t T);
AGCN_API int agcn_multitask_ce_forward(int32_t B, int32_t F, int32_t T, const float* feat, int32_t ldf, const float* W,
                              const float* bias, const float* labels, const float* weights, float inv_batch,
                              float* logits, float* loss, void* workspace, size_t workspace_bytes,
                              void* hip_stream);
AGCN_API int agcn_multitask_ce_backward(int32_t B, int32_t F, int32_t T, const float* feat, int32_t ldf, const float* W,
                               const float* logits, const float* labels, const float* weights, float inv_batch,
                               const float* dloss, float* dfeat, float* dW, float* dbias, void* workspace,
                               size_t workspace_bytes, void* hip_stream);

/* GraphPoolMol (models/layers/graphpool.py:55-110): out[i] = max over {x_j : L_ij != 0} (x_i if the row
 * is empty), pad rows 0.  `lap`: only the L0_* members are read.  argmax [rows, F] int32 is optional.
 * The reference passes NO gradient through this layer (py_func); agcn_graph_pool_backward is the
 * arg-max-routing extension and needs a symmetric pattern. */
AGCN_API int agcn_graph_pool_forward(const void* plan_host, const void* plan_dev, const AgcnGraphBatch* lap,
                            const float* X, int32_t F, int32_t ldx, float* out, int32_t* argmax,
                            void* hip_stream);
AGCN_API int agcn_graph_pool_backward(const void* plan_host, const void* plan_dev, const AgcnGraphBatch* lap,
                             const int32_t* argmax, const float* dOut, int32_t F, int32_t ldx, float* dX,
                             void* hip_stream);

/* Storage-dtype conversion of a [rows, cols] feature matrix (bf16 <-> f32) for the "bf16 features + fp32
 * Laplacian" configuration: features / activations / their gradients may be stored in bf16 between layers,
 * every kernel of the path computes in fp32.  f32 -> bf16 rounds to nearest even. */
#define AGCN_DTYPE_F32 0
#define AGCN_DTYPE_BF16 1
AGCN_API int agcn_cast_rows(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t rows, int32_t cols,
                   int32_t ld_src, int32_t ld_dst, void* hip_stream);

/* Row layout conversion between the reference's padded feed and packed rows.  The reference hands every layer
 * lists of B [max_atom, F] tensors (graph_topology.py:84-91) and slices the real rows off again inside the layer
 * (graphconv.py:145-154); here the first layer of a network packs them once (agcn_rows_pack), the whole network
 * runs on packed rows, and a padded copy is produced only if the caller looks at per-graph outputs
 * (agcn_rows_unpack; pad rows = 0, or pad_fill[F] for DenseMol whose bias also lands on pad rows,
 * dense_layer.py:42-50).  `plan` must be a packed plan (row_stride 0) of the same node counts. */
AGCN_API int agcn_rows_pack(const void* plan_host, const void* plan_dev, const float* padded, int32_t stride_rows, int32_t F,
                   int32_t ld_padded, float* packed, int32_t ld_packed, void* hip_stream);
AGCN_API int agcn_rows_unpack(const void* plan_host, const void* plan_dev, const float* packed, int32_t F, int32_t ld_packed,
                     float* padded, int32_t stride_rows, int32_t ld_padded, const float* pad_fill, void* hip_stream);

/* ---- packed-batch ingest: the step before the path, host only -------------------
 * Replaces the per-step feed construction of models/tf_modules/graph_topology.py:84-135 (pad every graph
 * to max_atom, densify every Laplacian) fed by DiskDataset.iterbatches (utils/dataset/diskdataset.py:239-295).
 * The dataset is laid out once as a graph store; a batch is a gather of contiguous segments into
 * caller-owned (pinned) host buffers, which the caller uploads and hands to agcn_plan_build + AgcnGraphBatch. */
typedef struct AgcnGraphStore {   /* all HOST pointers, owned by the caller */
  int64_t n_graphs;
  int32_t F;                 /* node-feature width */
  int32_t reserved;
  const int32_t* n;          /* [G]     nodes per graph (degree_list length, graph_structure.py:43) */
  const int64_t* row_off;    /* [G+1]   first row of graph g in X / rowptr */
  const float* X;            /* [R][F]  node features of all graphs, R = row_off[G] */
  const int64_t* rowptr;     /* [R+1]   CSR over ALL rows: entries of row r are rowptr[r] .. rowptr[r+1] */
  const int32_t* col;        /* column index inside the graph */
  const float* val;          /* float32(float64 Laplacian value), graph_structure.py:124-127 */
  const uint8_t* symmetric;  /* [G] 1 = L_orig == L_orig^T bit for bit (NULL = unknown: batches report 0) */
} AgcnGraphStore;
/* rows / nnz (and the largest graph) of the batch `ids` [B]: sizes of the staging buffers */
AGCN_API int agcn_store_batch_sizes(const AgcnGraphStore* store, const int64_t* ids, int32_t B, int64_t* rows,
                           int64_t* nnz, int32_t* max_n);
/* n_out [B]; X_out [rows][ldx_out]; rowptr_out [rows+1] (batch-global offsets, as AgcnGraphBatch wants);
 * col_out / val_out [nnz]; symmetric_out = AND of the graphs' flags.  n_threads <= 1: single thread. */
AGCN_API int agcn_store_pack_batch(const AgcnGraphStore* store, const int64_t* ids, int32_t B, int32_t* n_out,
                          float* X_out, int32_t ldx_out, int32_t* rowptr_out, int32_t* col_out, float* val_out,
                          int32_t* symmetric_out, int32_t n_threads);

/* ---- graph store resident in HBM ---------------------------------------------------------------------
 * 288 GB of HBM hold whole datasets (HIV: 41 127 graphs = 0.3 GB), so the store's X / CSR arrays are uploaded ONCE
 * and a batch never crosses PCIe again: agcn_store_stage_batch builds the plan of the batch `ids` plus a per-graph
 * gather table into caller-owned PINNED host memory (`stage_host`), enqueues ONE host-to-device copy of that blob
 * (tens of KB) and ONE gather launch that writes the packed batch (X rows, batch-global rowptr, col, val) into
 * caller-owned device buffers -- what agcn_store_pack_batch + five uploads do through the host.  The plan sits at
 * offset 0 of the blob: `stage_host` / `stage_dev` are the AgcnGraphBatch's plan_host / plan_dev.  The caller keeps
 * `stage_host` untouched until the copy has executed (an event on `hip_stream`); everything else is stream-ordered. */
typedef struct AgcnResidentStore {
  AgcnGraphStore host;       /* host index of the store: n, row_off, rowptr, symmetric are read (X / col / val may be NULL) */
  const void* X_dev;         /* [R][F]  device copy of the node features: float32, or bfloat16 if x_bf16 */
  const int64_t* rowptr_dev; /* [R+1]   device copies of the CSR arrays */
  const int32_t* col_dev;
  const float* val_dev;
  int32_t x_bf16;            /* 1: X_dev holds bfloat16 (features at rest in bf16, "bf16 features + fp32 Laplacian") */
  int32_t stage_bf16;        /* 0: the gather writes float32 rows (bf16 at rest is widened: the layer then runs with fp32 features);
                              * 1 (ABI 106): X_out is a BFLOAT16 buffer ([rows][ldx_out] elements) -- bf16 rows are copied as they
                              * are (half the bytes in AND out), float32 rows are rounded to nearest even -- for a layer call with
                              * AgcnLayerDesc.feat_dtype = AGCN_DTYPE_BF16 */
} AgcnResidentStore;
typedef struct AgcnStagedBatch {
  int64_t rows, nnz;
  int32_t max_n, symmetric;
  int64_t stage_bytes;       /* bytes of the blob that were written / copied */
  AgcnPlanInfo plan;         /* what agcn_plan_info would return for the staged plan */
} AgcnStagedBatch;
/* upper bound of the blob size for a batch of B graphs with `rows` rows in total */
AGCN_API size_t agcn_store_stage_bytes(int32_t B, int64_t rows);
AGCN_API int agcn_store_stage_batch(const AgcnResidentStore* store, const int64_t* ids, int32_t B, void* stage_host,
                           void* stage_dev, size_t stage_cap, void* X_out, int32_t ldx_out, int64_t rows_cap,
                           int32_t* rowptr_out, int32_t* col_out, float* val_out, int64_t nnz_cap,
                           AgcnStagedBatch* info, void* hip_stream);

/* ---- one batch of an epoch in ONE call ------------------------------------------------------------------------------
 * agcn_store_stage_batch + agcn_sgc_ll_forward + agcn_sgc_ll_backward (SGC_LL; rows packed: ldx = Fin, ldy = Fout) on capacity
 * buffers the caller allocated once: what a host loop does per batch around the three entry points -- plan objects, size
 * queries, argument structs -- happens here, in C (the Python loop spent 44 us of host time per batch there; with two batches
 * in flight the epoch was host-bound).  All pointers caller-owned; device buffers unless noted.
 *   restage = 1: stage the batch `ids` first (slot_event: a hipEvent_t, or NULL -- it is WAITED for before the pinned blob
 *                `stage_host` is rewritten and RECORDED behind the copy that reads it, so that a blob is reused safely);
 *   restage = 0: the batch staged by the previous call with the same buffers (`info` as that call left it) -- the retry
 *                after AGCN_ENOSPACE.
 * Returns AGCN_ENOSPACE with `needs` filled in when `reserve` / `workspace` are too small for this batch (nothing of the
 * layer has been launched; a new reserve must be zero-filled in its first AGCN_RESERVE_HEADER_BYTES). */
typedef struct AgcnEpochBuffers {
  void* stage_host;          /* PINNED host blob (agcn_store_stage_bytes) */
  void* stage_dev;           /* its device twin */
  size_t stage_cap;
  /* X, Y, dY, dX: float32, or BFLOAT16 buffers of the same shapes when desc->feat_dtype == AGCN_DTYPE_BF16 (the store must then
   * stage bf16: AgcnResidentStore.stage_bf16 = 1; ABI 106).  A bf16 batch the specialised kernels do not serve (dispatch bit 256
   * unset: a molecule beyond the members' reach) is converted inside the call -- four agcn_cast_rows launches around the fp32 layer
   * kernels, scratch from `workspace` (counted in needs->workspace_bytes) --, the results are the same bf16 tensors */
  float* X; int32_t* rowptr; int32_t* col; float* val;   /* the staged batch: [rows_cap][Fin], [rows_cap + 1], [nnz_cap] x 2 */
  int64_t rows_cap, nnz_cap, tiles_cap;
  float *Y, *res_L, *res_W;  /* [rows_cap][Fout], [tiles_cap] x 2 */
  const float* dY;           /* [rows_cap][Fout] upstream gradient */
  float *dX, *dweight, *dbias, *dM_L, *dalpha;
  void* reserve; size_t reserve_bytes;
  void* workspace; size_t workspace_bytes;
} AgcnEpochBuffers;
typedef struct AgcnEpochNeeds { size_t reserve_bytes, workspace_bytes; } AgcnEpochNeeds;
AGCN_API int agcn_epoch_step(const AgcnResidentStore* store, const int64_t* ids, int32_t B, const AgcnLayerDesc* desc,
                    const AgcnParams* params, const AgcnEpochBuffers* buf, int32_t restage, AgcnStagedBatch* info,
                    AgcnEpochNeeds* needs, void* slot_event, void* hip_stream);

/* Optional per-kernel timing (HIP events recorded on the launch stream around every kernel of
 * the path).  Off by default.  agcn_profile_collect waits for the events recorded since the last
 * collect and returns, per kernel id < agcn_profile_kernels(), the total milliseconds and the
 * number of launches.  Debug facility: not thread safe, not capturable into a HIP graph. */
AGCN_API void agcn_profile_enable(int on);
AGCN_API int32_t agcn_profile_kernels(void);
AGCN_API const char* agcn_profile_kernel_name(int32_t id);
AGCN_API int agcn_profile_collect(double* total_ms, int64_t* calls);

AGCN_API int32_t agcn_version(void);
AGCN_API const char* agcn_last_error(void);
/* LDS bytes per workgroup above which a graph is processed with its tile in
 * global memory instead of LDS (default 160 KiB; settable for tests). */
AGCN_API int32_t agcn_get_lds_limit(void);
AGCN_API void agcn_set_lds_limit(int32_t bytes);
/* 1 (default): SGC_LL with K = 2 on packed rows with a symmetric CSR L_orig runs on the
 * slab-parallel kernels (16-row strips of every graph's tile, balanced over the CUs);
 * 0: always one workgroup per graph.  Same results either way (tested). */
AGCN_API int32_t agcn_get_slab_mode(void);
AGCN_API void agcn_set_slab_mode(int32_t mode);
/* 1 (default): batches of small graphs (the largest graph's X, M_L, W, n x n tile and Z^T fit
 * LDS) run SGC_LL K = 2 on the fused one-launch-per-direction kernels; 0: never; 2 / 3: additionally,
 * when only a few graphs are too large, run those on a side stream next to the fused kernels ("mixed"
 * dispatch): 2 = on the per-graph LDS-tile kernels when their tile fits LDS (else as 3), 3 = on the slab
 * kernels. */
AGCN_API int32_t agcn_get_fused_mode(void);
AGCN_API void agcn_set_fused_mode(int32_t mode);
/* Flavour of the fused kernels: -1 (default) the library picks per direction from measurements;
 * 0: always stage M_L / W as LDS images; 1: always read them from global memory inside the k-loops (whenever the
 * graph's own working set fits LDS).  Same results to the last few ulps (different summation grouping). */
AGCN_API int32_t agcn_get_fused_global_b(void);
AGCN_API void agcn_set_fused_global_b(int32_t mode);
/* 2 (default): parameter gradients in one launch, the row splits' partial tiles handed to the workgroup of the tile's last
 * split as tagged 8-byte granules (where the per-graph kernel of the same call can clear them: whole backward calls on the
 * shape-specialised kernels, batches below 8192 rows; elsewhere as 1); 1: one launch, arrival counters (the last workgroup
 * to arrive adds the splits); 0: partial + reduce launches.  All three add in fixed order: bit-reproducible, and 1 and 2
 * bit-identical to each other. */
AGCN_API int32_t agcn_get_param_grad_onepass(void);
AGCN_API void agcn_set_param_grad_onepass(int32_t on);
/* Shape-specialised fused kernels (csrc/spec_*.hip: SGC_LL and SGC_LL_Reslap, K = 2, layer widths 75->64, 64->128, 128->128, 128->64,
 * 64->64; 1024-thread workgroups, feature widths compiled in).  A bit mask, default 7: bit 0 = forward, bit 1 =
 * backward (used wherever the fused path would be taken and the shape is covered; 0: always the general fused
 * kernels), bit 2 = the cooperative launch layout (csrc/spec_backward_coop.h): graphs of three 16-row tiles and up
 * (five, when the launch does not fit the chip any more; up to 128 nodes, up to 192 in a kernel variant of its own as far
 * as the members' LDS plan fits: 64->64 192, 75->64 176, 64->128 160, 128 input features 128) spread over one workgroup
 * per tile of the same launch, the smallest graphs packed two to a workgroup to make room.  Graphs beyond the one-workgroup
 * kernels' 96 nodes are spread whatever the batch size or Laplacian mode (only members can take them).  Same results within the last few ulps
 * (hardware sqrt / exp2 in the similarity phase; summation orders). */
AGCN_API int32_t agcn_get_spec_mode(void);
AGCN_API void agcn_set_spec_mode(int32_t mode);
/* In batches of >= 1024 graphs the graphs of at most 24 nodes (32 from 3072 graphs on) take an 8-wave flavour of the
 * specialised kernels in a launch of their own (three such workgroups share a CU).  2 (default): the launch of the few larger
 * graphs runs beside it on the library's per-device side stream (fork / join with events, capturable into a HIP
 * graph); 1: before it on the caller's stream; 0: one launch for all graphs. */
AGCN_API int32_t agcn_get_spec_small(void);
AGCN_API void agcn_set_spec_small(int32_t on);
/* Member workgroups per launch of the cooperative layout (spec_mode bit 2): all members of a graph must be RESIDENT
 * together, so concurrent member launches on one device have to fit the chip.  Default 120 (two concurrent launches on
 * 256 CUs: two epoch lanes, or two ranks sharing a GPU); with L launches in flight set 240 / L (0: no members).  Clamped
 * to [0, 120].  Process-wide, like the other knobs. */
AGCN_API int32_t agcn_get_coop_members(void);
AGCN_API void agcn_set_coop_members(int32_t members);
/* 0 (default; env AGCN_SPEC_TAIL=1 opts in): 1 makes a whole-backward call on a batch the specialised kernels take in ONE
 * 16-wave launch (SGC_LL, paper mode) compute the parameter gradients in trailing workgroups of that launch -- dispatch bit
 * 128 --; 0: always the separate parameter-gradient kernel (measured faster: profiles/EXPERIMENTS.md).  Both are
 * bit-reproducible; they differ in summation order. */
AGCN_API int32_t agcn_get_spec_tail(void);
AGCN_API void agcn_set_spec_tail(int32_t on);
/* Bound of a member's wait for a sibling's word, in ticks of the device's 100 MHz constant clock (default 1 000 000 = 10 ms:
 * three orders of magnitude beyond a step).  0 makes every wait that is not satisfied by its first poll run out -- the
 * deterministic way to exercise the reporting path (status words, NaN poisoning) in a test.  Process-wide. */
AGCN_API uint32_t agcn_get_coop_timeout_ticks(void);
AGCN_API void agcn_set_coop_timeout_ticks(uint32_t ticks);

/* 1 (default; env AGCN_ROWS_SPLIT=0; per call: AGCN_FLAG_ROWS_SPLIT): packed-row products of point-cloud sized batches (>= 1024
 * packed rows, K = 2) run on the bf16 matrix pipe at fp32 accuracy -- every fp32 operand split exactly into three bf16 pieces, six
 * v_mfma_f32_16x16x32_bf16 per 32 k (csrc/rows_split.hip).  WHICH products, exactly as the dispatch has it:
 *   Fin = 128 (forward: rows a multiple of 64; backward: rows a multiple of 128): Z = X M_L, the filter, its adjoint,
 *              dX += dZ M_L^T and the parameter-gradient partials;
 *   Fin = 64:  the forward stays on the fused point-cloud kernels (cloud_lap computes the filter itself); the backward takes the
 *              filter adjoint and the parameter-gradient partials;
 *   dense layers (agcn_rows_dense_*): >= AGCN_ROWS_SPLIT_DENSE_MIN packed rows (default 8192), 64 / 128 / 256 features.
 * 0: the fp32-MFMA kernels (csrc/rows_wide.hip, csrc/cloud_kernels.hip, csrc/rows_gemm.hip).  Results differ by fp32 summation order
 * only; both bit-reproducible.  A non-finite or near-FLT_MAX operand (|x| within half a bf16 ulp of the bf16 maximum, 3.39e38)
 * turns its row of the product into NaN on the split kernels (the high piece rounds to infinity, inf - inf), where the fp32-MFMA
 * kernels would propagate the infinity: overflow diagnostics differ between the two settings, finite results do not. */
AGCN_API int32_t agcn_get_rows_split(void);
AGCN_API void agcn_set_rows_split(int32_t on);

/* Environment switches of the point-cloud kernels, read once per process (A/B measurements; results within the same
 * tolerances either way): AGCN_SLAB_SMALL=0 -- slab batches with Fin <= 4 and a dense L_orig take the general slab
 * kernels instead of the streaming small-feature ones (which keep nothing n x n for the backward: r, S, R and dA are
 * rebuilt from the Z / g_1 / X_0 rows); AGCN_ROWS_NARROW=0 -- the projection (Fin <= 4) and the filter adjoint
 * (K * Fin <= 8 columns) of >= 1024 packed rows take the 64-row MFMA workgroups instead of the streaming row kernels;
 * AGCN_CLOUD_RBWD_BOTH=0 -- the backward passes of a 128-feature SGC_LL_Reslap layer run once per 64-column feature half, as before
 * round 5, instead of pass A over both halves / pass C by parts / pass B element-wise from the P = g_1 X_0^T tile;
 * AGCN_CLOUD_BWD2_BF = 0 | 2 -- backward 2's product c Z never | always on the bf16 matrix pipe (default 1: 128-feature layers only);
 * AGCN_CLOUD_BWD1_PARTS=0 -- backward 1 of a 128-feature SGC_LL layer once per feature half instead of by parts (P tile);
 * AGCN_CLOUD_SIM2=0 -- the 128-feature similarity with all 128 features of a lane's column row in registers (two waves per SIMD)
 * instead of two passes of 64 (four);
 * AGCN_BFPIPE=<mask> -- which feature contractions of the per-graph (molecule) kernels run on the bf16 matrix pipe at fp32 accuracy
 * (csrc/spec_bfpipe.h; default 12): 4 the backward's filter adjoint Gp W^T for calls with bf16 features (Gp is one bf16 piece, W^T a
 * three-piece image in MFMA operand order left in the reserve by the forward), 8 the same for fp32 features (Gp split into three piece
 * planes), 1 | 2 the members' forward projection Z = X M_L with bf16 | fp32 features (image built in LDS: measured no gain, off);
 * AGCN_ROWS_SPLIT_DENSE_MIN=<rows> -- fewest packed rows at which the dense layers (agcn_rows_dense_*) take the split-bf16 row
 * kernels (default 8192; values below 128 -- one workgroup's rows -- or unparsable ones are clamped to 128). */

#ifdef __cplusplus
}
#endif
#endif /* AGCN_H_ */
