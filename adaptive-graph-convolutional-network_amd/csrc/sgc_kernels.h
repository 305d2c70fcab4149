// Kernel argument blocks + launcher prototypes (host <-> .hip translation units).
#ifndef AGCN_SGC_KERNELS_H_
#define AGCN_SGC_KERNELS_H_

#include "agcn_internal.h"

namespace agcn {

// Everything is a device pointer unless noted.
struct SgcArgs {
  // plan
  const int32_t* n;
  const int32_t* row_off;
  const int64_t* tile_off;
  const int32_t* order;
  const AgcnGraphDesc* graph_desc;   // [B] in schedule order: one 32-byte load per workgroup (fused kernels)
  const int64_t* wtile_off;
  int32_t B, rows;
  // layer
  int32_t Fin, Fout, K, mode, variant, activation, ldx, ldy;
  // inputs
  int32_t feat_bf16;   // 1: X (forward), Y, dY, dX are bf16 matrices (specialised kernels only); the forward then leaves a
  float* sv_X0;        // widened fp32 copy of X in the reserve ([rows][Fin]) and the backward / parameter gradients read that
  int32_t lddx;        // row stride of dX in elements (the backward's X may be the reserve copy with stride Fin)
  const float* X;
  const float* L0_dense;
  int32_t L0_ld;
  const int32_t* L0_rowptr;
  const int32_t* L0_col;
  const float* L0_val;
  const float* L_prev;
  const float *weight, *bias, *M_L, *alpha, *beta;
  // forward outputs (const in backward)
  float *Y, *res_L, *res_W, *L_sum;
  // reserve (forward writes, backward reads)
  float* sv_Z;     // [rows][Fin]  Z = X . M_L
  float* sv_s;     // [rows]       s_i = d_i^-1/2
  float* sv_scal;  // [B][4]       ssR, scR, ssL, scL
  float* sv_X;     // [K-1][rows][Fin]  Chebyshev terms X_1..X_{K-1}
  float* sv_Lpre;  // [tile_elems] Reslap: L before the second clip
  // transposed parameter copies written by the specialised forward kernel for the specialised backward one (row-major
  // with the MFMA k index as the row: a fragment load then touches 4 cache lines instead of 16)
  float* sv_MT;    // [F4/4][Fin][4]        MT[k/4][f][k%4] = M_L[f][k], zero for k >= Fin (k-blocked: spec_kernels.h kb_frag)
  float* sv_WT;    // [K][Fout/4][Fin][4]   WT[kk][c/4][f][c%4] = W[f*K + kk][c]      (cloud kernels: [K][Fout][Fin])
  int32_t tp_first; // specialised forward: the graphs at schedule positions tp_first .. B-1 write sv_MT / sv_WT
  // exchange words of the cooperative backward (spec_backward_coop.h): [B][97] 64-bit words, zeroed per graph by the
  // specialised forward kernel
  uint64_t* sv_xch;
  uint32_t* dev_status; // the device's persistent status word (agcn_set_device_status_word) or nullptr: a wait that runs out
                        // counts itself there as well (it outlives the reserve)
  uint32_t coop_ticks;  // bound of such a wait in ticks of the 100 MHz constant clock (agcn_set_coop_timeout_ticks)
  // parameter gradients as trailing workgroups of the specialised backward launch (spec_tail.h); tail == nullptr: off
  uint32_t* tail;        // generation word, exit counter and the producer units' flags (reserve header)
  int32_t tail_tiles;    // trailing workgroups: tail_splits x tail_tiles_per_split
  int32_t tail_splits, tail_tiles_per_split;
  float* tail_part;      // [tail_splits - 1][tiles][TAIL_PSTRIDE] partial tiles of the non-reducing splits (workspace)
  int32_t tail_debug;    // AGCN_TAIL_DEBUG (timing experiments only): 1 tail workgroups exit at once, 2 no flag waits, 4 no products
  int32_t tail_first;    // schedule position of the first graph that is a unit of its own (the graphs before it are spread
  int32_t tail_graphs;   // over member slabs), and how many such graphs
  float *g_dweight, *g_dbias, *g_dM, *g_dalpha;
  int32_t coop_wide;    // 1: a cooperative graph has more than COOP_MAX_N nodes (the launch takes the kernel variant with the wide members)
  int32_t coop_slabs;   // 16-wave launch: its first coop_slabs workgroups are members (slabs of the plan's slab table),
  int32_t plain_count;  // the next plain_count one graph each, the rest TWO small graphs each (spec_forward.hip)
  int32_t dual_stride;  // floats of LDS between the two halves of such a workgroup
  int32_t dual_pairs;   // number of such workgroups (set by the caller; plain_count is set by the launcher)
  // feature contractions of the specialised kernels on the bf16 matrix pipe at fp32 accuracy (spec_bfpipe.h): bit 0 = Z = X M_L of
  // the forward; a workgroup takes it where its piece planes + parameter image fit `lds_bytes` (the launch's dynamic LDS)
  int32_t bfp_fwd;
  int32_t lds_bytes;
  // bf16 features: the adjoint of the filter, [g_0 | g_1] = Gp W^T, on the bf16 pipe -- Gp = dY act'(Y) is ONE bf16 piece per
  // element there, W^T comes as a three-piece image in MFMA operand order that the forward's last blocks leave in the reserve
  // beside the fp32 copies: tiles [K][Fout / 32][F4 / 16] x (3 pieces x 64 lanes x 16 bytes); nullptr: fp32 matrix instruction
  void* sv_WTI;
  int32_t g_np;       // backward: 0 = that phase on the fp32 instruction; 1 = bf16 features (Gp one piece); 3 = fp32 features (Gp split
                      // into three piece planes: more LDS, decided per launch by the host)
  // global-tile mode scratch
  float* ws_tile;  // [wtile_off[B]]
  float* ws_buf;   // per-graph image scratch, see graph_scratch_floats()
  // backward only
  const float* dY;
  const float* dL_sum;
  float *dX, *dL_prev;
  float* ws_g;     // [K-1][rows][Fin]  adjoints of X_1..X_{K-1}
  float* ws_dZ;    // [rows][Fin]       (paper mode)
  float* ws_pg;    // [B or n_slabs][2] per-graph / per-slab dalpha, dbeta
  float* ws_Gp;    // [rows][Fout]      masked upstream gradient (fused backward writes it)
  // slab-parallel path (slab_kernels.hip)
  const int32_t* slab_graph;
  const int32_t* slab_row;
  const AgcnSlabDesc* slab_desc;     // [n_slabs]: one 32-byte load per member workgroup (spec_backward_coop.h)
  int32_t n_slabs;
  float* sv_r;     // [tile_elems]  r_ij = ||Z_i - Z_j||  (paper mode, forward -> backward)
  float* ws_dA;    // [tile_elems]  dA = -dR               (paper mode, bwd1 -> bwd2)
  float* ws_dAt;   // [tile_elems]  dAt[i][j] = dA[j][i], written by the strip that owns row i (bwd2 then reads rows only)
  float* ws_dd;    // [rows]        dd_j                   (paper mode, bwd1 -> bwd2)
  int32_t pg_slab_base;  // mixed dispatch: slab s writes its partials at ws_pg[2 * (pg_slab_base + s)]
  int32_t slab_quad;     // 1: every graph has a multiple of 64 rows (four consecutive slabs always share a graph)
  // arrival counters of the one-pass parameter-gradient kernel that follows: zeroed by workgroup 0 of the
  // fused backward kernel (saves a memset node in front of it)
  int32_t* zero_ptr;
  int32_t zero_count;
  // ... or the granules of its tagged hand-off (param_grad_tagged_kernel): `ztag_count16` 16-byte units at ztag_ptr, cleared by
  // ALL workgroups of the specialised backward launch, a slice each (spec::zero_tags)
  float* ztag_ptr;
  int32_t ztag_count16;
  // point-cloud kernels (cloud_kernels.hip) on layers of 128 input features: the strip kernels work on 64 feature columns at a
  // time -- `foff` = first column of the half (0 / 64), `fstride` = row stride of the packed feature arrays (Z, g_1, X_1, dZ: Fin)
  // -- and run once per half.  Everything that couples the halves is linear in the per-half products, so the second run ADDS
  // (`half_acc`: n x n outputs read-modify-write, per-graph sums and ds added to the first run's); `half_last` = this run
  // finishes the quantities that need both halves (dd).  64-feature layers: foff 0, fstride 64, half_acc 0, half_last 1.
  int32_t foff, fstride, half_acc, half_last;
  // backward: workspace for the parameter images of the split-bf16 packed-row kernels (rows_split.hip), or nullptr
  void* rs_img;
  // 128 features, point-cloud kernels, paper mode: P = g_1 X_0^T [tile_elems] over all 128 features -- SGC_LL_Reslap: written by
  // part 2 of pass C (which runs first), read element-wise by pass B; SGC_LL: written by part 1 of backward 1, read by its part 2 --
  // or nullptr (those passes then form the product themselves, once per 64-column half)
  float* ws_P;
};

enum { RG_PROJECT = 0, RG_FILTER = 1, RG_FILTER_BWD = 2, RG_PROJECT_BWD = 3, RG_DENSE = 4 };

struct RowsGemmArgs {
  int32_t rows, Kd, N;
  int32_t Fin, Fout, K, activation;
  int32_t row_stride;       // 0 = packed rows (all valid)
  const int32_t* n;         // [B] (padded rows only)
  const float* A0; int32_t lda0;
  const float* A1; int64_t a1_plane;
  const float* Yact; int32_t ldy;
  const float* Bm;
  const float* bias;
  float* Out0; int32_t ldo0;
  float* Out1; int64_t o1_plane;
  float* Gp;                // RG_FILTER_BWD: [rows][Fout] masked upstream gradient
  const float* Add; int32_t ldadd;   // RG_DENSE: optional addend [rows][ldadd]
  const float* scale;       // RG_DENSE / RG_FILTER_BWD: optional device scalar on the product (DenseBlockEnd's beta)
  int32_t keep_pad;         // RG_DENSE / RG_FILTER_BWD: 1 = pad rows are ordinary rows (DenseMol, dense_layer.py:42-50)
  // mixed dispatch: only rows of graphs with more than big_n nodes are processed (0 = all rows)
  int32_t big_n;
  const int32_t* row_graph; // [rows]
  int32_t* zero_ptr;        // RG_FILTER_BWD: counters to zero for the parameter-gradient kernel that follows
  int32_t zero_count;
  float* ztag_ptr;          // ... or the granules of its tagged hand-off (param_grad_tagged_kernel): ztag_count16 16-byte units,
  int32_t ztag_count16;     // a slice per workgroup, behind everything else it does (big_n == 0 only)
  int32_t kc;               // K chunk staged in LDS (set by the launcher)
  int32_t nt;               // column tiles per workgroup (set by the launcher)
};

struct ParamGradArgs {
  int32_t rows, Fin, Fout, K, ldx, paper;
  int32_t n_chunks, chunk_rows, B;   // B = number of [dalpha, dbeta] partial pairs in pg
  int32_t wgs_per_chunk; // set by the launcher
  int64_t part_stride;  // floats per chunk in `part`
  const float* X;
  const float* sv_X;
  const float* Gp;      // [rows][Fout]
  const float* dZ;
  float* part;          // [n_chunks][part_stride]
  size_t part_floats;   // capacity of `part` (set by the caller; 0 = unknown: two-launch path)
  int32_t* counters;    // one-pass kernel: per-tile arrival counters (set by the launcher)
  int32_t counters_zeroed;  // 1 = a preceding kernel of the same call already zeroed them
  int32_t xcd;              // one-pass kernel: XCD-aware workgroup order (set by the launcher)
  // tagged hand-off (param_grad_tagged_kernel): the granules [n_tiles][S - 1][512] of (value, tag) words inside `part`, cleared
  // by the preceding kernel of the same call (tags_zeroed = 1: set by the caller, who also passed SgcArgs::ztag_ptr to it)
  int32_t tags_zeroed;
  uint64_t* gran;           // (set by the launcher)
  int32_t n_tiles;          // (set by the launcher)
  uint32_t* dev_status;     // the device's persistent status word or nullptr: a wait that runs out counts itself there
  uint32_t ticks;           // bound of such a wait (100 MHz constant clock)
  const float* pg;      // [B][2]
  float *dweight, *dbias, *dM_L, *dalpha, *dbeta;
};

// profiling ids (profile.hip)
enum { PK_PROJECT = 0, PK_GRAPH_FWD, PK_FILTER, PK_FILTER_BWD, PK_GRAPH_BWD, PK_PROJECT_BWD, PK_PARAM_PARTIAL,
       PK_PARAM_REDUCE, PK_SLAB_SIM, PK_SLAB_LAP, PK_SLAB_BWD1, PK_SLAB_BWD2, PK_FUSED_FWD, PK_FUSED_BWD,
       PK_PARAM_ONEPASS, PK_SPEC_FWD, PK_SPEC_BWD, PK_CLOUD_SIM, PK_CLOUD_DEG, PK_CLOUD_LAP, PK_CLOUD_BWD1,
       PK_CLOUD_BWD2, PK_CLOUD_RNORM, PK_CLOUD_RLAP1, PK_CLOUD_RBWD_A, PK_CLOUD_RBWD_B, PK_CLOUD_RBWD_C, PK_PARAM_TAGGED, PK_ROWS_PRESPLIT, PK_COUNT };
// Raises a kernel's dynamic-LDS cap (hipFuncAttributeMaxDynamicSharedMemorySize) once per kernel and size: the
// driver call is skipped when the cap already covers `bytes` (it costs several microseconds per launch otherwise).
// Tracked per (device, kernel).  Returns hipSuccess (0) or the HIP error code.
int ensure_dynamic_lds(const void* kernel, size_t bytes);
bool prof_on();
void prof_begin(int id, void* stream);
void prof_end(void* stream);

// threads per workgroup of the per-graph kernels in LDS-tile mode (256..1024)
int graph_threads(int32_t n_graphs);

// LDS bytes one graph of n nodes needs in LDS-tile mode.
size_t sgc_lds_bytes(int32_t n, int32_t Fin);
size_t sgc_lds_bytes_bwd(int32_t n, int32_t Fin);
// floats of per-graph image scratch (global-tile mode) for the whole batch
size_t graph_scratch_floats(int32_t rows, int32_t B, int32_t Fin);

// `first`/`count` select a slice of plan.order; lds_tile picks the kernel flavour.
int launch_sgc_forward(const SgcArgs& a, int32_t first, int32_t count, bool lds_tile, size_t lds_bytes,
                       void* stream);
int launch_sgc_backward(const SgcArgs& a, int32_t first, int32_t count, bool lds_tile, size_t lds_bytes,
                        void* stream);
int launch_rows_gemm(const RowsGemmArgs& a, int mode, void* stream);
// fused per-graph SGC_LL (K = 2) for graphs whose whole working set fits LDS
size_t fused_lds_bytes(int32_t n, int32_t Fin, int32_t Fout, int32_t K);
size_t fused_lds_bytes_gb(int32_t n, int32_t Fin, int32_t Fout, int32_t K);   // parameters read from global memory
int launch_fused_forward(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, bool global_b,
                         void* stream);
size_t fused_bwd_lds_bytes(int32_t n, int32_t Fin, int32_t Fout, int32_t K);
// global_w / global_m: W / M_L read from global memory instead of an LDS image
size_t fused_bwd_lds_bytes_gb(int32_t n, int32_t Fin, int32_t Fout, int32_t K, bool global_w, bool global_m);
int launch_fused_backward(const SgcArgs& a, int32_t first, int32_t count, size_t lds_bytes, bool global_w,
                          bool global_m, void* stream);
// slab-parallel SGC_LL (K = 2, packed rows, symmetric CSR L_orig)
bool slab_path_fits(int32_t Fin);
int launch_slab_forward(const SgcArgs& a, void* stream);
int launch_slab_backward(const SgcArgs& a, void* stream, int32_t max_n = 0);   // max_n: largest graph (0 = unknown)
// point-cloud kernels (cloud_kernels.hip): slab batches with Fin = 64, dense L_orig, every graph a multiple of 64 nodes
bool cloud_path(const SgcArgs& a, bool quad);
size_t cloud_dpart_floats(int64_t tile_elems);   // forward workspace: partial degrees
int launch_cloud_forward(const SgcArgs& a, float* dpart, int32_t max_n, int64_t tile_elems, void* stream);
int launch_cloud_backward_reslap(const SgcArgs& a, float* WT, float* spart, void* stream);   // spart: 8 floats per 64-row strip
int launch_cloud_backward(const SgcArgs& a, float* WT, void* stream);   // (incl. the packed filter adjoint)
bool cloud_param_grads_ok(const ParamGradArgs& a);
int launch_cloud_param_partials(const ParamGradArgs& a, void* stream);   // row-chunk partials + their fixed-order sum
// packed-row products of the 128-input-feature point-cloud layers (rows_wide.hip)
bool rows_wide_ok(int rows, int Fin, int Fout, int K, int ldx, int ldy);
int launch_rows_wide_project(const float* X, int ldx, const float* M, float* Z, int rows, void* stream);
int launch_rows_wide_filter(const float* X0, int ldx, const float* X1, const float* W, const float* bias, int relu, float* Y, int ldy,
                            int Fout, int rows, void* stream);
int launch_rows_wide_filter_bwd(const float* dY, const float* Y, int ldy, int relu, const float* W, const float* M, float* WT, float* MT,
                                float* Gp, float* dX, int ldx, float* g1, int Fout, int rows, void* stream);
int launch_rows_wide_project_bwd(const float* dZ, const float* MT, float* dX, int ldx, int rows, void* stream);
int launch_rows_wide_param_partials(const ParamGradArgs& a, void* stream);
// the same products on the bf16 matrix pipe at fp32 accuracy (rows_split.hip): parameter images built once per call into workspace
bool rows_split_ok(int rows, int Fin, int Fout, int K, int ldx, int ldy);
size_t rows_split_fwd_image_bytes(int Fin, int Fout);
size_t rows_split_bwd_image_bytes(int Fin, int Fout);
int launch_rows_split_fwd_images(const float* M, const float* W, int Fin, int Fout, void* img, void* stream);
int launch_rows_split_bwd_images(const float* M, const float* W, int Fin, int Fout, bool with_m, void* img, void* stream);
int launch_rows_split_project(const float* X, int ldx, float* Z, int Fin, int rows, const void* img, void* stream);
int launch_rows_split_filter(const float* X0, int ldx, const float* X1, int ldx1, const float* bias, int relu, float* Y, int ldy, int Fin,
                             int Fout, int rows, const void* img, void* stream);
int launch_rows_split_filter_bwd(const float* dY, const float* Y, int ldy, int relu, float* Gp, float* dX, int ldx, float* g1, int Fin,
                                 int Fout, int rows, const void* img, void* stream);
int launch_rows_split_project_bwd(const float* dZ, float* dX, int ldx, int Fin, int Fout, int rows, const void* img, void* stream);
int launch_rows_split_param_partials(ParamGradArgs& a, void* stream);   // sets a.n_chunks / a.chunk_rows for the reduce
// ... and of the dense layers (rows_ops.hip) at >= 8192 packed rows, 64 / 128 / 256 features: one parameter image per call
bool rows_split_dense_ok(int rows, int Fin, int Fout, int ldx, int ldy, int ldadd);
size_t rows_split_dense_image_bytes(int Fin, int Fout);
int launch_rows_split_dense_fwd(const float* A, int lda, const float* W, const float* scale, const float* bias, const float* Add, int ldadd,
                                int relu, float* Y, int ldy, int Fin, int Fout, int rows, void* img, void* stream);
int launch_rows_split_dense_bwd(const float* dY, const float* Y, int ldy, int relu, const float* W, const float* scale, float* Gp, float* dX,
                                int ldx, int Fin, int Fout, int rows, void* img, int32_t* zero_ptr, int32_t zero_count, void* stream);
int launch_param_grads(const ParamGradArgs& a, void* stream);
size_t param_grad_onepass_floats(int32_t rows, int32_t Fin, int32_t Fout, int32_t K, int32_t paper);
// the arrival counters inside `part` (nullptr when the one-pass kernel will not be used) and their number
int32_t* param_grad_counters(float* part, size_t part_floats, int32_t rows, int32_t Fin, int32_t Fout, int32_t K,
                             int32_t paper, int32_t* count);
// the granules of the tagged hand-off inside `part` (nullptr when that kernel will not be used) and their size in 16-byte units
float* param_grad_tag_region(float* part, size_t part_floats, int32_t rows, int32_t Fin, int32_t Fout, int32_t K, int32_t paper,
                             int32_t* count16);

}  // namespace agcn
#endif
