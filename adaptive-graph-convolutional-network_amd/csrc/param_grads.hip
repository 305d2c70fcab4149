// Parameter gradients that reduce over every node of the batch:
//   dweight[j, c] = sum_r C[r, j] Gp[r, c],  C[r, f*K + k] = X_k[r, f]   (graphconv.py:243-247 under autodiff)
//   dbias[c]      = sum_r Gp[r, c]
//   dM_L[f, c]    = sum_r X[r, f] dZ[r, c]                                (paper mode only)
//   dalpha, dbeta = sum over graphs of the per-graph partials
// with Gp = dY . act'(Y) (written by the RG_FILTER_BWD launch).  Two launches: per-chunk partial
// products on the matrix cores (A^T B with the row index as the MFMA k dimension), then a
// fixed-order reduction -- no atomics, so the result is reproducible bit for bit.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "agcn_device.h"
#include "sgc_kernels.h"

namespace agcn {

namespace {
constexpr int PG_UNROLL = 4;   // MFMA k-steps (of 4 rows) whose operand loads are issued together
}

// grid: (items / 4, problems).  problem 0: dweight; 1: dbias; 2: dM_L.
// One wave per (row chunk, 16 x 32 output strip).
__global__ __launch_bounds__(256) void param_grad_partial_kernel(ParamGradArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int p = blockIdx.y;
  const int K = a.K, Fin = a.Fin, Fout = a.Fout;
  int Fa, Fb;       // output is [Fa x Fb]
  int64_t out_base; // offset inside one chunk's partial block
  if (p == 0) { Fa = Fin * K; Fb = Fout; out_base = 0; }
  else if (p == 1) { Fa = 1; Fb = Fout; out_base = (int64_t)K * Fin * Fout; }
  else { Fa = Fin; Fb = Fin; out_base = (int64_t)K * Fin * Fout + Fout; }
  const int tf = (Fa + 15) >> 4, tc = (Fb + 31) >> 5;
  // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (blockIdx.x % 8 share one, the
  // grid's x extent is a multiple of 8), and every output tile of one ROW CHUNK is computed on the same
  // XCD, so the chunk's rows of X / Gp / dZ are fetched into ONE L2 instead of up to eight.
  const int W = a.wgs_per_chunk;                       // workgroups (of 4 wave items) per chunk
  const int g = blockIdx.x;
  const int chunk = (g / (8 * W)) * 8 + (g & 7);
  const int rem = ((g >> 3) % W) * 4 + wave;           // tile index inside the chunk
  if (chunk >= a.n_chunks || rem >= tf * tc) return;
  const int itc = rem / tf, itf = rem - itc * tf;
  const int f0 = itf * 16, c0 = itc * 32;
  const int r_begin = chunk * a.chunk_rows;
  const int r_end = min(a.rows, r_begin + a.chunk_rows);
  const int last_row = a.rows - 1;

  // A operand: column j of C / X / ones.  Addresses are clamped so that loads are unconditional
  // (they can all be issued before the first MFMA); out-of-range lanes are zeroed afterwards.
  const int j = min(f0 + r, Fa - 1);
  const bool jv = (f0 + r) < Fa;
  const float* Ap;
  int64_t lda;
  if (p == 0) {
    const int f = j / K, kx = j - f * K;
    if (kx == 0) { Ap = a.X + f; lda = a.ldx; }
    else { Ap = a.sv_X + (int64_t)(kx - 1) * a.rows * Fin + f; lda = Fin; }
  } else if (p == 2) { Ap = a.X + j; lda = a.ldx; }
  else { Ap = a.X; lda = 0; }
  const float* Bp = (p == 2) ? a.dZ : a.Gp;
  const int ldb = (p == 2) ? Fin : Fout;
  const int cA = min(c0 + r, Fb - 1), cB = min(c0 + 16 + r, Fb - 1);
  const bool cAv = (c0 + r) < Fb, cBv = (c0 + 16 + r) < Fb;

  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  for (int rr = r_begin; rr < r_end; rr += 4 * PG_UNROLL) {
    float av[PG_UNROLL], b0[PG_UNROLL], b1[PG_UNROLL];
#pragma unroll
    for (int u = 0; u < PG_UNROLL; ++u) {
      const int row = min(rr + 4 * u + q, last_row);
      av[u] = (p == 1) ? 1.0f : Ap[(int64_t)row * lda];
      b0[u] = Bp[(int64_t)row * ldb + cA];
      b1[u] = Bp[(int64_t)row * ldb + cB];
    }
#pragma unroll
    for (int u = 0; u < PG_UNROLL; ++u) {
      const bool rv = (rr + 4 * u + q) < r_end;
      const float x = (rv && jv) ? av[u] : 0.f;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, (rv && cAv) ? b0[u] : 0.f, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, (rv && cBv) ? b1[u] : 0.f, acc1, 0, 0, 0);
    }
  }
  float* out = a.part + (int64_t)chunk * a.part_stride + out_base;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ff = f0 + q * 4 + i;
    if (ff < Fa) {
      if (c0 + r < Fb) out[(int64_t)ff * Fb + c0 + r] = acc0[i];
      if (c0 + 16 + r < Fb) out[(int64_t)ff * Fb + c0 + 16 + r] = acc1[i];
    }
  }
}

__global__ __launch_bounds__(256) void param_grad_reduce_kernel(ParamGradArgs a) {
  const int64_t nW = (int64_t)a.K * a.Fin * a.Fout, nB = a.Fout, nM = a.paper ? (int64_t)a.Fin * a.Fin : 0;
  const int64_t total = nW + nB + nM;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < total) {
    float s = 0.f;
#pragma unroll 8
    for (int c = 0; c < a.n_chunks; ++c) s += a.part[(int64_t)c * a.part_stride + e];
    if (e < nW) a.dweight[e] = s;
    else if (e < nW + nB) a.dbias[e - nW] = s;
    else a.dM_L[e - nW - nB] = s;
  }
  // scalars: one wave each, fixed order
  if (blockIdx.x == 0) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 2) {
      float* dst = wave == 0 ? a.dalpha : a.dbeta;
      if (dst != nullptr && a.pg != nullptr) {
        const float s = pg_wave_sum(a.pg, a.B, wave, lane);
        if (lane == 0) dst[0] = s;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// One-pass variant: ONE launch instead of partial + reduce.
//   grid.x = n_tiles * S workgroups of 1024 threads.  Workgroup (tile t, split s) owns rows
//   [s * rps, (s + 1) * rps) of the batch; its 16 waves take 32-row slices (all 24 operand loads of a slice are
//   issued before the first MFMA: one memory round trip), the 16 wave accumulators are summed through LDS in
//   wave order, and the workgroup's [16 x 32] partial goes to `part`.  The LAST workgroup of a tile to arrive
//   (one agent-scope atomic counter per tile) adds the S partials in split order -- the sum order is
//   fixed whichever workgroup performs it, so the result is reproducible bit for bit -- and writes the
//   gradient.  Counters are zeroed by the launcher (or by the preceding kernel of the same call).
// Tiles: dweight tf x tc, then dbias tc, then (paper) dM_L tfM x tcM; 16 x 32 outputs each.
// (An XCD-aware workgroup order -- every workgroup of one row split on the same XCD, AGCN_PG_XCD=1 -- cuts this
// kernel's HBM fetch from 22 MB to 6 MB but makes it SLOWER, 9.4 -> 12.0 us eager / -1.7 % graphs/s on the whole
// step: the splits' rows then come out of one L2 instead of eight.  Measured; off by default.)
// TF = 16 | 32 output rows per tile.  32 (two A fragments per wave, four accumulators) halves the number of tiles that
// re-read every row of X / Gp / dZ; it paid only while the large-batch kernel was bound by HBM re-fetches (plain
// workgroup order) and is kept for diagnostics (AGCN_PG_TF=32).
// KS = MFMA k-steps (of 4 rows) whose operand loads a wave issues in one go.  8; 16 (both round trips of a 64-row slice
// in flight together) was measured SLOWER -- B = 2048: 51 vs 38 us -- the registers cost the second resident workgroup.
template <int TF, int KS>
__global__ __launch_bounds__(1024) void param_grad_onepass_kernel(ParamGradArgs a) {
  constexpr int NA = TF / 16, TE = TF * 32;   // A fragments per wave, floats per tile
  __shared__ float wsum[16][512 + 16];
  __shared__ int last_flag;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int K = a.K, Fin = a.Fin, Fout = a.Fout;
  const int S = a.n_chunks;                 // row splits
  // a.xcd != 0: XCD-aware order -- consecutive workgroup ids go round-robin over the 8 XCDs, so with the split
  // index running fastest over a multiple of 8 slots every workgroup of split s lands on XCD s % 8 and a row
  // range of X / Gp / dZ is fetched into ONE L2
  const int S8 = a.xcd ? ((S + 7) & ~7) : S;
  const int tfW = (Fin * K + TF - 1) / TF, tcW = (Fout + 31) >> 5;
  const int tfM = (Fin + TF - 1) / TF;
  int t, s;
  if (a.xcd == 2) {
    // split-major inside an XCD: the workgroups an XCD runs at any one time belong to ONE row split (all its output
    // tiles), so that split's rows of X / Gp / dZ are fetched from HBM once and re-read from that XCD's L2
    const int nt = tfW * tcW + tcW + (a.paper ? tfM * ((Fin + 31) >> 5) : 0);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int sl = j / nt;
    t = j - sl * nt;
    s = sl * 8 + xcd;
  } else {
    t = blockIdx.x / S8;
    s = blockIdx.x - t * S8;
  }
  if (s >= S) return;
  AGCN_STAMP(7, 0);
  int p, rem;                               // problem, tile index inside it
  if (t < tfW * tcW) { p = 0; rem = t; }
  else if (t < tfW * tcW + tcW) { p = 1; rem = t - tfW * tcW; }
  else { p = 2; rem = t - tfW * tcW - tcW; }
  const int Fa = p == 0 ? Fin * K : (p == 1 ? 1 : Fin), Fb = p == 2 ? Fin : Fout;
  const int tf = p == 0 ? tfW : (p == 1 ? 1 : tfM);
  const int itc = rem / tf, itf = rem - itc * tf;
  const int f0 = itf * TF, c0 = itc * 32;
  const int last_row = a.rows - 1;
  const int r_begin = s * a.chunk_rows + wave * a.wgs_per_chunk;   // chunk_rows: rows per split, wgs_per_chunk: rows per wave
  const int r_end = min(min(a.rows, (s + 1) * a.chunk_rows), r_begin + a.wgs_per_chunk);

  const float* Ap[NA];
  int64_t lda = 0;
  bool jv[NA];
#pragma unroll
  for (int x = 0; x < NA; ++x) {
    const int j = min(f0 + 16 * x + r, Fa - 1);
    jv[x] = (f0 + 16 * x + r) < Fa;
    if (p == 0) {
      const int f = j / K, kx = j - f * K;
      if (kx == 0) { Ap[x] = a.X + f; lda = a.ldx; }
      else { Ap[x] = a.sv_X + (int64_t)(kx - 1) * a.rows * Fin + f; lda = Fin; }
    } else if (p == 2) { Ap[x] = a.X + j; lda = a.ldx; }
    else { Ap[x] = a.X; lda = 0; }
  }
  // (K == 2 alternates X / sv_X columns with different strides: keep one stride per fragment)
  int64_t ldax[NA];
#pragma unroll
  for (int x = 0; x < NA; ++x) {
    const int j = min(f0 + 16 * x + r, Fa - 1);
    ldax[x] = (p == 0) ? ((j % K) == 0 ? (int64_t)a.ldx : (int64_t)Fin) : lda;
  }
  const float* Bp = (p == 2) ? a.dZ : a.Gp;
  const int ldb = (p == 2) ? Fin : Fout;
  const int cA = min(c0 + r, Fb - 1), cB = min(c0 + 16 + r, Fb - 1);
  const bool cAv = (c0 + r) < Fb, cBv = (c0 + 16 + r) < Fb;

  f32x4 acc[NA][2];
#pragma unroll
  for (int x = 0; x < NA; ++x) acc[x][0] = acc[x][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int rr = r_begin; rr < r_end; rr += 4 * KS) {
    float av[NA][KS], b0[KS], b1[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int row = min(rr + 4 * u + q, last_row);
#pragma unroll
      for (int x = 0; x < NA; ++x) av[x][u] = (p == 1) ? 1.0f : Ap[x][(int64_t)row * ldax[x]];
      b0[u] = Bp[(int64_t)row * ldb + cA];
      b1[u] = Bp[(int64_t)row * ldb + cB];
    }
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const bool rv = (rr + 4 * u + q) < r_end;
      const float y0 = (rv && cAv) ? b0[u] : 0.f, y1 = (rv && cBv) ? b1[u] : 0.f;
#pragma unroll
      for (int x = 0; x < NA; ++x) {
        const float xv = (rv && jv[x]) ? av[x][u] : 0.f;
        acc[x][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, y0, acc[x][0], 0, 0, 0);
        acc[x][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, y1, acc[x][1], 0, 0, 0);
      }
    }
  }
  AGCN_STAMP(7, 1);
  // The hand-off to the tile's last workgroup uses agent-scope (sc1) stores and loads of exactly the bytes
  // handed off -- no release / acquire fence: an agent-scope release would write back the whole L2, which
  // holds megabytes of this step's outputs (measured: 158 us for this kernel with __threadfence()).
  float* mine = a.part + ((int64_t)t * S + s) * TE;
#pragma unroll
  for (int x = 0; x < NA; ++x) {
    // element e = ff * 32 + c of the 16 x 32 half tile x; lane holds rows 4q .. 4q+3 of columns r and 16 + r
    if (x > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wsum[wave][(4 * q + i) * 32 + r] = acc[x][0][i];
      wsum[wave][(4 * q + i) * 32 + 16 + r] = acc[x][1][i];
    }
    __syncthreads();
    if (tid < 512) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 16; ++w) v += wsum[w][tid];
      __hip_atomic_store(mine + 512 * x + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // Every storing wave drains its sc1 stores BEFORE the barrier that precedes the arrival count: a workgroup-scope
  // barrier alone emits no vmcnt wait on gfx950, and the relaxed counter add could then overtake the partial tile
  // (MI355X_MICROARCH.md, inter-workgroup visibility, condition (3)).  Inline asm: invisible to the waitcnt pass.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  AGCN_STAMP(7, 2);
  __syncthreads();
  if (tid == 0) {
    const int old = __hip_atomic_fetch_add(a.counters + t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last_flag = (old == S - 1) ? 1 : 0;
  }
  __syncthreads();
  AGCN_STAMP(7, 3);
  if (last_flag) {
    for (int e = tid; e < TE; e += 1024) {
      const float* base = a.part + (int64_t)t * S * TE + e;
      float v = 0.f;
      for (int s0 = 0; s0 < S; s0 += 8) {   // eight loads in flight, added in split order
        float l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          l[u] = __hip_atomic_load(base + (int64_t)min(s0 + u, S - 1) * TE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (s0 + u < S) v += l[u];
      }
      const int ff = f0 + (e >> 5), c = c0 + (e & 31);
      if (ff < Fa && c < Fb) {
        if (p == 0) a.dweight[(int64_t)ff * Fout + c] = v;
        else if (p == 1) a.dbias[c] = v;
        else a.dM_L[(int64_t)ff * Fin + c] = v;
      }
    }
  }
  AGCN_STAMP(7, 4);
  // scalars: one wave each of workgroup 0, fixed order (the per-graph partials come from the previous kernel)
  if (blockIdx.x == 0 && wave < 2) {
    float* dst = wave == 0 ? a.dalpha : a.dbeta;
    if (dst != nullptr && a.pg != nullptr) {
      const float v = pg_wave_sum(a.pg, a.B, wave, lane);
      if (lane == 0) dst[0] = v;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Tagged hand-off variant of the one-pass kernel (batches below 8k rows, TF = 16): the same tiles and products over eight row
// splits (sixteen from 5121 rows on), split % 8 = its XCD -- and the splits' partial tiles travel as 8-byte (value, tag) GRANULES:
// one agent-scope store per element, no drain of the stores, no arrival counter, no second round of loads.  The workgroup of
// a tile's LAST split (the shortest one; behind its tile's producers in dispatch order: they have all been dispatched) is the
// tile's reducer: it polls the S - 1 granules of each of its elements (all requested at once), adds them in split order, its own
// partial last, and writes the gradient.  Against the arrival-counter hand-off that removes two of the four memory round trips
// (MI355X_MICROARCH.md, hand-off price list: data-tagged granules ~1 us against ~1.8x that for payload + drained flag, + the
// counter's fan-in, + the reducer's dependent loads).  The granules' tags are CLEARED by the preceding kernel of the same call
// (the specialised backward: spec::zero_tags, a slice per workgroup), so no state outlives a call and a tag of 1 can only come
// from this launch's producers.  Waits are bounded as the cooperative kernels' (10 ms; NaN + the device status word).
// (two workgroups per CU -- at most 64 VGPRs -- so that every workgroup of a 256-graph batch is resident at once: with one per CU the
//  reducers, dispatched last, started 5 us into the launch)
template <int KS>
__global__ __launch_bounds__(1024, 8) void param_grad_tagged_kernel(ParamGradArgs a) {
  __shared__ float wsum[16][512 + 16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const int K = a.K, Fin = a.Fin, Fout = a.Fout;
  const int S = a.n_chunks;
  // split = block % S: with the EIGHT splits of a batch of 128 rows and more, and workgroups dealt round-robin over the 8 XCDs,
  // every workgroup of one row split runs on one XCD -- a row of X / X_1 / Gp / dZ is then fetched into ONE L2 instead of all
  // eight (the plain order's 370 workgroups waited 3.5 - 6.6 us for their operands: 23 MB over the fabric for 3.6 MB of rows).
  // A speed matter only.  The reducer (s = S - 1) follows its tile's producers in dispatch order.
  const int t = (int)blockIdx.x / S, s = (int)blockIdx.x - t * S;
  AGCN_STAMP(7, 0);
  // dweight tiles: 16 consecutive FEATURES of ONE Chebyshev term x 32 columns (rows f K + k of dweight, k fixed per tile) -- an A
  // fragment load then takes 64 contiguous bytes of a row of ONE array instead of 32 bytes of X_0 and 32 of X_1 (half the cache
  // lines); the sums of an output element are the same rows in the same order whatever the tiling
  const int tfM = (Fin + 15) / 16;
  const int tfW = K * tfM, tcW = (Fout + 31) >> 5;
  int p, rem;
  if (t < tfW * tcW) { p = 0; rem = t; }
  else if (t < tfW * tcW + tcW) { p = 1; rem = t - tfW * tcW; }
  else { p = 2; rem = t - tfW * tcW - tcW; }
  const int Fa = p == 1 ? 1 : Fin, Fb = p == 2 ? Fin : Fout;          // rows (features) / columns of this tile's matrix
  const int tf = p == 0 ? tfW : (p == 1 ? 1 : tfM);
  const int itc = rem / tf, itf = rem - itc * tf;
  const int kx = p == 0 ? itf / tfM : 0;                                // Chebyshev term of a dweight tile
  const int f0 = (p == 0 ? itf - kx * tfM : itf) * 16, c0 = itc * 32;
  const int last_row = a.rows - 1;
  const int r_begin = s * a.chunk_rows + wave * a.wgs_per_chunk;
  const int r_end = min(min(a.rows, (s + 1) * a.chunk_rows), r_begin + a.wgs_per_chunk);

  const int j = min(f0 + r, Fa - 1);
  const bool jv = (f0 + r) < Fa;
  const float* Ap;
  int64_t lda;
  if (p == 0) {
    if (kx == 0) { Ap = a.X + j; lda = a.ldx; }
    else { Ap = a.sv_X + (int64_t)(kx - 1) * a.rows * Fin + j; lda = Fin; }
  } else if (p == 2) { Ap = a.X + j; lda = a.ldx; }
  else { Ap = a.X; lda = 0; }
  const float* Bp = (p == 2) ? a.dZ : a.Gp;
  const int ldb = (p == 2) ? Fin : Fout;
  const int cA = min(c0 + r, Fb - 1), cB = min(c0 + 16 + r, Fb - 1);
  const bool cAv = (c0 + r) < Fb, cBv = (c0 + 16 + r) < Fb;

  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
  for (int rr = r_begin; rr < r_end; rr += 4 * KS) {
    float av[KS], b0[KS], b1[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const int row = min(rr + 4 * u + q, last_row);
      av[u] = (p == 1) ? 1.0f : Ap[(int64_t)row * lda];
      b0[u] = Bp[(int64_t)row * ldb + cA];
      b1[u] = Bp[(int64_t)row * ldb + cB];
    }
#pragma unroll
    for (int u = 0; u < KS; ++u) {
      const bool rv = (rr + 4 * u + q) < r_end;
      const float xv = (rv && jv) ? av[u] : 0.f;
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, (rv && cAv) ? b0[u] : 0.f, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, (rv && cBv) ? b1[u] : 0.f, acc1, 0, 0, 0);
    }
  }
  AGCN_STAMP(7, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    wsum[wave][(4 * q + i) * 32 + r] = acc0[i];
    wsum[wave][(4 * q + i) * 32 + 16 + r] = acc1[i];
  }
  __syncthreads();
  if (tid < 512) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) v += wsum[w][tid];
    if (s != S - 1) {
      // producer: one granule per element, fire and forget
      __hip_atomic_store(a.gran + ((int64_t)t * (S - 1) + s) * 512 + tid, ((uint64_t)1u << 32) | (uint64_t)__float_as_uint(v),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      AGCN_STAMP(7, 2);
      // reducer: the other splits' granules of this element -- ALL requested back to back (clamped addresses, no branch between
      // the loads: a load inside a branch region of its own is waited for on the spot, and nine dependent round trips of
      // agent-scope loads were 50 us), re-requested together until every tag is set
      const uint64_t* base = a.gran + (int64_t)t * (S - 1) * 512;     // (uniform: scalar base + one vector offset per load)
      const uint32_t toff = (uint32_t)tid;
      const unsigned long long t0 = wall_clock64();
      bool ok = true;
      float tot = 0.f;
      // (S <= 8: seven granules per element, one round; S = 16: two rounds of eight -- the registers of fifteen loads in flight
      //  would cost the second resident workgroup)
      for (int u0 = 0; u0 < S - 1 && ok; u0 += 8) {
        uint64_t w[8];
        for (;;) {
#pragma unroll
          for (int u = 0; u < 8; ++u)
            w[u] = __hip_atomic_load(base + min(u0 + u, S - 2) * 512 + toff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bool all = true;
#pragma unroll
          for (int u = 0; u < 8; ++u) all = all && ((uint32_t)(w[u] >> 32) == 1u);
          if (all) break;
          if (wall_clock64() - t0 >= (unsigned long long)a.ticks) { ok = false; break; }
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (u0 + u < S - 1) tot += __uint_as_float((uint32_t)w[u]);
      }
      AGCN_STAMP(7, 3);
      tot += v;
      if (!ok) {     // (never silently: NaN in the gradient + the device's status word)
        tot = __builtin_nanf("");
        if (a.dev_status != nullptr) __hip_atomic_fetch_add(a.dev_status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const int ff = f0 + (tid >> 5), c = c0 + (tid & 31);
      if (ff < Fa && c < Fb) {
        if (p == 0) a.dweight[(int64_t)(ff * K + kx) * Fout + c] = tot;
        else if (p == 1) a.dbias[c] = tot;
        else a.dM_L[(int64_t)ff * Fin + c] = tot;
      }
    }
  }
  AGCN_STAMP(7, 4);
  // scalars: one wave each of workgroup 0, fixed order (the per-graph partials come from the previous kernel)
  if (blockIdx.x == 0 && wave >= 8 && wave < 10) {
    float* dst = wave == 8 ? a.dalpha : a.dbeta;
    if (dst != nullptr && a.pg != nullptr) {
      const float v = pg_wave_sum(a.pg, a.B, wave - 8, lane);
      if (lane == 0) dst[0] = v;
    }
  }
}


namespace {
// output rows per tile of the one-pass kernel: 32 when the row count makes the kernel bandwidth-bound
int pg_tf(int rows) {
  static const int force = [] { const char* e = std::getenv("AGCN_PG_TF"); return e ? std::atoi(e) : 0; }();
  if (force == 16 || force == 32) return force;
  // 16 | 32-row tiles, plain workgroup order: 18k rows 28.8 | 34.4 us, 36k: 49 | 48.5, 73k: 165 | 110; with the
  // split-major XCD order (the default from 8k rows on) the 16-row tile wins everywhere: 73k rows 62.8 | 74.5 us
  (void)rows;
  return 16;
}
int pg_tiles(int Fin, int Fout, int K, bool paper, int TF) {
  const int tfW = (Fin * K + TF - 1) / TF, tcW = (Fout + 31) / 32;
  const int tfM = (Fin + TF - 1) / TF, tcM = (Fin + 31) / 32;
  return tfW * tcW + tcW + (paper ? tfM * tcM : 0);
}
// rows per wave (multiple of 32) and number of row splits, for `nt` output tiles
void pg_split(int rows, int nt, int* rows_per_wave, int* S) {
  static const int force = [] { const char* e = std::getenv("AGCN_PG_RPW"); return e ? std::atoi(e) : 0; }();
  int rpw = 32;   // measured: 32 / 64 / 96 rows per wave are within 5 %, 128 is 30 % slower
  // wide layers have many tiles: beyond ~800 workgroups (three rounds of 1024 threads per CU) twice the rows per
  // wave win -- 256 graphs, 128 -> 128 (100 tiles): 20.5 -> 17.1 us; 75 -> 64 (37 tiles): 12.7 -> 13.0 us
  if (rows < 8192 && nt * ((rows + 16 * rpw - 1) / (16 * rpw)) > 800) rpw = 64;   // (larger batches: measured with 32)
  while ((rows + 16 * rpw - 1) / (16 * rpw) > 64) rpw += 32;
  if (force >= 4) rpw = force / 4 * 4;
  *rows_per_wave = rpw;
  *S = std::max(1, (rows + 16 * rpw - 1) / (16 * rpw));
}
int pg_mode() { return knob_pg_onepass(); }
}  // namespace

// the tagged hand-off serves the batches below 8k rows: 16-row tiles, eight or sixteen row splits (split % 8 = its XCD) of 16 waves
// x rpw rows each.
// (32 x 32 tiles of 32 consecutive features of one Chebyshev term -- half the cache lines per output, one workgroup per CU --
//  measured equal at 75 -> 64 and slower at 128 -> 64, 67.1 against 62.9 us per step: profiles/EXPERIMENTS.md)
static bool pg_tagged_shape(int32_t rows, int32_t Fin, int32_t Fout, int32_t K, int32_t paper, int* nt_out, int* S_out, int* rpw_out) {
  static const int xcd_force = [] { const char* e = std::getenv("AGCN_PG_XCD"); return e ? std::atoi(e) : -1; }();
  if (rows < 1 || rows >= 8192 || xcd_force > 0 || pg_tf(rows) != 16) return false;
  // rows per wave: any number (the k-steps of 4 rows are masked), at most 40 = ONE round of operand loads (10 k-steps in
  // flight per lane); from 5121 rows on two splits per XCD (HIV-shaped batches, ~6.4 k rows: 50 rows per wave in two rounds
  // measured 1 % SLOWER than the arrival-counter kernel, profiles/EXPERIMENTS.md)
  static const int s_force = [] { const char* e = std::getenv("AGCN_PG_TAG_SPLITS"); return e ? std::atoi(e) : 0; }();
  const int s_target = (s_force >= 2 && s_force <= 16) ? s_force : (rows <= 5120 ? 8 : 16);
  const int rpw = (rows + 16 * s_target - 1) / (16 * s_target);
  const int tfM = (Fin + 15) / 16, tc = (Fout + 31) / 32;   // dweight tiles per Chebyshev term (see the kernel)
  *nt_out = K * tfM * tc + tc + (paper ? tfM * ((Fin + 31) / 32) : 0);
  *S_out = (rows + 16 * rpw - 1) / (16 * rpw);              // <= s_target
  *rpw_out = rpw;
  return true;
}
// floats of the granules: 512 8-byte words per (tile, producing split)
static size_t pg_tagged_floats(int nt, int S) { return (size_t)nt * (S - 1) * 1024 + 4; }

// floats of `part` the one-pass kernel needs: S partial tiles per output tile + the arrival counters, or (tagged hand-off)
// S - 1 tiles of 8-byte granules per output tile
size_t param_grad_onepass_floats(int32_t rows, int32_t Fin, int32_t Fout, int32_t K, int32_t paper) {
  int rpw, S;
  const int TF = pg_tf(rows);
  const size_t nt = (size_t)pg_tiles(Fin, Fout, K, paper != 0, TF);
  pg_split(rows, (int)nt, &rpw, &S);
  size_t need = nt * S * (TF * 32) + ((nt + 63) / 64) * 64;
  int nt2, S2, rpw2;
  if (pg_tagged_shape(rows, Fin, Fout, K, paper, &nt2, &S2, &rpw2)) need = std::max(need, pg_tagged_floats(nt2, S2));
  return need;
}

float* param_grad_tag_region(float* part, size_t part_floats, int32_t rows, int32_t Fin, int32_t Fout, int32_t K, int32_t paper,
                             int32_t* count16) {
  *count16 = 0;
  int nt, S, rpw;
  if (pg_mode() != 2 || !part || !pg_tagged_shape(rows, Fin, Fout, K, paper, &nt, &S, &rpw)) return nullptr;
  if (part_floats < pg_tagged_floats(nt, S) || (reinterpret_cast<uintptr_t>(part) & 15) != 0) return nullptr;
  *count16 = nt * (S - 1) * 256;          // 512 granules of 8 bytes per (tile, split)
  return part;
}

int32_t* param_grad_counters(float* part, size_t part_floats, int32_t rows, int32_t Fin, int32_t Fout, int32_t K,
                             int32_t paper, int32_t* count) {
  *count = 0;
  if (pg_mode() < 1 || !part || part_floats < param_grad_onepass_floats(rows, Fin, Fout, K, paper)) return nullptr;
  int rpw, S;
  const int TF = pg_tf(rows);
  const int nt = pg_tiles(Fin, Fout, K, paper != 0, TF);
  pg_split(rows, nt, &rpw, &S);
  *count = nt;
  return reinterpret_cast<int32_t*>(part + (size_t)nt * S * (TF * 32));
}

int launch_param_grads(const ParamGradArgs& a, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a.tags_zeroed) {     // the caller had the preceding kernel clear the granules: param_grad_tag_region said yes
    int nt, S, rpw;
    if (!pg_tagged_shape(a.rows, a.Fin, a.Fout, a.K, a.paper, &nt, &S, &rpw) || a.part_floats < pg_tagged_floats(nt, S))
      return fail(AGCN_EINVAL, "param_grads: tagged hand-off requested for a shape it does not serve");
    ParamGradArgs aa = a;
    aa.n_chunks = S; aa.wgs_per_chunk = rpw; aa.chunk_rows = 16 * rpw;
    aa.n_tiles = nt;
    aa.gran = reinterpret_cast<uint64_t*>(a.part);
    prof_begin(PK_PARAM_TAGGED, stream);
    hipLaunchKernelGGL((param_grad_tagged_kernel<10>), dim3((unsigned)(nt * S)), dim3(1024), 0, s, aa);
    prof_end(stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "param_grad_tagged launch: %s", hipGetErrorString(e));
    return AGCN_OK;
  }
  if (pg_mode() >= 1 && a.part_floats >= param_grad_onepass_floats(a.rows, a.Fin, a.Fout, a.K, a.paper)) {
    ParamGradArgs aa = a;
    int rpw, S;
    const int TF = pg_tf(a.rows);
    const int nt = pg_tiles(a.Fin, a.Fout, a.K, a.paper != 0, TF);
    pg_split(a.rows, nt, &rpw, &S);
    aa.n_chunks = S; aa.wgs_per_chunk = rpw; aa.chunk_rows = 16 * rpw;
    aa.counters = reinterpret_cast<int32_t*>(a.part + (size_t)nt * S * (TF * 32));
    if (!a.counters_zeroed) {
      hipError_t e = hipMemsetAsync(aa.counters, 0, sizeof(int32_t) * (size_t)nt, s);
      if (e != hipSuccess) return fail(AGCN_EDEVICE, "param_grad counters memset: %s", hipGetErrorString(e));
    }
    prof_begin(PK_PARAM_ONEPASS, stream);
    // XCD-aware order: off for small batches (the rows fit every L2; the plain order has the shorter tail:
    // B = 256: 13.6 vs 15.4 us), split-major from ~8k rows on (512 / 1024 / 2048 / 4096 graphs: 19.5 / 26.3 / 38.3 /
    // 74.6 us against 19.9 / 29.7 / 47.7 / 111 us)
    static const int xcd_force = [] { const char* e = std::getenv("AGCN_PG_XCD"); return e ? std::atoi(e) : -1; }();
    const int xcd = xcd_force >= 0 ? xcd_force : (a.rows >= 8192 ? 2 : 0);
    aa.xcd = xcd;
    const dim3 grid((unsigned)(nt * (xcd ? ((S + 7) & ~7) : S)));
    if (TF == 32) hipLaunchKernelGGL((param_grad_onepass_kernel<32, 8>), grid, dim3(1024), 0, s, aa);
    else hipLaunchKernelGGL((param_grad_onepass_kernel<16, 8>), grid, dim3(1024), 0, s, aa);
    prof_end(stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AGCN_EDEVICE, "param_grad_onepass launch: %s", hipGetErrorString(e));
    return AGCN_OK;
  }
  const int tfW = (a.Fin * a.K + 15) / 16, tcW = (a.Fout + 31) / 32;
  const int tfM = (a.Fin + 15) / 16, tcM = (a.Fin + 31) / 32;
  int items = tfW * tcW;
  if (a.paper && tfM * tcM > items) items = tfM * tcM;
  ParamGradArgs aa = a;
  aa.wgs_per_chunk = (items + 3) / 4;
  const int gx = 8 * aa.wgs_per_chunk * ((a.n_chunks + 7) / 8);
  const int nprob = a.paper ? 3 : 2;
  prof_begin(PK_PARAM_PARTIAL, stream);
  hipLaunchKernelGGL(param_grad_partial_kernel, dim3(gx, nprob), dim3(256), 0, s, aa);
  prof_end(stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "param_grad_partial launch: %s", hipGetErrorString(e));
  const int64_t total = (int64_t)a.K * a.Fin * a.Fout + a.Fout + (a.paper ? (int64_t)a.Fin * a.Fin : 0);
  prof_begin(PK_PARAM_REDUCE, stream);
  hipLaunchKernelGGL(param_grad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
  prof_end(stream);
  e = hipGetLastError();
  if (e != hipSuccess) return fail(AGCN_EDEVICE, "param_grad_reduce launch: %s", hipGetErrorString(e));
  return AGCN_OK;
}

}  // namespace agcn
