// kernels.hpp -- hand-written gfx950 (CDNA4, wave64) kernels of the GMG hot path.
//
// All kernels are HBM-bandwidth bound (~0.16 flop/byte): no MFMA.  Four operator layouts,
// all produced at setup from the caller's CSR/CSC (profiles/r01_tuning.md has the numbers):
//   * SELL-64 (`sell_kernel`): lane = row, slices of 64 rows stored column-major, so the
//     (col,val) stream, the row-wise operands and (for banded matrices) the gather are all
//     coalesced; no LDS, barrier or shuffle; rows summed left to right (bit-identical to a
//     sequential CPU SpMV).  Chosen when the padding is small.
//   * SELL-C (`sellc_kernel`): the same kernel fed by a losslessly compressed stream -- 8-bit
//     value dictionary (LDS) + 16-bit column offsets per slice column: 12 -> 3 B/nnz.
//   * SELL-P (`sellp_kernel`): row-pattern dictionary for structured constant-coefficient
//     operators: a 16-bit pattern id per row, the (offset,value) patterns in LDS.
//   * CSR-stream (`csr_stream1_kernel`): block-wide coalesced loads of a 2048-nnz tile,
//     products staged in LDS, G lanes per row with a wave64 __shfl_xor tail -- the generic
//     path for ragged matrices (P) and rows longer than a tile.
// Element-wise work of the reference's Richardson sweep is fused into the mat-vec
// (EPI_SWEEP); the gathered vector lives in L2 / Infinity Cache, the matrix stream is read
// with non-temporal loads when it is larger than the cache.
//
// Reference operations realised here (GridapSolvers.jl v0.7.1, src/):
//   K1 mul!(y,A,x)            LinearSolvers/RichardsonSmoothers.jl:94, GMGLinearSolvers.jl:495, CGSolvers.jl:79,104
//   K2 dx=w*Dinv*r; x+=dx     RichardsonSmoothers.jl:91-93, JacobiLinearSolvers.jl:43-47
//   K3 r -= A dx              RichardsonSmoothers.jl:95, GMGLinearSolvers.jl:496
//   K4 rH = R rh              GMGLinearSolvers.jl:484
//   K5 dxh = P dxH; xh += dxh GMGLinearSolvers.jl:491,494
//   K6 coarse solve           GMGLinearSolvers.jl:474
//   K7 dot / norm             CGSolvers.jl:85,95,105,111 ; FGMRESSolvers.jl:141,161,164
//   K8 axpy-class             CGSolvers.jl:101,108-109 ; FGMRESSolvers.jl:146,162,165,192
//   K9/K10 patch solves       PatchBasedSmoothers/PatchSolvers.jl:279-300, BlockJacobiSolvers.jl:141-170
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gmg {

constexpr int kBlock = 256;   // 4 waves of 64
constexpr int kTile = 2048;   // nnz staged per workgroup (16 KiB of fp64 products in LDS)

enum Epi : int {
  EPI_SET = 0,     // y = A x
  EPI_SUB = 1,     // y = y - A x                  (K3)
  EPI_RESID = 2,   // y = b - A x                  (CGSolvers.jl:79, GMGLinearSolvers.jl:623-624)
  EPI_SWEEP = 3,   // one Richardson-Jacobi sweep  (K2+K1+K3 fused, see below)
  EPI_ADDTO = 4    // y = A x ; x2 += y            (K5); with omega != 0: y = omega (A x) ; x2 += y  (relaxed patch-operator update)
};

__device__ __forceinline__ int remap_block(int b, int nb, int on)
{
  // Observed dispatch: workgroup b runs on XCD b % 8 (performance only, never
  // correctness).  Give XCD k the contiguous chunk [k*nb/8, (k+1)*nb/8).
  if (!on || nb < 64) return b;
  if (on > 1) {
    // chunked: inside every group of 8*on consecutive blocks XCD k takes the on blocks [k*on, (k+1)*on) -- each L2 still sees
    // contiguous rows, but the eight XCDs stream from one window of 8*on blocks instead of eight far-apart eighths (HBM locality)
    const int G = on << 3, g = b / G;
    if ((g + 1) * G > nb) return b;                           // partial last group: identity
    const int wi = b - g * G;
    return g * G + (wi & 7) * on + (wi >> 3);
  }
  const int per = nb >> 3, rem = nb & 7;
  const int xcd = b & 7, slot = b >> 3;
  // XCDs < rem own (per+1) blocks
  const int base = xcd * per + (xcd < rem ? xcd : rem);
  return base + slot;
}

// ---------------------------------------------------------------------------
// CSR-stream kernel: the generic path (ragged matrices such as P, rows longer than a tile).
//
// The host partitions the rows so that every workgroup owns <= (256 >> lanes_log2)
// rows and <= TILE nnz (a row longer than TILE is a workgroup of its own).  Every
// row then has its G = 1<<lanes_log2 lanes for the whole kernel, which lets the
// row pointers and all row-wise epilogue operands be fetched BEFORE the stream
// phase: after the barrier the kernel touches only LDS and issues its stores.
//
// EPI_SWEEP has two forms:
//   ONEG = false : two gathers per nnz (dinv_j, r_j), r ping-pong, 12Z+44N bytes;
//   ONEG = true  : one gather per nnz from s = omega*(dinv.*r), which the previous
//                  sweep's epilogue produced (s ping-pong, r in place), 12Z+60N bytes.
// Both give bit-identical dx_j = omega*(dinv_j*r_j) (same two roundings as
// RichardsonSmoothers.jl:91-92 / JacobiLinearSolvers.jl:45).
// ---------------------------------------------------------------------------
struct StreamArgs2 {
  const void *rowptr;
  const int32_t *col;
  const double *val;
  const int32_t *blk_row;
  int nblocks;
  int lanes_log2;
  int xcd_remap;
  int x_zero;
  const double *x;     // gather source (ONEG sweep: s_old)
  const double *dinv;
  double omega;
  double *y;           // output (sweep: r_new; ONEG: r in place)
  const double *b;     // RESID: b ; sweep: r_old (row-wise)
  double *x2;          // sweep / ADDTO: x
  double *s_out;       // ONEG sweep and *_S epilogues: s_new = omega*(dinv.*r_new)
  int xmode;           // sells_kernel sweeps only: see SellSArgs::xmode
  int dinv_from_table; // sells_kernel sweeps only: the level's dinv is 1/diag of this very matrix
};

template <int EPI, typename PtrT, int TILE, bool ONEG, bool EMIT_S>
__global__ __launch_bounds__(kBlock) void csr_stream1_kernel(StreamArgs2 a)
{
  __shared__ double prod[TILE];
  const int tid = threadIdx.x;
  const int blk = remap_block(blockIdx.x, a.nblocks, a.xcd_remap);
  const PtrT *__restrict__ rowptr = reinterpret_cast<const PtrT *>(a.rowptr);
  const int r0 = a.blk_row[blk];
  const int r1 = a.blk_row[blk + 1];
  const PtrT nz0 = rowptr[r0];
  const PtrT nz1 = rowptr[r1];
  const int64_t cnt = (int64_t)(nz1 - nz0);
  const int32_t *__restrict__ col = a.col;
  const double *__restrict__ val = a.val;
  const double *__restrict__ xg = a.x;
  const double *__restrict__ dinv = a.dinv;
  const double omega = a.omega;

  if (cnt <= TILE) {
    // ---- row-owner prefetch (independent of the stream phase) ----
    const int lg = a.lanes_log2;
    const int G = 1 << lg;
    const int sub = tid & (G - 1);
    const int row = r0 + (tid >> lg);
    const bool active = row < r1;
    const bool owner = active && sub == 0;
    int k0 = 0, k1 = 0;
    double e0 = 0.0, e1 = 0.0, e2 = 0.0; // epilogue operands
    if (active) {
      k0 = (int)(rowptr[row] - nz0);
      k1 = (int)(rowptr[row + 1] - nz0);
    }
    if (owner) {
      if (EPI == EPI_SUB) e0 = a.y[row];
      else if (EPI == EPI_RESID) e0 = a.b[row];
      else if (EPI == EPI_ADDTO) e0 = a.x2[row];
      else if (EPI == EPI_SWEEP) {
        e0 = a.b[row];                               // r_old
        e1 = ONEG ? xg[row] : dinv[row];             // s_old | dinv
        e2 = a.x_zero ? 0.0 : a.x2[row];             // x
      }
    }
    double dinv_row = 0.0;
    if (owner && (EMIT_S || (EPI == EPI_SWEEP && ONEG))) dinv_row = dinv[row];

    // ---- phase 1: coalesced stream, gather, products -> LDS ----
    constexpr int U = TILE / kBlock;
    int32_t c[U];
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = tid + u * kBlock;
      const bool ok = k < cnt;
      c[u] = ok ? __builtin_nontemporal_load(col + nz0 + k) : -1;
      v[u] = ok ? __builtin_nontemporal_load(val + nz0 + k) : 0.0;
    }
    double g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int cc = c[u] >= 0 ? c[u] : 0;
      if (EPI == EPI_SWEEP && !ONEG) g[u] = omega * (dinv[cc] * xg[cc]);
      else g[u] = xg[cc];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = tid + u * kBlock;
      if (c[u] >= 0) prod[k] = v[u] * g[u];
    }
    __syncthreads();
    // ---- phase 2: G lanes per row, LDS only ----
    double s = 0.0;
    for (int k = k0 + sub; k < k1; k += G) s += prod[k];
    for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (owner) {
      double rn = 0.0;
      if (EPI == EPI_SET) { rn = s; a.y[row] = rn; }
      else if (EPI == EPI_SUB) { rn = e0 - s; a.y[row] = rn; }
      else if (EPI == EPI_RESID) { rn = e0 - s; a.y[row] = rn; }
      else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * s : s; a.y[row] = t; a.x2[row] = e0 + t; }
      else { // EPI_SWEEP
        const double dxi = ONEG ? e1 : omega * (e1 * e0);
        a.x2[row] = e2 + dxi;
        rn = e0 - s;
        a.y[row] = rn;
      }
      if (EMIT_S || (EPI == EPI_SWEEP && ONEG)) a.s_out[row] = omega * (dinv_row * rn);
    }
  } else {
    // ---- long row: the range is a single row; whole workgroup strides it ----
    double s = 0.0;
    for (int64_t k = tid; k < cnt; k += kBlock) {
      const int32_t cc = col[nz0 + k];
      double xv;
      if (EPI == EPI_SWEEP && !ONEG) xv = omega * (dinv[cc] * xg[cc]);
      else xv = xg[cc];
      s += val[nz0 + k] * xv;
    }
    prod[tid] = s;
    __syncthreads();
    for (int w = kBlock >> 1; w > 0; w >>= 1) {
      if (tid < w) prod[tid] += prod[tid + w];
      __syncthreads();
    }
    if (tid == 0) {
      const int row = r0;
      s = prod[0];
      double rn = 0.0;
      if (EPI == EPI_SET) { rn = s; a.y[row] = rn; }
      else if (EPI == EPI_SUB) { rn = a.y[row] - s; a.y[row] = rn; }
      else if (EPI == EPI_RESID) { rn = a.b[row] - s; a.y[row] = rn; }
      else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * s : s; a.y[row] = t; a.x2[row] = a.x2[row] + t; }
      else {
        const double ro = a.b[row];
        const double dxi = ONEG ? xg[row] : omega * (dinv[row] * ro);
        const double xo = a.x_zero ? 0.0 : a.x2[row];
        a.x2[row] = xo + dxi;
        rn = ro - s;
        a.y[row] = rn;
      }
      if (EMIT_S || (EPI == EPI_SWEEP && ONEG)) a.s_out[row] = omega * (dinv[row] * rn);
    }
  }
}

// ---------------------------------------------------------------------------
// SELL-64 kernels (sliced ELLPACK, slice height = one wave64, column-major inside a
// slice, built at setup from the caller's CSR when the padding it needs is small).
//
// lane == row: the j-th stored entries of 64 consecutive rows are contiguous, so the
// (col,val) stream AND all row-wise epilogue operands are perfectly coalesced, the
// gather x[col] is contiguous across lanes for banded/stencil matrices, and there is
// no LDS staging, barrier or shuffle at all.  Each row is summed left to right in
// its original CSR order -- the same order as a sequential CPU SpMV (bit-identical
// to the oracle's mul!).  Memory-level parallelism comes from UN independent
// (col,val) loads + UN gathers in flight per lane and many waves per CU.
// ---------------------------------------------------------------------------
struct SellArgs {
  const int64_t *soff;     // [nslices+1] offsets into scol/sval (multiples of 64)
  const int32_t *scol;     // padded, column-major per slice; padding points at a valid column
  const double *sval;      // padding = 0.0 (masked out by rowlen, never added)
  const int32_t *rowlen;   // [nrows]
  int64_t nrows;
  int nslices;
  int x_zero;
  int xcd_remap;           // contiguous slice ranges per XCD (L2 locality of the gather on big levels)
  const double *x;         // gather source (ONEG sweep: s_old)
  const double *dinv;
  double omega;
  double *y;
  const double *b;         // RESID: b ; sweep: r_old
  double *x2;              // sweep / ADDTO: x
  double *s_out;           // ONEG sweep: s_new
};

// NT: 0 cached matrix stream ; 1 non-temporal matrix stream ; 2 also the row-wise operands that are touched once per sweep
// (r, x, 1/diag in; r, x out: non-temporal; s stays cacheable -- the next sweep gathers it)
// XM (fused one-gather sweeps only): how x is updated, as in the row-pattern kernels -- 0: x += s_k ; 1: x untouched (the
// increment is applied by the next sweep) ; 2: x = (x + s_{k-1}) + s_k with s_{k-1} read from s_out before it is overwritten:
// the same two roundings, one read + one write of x less per pair of sweeps (writes are what this kernel pays most for).
template <int EPI, bool ONEG, int UN, int NT, int XM = 0>
__global__ __launch_bounds__(kBlock) void sell_kernel(SellArgs a)
{
  const int lane = threadIdx.x & 63;
  const int slice = remap_block(blockIdx.x, gridDim.x, a.xcd_remap) * (blockDim.x >> 6) + (threadIdx.x >> 6);   // any block size that is a multiple of 64
  if (slice >= a.nslices) return;
  const int64_t base = a.soff[slice];
  const int w = (int)((a.soff[slice + 1] - base) >> 6);
  const int64_t row = (int64_t)slice * 64 + lane;
  const bool valid = row < a.nrows;
  const int len = valid ? a.rowlen[row] : 0;
  const double *__restrict__ xg = a.x;
  const double *__restrict__ dinv = a.dinv;
  const double omega = a.omega;
  // row-wise epilogue operands (coalesced), issued before the stream
  double e0 = 0.0, e1 = 0.0, e2 = 0.0, dinv_row = 0.0, sp = 0.0;
  if (valid) {
    if (EPI == EPI_SUB) e0 = a.y[row];
    else if (EPI == EPI_RESID) e0 = a.b[row];
    else if (EPI == EPI_ADDTO) e0 = a.x2[row];
    else if (EPI == EPI_SWEEP) {
      if (NT >= 2) {
        e0 = __builtin_nontemporal_load(a.b + row);
        e1 = ONEG ? xg[row] : __builtin_nontemporal_load(dinv + row);
        if (XM != 1) e2 = a.x_zero ? 0.0 : __builtin_nontemporal_load(a.x2 + row);
        if (ONEG) dinv_row = __builtin_nontemporal_load(dinv + row);
      } else {
      e0 = a.b[row];
      e1 = ONEG ? xg[row] : dinv[row];
      if (XM != 1) e2 = a.x_zero ? 0.0 : a.x2[row];
      if (ONEG) dinv_row = dinv[row];
      }
      if (XM == 2) sp = a.s_out[row];                        // s_{k-1}, about to be overwritten by s_{k+1}
    }
  }
  const int32_t *cp = a.scol + base + lane;
  const double *vp = a.sval + base + lane;
  double s = 0.0;
  int j = 0;
  for (; j + UN <= w; j += UN) {
    int32_t c[UN];
    double v[UN], g[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (NT) { c[u] = __builtin_nontemporal_load(cp + (int64_t)(j + u) * 64); v[u] = __builtin_nontemporal_load(vp + (int64_t)(j + u) * 64); }
      else { c[u] = cp[(int64_t)(j + u) * 64]; v[u] = vp[(int64_t)(j + u) * 64]; }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {    // unconditional gathers: padding points at a valid column
      if (EPI == EPI_SWEEP && !ONEG) g[u] = omega * (dinv[c[u]] * xg[c[u]]);
      else g[u] = xg[c[u]];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const double pr = v[u] * g[u];
      s = (j + u < len) ? s + pr : s;   // masked: padding is never added
    }
  }
  for (; j < w; ++j) {
    const int32_t c = cp[(int64_t)j * 64];
    const double v = vp[(int64_t)j * 64];
    double g;
    if (EPI == EPI_SWEEP && !ONEG) g = omega * (dinv[c] * xg[c]);
    else g = xg[c];
    const double pr = v * g;
    s = (j < len) ? s + pr : s;
  }
  if (valid) {
    if (EPI == EPI_SET) a.y[row] = s;
    else if (EPI == EPI_SUB) a.y[row] = e0 - s;
    else if (EPI == EPI_RESID) a.y[row] = e0 - s;
    else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * s : s; a.y[row] = t; a.x2[row] = e0 + t; }
    else {
      const double dxi = ONEG ? e1 : omega * (e1 * e0);
      const double rn = e0 - s;
      const double xn = XM == 2 ? (e2 + sp) + dxi : e2 + dxi;
      if (NT >= 2) { if (XM != 1) __builtin_nontemporal_store(xn, a.x2 + row); __builtin_nontemporal_store(rn, a.y + row); }
      else { if (XM != 1) a.x2[row] = xn; a.y[row] = rn; }
      if (ONEG) a.s_out[row] = omega * (dinv_row * rn);
    }
  }
}

// ---------------------------------------------------------------------------
// sell_kernel with the matrix stream kept in flight.  A lane's work is a chain  (col,val) load -> gather x[col] -> multiply-add
// per batch of UN entries: in sell_kernel the next batch's stream loads are only issued after the current gathers have been
// consumed, so per batch a wave exposes L_stream + L_gather and carries stream bytes in flight for L_stream of it.  Once the
// gathered vector leaves the L2s (levels >= 256^3: L_gather roughly doubles) that share drops and with it the achieved
// bandwidth (0.78 -> 0.64 of peak; HBM traffic stays at the algorithmic bytes, so it is latency, not re-reads).  Here the
// stream loads of batch k+PD are issued BEFORE the gathers of batch k are waited for (loads return in order, so the wait for
// the gathers -- vmcnt(2*UN*PD) -- leaves the younger stream loads in flight).  Same products, same left-to-right order:
// bit-identical to sell_kernel.  Column / value indices past the slice width are clamped to its last column (masked by rowlen).
// ---------------------------------------------------------------------------
template <int EPI, bool ONEG, int UN, int PD, bool NT>
__global__ __launch_bounds__(kBlock) void sell_pipe_kernel(SellArgs a)
{
  const int lane = threadIdx.x & 63;
  const int slice = __builtin_amdgcn_readfirstlane((int)(remap_block(blockIdx.x, gridDim.x, a.xcd_remap) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  if (slice >= a.nslices) return;
  const int64_t base = a.soff[slice];
  const int w = (int)((a.soff[slice + 1] - base) >> 6);
  const int64_t row = (int64_t)slice * 64 + lane;
  const bool valid = row < a.nrows;
  const int64_t rc = valid ? row : a.nrows - 1;
  const double *__restrict__ xg = a.x;
  const double *__restrict__ dinv = a.dinv;
  const double omega = a.omega;
  const int32_t *cp = a.scol + base + lane;
  const double *vp = a.sval + base + lane;
  const int wl = w - 1;
  int32_t c[PD + 1][UN];
  double v[PD + 1][UN];
  auto issue = [&](int slot, int j0) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t o = (int64_t)min(j0 + u, wl) * 64;
      if (NT) { c[slot][u] = __builtin_nontemporal_load(cp + o); v[slot][u] = __builtin_nontemporal_load(vp + o); }
      else { c[slot][u] = cp[o]; v[slot][u] = vp[o]; }
    }
  };
  // the first PD batches of the stream, then the row-wise operands (clamped addresses: no divergent loads)
#pragma unroll
  for (int p = 0; p < PD; ++p) issue(p, p * UN);
  int len = a.rowlen[rc];
  if (!valid) len = 0;
  double e0 = 0.0, e1 = 0.0, e2 = 0.0, dinv_row = 0.0;
  if (EPI == EPI_SUB) e0 = a.y[rc];
  else if (EPI == EPI_RESID) e0 = a.b[rc];
  else if (EPI == EPI_ADDTO) e0 = a.x2[rc];
  else if (EPI == EPI_SWEEP) {
    e0 = a.b[rc];
    e1 = ONEG ? xg[rc] : dinv[rc];
    { const double xl = a.x2[rc]; e2 = a.x_zero ? 0.0 : xl; }
    if (ONEG) dinv_row = dinv[rc];
  }
  double s = 0.0;
  // steady state: the stream loads of batch j/UN + PD are issued unconditionally inside the loop body (a conditional issue makes
  // the compiler wait at the join as if they had not been issued: vmcnt(UN-1) instead of vmcnt(2*UN*PD + UN-1)); drain loop after
  auto batch = [&](int j, bool more) {
    double g[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (EPI == EPI_SWEEP && !ONEG) g[u] = omega * (dinv[c[0][u]] * xg[c[0][u]]);
      else g[u] = xg[c[0][u]];
    }
    if (more) issue(PD, j + PD * UN);                      // stays in flight across the gather wait below
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const double pr = v[0][u] * g[u];
      s = (j + u < len) ? s + pr : s;                      // masked: padding is never added
    }
#pragma unroll
    for (int p = 0; p < PD; ++p) {
#pragma unroll
      for (int u = 0; u < UN; ++u) { c[p][u] = c[p + 1][u]; v[p][u] = v[p + 1][u]; }
    }
  };
  int j = 0;
  for (; j + PD * UN < w; j += UN) batch(j, true);
  for (; j < w; j += UN) batch(j, false);
  if (valid) {
    if (EPI == EPI_SET) a.y[row] = s;
    else if (EPI == EPI_SUB) a.y[row] = e0 - s;
    else if (EPI == EPI_RESID) a.y[row] = e0 - s;
    else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * s : s; a.y[row] = t; a.x2[row] = e0 + t; }
    else {
      const double dxi = ONEG ? e1 : omega * (e1 * e0);
      a.x2[row] = e2 + dxi;
      const double rn = e0 - s;
      a.y[row] = rn;
      if (ONEG) a.s_out[row] = omega * (dinv_row * rn);
    }
  }
}

// Whole rows in flight: slices of width <= WMAX request every (col,val) pair of the lane's row at once, then every gather,
// then sum left to right -- one stream latency + one gather latency per slice instead of one pair per batch, at the price of
// ~3*WMAX registers (4 waves per SIMD at WMAX = 27).  Wider slices take the batched loop.  Bit-identical to sell_kernel.
template <int EPI, bool ONEG, int WMAX, bool NT>
__global__ __launch_bounds__(kBlock) void sell_row_kernel(SellArgs a)
{
  const int lane = threadIdx.x & 63;
  const int slice = __builtin_amdgcn_readfirstlane((int)(remap_block(blockIdx.x, gridDim.x, a.xcd_remap) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  if (slice >= a.nslices) return;
  const int64_t base = a.soff[slice];
  const int w = (int)((a.soff[slice + 1] - base) >> 6);
  const int64_t row = (int64_t)slice * 64 + lane;
  const bool valid = row < a.nrows;
  const int64_t rc = valid ? row : a.nrows - 1;
  const double *__restrict__ xg = a.x;
  const double *__restrict__ dinv = a.dinv;
  const double omega = a.omega;
  const int32_t *cp = a.scol + base + lane;
  const double *vp = a.sval + base + lane;
  double s = 0.0;
  int len;
  double e0 = 0.0, e1 = 0.0, e2 = 0.0, dinv_row = 0.0;
  auto head = [&]() {
    len = a.rowlen[rc];
    if (!valid) len = 0;
    if (EPI == EPI_SUB) e0 = a.y[rc];
    else if (EPI == EPI_RESID) e0 = a.b[rc];
    else if (EPI == EPI_ADDTO) e0 = a.x2[rc];
    else if (EPI == EPI_SWEEP) {
      e0 = a.b[rc];
      e1 = ONEG ? xg[rc] : dinv[rc];
      { const double xl = a.x2[rc]; e2 = a.x_zero ? 0.0 : xl; }
      if (ONEG) dinv_row = dinv[rc];
    }
  };
  if (w <= WMAX) {
    const int wl = w - 1;
    int32_t c[WMAX];
    double v[WMAX], g[WMAX];
#pragma unroll
    for (int u = 0; u < WMAX; ++u) {
      const int64_t o = (int64_t)min(u, wl) * 64;
      c[u] = NT ? __builtin_nontemporal_load(cp + o) : cp[o];
    }
#pragma unroll
    for (int u = 0; u < WMAX; ++u) {
      const int64_t o = (int64_t)min(u, wl) * 64;
      v[u] = NT ? __builtin_nontemporal_load(vp + o) : vp[o];
    }
    head();
#pragma unroll
    for (int u = 0; u < WMAX; ++u) {
      if (EPI == EPI_SWEEP && !ONEG) g[u] = omega * (dinv[c[u]] * xg[c[u]]);
      else g[u] = xg[c[u]];
    }
#pragma unroll
    for (int u = 0; u < WMAX; ++u) {
      const double pr = v[u] * g[u];
      s = (u < len) ? s + pr : s;
    }
  } else {
    head();
    constexpr int UN = 9;
    for (int j = 0; j < w; j += UN) {
      int32_t c[UN];
      double v[UN], g[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int64_t o = (int64_t)min(j + u, w - 1) * 64;
        c[u] = cp[o]; v[u] = vp[o];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (EPI == EPI_SWEEP && !ONEG) g[u] = omega * (dinv[c[u]] * xg[c[u]]);
        else g[u] = xg[c[u]];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const double pr = v[u] * g[u];
        s = (j + u < len) ? s + pr : s;
      }
    }
  }
  if (valid) {
    if (EPI == EPI_SET) a.y[row] = s;
    else if (EPI == EPI_SUB) a.y[row] = e0 - s;
    else if (EPI == EPI_RESID) a.y[row] = e0 - s;
    else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * s : s; a.y[row] = t; a.x2[row] = e0 + t; }
    else {
      const double dxi = ONEG ? e1 : omega * (e1 * e0);
      a.x2[row] = e2 + dxi;
      const double rn = e0 - s;
      a.y[row] = rn;
      if (ONEG) a.s_out[row] = omega * (dinv_row * rn);
    }
  }
}

// x[byte offset]: uniform base + 32-bit lane offset (the saddr + voffset form of global_load)
__device__ __forceinline__ double ld_off(const double *__restrict__ base, uint32_t byteoff)
{
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byteoff);
}

// ---------------------------------------------------------------------------
// SELL-O ("offset patterns"): the SELL-64 kernel without its column stream.  Variable-coefficient operators on
// structured meshes have as many distinct VALUES as entries, but their column STRUCTURE is a handful of offset
// patterns (27-point stencil: 27 boundary types).  The setup detects that (rows equal up to their values), keeps the
// value stream of SELL-64 (8 B/nnz, coalesced) and replaces the column stream (4 B/nnz) by a 16-bit offset-pattern id
// per row; the offsets come from a small LDS table.  Same products, same left-to-right order as sell_kernel / the CSR
// gather: bit-identical.  12 -> 8 B per stored nonzero.
// ---------------------------------------------------------------------------
struct SellOArgs {
  const int64_t *soff;      // [nslices+1] offsets into sval (multiples of 64)
  const double *sval;       // padded, column-major per slice (padding = 0.0, masked by rowlen)
  const int32_t *rowlen;
  const uint16_t *rowpid;   // [nrows] offset pattern of each row
  const int32_t *rowbase;   // [nrows] base column, or nullptr: offsets are relative to the row index
  const int32_t *poff;      // [np*W] BYTE offsets (8 * column offset), zero padded (the last pattern is empty)
  int np, W;
  int64_t nrows;
  int nslices;
  int x_zero;
  int xcd_remap;
  const double *x;
  const double *dinv;
  double omega;
  double *y;
  const double *b;
  double *x2;
  double *s_out;
};

template <int EPI, bool ONEG, int UN, int NT, int XM = 0>
__global__ __launch_bounds__(kBlock) void sello_kernel(SellOArgs a)
{
  extern __shared__ double sp_smem[];
  int32_t *s_off = reinterpret_cast<int32_t *>(sp_smem);
  const int tot = a.np * a.W;
  for (int i = threadIdx.x; i < tot; i += blockDim.x) s_off[i] = a.poff[i];
  const int lane = threadIdx.x & 63;
  const int slice = __builtin_amdgcn_readfirstlane((int)(remap_block(blockIdx.x, gridDim.x, a.xcd_remap) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  const bool live = slice < a.nslices;
  const int sc = live ? slice : a.nslices - 1;
  const int64_t base = a.soff[sc];
  const int w = (int)((a.soff[sc + 1] - base) >> 6);
  const int64_t row = (int64_t)sc * 64 + lane;
  const bool valid = live && row < a.nrows;
  const int64_t rc = min(row, a.nrows - 1);
  const int len = valid ? a.rowlen[rc] : 0;
  const int pid = (int)a.rowpid[rc];
  const uint32_t base8 = 8u * (uint32_t)(a.rowbase ? a.rowbase[rc] : (int32_t)rc);
  const double *__restrict__ xg = a.x;
  const double *__restrict__ dinv = a.dinv;
  const double omega = a.omega;
  // row-wise epilogue operands (coalesced), requested before the stream; clamped addresses, no divergence
  double e0 = 0.0, e1 = 0.0, e2 = 0.0, dinv_row = 0.0, sp = 0.0;
  if (EPI == EPI_SUB) e0 = a.y[rc];
  else if (EPI == EPI_RESID) e0 = a.b[rc];
  else if (EPI == EPI_ADDTO) e0 = a.x2[rc];
  else if (EPI == EPI_SWEEP) {
    if (NT >= 2) {
      e0 = __builtin_nontemporal_load(a.b + rc);
      e1 = ONEG ? xg[rc] : __builtin_nontemporal_load(dinv + rc);
      if (XM != 1) { const double xl = __builtin_nontemporal_load(a.x2 + rc); e2 = a.x_zero ? 0.0 : xl; }
      if (ONEG) dinv_row = __builtin_nontemporal_load(dinv + rc);
    } else {
    e0 = a.b[rc];
    e1 = ONEG ? xg[rc] : dinv[rc];
    if (XM != 1) { const double xl = a.x2[rc]; e2 = a.x_zero ? 0.0 : xl; }
    if (ONEG) dinv_row = dinv[rc];
    }
    if (XM == 2) sp = a.s_out[rc];
  }
  __syncthreads();
  const double *vp = a.sval + base + lane;
  const int32_t *to = s_off + pid * a.W;
  double s = 0.0;
  int j = 0;
  for (; j + UN <= w; j += UN) {
    double v[UN], g[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = NT ? __builtin_nontemporal_load(vp + (int64_t)(j + u) * 64) : vp[(int64_t)(j + u) * 64];
#pragma unroll
    for (int u = 0; u < UN; ++u) {    // entries past the end of the row have offset 0: they gather the row's base column (valid)
      const uint32_t c8 = base8 + (uint32_t)((j + u < a.W) ? to[j + u] : 0);
      if (EPI == EPI_SWEEP && !ONEG) g[u] = omega * (ld_off(dinv, c8) * ld_off(xg, c8));
      else g[u] = ld_off(xg, c8);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const double pr = v[u] * g[u];
      s = (j + u < len) ? s + pr : s;   // masked: padding is never added
    }
  }
  for (; j < w; ++j) {
    const double v = vp[(int64_t)j * 64];
    const uint32_t c8 = base8 + (uint32_t)((j < a.W) ? to[j] : 0);
    double g;
    if (EPI == EPI_SWEEP && !ONEG) g = omega * (ld_off(dinv, c8) * ld_off(xg, c8));
    else g = ld_off(xg, c8);
    const double pr = v * g;
    s = (j < len) ? s + pr : s;
  }
  if (valid) {
    if (EPI == EPI_SET) a.y[row] = s;
    else if (EPI == EPI_SUB) a.y[row] = e0 - s;
    else if (EPI == EPI_RESID) a.y[row] = e0 - s;
    else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * s : s; a.y[row] = t; a.x2[row] = e0 + t; }
    else {
      const double dxi = ONEG ? e1 : omega * (e1 * e0);
      const double rn = e0 - s;
      const double xn = XM == 2 ? (e2 + sp) + dxi : e2 + dxi;
      if (NT >= 2) { if (XM != 1) __builtin_nontemporal_store(xn, a.x2 + row); __builtin_nontemporal_store(rn, a.y + row); }
      else { if (XM != 1) a.x2[row] = xn; a.y[row] = rn; }
      if (ONEG) a.s_out[row] = omega * (dinv_row * rn);
    }
  }
}

// sello_kernel with the next batch (values AND gathers: the offsets come from LDS, nothing depends on the stream) requested
// before the current batch is consumed -- see sell_pipe_kernel.  WMAX > 0: slices of width <= WMAX request their whole row.
template <int EPI, bool ONEG, int UN, int WMAX, bool NT>
__global__ __launch_bounds__(kBlock) void sello_pipe_kernel(SellOArgs a)
{
  extern __shared__ double sp_smem[];
  int32_t *s_off = reinterpret_cast<int32_t *>(sp_smem);
  const int tot = a.np * a.W;
  for (int i = threadIdx.x; i < tot; i += blockDim.x) s_off[i] = a.poff[i];
  const int lane = threadIdx.x & 63;
  const int slice = __builtin_amdgcn_readfirstlane((int)(remap_block(blockIdx.x, gridDim.x, a.xcd_remap) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  const bool live = slice < a.nslices;
  const int sc = live ? slice : a.nslices - 1;
  const int64_t base = a.soff[sc];
  const int w = (int)((a.soff[sc + 1] - base) >> 6);
  const int64_t row = (int64_t)sc * 64 + lane;
  const bool valid = live && row < a.nrows;
  const int64_t rc = min(row, a.nrows - 1);
  const int len = valid ? a.rowlen[rc] : 0;
  const int pid = (int)a.rowpid[rc];
  const uint32_t base8 = 8u * (uint32_t)(a.rowbase ? a.rowbase[rc] : (int32_t)rc);
  const double *__restrict__ xg = a.x;
  const double *__restrict__ dinv = a.dinv;
  const double omega = a.omega;
  const double *vp = a.sval + base + lane;
  const int wl = w - 1;
  // the value stream does not depend on the table: first batch (or the whole row) in flight while the table is staged
  constexpr int NV = WMAX > 0 ? WMAX : UN;
  double v0[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int64_t o = (int64_t)min(u, wl) * 64;
    v0[u] = NT ? __builtin_nontemporal_load(vp + o) : vp[o];
  }
  double e0 = 0.0, e1 = 0.0, e2 = 0.0, dinv_row = 0.0;
  if (EPI == EPI_SUB) e0 = a.y[rc];
  else if (EPI == EPI_RESID) e0 = a.b[rc];
  else if (EPI == EPI_ADDTO) e0 = a.x2[rc];
  else if (EPI == EPI_SWEEP) {
    e0 = a.b[rc];
    e1 = ONEG ? xg[rc] : dinv[rc];
    { const double xl = a.x2[rc]; e2 = a.x_zero ? 0.0 : xl; }
    if (ONEG) dinv_row = dinv[rc];
  }
  __syncthreads();
  const int32_t *to = s_off + pid * a.W;
  auto gat = [&](int j) -> double {
    const uint32_t c8 = base8 + (uint32_t)((j < a.W) ? to[j] : 0);
    if (EPI == EPI_SWEEP && !ONEG) return omega * (ld_off(dinv, c8) * ld_off(xg, c8));
    return ld_off(xg, c8);
  };
  double s = 0.0;
  if (WMAX > 0 && w <= WMAX) {
    double g[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) g[u] = gat(min(u, wl));
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const double pr = v0[u] * g[u];
      s = (u < len) ? s + pr : s;
    }
  } else {
    double v[UN], g[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) { v[u] = v0[u]; g[u] = gat(min(u, wl)); }
    int j = 0;
    for (; j + UN < w; j += UN) {                            // steady state: the next batch is requested unconditionally
      double vn[UN], gn[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int jj = min(j + UN + u, wl);
        vn[u] = NT ? __builtin_nontemporal_load(vp + (int64_t)jj * 64) : vp[(int64_t)jj * 64];
        gn[u] = gat(jj);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const double pr = v[u] * g[u];
        s = (j + u < len) ? s + pr : s;
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) { v[u] = vn[u]; g[u] = gn[u]; }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {                           // last batch
      const double pr = v[u] * g[u];
      s = (j + u < len) ? s + pr : s;
    }
  }
  if (valid) {
    if (EPI == EPI_SET) a.y[row] = s;
    else if (EPI == EPI_SUB) a.y[row] = e0 - s;
    else if (EPI == EPI_RESID) a.y[row] = e0 - s;
    else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * s : s; a.y[row] = t; a.x2[row] = e0 + t; }
    else {
      const double dxi = ONEG ? e1 : omega * (e1 * e0);
      a.x2[row] = e2 + dxi;
      const double rn = e0 - s;
      a.y[row] = rn;
      if (ONEG) a.s_out[row] = omega * (dinv_row * rn);
    }
  }
}

// ---------------------------------------------------------------------------
// Compressed SELL-64 ("SELL-C"): the same lane-per-row kernel fed by a losslessly
// compressed matrix stream.  Two independent, automatically detected compressions:
//   IDX16 : per slice column (64 entries) a base column + 16-bit offsets, when the 64
//           columns span < 65536 (true for banded / stencil matrices); decided per
//           slice, slices that do not fit keep their 32-bit columns.     4 -> 2 B/nnz
//   VDICT : when the whole matrix holds <= 256 distinct fp64 values (constant-
//           coefficient operators on uniform meshes) an 8-bit code into a dictionary
//           kept in LDS.                                                   8 -> 1 B/nnz
// Both are exact: the decoded (col,val) pairs are bit-identical to the CSR input, and
// rows are still summed left to right.  Packed layout: entry (j,lane) of a slice sits at
// poff + (j/4)*256 + lane*4 + (j%4), so one lane fetches four consecutive entries of
// its row with a single 4-byte (codes) / 8-byte (offsets) load and a wave reads 256 /
// 512 contiguous bytes.
// ---------------------------------------------------------------------------
struct SellCArgs {
  const int64_t *soff;      // column-major offsets (scol / sval), width w
  const int64_t *poff;      // packed offsets (pidx / pcode), width rounded up to 4
  const int32_t *scol;
  const double *sval;
  const uint16_t *pidx;
  const uint8_t *pcode;
  const int32_t *pbase;     // [poff/64 + j] base column of slice column j
  const uint8_t *smode;     // per slice: bit0 = 16-bit offsets valid
  const double *dict;       // [256]
  const int32_t *rowlen;
  int64_t nrows;
  int nslices;
  int x_zero;
  int xcd_remap;
  const double *x;
  const double *dinv;
  double omega;
  double *y;
  const double *b;
  double *x2;
  double *s_out;
};

template <int EPI, bool ONEG, bool VDICT, bool NT>
__global__ __launch_bounds__(kBlock) void sellc_kernel(SellCArgs a)
{
  __shared__ double sdict[VDICT ? 256 : 1];
  if (VDICT) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) sdict[i] = a.dict[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  // wave-uniform by construction; readfirstlane makes it provably so (scalar loads, SGPR loop bounds)
  const int slice = __builtin_amdgcn_readfirstlane((int)(remap_block(blockIdx.x, gridDim.x, a.xcd_remap) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  if (slice >= a.nslices) return;
  const int64_t base = a.soff[slice];
  const int w = (int)((a.soff[slice + 1] - base) >> 6);
  const int64_t pb = a.poff[slice];
  const int npack = (w + 3) >> 2;
  const bool i16 = (a.smode[slice] & 1) != 0;
  const int64_t row = (int64_t)slice * 64 + lane;
  const bool valid = row < a.nrows;
  const int len = valid ? a.rowlen[row] : 0;
  const double *__restrict__ xg = a.x;
  const double *__restrict__ dinv = a.dinv;
  const double omega = a.omega;
  double e0 = 0.0, e1 = 0.0, e2 = 0.0, dinv_row = 0.0;
  if (valid) {
    if (EPI == EPI_SUB) e0 = a.y[row];
    else if (EPI == EPI_RESID) e0 = a.b[row];
    else if (EPI == EPI_ADDTO) e0 = a.x2[row];
    else if (EPI == EPI_SWEEP) {
      e0 = a.b[row];
      e1 = ONEG ? xg[row] : dinv[row];
      e2 = a.x_zero ? 0.0 : a.x2[row];
      if (ONEG) dinv_row = dinv[row];
    }
  }
  // base column of slice column j lives in lane j (w <= 64 is the common case; wider slices reload)
  const int32_t *pbp = a.pbase + (pb >> 6);
  int32_t mybase = (i16 && lane < 4 * npack) ? pbp[lane] : 0;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 *ip = reinterpret_cast<const u32x2 *>(a.pidx + pb) + lane;      // 4 x uint16 per lane per pack
  const uint32_t *kp = reinterpret_cast<const uint32_t *>(a.pcode + pb) + lane; // 4 x uint8  per lane per pack
  const int32_t *cp = a.scol + base + lane;
  const double *vp = a.sval + base + lane;
  double s = 0.0;
  // The compressed stream is so small (12 B per 4 entries) that a lane can afford to request
  // the packs of its WHOLE row at once (PB packs = 32 entries per round, 3 VGPRs per pack); the
  // gathers then run in sub-batches of 8 behind a single HBM latency instead of one per batch.
  constexpr int PB = 8;   // packs requested per round
  constexpr int PK = 2;   // packs per gather batch (8 gathers in flight)
  for (int r0 = 0; r0 < npack; r0 += PB) {
    u32x2 iw[PB];
    uint32_t kw[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) {
      const int pk = (r0 + q < npack) ? r0 + q : npack - 1;   // clamp (masked below)
      if (i16) iw[q] = NT ? __builtin_nontemporal_load(ip + (int64_t)pk * 64) : ip[(int64_t)pk * 64];
      if (VDICT) kw[q] = NT ? __builtin_nontemporal_load(kp + (int64_t)pk * 64) : kp[(int64_t)pk * 64];
    }
#pragma unroll
    for (int b0 = 0; b0 < PB; b0 += PK) {
      if (r0 + b0 < npack) {   // wave-uniform
        int32_t c[4 * PK];
        double v[4 * PK], g[4 * PK];
#pragma unroll
        for (int q = 0; q < PK; ++q) {
          const int pk = (r0 + b0 + q < npack) ? r0 + b0 + q : npack - 1;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = 4 * pk + u;
            const int jc = j < w ? j : w - 1;
            if (i16) {
              const uint32_t word = (u < 2) ? iw[b0 + q].x : iw[b0 + q].y;
              const int32_t off = (int32_t)((u & 1) ? (word >> 16) : (word & 0xffffu));
              int32_t bj;
              if (j < 64) bj = __builtin_amdgcn_readlane(mybase, j);
              else bj = pbp[j];
              c[4 * q + u] = bj + off;
            } else {
              c[4 * q + u] = NT ? __builtin_nontemporal_load(cp + (int64_t)jc * 64) : cp[(int64_t)jc * 64];
            }
            if (VDICT) v[4 * q + u] = sdict[(kw[b0 + q] >> (8 * u)) & 0xffu];
            else v[4 * q + u] = NT ? __builtin_nontemporal_load(vp + (int64_t)jc * 64) : vp[(int64_t)jc * 64];
          }
        }
#pragma unroll
        for (int e = 0; e < 4 * PK; ++e) {
          if (EPI == EPI_SWEEP && !ONEG) g[e] = omega * (dinv[c[e]] * xg[c[e]]);
          else g[e] = xg[c[e]];
        }
#pragma unroll
        for (int q = 0; q < PK; ++q) {
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = 4 * (r0 + b0 + q) + u;
            const double pr = v[4 * q + u] * g[4 * q + u];
            s = (r0 + b0 + q < npack && j < len) ? s + pr : s;
          }
        }
      }
    }
  }
  if (valid) {
    if (EPI == EPI_SET) a.y[row] = s;
    else if (EPI == EPI_SUB) a.y[row] = e0 - s;
    else if (EPI == EPI_RESID) a.y[row] = e0 - s;
    else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * s : s; a.y[row] = t; a.x2[row] = e0 + t; }
    else {
      const double dxi = ONEG ? e1 : omega * (e1 * e0);
      a.x2[row] = e2 + dxi;
      const double rn = e0 - s;
      a.y[row] = rn;
      if (ONEG) a.s_out[row] = omega * (dinv_row * rn);
    }
  }
}

// ---------------------------------------------------------------------------
// Row-pattern SELL ("SELL-P"): operators assembled on structured grids with constant
// coefficients hold only a handful of DISTINCT ROWS once a row is written as
// (column offsets relative to a base, values): a 27-point Q1 stiffness matrix with
// Dirichlet rows eliminated has 27 such patterns, its prolongation ~64.  The setup
// detects them exactly (bitwise equal values, equal offsets) and the matrix stream
// collapses to a 16-bit pattern id per row (+ a 32-bit base column when offsets are
// not relative to the row index): 324 B/row -> 2 B/row for the 27-point operator.
// The pattern table (offsets + fp64 values, a few KB) lives in LDS; lane = row as in
// SELL-64, rows are summed left to right in CSR order (bit-identical to the oracle),
// lanes of a wave mostly share a pattern, so the table reads are LDS broadcasts and the
// gather x[base+off] is contiguous across lanes.  What is left is the vector traffic.
// ---------------------------------------------------------------------------
struct SellPArgs {
  const uint16_t *rowpid;   // [nrows] pattern of each row
  const int32_t *rowbase;   // [nrows] base column, or nullptr: offsets are relative to the row index
  const int32_t *plen;      // [np]  (the last pattern is an empty row, used by the lanes past nrows)
  const int32_t *poff;      // [np*W] BYTE offsets (8*column offset), zero padded
  const double *pval;       // [np*W] zero padded
  int np, W;                // W = row stride of the table, a multiple of the kernel's UN
  int64_t nrows;
  int nslices;
  int x_zero;
  int xcd_remap;
  const double *x;
  const double *dinv;
  double omega;
  double *y;
  const double *b;
  double *x2;
  double *s_out;
};

// GT = true: the pattern table is too big for LDS (many patterns x wide rows, e.g. Q2 transfer operators) and is read
// from global memory instead -- it is a few hundred KB and stays in L2; same arithmetic, same order.
template <int EPI, bool ONEG, int UN, bool GT = false>
__global__ __launch_bounds__(kBlock) void sellp_kernel(SellPArgs a)
{
  extern __shared__ double sp_smem[];
  const int tot = a.np * a.W;
  const double *s_val = GT ? a.pval : sp_smem;
  const int32_t *s_off = GT ? a.poff : reinterpret_cast<const int32_t *>(sp_smem + tot);
  const int32_t *s_len = GT ? a.plen : s_off + tot;
  if (!GT) {
    double *w_val = sp_smem;
    int32_t *w_off = reinterpret_cast<int32_t *>(sp_smem + tot);
    int32_t *w_len = w_off + tot;
    for (int i = threadIdx.x; i < tot; i += blockDim.x) { w_val[i] = a.pval[i]; w_off[i] = a.poff[i]; }
    for (int i = threadIdx.x; i < a.np; i += blockDim.x) w_len[i] = a.plen[i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6;
  // every workgroup owns a contiguous chunk of slices (its waves interleave inside it); with
  // xcd_remap the chunks of one XCD are contiguous too, so each L2 sees 1/8 of the gathered vector
  const int nwg = gridDim.x;
  const int blk = remap_block(blockIdx.x, nwg, a.xcd_remap);
  // balanced chunks: nslices / nwg slices each, the remainder one more for the first workgroups (with ceil-sized chunks every
  // workgroup of the 128^3 level had 17 slices for its 4 waves -- a fifth, nearly empty round for the whole chip)
  const int chunk_lo = a.nslices / nwg, chunk_rem = a.nslices % nwg;
  const int s_begin = blk * chunk_lo + min(blk, chunk_rem);
  const int s_end = s_begin + chunk_lo + (blk < chunk_rem ? 1 : 0);
  const double *__restrict__ xg = a.x;
  const double *__restrict__ dinv = a.dinv;
  const double omega = a.omega;
  const int W = a.W;
  for (int slice = s_begin + wave; slice < s_end; slice += wpb) {
    const int64_t row = (int64_t)slice * 64 + lane;
    const bool valid = row < a.nrows;
    const int pid = valid ? (int)a.rowpid[row] : a.np - 1;    // the last pattern is empty: lanes past the end gather x[0] * 0
    const int len = s_len[pid];
    const uint32_t base8 = valid ? 8u * (uint32_t)(a.rowbase ? a.rowbase[row] : (int32_t)row) : 0u;
    double e0 = 0.0, e1 = 0.0, e2 = 0.0, dinv_row = 0.0;
    if (valid) {
      if (EPI == EPI_SUB) e0 = a.y[row];
      else if (EPI == EPI_RESID) e0 = a.b[row];
      else if (EPI == EPI_ADDTO) e0 = a.x2[row];
      else if (EPI == EPI_SWEEP) {
        e0 = a.b[row];
        e1 = ONEG ? xg[row] : dinv[row];
        e2 = a.x_zero ? 0.0 : a.x2[row];
        if (ONEG) dinv_row = dinv[row];
      }
    }
    int lmax = len;                                        // longest row of the wave
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, __shfl_xor(lmax, o));
    lmax = __builtin_amdgcn_readfirstlane(lmax);
    const double *tv = s_val + pid * W;
    const int32_t *to = s_off + pid * W;
    double s = 0.0;
    for (int j = 0; j < lmax; j += UN) {                   // j + UN <= W: the table rows are padded to a multiple of UN
      double v[UN], g[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        // entries past the end of a row are (offset 0, value 0.0): they gather the row's base column
        // (valid) and multiply it by zero; clearing the high word of the gathered value for those
        // lanes keeps the product an exact zero even if that entry of x is Inf/NaN.
        v[u] = tv[j + u];
        const uint32_t c8 = base8 + (uint32_t)to[j + u];
        if (EPI == EPI_SWEEP && !ONEG) g[u] = omega * (ld_off(dinv, c8) * ld_off(xg, c8));
        else g[u] = ld_off(xg, c8);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        double gu = g[u];
        if (!(j + u < len)) gu = __hiloint2double(0, __double2loint(gu));
        s = s + v[u] * gu;                                 // padded entries add an exact +0.0: s is unchanged
      }
    }
    if (valid) {
      if (EPI == EPI_SET) { a.y[row] = s; if (a.s_out) a.s_out[row] = omega * (dinv[row] * s); }      // optional s emission, see sells_kernel
      else if (EPI == EPI_SUB) { const double yn = e0 - s; a.y[row] = yn; if (a.s_out) a.s_out[row] = omega * (dinv[row] * yn); }
      else if (EPI == EPI_RESID) a.y[row] = e0 - s;
      else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * s : s; a.y[row] = t; a.x2[row] = e0 + t; }
      else {
        const double dxi = ONEG ? e1 : omega * (e1 * e0);
        a.x2[row] = e2 + dxi;
        const double rn = e0 - s;
        a.y[row] = rn;
        if (ONEG) a.s_out[row] = omega * (dinv_row * rn);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Prolongation + correction  dxh = P dxH ; xh += dxh  (GMGLinearSolvers.jl:491,494) of a big level with TWO rows per lane.  The
// one-row kernel is a chain of three memory latencies per 64-row slice (pattern id / base / x -> gathers -> stores, and the next
// slice's loads wait for the stores: one vmcnt) with four slices per wave at 128^3: 24 us for 62 MB.  Here a lane owns rows 2l and
// 2l+1 of a 128-row slice: half the chains, twice the gathers in flight per chain, 16-byte loads / stores of x and dx, the two
// pattern ids in one 4-byte load and the two base columns in one 8-byte load.  Rows are summed left to right as before: bit-identical.
//   a.nslices = ceil(nrows / 128) ; a.rowbase != nullptr ; table in LDS (not the GT form)
// ---------------------------------------------------------------------------
typedef double gmg_pd2 __attribute__((ext_vector_type(2)));
template <int UN>
__global__ __launch_bounds__(kBlock) void sellp_pair_addto_kernel(SellPArgs a)
{
  extern __shared__ double sp_smem[];
  const int tot = a.np * a.W;
  {
    double *w_val = sp_smem;
    int32_t *w_off = reinterpret_cast<int32_t *>(sp_smem + tot);
    int32_t *w_len = w_off + tot;
    for (int i = threadIdx.x; i < tot; i += blockDim.x) { w_val[i] = a.pval[i]; w_off[i] = a.poff[i]; }
    for (int i = threadIdx.x; i < a.np; i += blockDim.x) w_len[i] = a.plen[i];
    __syncthreads();
  }
  const double *s_val = sp_smem;
  const int32_t *s_off = reinterpret_cast<const int32_t *>(sp_smem + tot);
  const int32_t *s_len = s_off + tot;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x;
  const int blk = remap_block(blockIdx.x, nwg, a.xcd_remap);
  const int chunk_lo = a.nslices / nwg, chunk_rem = a.nslices % nwg;
  const int s_begin = blk * chunk_lo + min(blk, chunk_rem);
  const int s_end = s_begin + chunk_lo + (blk < chunk_rem ? 1 : 0);
  const double *__restrict__ xg = a.x;
  const int W = a.W;
  const double omega = a.omega;
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  for (int slice = s_begin + wave; slice < s_end; slice += wpb) {
    const int64_t row = (int64_t)slice * 128 + 2 * lane;
    const bool inner = (int64_t)slice * 128 + 128 <= a.nrows;   // wave-uniform: every lane owns two rows
    const bool vA = row < a.nrows, vB = row + 1 < a.nrows;
    int pidA = a.np - 1, pidB = a.np - 1;                  // the last pattern is empty
    uint32_t baseA = 0u, baseB = 0u;
    gmg_pd2 e0 = gmg_pd2{0.0, 0.0};
    if (inner) {
      const uint32_t pp = *reinterpret_cast<const uint32_t *>(a.rowpid + row);
      pidA = (int)(pp & 0xffffu); pidB = (int)(pp >> 16);
      const int2 bb = *reinterpret_cast<const int2 *>(a.rowbase + row);
      baseA = 8u * (uint32_t)bb.x; baseB = 8u * (uint32_t)bb.y;
      e0 = *reinterpret_cast<const d2u *>(a.x2 + row);
    } else {
      if (vA) { pidA = (int)a.rowpid[row]; baseA = 8u * (uint32_t)a.rowbase[row]; e0.x = a.x2[row]; }
      if (vB) { pidB = (int)a.rowpid[row + 1]; baseB = 8u * (uint32_t)a.rowbase[row + 1]; e0.y = a.x2[row + 1]; }
    }
    const int lenA = s_len[pidA], lenB = s_len[pidB];
    int lmax = max(lenA, lenB);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = max(lmax, __shfl_xor(lmax, o));
    lmax = __builtin_amdgcn_readfirstlane(lmax);
    const double *tvA = s_val + pidA * W, *tvB = s_val + pidB * W;
    const int32_t *toA = s_off + pidA * W, *toB = s_off + pidB * W;
    double sA = 0.0, sB = 0.0;
    for (int j = 0; j < lmax; j += UN) {
      double cA[UN], cB[UN], gA[UN], gB[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        cA[u] = tvA[j + u]; cB[u] = tvB[j + u];
        gA[u] = ld_off(xg, baseA + (uint32_t)toA[j + u]);
        gB[u] = ld_off(xg, baseB + (uint32_t)toB[j + u]);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {                       // entries past the end of a row: (offset 0, value 0.0), high word cleared -- see sellp_kernel
        double ga = gA[u], gb = gB[u];
        if (!(j + u < lenA)) ga = __hiloint2double(0, __double2loint(ga));
        if (!(j + u < lenB)) gb = __hiloint2double(0, __double2loint(gb));
        sA = sA + cA[u] * ga;
        sB = sB + cB[u] * gb;
      }
    }
    const gmg_pd2 t = gmg_pd2{omega != 0.0 ? omega * sA : sA, omega != 0.0 ? omega * sB : sB};
    const gmg_pd2 xn = gmg_pd2{e0.x + t.x, e0.y + t.y};
    if (inner) {
      *reinterpret_cast<d2u *>(a.y + row) = t;
      *reinterpret_cast<d2u *>(a.x2 + row) = xn;
    } else {
      if (vA) { a.y[row] = t.x; a.x2[row] = xn.x; }
      if (vB) { a.y[row + 1] = t.y; a.x2[row + 1] = xn.y; }
    }
  }
}

// ---------------------------------------------------------------------------
// SELL-P, shared-offset form ("stencil mode"): when the offsets are relative to the row index, every
// row pattern is a subset of the union U of all offsets, so ALL rows can share one offset list and
// differ only in their coefficients (absent entries become explicit zeros, which add an exact +0.0;
// a per-entry mask clears the high word of the gathered value so that the product stays an exact zero
// even for a non-finite x).  U is covered by runs of 3 consecutive offsets (o,o+1,o+2).  Per run a
// lane loads x[row+o] once; x[row+o+1] and x[row+o+2] are its right neighbours' values, fetched with
// DPP wave shifts (a slice is 62 rows: lanes 62/63 only carry the halo).  A 27-point operator needs 9+9
// load instructions per 64 rows instead of 27, which is what bounded the generic pattern kernel
// (L1 line traffic of the overlapping gathers).  Rows are still summed in ascending column order.
// ---------------------------------------------------------------------------
struct alignas(16) PatEntry {
  double v;
  uint32_t m;      // 0xffffffff: entry present ; 0: padding
  uint32_t pad;
};

struct SellSArgs {
  const uint16_t *rowpid;
  const PatEntry *tab;      // [np * nu], nu = K * nruns ; the last pattern is empty
  const double *tab8;       // the coefficients alone (unmasked sweep kernels)
  const uint8_t *codes;     // coded form (VD): [np * nu] index into dict, 255 = entry absent
  const uint16_t *codes16;  // walk form of a table with more than 255 distinct values (sellw_zwalk_kernel only): 16-bit codes, 65535 = absent; dict then holds 65536 doubles
  const double *dict;       // coded form: [256] distinct values, dict[255] = 0.0
  const uint32_t *runmask;  // coded form: [np] bit r set when the pattern has an entry in run r (nruns <= 32)
  const int32_t *run_off;   // [nruns] first offset of each run (elements)
  int np, nruns;            // nruns is a multiple of the kernel's RB
  int minoff, maxoff;       // smallest / largest first offset of a run
  int64_t nrows, ncols;
  int nslices;
  int x_zero;
  int xmode;                // sweep: 0 x += s_k ; 1 x untouched (deferred) ; 2 x = (x + s_{k-1}) + s_k, s_{k-1} read from s_out
  int xcd_remap;
  const double *pdinv;      // [np] 1/diag of each pattern (Jacobi inverse diagonal without its 8 B/row stream), or nullptr
  const double *x;
  const double *dinv;
  double omega;
  double *y;
  const double *b;
  double *x2;
  double *s_out;
  // wide rows, per-workgroup value tables (sells_kernel<..., WL = true>): the patterns chunk c of the launch geometry uses
  const uint16_t *wl_pids;  // [nwg * wl_stride] pattern ids, ascending
  const int32_t *wl_cnt;    // [nwg]
  int wl_stride, wl_max;    // list stride; largest list of the launch (sizes the LDS table)
};

// lane i <- lane i+1 ; lane 63 <- 0 (bound_ctrl: no separate initialisation of the destination)
__device__ __forceinline__ double wave_shl1(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast_lane(double v, int l)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// A slice is kSellsRows = 62 rows: the 64 lanes load x[row0+o .. row0+o+63], lanes 0..61 own a row and find their
// two right neighbours inside the wave, lanes 62/63 only carry the halo (no separate tail loads: 9 gather
// instructions per slice for a 27-point operator).
// K = offsets per run (3: 27-point operators; 5: Q2-type operators whose lines hold 5 consecutive offsets) and the slice is
// 65-K rows.  VD = coded table: one byte per entry indexing a dictionary of <= 255 distinct values (wide rows / many
// patterns would not fit LDS otherwise: Q2 stiffness = 217 patterns x 125 entries = 27 KB of codes).
constexpr int kSellsRows = 62;
// DBG (tools/mb_sells.hip only; the library instantiates DBG = 0): timing ablations that drop one ingredient each --
// bit0 no high-word mask, bit1 coefficients not read from LDS, bit2 no DPP shifts, bit3 no gathers.  Wrong results by design.
// WL (coded tables only): wide rows -- Q2 stiffness and additive-Schwarz operators, 125 entries per row -- spend their time on
// instruction issue, not on bytes: per entry the coded form pays a code read, a dictionary read, a compare + select for the mask
// and the address arithmetic of both (8 vector instructions + 2 LDS reads).  A workgroup's chunk of consecutive rows uses only a
// few of the patterns (6 x <= 4 x <= 2 of the 217 of a Q2 operator), so each workgroup decodes THOSE into a plain value table of
// its own (list prepared once per launch geometry: sellw_chunk_patterns_kernel) and an entry costs one LDS read, a multiply and
// an add.  Absent entries hold 0.0 and are NOT masked: with finite gathered values the products are exact zeros, which leave every
// partial sum unchanged (a sum that starts at +0.0 never becomes -0.0) -- bit-identical to the masked form; a batch of runs whose
// gathered values are not all finite takes the masked path with the codes read from global memory, so a non-finite entry of x
// still reaches exactly the rows that store a coefficient for it.
template <int EPI, bool ONEG, int RB, int K = 3, bool VD = false, int DBG = 0, bool WL = false>
__global__ __launch_bounds__(kBlock) void sells_kernel(SellSArgs a)
{
  static_assert(!WL || VD, "per-workgroup value tables are built from the coded form");
  constexpr int ROWS = 65 - K;
  extern __shared__ double sp_smem[];
  const int nu = K * a.nruns;
  const int tot = a.np * nu;
  // plain: [np*nu] coefficients | [np*nu] high-word masks | [np] 1/diag ; coded: [256] dictionary | [np*nu] codes | [np] 1/diag
  double *s_val = sp_smem;
  uint32_t *s_msk = reinterpret_cast<uint32_t *>(sp_smem + tot);
  const uint8_t *s_code = reinterpret_cast<const uint8_t *>(sp_smem + 256);
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int nwg = gridDim.x;
  const int blk = remap_block(blockIdx.x, nwg, a.xcd_remap);
  // balanced chunks: nslices / nwg slices each, the remainder one more for the first workgroups (with ceil-sized chunks every
  // workgroup of the 128^3 level had 17 slices for its 4 waves -- a fifth, nearly empty round for the whole chip)
  const int chunk_lo = a.nslices / nwg, chunk_rem = a.nslices % nwg;
  const int s_begin = blk * chunk_lo + min(blk, chunk_rem);
  const int s_end = s_begin + chunk_lo + (blk < chunk_rem ? 1 : 0);
  const double *__restrict__ xg = a.x;
  const double *__restrict__ dinv = a.dinv;
  const double omega = a.omega;
  const int last = (int)a.ncols - 1;

  // row-wise operands of a slice ("head"): requested at the top of the slice's iteration with CLAMPED addresses and no
  // divergent or conditional control flow around the loads (rows past the end / the two halo lanes re-read a valid row and
  // discard it).  Conditional loads made the compiler wait at every join, and a prefetch across iterations cannot overlap
  // anyway: on gfx9 loads and stores share vmcnt and return out of order with respect to each other, so a wave with stores in
  // flight waits for a load with vmcnt(0) (see sells_sweep_kernel and profiles/r02_tuning.md).
  int pid_n = a.np - 1;
  double e0_n = 0.0, e1_n = 0.0, e2_n = 0.0, dr_n = 0.0, sp_n = 0.0;
  const int xmode = a.xmode;
  const bool tab_dinv = a.pdinv != nullptr;
  const int64_t lastrow = a.nrows - 1;
  const bool xzero = a.x_zero != 0;
  auto load_head = [&](int slice) {
    const int64_t row = min((int64_t)slice * ROWS + lane, lastrow);
    if (!VD) pid_n = (int)a.rowpid[row];                   // (coded form: the pattern id is requested one slice ahead, below)
    if (EPI == EPI_SUB) e0_n = a.y[row];
    else if (EPI == EPI_RESID) e0_n = a.b[row];
    else if (EPI == EPI_ADDTO) e0_n = a.x2[row];
    else if (EPI == EPI_SWEEP) {
      e0_n = a.b[row];
      e1_n = xg[row];
      if (xmode != 1) { const double xl = a.x2[row]; e2_n = xzero ? 0.0 : xl; }
      if (xmode == 2) sp_n = a.s_out[row];                 // s_{k-1}, about to be overwritten by s_{k+1}
      if (!tab_dinv) dr_n = dinv[row];
    }
  };
  // WL: [wl_max * (nu + K)] values, the last run of every pattern all zeros (the multiply code is branch-free: a batch's missing
  // runs use it) | [wl_max] 1/diag | [wl_max] run masks | [np] pattern id -> local id
  double *s_dinv = WL ? sp_smem + (size_t)a.wl_max * (nu + K) : VD ? sp_smem + 256 + ((tot + 7) >> 3) : reinterpret_cast<double *>(s_msk + tot + (tot & 1));   // [np]
  uint32_t *s_rmask = reinterpret_cast<uint32_t *>(s_dinv + (WL ? a.wl_max : a.np));                               // [np] (coded form)
  uint8_t *s_map = reinterpret_cast<uint8_t *>(s_rmask + (WL ? a.wl_max : 0));
  if (WL) {
    const int lnp = a.wl_cnt[blk];
    const uint16_t *mine = a.wl_pids + (size_t)blk * a.wl_stride;
    for (int i = threadIdx.x; i < lnp; i += blockDim.x) {
      const int p = mine[i];
      s_map[p] = (uint8_t)i;
      s_rmask[i] = a.runmask[p];
      if (EPI == EPI_SWEEP && tab_dinv) s_dinv[i] = a.pdinv[p];
    }
    for (int i = threadIdx.x; i < lnp * (nu + K); i += blockDim.x) {
      const int l = i / (nu + K), e = i - l * (nu + K);
      sp_smem[i] = e < nu ? a.dict[a.codes[(size_t)mine[l] * nu + e]] : 0.0;
    }
  } else if (VD) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) sp_smem[i] = a.dict[i];
    uint8_t *wc = reinterpret_cast<uint8_t *>(sp_smem + 256);
    for (int i = threadIdx.x; i < tot; i += blockDim.x) wc[i] = a.codes[i];
    for (int i = threadIdx.x; i < a.np; i += blockDim.x) s_rmask[i] = a.runmask[i];
  } else
  for (int i = threadIdx.x; i < tot; i += blockDim.x) { s_val[i] = a.tab[i].v; s_msk[i] = a.tab[i].m; }
  if (!WL && EPI == EPI_SWEEP && tab_dinv)
    for (int i = threadIdx.x; i < a.np; i += blockDim.x) s_dinv[i] = a.pdinv[i];
  __syncthreads();

  if constexpr (WL) {
    // Batches of WB runs flow through a two-deep pipeline that runs ACROSS the slices of the wave: while batch k is multiplied the
    // gathers of batch k + 1 are in flight, and when batch k is the last of its slice, batch k + 1 is the first of the next one
    // (set up -- pattern ids -> local ids -> wave-OR of the run masks, row-wise operands requested -- before batch k is waited
    // for).  Every batch issues exactly WB gathers (missing runs re-read run 0 and meet the zero run of the table) and each path has
    // its own copy of the multiply code, so the waits are static vmcnt(#loads issued since).  Measured alternatives
    // (profiles/r03_tuning.md section 8): 10 runs per batch (twice the registers: two waves per SIMD instead of four) and a
    // three-slice pipeline with a register pair per run (two waves per SIMD) are both slower.
    constexpr int WB = 5;
    struct Ctx { int row0, gpid, lid; double e0, e1, e2, sp, dr; };
    auto epilogue = [&](const Ctx &c, double sum) {
      const int64_t row = (int64_t)c.row0 + lane;
      if (!(lane < ROWS && row < a.nrows)) return;
      if (EPI == EPI_SET) { a.y[row] = sum; if (a.s_out) a.s_out[row] = omega * (dinv[row] * sum); }
      else if (EPI == EPI_SUB) { const double yn = c.e0 - sum; a.y[row] = yn; if (a.s_out) a.s_out[row] = omega * (dinv[row] * yn); }
      else if (EPI == EPI_RESID) a.y[row] = c.e0 - sum;
      else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * sum : sum; a.y[row] = t; a.x2[row] = c.e0 + t; }
      else {
        if (xmode == 0) a.x2[row] = c.e2 + c.e1;
        else if (xmode == 2) a.x2[row] = (c.e2 + c.sp) + c.e1;
        const double rn = c.e0 - sum;
        a.y[row] = rn;
        a.s_out[row] = omega * (((EPI == EPI_SWEEP && tab_dinv) ? s_dinv[c.lid] : c.dr) * rn);
      }
    };
    int is = s_begin + wave;                                 // slice on the issue side
    if (is >= s_end) return;
    int pid_next = (int)a.rowpid[min((int64_t)is * ROWS + lane, lastrow)];
    // run offsets: lane r of a register holds run_off[r] (nruns <= 32), read with v_readlane -- a scalar load per run put a
    // constant-cache round trip in front of every batch of gathers
    const int voff = a.run_off[min(lane, a.nruns - 1)];
    uint32_t iM = 0;
    Ctx nc;
    auto setup = [&]() {                                     // slice `is`
      const int64_t row = (int64_t)is * ROWS + lane;
      nc.row0 = is * ROWS;
      nc.gpid = (lane < ROWS && row <= lastrow) ? pid_next : a.np - 1;
      nc.lid = (int)s_map[nc.gpid];
      uint32_t m = s_rmask[nc.lid];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m |= (uint32_t)__shfl_xor((int)m, o);
      iM = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
      const int64_t r = min(row, lastrow);
      nc.e0 = nc.e1 = nc.e2 = nc.sp = nc.dr = 0.0;
      if (EPI == EPI_SUB) nc.e0 = a.y[r];
      else if (EPI == EPI_RESID) nc.e0 = a.b[r];
      else if (EPI == EPI_ADDTO) nc.e0 = a.x2[r];
      else if (EPI == EPI_SWEEP) {
        nc.e0 = a.b[r];
        nc.e1 = xg[r];
        if (xmode != 1) { const double xl = a.x2[r]; nc.e2 = xzero ? 0.0 : xl; }
        if (xmode == 2) nc.sp = a.s_out[r];
        if (!tab_dinv) nc.dr = dinv[r];
      }
      pid_next = (int)a.rowpid[min((int64_t)min(is + wpb, s_end - 1) * ROWS + lane, lastrow)];
    };
    auto take = [&](int (&rr)[WB]) {
#pragma unroll
      for (int q = 0; q < WB; ++q) { rr[q] = iM ? (int)__builtin_ctz(iM) : -1; iM &= iM - 1; }
      return iM == 0;
    };
    auto issue = [&](const int (&rr)[WB], double (&G)[WB]) {
      const int row = nc.row0 + lane;
#pragma unroll
      for (int q = 0; q < WB; ++q) {
        const int o = __builtin_amdgcn_readlane(voff, __builtin_amdgcn_readfirstlane(max(rr[q], 0)));
        if (DBG & 8) G[q] = (double)(o + lane);              // timing ablation: no gathers (wrong results)
        else
        G[q] = ld_off(xg, 8u * (uint32_t)min(max(row + o, 0), last));
      }
    };
    double sum = 0.0;
    auto consume = [&](const int (&rr)[WB], const double (&G)[WB], const Ctx &c) {
      const double *tv = sp_smem + c.lid * (nu + K);
      bool fin = true;
#pragma unroll
      for (int q = 0; q < WB; ++q) fin = fin && __builtin_isfinite(G[q]);
      if (__all(fin)) {
        // five runs at a time: their 5 x K coefficients first, no branch in between (a branch per run made the compiler wait for each
        // run's LDS reads right where it issued them); missing runs take the zero run -- exact zero products, the sum does not move
#pragma unroll
        for (int h = 0; h < WB; h += 5) {
          if (rr[h] < 0) break;                              // uniform: the batch's runs are packed to the front
          double cf[5][K];
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            const double *tr = tv + (rr[h + q] >= 0 ? rr[h + q] : a.nruns) * K;
#pragma unroll
            for (int t = 0; t < K; ++t) cf[q][t] = tr[t];
          }
#pragma unroll
          for (int q = 0; q < 5; ++q) {
            double cur = G[h + q];
#pragma unroll
            for (int t = 0; t < K; ++t) {
              if (t > 0) cur = wave_shl1(cur);
              sum = sum + cf[q][t] * cur;
            }
          }
        }
      } else {                                               // a non-finite value in reach: masked products, codes from global memory
        const uint8_t *gc = a.codes + (size_t)c.gpid * nu;
#pragma unroll 1
        for (int q = 0; q < WB; ++q) {
          if (rr[q] < 0) continue;
          double cur = G[q];
#pragma unroll 1
          for (int t = 0; t < K; ++t) {
            if (t > 0) cur = wave_shl1(cur);
            const int code = gc[rr[q] * K + t];
            const double gv = __hiloint2double(__double2hiint(cur) & ((code == 255) ? 0 : -1), __double2loint(cur));
            sum = sum + tv[rr[q] * K + t] * gv;
          }
        }
      }
    };
    int r0[WB], r1[WB];
    double G0[WB], G1[WB];
    bool l0, l1;                                             // the batch is the last of its slice
    setup();
    Ctx cc = nc;                                             // slice on the multiply side
    l0 = take(r0);
    issue(r0, G0);
    for (;;) {
      if (!l0) { l1 = take(r1); issue(r1, G1); consume(r0, G0, cc); }
      else {
        is += wpb;
        if (is < s_end) { setup(); l1 = take(r1); issue(r1, G1); consume(r0, G0, cc); epilogue(cc, sum); sum = 0.0; cc = nc; }
        else { consume(r0, G0, cc); epilogue(cc, sum); break; }
      }
      if (!l1) { l0 = take(r0); issue(r0, G0); consume(r1, G1, cc); }
      else {
        is += wpb;
        if (is < s_end) { setup(); l0 = take(r0); issue(r0, G0); consume(r1, G1, cc); epilogue(cc, sum); sum = 0.0; cc = nc; }
        else { consume(r1, G1, cc); epilogue(cc, sum); break; }
      }
    }
    return;
  }
  // coded form: the run mask of a slice hangs off its pattern ids (load -> LDS -> wave OR -> gathers), so the ids alone
  // are requested one slice ahead
  int pid_ahead = a.np - 1;
  if (VD && s_begin + wave < s_end) pid_ahead = (int)a.rowpid[min((int64_t)(s_begin + wave) * ROWS + lane, lastrow)];
  for (int slice = s_begin + wave; slice < s_end; slice += wpb) {
    const int row0 = slice * ROWS;
    const int64_t row = (int64_t)row0 + lane;
    if (VD) {
      pid_n = pid_ahead;
      pid_ahead = (int)a.rowpid[min((int64_t)min(slice + wpb, s_end - 1) * ROWS + lane, lastrow)];
    }
    load_head(slice);
    const int pid = (lane < ROWS && row <= lastrow) ? pid_n : a.np - 1;   // halo lanes / rows past the end: the empty pattern
    const double e0 = e0_n, e1 = e1_n, e2 = e2_n, sp = sp_n;
    const double dinv_row = (EPI == EPI_SWEEP && tab_dinv) ? s_dinv[pid] : dr_n;
    double s = 0.0;
    double A[RB];
    auto gather = [&](int r0) {
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int o = a.run_off[r0 + q];                   // uniform: scalar load
        // column index clamped to [0,ncols) (the clamped entries have zero coefficients); uniform base + 32-bit lane offset
        if (DBG & 8) A[q] = (double)o;
        else
        A[q] = ld_off(xg, 8u * (uint32_t)min(max((int)row + o, 0), last));
      }
    };
    if (VD) {
      // Coded form: rows of a wave are consecutive, so whole runs (lines of the stencil) are absent for the entire wave
      // (Q2: a line of cell-interior dofs has 9 of the 25 runs).  OR the per-pattern run masks over the wave and visit
      // only the runs some lane needs -- in ascending order, so the sums keep their order (skipped terms are +0.0).
      uint32_t M = s_rmask[pid];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) M |= (uint32_t)__shfl_xor((int)M, o);
      M = (uint32_t)__builtin_amdgcn_readfirstlane((int)M);
      const uint8_t *tc = s_code + pid * nu;
      while (M) {
        int rr[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          rr[q] = M ? (int)__builtin_ctz(M) : -1;
          M &= M - 1;
        }
#pragma unroll
        for (int q = 0; q < RB; ++q)
          if (rr[q] >= 0) A[q] = ld_off(xg, 8u * (uint32_t)min(max((int)row + a.run_off[rr[q]], 0), last));
        {
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          if (rr[q] < 0) continue;
          double cur = A[q];
#pragma unroll
          for (int t = 0; t < K; ++t) {
            if (t > 0) cur = wave_shl1(cur);
            const int code = tc[rr[q] * K + t];
            const double g = __hiloint2double(__double2hiint(cur) & ((code == 255) ? 0 : -1), __double2loint(cur));
            s = s + sp_smem[code] * g;
          }
        }
        }
      }
    } else {
    gather(0);
    const double *tv = s_val + pid * nu;
    const uint32_t *tm = s_msk + pid * nu;
    for (int r0 = 0; r0 < a.nruns; r0 += RB) {
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        double cur = A[q];
#pragma unroll
        for (int t = 0; t < K; ++t) {
          if (t > 0 && !(DBG & 4)) cur = wave_shl1(cur);   // the last K-1 lanes receive junk: they own no row
          const int j = (r0 + q) * K + t;
          const double g = (DBG & 1) ? cur : __hiloint2double(__double2hiint(cur) & (int)tm[j], __double2loint(cur));
          s = s + ((DBG & 2) ? (double)(j + 1) : tv[j]) * g;
        }
      }
      if (r0 + RB < a.nruns) gather(r0 + RB);
    }
    }
    if (lane < ROWS && row < a.nrows) {
      // non-sweep epilogues may also emit s = omega*(Dinv*y) for the smoothing pass that follows (saves its
      // scaled_jacobi launch): requested by passing s_out / dinv / omega
      if (EPI == EPI_SET) { a.y[row] = s; if (a.s_out) a.s_out[row] = omega * (dinv[row] * s); }
      else if (EPI == EPI_SUB) { const double yn = e0 - s; a.y[row] = yn; if (a.s_out) a.s_out[row] = omega * (dinv[row] * yn); }
      else if (EPI == EPI_RESID) a.y[row] = e0 - s;
      else if (EPI == EPI_ADDTO) { const double t = a.omega != 0.0 ? a.omega * s : s; a.y[row] = t; a.x2[row] = e0 + t; }
      else {
        if (xmode == 0) a.x2[row] = e2 + e1;               // x += s_k
        else if (xmode == 2) a.x2[row] = (e2 + sp) + e1;   // the deferred x += s_{k-1}, then x += s_k: same two roundings
        const double rn = e0 - s;
        a.y[row] = rn;
        a.s_out[row] = omega * (dinv_row * rn);
      }
    }
  }
}

// The patterns each chunk of sells_kernel's launch geometry uses (chunk c = the slices workgroup c walks after the XCD remap):
// ids ascending, the empty pattern np - 1 always among them (halo lanes and rows past the end take it).  cnt[c] may exceed
// `stride` (then the list is truncated and the caller does not use the per-workgroup tables).
__global__ __launch_bounds__(256) void sellw_chunk_patterns_kernel(const uint16_t *__restrict__ rowpid, int64_t nrows, int nslices, int rows_per_slice,
                                                                    int np, int stride, uint16_t *__restrict__ pids, int32_t *__restrict__ cnt)
{
  __shared__ uint32_t bits[128];                             // np <= 4096
  const int nwg = gridDim.x, blk = blockIdx.x;
  const int chunk_lo = nslices / nwg, chunk_rem = nslices % nwg;
  const int s_begin = blk * chunk_lo + min(blk, chunk_rem);
  const int s_end = s_begin + chunk_lo + (blk < chunk_rem ? 1 : 0);
  for (int i = threadIdx.x; i < 128; i += blockDim.x) bits[i] = 0u;
  __syncthreads();
  const int64_t r0 = (int64_t)s_begin * rows_per_slice, r1 = min(nrows, (int64_t)s_end * rows_per_slice);
  for (int64_t i = r0 + threadIdx.x; i < r1; i += blockDim.x) {
    const int p = rowpid[i];
    atomicOr(&bits[p >> 5], 1u << (p & 31));
  }
  if (threadIdx.x == 0) atomicOr(&bits[(np - 1) >> 5], 1u << ((np - 1) & 31));
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < (np + 31) / 32; ++w)
      for (uint32_t m = bits[w]; m; m &= m - 1) {
        if (c < stride) pids[(size_t)blk * stride + c] = (uint16_t)(w * 32 + __builtin_ctz(m));
        ++c;
      }
    cnt[blk] = c;
  }
}

// ---------------------------------------------------------------------------
// The fused Richardson-Jacobi sweep on the shared-offset table, batched: every wave takes NB slices per iteration --
// all their row-wise operands and first gathers are requested up front, the taps run slice after slice, all stores go
// last.  Why: on gfx9 loads and stores share one counter (vmcnt) and return out of order with respect to each other, so
// a wave that still has stores in flight can only wait for a load with vmcnt(0); the one-slice-per-iteration loop of
// sells_kernel therefore drained its stores once per slice (four exposed store latencies per wave at 128^3) and its
// conditional operand loads forced waits at every join.  Here: no divergent or runtime-conditional loads (clamped
// addresses, XM and TD are template parameters), one drain per NB slices.  Same arithmetic, same order, same bits.
//   XM = SellSArgs::xmode (0: x += s_k ; 1: x deferred ; 2: x = (x + s_{k-1}) + s_k)   TD: 1/diag from the pattern table
// Plain (non-coded) table, K = 3, runs in batches of 3 (the 27-point operators).
// ---------------------------------------------------------------------------
// MK = true: 16-byte {coefficient, mask} table entries, the mask clears the high word of the gathered value of an ABSENT entry, so
//   its product with the table's +0.0 is an exact zero even when that value is Inf/NaN (the operator kernels always do this).
// MK = false (default for the fused sweeps): 8-byte entries, absent entries are plain +0.0 coefficients.  The LDS return
//   path (128 B/clk per CU) is what the tap loop waits for -- 27 reads x 16 B per lane were 11.6 us of LDS time per 128^3
//   sweep -- so halving the entry halves that.  Bit-identical for finite vectors (a +-0.0 term never changes a sum that
//   started from +0.0); with a non-finite entry in s the rows whose zero-padded taps touch it turn NaN one application
//   earlier than in the reference -- the iteration is lost either way (every dot product is NaN).  GMG_PAT_STRICT=1 = MK.
// NT = 1 (levels whose vectors no longer fit the L2s): the operands touched once per sweep -- pattern id, r in / out, x, 1/diag --
//   move with non-temporal loads / stores; s stays cacheable (this sweep gathers s_k, the next one s_{k+1}).
template <int XM, int NB, bool TD, bool MK, int NT = 0>
__global__ __launch_bounds__(kBlock) void sells_sweep_kernel(SellSArgs a)
{
  constexpr int K = 3, ROWS = 65 - K, RB = 3;
  extern __shared__ double sp_smem[];
  const int nu = K * a.nruns;
  const int tot = a.np * nu;
  // LDS: [np*nu] {coefficient, high-word mask} as 16-byte entries (one ds_read_b128 per tap), or the coefficients alone | [np] 1/diag
  PatEntry *s_tab = reinterpret_cast<PatEntry *>(sp_smem);
  double *s_tab8 = sp_smem;
  double *s_dinv = sp_smem + (MK ? 2 : 1) * (size_t)tot;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int nwg = gridDim.x;
  const int blk = remap_block(blockIdx.x, nwg, a.xcd_remap);
  // balanced chunks: nslices / nwg slices each, the remainder one more for the first workgroups (with ceil-sized chunks every
  // workgroup of the 128^3 level had 17 slices for its 4 waves -- a fifth, nearly empty round for the whole chip)
  const int chunk_lo = a.nslices / nwg, chunk_rem = a.nslices % nwg;
  const int s_begin = blk * chunk_lo + min(blk, chunk_rem);
  const int s_end = s_begin + chunk_lo + (blk < chunk_rem ? 1 : 0);
  const double *__restrict__ xg = a.x;
  const double omega = a.omega;
  const int last = (int)a.ncols - 1;
  const int lastrow = (int)a.nrows - 1;
  const bool xz = a.x_zero != 0;
  int pid[NB], row[NB];
  double e0[NB], e1[NB], e2[NB], sp[NB], dr[NB], A[NB][RB], acc[NB];
  // ---- phase 1: every load of a batch (row-wise operands + the first gathers; none depends on the pattern table) ----
  auto load_batch = [&](int sb) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int slice = min(sb + i * wpb, s_end - 1);        // a short last batch re-reads its last slice (results unused)
      row[i] = slice * ROWS + lane;
      const int rc = min(row[i], lastrow);
      pid[i] = NT ? (int)__builtin_nontemporal_load(a.rowpid + rc) : (int)a.rowpid[rc];
      e0[i] = NT ? __builtin_nontemporal_load(a.b + rc) : a.b[rc];
      e1[i] = xg[rc];
      e2[i] = 0.0; sp[i] = 0.0; dr[i] = 0.0;
      if (XM != 1) { const double xl = NT ? __builtin_nontemporal_load(a.x2 + rc) : a.x2[rc]; e2[i] = xz ? 0.0 : xl; }
      if (XM == 2) sp[i] = a.s_out[rc];
      if (!TD) dr[i] = NT ? __builtin_nontemporal_load(a.dinv + rc) : a.dinv[rc];
#pragma unroll
      for (int q = 0; q < RB; ++q) A[i][q] = ld_off(xg, 8u * (uint32_t)min(max(row[i] + a.run_off[q], 0), last));
    }
  };
  int sb = s_begin + wave;
  if (sb < s_end) load_batch(sb);                            // in flight while the table is staged (one round trip, not two)
  if (MK) { for (int i = threadIdx.x; i < tot; i += blockDim.x) s_tab[i] = a.tab[i]; }
  else { for (int i = threadIdx.x; i < tot; i += blockDim.x) s_tab8[i] = a.tab8[i]; }
  if (TD)
    for (int i = threadIdx.x; i < a.np; i += blockDim.x) s_dinv[i] = a.pdinv[i];
  __syncthreads();
  while (sb < s_end) {
    // ---- phase 2: taps, slice after slice (rows summed in ascending column order) ----
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const PatEntry *te = s_tab + pid[i] * nu;
      const double *tv = s_tab8 + pid[i] * nu;
      double s = 0.0;
      for (int r0 = 0; r0 < a.nruns; r0 += RB) {
        double cur[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) cur[q] = A[i][q];
        if (r0 + RB < a.nruns) {
#pragma unroll
          for (int q = 0; q < RB; ++q) A[i][q] = ld_off(xg, 8u * (uint32_t)min(max(row[i] + a.run_off[r0 + RB + q], 0), last));
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          double c = cur[q];
#pragma unroll
          for (int t = 0; t < K; ++t) {
            if (t > 0) c = wave_shl1(c);                     // the last K-1 lanes receive junk: they own no row
            const int j = (r0 + q) * K + t;
            if (MK) {
              const PatEntry en = te[j];
              const double g = __hiloint2double(__double2hiint(c) & (int)en.m, __double2loint(c));
              s = s + en.v * g;
            } else
              s = s + tv[j] * c;
          }
        }
      }
      acc[i] = s;
    }
    // ---- phase 3: every store of the batch ----
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (sb + i * wpb < s_end && lane < ROWS && row[i] <= lastrow) {
        const int r = row[i];
        const double dinv_row = TD ? s_dinv[pid[i]] : dr[i];
        const double rn = e0[i] - acc[i];
        if (NT) {
          if (XM == 0) __builtin_nontemporal_store(e2[i] + e1[i], a.x2 + r);
          else if (XM == 2) __builtin_nontemporal_store((e2[i] + sp[i]) + e1[i], a.x2 + r);
          __builtin_nontemporal_store(rn, a.y + r);
        } else {
          if (XM == 0) a.x2[r] = e2[i] + e1[i];
          else if (XM == 2) a.x2[r] = (e2[i] + sp[i]) + e1[i];
          a.y[r] = rn;
        }
        a.s_out[r] = omega * (dinv_row * rn);
      }
    }
    sb += wpb * NB;
    if (sb < s_end) load_batch(sb);
  }
}

// ---------------------------------------------------------------------------
// The same sweep WITHOUT the s vector, for operators whose rows all carry the same diagonal (constant-coefficient operators on
// uniform meshes: every free node sees the same cells, so 1/diag is ONE number d).  s_k = omega*(d*r_k) is then a function of the
// gathered value alone: the kernel gathers r_k itself and forms s once per LOADED value -- before the DPP shifts hand it to the
// neighbouring lanes --, two multiplies per run and lane, the very two roundings of RichardsonSmoothers.jl:91-92 /
// JacobiLinearSolvers.jl:45.  Gone: the s store and the s load of every row and sweep, the scaled-Jacobi launch in front of every
// pass, and the ping-pong of s (r ping-pongs instead: the gathers read r_k while the rows write r_{k+1}).  Per row and sweep
// 2 + 8 B read and 8 B written when x is untouched (XM = 1), + 16 B + 8 B when it is (XM = 2): 30 B on average against 46 B --
// and 12 B instead of 20 B of WRITES, which is what this kernel pays most for once the level has left the caches
// (profiles/r03_tuning.md).  Same products, same order, same roundings as sells_sweep_kernel: bit-identical (tested).
//   a.x = r_k (gathered, incl. the row's own value) ; a.y = r_{k+1} ; a.s_out = r_{k-1} (XM = 2 only; may be a.y) ; a.pdinv[0] = d
// ---------------------------------------------------------------------------
// clamp(x, 0, hi) in one instruction (the compiler emits v_max + v_min for min(max(x, 0), hi) with a run-time bound)
__device__ __forceinline__ int clamp0_med3(int x, int hi)
{
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
  return r;
}

template <int XM, int NB, bool MK, int NT = 0, bool FM = false>
__global__ __launch_bounds__(kBlock) void sells_rsweep_kernel(SellSArgs a)
{
  constexpr int K = 3, ROWS = 65 - K, RB = 3;
  extern __shared__ double sp_smem[];
  const int nu = K * a.nruns;
  const int tot = a.np * nu;
  // LDS: [np*nu] coefficients, dense (absent entries hold 0.0) | strict form (MK): [np*nu] high-word masks behind them, read only by
  // the rare batches that hold a non-finite value -- the finite path reads 8 B per tap, never 16
  double *s_tab8 = sp_smem;
  uint32_t *s_msk = reinterpret_cast<uint32_t *>(sp_smem + tot);
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int nwg = gridDim.x;
  const int blk = remap_block(blockIdx.x, nwg, a.xcd_remap);
  const int chunk_lo = a.nslices / nwg, chunk_rem = a.nslices % nwg;
  const int s_begin = blk * chunk_lo + min(blk, chunk_rem);
  const int s_end = s_begin + chunk_lo + (blk < chunk_rem ? 1 : 0);
  const double *__restrict__ rg = a.x;
  const double omega = a.omega;
  const double du = a.pdinv[0];                              // the one 1/diag (uniform: scalar load)
  const int last8 = 8 * ((int)a.ncols - 1);
  const int lastrow = (int)a.nrows - 1;
  const bool xz = a.x_zero != 0;
  int pid[NB], row[NB];
  double e0[NB], e2[NB], rp[NB], A[NB][RB], acc[NB];
  auto load_batch = [&](int sb) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int slice = min(sb + i * wpb, s_end - 1);
      row[i] = slice * ROWS + lane;
      const int rc = min(row[i], lastrow);
      pid[i] = NT ? (int)__builtin_nontemporal_load(a.rowpid + rc) : (int)a.rowpid[rc];
      e0[i] = rg[rc];
      e2[i] = 0.0; rp[i] = 0.0;
      if (XM != 1) { const double xl = NT ? __builtin_nontemporal_load(a.x2 + rc) : a.x2[rc]; e2[i] = xz ? 0.0 : xl; }
      if (XM == 2) rp[i] = a.s_out[rc];                      // r_{k-1} of the row, read before this sweep overwrites it
#pragma unroll
      for (int q = 0; q < RB; ++q) A[i][q] = ld_off(rg, (uint32_t)clamp0_med3(8 * row[i] + 8 * a.run_off[q], last8));   // byte offsets: add + clamp (ncols < 2^28)
    }
  };
  int sb = s_begin + wave;
  if (sb < s_end) load_batch(sb);
  if (MK) { for (int i = threadIdx.x; i < tot; i += blockDim.x) { const PatEntry en = a.tab[i]; s_tab8[i] = en.v; s_msk[i] = en.m; } }
  else { for (int i = threadIdx.x; i < tot; i += blockDim.x) s_tab8[i] = a.tab8[i]; }
  __syncthreads();
  while (sb < s_end) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const uint32_t *tm = s_msk + pid[i] * nu;
      const double *tv = s_tab8 + pid[i] * nu;
      double s = 0.0;
      for (int r0 = 0; r0 < a.nruns; r0 += RB) {
        double cur[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) cur[q] = omega * (du * A[i][q]);   // s = omega*(Dinv*r): once per loaded value
        if (r0 + RB < a.nruns) {
#pragma unroll
          for (int q = 0; q < RB; ++q) A[i][q] = ld_off(rg, (uint32_t)clamp0_med3(8 * row[i] + 8 * a.run_off[r0 + RB + q], last8));
        }
        // MK (strict masks): absent entries hold 0.0, so with finite values in every lane of the three windows the products are
        // exact zeros and leave the partial sums as they are -- the mask is only applied (27 v_and + 27 mask reads per slice)
        // when a window holds a non-finite value, which must reach exactly the rows that store a coefficient for it
        bool fin = true;
        if (MK) {
#pragma unroll
          for (int q = 0; q < RB; ++q) fin = fin && __builtin_isfinite(cur[q]);
          fin = __all(fin);
        }
        if (MK && !fin) {
#pragma unroll
          for (int q = 0; q < RB; ++q) {
            double c = cur[q];
#pragma unroll
            for (int t = 0; t < K; ++t) {
              if (t > 0) c = wave_shl1(c);
              const int j = (r0 + q) * K + t;
              const double g = __hiloint2double(__double2hiint(c) & (int)tm[j], __double2loint(c));
              s = FM ? __builtin_fma(tv[j], g, s) : s + tv[j] * g;
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < RB; ++q) {
            double c = cur[q];
#pragma unroll
            for (int t = 0; t < K; ++t) {
              if (t > 0) c = wave_shl1(c);
              const int j = (r0 + q) * K + t;
              s = FM ? __builtin_fma(tv[j], c, s) : s + tv[j] * c;
            }
          }
        }
      }
      acc[i] = s;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (sb + i * wpb < s_end && lane < ROWS && row[i] <= lastrow) {
        const int r = row[i];
        const double rn = e0[i] - acc[i];
        const double sk = omega * (du * e0[i]);              // the row's own s_k
        if (XM == 0) { const double xn = e2[i] + sk; if (NT) __builtin_nontemporal_store(xn, a.x2 + r); else a.x2[r] = xn; }
        else if (XM == 2) { const double xn = (e2[i] + omega * (du * rp[i])) + sk; if (NT) __builtin_nontemporal_store(xn, a.x2 + r); else a.x2[r] = xn; }
        a.y[r] = rn;
      }
    }
    sb += wpb * NB;
    if (sb < s_end) load_batch(sb);
  }
}

// ---------------------------------------------------------------------------
// The r-gather sweep with TWO rows per lane ("pair sweep").  What sells_rsweep_kernel pays per 62-row slice, by ablation at 128^3
// (tools/mb_psweep.hip, profiles/r04_tuning.md): ~4 us of launch + skeleton, 2 us of row-wise loads, 3.5 us of stores, 6 us of
// gathers (every vector-memory instruction moves 8 B per lane: the texture-address path runs at half its rate), 7.5 us of taps of
// which 54 multiply / add instructions are the arithmetic itself and 18 + 36 + 18 convert the loaded values, shift them across
// lanes and form addresses.  Here a lane owns rows 2l and 2l+1 of a slice of 126 rows: one 16-byte load per run brings its two
// window values (the next two come from lane l+1: still 4 DPP moves, now per TWO rows), one 16-byte load / store moves the pair's
// r, x; per row 9 instead of 18 conversions, 18 instead of 36 DPP moves, half the vector-memory instructions at twice the width.
// Slices whose windows could leave the vector (the first and last few of a level) and the ragged last slice take element-wise
// clamped loads and stores; all others need no clamp at all.  Same taps in the same order on the same values: bit-identical to
// sells_rsweep_kernel (FM = false).  FM: fused multiply-add taps (option pat_fma; one rounding per tap -- not the reference's mul!).
//   a.x = r_k (gathered) ; a.y = r_{k+1} ; a.s_out = r_{k-1} (XM = 2) ; a.pdinv[0] = d ; a.nslices = ceil(nrows / 126)
// ---------------------------------------------------------------------------
typedef double gmg_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gmg_d2 ld2_unaligned(const double *p)
{
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  return *reinterpret_cast<const d2u *>(p);
}
__device__ __forceinline__ void st2_unaligned(double *p, gmg_d2 v)
{
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  *reinterpret_cast<d2u *>(p) = v;
}

// NR = number of runs (9: 27-point operators; 3: 9-point) -- compile time, so that the run offsets live in scalar registers (read in the
// loop from the argument array they are vector loads the gathers then wait for) and the run loop is fully unrolled.
// OCC = 1: the run loop stays rolled (groups of three runs; unrolled, the compiler hoists the 54 coefficient reads of a slice and the
// kernel takes 105 registers: four waves per SIMD) and the kernel is capped at 64 registers -- eight waves per SIMD.
// OCC = 2: the same for workgroups of eight waves (four per CU): at one slice per wave the table is staged once per eight slices
// (288^3: 164 / 236 -> 153 / 225 us by variant; sixteen waves: no further gain)
template <int XM, bool MK, bool FM, int NR, int OCC = 0>
__global__ __launch_bounds__(OCC == 2 ? 2 * kBlock : kBlock, OCC == 2 ? 4 : (OCC ? 8 : 1)) void sells_r2sweep_kernel(SellSArgs a)
{
  constexpr int K = 3, ROWS2 = 126, RB = 3;
  extern __shared__ double sp_smem[];
  const int nu = K * NR;
  const int tot = a.np * nu;
  double *s_tab8 = sp_smem;                                   // [np*nu] coefficients, dense | MK: [np*nu] high-word masks
  uint32_t *s_msk = reinterpret_cast<uint32_t *>(sp_smem + tot);
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: `inner` is a uniform branch
  const int nwg = gridDim.x;
  const int blk = remap_block(blockIdx.x, nwg, a.xcd_remap);
  const int chunk_lo = a.nslices / nwg, chunk_rem = a.nslices % nwg;
  const int s_begin = blk * chunk_lo + min(blk, chunk_rem);
  const int s_end = s_begin + chunk_lo + (blk < chunk_rem ? 1 : 0);
  const double *__restrict__ rg = a.x;
  const double omega = a.omega;
  const double du = a.pdinv[0];
  const int last = (int)a.ncols - 1;
  const int lastrow = (int)a.nrows - 1;
  const bool xz = a.x_zero != 0;
  // a slice is "inner" when every window of every run and the pair loads of all 64 lanes stay inside the vectors
  const int lo_need = -a.minoff, hi_need = a.maxoff + 2 * 64 + 2;
  int roff[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) roff[q] = __builtin_amdgcn_readfirstlane(a.run_off[q]);
  const int voff = a.run_off[min(lane, NR - 1)];             // OCC: lane q holds run_off[q], read with v_readlane in the rolled loop
  int pidA = 0, pidB = 0, row = 0;
  bool inner = false;
  gmg_d2 e0, e2, rp, A[RB];
  auto load_slice = [&](int slice) {
    const int r0 = slice * ROWS2;
    row = r0 + 2 * lane;
    inner = r0 >= lo_need && r0 + hi_need <= last && r0 + 2 * 64 <= lastrow;      // wave-uniform
    e2 = gmg_d2{0.0, 0.0}; rp = gmg_d2{0.0, 0.0};
    if (inner) {
      const uint32_t pp = *reinterpret_cast<const uint32_t *>(a.rowpid + row);   // row is even: 4-byte aligned
      pidA = (int)(pp & 0xffffu); pidB = (int)(pp >> 16);
      e0 = ld2_unaligned(rg + row);
      if (XM != 1) { const gmg_d2 xl = ld2_unaligned(a.x2 + row); e2 = xz ? gmg_d2{0.0, 0.0} : xl; }
      if (XM == 2) rp = ld2_unaligned(a.s_out + row);
#pragma unroll
      for (int q = 0; q < RB; ++q) A[q] = ld2_unaligned(rg + row + roff[q]);
    } else {
      const int ra = min(row, lastrow), rb = min(row + 1, lastrow);
      pidA = (int)a.rowpid[ra]; pidB = (int)a.rowpid[rb];
      e0 = gmg_d2{rg[ra], rg[rb]};
      if (XM != 1) { const gmg_d2 xl = gmg_d2{a.x2[ra], a.x2[rb]}; e2 = xz ? gmg_d2{0.0, 0.0} : xl; }
      if (XM == 2) rp = gmg_d2{a.s_out[ra], a.s_out[rb]};
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int c = row + roff[q];
        A[q] = gmg_d2{rg[min(max(c, 0), last)], rg[min(max(c + 1, 0), last)]};
      }
    }
  };
  int sb = s_begin + wave;
  if (sb < s_end) load_slice(sb);
  // GM: the masks stay in global memory -- only the rare redo of a slice whose sum came out non-finite reads them; staging them is a
  // third of every workgroup's table traffic, 4-5 % of the sweep at one slice per wave (288^3: 172 -> 164 us).  Only where the redo
  // path's addresses fit the registers: the x-updating forms of the 64-register kernel have none to spare (they spill with it).
  constexpr bool GM = MK && OCC != 0 && XM == 1;
  if (MK && !GM) { for (int i = threadIdx.x; i < tot; i += blockDim.x) { const PatEntry en = a.tab[i]; s_tab8[i] = en.v; s_msk[i] = en.m; } }
  else if (GM) { for (int i = threadIdx.x; i < tot; i += blockDim.x) s_tab8[i] = a.tab8 ? a.tab8[i] : a.tab[i].v; }
  else { for (int i = threadIdx.x; i < tot; i += blockDim.x) s_tab8[i] = a.tab8[i]; }
  __syncthreads();
  while (sb < s_end) {
    const uint32_t *tmA = s_msk + pidA * nu, *tmB = s_msk + pidB * nu;
    const double *tvA = s_tab8 + pidA * nu, *tvB = s_tab8 + pidB * nu;
    double sA = 0.0, sB = 0.0;
    if constexpr (OCC != 0) {
#pragma unroll 1
      for (int r0 = 0; r0 < NR; r0 += RB) {
        gmg_d2 cur[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) cur[q] = gmg_d2{omega * (du * A[q].x), omega * (du * A[q].y)};
        if (r0 + RB < NR) {
          if (inner) {
#pragma unroll
            for (int q = 0; q < RB; ++q) A[q] = ld2_unaligned(rg + row + __builtin_amdgcn_readlane(voff, r0 + RB + q));
          } else {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
              const int c = row + __builtin_amdgcn_readlane(voff, r0 + RB + q);
              A[q] = gmg_d2{rg[min(max(c, 0), last)], rg[min(max(c + 1, 0), last)]};
            }
          }
        }
        const double *ta = tvA + r0 * K, *tb = tvB + r0 * K;
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          const double w0 = cur[q].x, w1 = cur[q].y;
          const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
          const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
          for (int t = 0; t < K; ++t) {
            const double ca = ta[q * K + t], cb = tb[q * K + t];
            sA = FM ? __builtin_fma(ca, wa[t], sA) : sA + ca * wa[t];
            sB = FM ? __builtin_fma(cb, wb[t], sB) : sB + cb * wb[t];
          }
        }
      }
    } else
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += RB) {
      gmg_d2 cur[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) cur[q] = gmg_d2{omega * (du * A[q].x), omega * (du * A[q].y)};   // s = omega*(Dinv*r): once per loaded value
      if (r0 + RB < NR) {
        if (inner) {
#pragma unroll
          for (int q = 0; q < RB; ++q) A[q] = ld2_unaligned(rg + row + roff[r0 + RB + q]);
        } else {
#pragma unroll
          for (int q = 0; q < RB; ++q) {
            const int c = row + roff[r0 + RB + q];
            A[q] = gmg_d2{rg[min(max(c, 0), last)], rg[min(max(c + 1, 0), last)]};
          }
        }
      }
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        // the four window values of this lane's two rows: w0, w1 its own, w2, w3 = lane l+1's w0, w1
        const double w0 = cur[q].x, w1 = cur[q].y;
        const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
        const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const int j = (r0 + q) * K + t;
          const double ca = tvA[j], cb = tvB[j];
          sA = FM ? __builtin_fma(ca, wa[t], sA) : sA + ca * wa[t];
          sB = FM ? __builtin_fma(cb, wb[t], sB) : sB + cb * wb[t];
        }
      }
    }
    // Strict form: absent entries hold 0.0, so with finite window values their products are exact zeros and the sums above ARE the
    // masked sums.  A non-finite value in reach turns every sum it touches into NaN / Inf -- through a stored coefficient or through
    // 0 x Inf = NaN of an absent one -- so "all sums finite" proves no mask was needed; otherwise (rare: a vector that already holds
    // Inf / NaN) the slice is redone with the masks, which confine the value to the rows that store a coefficient for it.
    if (MK && !__all(__builtin_isfinite(sA) && __builtin_isfinite(sB))) {
      sA = 0.0; sB = 0.0;
#pragma unroll 1
      for (int q = 0; q < NR; ++q) {
        const int c = row + roff[q];
        const double w0 = omega * (du * rg[min(max(c, 0), last)]), w1 = omega * (du * rg[min(max(c + 1, 0), last)]);
        const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
        const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const int j = q * K + t;
          const int ma = GM ? (int)a.tab[pidA * nu + j].m : (int)tmA[j], mb = GM ? (int)a.tab[pidB * nu + j].m : (int)tmB[j];
          const double ga = __hiloint2double(__double2hiint(wa[t]) & ma, __double2loint(wa[t]));
          const double gb = __hiloint2double(__double2hiint(wb[t]) & mb, __double2loint(wb[t]));
          sA = FM ? __builtin_fma(tvA[j], ga, sA) : sA + tvA[j] * ga;
          sB = FM ? __builtin_fma(tvB[j], gb, sB) : sB + tvB[j] * gb;
        }
      }
    }
    // results of the pair
    const gmg_d2 rn = gmg_d2{e0.x - sA, e0.y - sB};
    const gmg_d2 sk = gmg_d2{omega * (du * e0.x), omega * (du * e0.y)};     // the rows' own s_k
    gmg_d2 xn = gmg_d2{0.0, 0.0};
    if (XM == 0) xn = gmg_d2{e2.x + sk.x, e2.y + sk.y};
    else if (XM == 2) xn = gmg_d2{(e2.x + omega * (du * rp.x)) + sk.x, (e2.y + omega * (du * rp.y)) + sk.y};
    if (lane < 63) {
      if (inner) {
        if (XM != 1) st2_unaligned(a.x2 + row, xn);
        st2_unaligned(a.y + row, rn);
      } else {
        if (row <= lastrow) { if (XM != 1) a.x2[row] = xn.x; a.y[row] = rn.x; }
        if (row + 1 <= lastrow) { if (XM != 1) a.x2[row + 1] = xn.y; a.y[row + 1] = rn.y; }
      }
    }
    sb += wpb;
    if (sb < s_end) load_slice(sb);
  }
}

// ---------------------------------------------------------------------------
// The pair sweep as a WALK along the slowest grid direction ("z-walk"; round 5).  The nine runs of a 27-point operator are a 3 x 3
// grid of offsets, run q = 3 (dz + 1) + (dy + 1) at dz P + dy L - 1 (P = rows per grid plane, L = rows per grid line), so the window
// of run q of the slice at row r0 + P IS the window of run q + 3 of the slice at r0.  A wave therefore keeps an interval of <= 126
// rows of the plane and walks T planes upwards: per step it loads THREE new windows instead of nine and converts 6 instead of 18
// loaded values (s = omega (d r)); the other six windows stay in registers, already converted.  The loads of step k + 1 (three
// windows, the pair's own r / x / r_{k-1}, the pattern ids) are issued BEFORE the taps of step k: a wave always has one step of
// memory requests in flight behind ~160 VALU instructions, which the per-slice kernels (three dependent load -> use phases per
// slice, nothing in flight while a slice is multiplied) leave to the other waves of the SIMD.
// A plane of P rows is cut into m = ceil(P / 126) intervals of floor / ceil (P / m) rows (the slices of the per-slice kernels are
// cut from the flattened row range instead: their starts drift against the grid planes by P mod 126 per plane, so nothing lines up).
// Chain c = (z-block c / m, interval c % m): neighbouring waves hold neighbouring intervals of the same planes and meet in L1 / L2.
// Same taps in the same order on the same values, same strict-mask rule: bit-identical to sells_rsweep_kernel / sells_r2sweep_kernel.
//   a.x = r_k (gathered) ; a.y = r_{k+1} ; a.s_out = r_{k-1} (XM = 2) ; a.pdinv[0] = d
// ---------------------------------------------------------------------------
// build-time switches of the z-walk (defaults = the product; profiles/r05_tuning.md has the A/B runs): NT bit 0 = results stored
// non-temporally, bit 1 = x / r_{k-1} loaded non-temporally (each is touched once per sweep); PF = steps the requests run ahead of the
// taps; PD = runs the coefficient reads run ahead of the taps
#ifndef GMG_ZW_NT
#define GMG_ZW_NT 3
#endif
#ifndef GMG_ZW_PF
#define GMG_ZW_PF 1
#endif
#ifndef GMG_ZW_PD
#define GMG_ZW_PD 1
#endif
struct ZWalkGeo {
  int P;        // rows per plane: run_off[q + 3] - run_off[q]
  int m;        // intervals per plane
  int T;        // planes per chain
  int nplanes;  // ceil(nrows / P)
  int nchains;  // ceil(nplanes / T) * m
};

// EPI = EPI_SWEEP: the sweep (XM = its x mode).  EPI_SET / EPI_SUB / EPI_RESID: the operator mat-vecs of the same levels in the same
// walk -- y = A x (CGSolvers.jl:104), y -= A x (GMGLinearSolvers.jl:495), y = b - A x (CGSolvers.jl:79): the gathered value IS the
// window value, a.x gathered, a.y result (SUB: also read), a.b (RESID); same taps in the same order as sells_kernel / sells_r2mv_kernel.
template <int XM, bool MK, bool FM, int EPI = EPI_SWEEP>
__global__ __launch_bounds__(kBlock, 4) void sells_zsweep_kernel(SellSArgs a, ZWalkGeo g)
{
  constexpr bool SW = EPI == EPI_SWEEP;
  static_assert(SW || (XM == 1 && (EPI == EPI_SET || EPI == EPI_SUB || EPI == EPI_RESID)), "mat-vec epilogues: instantiate with XM = 1");
  constexpr int K = 3, NR = 9, nu = K * NR, NUP = 28;       // NUP: LDS stride of a pattern (doubles) -- even, so that pairs of coefficients are 16-byte aligned
  extern __shared__ double sp_smem[];
  const int tot = a.np * NUP;
  double *s_tab8 = sp_smem;                                   // [np*NUP] coefficients, dense (the masks stay in global memory: rare path)
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int blk = remap_block(blockIdx.x, gridDim.x, a.xcd_remap);
  const int chain = blk * wpb + wave;                         // wave-uniform
  const double *__restrict__ rg = a.x;
  const double omega = a.omega;
  const double du = SW ? a.pdinv[0] : 1.0;
  const double *__restrict__ eg = SW ? a.x : (EPI == EPI_RESID ? a.b : a.y);   // the rows' own operand (SET: none)
  const int last = (int)a.ncols - 1;
  const int lastrow = (int)a.nrows - 1;
  const bool xz = a.x_zero != 0;
  int roff[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) roff[q] = __builtin_amdgcn_readfirstlane(a.run_off[q]);
  const bool live = chain < g.nchains;
  const int zb = live ? chain / g.m : 0, iv = live ? chain - zb * g.m : 0;
  const int b0 = (int)(((int64_t)iv * g.P) / g.m), b1 = (int)(((int64_t)(iv + 1) * g.P) / g.m);
  const int len = b1 - b0;                                    // <= 126
  const int z0 = zb * g.T, z1 = min(g.nplanes, z0 + g.T);
  int r0 = z0 * g.P + b0;                                     // first row of the step's slice (scalar)
  auto conv = [&](gmg_d2 v) -> gmg_d2 { return SW ? gmg_d2{omega * (du * v.x), omega * (du * v.y)} : v; };   // sweep: s = omega*(Dinv*r), once per loaded value
  struct RowOps { int pidA, pidB; gmg_d2 e0, e2, rp; };
  const int nsteps = z1 - z0;
  // A chain is "inner" when every window of every step and the pair loads of all 64 lanes stay inside the vectors: no clamp anywhere.
  // The two forms of the loop differ ONLY in how a value is addressed; each issues a fixed number of memory instructions per step
  // on a single control path -- with a branch around the requests the wait-count pass must assume the path without them and parks
  // the wave (vmcnt(0)) on the requests of step k + 1 before the taps of step k: nothing overlaps (measured: 68 % of the wave
  // cycles in s_waitcnt).  The last step re-requests its own operands instead of branching around the prefetch.
  const bool inner_chain = live && nsteps > 0 && r0 + roff[0] >= 0 && r0 + (nsteps - 1) * g.P + roff[NR - 1] + 2 * 63 + 1 <= last &&
                           r0 + (nsteps - 1) * g.P + 2 * 63 + 1 <= lastrow;
  auto run_chain = [&](auto in_tag) {
    constexpr bool IN = decltype(in_tag)::value;
    // one window: the lane's two values at base + 2 lane, base + 2 lane + 1 (lane l + 1 holds the next two)
    auto loadw = [&](int base) -> gmg_d2 {
      if (IN) return ld2_unaligned(rg + base + 2 * lane);
      const int c = base + 2 * lane;
      return gmg_d2{rg[min(max(c, 0), last)], rg[min(max(c + 1, 0), last)]};
    };
    auto load_rows = [&](int rbase) -> RowOps {
      RowOps o;
      const int row = rbase + 2 * lane;
      o.e2 = gmg_d2{0.0, 0.0}; o.rp = gmg_d2{0.0, 0.0};
      if (IN) {
        o.pidA = (int)a.rowpid[row]; o.pidB = (int)a.rowpid[row + 1];
        o.e0 = gmg_d2{0.0, 0.0};
        if (EPI != EPI_SET) o.e0 = ld2_unaligned(eg + row);
        typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
        if (GMG_ZW_NT & 2) {
          if (XM != 1) { const gmg_d2 xl = __builtin_nontemporal_load(reinterpret_cast<const d2u *>(a.x2 + row)); o.e2 = xz ? gmg_d2{0.0, 0.0} : xl; }
          if (XM == 2) o.rp = __builtin_nontemporal_load(reinterpret_cast<const d2u *>(a.s_out + row));
        } else {
        if (XM != 1) { const gmg_d2 xl = ld2_unaligned(a.x2 + row); o.e2 = xz ? gmg_d2{0.0, 0.0} : xl; }
        if (XM == 2) o.rp = ld2_unaligned(a.s_out + row);
        }
      } else {
        const int ra = min(row, lastrow), rb = min(row + 1, lastrow);
        o.pidA = (int)a.rowpid[ra]; o.pidB = (int)a.rowpid[rb];
        o.e0 = gmg_d2{0.0, 0.0};
        if (EPI != EPI_SET) o.e0 = gmg_d2{eg[ra], eg[rb]};
        if (XM != 1) { const gmg_d2 xl = gmg_d2{a.x2[ra], a.x2[rb]}; o.e2 = xz ? gmg_d2{0.0, 0.0} : xl; }
        if (XM == 2) o.rp = gmg_d2{a.s_out[ra], a.s_out[rb]};
      }
      return o;
    };
    gmg_d2 C[NR];
    RowOps cur;
#pragma unroll
    for (int q = 0; q < NR; ++q) C[q] = loadw(r0 + roff[q]);
    cur = load_rows(r0);
    // requests run PF steps ahead of the taps (GMG_ZW_PF; 1: the operands of step k + 1 are requested before the taps of step k)
    constexpr int PF = XM == 1 ? GMG_ZW_PF : 1;
    const int rfirst = r0;
    gmg_d2 N[PF][3];
    RowOps nx[PF];
#pragma unroll
    for (int d = 1; d < PF; ++d) {
      const int rb = rfirst + min(d, nsteps - 1) * g.P;
#pragma unroll
      for (int q = 0; q < 3; ++q) N[d - 1][q] = loadw(rb + roff[6 + q]);
      nx[d - 1] = load_rows(rb);
    }
#pragma unroll
    for (int q = 0; q < NR; ++q) C[q] = conv(C[q]);
    // results of a step are stored at the top of the NEXT step, in front of that step's requests: the wait for the requests at the
    // end of a step (in-order counter) then never includes a store that was issued a few cycles earlier
    gmg_d2 prn = gmg_d2{0.0, 0.0}, pxn = gmg_d2{0.0, 0.0};
    int prow = 0, pnm = 0;
    auto put = [&]() {
      if (pnm >= 2) {
        typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
        if (GMG_ZW_NT & 1) {
          if (XM != 1) __builtin_nontemporal_store(pxn, reinterpret_cast<d2u *>(a.x2 + prow));
          __builtin_nontemporal_store(prn, reinterpret_cast<d2u *>(a.y + prow));
        } else {
        if (XM != 1) st2_unaligned(a.x2 + prow, pxn);
        st2_unaligned(a.y + prow, prn);
        }
      } else if (pnm == 1) {
        if (XM != 1) a.x2[prow] = pxn.x;
        a.y[prow] = prn.x;
      }
    };
#pragma unroll 1
    for (int z = z0; z < z1; ++z) {
      put();
      // step z + PF (past the end of the chain: the last step again, unused): requested before the taps of step z
      {
        const int rb = rfirst + min(z - z0 + PF, nsteps - 1) * g.P;
#pragma unroll
        for (int q = 0; q < 3; ++q) N[PF - 1][q] = loadw(rb + roff[6 + q]);
        nx[PF - 1] = load_rows(rb);
      }
      const double *tvA = s_tab8 + cur.pidA * NUP, *tvB = s_tab8 + cur.pidB * NUP;
      double sA = 0.0, sB = 0.0;
      // the coefficients of run q + PD are requested before the taps of run q (the scheduler, left alone, issues every LDS read right
      // in front of its first use and the wave waits out the LDS latency 14 times per step)
      constexpr int PD = GMG_ZW_PD;
      double ca[NR][K], cb[NR][K];
      // the three coefficients of run q: one 16-byte read (the pair at an even index) + one 8-byte read
      auto coef3 = [&](const double *tv, int q, double *out) {
        typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));
        if ((q & 1) == 0) { const d2a pr = *reinterpret_cast<const d2a *>(tv + q * K); out[0] = pr.x; out[1] = pr.y; out[2] = tv[q * K + 2]; }
        else { const d2a pr = *reinterpret_cast<const d2a *>(tv + q * K + 1); out[0] = tv[q * K]; out[1] = pr.x; out[2] = pr.y; }
      };
#pragma unroll
      for (int q = 0; q < PD; ++q) { coef3(tvA, q, ca[q]); coef3(tvB, q, cb[q]); }
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        if (q + PD < NR) { coef3(tvA, q + PD, ca[q + PD]); coef3(tvB, q + PD, cb[q + PD]); }
        __builtin_amdgcn_sched_barrier(0);
        // the four window values of this lane's two rows: w0, w1 its own, w2, w3 = lane l+1's w0, w1
        const double w0 = C[q].x, w1 = C[q].y;
        const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
        const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
        for (int t = 0; t < K; ++t) {
          sA = FM ? __builtin_fma(ca[q][t], wa[t], sA) : sA + ca[q][t] * wa[t];
          sB = FM ? __builtin_fma(cb[q][t], wb[t], sB) : sB + cb[q][t] * wb[t];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // strict form (see sells_r2sweep_kernel): "all sums of the slice finite" proves that no mask was needed; otherwise the slice is
      // redone from memory with the masks (read from global memory: a vector that already holds Inf / NaN)
      if (MK && !__all(__builtin_isfinite(sA) && __builtin_isfinite(sB))) {
        sA = 0.0; sB = 0.0;
        const int row = r0 + 2 * lane;
#pragma unroll 1
        for (int q = 0; q < NR; ++q) {
          const int c = row + roff[q];
          const double g0 = rg[min(max(c, 0), last)], g1 = rg[min(max(c + 1, 0), last)];
          const double w0 = SW ? omega * (du * g0) : g0, w1 = SW ? omega * (du * g1) : g1;
          const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
          const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
          for (int t = 0; t < K; ++t) {
            const int j = q * K + t;
            const int ma = (int)a.tab[cur.pidA * nu + j].m, mb = (int)a.tab[cur.pidB * nu + j].m;
            const double ga = __hiloint2double(__double2hiint(wa[t]) & ma, __double2loint(wa[t]));
            const double gb = __hiloint2double(__double2hiint(wb[t]) & mb, __double2loint(wb[t]));
            sA = FM ? __builtin_fma(tvA[j], ga, sA) : sA + tvA[j] * ga;
            sB = FM ? __builtin_fma(tvB[j], gb, sB) : sB + tvB[j] * gb;
          }
        }
      }
      // results of the pair
      const gmg_d2 rn = EPI == EPI_SET ? gmg_d2{sA, sB} : gmg_d2{cur.e0.x - sA, cur.e0.y - sB};
      const gmg_d2 sk = gmg_d2{omega * (du * cur.e0.x), omega * (du * cur.e0.y)};     // the rows' own s_k
      gmg_d2 xn = gmg_d2{0.0, 0.0};
      if (XM == 0) xn = gmg_d2{cur.e2.x + sk.x, cur.e2.y + sk.y};
      else if (XM == 2) xn = gmg_d2{(cur.e2.x + omega * (du * cur.rp.x)) + sk.x, (cur.e2.y + omega * (du * cur.rp.y)) + sk.y};
      prow = r0 + 2 * lane;
      pnm = min(len - 2 * lane, lastrow + 1 - prow);                       // rows of this lane inside the interval and the level: <= 0, 1, >= 2
      prn = rn; pxn = xn;
#pragma unroll
      for (int q = 0; q < 6; ++q) C[q] = C[q + 3];
#pragma unroll
      for (int q = 0; q < 3; ++q) C[6 + q] = conv(N[0][q]);
      cur = nx[0];
#pragma unroll
      for (int d = 0; d + 1 < PF; ++d) {
#pragma unroll
        for (int q = 0; q < 3; ++q) N[d][q] = N[d + 1][q];
        nx[d] = nx[d + 1];
      }
      r0 += g.P;
    }
    put();
  };
  // the coefficient table: staged by the whole workgroup (also by the waves of the last workgroup that have no chain)
  for (int i = threadIdx.x; i < tot; i += blockDim.x) {
    const int pq = i / NUP, j = i - pq * NUP;
    s_tab8[i] = j < nu ? (a.tab8 ? a.tab8[pq * nu + j] : a.tab[pq * nu + j].v) : 0.0;
  }
  __syncthreads();
  if (!live || nsteps <= 0) return;
  if (inner_chain) run_chain(std::integral_constant<bool, true>{});
  else run_chain(std::integral_constant<bool, false>{});
}

// ---------------------------------------------------------------------------
// TWO sweeps per pass over the data (round 6).  RichardsonSmoothers.jl:90-97 is ten sweeps in a row over the same vectors; launched one
// by one, every sweep reads r_k and writes r_{k+1} through memory (and every second one x).  On a level whose rows are the nodes of a
// grid (row = z P + y L + x: the 3 x 3 grid of run offsets says so) sweep k + 1 of a tile needs r_{k+1} only one node further out
// than the tile, so a workgroup can run BOTH sweeps on a tile with r_{k+1} never leaving the CU:
//   workgroup = W consecutive grid lines (one per wave) x one segment of x x a block of T planes, walked upwards plane by plane;
//   step z:  phase 1  every wave computes sweep k (the x-untouched form, XM = 1) for its line of plane z from r_k in memory and puts the
//                     line of r_{k+1} into a ring of four planes in LDS (zeros where the grid has no node: absent taps need finite values)
//            barrier
//            phase 2  the W - 2 inner waves compute sweep k + 1 (the form that updates x with both increments, XM = 2) for their line of
//                     plane z - 1 from the three planes z - 2, z - 1, z of the ring and store r_{k+2} and x.
// The rim lines (first / last wave), the planes below / above the block and one node left / right of an x segment are computed
// redundantly by the neighbouring workgroups -- W / (W - 2) x (T + 2) / T more arithmetic on sweep k, none on sweep k + 1 -- and per pair
// of sweeps memory sees r_k in, r_{k+2} out, x in / out: 32 bytes per row instead of 16 + 40.
// A ring of FOUR planes needs one barrier per step: phase 1 of step z + 1 writes the slot phase 2 of step z does not read.
// Every row is summed exactly as sells_r2sweep_kernel / sells_zsweep_kernel sum it (same taps, same order, same strict-mask rule; the
// rim rows are recomputed from the same inputs), so the pair is bit-identical to the two single sweeps -- except that a tap the row does
// not store may meet 0.0 instead of a neighbouring line's value: its product is an exact zero either way.
//   a.x = r_k ; a.y = r_{k+2} ; a.x2 = x (in / out) ; a.pdinv[0] = d ; a.x_zero: x is zero on entry
// whole != 0 (L <= 127): one wave holds a whole grid line, lane l rows 2 l, 2 l + 1 in both sweeps.  Otherwise a segment of xlen <= 124
// rows: sweep k on the segment and one node either side (lane l rows x0 - 1 + 2 l, x0 + 2 l; lane 63 only supplies windows), sweep k + 1
// on rows x0 + 2 l, x0 + 2 l + 1.  The ring stores position p of a line at p - (first row of sweep k + 1) + 1.
// ---------------------------------------------------------------------------
// BC ("box, constant"): on a constant-coefficient operator every row of a grid line stores the SAME 27 numbers -- those of the line's
// class (first / inner / last line of a plane) x (first / inner / last plane) -- except that the first / last row of the line does not
// store the taps that would leave the line.  With exact zeros at those positions (the ring's pads in phase 2, a select on the window
// values in phase 1) one set of coefficients serves the whole wave, and it comes out of the KERNEL ARGUMENTS through scalar loads
// instead of 27 LDS reads per row: the sweeps of a row-pattern level are bound by LDS bandwidth (54 x 8 B per lane and step against
// ~150 vector instructions), not by memory.  gmg_solver::z2_geo verifies the property row by row on the device (z2_box_check_kernel)
// before it sets `box`; the sums are those of the general form except for the sign of an exact zero product.
struct Z2Geo {
  int P, L, ny, nz;       // rows per plane, per line ; lines per plane ; planes
  int W, T;               // waves per workgroup (lines incl. the two rim lines) ; planes per block
  int nyt, nzb, nxs;      // tiles in y, blocks in z, segments in x
  int xlen, whole;
  int box;                // 1: coef / cmask below describe every row (BC kernels)
  double coef[9][27];     // class c = 3 * (plane class) + (line class), class 0 first / 1 inner / 2 last ; entry j = 3 * run + tap
  uint32_t cmask[9][27];  // 0xffffffff stored, 0 absent
};
constexpr int kZ2Slot = 130;     // doubles per line slot of the ring (128 rows + one either side)

// bad[0] != 0 afterwards: some row's pattern is not the class pattern with the taps that leave the line removed
__global__ void z2_box_check_kernel(int64_t nrows, int L, int ny, int nz, int P, const uint16_t *__restrict__ rowpid, const PatEntry *__restrict__ tab,
                                    const double *__restrict__ coef, const uint32_t *__restrict__ cmask, int *__restrict__ bad)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += (int64_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % L), y = (int)((i / L) % ny), z = (int)(i / P);
    const int c = 3 * (z == 0 ? 0 : z == nz - 1 ? 2 : 1) + (y == 0 ? 0 : y == ny - 1 ? 2 : 1);
    const PatEntry *te = tab + (size_t)rowpid[i] * 27;
    bool ok = true;
    for (int j = 0; j < 27; ++j) {
      const int dx = j % 3 - 1;
      const bool inside = x + dx >= 0 && x + dx < L;
      const unsigned long long want = inside ? (unsigned long long)__double_as_longlong(coef[c * 27 + j]) : 0ull;   // absent entries hold +0.0
      const uint32_t wm = inside ? cmask[c * 27 + j] : 0u;
      ok = ok && (unsigned long long)__double_as_longlong(te[j].v) == want && te[j].m == wm;
    }
    if (!ok) atomicOr(bad, 1);
  }
}

template <bool MK, bool FM, bool BC = false>
__global__ __launch_bounds__(1024) void sells_z2sweep_kernel(SellSArgs a, Z2Geo g)
{
  constexpr int K = 3, NR = 9, nu = K * NR, NUP = 28, SL = kZ2Slot;
  extern __shared__ double sp_smem[];
  const int tot = BC ? 0 : a.np * NUP;
  double *s_tab8 = sp_smem;                                   // [np*NUP] coefficients (absent entries 0.0); the masks stay in global memory
  double *ring = sp_smem + ((tot + 1) & ~1);                  // [4][W][SL] r_{k+1}
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = g.W;
  const int blk = remap_block(blockIdx.x, gridDim.x, a.xcd_remap);
  const int xs = blk % g.nxs, t2 = blk / g.nxs, yt = t2 % g.nyt, zb = t2 / g.nyt;
  const int y = yt * (W - 2) - 1 + wave;
  const bool vline = y >= 0 && y < g.ny;
  const bool inner = wave >= 1 && wave <= W - 2 && vline;
  const int x0 = g.whole ? 0 : xs * g.xlen;
  const int len = g.whole ? g.L : min(g.xlen, g.L - x0);      // rows of sweep k + 1 in this segment
  const int ka = g.whole ? 0 : x0 - 1;                        // x of lane 0's first row in sweep k
  const int klo = g.whole ? 0 : max(0, x0 - 1), khi = g.whole ? g.L : min(g.L, x0 + len + 1);
  const int xk = ka + 2 * lane;
  const bool kvA = xk >= klo && xk < khi, kvB = xk + 1 >= klo && xk + 1 < khi;
  const int xb = x0 + 2 * lane;
  const bool nvA = xb < x0 + len, nvB = xb + 1 < x0 + len;
  const int sidx = g.whole ? 2 * lane + 1 : 2 * lane;         // where the lane's two rows of sweep k sit in a line slot
  const double *__restrict__ rg = a.x;
  const double omega = a.omega, du = a.pdinv[0];
  const int last = (int)a.ncols - 1, lastrow = (int)a.nrows - 1;
  const bool xz = a.x_zero != 0;
  int roff[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) roff[q] = __builtin_amdgcn_readfirstlane(a.run_off[q]);
  auto conv = [&](gmg_d2 v) -> gmg_d2 { return gmg_d2{omega * (du * v.x), omega * (du * v.y)}; };   // s = omega * (Dinv * r)
  // the 27 taps of the lane's two rows from nine converted windows: w0, w1 its own pair, w2, w3 = lane l + 1's (lane 63: 0.0)
  // (BC: cls = the class of the wave's line and the step's plane, wave-uniform -- the coefficients are scalar loads from the arguments)
  auto taps = [&](const gmg_d2 *C, const double *tvA, const double *tvB, int cls, double &sA, double &sB) {
    sA = 0.0; sB = 0.0;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const double w0 = C[q].x, w1 = C[q].y;
      const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
      const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
      for (int t = 0; t < K; ++t) {
        const double ca = BC ? g.coef[cls][q * K + t] : tvA[q * K + t], cb = BC ? ca : tvB[q * K + t];
        sA = FM ? __builtin_fma(ca, wa[t], sA) : sA + ca * wa[t];
        sB = FM ? __builtin_fma(cb, wb[t], sB) : sB + cb * wb[t];
      }
    }
  };
  // strict form: a sum that is not finite is redone with the masks (a vector that already holds Inf / NaN: rare).  The windows are
  // fetched again, run by run (getw(q): from memory in phase 1, from the ring in phase 2) -- indexing the register array of the fast
  // path with a loop counter would move it into scratch memory for the whole kernel
  auto taps_masked = [&](auto getw, int pidA, int pidB, int cls, double &sA, double &sB) {
    sA = 0.0; sB = 0.0;
    const double *tvA = s_tab8 + pidA * NUP, *tvB = s_tab8 + pidB * NUP;
#pragma unroll 1
    for (int q = 0; q < NR; ++q) {
      const gmg_d2 c = conv(getw(q));
      const double w0 = c.x, w1 = c.y;
      const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
      const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
      for (int t = 0; t < K; ++t) {
        const int j = q * K + t;
        const int ma = BC ? (int)g.cmask[cls][j] : (int)a.tab[pidA * nu + j].m, mb = BC ? ma : (int)a.tab[pidB * nu + j].m;
        const double ca = BC ? g.coef[cls][j] : tvA[j], cb = BC ? ca : tvB[j];
        const double ga = __hiloint2double(__double2hiint(wa[t]) & ma, __double2loint(wa[t]));
        const double gb = __hiloint2double(__double2hiint(wb[t]) & mb, __double2loint(wb[t]));
        sA = FM ? __builtin_fma(ca, ga, sA) : sA + ca * ga;
        sB = FM ? __builtin_fma(cb, gb, sB) : sB + cb * gb;
      }
    }
  };
  // BC, phase 1: window positions that leave the grid line read as exact zeros (lane l holds positions ka - 1 + 2 l and ka + 2 l)
  const bool zx0 = BC && (ka - 1 + 2 * lane < 0 || ka - 1 + 2 * lane >= g.L), zx1 = BC && (ka + 2 * lane < 0 || ka + 2 * lane >= g.L);
  const bool zx_any = BC && __any(zx0 || zx1);
  const int cy = y <= 0 ? 0 : (y >= g.ny - 1 ? 2 : 1);
  if (!BC)
    for (int i = threadIdx.x; i < tot; i += blockDim.x) {
      const int pq = i / NUP, j = i - pq * NUP;
      s_tab8[i] = j < nu ? (a.tab8 ? a.tab8[pq * nu + j] : a.tab[pq * nu + j].v) : 0.0;
    }
  for (int i = threadIdx.x; i < 4 * W * SL; i += blockDim.x) ring[i] = 0.0;
  __syncthreads();
  const int zb0 = zb * g.T, zb1 = min(g.nz, zb0 + g.T);
  typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));
  // Requests run ahead of the arithmetic (as in sells_zsweep_kernel): the nine windows of sweep k live in registers, converted, and walk
  // up the planes with the wave -- a step fetches only the three windows of the plane above, and it asks for them, for the rows' own
  // r_k and for the operands of phase 2 (x, r_k of plane z - 1) BEFORE the arithmetic of the step, so that a workgroup whose waves
  // all sit in the same phase (one barrier per step) still has its memory requests in flight behind ~300 vector instructions.
  // Every address is clamped into the vectors: one control path, the same number of requests in the first and last planes.
  auto loadw = [&](int plane, int q) -> gmg_d2 {                // window q of the lane's rows taken at `plane` (any integer: clamped)
    const int c = plane * g.P + y * g.L + ka + roff[q] + 2 * lane;
    const gmg_d2 v = gmg_d2{rg[min(max(c, 0), last)], rg[min(max(c + 1, 0), last)]};
    return zx_any ? gmg_d2{zx0 ? 0.0 : v.x, zx1 ? 0.0 : v.y} : v;
  };
  struct Own { gmg_d2 e0; int pidA, pidB; };
  auto load_own = [&](int plane) -> Own {                       // sweep k: the rows' own r_k (and pattern ids)
    Own o;
    const int row = plane * g.P + y * g.L + ka + 2 * lane;
    const int ra = min(max(row, 0), lastrow), rb = min(max(row + 1, 0), lastrow);
    o.e0 = gmg_d2{rg[ra], rg[rb]};
    o.pidA = BC ? 0 : (int)a.rowpid[ra]; o.pidB = BC ? 0 : (int)a.rowpid[rb];
    return o;
  };
  struct Ops2 { gmg_d2 rp, e2; int pidA, pidB; };
  auto load_ops2 = [&](int plane) -> Ops2 {                     // sweep k + 1: r_k and x of the rows, pattern ids
    Ops2 o;
    const int row = plane * g.P + y * g.L + x0 + 2 * lane;
    const int ra = min(max(row, 0), lastrow), rb = min(max(row + 1, 0), lastrow);
    o.rp = gmg_d2{rg[ra], rg[rb]};
    o.e2 = gmg_d2{a.x2[ra], a.x2[rb]};
    o.pidA = BC ? 0 : (int)a.rowpid[ra]; o.pidB = BC ? 0 : (int)a.rowpid[rb];
    return o;
  };
  gmg_d2 C[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) C[q] = conv(loadw(zb0 - 1, q));
  Own cur = load_own(zb0 - 1);
#pragma unroll 1
  for (int z = zb0 - 1; z <= zb1; ++z) {
    // requests of this step: the three windows of plane z + 2 (run 6..8 of step z + 1), the rows' own r_k of plane z + 1, phase 2's operands
    gmg_d2 N[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) N[q] = loadw(z + 1, 6 + q);
    const Own nxt = load_own(z + 1);
    const int zz = z - 1;
    const Ops2 o2 = load_ops2(zz);
    // ---- phase 1: sweep k, plane z, this wave's line -> ring slot z & 3
    gmg_d2 rk1 = gmg_d2{0.0, 0.0};
    if (vline && z >= 0 && z < g.nz) {                        // (wave-uniform)
      const int cls1 = 3 * (z == 0 ? 0 : (z == g.nz - 1 ? 2 : 1)) + cy;
      double sA, sB;
      taps(C, s_tab8 + cur.pidA * NUP, s_tab8 + cur.pidB * NUP, cls1, sA, sB);
      if (MK && !__all(__builtin_isfinite(sA) && __builtin_isfinite(sB)))
        taps_masked([&](int q) -> gmg_d2 { const int c = z * g.P + y * g.L + ka + a.run_off[q] + 2 * lane;
                                           const gmg_d2 v = gmg_d2{rg[min(max(c, 0), last)], rg[min(max(c + 1, 0), last)]};
                                           return gmg_d2{zx0 ? 0.0 : v.x, zx1 ? 0.0 : v.y}; },
                    cur.pidA, cur.pidB, cls1, sA, sB);
      rk1 = gmg_d2{kvA ? cur.e0.x - sA : 0.0, kvB ? cur.e0.y - sB : 0.0};
    }
    {
      double *slot = ring + ((z & 3) * W + wave) * SL;
      slot[sidx] = rk1.x; slot[sidx + 1] = rk1.y;
    }
    __syncthreads();
    // ---- phase 2: sweep k + 1, plane z - 1, inner lines, from the ring
    if (inner && zz >= zb0 && zz < zb1) {                     // (wave-uniform)
      const int cls2 = 3 * (zz == 0 ? 0 : (zz == g.nz - 1 ? 2 : 1)) + cy;
      const double *tvA = s_tab8 + o2.pidA * NUP, *tvB = s_tab8 + o2.pidB * NUP;
      double sA = 0.0, sB = 0.0;
      gmg_d2 raw4 = gmg_d2{0.0, 0.0};
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        const int dz = q / 3 - 1, dy = q % 3 - 1;
        const double *sl = ring + (((zz + dz) & 3) * W + wave + dy) * SL;
        const gmg_d2 v = *reinterpret_cast<const d2a *>(sl + 2 * lane);
        if (q == 4) raw4 = v;
        const gmg_d2 c = conv(v);
        const double w0 = c.x, w1 = c.y;
        const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
        const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const double ca = BC ? g.coef[cls2][q * K + t] : tvA[q * K + t], cb = BC ? ca : tvB[q * K + t];
          sA = FM ? __builtin_fma(ca, wa[t], sA) : sA + ca * wa[t];
          sB = FM ? __builtin_fma(cb, wb[t], sB) : sB + cb * wb[t];
        }
      }
      if (MK && !__all(__builtin_isfinite(sA) && __builtin_isfinite(sB)))
        taps_masked([&](int q) -> gmg_d2 { const double *sl = ring + (((zz + q / 3 - 1) & 3) * W + wave + q % 3 - 1) * SL;
                                           return *reinterpret_cast<const d2a *>(sl + 2 * lane); }, o2.pidA, o2.pidB, cls2, sA, sB);
      const gmg_d2 e0 = gmg_d2{raw4.y, wave_shl1(raw4.x)};     // the rows' own r_{k+1}: positions x, x + 1 of the centre window
      const gmg_d2 e2 = xz ? gmg_d2{0.0, 0.0} : o2.e2;
      const gmg_d2 rn = gmg_d2{e0.x - sA, e0.y - sB};
      const gmg_d2 sk = gmg_d2{omega * (du * e0.x), omega * (du * e0.y)};
      const gmg_d2 xn = gmg_d2{(e2.x + omega * (du * o2.rp.x)) + sk.x, (e2.y + omega * (du * o2.rp.y)) + sk.y};
      const int row = zz * g.P + y * g.L + x0 + 2 * lane;
      if (nvA) { a.x2[row] = xn.x; a.y[row] = rn.x; }
      if (nvB) { a.x2[row + 1] = xn.y; a.y[row + 1] = rn.y; }
    }
    // the walk: plane z's windows 3..8 are plane z + 1's windows 0..5
#pragma unroll
    for (int q = 0; q < 6; ++q) C[q] = C[q + 3];
#pragma unroll
    for (int q = 0; q < 3; ++q) C[6 + q] = conv(N[q]);
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------
// The single sweep of a constant-coefficient box (round 6): what phase 1 of sells_z2sweep_kernel<.., BC> does, without the ring, the
// barrier and the rim rows.  One wave = one grid line (a segment of <= 126 rows of it) walked up T planes: nine converted windows in
// registers, three new ones per step requested a step ahead, the 27 coefficients of the line's class from the KERNEL ARGUMENTS (scalar
// loads) -- no pattern ids, no coefficient table in LDS, no LDS at all.  Window positions that leave the line read as exact zeros, so
// the line's first and last rows use the same coefficients as the rest.  Sums as in sells_r2sweep_kernel / sells_zsweep_kernel (same
// taps, same order) up to the sign of an exact zero product; strict form: cmask of the class.
//   a.x = r_k ; a.y = r_{k+1} ; a.x2 = x ; a.s_out = r_{k-1} (XM = 2) ; a.pdinv[0] = d ; g.T = planes per chain, g.W unused
// ---------------------------------------------------------------------------
template <int XM, bool MK, bool FM>
__global__ __launch_bounds__(kBlock, 4) void sells_boxsweep_kernel(SellSArgs a, Z2Geo g)
{
  constexpr int K = 3, NR = 9;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int blk = remap_block(blockIdx.x, gridDim.x, a.xcd_remap);
  const int chain = blk * wpb + wave;                         // (zb, y, xs), xs fastest: neighbouring waves hold neighbouring lines of the same planes
  const int per_zb = g.ny * g.nxs;
  if (chain >= g.nzb * per_zb) return;
  const int zb = chain / per_zb, c2 = chain - zb * per_zb, y = c2 / g.nxs, xs = c2 - y * g.nxs;
  const int x0 = g.whole ? 0 : xs * g.xlen;
  const int len = g.whole ? g.L : min(g.xlen, g.L - x0);
  const double *__restrict__ rg = a.x;
  const double omega = a.omega, du = a.pdinv[0];
  const int last = (int)a.ncols - 1, lastrow = (int)a.nrows - 1;
  const bool xz = a.x_zero != 0;
  int roff[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) roff[q] = __builtin_amdgcn_readfirstlane(a.run_off[q]);
  auto conv = [&](gmg_d2 v) -> gmg_d2 { return gmg_d2{omega * (du * v.x), omega * (du * v.y)}; };
  // lane l: rows x0 + 2 l, x0 + 2 l + 1 ; window positions x0 - 1 + 2 l, x0 + 2 l (lane 63 of a segment only supplies windows)
  const int xr = x0 + 2 * lane;
  const bool vA = xr < x0 + len, vB = xr + 1 < x0 + len;
  const bool zx0 = x0 - 1 + 2 * lane < 0 || x0 - 1 + 2 * lane >= g.L, zx1 = x0 + 2 * lane >= g.L;
  const bool zx_any = __any(zx0 || zx1);
  const int cy = y <= 0 ? 0 : (y >= g.ny - 1 ? 2 : 1);
  const int z0 = zb * g.T, z1 = min(g.nz, z0 + g.T);
  const int lbase = y * g.L + x0;
  auto loadw = [&](int plane, int q) -> gmg_d2 {
    const int c = plane * g.P + lbase + roff[q] + 2 * lane;
    const gmg_d2 v = gmg_d2{rg[min(max(c, 0), last)], rg[min(max(c + 1, 0), last)]};
    return zx_any ? gmg_d2{zx0 ? 0.0 : v.x, zx1 ? 0.0 : v.y} : v;
  };
  struct Own { gmg_d2 e0, e2, rp; };
  auto load_own = [&](int plane) -> Own {
    Own o;
    const int row = plane * g.P + lbase + 2 * lane;
    const int ra = min(max(row, 0), lastrow), rb = min(max(row + 1, 0), lastrow);
    o.e0 = gmg_d2{rg[ra], rg[rb]};
    o.e2 = gmg_d2{0.0, 0.0}; o.rp = gmg_d2{0.0, 0.0};
    if (XM != 1) o.e2 = gmg_d2{a.x2[ra], a.x2[rb]};
    if (XM == 2) o.rp = gmg_d2{a.s_out[ra], a.s_out[rb]};
    return o;
  };
  gmg_d2 C[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) C[q] = conv(loadw(z0, q));
  Own cur = load_own(z0);
#pragma unroll 1
  for (int z = z0; z < z1; ++z) {
    gmg_d2 N[3];
    const int zn = min(z + 1, z1 - 1);                        // (past the end of the chain: the last step again, unused)
#pragma unroll
    for (int q = 0; q < 3; ++q) N[q] = loadw(zn, 6 + q);
    const Own nxt = load_own(zn);
    const int cls = 3 * (z == 0 ? 0 : (z == g.nz - 1 ? 2 : 1)) + cy;
    double sA = 0.0, sB = 0.0;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      const double w0 = C[q].x, w1 = C[q].y;
      const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
      const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
      for (int t = 0; t < K; ++t) {
        const double c = g.coef[cls][q * K + t];
        sA = FM ? __builtin_fma(c, wa[t], sA) : sA + c * wa[t];
        sB = FM ? __builtin_fma(c, wb[t], sB) : sB + c * wb[t];
      }
    }
    if (MK && !__all(__builtin_isfinite(sA) && __builtin_isfinite(sB))) {
      sA = 0.0; sB = 0.0;
#pragma unroll 1
      for (int q = 0; q < NR; ++q) {
        const int cc = z * g.P + lbase + a.run_off[q] + 2 * lane;
        gmg_d2 v = gmg_d2{rg[min(max(cc, 0), last)], rg[min(max(cc + 1, 0), last)]};
        v = conv(gmg_d2{zx0 ? 0.0 : v.x, zx1 ? 0.0 : v.y});
        const double w0 = v.x, w1 = v.y;
        const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
        const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const int j = q * K + t;
          const int m = (int)g.cmask[cls][j];
          const double c = g.coef[cls][j];
          const double ga = __hiloint2double(__double2hiint(wa[t]) & m, __double2loint(wa[t]));
          const double gb = __hiloint2double(__double2hiint(wb[t]) & m, __double2loint(wb[t]));
          sA = FM ? __builtin_fma(c, ga, sA) : sA + c * ga;
          sB = FM ? __builtin_fma(c, gb, sB) : sB + c * gb;
        }
      }
    }
    const gmg_d2 rn = gmg_d2{cur.e0.x - sA, cur.e0.y - sB};
    const gmg_d2 sk = gmg_d2{omega * (du * cur.e0.x), omega * (du * cur.e0.y)};
    const gmg_d2 e2 = xz ? gmg_d2{0.0, 0.0} : cur.e2;
    gmg_d2 xn = gmg_d2{0.0, 0.0};
    if (XM == 0) xn = gmg_d2{e2.x + sk.x, e2.y + sk.y};
    else if (XM == 2) xn = gmg_d2{(e2.x + omega * (du * cur.rp.x)) + sk.x, (e2.y + omega * (du * cur.rp.y)) + sk.y};
    const int row = z * g.P + lbase + 2 * lane;
    if (vA) { if (XM != 1) a.x2[row] = xn.x; a.y[row] = rn.x; }
    if (vB) { if (XM != 1) a.x2[row + 1] = xn.y; a.y[row + 1] = rn.y; }
#pragma unroll
    for (int q = 0; q < 6; ++q) C[q] = C[q + 3];
#pragma unroll
    for (int q = 0; q < 3; ++q) C[6 + q] = conv(N[q]);
    cur = nxt;
  }
}

// ---------------------------------------------------------------------------
// Wide rows (Q2: 25 runs of 5 consecutive offsets, 125 entries per row) in the z-walk form (round 5).  The runs are a 5 x 5 grid,
// run q = 5 (dz + 2) + (dy + 2) at dz P + dy L - 2, so 20 of the 25 windows of the slice at r0 + P are windows of the slice at r0:
// a wave keeps an interval of <= 60 rows of a grid plane, walks T planes upwards and gathers FIVE new windows per step instead of
// up to 25 (sells_kernel<..., WL>: 16 gathers per slice on average -- at 8 bytes per lane the texture-address path ran as long as
// the multiply / add stream, and the two added up).  One row per lane, the 25 windows in registers; the per-workgroup value tables
// of the wide-row kernel (the patterns a workgroup's four chains touch, decoded into LDS; absent entries are 0.0) and its skipping
// of runs no row of the wave has, by grid plane: the five runs of a dz are skipped when absent for the whole wave, and cut to the
// middle three when the outer two are (a line of Q2 mid-edge dofs).  Requests run one step ahead on a single control path (every
// address clamped: the same instruction count in the first / last planes as inside), results are stored one step late.
// Rows are summed over the runs in ascending order with exact-zero products for absent entries, and a step whose sums are not all
// finite is redone from memory with the masks of the coded table: bit-identical to sells_kernel<EPI, false, 5, 5, true, 0, true>.
//   a.x gathered ; a.y result (SUB: also read) ; a.b (RESID) ; a.x2 (ADDTO: x2 += omega * A x, y = omega * A x)
// ---------------------------------------------------------------------------
// FM: fused multiply-add taps (option pat_fma: one rounding per tap -- not the reference's mul!, inside the parity gates)
template <int EPI, bool FM = false>
__global__ __launch_bounds__(kBlock, 4) void sellw_zwalk_kernel(SellSArgs a, ZWalkGeo g)
{
  static_assert(EPI == EPI_SET || EPI == EPI_SUB || EPI == EPI_RESID || EPI == EPI_ADDTO, "operator applications only");
  constexpr int K = 5, NR = 25, nu = K * NR, NUT = nu + 1;    // NUT: LDS stride of a pattern (even: 16-byte aligned rows)
  extern __shared__ double sp_smem[];
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int blk = remap_block(blockIdx.x, gridDim.x, a.xcd_remap);
  const int chain = blk * wpb + wave;
  const double *__restrict__ xg = a.x;
  const int last = (int)a.ncols - 1;
  const int lastrow = (int)a.nrows - 1;
  // [wl_max * NUT] values | [wl_max] run masks | [np] pattern id -> local id
  uint32_t *s_rmask = reinterpret_cast<uint32_t *>(sp_smem + (size_t)a.wl_max * NUT);
  uint8_t *s_map = reinterpret_cast<uint8_t *>(s_rmask + a.wl_max);
  {
    const int lnp = a.wl_cnt[blk];
    const uint16_t *mine = a.wl_pids + (size_t)blk * a.wl_stride;
    for (int i = threadIdx.x; i < lnp; i += blockDim.x) {
      const int p = mine[i];
      s_map[p] = (uint8_t)i;
      s_rmask[i] = a.runmask[p];
    }
    for (int i = threadIdx.x; i < lnp * NUT; i += blockDim.x) {
      const int l = i / NUT, e = i - l * NUT;
      sp_smem[i] = e < nu ? a.dict[a.codes16 ? (int)a.codes16[(size_t)mine[l] * nu + e] : (int)a.codes[(size_t)mine[l] * nu + e]] : 0.0;
    }
  }
  __syncthreads();
  if (chain >= g.nchains) return;
  int roff[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) roff[q] = __builtin_amdgcn_readfirstlane(a.run_off[q]);
  const int zb = chain / g.m, iv = chain - zb * g.m;
  const int b0 = (int)(((int64_t)iv * g.P) / g.m), b1 = (int)(((int64_t)(iv + 1) * g.P) / g.m);
  const int len = b1 - b0;                                    // <= 60
  const int z0 = zb * g.T, z1 = min(g.nplanes, z0 + g.T);
  const int nsteps = z1 - z0;
  if (nsteps <= 0) return;
  const int rfirst = z0 * g.P + b0;
  const double *__restrict__ eg = EPI == EPI_RESID ? a.b : (EPI == EPI_ADDTO ? a.x2 : a.y);
  auto loadw = [&](int base) -> double { return xg[min(max(base + lane, 0), last)]; };
  struct RowOps { int pid; double e0; };
  auto load_rows = [&](int rbase) -> RowOps {
    RowOps o;
    const int r = min(rbase + lane, lastrow);
    o.pid = (int)a.rowpid[r];
    o.e0 = EPI == EPI_SET ? 0.0 : eg[r];
    return o;
  };
  // The 25 windows live in a ring of registers: logical window q of the step with rotation R is C[(q + K R) % NR].  The step loop is
  // unrolled over the five rotations, so "the windows move up one plane" costs nothing (shifting 20 doubles was 40 of the ~420 vector
  // instructions of a step); the five new windows are requested into their own registers and take the places of the oldest plane
  // group after the step.
  double C[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) C[q] = loadw(rfirst + roff[q]);
  RowOps cur = load_rows(rfirst);
  int r0 = rfirst;
  double p0 = 0.0, p1 = 0.0;                                   // results of the previous step (stored at the top of the next one)
  int prow = 0;
  bool pok = false;
  auto put = [&]() {
    if (pok) {
      a.y[prow] = p0;
      if (EPI == EPI_ADDTO) a.x2[prow] = p1;
    }
  };
  auto step = [&](auto rot_tag, int k) {
    constexpr int R = decltype(rot_tag)::value;
    put();
    // step k + 1 (the last step: this step again, unused): requested before the taps of step k
    const int rb = rfirst + min(k + 1, nsteps - 1) * g.P;
    double N[K];
#pragma unroll
    for (int q = 0; q < K; ++q) N[q] = loadw(rb + roff[NR - K + q]);
    const RowOps nxt = load_rows(rb);
    const int row = r0 + lane;
    const bool mine = lane < len && row <= lastrow;
    const int gpid = mine ? cur.pid : a.np - 1;                // halo lanes / rows of other chains: the empty pattern
    const int lid = (int)s_map[gpid];
    uint32_t m = s_rmask[lid];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m |= (uint32_t)__shfl_xor((int)m, o);
    const uint32_t M = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
    const double *tv = sp_smem + lid * NUT;
    double sum = 0.0;
    // the runs LO .. HI of plane group ZI: coefficients of the next run requested before the taps of the current one
    auto group = [&](auto zi_tag, auto lo_tag, auto hi_tag) {
      constexpr int ZI = decltype(zi_tag)::value, LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
      double cf[2][K];
      // the five coefficients of a run: two 16-byte reads + one 8-byte read (a pattern's table row is 16-byte aligned, a run starts at
      // 40 q bytes: the pairs sit at even indices for even q, at odd ones for odd q) -- 10 LDS cycles instead of the 18 of two
      // ds_read2_b64 + one ds_read_b64
      auto rd = [&](auto q_tag, double *out) {
        constexpr int Q = decltype(q_tag)::value;
        typedef double d2a __attribute__((ext_vector_type(2), aligned(16)));
        const double *tr = tv + Q * K;
        if ((Q & 1) == 0) {
          const d2a p0 = *reinterpret_cast<const d2a *>(tr), p1 = *reinterpret_cast<const d2a *>(tr + 2);
          out[0] = p0.x; out[1] = p0.y; out[2] = p1.x; out[3] = p1.y; out[4] = tr[4];
        } else {
          const d2a p0 = *reinterpret_cast<const d2a *>(tr + 1), p1 = *reinterpret_cast<const d2a *>(tr + 3);
          out[0] = tr[0]; out[1] = p0.x; out[2] = p0.y; out[3] = p1.x; out[4] = p1.y;
        }
      };
      rd(std::integral_constant<int, ZI * K + LO>{}, cf[LO & 1]);
#pragma unroll
      for (int y = LO; y <= HI; ++y) {
        if (y < HI) {
          if (y == 0) rd(std::integral_constant<int, ZI * K + 1>{}, cf[1]);
          else if (y == 1) rd(std::integral_constant<int, ZI * K + 2>{}, cf[0]);
          else if (y == 2) rd(std::integral_constant<int, ZI * K + 3>{}, cf[1]);
          else rd(std::integral_constant<int, ZI * K + 4>{}, cf[0]);
        }
        __builtin_amdgcn_sched_barrier(0);
        double v = C[((ZI + R) * K + y) % NR];
#pragma unroll
        for (int t = 0; t < K; ++t) {
          if (t > 0) v = wave_shl1(v);
          sum = FM ? __builtin_fma(cf[y & 1][t], v, sum) : sum + cf[y & 1][t] * v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto plane = [&](auto zi_tag) {
      constexpr int ZI = decltype(zi_tag)::value;
      const uint32_t gm = (M >> (ZI * K)) & 31u;              // wave-uniform
      if (gm == 0u) return;
      if ((gm & 0x11u) == 0u) group(zi_tag, std::integral_constant<int, 1>{}, std::integral_constant<int, 3>{});
      else group(zi_tag, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
    };
    plane(std::integral_constant<int, 0>{});
    plane(std::integral_constant<int, 1>{});
    plane(std::integral_constant<int, 2>{});
    plane(std::integral_constant<int, 3>{});
    plane(std::integral_constant<int, 4>{});
    // "all sums finite" proves that no mask was needed (absent entries are exact zeros against finite values); otherwise -- a vector
    // that already holds Inf / NaN -- the step is redone from memory with the masks of the coded table
    if (!__all(__builtin_isfinite(sum))) {
      sum = 0.0;
      const size_t gb = (size_t)gpid * nu;
#pragma unroll 1
      for (int q = 0; q < NR; ++q) {
        if (!((M >> q) & 1u)) continue;
        double v = xg[min(max(row + roff[q], 0), last)];
#pragma unroll 1
        for (int t = 0; t < K; ++t) {
          if (t > 0) v = wave_shl1(v);
          const bool absent = a.codes16 ? a.codes16[gb + q * K + t] == 65535 : a.codes[gb + q * K + t] == 255;
          const double gv = __hiloint2double(__double2hiint(v) & (absent ? 0 : -1), __double2loint(v));
          sum = FM ? __builtin_fma(tv[q * K + t], gv, sum) : sum + tv[q * K + t] * gv;
        }
      }
    }
    if (EPI == EPI_SET) p0 = sum;
    else if (EPI == EPI_SUB || EPI == EPI_RESID) p0 = cur.e0 - sum;
    else { const double t = a.omega != 0.0 ? a.omega * sum : sum; p0 = t; p1 = cur.e0 + t; }
    prow = row; pok = mine;
    // the oldest plane group (logical runs 0 .. 4 of this step) makes room for the new one (logical 20 .. 24 of the next step)
#pragma unroll
    for (int q = 0; q < K; ++q) C[(R * K + q) % NR] = N[q];
    cur = nxt;
    r0 += g.P;
  };
#pragma unroll 1
  for (int k = 0; k < nsteps; k += K) {
    step(std::integral_constant<int, 0>{}, k);
    if (k + 1 < nsteps) step(std::integral_constant<int, 1>{}, k + 1);
    if (k + 2 < nsteps) step(std::integral_constant<int, 2>{}, k + 2);
    if (k + 3 < nsteps) step(std::integral_constant<int, 3>{}, k + 3);
    if (k + 4 < nsteps) step(std::integral_constant<int, 4>{}, k + 4);
  }
  put();
}

// the patterns the four chains of every workgroup of sellw_zwalk_kernel touch (one list per workgroup, ascending; the empty pattern
// np - 1 is always in it)
__global__ __launch_bounds__(256) void sellwz_patterns_kernel(const uint16_t *__restrict__ rowpid, int64_t nrows, ZWalkGeo g, int wpb, int np, int stride,
                                                               uint16_t *__restrict__ pids, int32_t *__restrict__ cnt)
{
  __shared__ uint32_t bits[128];                             // np <= 4096
  const int blk = blockIdx.x;
  for (int i = threadIdx.x; i < 128; i += blockDim.x) bits[i] = 0u;
  __syncthreads();
  for (int w = 0; w < wpb; ++w) {
    const int chain = blk * wpb + w;
    if (chain >= g.nchains) break;
    const int zb = chain / g.m, iv = chain - zb * g.m;
    const int b0 = (int)(((int64_t)iv * g.P) / g.m), b1 = (int)(((int64_t)(iv + 1) * g.P) / g.m);
    const int len = b1 - b0;
    const int z0 = zb * g.T, z1 = min(g.nplanes, z0 + g.T);
    const int tot = (z1 - z0) * len;
    for (int i = threadIdx.x; i < tot; i += blockDim.x) {
      const int z = z0 + i / len, j = i - (i / len) * len;
      const int64_t r = (int64_t)z * g.P + b0 + j;
      if (r < nrows) {
        const int p = rowpid[r];
        atomicOr(&bits[p >> 5], 1u << (p & 31));
      }
    }
  }
  if (threadIdx.x == 0) atomicOr(&bits[(np - 1) >> 5], 1u << ((np - 1) & 31));
  __syncthreads();
  if (threadIdx.x == 0) {
    int c = 0;
    for (int w = 0; w < (np + 31) / 32; ++w)
      for (uint32_t m = bits[w]; m; m &= m - 1) {
        if (c < stride) pids[(size_t)blk * stride + c] = (uint16_t)(w * 32 + __builtin_ctz(m));
        ++c;
      }
    cnt[blk] = c;
  }
}

// ---------------------------------------------------------------------------
// The operator mat-vecs of a row-pattern level with two rows per lane: y = A x (EPI_SET: CGSolvers.jl:104), y -= A x (EPI_SUB:
// GMGLinearSolvers.jl:495), y = b - A x (EPI_RESID: CGSolvers.jl:79).  The layout, the windows and the strict-mask rule of
// sells_r2sweep_kernel without the omega*(d*.) of a sweep: the gathered value IS the window value.  Same taps in the same order as
// sells_kernel<EPI, false, 3>: bit-identical (FM = false).
//   a.x gathered ; a.y result (SUB: also read) ; a.b (RESID) ; a.nslices = ceil(nrows / 126)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double s, double *sh /*[4]*/);
// DOT (EPI_SET): also the first stage of dot(x, y) = dot(p, w) of CGSolvers.jl:105 -- every workgroup stores the sum over its rows in
// a.s_out[blockIdx.x] (the consumer sums the partials: sum_partials_all); p and w are not read again by a dot kernel.
// OCC = 2 (big levels, one slice per wave in workgroups of eight waves): the 64-register form -- rolled run loop -- of sells_r2sweep_kernel
template <int EPI, bool MK, bool FM, int NR, bool DOT = false, int OCC = 0>
__global__ __launch_bounds__(OCC == 2 ? 2 * kBlock : kBlock, OCC == 2 ? 4 : 1) void sells_r2mv_kernel(SellSArgs a)
{
  static_assert(EPI == EPI_SET || EPI == EPI_SUB || EPI == EPI_RESID, "mat-vec epilogues only");
  static_assert(!DOT || EPI == EPI_SET, "the fused dot is dot(x, A x)");
  double dsum = 0.0;
  constexpr int K = 3, ROWS2 = 126, RB = 3;
  extern __shared__ double sp_smem[];
  const int nu = K * NR;
  const int tot = a.np * nu;
  double *s_tab8 = sp_smem;
  uint32_t *s_msk = reinterpret_cast<uint32_t *>(sp_smem + tot);
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nwg = gridDim.x;
  const int blk = remap_block(blockIdx.x, nwg, a.xcd_remap);
  const int chunk_lo = a.nslices / nwg, chunk_rem = a.nslices % nwg;
  const int s_begin = blk * chunk_lo + min(blk, chunk_rem);
  const int s_end = s_begin + chunk_lo + (blk < chunk_rem ? 1 : 0);
  const double *__restrict__ xg = a.x;
  const double *__restrict__ eg = EPI == EPI_RESID ? a.b : a.y;
  const int last = (int)a.ncols - 1;
  const int lastrow = (int)a.nrows - 1;
  const int lo_need = -a.minoff, hi_need = a.maxoff + 2 * 64 + 2;
  int roff[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) roff[q] = __builtin_amdgcn_readfirstlane(a.run_off[q]);
  const int voff = a.run_off[min(lane, NR - 1)];             // OCC: lane q holds run_off[q] (rolled run loop)
  int pidA = 0, pidB = 0, row = 0;
  bool inner = false;
  gmg_d2 e0, A[RB];
  auto load_slice = [&](int slice) {
    const int r0 = slice * ROWS2;
    row = r0 + 2 * lane;
    inner = r0 >= lo_need && r0 + hi_need <= last && r0 + 2 * 64 <= lastrow;      // wave-uniform
    e0 = gmg_d2{0.0, 0.0};
    if (inner) {
      const uint32_t pp = *reinterpret_cast<const uint32_t *>(a.rowpid + row);
      pidA = (int)(pp & 0xffffu); pidB = (int)(pp >> 16);
      if (EPI != EPI_SET) e0 = ld2_unaligned(eg + row);
      else if (DOT) e0 = ld2_unaligned(xg + row);
#pragma unroll
      for (int q = 0; q < RB; ++q) A[q] = ld2_unaligned(xg + row + roff[q]);
    } else {
      const int ra = min(row, lastrow), rb = min(row + 1, lastrow);
      pidA = (int)a.rowpid[ra]; pidB = (int)a.rowpid[rb];
      if (EPI != EPI_SET) e0 = gmg_d2{eg[ra], eg[rb]};
      else if (DOT) e0 = gmg_d2{xg[ra], xg[rb]};
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const int c = row + roff[q];
        A[q] = gmg_d2{xg[min(max(c, 0), last)], xg[min(max(c + 1, 0), last)]};
      }
    }
  };
  int sb = s_begin + wave;
  if (sb < s_end) load_slice(sb);
  if (MK) { for (int i = threadIdx.x; i < tot; i += blockDim.x) { const PatEntry en = a.tab[i]; s_tab8[i] = en.v; s_msk[i] = en.m; } }
  else { for (int i = threadIdx.x; i < tot; i += blockDim.x) s_tab8[i] = a.tab8[i]; }
  __syncthreads();
  while (sb < s_end) {
    const uint32_t *tmA = s_msk + pidA * nu, *tmB = s_msk + pidB * nu;
    const double *tvA = s_tab8 + pidA * nu, *tvB = s_tab8 + pidB * nu;
    double sA = 0.0, sB = 0.0;
    if constexpr (OCC != 0) {
#pragma unroll 1
      for (int r0 = 0; r0 < NR; r0 += RB) {
        gmg_d2 cur[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) cur[q] = A[q];
        if (r0 + RB < NR) {
          if (inner) {
#pragma unroll
            for (int q = 0; q < RB; ++q) A[q] = ld2_unaligned(xg + row + __builtin_amdgcn_readlane(voff, r0 + RB + q));
          } else {
#pragma unroll
            for (int q = 0; q < RB; ++q) {
              const int c = row + __builtin_amdgcn_readlane(voff, r0 + RB + q);
              A[q] = gmg_d2{xg[min(max(c, 0), last)], xg[min(max(c + 1, 0), last)]};
            }
          }
        }
        const double *ta = tvA + r0 * K, *tb = tvB + r0 * K;
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          const double w0 = cur[q].x, w1 = cur[q].y;
          const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
          const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
          for (int t = 0; t < K; ++t) {
            const double ca = ta[q * K + t], cb = tb[q * K + t];
            sA = FM ? __builtin_fma(ca, wa[t], sA) : sA + ca * wa[t];
            sB = FM ? __builtin_fma(cb, wb[t], sB) : sB + cb * wb[t];
          }
        }
      }
    } else
#pragma unroll
    for (int r0 = 0; r0 < NR; r0 += RB) {
      gmg_d2 cur[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) cur[q] = A[q];
      if (r0 + RB < NR) {
        if (inner) {
#pragma unroll
          for (int q = 0; q < RB; ++q) A[q] = ld2_unaligned(xg + row + roff[r0 + RB + q]);
        } else {
#pragma unroll
          for (int q = 0; q < RB; ++q) {
            const int c = row + roff[r0 + RB + q];
            A[q] = gmg_d2{xg[min(max(c, 0), last)], xg[min(max(c + 1, 0), last)]};
          }
        }
      }
#pragma unroll
      for (int q = 0; q < RB; ++q) {
        const double w0 = cur[q].x, w1 = cur[q].y;
        const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
        const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const int j = (r0 + q) * K + t;
          const double ca = tvA[j], cb = tvB[j];
          sA = FM ? __builtin_fma(ca, wa[t], sA) : sA + ca * wa[t];
          sB = FM ? __builtin_fma(cb, wb[t], sB) : sB + cb * wb[t];
        }
      }
    }
    // strict form: see sells_r2sweep_kernel -- "all sums finite" proves that no mask was needed
    if (MK && !__all(__builtin_isfinite(sA) && __builtin_isfinite(sB))) {
      sA = 0.0; sB = 0.0;
#pragma unroll 1
      for (int q = 0; q < NR; ++q) {
        const int c = row + roff[q];
        const double w0 = xg[min(max(c, 0), last)], w1 = xg[min(max(c + 1, 0), last)];
        const double w2 = wave_shl1(w0), w3 = wave_shl1(w1);
        const double wa[3] = {w0, w1, w2}, wb[3] = {w1, w2, w3};
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const int j = q * K + t;
          const double ga = __hiloint2double(__double2hiint(wa[t]) & (int)tmA[j], __double2loint(wa[t]));
          const double gb = __hiloint2double(__double2hiint(wb[t]) & (int)tmB[j], __double2loint(wb[t]));
          sA = FM ? __builtin_fma(tvA[j], ga, sA) : sA + tvA[j] * ga;
          sB = FM ? __builtin_fma(tvB[j], gb, sB) : sB + tvB[j] * gb;
        }
      }
    }
    const gmg_d2 yn = EPI == EPI_SET ? gmg_d2{sA, sB} : gmg_d2{e0.x - sA, e0.y - sB};
    if (lane < 63) {
      if (inner) { st2_unaligned(a.y + row, yn); if (DOT) dsum += e0.x * yn.x + e0.y * yn.y; }
      else {
        if (row <= lastrow) { a.y[row] = yn.x; if (DOT) dsum += e0.x * yn.x; }
        if (row + 1 <= lastrow) { a.y[row + 1] = yn.y; if (DOT) dsum += e0.y * yn.y; }
      }
    }
    sb += wpb;
    if (sb < s_end) load_slice(sb);
  }
  if (DOT) {                                                 // blockDim.x == kBlock
    __shared__ double sh[4];
    const double t = block_sum(dsum, sh);
    if (threadIdx.x == 0) a.s_out[blockIdx.x] = t;
  }
}

// ---------------------------------------------------------------------------
// The r-gather sweep with the gathers shared through LDS ("tile sweep").  sells_rsweep_kernel fetches every r value nine times out of
// L2 (once per run: 3 x 3 neighbouring grid lines; only the three taps of a run are shared inside the wave) -- at 128^3 that is
// 150 MB of L2 -> L1 traffic per sweep on top of the 61 MB the sweep has to move, and neither fewer instructions nor another launch
// geometry changes its 21-22 us (profiles/r03_tuning.md section 10).  Here a workgroup of WPB waves takes TILES of T <= 3 WPB
// consecutive slices (62 T rows; T is chosen by the launcher so that the tiles fill whole rounds of resident workgroups).  The windows of all slices of a tile for one run are one contiguous stretch of r (62 T + 2 values),
// and the stretches of runs whose offsets differ by less than that -- the three lines of a plane -- overlap: the host merges them
// into segments (SellTile), the workgroup loads each segment ONCE, coalesced, converts it (s = omega*(d*r): two multiplies per loaded
// value instead of per gathered value) and keeps it in LDS; a slice then reads its nine windows from LDS.  2.7 x fewer values fetched
// at 128^3, none of them a gather.  Same taps in the same order on the same values: bit-identical to sells_rsweep_kernel.
// ---------------------------------------------------------------------------
struct SellTile {
  int T;                    // slices per tile
  int nseg;                 // merged stretches of r per tile
  int seg_off[12];          // first column of the stretch relative to the tile's first row
  int seg_len[12];
  int seg_base[12];         // its place in the LDS staging array (elements)
  int run_lds[32];          // window of run r for the tile's first slice: staging index of its lane 0
  int elems;                // staged values per tile
};

// BC: coefficients broadcast inside DPP rows.  27 of the 36 LDS reads of a slice are coefficient reads (216 of 290 B per row: at 288^3
// the LDS pipes are busy for 99 us of a 203 us sweep).  When the 16 lanes of every DPP row of the slice carry ONE pattern (no grid-line
// end inside the slice: 4 of 5 slices at 288^3) the lanes of a row load that pattern's 27 coefficients ONCE -- lane l the entries
// l % 16 and 16 + l % 16 -- and a tap takes its coefficient from lane j of the row with ONE `v_mov_b64_dpp row_newbcast:j` (gfx950 has
// no DPP form of the 64-bit multiply): a vector instruction on one of four SIMDs instead of a 512-byte read on the CU's one LDS pipe.
// Same coefficient values, same products, same order: bit-identical.  Other slices keep the per-lane LDS reads.
template <int J>
__device__ __forceinline__ double row_bcast(double v)        // lane J of the caller's DPP row (16 lanes) to every lane of that row
{
  double r;
  asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
  return r;
}
// the three taps of run Q of a 27-point slice with coefficients c0 (entries 0..15 in lanes 0..15 of every DPP row) and c1 (entries 16..26)
template <bool FM, int Q>
__device__ __forceinline__ double taps_bcast_run(double s, double c, double c0, double c1)
{
  constexpr int J0 = 3 * Q, J1 = 3 * Q + 1, J2 = 3 * Q + 2;
  const double k0 = row_bcast<J0 & 15>(J0 < 16 ? c0 : c1), k1 = row_bcast<J1 & 15>(J1 < 16 ? c0 : c1), k2 = row_bcast<J2 & 15>(J2 < 16 ? c0 : c1);
  s = FM ? __builtin_fma(k0, c, s) : s + k0 * c;
  c = wave_shl1(c);
  s = FM ? __builtin_fma(k1, c, s) : s + k1 * c;
  c = wave_shl1(c);
  s = FM ? __builtin_fma(k2, c, s) : s + k2 * c;
  return s;
}
template <int XM, bool MK, int WPB, bool FM = false, bool BC = false>
__global__ __launch_bounds__(64 * WPB, 8) void sells_tsweep_kernel(SellSArgs a, SellTile tl)   // 8 waves per SIMD = two resident workgroups of 16 waves per CU: <= 64 VGPRs
{
  constexpr int K = 3, ROWS = 65 - K, RB = 3, SPW = 3;       // T <= SPW * WPB slices per tile: slice j of the tile belongs to wave j % WPB
  const int T = tl.T;
  extern __shared__ double sp_smem[];
  const int nu = K * a.nruns;
  const int tot = a.np * nu;
  // LDS: [np*nu] coefficients, dense | strict form (MK): [np*nu] high-word masks (only read by batches holding a non-finite value) | stage
  double *s_tab8 = sp_smem;
  uint32_t *s_msk = reinterpret_cast<uint32_t *>(sp_smem + tot);
  double *s_stage = sp_smem + (MK ? 2 : 1) * (size_t)tot;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = (a.nslices + T - 1) / T;
  const double *__restrict__ rg = a.x;
  const double omega = a.omega;
  const double du = a.pdinv[0];
  const int last = (int)a.ncols - 1;
  const int lastrow = (int)a.nrows - 1;
  const bool xz = a.x_zero != 0;
  if (MK) { for (int i = threadIdx.x; i < tot; i += blockDim.x) { const PatEntry en = a.tab[i]; s_tab8[i] = en.v; s_msk[i] = en.m; } }
  else { for (int i = threadIdx.x; i < tot; i += blockDim.x) s_tab8[i] = a.tab8[i]; }
  // (requesting the next tile's stretches and row-wise operands a tile ahead was tried: the registers it takes leave ONE workgroup
  // per CU instead of two, and the sweep got slower -- 223 instead of 210 us at 288^3; two resident workgroups overlap their phases)
  // Workgroups are dealt to the eight XCDs round-robin; each XCD walks its own contiguous eighth of the tiles with all its workgroups
  // side by side, so the stretches of the planes above and below (the own stretches of tiles one plane away) are in that XCD's L2.
  // gridDim.x is a multiple of 8.
  const int wpx = gridDim.x >> 3, xcd = blockIdx.x & 7, wq = blockIdx.x >> 3;
  const int tpx = (ntiles + 7) >> 3;
  const int t_hi = min(ntiles, (xcd + 1) * tpx);
  for (int tile = xcd * tpx + wq; tile < t_hi; tile += wpx) {
    const int ts = tile * T;                                 // first slice of the tile
    const int R0 = ts * ROWS;
    // row-wise operands of this wave's slices: requested before the staging loads, used after the taps
    int pid[SPW], row[SPW];
    bool live[SPW];
    double e0[SPW], e2[SPW], rp[SPW];
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      const int slice = ts + wave + i * WPB;
      live[i] = wave + i * WPB < T && slice < a.nslices;
      row[i] = min(slice, a.nslices - 1) * ROWS + lane;
      const int rc = min(row[i], lastrow);
      pid[i] = (int)a.rowpid[rc];
      e0[i] = rg[rc];
      e2[i] = 0.0; rp[i] = 0.0;
      if (XM != 1) { const double xl = a.x2[rc]; e2[i] = xz ? 0.0 : xl; }
      if (XM == 2) rp[i] = a.s_out[rc];
    }
    __syncthreads();                                         // the previous tile's windows have been read (first tile: the table is complete)
    for (int sg = 0; sg < tl.nseg; ++sg) {
      const int c0 = R0 + tl.seg_off[sg];
      double *dst = s_stage + tl.seg_base[sg];
      for (int p = threadIdx.x; p < tl.seg_len[sg]; p += blockDim.x) dst[p] = omega * (du * rg[min(max(c0 + p, 0), last)]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      if (!live[i]) continue;                                // uniform in the wave
      const int rel = (wave + i * WPB) * ROWS + lane;
      const uint32_t *tm = s_msk + pid[i] * nu;
      const double *tv = s_tab8 + pid[i] * nu;
      double s = 0.0;
      bool done = false;
      if (BC) {                                              // (the launcher sets BC only for nine runs of three)
        // the pattern of the DPP row's first lane; the two halo lanes (no row of their own) follow their row
        const int lead = __builtin_amdgcn_update_dpp(0, pid[i], 0x150, 0xf, 0xf, true);
        if (__all(lane >= ROWS || pid[i] == lead)) {
          const double *tl16 = s_tab8 + lead * nu + (lane & 15);
          const double c0 = tl16[0], c1 = (lane & 15) < 11 ? tl16[16] : 0.0;
          // (three windows in flight at a time: more would cost the second resident workgroup its registers)
          double w0 = s_stage[tl.run_lds[0] + rel], w1 = s_stage[tl.run_lds[1] + rel], w2 = s_stage[tl.run_lds[2] + rel];
          double v0 = s_stage[tl.run_lds[3] + rel], v1 = s_stage[tl.run_lds[4] + rel], v2 = s_stage[tl.run_lds[5] + rel];
          s = taps_bcast_run<FM, 0>(s, w0, c0, c1); s = taps_bcast_run<FM, 1>(s, w1, c0, c1); s = taps_bcast_run<FM, 2>(s, w2, c0, c1);
          w0 = s_stage[tl.run_lds[6] + rel]; w1 = s_stage[tl.run_lds[7] + rel]; w2 = s_stage[tl.run_lds[8] + rel];
          s = taps_bcast_run<FM, 3>(s, v0, c0, c1); s = taps_bcast_run<FM, 4>(s, v1, c0, c1); s = taps_bcast_run<FM, 5>(s, v2, c0, c1);
          s = taps_bcast_run<FM, 6>(s, w0, c0, c1); s = taps_bcast_run<FM, 7>(s, w1, c0, c1); s = taps_bcast_run<FM, 8>(s, w2, c0, c1);
          done = true;
        }
      }
      if (!done)
      for (int r0 = 0; r0 < a.nruns; r0 += RB) {
        double cur[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) cur[q] = s_stage[tl.run_lds[r0 + q] + rel];
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          double c = cur[q];
#pragma unroll
          for (int t = 0; t < K; ++t) {
            if (t > 0) c = wave_shl1(c);
            const int j = (r0 + q) * K + t;
            s = FM ? __builtin_fma(tv[j], c, s) : s + tv[j] * c;
          }
        }
      }
      // strict form: "every sum of the slice finite" proves that no mask was needed (see sells_r2sweep_kernel); otherwise the slice
      // is redone from the staged windows with the masks
      if (MK && !__all(__builtin_isfinite(s))) {
        s = 0.0;
        for (int r0 = 0; r0 < a.nruns; ++r0) {
          double c = s_stage[tl.run_lds[r0] + rel];
#pragma unroll
          for (int t = 0; t < K; ++t) {
            if (t > 0) c = wave_shl1(c);
            const int j = r0 * K + t;
            const double g = __hiloint2double(__double2hiint(c) & (int)tm[j], __double2loint(c));
            s = FM ? __builtin_fma(tv[j], g, s) : s + tv[j] * g;
          }
        }
      }
      if (lane < ROWS && row[i] <= lastrow) {
        const int r = row[i];
        const double rn = e0[i] - s;
        const double sk = omega * (du * e0[i]);              // the row's own s_k
        if (XM == 0) a.x2[r] = e2[i] + sk;
        else if (XM == 2) a.x2[r] = (e2[i] + omega * (du * rp[i])) + sk;
        a.y[r] = rn;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// A whole Richardson-Jacobi smoothing pass (niter sweeps) of a SMALL level in ONE launch.
//
// On levels of a few 10^4 .. 10^5 rows a sweep kernel runs 4.6-5.9 us + ~1.5 us of dependent-launch gap against
// < 1 us of actual work (profiles/r02_tuning.md, section 2): the level is bound by per-launch latency.  Here every wave keeps
// NS slices for the whole pass: r, x, 1/diag, the pattern id and the row's own s stay in REGISTERS across the sweeps; the
// only data that crosses lanes between sweeps is s (the gathered vector), which goes through memory with
// agent-scope (sc1) 8-byte stores and loads -- coherent across CUs and XCDs without any cache fence -- and a sweep of a
// workgroup starts as soon as the workgroups that own the rows it gathers from have published the previous sweep
// (one flag word per workgroup, polled by one lane per neighbour): no grid-wide barrier.
//   flag[w] = epoch + (number of sweeps whose s-stores of workgroup w are complete)
//   WAR on the two s buffers: w rewrites s[(k+1)&1] in sweep k+1 only after its neighbours' flags reached k+1, i.e.
//   after they finished the gathers of sweep k (loads are waited for before a wave stores).
// Placement independent (no assumption on workgroup -> XCD mapping or dispatch order); needs all workgroups resident
// at once (the launcher keeps the grid <= one workgroup per CU) and every wait is bounded (timeout -> *err).
// Arithmetic, operand order and roundings are those of sells_sweep_kernel, sweep after sweep: bit-identical.
// ---------------------------------------------------------------------------
struct SellSmoothArgs {
  const uint16_t *rowpid;
  const PatEntry *tab;
  const double *tab8;
  const int32_t *run_off;
  int np, nruns;
  int64_t nrows, ncols;
  int nslices;
  const double *pdinv;      // TD: 1/diag per pattern
  const double *dinv;       // !TD: 1/diag per row
  double omega;
  int niter, x_zero;
  const double *r_in;
  double *r_out;
  double *x;
  double *s_a, *s_b;        // sweep k gathers from (k even ? s_a : s_b) and writes the other; s_a holds s_0 on entry
  uint32_t *flags;          // [gridDim.x * 16] one word per workgroup, 64 B apart; never reset: epoch advances by niter per launch
  uint32_t epoch;
  uint32_t *err;            // host-visible: set when a wait timed out
  int halo_wg;              // workgroup w waits for w-halo_wg .. w+halo_wg
  int fenced;               // progress words: release store / acquire after the poll (agent scope) on top of the explicit store drain
  uint32_t *err_dev;        // device-memory twin of *err (read at the end of the pass: a host-mapped word would cost a PCIe round trip)
};

__device__ __forceinline__ double ld_agent(const double *p)
{
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double *p, double v)
{
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// DBG (tools/mb_smooth.hip only, never instantiated by the library): 1 no neighbour waits, 2 plain gather loads, 4 plain s stores,
// 8 no taps, 16 no store drain / flag publish
template <int NS, bool TD, bool MK, int DBG = 0>
__global__ __launch_bounds__(1024) void sells_smooth_kernel(SellSmoothArgs a)
{
  constexpr int K = 3, ROWS = 65 - K, RB = 3;
  extern __shared__ double sp_smem[];
  const int nu = K * a.nruns;
  const int tot = a.np * nu;
  // [np*nu] coefficients, dense (absent entries hold 0.0) | strict form (MK): [np*nu] high-word masks behind them, read only by a
  // slice whose sum came out non-finite (see sells_r2sweep_kernel: "all sums finite" proves that no mask was needed)
  double *s_tab8 = sp_smem;
  uint32_t *s_msk = reinterpret_cast<uint32_t *>(sp_smem + tot);
  double *s_dinv = sp_smem + (MK ? 2 : 1) * (size_t)tot;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = threadIdx.x >> 6;
  const int w = blockIdx.x;
  const int last = (int)a.ncols - 1;
  const int lastrow = (int)a.nrows - 1;
  const double omega = a.omega;
  int pid[NS], row[NS];
  bool own[NS];
  double r[NS], xr[NS], so[NS], dv[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int slice = w * (wpb * NS) + i * wpb + wave;
    row[i] = min(slice, a.nslices - 1) * ROWS + lane;       // slices past the end recompute the last one, store nothing
    own[i] = slice < a.nslices && lane < ROWS && row[i] <= lastrow;
    const int rc = min(row[i], lastrow);
    pid[i] = (int)a.rowpid[rc];
    r[i] = a.r_in[rc];
    { const double xl = a.x[rc]; xr[i] = a.x_zero ? 0.0 : xl; }
    so[i] = a.s_a[rc];
    dv[i] = TD ? 0.0 : a.dinv[rc];
  }
  if (MK) { for (int i = threadIdx.x; i < tot; i += blockDim.x) { const PatEntry en = a.tab[i]; s_tab8[i] = en.v; s_msk[i] = en.m; } }
  else { for (int i = threadIdx.x; i < tot; i += blockDim.x) s_tab8[i] = a.tab8[i]; }
  if (TD)
    for (int i = threadIdx.x; i < a.np; i += blockDim.x) s_dinv[i] = a.pdinv[i];
  __syncthreads();
  if (TD) {
#pragma unroll
    for (int i = 0; i < NS; ++i) dv[i] = s_dinv[pid[i]];
  }
  const double *sin = a.s_a;
  double *sout = a.s_b;
  const int nb_lo = max(0, w - a.halo_wg), nb_hi = min((int)gridDim.x - 1, w + a.halo_wg);
  for (int k = 0; k < a.niter; ++k) {
    if (k > 0 && !(DBG & 1)) {
      // neighbours' sweep k-1 published?  one lane per neighbour, relaxed agent-scope polls
      if (wave == 0 && nb_lo + lane <= nb_hi && nb_lo + lane != w) {
        const uint32_t *f = a.flags + (size_t)(nb_lo + lane) * 16;
        const uint32_t want = a.epoch + (uint32_t)k;
        unsigned spins = 0;
        while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
          if (++spins > (1u << 22)) {
            __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (a.err_dev) __hip_atomic_store(a.err_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        if (a.fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // pairs with the neighbours' release below
      }
      __syncthreads();
    }
    double acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      double A[RB];
#pragma unroll
      for (int q = 0; q < RB; ++q) { const double *ga = sin + min(max(row[i] + a.run_off[q], 0), last); A[q] = (DBG & 2) ? *ga : ld_agent(ga); }
      const uint32_t *tm = s_msk + pid[i] * nu;
      const double *tv = s_tab8 + pid[i] * nu;
      double s = 0.0;
      for (int r0 = 0; r0 < ((DBG & 8) ? RB : a.nruns); r0 += RB) {
        double cur[RB];
#pragma unroll
        for (int q = 0; q < RB; ++q) cur[q] = A[q];
        if (r0 + RB < a.nruns) {
#pragma unroll
          for (int q = 0; q < RB; ++q) { const double *ga = sin + min(max(row[i] + a.run_off[r0 + RB + q], 0), last); A[q] = (DBG & 2) ? *ga : ld_agent(ga); }
        }
#pragma unroll
        for (int q = 0; q < RB; ++q) {
          double c = cur[q];
#pragma unroll
          for (int t = 0; t < K; ++t) {
            if (t > 0) c = wave_shl1(c);
            const int j = (r0 + q) * K + t;
            s = s + tv[j] * c;
          }
        }
      }
      if (MK && !(DBG & 8) && !__all(__builtin_isfinite(s))) {   // rare: a vector that already holds Inf / NaN -- redo the slice with the masks
        s = 0.0;
#pragma unroll 1
        for (int r0 = 0; r0 < a.nruns; ++r0) {
          const double *ga = sin + min(max(row[i] + a.run_off[r0], 0), last);
          double c = (DBG & 2) ? *ga : ld_agent(ga);
#pragma unroll
          for (int t = 0; t < K; ++t) {
            if (t > 0) c = wave_shl1(c);
            const int j = r0 * K + t;
            const double g = __hiloint2double(__double2hiint(c) & (int)tm[j], __double2loint(c));
            s = s + tv[j] * g;
          }
        }
      }
      acc[i] = s;
    }
    const bool publish = k + 1 < a.niter;                    // the s of the last sweep has no reader
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      xr[i] = xr[i] + so[i];                                 // x += s_k
      const double rn = r[i] - acc[i];                       // r -= A s_k
      r[i] = rn;
      so[i] = omega * (dv[i] * rn);                          // s_{k+1} = omega * Dinv r
      if (publish && own[i]) { if (DBG & 4) sout[row[i]] = so[i]; else st_agent(sout + row[i], so[i]); }
    }
    if (publish && !(DBG & 16)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave: the stores have left the CU ...
      __syncthreads();
      if (threadIdx.x == 0) {                                // ... before the flag
        if (a.fenced) __hip_atomic_store(a.flags + (size_t)w * 16, a.epoch + (uint32_t)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_store(a.flags + (size_t)w * 16, a.epoch + (uint32_t)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const double *t = sin; sin = sout; sout = const_cast<double *>(t);
  }
  // a wait timed out somewhere (as far as this workgroup can see): the pass is void -- leave x and r as they were, the host
  // re-runs the solve sweep by sweep (with_persist_retry)
  if (a.err_dev && __hip_atomic_load(a.err_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    if (own[i]) { a.x[row[i]] = xr[i]; a.r_out[row[i]] = r[i]; }
  }
}

// ---------------------------------------------------------------------------
// Setup of the additive-Schwarz operator in row-pattern form (gmg_solver::build_patch_operator) on the device: which patches touch a
// dof, in ascending patch order (a counting sort of the patch slots by dof), and one 64-bit signature per row over
// (inverse block, local row, patch shape, first dof - row) of its patches -- 4.6e8 slots / 1.3e8 rows at 256^3 Q2, 2.1 s on 16 host cores.
// ---------------------------------------------------------------------------
__global__ void slot_count_kernel(int64_t ne, const int32_t *__restrict__ prow, int32_t *__restrict__ cnt)
{
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < ne; q += (int64_t)gridDim.x * blockDim.x) atomicAdd(cnt + prow[q], 1);
}
// exclusive prefix sum of n int32 counts -> n + 1 offsets: (1) per block of 4096 a local scan + the block's total, (2) the totals
// scanned by one block, (3) added back
constexpr int kScanItems = 16, kScanBlock = 256, kScanTile = kScanItems * kScanBlock;
__global__ __launch_bounds__(kScanBlock) void scan_local_kernel(int64_t n, const int32_t *__restrict__ in, int32_t *__restrict__ out, int32_t *__restrict__ totals)
{
  __shared__ int32_t sh[kScanBlock];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int32_t v[kScanItems], s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) { v[k] = base + k < n ? in[base + k] : 0; s += v[k]; }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = 1; d < kScanBlock; d <<= 1) {
    const int32_t t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  int32_t run = sh[threadIdx.x] - s;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
  if (threadIdx.x == kScanBlock - 1) totals[blockIdx.x] = sh[threadIdx.x];
}
__global__ __launch_bounds__(1024) void scan_totals_kernel(int nb, int32_t *__restrict__ totals)
{
  __shared__ int32_t sh[1024];
  int32_t carry = 0;
  for (int b0 = 0; b0 < nb; b0 += 1024) {
    const int i = b0 + (int)threadIdx.x;
    const int32_t mine = i < nb ? totals[i] : 0;
    sh[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int32_t t = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nb) totals[i] = carry + sh[threadIdx.x] - mine;
    carry += sh[1023];
    __syncthreads();
  }
}
__global__ __launch_bounds__(kScanBlock) void scan_add_kernel(int64_t n, int32_t *__restrict__ out, const int32_t *__restrict__ totals, int32_t total_all)
{
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  const int32_t add = totals[blockIdx.x];
#pragma unroll
  for (int k = 0; k < kScanItems; ++k)
    if (base + k < n) out[base + k] += add;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = total_all;
}
// inc[iptr[dof] + (arrival order)] = slot ; the short lists are put into ascending slot (= patch) order afterwards
__global__ void slot_scatter_kernel(int64_t ne, const int32_t *__restrict__ prow, const int32_t *__restrict__ iptr, int32_t *__restrict__ fill, int32_t *__restrict__ inc)
{
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < ne; q += (int64_t)gridDim.x * blockDim.x) {
    const int32_t i = prow[q];
    inc[iptr[i] + atomicAdd(fill + i, 1)] = (int32_t)q;
  }
}
__global__ void slot_sort_kernel(int64_t n, const int32_t *__restrict__ iptr, int32_t *__restrict__ inc)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int32_t *lo = inc + iptr[i];
    const int len = iptr[i + 1] - iptr[i];
    for (int a = 1; a < len; ++a) {                          // insertion sort: a handful of entries
      const int32_t v = lo[a];
      int b = a - 1;
      for (; b >= 0 && lo[b] > v; --b) lo[b + 1] = lo[b];
      lo[b + 1] = v;
    }
  }
}
__global__ void slot_patch_kernel(int64_t npatch, const int64_t *__restrict__ pptr, int32_t *__restrict__ s2p)
{
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npatch; p += (int64_t)gridDim.x * blockDim.x)
    for (int64_t q = pptr[p]; q < pptr[p + 1]; ++q) s2p[q] = (int32_t)p;
}
// signature words of row i, slot k of its list: (inverse block of the patch, local row, patch shape, first dof of the patch - i)
__device__ __forceinline__ void row_sig_words(int64_t i, int32_t q, const int32_t *__restrict__ s2p, const int64_t *__restrict__ pptr, const int32_t *__restrict__ prow,
                                              const int32_t *__restrict__ ublock, const int32_t *__restrict__ shape, int64_t w[4])
{
  const int32_t p = s2p[q];
  w[0] = ublock[p]; w[1] = (int64_t)q - pptr[p]; w[2] = shape[p]; w[3] = (int64_t)prow[pptr[p]] - i;
}
__global__ void row_sig_hash_kernel(int64_t n, const int32_t *__restrict__ iptr, const int32_t *__restrict__ inc, const int32_t *__restrict__ s2p,
                                    const int64_t *__restrict__ pptr, const int32_t *__restrict__ prow, const int32_t *__restrict__ ublock,
                                    const int32_t *__restrict__ shape, unsigned long long *__restrict__ hash)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int len = iptr[i + 1] - iptr[i];
    unsigned long long h = 1469598103934665603ull ^ (unsigned long long)(4 * len);
    for (int k = 0; k < len; ++k) {
      int64_t w[4];
      row_sig_words(i, inc[iptr[i] + k], s2p, pptr, prow, ublock, shape, w);
#pragma unroll
      for (int j = 0; j < 4; ++j) { h = (h ^ (unsigned long long)w[j]) * 1099511628211ull; h ^= h >> 29; }
    }
    hash[i] = h;
  }
}
// exact check of the grouping by hash + the row's pattern id: row i against the representative of its hash (table sorted by hash)
__global__ void row_sig_assign_kernel(int64_t n, const int32_t *__restrict__ iptr, const int32_t *__restrict__ inc, const int32_t *__restrict__ s2p,
                                      const int64_t *__restrict__ pptr, const int32_t *__restrict__ prow, const int32_t *__restrict__ ublock,
                                      const int32_t *__restrict__ shape, const unsigned long long *__restrict__ hash, int ngroups,
                                      const unsigned long long *__restrict__ ghash, const int64_t *__restrict__ grep, const int32_t *__restrict__ gid,
                                      uint16_t *__restrict__ rowpid, int *__restrict__ nmis)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long h = hash[i];
    int lo = 0, hi = ngroups - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (ghash[mid] < h) lo = mid + 1; else hi = mid; }
    bool ok = ghash[lo] == h;
    const int64_t r = grep[lo];
    const int len = iptr[i + 1] - iptr[i];
    ok = ok && len == iptr[r + 1] - iptr[r];
    for (int k = 0; ok && k < len; ++k) {
      int64_t a[4], b[4];
      row_sig_words(i, inc[iptr[i] + k], s2p, pptr, prow, ublock, shape, a);
      row_sig_words(r, inc[iptr[r] + k], s2p, pptr, prow, ublock, shape, b);
      ok = a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3];
    }
    if (!ok) atomicAdd(nmis, 1);
    rowpid[i] = (uint16_t)gid[lo];
  }
}

// dx -= c ; x += dx   (patch-corrected prolongation: y = P x - sum_p A_pp^-1 (A P x)_p,
// PatchBasedSmoothers/PatchTransferOperators.jl:153-172, then xh .+= dxh GMGLinearSolvers.jl:494)
__global__ void prolong_correct_kernel(int64_t n, const double *__restrict__ c, double *__restrict__ dx, double *__restrict__ x)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double d = dx[i] - c[i];
    dx[i] = d;
    x[i] = x[i] + d;
  }
}

// dst = src, 16 B per lane, non-temporal: the streaming-ceiling probe (gmg_stream_probe).  Moves exactly 32*n2 bytes.
__global__ __launch_bounds__(256) void stream_copy_kernel(int64_t n2, const double *__restrict__ src, double *__restrict__ dst)
{
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 *s2 = reinterpret_cast<const d2 *>(src);
  d2 *t2 = reinterpret_cast<d2 *>(dst);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(__builtin_nontemporal_load(s2 + i), t2 + i);
}

// read-only twin of the probe (gmg_stream_probe_read): 16 B per lane, non-temporal loads, one 8-byte result per thread
__global__ __launch_bounds__(256) void stream_read_kernel(int64_t n2, const double *__restrict__ src, double *__restrict__ sink)
{
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 *s2 = reinterpret_cast<const d2 *>(src);
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    const d2 v = __builtin_nontemporal_load(s2 + i);
    acc += v.x + v.y;
  }
  sink[(int64_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// s = omega*(dinv.*r)  (first sweep of a ONEG smoothing pass)
__global__ void scaled_jacobi_kernel(int64_t n, double omega, const double *__restrict__ dinv,
                                     const double *__restrict__ r, double *__restrict__ s)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    s[i] = omega * (dinv[i] * r[i]);
}

// ---------------------------------------------------------------------------
// inv_diag = 1.0 ./ diag(A)   (JacobiLinearSolvers.jl:20-23)
// ---------------------------------------------------------------------------
template <typename PtrT>
__global__ void inv_diag_kernel(int64_t n, const PtrT *__restrict__ rowptr, const int32_t *__restrict__ col,
                                const double *__restrict__ val, double *__restrict__ dinv, int *__restrict__ nzero)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = 0.0;
  for (PtrT k = rowptr[i]; k < rowptr[i + 1]; ++k)
    if (col[k] == (int32_t)i) d += val[k];
  if (d == 0.0) atomicAdd(nzero, 1);
  dinv[i] = 1.0 / d;
}

// ---------------------------------------------------------------------------
// K7: deterministic two-stage fp64 reductions.  Stage 1 writes one partial per
// workgroup (fixed grid => fixed summation tree => bit-reproducible run to
// run); stage 2 is a single workgroup.
// ---------------------------------------------------------------------------
constexpr int kRedBlocks = 1024;

__device__ __forceinline__ double wave_sum(double s)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  return s;
}

__device__ __forceinline__ double block_sum(double s, double *sh /*[4]*/)
{
  s = wave_sum(s);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = s;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  return t; // valid on thread 0
}

// The second stage of a two-stage reduction done by the CONSUMER: every workgroup of the kernel that needs the scalar sums the
// producer's partials itself -- the very sum reduce_final_kernel forms (same strided accumulation over kBlock lanes, same tree), so
// every workgroup holds the same bits as the one-workgroup kernel would have written -- instead of a dependent 4.5 us launch in
// between (three per CG iteration; a ticket scheme that lets the producer's last workgroup finalise was measured slower: +12 us per
// producer for the agent-scope store / ticket / reload chain).  blockDim.x must be kBlock.  Returns the sum to every thread.
__device__ __forceinline__ double sum_partials_all(const double *__restrict__ parts, int nparts, double *sh /*[5]*/)
{
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) s += parts[i];
  s = wave_sum(s);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = s;
  __syncthreads();
  if (threadIdx.x == 0) sh[4] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return sh[4];
}

__global__ __launch_bounds__(kBlock) void dot_partial_kernel(int64_t n, const double *__restrict__ a,
                                                             const double *__restrict__ b,
                                                             double *__restrict__ partials, int vec)
{
  __shared__ double sh[4];
  double s = 0.0;
  if (vec) { // both pointers 16-byte aligned: 16 B/lane loads
    const int64_t n2 = n >> 1;
    const double2 *a2 = reinterpret_cast<const double2 *>(a);
    const double2 *b2 = reinterpret_cast<const double2 *>(b);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
      const double2 u = a2[i], v = b2[i];
      s += u.x * v.x;
      s += u.y * v.y;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && (n & 1)) s += a[n - 1] * b[n - 1];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) s += a[i] * b[i];
  }
  const double t = block_sum(s, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// out[slot] = (take_sqrt ? sqrt : id)(sum partials)
__global__ __launch_bounds__(kBlock) void reduce_final_kernel(int nparts, const double *__restrict__ partials,
                                                              double *__restrict__ out, int take_sqrt)
{
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) s += partials[i];
  const double t = block_sum(s, sh);
  if (threadIdx.x == 0) out[0] = take_sqrt ? sqrt(t) : t;
}

// reduce_final_kernel + post_scalar_kernel in one launch: the scalar is also posted into host-mapped memory (fetch_scalar polls the number)
__global__ __launch_bounds__(kBlock) void reduce_post_kernel(int nparts, const double *__restrict__ partials, double *__restrict__ out,
                                                             int take_sqrt, double *value, unsigned long long *seq, unsigned long long want)
{
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nparts; i += kBlock) s += partials[i];
  const double t = block_sum(s, sh);
  if (threadIdx.x == 0) {
    const double res = take_sqrt ? sqrt(t) : t;
    out[0] = res;
    __hip_atomic_store(value, res, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(seq, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---------------------------------------------------------------------------
// K8: axpy-class kernels
// ---------------------------------------------------------------------------
// p = z + beta*p                     (CGSolvers.jl:101)
__global__ void xpby_kernel(int64_t n, const double *__restrict__ z, double beta, double *__restrict__ p)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = z[i] + beta * p[i];
}
// The same with beta formed on the device from reduction results: beta = gamma/gamma_old, or
// (gamma - delta)/gamma_old for the flexible variant (CGSolvers.jl:95,99) -- no host round trip.
// first != 0: p is the zero vector of CGSolvers.jl:80 and is not read.
// gparts != nullptr: gamma is still in the producer's partials (dot_partial_kernel): every workgroup sums them (sum_partials_all) and
// workgroup 0 stores the scalar for the kernels that read it later.  blockDim.x == kBlock.
__global__ __launch_bounds__(kBlock) void xpby_dev_kernel(int64_t n, const double *__restrict__ z, double *__restrict__ gamma,
                                                          const double *__restrict__ gamma_old, const double *__restrict__ delta,
                                                          double *__restrict__ p, int first, const double *__restrict__ gparts, int ngparts)
{
  __shared__ double sh[5];
  double g;
  if (gparts) {
    g = sum_partials_all(gparts, ngparts, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) gamma[0] = g;
  } else g = gamma[0];
  const double beta = delta ? (g - delta[0]) / gamma_old[0] : g / gamma_old[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    p[i] = z[i] + beta * (first ? 0.0 : p[i]);
}
__global__ void set_scalar_kernel(double *__restrict__ dst, double v) { dst[0] = v; }
// one lane: *value = *src ; *seq = want  in host-mapped memory, system scope, the number released after the value (fetch_scalar polls it)
__global__ void post_scalar_kernel(const double *__restrict__ src, double *value, unsigned long long *seq, unsigned long long want)
{
  const double v = src[0];
  __hip_atomic_store(value, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(seq, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// x += alpha*p ; r -= alpha*w ; partial ||r||^2   (CGSolvers.jl:108-111)
// alpha = gamma / dot(p,w) (CGSolvers.jl:105) formed on the device from the two reduction results
// pwparts != nullptr: dot(p,w) is still in dot_partial_kernel's partials: summed by every workgroup (sum_partials_all), stored by
// workgroup 0; `partials` (the output) must then be a different array
__global__ __launch_bounds__(kBlock) void cg_update_kernel(int64_t n, const double *__restrict__ gamma,
                                                           double *__restrict__ pw, const double *__restrict__ p,
                                                           const double *__restrict__ w, double *__restrict__ x,
                                                           double *__restrict__ r, double *__restrict__ partials,
                                                           const double *__restrict__ pwparts, int npwparts)
{
  __shared__ double sh[5];
  double pwv;
  if (pwparts) {
    pwv = sum_partials_all(pwparts, npwparts, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0) pw[0] = pwv;
    __syncthreads();                                         // sh is reused below
  } else pwv = pw[0];
  const double alpha = gamma[0] / pwv;
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    x[i] += alpha * p[i];
    const double rn = r[i] - alpha * w[i];
    r[i] = rn;
    s += rn * rn;
  }
  const double t = block_sum(s, sh);
  if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
// y += alpha*x                        (FGMRESSolvers.jl:162,192)
__global__ void axpy_kernel(int64_t n, double alpha, const double *__restrict__ x, double *__restrict__ y)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = y[i] + alpha * x[i];
}
// y = y - alpha*x  with alpha read from device memory (MGS step without a host round trip)
__global__ void axmy_dev_kernel(int64_t n, const double *__restrict__ alpha, const double *__restrict__ x,
                                double *__restrict__ y)
{
  const double al = alpha[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = y[i] - al * x[i];
}
// x ./= s                             (FGMRESSolvers.jl:146,165) -- true division as the reference
__global__ void div_kernel(int64_t n, double s, double *__restrict__ x)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = x[i] / s;
}
__global__ void div_dev_kernel(int64_t n, const double *__restrict__ s, double *__restrict__ x)
{
  const double d = s[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = x[i] / d;
}
// dx = omega*(dinv.*r) ; optionally x += dx     (unfused Jacobi apply, gmg_precond_apply)
__global__ void jacobi_apply_kernel(int64_t n, const double *__restrict__ dinv, const double *__restrict__ r,
                                    double *__restrict__ dx)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = dinv[i] * r[i];
}

// ---------------------------------------------------------------------------
// K6: coarsest solve x = Ainv * r, Ainv dense row-major n x n.  One wave per
// row, 16 B/lane loads, wave64 shuffle reduction.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void dense_gemv_kernel(int n, const double *__restrict__ Ainv,
                                                            const double *__restrict__ r, double *__restrict__ x)
{
  const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n) return;
  const double *row = Ainv + (size_t)wave * n;
  // 4 independent 16-byte loads per lane and iteration (4 KB of the row in flight per wave); rows of an odd n are only
  // 8-byte aligned, which global loads allow
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int base = 0;
  for (; base + 512 <= n; base += 512) {
    const int j = base + 2 * lane;
    const d2u a0 = *reinterpret_cast<const d2u *>(row + j), a1 = *reinterpret_cast<const d2u *>(row + j + 128);
    const d2u a2 = *reinterpret_cast<const d2u *>(row + j + 256), a3 = *reinterpret_cast<const d2u *>(row + j + 384);
    const d2u r0 = *reinterpret_cast<const d2u *>(r + j), r1 = *reinterpret_cast<const d2u *>(r + j + 128);
    const d2u r2 = *reinterpret_cast<const d2u *>(r + j + 256), r3 = *reinterpret_cast<const d2u *>(r + j + 384);
    s0 += a0.x * r0.x + a0.y * r0.y;
    s1 += a1.x * r1.x + a1.y * r1.y;
    s2 += a2.x * r2.x + a2.y * r2.y;
    s3 += a3.x * r3.x + a3.y * r3.y;
  }
  for (int j = base + lane; j < n; j += 64) s0 += row[j] * r[j];
  const double s = wave_sum((s0 + s1) + (s2 + s3));
  if (lane == 0) x[wave] = s;
}

// ---------------------------------------------------------------------------
// Device-side dense inversion of a large coarsest-level matrix (setup only): blocked
// Gauss-Jordan without pivoting, panel width GJ_B.  For panel K = [k0,k0+b):
//   Pinv = A[K,K]^-1 ; R = Pinv*A[K,:] ; C = A[:,K]
//   A[i,j] -= C[i,:]*R[:,j]  (i,j not in K) ; A[K,notK] = R ; A[notK,K] = -C*Pinv ; A[K,K] = Pinv
// After the last panel A holds A^-1.  Valid for matrices that need no pivoting (SPD /
// diagonally dominant coarse operators); the caller verifies the result.
// ---------------------------------------------------------------------------
constexpr int GJ_B = 32;

// densify: D[i*n + col] += val
template <typename PtrT>
__global__ void densify_kernel(int64_t n, const PtrT *__restrict__ rowptr, const int32_t *__restrict__ col,
                               const double *__restrict__ val, double *__restrict__ D)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (PtrT k = rowptr[i]; k < rowptr[i + 1]; ++k) D[(size_t)i * n + col[k]] += val[k];
}

// one workgroup: Pinv = A[K,K]^-1 (Gauss-Jordan in LDS, no pivoting); flags a zero pivot
__global__ __launch_bounds__(GJ_B *GJ_B) void gj_diag_kernel(int n, int k0, int b, const double *__restrict__ A,
                                                               double *__restrict__ Pinv, int *__restrict__ bad)
{
  __shared__ double M[GJ_B][GJ_B + 1], X[GJ_B][GJ_B + 1];
  const int r = threadIdx.x / GJ_B, c = threadIdx.x % GJ_B;
  const bool in = r < b && c < b;
  M[r][c] = in ? A[(size_t)(k0 + r) * n + k0 + c] : (r == c ? 1.0 : 0.0);
  X[r][c] = (r == c) ? 1.0 : 0.0;
  __syncthreads();
  for (int j = 0; j < GJ_B; ++j) {
    // all reads of step j first, then one barrier, then the writes
    const double d = M[j][j];
    const double mjc = M[j][c] / d, xjc = X[j][c] / d;
    const double f = M[r][j];
    __syncthreads();
    if (d == 0.0) { if (threadIdx.x == 0) atomicAdd(bad, 1); return; }   // uniform
    if (r == j) { M[j][c] = mjc; X[j][c] = xjc; }
    else { M[r][c] -= f * mjc; X[r][c] -= f * xjc; }
    __syncthreads();
  }
  if (in) Pinv[r * GJ_B + c] = X[r][c];
}

// panels: R[t][j] = sum_u Pinv[t][u]*A[k0+u][j] ; C[i][t] = A[i][k0+t] ; Cp[i][t] = -sum_u C[i][u]*Pinv[u][t]
__global__ void gj_panels_kernel(int n, int k0, int b, const double *__restrict__ A, const double *__restrict__ Pinv,
                                 double *__restrict__ R, double *__restrict__ C, double *__restrict__ Cp)
{
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over n*b
  if (idx >= (int64_t)n * b) return;
  const int t = (int)(idx % b);
  const int64_t q = idx / b;                                              // column j (for R) / row i (for C)
  double s = 0.0;
  for (int u = 0; u < b; ++u) s += Pinv[t * GJ_B + u] * A[(size_t)(k0 + u) * n + q];
  R[(size_t)t * n + q] = s;
  C[(size_t)q * GJ_B + t] = A[(size_t)q * n + k0 + t];
  double sp = 0.0;
  for (int u = 0; u < b; ++u) sp += A[(size_t)q * n + k0 + u] * Pinv[u * GJ_B + t];
  Cp[(size_t)q * GJ_B + t] = -sp;
}

// trailing update, 64x64 tile per workgroup, 4x4 outputs per thread
__global__ __launch_bounds__(256) void gj_update_kernel(int n, int k0, int b, double *__restrict__ A,
                                                        const double *__restrict__ Pinv, const double *__restrict__ R,
                                                        const double *__restrict__ C, const double *__restrict__ Cp)
{
  __shared__ double sC[64][GJ_B + 1], sR[GJ_B][64 + 1];
  const int ti = blockIdx.y * 64, tj = blockIdx.x * 64;
  for (int e = threadIdx.x; e < 64 * GJ_B; e += 256) {
    const int i = e / GJ_B, t = e % GJ_B;
    sC[i][t] = (ti + i < n && t < b) ? C[(size_t)(ti + i) * GJ_B + t] : 0.0;
    const int tt = e / 64, j = e % 64;
    sR[tt][j] = (tj + j < n && tt < b) ? R[(size_t)tt * n + tj + j] : 0.0;
  }
  __syncthreads();
  const int li = (threadIdx.x / 16) * 4, lj = (threadIdx.x % 16) * 4;
  double acc[4][4] = {};
  for (int t = 0; t < GJ_B; ++t) {
    double cv[4], rv[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { cv[a] = sC[li + a][t]; rv[a] = sR[t][lj + a]; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[a][c] += cv[a] * rv[c];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int i = ti + li + a;
    if (i >= n) continue;
    const bool iK = i >= k0 && i < k0 + b;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = tj + lj + c;
      if (j >= n) continue;
      const bool jK = j >= k0 && j < k0 + b;
      double v;
      if (iK && jK) v = Pinv[(i - k0) * GJ_B + (j - k0)];
      else if (iK) v = R[(size_t)(i - k0) * n + j];
      else if (jK) v = Cp[(size_t)i * GJ_B + (j - k0)];
      else v = A[(size_t)i * n + j] - acc[a][c];
      A[(size_t)i * n + j] = v;
    }
  }
}

// The same trailing update on the FP64 matrix cores: A[i][j] -= sum_t C[i][t] * R[t][j], t < GJ_B = 32, as eight
// v_mfma_f64_16x16x4_f64 steps per 16x16 output tile.  A workgroup of 4 waves owns a 64x64 tile (wave w: rows 16w..16w+15,
// four column tiles); the operands are read straight from the two n x 32 panels (L2-resident, a lane needs ONE double per
// step: A-operand C[row = lane%16][k = lane/16], B-operand R[k = lane/16][col = lane%16]), the accumulator register a of a
// lane holds row (lane/16) + 4a of column lane%16.  Setup only (the dense inverse of the coarsest level; verified against the sparse
// operator afterwards); sums are FMA chains, so the inverse differs from the scalar kernel's in the last bits.
// 3 375 dofs: 252 -> ~45 us per panel.
__global__ __launch_bounds__(256) void gj_update_mfma_kernel(int n, int k0, int b, double *__restrict__ A,
                                                             const double *__restrict__ Pinv, const double *__restrict__ R,
                                                             const double *__restrict__ C, const double *__restrict__ Cp)
{
  typedef double d4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ti = blockIdx.y * 64 + wave * 16, tj = blockIdx.x * 64;
  const int lr = lane & 15, lk = lane >> 4;
  // A-operands of the eight k-steps: C[ti + lr][4 s + lk]
  double ca[GJ_B / 4];
  const int ia = ti + lr;
#pragma unroll
  for (int s = 0; s < GJ_B / 4; ++s) {
    const int t = 4 * s + lk;
    ca[s] = (ia < n && t < b) ? C[(size_t)ia * GJ_B + t] : 0.0;
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int j = tj + 16 * c + lr;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < GJ_B / 4; ++s) {
      const int t = 4 * s + lk;
      const double rb = (j < n && t < b) ? R[(size_t)t * n + j] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[s], rb, acc, 0, 0, 0);
    }
    if (j >= n) continue;
    const bool jK = j >= k0 && j < k0 + b;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int i = ti + lk + 4 * a;                         // f64 MFMA: row = (lane >> 4) + 4 * reg, col = lane & 15
      if (i >= n) continue;
      const bool iK = i >= k0 && i < k0 + b;
      double v;
      if (iK && jK) v = Pinv[(i - k0) * GJ_B + (j - k0)];
      else if (iK) v = R[(size_t)(i - k0) * n + j];
      else if (jK) v = Cp[(size_t)i * GJ_B + (j - k0)];
      else v = A[(size_t)i * n + j] - acc[a];
      A[(size_t)i * n + j] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// The same inversion with 64-wide panels for LARGE coarsest levels (tens of thousands of dofs: the 31^3-node coarsest level
// of BASELINE config 3 is a 7.1 GB dense inverse).  A pass over the matrix is bound by HBM (read + write of n^2 doubles), so
// the panel width sets the number of passes; 64 columns cost 16 v_mfma_f64_16x16x4_f64 per 16 x 16 tile, still far below the
// time the tile's 4 KB take to stream.  The matrix is held with a leading dimension lda (a multiple of 16: every 16-double
// row segment a wave touches is one aligned 128-byte line; n itself is usually odd), R shares it, C / Cp are n x 64; all of
// them padded with zeros to whole 64 x 128 tiles.
//   gj_diag64_kernel     Pinv = A[K,K]^-1, one workgroup of 1024 threads, Gauss-Jordan in LDS (64 KB)
//   gj_panel_rows_kernel R = Pinv * A[K,:]                 (thread = column, 16 rows of R per thread, Pinv uniform)
//   gj_panel_cols_kernel C = A[:,K] ; Cp = -C * Pinv       (wave = 16 rows of A, lane = column of the panel)
//   gj_update64_kernel   A -= C * R off the panel; panel rows / columns / block replaced as in gj_update_kernel.
//                        A wave owns 32 rows x 64 columns (2 x 4 MFMA tiles: the C operand stays in registers for the four
//                        column tiles), a workgroup 64 x 128.  k index of MFMA step s in lane group g (= lane / 16): 16 g + s,
//                        so a lane's C operands are 16 consecutive doubles.
// ---------------------------------------------------------------------------
constexpr int GJ_W = 64;
constexpr int GJ_SX = 8;                                     // super-tile of gj_update64_kernel: 8 workgroup tiles of 128 columns (x 1024 rows)

__global__ __launch_bounds__(1024) void gj_diag64_kernel(int n, int64_t lda, int k0, int b, const double *__restrict__ A,
                                                         double *__restrict__ Pinv, int *__restrict__ bad)
{
  __shared__ double M[GJ_W][GJ_W], X[GJ_W][GJ_W];
  const int r0 = threadIdx.x >> 6, c = threadIdx.x & 63;     // rows r0 + 16 q of column c (r0 is uniform in a wave)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = r0 + 16 * q;
    M[r][c] = (r < b && c < b) ? A[(size_t)(k0 + r) * lda + k0 + c] : (r == c ? 1.0 : 0.0);
    X[r][c] = (r == c) ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int j = 0; j < GJ_W; ++j) {
    const double d = M[j][j];
    const double mjc = M[j][c] / d, xjc = X[j][c] / d;
    double f[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) f[q] = M[r0 + 16 * q][j];
    __syncthreads();
    if (d == 0.0) { if (threadIdx.x == 0) atomicAdd(bad, 1); return; }   // uniform
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + 16 * q;
      if (r == j) { M[j][c] = mjc; X[j][c] = xjc; }
      else { M[r][c] -= f[q] * mjc; X[r][c] -= f[q] * xjc; }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = r0 + 16 * q;
    if (r < b && c < b) Pinv[r * GJ_W + c] = X[r][c];
  }
}

// R[t][j] = sum_u Pinv[t][u] * A[k0 + u][j]: blockIdx.y picks 16 rows t of R, a thread one column j
__global__ __launch_bounds__(64) void gj_panel_rows_kernel(int n, int64_t lda, int k0, int b, const double *__restrict__ A,
                                                           const double *__restrict__ Pinv, double *__restrict__ R)
{
  const int j = blockIdx.x * 64 + threadIdx.x;
  const int t0 = blockIdx.y * 16;
  if (j >= n) return;
  double acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.0;
  for (int u = 0; u < b; ++u) {
    const double a = A[(size_t)(k0 + u) * lda + j];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] += Pinv[(t0 + t) * GJ_W + u] * a;   // uniform address: scalar loads
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) R[(size_t)(t0 + t) * lda + j] = (t0 + t < b) ? acc[t] : 0.0;   // rows >= b: zeros (the update reads all 64)
}

// C[i][t] = A[i][k0 + t] ; Cp[i][t] = -sum_u C[i][u] * Pinv[u][t]: a wave takes 16 rows, lane = t
__global__ __launch_bounds__(256) void gj_panel_cols_kernel(int n, int64_t lda, int k0, int b, const double *__restrict__ A,
                                                            const double *__restrict__ Pinv, double *__restrict__ C,
                                                            double *__restrict__ Cp)
{
  __shared__ double sC[4][GJ_W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i0 = (blockIdx.x * 4 + wave) * 16;
  for (int q = 0; q < 16; ++q) {
    const int i = i0 + q;
    if (i >= n) break;                                       // uniform in the wave
    const double v = (lane < b) ? A[(size_t)i * lda + k0 + lane] : 0.0;
    sC[wave][lane] = v;
    __builtin_amdgcn_wave_barrier();
    double sp = 0.0;
    if (lane < b)
      for (int u = 0; u < b; ++u) sp += sC[wave][u] * Pinv[u * GJ_W + lane];
    C[(size_t)i * GJ_W + lane] = v;
    Cp[(size_t)i * GJ_W + lane] = -sp;
    __builtin_amdgcn_wave_barrier();
  }
}

template <int RT>                                            // a wave owns 16 RT rows x 64 columns, a workgroup 32 RT x 128
__global__ __launch_bounds__(256) void gj_update64_kernel(int n, int64_t lda, int k0, int b, double *__restrict__ A,
                                                          const double *__restrict__ Pinv, const double *__restrict__ R,
                                                          const double *__restrict__ C, const double *__restrict__ Cp)
{
  // Every buffer is padded with zeros to whole tiles (lda a multiple of 128, rows of A / C / Cp a multiple of 128, R rows >= b and
  // C columns >= b zero): no load of the hot path is conditional -- a conditional load is a branch, and the compiler then waits for
  // each operand before the MFMA that uses it (64 serialised L2 round trips per wave: 8.5 ms per pass instead of 4.5).
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int WR = 32 * RT;                                // rows of a workgroup tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Workgroups are dealt to the eight XCDs round-robin; each XCD walks its own contiguous share of the super-tiles (GJ_SX x
  // 1024 / WR workgroup tiles = 1024 x 1024 entries: the 512 KB of C rows and 512 KB of R columns they share stay in that XCD's L2).
  constexpr int SY = 1024 / WR;
  const int nbx = (n + 127) / 128, nby = (n + WR - 1) / WR;
  const int nsx = (nbx + GJ_SX - 1) / GJ_SX;
  const int64_t per_xcd = (int64_t)gridDim.x / 8;            // the launcher rounds the grid up to a multiple of 8 super-tiles
  const int64_t L = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int st = (int)(L / (GJ_SX * SY)), in = (int)(L % (GJ_SX * SY));
  const int bx = (st % nsx) * GJ_SX + in % GJ_SX, by = (st / nsx) * SY + in / GJ_SX;
  if (bx >= nbx || by >= nby) return;
  const int ti = by * WR + (wave >> 1) * (16 * RT);
  const int tj = bx * 128 + (wave & 1) * 64;
  const int lr = lane & 15, lk = lane >> 4;
  // A-operands: C[ti + 16 rt + lr][16 lk + s]
  double ca[RT][16];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const d4 *src = reinterpret_cast<const d4 *>(C + (size_t)(ti + 16 * rt + lr) * GJ_W + 16 * lk);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const d4 x = src[v];
      ca[rt][4 * v + 0] = x[0]; ca[rt][4 * v + 1] = x[1]; ca[rt][4 * v + 2] = x[2]; ca[rt][4 * v + 3] = x[3];
    }
  }
  const bool panel = (ti < k0 + b && ti + 16 * RT > k0) || (tj < k0 + b && tj + 64 > k0);   // uniform in the wave
  double *Aw = A + (size_t)(ti + lk) * lda + tj + lr;         // register (rt, a) of column tile c: Aw[(16 rt + 4 a) * lda + 16 c]
  const double *Rw = R + (size_t)(16 * lk) * lda + tj + lr;   // step s of column tile c: Rw[s * lda + 16 c]
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double av[RT][4], rb[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) rb[s] = Rw[(size_t)s * lda + 16 * c];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int a = 0; a < 4; ++a) av[rt][a] = Aw[(size_t)(16 * rt + 4 * a) * lda + 16 * c];   // f64 MFMA: row = (lane >> 4) + 4 * reg, col = lane & 15
    d4 acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[rt] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[rt][s], rb[s], acc[rt], 0, 0, 0);
    if (!panel) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int a = 0; a < 4; ++a) Aw[(size_t)(16 * rt + 4 * a) * lda + 16 * c] = av[rt][a] - acc[rt][a];
    } else {
      const int j = tj + 16 * c + lr;
      if (j >= n) continue;
      const bool jK = j >= k0 && j < k0 + b;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int i = ti + 16 * rt + lk + 4 * a;
          if (i >= n) continue;
          const bool iK = i >= k0 && i < k0 + b;
          double v;
          if (iK && jK) v = Pinv[(i - k0) * GJ_W + (j - k0)];
          else if (iK) v = R[(size_t)(i - k0) * lda + j];
          else if (jK) v = Cp[(size_t)i * GJ_W + (j - k0)];
          else v = av[rt][a] - acc[rt][a];
          A[(size_t)i * lda + j] = v;
        }
    }
  }
}

// densify into a padded leading dimension: D[i * lda + col] += val
template <typename PtrT>
__global__ void densify_ld_kernel(int64_t n, int64_t lda, const PtrT *__restrict__ rowptr, const int32_t *__restrict__ col,
                                  const double *__restrict__ val, double *__restrict__ D)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (PtrT k = rowptr[i]; k < rowptr[i + 1]; ++k) D[(size_t)i * lda + col[k]] += val[k];
}

// ---------------------------------------------------------------------------
// K9/K10: patch smoother.
//  setup : extract A[p,p] (BlockJacobiSolvers.jl:160), factorise (LU with partial
//          pivoting = PatchSolvers.jl:176 lu!, or NoPivot = BlockJacobiSolvers.jl:162)
//          and store the explicit inverse block (n_p x n_p, row-major).
//  apply : per patch  xp = inv(A_pp) * b[rows_p]  -> contribution buffer;
//          per dof    dx_i = omega * sum_{p contains i, ascending p} xp ; x_i += dx_i
//          (same accumulation order as the reference's sequential patch loop
//           PatchSolvers.jl:288-297 => deterministic, no atomics).
// ---------------------------------------------------------------------------
// One thread per patch builds the inverse by Gauss-Jordan on an identity-augmented
// copy held in global scratch (setup only, not on the timed path).
template <typename PtrT>
__global__ void patch_factor_kernel(int64_t npatch, const int64_t *__restrict__ pptr, const int32_t *__restrict__ pdofs,
                                    const int64_t *__restrict__ boff, const PtrT *__restrict__ rowptr,
                                    const int32_t *__restrict__ col, const double *__restrict__ val,
                                    double *__restrict__ binv, double *__restrict__ scratch, int max_np,
                                    int pivoting, int *__restrict__ nsing)
{
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npatch) return;
  const int np = (int)(pptr[p + 1] - pptr[p]);
  if (np == 0) return;
  const int32_t *dofs = pdofs + pptr[p];
  double *M = scratch + (size_t)p * max_np * max_np; // row-major np x np copy of A[p,p]
  double *X = binv + boff[p];                        // row-major np x np, becomes the inverse
  for (int r = 0; r < np; ++r) {
    for (int c = 0; c < np; ++c) { M[r * np + c] = 0.0; X[r * np + c] = (r == c) ? 1.0 : 0.0; }
    const int32_t gr = dofs[r];
    for (PtrT k = rowptr[gr]; k < rowptr[gr + 1]; ++k) {
      const int32_t gc = col[k];
      for (int c = 0; c < np; ++c)
        if (dofs[c] == gc) M[r * np + c] += val[k];
    }
  }
  // LU-ordered elimination (forward), then back-substitution on all columns of X
  for (int j = 0; j < np; ++j) {
    int piv = j;
    if (pivoting) {
      double mx = fabs(M[j * np + j]);
      for (int i = j + 1; i < np; ++i) {
        const double v = fabs(M[i * np + j]);
        if (v > mx) { mx = v; piv = i; }
      }
    }
    if (piv != j)
      for (int c = 0; c < np; ++c) {
        double t = M[j * np + c]; M[j * np + c] = M[piv * np + c]; M[piv * np + c] = t;
        t = X[j * np + c]; X[j * np + c] = X[piv * np + c]; X[piv * np + c] = t;
      }
    const double d = M[j * np + j];
    if (d == 0.0) { atomicAdd(nsing, 1); return; }
    for (int i = j + 1; i < np; ++i) {
      const double l = M[i * np + j] / d;
      if (l != 0.0) {
        for (int c = j + 1; c < np; ++c) M[i * np + c] -= l * M[j * np + c];
        for (int c = 0; c < np; ++c) X[i * np + c] -= l * X[j * np + c];
      }
      M[i * np + j] = 0.0;
    }
  }
  for (int j = np - 1; j >= 0; --j) {
    const double d = M[j * np + j];
    for (int c = 0; c < np; ++c) {
      double s = X[j * np + c];
      for (int k = j + 1; k < np; ++k) s -= M[j * np + k] * X[k * np + c];
      X[j * np + c] = s / d;
    }
  }
}

// ---------------------------------------------------------------------------
// Wave-per-patch inversion (setup): the same LU-ordered elimination as patch_factor_kernel (same operation order per
// entry, hence the same bits), with the n_p x 2 n_p augmented block [M | X] in LDS and lane = column.  Block sources:
//   PSRC_CSR     A[rows_p, cols_p] gathered from the level's device CSR              (BlockJacobiSolvers.jl:160)
//   PSRC_PATTERN the same from the row-pattern form of A (streamed operators never hold a CSR)
//   PSRC_SELL    the same from the SELL-64 arrays (numerical_setup!: the CSR copy was dropped after the first setup)
//   PSRC_DENSE   the caller's own patch matrices, column-major as Julia stores them  (PatchSolvers.jl:137-150:
//                assemble_matrix(biform, assem, trial, test) of the SOLVER's weak form)
// n_p <= 64.  Output: explicit inverse, row-major, at binv[boff[p] - boff0].
// ---------------------------------------------------------------------------
enum PatchSrcKind : int { PSRC_CSR = 0, PSRC_PATTERN = 1, PSRC_DENSE = 2, PSRC_SELL = 3 };
struct PatchSrc {
  const void *rowptr; const int32_t *col; const double *val; int ptr64;                                  // CSR
  const uint16_t *rowpid; const int32_t *rowbase; const int32_t *plen, *poff8; const double *pval; int W; // pattern table (byte offsets)
  const double *dense; int64_t dense_off0;                                                                // blocks of this batch
  const int64_t *soff; const int32_t *scol; const double *sval; const int32_t *rowlen;                    // SELL-64 arrays (value refresh: the CSR stream is gone)
};

template <int SRC>
__global__ __launch_bounds__(64) void patch_invert_kernel(int64_t npatch, const int64_t *__restrict__ pptr,
                                                          const int32_t *__restrict__ prow, const int32_t *__restrict__ pcol,
                                                          const int64_t *__restrict__ boff, int64_t boff0, PatchSrc src,
                                                          int pivoting, double *__restrict__ binv, int max_np,
                                                          int *__restrict__ nsing, const int64_t *__restrict__ plist = nullptr,
                                                          const int64_t *__restrict__ ooff = nullptr)
{
  // plist: invert the listed patches only (representatives of the source de-duplication), block k of the list to binv[ooff[k]]
  extern __shared__ double pi_smem[];
  if ((int64_t)blockIdx.x >= npatch) return;
  const int64_t p = plist ? plist[blockIdx.x] : (int64_t)blockIdx.x;
  const int64_t q0 = pptr[p];
  const int np = (int)(pptr[p + 1] - q0);
  if (np == 0) return;
  double *M = pi_smem, *X = pi_smem + (size_t)max_np * max_np;
  int32_t *cols = reinterpret_cast<int32_t *>(X + (size_t)max_np * max_np);
  const int lane = threadIdx.x;
  if (lane < np) cols[lane] = pcol[q0 + lane];
  for (int e = lane; e < np * np; e += 64) { M[e] = 0.0; X[e] = ((e / np) == (e % np)) ? 1.0 : 0.0; }
  __syncthreads();
  if (lane < np) {
    const int r = lane;
    const int32_t gr = prow[q0 + r];
    if (SRC == PSRC_CSR) {
      int64_t k0, k1;
      if (src.ptr64) { k0 = reinterpret_cast<const int64_t *>(src.rowptr)[gr]; k1 = reinterpret_cast<const int64_t *>(src.rowptr)[gr + 1]; }
      else { k0 = reinterpret_cast<const int32_t *>(src.rowptr)[gr]; k1 = reinterpret_cast<const int32_t *>(src.rowptr)[gr + 1]; }
      for (int64_t k = k0; k < k1; ++k) {
        const int32_t gc = src.col[k];
        for (int c = 0; c < np; ++c)
          if (cols[c] == gc) M[r * np + c] += src.val[k];
      }
    } else if (SRC == PSRC_PATTERN) {
      const int pid = src.rowpid[gr];
      const int len = src.plen[pid];
      const int32_t base = src.rowbase ? src.rowbase[gr] : gr;
      for (int j = 0; j < len; ++j) {
        const int32_t gc = base + src.poff8[(size_t)pid * src.W + j] / 8;
        const double v = src.pval[(size_t)pid * src.W + j];
        for (int c = 0; c < np; ++c)
          if (cols[c] == gc) M[r * np + c] += v;
      }
    } else if (SRC == PSRC_SELL) {
      const int64_t base = src.soff[gr >> 6];
      const int len = src.rowlen[gr];
      for (int j = 0; j < len; ++j) {
        const int64_t q = base + (int64_t)j * 64 + (gr & 63);
        const int32_t gc = src.scol[q];
        const double v = src.sval[q];
        for (int c = 0; c < np; ++c)
          if (cols[c] == gc) M[r * np + c] += v;
      }
    } else {
      const double *B = src.dense + (boff[p] - src.dense_off0);
      for (int c = 0; c < np; ++c) M[r * np + c] = B[r + (size_t)c * np];
    }
  }
  __syncthreads();
  const int c = lane;                                        // this lane's column of M and of X
  for (int j = 0; j < np; ++j) {
    int piv = j;
    if (pivoting) {
      double v = (lane >= j && lane < np) ? fabs(M[lane * np + j]) : -1.0;
      int idx = lane;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(v, o);
        const int i2 = __shfl_xor(idx, o);
        if (v2 > v || (v2 == v && i2 < idx)) { v = v2; idx = i2; }
      }
      piv = idx;                                             // first row holding the maximum (dgetf2's idamax)
    }
    __syncthreads();
    if (piv != j && c < np) {
      double t = M[j * np + c]; M[j * np + c] = M[piv * np + c]; M[piv * np + c] = t;
      t = X[j * np + c]; X[j * np + c] = X[piv * np + c]; X[piv * np + c] = t;
    }
    __syncthreads();
    const double d = M[j * np + j];
    if (d == 0.0) { if (lane == 0) atomicAdd(nsing, 1); return; }   // uniform
    for (int i = j + 1; i < np; ++i) {
      const double l = M[i * np + j] / d;
      __syncthreads();                                       // everyone has read M[i][j] before column j clears it
      if (l != 0.0 && c < np) {
        if (c > j) M[i * np + c] -= l * M[j * np + c];
        X[i * np + c] -= l * X[j * np + c];
      }
      if (c == j) M[i * np + j] = 0.0;
    }
    __syncthreads();
  }
  if (c < np)
    for (int j = np - 1; j >= 0; --j) {
      const double d = M[j * np + j];
      double sacc = X[j * np + c];
      for (int k = j + 1; k < np; ++k) sacc -= M[j * np + k] * X[k * np + c];
      X[j * np + c] = sacc / d;
    }
  __syncthreads();
  double *out = plist ? binv + ooff[blockIdx.x] : binv + (boff[p] - boff0);
  for (int e = lane; e < np * np; e += 64) out[e] = X[e];
}

// ---- source de-duplication of patch blocks taken from a row-pattern operator ---------------------------------------------
// A[rows_p, cols_p] gathered from the pattern form is a function of (n_p; per row r: pattern id of the row, its base column
// relative to cols_p[0]; per column c: cols_p[c] - cols_p[0]) -- patch_invert_kernel<PSRC_PATTERN> reads nothing else.  Patches
// with equal signatures therefore have bitwise equal blocks and bitwise equal inverses: only one per signature is inverted
// (64 of 1.7 x 10^7 on the uniform Q2 mesh of BASELINE config 3).  Hash per patch (wave = patch, lane = row) ...
__device__ __forceinline__ unsigned long long patch_sig_word(int r, int pid, int32_t dbase, int32_t dcol)
{
  unsigned long long a = (unsigned long long)(unsigned)pid * 0x9E3779B97F4A7C15ull ^ (unsigned long long)(uint32_t)dbase * 0xC2B2AE3D27D4EB4Full ^
                         (((unsigned long long)(uint32_t)dcol << 32) | (unsigned)r) * 0x165667B19E3779F9ull;
  a ^= a >> 29; a *= 0xBF58476D1CE4E5B9ull; a ^= a >> 32;
  return a;
}
__global__ __launch_bounds__(256) void patch_sig_hash_kernel(int64_t npatch, const int64_t *__restrict__ pptr, const int32_t *__restrict__ prow,
                                                             const int32_t *__restrict__ pcol, PatchSrc src, unsigned long long *__restrict__ hash)
{
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= npatch) return;
  const int64_t q0 = pptr[p];
  const int np = (int)(pptr[p + 1] - q0);
  unsigned long long h = 0;
  if (lane < np) {
    const int32_t ref = pcol[q0];
    const int32_t gr = prow[q0 + lane];
    h = patch_sig_word(lane, src.rowpid[gr], (src.rowbase ? src.rowbase[gr] : gr) - ref, pcol[q0 + lane] - ref);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
  if (lane == 0) hash[p] = h ^ ((unsigned long long)np * 0xD6E8FEB86659FD93ull);
}
// ... and the exact comparison of every patch with the representative of its group (rep[g] = patch index): hash collisions -> mismatch
__global__ __launch_bounds__(256) void patch_sig_verify_kernel(int64_t npatch, const int64_t *__restrict__ pptr, const int32_t *__restrict__ prow,
                                                               const int32_t *__restrict__ pcol, PatchSrc src, const int32_t *__restrict__ grp,
                                                               const int64_t *__restrict__ rep, int *__restrict__ nmis)
{
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= npatch) return;
  const int64_t q = rep[grp[p]];
  if (q == p) return;
  const int64_t q0 = pptr[p], r0 = pptr[q];
  const int np = (int)(pptr[p + 1] - q0);
  bool same = (int)(pptr[q + 1] - r0) == np;
  if (same && lane < np) {
    const int32_t ref = pcol[q0], rref = pcol[r0];
    const int32_t gr = prow[q0 + lane], rgr = prow[r0 + lane];
    same = src.rowpid[gr] == src.rowpid[rgr] && (src.rowbase ? src.rowbase[gr] : gr) - ref == (src.rowbase ? src.rowbase[rgr] : rgr) - rref &&
           pcol[q0 + lane] - ref == pcol[r0 + lane] - rref;
  }
  if (__any(!same) && lane == 0) atomicAdd(nmis, 1);
}

// CSR -> SELL-64 on the device (setup): one wave per slice, lane = row; entry j of the slice's rows goes to
// soff[slice] + 64 j + lane (coalesced writes).  Padding: value 0.0, column = the row's first column (any valid one).
template <typename PtrT>
__global__ __launch_bounds__(256) void sell_build_kernel(int64_t nrows, int64_t nslices, const PtrT *__restrict__ rowptr,
                                                         const int32_t *__restrict__ col, const double *__restrict__ val,
                                                         const int64_t *__restrict__ soff, int32_t *__restrict__ scol,
                                                         double *__restrict__ sval, int32_t *__restrict__ rowlen)
{
  const int lane = threadIdx.x & 63;
  const int64_t sl = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (sl >= nslices) return;
  const int64_t base = soff[sl];
  const int64_t w = (soff[sl + 1] - base) >> 6;
  const int64_t i = sl * 64 + lane;
  int64_t k0 = 0, len = 0;
  if (i < nrows) { k0 = (int64_t)rowptr[i]; len = (int64_t)rowptr[i + 1] - k0; rowlen[i] = (int32_t)len; }
  const int32_t padcol = len > 0 ? col[k0] : 0;
  for (int64_t j = 0; j < w; ++j) {
    const int64_t q = base + j * 64 + lane;
    if (j < len) { scol[q] = col[k0 + j]; sval[q] = val[k0 + j]; }
    else { scol[q] = padcol; sval[q] = 0.0; }
  }
}

// numerical_setup!: new values into an existing SELL-64 layout (sval[slice, j, lane] = val[rowptr[row] + j]) and the
// Jacobi inverse diagonal recomputed from the refreshed rows.  PtrT = row pointer type of the kept CSR row pointers.
template <typename PtrT>
__global__ void sell_refill_kernel(int64_t nrows, const PtrT *__restrict__ rowptr, const int64_t *__restrict__ soff,
                                   const int32_t *__restrict__ scol, const double *__restrict__ val, double *__restrict__ sval,
                                   double *__restrict__ dinv, int *__restrict__ nzero)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  const int64_t base = soff[i >> 6] + (i & 63);
  const PtrT k0 = rowptr[i], k1 = rowptr[i + 1];
  double d = 0.0;
  for (PtrT k = k0; k < k1; ++k) {
    const int64_t q = base + (int64_t)(k - k0) * 64;
    const double v = val[k];
    sval[q] = v;
    if (scol[q] == (int32_t)i) d += v;
  }
  if (dinv) {
    if (d == 0.0) atomicAdd(nzero, 1);
    dinv[i] = 1.0 / d;
  }
}

// dinv[i] = pdinv[rowpid[i]]: the Jacobi inverse diagonal of a row-pattern operator (no CSR needed)
__global__ void expand_pattern_dinv_kernel(int64_t n, const uint16_t *__restrict__ rowpid, const double *__restrict__ pdinv,
                                           double *__restrict__ dinv)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dinv[i] = pdinv[rowpid[i]];
}

// One wave per patch: lanes = rows of the block (n_p <= 64), b_p staged in LDS.
__global__ __launch_bounds__(kBlock) void patch_apply_kernel(int64_t npatch, const int64_t *__restrict__ pptr,
                                                             const int32_t *__restrict__ pdofs,
                                                             const int64_t *__restrict__ boff,
                                                             const double *__restrict__ binv,
                                                             const double *__restrict__ b,
                                                             double *__restrict__ contrib)
{
  __shared__ double sb[4][64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + w;
  if (p >= npatch) return;
  const int64_t q0 = pptr[p];
  const int np = (int)(pptr[p + 1] - q0);
  if (np == 0) return;
  if (lane < np) sb[w][lane] = b[pdofs[q0 + lane]];
  __builtin_amdgcn_wave_barrier();
  if (lane < np) {
    const double *row = binv + boff[p] + (size_t)lane * np;
    double s = 0.0;
    for (int k = 0; k < np; ++k) s += row[k] * sb[w][k];
    contrib[q0 + lane] = s;
  }
}

// ---- patch-block de-duplication (uniform meshes: a handful of distinct blocks) -------------
// 64-bit hash of every inverse block (bit patterns, FNV-1a style), one thread per patch.
__global__ void block_hash_kernel(int64_t npatch, const int64_t *__restrict__ pptr, const int64_t *__restrict__ boff,
                                  const double *__restrict__ binv, unsigned long long *__restrict__ hash)
{
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npatch) return;
  const int np = (int)(pptr[p + 1] - pptr[p]);
  const unsigned long long *b = reinterpret_cast<const unsigned long long *>(binv + boff[p]);
  unsigned long long h = 1469598103934665603ull ^ (unsigned long long)np;
  for (int k = 0; k < np * np; ++k) { h ^= b[k]; h *= 1099511628211ull; }
  hash[p] = h;
}
// bitwise comparison of every block with its group's representative (hash collisions -> mismatch)
__global__ void block_verify_kernel(int64_t npatch, const int64_t *__restrict__ pptr, const int64_t *__restrict__ boff,
                                    const double *__restrict__ binv, const int64_t *__restrict__ rep, int *__restrict__ nmis)
{
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npatch) return;
  const int64_t q = rep[p];
  if (q == p) return;
  const int np = (int)(pptr[p + 1] - pptr[p]);
  const unsigned long long *a = reinterpret_cast<const unsigned long long *>(binv + boff[p]);
  const unsigned long long *b = reinterpret_cast<const unsigned long long *>(binv + boff[q]);
  bool same = (pptr[q + 1] - pptr[q]) == np;
  for (int k = 0; k < np * np && same; ++k) same = a[k] == b[k];
  if (!same) atomicAdd(nmis, 1);
}
// the same against a compact store: rep[p] = element offset of the representative block
__global__ void block_verify_store_kernel(int64_t npatch, const int64_t *__restrict__ pptr, const int64_t *__restrict__ boff,
                                          const double *__restrict__ binv, const double *__restrict__ store,
                                          const int64_t *__restrict__ rep, int *__restrict__ nmis)
{
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npatch) return;
  const int np = (int)(pptr[p + 1] - pptr[p]);
  const unsigned long long *a = reinterpret_cast<const unsigned long long *>(binv + boff[p]);
  const unsigned long long *b = reinterpret_cast<const unsigned long long *>(store + rep[p]);
  bool same = true;
  for (int k = 0; k < np * np && same; ++k) same = a[k] == b[k];
  if (!same) atomicAdd(nmis, 1);
}
// gather the representatives' blocks into the compact store
__global__ void block_compact_kernel(int64_t nuniq, const int64_t *__restrict__ src_off, const int64_t *__restrict__ dst_off,
                                     const double *__restrict__ binv, double *__restrict__ ubinv)
{
  const int64_t u = blockIdx.x;
  const int64_t len = dst_off[u + 1] - dst_off[u];
  for (int64_t k = threadIdx.x; k < len; k += blockDim.x) ubinv[dst_off[u] + k] = binv[src_off[u] + k];
}

// De-duplicated patch solve.  A workgroup walks CHUNK consecutive patches; the block of the
// chunk's first patch is cached in LDS (on a uniform mesh almost every patch of the chunk
// shares it) and each half-wave (32 lanes, n_p <= 32) solves one patch per step:
// lane = row of the block, b_p staged in LDS.  Blocks that differ from the cached one are
// read from the compact store (a few KB in L2).
constexpr int kPatchChunk = 64;
__global__ __launch_bounds__(kBlock) void patch_apply_dedup_kernel(int64_t npatch, const int64_t *__restrict__ pptr,
                                                                   const int32_t *__restrict__ pdofs,
                                                                   const int32_t *__restrict__ ublock,
                                                                   const int64_t *__restrict__ uboff,
                                                                   const double *__restrict__ ubinv,
                                                                   const double *__restrict__ b,
                                                                   double *__restrict__ contrib)
{
  __shared__ double sB[32 * 32];
  __shared__ double sb[8][32];
  const int64_t p0 = (int64_t)blockIdx.x * kPatchChunk;
  const int64_t p1 = min(p0 + (int64_t)kPatchChunk, npatch);
  const int cached = ublock[p0];
  const int64_t co = uboff[cached];
  const int clen = (int)(uboff[cached + 1] - co);
  for (int k = threadIdx.x; k < clen; k += kBlock) sB[k] = ubinv[co + k];
  __syncthreads();
  const int hw = threadIdx.x >> 5, lane = threadIdx.x & 31;   // 8 half-waves
  // Every half-wave solves TWO patches per step (p and p + 8): the chunk's block row is read from LDS once for both
  // (the LDS pipe bounds this kernel: 27 x 27 block entries per patch), their right-hand sides are gathered together.
  __shared__ double sb2[8][32];
  for (int64_t p = p0 + hw; p < p1; p += 16) {
    const int64_t pB = p + 8;
    const bool hasB = pB < p1;
    const int64_t q0 = pptr[p], q0B = hasB ? pptr[pB] : 0;
    const int np = (int)(pptr[p + 1] - q0), npB = hasB ? (int)(pptr[pB + 1] - q0B) : 0;
    const int ub = ublock[p], ubB = hasB ? ublock[pB] : -1;
    double bA = 0.0, bB = 0.0;
    if (lane < np) bA = b[pdofs[q0 + lane]];
    if (lane < npB) bB = b[pdofs[q0B + lane]];
    if (lane < np) sb[hw][lane] = bA;
    if (lane < npB) sb2[hw][lane] = bB;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (np > 0 && np == npB && ub == cached && ubB == cached) {
      if (lane < np) {
        const double *rowp = sB + lane * np;
        double sA = 0.0, sBv = 0.0;
        for (int k = 0; k < np; ++k) {
          const double rk = rowp[k];
          sA += rk * sb[hw][k];
          sBv += rk * sb2[hw][k];
        }
        contrib[q0 + lane] = sA;
        contrib[q0B + lane] = sBv;
      }
    } else {
      if (lane < np) {
        const double *rowp = (ub == cached) ? sB + lane * np : ubinv + uboff[ub] + (size_t)lane * np;
        double s = 0.0;
        for (int k = 0; k < np; ++k) s += rowp[k] * sb[hw][k];
        contrib[q0 + lane] = s;
      }
      if (lane < npB) {
        const double *rowp = (ubB == cached) ? sB + lane * npB : ubinv + uboff[ubB] + (size_t)lane * npB;
        double s = 0.0;
        for (int k = 0; k < npB; ++k) s += rowp[k] * sb2[hw][k];
        contrib[q0B + lane] = s;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// Generic fallback for patches with n_p > 64: one workgroup per patch.
__global__ __launch_bounds__(kBlock) void patch_apply_big_kernel(int64_t npatch, const int64_t *__restrict__ pptr,
                                                                 const int32_t *__restrict__ pdofs,
                                                                 const int64_t *__restrict__ boff,
                                                                 const double *__restrict__ binv,
                                                                 const double *__restrict__ b,
                                                                 double *__restrict__ contrib)
{
  const int64_t p = blockIdx.x;
  const int64_t q0 = pptr[p];
  const int np = (int)(pptr[p + 1] - q0);
  for (int r = threadIdx.x; r < np; r += kBlock) {
    const double *row = binv + boff[p] + (size_t)r * np;
    double s = 0.0;
    for (int k = 0; k < np; ++k) s += row[k] * b[pdofs[q0 + k]];
    contrib[q0 + r] = s;
  }
}

// dof-centric gather: dx_i = omega * sum contrib[inc[k]] ; optionally x_i += dx_i
__global__ void patch_gather_kernel(int64_t n, const int64_t *__restrict__ iptr, const int64_t *__restrict__ inc,
                                    const double *__restrict__ contrib, double omega, int relax,
                                    double *__restrict__ dx, double *__restrict__ x)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int64_t k = iptr[i]; k < iptr[i + 1]; ++k) s += contrib[inc[k]];
  if (relax) {
    s = omega * s;
    x[i] = x[i] + s;
  }
  dx[i] = s;
}

// The same gather with the incidence lists in sliced-ELL form (64 dofs per slice, column-major, 32-bit slots,
// padding -> a slot that holds 0.0): the index loads are coalesced.  Slots of a dof keep their ascending patch order.
__global__ void patch_gather_sell_kernel(int64_t n, const int64_t *__restrict__ soff, const int32_t *__restrict__ sinc,
                                         const double *__restrict__ contrib, double omega, int relax,
                                         double *__restrict__ dx, double *__restrict__ x)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t slice = i >> 6;
  const int lane = (int)(i & 63);
  const int64_t base = soff[slice];
  const int w = (int)((soff[slice + 1] - base) >> 6);
  const int32_t *ip = sinc + base + lane;
  double s = 0.0;
  int j = 0;
  for (; j + 4 <= w; j += 4) {                               // four index loads, then four gathers in flight; summed in slot order
    const int32_t i0 = ip[(int64_t)j * 64], i1 = ip[(int64_t)(j + 1) * 64], i2 = ip[(int64_t)(j + 2) * 64], i3 = ip[(int64_t)(j + 3) * 64];
    const double c0 = contrib[i0], c1 = contrib[i1], c2 = contrib[i2], c3 = contrib[i3];
    s += c0; s += c1; s += c2; s += c3;
  }
  for (; j < w; ++j) s += contrib[ip[(int64_t)j * 64]];
  if (relax) {
    s = omega * s;
    x[i] = x[i] + s;
  }
  dx[i] = s;
}

} // namespace gmg
