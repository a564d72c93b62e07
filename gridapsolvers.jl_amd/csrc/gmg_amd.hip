// gmg_amd.hip -- host side of libgmgamd.so: the C ABI declared in include/gmg_amd.h.
//
// Owns, per handle, the device-resident multigrid hierarchy (CSR A_l, D^-1_l,
// explicit P_l and R_l = P_l^T, patch inverse blocks, dense coarse inverse,
// work vectors) and drives the gfx950 kernels of kernels.hpp on ONE HIP stream.
// Host code only orchestrates: every fp64 operation on vectors / matrices of the
// hot path runs on the GPU.  There is no CPU fallback: without a HIP device
// gmg_create fails with GMG_ERR_HIP.
//
// Reference call structure reproduced (GridapSolvers.jl v0.7.1, src/LinearSolvers):
//   gmg_apply        = solve!(x,::GMGNumericalSetup,b)        GMGLinearSolvers.jl:612-645
//   Solver::cycle    = gmg_v_cycle!/gmg_w_cycle!/gmg_f_cycle!  GMGLinearSolvers.jl:468-610
//   Solver::smooth   = solve!(x,::RichardsonSmootherNS,r)      RichardsonSmoothers.jl:84-98
//   gmg_cg_solve     = solve!(x,::CGNumericalSetup,b)          Krylov/CGSolvers.jl:73-120
//   gmg_fgmres_solve = solve!(x,::FGMRESNumericalSetup,b)      Krylov/FGMRESSolvers.jl:130-199
//   ConvLog          = ConvergenceLog / SolverTolerances       SolverInterfaces/ConvergenceLogs.jl:101-150,
//                                                              SolverTolerances.jl:97-128
#include "../../include/gmg_amd.h"
#include "kernels.hpp"
#include "comm.hpp"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <exception>
#include <string>
#include <thread>
#include <atomic>
#include <chrono>
#include <unordered_map>
#include <vector>

using namespace gmg;

namespace {

std::string g_last_error;

struct GmgError {
  int code;
  std::string msg;
};

#define HIP_CHECK(expr)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      throw GmgError{GMG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)};         \
  } while (0)

#define REQUIRE(cond, code, text)                  \
  do {                                             \
    if (!(cond)) throw GmgError{(code), (text)};   \
  } while (0)

// ----------------------------------------------------------------------------
// host-side sparse containers and input conversion
// ----------------------------------------------------------------------------
struct HostCSR {
  int64_t nrows = 0, ncols = 0;
  std::vector<int64_t> ptr;
  std::vector<int32_t> col;
  std::vector<double> val;
  int64_t nnz() const { return ptr.empty() ? 0 : ptr.back(); }
};

int64_t read_index(const void *p, int64_t i, int bytes)
{
  return bytes == 8 ? reinterpret_cast<const int64_t *>(p)[i] : (int64_t) reinterpret_cast<const int32_t *>(p)[i];
}

// setup-time host loops over independent chunks (slices, columns): plain std::thread fan-out.
// run_threads(nt, body): body(0) .. body(nt-1), body(0) on the calling thread.  Exception-safe on both sides -- an exception in a
// worker (a REQUIRE, std::bad_alloc on a box with little host memory) is carried to the caller and rethrown after every thread
// has been joined, and a thread that cannot be created (EAGAIN under a container's pid limit) has its share run by the caller:
// either of the two used to end the whole process through std::terminate (an uncaught exception in a std::thread body; a
// joinable std::thread destroyed while the vector unwinds) -- SIGABRT with no message, which no ABI-level try/catch can intercept.
template <typename F>
void run_threads(int nt, F &&body)
{
  if (nt <= 1) { body(0); return; }
  std::vector<std::thread> th;
  std::exception_ptr first;
  std::mutex mu;
  auto guarded_body = [&](int t) {
    try { body(t); }
    catch (...) { std::lock_guard<std::mutex> lk(mu); if (!first) first = std::current_exception(); }
  };
  int started = 1;                                    // index 0 is the caller's
  try {
    th.reserve((size_t)nt - 1);
    for (; started < nt; ++started) th.emplace_back(guarded_body, started);
  } catch (...) { /* thread creation failed: the caller runs the indices that have no thread */ }
  guarded_body(0);
  for (int t = started; t < nt; ++t) guarded_body(t);
  for (auto &x : th) x.join();
  if (first) std::rethrow_exception(first);
}
inline int host_threads(int64_t cap)
{
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(hw, 32u), cap));
}
template <typename F>
void parallel_for(int64_t n, F &&fn)
{
  const int nt = host_threads(n / 256);
  if (nt <= 1) { for (int64_t i = 0; i < n; ++i) fn(i); return; }
  const int64_t chunk = (n + nt - 1) / nt;
  run_threads(nt, [&](int t) { for (int64_t i = t * chunk; i < std::min(n, (t + 1) * chunk); ++i) fn(i); });
}

// the same for a handful of BIG work items (one chunk of rows each): one thread per item, at most 32
template <typename F>
void parallel_chunks(int64_t n, F &&fn)
{
  const int nt = host_threads(n);
  if (nt <= 1) { for (int64_t i = 0; i < n; ++i) fn(i); return; }
  std::atomic<int64_t> next(0);
  run_threads(nt, [&](int) { for (int64_t i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); });
}

// Accepts {CSR|CSC} x {0|1}-based x {int32|int64} (SparseMatrixCSC{Float64,Int}: CSC,1,8;
// SparseMatrixCSR{1,Float64,Int32}: CSR,1,4 ...) and produces 0-based CSR.
HostCSR convert_input(int64_t nrows, int64_t ncols, int64_t nnz, const void *ptr, const void *idx,
                      const double *val, int layout, int base, int bytes)
{
  REQUIRE(ptr && idx && val, GMG_ERR_INVALID, "null matrix array");
  REQUIRE(nrows >= 0 && ncols >= 0 && nnz >= 0, GMG_ERR_INVALID, "negative matrix size");
  REQUIRE(bytes == 4 || bytes == 8, GMG_ERR_INVALID, "index_bytes must be 4 or 8");
  REQUIRE(base == 0 || base == 1, GMG_ERR_INVALID, "index_base must be 0 or 1");
  REQUIRE(layout == GMG_CSR || layout == GMG_CSC, GMG_ERR_INVALID, "layout must be GMG_CSR or GMG_CSC");
  REQUIRE(ncols < (int64_t)INT32_MAX && nrows < (int64_t)INT32_MAX, GMG_ERR_UNSUPPORTED,
          "more than 2^31-1 rows/cols per device");
  const int64_t nmajor = layout == GMG_CSR ? nrows : ncols;
  const int64_t nminor = layout == GMG_CSR ? ncols : nrows;
  REQUIRE(read_index(ptr, 0, bytes) == base, GMG_ERR_INVALID, "pointer array does not start at index_base");
  REQUIRE(read_index(ptr, nmajor, bytes) - base == nnz, GMG_ERR_INVALID, "pointer array end != nnz");
  HostCSR out;
  out.nrows = nrows;
  out.ncols = ncols;
  out.ptr.assign(nrows + 1, 0);
  out.col.resize(nnz);
  out.val.resize(nnz);
  if (layout == GMG_CSR) {
    for (int64_t i = 0; i <= nrows; ++i) out.ptr[i] = read_index(ptr, i, bytes) - base;
    for (int64_t i = 0; i < nrows; ++i) REQUIRE(out.ptr[i] <= out.ptr[i + 1], GMG_ERR_INVALID, "row pointers not monotone");
    std::atomic<int> badcol(0);
    const int64_t nchunk = (nnz + (1 << 20) - 1) >> 20;
    parallel_chunks(nchunk, [&](int64_t t) {
      const int64_t k0 = t << 20, k1 = std::min(nnz, k0 + (1 << 20));
      for (int64_t k = k0; k < k1; ++k) {
        const int64_t c = read_index(idx, k, bytes) - base;
        if (c < 0 || c >= nminor) { badcol.store(1); return; }
        out.col[k] = (int32_t)c;
      }
      std::memcpy(out.val.data() + k0, val + k0, sizeof(double) * (size_t)(k1 - k0));
    });
    REQUIRE(badcol.load() == 0, GMG_ERR_INVALID, "column index out of range");
  } else {
    // transpose CSC -> CSR by counting sort (keeps columns sorted within a row)
    for (int64_t k = 0; k < nnz; ++k) {
      const int64_t r = read_index(idx, k, bytes) - base;
      REQUIRE(r >= 0 && r < nminor, GMG_ERR_INVALID, "row index out of range");
      out.ptr[r + 1]++;
    }
    for (int64_t i = 0; i < nrows; ++i) out.ptr[i + 1] += out.ptr[i];
    std::vector<int64_t> fill(out.ptr.begin(), out.ptr.end() - 1);
    for (int64_t c = 0; c < ncols; ++c) {
      const int64_t k0 = read_index(ptr, c, bytes) - base, k1 = read_index(ptr, c + 1, bytes) - base;
      REQUIRE(k0 <= k1, GMG_ERR_INVALID, "column pointers not monotone");
      for (int64_t k = k0; k < k1; ++k) {
        const int64_t r = read_index(idx, k, bytes) - base;
        const int64_t q = fill[r]++;
        out.col[q] = (int32_t)c;
        out.val[q] = val[k];
      }
    }
  }
  return out;
}

HostCSR transpose(const HostCSR &A)
{
  HostCSR T;
  T.nrows = A.ncols;
  T.ncols = A.nrows;
  T.ptr.assign(T.nrows + 1, 0);
  const int64_t nnz = A.nnz();
  T.col.resize(nnz);
  T.val.resize(nnz);
  for (int64_t k = 0; k < nnz; ++k) T.ptr[A.col[k] + 1]++;
  for (int64_t i = 0; i < T.nrows; ++i) T.ptr[i + 1] += T.ptr[i];
  std::vector<int64_t> fill(T.ptr.begin(), T.ptr.end() - 1);
  for (int64_t i = 0; i < A.nrows; ++i)
    for (int64_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
      const int64_t q = fill[A.col[k]]++;
      T.col[q] = (int32_t)i;
      T.val[q] = A.val[k];
    }
  return T;
}

// ----------------------------------------------------------------------------
// Streamed operators (gmg_set_operator_rows): the caller hands over consecutive row blocks and the library keeps only the
// row-pattern form -- a 16-bit pattern id (+ a 32-bit base column) per row and the dictionary of distinct rows -- never a
// CSR.  Host memory is O(block), which is what lets BASELINE config 3 (8.5e9 nonzeros) through the setup.
// mode 0: column offsets relative to the row index (square level matrices); mode 1: relative to the row's first column.
// ----------------------------------------------------------------------------
struct PatStream {
  int mode = 0;
  int64_t nrows = 0, ncols = 0, rows_seen = 0, nnz = 0, wmax = 0;
  std::vector<int32_t> len, start, off;
  std::vector<uint64_t> val;
  std::unordered_map<uint64_t, std::vector<int32_t>> index;
  std::vector<uint16_t> rowpid;
  std::vector<int32_t> rowbase;
  bool complete() const { return rows_seen == nrows; }
};

// ----------------------------------------------------------------------------
// device containers
// ----------------------------------------------------------------------------
struct DevCSR {
  mutable Z2Geo z2;                 // fused pair of sweeps (sells_z2sweep_kernel): geometry, valid when z2_ok > 0
  mutable Z2Geo zbx;                // single sweep of a constant-coefficient box (sells_boxsweep_kernel): the same grid, its own chains
  mutable int z2_ok = -1;           // -1 not examined yet, 0 no, 1 yes
  int64_t nrows = 0, ncols = 0, nnz = 0;
  bool ptr64 = false;
  void *rowptr = nullptr;
  int32_t *col = nullptr;
  double *val = nullptr;
  int32_t *blk_row = nullptr;
  int64_t *blk_nz = nullptr;
  int nblocks = 0;
  int lanes_log2 = 0;
  int tile = kTile;
  // SELL-64 copy (built when the padding is small): see kernels.hpp sell_kernel
  bool sell = false;
  int64_t *soff = nullptr;
  int32_t *scol = nullptr;
  double *sval = nullptr;
  int32_t *rowlen = nullptr;
  int nslices = 0;
  int64_t zpad = 0;
  // compressed stream (SELL-C): 16-bit column offsets per slice column, 8-bit value codes
  int64_t nnz_model = -1;   // nnz of the caller's operator when A was split (byte model)
  int64_t band = 0;         // SELL-64 / SELL-O: largest |column - row|
  bool comp_idx = false, vdict = false;
  int64_t *poff = nullptr;
  uint16_t *pidx = nullptr;
  uint8_t *pcode = nullptr;
  int32_t *pbase = nullptr;
  uint8_t *smode = nullptr;
  double *dict = nullptr;
  int64_t zpack = 0;
  double stream_bytes_per_nnz = 12.0;
  // row-pattern dictionary (SELL-P): see kernels.hpp sellp_kernel
  bool pat = false;
  int pat_np = 0, pat_w = 0;
  uint16_t *rowpid = nullptr;
  int32_t *rowbase = nullptr;   // nullptr: offsets relative to the row index
  int32_t *plen = nullptr, *ppoff = nullptr;
  double *ppval = nullptr;
  // shared-offset ("stencil") form of the pattern table: see kernels.hpp sells_kernel
  bool pat_shared = false;
  PatEntry *ptab = nullptr;
  double *ptab8 = nullptr;        // the coefficients of ptab alone (8 B per entry): unmasked sweep kernels
  int32_t *prun = nullptr;
  int pat_nruns = 0, pat_minoff = 0, pat_maxoff = 0;
  double *pdinv = nullptr;      // [np] 1/diag per pattern (nullptr: some pattern has no diagonal entry)
  double pdinv_u = 0.0;         // that number
  bool pdinv_uniform = false;   // every pattern has the same 1/diag, bit for bit (constant-coefficient operator on a uniform mesh): sells_rsweep_kernel
  bool pat_generic = false;     // the per-lane-offset table (sellp_kernel) fits LDS
  int pat_k = 3;                // offsets per run of the shared form
  bool pat_coded = false;       // shared form with one-byte value codes + dictionary
  uint8_t *pcodes = nullptr;
  double *pdict = nullptr;
  uint32_t *prunmask = nullptr;
  // per-workgroup value tables of the coded form (sells_kernel<..., WL>): pattern lists of the launch geometry they were built for
  mutable std::vector<int32_t> h_run_off;   // host copy of prun (tile sweep geometry), fetched on first use
  mutable int wl_state = 0;     // 0 not looked at, 1 in use, 2 not applicable (too many patterns in one chunk / switched off)
  mutable int wl_req = 0, wl_nwg = 0, wl_wpb = 0, wl_max = 0;   // requested / launched workgroups
  mutable uint16_t *wl_pids = nullptr;
  // the same for the z-walk form of the wide-row kernels (sellw_zwalk_kernel): one list per workgroup of four chains
  mutable int wz_state = 0, wz_T = 0, wz_nwg = 0, wz_max = 0;   // 0 not tried, 1 lists ready, 2 does not apply
  mutable uint16_t *wz_pids = nullptr;
  mutable int32_t *wz_cnt = nullptr;
  mutable ZWalkGeo wz_geo = {0, 0, 0, 0, 0};
  // auxiliary walk table (round 6): a plain (K = 3) table whose offsets form the 5 x 5 run grid but whose values are too many for 8-bit
  // codes also carries 16-bit codes over those 25 runs of five -- sellw_zwalk_kernel decodes per workgroup, so the dictionary never has to
  // fit LDS; the per-slice kernels keep using the plain table
  bool aux_ok = false;
  int32_t *aux_run = nullptr;        // [25] run offsets
  uint16_t *aux_codes16 = nullptr;   // [np * 125]
  double *aux_dict = nullptr;        // [65536], [65535] = 0.0
  uint32_t *aux_rmask = nullptr;     // [np]
  std::vector<int32_t> h_aux_run;
  mutable int32_t *wl_cnt = nullptr;
  // offset-pattern layout (SELL-O): SELL-64 value stream + a 16-bit offset-pattern id per row, no column stream
  bool opat = false;
  uint16_t *orowpid = nullptr;
  int32_t *orowbase = nullptr, *opoff = nullptr;
  int opat_np = 0, opat_w = 0;
  // kernel + template arguments of the fused sweep last launched on this operator (x-update variant excluded): bench.py only
  // attaches committed PMC traffic to a kernel whose signature matches the profiled one (gmg_sweep_signature)
  mutable char sweep_sig[112] = {0};
  void note_sweep(const char *fmt, ...) const __attribute__((format(printf, 2, 3)))
  {
    if (sweep_sig[0]) return;
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(sweep_sig, sizeof(sweep_sig), fmt, ap);
    va_end(ap);
  }
  bool present() const { return rowptr != nullptr; }
};

enum SmootherKind { SM_JACOBI = 0, SM_PATCH = 1 };

struct Smoother {
  int kind = SM_JACOBI;
  int niter = 10;             // GMGLinearSolvers.jl:52 default RichardsonSmoother(Jacobi,10)
  double omega = 1.0;         // RichardsonSmoothers.jl:28 default
  // patch data (host copy kept until setup; shared between the copies of a smoother)
  int patch_kind = GMG_PATCH_LU;
  struct Tables {
    std::vector<int64_t> pptr;
    std::vector<int32_t> prow;     // patch_rows: b[rows_p] is the local right-hand side   (PatchSolvers.jl:237-240)
    std::vector<int32_t> pcol;     // patch_cols: x[cols_p] += x_p (:296); empty = patch_rows
    // caller-assembled patch matrices (column-major n_p x n_p, concatenated) or their lu! factors + LAPACK pivots (1-based)
    std::vector<double> blocks;
    std::vector<int32_t> piv;
    bool has_blocks = false, are_factors = false;
  };
  std::shared_ptr<Tables> tab;
  // device
  int64_t npatch = 0;
  int max_np = 0;
  int64_t *d_pptr = nullptr;
  int32_t *d_pdofs = nullptr;
  int32_t *d_pcol = nullptr;    // == d_pdofs when patch_cols == patch_rows
  int64_t *d_boff = nullptr;
  double *d_binv = nullptr;
  int64_t n_binv = 0, n_ubinv = 0, n_uboff = 0;   // allocation sizes (value refresh releases and rebuilds the blocks)
  int64_t n_isoff = 0, n_isinc = 0;
  // de-duplicated blocks (uniform meshes): patch -> unique block id, compact block store
  bool dedup = false;
  int64_t nuniq = 0;
  int32_t *d_ublock = nullptr;
  int64_t *d_uboff = nullptr;
  double *d_ubinv = nullptr;
  int64_t *d_iptr = nullptr, *d_inc = nullptr;
  int64_t *d_isoff = nullptr;   // sliced-ELL incidence (patch_gather_sell_kernel); nullptr: CSR-like lists
  int32_t *d_isinc = nullptr;
  double *d_contrib = nullptr;
  bool built = false;
  // the additive-Schwarz operator M = sum_p R_p^T inv(A_pp) R_p itself, in row-pattern form (uniform meshes: a few hundred
  // distinct rows): dx = M r is then ONE pattern mat-vec instead of patch solve + contribution buffer + gather
  bool use_M = false;
  DevCSR M;
  std::vector<int32_t> h_ublock;   // host copies of the de-duplication result (setup only)
  std::vector<int64_t> h_uboff;
  // every device array above was freed with the handle's allocations
  void reset_device()
  {
    use_M = false; M = DevCSR();
    d_pptr = nullptr; d_pdofs = nullptr; d_pcol = nullptr; d_boff = nullptr; d_binv = nullptr;
    d_ublock = nullptr; d_uboff = nullptr; d_ubinv = nullptr; d_iptr = nullptr; d_inc = nullptr; d_isoff = nullptr; d_isinc = nullptr;
    d_contrib = nullptr;
    n_binv = n_ubinv = n_uboff = 0; dedup = false; nuniq = 0; npatch = 0; built = false;
  }
};

// rows of an operator that reference ghost columns, in the form ghost_fix_sell_kernel reads (slices of 64 rows, column-major, values as
// doubles or as one-byte codes into a dictionary)
struct GhostFix {
  int64_t nb = 0;
  int32_t *rows = nullptr, *len = nullptr, *scol = nullptr;
  int64_t *soff = nullptr;
  double *sval = nullptr, *dict = nullptr;
  uint8_t *scode = nullptr;
};

struct Level {
  // s = omega*(Dinv*r) already written into sbuf[0] by the kernel that produced r (restriction / r -= A dx)
  bool s0_ready = false;
  const double *s0_src = nullptr;
  double s0_omega = 0.0;
  HostCSR hA, hP, hR;
  bool hasA = false, hasP = false, hasR = false;
  std::shared_ptr<PatStream> sA, sP, sR;   // streamed operators (hA/hP/hR then carry the shape only)
  // streamed matrix of an own | ghost level: the own x own part goes to the stream, the entries in ghost columns of the rows that
  // have any are kept as the small CSR the boundary fix-up applies (gmg_set_operator_rows splits every block)
  bool rs_forbid = false;           // several ranks: the r-gather sweep form is a joint decision (gmg_solver::setup)
  // overlapping layout, gmg_set_partition_overlap_hints: the caller states GEOMETRY only -- how many exact node layers the halo holds,
  // how many of them one sweep of this level's smoother consumes, how far the restriction of an owned coarse row reads -- and (b) that P's
  // rows are complete wherever r_own -= (A dxh)_own reads dxh.  Whether the consistent!(r) before the restriction can be skipped is
  // derived per smoothing pass from the depth and the niter of the smoother that just ran (skip_r_after); cleared by
  // gmg_set_partition_overlap and gmg_set_smoother_* of the level.
  int ovl_layers = 0, ovl_sweep_reach = 0, ovl_r_reach = 0;
  bool ovl_skip_dx = false;
  void clear_overlap_hints() { ovl_layers = ovl_sweep_reach = ovl_r_reach = 0; ovl_skip_dx = false; }
  bool sA_split = false;
  std::vector<int32_t> g_rows, g_col;
  std::vector<int64_t> g_ptr;
  std::vector<double> g_val;
  // how the caller interleaved own and ghost columns inside the rows of g_rows (1 = ghost column), row t at g_mptr[t]..g_mptr[t+1]:
  // what gmg_update_values needs to route a full-row value array into the stream values and g_val
  std::vector<uint8_t> g_mask;
  std::vector<int64_t> g_mptr;
  void clear_split() { sA_split = false; g_rows.clear(); g_col.clear(); g_val.clear(); g_ptr.assign(1, 0); g_mask.clear(); g_mptr.assign(1, 0); }
  bool values_dirty = false;               // gmg_update_values since the last setup
  int input_layout = GMG_CSR;              // layout gmg_set_matrix was called with
  std::vector<uint32_t> csc_perm;          // gmg_update_values_csc: CSR position (the handle's order) of every CSC entry, built on first use
  DevCSR A, P, R;
  double *dinv = nullptr;
  Smoother pre, post;
  Smoother pcorr;                 // patch tables of a patch-corrected prolongation (kind == SM_PATCH when set)
  bool has_pcorr = false;
  HostCSR hG;                     // rhs form of the correction when it is not the level operator (StokesGMG.jl:125-127: graddiv)
  bool hasG = false;
  DevCSR G;
  double *ptmp = nullptr, *pcor = nullptr;
  bool post_shares_pre = true;
  int64_t n = 0;                  // owned rows
  int64_t nvec = 0;               // vector length = owned + ghost
  HaloPlan halo;                  // neighbour exchange plan (multi-GPU)
  // own x ghost part of A on the rows that have ghost columns (A itself then holds own x own only)
  bool split = false;
  struct HostXfer { struct ::gmg_solver *solver; int level; } hostctx{nullptr, 0};   // hipLaunchHostFunc payload
  int64_t nbnd = 0;
  int32_t *gh_rows = nullptr;
  int64_t *gh_ptr = nullptr;
  int32_t *gh_col = nullptr;
  double *gh_val = nullptr;
  // the same entries in slices of 64 boundary rows, column-major (ghost_fix_sell_kernel); null: the CSR form is used
  int32_t *gh_len = nullptr, *gh_scol = nullptr;
  int64_t *gh_soff = nullptr;
  double *gh_sval = nullptr;
  uint8_t *gh_scode = nullptr;      // one byte per entry into gh_dict (<= 256 distinct values) instead of gh_sval
  double *gh_dict = nullptr;
  // the restriction of an own | ghost level split the same way (round 5): R keeps the own columns -- and with them a row-pattern
  // layout -- and runs while consistent!(r) is in flight, the ghost columns of the coarse rows next to the box faces are added afterwards
  bool r_split = false;
  GhostFix rfix;
  double *x = nullptr;            // correction at this level (levels > 0)
  double *rbuf[2] = {nullptr, nullptr};
  double *sbuf[2] = {nullptr, nullptr}; // s = omega*(dinv.*r) ping-pong (one-gather sweep)
  uint32_t *pflags = nullptr;     // persistent smoothing pass (sells_smooth_kernel): one progress word per workgroup, 64 B apart
  int pf_nwg = 0;
  uint32_t pf_epoch = 1;
  double *dx = nullptr;
  double *rcur = nullptr;         // buffer holding the current residual after a cycle
};

// ConvergenceLog + SolverTolerances (ConvergenceLogs.jl:42-150, SolverTolerances.jl:97-128)
struct ConvLog {
  int maxiter = 0;
  double atol = 0, rtol = 0;
  int num_iters = 0;
  std::vector<double> residuals;
  void configure(int mi, double a, double r)
  {
    maxiter = mi; atol = a; rtol = r;
    residuals.assign((size_t)std::max(mi, 0) + 1, 0.0);
  }
  bool finished(int niter, double e_a, double e_r) const
  {
    return (niter >= maxiter) || (e_r < rtol) || (e_a < atol); // strict <, SolverTolerances.jl:126-128
  }
  bool init(double r0)
  {
    num_iters = 0;
    std::fill(residuals.begin(), residuals.end(), 0.0);
    residuals[0] = r0;
    return finished(num_iters, r0, 1.0);
  }
  bool update(double r)
  {
    num_iters += 1;
    if ((size_t)num_iters < residuals.size()) residuals[num_iters] = r;
    return finished(num_iters, r, r / residuals[0]);
  }
  int finalize(double r) const
  {
    const double r_rel = r / residuals[0];
    if (r_rel < rtol) return GMG_CONVERGED_RTOL;
    if (r < atol) return GMG_CONVERGED_ATOL;
    if (num_iters >= maxiter) return GMG_DIVERGED_MAXITER;
    return GMG_DIVERGED_BREAKDOWN;
  }
  void export_to(gmg_result *res, double *hist, int cap, double last) const
  {
    if (res) {
      res->niters = num_iters;
      res->flag = finalize(last);
      res->res0 = residuals[0];
      res->res = last;
    }
    if (hist)
      for (int i = 0; i <= num_iters && i < cap; ++i) hist[i] = residuals[i];
  }
};

constexpr int kScalarSlots = 4096;

} // namespace

// ----------------------------------------------------------------------------
// the solver object behind gmg_handle_t
// ----------------------------------------------------------------------------
// operator / preconditioner callbacks of the Krylov drivers (cg_core, fgmres_core)
struct KrylovOps {
  std::function<void(double *x, const double *b, double *r)> resid;          // r = b - A x
  std::function<void(double *x, double *y)> apply;                           // y = A x
  std::function<int(double *x, double *y, double *parts)> apply_dot;         // optional: y = A x + first stage of dot(x, y) -> #partials (0: not done)
  std::function<void(double *z, const double *r, double known_res)> precond; // empty: Pl === nothing (CG) / Pr === nothing (FGMRES)
  std::function<void(double *z, const double *r)> precond_left;              // FGMRES only: Pl of KrylovUtils.jl:14-18,46-50
  double *zl = nullptr;                                                      // its work vector (FGMRESSolvers.jl:66)
};

struct gmg_block_solver;
static void block_forget(gmg_block_solver *B, gmg_solver *g);   // block.inc.hpp
static double cg_core(gmg_solver &S, int64_t n, const double *db, double *dx, double *w, double *p, double *z, double *r,
                      const KrylovOps &ops, bool flexible, ConvLog &log);

struct gmg_solver {
  gmg_block_solver *attached_to = nullptr;   // block preconditioner that borrowed this handle (gmg_block_set_diag_gmg + setup)
  int device = 0;
  hipStream_t stream = nullptr;      // the stream work is issued on (a block solver re-points it at its own)
  hipStream_t own_stream = nullptr;  // the stream this handle created and destroys
  int nlev = 0;
  std::vector<Level> lev;            // nlev levels + one slot (index nlev) for a separate Krylov operator, see has_outer()
  std::string err;
  // Finest level in the overlapping layout (gmg_set_partition_overlap on level 0): the Krylov solver keeps the caller's own | ghost
  // vectors and its own finest operator (lev[nlev]: gmg_set_matrix / gmg_set_partition with GMG_LEVEL_KRYLOV), the preconditioner's
  // level 0 lives in the overlapping numbering; own2loc maps every owned entry to its place there (gmg_set_krylov_map).
  std::vector<int64_t> h_own2loc;
  int64_t *d_own2loc = nullptr;
  // Levels sub_from .. rep_from-1 on a RANK SUBSET (gmg_set_redistribution; np_per_level / redistribute!, ModelHierarchies.jl:80-148,
  // GridTransferOperators.jl:447-532): level sub_from exists in the glued partition of all ranks (P / R of level sub_from-1 are built
  // against it) and in the subset's; two p2p plans move the restricted residual to the subset owners and the correction back to the
  // glued own AND ghost entries.  Ranks outside the subset hold nothing of those levels and only shadow their collectives.
  struct Redist {
    bool present = false, member = false;
    int64_t n_glue_own = 0, n_glue_ghost = 0;
    HaloPlan to_sub, from_sub;                      // snd_idx: local ids in the source space, rcv_idx: in the destination space
    std::vector<int64_t> h_self[4];                 // to_sub (src, dst), from_sub (src, dst): entries that stay on this rank
    int64_t *d_self[4] = {nullptr, nullptr, nullptr, nullptr};
    double *glue_r = nullptr, *glue_x = nullptr;
  } redist;
  int sub_from = -1;
  int64_t n_redist = 0;                             // redistributions issued (gmg_get_comm_stats counts them with the exchanges)
  bool inactive(int l) const { return redist.present && !redist.member && l >= sub_from && l < rep_from; }
  bool has_outer() const { return (int)lev.size() > nlev && lev[nlev].hasA; }
  int kl() const { return has_outer() ? nlev : 0; }                 // the level the Krylov solvers' operator and vectors belong to
  int64_t user_n() const { return lev[kl()].n; }
  bool setup_done = false;
  bool structure_dirty = true;       // anything but values changed since the last setup
  bool was_setup = false;
  void touch() { setup_done = false; structure_dirty = true; }
  int64_t dev_bytes = 0;
  std::vector<void *> allocs;

  // Layout / schedule policy of this handle (gmg_set_option).  The reference configures a solver through constructor keywords only
  // (GMGLinearSolvers.jl:48-58); a process-wide environment variable cannot choose a layout per solver, so every switch the library
  // reads is a per-handle option first.  The environment variable of the same name, when set, overrides it (debugging / A-B runs).
  std::map<std::string, double> options;
  double opt_num(const char *name, double dflt) const
  {
    const char *s = std::getenv(name);
    if (s && *s) return std::atof(s);
    auto it = options.find(name);
    return it != options.end() ? it->second : dflt;
  }
  int opt_int(const char *name, int dflt) const { return (int)opt_num(name, (double)dflt); }

  // GMGLinearSolver kwargs (GMGLinearSolvers.jl:56-58)
  int mode = GMG_MODE_PRECONDITIONER, cycle_type = GMG_V_CYCLE;
  ConvLog log;
  double log_last = 0.0;             // last residual norm the GMG's own log saw (NaN: not evaluated, see gmg_set_verbose)
  int verbose = 0;                   // GMGLinearSolver(...; verbose): > 0 keeps the log complete on every path

  // coarse solver (coarsest_solver kwarg, GMGLinearSolvers.jl:54,423-434)
  double *d_Ainv = nullptr;
  int coarse_kind = GMG_COARSE_DENSE_INVERSE;
  // what the setup actually builds: a dense-inverse request (LUSolver()) on a coarsest level of >= GMG_COARSE_AUTO_CG_MIN dofs is
  // served by CGSolver(JacobiLinearSolver()) on the device, run to rtol 1e-10 -- the n x n inverse of such a level costs seconds of
  // setup and gigabytes (29 791 dofs, BASELINE config 3: 1.9 s, 7.1 GB) for a solve the outer Krylov method only needs to ~1e-7
  int coarse_eff = GMG_COARSE_DENSE_INVERSE;
  bool coarse_auto = false;
  void resolve_coarse_kind()
  {
    coarse_eff = coarse_kind; coarse_auto = false;
    const int amin = opt_int("GMG_COARSE_AUTO_CG_MIN", 20000);
    if (coarse_kind == GMG_COARSE_DENSE_INVERSE && amin > 0 && nlev > 0 && lev[nlev - 1].hA.nrows >= amin) { coarse_eff = GMG_COARSE_CG_JACOBI; coarse_auto = true; }
  }
  int coarse_maxiter = 1000;                 // CGSolver defaults, CGSolvers.jl:19
  double coarse_atol = 1e-12, coarse_rtol = 1e-6;
  gmg_coarse_solve_fn coarse_fn = nullptr;
  void *coarse_ctx = nullptr;
  ConvLog coarse_log;
  double coarse_last = 0.0;
  double *cc_w = nullptr, *cc_p = nullptr, *cc_z = nullptr, *cc_r = nullptr;   // CGSolvers.jl:42-48 on the coarsest level
  double *h_cr = nullptr, *h_cx = nullptr;   // pinned staging of the host callback
  bool reduce_local = false;                 // reductions over a REPLICATED vector: no all-reduce across ranks
  int krylov_depth = 0;                      // nesting of cg_core calls (each level owns its scalar slots)
  // distributed runs: levels >= rep_from are REPLICATED (every rank holds the global operators and
  // computes them redundantly, no halo traffic on small levels).  The restricted residual of the last
  // distributed level is assembled with one all-reduce (own rows scattered by global id).
  int rep_from = -1;
  std::vector<int64_t> h_rep_gid;     // global row id (level rep_from) of every row this rank's R produces
  int64_t *d_rep_gid = nullptr;
  double *d_rep_tmp = nullptr;
  double *h_rep_full = nullptr;       // pinned, host transport only

  // inter-GPU transport
  Comm comm;
  hipStream_t comm_stream = nullptr;   // halo traffic overlapping the own x own mat-vec (RCCL only)
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  int64_t n_exchanges = 0;             // halo exchanges issued since gmg_create (gmg_get_comm_stats)
  int64_t n_allreduces = 0;            // all-reduces issued (scalars and the replication boundary)
  int halo_fuse_pack = 1;              // GMG_HALO_FUSE_PACK: the boundary fix-up of a sweep packs the next sweep's send buffer
  int overlap = 1;                     // GMG_OVERLAP
  int host_async = 0;                  // GMG_HOST_ASYNC: run the overlapped schedule with the host transport (tests)
  double *cg_x = nullptr;       // solution with ghost space (distributed runs)

  // reductions
  double *d_partials = nullptr;
  double *d_scalars = nullptr;
  double *h_scalars = nullptr; // pinned

  // Krylov work vectors (CGSolvers.jl:42-48 ; FGMRESSolvers.jl:58-70)
  double *cg_w = nullptr, *cg_p = nullptr, *cg_z = nullptr, *cg_r = nullptr;
  std::vector<double *> fg_V, fg_Z;
  // staging buffers for host-memory callers
  double *st_b = nullptr, *st_x = nullptr;
  std::vector<double *> st_extra;

  // tuning
  int xcd_remap = 0;
  int lanes_override = -1;
  int use_sell = 1;     // GMG_SELL: SELL-64 layout for matrices with padding <= sell_maxpad
  double sell_maxpad = 1.25;
  int use_idx16 = 1;    // GMG_IDX16: 16-bit column offsets where a slice column spans < 65536
  int use_vdict = 1;    // GMG_VDICT: 8-bit value dictionary when the matrix has <= 256 distinct values
  int sell_block = 0;   // GMG_SELL_BLOCK: 0 = auto (256 threads on big levels, 64 on small ones)
  int sell_un = 6;      // GMG_SELL_UN: independent (col,val,gather) triples in flight per lane
  int nt_loads = 1;     // GMG_NT: non-temporal matrix stream
  int nt_rowwise = 1;   // GMG_NT_ROWWISE: SELL-64 / SELL-O sweeps of big levels read r, x, 1/diag and write r, x non-temporally
  int xcd_remap_big = -1; // GMG_XCD_REMAP_BIG: workgroup -> rows mapping on levels whose gathered vector exceeds the L2s (SELL-64 / SELL-O): 0 launch order, 1 contiguous eighths, n > 1 chunks of n workgroups per XCD, -1 chunk from the operator's gather reach
  int64_t big_rows = 4000000;   // GMG_BIG_ROWS: a level is "big" above this many rows (8 B/row against 8 x 4 MB of L2)
  int sell_defer = 1;   // GMG_SELL_DEFER: x updated every second sweep in the SELL-64 / SELL-O sweeps as well
  bool big_level(const DevCSR &M) const { return M.nrows > big_rows; }
  // Workgroup -> row-range mapping of the SELL-64 / SELL-O kernels on a level whose gathered vector exceeds the L2s.  Launch order
  // (0) lets all eight XCDs stream inside one window of the arrays, but neighbouring workgroups sit on different XCDs, so every L2
  // fetches (nearly) the whole gathered vector: 1.095 x the algorithmic bytes at 256^3 (7 075 MB per sweep, profiles/r04_tuning.md).
  // Contiguous eighths (1) fetch it once but stream from eight far-apart windows.  Chunked (n > 1: n consecutive workgroups per XCD
  // inside a group of 8 chunks) with a chunk two gather reaches deep (auto, -1): 6 483 MB = 1.003 x at 256^3 in the same time as
  // launch order (1 106 us); four reaches deep moves the same bytes 4 % slower, so the chunk is kept as small as the reach allows.
  int remap_for_big(const DevCSR &M, int rows_per_wg, int nwg) const
  {
    if (xcd_remap_big >= 0) return xcd_remap_big == 1 ? xcd_remap : xcd_remap_big;
    const int64_t want = (2 * std::max<int64_t>(M.band, 1) + rows_per_wg - 1) / rows_per_wg;
    if (want < 2 || want * 8 * 4 > nwg) return 0;           // no locality to win / the window would span a quarter of the level: launch order
    return (int)want;
  }
  int use_pattern = 1;  // GMG_PATTERN: row-pattern dictionary (SELL-P) when the matrix has few distinct rows
  int pat_un = 9;       // GMG_PAT_UN: gathers in flight per lane in sellp_kernel
  int pat_wgs = 2048;   // GMG_PAT_WGS: resident workgroups of the persistent sellp launch
  int pat_rb = 3;       // GMG_PAT_RB: runs (of 3 offsets) loaded per batch in sells_kernel (3 or 9)
  int use_opattern = 1; // GMG_OPATTERN: offset-pattern layout (SELL-O) for SELL-64 matrices whose column structure repeats
  int pat_batched = 1;  // GMG_PAT_BATCHED: the restructured sweep kernel (sells_sweep_kernel) for plain shared-offset tables
  int pat_nb = 0;       // GMG_PAT_NB: slices per batch of that kernel (0 = auto: 1, or 2 on levels with >= 200000 slices)
  int pat_small_wpb = 4;   // GMG_PAT_SMALL_WPB: waves per workgroup of sells_kernel on levels with < 8192 slices (table staging amortised)
  int pat_small_wpb2 = 2;  // GMG_PAT_SMALL_WPB2: the same for sellp_kernel
  int pat_emit = 1;     // GMG_PAT_EMIT: restriction / r -= A dx kernels also write the next smoothing pass' s_0
  int64_t pat_coded_min_rows = 500000;   // GMG_PAT_CODED_MIN_ROWS
  int gj_mfma = 1;      // GMG_GJ_MFMA: trailing update of the device coarse inversion on the FP64 matrix cores
  int pat_tile = 0;     // GMG_PAT_TILE: r-gather sweeps share their gathers through LDS (sells_tsweep_kernel): 0 never (default since the 64-register pair sweep beats it), 1 on levels of >= GMG_PAT_TILE_ROWS rows, 2 wherever it applies
  int64_t pat_tile_rows = 3500000;   // "big" row-pattern levels: pair sweep / mat-vecs at one slice per wave in eight-wave workgroups (and the tile sweep with pat_tile = 1)
  int pat_wide = 1;     // GMG_PAT_WIDE: coded (wide-row) operators decode the patterns of each workgroup's chunk into a plain LDS value table
  int pat_strict = 1;   // GMG_PAT_STRICT: fused sweeps keep the per-entry mask (exact zero products even for non-finite vectors); 0 = 8-byte table entries, 2-3 % faster
  int persist_wpb_min = 1;   // GMG_PERSIST_WPB: smallest workgroup (in waves) of a one-launch pass
  int persist = 1;      // GMG_PERSIST: small levels run a whole smoothing pass in one launch (sells_smooth_kernel)
  int persist_fenced = 0; // GMG_PERSIST_FENCED: progress words published with release / polled with acquire semantics (agent scope)
  int persist_max_slices = 0;  // GMG_PERSIST_MAX_SLICES (0: what one workgroup per CU holds)
  int n_cus = 0;
  uint32_t *d_perr_dev = nullptr;                  // device-memory twin (the kernel's end-of-pass check)
  uint32_t *h_perr = nullptr, *d_perr = nullptr;   // pinned + mapped: a bounded wait of the persistent kernel timed out
  int pat_bcast = 1;    // GMG_PAT_BCAST: tile sweep: slices whose DPP rows are single-pattern take their coefficients by row broadcast (no LDS read per tap)
  int pat_r2 = 1;       // GMG_PAT_R2: r-gather sweeps with two rows per lane (sells_r2sweep_kernel)
  int pat_zwalk = 1;    // GMG_PAT_ZWALK: the pair sweep as a walk along the slowest grid direction (kernels.hpp: sells_zsweep_kernel); 1: levels of >= pat_zwalk_rows rows, 2: every level
  int64_t pat_zwalk_wide_rows = 1000000;   // GMG_PAT_ZWALK_WIDE_ROWS: smallest wide-row level that takes the walk -- it wins wherever there are enough chains (it removes gathers,
                                           // not bytes): Q2 64^3 (2.05e6 rows) 18.6 -> 16.0 ms per solve, 96^3 (7.0e6) 40.3 -> 30.0, 128^3 89 -> 70
  int pat_zwalk_wide = 1; // GMG_PAT_ZWALK_WIDE: the wide-row (Q2) operator applications of those levels (sellw_zwalk_kernel)
  int pat_zwalk_mv = 1; // GMG_PAT_ZWALK_MV: also the operator mat-vecs of those levels
  // GMG_PAT_FUSE2 (opt-in): two sweeps per pass over the data (kernels.hpp: sells_z2sweep_kernel) on grid levels of one GPU: 1 = levels of >=
  // pat_fuse2_rows rows, 2 = every level that qualifies, 0 = off (default).  Bit-identical to the single sweeps and 32 instead of 56 bytes
  // per row and pair -- but measured SLOWER on MI355X (profiles/r06_ab_fuse2.txt): 128^3 21.9 against 18.5 us per sweep, 288^3 185 against
  // 156: the rim rows (W / (W - 2) x (T + 2) / T more sweep-k rows), 77 % lane use on 287-row lines and one workgroup barrier per plane
  // cost more issue slots than the saved bytes buy; the single sweeps already run at their memory bound at 288^3.
  // pat_fuse2_w = waves (grid lines) per workgroup, pat_fuse2_t = planes per block (0: chosen from the level's size); pat_fuse2_box = 0
  // keeps the per-row patterns in LDS even on constant-coefficient boxes (the BC form reads one coefficient set per wave from the arguments)
  int pat_fuse2 = 0, pat_fuse2_w = 0, pat_fuse2_t = 0;
  // GMG_PAT_BOX (opt-in): constant-coefficient box levels (verified row by row at setup) sweep with one coefficient set per wave from the kernel
  // arguments -- no pattern ids, no LDS (kernels.hpp: sells_boxsweep_kernel): 1 = levels of pat_box_min_rows .. pat_box_max_rows rows, 2 = every
  // level that qualifies, 0 = off (default).  Measured on MI355X (profiles/r06_ab_box.txt): 128^3 20.1-21.0 us per sweep against 18.8 for
  // sells_r2sweep_kernel, 288^3 208 against 165 for sells_zsweep_kernel -- the per-row LDS coefficient reads it removes are not what bounds the
  // single sweeps.  pat_box_t = planes per chain (0: from the level's size)
  int pat_box = 0, pat_box_t = 0;
  int64_t pat_box_min_rows = 1000000, pat_box_max_rows = 9000000;
  int64_t pat_fuse2_rows = 1000000;
  int pat_zwalk_T = 12; // GMG_PAT_ZWALK_T: planes per chain (288^3: 8 / 12 / 16 / 24 / 32 -> 149 / 114 / 120 / 118 / 155 us for the x-untouched form)
  int64_t pat_zwalk_rows = 9000000;   // the walk pays once r, r', x and the pattern ids (26 B per row) no longer fit the 256 MB Infinity Cache: 224^3 (1.09e7 rows) 8.13 -> 7.59 ms per solve,
                                      // 192^3 (7.0e6) 4.89 -> 5.23-5.50 with every chain length (profiles/r05_sizes.txt)
  int pat_r2_wgs = 0;   // GMG_PAT_R2_WGS: its resident workgroups (0: four per CU, eight with pat_r2_occ)
  int pat_r2_occ = 2;   // GMG_PAT_R2_OCC: the 64-register form of the pair sweep (rolled run loop, eight waves per SIMD); 2: workgroups of eight waves at one slice per wave (big levels)
  int pat_r2mv_dot = 1; // GMG_PAT_R2MV_DOT: dot(p, A p) of CG formed by the mat-vec kernel (first stage; order of the sum differs from dot_partial_kernel's)
  int pat_pair_p = 1;   // GMG_PAT_PAIR_P: prolongation + correction with two rows per lane (sellp_pair_addto_kernel), levels of >= pat_r2mv_min rows
  int pat_r2mv = 1;     // GMG_PAT_R2MV: mat-vecs (y = A x, y -= A x, y = b - A x) with two rows per lane (sells_r2mv_kernel)
  int64_t pat_r2mv_min = 100000;   // GMG_PAT_R2MV_MIN: smallest level (rows) that takes it
  int pat_fma = 0;      // GMG_PAT_FMA: fused multiply-add taps in the row-pattern sweeps (one rounding per tap: not the reference's mul! arithmetic)
  int pat_rsweep = 1;   // GMG_PAT_RSWEEP: sweeps of uniform-diagonal row-pattern levels gather r itself (no s vector: sells_rsweep_kernel)
  int pat_defer = 1;    // GMG_PAT_DEFER: x updated every second sweep (shared-offset pattern kernel)
  int pat_dinv = 1;     // GMG_PAT_DINV: Jacobi inverse diagonal from the pattern table instead of its vector
  int pat_shared = 1;   // GMG_PAT_SHARED: shared-offset (stencil) form when the offsets are row-relative
  int one_gather_sweep = 1;   // GMG_ONE_GATHER: sweep gathers s = w*Dinv*r (1) or r and Dinv (0)
  int tile = kTile;

  // profiling of the fused sweep
  int prof_level = -1;
  int prof_stride = 7;          // GMG_PROF_STRIDE (odd: the sweeps alternate between two variants, an even stride would sample one of them only)
  uint64_t prof_seq = 0;
  std::vector<hipEvent_t> prof_ev;
  size_t prof_used = 0;
  double prof_ms = 0.0;
  int64_t prof_launches = 0, prof_fused = 0;
  bool prof_patch = false;      // the timed launches were `r -= A dx` mat-vecs of a patch-smoother sweep, not fused Jacobi sweeps
  std::vector<int> prof_w;     // sweeps bracketed by each event pair (1, or niter for a pass run as one launch)
  std::vector<int8_t> prof_xm; // variant (xmode 0 / 1 / 2) of the sweep each event pair brackets
  double prof_ms_v[3] = {0, 0, 0}; int64_t prof_n_v[3] = {0, 0, 0};

  // ---- memory -------------------------------------------------------------
  template <typename T>
  T *dalloc(size_t count)
  {
    void *p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T) + 64; // slack for vector loads
    HIP_CHECK(hipMalloc(&p, bytes));
    allocs.push_back(p);
    dev_bytes += (int64_t)bytes;
    return reinterpret_cast<T *>(p);
  }
  template <typename T>
  T *upload(const std::vector<T> &v)
  {
    T *p = dalloc<T>(v.size());
    // blocking copy: callers pass temporaries (setup path, not timed)
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
  }
  template <typename T>
  T *upload_padded(const std::vector<T> &v, size_t pad)
  {
    T *p = dalloc<T>(v.size() + pad);
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(p + v.size(), 0, pad * sizeof(T)));
    return p;
  }
  // give device memory back early (setup-only arrays)
  template <typename T>
  void release(T *&p, size_t count)
  {
    if (!p) return;
    auto it = std::find(allocs.begin(), allocs.end(), (void *)p);
    if (it != allocs.end()) {
      allocs.erase(it);
      (void)hipFree((void *)p);
      dev_bytes -= (int64_t)(std::max<size_t>(count, 1) * sizeof(T) + 64);
    }
    p = nullptr;
  }
  // a SELL matrix never streams its CSR copy: drop (col,val) once D^-1 / patch blocks are built
  void drop_csr_stream(DevCSR &M)
  {
    if (!M.sell) return;
    HIP_CHECK(hipStreamSynchronize(stream));
    release(M.col, (size_t)M.nnz + 4096);
    release(M.val, (size_t)M.nnz + 4096);
    if (M.vdict && M.sval) release(M.sval, (size_t)M.zpad);          // values come from the dictionary
  }
  double *dvec(int64_t n)
  {
    double *p = dalloc<double>((size_t)n);
    HIP_CHECK(hipMemsetAsync(p, 0, sizeof(double) * (size_t)std::max<int64_t>(n, 1), stream));
    return p;
  }
  void free_all()
  {
    dump_host_steps();
    for (void *p : allocs) (void)hipFree(p);
    allocs.clear();
    d_perr_dev = nullptr;
    dev_bytes = 0;
    for (auto &L : lev) {
      L.A = DevCSR(); L.P = DevCSR(); L.R = DevCSR(); L.G = DevCSR();
      L.dinv = L.x = L.dx = L.rcur = nullptr;
      L.rbuf[0] = L.rbuf[1] = nullptr;
      L.ptmp = L.pcor = nullptr; L.pcorr.built = false;
      L.split = false; L.nbnd = 0; L.gh_rows = nullptr; L.gh_ptr = nullptr; L.gh_col = nullptr; L.gh_val = nullptr;
      L.gh_len = nullptr; L.gh_scol = nullptr; L.gh_soff = nullptr; L.gh_sval = nullptr; L.gh_scode = nullptr; L.gh_dict = nullptr;
      L.r_split = false; L.rfix = GhostFix();
      L.sbuf[0] = L.sbuf[1] = nullptr;
      L.pflags = nullptr; L.pf_nwg = 0; L.pf_epoch = 1;
      for (Smoother *sp : {&L.pre, &L.post, &L.pcorr}) sp->reset_device();
      L.s0_ready = false;
    }
    d_Ainv = d_partials = d_scalars = nullptr;
    d_partials2 = nullptr;
    cc_w = cc_p = cc_z = cc_r = nullptr;
    d_rep_gid = nullptr; d_rep_tmp = nullptr; cg_x = nullptr; d_own2loc = nullptr;
    redist.glue_r = redist.glue_x = nullptr;
    for (auto &q : redist.d_self) q = nullptr;
    for (HaloPlan *H : {&redist.to_sub, &redist.from_sub}) { H->d_snd_idx = nullptr; H->d_sendbuf = nullptr; H->d_recvbuf = nullptr; H->d_pk_ptr = nullptr; H->d_pk_slot = nullptr; }
    for (auto &L : lev) { L.halo.d_snd_idx = nullptr; L.halo.d_sendbuf = nullptr; L.halo.d_recvbuf = nullptr; L.halo.d_pk_ptr = nullptr; L.halo.d_pk_slot = nullptr; }
    cg_w = cg_p = cg_z = cg_r = st_b = st_x = nullptr;
    fg_V.clear(); fg_Z.clear(); st_extra.clear();
    setup_done = false;
  }

  // ---- uploads --------------------------------------------------------------
  DevCSR upload_csr(const HostCSR &H)
  {
    DevCSR D;
    D.nrows = H.nrows; D.ncols = H.ncols; D.nnz = H.nnz();
    D.ptr64 = D.nnz >= (int64_t)INT32_MAX || opt_int("GMG_FORCE_PTR64", 0) != 0;   // 64-bit row pointers for >= 2^31 nnz (config 3 scale)
    if (D.ptr64) D.rowptr = upload(H.ptr);
    else {
      std::vector<int32_t> p32(H.ptr.begin(), H.ptr.end());
      D.rowptr = upload(p32);
    }
    // + one tile of slack: the stream kernels load whole tiles unconditionally
    D.col = upload_padded(H.col, 4096);
    D.val = upload_padded(H.val, 4096);
    // lanes per row in the reduce phase: ~ (avg nnz/row)/8, power of two in [1,64]
    const double avg = H.nrows ? (double)D.nnz / (double)H.nrows : 1.0;
    int lg = 0;
    while (lg < 6 && (double)(1 << (lg + 1)) * 6.0 <= avg) ++lg;
    if (lanes_override >= 0) lg = std::min(lanes_override, 6);
    D.tile = tile;
    // workgroup row ranges: greedy fill of the LDS tile, at most 256>>lg rows so that every
    // row owns its G lanes for the whole kernel (single-pass reduce)
    const int64_t max_rows = kBlock >> lg;
    std::vector<int32_t> blk;
    blk.push_back(0);
    int64_t r = 0;
    while (r < H.nrows) {
      int64_t e = r;
      const int64_t base = H.ptr[r];
      while (e < H.nrows && H.ptr[e + 1] - base <= D.tile && (e - r) < max_rows) ++e;
      if (e == r) e = r + 1; // single long row
      blk.push_back((int32_t)e);
      r = e;
    }
    D.nblocks = (int)blk.size() - 1;
    D.blk_row = upload(blk);
    {
      std::vector<int64_t> bnz(blk.size() + 1);
      for (size_t i = 0; i < blk.size(); ++i) bnz[i] = H.ptr[blk[i]];
      bnz[blk.size()] = bnz[blk.size() - 1];
      D.blk_nz = upload(bnz);
    }
    build_sell(H, D);
    D.lanes_log2 = lg;
    return D;
  }

  // Row-pattern detection (setup): rows are equal when their lengths, their column offsets (relative to
  // the row index, mode 0, or to their first column, mode 1) and the BITS of their values are equal.
  // Chunks of rows are scanned in parallel with thread-local tables that are merged in chunk order
  // (deterministic ids).  Gives up as soon as the table outgrows LDS.
  // ignore_values: rows are equal when their lengths and column offsets are (values may differ): the offset-pattern layout
  // of variable-coefficient operators on structured meshes (SELL-O, sello_kernel)
  bool detect_patterns(const HostCSR &H, int mode, std::vector<uint16_t> &rowpid, std::vector<int32_t> &rowbase,
                       std::vector<int32_t> &plen, std::vector<int32_t> &poff, std::vector<double> &pval, int &W,
                       bool allow_big, bool &generic_fits, bool ignore_values = false)
  {
    constexpr int kMaxTableBytes = 48 * 1024;
    generic_fits = false;
    const int64_t n = H.nrows;
    int64_t wmax = 0;
    for (int64_t i = 0; i < n; ++i) wmax = std::max(wmax, H.ptr[i + 1] - H.ptr[i]);
    if (wmax == 0 || wmax > 1024) return false;
    W = (int)wmax;
    // allow_big: the caller may still find a compact (shared-offset, coded) table for many / wide patterns
    const int max_np = allow_big ? 4096 : (int)std::min<int64_t>(65534, kMaxTableBytes / (12 * wmax + 4) - 1);
    if (max_np < 1) return false;
    struct Local {
      std::vector<int32_t> len, start;       // per local pattern
      std::vector<int32_t> off;
      std::vector<uint64_t> val;
      std::vector<uint64_t> hash;
      std::unordered_map<uint64_t, std::vector<int32_t>> index;
    };
    const int T = host_threads((n + 4095) / 4096);
    std::vector<Local> loc((size_t)T);
    std::vector<int32_t> lid((size_t)n);
    std::atomic<bool> fail(false);
    const int64_t per = (n + T - 1) / T;
    auto row_equal = [&](const Local &L, int32_t p, int64_t i, int32_t base) {
      const int64_t k0 = H.ptr[i], len = H.ptr[i + 1] - k0;
      if (L.len[p] != (int32_t)len) return false;
      const int32_t st = L.start[p];
      for (int64_t j = 0; j < len; ++j) {
        uint64_t bits;
        std::memcpy(&bits, &H.val[k0 + j], 8);
        if (L.off[st + j] != H.col[k0 + j] - base || (!ignore_values && L.val[st + j] != bits)) return false;
      }
      return true;
    };
    parallel_chunks(T, [&](int64_t t) {
      Local &L = loc[(size_t)t];
      for (int64_t i = t * per; i < std::min(n, (t + 1) * per); ++i) {
        if (fail.load(std::memory_order_relaxed)) return;
        const int64_t k0 = H.ptr[i], len = H.ptr[i + 1] - k0;
        const int32_t base = mode == 0 ? (int32_t)i : (len > 0 ? H.col[k0] : 0);
        uint64_t h = 1469598103934665603ull ^ (uint64_t)len;
        for (int64_t j = 0; j < len; ++j) {
          uint64_t bits;
          std::memcpy(&bits, &H.val[k0 + j], 8);
          h = (h ^ (uint64_t)(uint32_t)(H.col[k0 + j] - base)) * 1099511628211ull;
          if (!ignore_values) h = (h ^ bits) * 1099511628211ull;
          h ^= h >> 29;
        }
        auto &bucket = L.index[h];
        int32_t found = -1;
        for (int32_t p : bucket)
          if (row_equal(L, p, i, base)) { found = p; break; }
        if (found < 0) {
          if ((int)L.len.size() >= max_np) { fail.store(true); return; }
          found = (int32_t)L.len.size();
          L.len.push_back((int32_t)len);
          L.start.push_back((int32_t)L.off.size());
          L.hash.push_back(h);
          for (int64_t j = 0; j < len; ++j) {
            uint64_t bits;
            std::memcpy(&bits, &H.val[k0 + j], 8);
            L.off.push_back(H.col[k0 + j] - base);
            L.val.push_back(bits);
          }
          bucket.push_back(found);
        }
        lid[(size_t)i] = found;
      }
    });
    if (fail.load()) return false;
    // merge in chunk order
    Local G;
    std::vector<std::vector<int32_t>> l2g((size_t)T);
    for (int t = 0; t < T; ++t) {
      const Local &L = loc[(size_t)t];
      l2g[(size_t)t].resize(L.len.size());
      for (size_t p = 0; p < L.len.size(); ++p) {
        auto &bucket = G.index[L.hash[p]];
        int32_t found = -1;
        for (int32_t q : bucket) {
          if (G.len[q] != L.len[p]) continue;
          bool eq = true;
          for (int32_t j = 0; j < L.len[p] && eq; ++j)
            eq = G.off[G.start[q] + j] == L.off[L.start[p] + j] && (ignore_values || G.val[G.start[q] + j] == L.val[L.start[p] + j]);
          if (eq) { found = q; break; }
        }
        if (found < 0) {
          if ((int)G.len.size() >= max_np) return false;
          found = (int32_t)G.len.size();
          G.len.push_back(L.len[p]);
          G.start.push_back((int32_t)G.off.size());
          G.off.insert(G.off.end(), L.off.begin() + L.start[p], L.off.begin() + L.start[p] + L.len[p]);
          G.val.insert(G.val.end(), L.val.begin() + L.start[p], L.val.begin() + L.start[p] + L.len[p]);
          bucket.push_back(found);
        }
        l2g[(size_t)t][p] = found;
      }
    }
    const int np = (int)G.len.size();
    rowpid.resize((size_t)n);
    if (mode == 1) rowbase.resize((size_t)n); else rowbase.clear();
    parallel_chunks(T, [&](int64_t t) {
      for (int64_t i = t * per; i < std::min(n, (t + 1) * per); ++i) {
        rowpid[(size_t)i] = (uint16_t)l2g[(size_t)t][(size_t)lid[(size_t)i]];
        if (mode == 1) rowbase[(size_t)i] = H.ptr[i + 1] > H.ptr[i] ? H.col[H.ptr[i]] : 0;
      }
    });
    plen = G.len;
    plen.push_back(0);                                      // trailing empty pattern for the lanes past nrows
    const int un = pat_un_eff();
    W = (W + un - 1) / un * un;                             // row stride: the kernel reads UN entries at a time
    generic_fits = (int64_t)(np + 1) * (12 * W + 4) <= kMaxTableBytes;
    if (!generic_fits && !allow_big) return false;
    poff.assign((size_t)(np + 1) * W, 0);
    pval.assign((size_t)(np + 1) * W, 0.0);
    for (int p = 0; p < np; ++p)
      for (int32_t j = 0; j < G.len[p]; ++j) {
        poff[(size_t)p * W + j] = 8 * G.off[G.start[p] + j]; // byte offsets (32-bit lane offset + uniform base)
        std::memcpy(&pval[(size_t)p * W + j], &G.val[G.start[p] + j], 8);
      }
    return true;
  }
  static int pat_un_of(int u) { return u >= 27 ? 27 : u >= 14 ? 14 : u >= 9 ? 9 : u >= 6 ? 6 : 3; }
  int pat_un_eff() const { return pat_un_of(pat_un); }
  bool build_pattern(const HostCSR &H, DevCSR &D)
  {
    // byte offsets into the gathered vector are 32-bit: ncols * 8 < 2^31
    if (!use_pattern || H.nrows < 64 || D.nnz == 0 || H.ncols >= (int64_t)(1 << 28)) return false;
    std::vector<uint16_t> rowpid;
    std::vector<int32_t> rowbase, plen, poff;
    std::vector<double> pval;
    int W = 0;
    bool generic = false;
    bool ok = (H.ncols >= H.nrows) && detect_patterns(H, 0, rowpid, rowbase, plen, poff, pval, W, pat_shared && one_gather(), generic);
    if (ok) {
      if (pat_shared) build_shared_offsets(D, plen, poff, pval, W, H.ncols, generic);
      if (!generic && !D.pat_shared) ok = false;            // neither table fits LDS
      // the coded kernel walks the whole superset of offsets (Q2: 125 taps per row where 62 are stored on average): it only
      // pays where the matrix stream is the bottleneck, i.e. on big levels; small ones keep SELL-C
      if (ok && !generic && D.pat_coded && H.nrows < pat_coded_min_rows) { ok = false; D.pat_shared = false; D.pat_coded = false; return false; }
    }
    if (!ok) {
      D.pat_shared = false;
      ok = detect_patterns(H, 1, rowpid, rowbase, plen, poff, pval, W, false, generic);
    }
    if (!ok) return false;
    upload_pattern(D, H.nrows, rowpid, rowbase, plen, poff, pval, W, generic, false);
    return true;
  }
  void upload_pattern(DevCSR &D, int64_t nrows, const std::vector<uint16_t> &rowpid, const std::vector<int32_t> &rowbase,
                      const std::vector<int32_t> &plen, const std::vector<int32_t> &poff, const std::vector<double> &pval, int W,
                      bool generic, bool keep_table)
  {
    D.rowpid = upload_padded(rowpid, 64);
    D.rowbase = rowbase.empty() ? nullptr : upload_padded(rowbase, 64);
    D.pat_generic = generic;
    // keep_table: the (offset,value) table also serves the patch-block extraction of operators that hold no CSR
    if (generic || keep_table) { D.plen = upload(plen); D.ppoff = upload(poff); D.ppval = upload(pval); }
    D.pat_np = (int)plen.size(); D.pat_w = W;
    D.nslices = (int)((nrows + 63) / 64);
    D.pat = true; D.sell = true;
    D.zpad = D.nnz;
    D.stream_bytes_per_nnz = (rowbase.empty() ? 2.0 : 6.0) * (double)nrows / (double)std::max<int64_t>(D.nnz, 1);
  }

  // ---- streamed operators ---------------------------------------------------------------------------------------------
  // One more block of rows: hash every row (parallel, dictionary read-only), then insert the rows whose pattern is new
  // (sequential; a handful per operator).  Pattern ids are assigned in row order => deterministic.
  void stream_append(PatStream &P, int64_t row0, int64_t nr, const void *ptr, const void *idx, const double *val, int base, int bytes)
  {
    REQUIRE(row0 == P.rows_seen, GMG_ERR_STATE, "row blocks must arrive in order (expected row " + std::to_string(P.rows_seen) + ")");
    REQUIRE(nr >= 0 && row0 + nr <= P.nrows, GMG_ERR_INVALID, "row block exceeds the operator");
    REQUIRE(ptr && (nr == 0 || (idx && val)), GMG_ERR_INVALID, "null row block");
    const int64_t p0 = read_index(ptr, 0, bytes);
    REQUIRE(p0 == base, GMG_ERR_INVALID, "row pointers of a block start at index_base");
    constexpr int kMaxPatterns = 4096;
    std::vector<int32_t> found((size_t)nr, -1);
    std::vector<uint64_t> hashes((size_t)nr);
    std::atomic<int> bad(0);
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(64, (nr + 16383) / 16384));
    const int64_t per = (nr + T - 1) / T;
    auto row_ref = [&](int64_t i, int64_t k0, int64_t len) {
      return P.mode == 0 ? row0 + i : (len > 0 ? read_index(idx, k0, bytes) - base : (int64_t)0);
    };
    auto row_is = [&](int32_t q, int64_t k0, int64_t len, int64_t ref) {
      if (P.len[q] != (int32_t)len) return false;
      const int32_t st = P.start[q];
      for (int64_t j = 0; j < len; ++j) {
        uint64_t bits;
        std::memcpy(&bits, &val[k0 + j], 8);
        if (P.off[st + j] != (int32_t)(read_index(idx, k0 + j, bytes) - base - ref) || P.val[st + j] != bits) return false;
      }
      return true;
    };
    // Rounds: (a) in parallel, every unresolved row is looked up in the dictionary (read-only) -- the first round also
    // validates the rows and computes their hashes --, each chunk remembering the FIRST row of every hash it could not
    // resolve; (b) those few rows are inserted sequentially in row order (pattern ids = order of first appearance, as a
    // sequential scan would give).  A whole matrix handed over as one block needs two rounds; later blocks of a stream one.
    std::vector<int64_t> wmax_t((size_t)T, 0);
    for (int round = 0;; ++round) {
      std::vector<std::vector<std::pair<int64_t, uint64_t>>> fresh((size_t)T);
      parallel_chunks(T, [&](int64_t t) {
        std::unordered_map<uint64_t, int64_t> first;
        struct Hit { uint64_t h; int32_t q; };
        Hit cache[64];
        for (auto &c : cache) c = Hit{0, -1};
        int64_t wloc = 0;                                        // (not wmax_t[t] in the loop: neighbouring chunks share its cache line)
        for (int64_t i = t * per; i < std::min(nr, (t + 1) * per); ++i) {
          if (found[(size_t)i] >= 0) continue;
          const int64_t k0 = read_index(ptr, i, bytes) - base, k1 = read_index(ptr, i + 1, bytes) - base;
          const int64_t len = k1 - k0;
          const int64_t ref = row_ref(i, k0, len);
          uint64_t h;
          if (round == 0) {
            if (k1 < k0 || len > 1024) { bad.store(1); return; }
            h = 1469598103934665603ull ^ (uint64_t)len;
            int64_t prev = -1;
            for (int64_t j = 0; j < len; ++j) {
              const int64_t c = read_index(idx, k0 + j, bytes) - base;
              if (c < 0 || c >= P.ncols) { bad.store(2); return; }
              if (c <= prev) { bad.store(3); return; }           // sorted, duplicate-free rows only (what assemblers emit)
              prev = c;
              uint64_t bits;
              std::memcpy(&bits, &val[k0 + j], 8);
              h = (h ^ (uint64_t)(uint32_t)(int32_t)(c - ref)) * 1099511628211ull;
              h = (h ^ bits) * 1099511628211ull;
              h ^= h >> 29;
            }
            hashes[(size_t)i] = h;
            wloc = std::max(wloc, len);
          } else
            h = hashes[(size_t)i];
          Hit &c = cache[h & 63];
          if (c.q >= 0 && c.h == h && row_is(c.q, k0, len, ref)) { found[(size_t)i] = c.q; continue; }
          auto it = P.index.find(h);
          if (it != P.index.end()) {
            bool ok = false;
            for (int32_t q : it->second)
              if (row_is(q, k0, len, ref)) { found[(size_t)i] = q; c = Hit{h, q}; ok = true; break; }
            if (ok) continue;
          }
          if (bad.load(std::memory_order_relaxed)) return;
          if ((int)first.size() > kMaxPatterns) { int z = 0; bad.compare_exchange_strong(z, 4); return; }   // not a pattern operator: stop early
          auto ins = first.emplace(h, i);                        // keeps the smallest row of the chunk (rows ascend)
          if (!ins.second) {
            // same hash as an earlier unresolved row r of this chunk: when the rows are equal, i takes r's id once r has one
            // (one pass over a matrix that arrives whole: the dictionary is empty in round 0)
            const int64_t r = ins.first->second;
            const int64_t q0 = read_index(ptr, r, bytes) - base, qlen = read_index(ptr, r + 1, bytes) - base - q0;
            bool eq = qlen == len;
            if (eq) {
              const int64_t rref = row_ref(r, q0, qlen);
              for (int64_t j = 0; j < len && eq; ++j)
                eq = (read_index(idx, k0 + j, bytes) - ref) == (read_index(idx, q0 + j, bytes) - rref) &&
                     std::memcmp(&val[k0 + j], &val[q0 + j], 8) == 0;
            }
            if (eq) found[(size_t)i] = (int32_t)(-2 - (r - t * per));   // offset of r inside the chunk (< 2^31)
          }
        }
        for (const auto &kv : first) fresh[(size_t)t].emplace_back(kv.second, kv.first);   // (row, hash)
        if (round == 0) wmax_t[(size_t)t] = wloc;
      });
      REQUIRE(bad.load() != 1, GMG_ERR_INVALID, "row pointers not monotone, or a row longer than 1024 entries");
      REQUIRE(bad.load() != 2, GMG_ERR_INVALID, "column index out of range");
      REQUIRE(bad.load() != 3, GMG_ERR_UNSUPPORTED, "streamed rows must have sorted, duplicate-free columns");
      REQUIRE(bad.load() != 4, GMG_ERR_UNSUPPORTED,
              "streamed operator has more than 4096 distinct rows: it is not pattern-compressible; pass it whole (gmg_set_matrix)");
      std::vector<std::pair<int64_t, uint64_t>> todo;
      for (auto &f : fresh) todo.insert(todo.end(), f.begin(), f.end());
      if (todo.empty()) break;
      REQUIRE(round < 64, GMG_ERR_STATE, "pattern dictionary did not converge");
      std::sort(todo.begin(), todo.end());
      for (auto &pr : todo) {
        const int64_t i = pr.first;
        const int64_t k0 = read_index(ptr, i, bytes) - base, len = read_index(ptr, i + 1, bytes) - base - k0;
        const int64_t ref = row_ref(i, k0, len);
        auto &bucket = P.index[pr.second];
        int32_t q = -1;
        for (int32_t cand : bucket)                            // inserted a moment ago for an earlier chunk's first row?
          if (row_is(cand, k0, len, ref)) { q = cand; break; }
        if (q < 0) {
          REQUIRE((int)P.len.size() < kMaxPatterns, GMG_ERR_UNSUPPORTED,
                  "streamed operator has more than 4096 distinct rows: it is not pattern-compressible; pass it whole (gmg_set_matrix)");
          q = (int32_t)P.len.size();
          P.len.push_back((int32_t)len);
          P.start.push_back((int32_t)P.off.size());
          for (int64_t j = 0; j < len; ++j) {
            uint64_t bits;
            std::memcpy(&bits, &val[k0 + j], 8);
            P.off.push_back((int32_t)(read_index(idx, k0 + j, bytes) - base - ref));
            P.val.push_back(bits);
          }
          bucket.push_back(q);
        }
        found[(size_t)i] = q;
      }
      parallel_chunks(T, [&](int64_t t) {                        // rows that were equal to a chunk-first row take its id
        for (int64_t i = t * per; i < std::min(nr, (t + 1) * per); ++i)
          if (found[(size_t)i] <= -2) found[(size_t)i] = found[(size_t)(t * per + (int64_t)(-2 - found[(size_t)i]))];
      });
    }
    for (int64_t w : wmax_t) P.wmax = std::max(P.wmax, w);
    P.rowpid.resize((size_t)(row0 + nr));
    if (P.mode == 1) P.rowbase.resize((size_t)(row0 + nr));
    parallel_for(nr, [&](int64_t i) {
      P.rowpid[(size_t)(row0 + i)] = (uint16_t)found[(size_t)i];
      if (P.mode == 1) {
        const int64_t k0 = read_index(ptr, i, bytes) - base, len = read_index(ptr, i + 1, bytes) - base - k0;
        P.rowbase[(size_t)(row0 + i)] = (int32_t)(len > 0 ? read_index(idx, k0, bytes) - base : 0);
      }
    });
    P.nnz += read_index(ptr, nr, bytes) - base;
    P.rows_seen = row0 + nr;
  }
  // gmg_set_matrix / _prolongation / _restriction on a big CSR operator: look at the structure right away, while the caller's
  // arrays are still borrowed.  When the operator turns out to be a row-pattern operator whose table fits LDS it is kept
  // in that form ONLY -- no host copy, no CSR upload at setup (the common case of the benchmark hierarchy: 27 distinct
  // rows).  Anything else (too many patterns, unsorted rows, table too big) returns false and takes the general path.
  bool try_eager_pattern(std::shared_ptr<PatStream> &out, int mode, int64_t nrows, int64_t ncols, int64_t nnz, const void *ptr,
                         const void *idx, const double *val, int layout, int base, int bytes)
  {
    if (layout != GMG_CSR || comm.nranks != 1 || nrows < opt_int("GMG_EAGER_MIN_ROWS", 20000) || !opt_int("GMG_EAGER", 1)) return false;
    if (!opt_int("GMG_PATTERN", 1) || !opt_int("GMG_SELL", 1) || !ptr || !idx || !val) return false;
    if (ncols >= (int64_t)(1 << 28) || nnz <= 0) return false;
    if (read_index(ptr, 0, bytes) != base || read_index(ptr, nrows, bytes) - base != nnz) return false;   // general path reports it
    auto S = std::make_shared<PatStream>();
    S->mode = mode; S->nrows = nrows; S->ncols = ncols;
    try {
      stream_append(*S, 0, nrows, ptr, idx, val, base, bytes);
    } catch (const GmgError &) {
      return false;
    }
    const int un = pat_un_of(opt_int("GMG_PAT_UN", 9));      // table stride granularity of the pattern kernel, as finish_stream will see it (read_tuning has not run yet)
    const int64_t W = (std::max<int64_t>(S->wmax, 1) + un - 1) / un * un;
    const bool generic = (int64_t)(S->len.size() + 1) * (12 * W + 4) <= 48 * 1024;
    if (!generic) return false;                              // wide / many patterns: let the general path pick the layout
    out = S;
    return true;
  }
  // CSR rows of an operator held in row-pattern form (setup only: R = P^T when the caller gave P alone)
  static HostCSR expand_stream(const PatStream &P)
  {
    REQUIRE(P.complete(), GMG_ERR_STATE, "row stream incomplete");
    HostCSR H;
    H.nrows = P.nrows; H.ncols = P.ncols;
    H.ptr.assign((size_t)P.nrows + 1, 0);
    for (int64_t i = 0; i < P.nrows; ++i) H.ptr[(size_t)i + 1] = H.ptr[(size_t)i] + P.len[P.rowpid[(size_t)i]];
    H.col.resize((size_t)H.ptr[(size_t)P.nrows]);
    H.val.resize((size_t)H.ptr[(size_t)P.nrows]);
    parallel_for(P.nrows, [&](int64_t i) {
      const int32_t q = P.rowpid[(size_t)i];
      const int64_t ref = P.mode == 0 ? i : (int64_t)P.rowbase[(size_t)i];
      for (int32_t j = 0; j < P.len[q]; ++j) {
        H.col[(size_t)(H.ptr[(size_t)i] + j)] = (int32_t)(ref + P.off[P.start[q] + j]);
        std::memcpy(&H.val[(size_t)(H.ptr[(size_t)i] + j)], &P.val[P.start[q] + j], 8);
      }
    });
    return H;
  }
  // split stream of an own | ghost level (own x own rows in the pattern form + ghost-column CSR of the boundary rows) back to
  // whole rows, n_own x (n_own + n_ghost): own columns first, then the ghost columns -- the order the own | ghost kernels sum in
  static HostCSR merge_split(const HostCSR &own, int64_t ncols, const std::vector<int32_t> &g_rows, const std::vector<int64_t> &g_ptr,
                             const std::vector<int32_t> &g_col, const std::vector<double> &g_val)
  {
    HostCSR H;
    H.nrows = own.nrows; H.ncols = ncols;
    H.ptr.assign((size_t)own.nrows + 1, 0);
    std::vector<int64_t> gcount((size_t)own.nrows, 0);
    for (size_t t = 0; t < g_rows.size(); ++t) gcount[(size_t)g_rows[t]] = g_ptr[t + 1] - g_ptr[t];
    for (int64_t i = 0; i < own.nrows; ++i) H.ptr[(size_t)i + 1] = H.ptr[(size_t)i] + (own.ptr[(size_t)i + 1] - own.ptr[(size_t)i]) + gcount[(size_t)i];
    H.col.resize((size_t)H.nnz()); H.val.resize((size_t)H.nnz());
    parallel_for(own.nrows, [&](int64_t i) {
      const int64_t k0 = own.ptr[(size_t)i], k1 = own.ptr[(size_t)i + 1];
      std::copy(own.col.begin() + k0, own.col.begin() + k1, H.col.begin() + H.ptr[(size_t)i]);
      std::copy(own.val.begin() + k0, own.val.begin() + k1, H.val.begin() + H.ptr[(size_t)i]);
    });
    for (size_t t = 0; t < g_rows.size(); ++t) {
      const int64_t i = g_rows[t], at = H.ptr[(size_t)i] + (own.ptr[(size_t)i + 1] - own.ptr[(size_t)i]);
      std::copy(g_col.begin() + g_ptr[t], g_col.begin() + g_ptr[t + 1], H.col.begin() + at);
      std::copy(g_val.begin() + g_ptr[t], g_val.begin() + g_ptr[t + 1], H.val.begin() + at);
    }
    return H;
  }
  // a level that cannot keep its stream gets its rows back (whole rows when the stream was split)
  static void rows_from_stream(Level &L)
  {
    HostCSR own = expand_stream(*L.sA);
    if (L.sA_split) { L.hA = merge_split(own, L.hA.ncols, L.g_rows, L.g_ptr, L.g_col, L.g_val); L.clear_split(); }
    else L.hA = std::move(own);
    L.sA.reset();
  }
  // device form of a completed stream (the counterpart of upload_csr)
  DevCSR finish_stream(PatStream &P, const char *what)
  {
    REQUIRE(P.complete(), GMG_ERR_STATE, std::string(what) + ": row stream incomplete (" + std::to_string(P.rows_seen) + " of " + std::to_string(P.nrows) + " rows)");
    REQUIRE(use_pattern && use_sell, GMG_ERR_UNSUPPORTED, "streamed operators need the row-pattern layout (GMG_PATTERN=1, GMG_SELL=1)");
    REQUIRE(P.ncols < (int64_t)(1 << 28) && P.nrows >= 1 && P.nnz > 0, GMG_ERR_UNSUPPORTED, "streamed operator: unsupported shape");
    DevCSR D;
    D.nrows = P.nrows; D.ncols = P.ncols; D.nnz = P.nnz;
    const int np = (int)P.len.size();
    std::vector<int32_t> plen(P.len);
    plen.push_back(0);                                      // trailing empty pattern for the lanes past nrows
    const int un = pat_un_eff();
    const int W = (int)((std::max<int64_t>(P.wmax, 1) + un - 1) / un * un);
    const bool generic = (int64_t)(np + 1) * (12 * W + 4) <= 48 * 1024;
    std::vector<int32_t> poff((size_t)(np + 1) * W, 0);
    std::vector<double> pval((size_t)(np + 1) * W, 0.0);
    for (int p = 0; p < np; ++p)
      for (int32_t j = 0; j < P.len[p]; ++j) {
        poff[(size_t)p * W + j] = 8 * P.off[P.start[p] + j];
        std::memcpy(&pval[(size_t)p * W + j], &P.val[P.start[p] + j], 8);
      }
    if (P.mode == 0 && pat_shared && one_gather()) build_shared_offsets(D, plen, poff, pval, W, P.ncols, generic);
    // neither LDS form fits (many patterns x wide rows, e.g. Q2 transfer operators): the table is read from global memory / L2
    if (P.mode == 0 && P.nrows == P.ncols && !D.pdinv) {    // 1 ./ diag per pattern (JacobiLinearSolvers.jl:20-23)
      std::vector<double> pd((size_t)np + 1, 0.0);
      bool all = true;
      for (int p = 0; p < np && all; ++p) {
        bool f = false;
        for (int32_t j = 0; j < P.len[p]; ++j)
          if (P.off[P.start[p] + j] == 0) { double v; std::memcpy(&v, &P.val[P.start[p] + j], 8); pd[p] = 1.0 / v; f = true; }
        all = f;
      }
      if (all) {
        D.pdinv = upload(pd);
        D.pdinv_u = np >= 1 ? pd[0] : 0.0;
        D.pdinv_uniform = np >= 1;
        for (int p = 1; p < np; ++p) D.pdinv_uniform = D.pdinv_uniform && std::memcmp(&pd[p], &pd[0], 8) == 0;
      }
    }
    upload_pattern(D, P.nrows, P.rowpid, P.rowbase, plen, poff, pval, W, generic, true);
    return D;                                                // P keeps its per-row ids (2-6 B/row): gmg_setup may run again
  }
  // 16-bit coded table over the 5 x 5 run grid for the walk form of the wide-row kernels (DevCSR::aux_*): offsets in five equally
  // spaced groups of <= 15, at most 65 535 distinct values
  void build_aux_walk_table(DevCSR &D, const std::vector<int32_t> &plen, const std::vector<int32_t> &poff8, const std::vector<double> &pval, int W,
                            const std::vector<int32_t> &U)
  {
    const int np = (int)plen.size();
    std::vector<int32_t> first, lastv;
    for (int32_t o : U) {
      if (first.empty() || o > first.back() + 24) { first.push_back(o); lastv.push_back(o); }
      else lastv.back() = o;
    }
    if (first.size() != 5 || first[1] - first[0] < 64) return;
    for (size_t gq = 0; gq < 5; ++gq)
      if (lastv[gq] - first[gq] > 14 || (gq > 0 && first[gq] - first[gq - 1] != first[1] - first[0])) return;
    std::vector<int32_t> runs;
    for (size_t gq = 0; gq < 5; ++gq)
      for (int j = 0; j < 5; ++j) runs.push_back(first[gq] - 5 + 5 * j);
    std::vector<uint64_t> keys;
    for (int p = 0; p < np; ++p)
      for (int j = 0; j < plen[p]; ++j) { uint64_t bits; std::memcpy(&bits, &pval[(size_t)p * W + j], 8); keys.push_back(bits); }
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    if (keys.size() > 65535) return;
    const int nu = 125;
    std::vector<uint16_t> codes((size_t)np * nu, (uint16_t)65535);
    std::vector<uint32_t> rmask((size_t)np, 0u);
    for (int p = 0; p < np; ++p)
      for (int j = 0; j < plen[p]; ++j) {
        const int32_t o = poff8[(size_t)p * W + j] / 8;
        const size_t r = (size_t)(std::upper_bound(runs.begin(), runs.end(), o) - runs.begin()) - 1;
        uint64_t bits;
        std::memcpy(&bits, &pval[(size_t)p * W + j], 8);
        codes[(size_t)p * nu + r * 5 + (size_t)(o - runs[r])] = (uint16_t)(std::lower_bound(keys.begin(), keys.end(), bits) - keys.begin());
        rmask[p] |= 1u << r;
      }
    std::vector<double> dict(65536, 0.0);
    for (size_t q = 0; q < keys.size(); ++q) std::memcpy(&dict[q], &keys[q], 8);
    codes.resize(((codes.size() + 63) / 64) * 64, (uint16_t)65535);
    D.aux_codes16 = upload(codes); D.aux_dict = upload(dict); D.aux_rmask = upload(rmask); D.aux_run = upload(runs);
    D.h_aux_run = runs;
    D.aux_ok = true;
  }
  // Shared-offset form of a row-relative pattern table (sells_kernel): the union of all offsets, covered
  // greedily by runs of three consecutive offsets; every pattern becomes a dense coefficient vector over it.
  void build_shared_offsets(DevCSR &D, const std::vector<int32_t> &plen, const std::vector<int32_t> &poff8,
                            const std::vector<double> &pval, int W, int64_t ncols, bool generic_fits)
  {
    const int np = (int)plen.size();                       // includes the trailing empty pattern
    if (ncols >= (int64_t)(1 << 28)) return;
    std::vector<int32_t> U;
    for (int p = 0; p < np; ++p)
      for (int j = 0; j < plen[p]; ++j) {
        const int32_t o = poff8[(size_t)p * W + j] / 8;
        if (j > 0 && o <= poff8[(size_t)p * W + j - 1] / 8) return;   // unsorted or duplicate columns: keep the generic form
        U.push_back(o);
      }
    std::sort(U.begin(), U.end());
    U.erase(std::unique(U.begin(), U.end()), U.end());
    if (U.empty()) return;
    size_t maxlen = 0;
    for (int p = 0; p < np; ++p) maxlen = std::max(maxlen, (size_t)plen[p]);
    // distinct values (coded form)
    std::vector<uint64_t> keys;
    bool few_values = true;
    for (int p = 0; p < np && few_values; ++p)
      for (int j = 0; j < plen[p] && few_values; ++j) {
        uint64_t bits;
        std::memcpy(&bits, &pval[(size_t)p * W + j], 8);
        auto it = std::lower_bound(keys.begin(), keys.end(), bits);
        if (it == keys.end() || *it != bits) {
          if (keys.size() == 255) few_values = false;
          else keys.insert(it, bits);
        }
      }
    // candidates: plain table with runs of 3 (12 B per entry), coded table with runs of 5 or 3 (1 B per entry)
    struct Cand { int k; bool coded; int rb; };
    const Cand cands[] = {{3, false, pat_rb}, {5, true, 5}, {3, true, 3}};
    for (const Cand &c : cands) {
      if (c.coded && !few_values) continue;
      std::vector<int32_t> runs;
      for (size_t i = 0; i < U.size();) {
        const int32_t o = U[i];
        runs.push_back(o);
        while (i < U.size() && U[i] <= o + c.k - 1) ++i;
      }
      // Round 6: a coded table whose offsets fall into FIVE equally spaced groups of <= 15 consecutive offsets each -- a 2-D operator of
      // reach +-2 lines on interleaved vector dofs (Stokes velocity, Q2: 5 lines x offsets -5 .. +5) -- is laid out as the 5 x 5 grid of
      // runs the z-walk form of the wide-row kernels takes (sellw_zwalk_kernel: "planes" = grid lines): per group the runs
      // first - 5 + 5 j, j = 0 .. 4, so that the entries sit in the middle three (the kernel cuts a group to those when the outer two are
      // absent for the wave).  The greedy cover above gives 3 runs per group, which only the per-slice kernel can use (16 gathers per
      // slice: 251 us per application on 8.4e6 rows against ~1/4 of that in the walk).
      if (c.k == 5 && c.coded && runs.size() != 25 && opt_int("GMG_PAT_WIDE_GRID", 1)) {
        std::vector<int32_t> first, lastv;
        for (int32_t o : U) {
          if (first.empty() || o > first.back() + 24) { first.push_back(o); lastv.push_back(o); }
          else lastv.back() = o;
        }
        bool grid = first.size() == 5 && first[1] - first[0] >= 64;
        for (size_t gq = 0; gq < first.size() && grid; ++gq)
          grid = lastv[gq] - first[gq] <= 14 && (gq == 0 || first[gq] - first[gq - 1] == first[1] - first[0]);
        if (grid) {
          runs.clear();
          for (size_t gq = 0; gq < 5; ++gq)
            for (int j = 0; j < 5; ++j) runs.push_back(first[gq] - 5 + 5 * j);
        }
      }
      // more loads than the generic kernel would issue: not worth it (when the generic kernel is an option)
      if (generic_fits && 2 * runs.size() > maxlen + 2 && runs.size() != 25) continue;
      const size_t nreal = runs.size();                     // sorted ascending
      while (runs.size() % (size_t)c.rb) runs.push_back(0); // dummy runs: zero coefficients on x[row..row+k-1]
      const int nruns = (int)runs.size(), nu = c.k * nruns;
      const int64_t lds = c.coded ? (int64_t)2048 + (((int64_t)np * nu + 7) / 8) * 8 + (int64_t)np * 12 + 8
                                  : (int64_t)np * nu * 12 + 8 + (int64_t)np * 8;
      if (lds > 48 * 1024) continue;
      std::vector<PatEntry> tab;
      std::vector<uint8_t> codes;
      if (c.coded) codes.assign((size_t)np * nu, (uint8_t)255);
      else { tab.resize((size_t)np * nu); std::memset(tab.data(), 0, tab.size() * sizeof(PatEntry)); }
      for (int p = 0; p < np; ++p)
        for (int j = 0; j < plen[p]; ++j) {
          const int32_t o = poff8[(size_t)p * W + j] / 8;
          const auto it = std::upper_bound(runs.begin(), runs.begin() + nreal, o);   // the run that holds o
          const size_t r = (size_t)(it - runs.begin()) - 1;
          const size_t e = (size_t)p * nu + r * c.k + (size_t)(o - runs[r]);
          if (c.coded) {
            uint64_t bits;
            std::memcpy(&bits, &pval[(size_t)p * W + j], 8);
            codes[e] = (uint8_t)(std::lower_bound(keys.begin(), keys.end(), bits) - keys.begin());
          } else { tab[e].v = pval[(size_t)p * W + j]; tab[e].m = 0xffffffffu; }
        }
      {   // inverse diagonal per pattern (JacobiLinearSolvers.jl:20-23: 1 ./ diag(A)); rows of the same pattern share it
        std::vector<double> pd((size_t)np, 0.0);
        // (own | ghost levels: the own x own part is n_own x (n_own + n_ghost) with the diagonal at column = row)
        bool all = D.nrows <= D.ncols;
        for (int p = 0; p < np - 1 && all; ++p) {
          bool found = false;
          for (int j = 0; j < plen[p]; ++j)
            if (poff8[(size_t)p * W + j] == 0) { pd[p] = 1.0 / pval[(size_t)p * W + j]; found = true; }
          all = found;
        }
        if (all) {
          D.pdinv = upload(pd);
          D.pdinv_u = np >= 2 ? pd[0] : 0.0;
          D.pdinv_uniform = np >= 2;                         // (the last pattern is the empty one)
          for (int p = 1; p < np - 1; ++p) D.pdinv_uniform = D.pdinv_uniform && std::memcmp(&pd[p], &pd[0], 8) == 0;
        }
      }
      if (c.coded) {
        if (nruns > 32) continue;                           // run masks are 32-bit
        std::vector<uint32_t> rmask((size_t)np, 0u);
        for (int p = 0; p < np; ++p)
          for (int e = 0; e < nu; ++e)
            if (codes[(size_t)p * nu + e] != 255) rmask[p] |= 1u << (e / c.k);
        D.prunmask = upload(rmask);
        std::vector<double> dict(256, 0.0);
        for (size_t q = 0; q < keys.size(); ++q) std::memcpy(&dict[q], &keys[q], 8);
        D.pcodes = upload_padded(codes, 64);
        D.pdict = upload(dict);
      } else {
        D.ptab = upload(tab);
        std::vector<double> tab8(tab.size());
        for (size_t q = 0; q < tab.size(); ++q) tab8[q] = tab[q].v;
        D.ptab8 = upload(tab8);
      }
      if (!c.coded && opt_int("GMG_PAT_WIDE_GRID", 1) && pat_zwalk && pat_zwalk_wide && (pat_zwalk >= 2 || D.nrows >= pat_zwalk_wide_rows) && np <= 4096)
        build_aux_walk_table(D, plen, poff8, pval, W, U);
      D.prun = upload(runs);
      D.pat_nruns = nruns;
      D.pat_k = c.k;
      D.pat_coded = c.coded;
      D.pat_minoff = *std::min_element(runs.begin(), runs.end());
      D.pat_maxoff = *std::max_element(runs.begin(), runs.end());
      D.pat_shared = true;
      return;
    }
  }
  static constexpr int kWideStride = 64;                     // patterns per chunk the lists hold
  // which patterns every chunk of the launch geometry touches.  Returns the number of workgroups to launch (0: the coded kernel
  // stays): the requested count, or -- when the tables let fewer workgroups be resident than that -- a whole number of resident
  // rounds (3 workgroups per CU fit with the 51 KB of a Q2 stiffness matrix: 2048 workgroups would run 3 + 3 + 2 per CU, the last
  // round a third empty; 1536 run 3 + 3).
  int prepare_wide(const DevCSR &M, int nwg_req, int wpb, int nsl, int rows)
  {
    if (M.wl_state != 0 && M.wl_req == nwg_req && M.wl_wpb == wpb) return M.wl_state == 1 ? M.wl_nwg : 0;
    if (M.wl_state == 1) { HIP_CHECK(hipStreamSynchronize(stream)); release(M.wl_pids, (size_t)M.wl_nwg * kWideStride); release(M.wl_cnt, (size_t)M.wl_nwg); }
    M.wl_state = 2; M.wl_req = nwg_req; M.wl_nwg = 0; M.wl_wpb = wpb; M.wl_max = 0;
    if (!pat_wide || !M.pat_coded || M.pat_np > 4096 || M.pat_np < 2 || !M.rowpid || !M.pcodes || !M.pdict || !M.prunmask) return 0;
    const size_t lds_cap = (size_t)opt_int("GMG_PAT_WIDE_LDS", 72 * 1024);
    int nwg = nwg_req;
    for (int attempt = 0; attempt < 2; ++attempt) {
      uint16_t *pids = dalloc<uint16_t>((size_t)nwg * kWideStride);
      int32_t *cnt = dalloc<int32_t>((size_t)nwg);
      hipLaunchKernelGGL(sellw_chunk_patterns_kernel, dim3(nwg), dim3(256), 0, stream, M.rowpid, M.nrows, nsl, rows, M.pat_np, kWideStride, pids, cnt);
      HIP_CHECK(hipGetLastError());
      std::vector<int32_t> h((size_t)nwg);
      HIP_CHECK(hipMemcpyAsync(h.data(), cnt, sizeof(int32_t) * (size_t)nwg, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      const int lmax = *std::max_element(h.begin(), h.end());
      const bool ok = lmax <= kWideStride && lmax <= 255 && wide_lds(M, lmax) <= lds_cap;
      // resident workgroups per CU: LDS (160 KB) and registers (128 per lane: 4 waves per SIMD)
      const int per_cu = ok ? (int)std::min<size_t>((size_t)(160 * 1024) / (wide_lds(M, lmax) + 512), (size_t)std::max(1, 16 / wpb)) : 0;
      const int resident = per_cu * 256;
      if (opt_int("GMG_SETUP_TIMING", 0))
        std::fprintf(stderr, "[gmg] wide-row tables: %lld rows, %d patterns x %d entries, %d workgroups, at most %d patterns per chunk -> %zu B of LDS, %d workgroups per CU\n",
                     (long long)M.nrows, M.pat_np, M.pat_k * M.pat_nruns, nwg, lmax, wide_lds(M, std::min(lmax, 255)), per_cu);
      if (!ok) { release(pids, (size_t)nwg * kWideStride); release(cnt, (size_t)nwg); return 0; }
      const int rounds = resident > 0 ? nwg / resident : 0;
      if (attempt == 0 && opt_int("GMG_PAT_WIDE_ROUNDS", 1) && rounds >= 1 && nwg % resident != 0) {
        // not a whole number of rounds: take the whole rounds below the request and list the patterns of THAT geometry
        release(pids, (size_t)nwg * kWideStride); release(cnt, (size_t)nwg);
        nwg = rounds * resident;
        continue;
      }
      M.wl_pids = pids; M.wl_cnt = cnt; M.wl_max = lmax; M.wl_nwg = nwg; M.wl_state = 1;
      return nwg;
    }
    return 0;
  }
  static size_t wide_lds(const DevCSR &M, int lmax)
  {
    const size_t nu = (size_t)M.pat_k * M.pat_nruns;
    return (size_t)lmax * (nu + M.pat_k) * 8 + (size_t)lmax * 12 + (size_t)M.pat_np + 16;
  }
  // z-walk form of the wide-row kernels (kernels.hpp: sellw_zwalk_kernel): the 25 runs must be a 5 x 5 grid of offsets whose rows
  // differ by a constant plane offset; lists of the patterns every workgroup (four chains) touches.  Returns the workgroups to launch.
  int prepare_wide_z(const DevCSR &M)
  {
    if (M.wz_state != 0 && M.wz_T == pat_zwalk_T) return M.wz_state == 1 ? M.wz_nwg : 0;
    if (M.wz_state == 1) { HIP_CHECK(hipStreamSynchronize(stream)); release(M.wz_pids, (size_t)M.wz_nwg * kWideStride); release(M.wz_cnt, (size_t)M.wz_nwg); }
    M.wz_state = 2; M.wz_T = pat_zwalk_T; M.wz_nwg = 0; M.wz_max = 0;
    const bool main_tab = M.pat_coded && M.pat_k == 5 && M.pat_nruns == 25 && M.pcodes && M.pdict && M.prunmask;
    if (!pat_wide || !(main_tab || M.aux_ok) || M.pat_np > 4096 || M.pat_np < 2 || !M.rowpid) return 0;
    if (M.ncols >= (int64_t)(1 << 28) || M.nrows + 64 + std::max<int64_t>(M.pat_maxoff, -(int64_t)M.pat_minoff) + 16 >= (int64_t)(1 << 28)) return 0;
    const std::vector<int32_t> &off = main_tab ? host_run_off(M) : M.h_aux_run;
    const int64_t P = (int64_t)off[5] - off[0];
    if (P < 64 || P > (int64_t)(1 << 24)) return 0;
    for (int q = 0; q < 20; ++q) if ((int64_t)off[(size_t)q + 5] - off[(size_t)q] != P) return 0;
    ZWalkGeo g;
    g.P = (int)P; g.m = (int)((P + 59) / 60); g.T = pat_zwalk_T;
    g.nplanes = (int)((M.nrows + P - 1) / P);
    if (g.nplanes < 5) return 0;
    const int64_t nch = (int64_t)((g.nplanes + g.T - 1) / g.T) * g.m;
    if (nch >= (int64_t)(1 << 30)) return 0;
    g.nchains = (int)nch;
    const int wpb = 4, nwg = (g.nchains + wpb - 1) / wpb;
    uint16_t *pids = dalloc<uint16_t>((size_t)nwg * kWideStride);
    int32_t *cnt = dalloc<int32_t>((size_t)nwg);
    hipLaunchKernelGGL(sellwz_patterns_kernel, dim3(nwg), dim3(256), 0, stream, M.rowpid, M.nrows, g, wpb, M.pat_np, kWideStride, pids, cnt);
    HIP_CHECK(hipGetLastError());
    std::vector<int32_t> h((size_t)nwg);
    HIP_CHECK(hipMemcpyAsync(h.data(), cnt, sizeof(int32_t) * (size_t)nwg, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    const int lmax = *std::max_element(h.begin(), h.end());
    const size_t lds_cap = (size_t)opt_int("GMG_PAT_WIDE_LDS", 72 * 1024);
    if (opt_int("GMG_SETUP_TIMING", 0))
      std::fprintf(stderr, "[gmg] wide-row z-walk: %lld rows, P = %d, %d intervals x %d z-blocks, at most %d patterns per workgroup -> %zu B of LDS\n",
                   (long long)M.nrows, g.P, g.m, (g.nplanes + g.T - 1) / g.T, lmax, wide_z_lds(M, std::min(lmax, 255)));
    if (!(lmax <= kWideStride && lmax <= 255 && wide_z_lds(M, lmax) <= lds_cap)) { release(pids, (size_t)nwg * kWideStride); release(cnt, (size_t)nwg); return 0; }
    M.wz_pids = pids; M.wz_cnt = cnt; M.wz_max = lmax; M.wz_nwg = nwg; M.wz_geo = g; M.wz_state = 1;
    return nwg;
  }
  static size_t wide_z_lds(const DevCSR &M, int lmax) { return (size_t)lmax * 126 * 8 + (size_t)lmax * 4 + (size_t)M.pat_np + 16; }
  // sells_r2mv_kernel applies: plain shared-offset table, 27- / 9-point runs, signed 32-bit offsets, table + masks within 64 KB of LDS
  bool r2mv_ok(const DevCSR &M) const
  {
    const int nu = M.pat_k * M.pat_nruns;
    return pat_r2mv && M.sell && M.pat && M.pat_shared && !M.pat_coded && M.pat_k == 3 && (M.pat_nruns == 9 || M.pat_nruns == 3) && M.nrows >= pat_r2mv_min &&
           M.ncols < (int64_t)(1 << 28) && M.nrows + 192 + std::max<int64_t>(M.pat_maxoff, -(int64_t)M.pat_minoff) < (int64_t)(1 << 28) &&
           (size_t)M.pat_np * nu * 16 + 16 <= 64 * 1024;
  }
  double *r2mv_dot_parts = nullptr;    // set by spmv_set_dot around its launch
  int r2mv_dot_n = 0;
  // y = M x and the first stage of dot(x, y) in one kernel; returns the number of partials written to `parts`, 0 = nothing done
  // (the caller applies the operator and runs the separate dot), -1 = y is done but the dot is not.  Single rank, no halo.
  int spmv_set_dot(const DevCSR &M, const double *x, double *y, double *parts)
  {
    if (!pat_r2mv_dot || comm.nranks > 1 || !r2mv_ok(M)) return 0;
    r2mv_dot_parts = parts; r2mv_dot_n = 0;
    try { spmv_set(M, x, y); } catch (...) { r2mv_dot_parts = nullptr; throw; }
    r2mv_dot_parts = nullptr;
    return r2mv_dot_n > 0 ? r2mv_dot_n : -1;                 // -1: y = M x is done, the dot is not (a grid of more than kRedBlocks workgroups)
  }
  template <int EPI, bool ONEG>
  void launch_sells(const DevCSR &M, const StreamArgs2 &a2)
  {
    SellSArgs a;
    std::memset(&a, 0, sizeof(a));
    a.rowpid = M.rowpid; a.tab = M.ptab; a.tab8 = M.ptab8; a.run_off = M.prun; a.np = M.pat_np; a.nruns = M.pat_nruns;
    a.minoff = M.pat_minoff; a.maxoff = M.pat_maxoff;
    a.xmode = a2.xmode; a.pdinv = (EPI == EPI_SWEEP && a2.dinv_from_table) ? M.pdinv : nullptr;
    a.codes = M.pcodes; a.dict = M.pdict; a.runmask = M.prunmask;
    const int rows = 65 - M.pat_k;
    const int nsl = (int)((M.nrows + rows - 1) / rows);
    a.nrows = M.nrows; a.ncols = M.ncols; a.nslices = nsl; a.xcd_remap = xcd_remap;
    a.x_zero = a2.x_zero; a.x = a2.x; a.dinv = a2.dinv; a.omega = a2.omega; a.y = a2.y; a.b = a2.b; a.x2 = a2.x2; a.s_out = a2.s_out;
    const int wpb = sell_block > 0 ? sell_block / 64 : (nsl >= 256 * 32 ? 4 : pat_small_wpb);
    const int nwg = std::max(1, std::min((nsl + wpb - 1) / wpb, pat_wgs));
    const int nu = M.pat_k * M.pat_nruns;
    const size_t lds = M.pat_coded ? (size_t)2048 + (((size_t)M.pat_np * nu + 7) / 8) * 8 + (size_t)M.pat_np * 12 + 8
                                   : (size_t)M.pat_np * nu * 12 + 8 + (size_t)M.pat_np * 8;
    const dim3 g(nwg), b(64 * wpb);
    if constexpr ((EPI == EPI_SET || EPI == EPI_SUB || EPI == EPI_RESID) && !ONEG) {
      // two rows per lane (sells_r2mv_kernel): the mat-vecs of CG and of the coarse-grid correction on the big row-pattern levels
      if (r2mv_ok(M) && !a2.s_out) {
        const int nsl2 = (int)((M.nrows + 125) / 126);
        a.nslices = nsl2;
        // (the 64-register form of the sweeps, pat_r2_occ, does not pay here: 14.8 -> 15.3 us)
        const dim3 gr(std::max(1, std::min((nsl2 + wpb - 1) / wpb, pat_r2_wgs > 0 ? pat_r2_wgs : 4 * n_cus)));
        const size_t lds2 = (size_t)M.pat_np * nu * 16 + 16;
        const bool mk = pat_strict || !M.ptab8;
        // big levels (>= pat_tile_rows rows), no fused dot asked for: one slice per wave in workgroups of eight waves, 64-register form
        // (as the sweeps of these levels: launch_rsweep)
        bool dot_here = false;
        if constexpr (EPI == EPI_SET) dot_here = r2mv_dot_parts != nullptr;
        // the same levels in the z-walk form (kernels.hpp: sells_zsweep_kernel<..., EPI>): three new windows per step, requests a step ahead
        ZWalkGeo zg;
        if (pat_zwalk && pat_zwalk_mv && !dot_here && (pat_zwalk >= 2 || M.nrows >= pat_zwalk_rows) && zwalk_geo(M, zg)) {
          const dim3 gz((unsigned)((zg.nchains + 3) / 4)), bz(256);
          const size_t ldsz = (size_t)M.pat_np * 28 * 8 + 16;
          if (mk) { if (pat_fma) hipLaunchKernelGGL((sells_zsweep_kernel<1, true, true, EPI>), gz, bz, ldsz, stream, a, zg);
                    else hipLaunchKernelGGL((sells_zsweep_kernel<1, true, false, EPI>), gz, bz, ldsz, stream, a, zg); }
          else { if (pat_fma) hipLaunchKernelGGL((sells_zsweep_kernel<1, false, true, EPI>), gz, bz, ldsz, stream, a, zg);
                 else hipLaunchKernelGGL((sells_zsweep_kernel<1, false, false, EPI>), gz, bz, ldsz, stream, a, zg); }
          HIP_CHECK(hipGetLastError());
          return;
        }
        if (pat_r2_occ >= 2 && M.nrows >= pat_tile_rows && M.pat_nruns == 9 && pat_r2_wgs <= 0 && !dot_here) {
          const dim3 g8((nsl2 + 7) / 8), b8(512);
          if (mk) { if (pat_fma) hipLaunchKernelGGL((sells_r2mv_kernel<EPI, true, true, 9, false, 2>), g8, b8, lds2, stream, a);
                    else hipLaunchKernelGGL((sells_r2mv_kernel<EPI, true, false, 9, false, 2>), g8, b8, lds2, stream, a); }
          else { if (pat_fma) hipLaunchKernelGGL((sells_r2mv_kernel<EPI, false, true, 9, false, 2>), g8, b8, lds2, stream, a);
                 else hipLaunchKernelGGL((sells_r2mv_kernel<EPI, false, false, 9, false, 2>), g8, b8, lds2, stream, a); }
          HIP_CHECK(hipGetLastError());
          return;
        }
#define GMG_R2MV_LAUNCH(MKV, FMV)                                                                              \
        do {                                                                                                     \
          if (M.pat_nruns == 9) hipLaunchKernelGGL((sells_r2mv_kernel<EPI, MKV, FMV, 9>), gr, b, lds2, stream, a);   \
          else hipLaunchKernelGGL((sells_r2mv_kernel<EPI, MKV, FMV, 3>), gr, b, lds2, stream, a);                \
        } while (0)
        if constexpr (EPI == EPI_SET) {
          if (r2mv_dot_parts && gr.x <= (unsigned)kRedBlocks) {   // the first stage of dot(x, A x) rides along (spmv_set_dot)
            a.s_out = r2mv_dot_parts;
            r2mv_dot_n = (int)gr.x;
#define GMG_R2MV_DOT(MKV, FMV)                                                                                 \
            do {                                                                                                 \
              if (M.pat_nruns == 9) hipLaunchKernelGGL((sells_r2mv_kernel<EPI_SET, MKV, FMV, 9, true>), gr, b, lds2, stream, a);   \
              else hipLaunchKernelGGL((sells_r2mv_kernel<EPI_SET, MKV, FMV, 3, true>), gr, b, lds2, stream, a);  \
            } while (0)
            if (mk) { if (pat_fma) GMG_R2MV_DOT(true, true); else GMG_R2MV_DOT(true, false); }
            else { if (pat_fma) GMG_R2MV_DOT(false, true); else GMG_R2MV_DOT(false, false); }
#undef GMG_R2MV_DOT
            HIP_CHECK(hipGetLastError());
            return;
          }
        }
        if (mk) { if (pat_fma) GMG_R2MV_LAUNCH(true, true); else GMG_R2MV_LAUNCH(true, false); }
        else { if (pat_fma) GMG_R2MV_LAUNCH(false, true); else GMG_R2MV_LAUNCH(false, false); }
#undef GMG_R2MV_LAUNCH
        HIP_CHECK(hipGetLastError());
        return;
      }
    }
    if (EPI == EPI_SWEEP && ONEG && !M.pat_coded && M.pat_k == 3 && pat_rb == 3 && pat_batched && M.pat_nruns % 3 == 0 &&
        (size_t)M.pat_np * nu * 16 + (size_t)M.pat_np * 8 + 16 <= 64 * 1024) {
      // restructured sweep (sells_sweep_kernel): no conditional operand loads, one store drain per NB slices
      const bool td = a.pdinv != nullptr;
      const int nb = pat_nb > 0 ? pat_nb : (nsl >= 200000 ? 2 : 1);
      const int wg2 = std::max(1, std::min((nsl + wpb - 1) / wpb, nb >= 2 && pat_wgs == 2048 ? 1024 : pat_wgs));
      const dim3 g2(wg2);
      const size_t lds2 = (size_t)M.pat_np * nu * 16 + (size_t)M.pat_np * 8 + 16;    // 16-byte {value, mask} entries
      const bool mk = pat_strict || !M.ptab8;
#define GMG_SWEEP_LAUNCH(XMV, NBV)                                                                             \
      do {                                                                                                       \
        if (mk) {                                                                                                \
          if (td) hipLaunchKernelGGL((sells_sweep_kernel<XMV, NBV, true, true>), g2, b, lds2, stream, a);        \
          else hipLaunchKernelGGL((sells_sweep_kernel<XMV, NBV, false, true>), g2, b, lds2, stream, a);          \
        } else {                                                                                                 \
          if (td) hipLaunchKernelGGL((sells_sweep_kernel<XMV, NBV, true, false>), g2, b, lds2, stream, a);       \
          else hipLaunchKernelGGL((sells_sweep_kernel<XMV, NBV, false, false>), g2, b, lds2, stream, a);         \
        }                                                                                                        \
      } while (0)
      M.note_sweep("sells_sweep_kernel<XM=*,NB=%d,TD=%d,MK=%d> wgs=%d wpb=%d", nb >= 2 ? 2 : 1, td ? 1 : 0, mk ? 1 : 0, wg2, wpb);
      if (nb >= 2) { if (a.xmode == 0) GMG_SWEEP_LAUNCH(0, 2); else if (a.xmode == 1) GMG_SWEEP_LAUNCH(1, 2); else GMG_SWEEP_LAUNCH(2, 2); }
      else { if (a.xmode == 0) GMG_SWEEP_LAUNCH(0, 1); else if (a.xmode == 1) GMG_SWEEP_LAUNCH(1, 1); else GMG_SWEEP_LAUNCH(2, 1); }
#undef GMG_SWEEP_LAUNCH
      HIP_CHECK(hipGetLastError());
      return;
    }
    if (EPI == EPI_SWEEP) M.note_sweep("sells_kernel<EPI_SWEEP,%s,RB=%d,K=%d,VD=%d> wgs=%d wpb=%d", ONEG ? "ONEG" : "2G", M.pat_coded ? M.pat_k : pat_rb, M.pat_k, M.pat_coded ? 1 : 0, nwg, wpb);
    if constexpr ((EPI == EPI_SET || EPI == EPI_SUB || EPI == EPI_RESID || EPI == EPI_ADDTO) && !ONEG) {
      // big wide-row levels: the z-walk form (five new windows per step instead of up to 25 gathers per slice)
      if (((M.pat_coded && M.pat_k == 5) || M.aux_ok) && pat_zwalk && pat_zwalk_wide && !a.s_out && (pat_zwalk >= 2 || M.nrows >= pat_zwalk_wide_rows)) {
        const int nwz = prepare_wide_z(M);
        if (nwz > 0) {
          if (!(M.pat_coded && M.pat_k == 5)) {               // the auxiliary 16-bit coded table over the 5 x 5 run grid
            a.codes = nullptr; a.codes16 = M.aux_codes16; a.dict = M.aux_dict; a.runmask = M.aux_rmask; a.run_off = M.aux_run; a.nruns = 25;
          }
          a.wl_pids = M.wz_pids; a.wl_cnt = M.wz_cnt; a.wl_stride = kWideStride; a.wl_max = M.wz_max;
          const size_t ldsz = wide_z_lds(M, M.wz_max);
          static bool attr_z[64] = {false};
          if (!attr_z[device & 63]) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sellw_zwalk_kernel<EPI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sellw_zwalk_kernel<EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            attr_z[device & 63] = true;
          }
          if (pat_fma) hipLaunchKernelGGL((sellw_zwalk_kernel<EPI, true>), dim3((unsigned)nwz), dim3(256), ldsz, stream, a, M.wz_geo);
          else
          hipLaunchKernelGGL((sellw_zwalk_kernel<EPI, false>), dim3((unsigned)nwz), dim3(256), ldsz, stream, a, M.wz_geo);
          HIP_CHECK(hipGetLastError());
          return;
        }
      }
    }
    const int nwg_wide = (M.pat_coded && M.pat_k == 5) ? prepare_wide(M, nwg, wpb, nsl, rows) : 0;
    if (nwg_wide > 0) {
      const dim3 g(nwg_wide);
      a.wl_pids = M.wl_pids; a.wl_cnt = M.wl_cnt; a.wl_stride = kWideStride; a.wl_max = M.wl_max;
      const size_t ldsw = wide_lds(M, M.wl_max);
      static bool attr_set[64] = {false};                   // per instantiation and device: LDS beyond the 64 KB default
      if (!attr_set[device & 63]) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sells_kernel<EPI, ONEG, 5, 5, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_set[device & 63] = true;
      }
      if (opt_int("GMG_DBG_NOGATHER", 0)) hipLaunchKernelGGL((sells_kernel<EPI, ONEG, 5, 5, true, 8, true>), g, b, ldsw, stream, a);   // timing ablation only
      else
      hipLaunchKernelGGL((sells_kernel<EPI, ONEG, 5, 5, true, 0, true>), g, b, ldsw, stream, a);
    } else
    if (M.pat_coded && M.pat_k == 5) hipLaunchKernelGGL((sells_kernel<EPI, ONEG, 5, 5, true>), g, b, lds, stream, a);
    else if (M.pat_coded) hipLaunchKernelGGL((sells_kernel<EPI, ONEG, 3, 3, true>), g, b, lds, stream, a);
    else if (pat_rb == 9) hipLaunchKernelGGL((sells_kernel<EPI, ONEG, 9>), g, b, lds, stream, a);
    else if (pat_rb == 1) hipLaunchKernelGGL((sells_kernel<EPI, ONEG, 1>), g, b, lds, stream, a);
    else hipLaunchKernelGGL((sells_kernel<EPI, ONEG, 3>), g, b, lds, stream, a);
    HIP_CHECK(hipGetLastError());
  }
  template <int EPI, bool ONEG>
  void launch_sellp(const DevCSR &M, const StreamArgs2 &a2)
  {
    if (M.pat_shared && (EPI != EPI_SWEEP || ONEG)) { launch_sells<EPI, ONEG>(M, a2); return; }
    REQUIRE(M.pat_generic || M.plen, GMG_ERR_UNSUPPORTED, "this operator only has the shared-offset pattern form (one-gather sweeps)");
    SellPArgs a;
    std::memset(&a, 0, sizeof(a));
    a.rowpid = M.rowpid; a.rowbase = M.rowbase; a.plen = M.plen; a.poff = M.ppoff; a.pval = M.ppval; a.np = M.pat_np; a.W = M.pat_w;
    a.nrows = M.nrows; a.nslices = M.nslices; a.xcd_remap = xcd_remap;
    a.x_zero = a2.x_zero; a.x = a2.x; a.dinv = a2.dinv; a.omega = a2.omega; a.y = a2.y; a.b = a2.b; a.x2 = a2.x2; a.s_out = a2.s_out;
    // persistent launch: the pattern table is staged into LDS once per workgroup
    const int wpb = sell_block > 0 ? sell_block / 64 : (M.nslices >= 256 * 32 ? 4 : pat_small_wpb2);
    const int nwg = std::max(1, std::min((M.nslices + wpb - 1) / wpb, pat_wgs));
    const size_t lds = (size_t)M.pat_np * M.pat_w * 12 + (size_t)M.pat_np * 4;
    const dim3 g(nwg), b(64 * wpb);
    if constexpr (EPI == EPI_ADDTO && !ONEG) {
      // two rows per lane (sellp_pair_addto_kernel): the prolongation + correction of the big levels
      if (pat_pair_p && M.pat_generic && M.rowbase && M.nrows >= pat_r2mv_min && M.ncols < (int64_t)(1 << 28)) {
        a.nslices = (int)((M.nrows + 127) / 128);
        const dim3 gp(std::max(1, std::min((a.nslices + wpb - 1) / wpb, pat_wgs)));
        switch (pat_un_eff()) {
        case 27: hipLaunchKernelGGL((sellp_pair_addto_kernel<9>), gp, b, lds, stream, a); break;
        case 14: hipLaunchKernelGGL((sellp_pair_addto_kernel<7>), gp, b, lds, stream, a); break;
        case 9: hipLaunchKernelGGL((sellp_pair_addto_kernel<9>), gp, b, lds, stream, a); break;
        case 6: hipLaunchKernelGGL((sellp_pair_addto_kernel<6>), gp, b, lds, stream, a); break;
        default: hipLaunchKernelGGL((sellp_pair_addto_kernel<3>), gp, b, lds, stream, a); break;
        }
        HIP_CHECK(hipGetLastError());
        return;
      }
    }
    if (!M.pat_generic) {                                   // table kept in global memory / L2 (does not fit LDS)
      switch (pat_un_eff()) {
      case 27: hipLaunchKernelGGL((sellp_kernel<EPI, ONEG, 27, true>), g, b, 0, stream, a); break;
      case 14: hipLaunchKernelGGL((sellp_kernel<EPI, ONEG, 14, true>), g, b, 0, stream, a); break;
      case 9: hipLaunchKernelGGL((sellp_kernel<EPI, ONEG, 9, true>), g, b, 0, stream, a); break;
      case 6: hipLaunchKernelGGL((sellp_kernel<EPI, ONEG, 6, true>), g, b, 0, stream, a); break;
      default: hipLaunchKernelGGL((sellp_kernel<EPI, ONEG, 3, true>), g, b, 0, stream, a); break;
      }
      HIP_CHECK(hipGetLastError());
      return;
    }
#define GMG_SELLP_LAUNCH(UNV) hipLaunchKernelGGL((sellp_kernel<EPI, ONEG, UNV>), g, b, lds, stream, a)
    switch (pat_un_eff()) {
    case 27: GMG_SELLP_LAUNCH(27); break;
    case 14: GMG_SELLP_LAUNCH(14); break;
    case 9: GMG_SELLP_LAUNCH(9); break;
    case 6: GMG_SELLP_LAUNCH(6); break;
    default: GMG_SELLP_LAUNCH(3); break;
    }
#undef GMG_SELLP_LAUNCH
    HIP_CHECK(hipGetLastError());
  }

  // SELL-64 conversion (setup): slices of 64 rows, width = longest row of the slice.
  void build_sell(const HostCSR &H, DevCSR &D)
  {
    if (!use_sell || H.nrows == 0 || D.nnz == 0) return;
    if (build_pattern(H, D)) return;
    const int64_t ns = (H.nrows + 63) / 64;
    std::vector<int64_t> soff((size_t)ns + 1, 0);
    int64_t band = 0;                                        // largest |column - row| (columns ascend inside a row): how far a gather reaches
    for (int64_t sl = 0; sl < ns; ++sl) {
      int64_t w = 0;
      for (int64_t i = sl * 64; i < std::min<int64_t>(H.nrows, sl * 64 + 64); ++i) {
        w = std::max(w, H.ptr[i + 1] - H.ptr[i]);
        if (H.ptr[i + 1] > H.ptr[i]) band = std::max(band, std::max<int64_t>(i - (int64_t)H.col[H.ptr[i]], (int64_t)H.col[H.ptr[i + 1] - 1] - i));
      }
      soff[sl + 1] = soff[sl] + w * 64;
    }
    D.band = band;
    const int64_t zp = soff[ns];
    if ((double)zp > sell_maxpad * (double)D.nnz) {
      // Ragged matrix.  SELL still wins when the stream compresses (<= 256 distinct values): a padded
      // entry then costs ~3 B instead of the 12 B a real entry costs in CSR (e.g. Q2 stiffness matrices:
      // padding 1.65, 38 distinct values).  Otherwise keep the CSR-stream kernel.
      bool few_values = use_vdict != 0;
      if (few_values) {
        std::vector<uint64_t> keys;
        for (int64_t k = 0; k < (int64_t)H.val.size() && few_values; ++k) {
          uint64_t bits;
          std::memcpy(&bits, &H.val[k], 8);
          auto it = std::lower_bound(keys.begin(), keys.end(), bits);
          if (it == keys.end() || *it != bits) {
            if (keys.size() == 256) few_values = false;
            else keys.insert(it, bits);
          }
        }
      }
      if (!few_values || 4.0 * (double)zp > 0.8 * 12.0 * (double)D.nnz) return;
    }
    // the SELL arrays are laid out ON THE DEVICE from the CSR copy that is already there (a host-side build wrote and
    // uploaded another 12 B per padded entry: 0.2 s for a 5 x 10^7-nonzero operator)
    D.soff = upload(soff);
    D.scol = dalloc<int32_t>((size_t)zp); D.sval = dalloc<double>((size_t)zp); D.rowlen = dalloc<int32_t>((size_t)H.nrows);
    {
      const int wpb = 4;
      const dim3 g((unsigned)((ns + wpb - 1) / wpb)), b(64 * wpb);
      if (D.ptr64) hipLaunchKernelGGL((sell_build_kernel<int64_t>), g, b, 0, stream, H.nrows, ns, (const int64_t *)D.rowptr, D.col, D.val, D.soff, D.scol, D.sval, D.rowlen);
      else hipLaunchKernelGGL((sell_build_kernel<int32_t>), g, b, 0, stream, H.nrows, ns, (const int32_t *)D.rowptr, D.col, D.val, D.soff, D.scol, D.sval, D.rowlen);
      HIP_CHECK(hipGetLastError());
    }
    D.nslices = (int)ns; D.zpad = zp; D.sell = true;
    auto rowlen_of = [&](int64_t i) { return (int32_t)(H.ptr[i + 1] - H.ptr[i]); };
    // (slice entry (j, lane l) of slice sl: column H.col[H.ptr[i] + j] for j < rowlen(i), else the row's first column)
    auto scol_at = [&](int64_t sl, int64_t j, int l) {
      const int64_t i = sl * 64 + l;
      if (i >= H.nrows) return (int32_t)0;
      const int64_t len = H.ptr[i + 1] - H.ptr[i];
      return len > 0 ? H.col[H.ptr[i] + (j < len ? j : 0)] : (int32_t)0;
    };
    // value dictionary first: without it the 16-bit offsets are not used either (unless forced), so their packing is skipped
    std::vector<uint64_t> keys;   // distinct bit patterns, sorted
    bool few_keys = use_vdict != 0;
    if (few_keys) {
      keys.reserve(257);
      for (int64_t k = 0; k < (int64_t)H.val.size() && few_keys; ++k) {
        uint64_t bits;
        std::memcpy(&bits, &H.val[k], 8);
        auto it = std::lower_bound(keys.begin(), keys.end(), bits);
        if (it == keys.end() || *it != bits) {
          if (keys.size() == 256) few_keys = false;
          else keys.insert(it, bits);
        }
      }
    }
    // ---- lossless compression of the stream (see sellc_kernel) ----
    std::vector<int64_t> poff((size_t)ns + 1, 0);
    for (int64_t sl = 0; sl < ns; ++sl) {
      const int64_t w = (soff[sl + 1] - soff[sl]) / 64;
      poff[sl + 1] = poff[sl] + ((w + 3) / 4) * 4 * 64;
    }
    const int64_t zpp = poff[ns];
    std::vector<uint8_t> smode((size_t)ns, 0);
    std::vector<uint16_t> pidx;
    std::vector<int32_t> pbase;
    int64_t n16 = 0;
    if (use_idx16 && (few_keys || use_idx16 > 1)) {
      pidx.assign((size_t)zpp, 0);
      pbase.assign((size_t)(zpp / 64), 0);
      parallel_for(ns, [&](int64_t sl) {
        const int64_t w = (soff[sl + 1] - soff[sl]) / 64;
        const int64_t w4 = (poff[sl + 1] - poff[sl]) / 64;
        bool ok = w > 0;
        for (int64_t j = 0; j < w && ok; ++j) {
          int32_t lo = INT32_MAX, hi = INT32_MIN;
          for (int l = 0; l < 64; ++l) {
            const int64_t i = sl * 64 + l;
            if (i < H.nrows && j < rowlen_of(i)) { const int32_t c = scol_at(sl, j, l); lo = std::min(lo, c); hi = std::max(hi, c); }
          }
          if (lo == INT32_MAX) lo = hi = scol_at(sl, j, 0);        // all-padding column
          if ((int64_t)hi - lo > 65535) ok = false;
          pbase[poff[sl] / 64 + j] = lo;
        }
        if (!ok) return;
        smode[sl] = 1;
        for (int64_t j = w; j < w4; ++j) pbase[poff[sl] / 64 + j] = pbase[poff[sl] / 64 + w - 1];
        for (int64_t j = 0; j < w4; ++j)
          for (int l = 0; l < 64; ++l) {
            const int64_t i = sl * 64 + l;
            const bool real = j < w && i < H.nrows && j < rowlen_of(i);
            const int32_t bj = pbase[poff[sl] / 64 + j];
            const int32_t c = real ? scol_at(sl, j, l) : bj;                 // padding decodes to the (valid) base column
            pidx[poff[sl] + (j / 4) * 256 + l * 4 + (j % 4)] = (uint16_t)(c - bj);
          }
      });
      for (int64_t sl = 0; sl < ns; ++sl) n16 += smode[sl];
    }
    std::vector<uint8_t> pcode;
    std::vector<double> dict;
    if (use_vdict) {
      if (few_keys) {
        dict.assign(256, 0.0);
        for (size_t q = 0; q < keys.size(); ++q) std::memcpy(&dict[q], &keys[q], 8);
        pcode.assign((size_t)zpp, 0);
        parallel_for(ns, [&](int64_t sl) {
          const int64_t w = (soff[sl + 1] - soff[sl]) / 64;
          for (int64_t j = 0; j < w; ++j)
            for (int l = 0; l < 64; ++l) {
              const int64_t i = sl * 64 + l;
              if (i < H.nrows && j < rowlen_of(i)) {
                uint64_t bits;
                std::memcpy(&bits, &H.val[H.ptr[i] + j], 8);
                const size_t code = std::lower_bound(keys.begin(), keys.end(), bits) - keys.begin();
                pcode[poff[sl] + (j / 4) * 256 + l * 4 + (j % 4)] = (uint8_t)code;
              }
            }
        });
      }
    }
    // 16-bit offsets pay off only together with the value dictionary (measured: with 8-byte
    // values the packed decode costs more than the 2 B/nnz it saves, profiles/r01_tuning.md)
    D.vdict = !pcode.empty();
    D.comp_idx = n16 > 0 && (D.vdict || use_idx16 > 1);
    if (!D.vdict && !D.comp_idx && use_opattern && H.nrows >= 64 && H.ncols < (int64_t)(1 << 28)) {
      // values differ from row to row (no dictionary) but the COLUMN structure may still be a handful of offset patterns
      // (variable coefficients on a structured mesh): then the 4 B/nnz column stream is replaced by 2 B/row
      std::vector<uint16_t> orowpid;
      std::vector<int32_t> orowbase, oplen, opoff;
      std::vector<double> opval;
      int OW = 0;
      bool gfit = false;
      bool ok = (H.ncols >= H.nrows) && detect_patterns(H, 0, orowpid, orowbase, oplen, opoff, opval, OW, false, gfit, true);
      if (!ok) ok = detect_patterns(H, 1, orowpid, orowbase, oplen, opoff, opval, OW, false, gfit, true);
      if (ok && (int64_t)oplen.size() * OW * 4 <= 32 * 1024) {
        D.orowpid = upload_padded(orowpid, 64);
        D.orowbase = orowbase.empty() ? nullptr : upload_padded(orowbase, 64);
        D.opoff = upload(opoff);
        D.opat_np = (int)oplen.size(); D.opat_w = OW;
        D.opat = true;
        D.stream_bytes_per_nnz = 8.0 + (orowbase.empty() ? 2.0 : 6.0) * (double)H.nrows / (double)D.nnz;
      }
    }
    if (!D.comp_idx) { std::fill(smode.begin(), smode.end(), 0); n16 = 0; }
    if (D.comp_idx || D.vdict) {
      D.poff = upload(poff); D.smode = upload(smode); D.zpack = zpp;
      if (pidx.empty()) pidx.assign(64, 0);
      if (pbase.empty()) pbase.assign((size_t)(zpp / 64) + 1, 0);
      if (pcode.empty()) pcode.assign(64, 0);
      if (dict.empty()) dict.assign(256, 0.0);
      D.pidx = upload_padded(pidx, 512); D.pbase = upload_padded(pbase, 64); D.pcode = upload_padded(pcode, 512); D.dict = upload(dict);
      const double frac16 = (double)n16 / (double)ns;
      D.stream_bytes_per_nnz = (D.vdict ? 1.0 : 8.0) + (2.0 * frac16 + 4.0 * (1.0 - frac16));
    }
  }
  template <int EPI, bool ONEG>
  void launch_sellc(const DevCSR &M, const StreamArgs2 &a2)
  {
    SellCArgs a;
    std::memset(&a, 0, sizeof(a));
    a.soff = M.soff; a.poff = M.poff; a.scol = M.scol; a.sval = M.sval; a.pidx = M.pidx; a.pcode = M.pcode; a.pbase = M.pbase;
    a.smode = M.smode; a.dict = M.dict; a.rowlen = M.rowlen; a.nrows = M.nrows; a.nslices = M.nslices; a.xcd_remap = xcd_remap;
    a.x_zero = a2.x_zero; a.x = a2.x; a.dinv = a2.dinv; a.omega = a2.omega; a.y = a2.y; a.b = a2.b; a.x2 = a2.x2; a.s_out = a2.s_out;
    const int wpb = sell_block > 0 ? sell_block / 64 : (M.nslices >= 256 * 32 ? 4 : 1);
    const dim3 g((M.nslices + wpb - 1) / wpb), b(64 * wpb);
    const bool nt = nt_loads && (M.stream_bytes_per_nnz * (double)M.zpad > 192.0e6);
    if (M.vdict) {
      if (nt) hipLaunchKernelGGL((sellc_kernel<EPI, ONEG, true, true>), g, b, 0, stream, a);
      else hipLaunchKernelGGL((sellc_kernel<EPI, ONEG, true, false>), g, b, 0, stream, a);
    } else {
      if (nt) hipLaunchKernelGGL((sellc_kernel<EPI, ONEG, false, true>), g, b, 0, stream, a);
      else hipLaunchKernelGGL((sellc_kernel<EPI, ONEG, false, false>), g, b, 0, stream, a);
    }
    HIP_CHECK(hipGetLastError());
  }
  template <int EPI, bool ONEG>
  void launch_sello(const DevCSR &M, const StreamArgs2 &a2)
  {
    SellOArgs a;
    std::memset(&a, 0, sizeof(a));
    a.soff = M.soff; a.sval = M.sval; a.rowlen = M.rowlen; a.rowpid = M.orowpid; a.rowbase = M.orowbase; a.poff = M.opoff;
    a.np = M.opat_np; a.W = M.opat_w; a.nrows = M.nrows; a.nslices = M.nslices; a.xcd_remap = xcd_remap;
    a.x_zero = a2.x_zero; a.x = a2.x; a.dinv = a2.dinv; a.omega = a2.omega; a.y = a2.y; a.b = a2.b; a.x2 = a2.x2; a.s_out = a2.s_out;
    const int wpb = sell_block > 0 ? sell_block / 64 : (M.nslices >= 256 * 32 ? 4 : 1);
    const dim3 g((M.nslices + wpb - 1) / wpb), b(64 * wpb);
    const size_t lds = (size_t)M.opat_np * M.opat_w * 4;
    const bool nt = nt_loads && (8.0 * (double)M.zpad > 128.0e6);
    const bool huge = big_level(M);                        // see launch_sell
    if (huge) a.xcd_remap = remap_for_big(M, 64 * wpb, (int)g.x);
    if constexpr (EPI == EPI_SWEEP && ONEG) {
      if (sell_un < 27 && (a2.xmode != 0 || (huge && nt && nt_rowwise))) {
#define GMG_SELLO_SWEEP(NTV)                                                                                                        \
        do {                                                                                                                          \
          if (a2.xmode == 1) hipLaunchKernelGGL((sello_kernel<EPI_SWEEP, true, 9, NTV, 1>), g, b, lds, stream, a);                    \
          else if (a2.xmode == 2) hipLaunchKernelGGL((sello_kernel<EPI_SWEEP, true, 9, NTV, 2>), g, b, lds, stream, a);               \
          else hipLaunchKernelGGL((sello_kernel<EPI_SWEEP, true, 9, NTV, 0>), g, b, lds, stream, a);                                  \
        } while (0)
        M.note_sweep("sello_kernel<EPI_SWEEP,ONEG,UN=9,NT=%d,XM=*> wpb=%d remap=%d", (huge && nt && nt_rowwise) ? 2 : (nt ? 1 : 0), wpb, a.xcd_remap);
        if (huge && nt && nt_rowwise) GMG_SELLO_SWEEP(2);
        else if (nt) GMG_SELLO_SWEEP(1);
        else GMG_SELLO_SWEEP(0);
#undef GMG_SELLO_SWEEP
        HIP_CHECK(hipGetLastError());
        return;
      }
    }
    REQUIRE(a2.xmode == 0 || EPI != EPI_SWEEP, GMG_ERR_STATE, "deferred x update needs the default unroll of the SELL-O sweep");
#define GMG_SELLO_LAUNCH(UNV)                                                                         \
    do {                                                                                                \
      if (nt) hipLaunchKernelGGL((sello_kernel<EPI, ONEG, UNV, 1>), g, b, lds, stream, a);               \
      else hipLaunchKernelGGL((sello_kernel<EPI, ONEG, UNV, 0>), g, b, lds, stream, a);                  \
    } while (0)
    if (EPI == EPI_SWEEP) M.note_sweep("sello_kernel<EPI_SWEEP,%s,UN=%d,NT=%d,XM=0> wpb=%d remap=%d", ONEG ? "ONEG" : "2G", sell_un >= 27 ? 27 : sell_un >= 9 ? 9 : 3, nt ? 1 : 0, wpb, a.xcd_remap);
    if (sell_un >= 27) GMG_SELLO_LAUNCH(27);
    else if (sell_un >= 9) GMG_SELLO_LAUNCH(9);
    else GMG_SELLO_LAUNCH(3);
#undef GMG_SELLO_LAUNCH
    HIP_CHECK(hipGetLastError());
  }
  template <int EPI, bool ONEG>
  void launch_sell(const DevCSR &M, const StreamArgs2 &a2)
  {
    if (M.pat) { launch_sellp<EPI, ONEG>(M, a2); return; }
    if (M.comp_idx || M.vdict) { launch_sellc<EPI, ONEG>(M, a2); return; }
    if (M.opat) { launch_sello<EPI, ONEG>(M, a2); return; }
    SellArgs a;
    std::memset(&a, 0, sizeof(a));
    a.soff = M.soff; a.scol = M.scol; a.sval = M.sval; a.rowlen = M.rowlen; a.nrows = M.nrows; a.nslices = M.nslices; a.xcd_remap = xcd_remap;
    a.x_zero = a2.x_zero; a.x = a2.x; a.dinv = a2.dinv; a.omega = a2.omega; a.y = a2.y; a.b = a2.b; a.x2 = a2.x2; a.s_out = a2.s_out;
    // one wave per slice; small levels get single-wave workgroups so that they spread over all CUs
    const int wpb = sell_block > 0 ? sell_block / 64 : (M.nslices >= 256 * 32 ? 4 : 1);
    const dim3 g((M.nslices + wpb - 1) / wpb), b(64 * wpb);
    const bool nt = nt_loads && (12.0 * (double)M.zpad > 192.0e6);   // keep cache-resident levels cacheable
    // Levels whose gathered vector no longer fits the eight L2s (profiles/r03_tuning.md): workgroups in launch order (all XCDs
    // stream -- and write -- inside one window of the arrays instead of eight far-apart eighths) and the once-per-sweep row-wise
    // operands non-temporal.  Both lose 1-2 % on levels that do fit, so they switch on by size.
    const bool huge = big_level(M);
    if (huge) a.xcd_remap = remap_for_big(M, 64 * wpb, (int)g.x);
    if constexpr (EPI == EPI_SWEEP && ONEG) {
      if (sell_un >= 6 && sell_un < 9 && (a2.xmode != 0 || (huge && nt && nt_rowwise))) {
        // default unroll: the variants with the deferred x update (a2.xmode) and non-temporal row-wise operands
#define GMG_SELL_SWEEP(NTV)                                                                                                         \
        do {                                                                                                                          \
          if (a2.xmode == 1) hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, NTV, 1>), g, b, 0, stream, a);                       \
          else if (a2.xmode == 2) hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, NTV, 2>), g, b, 0, stream, a);                  \
          else hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, NTV, 0>), g, b, 0, stream, a);                                     \
        } while (0)
        M.note_sweep("sell_kernel<EPI_SWEEP,ONEG,UN=6,NT=%d,XM=*> wpb=%d remap=%d", (huge && nt && nt_rowwise) ? 2 : (nt ? 1 : 0), wpb, a.xcd_remap);
        if (huge && nt && nt_rowwise) GMG_SELL_SWEEP(2);
        else if (nt) GMG_SELL_SWEEP(1);
        else GMG_SELL_SWEEP(0);
#undef GMG_SELL_SWEEP
        HIP_CHECK(hipGetLastError());
        return;
      }
    }
    REQUIRE(a2.xmode == 0 || EPI != EPI_SWEEP, GMG_ERR_STATE, "deferred x update needs the default unroll of the SELL-64 sweep");
#define GMG_SELL_LAUNCH(UNV)                                                                          \
    do {                                                                                                \
      if (nt) hipLaunchKernelGGL((sell_kernel<EPI, ONEG, UNV, 1>), g, b, 0, stream, a);                  \
      else hipLaunchKernelGGL((sell_kernel<EPI, ONEG, UNV, 0>), g, b, 0, stream, a);                     \
    } while (0)
    if (EPI == EPI_SWEEP) M.note_sweep("sell_kernel<EPI_SWEEP,%s,UN=%d,NT=%d,XM=0> wpb=%d remap=%d", ONEG ? "ONEG" : "2G", sell_un, nt ? 1 : 0, wpb, a.xcd_remap);
    if (sell_un >= 27) GMG_SELL_LAUNCH(27);
    else if (sell_un >= 9) GMG_SELL_LAUNCH(9);
    else if (sell_un >= 6) GMG_SELL_LAUNCH(6);
    else if (sell_un >= 4) GMG_SELL_LAUNCH(4);
    else if (sell_un >= 3) GMG_SELL_LAUNCH(3);
    else GMG_SELL_LAUNCH(2);
#undef GMG_SELL_LAUNCH
    HIP_CHECK(hipGetLastError());
  }

  // ---- kernel launchers ------------------------------------------------------
  static int grid_for(int64_t n, int block = 256)
  {
    const int64_t g = (n + block - 1) / block;
    return (int)std::max<int64_t>(1, std::min<int64_t>(g, 256 * 8));
  }

  template <int EPI, bool ONEG, bool EMIT_S>
  void launch_stream1(const DevCSR &M, const StreamArgs2 &a)
  {
    if (M.nblocks == 0) return;
    const dim3 g(M.nblocks), b(kBlock);
    if (M.ptr64) hipLaunchKernelGGL((csr_stream1_kernel<EPI, int64_t, 2048, ONEG, EMIT_S>), g, b, 0, stream, a);
    else hipLaunchKernelGGL((csr_stream1_kernel<EPI, int32_t, 2048, ONEG, EMIT_S>), g, b, 0, stream, a);
    HIP_CHECK(hipGetLastError());
  }
  StreamArgs2 base_args1(const DevCSR &M) const
  {
    StreamArgs2 a;
    std::memset(&a, 0, sizeof(a));
    a.rowptr = M.rowptr; a.col = M.col; a.val = M.val; a.blk_row = M.blk_row;
    a.nblocks = M.nblocks; a.lanes_log2 = M.lanes_log2; a.xcd_remap = xcd_remap;
    return a;
  }
  // y = M x
  // emit_s/emit_dinv/emit_omega: pattern layouts can also write s = omega*(Dinv*y) for the smoothing pass that consumes y
  void spmv_set(const DevCSR &M, const double *x, double *y, double *emit_s = nullptr, const double *emit_dinv = nullptr, double emit_omega = 0.0)
  {
    StreamArgs2 a = base_args1(M); a.x = x; a.y = y;
    if (emit_s && M.pat) { a.s_out = emit_s; a.dinv = emit_dinv; a.omega = emit_omega; }
    if (M.sell) { launch_sell<EPI_SET, false>(M, a); return; }
    launch_stream1<EPI_SET, false, false>(M, a);
  }
  // y -= M x
  void spmv_sub(const DevCSR &M, const double *x, double *y, double *emit_s = nullptr, const double *emit_dinv = nullptr, double emit_omega = 0.0)
  {
    StreamArgs2 a = base_args1(M); a.x = x; a.y = y;
    if (emit_s && M.pat) { a.s_out = emit_s; a.dinv = emit_dinv; a.omega = emit_omega; }
    if (M.sell) { launch_sell<EPI_SUB, false>(M, a); return; }
    launch_stream1<EPI_SUB, false, false>(M, a);
  }
  // y = b - M x
  void spmv_resid(const DevCSR &M, const double *x, const double *b, double *y)
  {
    StreamArgs2 a = base_args1(M); a.x = x; a.y = y; a.b = b;
    if (M.sell) { launch_sell<EPI_RESID, false>(M, a); return; }
    launch_stream1<EPI_RESID, false, false>(M, a);
  }
  // y = M x ; x2 += y      (omega != 0: y = omega (M x) ; x2 += y)
  void spmv_addto(const DevCSR &M, const double *x, double *y, double *x2, double omega = 0.0)
  {
    StreamArgs2 a = base_args1(M); a.x = x; a.y = y; a.x2 = x2; a.omega = omega;
    if (M.sell) { launch_sell<EPI_ADDTO, false>(M, a); return; }
    launch_stream1<EPI_ADDTO, false, false>(M, a);
  }
  bool one_gather() const { return one_gather_sweep != 0; }
  // levels whose fused sweeps gather r itself (uniform 1/diag, plain shared-offset table): see sells_rsweep_kernel
  bool rsweep_level(const Level &L) const
  {
    const DevCSR &M = L.A;
    if (L.rs_forbid) return false;                           // several ranks: some rank's part does not qualify (decided jointly at setup)
    if (!(pat_rsweep && one_gather_sweep && pat_dinv && M.sell && M.pat && M.pat_shared && !M.pat_coded && M.pat_k == 3 && pat_rb == 3 && pat_batched)) return false;
    if (M.pat_nruns % 3 != 0 || !M.pdinv || !M.pdinv_uniform || !L.rbuf[0] || !L.rbuf[1]) return false;
    // signed 32-bit byte offsets in the gathers: 8 * (row + run offset) is formed BEFORE the clamp, and row reaches 63 past the last slice
    if (M.ncols >= (int64_t)(1 << 28) || M.nrows + 64 + std::max<int64_t>(M.pat_maxoff, -(int64_t)M.pat_minoff) >= (int64_t)(1 << 28)) return false;
    const int nu = M.pat_k * M.pat_nruns;
    return (size_t)M.pat_np * nu * 16 + (size_t)M.pat_np * 8 + 16 <= 64 * 1024;
  }
  // the tile form of that sweep (kernels.hpp: sells_tsweep_kernel): segments of r a tile of T slices needs, merged over the runs
  bool launch_tsweep(const DevCSR &M, const SellSArgs &a, int nsl)
  {
    constexpr int WPB = 16, ROWS = 62, TMAX = 3 * WPB;
    if (M.pat_nruns > 32 || nsl < opt_int("GMG_PAT_TILE_MIN", 512)) return false;
    std::vector<int32_t> &off = M.h_run_off;
    if (off.empty()) {
      off.resize((size_t)M.pat_nruns);
      HIP_CHECK(hipMemcpyAsync(off.data(), M.prun, sizeof(int32_t) * (size_t)M.pat_nruns, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    }
    const bool mk = pat_strict || !M.ptab8;
    const int nu = M.pat_k * M.pat_nruns;
    const size_t lds_cap = (size_t)opt_int("GMG_PAT_TILE_LDS", 78 * 1024);      // two workgroups of 16 waves per CU
    std::vector<int> order((size_t)M.pat_nruns);
    for (int r = 0; r < M.pat_nruns; ++r) order[(size_t)r] = r;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return off[(size_t)x] < off[(size_t)y]; });
    // segments of r a tile of T slices needs: the runs' stretches [off, off + 62 T + 2), merged where they overlap or touch
    auto geometry = [&](int T, SellTile &tl) -> size_t {
      std::memset(&tl, 0, sizeof(tl));
      tl.T = T;
      const int RT = ROWS * T;
      int nseg = 0;
      int64_t elems = 0;
      for (int k = 0; k < M.pat_nruns; ++k) {
        const int r = order[(size_t)k];
        const int64_t lo = off[(size_t)r], hi = (int64_t)off[(size_t)r] + RT + 2;
        if (nseg > 0 && lo <= (int64_t)tl.seg_off[nseg - 1] + tl.seg_len[nseg - 1])
          tl.seg_len[nseg - 1] = (int)std::max<int64_t>(tl.seg_len[nseg - 1], hi - tl.seg_off[nseg - 1]);
        else {
          if (nseg == 12) return 0;
          tl.seg_off[nseg] = (int)lo; tl.seg_len[nseg] = (int)(hi - lo);
          ++nseg;
        }
      }
      for (int sg = 0; sg < nseg; ++sg) { tl.seg_base[sg] = (int)elems; elems += tl.seg_len[sg]; }
      for (int r = 0; r < M.pat_nruns; ++r) {
        int sg = 0;
        while (!(off[(size_t)r] >= tl.seg_off[sg] && (int64_t)off[(size_t)r] + RT + 2 <= (int64_t)tl.seg_off[sg] + tl.seg_len[sg])) ++sg;
        tl.run_lds[r] = tl.seg_base[sg] + (off[(size_t)r] - tl.seg_off[sg]);
      }
      tl.nseg = nseg; tl.elems = (int)elems;
      return (size_t)M.pat_np * nu * (mk ? 16 : 8) + (size_t)elems * 8 + 16;
    };
    // the largest tile that fits the LDS budget, then the tile size that fills whole rounds of the resident workgroups
    SellTile tl;
    int tmax = opt_int("GMG_PAT_TILE_T", TMAX);
    tmax = std::max(WPB, std::min(tmax, TMAX));
    size_t lds = 0;
    for (; tmax >= WPB; --tmax) { lds = geometry(tmax, tl); if (lds != 0 && lds <= lds_cap) break; }
    if (tmax < WPB) return false;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (size_t)(160 * 1024) / (lds + 512)));
    const int resident = per_cu * 256;
    const int rounds = (nsl + resident * tmax - 1) / (resident * tmax);
    const int T = std::max(WPB, (nsl + rounds * resident - 1) / (rounds * resident));
    lds = geometry(T, tl);
    if (lds == 0 || lds > lds_cap) return false;
    const int ntiles = (nsl + T - 1) / T;
    const int nwg = std::max(8, (std::min(ntiles, resident) / 8) * 8);          // a multiple of 8: one share per XCD
    const dim3 g(nwg), b(64 * WPB);
    // coefficients broadcast inside DPP rows (27-point operators), in the sweeps that leave x alone: 190.7 -> 184.8 us at 288^3; the sweeps
    // that also update x have no registers to spare under the 64 the two resident workgroups allow (252.8 -> 282 us with it)
    const bool bc = pat_bcast && M.pat_nruns == 9 && M.pat_k == 3 && a.xmode == 1;
    M.note_sweep("sells_tsweep_kernel<XM=*,MK=%d,WPB=%d,FM=%d,BC=%d> T=%d tiles=%d wgs=%d segs=%d", mk ? 1 : 0, WPB, pat_fma ? 1 : 0, bc ? 1 : 0, T, ntiles, nwg, tl.nseg);
#define GMG_TSWEEP_LAUNCH3(XMV, MKV, FMV, BCV)                                                                      \
    do {                                                                                                              \
      static bool attr[64] = {false};                                                                                 \
      if (!attr[device & 63]) { HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sells_tsweep_kernel<XMV, MKV, WPB, FMV, BCV>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); attr[device & 63] = true; } \
      hipLaunchKernelGGL((sells_tsweep_kernel<XMV, MKV, WPB, FMV, BCV>), g, b, lds, stream, a, tl);                   \
    } while (0)
#define GMG_TSWEEP_LAUNCH2(XMV, MKV, FMV) do { if (bc) GMG_TSWEEP_LAUNCH3(XMV, MKV, FMV, true); else GMG_TSWEEP_LAUNCH3(XMV, MKV, FMV, false); } while (0)
#define GMG_TSWEEP_LAUNCH(XMV)                                                                                      \
    do {                                                                                                              \
      if (mk) { if (pat_fma) GMG_TSWEEP_LAUNCH2(XMV, true, true); else GMG_TSWEEP_LAUNCH2(XMV, true, false); }        \
      else { if (pat_fma) GMG_TSWEEP_LAUNCH2(XMV, false, true); else GMG_TSWEEP_LAUNCH2(XMV, false, false); }         \
    } while (0)
    if (a.xmode == 0) GMG_TSWEEP_LAUNCH(0); else if (a.xmode == 1) GMG_TSWEEP_LAUNCH(1); else GMG_TSWEEP_LAUNCH(2);
#undef GMG_TSWEEP_LAUNCH
#undef GMG_TSWEEP_LAUNCH3
#undef GMG_TSWEEP_LAUNCH2
#undef GMG_TSWEEP_LAUNCH2
    HIP_CHECK(hipGetLastError());
    return true;
  }
  const std::vector<int32_t> &host_run_off(const DevCSR &M)
  {
    std::vector<int32_t> &off = M.h_run_off;
    if (off.empty()) {
      off.resize((size_t)M.pat_nruns);
      HIP_CHECK(hipMemcpyAsync(off.data(), M.prun, sizeof(int32_t) * (size_t)M.pat_nruns, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    }
    return off;
  }
  // geometry of the z-walk (kernels.hpp: sells_zsweep_kernel): the nine runs must be a 3 x 3 grid of offsets whose rows differ by a
  // constant plane offset P -- window q of the slice at r0 + P is window q + 3 of the slice at r0
  bool zwalk_geo(const DevCSR &M, ZWalkGeo &g)
  {
    if (M.pat_nruns != 9 || M.pat_k != 3 || !M.ptab) return false;
    const std::vector<int32_t> &off = host_run_off(M);
    const int64_t P = (int64_t)off[3] - off[0];
    if (P < 64 || P > (int64_t)(1 << 24)) return false;
    for (int q = 0; q < 6; ++q) if ((int64_t)off[(size_t)q + 3] - off[(size_t)q] != P) return false;
    g.P = (int)P;
    g.m = (int)((P + 125) / 126);
    g.T = pat_zwalk_T;
    g.nplanes = (int)((M.nrows + P - 1) / P);
    if (g.nplanes < 3) return false;
    const int64_t nch = (int64_t)((g.nplanes + g.T - 1) / g.T) * g.m;
    if (nch >= (int64_t)(1 << 30)) return false;
    g.nchains = (int)nch;
    return true;
  }
  // geometry of the fused pair of sweeps (kernels.hpp: sells_z2sweep_kernel): the nine runs are the 3 x 3 neighbours (dz, dy) of a grid
  // whose rows are numbered z P + y L + x, every plane has whole lines and the level whole planes
  bool z2_geo(const DevCSR &M, Z2Geo &g)
  {
    if (M.pat_nruns != 9 || M.pat_k != 3 || !M.ptab) return false;
    const std::vector<int32_t> &off = host_run_off(M);
    const int64_t P = (int64_t)off[3] - off[0], L = (int64_t)off[1] - off[0];
    if (L < 8 || P < 2 * L || P % L != 0 || M.nrows % P != 0 || M.nrows != M.ncols) return false;
    for (int q = 0; q < 9; ++q) if ((int64_t)off[(size_t)q] != (int64_t)(q / 3 - 1) * P + (int64_t)(q % 3 - 1) * L - 1) return false;
    g.P = (int)P; g.L = (int)L; g.ny = (int)(P / L); g.nz = (int)(M.nrows / P);
    if (g.nz < 3 || g.ny < 3) return false;
    g.whole = L <= 127 ? 1 : 0;
    g.nxs = g.whole ? 1 : (int)((L + 123) / 124);
    g.xlen = g.whole ? (int)L : (int)((L + g.nxs - 1) / g.nxs);
    // W lines per workgroup (two of them rim lines), T planes per block: enough workgroups for two rounds of the chip, as little redundancy as that allows
    int W = pat_fuse2_w > 0 ? pat_fuse2_w : (M.nrows >= 8000000 ? 16 : 8);
    W = std::max(3, std::min(16, std::min(W, g.ny + 2)));
    g.W = W;
    g.nyt = (g.ny + W - 3) / (W - 2);
    int T = pat_fuse2_t;
    if (T <= 0) {
      const int64_t per_plane_block = (int64_t)g.nyt * g.nxs;
      const int64_t want = 2 * (int64_t)n_cus;
      T = (int)std::max<int64_t>(4, std::min<int64_t>(16, (int64_t)g.nz * per_plane_block / std::max<int64_t>(1, want)));
    }
    g.T = std::max(1, std::min(T, g.nz));
    g.nzb = (g.nz + g.T - 1) / g.T;
    if ((((size_t)M.pat_np * 28 + 1) & ~(size_t)1) * 8 + (size_t)4 * g.W * kZ2Slot * 8 > (size_t)150 * 1024) return false;   // table + ring in one CU's LDS
    if ((int64_t)g.nzb * g.nyt * g.nxs >= (int64_t)(1 << 30)) return false;
    // constant-coefficient box?  The nine (plane class, line class) patterns from nine rows in the middle of their lines, then every
    // row checked against them on the device (kernels.hpp: z2_box_check_kernel)
    g.box = 0;
    std::memset(g.coef, 0, sizeof(g.coef)); std::memset(g.cmask, 0, sizeof(g.cmask));
    if (opt_int("GMG_PAT_FUSE2_BOX", 1) && g.L >= 3) {
      std::vector<PatEntry> tab((size_t)M.pat_np * 27);
      HIP_CHECK(hipMemcpyAsync(tab.data(), M.ptab, tab.size() * sizeof(PatEntry), hipMemcpyDeviceToHost, stream));
      uint16_t pid[9];
      const int zc[3] = {0, g.nz / 2, g.nz - 1}, yc[3] = {0, g.ny / 2, g.ny - 1};
      for (int c = 0; c < 9; ++c) {
        const int64_t row = (int64_t)zc[c / 3] * g.P + (int64_t)yc[c % 3] * g.L + g.L / 2;
        HIP_CHECK(hipMemcpyAsync(&pid[c], M.rowpid + row, sizeof(uint16_t), hipMemcpyDeviceToHost, stream));
      }
      HIP_CHECK(hipStreamSynchronize(stream));
      std::vector<double> hc(9 * 27);
      std::vector<uint32_t> hm(9 * 27);
      for (int c = 0; c < 9; ++c)
        for (int j = 0; j < 27; ++j) {
          const PatEntry &e = tab[(size_t)pid[c] * 27 + j];
          g.coef[c][j] = e.v; g.cmask[c][j] = e.m; hc[(size_t)c * 27 + j] = e.v; hm[(size_t)c * 27 + j] = e.m;
        }
      double *d_c = nullptr; uint32_t *d_m = nullptr; int *d_bad = nullptr;
      HIP_CHECK(hipMalloc((void **)&d_c, hc.size() * sizeof(double)));
      HIP_CHECK(hipMalloc((void **)&d_m, hm.size() * sizeof(uint32_t)));
      HIP_CHECK(hipMalloc((void **)&d_bad, sizeof(int)));
      int bad = 0;
      hipError_t e1 = hipMemcpyAsync(d_c, hc.data(), hc.size() * sizeof(double), hipMemcpyHostToDevice, stream);
      hipError_t e2 = hipMemcpyAsync(d_m, hm.data(), hm.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream);
      hipError_t e3 = hipMemsetAsync(d_bad, 0, sizeof(int), stream);
      if (e1 == hipSuccess && e2 == hipSuccess && e3 == hipSuccess) {
        hipLaunchKernelGGL(z2_box_check_kernel, dim3((unsigned)std::min<int64_t>(4096, (M.nrows + 255) / 256)), dim3(256), 0, stream, M.nrows, g.L, g.ny, g.nz, g.P,
                           M.rowpid, M.ptab, d_c, d_m, d_bad);
        e1 = hipGetLastError();
        if (e1 == hipSuccess) e1 = hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream);
        if (e1 == hipSuccess) e1 = hipStreamSynchronize(stream);
      }
      (void)hipFree(d_c); (void)hipFree(d_m); (void)hipFree(d_bad);
      HIP_CHECK(e1); HIP_CHECK(e2); HIP_CHECK(e3);
      g.box = bad == 0 ? 1 : 0;
    }
    return true;
  }
  bool fuse2_level(const Level &L) const
  {
    return pat_fuse2 && pat_defer && comm.nranks == 1 && rsweep_level(L) && (pat_fuse2 >= 2 || L.A.nrows >= pat_fuse2_rows) && L.A.z2_ok > 0;
  }
  // sweeps k and k + 1 of a pass in one launch: r_k -> r_{k+2}, x += s_k + s_{k+1}
  void launch_z2sweep(const DevCSR &M, const double *r_cur, double *r_next, double *x, bool x_zero, double omega)
  {
    SellSArgs a;
    std::memset(&a, 0, sizeof(a));
    a.rowpid = M.rowpid; a.tab = M.ptab; a.tab8 = M.ptab8; a.run_off = M.prun; a.np = M.pat_np; a.nruns = M.pat_nruns;
    a.minoff = M.pat_minoff; a.maxoff = M.pat_maxoff; a.xmode = 2; a.pdinv = M.pdinv;
    a.nrows = M.nrows; a.ncols = M.ncols; a.xcd_remap = xcd_remap;
    a.x_zero = x_zero ? 1 : 0; a.x = r_cur; a.omega = omega; a.y = r_next; a.b = r_cur; a.x2 = x;
    const Z2Geo &g = M.z2;
    const bool mk = pat_strict || !M.ptab8;
    const bool bc = g.box != 0;
    const size_t lds = (bc ? 0 : (((size_t)M.pat_np * 28 + 1) & ~(size_t)1) * 8) + (size_t)4 * g.W * kZ2Slot * 8;
    const dim3 gr((unsigned)(g.nzb * g.nyt * g.nxs)), b(64 * g.W);
    M.note_sweep("sells_z2sweep_kernel<MK=%d,FM=%d,BC=%d> (two sweeps per launch) wgs=%u W=%d T=%d tiles=%dx%dx%d L=%d", mk ? 1 : 0, pat_fma ? 1 : 0, bc ? 1 : 0, gr.x, g.W, g.T, g.nxs, g.nyt, g.nzb, g.L);
    REQUIRE(lds <= (size_t)150 * 1024, GMG_ERR_UNSUPPORTED, "sells_z2sweep_kernel: pattern table + ring exceed the LDS of a CU");
    static bool attr[64] = {false};
    if (!attr[device & 63]) {
#define GMG_Z2_ATTR(MKV, FMV, BCV) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&sells_z2sweep_kernel<MKV, FMV, BCV>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
      GMG_Z2_ATTR(true, true, true); GMG_Z2_ATTR(true, false, true); GMG_Z2_ATTR(false, true, true); GMG_Z2_ATTR(false, false, true);
      GMG_Z2_ATTR(true, true, false); GMG_Z2_ATTR(true, false, false); GMG_Z2_ATTR(false, true, false); GMG_Z2_ATTR(false, false, false);
#undef GMG_Z2_ATTR
      attr[device & 63] = true;
    }
#define GMG_Z2_LAUNCH(BCV)                                                                                                   \
    do {                                                                                                                     \
      if (mk) { if (pat_fma) hipLaunchKernelGGL((sells_z2sweep_kernel<true, true, BCV>), gr, b, lds, stream, a, g);          \
                else hipLaunchKernelGGL((sells_z2sweep_kernel<true, false, BCV>), gr, b, lds, stream, a, g); }               \
      else { if (pat_fma) hipLaunchKernelGGL((sells_z2sweep_kernel<false, true, BCV>), gr, b, lds, stream, a, g);            \
             else hipLaunchKernelGGL((sells_z2sweep_kernel<false, false, BCV>), gr, b, lds, stream, a, g); }                 \
    } while (0)
    if (bc) GMG_Z2_LAUNCH(true); else GMG_Z2_LAUNCH(false);
#undef GMG_Z2_LAUNCH
    HIP_CHECK(hipGetLastError());
  }
  void ensure_z2(const DevCSR &M)
  {
    if (M.z2_ok >= 0) return;
    M.z2_ok = z2_geo(M, M.z2) ? 1 : 0;
    if (M.z2_ok > 0) {
      // the single box sweep walks the same grid with chains of its own: segments of <= 126 rows, enough chains for ~4 waves per SIMD
      Z2Geo &b = M.zbx;
      b = M.z2;
      b.nxs = b.whole ? 1 : (b.L + 125) / 126;
      b.xlen = b.whole ? b.L : (b.L + b.nxs - 1) / b.nxs;
      int T = pat_box_t;
      if (T <= 0) T = (int)std::max<int64_t>(2, std::min<int64_t>(12, (int64_t)b.nz * b.ny * b.nxs / std::max(1, 16 * n_cus)));
      b.T = std::max(1, std::min(T, b.nz));
      b.nzb = (b.nz + b.T - 1) / b.T;
    }
  }
  // (what box_level decided, for the byte counts: valid once a sweep of the level has been launched)
  bool box_active(const DevCSR &M) const
  {
    return pat_box && pat_r2 && (pat_box >= 2 || (M.nrows >= pat_box_min_rows && M.nrows <= pat_box_max_rows)) && M.z2_ok > 0 && M.z2.box != 0;
  }
  bool box_level(const DevCSR &M)
  {
    if (!pat_box || !pat_r2) return false;
    if (pat_box < 2 && (M.nrows < pat_box_min_rows || M.nrows > pat_box_max_rows)) return false;
    ensure_z2(M);
    return M.z2_ok > 0 && M.z2.box != 0;
  }
  void launch_rsweep(const DevCSR &M, const double *r_cur, double *r_next, const double *r_prev, double *x, bool x_zero, double omega, int xmode)
  {
    SellSArgs a;
    std::memset(&a, 0, sizeof(a));
    a.rowpid = M.rowpid; a.tab = M.ptab; a.tab8 = M.ptab8; a.run_off = M.prun; a.np = M.pat_np; a.nruns = M.pat_nruns;
    a.minoff = M.pat_minoff; a.maxoff = M.pat_maxoff; a.xmode = xmode; a.pdinv = M.pdinv;
    const int rows = 65 - M.pat_k;
    const int nsl = (int)((M.nrows + rows - 1) / rows);
    a.nrows = M.nrows; a.ncols = M.ncols; a.nslices = nsl; a.xcd_remap = xcd_remap;
    a.x_zero = x_zero ? 1 : 0; a.x = r_cur; a.omega = omega; a.y = r_next; a.b = r_cur; a.x2 = x; a.s_out = const_cast<double *>(r_prev);
    if ((pat_tile >= 2 || (pat_tile == 1 && M.nrows >= pat_tile_rows)) && launch_tsweep(M, a, nsl)) return;
    const int wpb = sell_block > 0 ? sell_block / 64 : (nsl >= 256 * 32 ? 4 : pat_small_wpb);
    const int nu = M.pat_k * M.pat_nruns;
    const int nb = pat_nb > 0 ? pat_nb : (nsl >= 200000 ? 2 : 1);
    const int wg2 = std::max(1, std::min((nsl + wpb - 1) / wpb, nb >= 2 && pat_wgs == 2048 ? 1024 : pat_wgs));
    const dim3 g2(wg2), b(64 * wpb);
    const size_t lds2 = (size_t)M.pat_np * nu * 16 + 16;
    const bool mk = pat_strict || !M.ptab8;
    // constant-coefficient box: one coefficient set per wave from the kernel arguments, no pattern ids, no LDS
    if (box_level(M)) {
      const Z2Geo &gb = M.zbx;
      const int nch = gb.nzb * gb.ny * gb.nxs;
      const dim3 gx((unsigned)((nch + 3) / 4)), bx(256);
      M.note_sweep("sells_boxsweep_kernel<XM=*,MK=%d,FM=%d> chains=%d T=%d segs=%d L=%d", mk ? 1 : 0, pat_fma ? 1 : 0, nch, gb.T, gb.nxs, gb.L);
#define GMG_BOX_LAUNCH(XMV)                                                                                      \
      do {                                                                                                       \
        if (mk) { if (pat_fma) hipLaunchKernelGGL((sells_boxsweep_kernel<XMV, true, true>), gx, bx, 0, stream, a, gb);      \
                  else hipLaunchKernelGGL((sells_boxsweep_kernel<XMV, true, false>), gx, bx, 0, stream, a, gb); }           \
        else { if (pat_fma) hipLaunchKernelGGL((sells_boxsweep_kernel<XMV, false, true>), gx, bx, 0, stream, a, gb);        \
               else hipLaunchKernelGGL((sells_boxsweep_kernel<XMV, false, false>), gx, bx, 0, stream, a, gb); }             \
      } while (0)
      if (xmode == 0) GMG_BOX_LAUNCH(0); else if (xmode == 1) GMG_BOX_LAUNCH(1); else GMG_BOX_LAUNCH(2);
#undef GMG_BOX_LAUNCH
      HIP_CHECK(hipGetLastError());
      return;
    }
    // the pair sweep as a walk along the slowest grid direction: three new windows per step instead of nine, one step of requests in flight
    ZWalkGeo zg;
    if (pat_r2 && pat_zwalk && (pat_zwalk >= 2 || M.nrows >= pat_zwalk_rows) && zwalk_geo(M, zg)) {
      const int wz = 4;
      const dim3 gz((unsigned)((zg.nchains + wz - 1) / wz)), bz(64 * wz);
      const size_t ldsz = (size_t)M.pat_np * 28 * 8 + 16;       // patterns padded to 28 doubles (16-byte aligned coefficient pairs)
      M.note_sweep("sells_zsweep_kernel<XM=*,MK=%d,FM=%d> chains=%d P=%d m=%d T=%d", mk ? 1 : 0, pat_fma ? 1 : 0, zg.nchains, zg.P, zg.m, zg.T);
#define GMG_ZW_LAUNCH(XMV)                                                                                       \
      do {                                                                                                       \
        if (mk) { if (pat_fma) hipLaunchKernelGGL((sells_zsweep_kernel<XMV, true, true>), gz, bz, ldsz, stream, a, zg);     \
                  else hipLaunchKernelGGL((sells_zsweep_kernel<XMV, true, false>), gz, bz, ldsz, stream, a, zg); }          \
        else { if (pat_fma) hipLaunchKernelGGL((sells_zsweep_kernel<XMV, false, true>), gz, bz, ldsz, stream, a, zg);       \
               else hipLaunchKernelGGL((sells_zsweep_kernel<XMV, false, false>), gz, bz, ldsz, stream, a, zg); }            \
      } while (0)
      if (xmode == 0) GMG_ZW_LAUNCH(0); else if (xmode == 1) GMG_ZW_LAUNCH(1); else GMG_ZW_LAUNCH(2);
#undef GMG_ZW_LAUNCH
      HIP_CHECK(hipGetLastError());
      return;
    }
    // two rows per lane (kernels.hpp: sells_r2sweep_kernel): slices of 126 rows, 16-byte loads / stores, half the conversions and shifts per row
    if (pat_r2 && (M.pat_nruns == 9 || M.pat_nruns == 3)) {
      const int nsl2 = (int)((M.nrows + 125) / 126);
      a.nslices = nsl2;
      // one round of workgroups: four per CU at 105 registers (128^3: 17.4 us per sweep; 768: 19.2, 1280: 20.8, 2048: 19.1 --
      // profiles/r04_tuning.md), eight per CU for the 64-register form (pat_r2_occ: rolled run loop; 15.7 -> 14.4 us, section 13)
      // Levels of >= pat_tile_rows rows (256^3 and up): the 64-register form with ONE slice per wave -- workgroups are dealt to the CUs
      // as they retire, neighbouring slices run at the same time on the same XCD and meet in its L2 -- beats the tile sweep that
      // these levels took before: 256^3 134 / 185 -> 123 / 166 us per sweep by variant, 288^3 184 / 271 -> 173 / 253 (section 14)
      const bool occ = pat_r2_occ && wpb == 4 && M.pat_nruns == 9;
      // one slice per wave: workgroups of eight waves stage the coefficient table once per eight slices (pat_r2_occ = 2)
      const bool occ8 = occ && pat_r2_occ >= 2 && M.nrows >= pat_tile_rows && pat_r2_wgs <= 0;
      if (occ8) {
        const dim3 g8((nsl2 + 7) / 8), b8(512);
        M.note_sweep("sells_r2sweep_kernel<XM=*,MK=%d,FM=%d,NR=%d,OCC=2> wgs=%d wpb=8", mk ? 1 : 0, pat_fma ? 1 : 0, M.pat_nruns, (int)g8.x);
#define GMG_R2_LAUNCH8(XMV)                                                                                      \
        do {                                                                                                     \
          if (mk) { if (pat_fma) hipLaunchKernelGGL((sells_r2sweep_kernel<XMV, true, true, 9, 2>), g8, b8, lds2, stream, a);   \
                    else hipLaunchKernelGGL((sells_r2sweep_kernel<XMV, true, false, 9, 2>), g8, b8, lds2, stream, a); }        \
          else { if (pat_fma) hipLaunchKernelGGL((sells_r2sweep_kernel<XMV, false, true, 9, 2>), g8, b8, lds2, stream, a);     \
                 else hipLaunchKernelGGL((sells_r2sweep_kernel<XMV, false, false, 9, 2>), g8, b8, lds2, stream, a); }          \
        } while (0)
        if (xmode == 0) GMG_R2_LAUNCH8(0); else if (xmode == 1) GMG_R2_LAUNCH8(1); else GMG_R2_LAUNCH8(2);
#undef GMG_R2_LAUNCH8
        HIP_CHECK(hipGetLastError());
        return;
      }
      const int full = (nsl2 + wpb - 1) / wpb;
      const int wgr = std::max(1, std::min(full, pat_r2_wgs > 0 ? pat_r2_wgs : (occ && M.nrows >= pat_tile_rows ? full : (occ ? 8 : 4) * n_cus)));
      const dim3 gr(wgr);
      M.note_sweep("sells_r2sweep_kernel<XM=*,MK=%d,FM=%d,NR=%d,OCC=%d> wgs=%d wpb=%d", mk ? 1 : 0, pat_fma ? 1 : 0, M.pat_nruns, occ ? 1 : 0, wgr, wpb);
#define GMG_R2_LAUNCH2(XMV, MKV, FMV)                                                                            \
      do {                                                                                                       \
        if (occ) hipLaunchKernelGGL((sells_r2sweep_kernel<XMV, MKV, FMV, 9, 1>), gr, b, lds2, stream, a);        \
        else if (M.pat_nruns == 9) hipLaunchKernelGGL((sells_r2sweep_kernel<XMV, MKV, FMV, 9>), gr, b, lds2, stream, a);   \
        else hipLaunchKernelGGL((sells_r2sweep_kernel<XMV, MKV, FMV, 3>), gr, b, lds2, stream, a);               \
      } while (0)
#define GMG_R2_LAUNCH(XMV)                                                                                       \
      do {                                                                                                       \
        if (mk) { if (pat_fma) GMG_R2_LAUNCH2(XMV, true, true); else GMG_R2_LAUNCH2(XMV, true, false); }         \
        else { if (pat_fma) GMG_R2_LAUNCH2(XMV, false, true); else GMG_R2_LAUNCH2(XMV, false, false); }          \
      } while (0)
      if (xmode == 0) GMG_R2_LAUNCH(0); else if (xmode == 1) GMG_R2_LAUNCH(1); else GMG_R2_LAUNCH(2);
#undef GMG_R2_LAUNCH2
#undef GMG_R2_LAUNCH
      HIP_CHECK(hipGetLastError());
      return;
    }
    M.note_sweep("sells_rsweep_kernel<XM=*,NB=%d,MK=%d,FM=%d> wgs=%d wpb=%d", nb >= 2 ? 2 : 1, mk ? 1 : 0, pat_fma ? 1 : 0, wg2, wpb);
#define GMG_RSWEEP_LAUNCH(XMV, NBV)                                                                            \
    do {                                                                                                         \
      if (mk) { if (pat_fma) hipLaunchKernelGGL((sells_rsweep_kernel<XMV, NBV, true, 0, true>), g2, b, lds2, stream, a);   \
                else hipLaunchKernelGGL((sells_rsweep_kernel<XMV, NBV, true>), g2, b, lds2, stream, a); }        \
      else { if (pat_fma) hipLaunchKernelGGL((sells_rsweep_kernel<XMV, NBV, false, 0, true>), g2, b, lds2, stream, a);     \
             else hipLaunchKernelGGL((sells_rsweep_kernel<XMV, NBV, false>), g2, b, lds2, stream, a); }          \
    } while (0)
    if (nb >= 2) { if (xmode == 0) GMG_RSWEEP_LAUNCH(0, 2); else if (xmode == 1) GMG_RSWEEP_LAUNCH(1, 2); else GMG_RSWEEP_LAUNCH(2, 2); }
    else { if (xmode == 0) GMG_RSWEEP_LAUNCH(0, 1); else if (xmode == 1) GMG_RSWEEP_LAUNCH(1, 1); else GMG_RSWEEP_LAUNCH(2, 1); }
#undef GMG_RSWEEP_LAUNCH
    HIP_CHECK(hipGetLastError());
  }
  // one sweep of that form: x (+)= s_k ; r_{k+1} = r_k - A s_k with s_k = omega*(d*r_k) formed from the gathered r_k
  void rsweep(int l, const Smoother &S, double *x, const double *r_cur, double *r_next, const double *r_prev, bool x_zero, int xmode,
              bool prepacked, bool pack_next)
  {
    Level &L = lev[l];
    const bool halo_sweep = comm.nranks > 1 && L.halo.present && !L.halo.ovl;
    if (halo_sweep) begin_exchange(l, const_cast<double *>(r_cur), prepacked);     // the ghosts of r_k
    const bool prof = (l == prof_level) && (prof_seq++ % (uint64_t)prof_stride == 0) && prof_used + 2 <= prof_ev.size();
    if (prof) HIP_CHECK(hipEventRecord(prof_ev[prof_used], stream));
    launch_rsweep(L.A, r_cur, r_next, r_prev, x, x_zero, S.omega, xmode);
    if (prof) {
      HIP_CHECK(hipEventRecord(prof_ev[prof_used + 1], stream));
      prof_w[prof_used / 2] = 1;
      prof_xm[prof_used / 2] = (int8_t)xmode;
      prof_used += 2;
    }
    if (halo_sweep) {
      if (overlapped()) HIP_CHECK(hipStreamWaitEvent(stream, ev_done, 0));
      if (L.split && L.nbnd > 0) {
        const bool pk = pack_next && L.halo.d_pk_ptr != nullptr;
        if (L.gh_scode)
          hipLaunchKernelGGL((ghost_fix_sell_kernel<3, true>), dim3((unsigned)((L.nbnd + 255) / 256)), dim3(256), 0, stream, L.nbnd, L.gh_rows,
                             L.gh_len, L.gh_soff, L.gh_scol, (const void *)L.gh_scode, L.gh_dict, r_cur, r_next, (const double *)nullptr, S.omega, (double *)nullptr,
                             pk ? L.halo.d_pk_ptr : nullptr, pk ? L.halo.d_pk_slot : nullptr, pk ? L.halo.d_sendbuf : nullptr, L.A.pdinv_u);
        else if (L.gh_sval)
          hipLaunchKernelGGL((ghost_fix_sell_kernel<3, false>), dim3((unsigned)((L.nbnd + 255) / 256)), dim3(256), 0, stream, L.nbnd, L.gh_rows,
                             L.gh_len, L.gh_soff, L.gh_scol, (const void *)L.gh_sval, (const double *)nullptr, r_cur, r_next, (const double *)nullptr, S.omega, (double *)nullptr,
                             pk ? L.halo.d_pk_ptr : nullptr, pk ? L.halo.d_pk_slot : nullptr, pk ? L.halo.d_sendbuf : nullptr, L.A.pdinv_u);
        else
        hipLaunchKernelGGL((ghost_fix_kernel<3>), dim3((unsigned)((L.nbnd + 255) / 256)), dim3(256), 0, stream, L.nbnd, L.gh_rows,
                           L.gh_ptr, L.gh_col, L.gh_val, r_cur, r_next, (const double *)nullptr, S.omega, (double *)nullptr,
                           pk ? L.halo.d_pk_ptr : nullptr, pk ? L.halo.d_pk_slot : nullptr, pk ? L.halo.d_sendbuf : nullptr, L.A.pdinv_u);
        HIP_CHECK(hipGetLastError());
      }
    }
  }
  // fused Richardson-Jacobi sweep: x += w*Dinv*r_old ; r_new = r_old - A*(w*Dinv*r_old)
  // one-gather form: s_old = w*Dinv*r_old is an input, s_new = w*Dinv*r_new an output.
  void sweep(int l, const Smoother &S, double *x, const double *r_old, double *r_new, bool x_zero,
             const double *s_old = nullptr, double *s_new = nullptr, int xmode = 0, bool prepacked = false, bool pack_next = false)
  {
    Level &L = lev[l];
    const bool halo_sweep = comm.nranks > 1 && !(L.halo.present && L.halo.ovl);   // overlapping layout: smooth() exchanges once per block of sweeps
    if (halo_sweep) begin_exchange(l, const_cast<double *>(s_old ? s_old : r_old), prepacked);
    // HIP events around every prof_stride-th sweep launch of the profiled level: an event pair costs ~4 us of
    // stream time, timing every launch would slow the solve it measures by > 10 %
    const bool prof = (l == prof_level) && (prof_seq++ % (uint64_t)prof_stride == 0) && prof_used + 2 <= prof_ev.size();
    if (prof) HIP_CHECK(hipEventRecord(prof_ev[prof_used], stream));
    {
      StreamArgs2 a = base_args1(L.A);
      a.b = r_old; a.dinv = L.dinv; a.omega = S.omega; a.y = r_new; a.x2 = x; a.x_zero = x_zero ? 1 : 0;
      a.xmode = xmode; a.dinv_from_table = pat_dinv;      // L.dinv is 1 ./ diag(L.A): the pattern table holds the same numbers
      if (L.A.sell) {
        if (s_old) { a.x = s_old; a.s_out = s_new; launch_sell<EPI_SWEEP, true>(L.A, a); }
        else { a.x = r_old; launch_sell<EPI_SWEEP, false>(L.A, a); }
      } else
      if (s_old) { a.x = s_old; a.s_out = s_new; launch_stream1<EPI_SWEEP, true, false>(L.A, a); }
      else { a.x = r_old; launch_stream1<EPI_SWEEP, false, false>(L.A, a); }
    }
    if (prof) {
      HIP_CHECK(hipEventRecord(prof_ev[prof_used + 1], stream));
      prof_w[prof_used / 2] = 1;
      prof_xm[prof_used / 2] = (int8_t)xmode;
      prof_used += 2;
    }
    if (halo_sweep) finish_ghost<2>(l, s_old, r_new, S.omega, s_new, pack_next);   // distributed => one-gather sweep
  }

  void copy(double *dst, const double *src, int64_t n)
  {
    if (n > 0 && dst != src) HIP_CHECK(hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
  }
  void zero(double *dst, int64_t n)
  {
    if (n > 0) HIP_CHECK(hipMemsetAsync(dst, 0, sizeof(double) * (size_t)n, stream));
  }

  // dot -> device scalar slot (no host sync)
  // first stage only: partial sums into `parts` (kRedBlocks doubles); returns how many.  The second stage is either
  // finish_reduction or the consumer kernel itself (sum_partials_all).
  int dot_partials(int64_t n, const double *a, const double *b, double *parts)
  {
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kRedBlocks, (n / 2 + kBlock - 1) / kBlock));
    const bool aligned = ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    hipLaunchKernelGGL(dot_partial_kernel, dim3(nb), dim3(kBlock), 0, stream, n, a, b, parts, aligned ? 1 : 0);
    HIP_CHECK(hipGetLastError());
    return nb;
  }
  void dot_async(int64_t n, const double *a, const double *b, int slot, bool take_sqrt, bool post = false)
  {
    const int nb = dot_partials(n, a, b, d_partials);
    finish_reduction(nb, slot, take_sqrt, d_partials, post);
  }
  // red_fused (default): inside cg_core the second stage of a reduction is done by the kernel that consumes the scalar, and the one
  // the host needs is reduced and posted by one launch -- single rank (or rank-local reductions) only
  int red_fused = 1;                   // GMG_RED_FUSED
  double *d_partials2 = nullptr;
  bool fuse_reductions() const { return red_fused != 0 && (comm.nranks == 1 || reduce_local) && d_partials2 != nullptr; }
  // partials -> d_scalars[slot] = (sqrt of) the GLOBAL sum.  Single GPU: one kernel.  Several
  // ranks: local sum, all-reduce over the parts (the reduction PartitionedArrays performs
  // inside dot/norm on a PVector), then the square root.
  // post: single rank with host_poll -- the scalar is also posted to the host by the same launch (fetch_scalar then only waits)
  void finish_reduction(int nb, int slot, bool take_sqrt, const double *parts = nullptr, bool post = false)
  {
    if (!parts) parts = d_partials;
    posted = 0;                                              // (a number left over by a call that threw before its fetch)
    const bool dist = comm.nranks > 1 && !reduce_local;
    if (post && !dist && red_fused && opt_int("GMG_HOST_POLL", 1)) {
      need_mail();
      posted = ++mail_seq;
      hipLaunchKernelGGL(reduce_post_kernel, dim3(1), dim3(kBlock), 0, stream, nb, parts, d_scalars + slot, take_sqrt ? 1 : 0,
                         &d_mail->value, &d_mail->seq, posted);
      HIP_CHECK(hipGetLastError());
      return;
    }
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(kBlock), 0, stream, nb, parts, d_scalars + slot,
                       (take_sqrt && !dist) ? 1 : 0);
    HIP_CHECK(hipGetLastError());
    if (!dist) return;
    ++n_allreduces;
    if (comm.kind == COMM_RCCL) {
      const int rc = comm.api.AllReduce(d_scalars + slot, d_scalars + slot, 1, kNcclDouble, kNcclSum, comm.comm, stream);
      REQUIRE(rc == 0, GMG_ERR_COMM, std::string("ncclAllReduce: ") + comm.api.GetErrorString(rc));
      if (take_sqrt) {
        hipLaunchKernelGGL(sqrt_inplace_kernel, dim3(1), dim3(1), 0, stream, d_scalars + slot);
        HIP_CHECK(hipGetLastError());
      }
    } else {
      HIP_CHECK(hipMemcpyAsync(h_scalars + slot, d_scalars + slot, sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      double v = h_scalars[slot];
      comm.rfn(comm.ctx, &v, 1);
      h_scalars[slot] = take_sqrt ? std::sqrt(v) : v;
      HIP_CHECK(hipMemcpyAsync(d_scalars + slot, h_scalars + slot, sizeof(double), hipMemcpyHostToDevice, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    }
  }
  // consistent!(v): owner -> ghost copy of the level-l vector `v` (length nvec)
  void exchange(int l, double *v) { exchange_on(l, v, stream); }
  // prepacked: the send buffer already holds v's boundary entries (written by the previous sweep's ghost_fix_kernel)
  void exchange_on(int l, double *v, hipStream_t stream, bool prepacked = false) { exchange_plan(lev[l].halo, v, stream, prepacked); }
  // the same for any exchange plan (block preconditioners keep one per block)
  // (dst: where the received values go when it is not the vector the sent ones come from -- redistribution between two partitions)
  void exchange_plan(HaloPlan &H, double *v, hipStream_t stream, bool prepacked = false, double *dst = nullptr)
  {
    if (comm.nranks <= 1 || !H.present || H.nbr.empty()) return;
    if (!dst) dst = v;
    const int64_t ns = H.nsend();
    if (ns > 0 && !prepacked) {
      hipLaunchKernelGGL(halo_pack_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, stream, ns, H.d_snd_idx, v, H.d_sendbuf);
      HIP_CHECK(hipGetLastError());
    }
    // own | ghost layout: received straight into the ghost segment; overlapping layout: into a landing buffer, then scattered
    double *ghost = H.ovl ? H.d_unpack : dst + H.n_own;
    if (comm.kind == COMM_RCCL) {
      int rc = comm.api.GroupStart();
      for (size_t k = 0; k < H.nbr.size() && rc == 0; ++k) {
        const int64_t sc = H.snd_ptr[k + 1] - H.snd_ptr[k], rcn = H.rcv_ptr[k + 1] - H.rcv_ptr[k];
        if (sc > 0) rc = comm.api.Send(H.d_sendbuf + H.snd_ptr[k], (size_t)sc, kNcclDouble, comm.peer(H.nbr[k]), comm.comm, stream);
        if (rc == 0 && rcn > 0) rc = comm.api.Recv(ghost + H.rcv_ptr[k], (size_t)rcn, kNcclDouble, comm.peer(H.nbr[k]), comm.comm, stream);
      }
      const int rc2 = comm.api.GroupEnd();
      REQUIRE(rc == 0 && rc2 == 0, GMG_ERR_COMM, std::string("RCCL halo exchange: ") + comm.api.GetErrorString(rc ? rc : rc2));
    } else {
      if (ns > 0) HIP_CHECK(hipMemcpyAsync(H.h_send, H.d_sendbuf, sizeof(double) * (size_t)ns, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      comm.xfn(comm.ctx, (int)H.nbr.size(), H.nbr.data(), H.h_send, H.snd_ptr.data(), H.h_recv, H.rcv_ptr.data());
      if (H.n_ghost > 0) HIP_CHECK(hipMemcpyAsync(ghost, H.h_recv, sizeof(double) * (size_t)H.n_ghost, hipMemcpyHostToDevice, stream));
    }
    if (H.ovl && H.n_ghost > 0) {
      hipLaunchKernelGGL(halo_unpack_kernel, dim3((unsigned)((H.n_ghost + 255) / 256)), dim3(256), 0, stream, H.n_ghost, H.d_rcv_idx, H.d_unpack, dst);
      HIP_CHECK(hipGetLastError());
    }
    ++n_exchanges;
  }
  // redistribute!(dst, src) between the two partitions of level sub_from: entries that stay on this rank are copied, the others travel
  // with one grouped send / receive (same primitive as a halo exchange: pack by local id, receive, scatter by local id)
  void redistribute(HaloPlan &H, int self, const double *src, double *dst)
  {
    const int64_t nself = (int64_t)redist.h_self[self].size();
    if (nself > 0) {
      hipLaunchKernelGGL(gather_scatter_kernel, dim3((unsigned)((nself + 255) / 256)), dim3(256), 0, stream, nself, redist.d_self[self], redist.d_self[self + 1], src, dst);
      HIP_CHECK(hipGetLastError());
    }
    const int64_t before = n_exchanges;
    exchange_plan(H, const_cast<double *>(src), stream, false, dst);
    n_exchanges = before;
    ++n_redist;
  }
  // a rank outside the subset takes part in the collectives below level sub_from-1 without holding the levels: the all-reduce that
  // assembles the replicated residual (zero contribution); mirrors cycle()'s recursion
  void shadow_cycle(int l, int ctype)
  {
    if (l >= nlev - 1 || l >= rep_from) return;
    const int passes = (ctype == GMG_V_CYCLE) ? 1 : 2;
    for (int pass = 0; pass < passes; ++pass) {
      const int child = (pass == 0) ? ctype : (ctype == GMG_W_CYCLE ? GMG_W_CYCLE : GMG_V_CYCLE);
      if (l + 1 == rep_from) restrict_replicate(l, nullptr, lev[l + 1].rbuf[0]);
      else shadow_cycle(l + 1, child);
    }
  }
  void alloc_plan_buffers(HaloPlan &H)
  {
    H.d_snd_idx = upload(H.h_snd_idx);
    H.d_sendbuf = dvec(H.nsend());
    H.d_recvbuf = dvec(H.nsend());
    if (H.ovl) { H.d_rcv_idx = upload(H.h_rcv_idx); H.d_unpack = dvec(H.n_ghost); }
    if (comm.kind == COMM_HOST) {
      if (!H.h_send) HIP_CHECK(hipHostMalloc((void **)&H.h_send, sizeof(double) * (size_t)std::max<int64_t>(1, H.nsend())));
      if (!H.h_recv) HIP_CHECK(hipHostMalloc((void **)&H.h_recv, sizeof(double) * (size_t)std::max<int64_t>(1, H.n_ghost)));
    }
  }
  // assemble!(v): ghost -> owner add (the reverse of consistent!), PatchSolvers.jl:254 / BlockJacobiSolvers.jl:134.
  // v has n_own + n_ghost entries; afterwards the owned entries hold own + all ghost copies' contributions.
  void assemble_add(int l, double *v)
  {
    HaloPlan &H = lev[l].halo;
    if (comm.nranks <= 1 || !H.present || H.nbr.empty()) return;
    REQUIRE(!H.ovl, GMG_ERR_UNSUPPORTED, "assemble! (reverse halo) is not available in the overlapping layout");
    const int64_t ns = H.nsend();
    double *ghost = v + H.n_own;
    if (comm.kind == COMM_RCCL) {
      int rc = comm.api.GroupStart();
      for (size_t k = 0; k < H.nbr.size() && rc == 0; ++k) {
        const int64_t sc = H.snd_ptr[k + 1] - H.snd_ptr[k], rcn = H.rcv_ptr[k + 1] - H.rcv_ptr[k];
        if (rcn > 0) rc = comm.api.Send(ghost + H.rcv_ptr[k], (size_t)rcn, kNcclDouble, comm.peer(H.nbr[k]), comm.comm, stream);
        if (rc == 0 && sc > 0) rc = comm.api.Recv(H.d_recvbuf + H.snd_ptr[k], (size_t)sc, kNcclDouble, comm.peer(H.nbr[k]), comm.comm, stream);
      }
      const int rc2 = comm.api.GroupEnd();
      REQUIRE(rc == 0 && rc2 == 0, GMG_ERR_COMM, std::string("RCCL reverse halo: ") + comm.api.GetErrorString(rc ? rc : rc2));
    } else {
      if (H.n_ghost > 0) HIP_CHECK(hipMemcpyAsync(H.h_recv, ghost, sizeof(double) * (size_t)H.n_ghost, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      // roles swapped: the ghost segment is what is sent, the send lists are what is received
      comm.xfn(comm.ctx, (int)H.nbr.size(), H.nbr.data(), H.h_recv, H.rcv_ptr.data(), H.h_send, H.snd_ptr.data());
      if (ns > 0) HIP_CHECK(hipMemcpyAsync(H.d_recvbuf, H.h_send, sizeof(double) * (size_t)ns, hipMemcpyHostToDevice, stream));
    }
    for (size_t k = 0; k < H.nbr.size(); ++k) {
      const int64_t sc = H.snd_ptr[k + 1] - H.snd_ptr[k];
      if (sc == 0) continue;
      hipLaunchKernelGGL(halo_unpack_add_kernel, dim3((unsigned)((sc + 255) / 256)), dim3(256), 0, stream, sc, H.d_snd_idx + H.snd_ptr[k],
                         H.d_recvbuf + H.snd_ptr[k], v);
      HIP_CHECK(hipGetLastError());
    }
  }
  // Split mat-vec with A_l: start the halo of `src`, (caller runs the own x own kernel), then
  // finish the boundary rows.  With RCCL the exchange runs on comm_stream concurrently with the
  // own x own kernel; with the host transport it is synchronous (same data flow, no overlap).
  bool overlapped() const { return overlap && comm_stream && (comm.kind == COMM_RCCL || host_async); }
  static void host_exchange_trampoline(void *p)
  {
    // runs on a HIP runtime thread, in comm_stream order; must not call HIP
    auto *c = static_cast<Level::HostXfer *>(p);
    HaloPlan &H = c->solver->lev[c->level].halo;
    c->solver->comm.xfn(c->solver->comm.ctx, (int)H.nbr.size(), H.nbr.data(), H.h_send, H.snd_ptr.data(), H.h_recv, H.rcv_ptr.data());
  }
  void begin_exchange(int l, double *src, bool prepacked = false)
  {
    if (comm.nranks <= 1 || !lev[l].halo.present) return;
    if (lev[l].halo.ovl) { exchange_on(l, src, stream, false); return; }   // overlapping layout: whole rows, nothing to overlap with
    if (overlapped() && comm.kind == COMM_RCCL) {
      HIP_CHECK(hipEventRecord(ev_ready, stream));
      HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_ready, 0));
      exchange_on(l, src, comm_stream, prepacked);
      HIP_CHECK(hipEventRecord(ev_done, comm_stream));
    } else if (overlapped()) {
      // host transport, asynchronous flavour (tests): the same two-stream / two-event schedule as the
      // RCCL path, with the host callback enqueued on comm_stream
      HaloPlan &H = lev[l].halo;
      HIP_CHECK(hipEventRecord(ev_ready, stream));
      HIP_CHECK(hipStreamWaitEvent(comm_stream, ev_ready, 0));
      const int64_t ns = H.nsend();
      if (ns > 0) {
        if (!prepacked) {
          hipLaunchKernelGGL(halo_pack_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, comm_stream, ns, H.d_snd_idx, src, H.d_sendbuf);
          HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipMemcpyAsync(H.h_send, H.d_sendbuf, sizeof(double) * (size_t)ns, hipMemcpyDeviceToHost, comm_stream));
      }
      lev[l].hostctx = {this, l};
      HIP_CHECK(hipLaunchHostFunc(comm_stream, host_exchange_trampoline, &lev[l].hostctx));
      if (H.n_ghost > 0) HIP_CHECK(hipMemcpyAsync(src + H.n_own, H.h_recv, sizeof(double) * (size_t)H.n_ghost, hipMemcpyHostToDevice, comm_stream));
      HIP_CHECK(hipEventRecord(ev_done, comm_stream));
      ++n_exchanges;
    } else {
      exchange_on(l, src, stream, prepacked);
    }
  }
  // pack_next (MODE 2): also fill the send buffer with the new s of the boundary rows for the next sweep's exchange
  template <int MODE>
  void finish_ghost(int l, const double *src, double *y, double omega = 0.0, double *s_out = nullptr, bool pack_next = false)
  {
    Level &L = lev[l];
    if (comm.nranks <= 1 || !L.halo.present || L.halo.ovl) return;
    if (overlapped()) HIP_CHECK(hipStreamWaitEvent(stream, ev_done, 0));
    if (!L.split || L.nbnd == 0) return;
    const bool pk = pack_next && MODE == 2 && L.halo.d_pk_ptr != nullptr;
    if (L.gh_scode)
      hipLaunchKernelGGL((ghost_fix_sell_kernel<MODE, true>), dim3((unsigned)((L.nbnd + 255) / 256)), dim3(256), 0, stream, L.nbnd, L.gh_rows,
                         L.gh_len, L.gh_soff, L.gh_scol, (const void *)L.gh_scode, L.gh_dict, src, y, L.dinv, omega, s_out, pk ? L.halo.d_pk_ptr : nullptr,
                         pk ? L.halo.d_pk_slot : nullptr, pk ? L.halo.d_sendbuf : nullptr);
    else if (L.gh_sval)
      hipLaunchKernelGGL((ghost_fix_sell_kernel<MODE, false>), dim3((unsigned)((L.nbnd + 255) / 256)), dim3(256), 0, stream, L.nbnd, L.gh_rows,
                         L.gh_len, L.gh_soff, L.gh_scol, (const void *)L.gh_sval, (const double *)nullptr, src, y, L.dinv, omega, s_out, pk ? L.halo.d_pk_ptr : nullptr,
                         pk ? L.halo.d_pk_slot : nullptr, pk ? L.halo.d_sendbuf : nullptr);
    else
    hipLaunchKernelGGL((ghost_fix_kernel<MODE>), dim3((unsigned)((L.nbnd + 255) / 256)), dim3(256), 0, stream, L.nbnd, L.gh_rows,
                       L.gh_ptr, L.gh_col, L.gh_val, src, y, L.dinv, omega, s_out, pk ? L.halo.d_pk_ptr : nullptr,
                       pk ? L.halo.d_pk_slot : nullptr, pk ? L.halo.d_sendbuf : nullptr);
    HIP_CHECK(hipGetLastError());
  }
  // boundary rows (CSR over ghost columns) -> the sliced form; values become dictionary codes when there are <= 256 distinct ones
  void build_ghost_fix(GhostFix &G, const std::vector<int32_t> &brows, const std::vector<int64_t> &bptr, const std::vector<int32_t> &bcol,
                       const std::vector<double> &bval)
  {
    G = GhostFix();
    const size_t nbr = brows.size(), nsl = (nbr + 63) / 64;
    if (nbr == 0) return;
    std::vector<int32_t> glen(nbr);
    std::vector<int64_t> soff(nsl + 1, 0);
    for (size_t sl = 0; sl < nsl; ++sl) {
      int64_t w = 0;
      for (size_t q = sl * 64; q < std::min(nbr, sl * 64 + 64); ++q) { glen[q] = (int32_t)(bptr[q + 1] - bptr[q]); w = std::max<int64_t>(w, glen[q]); }
      soff[sl + 1] = soff[sl] + 64 * w;
    }
    std::vector<int32_t> scol((size_t)soff[nsl], 0);
    std::vector<double> sval((size_t)soff[nsl], 0.0);
    for (size_t q = 0; q < nbr; ++q)
      for (int64_t k = bptr[q]; k < bptr[q + 1]; ++k) {
        const size_t at = (size_t)soff[q / 64] + (size_t)(k - bptr[q]) * 64 + (q & 63);
        scol[at] = bcol[(size_t)k]; sval[at] = bval[(size_t)k];
      }
    std::map<uint64_t, int> seen;
    std::vector<double> dict;
    std::vector<uint8_t> code(sval.size(), 0);
    bool small = opt_int("GMG_HALO_FIX_DICT", 1) != 0;
    for (size_t q = 0; q < sval.size() && small; ++q) {
      uint64_t bits; std::memcpy(&bits, &sval[q], 8);
      auto it = seen.find(bits);
      if (it == seen.end()) {
        if (dict.size() == 256) { small = false; break; }
        it = seen.emplace(bits, (int)dict.size()).first;
        dict.push_back(sval[q]);
      }
      code[q] = (uint8_t)it->second;
    }
    G.nb = (int64_t)nbr; G.rows = upload(brows); G.len = upload(glen); G.soff = upload(soff); G.scol = upload(scol);
    if (small) { dict.resize(256, 0.0); G.scode = upload(code); G.dict = upload(dict); }
    else G.sval = upload(sval);
  }
  // y[row] += sum over the row's ghost entries of val * v[col]
  void add_ghost_part(const GhostFix &G, const double *v, double *y)
  {
    if (G.nb <= 0) return;
    const dim3 g((unsigned)((G.nb + 255) / 256)), b(256);
    if (G.scode) hipLaunchKernelGGL((ghost_fix_sell_kernel<0, true>), g, b, 0, stream, G.nb, G.rows, G.len, G.soff, G.scol, (const void *)G.scode, G.dict, v, y,
                                    (const double *)nullptr, 0.0, (double *)nullptr);
    else hipLaunchKernelGGL((ghost_fix_sell_kernel<0, false>), g, b, 0, stream, G.nb, G.rows, G.len, G.soff, G.scol, (const void *)G.sval, (const double *)nullptr, v, y,
                            (const double *)nullptr, 0.0, (double *)nullptr);
    HIP_CHECK(hipGetLastError());
  }
  // dst = R_l r on a level whose R is split (own | ghost layout, several ranks): consistent!(r) runs on the communication stream while
  // the own columns are applied, the ghost columns follow -- the exchange is the caller's otherwise
  void restrict_into(int l, double *r, double *dst)
  {
    Level &L = lev[l];
    if (!L.r_split) { spmv_set(L.R, r, dst); return; }
    begin_exchange(l, r);
    spmv_set(L.R, r, dst);
    if (overlapped()) HIP_CHECK(hipStreamWaitEvent(stream, ev_done, 0));
    add_ghost_part(L.rfix, r, dst);
  }
  bool can_fuse_pack(int l) const { return comm.nranks > 1 && lev[l].halo.present && lev[l].split && lev[l].nbnd > 0 && lev[l].halo.d_pk_ptr != nullptr; }
  // y = A_l x ; y -= A_l x ; y = b - A_l x   with the halo of x folded in
  void apply_A_set(int l, double *x, double *y)
  {
    begin_exchange(l, x);
    spmv_set(lev[l].A, x, y);
    finish_ghost<0>(l, x, y);
  }
  void apply_A_sub(int l, double *x, double *y, const Smoother *next = nullptr)
  {
    begin_exchange(l, x);
    Level &L = lev[l];
    // the smoothing pass that follows starts from s = omega*(Dinv*y): let this kernel write it
    if (next && emits_s0(L, *next, L.A)) {
      spmv_sub(L.A, x, y, L.sbuf[0], L.dinv, next->omega);
      L.s0_ready = true; L.s0_src = y; L.s0_omega = next->omega;
    } else
    spmv_sub(L.A, x, y);
    finish_ghost<1>(l, x, y);
  }
  bool emits_s0(const Level &L, const Smoother &S, const DevCSR &producer)
  {
    if (!(pat_emit && comm.nranks == 1 && producer.pat && S.kind == SM_JACOBI && one_gather() && S.niter > 0 && L.sbuf[0] != nullptr && L.dinv != nullptr)) return false;
    // levels whose sweeps gather r itself have no use for s_0 -- unless the pass runs as one launch (small levels), which starts from it
    const int l = (int)(&L - &lev[0]);
    if (rsweep_level(L) && !smooth_persistent(l, S, nullptr, nullptr, nullptr, false, S.niter, true)) return false;
    return true;
  }
  void apply_A_resid(int l, double *x, const double *b, double *y)
  {
    begin_exchange(l, x);
    spmv_resid(lev[l].A, x, b, y);
    finish_ghost<1>(l, x, y);
  }
  // The one value per Krylov iteration the host needs (the residual norm: the stopping rule of ConvergenceLogs.jl:101-150 is evaluated
  // on the host).  A copy + hipStreamSynchronize leaves the GPU idle for 23-39 us per iteration (profiles/r04_tuning.md: the blocking
  // wait's wake-up); instead a one-lane kernel posts {value, sequence number} into page-locked host memory the device can write
  // (system-scope stores: value first, then the number, release) and the host spins on the number -- the stream stays in order, so
  // everything issued before has completed when the number arrives.  Option host_poll = 0 restores the copy + synchronize.
  struct Mail { double value; unsigned long long seq; };
  Mail *h_mail = nullptr, *d_mail = nullptr;
  unsigned long long mail_seq = 0;
  void need_mail()
  {
    if (h_mail) return;
    HIP_CHECK(hipHostMalloc((void **)&h_mail, 64, hipHostMallocMapped));
    h_mail->value = 0.0; h_mail->seq = 0;
    HIP_CHECK(hipHostGetDevicePointer((void **)&d_mail, h_mail, 0));
  }
  // finish_reduction(..., post = true) posts the scalar with the launch that reduces it: `posted` != 0 tells fetch_scalar to wait for that number
  unsigned long long posted = 0;
  double fetch_scalar(int slot)
  {
    if (opt_int("GMG_HOST_POLL", 1)) {
      need_mail();
      unsigned long long want = posted;
      posted = 0;
      if (!want) {
        want = ++mail_seq;
        hipLaunchKernelGGL(post_scalar_kernel, dim3(1), dim3(1), 0, stream, d_scalars + slot, &d_mail->value, &d_mail->seq, want);
        HIP_CHECK(hipGetLastError());
      }
      volatile unsigned long long *seq = &h_mail->seq;
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      while (__atomic_load_n(seq, __ATOMIC_ACQUIRE) < want) {       // (the numbers only grow: a later post also releases an earlier wait)
        if ((++spins & 0x3ff) == 0) {
          // a launch that failed would never post: look at the stream every ~1000 polls, give up on polling after 2 s
          const hipError_t q = hipStreamQuery(stream);
          if (q != hipSuccess && q != hipErrorNotReady) HIP_CHECK(q);
          if (q == hipSuccess && __atomic_load_n(seq, __ATOMIC_ACQUIRE) < want &&
              std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0)
            throw GmgError{GMG_ERR_HIP, "the device never posted the residual norm (host-mapped memory not visible?): set option host_poll = 0"};
        }
      }
      check_persistent();
      return h_mail->value;
    }
    posted = 0;
    HIP_CHECK(hipMemcpyAsync(h_scalars + slot, d_scalars + slot, sizeof(double), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    check_persistent();
    return h_scalars[slot];
  }
  double dot(int64_t n, const double *a, const double *b)
  {
    dot_async(n, a, b, 0, false, true);
    return fetch_scalar(0);
  }
  double norm(int64_t n, const double *a)
  {
    dot_async(n, a, a, 0, true, true);
    return fetch_scalar(0);
  }

  // ---- smoother ---------------------------------------------------------------
  void build_patch(Level &L, Smoother &S, bool blocks_only = false);
  void build_patch_operator(Level &L, Smoother &S);
  void free_pattern(DevCSR &M);
  void patch_precond(Level &L, Smoother &S, const double *r, double omega, bool relax, double *dx, double *x);

  // One launch for the whole pass on small single-GPU levels in the shared-offset pattern form (see sells_smooth_kernel).
  // s_0 is in L.sbuf[0].  Returns false when the level does not qualify (the caller then runs sweep by sweep).
  // dry = true: only answer whether the pass WOULD run as one launch (nothing is allocated or launched)
  bool smooth_persistent(int l, const Smoother &S, double *x, const double *r_in, double *r_out, bool x_zero, int niter, bool dry = false)
  {
    Level &L = lev[l];
    const DevCSR &M = L.A;
    // replicated levels have no halo; levels in the overlapping layout run the sweeps between two exchanges like a single-GPU level
    if (!persist || (comm.nranks > 1 && L.halo.present && !L.halo.ovl) || niter < 2) return false;
    if (!(M.sell && M.pat && M.pat_shared && !M.pat_coded && M.pat_k == 3 && M.pat_nruns % 3 == 0)) return false;
    const int nu = M.pat_k * M.pat_nruns;
    const size_t lds = (size_t)M.pat_np * nu * 16 + (size_t)M.pat_np * 8 + 16;
    if (lds > 64 * 1024) return false;
    if (n_cus <= 0) {
      hipDeviceProp_t prop;
      HIP_CHECK(hipGetDeviceProperties(&prop, device));
      n_cus = std::max(1, prop.multiProcessorCount);
    }
    const int rows = 65 - M.pat_k;
    const int nsl = (int)((M.nrows + rows - 1) / rows);
    const int cap = persist_max_slices > 0 ? persist_max_slices : n_cus * 16 * 2;
    if (nsl > cap || nsl > n_cus * 16 * 2 || M.nrows >= (int64_t)1 << 30) return false;
    // geometry: as many workgroups as CUs allow (latency-bound: spread the waves), one or two slices per wave
    int wpb = std::max(1, std::min(16, persist_wpb_min));
    while (wpb < 16 && (nsl + wpb - 1) / wpb > n_cus) wpb *= 2;
    int ns = (nsl + wpb - 1) / wpb > n_cus ? 2 : 1;
    const int64_t reach = std::max<int64_t>(-(int64_t)M.pat_minoff, (int64_t)M.pat_maxoff + 2);
    int halo = (int)((reach + (int64_t)wpb * ns * rows - 1) / ((int64_t)wpb * ns * rows));
    while (2 * halo + 1 > 64 && wpb < 16) { wpb *= 2; halo = (int)((reach + (int64_t)wpb * ns * rows - 1) / ((int64_t)wpb * ns * rows)); }
    if (2 * halo + 1 > 64) return false;
    const int nwg = (nsl + wpb * ns - 1) / (wpb * ns);
    if (nwg > n_cus) return false;
    if (dry) return true;
    if (!h_perr) {
      HIP_CHECK(hipHostMalloc((void **)&h_perr, 64, hipHostMallocMapped));
      *h_perr = 0;
      HIP_CHECK(hipHostGetDevicePointer((void **)&d_perr, h_perr, 0));
    }
    if (!d_perr_dev) {                                       // (released with the other device arrays by free_all)
      d_perr_dev = dalloc<uint32_t>(16);
      HIP_CHECK(hipMemsetAsync(d_perr_dev, 0, 64, stream));
    }
    if (!L.pflags || L.pf_nwg < nwg) {
      if (L.pflags) { HIP_CHECK(hipStreamSynchronize(stream)); release(L.pflags, (size_t)L.pf_nwg * 16); }
      L.pflags = dalloc<uint32_t>((size_t)nwg * 16);
      HIP_CHECK(hipMemsetAsync(L.pflags, 0, sizeof(uint32_t) * (size_t)nwg * 16, stream));
      L.pf_nwg = nwg; L.pf_epoch = 1;
    }
    if (!h_perr) {
      HIP_CHECK(hipHostMalloc((void **)&h_perr, 64, hipHostMallocMapped));
      *h_perr = 0;
      HIP_CHECK(hipHostGetDevicePointer((void **)&d_perr, h_perr, 0));
    }
    if (!d_perr_dev) {                                       // (released with the other device arrays by free_all)
      d_perr_dev = dalloc<uint32_t>(16);
      HIP_CHECK(hipMemsetAsync(d_perr_dev, 0, 64, stream));
    }
    SellSmoothArgs a;
    std::memset(&a, 0, sizeof(a));
    a.rowpid = M.rowpid; a.tab = M.ptab; a.tab8 = M.ptab8; a.run_off = M.prun; a.np = M.pat_np; a.nruns = M.pat_nruns;
    a.nrows = M.nrows; a.ncols = M.ncols; a.nslices = nsl;
    a.pdinv = pat_dinv ? M.pdinv : nullptr; a.dinv = L.dinv; a.omega = S.omega;
    a.niter = niter; a.x_zero = x_zero ? 1 : 0;
    a.r_in = r_in; a.r_out = r_out; a.x = x; a.s_a = L.sbuf[0]; a.s_b = L.sbuf[1];
    a.flags = L.pflags; a.epoch = L.pf_epoch; a.err = d_perr; a.err_dev = d_perr_dev; a.halo_wg = halo; a.fenced = persist_fenced;
    L.pf_epoch += (uint32_t)niter;
    const dim3 g(nwg), b(64 * wpb);
    const bool td = a.pdinv != nullptr;
    const bool mk = pat_strict || !M.ptab8;
    const bool prof = (l == prof_level) && prof_used + 2 <= prof_ev.size();       // every pass: there are niter times fewer of them
    if (prof) HIP_CHECK(hipEventRecord(prof_ev[prof_used], stream));
#define GMG_SMOOTH_LAUNCH(NSV, TDV)                                                                            \
    do {                                                                                                         \
      if (mk) hipLaunchKernelGGL((sells_smooth_kernel<NSV, TDV, true>), g, b, lds, stream, a);                   \
      else hipLaunchKernelGGL((sells_smooth_kernel<NSV, TDV, false>), g, b, lds, stream, a);                     \
    } while (0)
    if (ns == 2) { if (td) GMG_SMOOTH_LAUNCH(2, true); else GMG_SMOOTH_LAUNCH(2, false); }
    else { if (td) GMG_SMOOTH_LAUNCH(1, true); else GMG_SMOOTH_LAUNCH(1, false); }
#undef GMG_SMOOTH_LAUNCH
    HIP_CHECK(hipGetLastError());
    if (prof) {
      HIP_CHECK(hipEventRecord(prof_ev[prof_used + 1], stream));
      prof_w[prof_used / 2] = niter;
      prof_xm[prof_used / 2] = 0;
      prof_used += 2;
    }
    return true;
  }
  // Runs a whole solve; if a one-launch smoothing pass timed out inside it (its workgroups were not co-resident: the GPU is shared
  // with other persistent kernels), the handle switches to per-sweep launches, the initial guess is restored and the solve runs
  // again instead of failing the call.  `x_user` is only overwritten in place for single-GPU device vectors: saved first, unless
  // the body never reads it (x_output_only: gmg_apply as a preconditioner -- the path a host-language Krylov loop calls once per
  // iteration).  Several ranks: a timed-out pass writes nothing, so the solve simply carries on with the values it had, NO rank
  // leaves the collective sequence in the middle (fetch_scalar does not throw), and at the end of the body every rank learns
  // through one all-reduce whether ANY rank tripped: either all of them run the solve again or none does.
  template <typename F>
  void with_persist_retry(double *x_user, int64_t n, int memspace, bool x_output_only, F &&body)
  {
    const bool armed = persist != 0;
    const bool multi = comm.nranks > 1;
    const bool inplace = memspace == GMG_MEM_DEVICE && !multi && !x_output_only;
    if (armed && inplace) copy(scratch_vec(6, lev[0].nvec), x_user, n);
    persist_defer_throw = armed && multi;
    bool again = false, forced = false;
    struct Undefer { gmg_solver &S; ~Undefer() { S.defer_out = false; S.pending_out.clear(); } } undefer{*this};
    defer_out = armed;
    pending_out.clear();
    try {
      if (armed && opt_int("GMG_PERSIST_FORCE_TIMEOUT", 0) > 0) {   // test hook: behave as if a pass timed out in this solve
        options["GMG_PERSIST_FORCE_TIMEOUT"] = opt_int("GMG_PERSIST_FORCE_TIMEOUT", 0) - 1;
        forced = true;
        force_pending = !multi;                              // one rank: the next check_persistent() (first scalar fetch of the solve) trips
      }
      body();
      if (!persist_defer_throw) {
        // out_vec is deferred while armed, so nothing above guarantees that the stream has drained: a one-launch pass issued after
        // the body's last scalar fetch must have posted its time-out word before the solve is declared good
        if (armed) HIP_CHECK(hipStreamSynchronize(stream));
        check_persistent();
      }
    } catch (const GmgError &) {
      persist_defer_throw = false;
      if (!(armed && persist_tripped)) throw;
      again = true;
    }
    if (armed && multi) {
      persist_defer_throw = false;
      HIP_CHECK(hipStreamSynchronize(stream));
      double flag = ((h_perr && *h_perr) || forced) ? 1.0 : 0.0;
      host_allreduce_sum(&flag);                             // joint decision (not counted in gmg_get_comm_stats: control traffic)
      if (flag > 0.0) {
        if (h_perr) *h_perr = 0;
        persist = 0;
        again = true;
      }
    }
    if (again) {
      persist_tripped = false;
      ++persist_retries;
      HIP_CHECK(hipStreamSynchronize(stream));
      if (inplace) copy(x_user, scratch_vec(6, lev[0].nvec), n);
      pending_out.clear();
      body();
    }
    defer_out = false;
    const std::vector<PendingOut> todo = pending_out;
    pending_out.clear();
    for (const PendingOut &o : todo) out_vec(o.user, o.dev, o.n, o.memspace);
  }
  bool persist_tripped = false;
  bool persist_defer_throw = false;   // several ranks: the time-out is acted on jointly at the end of the solve
  int64_t persist_retries = 0;
  // one double summed over the ranks through the host (control decisions; solves use the in-stream reductions)
  void host_allreduce_sum(double *v)
  {
    if (comm.nranks <= 1) return;
    if (comm.kind == COMM_RCCL) {
      double *slot = d_scalars + kScalarSlots - 1;
      HIP_CHECK(hipMemcpyAsync(slot, v, sizeof(double), hipMemcpyHostToDevice, stream));
      const int rc = comm.api.AllReduce(slot, slot, 1, kNcclDouble, kNcclSum, comm.comm, stream);
      REQUIRE(rc == 0, GMG_ERR_COMM, std::string("ncclAllReduce: ") + comm.api.GetErrorString(rc));
      HIP_CHECK(hipMemcpyAsync(v, slot, sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    } else
      comm.rfn(comm.ctx, v, 1);
  }
  bool force_pending = false;
  void check_persistent()
  {
    if (!((h_perr && *h_perr) || force_pending)) return;
    if (persist_defer_throw) return;                         // multi-rank solve in flight: decided jointly in with_persist_retry
    force_pending = false;
    if (h_perr) *h_perr = 0;
    persist = 0;                                             // later solves of this handle sweep launch by launch
    persist_tripped = true;
    throw GmgError{GMG_ERR_STATE, "one-launch smoothing pass: a neighbour wait timed out (its workgroups were not all resident -- is the GPU shared with "
                                  "other persistent kernels?); the results of this call are invalid, the handle now runs with GMG_PERSIST=0"};
  }

  // solve!(x,ns::RichardsonSmootherNumericalSetup,r), RichardsonSmoothers.jl:84-98.
  // r_in may be a caller-owned read-only vector; returns the buffer holding the
  // updated residual (one of L.rbuf[]).
  double *smooth(int l, Smoother &S, double *x, const double *r_in, bool x_zero)
  {
    Level &L = lev[l];
    const int64_t n = L.n;
    const bool r_internal = (r_in == L.rbuf[0] || r_in == L.rbuf[1]);
    if (S.niter <= 0) {
      if (x_zero) zero(x, n);
      if (!r_internal) { copy(L.rbuf[0], r_in, n); return L.rbuf[0]; }
      return const_cast<double *>(r_in);
    }
    if (S.kind == SM_JACOBI && one_gather()) {
      // s_0 = w*Dinv*r_in ; then each sweep: x += s_k ; r -= A s_k (in place after the
      // first sweep) ; s_{k+1} = w*Dinv*r
      // Overlapping layout (gmg_set_partition_overlap): the pass runs in blocks of `depth` sweeps -- consistent!(r) on all ghost
      // layers, s_0 from it on every local row, then `depth` sweeps over every local row with no communication (ghost layer j
      // stays exact for depth - j sweeps; the owned rows are exact throughout).  Same arithmetic per row as on one GPU.
      const bool ovl = comm.nranks > 1 && L.halo.present && L.halo.ovl;
      const double *cur = r_in;
      double *out = r_internal ? const_cast<double *>(r_in) : L.rbuf[0];
      const int block = ovl ? std::max(1, std::min(L.halo.depth, S.niter)) : S.niter;
      // shared-offset pattern kernel: x is updated every second sweep with both increments,
      // x = (x + s_{k-1}) + s_k (the same two roundings), which saves one read+write of x per pair
      const bool rsw = pat_defer && rsweep_level(L);
      const bool sell64 = L.A.sell && !L.A.pat && !L.A.comp_idx && !L.A.vdict && !L.A.opat;
      const bool defer = (pat_defer && L.A.pat_shared) ||
                         (sell_defer && !(comm.nranks > 1 && L.split) && ((sell64 && sell_un >= 6 && sell_un < 9) || (L.A.sell && L.A.opat && sell_un < 27)));
      for (int done = 0; done < S.niter; done += block) {
        const int nb = std::min(block, S.niter - done);
        const bool first = done == 0;
        if (ovl) {
          if (cur != out) { copy(out, cur, n); cur = out; }   // (levels >= 1 smooth in place: no copy)
          exchange(l, out);
        }
        const bool xz0 = x_zero && first;
        const bool one_launch = smooth_persistent(l, S, x, cur, out, xz0, nb, true);
        if (rsw && !one_launch) {
          // sweeps that gather r itself (uniform 1/diag): no s vector, no scaled-Jacobi launch; r ping-pongs between the level's
          // two residual buffers (the gathers read r_k while the rows write r_{k+1}), the result is wherever the last sweep wrote
          L.s0_ready = false;
          const double *prev = nullptr;
          if (comm.nranks > 1 && L.halo.present && !L.halo.ovl && cur != L.rbuf[0] && cur != L.rbuf[1]) {
            copy(L.rbuf[0], cur, n);                           // the exchange writes the ghost entries of r_k: never into a caller's vector
            cur = L.rbuf[0];
          }
          if (pat_fuse2) ensure_z2(L.A);
          const bool fuse2 = fuse2_level(L);
          for (int it = 0; it < nb; ++it) {
            double *next = (cur == L.rbuf[0]) ? L.rbuf[1] : L.rbuf[0];
            if (fuse2 && (it & 1) == 0 && it + 1 < nb) {
              // sweeps it (x untouched) and it + 1 (x += both increments) in ONE launch: r_k -> r_{k+2}, r_{k+1} never leaves the CUs
              const bool prof = (l == prof_level) && (prof_seq++ % (uint64_t)prof_stride == 0) && prof_used + 2 <= prof_ev.size();
              if (prof) HIP_CHECK(hipEventRecord(prof_ev[prof_used], stream));
              launch_z2sweep(L.A, cur, next, x, xz0 && it == 0, S.omega);
              if (prof) {
                HIP_CHECK(hipEventRecord(prof_ev[prof_used + 1], stream));
                prof_w[prof_used / 2] = 2;
                prof_xm[prof_used / 2] = 2;
                prof_used += 2;
              }
              prev = nullptr;
              cur = next;
              ++it;
              continue;
            }
            int xmode = 0;
            bool xz = xz0 && it == 0;
            if ((it & 1) == 0 && it + 1 < nb) xmode = 1;
            else if (it & 1) { xmode = 2; xz = xz0 && it == 1; }
            const bool fp = can_fuse_pack(l);
            rsweep(l, S, x, cur, next, prev, xz, xmode, fp && it > 0, fp && it + 1 < nb);
            prev = cur;
            cur = next;
          }
          out = const_cast<double *>(cur);
          continue;
        }
        if (!(first && !ovl && L.s0_ready && L.s0_src == r_in && L.s0_omega == S.omega)) {
          hipLaunchKernelGGL(scaled_jacobi_kernel, dim3(grid_for(n)), dim3(256), 0, stream, n, S.omega, L.dinv, cur, L.sbuf[0]);
          HIP_CHECK(hipGetLastError());
        }
        L.s0_ready = false;
        if (one_launch && smooth_persistent(l, S, x, cur, out, xz0, nb)) { cur = out; continue; }
        for (int it = 0; it < nb; ++it) {
          int xmode = 0;
          bool xz = xz0 && it == 0;
          if (defer) {
            if ((it & 1) == 0 && it + 1 < nb) xmode = 1;
            else if (it & 1) { xmode = 2; xz = xz0 && it == 1; }
          }
          const bool fp = can_fuse_pack(l);
          sweep(l, S, x, cur, out, xz, L.sbuf[it & 1], L.sbuf[(it + 1) & 1], xmode, fp && it > 0, fp && it + 1 < nb);
          cur = out;
        }
      }
      return out;
    }
    if (S.kind == SM_JACOBI) {
      const double *cur = r_in;
      double *out = nullptr;
      for (int it = 0; it < S.niter; ++it) {              // :90 while iter <= niter
        out = (cur == L.rbuf[0]) ? L.rbuf[1] : L.rbuf[0];
        sweep(l, S, x, cur, out, x_zero && it == 0);       // :91-95 fused
        cur = out;
      }
      return out;
    }
    // patch smoother: r updated in place
    double *r = nullptr;
    if (r_internal) r = const_cast<double *>(r_in);
    else { r = L.rbuf[0]; copy(r, r_in, n); }
    if (x_zero) zero(x, n);
    // Overlapping layout (gmg_set_partition_overlap): consistent!(r) once per block of `depth` sweeps; inside a block every local row
    // is swept with the patches of the local box -- no assemble!, no exchange of dx (ghost layers are recomputed redundantly and stay
    // exact for depth - j blocks of 3 order - 2 node layers, see partition._OverlapGeom).  Same arithmetic per owned row as on one GPU.
    const bool ovl_patch = comm.nranks > 1 && L.halo.present && L.halo.ovl;
    const int pblock = ovl_patch ? std::max(1, std::min(L.halo.depth, S.niter)) : 1;
    for (int it = 0; it < S.niter; ++it) {
      if (!ovl_patch || it % pblock == 0)
      exchange(l, r);                                      // consistent!(b) PatchSolvers.jl:231 (no-op on one GPU)
      patch_precond(L, S, r, S.omega, true, L.dx, x);      // :91-93
      // profiled level: HIP events around the operator mat-vec of every prof_stride-th patch sweep (r -= A dx, the wide-row kernel)
      const bool prof = (l == prof_level) && (prof_seq++ % (uint64_t)prof_stride == 0) && prof_used + 2 <= prof_ev.size();
      if (prof) HIP_CHECK(hipEventRecord(prof_ev[prof_used], stream));
      if (ovl_patch) spmv_sub(L.A, L.dx, r);               // every local row, no exchange
      else
      apply_A_sub(l, L.dx, r);                             // :94-95 (exchanges dx itself: consistent!(x), PatchSolvers.jl:256)
      if (prof) {
        HIP_CHECK(hipEventRecord(prof_ev[prof_used + 1], stream));
        prof_w[prof_used / 2] = 1;
        prof_xm[prof_used / 2] = 0;
        prof_used += 2;
        prof_patch = true;
      }
    }
    return r;
  }

  // rH = R rh across the distributed -> replicated boundary: own coarse rows, scattered into the
  // global vector by global id, summed over ranks (disjoint contributions).
  void restrict_replicate(int l, const double *r, double *rH_global)
  {
    Level &L = lev[l];
    const int64_t nrows = r ? L.R.nrows : 0, ng = lev[l + 1].n;   // (r == nullptr: a rank that holds no part of level l, shadow_cycle)
    if (r) restrict_into(l, const_cast<double *>(r), d_rep_tmp);
    zero(rH_global, ng);
    if (nrows > 0) {
      hipLaunchKernelGGL(scatter_gid_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, stream, nrows, d_rep_gid, d_rep_tmp, rH_global);
      HIP_CHECK(hipGetLastError());
    }
    ++n_allreduces;
    if (comm.kind == COMM_RCCL) {
      const int rc = comm.api.AllReduce(rH_global, rH_global, (size_t)ng, kNcclDouble, kNcclSum, comm.comm, stream);
      REQUIRE(rc == 0, GMG_ERR_COMM, std::string("ncclAllReduce(restriction): ") + comm.api.GetErrorString(rc));
    } else {
      HIP_CHECK(hipMemcpyAsync(h_rep_full, rH_global, sizeof(double) * (size_t)ng, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      comm.rfn(comm.ctx, h_rep_full, (int)ng);
      HIP_CHECK(hipMemcpyAsync(rH_global, h_rep_full, sizeof(double) * (size_t)ng, hipMemcpyHostToDevice, stream));
    }
  }

  // x = Ainv r (row-major dense inverse, one wave per row)
  void dense_solve(const double *Ainv, int n, const double *r, double *x)
  {
    const int waves_per_block = kBlock / 64;
    const int grid = (n + waves_per_block - 1) / waves_per_block;
    hipLaunchKernelGGL(dense_gemv_kernel, dim3(std::max(grid, 1)), dim3(kBlock), 0, stream, n, Ainv, r, x);
    HIP_CHECK(hipGetLastError());
  }
  // solve!(xh, ns.coarsest_solver_cache, rh), GMGLinearSolvers.jl:474.  x arrives as fill!(dxH,0) (:487).
  // (in distributed runs the coarsest level is replicated: same kernels, global matrix, local reductions)
  void coarse_solve(const double *r, double *x)
  {
    Level &L = lev[nlev - 1];
    const int64_t n = L.n;
    if (coarse_eff == GMG_COARSE_DENSE_INVERSE) { dense_solve(d_Ainv, (int)n, r, x); return; }
    if (coarse_eff == GMG_COARSE_CG_JACOBI) {
      // CGSolver(JacobiLinearSolver(); maxiter, atol, rtol) on A_L, initial guess 0
      zero(x, n);
      KrylovOps ops;
      ops.resid = [&](double *xx, const double *b, double *rr) { spmv_resid(L.A, xx, b, rr); };
      ops.apply = [&](double *xx, double *yy) { spmv_set(L.A, xx, yy); };
      ops.precond = [&, n](double *z, const double *rr, double) {
        hipLaunchKernelGGL(jacobi_apply_kernel, dim3(grid_for(n)), dim3(256), 0, stream, n, L.dinv, rr, z);
        HIP_CHECK(hipGetLastError());
      };
      if (coarse_auto) coarse_log.configure(10000, 0.0, 1e-10);
      else coarse_log.configure(coarse_maxiter, coarse_atol, coarse_rtol);
      const bool saved = reduce_local;
      reduce_local = true;
      try { coarse_last = cg_core(*this, n, r, x, cc_w, cc_p, cc_z, cc_r, ops, false, coarse_log); }
      catch (...) { reduce_local = saved; throw; }
      reduce_local = saved;
      return;
    }
    // host callback: the caller's own LinearSolver (PETSc, UMFPACK ... in the Julia host) on host vectors
    HIP_CHECK(hipMemcpyAsync(h_cr, r, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    std::memset(h_cx, 0, sizeof(double) * (size_t)n);
    const int rc = coarse_fn(coarse_ctx, n, h_cr, h_cx);
    REQUIRE(rc == 0, GMG_ERR_INVALID, "coarse-solver callback reported failure (" + std::to_string(rc) + ")");
    HIP_CHECK(hipMemcpyAsync(x, h_cx, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, stream));
  }

  // GMG_HOST_TIMELINE=1 (debug): host-side duration of every step of the cycles (is a launch blocking?), printed when the handle dies
  struct HostStep { const char *what; int lev; double t0, t1; };
  std::vector<HostStep> host_steps;
  int host_timeline = -1;
  struct StepTimer {
    gmg_solver &S; const char *what; int lev; double t0;
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    StepTimer(gmg_solver &s, const char *w, int l) : S(s), what(w), lev(l), t0(0.0) { if (S.host_timeline > 0) t0 = now(); }
    ~StepTimer() { if (S.host_timeline > 0 && S.host_steps.size() < 4096) S.host_steps.push_back({what, lev, t0, now()}); }
  };
  void dump_host_steps()
  {
    if (host_steps.empty()) return;
    const size_t first = host_steps.size() > 80 ? host_steps.size() - 80 : 0;
    for (size_t i = first; i < host_steps.size(); ++i)
      std::fprintf(stderr, "[gmg host] %10.1f us  +%7.1f  lev %d  %s\n", host_steps[i].t0 - host_steps[first].t0, host_steps[i].t1 - host_steps[i].t0, host_steps[i].lev, host_steps[i].what);
  }
  // Overlapping level: is the residual the pass of smoother S left behind still exact on every ghost entry the restriction of an owned
  // coarse row reads?  smooth() runs S.niter sweeps in blocks of `depth` (one exchange per block); the LAST block has
  // niter - depth * ((niter - 1) / depth) sweeps, each of which makes `ovl_sweep_reach` more node layers inexact.
  static bool skip_r_after(const Level &L, const Smoother &S)
  {
    if (L.ovl_layers <= 0 || L.ovl_sweep_reach <= 0 || L.ovl_r_reach <= 0 || S.niter <= 0) return false;
    const int k = std::max(1, std::min(L.halo.depth, S.niter));
    const int last = S.niter - k * ((S.niter - 1) / k);
    return L.ovl_layers - last * L.ovl_sweep_reach >= L.ovl_r_reach;
  }
  // gmg_v_cycle! / gmg_w_cycle! / gmg_f_cycle!, GMGLinearSolvers.jl:468-610
  void cycle(int l, double *x, const double *r_in, bool x_zero, int ctype)
  {
    if (host_timeline < 0) host_timeline = opt_int("GMG_HOST_TIMELINE", 0);
    Level &L = lev[l];
    if (l == nlev - 1) {                                   // :472-474
      StepTimer tm(*this, "coarse", l);
      coarse_solve(r_in, x);
      return;
    }
    Level &C = lev[l + 1];
    double *r;
    { StepTimer tm(*this, "pre-smooth", l); r = smooth(l, L.pre, x, r_in, x_zero); }         // :481
    const int passes = (ctype == GMG_V_CYCLE) ? 1 : 2;
    for (int pass = 0; pass < passes; ++pass) {
      if (pass == 1) r = smooth(l, L.post, x, r, false);   // W :531 / F :584 re-smooth
      const bool ovl_l = comm.nranks > 1 && L.halo.present && L.halo.ovl;
      if (!(ovl_l && skip_r_after(L, pass == 0 ? L.pre : L.post)) && !L.r_split) exchange(l, r);      // (a split R starts its own exchange and hides it: restrict_into)
      if (redist.present && l + 1 == sub_from) {
        // level l+1 lives on a rank subset: restrict in the glued partition, redistribute! to the subset owners, recurse there (the
        // other ranks shadow the collectives), bring the correction back to the glued own and ghost entries, prolongate from those
        const int child = (pass == 0) ? ctype : (ctype == GMG_W_CYCLE ? GMG_W_CYCLE : GMG_V_CYCLE);
        restrict_into(l, r, redist.glue_r);
        redistribute(redist.to_sub, 0, redist.glue_r, C.rbuf[0]);
        if (redist.member) cycle(l + 1, C.x, C.rbuf[0], true, child);
        else shadow_cycle(l + 1, child);
        redistribute(redist.from_sub, 2, C.x, redist.glue_x);
        spmv_addto(L.P, redist.glue_x, L.dx, x);
        apply_A_sub(l, L.dx, r, &L.post);
        continue;
      }
      if (comm.nranks > 1 && l + 1 == rep_from) restrict_replicate(l, r, C.rbuf[0]);
      else if (l + 1 < nlev - 1 && emits_s0(C, C.pre, L.R)) {
        StepTimer tm(*this, "restrict+s0", l);
        spmv_set(L.R, r, C.rbuf[0], C.sbuf[0], C.dinv, C.pre.omega);   // :484 rH = R rh (+ the child's first s)
        C.s0_ready = true; C.s0_src = C.rbuf[0]; C.s0_omega = C.pre.omega;
      } else {
      StepTimer tm(*this, "restrict", l);
      restrict_into(l, r, C.rbuf[0]);                      // :484 rH = R rh
      }
      // :487 fill!(dxH,0) is implicit: the first sweep below / the coarse solve write dxH
      const int child = (pass == 0) ? ctype : (ctype == GMG_W_CYCLE ? GMG_W_CYCLE : GMG_V_CYCLE);
      cycle(l + 1, C.x, C.rbuf[0], true, child);           // :488
      exchange(l + 1, C.x);
      if (L.has_pcorr) {
        // PatchProlongationOperator: dxh = P dxH - sum_p A_pp^-1 (A P dxH)_p  (PatchTransferOperators.jl:153-172)
        spmv_set(L.P, C.x, L.dx);
        if (L.hasG) { exchange(l, L.dx); spmv_set(L.G, L.dx, L.ptmp); }   // rhs(uH,v_i) with the caller's rhs form (consistent!(dxh) first when distributed)
        else apply_A_set(l, L.dx, L.ptmp);
        patch_precond(L, L.pcorr, L.ptmp, 1.0, false, L.pcor, nullptr);
        hipLaunchKernelGGL(prolong_correct_kernel, dim3(grid_for(L.n)), dim3(256), 0, stream, L.n, L.pcor, L.dx, x);
        HIP_CHECK(hipGetLastError());
      } else
      { StepTimer tm(*this, "prolong", l); spmv_addto(L.P, C.x, L.dx, x); }                       // :491,494 dxh = P dxH ; xh += dxh
      if (ovl_l && L.ovl_skip_dx && !L.has_pcorr) spmv_sub(L.A, L.dx, r);                            // (every local row; the pass that follows opens with consistent!(r))
      else
      { StepTimer tm(*this, "r -= A dx", l); apply_A_sub(l, L.dx, r, &L.post); }                    // :495-496 rh -= Ah dxh (+ the post-smoother's first s)
    }
    { StepTimer tm(*this, "post-smooth", l); r = smooth(l, L.post, x, r, false); }                    // :499
    L.rcur = r;
  }

  // solve!(x,ns::GMGNumericalSetup,b), GMGLinearSolvers.jl:612-645 (device pointers).
  // known_res0 >= 0: the caller already holds norm(b) (outer CG's `res`), which
  // removes the only host round trip of the preconditioner application.
  double gmg_solve_dev(double *x, const double *b, double known_res0)
  {
    Level &L0 = lev[0];
    const int64_t n = L0.n;
    const double *r_in;
    bool x_zero;
    if (mode == GMG_MODE_PRECONDITIONER) {                 // :618-620
      x_zero = true;  // fill!(x,0) folded into the first sweep
      r_in = b;       // copy!(rh,b) folded: the first sweep reads b, writes rh
    } else {                                               // :621-625
      apply_A_resid(0, x, b, L0.rbuf[0]);                  // callers pass x with ghost space when distributed
      r_in = L0.rbuf[0];
      x_zero = false;
    }
    double res = known_res0;
    if (!(res >= 0.0)) res = norm(n, r_in);                // :627
    bool done = log.init(res);                             // :628
    if (done && x_zero) zero(x, n);
    const bool single = (log.maxiter == 1);
    while (!done) {
      cycle(0, x, r_in, x_zero, cycle_type);               // :630-637
      r_in = L0.rcur;
      x_zero = false;
      if (single && known_res0 >= 0.0 && (verbose <= 0 || has_outer())) {
        // maxiter == 1: update! returns true whatever the norm is (:640); the post-cycle norm is a
        // logging-only quantity here and is not evaluated unless the solver is verbose (gmg_set_verbose):
        // residuals[1] of the GMG's own log then reads NaN (documented in INTEGRATION.md).
        log.num_iters = 1;
        if (log.residuals.size() > 1) log.residuals[1] = NAN;
        res = NAN;
        done = true;
      } else {
        res = norm(n, r_in);                               // :639
        done = log.update(res);                            // :640
      }
    }
    log_last = res;
    return res;
  }

  // Krylov preconditioner dispatch: 0 = nothing (z .= r, CGSolvers.jl:91), 1 = this GMG,
  // 2 = JacobiLinearSolver() on the finest matrix (JacobiLinearSolvers.jl:43-47)
  void krylov_precond(int kind, double *z, const double *r, double known_res0)
  {
    const int64_t n = user_n();
    if (has_outer()) {
      // finest level in the overlapping layout: r (own | ghost numbering) -> level 0's numbering, cycle, owned entries of the
      // correction back.  The ghost layers of r are filled by the exchange that opens the first smoothing block.
      REQUIRE(kind == 0 || kind == 1, GMG_ERR_UNSUPPORTED, "finest level in the overlapping layout: the preconditioner is the GMG (use_precond 0 or 1)");
      if (kind == 0) { copy(z, r, n); return; }
      Level &L0 = lev[0];
      const unsigned grid = (unsigned)std::max<int64_t>(1, (n + 255) / 256);
      hipLaunchKernelGGL(halo_unpack_kernel, dim3(grid), dim3(256), 0, stream, n, d_own2loc, r, L0.rbuf[0]);
      HIP_CHECK(hipGetLastError());
      gmg_solve_dev(L0.x, L0.rbuf[0], known_res0 >= 0.0 ? known_res0 : norm(n, r));   // (the GMG's own norm would count ghost entries)
      hipLaunchKernelGGL(halo_pack_kernel, dim3(grid), dim3(256), 0, stream, n, d_own2loc, L0.x, z);
      HIP_CHECK(hipGetLastError());
      return;
    }
    if (kind == 1) gmg_solve_dev(z, r, known_res0);
    else if (kind == 3) {
      // LinearSolverFromSmoother(pre_smoothers[1]): x = 0 ; r = copy(b) ; solve!(x,smoother,r)  (LinearSolverFromSmoothers.jl:44-50)
      smooth(0, lev[0].pre, z, r, true);
    } else if (kind == 2) {
      hipLaunchKernelGGL(jacobi_apply_kernel, dim3(grid_for(n)), dim3(256), 0, stream, n, lev[0].dinv, r, z);
      HIP_CHECK(hipGetLastError());
    } else copy(z, r, n);
  }

  // ---- staging for host-memory callers -----------------------------------------
  // The Julia binding hands over Vector{Float64} (GMG_MEM_HOST): b and the initial guess go up, x comes back -- 24 N bytes per solve
  // over PCIe.  Three things keep that near the link rate: (i) caller arrays registered once (gmg_host_register: page-locked and
  // mapped, the pattern of ext/GridapPETScExt/PETScCaches.jl:23-36, which pins the exact x / b objects) move by DMA straight from /
  // into the caller's pages; (ii) unregistered (pageable) arrays are pipelined through two page-locked chunks of the library's own
  // (the memcpy of chunk k+1 overlaps the DMA of chunk k) instead of the runtime's synchronous pageable path; (iii) with the option
  // GMG_X0_ZERO the initial guess is not uploaded at all (hipMemsetAsync) -- the reference CG reads x (CGSolvers.jl:79), so that is
  // an opt-in for callers that know x0 = 0.
  struct HostReg { const char *base; size_t bytes; bool ours; };
  std::vector<HostReg> host_regs;
  bool host_registered(const void *p, size_t bytes) const
  {
    const char *c = static_cast<const char *>(p);
    for (const HostReg &r : host_regs)
      if (c >= r.base && c + bytes <= r.base + r.bytes) return true;
    return false;
  }
  char *h_chunk[2] = {nullptr, nullptr};     // page-locked staging chunks (pageable callers)
  hipEvent_t ev_chunk[2] = {nullptr, nullptr};
  size_t chunk_bytes = 0;
  int64_t host_bytes_up = 0, host_bytes_down = 0;   // bytes moved for host-memory callers since gmg_create (gmg_get_host_io_stats)
  void ensure_chunks()
  {
    const size_t want = (size_t)std::max(1 << 16, opt_int("GMG_HOST_CHUNK_BYTES", 4 << 20));
    if (h_chunk[0] && chunk_bytes == want) return;
    for (int i = 0; i < 2; ++i) {
      if (h_chunk[i]) (void)hipHostFree(h_chunk[i]);
      HIP_CHECK(hipHostMalloc((void **)&h_chunk[i], want, hipHostMallocDefault));
      if (!ev_chunk[i]) HIP_CHECK(hipEventCreateWithFlags(&ev_chunk[i], hipEventDisableTiming));
    }
    chunk_bytes = want;
  }
  void h2d(double *dev, const double *host, int64_t n)
  {
    const size_t bytes = sizeof(double) * (size_t)n;
    host_bytes_up += (int64_t)bytes;
    if (host_registered(host, bytes) || bytes <= (1 << 16)) {
      HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream));
      return;
    }
    ensure_chunks();
    const char *src = reinterpret_cast<const char *>(host);
    char *dst = reinterpret_cast<char *>(dev);
    int k = 0;
    for (size_t off = 0; off < bytes; off += chunk_bytes, k ^= 1) {
      const size_t len = std::min(chunk_bytes, bytes - off);
      HIP_CHECK(hipEventSynchronize(ev_chunk[k]));           // the DMA that last read this chunk has finished
      std::memcpy(h_chunk[k], src + off, len);
      HIP_CHECK(hipMemcpyAsync(dst + off, h_chunk[k], len, hipMemcpyHostToDevice, stream));
      HIP_CHECK(hipEventRecord(ev_chunk[k], stream));
    }
  }
  // blocks until the data is in the caller's array
  void d2h(double *host, const double *dev, int64_t n)
  {
    const size_t bytes = sizeof(double) * (size_t)n;
    host_bytes_down += (int64_t)bytes;
    if (host_registered(host, bytes) || bytes <= (1 << 16)) {
      HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      return;
    }
    ensure_chunks();
    char *dst = reinterpret_cast<char *>(host);
    const char *src = reinterpret_cast<const char *>(dev);
    const size_t nchunks = (bytes + chunk_bytes - 1) / chunk_bytes;
    for (size_t c = 0; c < nchunks + 1; ++c) {               // DMA of chunk c is in flight while chunk c-1 is copied out
      if (c < nchunks) {
        const size_t off = c * chunk_bytes, len = std::min(chunk_bytes, bytes - off);
        // (chunk buffer c & 1 was drained when chunk c-2 was copied out, one iteration ago)
        HIP_CHECK(hipMemcpyAsync(h_chunk[c & 1], src + off, len, hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipEventRecord(ev_chunk[c & 1], stream));
      }
      if (c >= 1) {
        const size_t off = (c - 1) * chunk_bytes, len = std::min(chunk_bytes, bytes - off);
        HIP_CHECK(hipEventSynchronize(ev_chunk[(c - 1) & 1]));
        std::memcpy(dst + off, h_chunk[(c - 1) & 1], len);
      }
    }
  }
  const double *in_vec(const double *p, int64_t n, int memspace, double *stage)
  {
    if (memspace == GMG_MEM_DEVICE) return p;
    h2d(stage, p, n);
    return stage;
  }
  // the initial guess of a Krylov / :solver call -> dx (a no-op for single-GPU device callers, whose x is used in place)
  void in_guess(double *dx, const double *x_user, int64_t n, int memspace)
  {
    const bool host = memspace == GMG_MEM_HOST;
    if (!host && dx == x_user) {
      if (opt_int("GMG_X0_ZERO", 0)) zero(dx, n);
      return;
    }
    if (opt_int("GMG_X0_ZERO", 0)) { zero(dx, n); return; }
    if (host) h2d(dx, x_user, n);
    else HIP_CHECK(hipMemcpyAsync(dx, x_user, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream));
  }
  // Inside with_persist_retry results are handed to the caller only once the solve is known to stand (single rank: no time-out seen;
  // several ranks: the joint decision has been taken) -- until then the caller's x still holds the initial guess a re-run starts from.
  struct PendingOut { double *user; const double *dev; int64_t n; int memspace; };
  std::vector<PendingOut> pending_out;
  bool defer_out = false;
  void out_vec(double *user, const double *dev, int64_t n, int memspace)
  {
    if (defer_out) { pending_out.push_back({user, dev, n, memspace}); return; }
    if (memspace == GMG_MEM_DEVICE) {
      if (user != dev) copy(user, dev, n);
      HIP_CHECK(hipStreamSynchronize(stream));
    } else
      d2h(user, dev, n);
  }
  double *scratch_vec(size_t i, int64_t n)
  {
    while (st_extra.size() <= i) st_extra.push_back(nullptr);
    if (!st_extra[i]) st_extra[i] = dvec(n);
    return st_extra[i];
  }

  void read_tuning()
  {
    xcd_remap = opt_int("GMG_XCD_REMAP", 1);   // XCD-contiguous row ranges: each L2 sees 1/8 of the gathered vector (pattern kernels: -6 % per solve)
    lanes_override = opt_int("GMG_LANES_LOG2", -1);
    one_gather_sweep = opt_int("GMG_ONE_GATHER", 1);
    use_sell = opt_int("GMG_SELL", 1);
    use_idx16 = opt_int("GMG_IDX16", 1);
    use_vdict = opt_int("GMG_VDICT", 1);
    sell_un = opt_int("GMG_SELL_UN", 6);
    sell_block = std::min(256, opt_int("GMG_SELL_BLOCK", 0)) / 64 * 64;
    sell_maxpad = opt_num("GMG_SELL_MAXPAD", 1.25);
    nt_loads = opt_int("GMG_NT", 1);
    nt_rowwise = opt_int("GMG_NT_ROWWISE", 1);
    xcd_remap_big = opt_int("GMG_XCD_REMAP_BIG", -1);
    big_rows = (int64_t)opt_int("GMG_BIG_ROWS", 4000000);
    sell_defer = opt_int("GMG_SELL_DEFER", 1);
    use_pattern = opt_int("GMG_PATTERN", 1);
    pat_un = opt_int("GMG_PAT_UN", 9);
    pat_wgs = std::max(1, opt_int("GMG_PAT_WGS", 2048));
    pat_shared = opt_int("GMG_PAT_SHARED", 1);
    halo_fuse_pack = opt_int("GMG_HALO_FUSE_PACK", 1);
    prof_stride = std::max(1, opt_int("GMG_PROF_STRIDE", 7));
    pat_defer = opt_int("GMG_PAT_DEFER", 1);
    pat_rsweep = opt_int("GMG_PAT_RSWEEP", 1);
    if (n_cus <= 0) {
      int v = 0;
      HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device));
      n_cus = std::max(1, v);
    }
    pat_bcast = opt_int("GMG_PAT_BCAST", 1);
    red_fused = opt_int("GMG_RED_FUSED", 1);
    pat_r2 = opt_int("GMG_PAT_R2", 1);
    pat_r2_wgs = opt_int("GMG_PAT_R2_WGS", 0);
    pat_r2_occ = opt_int("GMG_PAT_R2_OCC", 2);
    pat_zwalk = opt_int("GMG_PAT_ZWALK", 1);
    pat_zwalk_T = std::max(1, opt_int("GMG_PAT_ZWALK_T", 12));
    pat_zwalk_mv = opt_int("GMG_PAT_ZWALK_MV", 1);
    pat_zwalk_wide = opt_int("GMG_PAT_ZWALK_WIDE", 1);
    pat_zwalk_wide_rows = opt_int("GMG_PAT_ZWALK_WIDE_ROWS", 1000000);
    pat_zwalk_rows = opt_int("GMG_PAT_ZWALK_ROWS", 9000000);
    pat_fuse2 = opt_int("GMG_PAT_FUSE2", 0);
    pat_fuse2_w = opt_int("GMG_PAT_FUSE2_W", 0);
    pat_fuse2_t = opt_int("GMG_PAT_FUSE2_T", 0);
    pat_fuse2_rows = opt_int("GMG_PAT_FUSE2_ROWS", 1000000);
    pat_box = opt_int("GMG_PAT_BOX", 0);
    pat_box_t = opt_int("GMG_PAT_BOX_T", 0);
    pat_box_min_rows = opt_int("GMG_PAT_BOX_MIN_ROWS", 1000000);
    pat_box_max_rows = opt_int("GMG_PAT_BOX_MAX_ROWS", 9000000);
    persist_wpb_min = opt_int("GMG_PERSIST_WPB", 1);
    pat_r2mv = opt_int("GMG_PAT_R2MV", 1);
    pat_pair_p = opt_int("GMG_PAT_PAIR_P", 1);
    pat_r2mv_dot = opt_int("GMG_PAT_R2MV_DOT", 1);
    pat_r2mv_min = opt_int("GMG_PAT_R2MV_MIN", 100000);
    pat_fma = opt_int("GMG_PAT_FMA", 0);
    persist = opt_int("GMG_PERSIST", 1);
    // several ranks on ONE device (host-staged transport: the test / debugging set-up): the one-launch passes of different processes
    // could keep each other from becoming fully resident, so they are off unless asked for
    if (comm.kind == COMM_HOST && !opt_int("GMG_PERSIST_SHARED", 0)) persist = 0;
    // release / acquire on the progress words costs 2.4 us per sweep (43 -> 68 us per pass of 10 on 63^3 rows, profiles/r03_mb_smooth.txt)
    // and adds nothing the explicit ordering does not already give: every datum that crosses workgroups moves with agent-scope
    // (sc1) atomics, the publishing lane stores the word after the workgroup's s_waitcnt vmcnt(0) + barrier, the polling lanes read
    // it with agent-scope loads and a barrier precedes the gathers.  Off by default, kept as a switch.
    persist_fenced = opt_int("GMG_PERSIST_FENCED", 0);
    pat_strict = opt_int("GMG_PAT_STRICT", 1);
    pat_wide = opt_int("GMG_PAT_WIDE", 1);
    pat_tile = opt_int("GMG_PAT_TILE", 0);
    pat_tile_rows = opt_int("GMG_PAT_TILE_ROWS", 3500000);
    gj_mfma = opt_int("GMG_GJ_MFMA", 1);
    persist_max_slices = opt_int("GMG_PERSIST_MAX_SLICES", 0);
    pat_coded_min_rows = opt_int("GMG_PAT_CODED_MIN_ROWS", 500000);
    pat_emit = opt_int("GMG_PAT_EMIT", 1);
    pat_small_wpb = std::min(4, std::max(1, opt_int("GMG_PAT_SMALL_WPB", 4)));
    pat_small_wpb2 = std::min(4, std::max(1, opt_int("GMG_PAT_SMALL_WPB2", 2)));
    pat_dinv = opt_int("GMG_PAT_DINV", 1);
    pat_rb = opt_int("GMG_PAT_RB", 3); pat_rb = pat_rb >= 9 ? 9 : (pat_rb <= 1 ? 1 : 3);
    pat_batched = opt_int("GMG_PAT_BATCHED", 1);
    use_opattern = opt_int("GMG_OPATTERN", 1);
    pat_nb = std::min(2, std::max(0, opt_int("GMG_PAT_NB", 0)));
    tile = kTile;
  }
  // inv_diag = 1 ./ diag(A) (JacobiLinearSolvers.jl:20-23); needs the CSR stream of A (before drop_csr_stream)
  double *build_inv_diag(const DevCSR &A, int &nzero)
  {
    const int64_t n = A.nrows;
    double *dinv = dalloc<double>((size_t)n);
    if (A.rowptr == nullptr) {                               // streamed operator: 1/diag per pattern, expanded per row
      nzero = 0;
      if (!A.pdinv) { nzero = 1; return dinv; }              // some row pattern has no diagonal entry
      hipLaunchKernelGGL(expand_pattern_dinv_kernel, dim3(grid_for(n)), dim3(256), 0, stream, n, A.rowpid, A.pdinv, dinv);
      HIP_CHECK(hipGetLastError());
      std::vector<double> pd((size_t)A.pat_np);
      HIP_CHECK(hipMemcpyAsync(pd.data(), A.pdinv, sizeof(double) * pd.size(), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      for (int q = 0; q + 1 < A.pat_np; ++q) if (!std::isfinite(pd[(size_t)q])) nzero++;
      return dinv;
    }
    int *d_nzero = dalloc<int>(1);
    HIP_CHECK(hipMemsetAsync(d_nzero, 0, sizeof(int), stream));
    const int grid = (int)std::max<int64_t>(1, (n + 255) / 256);
    if (A.ptr64)
      hipLaunchKernelGGL((inv_diag_kernel<int64_t>), dim3(grid), dim3(256), 0, stream, n, (const int64_t *)A.rowptr, A.col, A.val, dinv, d_nzero);
    else
      hipLaunchKernelGGL((inv_diag_kernel<int32_t>), dim3(grid), dim3(256), 0, stream, n, (const int32_t *)A.rowptr, A.col, A.val, dinv, d_nzero);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(&nzero, d_nzero, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    return dinv;
  }
  // reductions + pinned scalars (also all a block-solver engine without levels needs)
  void init_reductions()
  {
    d_partials = dvec(kRedBlocks);
    d_partials2 = dvec(kRedBlocks);
    d_scalars = dvec(kScalarSlots);
    if (!h_scalars) HIP_CHECK(hipHostMalloc((void **)&h_scalars, kScalarSlots * sizeof(double)));
  }
  // Bytes one fused sweep launch moves with the storage layout chosen at setup, every operand once, no cache credit:
  // the matrix stream as stored (padding included) + the row-wise vector traffic of the one-gather sweep
  // (r in/out, s in/out, x in/out, D^-1) -- what `traffic` (PMC) should show when nothing is re-read.
  double sweep_layout_bytes(int l) const
  {
    const Level &L = lev[l];
    const DevCSR &A = L.A;
    const double N = (double)L.n;
    double vec = 8.0 * 7.0 * N;                            // r, r', s, s', x, x', dinv
    double mat;
    if (A.pat && A.pat_shared) {
      mat = 2.0 * N;                                       // 16-bit pattern id per row; the table lives in LDS
      if (pat_dinv && A.pdinv) vec -= 8.0 * N;             // 1/diag from the pattern table
      if (pat_defer) vec -= 4.0 * N;                       // x touched every second sweep: (8+8+8)/2 instead of 8+8
      if (pat_defer && rsweep_level(L)) vec = 28.0 * N;    // sells_rsweep_kernel: r in + r out, (x in + x out + r_prev) every second sweep; no s, no 1/diag
      if (fuse2_level(L)) vec = 16.0 * N;                  // sells_z2sweep_kernel, per sweep of the pair: (r_k in + r_{k+2} out + x in + x out) / 2
      if (box_active(A)) mat = 0.0;                        // sells_boxsweep_kernel: no pattern ids either -- the coefficients are kernel arguments
    } else if (A.pat) mat = (A.rowbase ? 6.0 : 2.0) * N;
    else if (A.sell && A.opat) { mat = 8.0 * (double)A.zpad + (A.orowbase ? 6.0 : 2.0) * N + 4.0 * N + 8.0 * (double)A.nslices; if (sell_defer && sell_un < 27) vec -= 4.0 * N; }
    else if (A.sell && (A.comp_idx || A.vdict)) mat = A.stream_bytes_per_nnz * (double)A.zpack + 4.0 * N + 4.0 * (double)(A.zpack / 64);
    else if (A.sell) { mat = 12.0 * (double)A.zpad + 4.0 * N + 8.0 * (double)A.nslices; if (sell_defer && sell_un >= 6 && sell_un < 9) vec -= 4.0 * N; }
    else mat = 12.0 * (double)A.nnz + (A.ptr64 ? 8.0 : 4.0) * N;
    return mat + vec;
  }
  // The same count for ONE form of the sweep (xmode 0: x updated every sweep, 1: x untouched, 2: x updated with both
  // pending increments).  On the layouts whose sweeps alternate between forms 1 and 2, sweep_layout_bytes() is their
  // launch-weighted mean: it carries 12 B/row for x (x in + x out + the previous increment on every second sweep).
  double sweep_layout_bytes_variant(int l, int xmode) const
  {
    const Level &L = lev[l];
    const DevCSR &A = L.A;
    const double N = (double)L.n, mean = sweep_layout_bytes(l);
    const bool defer = (A.pat && A.pat_shared) ? pat_defer != 0
                     : (A.pat) ? false
                     : (A.sell && A.opat) ? (sell_defer && sell_un < 27)
                     : (A.sell && (A.comp_idx || A.vdict)) ? false
                     : (A.sell) ? (sell_defer && sell_un >= 6 && sell_un < 9) : false;
    if (!defer || fuse2_level(L)) return mean;              // (a fused pair is one form: half of the pair's bytes per sweep)
    return mean - 12.0 * N + (xmode == 1 ? 0.0 : xmode == 2 ? 24.0 * N : 16.0 * N);
  }
  KrylovOps level0_ops(int use_precond);
  bool can_refresh() const;
  void refresh_values();
  void setup();
  void build_coarse();
  double *build_dense_inverse(const HostCSR &A, const std::string &what);
  double *build_coarse_device(const HostCSR &A, const std::string &what);
};

// ----------------------------------------------------------------------------
// Krylov drivers, written against callbacks so that the single-matrix solvers
// (gmg_cg_solve / gmg_fgmres_solve) and the block solvers (gmg_block_*) share them.
// `S` supplies the stream, the reductions and the vector kernels.
// ----------------------------------------------------------------------------

KrylovOps gmg_solver::level0_ops(int use_precond)
{
  KrylovOps ops;
  const int k = kl();
  ops.resid = [this, k](double *x, const double *b, double *r) { apply_A_resid(k, x, b, r); };
  ops.apply = [this, k](double *x, double *y) { apply_A_set(k, x, y); };
  if (comm.nranks == 1) ops.apply_dot = [this](double *x, double *y, double *parts) { return spmv_set_dot(lev[0].A, x, y, parts); };
  if (use_precond) ops.precond = [this, use_precond](double *z, const double *r, double known) { krylov_precond(use_precond, z, r, known); };
  return ops;
}

// solve!(x,ns::CGNumericalSetup,b), Krylov/CGSolvers.jl:73-120
static double cg_core(gmg_solver &S, int64_t n, const double *db, double *dx, double *w, double *p, double *z, double *r,
               const KrylovOps &ops, bool flexible, ConvLog &log)
{
  // Scalars stay on the device: gamma (ping-pong), delta and dot(p,w) live in d_scalars and the vector
  // kernels form beta / alpha from them, so an iteration has ONE host round trip (the residual norm that
  // the stopping rule needs) instead of three.
  // A CG can run inside another Krylov solve (CG-Jacobi diagonal block or coarsest solver under an outer CG): every nesting
  // level owns its four slots.
  REQUIRE(S.krylov_depth < 4, GMG_ERR_UNSUPPORTED, "Krylov solvers nested deeper than 4");
  struct Depth { int &d; Depth(int &x) : d(x) { ++d; } ~Depth() { --d; } } depth_guard(S.krylov_depth);
  const int sbase = kScalarSlots - 8 * S.krylov_depth;
  const int kG0 = sbase, kG1 = sbase + 1, kDelta = sbase + 2, kPW = sbase + 3;
  int g_old = kG0, g_new = kG1;
  ops.resid(dx, db, r);                                  // CGSolvers.jl:79  w = A x ; r = b - w
  // :80 fill!(p,0): folded into the first p = z + beta*p ; :81 fill!(z,0): only the flexible variant reads z before writing it
  if (flexible) S.zero(z, n);
  hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, S.stream, S.d_scalars + g_old, 1.0);   // :82 gamma = 1
  HIP_CHECK(hipGetLastError());
  double resn = S.norm(n, r);                            // :85
  bool done = log.init(resn);                            // :86
  bool first = true;
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(kRedBlocks, (n + kBlock - 1) / kBlock));
  // fused second stages (sum_partials_all): gamma = dot(z,r) is summed by xpby_dev_kernel, dot(p,w) by cg_update_kernel, and
  // ||r|| is reduced and posted to the host by one launch -- 4 launches per iteration instead of 8 for the scalars
  const bool fuse = S.fuse_reductions();
  const unsigned xgrid = fuse ? (unsigned)nb : (unsigned)gmg_solver::grid_for(n);
  while (!done) {
    int ngp = 0;                                         // gamma still in partials?
    if (!ops.precond) {                                  // :90-92
      S.copy(z, r, n);
      if (fuse) ngp = S.dot_partials(n, r, r, S.d_partials); else S.dot_async(n, r, r, g_new, false);
    } else if (!flexible) {                              // :93-95
      ops.precond(z, r, resn);
      if (fuse) ngp = S.dot_partials(n, z, r, S.d_partials); else S.dot_async(n, z, r, g_new, false);
    } else {                                             // :96-99
      S.dot_async(n, z, r, kDelta, false);
      ops.precond(z, r, resn);
      if (fuse) ngp = S.dot_partials(n, z, r, S.d_partials); else S.dot_async(n, z, r, g_new, false);
    }
    hipLaunchKernelGGL(xpby_dev_kernel, dim3(xgrid), dim3(kBlock), 0, S.stream, n, z, S.d_scalars + g_new,
                       S.d_scalars + g_old, (ops.precond && flexible) ? S.d_scalars + kDelta : nullptr, p, first ? 1 : 0,
                       ngp ? S.d_partials : nullptr, ngp); // :101
    HIP_CHECK(hipGetLastError());
    int npw = 0;
    if (fuse && ops.apply_dot) npw = ops.apply_dot(p, w, S.d_partials);   // :104-105 in one kernel (> 0), or :104 alone (-1)
    if (npw <= 0) {
      if (npw == 0) ops.apply(p, w);                     // :104
      if (fuse) npw = S.dot_partials(n, p, w, S.d_partials); else { S.dot_async(n, p, w, kPW, false); npw = 0; }   // :105
    }
    double *nparts = fuse ? S.d_partials2 : S.d_partials;
    hipLaunchKernelGGL(cg_update_kernel, dim3(nb), dim3(kBlock), 0, S.stream, n, S.d_scalars + g_new, S.d_scalars + kPW, p, w, dx, r,
                       nparts, npw ? S.d_partials : nullptr, npw); // :108-109
    HIP_CHECK(hipGetLastError());
    S.finish_reduction(nb, 0, true, nparts, true);
    resn = S.fetch_scalar(0);                            // :111
    done = log.update(resn);                             // :112
    std::swap(g_old, g_new);
    first = false;
  }
  return resn;
}

// solve!(x,ns::FGMRESNumericalSetup,b), Krylov/FGMRESSolvers.jl:130-199.  V/Z are the caller's basis
// caches (:58-70); they grow by m_add when the basis outgrows them (:151-154).  nv = allocation length
// of a basis vector (n plus ghost space in distributed runs).
static double fgmres_core(gmg_solver &S, int64_t n, int64_t nv, const double *db, double *dx, std::vector<double *> &V,
                   std::vector<double *> &Z, const KrylovOps &ops, int m0, bool restart, int m_add, ConvLog &log)
{
  int m = std::max<int>(m0, (int)Z.size());
  while ((int)V.size() < m + 1) V.push_back(S.dvec(nv));
  while ((int)Z.size() < m) Z.push_back(S.dvec(nv));
  m = (int)Z.size();
  // Hessenberg / rotations are (m+1) x m like the reference's caches (FGMRESSolvers.jl:58-70) and grow with the
  // basis (expand_krylov_caches!, :77-94); with restart=true j never exceeds m0.
  int hcap = m + 1;                                      // rows of H = columns + 1
  int ldh = hcap + 1;
  std::vector<double> H((size_t)ldh * hcap, 0.0), g((size_t)hcap + 1, 0.0), c((size_t)hcap, 0.0), s((size_t)hcap, 0.0);
  auto Hm = [&](int i, int j) -> double & { return H[(size_t)(i - 1) + (size_t)(j - 1) * ldh]; };
  auto grow_small = [&](int newcap) {
    const int nld = newcap + 1;
    std::vector<double> H2((size_t)nld * newcap, 0.0);
    for (int jj = 0; jj < hcap; ++jj)
      for (int ii = 0; ii < ldh; ++ii) H2[(size_t)ii + (size_t)jj * nld] = H[(size_t)ii + (size_t)jj * ldh];
    H.swap(H2);
    g.resize((size_t)newcap + 1, 0.0); c.resize((size_t)newcap, 0.0); s.resize((size_t)newcap, 0.0);
    hcap = newcap; ldh = nld;
  };
  const int grid = gmg_solver::grid_for(n);

  // krylov_residual!(V[1],x,A,b,Pl,zl): KrylovUtils.jl:46-54 ; FGMRESSolvers.jl:136-140
  auto residual = [&](double *out) {
    if (ops.precond_left) { ops.resid(dx, db, ops.zl); ops.precond_left(out, ops.zl); }
    else ops.resid(dx, db, out);
  };
  residual(V[0]);
  double beta = S.norm(n, V[0]);                         // :141
  bool done = log.init(beta);                            // :142
  while (!done) {
    int j = 1;                                           // :145
    hipLaunchKernelGGL(div_kernel, dim3(grid), dim3(256), 0, S.stream, n, beta, V[0]); // :146
    HIP_CHECK(hipGetLastError());
    std::fill(H.begin(), H.end(), 0.0);                  // :147
    std::fill(g.begin(), g.end(), 0.0); g[0] = beta;     // :148
    while (!done && !(restart && j > m0)) {              // :149
      if (j > m) {                                       // :151-154
        for (int q = 0; q < m_add; ++q) { V.push_back(S.dvec(nv)); Z.push_back(S.dvec(nv)); }
        m += m_add;
      }
      if (j + 1 > hcap) grow_small(m + 1);
      REQUIRE(j + 3 < kScalarSlots - 40, GMG_ERR_UNSUPPORTED, "Krylov basis larger than the scalar buffer (use restart=true)");
      double *Vn = V[j], *Zj = Z[j - 1];
      // krylov_mul!(V[j+1],A,V[j],Pr,nothing,Z[j],zl): KrylovUtils.jl:22-25
      if (ops.precond) ops.precond(Zj, V[j - 1], -1.0);
      else S.copy(Zj, V[j - 1], n);
      if (ops.precond_left) { ops.apply(Zj, ops.zl); ops.precond_left(Vn, ops.zl); }   // krylov_mul! with Pl, KrylovUtils.jl:14-18
      else ops.apply(Zj, Vn);                            // :159
      for (int i = 1; i <= j; ++i) {                     // :160-163 modified Gram-Schmidt
        S.dot_async(n, Vn, V[i - 1], i, false);
        hipLaunchKernelGGL(axmy_dev_kernel, dim3(grid), dim3(256), 0, S.stream, n, S.d_scalars + i, V[i - 1], Vn);
        HIP_CHECK(hipGetLastError());
      }
      S.dot_async(n, Vn, Vn, j + 1, true);               // :164
      hipLaunchKernelGGL(div_dev_kernel, dim3(grid), dim3(256), 0, S.stream, n, S.d_scalars + (j + 1), Vn); // :165
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(S.h_scalars + 1, S.d_scalars + 1, sizeof(double) * (size_t)(j + 1), hipMemcpyDeviceToHost, S.stream));
      HIP_CHECK(hipStreamSynchronize(S.stream));
      for (int i = 1; i <= j + 1; ++i) Hm(i, j) = S.h_scalars[i];
      for (int i = 1; i <= j - 1; ++i) {                 // :168-172
        const double gm = c[i - 1] * Hm(i, j) + s[i - 1] * Hm(i + 1, j);
        Hm(i + 1, j) = -s[i - 1] * Hm(i, j) + c[i - 1] * Hm(i + 1, j);
        Hm(i, j) = gm;
      }
      { // LinearAlgebra.givensAlgorithm (:175), normal-range branch + LAPACK sign rule
        const double f = Hm(j, j), gg = Hm(j + 1, j);
        double cs, sn;
        if (gg == 0.0) { cs = 1.0; sn = 0.0; }
        else if (f == 0.0) { cs = 0.0; sn = 1.0; }
        else {
          const double safmn2 = std::ldexp(1.0, -485), safmx2 = 1.0 / safmn2;
          double f1 = f, g1 = gg, scale = std::max(std::fabs(f1), std::fabs(g1));
          while (scale >= safmx2) { f1 *= safmn2; g1 *= safmn2; scale = std::max(std::fabs(f1), std::fabs(g1)); }
          while (scale <= safmn2) { f1 *= safmx2; g1 *= safmx2; scale = std::max(std::fabs(f1), std::fabs(g1)); }
          const double rr = std::sqrt(f1 * f1 + g1 * g1);
          cs = f1 / rr; sn = g1 / rr;
          if (std::fabs(f) > std::fabs(gg) && cs < 0.0) { cs = -cs; sn = -sn; }
        }
        c[j - 1] = cs; s[j - 1] = sn;
      }
      Hm(j, j) = c[j - 1] * Hm(j, j) + s[j - 1] * Hm(j + 1, j); Hm(j + 1, j) = 0.0; // :176
      g[j] = -s[j - 1] * g[j - 1]; g[j - 1] = c[j - 1] * g[j - 1];                  // :177
      beta = std::fabs(g[j]);                            // :179
      j += 1;                                            // :180
      done = log.update(beta);                           // :181
    }
    j = j - 1;                                           // :183
    for (int i = j; i >= 1; --i) {                       // :186-188
      double acc = 0.0;
      for (int k = i + 1; k <= j; ++k) acc += Hm(i, k) * g[k - 1];
      g[i - 1] = (g[i - 1] - acc) / Hm(i, i);
    }
    for (int i = 1; i <= j; ++i) {                       // :191-193
      hipLaunchKernelGGL(axpy_kernel, dim3(grid), dim3(256), 0, S.stream, n, g[i - 1], Z[i - 1], dx);
      HIP_CHECK(hipGetLastError());
    }
    residual(V[0]);                                      // :194
  }
  return beta;
}

// ----------------------------------------------------------------------------
// patch smoother: setup + application
// ----------------------------------------------------------------------------
// blocks_only: numerical_setup! -- the patch tables and incidence lists stay, the inverse blocks are rebuilt from the new values
void gmg_solver::build_patch(Level &L, Smoother &S, bool blocks_only)
{
  REQUIRE(S.tab, GMG_ERR_INVALID, "patch tables missing");
  const Smoother::Tables &T = *S.tab;
  const bool bp_timing = opt_int("GMG_SETUP_TIMING", 0) != 0;
  auto bp_last = std::chrono::steady_clock::now();
  auto bp_lap = [&](const char *what) {
    if (!bp_timing) return;
    (void)hipStreamSynchronize(stream);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[gmg_setup]   patch tables: %-34s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - bp_last).count());
    bp_last = now;
  };
  const int64_t npatch = (int64_t)T.pptr.size() - 1;
  REQUIRE(npatch >= 0, GMG_ERR_INVALID, "patch_ptr missing");
  S.npatch = npatch;
  const std::vector<int32_t> &pcolv = T.pcol.empty() ? T.prow : T.pcol;
  std::vector<int64_t> boff((size_t)npatch + 1, 0);
  int max_np = 0;
  for (int64_t p = 0; p < npatch; ++p) {
    const int64_t np = T.pptr[p + 1] - T.pptr[p];
    REQUIRE(np >= 0, GMG_ERR_INVALID, "patch_ptr not monotone");
    boff[p + 1] = boff[p] + np * np;
    max_np = std::max<int>(max_np, (int)np);
  }
  const int64_t ndof_entries = T.pptr[npatch];
  const int64_t nloc = L.nvec;                              // patches of a distributed level reach into the ghost dofs (own | ghost numbering)
  REQUIRE((int64_t)T.prow.size() == ndof_entries && (int64_t)pcolv.size() == ndof_entries, GMG_ERR_INVALID, "patch tables have the wrong length");
  {
    std::atomic<int> oob(0);
    parallel_chunks(64, [&](int64_t t) {
      const int64_t per = (ndof_entries + 63) / 64;
      for (int64_t q = t * per; q < std::min(ndof_entries, (t + 1) * per); ++q)
        if (!(T.prow[q] >= 0 && T.prow[q] < nloc && pcolv[q] >= 0 && pcolv[q] < nloc)) { oob.store(1); return; }
    });
    REQUIRE(oob.load() == 0, GMG_ERR_INVALID, "patch dof out of range");
  }
  if (T.has_blocks) REQUIRE((int64_t)T.blocks.size() == boff[npatch], GMG_ERR_INVALID, "patch blocks have the wrong total size");
  S.max_np = max_np;
  if (blocks_only) {
    REQUIRE(S.built && S.d_pptr, GMG_ERR_STATE, "patch smoother was never set up");
    HIP_CHECK(hipStreamSynchronize(stream));
    release(S.d_binv, (size_t)S.n_binv);
    release(S.d_ubinv, (size_t)S.n_ubinv);
    release(S.d_uboff, (size_t)S.n_uboff);
    release(S.d_ublock, (size_t)npatch);
    release(S.d_boff, (size_t)npatch + 1);
    S.dedup = false; S.nuniq = 0;
  } else {
  S.d_pptr = upload(T.pptr);
  S.d_pdofs = upload(T.prow);
  S.d_pcol = T.pcol.empty() ? S.d_pdofs : upload(T.pcol);
  }
  bp_lap("validation + upload");
  int32_t *d_pcol = S.d_pcol;
  // dof -> contribution slots, ascending patch order (= reference loop order PatchSolvers.jl:288).  Built at the first setup;
  // again after a value refresh when the row-pattern form of the patch operator (which had released these buffers) can no
  // longer be formed.
  auto make_incidence = [&]() {
    S.d_contrib = dvec(ndof_entries + 1);                  // + one slot that stays 0.0: the padding target of the sliced incidence
    std::vector<int64_t> iptr((size_t)nloc + 1, 0), inc((size_t)ndof_entries);
    for (int64_t q = 0; q < ndof_entries; ++q) iptr[pcolv[q] + 1]++;
    for (int64_t i = 0; i < nloc; ++i) iptr[i + 1] += iptr[i];
    {
      std::vector<int64_t> fill(iptr.begin(), iptr.end() - 1);
      for (int64_t q = 0; q < ndof_entries; ++q) inc[fill[pcolv[q]]++] = q;
    }
    if (ndof_entries < (int64_t)INT32_MAX && nloc > 0) {
      const int64_t ns = (nloc + 63) / 64;
      std::vector<int64_t> soff((size_t)ns + 1, 0);
      for (int64_t sl = 0; sl < ns; ++sl) {
        int64_t w = 0;
        for (int64_t i = sl * 64; i < std::min<int64_t>(nloc, sl * 64 + 64); ++i) w = std::max(w, iptr[i + 1] - iptr[i]);
        soff[sl + 1] = soff[sl] + w * 64;
      }
      std::vector<int32_t> sinc((size_t)soff[ns], (int32_t)ndof_entries);   // padding -> the zero slot
      parallel_for(ns, [&](int64_t sl) {
        for (int l = 0; l < 64; ++l) {
          const int64_t i = sl * 64 + l;
          if (i >= nloc) break;
          for (int64_t k = iptr[i]; k < iptr[i + 1]; ++k) sinc[(size_t)(soff[sl] + (k - iptr[i]) * 64 + l)] = (int32_t)inc[k];
        }
      });
      S.d_isoff = upload(soff);
      S.d_isinc = upload_padded(sinc, 64);
      S.n_isoff = (int64_t)soff.size(); S.n_isinc = (int64_t)sinc.size() + 64;
      S.d_iptr = nullptr; S.d_inc = nullptr;
    } else {
      S.d_iptr = upload(iptr);
      S.d_inc = upload(inc);
    }
  };
  // (built after the blocks, and only when the patch-by-patch kernels will run: the row-pattern form of the additive-Schwarz
  // operator needs none of it, and at 4.6 x 10^8 patch entries the lists take 2.8 s of host time)
  if (npatch == 0 || max_np == 0) { if (!blocks_only) make_incidence(); S.built = true; return; }

  // ---- inverse blocks, built in batches (bounded scratch) and de-duplicated on the fly -------------------------------
  // Block sources: caller's lu! factors (inverted on the host exactly as ldiv! would solve against the identity), caller's
  // patch matrices, or A[rows_p, cols_p] gathered on the device from the CSR / row-pattern form of the level operator.
  const bool from_factors = T.has_blocks && T.are_factors;
  const bool from_dense = T.has_blocks && !T.are_factors;
  const bool from_sell = !T.has_blocks && L.A.col == nullptr && L.A.sell && !L.A.pat && L.A.scol && L.A.sval;   // CSR stream dropped after the first setup
  const bool from_pattern = !T.has_blocks && !from_sell && (L.A.rowptr == nullptr || L.A.col == nullptr);
  if (from_pattern) REQUIRE(L.A.pat && L.A.plen && L.A.ppoff && L.A.ppval, GMG_ERR_STATE, "level operator holds neither a CSR nor a pattern table for the patch blocks");
  const bool wave_kernel = max_np <= 64;
  REQUIRE(wave_kernel || (!T.has_blocks && !from_pattern && !from_sell), GMG_ERR_UNSUPPORTED, "patches larger than 64 dofs need the level's CSR");
  const int pivoting = (S.patch_kind == GMG_PATCH_LU) ? 1 : 0;
  const size_t per = (size_t)max_np * max_np;
  const int64_t batch = std::max<int64_t>(1, std::min<int64_t>(npatch, (int64_t)((512u << 20) / (per * sizeof(double)))));
  const bool want_dedup = opt_int("GMG_PATCH_DEDUP", 1) && max_np <= 32 && npatch >= 64;
  int *d_nsing = dalloc<int>(1);
  S.d_boff = upload(boff);
  int64_t tmp_elems = 1;
  for (int64_t p0 = 0; p0 < npatch; p0 += batch) tmp_elems = std::max(tmp_elems, boff[std::min(npatch, p0 + batch)] - boff[p0]);

  struct DevTmp {                                           // setup-only scratch, freed on every exit path
    std::vector<void *> ptrs;
    void *get(size_t bytes) { void *q = nullptr; if (hipMalloc(&q, std::max<size_t>(bytes, 8)) != hipSuccess) throw GmgError{GMG_ERR_ALLOC, "patch setup scratch"}; ptrs.push_back(q); return q; }
    void drop(void *q) { auto it = std::find(ptrs.begin(), ptrs.end(), q); if (it != ptrs.end()) { ptrs.erase(it); (void)hipFree(q); } }
    ~DevTmp() { for (void *q : ptrs) (void)hipFree(q); }
  };

  // returns false when de-duplication should be abandoned (almost all blocks distinct, or a hash collision)
  auto build_blocks = [&](bool dedup) -> bool {
    DevTmp tmp;
    HIP_CHECK(hipMemsetAsync(d_nsing, 0, sizeof(int), stream));
    double *d_tmp = nullptr, *d_dense = nullptr, *d_scratch = nullptr, *d_ustore = nullptr;
    unsigned long long *d_hash = nullptr;
    int64_t *d_rep = nullptr;
    int *d_nmis = nullptr;
    std::unordered_map<unsigned long long, int32_t> uniq;   // hash -> unique id
    std::vector<int64_t> uboff(1, 0);
    std::vector<int32_t> ublock;
    int64_t ucap = 0, uused = 0;
    if (dedup) {
      ublock.assign((size_t)npatch, 0);
      d_tmp = (double *)tmp.get(sizeof(double) * (size_t)tmp_elems);
      d_hash = (unsigned long long *)tmp.get(sizeof(unsigned long long) * (size_t)batch);
      d_rep = (int64_t *)tmp.get(sizeof(int64_t) * (size_t)batch);
      d_nmis = (int *)tmp.get(sizeof(int));
      HIP_CHECK(hipMemsetAsync(d_nmis, 0, sizeof(int), stream));
    } else if (!S.d_binv) { S.d_binv = dalloc<double>((size_t)boff[npatch]); S.n_binv = boff[npatch]; }
    if (from_dense) d_dense = (double *)tmp.get(sizeof(double) * (size_t)tmp_elems);
    if (!wave_kernel) d_scratch = (double *)tmp.get(per * sizeof(double) * (size_t)batch);
    std::vector<double> hinv;                               // host-inverted batch (caller's factors)
    std::vector<unsigned long long> hash((size_t)batch);
    std::vector<int64_t> rep((size_t)batch);
    for (int64_t p0 = 0; p0 < npatch; p0 += batch) {
      const int64_t cnt = std::min(batch, npatch - p0);
      const int64_t e0 = boff[p0], ne = boff[p0 + cnt] - e0;
      double *out = dedup ? d_tmp : S.d_binv + e0;           // batch-relative base: block p sits at out[boff[p]-e0]
      if (from_factors) {
        // inverse = ldiv!(F, I): the row-permuted identity through the unit-lower and the upper factor (LAPACK getrs
        // order), column by column; factors column-major as lu! leaves them, pivots 1-based (ipiv)
        hinv.assign((size_t)ne, 0.0);
        std::atomic<int> bad(0);
        parallel_for(cnt, [&](int64_t i) {
          const int64_t pg = p0 + i;
          const int np = (int)(T.pptr[pg + 1] - T.pptr[pg]);
          if (np == 0) return;
          const double *F = T.blocks.data() + boff[pg];
          const int32_t *ip = T.piv.empty() ? nullptr : T.piv.data() + T.pptr[pg];
          double *Xo = hinv.data() + (boff[pg] - e0);
          std::vector<double> col((size_t)np);
          for (int c = 0; c < np; ++c) {
            for (int r = 0; r < np; ++r) col[r] = (r == c) ? 1.0 : 0.0;
            if (ip) for (int r = 0; r < np; ++r) { const int q = ip[r] - 1; if (q != r && q >= 0 && q < np) std::swap(col[r], col[q]); }
            for (int k = 0; k < np; ++k)                     // L y = P e_c (unit diagonal)
              if (col[k] != 0.0) for (int r = k + 1; r < np; ++r) col[r] -= col[k] * F[r + (size_t)k * np];
            for (int k = np - 1; k >= 0; --k) {              // U x = y
              if (F[k + (size_t)k * np] == 0.0) { bad.store(1); return; }
              if (col[k] != 0.0) {
                col[k] /= F[k + (size_t)k * np];
                for (int r = 0; r < k; ++r) col[r] -= col[k] * F[r + (size_t)k * np];
              }
            }
            for (int r = 0; r < np; ++r) Xo[(size_t)r * np + c] = col[r];   // row-major inverse
          }
        });
        REQUIRE(bad.load() == 0, GMG_ERR_SINGULAR, "singular patch factor (zero on the diagonal of U)");
        HIP_CHECK(hipMemcpyAsync(out, hinv.data(), sizeof(double) * (size_t)ne, hipMemcpyHostToDevice, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
      } else if (wave_kernel) {
        PatchSrc src;
        std::memset(&src, 0, sizeof(src));
        const size_t lds = 2 * per * sizeof(double) + (size_t)max_np * sizeof(int32_t) + 8;
        const dim3 g((unsigned)cnt), b(64);
        if (from_dense) {
          HIP_CHECK(hipMemcpyAsync(d_dense, T.blocks.data() + e0, sizeof(double) * (size_t)ne, hipMemcpyHostToDevice, stream));
          src.dense = d_dense; src.dense_off0 = e0;
          hipLaunchKernelGGL((patch_invert_kernel<PSRC_DENSE>), g, b, lds, stream, cnt, S.d_pptr + p0, S.d_pdofs, d_pcol, S.d_boff + p0, e0, src,
                             pivoting, out, max_np, d_nsing);
        } else if (from_sell) {
          src.soff = L.A.soff; src.scol = L.A.scol; src.sval = L.A.sval; src.rowlen = L.A.rowlen;
          hipLaunchKernelGGL((patch_invert_kernel<PSRC_SELL>), g, b, lds, stream, cnt, S.d_pptr + p0, S.d_pdofs, d_pcol, S.d_boff + p0, e0, src,
                             pivoting, out, max_np, d_nsing);
        } else if (from_pattern) {
          src.rowpid = L.A.rowpid; src.rowbase = L.A.rowbase; src.plen = L.A.plen; src.poff8 = L.A.ppoff; src.pval = L.A.ppval; src.W = L.A.pat_w;
          hipLaunchKernelGGL((patch_invert_kernel<PSRC_PATTERN>), g, b, lds, stream, cnt, S.d_pptr + p0, S.d_pdofs, d_pcol, S.d_boff + p0, e0, src,
                             pivoting, out, max_np, d_nsing);
        } else {
          src.rowptr = L.A.rowptr; src.col = L.A.col; src.val = L.A.val; src.ptr64 = L.A.ptr64 ? 1 : 0;
          hipLaunchKernelGGL((patch_invert_kernel<PSRC_CSR>), g, b, lds, stream, cnt, S.d_pptr + p0, S.d_pdofs, d_pcol, S.d_boff + p0, e0, src,
                             pivoting, out, max_np, d_nsing);
        }
        HIP_CHECK(hipGetLastError());
        if (from_dense) HIP_CHECK(hipStreamSynchronize(stream));   // d_dense is refilled by the next batch
      } else {
        // patches with more than 64 dofs: one thread per patch on global scratch (rows = cols, CSR source)
        const int grid = (int)((cnt + 63) / 64);
        if (L.A.ptr64)
          hipLaunchKernelGGL((patch_factor_kernel<int64_t>), dim3(grid), dim3(64), 0, stream, cnt, S.d_pptr + p0, S.d_pdofs,
                             S.d_boff + p0, (const int64_t *)L.A.rowptr, L.A.col, L.A.val, out - e0, d_scratch, max_np, pivoting, d_nsing);
        else
          hipLaunchKernelGGL((patch_factor_kernel<int32_t>), dim3(grid), dim3(64), 0, stream, cnt, S.d_pptr + p0, S.d_pdofs,
                             S.d_boff + p0, (const int32_t *)L.A.rowptr, L.A.col, L.A.val, out - e0, d_scratch, max_np, pivoting, d_nsing);
        HIP_CHECK(hipGetLastError());
      }
      if (!dedup) continue;
      // ---- de-duplicate bitwise-identical inverse blocks (lossless; uniform meshes hold a few dozen distinct blocks) ----
      hipLaunchKernelGGL(block_hash_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, cnt, S.d_pptr + p0, S.d_boff + p0, d_tmp - e0, d_hash);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(hash.data(), d_hash, sizeof(unsigned long long) * (size_t)cnt, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      std::vector<int64_t> new_src, new_dst, new_len;       // unique blocks first seen in this batch
      for (int64_t i = 0; i < cnt; ++i) {
        const int64_t pg = p0 + i;
        const int64_t np = T.pptr[pg + 1] - T.pptr[pg];
        auto it = uniq.find(hash[(size_t)i]);
        if (it == uniq.end()) {
          it = uniq.emplace(hash[(size_t)i], (int32_t)(uboff.size() - 1)).first;
          new_src.push_back(boff[pg] - e0); new_dst.push_back(uboff.back()); new_len.push_back(np * np);
          uboff.push_back(uboff.back() + np * np);
        }
        ublock[(size_t)pg] = it->second;
        rep[(size_t)i] = uboff[(size_t)it->second];
      }
      // (almost) every block distinct: the compact store would be the full store plus overhead
      if ((int64_t)(uboff.size() - 1) * 4 > p0 + cnt) return false;
      if (uboff.back() > ucap) {
        const int64_t ncap = std::max<int64_t>(uboff.back() * 2, 1 << 16);
        double *nst = (double *)tmp.get(sizeof(double) * (size_t)ncap);
        if (d_ustore) {
          HIP_CHECK(hipMemcpyAsync(nst, d_ustore, sizeof(double) * (size_t)uused, hipMemcpyDeviceToDevice, stream));
          HIP_CHECK(hipStreamSynchronize(stream));
          tmp.drop(d_ustore);
        }
        d_ustore = nst; ucap = ncap;
      }
      for (size_t q = 0; q < new_src.size(); ++q)
        HIP_CHECK(hipMemcpyAsync(d_ustore + new_dst[q], d_tmp + new_src[q], sizeof(double) * (size_t)new_len[q], hipMemcpyDeviceToDevice, stream));
      uused = uboff.back();
      HIP_CHECK(hipMemcpyAsync(d_rep, rep.data(), sizeof(int64_t) * (size_t)cnt, hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(block_verify_store_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, cnt, S.d_pptr + p0, S.d_boff + p0,
                         d_tmp - e0, d_ustore, d_rep, d_nmis);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipStreamSynchronize(stream));              // rep / hash / d_tmp are reused by the next batch
    }
    int nsing = 0, nmis = 0;
    HIP_CHECK(hipMemcpyAsync(&nsing, d_nsing, sizeof(int), hipMemcpyDeviceToHost, stream));
    if (d_nmis) HIP_CHECK(hipMemcpyAsync(&nmis, d_nmis, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    REQUIRE(nsing == 0, GMG_ERR_SINGULAR, "singular patch block (BlockJacobiSolvers.jl:163 'Factorization failed')");
    if (!dedup) return true;
    if (nmis != 0) return false;                            // hash collision: every patch must be bitwise equal to its representative
    S.d_uboff = upload(uboff); S.n_uboff = (int64_t)uboff.size();
    S.d_ubinv = dalloc<double>((size_t)uboff.back()); S.n_ubinv = uboff.back();
    HIP_CHECK(hipMemcpyAsync(S.d_ubinv, d_ustore, sizeof(double) * (size_t)uboff.back(), hipMemcpyDeviceToDevice, stream));
    S.d_ublock = upload(ublock);
    S.h_ublock = ublock; S.h_uboff = uboff;
    S.nuniq = (int64_t)uboff.size() - 1;
    S.dedup = true;
    HIP_CHECK(hipStreamSynchronize(stream));
    return true;
  };
  // Row-pattern sources: group the patches by the SIGNATURE of their source block first (kernels.hpp: patch_sig_hash_kernel) and
  // invert one representative per group -- the groups' inverses are then de-duplicated bitwise exactly as above, so block ids and
  // stores come out as the batch-by-batch path gives them (that one inverts every patch: 5.1 s of the 8.2 s level-0 setup of
  // BASELINE config 3, this one 64 of 1.7 x 10^7).  Returns false when the signatures do not repeat (or a hash collided).
  auto bp_sub = std::chrono::steady_clock::now();
  auto build_blocks_by_source = [&]() -> bool {
    DevTmp tmp;
    auto sub = [&](const char *what) { if (bp_timing) { (void)hipStreamSynchronize(stream); const auto now = std::chrono::steady_clock::now(); std::fprintf(stderr, "[gmg_setup]     source de-duplication: %-24s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - bp_sub).count()); bp_sub = now; } };
    PatchSrc src;
    std::memset(&src, 0, sizeof(src));
    src.rowpid = L.A.rowpid; src.rowbase = L.A.rowbase; src.plen = L.A.plen; src.poff8 = L.A.ppoff; src.pval = L.A.ppval; src.W = L.A.pat_w;
    unsigned long long *d_hash = (unsigned long long *)tmp.get(sizeof(unsigned long long) * (size_t)npatch);
    hipLaunchKernelGGL(patch_sig_hash_kernel, dim3((unsigned)((npatch + 3) / 4)), dim3(256), 0, stream, npatch, S.d_pptr, S.d_pdofs, d_pcol, src, d_hash);
    HIP_CHECK(hipGetLastError());
    std::vector<unsigned long long> hash((size_t)npatch);
    HIP_CHECK(hipMemcpyAsync(hash.data(), d_hash, sizeof(unsigned long long) * (size_t)npatch, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    tmp.drop(d_hash);
    sub("signature hashes");
    // groups numbered by first appearance: per chunk the distinct hashes with their first patch, merged in chunk order
    const int64_t NT = std::max<int64_t>(1, std::min<int64_t>(64, npatch / 65536));
    const int64_t chunk = (npatch + NT - 1) / NT;
    std::vector<std::vector<std::pair<unsigned long long, int64_t>>> firsts((size_t)NT);
    parallel_chunks(NT, [&](int64_t t) {
      std::unordered_map<unsigned long long, char> seen;
      unsigned long long last = 0; bool have = false;
      for (int64_t pp = t * chunk; pp < std::min(npatch, (t + 1) * chunk); ++pp) {
        const unsigned long long h = hash[(size_t)pp];
        if (have && h == last) continue;
        last = h; have = true;
        if (seen.emplace(h, 0).second) firsts[(size_t)t].emplace_back(h, pp);
      }
    });
    std::unordered_map<unsigned long long, int32_t> gid;
    std::vector<int64_t> rep;                                // group -> first patch
    for (int64_t t = 0; t < NT; ++t)
      for (const auto &f : firsts[(size_t)t])
        if (gid.emplace(f.first, (int32_t)rep.size()).second) rep.push_back(f.second);
    const int64_t ngrp = (int64_t)rep.size();
    if (ngrp * 4 > npatch) return false;
    std::vector<int32_t> grp((size_t)npatch);
    parallel_chunks(NT, [&](int64_t t) {
      unsigned long long last = 0; int32_t lastg = -1;
      for (int64_t pp = t * chunk; pp < std::min(npatch, (t + 1) * chunk); ++pp) {
        const unsigned long long h = hash[(size_t)pp];
        if (lastg < 0 || h != last) { last = h; lastg = gid.find(h)->second; }
        grp[(size_t)pp] = lastg;
      }
    });
    hash.clear(); hash.shrink_to_fit();
    sub("grouping (host)");
    // exact check of every patch against its group's representative
    int32_t *d_grp = (int32_t *)tmp.get(sizeof(int32_t) * (size_t)npatch);
    int64_t *d_replist = (int64_t *)tmp.get(sizeof(int64_t) * (size_t)ngrp);
    int *d_nmis = (int *)tmp.get(sizeof(int));
    HIP_CHECK(hipMemsetAsync(d_nmis, 0, sizeof(int), stream));
    HIP_CHECK(hipMemsetAsync(d_nsing, 0, sizeof(int), stream));
    HIP_CHECK(hipMemcpyAsync(d_grp, grp.data(), sizeof(int32_t) * (size_t)npatch, hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(d_replist, rep.data(), sizeof(int64_t) * (size_t)ngrp, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(patch_sig_verify_kernel, dim3((unsigned)((npatch + 3) / 4)), dim3(256), 0, stream, npatch, S.d_pptr, S.d_pdofs, d_pcol, src, d_grp, d_replist, d_nmis);
    HIP_CHECK(hipGetLastError());
    // the representatives' inverses, compact: group g at roff[g]
    std::vector<int64_t> rpptr((size_t)ngrp + 1, 0), roff((size_t)ngrp + 1, 0);
    for (int64_t g = 0; g < ngrp; ++g) {
      const int64_t np = T.pptr[(size_t)rep[(size_t)g] + 1] - T.pptr[(size_t)rep[(size_t)g]];
      rpptr[(size_t)g + 1] = rpptr[(size_t)g] + np;
      roff[(size_t)g + 1] = roff[(size_t)g] + np * np;
    }
    double *d_rinv = (double *)tmp.get(sizeof(double) * (size_t)std::max<int64_t>(1, roff[(size_t)ngrp]));
    int64_t *d_rpptr = (int64_t *)tmp.get(sizeof(int64_t) * ((size_t)ngrp + 1));
    int64_t *d_roff = (int64_t *)tmp.get(sizeof(int64_t) * ((size_t)ngrp + 1));
    HIP_CHECK(hipMemcpyAsync(d_rpptr, rpptr.data(), sizeof(int64_t) * ((size_t)ngrp + 1), hipMemcpyHostToDevice, stream));
    HIP_CHECK(hipMemcpyAsync(d_roff, roff.data(), sizeof(int64_t) * ((size_t)ngrp + 1), hipMemcpyHostToDevice, stream));
    {
      const size_t lds = 2 * per * sizeof(double) + (size_t)max_np * sizeof(int32_t) + 8;
      hipLaunchKernelGGL((patch_invert_kernel<PSRC_PATTERN>), dim3((unsigned)ngrp), dim3(64), lds, stream, ngrp, S.d_pptr, S.d_pdofs, d_pcol, S.d_boff, (int64_t)0, src,
                         pivoting, d_rinv, max_np, d_nsing, d_replist, d_roff);
      HIP_CHECK(hipGetLastError());
    }
    // second stage: groups with bitwise equal inverses share one stored block (ids by first appearance, as the batch path numbers them)
    unsigned long long *d_ghash = (unsigned long long *)tmp.get(sizeof(unsigned long long) * (size_t)ngrp);
    hipLaunchKernelGGL(block_hash_kernel, dim3((unsigned)((ngrp + 255) / 256)), dim3(256), 0, stream, ngrp, d_rpptr, d_roff, d_rinv, d_ghash);
    HIP_CHECK(hipGetLastError());
    std::vector<unsigned long long> ghash((size_t)ngrp);
    int nsing = 0, nmis = 0;
    HIP_CHECK(hipMemcpyAsync(ghash.data(), d_ghash, sizeof(unsigned long long) * (size_t)ngrp, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(&nsing, d_nsing, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipMemcpyAsync(&nmis, d_nmis, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    REQUIRE(nsing == 0, GMG_ERR_SINGULAR, "singular patch block (BlockJacobiSolvers.jl:163 'Factorization failed')");
    if (nmis != 0) return false;
    sub("verify + invert groups");
    std::unordered_map<unsigned long long, int32_t> uniq;
    std::vector<int64_t> uboff(1, 0), usrc, goff((size_t)ngrp);
    std::vector<int32_t> uid((size_t)ngrp);
    for (int64_t g = 0; g < ngrp; ++g) {
      const int64_t len = roff[(size_t)g + 1] - roff[(size_t)g];
      auto it = uniq.find(ghash[(size_t)g]);
      if (it == uniq.end()) {
        it = uniq.emplace(ghash[(size_t)g], (int32_t)(uboff.size() - 1)).first;
        usrc.push_back(roff[(size_t)g]);
        uboff.push_back(uboff.back() + len);
      }
      uid[(size_t)g] = it->second;
      goff[(size_t)g] = uboff[(size_t)it->second];
    }
    const int64_t nu = (int64_t)uboff.size() - 1;
    double *d_ustore = (double *)tmp.get(sizeof(double) * (size_t)std::max<int64_t>(1, uboff.back()));
    for (int64_t u = 0; u < nu; ++u)
      HIP_CHECK(hipMemcpyAsync(d_ustore + uboff[(size_t)u], d_rinv + usrc[(size_t)u], sizeof(double) * (size_t)(uboff[(size_t)u + 1] - uboff[(size_t)u]), hipMemcpyDeviceToDevice, stream));
    int64_t *d_goff = (int64_t *)tmp.get(sizeof(int64_t) * (size_t)ngrp);
    HIP_CHECK(hipMemcpyAsync(d_goff, goff.data(), sizeof(int64_t) * (size_t)ngrp, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(block_verify_store_kernel, dim3((unsigned)((ngrp + 255) / 256)), dim3(256), 0, stream, ngrp, d_rpptr, d_roff, d_rinv, d_ustore, d_goff, d_nmis);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(&nmis, d_nmis, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (nmis != 0) return false;
    std::vector<int32_t> ublock((size_t)npatch);
    parallel_chunks(NT, [&](int64_t t) {
      for (int64_t pp = t * chunk; pp < std::min(npatch, (t + 1) * chunk); ++pp) ublock[(size_t)pp] = uid[(size_t)grp[(size_t)pp]];
    });
    sub("second stage + block ids");
    S.d_uboff = upload(uboff); S.n_uboff = (int64_t)uboff.size();
    S.d_ubinv = dalloc<double>((size_t)uboff.back()); S.n_ubinv = uboff.back();
    HIP_CHECK(hipMemcpyAsync(S.d_ubinv, d_ustore, sizeof(double) * (size_t)uboff.back(), hipMemcpyDeviceToDevice, stream));
    S.d_ublock = upload(ublock);
    S.h_ublock = std::move(ublock); S.h_uboff = uboff;
    S.nuniq = nu;
    S.dedup = true;
    HIP_CHECK(hipStreamSynchronize(stream));
    sub("uploads");
    if (bp_timing) std::fprintf(stderr, "[gmg_setup]   patch tables: %lld source signatures, %lld distinct inverse blocks of %lld patches\n", (long long)ngrp, (long long)nu, (long long)npatch);
    return true;
  };
  const bool by_source = want_dedup && from_pattern && wave_kernel && opt_int("GMG_PATCH_SOURCE_DEDUP", 1);
  bp_lap("block offsets");
  bp_sub = std::chrono::steady_clock::now();
  if (!(by_source && build_blocks_by_source()))
    if (!(want_dedup && build_blocks(true))) build_blocks(false);
  bp_lap("inverse blocks + de-duplication");
  if (S.dedup) release(S.d_boff, (size_t)npatch + 1);        // the de-duplicated solve addresses blocks through ublock / uboff
  S.built = true;
  build_patch_operator(L, S);
  S.h_ublock.clear(); S.h_ublock.shrink_to_fit(); S.h_uboff.clear();
  bp_lap("patch operator");
  if (!S.use_M && !S.d_contrib) make_incidence();           // (value refresh: the patch-by-patch kernels are needed again)
  bp_lap("incidence lists");
}

// ----------------------------------------------------------------------------
// Additive-Schwarz operator in row-pattern form.
//   dx = sum_p R_p^T inv(A_pp) R_p r  (PatchSolvers.jl:288-297: x_p = A_pp \ r[rows_p]; dx[rows_p] += x_p in patch order)
// is a sparse operator M with the sparsity of the level matrix (Q2 vertex stars: 27..125 entries per row) and, once the patch
// blocks have been de-duplicated, only as many DISTINCT rows as there are combinations of (block, local row, patch shape,
// position) around a dof -- a few hundred on a uniform mesh.  Every row gets a signature (those words for each of its
// patches, in patch order); equal signatures are exactly equal rows, so the signatures are numbered (exact comparison, ids
// by first appearance), ONE row per signature is merged -- entries of the same column summed in patch order -- and M is
// handed to the row-pattern machinery of the level operators (16-bit id per row + table; coded shared-offset form for wide
// rows).  dx = M r then costs one pattern mat-vec (1.7 ms on 1.3 x 10^8 Q2 dofs) instead of patch solves + a 27-double
// contribution per patch + the incidence gather (8.4 ms).  Rounding: the coefficients of a column are pre-summed over the
// patches, the products are summed in column order -- same tolerance class as the explicit inverses the patch solve
// already uses instead of the reference's LU solve (tests: <= 1e-12 against the oracle).
// Not taken (the patch kernels stay): distributed levels, patch_rows != patch_cols, blocks that did not de-duplicate,
// more than 4096 signatures, GMG_PATCH_OPERATOR=0.
// ----------------------------------------------------------------------------
void gmg_solver::free_pattern(DevCSR &M)
{
  HIP_CHECK(hipStreamSynchronize(stream));
  release(M.rowpid, (size_t)M.nrows + 64);
  release(M.rowbase, (size_t)M.nrows + 64);
  release(M.plen, (size_t)1); release(M.ppoff, (size_t)1); release(M.ppval, (size_t)1);
  release(M.ptab, (size_t)1); release(M.ptab8, (size_t)1); release(M.prun, (size_t)1); release(M.pdinv, (size_t)1);
  release(M.pcodes, (size_t)1); release(M.pdict, (size_t)1); release(M.prunmask, (size_t)1);
  if (M.wl_state == 1) { release(M.wl_pids, (size_t)1); release(M.wl_cnt, (size_t)1); }
  if (M.wz_state == 1) { release(M.wz_pids, (size_t)1); release(M.wz_cnt, (size_t)1); }
  M = DevCSR();
}

void gmg_solver::build_patch_operator(Level &L, Smoother &S)
{
  if (S.use_M) free_pattern(S.M);
  S.use_M = false;
  const Smoother::Tables &T = *S.tab;
  const int64_t npatch = S.npatch, n = L.n;
  if (!opt_int("GMG_PATCH_OPERATOR", 1) || !S.dedup || !T.pcol.empty() || (comm.nranks > 1 && L.halo.present && !L.halo.ovl) || !use_pattern || !use_sell) return;
  if (npatch < 64 || n < 64 || n >= (int64_t)(1 << 28) || S.max_np > 32 || S.h_ublock.size() != (size_t)npatch) return;
  const auto t_begin = std::chrono::steady_clock::now();
  auto t_sub = t_begin;
  const bool po_timing = opt_int("GMG_SETUP_TIMING", 0) != 0;
  auto sub = [&](const char *what) { if (!po_timing) return; const auto now = std::chrono::steady_clock::now(); std::fprintf(stderr, "[gmg_setup]     patch operator: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_sub).count()); t_sub = now; };
  const int64_t ne = T.pptr[npatch];
  if (ne >= (int64_t)INT32_MAX) return;
  // unique inverse blocks back on the host (a few KB .. MB)
  std::vector<double> ubinv((size_t)S.n_ubinv);
  HIP_CHECK(hipMemcpyAsync(ubinv.data(), S.d_ubinv, sizeof(double) * (size_t)S.n_ubinv, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  // exact numbering of integer sequences: ids by first appearance (two parallel passes + a short sequential one)
  auto sequence_ids = [&](int64_t nitems, auto len, auto word, std::vector<int32_t> &ids, std::vector<int64_t> &rep, int max_ids) -> bool {
    ids.assign((size_t)nitems, -1);
    rep.clear();
    std::vector<uint64_t> hashes((size_t)nitems);
    const int Tn = (int)std::max<int64_t>(1, std::min<int64_t>(64, (nitems + 16383) / 16384));
    const int64_t per = (nitems + Tn - 1) / Tn;
    std::unordered_map<uint64_t, std::vector<int32_t>> index;
    auto same = [&](int64_t a, int64_t b) {
      const int la = len(a);
      if (la != len(b)) return false;
      for (int j = 0; j < la; ++j)
        if (word(a, j) != word(b, j)) return false;
      return true;
    };
    std::atomic<bool> over(false);
    for (int round = 0; round < 64; ++round) {
      std::vector<std::vector<std::pair<int64_t, uint64_t>>> fresh((size_t)Tn);
      parallel_chunks(Tn, [&](int64_t t) {
        std::unordered_map<uint64_t, int64_t> first;
        for (int64_t i = t * per; i < std::min(nitems, (t + 1) * per); ++i) {
          if (ids[(size_t)i] >= 0) continue;
          uint64_t h;
          if (round == 0) {
            const int li = len(i);
            h = 1469598103934665603ull ^ (uint64_t)li;
            for (int j = 0; j < li; ++j) { h = (h ^ (uint64_t)word(i, j)) * 1099511628211ull; h ^= h >> 29; }
            hashes[(size_t)i] = h;
          } else
            h = hashes[(size_t)i];
          auto it = index.find(h);
          bool ok = false;
          if (it != index.end())
            for (int32_t q : it->second)
              if (same(i, rep[(size_t)q])) { ids[(size_t)i] = q; ok = true; break; }
          if (ok) continue;
          if (over.load(std::memory_order_relaxed)) return;
          if ((int)first.size() > max_ids) { over.store(true); return; }
          auto ins = first.emplace(h, i);
          if (!ins.second && same(i, ins.first->second)) ids[(size_t)i] = (int32_t)(-2 - (ins.first->second - t * per));
        }
        for (const auto &kv : first) fresh[(size_t)t].emplace_back(kv.second, kv.first);
      });
      if (over.load()) return false;
      std::vector<std::pair<int64_t, uint64_t>> todo;
      for (auto &f : fresh) todo.insert(todo.end(), f.begin(), f.end());
      if (todo.empty()) return true;
      std::sort(todo.begin(), todo.end());
      for (auto &pr : todo) {
        auto &bucket = index[pr.second];
        int32_t q = -1;
        for (int32_t cand : bucket)
          if (same(pr.first, rep[(size_t)cand])) { q = cand; break; }
        if (q < 0) {
          if ((int)rep.size() >= max_ids) return false;
          q = (int32_t)rep.size();
          rep.push_back(pr.first);
          bucket.push_back(q);
        }
        ids[(size_t)pr.first] = q;
      }
      parallel_chunks(Tn, [&](int64_t t) {
        for (int64_t i = t * per; i < std::min(nitems, (t + 1) * per); ++i)
          if (ids[(size_t)i] <= -2) ids[(size_t)i] = ids[(size_t)(t * per + (int64_t)(-2 - ids[(size_t)i]))];
      });
    }
    return false;
  };
  // patch shapes: dof offsets relative to the patch's first dof
  std::vector<int32_t> shape;
  std::vector<int64_t> shape_rep;
  if (!sequence_ids(npatch, [&](int64_t p) { return (int)(T.pptr[p + 1] - T.pptr[p]); },
                    [&](int64_t p, int j) { return (int64_t)T.prow[(size_t)(T.pptr[p] + j)] - (int64_t)T.prow[(size_t)T.pptr[p]]; }, shape, shape_rep, 4096))
    return;
  sub("patch shapes");
  // ---- which patches touch a dof (ascending patch order) and one signature per row: on the device (option patch_op_device, default)
  //      -- 4.6e8 slots / 1.3e8 rows at 256^3 Q2 took 2.1 s on 16 host cores -- or on the host
  std::vector<int64_t> iptr;
  std::vector<int32_t> inc, s2p, sig;
  std::vector<int64_t> sig_rep;
  std::vector<std::vector<int32_t>> rep_slots;               // slots of every representative row, ascending
  std::vector<uint16_t> rowpid_dev;
  bool on_device = opt_int("GMG_PATCH_OP_DEVICE", 1) != 0 && S.d_pptr && S.d_pdofs && S.d_ublock && n + 1 < (int64_t)INT32_MAX;
  if (on_device) {
    const size_t mark = allocs.size();
    const int64_t bytes0 = dev_bytes;
    auto drop_scratch = [&]() { HIP_CHECK(hipStreamSynchronize(stream)); while (allocs.size() > mark) { (void)hipFree(allocs.back()); allocs.pop_back(); } dev_bytes = bytes0; };
    int32_t *d_shape = upload(shape);
    int32_t *d_iptr = dalloc<int32_t>((size_t)n + 2), *d_fill = dalloc<int32_t>((size_t)n + 1), *d_inc = dalloc<int32_t>((size_t)ne), *d_s2p = dalloc<int32_t>((size_t)ne);
    const int nb = (int)((n + kScanTile - 1) / kScanTile);
    int32_t *d_tot = dalloc<int32_t>((size_t)nb + 1);
    unsigned long long *d_hash = dalloc<unsigned long long>((size_t)n);
    HIP_CHECK(hipMemsetAsync(d_fill, 0, sizeof(int32_t) * ((size_t)n + 1), stream));
    const dim3 gs((unsigned)std::min<int64_t>((ne + 255) / 256, 1 << 20)), gn((unsigned)std::min<int64_t>((n + 255) / 256, 1 << 20)), b256(256);
    hipLaunchKernelGGL(slot_count_kernel, gs, b256, 0, stream, ne, S.d_pdofs, d_fill);
    hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)nb), dim3(kScanBlock), 0, stream, n, d_fill, d_iptr, d_tot);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1024), 0, stream, nb, d_tot);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(kScanBlock), 0, stream, n, d_iptr, d_tot, (int32_t)ne);
    HIP_CHECK(hipMemsetAsync(d_fill, 0, sizeof(int32_t) * ((size_t)n + 1), stream));
    hipLaunchKernelGGL(slot_scatter_kernel, gs, b256, 0, stream, ne, S.d_pdofs, d_iptr, d_fill, d_inc);
    hipLaunchKernelGGL(slot_sort_kernel, gn, b256, 0, stream, n, d_iptr, d_inc);
    hipLaunchKernelGGL(slot_patch_kernel, dim3((unsigned)std::min<int64_t>((npatch + 255) / 256, 1 << 20)), b256, 0, stream, npatch, S.d_pptr, d_s2p);
    hipLaunchKernelGGL(row_sig_hash_kernel, gn, b256, 0, stream, n, d_iptr, d_inc, d_s2p, S.d_pptr, S.d_pdofs, S.d_ublock, d_shape, d_hash);
    HIP_CHECK(hipGetLastError());
    std::vector<unsigned long long> hash((size_t)n);
    HIP_CHECK(hipMemcpyAsync(hash.data(), d_hash, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    sub("slot lists + row hashes (device)");
    // distinct hashes and the first row that carries each: per chunk, then merged
    const int TN = (int)std::max<int64_t>(1, std::min<int64_t>(64, n / (1 << 18)));
    const int64_t pern = (n + TN - 1) / TN;
    std::vector<std::vector<std::pair<unsigned long long, int64_t>>> found((size_t)TN);
    std::atomic<bool> over(false);
    parallel_chunks(TN, [&](int64_t t) {
      std::unordered_map<unsigned long long, int64_t> first;
      unsigned long long last_h = 0; bool have_last = false;
      for (int64_t i = t * pern; i < std::min(n, (t + 1) * pern); ++i) {
        const unsigned long long h = hash[(size_t)i];
        if (have_last && h == last_h) continue;
        last_h = h; have_last = true;
        if (first.emplace(h, i).second && first.size() > 8192) { over.store(true); return; }
      }
      found[(size_t)t].assign(first.begin(), first.end());
    });
    std::unordered_map<unsigned long long, int64_t> groups;
    if (!over.load())
      for (auto &f : found)
        for (auto &kv : f) { auto it = groups.find(kv.first); if (it == groups.end()) groups.emplace(kv.first, kv.second); else it->second = std::min(it->second, kv.second); }
    if (over.load() || groups.size() > 4096) { drop_scratch(); return; }
    std::vector<std::pair<int64_t, unsigned long long>> byrow;
    for (auto &kv : groups) byrow.emplace_back(kv.second, kv.first);
    std::sort(byrow.begin(), byrow.end());                    // ids by first appearance
    const int ng = (int)byrow.size();
    std::vector<std::pair<unsigned long long, int32_t>> byhash((size_t)ng);
    for (int g = 0; g < ng; ++g) { byhash[(size_t)g] = {byrow[(size_t)g].second, (int32_t)g}; sig_rep.push_back(byrow[(size_t)g].first); }
    std::sort(byhash.begin(), byhash.end());
    std::vector<unsigned long long> ghash((size_t)ng);
    std::vector<int64_t> grep((size_t)ng);
    std::vector<int32_t> gid((size_t)ng);
    for (int g = 0; g < ng; ++g) { ghash[(size_t)g] = byhash[(size_t)g].first; gid[(size_t)g] = byhash[(size_t)g].second; grep[(size_t)g] = sig_rep[(size_t)gid[(size_t)g]]; }
    unsigned long long *d_ghash = upload(ghash);
    int64_t *d_grep = upload(grep);
    int32_t *d_gid = upload(gid);
    uint16_t *d_rowpid = dalloc<uint16_t>((size_t)n);
    int *d_nmis = dalloc<int>(1);
    HIP_CHECK(hipMemsetAsync(d_nmis, 0, sizeof(int), stream));
    hipLaunchKernelGGL(row_sig_assign_kernel, gn, b256, 0, stream, n, d_iptr, d_inc, d_s2p, S.d_pptr, S.d_pdofs, S.d_ublock, d_shape, d_hash, ng, d_ghash, d_grep,
                       d_gid, d_rowpid, d_nmis);
    HIP_CHECK(hipGetLastError());
    int nmis = 0;
    HIP_CHECK(hipMemcpyAsync(&nmis, d_nmis, sizeof(int), hipMemcpyDeviceToHost, stream));
    rowpid_dev.resize((size_t)n);
    HIP_CHECK(hipMemcpyAsync(rowpid_dev.data(), d_rowpid, sizeof(uint16_t) * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    if (nmis != 0) { rowpid_dev.clear(); sig_rep.clear(); on_device = false; }   // a 64-bit hash collision: the exact host path decides
    else {
      // the slot lists of the representative rows (a few dozen short copies)
      rep_slots.resize((size_t)ng);
      for (int g = 0; g < ng; ++g) {
        int32_t be[2];
        HIP_CHECK(hipMemcpy(be, d_iptr + sig_rep[(size_t)g], sizeof(be), hipMemcpyDeviceToHost));
        rep_slots[(size_t)g].resize((size_t)(be[1] - be[0]));
        if (be[1] > be[0]) HIP_CHECK(hipMemcpy(rep_slots[(size_t)g].data(), d_inc + be[0], sizeof(int32_t) * (size_t)(be[1] - be[0]), hipMemcpyDeviceToHost));
      }
    }
    drop_scratch();
    sub("row signatures: groups + exact check (device)");
  }
  if (!on_device) {
  // slot -> patch, dof -> slots in ascending patch order
  s2p.assign((size_t)ne, 0);
  parallel_for(npatch, [&](int64_t p) { for (int64_t q = T.pptr[p]; q < T.pptr[p + 1]; ++q) s2p[(size_t)q] = (int32_t)p; });
  // (counting sort over 4.6e8 slots at 256^3 Q2: counted and scattered by all cores with relaxed atomic increments, then every
  // dof's short list is sorted back into ascending slot = patch order -- the result does not depend on the interleaving)
  iptr.assign((size_t)n + 1, 0);
  inc.assign((size_t)ne, 0);
  {
    const int64_t TQ = std::max<int64_t>(1, std::min<int64_t>(64, ne / (1 << 20)));
    const int64_t perq = (ne + TQ - 1) / TQ;
    parallel_chunks(TQ, [&](int64_t t) {
      for (int64_t q = t * perq; q < std::min(ne, (t + 1) * perq); ++q) __atomic_fetch_add(&iptr[(size_t)T.prow[(size_t)q] + 1], (int64_t)1, __ATOMIC_RELAXED);
    });
    for (int64_t i = 0; i < n; ++i) iptr[(size_t)i + 1] += iptr[(size_t)i];
    std::vector<int64_t> fill(iptr.begin(), iptr.end() - 1);
    parallel_chunks(TQ, [&](int64_t t) {
      for (int64_t q = t * perq; q < std::min(ne, (t + 1) * perq); ++q)
        inc[(size_t)__atomic_fetch_add(&fill[(size_t)T.prow[(size_t)q]], (int64_t)1, __ATOMIC_RELAXED)] = (int32_t)q;
    });
    const int64_t TD = std::max<int64_t>(1, std::min<int64_t>(64, n / (1 << 18)));
    const int64_t perd = (n + TD - 1) / TD;
    parallel_chunks(TD, [&](int64_t t) {
      for (int64_t i = t * perd; i < std::min(n, (t + 1) * perd); ++i) {
        int32_t *lo = inc.data() + iptr[(size_t)i], *hi = inc.data() + iptr[(size_t)i + 1];
        if (hi - lo > 1) std::sort(lo, hi);
      }
    });
  }
  sub("slot lists (counting sort)");
  // row signatures: (block, local row, shape, first dof - row) of every patch of the row, in patch order
  auto sig_len = [&](int64_t i) { return (int)(4 * (iptr[(size_t)i + 1] - iptr[(size_t)i])); };
  auto sig_word = [&](int64_t i, int j) -> int64_t {
    const int64_t q = inc[(size_t)(iptr[(size_t)i] + (j >> 2))];
    const int64_t p = s2p[(size_t)q];
    switch (j & 3) {
    case 0: return S.h_ublock[(size_t)p];
    case 1: return q - T.pptr[p];
    case 2: return shape[(size_t)p];
    default: return (int64_t)T.prow[(size_t)T.pptr[p]] - i;
    }
  };
  if (!sequence_ids(n, sig_len, sig_word, sig, sig_rep, 4096)) return;
  sub("row signatures");
    rep_slots.resize(sig_rep.size());
    for (size_t g = 0; g < sig_rep.size(); ++g) {
      const int64_t i = sig_rep[g];
      rep_slots[g].assign(inc.begin() + iptr[(size_t)i], inc.begin() + iptr[(size_t)i + 1]);
    }
  }
  // one merged row per signature
  auto S_ = std::make_shared<PatStream>();
  PatStream &P = *S_;
  P.mode = 0; P.nrows = n; P.ncols = n;
  for (size_t sgi = 0; sgi < sig_rep.size(); ++sgi) {
    const int64_t i = sig_rep[sgi];
    std::vector<std::pair<int64_t, double>> ent;             // (column offset, value) in patch order
    for (const int32_t qs : rep_slots[sgi]) {
      const int64_t q = qs, p = (int64_t)(std::upper_bound(T.pptr.begin(), T.pptr.end(), q) - T.pptr.begin()) - 1;
      const int64_t np = T.pptr[p + 1] - T.pptr[p], li = q - T.pptr[p];
      const double *blk = ubinv.data() + S.h_uboff[(size_t)S.h_ublock[(size_t)p]];
      for (int64_t j = 0; j < np; ++j) ent.emplace_back((int64_t)T.prow[(size_t)(T.pptr[p] + j)] - i, blk[li * np + j]);
    }
    std::stable_sort(ent.begin(), ent.end(), [](const std::pair<int64_t, double> &a, const std::pair<int64_t, double> &b) { return a.first < b.first; });
    P.start.push_back((int32_t)P.off.size());
    int32_t len = 0;
    for (size_t e = 0; e < ent.size();) {
      double v = ent[e].second;
      size_t f = e + 1;
      for (; f < ent.size() && ent[f].first == ent[e].first; ++f) v = v + ent[f].second;     // patch order
      uint64_t bits;
      std::memcpy(&bits, &v, 8);
      P.off.push_back((int32_t)ent[e].first);
      P.val.push_back(bits);
      ++len;
      e = f;
    }
    if (len > 1024) return;
    P.len.push_back(len);
    P.wmax = std::max<int64_t>(P.wmax, len);
  }
  if (on_device) P.rowpid.swap(rowpid_dev);
  else {
    P.rowpid.resize((size_t)n);
    parallel_for(n, [&](int64_t i) { P.rowpid[(size_t)i] = (uint16_t)sig[(size_t)i]; });
  }
  {
    const int64_t TN = std::max<int64_t>(1, std::min<int64_t>(64, n / (1 << 18)));
    const int64_t pern = (n + TN - 1) / TN;
    std::vector<int64_t> part((size_t)TN, 0);
    parallel_chunks(TN, [&](int64_t t) {
      int64_t c = 0;
      for (int64_t i = t * pern; i < std::min(n, (t + 1) * pern); ++i) c += P.len[(size_t)P.rowpid[(size_t)i]];
      part[(size_t)t] = c;
    });
    for (int64_t c : part) P.nnz += c;
  }
  P.rows_seen = n;
  if (P.nnz <= 0) return;
  sub("merged rows + row ids");
  try {
    S.M = finish_stream(P, "patch operator");
  } catch (const GmgError &) {
    S.M = DevCSR();
    return;
  }
  sub("device form (finish_stream)");
  S.use_M = true;
  // the patch-solve path's buffers are not needed any more
  HIP_CHECK(hipStreamSynchronize(stream));
  release(S.d_contrib, (size_t)ne + 1);
  if (S.d_isoff) { release(S.d_isoff, (size_t)S.n_isoff); release(S.d_isinc, (size_t)S.n_isinc); }
  if (S.d_iptr) { release(S.d_iptr, (size_t)n + 1); release(S.d_inc, (size_t)ne); }
  if (opt_int("GMG_SETUP_TIMING", 0))
    std::fprintf(stderr, "[gmg_setup] patch operator: %lld rows, %zu shapes, %zu distinct rows, %lld nnz, %.1f ms\n", (long long)n, shape_rep.size(),
                 sig_rep.size(), (long long)P.nnz, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
}

// dx = (omega *) sum_p scatter(inv(A_pp) r_p) ; relax: x += dx
void gmg_solver::patch_precond(Level &L, Smoother &S, const double *r, double omega, bool relax, double *dx, double *x)
{
  if (S.use_M) {
    // dx = sum_p R_p^T inv(A_pp) R_p r as one pattern mat-vec; the relaxation dx .= omega .* dx ; x .+= dx
    // (RichardsonSmoothers.jl:92-93) rides in its epilogue
    if (relax && omega != 0.0) spmv_addto(S.M, r, dx, x, omega);
    else {
      spmv_set(S.M, r, dx);
      if (relax) hipLaunchKernelGGL(relax_update_kernel, dim3(grid_for(L.n)), dim3(256), 0, stream, L.n, omega, dx, x);
      else if (omega != 1.0) hipLaunchKernelGGL(scale_inplace_kernel, dim3(grid_for(L.n)), dim3(256), 0, stream, L.n, omega, dx);
      HIP_CHECK(hipGetLastError());
    }
    return;
  }
  if (S.npatch > 0) {
    if (S.dedup) {
      const int grid = (int)((S.npatch + kPatchChunk - 1) / kPatchChunk);
      hipLaunchKernelGGL(patch_apply_dedup_kernel, dim3(grid), dim3(kBlock), 0, stream, S.npatch, S.d_pptr, S.d_pdofs, S.d_ublock,
                         S.d_uboff, S.d_ubinv, r, S.d_contrib);
    } else if (S.max_np <= 64) {
      const int grid = (int)((S.npatch + 3) / 4);
      hipLaunchKernelGGL(patch_apply_kernel, dim3(grid), dim3(kBlock), 0, stream, S.npatch, S.d_pptr, S.d_pdofs, S.d_boff,
                         S.d_binv, r, S.d_contrib);
    } else {
      hipLaunchKernelGGL(patch_apply_big_kernel, dim3((int)S.npatch), dim3(kBlock), 0, stream, S.npatch, S.d_pptr, S.d_pdofs,
                         S.d_boff, S.d_binv, r, S.d_contrib);
    }
    HIP_CHECK(hipGetLastError());
  }
  // distributed level: the owned patches reach into ghost dofs.  consistent!(b) happened at the caller (see smooth()); here the
  // local contributions are gathered for ALL local dofs, the ghost ones are added to their owners (assemble!,
  // PatchSolvers.jl:251-258), then the relaxation runs on the owned entries.
  const int l = (int)(&L - &lev[0]);
  const bool dist = comm.nranks > 1 && L.halo.present && !L.halo.ovl;   // (overlapping layout: a single-GPU level between two exchanges)
  const int64_t ng = dist ? L.nvec : L.n;
  const int grid = (int)((ng + 255) / 256);
  const int fused_relax = (relax && !dist) ? 1 : 0;
  if (S.d_isoff)
    hipLaunchKernelGGL(patch_gather_sell_kernel, dim3(std::max(grid, 1)), dim3(256), 0, stream, ng, S.d_isoff, S.d_isinc, S.d_contrib,
                       omega, fused_relax, dx, x);
  else
  hipLaunchKernelGGL(patch_gather_kernel, dim3(std::max(grid, 1)), dim3(256), 0, stream, ng, S.d_iptr, S.d_inc, S.d_contrib,
                     omega, fused_relax, dx, x);
  HIP_CHECK(hipGetLastError());
  if (dist) {
    assemble_add(l, dx);
    if (relax) {
      hipLaunchKernelGGL(relax_update_kernel, dim3(grid_for(L.n)), dim3(256), 0, stream, L.n, omega, dx, x);
      HIP_CHECK(hipGetLastError());
    }
  }
}

// ----------------------------------------------------------------------------
// coarse solver: coarsest_solver = LUSolver() (GMGLinearSolvers.jl:54,423-434).
// The exact factorisation is done once on the host (banded LU with partial
// pivoting) and turned into a dense inverse so that the per-cycle solve is a
// single bandwidth-bound GEMV on the device.
// ----------------------------------------------------------------------------
namespace {
struct BandLU {
  int n = 0, kl = 0, ku = 0, ld = 0;
  std::vector<double> ab;
  std::vector<int> piv;
  double &at(int i, int j) { return ab[(size_t)(kl + ku + i - j) + (size_t)j * ld]; }
  double get(int i, int j) const { return ab[(size_t)(kl + ku + i - j) + (size_t)j * ld]; }

  bool factor(const HostCSR &A)
  {
    n = (int)A.nrows;
    for (int i = 0; i < n; ++i)
      for (int64_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k) {
        kl = std::max(kl, i - A.col[k]);
        ku = std::max(ku, A.col[k] - i);
      }
    ld = 2 * kl + ku + 1;
    ab.assign((size_t)ld * n, 0.0);
    piv.assign(n, 0);
    for (int i = 0; i < n; ++i)
      for (int64_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k) at(i, A.col[k]) += A.val[k];
    int ju = 0;
    for (int j = 0; j < n; ++j) {
      const int km = std::min(kl, n - 1 - j);
      int p = j;
      double best = std::fabs(get(j, j));
      for (int i = j + 1; i <= j + km; ++i)
        if (std::fabs(get(i, j)) > best) { best = std::fabs(get(i, j)); p = i; }
      if (best == 0.0) return false;
      piv[j] = p;
      ju = std::max(ju, std::min(p + ku, n - 1));
      ju = std::max(ju, j);
      if (p != j)
        for (int c = j; c <= ju; ++c) std::swap(at(j, c), at(p, c));
      const double inv = 1.0 / get(j, j);
      for (int i = j + 1; i <= j + km; ++i) at(i, j) *= inv;
      for (int c = j + 1; c <= ju; ++c) {
        const double f = get(j, c);
        if (f != 0.0)
          for (int i = j + 1; i <= j + km; ++i) at(i, c) -= get(i, j) * f;
      }
    }
    return true;
  }
  void solve(double *x) const
  {
    for (int j = 0; j < n; ++j) {
      const int km = std::min(kl, n - 1 - j);
      if (piv[j] != j) std::swap(x[j], x[piv[j]]);
      const double xj = x[j];
      if (xj != 0.0)
        for (int i = j + 1; i <= j + km; ++i) x[i] -= get(i, j) * xj;
    }
    const int kv = kl + ku;
    for (int j = n - 1; j >= 0; --j) {
      x[j] /= get(j, j);
      const double xj = x[j];
      for (int i = std::max(0, j - kv); i < j; ++i) x[i] -= get(i, j) * xj;
    }
  }
};
} // namespace

void gmg_solver::build_coarse()
{
  const int64_t n = lev[nlev - 1].n;
  REQUIRE(!(lev[nlev - 1].sA && coarse_eff == GMG_COARSE_DENSE_INVERSE), GMG_ERR_UNSUPPORTED,
          "the dense-inverse coarse solver needs the coarsest matrix whole (gmg_set_matrix), not streamed");
  if (coarse_eff == GMG_COARSE_DENSE_INVERSE) d_Ainv = build_dense_inverse(lev[nlev - 1].hA, "coarsest-level matrix");
  else if (coarse_eff == GMG_COARSE_CG_JACOBI) { cc_w = dvec(n); cc_p = dvec(n); cc_z = dvec(n); cc_r = dvec(n); }
  else {
    REQUIRE(coarse_fn, GMG_ERR_STATE, "coarse-solver callback missing");
    if (h_cr) { (void)hipHostFree(h_cr); (void)hipHostFree(h_cx); h_cr = h_cx = nullptr; }
    HIP_CHECK(hipHostMalloc((void **)&h_cr, sizeof(double) * (size_t)std::max<int64_t>(1, n)));
    HIP_CHECK(hipHostMalloc((void **)&h_cx, sizeof(double) * (size_t)std::max<int64_t>(1, n)));
  }
}

// LUSolver() on a small sparse matrix: row-major dense inverse on the device.
double *gmg_solver::build_dense_inverse(const HostCSR &A, const std::string &what)
{
  const int n = (int)A.nrows;
  // Small matrices: exact banded LU with partial pivoting on the host (O(n^2 * bandwidth)).
  // Large ones: blocked Gauss-Jordan on the device (no pivoting; result verified below).
  if (n > opt_int("GMG_COARSE_HOST_MAX", 1500)) {
    REQUIRE((double)n * n * 8.0 <= 64.0e9, GMG_ERR_UNSUPPORTED,
            what + " has " + std::to_string(n) + " dofs: its dense inverse would not fit; add multigrid levels");
    // the device inversion does not pivot and verifies its result: a matrix that needs pivoting is rejected there -- within
    // reach of the host's pivoted banded LU (seconds up to ~6000 dofs) take that instead of failing the setup
    if (n > opt_int("GMG_COARSE_HOST_FALLBACK_MAX", 6000)) return build_coarse_device(A, what);
    const size_t mark = allocs.size();
    const int64_t bytes0 = dev_bytes;
    try {
      return build_coarse_device(A, what);
    } catch (const GmgError &e) {
      // the scratch of the rejected inversion (n^2 result, padded work copy, panels, CSR upload) goes back before the host takes over
      HIP_CHECK(hipStreamSynchronize(stream));
      while (allocs.size() > mark) { (void)hipFree(allocs.back()); allocs.pop_back(); }
      dev_bytes = bytes0;
      if (e.code != GMG_ERR_SINGULAR) throw;
    }
  }
  BandLU lu;
  REQUIRE(lu.factor(A), GMG_ERR_SINGULAR, what + " is singular");
  std::vector<double> inv((size_t)n * n);
  const int nthreads = host_threads(std::max(1, n / 16));
  auto work = [&](int t) {
    std::vector<double> e(n);
    for (int c = t; c < n; c += nthreads) {
      std::fill(e.begin(), e.end(), 0.0);
      e[c] = 1.0;
      lu.solve(e.data());
      for (int i = 0; i < n; ++i) inv[(size_t)i * n + c] = e[i]; // row-major inverse
    }
  };
  run_threads(nthreads, work);
  double *d = upload(inv);
  HIP_CHECK(hipStreamSynchronize(stream));
  return d;
}

double *gmg_solver::build_coarse_device(const HostCSR &A, const std::string &what)
{
  const int n = (int)A.nrows;
  const size_t mark = allocs.size();
  const int64_t bytes0 = dev_bytes;
  // large levels: 64-wide panels on a padded leading dimension (half the passes over the matrix, aligned 128-byte row
  // segments), compacted into the n x n result at the end; small ones: 32-wide panels in place
  const bool wide = gj_mfma && n >= opt_int("GMG_GJ_WIDE_MIN", 4096);
  const int64_t lda = wide ? (((int64_t)n + 127) / 128) * 128 : n;        // whole tiles: the update kernel loads unconditionally
  const int64_t nr = wide ? (((int64_t)n + 127) / 128) * 128 : n;
  double *D = dalloc<double>((size_t)n * n);
  const int64_t bytesD = dev_bytes - bytes0;
  double *W = wide ? dalloc<double>((size_t)nr * lda) : D;
  HIP_CHECK(hipMemsetAsync(W, 0, sizeof(double) * (size_t)nr * lda, stream));
  int64_t *d_ptr = upload(A.ptr);
  int32_t *d_col = upload(A.col);
  double *d_val = upload(A.val);
  hipLaunchKernelGGL((densify_ld_kernel<int64_t>), dim3((n + 255) / 256), dim3(256), 0, stream, (int64_t)n, lda, d_ptr, d_col, d_val, W);
  HIP_CHECK(hipGetLastError());
  int *d_bad = dalloc<int>(1);
  HIP_CHECK(hipMemsetAsync(d_bad, 0, sizeof(int), stream));
  if (wide) {
    double *Pinv = dvec(GJ_W * GJ_W), *R = dvec((int64_t)GJ_W * lda), *C = dvec((int64_t)GJ_W * nr), *Cp = dvec((int64_t)GJ_W * nr);
    constexpr int GJ_RT = 4;                                 // 64 x 64 per wave, 128 x 128 per workgroup
    const int64_t nst = (int64_t)(((n + 127) / 128 + GJ_SX - 1) / GJ_SX) * ((n + 1023) / 1024);
    const dim3 tiles((unsigned)(((nst + 7) / 8) * 8 * GJ_SX * (1024 / (32 * GJ_RT))));
    for (int k0 = 0; k0 < n; k0 += GJ_W) {
      const int b = std::min(GJ_W, n - k0);
      hipLaunchKernelGGL(gj_diag64_kernel, dim3(1), dim3(1024), 0, stream, n, lda, k0, b, W, Pinv, d_bad);
      hipLaunchKernelGGL(gj_panel_rows_kernel, dim3((n + 63) / 64, 4), dim3(64), 0, stream, n, lda, k0, b, W, Pinv, R);
      hipLaunchKernelGGL(gj_panel_cols_kernel, dim3((n + 63) / 64), dim3(256), 0, stream, n, lda, k0, b, W, Pinv, C, Cp);
      hipLaunchKernelGGL((gj_update64_kernel<GJ_RT>), tiles, dim3(256), 0, stream, n, lda, k0, b, W, Pinv, R, C, Cp);
      HIP_CHECK(hipGetLastError());
    }
    HIP_CHECK(hipMemcpy2DAsync(D, sizeof(double) * (size_t)n, W, sizeof(double) * (size_t)lda, sizeof(double) * (size_t)n, (size_t)n, hipMemcpyDeviceToDevice, stream));
  } else {
    double *Pinv = dvec(GJ_B * GJ_B), *R = dvec((int64_t)GJ_B * n), *C = dvec((int64_t)GJ_B * n), *Cp = dvec((int64_t)GJ_B * n);
    const dim3 tiles((n + 63) / 64, (n + 63) / 64);
    for (int k0 = 0; k0 < n; k0 += GJ_B) {
      const int b = std::min(GJ_B, n - k0);
      hipLaunchKernelGGL(gj_diag_kernel, dim3(1), dim3(GJ_B * GJ_B), 0, stream, n, k0, b, D, Pinv, d_bad);
      hipLaunchKernelGGL(gj_panels_kernel, dim3((unsigned)(((int64_t)n * b + 255) / 256)), dim3(256), 0, stream, n, k0, b, D, Pinv, R, C, Cp);
      if (gj_mfma) hipLaunchKernelGGL(gj_update_mfma_kernel, tiles, dim3(256), 0, stream, n, k0, b, D, Pinv, R, C, Cp);
      else hipLaunchKernelGGL(gj_update_kernel, tiles, dim3(256), 0, stream, n, k0, b, D, Pinv, R, C, Cp);
      HIP_CHECK(hipGetLastError());
    }
  }
  int bad = 0;
  HIP_CHECK(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  REQUIRE(bad == 0, GMG_ERR_SINGULAR, what + " needs pivoting (zero pivot in the device inversion); add multigrid levels");
  // verify: A*(Ainv*v) == v for a deterministic vector (host sparse product, device GEMV)
  std::vector<double> v((size_t)n), w((size_t)n);
  for (int i = 0; i < n; ++i) v[i] = 1.0 + 0.5 * std::sin(0.37 * i);
  double *dv = upload(v), *dw = dvec(n);
  const int grid = (n + 3) / 4;
  hipLaunchKernelGGL(dense_gemv_kernel, dim3(grid), dim3(kBlock), 0, stream, n, D, dv, dw);
  HIP_CHECK(hipGetLastError());
  HIP_CHECK(hipMemcpyAsync(w.data(), dw, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream));
  HIP_CHECK(hipStreamSynchronize(stream));
  double err = 0.0, nv = 0.0;
  for (int i = 0; i < n; ++i) {
    double sAw = 0.0;
    for (int64_t k = A.ptr[i]; k < A.ptr[i + 1]; ++k) sAw += A.val[k] * w[A.col[k]];
    err += (sAw - v[i]) * (sAw - v[i]);
    nv += v[i] * v[i];
  }
  REQUIRE(std::isfinite(err) && std::sqrt(err / nv) < 1.0e-8, GMG_ERR_SINGULAR,
          "device inversion of the " + what + " is inaccurate (matrix needs pivoting); add multigrid levels");
  // panels, the CSR copy and the check vectors were setup scratch
  for (size_t i = allocs.size(); i-- > mark;)
    if (allocs[i] != (void *)D) { (void)hipFree(allocs[i]); allocs.erase(allocs.begin() + (long)i); }
  dev_bytes = bytes0 + bytesD;
  return D;
}

// numerical_setup(ss::GMGSymbolicSetup,mat): GMGLinearSolvers.jl:183-210
void gmg_solver::setup()
{
  HIP_CHECK(hipSetDevice(device));
  free_all();
  read_tuning();
  resolve_coarse_kind();
  // GMG_SETUP_TIMING=1: per-phase wall times of the numerical setup on stderr
  const bool timing = opt_int("GMG_SETUP_TIMING", 0) != 0;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what, int l) {
    if (!timing) return;
    (void)hipStreamSynchronize(stream);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[gmg_setup] level %d %-28s %8.1f ms\n", l, what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // patch smoothers / patch corrections with patches of more than 64 dofs take their blocks from the level's CSR (one thread block
  // per patch): a level kept in row-pattern form only gets its rows back
  for (int l = 0; l < nlev - 1; ++l) {
    Level &L = lev[l];
    if (!(L.sA && L.sA->complete())) continue;
    int64_t big = 0;
    for (const Smoother *sp : {&L.pre, &L.post, &L.pcorr})
      if (sp->kind == SM_PATCH && sp->tab && !sp->tab->has_blocks)
        for (size_t q = 0; q + 1 < sp->tab->pptr.size(); ++q) big = std::max(big, sp->tab->pptr[q + 1] - sp->tab->pptr[q]);
    if (big > 64) rows_from_stream(L);
  }
  if (comm.nranks > 1) {
    // Row-pattern-only operators (streamed with gmg_set_operator_rows, or handed over whole before the communicator was initialised)
    // stay as they are on levels that are laid out like a single-GPU level: the overlapping layout (one square local operator over the
    // extended box) and the replicated levels.  An own | ghost level splits own / ghost columns on the CSR: its matrix gets its rows
    // back.  Transfers are applied whole (the vectors carry their ghost space), so their pattern form is kept everywhere.
    for (int l = 0; l < nlev; ++l) {
      Level &L = lev[l];
      const bool own_ghost = L.halo.present && !L.halo.ovl;
      if (own_ghost && L.sA && L.sA->complete() && !L.sA_split) rows_from_stream(L);
    }
  }
  if (redist.present) {
    REQUIRE(comm.nranks > 1 && rep_from >= 2 && sub_from >= 1 && sub_from < rep_from, GMG_ERR_STATE,
            "gmg_set_redistribution: the subset levels lie between level 1 and the first replicated level (gmg_set_replication first)");
    REQUIRE(!lev[sub_from - 1].has_pcorr, GMG_ERR_UNSUPPORTED, "no patch-corrected prolongation across a redistribution");
  }
  for (int l = 0; l < nlev; ++l) {
    Level &L = lev[l];
    if (inactive(l)) {                                       // a level of a rank subset this rank is not part of
      REQUIRE(!L.hasA && !L.halo.present, GMG_ERR_INVALID, "this rank is not a member of the subset that holds level " + std::to_string(l));
      L.n = L.nvec = 0;
      continue;
    }
    REQUIRE(L.hasA, GMG_ERR_STATE, "gmg_set_matrix missing for level " + std::to_string(l));
    L.n = L.hA.nrows;
    L.nvec = L.hA.ncols;
    const bool replicated = comm.nranks > 1 && rep_from >= 0 && l >= rep_from;
    if (L.halo.present) {
      REQUIRE(comm.nranks > 1, GMG_ERR_STATE, "gmg_set_partition needs gmg_comm_init_* first");
      REQUIRE(!replicated, GMG_ERR_INVALID, "replicated levels take global operators, not a partition");
      if (L.halo.ovl) {
        REQUIRE(l >= 1 || has_outer(), GMG_ERR_UNSUPPORTED,
                "finest level in the overlapping layout: the Krylov operator must be given in the own | ghost layout as well "
                "(gmg_set_matrix / gmg_set_partition with GMG_LEVEL_KRYLOV, gmg_set_krylov_map)");
        REQUIRE(L.hA.nrows == L.hA.ncols && L.hA.nrows == L.halo.n_own + L.halo.n_ghost, GMG_ERR_INVALID,
                "overlapping layout: the local matrix must be square over all local entries on level " + std::to_string(l));
        // patch smoothers: the caller lists every patch whose dofs all lie in the local box and the blocks A[p,p] come from the local
        // matrix (exact there) -- no assemble!; patches with their own matrices / separate column tables keep the own | ghost layout
        for (const Smoother *sp : {&L.pre, &L.post})
          REQUIRE(l == nlev - 1 || sp->kind == SM_JACOBI || (sp->tab && sp->tab->pcol.empty() && !sp->tab->has_blocks), GMG_ERR_UNSUPPORTED,
                  "overlapping layout: patch smoothers take their blocks from the local matrix (gmg_set_smoother_patch), patch_cols = patch_rows");
        // patch-corrected prolongation (round 5): the correction patches a rank lists are the ones inside its local entries, blocks from
        // the local matrix; dxh = P dxH is made consistent before the rhs form is applied and again before r -= A dxh (cycle())
        REQUIRE(!L.has_pcorr || (L.pcorr.tab && !L.pcorr.tab->has_blocks), GMG_ERR_UNSUPPORTED,
                "overlapping layout: the patch-corrected prolongation takes its blocks from the local matrix");
      } else
      REQUIRE(L.halo.n_own == L.hA.nrows && L.halo.n_own + L.halo.n_ghost == L.hA.ncols, GMG_ERR_INVALID,
              "local matrix must be n_own x (n_own+n_ghost) on level " + std::to_string(l));
    } else {
      REQUIRE(comm.nranks == 1 || replicated, GMG_ERR_STATE, "gmg_set_partition missing on level " + std::to_string(l));
      REQUIRE(L.hA.nrows == L.hA.ncols, GMG_ERR_INVALID, "level matrix must be square");
    }
  }
  if (comm.nranks > 1)
    REQUIRE(rep_from >= 1 && rep_from <= nlev - 1, GMG_ERR_STATE,
            "distributed runs need gmg_set_replication: at least the coarsest level must be replicated");
  if (has_outer()) {
    // separate Krylov operator (GMG_LEVEL_KRYLOV): own | ghost layout over the caller's vectors, level 0 in the overlapping layout
    Level &K = lev[nlev];
    K.n = K.hA.nrows; K.nvec = K.hA.ncols;
    REQUIRE(comm.nranks > 1 && lev[0].halo.present && lev[0].halo.ovl, GMG_ERR_STATE,
            "a separate Krylov operator goes with a finest level in the overlapping layout (gmg_set_partition_overlap on level 0)");
    REQUIRE(K.halo.present && !K.halo.ovl, GMG_ERR_STATE, "gmg_set_partition(h, GMG_LEVEL_KRYLOV, ...) missing");
    REQUIRE(K.halo.n_own == K.hA.nrows && K.halo.n_own + K.halo.n_ghost == K.hA.ncols, GMG_ERR_INVALID,
            "Krylov operator: the local matrix must be n_own x (n_own + n_ghost)");
    REQUIRE((int64_t)h_own2loc.size() == K.n && K.n == lev[0].halo.n_own, GMG_ERR_INVALID,
            "gmg_set_krylov_map: one local id of level 0 per owned entry");
    for (int64_t q : h_own2loc) REQUIRE(q >= 0 && q < lev[0].n, GMG_ERR_INVALID, "gmg_set_krylov_map: local id out of range");
    REQUIRE(mode == GMG_MODE_PRECONDITIONER && log.maxiter == 1, GMG_ERR_UNSUPPORTED,
            "finest level in the overlapping layout: the GMG runs as a preconditioner with maxiter = 1");
  } else
    REQUIRE((int)lev.size() <= nlev || !lev[nlev].halo.present, GMG_ERR_STATE, "gmg_set_partition(GMG_LEVEL_KRYLOV) without gmg_set_matrix(GMG_LEVEL_KRYLOV)");
  for (int l = 0; l < nlev - 1; ++l) {
    Level &L = lev[l];
    if (inactive(l)) continue;
    REQUIRE(L.hasP, GMG_ERR_STATE, "gmg_set_prolongation missing for level " + std::to_string(l));
    const bool to_glue = redist.present && l + 1 == sub_from;   // P / R against the glued partition of level l+1 (own | ghost numbering)
    REQUIRE(L.hP.nrows == L.n && L.hP.ncols == (to_glue ? redist.n_glue_own + redist.n_glue_ghost : lev[l + 1].nvec), GMG_ERR_INVALID,
            "prolongation shape does not match level sizes (GMGLinearSolvers.jl:59-61)");
    if (comm.nranks > 1)
      REQUIRE(L.hasR, GMG_ERR_STATE, "distributed runs need the local rows of R (gmg_set_restriction): a local P^T misses off-rank rows");
    if (L.hasR) {
      const bool boundary = comm.nranks > 1 && l + 1 == rep_from;   // R yields this rank's rows of the replicated level
      REQUIRE((boundary ? L.hR.nrows == (int64_t)h_rep_gid.size() : to_glue ? L.hR.nrows == redist.n_glue_own : L.hR.nrows == lev[l + 1].n) && L.hR.ncols == L.nvec,
              GMG_ERR_INVALID, "restriction shape mismatch");
    }
    if (comm.nranks > 1 && L.halo.present && !L.halo.ovl)
      for (const Smoother *sp : {&L.pre, &L.post})
        REQUIRE(sp->kind == SM_JACOBI || (sp->tab && sp->tab->has_blocks), GMG_ERR_UNSUPPORTED,
                "distributed patch smoothers need the caller's patch matrices (gmg_set_smoother_patch_matrices): a rank's local matrix "
                "holds the owned rows only, the blocks of patches reaching into ghost dofs cannot be gathered from it");
  }
  if (comm.nranks > 1) {
    one_gather_sweep = 1;   // the one-gather sweep needs only s-ghosts
    for (int64_t g : h_rep_gid) REQUIRE(g >= 0 && g < lev[rep_from].n, GMG_ERR_INVALID, "replication: global id out of range");
  }
  init_reductions();
  for (int l = 0; l < nlev + (has_outer() ? 1 : 0); ++l) {   // (index nlev: the separate Krylov operator -- matrix and exchange plan only)
    Level &L = lev[l];
    if (l < nlev && inactive(l)) continue;
    if (L.halo.present && comm.nranks > 1 && L.halo.ovl) {
      // overlapping layout: one square local operator over all local entries -- laid out like a single-GPU level (row patterns,
      // shared offsets, one-launch smoothing passes all apply); ghost rows are recomputed redundantly between exchanges
      L.split = false; L.nbnd = 0;
      L.A = L.sA ? finish_stream(*L.sA, "level matrix") : upload_csr(L.hA);
    } else if (L.halo.present && comm.nranks > 1) {
      REQUIRE(!L.sA || L.sA_split, GMG_ERR_UNSUPPORTED, "own | ghost levels hold their matrix as CSR or as a stream split by gmg_set_operator_rows");
      // own x own / own x ghost split: A keeps the owned columns, the ghost columns of the rows
      // that have any go to a small CSR applied after the halo has arrived (finish_ghost)
      HostCSR loc;
      loc.nrows = L.hA.nrows; loc.ncols = L.hA.ncols;
      std::vector<int32_t> brows, bcol;
      std::vector<int64_t> bptr(1, 0);
      std::vector<double> bval;
      if (L.sA) {                                            // streamed: gmg_set_operator_rows has split every block already
        REQUIRE(L.sA->complete(), GMG_ERR_STATE, "row stream incomplete");
        brows = L.g_rows; bcol = L.g_col; bptr = L.g_ptr; bval = L.g_val;
      } else {
      loc.ptr.assign((size_t)L.n + 1, 0);
      for (int64_t i = 0; i < L.n; ++i) {
        bool any = false;
        for (int64_t k = L.hA.ptr[i]; k < L.hA.ptr[i + 1]; ++k) {
          if (L.hA.col[k] < L.n) { loc.col.push_back(L.hA.col[k]); loc.val.push_back(L.hA.val[k]); }
          else { bcol.push_back(L.hA.col[k]); bval.push_back(L.hA.val[k]); any = true; }
        }
        loc.ptr[i + 1] = (int64_t)loc.col.size();
        if (any) { brows.push_back((int32_t)i); bptr.push_back((int64_t)bcol.size()); }
      }
      }
      L.split = true;
      L.nbnd = (int64_t)brows.size();
      L.gh_rows = upload(brows); L.gh_ptr = upload(bptr); L.gh_col = upload(bcol); L.gh_val = upload(bval);
      if (opt_int("GMG_HALO_FIX_SELL", 1) && !brows.empty()) {
        // slices of 64 boundary rows, column-major, one byte per entry into a dictionary when the values allow: ONE builder for this
        // level's fix-up and the split restriction's (build_ghost_fix) -- ghost_fix_sell_kernel reads both
        GhostFix gf;
        build_ghost_fix(gf, brows, bptr, bcol, bval);
        L.gh_len = gf.len; L.gh_soff = gf.soff; L.gh_scol = gf.scol; L.gh_scode = gf.scode; L.gh_dict = gf.dict; L.gh_sval = gf.sval;
      }
      {   // send slots of every boundary row (fused pack): valid when every sent row is a boundary row
        HaloPlan &Hp = L.halo;
        std::vector<int32_t> bidx((size_t)L.n, -1);
        for (size_t q = 0; q < brows.size(); ++q) bidx[brows[q]] = (int32_t)q;
        bool ok = halo_fuse_pack != 0 && Hp.nsend() < (int64_t)INT32_MAX;
        std::vector<int64_t> pkp(brows.size() + 1, 0);
        for (int64_t sidx = 0; sidx < Hp.nsend() && ok; ++sidx) {
          const int64_t row = Hp.h_snd_idx[sidx];
          if (row < 0 || row >= L.n || bidx[row] < 0) ok = false;
          else pkp[(size_t)bidx[row] + 1]++;
        }
        if (ok && !brows.empty()) {
          for (size_t q = 0; q < brows.size(); ++q) pkp[q + 1] += pkp[q];
          std::vector<int32_t> slot((size_t)Hp.nsend());
          std::vector<int64_t> fill(pkp.begin(), pkp.end() - 1);
          for (int64_t sidx = 0; sidx < Hp.nsend(); ++sidx) slot[(size_t)fill[bidx[Hp.h_snd_idx[sidx]]]++] = (int32_t)sidx;
          Hp.d_pk_ptr = upload(pkp);
          Hp.d_pk_slot = upload(slot);
        } else { Hp.d_pk_ptr = nullptr; Hp.d_pk_slot = nullptr; }
        if (timing) std::fprintf(stderr, "[gmg_setup] level %d halo: %lld boundary rows, %lld send slots, pack fused into the fix-up: %s\n", l,
                                 (long long)brows.size(), (long long)Hp.nsend(), Hp.d_pk_ptr ? "yes" : "no");
      }
      if (L.sA) {
        L.A = finish_stream(*L.sA, "level matrix (own x own part)");
        L.A.nnz_model = L.sA->nnz + (int64_t)bcol.size();
      } else {
      L.A = upload_csr(loc);
      L.A.nnz_model = L.hA.nnz();
      }
    } else if (L.sA) L.A = finish_stream(*L.sA, "level matrix");
    else
    L.A = upload_csr(L.hA);                                 // :185 gmg_compute_matrices
    lap("A: upload + layout", l);
    if (l < nlev) L.rbuf[0] = dvec(L.nvec);                 // :187,188 rh / rH
    if (l > 0 ? l < nlev : has_outer()) L.x = dvec(L.nvec); // :188 dxH (level 0: only when the Krylov vectors live in another numbering)
    if (L.halo.present) {
      alloc_plan_buffers(L.halo);
    }
    if (l < nlev - 1) {
      L.rbuf[1] = dvec(L.nvec);
      L.dx = dvec(L.nvec);
      if (one_gather()) { L.sbuf[0] = dvec(L.nvec); L.sbuf[1] = dvec(L.nvec); }                                     // :188 dxh (Adxh is fused away)
      L.P = L.sP ? finish_stream(*L.sP, "prolongation") : upload_csr(L.hP);
      if (L.hasR && !L.sR && comm.nranks > 1 && L.halo.present && !L.halo.ovl && opt_int("GMG_HALO_SPLIT_R", 1)) {
        // own | ghost level: R's own columns (all rows: row-pattern layouts apply) + the ghost columns of the rows that have any
        HostCSR loc;
        loc.nrows = L.hR.nrows; loc.ncols = L.hR.ncols;
        loc.ptr.assign((size_t)L.hR.nrows + 1, 0);
        std::vector<int32_t> brows, bcol;
        std::vector<int64_t> bptr(1, 0);
        std::vector<double> bval;
        loc.col.reserve((size_t)L.hR.nnz()); loc.val.reserve((size_t)L.hR.nnz());   // (the ghost part is a boundary layer: it grows on demand)
        for (int64_t i = 0; i < L.hR.nrows; ++i) {
          bool any = false;
          for (int64_t k = L.hR.ptr[i]; k < L.hR.ptr[i + 1]; ++k) {
            if (L.hR.col[k] < L.n) { loc.col.push_back(L.hR.col[k]); loc.val.push_back(L.hR.val[k]); }
            else { bcol.push_back(L.hR.col[k]); bval.push_back(L.hR.val[k]); any = true; }
          }
          loc.ptr[i + 1] = (int64_t)loc.col.size();
          if (any) { brows.push_back((int32_t)i); bptr.push_back((int64_t)bcol.size()); }
        }
        L.R = upload_csr(loc);
        L.R.nnz_model = L.hR.nnz();
        build_ghost_fix(L.rfix, brows, bptr, bcol, bval);
        L.r_split = true;
      } else
      if (L.hasR) L.R = L.sR ? finish_stream(*L.sR, "restriction") : upload_csr(L.hR);
      else {
        HostCSR Rt = L.sP ? transpose(expand_stream(*L.sP)) : transpose(L.hP);   // R = P^T, GridTransferOperators.jl:536-547
        L.R = upload_csr(Rt);
      }
      lap("P, R: upload + layout", l);
      // :189-190 smoother caches: inv_diag = 1 ./ diag(A) (JacobiLinearSolvers.jl:20-23)
      int nzero = 0;
      L.dinv = build_inv_diag(L.A, nzero);
      const bool need_diag = (L.pre.kind == SM_JACOBI) || (L.post.kind == SM_JACOBI);
      REQUIRE(!(need_diag && nzero > 0), GMG_ERR_SINGULAR, "zero diagonal entry on level " + std::to_string(l));
      if (L.has_pcorr) {
        if (comm.nranks > 1 && L.halo.present && !L.halo.ovl) {
          // distributed (PatchTransferOperators.jl:153-172 on PVectors): the correction patches (coarse-cell interiors) must lie
          // inside the owned dofs -- their blocks are then rows / columns of the own x own part of the local matrix
          REQUIRE(L.pcorr.tab && !L.pcorr.tab->has_blocks, GMG_ERR_UNSUPPORTED, "distributed patch prolongation: blocks come from the level matrix");
          for (int32_t q : L.pcorr.tab->prow) REQUIRE(q < L.n, GMG_ERR_UNSUPPORTED, "distributed patch prolongation: every patch dof must be owned by this rank");
          for (int32_t q : L.pcorr.tab->pcol) REQUIRE(q < L.n, GMG_ERR_UNSUPPORTED, "distributed patch prolongation: every patch dof must be owned by this rank");
        }
        build_patch(L, L.pcorr);
        if (L.hasG) {
          REQUIRE(L.hG.nrows == L.n && L.hG.ncols == L.nvec, GMG_ERR_INVALID, "rhs operator of the prolongation correction has the wrong shape");
          L.G = upload_csr(L.hG);
          drop_csr_stream(L.G);
        }
        L.ptmp = dvec(L.nvec); L.pcor = dvec(L.nvec);
      }
      if (L.pre.kind == SM_PATCH) build_patch(L, L.pre);
      if (L.post_shares_pre) L.post = L.pre;
      else if (L.post.kind == SM_PATCH) build_patch(L, L.post);
      drop_csr_stream(L.P);
      drop_csr_stream(L.R);
      lap("D^-1, patch blocks", l);
    }
    if (l == nlev - 1 && coarse_eff == GMG_COARSE_CG_JACOBI) {
      int nzero = 0;
      L.dinv = build_inv_diag(L.A, nzero);                  // JacobiLinearSolvers.jl:20-23 on the coarsest matrix
      REQUIRE(nzero == 0, GMG_ERR_SINGULAR, "zero diagonal entry on the coarsest level");
    }
    drop_csr_stream(L.A);
  }
  build_coarse();                                           // :195 gmg_coarse_solver_caches
  lap("coarse inverse", nlev - 1);
  const int64_t n0 = lev[kl()].nvec;
  if (redist.present) {
    const int64_t nsub = lev[sub_from].nvec, nglue = redist.n_glue_own + redist.n_glue_ghost;
    const int64_t src_n[2] = {redist.n_glue_own, nsub}, dst_n[2] = {nsub, nglue};
    HaloPlan *plans[2] = {&redist.to_sub, &redist.from_sub};
    for (int k = 0; k < 2; ++k) {
      for (int64_t q : plans[k]->h_snd_idx) REQUIRE(q >= 0 && q < src_n[k], GMG_ERR_INVALID, "redistribution plan: sent id out of range");
      for (int64_t q : plans[k]->h_rcv_idx) REQUIRE(q >= 0 && q < dst_n[k], GMG_ERR_INVALID, "redistribution plan: received id out of range");
      for (int64_t q : redist.h_self[2 * k]) REQUIRE(q >= 0 && q < src_n[k], GMG_ERR_INVALID, "redistribution plan: local id out of range");
      for (int64_t q : redist.h_self[2 * k + 1]) REQUIRE(q >= 0 && q < dst_n[k], GMG_ERR_INVALID, "redistribution plan: local id out of range");
      alloc_plan_buffers(*plans[k]);
    }
    for (int k = 0; k < 4; ++k) redist.d_self[k] = upload(redist.h_self[k]);
    redist.glue_r = dvec(redist.n_glue_own);
    redist.glue_x = dvec(nglue);
  }
  if (has_outer()) d_own2loc = upload(h_own2loc);
  cg_w = dvec(n0); cg_p = dvec(n0); cg_z = dvec(n0); cg_r = dvec(n0);
  st_b = dvec(n0); st_x = dvec(n0);
  if (comm.nranks > 1) {
    cg_x = dvec(n0);
    d_rep_gid = upload(h_rep_gid);
    d_rep_tmp = dvec((int64_t)h_rep_gid.size());
    if (comm.kind == COMM_HOST && !h_rep_full)
      HIP_CHECK(hipHostMalloc((void **)&h_rep_full, sizeof(double) * (size_t)std::max<int64_t>(1, lev[rep_from].n)));
  }
  if (comm.nranks > 1) {
    // The sweep form of a partitioned level decides WHAT its halo carries (r_k for the r-gather sweeps, s_k otherwise), so it is a
    // joint decision: a level sweeps in the r-gather form only if EVERY rank's part qualifies (a rank with fewer than 64 rows has no
    // row-pattern table).  One all-reduce per own | ghost level at setup.
    for (int l = 0; l + 1 < nlev; ++l) {
      Level &L = lev[l];
      // (levels of a rank subset: every rank votes -- the ranks that hold nothing of the level, or hold it in the overlapping layout,
      // with a yes -- so that the collective sequence is the same everywhere whatever a rank knows about the level)
      const bool subset_level = redist.present && l >= sub_from && l < rep_from;
      const bool own_ghost = !inactive(l) && L.halo.present && !L.halo.ovl;
      if (!own_ghost && !subset_level) continue;
      if (own_ghost) L.rs_forbid = false;
      double ok = (!own_ghost || rsweep_level(L)) ? 1.0 : 0.0;
      host_allreduce_sum(&ok);
      if (own_ghost) L.rs_forbid = ok < (double)comm.real_ranks() - 0.5;
    }
  }
  HIP_CHECK(hipStreamSynchronize(stream));
  setup_done = true;
  was_setup = true;
  structure_dirty = false;
  for (auto &L : lev) L.values_dirty = false;
}

// numerical_setup!(ns, A): same sparsity, new values (GMGLinearSolvers.jl:260-297, JacobiLinearSolvers.jl:25-27).  Keeps every
// layout decision, table and work vector; rewrites the value arrays on the device, recomputes D^-1, re-factorises the patch
// blocks and the coarse solver.  Possible when the operator of every refreshed level is stored with explicit values
// (SELL-64 / CSR-stream: what variable-coefficient operators get); dictionary / pattern layouts depend on the values
// themselves, so those fall back to a full setup.  Results are bit-identical to a fresh setup with the new values (tested).
bool gmg_solver::can_refresh() const
{
  if (!was_setup || structure_dirty || comm.nranks != 1 || !opt_int("GMG_REFRESH", 1)) return false;
  for (int l = 0; l < nlev; ++l) {
    const Level &L = lev[l];
    if (!L.values_dirty) continue;
    if (L.sA) return false;
    const DevCSR &A = L.A;
    if (A.sell && (A.pat || A.vdict || A.comp_idx || !A.sval)) return false;
    if (!A.rowptr) return false;
  }
  return true;
}

void gmg_solver::refresh_values()
{
  HIP_CHECK(hipSetDevice(device));
  const bool timing = opt_int("GMG_SETUP_TIMING", 0) != 0;
  const auto t_begin = std::chrono::steady_clock::now();
  int *d_nzero = nullptr;
  HIP_CHECK(hipMalloc((void **)&d_nzero, sizeof(int)));
  double *d_val = nullptr;
  size_t cap = 0;
  try {
    for (int l = 0; l < nlev; ++l) {
      Level &L = lev[l];
      if (!L.values_dirty) continue;
      DevCSR &A = L.A;
      const int64_t nnz = L.hA.nnz(), n = L.n;
      HIP_CHECK(hipMemsetAsync(d_nzero, 0, sizeof(int), stream));
      const bool want_dinv = L.dinv != nullptr;
      if (!A.sell) {
        // CSR-stream layout: the value array itself is streamed
        HIP_CHECK(hipMemcpyAsync(A.val, L.hA.val.data(), sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, stream));
        if (want_dinv) {
          const int grid = (int)std::max<int64_t>(1, (n + 255) / 256);
          if (A.ptr64) hipLaunchKernelGGL((inv_diag_kernel<int64_t>), dim3(grid), dim3(256), 0, stream, n, (const int64_t *)A.rowptr, A.col, A.val, L.dinv, d_nzero);
          else hipLaunchKernelGGL((inv_diag_kernel<int32_t>), dim3(grid), dim3(256), 0, stream, n, (const int32_t *)A.rowptr, A.col, A.val, L.dinv, d_nzero);
          HIP_CHECK(hipGetLastError());
        }
      } else {
        if ((size_t)nnz > cap) {
          if (d_val) { HIP_CHECK(hipStreamSynchronize(stream)); (void)hipFree(d_val); d_val = nullptr; }
          HIP_CHECK(hipMalloc((void **)&d_val, sizeof(double) * (size_t)std::max<int64_t>(nnz, 1)));
          cap = (size_t)nnz;
        }
        HIP_CHECK(hipMemcpyAsync(d_val, L.hA.val.data(), sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, stream));
        const int grid = (int)std::max<int64_t>(1, (n + 255) / 256);
        if (A.ptr64) hipLaunchKernelGGL((sell_refill_kernel<int64_t>), dim3(grid), dim3(256), 0, stream, n, (const int64_t *)A.rowptr, A.soff, A.scol, d_val, A.sval, L.dinv, d_nzero);
        else hipLaunchKernelGGL((sell_refill_kernel<int32_t>), dim3(grid), dim3(256), 0, stream, n, (const int32_t *)A.rowptr, A.soff, A.scol, d_val, A.sval, L.dinv, d_nzero);
        HIP_CHECK(hipGetLastError());
      }
      int nzero = 0;
      HIP_CHECK(hipMemcpyAsync(&nzero, d_nzero, sizeof(int), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      const bool need_diag = l < nlev - 1 ? (L.pre.kind == SM_JACOBI || L.post.kind == SM_JACOBI) : coarse_eff == GMG_COARSE_CG_JACOBI;
      REQUIRE(!(want_dinv && need_diag && nzero > 0), GMG_ERR_SINGULAR, "zero diagonal entry on level " + std::to_string(l));
      L.s0_ready = false;
      if (l < nlev - 1) {
        if (L.has_pcorr) build_patch(L, L.pcorr, true);
        if (L.pre.kind == SM_PATCH) build_patch(L, L.pre, true);
        if (L.post_shares_pre) L.post = L.pre;
        else if (L.post.kind == SM_PATCH) build_patch(L, L.post, true);
      } else if (coarse_eff == GMG_COARSE_DENSE_INVERSE) {
        release(d_Ainv, (size_t)n * (size_t)n);
        build_coarse();
      }
      L.values_dirty = false;
    }
  } catch (...) {
    if (d_val) (void)hipFree(d_val);
    (void)hipFree(d_nzero);
    throw;
  }
  HIP_CHECK(hipStreamSynchronize(stream));
  if (d_val) (void)hipFree(d_val);
  (void)hipFree(d_nzero);
  setup_done = true;
  if (timing) std::fprintf(stderr, "[gmg_setup] value refresh %8.1f ms\n",
                           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
}

// ----------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------
namespace {
template <typename F>
int guarded(gmg_handle_t h, F &&f)
{
  try {
    if (h) HIP_CHECK(hipSetDevice(h->device));
    f();
    if (h) h->check_persistent();                           // a bounded wait of the one-launch smoothing pass timed out (host-mapped flag)
    return GMG_OK;
  } catch (const GmgError &e) {
    if (h) h->err = e.msg;
    g_last_error = e.msg;
    return e.code;
  } catch (const std::bad_alloc &) {
    if (h) h->err = "host allocation failed";
    g_last_error = "host allocation failed";
    return GMG_ERR_ALLOC;
  } catch (const std::exception &e) {
    if (h) h->err = e.what();
    g_last_error = e.what();
    return GMG_ERR_INVALID;
  } catch (...) {                                            // nothing may cross the C ABI
    if (h) h->err = "unknown exception";
    g_last_error = "unknown exception";
    return GMG_ERR_INVALID;
  }
}
void check_level(gmg_handle_t h, int lev, bool not_coarsest)
{
  REQUIRE(h, GMG_ERR_INVALID, "null handle");
  REQUIRE(lev >= 0 && lev < h->nlev, GMG_ERR_INVALID, "level out of range");
  if (not_coarsest) REQUIRE(lev < h->nlev - 1, GMG_ERR_INVALID, "level must not be the coarsest");
}
// level index of the entry points that also take the separate Krylov operator (GMG_LEVEL_KRYLOV -> the slot behind the levels)
int slot_of(gmg_handle_t h, int lev)
{
  REQUIRE(h, GMG_ERR_INVALID, "null handle");
  if (lev == GMG_LEVEL_KRYLOV) return h->nlev;
  check_level(h, lev, false);
  return lev;
}
// exchange plan of one vector space (PartitionedArrays: assembly neighbours + local indices), validated
void fill_plan_checked(HaloPlan &H, const Comm &comm, int64_t n_own, int64_t n_ghost, int nnbr, const int32_t *nbr_rank,
                       const int64_t *snd_ptr, const int64_t *snd_idx, const int64_t *rcv_ptr, const int64_t *rcv_idx, int depth);
// the plan is built aside and replaces the level's only when every check has passed (a rejected call leaves no half-filled plan)
void fill_plan(HaloPlan &H, const Comm &comm, int64_t n_own, int64_t n_ghost, int nnbr, const int32_t *nbr_rank,
               const int64_t *snd_ptr, const int64_t *snd_idx, const int64_t *rcv_ptr, const int64_t *rcv_idx = nullptr, int depth = 1)
{
  HaloPlan T;
  fill_plan_checked(T, comm, n_own, n_ghost, nnbr, nbr_rank, snd_ptr, snd_idx, rcv_ptr, rcv_idx, depth);
  if (H.h_send) (void)hipHostFree(H.h_send);               // pinned buffers of the plan that is replaced
  if (H.h_recv) (void)hipHostFree(H.h_recv);
  H = std::move(T);
}
void fill_plan_checked(HaloPlan &H, const Comm &comm, int64_t n_own, int64_t n_ghost, int nnbr, const int32_t *nbr_rank,
                       const int64_t *snd_ptr, const int64_t *snd_idx, const int64_t *rcv_ptr, const int64_t *rcv_idx, int depth)
{
  REQUIRE(n_own >= 0 && n_ghost >= 0 && nnbr >= 0, GMG_ERR_INVALID, "negative sizes");
  REQUIRE(nnbr == 0 || (nbr_rank && snd_ptr && rcv_ptr), GMG_ERR_INVALID, "null neighbour arrays");
  H = HaloPlan();
  H.present = true; H.n_own = n_own; H.n_ghost = n_ghost;
  H.nbr.assign(nbr_rank, nbr_rank + nnbr);
  H.snd_ptr.assign(1, 0); H.rcv_ptr.assign(1, 0);
  if (nnbr > 0) { H.snd_ptr.assign(snd_ptr, snd_ptr + nnbr + 1); H.rcv_ptr.assign(rcv_ptr, rcv_ptr + nnbr + 1); }
  REQUIRE(H.snd_ptr[0] == 0 && H.rcv_ptr[0] == 0, GMG_ERR_INVALID, "snd_ptr / rcv_ptr must start at 0");
  for (int k = 0; k < nnbr; ++k) {
    REQUIRE(H.snd_ptr[k] <= H.snd_ptr[k + 1] && H.rcv_ptr[k] <= H.rcv_ptr[k + 1], GMG_ERR_INVALID, "pointers not monotone");
    REQUIRE(nbr_rank[k] >= 0 && nbr_rank[k] < comm.nranks && nbr_rank[k] != comm.rank, GMG_ERR_INVALID, "bad neighbour rank");
  }
  REQUIRE(H.rcv_ptr.back() == n_ghost, GMG_ERR_INVALID, "rcv_ptr must cover exactly the ghost segment");
  const int64_t ns = H.snd_ptr.back();
  REQUIRE(ns == 0 || snd_idx, GMG_ERR_INVALID, "null snd_idx");
  H.h_snd_idx.assign(snd_idx, snd_idx + ns);
  if (!rcv_idx) {
    for (int64_t i = 0; i < ns; ++i) REQUIRE(snd_idx[i] >= 0 && snd_idx[i] < n_own, GMG_ERR_INVALID, "snd_idx must address owned entries");
    return;
  }
  // overlapping layout: owned and ghost entries share one local numbering of n_own + n_ghost entries; every ghost entry is received
  // exactly once, no sent entry is a ghost
  REQUIRE(depth >= 1, GMG_ERR_INVALID, "halo depth must be >= 1");
  const int64_t nloc = n_own + n_ghost;
  H.ovl = true; H.depth = depth;
  H.h_rcv_idx.assign(rcv_idx, rcv_idx + n_ghost);
  std::vector<uint8_t> is_ghost((size_t)nloc, 0);
  for (int64_t i = 0; i < n_ghost; ++i) {
    REQUIRE(rcv_idx[i] >= 0 && rcv_idx[i] < nloc, GMG_ERR_INVALID, "rcv_idx out of range");
    REQUIRE(!is_ghost[(size_t)rcv_idx[i]], GMG_ERR_INVALID, "a ghost entry is received twice");
    is_ghost[(size_t)rcv_idx[i]] = 1;
  }
  for (int64_t i = 0; i < ns; ++i)
    REQUIRE(snd_idx[i] >= 0 && snd_idx[i] < nloc && !is_ghost[(size_t)snd_idx[i]], GMG_ERR_INVALID, "snd_idx must address owned entries");
}
void check_ready(gmg_handle_t h)
{
  REQUIRE(h, GMG_ERR_INVALID, "null handle");
  REQUIRE(h->setup_done, GMG_ERR_STATE, "gmg_setup has not been called (numerical_setup missing)");
}
} // namespace

extern "C" {

int gmg_version(void) { return 100; }

const char *gmg_last_error(gmg_handle_t h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int gmg_create(gmg_handle_t *out, int nlevels, int device_id)
{
  return guarded(nullptr, [&] {
    REQUIRE(out, GMG_ERR_INVALID, "null handle pointer");
    *out = nullptr;
    REQUIRE(nlevels >= 2, GMG_ERR_INVALID, "at least two levels required");
    int ndev = 0;
    HIP_CHECK(hipGetDeviceCount(&ndev));
    REQUIRE(ndev > 0, GMG_ERR_HIP, "no HIP device visible: libgmgamd has no CPU path");
    REQUIRE(device_id >= 0 && device_id < ndev, GMG_ERR_INVALID, "device_id out of range");
    HIP_CHECK(hipSetDevice(device_id));
    gmg_solver *s = new gmg_solver();
    s->device = device_id;
    s->nlev = nlevels;
    s->lev.resize(nlevels + 1);                             // + the slot of a separate Krylov operator (GMG_LEVEL_KRYLOV)
    s->log.configure(100, 1.0e-14, 1.0e-8); // GMGLinearSolvers.jl:58 defaults
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete s;
      throw GmgError{GMG_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)};
    }
    s->own_stream = s->stream;
    *out = s;
  });
}

int gmg_destroy(gmg_handle_t h)
{
  if (!h) return GMG_OK;
  // a block preconditioner still borrows this handle: make it forget the pointer (it then needs a new
  // gmg_block_set_diag_gmg + gmg_block_setup) instead of leaving it dangling -- destruction order is free
  if (h->attached_to) block_forget(h->attached_to, h);
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  h->free_all();
  for (auto &L : h->lev) {
    if (L.halo.h_send) (void)hipHostFree(L.halo.h_send);
    if (L.halo.h_recv) (void)hipHostFree(L.halo.h_recv);
  }
  for (HaloPlan *H : {&h->redist.to_sub, &h->redist.from_sub}) {
    if (H->h_send) (void)hipHostFree(H->h_send);
    if (H->h_recv) (void)hipHostFree(H->h_recv);
  }
  if (h->h_rep_full) (void)hipHostFree(h->h_rep_full);
  if (h->h_cr) (void)hipHostFree(h->h_cr);
  if (h->h_cx) (void)hipHostFree(h->h_cx);
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  if (h->comm.kind == COMM_RCCL && h->comm.comm) (void)h->comm.api.CommDestroy(h->comm.comm);
  if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
  if (h->ev_done) (void)hipEventDestroy(h->ev_done);
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  for (auto ev : h->prof_ev) (void)hipEventDestroy(ev);
  if (h->h_scalars) (void)hipHostFree(h->h_scalars);
  if (h->h_perr) (void)hipHostFree(h->h_perr);
  if (h->h_mail) (void)hipHostFree(h->h_mail);
  for (const auto &r : h->host_regs)
    if (r.ours) (void)hipHostUnregister(const_cast<char *>(r.base));   // the caller's pages are unpinned, never freed
  for (int i = 0; i < 2; ++i) {
    if (h->h_chunk[i]) (void)hipHostFree(h->h_chunk[i]);
    if (h->ev_chunk[i]) (void)hipEventDestroy(h->ev_chunk[i]);
  }
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return GMG_OK;
}

int gmg_set_matrix(gmg_handle_t h, int lev, int64_t nrows, int64_t ncols, int64_t nnz, const void *ptr,
                   const void *idx, const double *val, int layout, int index_base, int index_bytes)
{
  return guarded(h, [&] {
    const int slot = slot_of(h, lev);
    REQUIRE(nrows == ncols || h->comm.nranks > 1, GMG_ERR_INVALID, "level matrix must be square");
    Level &L = h->lev[slot];
    L.sA.reset();
    L.clear_split();                                        // (a split stream of an earlier gmg_set_operator_rows is gone with it)
    L.input_layout = layout;
    L.csc_perm.clear(); L.csc_perm.shrink_to_fit();
    if (slot < h->nlev - 1 && h->try_eager_pattern(L.sA, 0, nrows, ncols, nnz, ptr, idx, val, layout, index_base, index_bytes)) {
      L.hA = HostCSR(); L.hA.nrows = nrows; L.hA.ncols = ncols;     // shape only
    } else
    L.hA = convert_input(nrows, ncols, nnz, ptr, idx, val, layout, index_base, index_bytes);
    L.hasA = true;
    h->touch();
  });
}

int gmg_set_operator_rows(gmg_handle_t h, int lev, int op, int64_t nrows_total, int64_t ncols, int64_t row0, int64_t nrows_block,
                          const void *ptr, const void *idx, const double *val, int index_base, int index_bytes)
{
  return guarded(h, [&] {
    check_level(h, lev, op != GMG_OP_A);
    REQUIRE(op == GMG_OP_A || op == GMG_OP_P || op == GMG_OP_R, GMG_ERR_INVALID, "unknown operator");
    REQUIRE(index_bytes == 4 || index_bytes == 8, GMG_ERR_INVALID, "index_bytes must be 4 or 8");
    REQUIRE(index_base == 0 || index_base == 1, GMG_ERR_INVALID, "index_base must be 0 or 1");
    REQUIRE(nrows_total >= 1 && ncols >= 1 && nrows_total < (int64_t)INT32_MAX && ncols < (int64_t)INT32_MAX, GMG_ERR_INVALID, "bad operator shape");
    Level &L = h->lev[lev];
    std::shared_ptr<PatStream> &S = op == GMG_OP_A ? L.sA : op == GMG_OP_P ? L.sP : L.sR;
    HostCSR &H = op == GMG_OP_A ? L.hA : op == GMG_OP_P ? L.hP : L.hR;
    bool &has = op == GMG_OP_A ? L.hasA : op == GMG_OP_P ? L.hasP : L.hasR;
    // own | ghost level (gmg_set_partition called before): rows are n_own x (n_own + n_ghost); the stream takes the own columns
    const bool split = op == GMG_OP_A && h->comm.nranks > 1 && L.halo.present && !L.halo.ovl;
    if (split)
      REQUIRE(nrows_total == L.halo.n_own && ncols == L.halo.n_own + L.halo.n_ghost, GMG_ERR_INVALID,
              "streamed matrix of an own | ghost level: n_own rows, n_own + n_ghost columns (declare the partition first)");
    if (row0 == 0) {                                        // first block: (re)start the stream
      if (op == GMG_OP_A && !split)
        REQUIRE(nrows_total == ncols, GMG_ERR_INVALID,
                h->comm.nranks > 1 ? "a streamed level matrix must be square: own | ghost levels take their rows whole (gmg_set_matrix), levels in the "
                                     "overlapping layout (gmg_set_partition_overlap) and replicated levels can be streamed"
                                   : "level matrix must be square");
      S = std::make_shared<PatStream>();
      S->mode = (op == GMG_OP_A) ? 0 : 1;
      S->nrows = nrows_total; S->ncols = split ? nrows_total : ncols;
      H = HostCSR();
      H.nrows = nrows_total; H.ncols = ncols;               // shape only: the rows are never kept
      has = false;
      if (op == GMG_OP_A) {
        L.clear_split();
        L.sA_split = split;
      }
      h->touch();
    }
    REQUIRE(S && S->nrows == nrows_total && S->ncols == (split ? nrows_total : ncols), GMG_ERR_STATE, "row block does not continue the stream started with row0 = 0");
    if (split) {
      REQUIRE(L.sA_split && ptr && (nrows_block == 0 || (idx && val)), GMG_ERR_INVALID, "null row block");
      REQUIRE(row0 + nrows_block <= nrows_total && nrows_block >= 0, GMG_ERR_INVALID, "row block exceeds the operator");
      const int64_t n_own = nrows_total;
      const int64_t k00 = read_index(ptr, 0, index_bytes);
      const int64_t kend = read_index(ptr, nrows_block, index_bytes);
      REQUIRE(k00 == index_base && kend >= k00, GMG_ERR_INVALID, "row pointers of a block start at index_base");
      std::vector<int64_t> fptr((size_t)nrows_block + 1, 0), fidx;
      std::vector<double> fval;
      fidx.reserve((size_t)(kend - k00)); fval.reserve((size_t)(kend - k00));
      for (int64_t i = 0; i < nrows_block; ++i) {
        const int64_t k0 = read_index(ptr, i, index_bytes) - index_base, k1 = read_index(ptr, i + 1, index_bytes) - index_base;
        REQUIRE(k1 >= k0, GMG_ERR_INVALID, "row pointers must ascend");
        bool any = false;
        for (int64_t k = k0; k < k1; ++k) {
          const int64_t c = read_index(idx, k, index_bytes) - index_base;
          REQUIRE(c >= 0 && c < ncols, GMG_ERR_INVALID, "column index out of range");
          if (c < n_own) { fidx.push_back(c); fval.push_back(val[k]); }
          else { L.g_col.push_back((int32_t)c); L.g_val.push_back(val[k]); any = true; }
        }
        fptr[(size_t)i + 1] = (int64_t)fidx.size();
        if (any) {
          L.g_rows.push_back((int32_t)(row0 + i)); L.g_ptr.push_back((int64_t)L.g_col.size());
          for (int64_t k = k0; k < k1; ++k) L.g_mask.push_back((read_index(idx, k, index_bytes) - index_base) >= n_own ? 1 : 0);
          L.g_mptr.push_back((int64_t)L.g_mask.size());
        }
      }
      h->stream_append(*S, row0, nrows_block, fptr.data(), fidx.data(), fval.data(), 0, 8);
    } else
    h->stream_append(*S, row0, nrows_block, ptr, idx, val, index_base, index_bytes);
    if (S->complete()) has = true;
  });
}

// Structured operators repeat themselves: on a uniform mesh all interior node planes of a level carry the same rows up to a
// shift of the column indices.  After a block of rows has been handed over (gmg_set_operator_rows), this call appends `count`
// further copies of the LAST `nrows_block` rows, copy k with every column index shifted by k * col_shift -- no arrays, no hashing
// (the copies have the pattern ids of the block).  Level matrices keep their offsets relative to the row index, so there
// col_shift must equal nrows_block.  BASELINE config 3 at 256^3: 511 planes of 2.6e5 rows x 125 entries each, 7 distinct ones.
int gmg_set_operator_rows_repeat(gmg_handle_t h, int lev, int op, int64_t nrows_block, int64_t count, int64_t col_shift)
{
  return guarded(h, [&] {
    check_level(h, lev, op != GMG_OP_A);
    REQUIRE(op == GMG_OP_A || op == GMG_OP_P || op == GMG_OP_R, GMG_ERR_INVALID, "unknown operator");
    Level &L = h->lev[lev];
    std::shared_ptr<PatStream> &S = op == GMG_OP_A ? L.sA : op == GMG_OP_P ? L.sP : L.sR;
    bool &has = op == GMG_OP_A ? L.hasA : op == GMG_OP_P ? L.hasP : L.hasR;
    REQUIRE(S && !S->complete(), GMG_ERR_STATE, "no row stream in progress (gmg_set_operator_rows first)");
    REQUIRE(!(op == GMG_OP_A && L.sA_split), GMG_ERR_UNSUPPORTED,
            "the matrix of an own | ghost level is split block by block (its ghost columns do not repeat with a shift): hand every block over");
    PatStream &P = *S;
    REQUIRE(nrows_block >= 1 && count >= 1 && nrows_block <= P.rows_seen, GMG_ERR_INVALID, "the repeated block must lie inside the rows already handed over");
    REQUIRE(P.rows_seen + nrows_block * count <= P.nrows, GMG_ERR_INVALID, "repeated rows exceed the operator");
    REQUIRE(P.mode == 1 || col_shift == nrows_block, GMG_ERR_INVALID, "level matrix: a repeated block shifts its columns by its own height");
    const int64_t r0 = P.rows_seen - nrows_block;
    // column range and nonzeros of the block (from the pattern table)
    int64_t cmin = P.ncols, cmax = -1, bnnz = 0;
    for (int64_t i = r0; i < P.rows_seen; ++i) {
      const int32_t q = P.rowpid[(size_t)i];
      const int32_t len = P.len[(size_t)q];
      if (len == 0) continue;
      const int64_t ref = P.mode == 0 ? i : (int64_t)P.rowbase[(size_t)i];
      cmin = std::min(cmin, ref + P.off[(size_t)P.start[(size_t)q]]);
      cmax = std::max(cmax, ref + P.off[(size_t)P.start[(size_t)q] + len - 1]);
      bnnz += len;
    }
    if (cmax >= 0)
      REQUIRE(cmin + std::min<int64_t>(col_shift, col_shift * count) >= 0 && cmax + std::max<int64_t>(col_shift, col_shift * count) < P.ncols,
              GMG_ERR_INVALID, "repeated rows reach outside the column range");
    const int64_t n_new = nrows_block * count;
    P.rowpid.resize((size_t)(P.rows_seen + n_new));
    if (P.mode == 1) P.rowbase.resize((size_t)(P.rows_seen + n_new));
    parallel_for(n_new, [&](int64_t t) {
      const int64_t k = t / nrows_block + 1, j = t % nrows_block;
      P.rowpid[(size_t)(P.rows_seen + t)] = P.rowpid[(size_t)(r0 + j)];
      if (P.mode == 1) P.rowbase[(size_t)(P.rows_seen + t)] = (int32_t)((int64_t)P.rowbase[(size_t)(r0 + j)] + k * col_shift);
    });
    P.nnz += bnnz * count;
    P.rows_seen += n_new;
    if (P.complete()) has = true;
  });
}

static void update_values_impl(gmg_handle_t h, int lev, const double *val)
{
  {
    Level &L = h->lev[lev];
    REQUIRE(L.hasA, GMG_ERR_STATE, "no matrix set on this level");
    REQUIRE(val, GMG_ERR_INVALID, "null values");
    if (L.sA) {
      // The level is held in row-pattern form only (no CSR copy: structured operators handed over whole, or streamed).  The
      // pattern ids depend on the values, so the rows are rebuilt from the pattern form with the new values and hashed again
      // (numerical_setup! keeps working on every level; a fresh gmg_set_matrix costs the same).
      REQUIRE(L.sA->complete(), GMG_ERR_STATE, "row stream incomplete");
      HostCSR S = gmg_solver::expand_stream(*L.sA);
      const int64_t nnz = S.nnz();
      if (L.sA_split) {
        // own | ghost level streamed with gmg_set_operator_rows: `val` follows the caller's whole rows (own and ghost columns
        // interleaved as they were handed over); own-column values go to the stream's rows, ghost-column values to the fix-up CSR
        int64_t k = 0, go = 0;
        size_t t = 0;
        for (int64_t i = 0; i < S.nrows; ++i) {
          int64_t so = S.ptr[(size_t)i];
          if (t < L.g_rows.size() && L.g_rows[t] == i) {
            for (int64_t m = L.g_mptr[t]; m < L.g_mptr[t + 1]; ++m) {
              if (L.g_mask[(size_t)m]) L.g_val[(size_t)go++] = val[k++];
              else S.val[(size_t)so++] = val[k++];
            }
            REQUIRE(so == S.ptr[(size_t)i + 1] && go == L.g_ptr[t + 1], GMG_ERR_STATE, "split row stream: inconsistent interleave record");
            ++t;
          } else {
            const int64_t c = S.ptr[(size_t)i + 1] - so;
            std::memcpy(&S.val[(size_t)so], val + k, sizeof(double) * (size_t)c);
            k += c;
          }
        }
      } else
      std::memcpy(S.val.data(), val, sizeof(double) * (size_t)nnz);
      std::shared_ptr<PatStream> N;
      bool eager = false;
      if (nnz < (int64_t)INT32_MAX) {
        std::vector<int32_t> p32(S.ptr.begin(), S.ptr.end());
        eager = h->try_eager_pattern(N, 0, S.nrows, S.ncols, nnz, p32.data(), S.col.data(), S.val.data(), GMG_CSR, 0, 4);
      }
      if (eager) L.sA = N;                                  // (hA keeps the shape only; a split stream keeps its refreshed ghost CSR)
      else if (L.sA_split) {                                // whole rows again: own columns first, then the ghost columns
        L.hA = gmg_solver::merge_split(S, L.hA.ncols, L.g_rows, L.g_ptr, L.g_col, L.g_val);
        L.clear_split(); L.sA.reset();
      }
      else { L.hA = std::move(S); L.sA.reset(); }
      h->touch();
      return;
    }
    // values are given in the 0-based CSR order held by the handle
    std::memcpy(L.hA.val.data(), val, sizeof(double) * (size_t)L.hA.nnz());
    L.values_dirty = true;
    h->setup_done = false;                                  // structure untouched: gmg_setup takes the value-refresh path when it can
  }
}

int gmg_update_values(gmg_handle_t h, int lev, const double *val)
{
  return guarded(h, [&] {
    check_level(h, lev, false);
    update_values_impl(h, lev, val);
  });
}

int gmg_update_values_csc(gmg_handle_t h, int lev, const void *colptr, const void *rowidx, const double *val, int index_base, int index_bytes)
{
  return guarded(h, [&] {
    check_level(h, lev, false);
    Level &L = h->lev[lev];
    REQUIRE(L.hasA, GMG_ERR_STATE, "no matrix set on this level");
    REQUIRE(val, GMG_ERR_INVALID, "null values");
    REQUIRE(L.input_layout == GMG_CSC && !L.sA_split, GMG_ERR_STATE,
            "gmg_update_values_csc: this level's matrix was not handed over with gmg_set_matrix in GMG_CSC layout");
    REQUIRE(index_bytes == 4 || index_bytes == 8, GMG_ERR_INVALID, "index_bytes must be 4 or 8");
    REQUIRE(index_base == 0 || index_base == 1, GMG_ERR_INVALID, "index_base must be 0 or 1");
    // the handle's CSR row pointer (a level kept in row-pattern form only: from the pattern lengths)
    std::vector<int64_t> rp;
    const int64_t nrows = L.hA.nrows, ncols = L.hA.ncols;
    if (L.sA) {
      REQUIRE(L.sA->complete(), GMG_ERR_STATE, "row stream incomplete");
      rp.assign((size_t)nrows + 1, 0);
      for (int64_t i = 0; i < nrows; ++i) rp[(size_t)i + 1] = rp[(size_t)i] + L.sA->len[L.sA->rowpid[(size_t)i]];
    } else rp = L.hA.ptr;
    const int64_t nnz = rp.back();
    if (L.csc_perm.size() != (size_t)nnz) {
      // first refresh: where the counting-sort transposition of gmg_set_matrix (convert_input) put every CSC entry
      REQUIRE(colptr && rowidx, GMG_ERR_INVALID, "gmg_update_values_csc: the first call on a level needs colptr and rowidx");
      REQUIRE(nnz < (int64_t)UINT32_MAX, GMG_ERR_UNSUPPORTED, "more than 2^32-1 stored entries on one level");
      REQUIRE(read_index(colptr, 0, index_bytes) == index_base && read_index(colptr, ncols, index_bytes) - index_base == nnz, GMG_ERR_INVALID,
              "gmg_update_values_csc: the column pointers do not describe the pattern this level was set with");
      std::vector<uint32_t> perm((size_t)nnz);
      std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
      for (int64_t c = 0; c < ncols; ++c) {
        const int64_t k0 = read_index(colptr, c, index_bytes) - index_base, k1 = read_index(colptr, c + 1, index_bytes) - index_base;
        REQUIRE(k0 <= k1 && k1 <= nnz, GMG_ERR_INVALID, "column pointers not monotone");
        for (int64_t k = k0; k < k1; ++k) {
          const int64_t r = read_index(rowidx, k, index_bytes) - index_base;
          REQUIRE(r >= 0 && r < nrows, GMG_ERR_INVALID, "row index out of range");
          REQUIRE(fill[(size_t)r] < rp[(size_t)r + 1], GMG_ERR_INVALID, "gmg_update_values_csc: the pattern differs from the one this level was set with");
          perm[(size_t)k] = (uint32_t)fill[(size_t)r]++;
        }
      }
      L.csc_perm.swap(perm);
    }
    std::vector<double> v((size_t)nnz);
    const uint32_t *pm = L.csc_perm.data();
    const int64_t nchunk = (nnz + (1 << 20) - 1) >> 20;
    parallel_chunks(nchunk, [&](int64_t t) {
      const int64_t k0 = t << 20, k1 = std::min(nnz, k0 + (1 << 20));
      for (int64_t k = k0; k < k1; ++k) v[pm[k]] = val[k];
    });
    std::vector<uint32_t> keep;
    keep.swap(L.csc_perm);                                   // (update_values_impl may re-hash the level; the pattern, hence the permutation, stays)
    update_values_impl(h, lev, v.data());
    L.csc_perm.swap(keep);
  });
}

int gmg_set_prolongation(gmg_handle_t h, int lev, int64_t nrows, int64_t ncols, int64_t nnz, const void *ptr,
                         const void *idx, const double *val, int layout, int index_base, int index_bytes)
{
  return guarded(h, [&] {
    check_level(h, lev, true);
    Level &L = h->lev[lev];
    L.sP.reset();
    if (h->try_eager_pattern(L.sP, 1, nrows, ncols, nnz, ptr, idx, val, layout, index_base, index_bytes)) {
      L.hP = HostCSR(); L.hP.nrows = nrows; L.hP.ncols = ncols;     // shape only (R = P^T, if needed, is formed from the pattern form)
    } else
    L.hP = convert_input(nrows, ncols, nnz, ptr, idx, val, layout, index_base, index_bytes);
    L.hasP = true;
    h->touch();
  });
}

int gmg_set_restriction(gmg_handle_t h, int lev, int64_t nrows, int64_t ncols, int64_t nnz, const void *ptr,
                        const void *idx, const double *val, int layout, int index_base, int index_bytes)
{
  return guarded(h, [&] {
    check_level(h, lev, true);
    Level &L = h->lev[lev];
    L.sR.reset();
    if (h->try_eager_pattern(L.sR, 1, nrows, ncols, nnz, ptr, idx, val, layout, index_base, index_bytes)) {
      L.hR = HostCSR(); L.hR.nrows = nrows; L.hR.ncols = ncols;
    } else
    L.hR = convert_input(nrows, ncols, nnz, ptr, idx, val, layout, index_base, index_bytes);
    L.hasR = true;
    h->touch();
  });
}

static void assign_smoother(gmg_handle_t h, int lev, int which, const Smoother &S)
{
  Level &L = h->lev[lev];
  REQUIRE(which == GMG_PRE || which == GMG_POST || which == GMG_PRE_AND_POST, GMG_ERR_INVALID, "bad `which`");
  if (which == GMG_PRE_AND_POST) { L.pre = S; L.post_shares_pre = true; }
  else if (which == GMG_PRE) { L.pre = S; if (L.post_shares_pre) { L.post_shares_pre = false; } }
  else { L.post = S; L.post_shares_pre = false; }
  L.clear_overlap_hints();          // layers-per-sweep was stated for the smoother that was here before: state it again after this call
  h->touch();
}

int gmg_set_smoother_jacobi(gmg_handle_t h, int lev, int which, int niter, double omega)
{
  return guarded(h, [&] {
    check_level(h, lev, true);
    REQUIRE(niter >= 0, GMG_ERR_INVALID, "niter < 0");
    Smoother S;
    S.kind = SM_JACOBI; S.niter = niter; S.omega = omega;
    assign_smoother(h, lev, which, S);
  });
}

// patch tables of PatchSolver / BlockJacobiSolver / PatchProlongationOperator (0-based int32 copies, shared by smoother copies)
static Smoother make_patch_smoother(int niter, double omega, int kind, int64_t npatch, const void *patch_ptr, const void *patch_rows,
                                    const void *patch_cols, int index_base, int index_bytes)
{
  REQUIRE(niter >= 0 && npatch >= 0, GMG_ERR_INVALID, "negative niter / npatch");
  REQUIRE(patch_ptr && (patch_rows || npatch == 0), GMG_ERR_INVALID, "null patch arrays");
  REQUIRE(index_bytes == 4 || index_bytes == 8, GMG_ERR_INVALID, "index_bytes must be 4 or 8");
  REQUIRE(index_base == 0 || index_base == 1, GMG_ERR_INVALID, "index_base must be 0 or 1");
  REQUIRE(kind == GMG_PATCH_LU || kind == GMG_PATCH_NOPIVOT, GMG_ERR_INVALID, "bad patch kind");
  Smoother S;
  S.kind = SM_PATCH; S.niter = niter; S.omega = omega; S.patch_kind = kind;
  S.tab = std::make_shared<Smoother::Tables>();
  Smoother::Tables &T = *S.tab;
  T.pptr.resize((size_t)npatch + 1);
  for (int64_t p = 0; p <= npatch; ++p) T.pptr[p] = read_index(patch_ptr, p, index_bytes) - index_base;
  REQUIRE(T.pptr[0] == 0, GMG_ERR_INVALID, "patch_ptr does not start at index_base");
  for (int64_t p = 0; p < npatch; ++p) REQUIRE(T.pptr[p] <= T.pptr[p + 1], GMG_ERR_INVALID, "patch_ptr not monotone");
  const int64_t tot = T.pptr[npatch];
  T.prow.resize((size_t)tot);
  parallel_for(tot, [&](int64_t q) { T.prow[(size_t)q] = (int32_t)(read_index(patch_rows, q, index_bytes) - index_base); });
  if (patch_cols && patch_cols != patch_rows) {
    T.pcol.resize((size_t)tot);
    bool same = true;
    for (int64_t q = 0; q < tot; ++q) { T.pcol[q] = (int32_t)(read_index(patch_cols, q, index_bytes) - index_base); same = same && T.pcol[q] == T.prow[q]; }
    if (same) { T.pcol.clear(); T.pcol.shrink_to_fit(); }
  }
  return S;
}

int gmg_set_smoother_patch(gmg_handle_t h, int lev, int which, int niter, double omega, int kind, int64_t npatch,
                           const void *patch_ptr, const void *patch_dofs, int index_base, int index_bytes)
{
  return guarded(h, [&] {
    check_level(h, lev, true);
    assign_smoother(h, lev, which, make_patch_smoother(niter, omega, kind, npatch, patch_ptr, patch_dofs, nullptr, index_base, index_bytes));
  });
}

int gmg_set_smoother_patch_matrices(gmg_handle_t h, int lev, int which, int niter, double omega, int kind, int64_t npatch,
                                    const void *patch_ptr, const void *patch_rows, const void *patch_cols, int index_base,
                                    int index_bytes, const double *blocks, int blocks_are_factors, const int32_t *pivots)
{
  return guarded(h, [&] {
    check_level(h, lev, true);
    Smoother S = make_patch_smoother(niter, omega, kind, npatch, patch_ptr, patch_rows, patch_cols, index_base, index_bytes);
    Smoother::Tables &T = *S.tab;
    if (blocks) {
      int64_t tot = 0;
      for (int64_t p = 0; p < npatch; ++p) { const int64_t np = T.pptr[p + 1] - T.pptr[p]; tot += np * np; }
      T.blocks.assign(blocks, blocks + tot);
      T.has_blocks = true;
      T.are_factors = blocks_are_factors != 0;
      if (T.are_factors && pivots) {
        T.piv.assign(pivots, pivots + T.pptr[npatch]);
        for (int64_t p = 0; p < npatch; ++p)
          for (int64_t q = T.pptr[p]; q < T.pptr[p + 1]; ++q)
            REQUIRE(T.piv[q] >= 1 && T.piv[q] <= T.pptr[p + 1] - T.pptr[p], GMG_ERR_INVALID, "pivot index out of range (LAPACK ipiv is 1-based)");
      }
    } else REQUIRE(!blocks_are_factors && !pivots, GMG_ERR_INVALID, "factors / pivots given without blocks");
    assign_smoother(h, lev, which, S);
  });
}

int gmg_set_prolongation_patch_correction(gmg_handle_t h, int lev, int kind, int64_t npatch, const void *patch_ptr,
                                          const void *patch_dofs, int index_base, int index_bytes)
{
  return guarded(h, [&] {
    check_level(h, lev, true);
    Smoother S = make_patch_smoother(0, 1.0, kind, npatch, patch_ptr, patch_dofs, nullptr, index_base, index_bytes);
    h->lev[lev].pcorr = S;
    h->lev[lev].has_pcorr = true;
    h->touch();
  });
}

int gmg_set_prolongation_patch_correction_rhs(gmg_handle_t h, int lev, int64_t n, int64_t nnz, const void *ptr, const void *idx,
                                              const double *val, int layout, int index_base, int index_bytes)
{
  return guarded(h, [&] {
    check_level(h, lev, true);
    // distributed level (gmg_set_partition first): this rank's rows, [own | ghost] columns
    const HaloPlan &Hp = h->lev[lev].halo;
    const int64_t ncols = (h->comm.nranks > 1 && Hp.present && !Hp.ovl) ? Hp.n_own + Hp.n_ghost : n;
    h->lev[lev].hG = convert_input(n, ncols, nnz, ptr, idx, val, layout, index_base, index_bytes);
    h->lev[lev].hasG = true;
    h->touch();
  });
}

int gmg_set_options(gmg_handle_t h, int mode, int cycle, int maxiter, double atol, double rtol)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    REQUIRE(mode == GMG_MODE_PRECONDITIONER || mode == GMG_MODE_SOLVER, GMG_ERR_INVALID,
            "mode must be :preconditioner or :solver (GMGLinearSolvers.jl:60)");
    REQUIRE(cycle == GMG_V_CYCLE || cycle == GMG_W_CYCLE || cycle == GMG_F_CYCLE, GMG_ERR_INVALID,
            "cycle_type must be :v_cycle, :w_cycle or :f_cycle (GMGLinearSolvers.jl:61)");
    REQUIRE(maxiter >= 0, GMG_ERR_INVALID, "maxiter < 0");
    h->mode = mode; h->cycle_type = cycle;
    h->log.configure(maxiter, atol, rtol);
  });
}

int gmg_set_coarse_solver(gmg_handle_t h, int kind, int maxiter, double atol, double rtol, gmg_coarse_solve_fn fn, void *ctx)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    REQUIRE(kind == GMG_COARSE_DENSE_INVERSE || kind == GMG_COARSE_CG_JACOBI || kind == GMG_COARSE_HOST_CALLBACK, GMG_ERR_INVALID,
            "unknown coarse solver kind");
    if (kind == GMG_COARSE_CG_JACOBI) REQUIRE(maxiter >= 0, GMG_ERR_INVALID, "maxiter < 0");
    if (kind == GMG_COARSE_HOST_CALLBACK) REQUIRE(fn, GMG_ERR_INVALID, "null callback");
    h->coarse_kind = kind;
    if (kind == GMG_COARSE_CG_JACOBI) { h->coarse_maxiter = maxiter; h->coarse_atol = atol; h->coarse_rtol = rtol; }
    h->coarse_fn = fn; h->coarse_ctx = ctx;
    h->touch();
  });
}

int gmg_get_coarse_log(gmg_handle_t h, gmg_result *res)
{
  return guarded(h, [&] {
    REQUIRE(h && res, GMG_ERR_INVALID, "null argument");
    REQUIRE(h->coarse_eff == GMG_COARSE_CG_JACOBI && !h->coarse_log.residuals.empty(), GMG_ERR_STATE, "no iterative coarse solve has run");
    h->coarse_log.export_to(res, nullptr, 0, h->coarse_last);
  });
}

// ---- per-handle policy options --------------------------------------------------------------------------------------------
namespace {
// every switch the library reads; `live` options take effect at the next call, the others at the next gmg_setup
struct OptionKey { const char *name; bool live; };
const OptionKey kOptionKeys[] = {
  {"GMG_BIG_ROWS", false}, {"GMG_COARSE_HOST_FALLBACK_MAX", false}, {"GMG_COARSE_HOST_MAX", false}, {"GMG_COARSE_AUTO_CG_MIN", false},
  {"GMG_DBG_NOGATHER", false}, {"GMG_EAGER", true}, {"GMG_EAGER_MIN_ROWS", true}, {"GMG_FORCE_PTR64", false}, {"GMG_GJ_MFMA", false}, {"GMG_GJ_WIDE_MIN", false},
  {"GMG_HALO_FUSE_PACK", false}, {"GMG_HALO_FIX_SELL", false}, {"GMG_HALO_FIX_DICT", false}, {"GMG_HALO_SPLIT_R", false}, {"GMG_HOST_ASYNC", false}, {"GMG_IDX16", false}, {"GMG_LANES_LOG2", false}, {"GMG_NT", false},
  {"GMG_NT_ROWWISE", false}, {"GMG_ONE_GATHER", false}, {"GMG_OPATTERN", false}, {"GMG_OVERLAP", false}, {"GMG_PATCH_DEDUP", false},
  {"GMG_PATCH_OPERATOR", false}, {"GMG_PATCH_OP_DEVICE", false}, {"GMG_PATCH_SOURCE_DEDUP", false}, {"GMG_PATTERN", false}, {"GMG_PAT_BATCHED", false},
  {"GMG_PAT_CODED_MIN_ROWS", false}, {"GMG_PAT_DEFER", false}, {"GMG_PAT_DINV", false}, {"GMG_PAT_EMIT", false}, {"GMG_PAT_NB", false},
  {"GMG_PAT_RB", false}, {"GMG_PAT_RSWEEP", false}, {"GMG_PAT_SHARED", false}, {"GMG_PAT_SMALL_WPB", false}, {"GMG_PAT_SMALL_WPB2", false},
  {"GMG_PAT_STRICT", false}, {"GMG_PAT_TILE", false}, {"GMG_PAT_TILE_LDS", false}, {"GMG_PAT_TILE_MIN", false}, {"GMG_PAT_TILE_ROWS", false},
  {"GMG_PAT_TILE_T", false}, {"GMG_PAT_UN", false}, {"GMG_PAT_WGS", false}, {"GMG_PAT_WIDE", false}, {"GMG_PAT_WIDE_LDS", false},
  {"GMG_PAT_WIDE_ROUNDS", false}, {"GMG_PERSIST", false}, {"GMG_PERSIST_FENCED", false}, {"GMG_PERSIST_MAX_SLICES", false},
  {"GMG_PERSIST_SHARED", false}, {"GMG_PROF_STRIDE", true}, {"GMG_REFRESH", true}, {"GMG_SELL", false}, {"GMG_SELL_BLOCK", false},
  {"GMG_SELL_DEFER", false}, {"GMG_SELL_MAXPAD", false}, {"GMG_SELL_UN", false}, {"GMG_SETUP_TIMING", true}, {"GMG_VDICT", false},
  {"GMG_XCD_REMAP", false}, {"GMG_XCD_REMAP_BIG", false}, {"GMG_X0_ZERO", true}, {"GMG_HOST_POLL", true}, {"GMG_HOST_CHUNK_BYTES", true}, {"GMG_PAT_FMA", false}, {"GMG_PAT_R2", false}, {"GMG_RED_FUSED", false}, {"GMG_PAT_R2MV", false}, {"GMG_PAT_R2_OCC", false}, {"GMG_PAT_PAIR_P", false}, {"GMG_PAT_R2MV_DOT", false}, {"GMG_PERSIST_WPB", false}, {"GMG_HOST_TIMELINE", true}, {"GMG_PAT_R2MV_MIN", false}, {"GMG_PAT_BCAST", false}, {"GMG_PAT_R2_WGS", false}, {"GMG_PAT_ZWALK", false}, {"GMG_PAT_ZWALK_T", false}, {"GMG_PAT_ZWALK_ROWS", false}, {"GMG_PAT_ZWALK_MV", false}, {"GMG_PAT_ZWALK_WIDE", false}, {"GMG_PAT_ZWALK_WIDE_ROWS", false}, {"GMG_PAT_FUSE2", false}, {"GMG_PAT_FUSE2_W", false}, {"GMG_PAT_FUSE2_T", false}, {"GMG_PAT_FUSE2_ROWS", false}, {"GMG_PAT_FUSE2_BOX", false}, {"GMG_PAT_BOX", false}, {"GMG_PAT_WIDE_GRID", false}, {"GMG_PAT_BOX_T", false}, {"GMG_PAT_BOX_MIN_ROWS", false}, {"GMG_PAT_BOX_MAX_ROWS", false},
  {"GMG_PERSIST_FORCE_TIMEOUT", true},
};
// "pat_tile", "PAT_TILE" and "GMG_PAT_TILE" name the same option
const OptionKey *find_option(const char *key)
{
  if (!key) return nullptr;
  std::string k(key);
  for (auto &c : k) c = (char)std::toupper((unsigned char)c);
  if (k.rfind("GMG_", 0) != 0) k = "GMG_" + k;
  for (const OptionKey &o : kOptionKeys)
    if (k == o.name) return &o;
  return nullptr;
}
} // namespace

int gmg_set_option(gmg_handle_t h, const char *key, double value)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    const OptionKey *o = find_option(key);
    REQUIRE(o, GMG_ERR_INVALID, std::string("unknown option '") + (key ? key : "(null)") + "' (see the option table in include/gmg_amd.h)");
    auto it = h->options.find(o->name);
    if (it != h->options.end() && it->second == value) return;
    h->options[o->name] = value;
    if (!o->live) h->touch();                                // layouts / schedules are chosen at gmg_setup
  });
}

int gmg_get_option(gmg_handle_t h, const char *key, double *value, int *source)
{
  return guarded(h, [&] {
    REQUIRE(h && value, GMG_ERR_INVALID, "null argument");
    const OptionKey *o = find_option(key);
    REQUIRE(o, GMG_ERR_INVALID, std::string("unknown option '") + (key ? key : "(null)") + "'");
    const char *e = std::getenv(o->name);
    const bool from_env = e && *e;
    const bool from_handle = h->options.count(o->name) != 0;
    *value = h->opt_num(o->name, std::nan(""));              // NaN: the library's built-in default applies
    if (source) *source = from_env ? 2 : (from_handle ? 1 : 0);
  });
}

int gmg_host_register(gmg_handle_t h, const void *ptr, int64_t nbytes)
{
  return guarded(h, [&] {
    REQUIRE(h && ptr && nbytes > 0, GMG_ERR_INVALID, "gmg_host_register: null handle / pointer or empty range");
    if (h->host_registered(ptr, (size_t)nbytes)) return;
    // page-locked by somebody else already (hipHostMalloc'd, or registered through another handle or library)?  DMA works as it is;
    // remembered as foreign so that this handle never unregisters it
    hipPointerAttribute_t attr;
    const hipError_t pa = hipPointerGetAttributes(&attr, ptr);
    (void)hipGetLastError();                                 // (a plain malloc'd pointer is "invalid value" for the query)
    hipError_t e = hipSuccess;
    const bool foreign = pa == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!foreign) e = hipHostRegister(const_cast<void *>(ptr), (size_t)nbytes, hipHostRegisterDefault);
    if (foreign || e == hipErrorHostMemoryAlreadyRegistered) {
      (void)hipGetLastError();
      h->host_regs.push_back({static_cast<const char *>(ptr), (size_t)nbytes, false});
      return;
    }
    HIP_CHECK(e);
    h->host_regs.push_back({static_cast<const char *>(ptr), (size_t)nbytes, true});
  });
}

int gmg_host_unregister(gmg_handle_t h, const void *ptr)
{
  return guarded(h, [&] {
    REQUIRE(h && ptr, GMG_ERR_INVALID, "gmg_host_unregister: null handle / pointer");
    for (size_t i = 0; i < h->host_regs.size(); ++i)
      if (h->host_regs[i].base == static_cast<const char *>(ptr)) {
        HIP_CHECK(hipStreamSynchronize(h->stream));
        if (h->host_regs[i].ours) HIP_CHECK(hipHostUnregister(const_cast<void *>(ptr)));
        h->host_regs.erase(h->host_regs.begin() + (long)i);
        return;
      }
    throw GmgError{GMG_ERR_INVALID, "gmg_host_unregister: this range was not registered with this handle (pass the base pointer given to gmg_host_register)"};
  });
}

int gmg_get_host_io_stats(gmg_handle_t h, int64_t *bytes_up, int64_t *bytes_down, int64_t *nregistered)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    if (bytes_up) *bytes_up = h->host_bytes_up;
    if (bytes_down) *bytes_down = h->host_bytes_down;
    if (nregistered) *nregistered = (int64_t)h->host_regs.size();
  });
}

int gmg_get_persist_retries(gmg_handle_t h, int64_t *retries, int *persist_active)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    if (retries) *retries = h->persist_retries;
    if (persist_active) *persist_active = h->persist;
  });
}

// The stream the handle issues its work on.  stream = NULL: back to the handle's own (created by gmg_create).
int gmg_set_stream(gmg_handle_t h, void *stream)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    REQUIRE(!h->attached_to, GMG_ERR_STATE, "this handle is a diagonal block of a block preconditioner, which issues its work on the block handle's stream");
    HIP_CHECK(hipSetDevice(h->device));
    HIP_CHECK(hipStreamSynchronize(h->stream));             // nothing of ours is left on the stream we leave
    // the two sentinels of the header are mapped by NAME (not by casting 1 / 2 to a handle: that would rest on the numeric values of
    // hipStreamLegacy / hipStreamPerThread in this runtime), and whatever comes out is checked before it is accepted
    hipStream_t s = h->own_stream;
    if (stream == GMG_STREAM_LEGACY) s = hipStreamLegacy;
    else if (stream == GMG_STREAM_PER_THREAD) s = hipStreamPerThread;
    else if (stream) s = static_cast<hipStream_t>(stream);
    if (s != h->own_stream) {
      const hipError_t q = hipStreamQuery(s);
      if (q != hipSuccess && q != hipErrorNotReady) {
        (void)hipGetLastError();
        throw GmgError{GMG_ERR_INVALID, std::string("gmg_set_stream: not a usable HIP stream (") + hipGetErrorString(q) + ")"};
      }
    }
    h->stream = s;
  });
}

int gmg_get_stream(gmg_handle_t h, void **stream)
{
  return guarded(h, [&] {
    REQUIRE(h && stream, GMG_ERR_INVALID, "null argument");
    *stream = static_cast<void *>(h->stream);
  });
}

int gmg_set_verbose(gmg_handle_t h, int verbose)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    h->verbose = verbose;
  });
}

int gmg_get_log(gmg_handle_t h, gmg_result *res, double *hist, int hist_cap)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    h->log.export_to(res, hist, hist_cap, h->log_last);
  });
}

int gmg_setup(gmg_handle_t h)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    if (h->can_refresh()) h->refresh_values();             // numerical_setup!: only values changed and the layouts carry values
    else h->setup();
  });
}

int gmg_apply(gmg_handle_t h, const double *b, double *x, int memspace, gmg_result *res, double *hist, int hist_cap)
{
  return guarded(h, [&] {
    check_ready(h);
    REQUIRE(b && x, GMG_ERR_INVALID, "null vector");
    h->with_persist_retry(x, h->user_n(), memspace, h->mode == GMG_MODE_PRECONDITIONER, [&] {
      const int64_t n = h->user_n();
      const double *db = h->in_vec(b, n, memspace, h->st_b);
      const bool dist = h->comm.nranks > 1;
      double *dx = dist ? h->cg_x : ((memspace == GMG_MEM_DEVICE) ? x : h->st_x);
      if (h->mode == GMG_MODE_SOLVER) h->in_guess(dx, x, n, memspace);   // :preconditioner overwrites x with zeros first (GMGLinearSolvers.jl:619)
      double last = NAN;
      if (h->has_outer()) h->krylov_precond(1, dx, db, -1.0);   // finest level in the overlapping layout: through level 0's numbering
      else last = h->gmg_solve_dev(dx, db, -1.0);
      h->out_vec(x, dx, n, memspace);
      h->log.export_to(res, hist, hist_cap, last);
    });
  });
}

int gmg_cg_solve(gmg_handle_t h, const double *b, double *x, int memspace, int maxiter, double atol, double rtol,
                 int flexible, int use_precond, gmg_result *res, double *hist, int hist_cap)
{
  return guarded(h, [&] {
    check_ready(h);
    REQUIRE(b && x, GMG_ERR_INVALID, "null vector");
    REQUIRE(maxiter >= 0, GMG_ERR_INVALID, "maxiter < 0");
    REQUIRE(use_precond >= 0 && use_precond <= 3, GMG_ERR_INVALID, "use_precond must be 0, 1, 2 or 3");
    h->with_persist_retry(x, h->user_n(), memspace, false, [&] {
      gmg_solver &S = *h;
      Level &L0 = S.lev[S.kl()];
      const int64_t n = L0.n;
      const double *db = S.in_vec(b, n, memspace, S.st_b);
      const bool dist = S.comm.nranks > 1;
      double *dx = dist ? S.cg_x : ((memspace == GMG_MEM_DEVICE) ? x : S.st_x);
      S.in_guess(dx, x, n, memspace);                        // x = initial guess on entry (CGSolvers.jl:79)
      ConvLog log;
      log.configure(maxiter, atol, rtol);
      KrylovOps ops = S.level0_ops(use_precond);
      const double resn = cg_core(S, n, db, dx, S.cg_w, S.cg_p, S.cg_z, S.cg_r, ops, flexible != 0, log);
      S.out_vec(x, dx, n, memspace);
      log.export_to(res, hist, hist_cap, resn);              // :118
    });
  });
}

int gmg_fgmres_solve(gmg_handle_t h, const double *b, double *x, int memspace, int m0, int restart, int m_add,
                     int maxiter, double atol, double rtol, int use_precond, gmg_result *res, double *hist, int hist_cap)
{
  return gmg_fgmres_solve_pl(h, b, x, memspace, m0, restart, m_add, maxiter, atol, rtol, use_precond, 0, res, hist, hist_cap);
}

int gmg_fgmres_solve_pl(gmg_handle_t h, const double *b, double *x, int memspace, int m0, int restart, int m_add,
                        int maxiter, double atol, double rtol, int use_precond, int use_precond_left, gmg_result *res,
                        double *hist, int hist_cap)
{
  return guarded(h, [&] {
    REQUIRE(use_precond_left >= 0 && use_precond_left <= 3, GMG_ERR_INVALID, "use_precond_left must be 0, 1, 2 or 3");
    REQUIRE(!(use_precond == 1 && use_precond_left == 1), GMG_ERR_UNSUPPORTED,
            "one handle holds one GMG: it can be Pr or Pl, not both (its work vectors are in use)");
    check_ready(h);
    REQUIRE(b && x, GMG_ERR_INVALID, "null vector");
    REQUIRE(m0 >= 1 && m_add >= 1 && maxiter >= 0, GMG_ERR_INVALID, "bad FGMRES sizes");
    REQUIRE(use_precond >= 0 && use_precond <= 3, GMG_ERR_INVALID, "use_precond must be 0, 1, 2 or 3");
    h->with_persist_retry(x, h->user_n(), memspace, false, [&] {
      gmg_solver &S = *h;
      Level &L0 = S.lev[S.kl()];
      const int64_t n = L0.n;
      const double *db = S.in_vec(b, n, memspace, S.st_b);
      const bool dist = S.comm.nranks > 1;
      const int64_t nv = L0.nvec;
      double *dx = dist ? S.cg_x : ((memspace == GMG_MEM_DEVICE) ? x : S.st_x);
      S.in_guess(dx, x, n, memspace);                        // x = initial guess on entry (CGSolvers.jl:79)
      ConvLog log;
      log.configure(maxiter, atol, rtol);
      KrylovOps ops = S.level0_ops(use_precond);
      if (use_precond_left) {
        ops.zl = S.scratch_vec(4, nv);
        ops.precond_left = [&S, use_precond_left](double *z, const double *r) { S.krylov_precond(use_precond_left, z, r, -1.0); };
      }
      const double beta = fgmres_core(S, n, nv, db, dx, S.fg_V, S.fg_Z, ops, m0, restart != 0, m_add, log);
      S.out_vec(x, dx, n, memspace);
      log.export_to(res, hist, hist_cap, beta);              // :197
    });
  });
}

int gmg_richardson_solve(gmg_handle_t h, const double *b, double *x, int memspace, double omega, int maxiter, double atol,
                         double rtol, int use_precond, gmg_result *res, double *hist, int hist_cap)
{
  return guarded(h, [&] {
    check_ready(h);
    REQUIRE(b && x, GMG_ERR_INVALID, "null vector");
    REQUIRE(maxiter >= 0, GMG_ERR_INVALID, "maxiter < 0");
    REQUIRE(use_precond >= 0 && use_precond <= 3, GMG_ERR_INVALID, "use_precond must be 0, 1, 2 or 3");
    REQUIRE(!h->has_outer(), GMG_ERR_UNSUPPORTED, "finest level in the overlapping layout: CG / FGMRES only");
    h->with_persist_retry(x, h->lev[0].n, memspace, false, [&] {
      gmg_solver &S = *h;
      const int64_t n = S.lev[0].n;
      const double *db = S.in_vec(b, n, memspace, S.st_b);
      const bool dist = S.comm.nranks > 1;
      double *dx = dist ? S.cg_x : ((memspace == GMG_MEM_DEVICE) ? x : S.st_x);
      S.in_guess(dx, x, n, memspace);                        // x = initial guess on entry (CGSolvers.jl:79)
      double *z = S.cg_z, *r = S.cg_r;
      ConvLog log;
      log.configure(maxiter, atol, rtol);
      const int grid = gmg_solver::grid_for(n);
      S.apply_A_resid(0, dx, db, r);                         // RichardsonLinearSolvers.jl:84-85
      double resn = S.norm(n, r);
      bool done = log.init(resn);                            // :86
      while (!done) {
        const double *dir = r;
        if (use_precond) { S.krylov_precond(use_precond, z, r, resn); dir = z; }   // :89
        hipLaunchKernelGGL(axpy_kernel, dim3(grid), dim3(256), 0, S.stream, n, omega, dir, dx);   // :90,98 x .+= w .* z
        HIP_CHECK(hipGetLastError());
        S.apply_A_resid(0, dx, db, r);                       // :91-92
        resn = S.norm(n, r);
        done = log.update(resn);                             // :93
      }
      S.out_vec(x, dx, n, memspace);
      log.export_to(res, hist, hist_cap, resn);              // :95
    });
  });
}

int gmg_op_apply(gmg_handle_t h, int lev, int op, const double *x, double *y, int memspace)
{
  return guarded(h, [&] {
    check_ready(h);
    check_level(h, lev, op != GMG_OP_A);
    REQUIRE(!h->inactive(lev), GMG_ERR_STATE, "this rank holds no part of that level (it lives on a rank subset, gmg_set_redistribution)");
    REQUIRE(x && y, GMG_ERR_INVALID, "null vector");
    Level &L = h->lev[lev];
    const DevCSR *M = op == GMG_OP_A ? &L.A : op == GMG_OP_P ? &L.P : op == GMG_OP_R ? &L.R : nullptr;
    REQUIRE(M, GMG_ERR_INVALID, "unknown operator");
    const int src = (op == GMG_OP_P) ? lev + 1 : lev;     // level the gathered vector lives on
    const double *dxv;
    if (h->comm.nranks > 1) {
      double *sx = h->scratch_vec(0, h->lev[0].nvec);
      HIP_CHECK(hipMemcpyAsync(sx, x, sizeof(double) * (size_t)h->lev[src].n,
                               memspace == GMG_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->stream));
      double *dy = memspace == GMG_MEM_DEVICE ? y : h->scratch_vec(1, h->lev[0].nvec);
      if (op == GMG_OP_A) h->apply_A_set(lev, sx, dy);      // own x own kernel overlapping the halo + boundary rows
      else {
        h->exchange(src, sx);
        h->spmv_set(*M, sx, dy);
        if (op == GMG_OP_R && L.r_split) h->add_ghost_part(L.rfix, sx, dy);     // (R of an own | ghost level: the ghost columns separately)
      }
      h->out_vec(y, dy, M->nrows, memspace);
      return;
    } else {
      dxv = h->in_vec(x, M->ncols, memspace, h->scratch_vec(0, h->lev[0].nvec));
    }
    double *dy = memspace == GMG_MEM_DEVICE ? y : h->scratch_vec(1, h->lev[0].nvec);
    h->spmv_set(*M, dxv, dy);
    h->out_vec(y, dy, M->nrows, memspace);
  });
}

int gmg_smooth(gmg_handle_t h, int lev, int which, double *x, double *r, int memspace)
{
  return guarded(h, [&] {
    check_ready(h);
    check_level(h, lev, true);
    REQUIRE(!h->inactive(lev), GMG_ERR_STATE, "this rank holds no part of that level (it lives on a rank subset, gmg_set_redistribution)");
    REQUIRE(x && r, GMG_ERR_INVALID, "null vector");
    REQUIRE(which == GMG_PRE || which == GMG_POST, GMG_ERR_INVALID, "which must be GMG_PRE or GMG_POST");
    Level &L = h->lev[lev];
    // a one-launch pass that times out writes nothing; x is saved, r is only replaced after the pass: the call is re-run per sweep
    h->with_persist_retry(x, L.n, memspace, false, [&] {
      double *dx = memspace == GMG_MEM_DEVICE ? x : h->scratch_vec(0, h->lev[0].nvec);
      if (memspace == GMG_MEM_HOST) h->h2d(dx, x, L.n);
      const double *dr = h->in_vec(r, L.n, memspace, h->scratch_vec(1, h->lev[0].nvec));
      double *rout = h->smooth(lev, which == GMG_PRE ? L.pre : L.post, dx, dr, false);
      HIP_CHECK(hipStreamSynchronize(h->stream));
      if (!h->persist_defer_throw) h->check_persistent();   // before the caller's r is replaced
      h->out_vec(r, rout, L.n, memspace);
      h->out_vec(x, dx, L.n, memspace);
    });
  });
}

int gmg_precond_apply(gmg_handle_t h, int lev, int which, const double *r, double *dx, int memspace)
{
  return guarded(h, [&] {
    check_ready(h);
    check_level(h, lev, true);
    REQUIRE(!h->inactive(lev), GMG_ERR_STATE, "this rank holds no part of that level (it lives on a rank subset, gmg_set_redistribution)");
    REQUIRE(r && dx, GMG_ERR_INVALID, "null vector");
    Level &L = h->lev[lev];
    Smoother &S = which == GMG_POST ? L.post : L.pre;
    const double *dr = h->in_vec(r, L.n, memspace, h->scratch_vec(1, h->lev[0].nvec));
    double *out = memspace == GMG_MEM_DEVICE ? dx : h->scratch_vec(0, h->lev[0].nvec);
    if (S.kind == SM_JACOBI) {
      hipLaunchKernelGGL(jacobi_apply_kernel, dim3(gmg_solver::grid_for(L.n)), dim3(256), 0, h->stream, L.n, L.dinv, dr, out);
      HIP_CHECK(hipGetLastError());
    } else if (h->comm.nranks > 1 && L.halo.present) {
      // solve!(x::PVector, ns::PatchNS, b::PVector), PatchSolvers.jl:227-236: consistent!(b), local solves, assemble!(x)
      double *rr = h->scratch_vec(2, h->lev[0].nvec), *oo = h->scratch_vec(3, h->lev[0].nvec);
      h->copy(rr, dr, L.n);
      h->exchange(lev, rr);
      h->patch_precond(L, S, rr, 1.0, false, oo, nullptr);
      out = oo;
    } else {
      h->patch_precond(L, S, dr, 1.0, false, out, nullptr);
    }
    h->out_vec(dx, out, L.n, memspace);
  });
}

int gmg_coarse_solve(gmg_handle_t h, const double *r, double *x, int memspace)
{
  return guarded(h, [&] {
    check_ready(h);
    REQUIRE(r && x, GMG_ERR_INVALID, "null vector");
    const int64_t n = h->lev[h->nlev - 1].n;
    const double *dr = h->in_vec(r, n, memspace, h->scratch_vec(1, h->lev[0].nvec));
    double *out = memspace == GMG_MEM_DEVICE ? x : h->scratch_vec(0, h->lev[0].nvec);
    h->coarse_solve(dr, out);
    h->out_vec(x, out, n, memspace);
  });
}

int gmg_dot(gmg_handle_t h, int64_t n, const double *a, const double *b, int memspace, double *out)
{
  return guarded(h, [&] {
    check_ready(h);
    REQUIRE(a && b && out && n >= 0, GMG_ERR_INVALID, "bad arguments");
    if (memspace == GMG_MEM_HOST) REQUIRE(n <= h->lev[0].n, GMG_ERR_INVALID, "host vectors longer than the finest level");
    const double *da = h->in_vec(a, n, memspace, h->scratch_vec(0, h->lev[0].nvec));
    const double *db = (a == b) ? da : h->in_vec(b, n, memspace, h->scratch_vec(1, h->lev[0].nvec));
    *out = h->dot(n, da, db);
  });
}

// ---- multi-GPU ------------------------------------------------------------------
int gmg_comm_unique_id(const char *rccl_path, char *id_out128)
{
  return guarded(nullptr, [&] {
    REQUIRE(id_out128, GMG_ERR_INVALID, "null id buffer");
    RcclApi api;
    std::string err;
    REQUIRE(api.load(rccl_path, err), GMG_ERR_COMM, err);
    NcclUniqueId id;
    const int rc = api.GetUniqueId(&id);
    REQUIRE(rc == 0, GMG_ERR_COMM, std::string("ncclGetUniqueId: ") + api.GetErrorString(rc));
    std::memcpy(id_out128, id.internal, 128);
  });
}

int gmg_comm_init_rccl(gmg_handle_t h, const char *rccl_path, const char *unique_id128, int rank, int nranks)
{
  return guarded(h, [&] {
    REQUIRE(h && unique_id128, GMG_ERR_INVALID, "null argument");
    REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, GMG_ERR_INVALID, "bad rank / nranks");
    REQUIRE(h->comm.kind == COMM_NONE, GMG_ERR_STATE, "communicator already initialised");
    std::string err;
    REQUIRE(h->comm.api.load(rccl_path, err), GMG_ERR_COMM, err);
    NcclUniqueId id;
    std::memcpy(id.internal, unique_id128, 128);
    const int rc = h->comm.api.CommInitRank(&h->comm.comm, nranks, id, rank);
    REQUIRE(rc == 0, GMG_ERR_COMM, std::string("ncclCommInitRank: ") + h->comm.api.GetErrorString(rc));
    h->comm.kind = COMM_RCCL; h->comm.rank = rank; h->comm.nranks = nranks;
    h->overlap = h->opt_int("GMG_OVERLAP", 1);
    if (!h->comm_stream) HIP_CHECK(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
    if (!h->ev_ready) HIP_CHECK(hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
    if (!h->ev_done) HIP_CHECK(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
    h->touch();
  });
}

// Diagnostic: one all-reduce and one grouped self send/recv through the RCCL binding
// (valid on a 1-rank communicator) -- proves the dlopen'ed signatures and enum values.
int gmg_comm_selftest(gmg_handle_t h, double *out2)
{
  return guarded(h, [&] {
    REQUIRE(h && out2, GMG_ERR_INVALID, "null argument");
    REQUIRE(h->comm.kind == COMM_RCCL, GMG_ERR_STATE, "RCCL communicator not initialised");
    double *d = nullptr;
    HIP_CHECK(hipMalloc((void **)&d, 4 * sizeof(double)));
    const double init[4] = {1.5, 42.0, 0.0, 0.0};
    HIP_CHECK(hipMemcpy(d, init, sizeof(init), hipMemcpyHostToDevice));
    int rc = h->comm.api.AllReduce(d, d, 1, kNcclDouble, kNcclSum, h->comm.comm, h->stream);
    REQUIRE(rc == 0, GMG_ERR_COMM, std::string("ncclAllReduce: ") + h->comm.api.GetErrorString(rc));
    rc = h->comm.api.GroupStart();
    if (rc == 0) rc = h->comm.api.Send(d + 1, 1, kNcclDouble, h->comm.rank, h->comm.comm, h->stream);
    if (rc == 0) rc = h->comm.api.Recv(d + 2, 1, kNcclDouble, h->comm.rank, h->comm.comm, h->stream);
    const int rc2 = h->comm.api.GroupEnd();
    REQUIRE(rc == 0 && rc2 == 0, GMG_ERR_COMM, std::string("self send/recv: ") + h->comm.api.GetErrorString(rc ? rc : rc2));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    double back[4];
    HIP_CHECK(hipMemcpy(back, d, sizeof(back), hipMemcpyDeviceToHost));
    (void)hipFree(d);
    out2[0] = back[0];   // = nranks * 1.5
    out2[1] = back[2];   // = 42.0
  });
}

// Latency of the two communication primitives as the solver issues them (DESIGN.md section 5's model is built on these):
//   out[0] stream time (us) of one halo exchange = ev_ready -> comm_stream: grouped ncclSend/ncclRecv of `nmsg` messages of `count`
//          doubles each -> ev_done -> main stream, with a 1-block kernel on the main stream between exchanges (the dependent
//          mat-vec of a latency-bound level); out[1] host time (us) to enqueue it;
//   out[2] stream time (us) of a 1-double ncclAllReduce + the square-root kernel (PVector norm); out[3] its host enqueue time;
//   out[4] the same exchange issued in-stream (no second stream, no events); out[5] an empty 1-block kernel (launch floor).
// Peers: rank +- 1, +- 2, ... (ring distance) on a multi-rank communicator, the rank itself on one rank (RCCL then runs its
// send/recv kernel without an xGMI hop: the launch + proxy floor of the path).
int gmg_comm_latency_probe(gmg_handle_t h, int nmsg, int64_t count, int reps, double *out6)
{
  return guarded(h, [&] {
    REQUIRE(h && out6, GMG_ERR_INVALID, "null argument");
    REQUIRE(h->comm.kind == COMM_RCCL, GMG_ERR_STATE, "RCCL communicator not initialised");
    REQUIRE(nmsg >= 1 && nmsg <= 26 && count >= 1 && reps >= 1, GMG_ERR_INVALID, "bad probe shape");
    const int nr = h->comm.nranks, me = h->comm.rank;
    double *snd = nullptr, *rcv = nullptr, *sc = nullptr;
    HIP_CHECK(hipMalloc((void **)&snd, sizeof(double) * (size_t)nmsg * (size_t)count));
    HIP_CHECK(hipMalloc((void **)&rcv, sizeof(double) * (size_t)nmsg * (size_t)count));
    HIP_CHECK(hipMalloc((void **)&sc, 8 * sizeof(double)));
    HIP_CHECK(hipMemset(snd, 0, sizeof(double) * (size_t)nmsg * (size_t)count));
    HIP_CHECK(hipMemset(sc, 0, 8 * sizeof(double)));
    hipEvent_t t0, t1;
    HIP_CHECK(hipEventCreate(&t0));
    HIP_CHECK(hipEventCreate(&t1));
    auto peer = [&](int k, bool up) {          // k-th message: ring distance k/2+1, alternating direction
      if (nr == 1) return me;
      const int d = (k / 2 + 1) % nr;
      return ((k & 1) == (up ? 0 : 1)) ? (me + d) % nr : (me - d + nr) % nr;
    };
    auto group = [&](hipStream_t st) {
      int rc = h->comm.api.GroupStart();
      for (int k = 0; k < nmsg && rc == 0; ++k) {
        rc = h->comm.api.Send(snd + (size_t)k * count, (size_t)count, kNcclDouble, peer(k, true), h->comm.comm, st);
        if (rc == 0) rc = h->comm.api.Recv(rcv + (size_t)k * count, (size_t)count, kNcclDouble, peer(k, false), h->comm.comm, st);
      }
      const int rc2 = h->comm.api.GroupEnd();
      REQUIRE(rc == 0 && rc2 == 0, GMG_ERR_COMM, std::string("latency probe send/recv: ") + h->comm.api.GetErrorString(rc ? rc : rc2));
    };
    auto tiny = [&]() { hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, h->stream, sc + 4, 1.0); };
    auto timed = [&](auto &&body, double &dev_us, double &host_us) {
      for (int w = 0; w < 3; ++w) body();                                   // warm-up (RCCL connects lazily)
      HIP_CHECK(hipStreamSynchronize(h->stream));
      if (h->comm_stream) HIP_CHECK(hipStreamSynchronize(h->comm_stream));
      HIP_CHECK(hipEventRecord(t0, h->stream));
      const auto c0 = std::chrono::steady_clock::now();
      for (int r = 0; r < reps; ++r) body();
      const auto c1 = std::chrono::steady_clock::now();
      HIP_CHECK(hipEventRecord(t1, h->stream));
      HIP_CHECK(hipEventSynchronize(t1));
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, t0, t1));
      dev_us = 1e3 * (double)ms / reps;
      host_us = std::chrono::duration<double, std::micro>(c1 - c0).count() / reps;
    };
    double d = 0, hst = 0;
    timed([&] { tiny(); HIP_CHECK(hipGetLastError()); }, d, hst);
    out6[5] = d;
    timed([&] {
      tiny();
      HIP_CHECK(hipEventRecord(h->ev_ready, h->stream));
      HIP_CHECK(hipStreamWaitEvent(h->comm_stream, h->ev_ready, 0));
      group(h->comm_stream);
      HIP_CHECK(hipEventRecord(h->ev_done, h->comm_stream));
      HIP_CHECK(hipStreamWaitEvent(h->stream, h->ev_done, 0));
    }, d, hst);
    out6[0] = d; out6[1] = hst;
    timed([&] {
      tiny();
      const int rc = h->comm.api.AllReduce(sc, sc, 1, kNcclDouble, kNcclSum, h->comm.comm, h->stream);
      REQUIRE(rc == 0, GMG_ERR_COMM, std::string("latency probe all-reduce: ") + h->comm.api.GetErrorString(rc));
      hipLaunchKernelGGL(sqrt_inplace_kernel, dim3(1), dim3(1), 0, h->stream, sc);
    }, d, hst);
    out6[2] = d; out6[3] = hst;
    timed([&] { tiny(); group(h->stream); }, d, hst);
    out6[4] = d;
    HIP_CHECK(hipStreamSynchronize(h->stream));
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    (void)hipFree(snd); (void)hipFree(rcv); (void)hipFree(sc);
  });
}

int gmg_comm_init_host(gmg_handle_t h, int rank, int nranks, gmg_host_exchange_fn xfn, gmg_host_allreduce_fn rfn, void *ctx)
{
  return guarded(h, [&] {
    REQUIRE(h && xfn && rfn, GMG_ERR_INVALID, "null argument");
    REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, GMG_ERR_INVALID, "bad rank / nranks");
    REQUIRE(h->comm.kind == COMM_NONE, GMG_ERR_STATE, "communicator already initialised");
    h->comm.kind = COMM_HOST; h->comm.rank = rank; h->comm.nranks = nranks;
    h->comm.xfn = xfn; h->comm.rfn = rfn; h->comm.ctx = ctx;
    h->host_async = h->opt_int("GMG_HOST_ASYNC", 0);
    h->overlap = h->opt_int("GMG_OVERLAP", 1);
    if (h->host_async) {
      if (!h->comm_stream) HIP_CHECK(hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
      if (!h->ev_ready) HIP_CHECK(hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
      if (!h->ev_done) HIP_CHECK(hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming));
    }
    h->touch();
  });
}

int gmg_comm_set_loopback(gmg_handle_t h, int virtual_nranks)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    REQUIRE(h->comm.kind != COMM_NONE, GMG_ERR_STATE, "gmg_comm_init_rccl / gmg_comm_init_host first");
    REQUIRE(h->comm.nranks == 1 && !h->comm.loopback, GMG_ERR_STATE, "loopback needs a communicator of exactly one rank");
    REQUIRE(virtual_nranks >= 2, GMG_ERR_INVALID, "virtual_nranks must be >= 2");
    h->comm.loopback = true; h->comm.rank = 0; h->comm.nranks = virtual_nranks;
    h->touch();
  });
}

int gmg_set_partition(gmg_handle_t h, int lev, int64_t n_own, int64_t n_ghost, int nnbr, const int32_t *nbr_rank,
                      const int64_t *snd_ptr, const int64_t *snd_idx, const int64_t *rcv_ptr)
{
  return guarded(h, [&] {
    const int slot = slot_of(h, lev);
    REQUIRE(n_own >= 0 && n_ghost >= 0 && nnbr >= 0, GMG_ERR_INVALID, "negative sizes");
    REQUIRE(nnbr == 0 || (nbr_rank && snd_ptr && rcv_ptr), GMG_ERR_INVALID, "null neighbour arrays");
    fill_plan(h->lev[slot].halo, h->comm, n_own, n_ghost, nnbr, nbr_rank, snd_ptr, snd_idx, rcv_ptr);
    h->touch();
  });
}

int gmg_set_partition_overlap_hints(gmg_handle_t h, int lev, int exact_node_layers, int layers_per_sweep, int restriction_reach,
                                    int correction_exact_near_owned)
{
  return guarded(h, [&] {
    check_level(h, lev, true);
    REQUIRE(exact_node_layers >= 0 && layers_per_sweep >= 0 && restriction_reach >= 0, GMG_ERR_INVALID, "negative layer counts");
    Level &L = h->lev[lev];
    REQUIRE(L.halo.present && L.halo.ovl, GMG_ERR_INVALID, "overlap hints need gmg_set_partition_overlap on this level first");
    L.ovl_layers = exact_node_layers; L.ovl_sweep_reach = layers_per_sweep; L.ovl_r_reach = restriction_reach;
    L.ovl_skip_dx = correction_exact_near_owned != 0;
    h->touch();
  });
}

int gmg_set_krylov_map(gmg_handle_t h, const int64_t *own_to_local, int64_t n_own)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    REQUIRE(n_own >= 0 && (n_own == 0 || own_to_local), GMG_ERR_INVALID, "null map");
    h->h_own2loc.assign(own_to_local, own_to_local + n_own);
    h->touch();
  });
}

int gmg_set_partition_overlap(gmg_handle_t h, int lev, int64_t n_local, int64_t n_ghost, int depth, int nnbr, const int32_t *nbr_rank,
                              const int64_t *snd_ptr, const int64_t *snd_idx, const int64_t *rcv_ptr, const int64_t *rcv_idx)
{
  return guarded(h, [&] {
    check_level(h, lev, false);
    REQUIRE(n_local >= 0 && n_ghost >= 0 && n_ghost <= n_local && nnbr >= 0, GMG_ERR_INVALID, "bad sizes");
    REQUIRE(nnbr == 0 || (nbr_rank && snd_ptr && rcv_ptr), GMG_ERR_INVALID, "null neighbour arrays");
    REQUIRE(n_ghost == 0 || rcv_idx, GMG_ERR_INVALID, "null rcv_idx");
    static const int64_t none = 0;
    fill_plan(h->lev[lev].halo, h->comm, n_local - n_ghost, n_ghost, nnbr, nbr_rank, snd_ptr, snd_idx, rcv_ptr, rcv_idx ? rcv_idx : &none, depth);
    h->lev[lev].clear_overlap_hints();                       // stated for the halo that was here before
    if (lev > 0) h->lev[lev - 1].ovl_skip_dx = false;        // "P's rows are complete" was a statement about THIS level's layout too
    h->touch();
  });
}

int gmg_get_comm_info(gmg_handle_t h, int *transport, int *rank, int *nranks, int *comm_count, int *comm_device)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    if (transport) *transport = h->comm.kind;
    if (rank) *rank = h->comm.rank;
    if (nranks) *nranks = h->comm.nranks;
    int cnt = -1, dev = -1;
    if (h->comm.kind == COMM_RCCL && h->comm.comm) {
      if (h->comm.api.CommCount && h->comm.api.CommCount(h->comm.comm, &cnt) != 0) cnt = -1;
      if (h->comm.api.CommCuDevice && h->comm.api.CommCuDevice(h->comm.comm, &dev) != 0) dev = -1;
    }
    if (comm_count) *comm_count = cnt;                       // what ncclCommCount says (-1: not an RCCL communicator)
    if (comm_device) *comm_device = dev >= 0 ? dev : h->device;
  });
}

int gmg_get_comm_stats(gmg_handle_t h, int64_t *n_exchanges, int64_t *n_allreduces)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    if (n_exchanges) *n_exchanges = h->n_exchanges + h->n_redist;   // (redistributions between two partitions are exchanges too)
    if (n_allreduces) *n_allreduces = h->n_allreduces;
  });
}

int gmg_set_redistribution(gmg_handle_t h, int lev, int member, int64_t n_glue_own, int64_t n_glue_ghost,
                           const gmg_redist_plan *to_sub, const gmg_redist_plan *from_sub)
{
  return guarded(h, [&] {
    check_level(h, lev, true);
    REQUIRE(lev >= 1, GMG_ERR_INVALID, "the finest level lives on all ranks");
    REQUIRE(h->comm.nranks > 1, GMG_ERR_STATE, "gmg_set_redistribution needs gmg_comm_init_* first");
    REQUIRE(to_sub && from_sub && n_glue_own >= 0 && n_glue_ghost >= 0, GMG_ERR_INVALID, "null plan / negative sizes");
    gmg_solver::Redist R;
    R.present = true; R.member = member != 0;
    R.n_glue_own = n_glue_own; R.n_glue_ghost = n_glue_ghost;
    const gmg_redist_plan *in[2] = {to_sub, from_sub};
    HaloPlan *out[2] = {&R.to_sub, &R.from_sub};
    for (int k = 0; k < 2; ++k) {
      const gmg_redist_plan &p = *in[k];
      HaloPlan &H = *out[k];
      REQUIRE(p.nnbr >= 0 && p.nself >= 0, GMG_ERR_INVALID, "negative sizes");
      REQUIRE(p.nnbr == 0 || (p.nbr_rank && p.snd_ptr && p.rcv_ptr), GMG_ERR_INVALID, "null neighbour arrays");
      REQUIRE(p.nself == 0 || (p.self_src && p.self_dst), GMG_ERR_INVALID, "null self lists");
      H.present = true; H.ovl = true; H.depth = 1;
      H.nbr.assign(p.nbr_rank, p.nbr_rank + p.nnbr);
      H.snd_ptr.assign(1, 0); H.rcv_ptr.assign(1, 0);
      if (p.nnbr > 0) { H.snd_ptr.assign(p.snd_ptr, p.snd_ptr + p.nnbr + 1); H.rcv_ptr.assign(p.rcv_ptr, p.rcv_ptr + p.nnbr + 1); }
      REQUIRE(H.snd_ptr[0] == 0 && H.rcv_ptr[0] == 0, GMG_ERR_INVALID, "snd_ptr / rcv_ptr must start at 0");
      for (int q = 0; q < p.nnbr; ++q) {
        REQUIRE(H.snd_ptr[q] <= H.snd_ptr[q + 1] && H.rcv_ptr[q] <= H.rcv_ptr[q + 1], GMG_ERR_INVALID, "pointers not monotone");
        REQUIRE(p.nbr_rank[q] >= 0 && p.nbr_rank[q] < h->comm.nranks && p.nbr_rank[q] != h->comm.rank, GMG_ERR_INVALID, "bad neighbour rank");
      }
      REQUIRE((H.snd_ptr.back() == 0 || p.snd_idx) && (H.rcv_ptr.back() == 0 || p.rcv_idx), GMG_ERR_INVALID, "null index lists");
      H.h_snd_idx.assign(p.snd_idx, p.snd_idx + H.snd_ptr.back());
      H.h_rcv_idx.assign(p.rcv_idx, p.rcv_idx + H.rcv_ptr.back());
      H.n_own = 0; H.n_ghost = H.rcv_ptr.back();             // (n_ghost: what the landing buffer of the scatter holds)
      R.h_self[2 * k].assign(p.self_src, p.self_src + p.nself);
      R.h_self[2 * k + 1].assign(p.self_dst, p.self_dst + p.nself);
    }
    for (HaloPlan *H : {&h->redist.to_sub, &h->redist.from_sub}) {
      if (H->h_send) (void)hipHostFree(H->h_send);
      if (H->h_recv) (void)hipHostFree(H->h_recv);
    }
    h->redist = std::move(R);
    h->sub_from = lev;
    h->touch();
  });
}

int gmg_set_replication(gmg_handle_t h, int lev, const int64_t *own_global_ids, int64_t n_own)
{
  return guarded(h, [&] {
    check_level(h, lev, false);
    REQUIRE(lev >= 1, GMG_ERR_INVALID, "the finest level cannot be replicated");
    REQUIRE(n_own == 0 || own_global_ids, GMG_ERR_INVALID, "null own_global_ids");
    h->rep_from = lev;
    h->h_rep_gid.assign(own_global_ids, own_global_ids + n_own);
    h->touch();
  });
}

int gmg_profile_enable(gmg_handle_t h, int lev, int enable)
{
  return guarded(h, [&] {
    REQUIRE(h, GMG_ERR_INVALID, "null handle");
    if (!enable) { h->prof_level = -1; return; }
    check_level(h, lev, true);
    if (h->prof_ev.empty()) {
      h->prof_ev.resize(2 * 8192);
      h->prof_w.assign(8192, 1);
      h->prof_xm.assign(8192, 0);
      for (auto &ev : h->prof_ev) HIP_CHECK(hipEventCreate(&ev));
    }
    h->prof_level = lev;
    h->prof_stride = std::max(1, h->opt_int("GMG_PROF_STRIDE", 7));   // live: a caller picks the sampling density per measurement
    h->prof_seq = 0;
    h->prof_used = 0;
    h->prof_ms = 0.0;
    h->prof_launches = 0; h->prof_fused = 0; h->prof_patch = false;
    for (int v = 0; v < 3; ++v) { h->prof_ms_v[v] = 0.0; h->prof_n_v[v] = 0; }
  });
}

int gmg_get_kernel_stats(gmg_handle_t h, gmg_kernel_stats *out)
{
  return guarded(h, [&] {
    check_ready(h);
    REQUIRE(out, GMG_ERR_INVALID, "null output");
    HIP_CHECK(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]));
      h->prof_ms += ms;
      h->prof_launches += h->prof_w[i / 2];
      if (h->prof_w[i / 2] > 2) h->prof_fused += 1;          // (a weight of 2 is a fused PAIR of sweeps, sells_z2sweep_kernel: not a one-launch pass)
      const int v = std::min(2, std::max(0, (int)h->prof_xm[i / 2]));
      h->prof_ms_v[v] += ms; h->prof_n_v[v] += h->prof_w[i / 2];
    }
    h->prof_used = 0;
    const int l = h->prof_level >= 0 ? h->prof_level : 0;
    const Level &L = h->lev[l];
    out->launches = h->prof_launches;
    out->fused_passes = h->prof_fused;
    out->total_ms = h->prof_ms;
    out->rows = L.n;
    out->nnz = L.A.nnz_model >= 0 ? L.A.nnz_model : L.A.nnz;
    // B_sweep = 12 Z + 68 N  (fp64 value + int32 column per nnz; row pointer,
    // r, D^-1, x, dx read / dx, x, Adx, r written as in RichardsonSmoothers.jl:91-95)
    out->alg_bytes = 12.0 * (double)out->nnz + 68.0 * (double)L.n;
    out->layout_bytes = h->sweep_layout_bytes(l);
    if (h->prof_patch) {
      // patch-smoother level: the timed kernel is the operator mat-vec r -= A dx (GMGLinearSolvers.jl:495-496 / RichardsonSmoothers.jl:94-95):
      // B = 12 Z + 28 N; as stored in the row-pattern layout: pattern id 2 B + dx 8 B + r in / out 16 B per row
      out->alg_bytes = 12.0 * (double)out->nnz + 28.0 * (double)L.n;
      if (L.A.pat) out->layout_bytes = ((L.A.rowbase ? 6.0 : 2.0) + 24.0) * (double)L.n;
      else out->layout_bytes = h->sweep_layout_bytes(l) - 32.0 * (double)L.n;
    }
  });
}

int gmg_get_kernel_stats_by_variant(gmg_handle_t h, double total_ms[3], int64_t launches[3], double layout_bytes[3])
{
  return guarded(h, [&] {
    check_ready(h);
    REQUIRE(total_ms && launches && layout_bytes, GMG_ERR_INVALID, "null output");
    gmg_kernel_stats tmp;
    const int st = gmg_get_kernel_stats(h, &tmp);            // folds the pending event pairs into the accumulators
    REQUIRE(st == GMG_OK, st, "gmg_get_kernel_stats failed");
    const int l = h->prof_level >= 0 ? h->prof_level : 0;
    for (int v = 0; v < 3; ++v) {
      total_ms[v] = h->prof_ms_v[v]; launches[v] = h->prof_n_v[v];
      layout_bytes[v] = h->prof_patch ? tmp.layout_bytes : h->sweep_layout_bytes_variant(l, v);
    }
  });
}

int gmg_model_bytes(gmg_handle_t h, double *vcycle_bytes, double *cg_iter_bytes)
{
  return guarded(h, [&] {
    check_ready(h);
    double BV = 0.0;
    for (int l = 0; l < h->nlev - 1; ++l) {
      const Level &L = h->lev[l];
      const double Z = (double)L.A.nnz, N = (double)L.n, ZP = (double)L.P.nnz, NH = (double)h->lev[l + 1].n;
      const double nu2 = (double)(L.pre.niter + L.post.niter);
      BV += nu2 * (12.0 * Z + 68.0 * N) + (12.0 * Z + 28.0 * N) + 24.0 * ZP + 36.0 * N + 28.0 * NH;
    }
    const double NL = (double)h->lev[h->nlev - 1].n;
    BV += 8.0 * NL * NL + 16.0 * NL;
    if (vcycle_bytes) *vcycle_bytes = BV;
    if (cg_iter_bytes) *cg_iter_bytes = BV + 12.0 * (double)h->lev[0].A.nnz + 132.0 * (double)h->lev[0].n;
  });
}

int gmg_sweep_signature(gmg_handle_t h, int lev, char *buf, int cap)
{
  return guarded(h, [&] {
    check_ready(h);
    check_level(h, lev, false);
    REQUIRE(!h->inactive(lev), GMG_ERR_STATE, "this rank holds no part of that level (it lives on a rank subset, gmg_set_redistribution)");
    REQUIRE(buf && cap > 0, GMG_ERR_INVALID, "null buffer");
    std::snprintf(buf, (size_t)cap, "%s", h->lev[lev].A.sweep_sig);
  });
}

int gmg_level_format(gmg_handle_t h, int lev, int *sell, int *vdict, int *idx16, double *stream_bytes_per_nnz, double *padding)
{
  return guarded(h, [&] {
    check_ready(h);
    check_level(h, lev, false);
    REQUIRE(!h->inactive(lev), GMG_ERR_STATE, "this rank holds no part of that level (it lives on a rank subset, gmg_set_redistribution)");
    const DevCSR &A = h->lev[lev].A;
    if (sell) *sell = A.pat ? 2 : (A.sell ? (A.opat ? 3 : 1) : 0);
    if (vdict) *vdict = A.vdict ? 1 : 0;
    if (idx16) *idx16 = A.comp_idx ? 1 : 0;
    if (stream_bytes_per_nnz) *stream_bytes_per_nnz = (A.sell && (A.pat || A.vdict || A.comp_idx || A.opat)) ? A.stream_bytes_per_nnz : 12.0;
    if (padding) *padding = A.sell && A.nnz > 0 ? (double)A.zpad / (double)A.nnz : 1.0;
  });
}

// Streaming ceiling of this device, measured with the library's own copy kernel (16 B/lane, non-temporal) on the handle's
// stream: the figure bench.py reports next to the 8 TB/s spec, and -- its byte count being exact -- the kernel the
// FETCH_SIZE / WRITE_SIZE counter corrections of profiles/summarize.py are calibrated on.
int gmg_stream_probe(gmg_handle_t h, int64_t nbytes, int reps, double *gbytes_per_s)
{
  return guarded(h, [&] {
    REQUIRE(h && gbytes_per_s, GMG_ERR_INVALID, "null argument");
    REQUIRE(nbytes >= 4096 && reps >= 1, GMG_ERR_INVALID, "bad probe size");
    const int64_t n2 = nbytes / 16;
    double *src = nullptr, *dst = nullptr;
    HIP_CHECK(hipMalloc((void **)&src, (size_t)n2 * 16));
    if (hipMalloc((void **)&dst, (size_t)n2 * 16) != hipSuccess) { (void)hipFree(src); throw GmgError{GMG_ERR_ALLOC, "probe buffers"}; }
    HIP_CHECK(hipMemsetAsync(src, 0, (size_t)n2 * 16, h->stream));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    const int grid = (int)std::min<int64_t>((n2 + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, h->stream, n2, src, dst);   // warm-up
    HIP_CHECK(hipEventRecord(e0, h->stream));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, h->stream, n2, src, dst);
    HIP_CHECK(hipEventRecord(e1, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(dst);
    *gbytes_per_s = 2.0 * 16.0 * (double)n2 * reps / ((double)ms * 1e-3) / 1e9;
  });
}

int gmg_stream_probe_read(gmg_handle_t h, int64_t nbytes, int reps, double *gbytes_per_s)
{
  return guarded(h, [&] {
    REQUIRE(h && gbytes_per_s, GMG_ERR_INVALID, "null argument");
    REQUIRE(nbytes >= 4096 && reps >= 1, GMG_ERR_INVALID, "bad probe size");
    const int64_t n2 = nbytes / 16;
    double *src = nullptr, *sink = nullptr;
    HIP_CHECK(hipMalloc((void **)&src, (size_t)n2 * 16));
    if (hipMalloc((void **)&sink, 256 * 32 * 256 * sizeof(double)) != hipSuccess) { (void)hipFree(src); throw GmgError{GMG_ERR_ALLOC, "probe buffers"}; }
    HIP_CHECK(hipMemsetAsync(src, 0, (size_t)n2 * 16, h->stream));
    hipEvent_t e0, e1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    const int grid = (int)std::min<int64_t>((n2 + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(256), 0, h->stream, n2, src, sink);   // warm-up
    HIP_CHECK(hipEventRecord(e0, h->stream));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(256), 0, h->stream, n2, src, sink);
    HIP_CHECK(hipEventRecord(e1, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(src); (void)hipFree(sink);
    *gbytes_per_s = 16.0 * (double)n2 * reps / ((double)ms * 1e-3) / 1e9;
  });
}

int gmg_device_bytes(gmg_handle_t h, int64_t *bytes)
{
  return guarded(h, [&] {
    REQUIRE(h && bytes, GMG_ERR_INVALID, "null argument");
    *bytes = h->dev_bytes;
  });
}

} // extern "C"

#include "block.inc.hpp"
