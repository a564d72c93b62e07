/*
 * gmg_amd.h -- C ABI of libgmgamd.so, the MI355X (gfx950) geometric-multigrid
 * V-cycle / CG / FGMRES hot path for GridapSolvers.jl.
 *
 * The reference (GridapSolvers.jl v0.7.1) is pure Julia and has no FFI for this
 * path; its drop-in boundary is the Gridap.Algebra interface
 *     LinearSolver -> symbolic_setup -> numerical_setup(!) -> solve!
 * Each entry point below names the reference method(s) (file:line) whose work
 * it takes over; the Julia wrapper that binds them with `ccall` is in
 * gridapsolvers.jl_amd/julia/GridapSolversAMD.jl and INTEGRATION.md.
 *
 * Conventions
 *   - every function returns an int status (GMG_OK == 0); nothing throws or
 *     longjmps across the boundary; gmg_last_error() gives the message.
 *   - all pointers passed to setters are HOST pointers borrowed for the call
 *     only.  Vector arguments of the solve / operator calls are host or device
 *     pointers according to `memspace`.
 *   - one handle = one HIP device + one HIP stream; not thread-safe per handle.
 *   - levels are numbered 0 .. nlevels-1, level 0 = finest (reference: 1 = finest).
 *   - fp64 values; integer indices int32 or int64, 0- or 1-based, CSR or CSC.
 */
#ifndef GMG_AMD_H
#define GMG_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMG_API __attribute__((visibility("default")))

typedef struct gmg_solver *gmg_handle_t;

enum gmg_status {
  GMG_OK = 0,
  GMG_ERR_INVALID = 1,     /* bad argument / inconsistent sizes (reference: @check failures) */
  GMG_ERR_HIP = 2,         /* a HIP runtime call failed */
  GMG_ERR_STATE = 3,       /* call order violated (e.g. solve before setup) */
  GMG_ERR_ALLOC = 4,
  GMG_ERR_COMM = 5,        /* RCCL failure */
  GMG_ERR_UNSUPPORTED = 6,
  GMG_ERR_SINGULAR = 7     /* zero pivot / zero diagonal */
};

enum gmg_layout { GMG_CSR = 0, GMG_CSC = 1 };
enum gmg_memspace { GMG_MEM_HOST = 0, GMG_MEM_DEVICE = 1 };
/* GMGLinearSolvers.jl:56 mode ; :57 cycle_type */
enum gmg_mode { GMG_MODE_PRECONDITIONER = 0, GMG_MODE_SOLVER = 1 };
enum gmg_cycle { GMG_V_CYCLE = 0, GMG_W_CYCLE = 1, GMG_F_CYCLE = 2 };
/* SolverInterfaces/SolverTolerances.jl:11-16 SolverConvergenceFlag */
enum gmg_conv_flag {
  GMG_CONVERGED_ATOL = 0,
  GMG_CONVERGED_RTOL = 1,
  GMG_DIVERGED_MAXITER = 2,
  GMG_DIVERGED_BREAKDOWN = 3
};
enum gmg_which { GMG_PRE = 0, GMG_POST = 1, GMG_PRE_AND_POST = 2 };
enum gmg_patch_kind {
  GMG_PATCH_LU = 0,        /* PatchSolver: lu!(A[p,p]) with partial pivoting, PatchSolvers.jl:176 */
  GMG_PATCH_NOPIVOT = 1    /* BlockJacobiSolver: lu!(A[p,p],NoPivot()), BlockJacobiSolvers.jl:162 */
};
enum gmg_op { GMG_OP_A = 0, GMG_OP_P = 1, GMG_OP_R = 2 };

/* Result of an iterative solve: mirrors ConvergenceLog (ConvergenceLogs.jl:42-60). */
typedef struct {
  int32_t niters;          /* log.num_iters */
  int32_t flag;            /* finalize!(log,res) -> gmg_conv_flag */
  double res0;             /* log.residuals[1] */
  double res;              /* last residual */
} gmg_result;

/* Timing / traffic statistics of the profiled kernel class (gmg_profile_enable). */
typedef struct {
  int64_t launches;        /* sweeps measured (one per launch, except on levels that run a whole smoothing pass as ONE
                              launch -- see fused_passes -- where a pass of niter sweeps counts niter) */
  double total_ms;         /* sum of HIP-event durations on the handle's stream */
  double alg_bytes;        /* algorithmic bytes of ONE launch (SURVEY 8d byte model) */
  int64_t rows, nnz;       /* shape of the operator the kernel streams */
  double layout_bytes;     /* bytes ONE launch moves with the storage layout chosen at setup (matrix stream as
                              stored + row-wise vectors, each once): equals alg_bytes only for the plain 12 B/nnz
                              layouts; far smaller for the row-pattern / dictionary layouts */
  int64_t fused_passes;    /* measured smoothing passes that ran as one launch (sells_smooth_kernel: small single-GPU
                              levels in the row-pattern layout); 0 on levels swept launch by launch */
} gmg_kernel_stats;

/* ---- lifetime ------------------------------------------------------------- */
/* GMGLinearSolver(...) constructor, GMGLinearSolvers.jl:48-69 (allocates nothing on device yet). */
GMG_API int gmg_create(gmg_handle_t *h, int nlevels, int device_id);
/* finalizer of the numerical setup (pattern: ext/PardisoExt.jl:54-61). */
GMG_API int gmg_destroy(gmg_handle_t h);
GMG_API const char *gmg_last_error(gmg_handle_t h); /* h may be NULL: last global error */
GMG_API int gmg_version(void);

/* ---- operators (numerical_setup inputs) ------------------------------------ */
/* smatrices[lev+1], GMGLinearSolvers.jl:183-185,336-340.  Square. */
GMG_API int gmg_set_matrix(gmg_handle_t h, int lev, int64_t nrows, int64_t ncols, int64_t nnz,
                           const void *ptr, const void *idx, const double *val,
                           int layout, int index_base, int index_bytes);
/* The same operators (op = GMG_OP_A, GMG_OP_P or GMG_OP_R of level lev) handed over as a STREAM of consecutive row blocks,
 * for hierarchies whose CSR the host cannot or need not hold at once (BASELINE config 3: 8.5e9 nonzeros on the finest level).
 * Each call passes rows [row0, row0 + nrows_block) as a standalone CSR (ptr has nrows_block+1 entries starting at
 * index_base, columns are global, sorted and duplicate-free); row0 = 0 (re)starts the stream, blocks arrive in order, the
 * operator is complete when row0 + nrows_block = nrows_total.  The library keeps only the row-pattern form (a pattern id
 * per row + the dictionary of distinct rows): host memory is O(block).  Operators with more than 4096 distinct rows
 * (variable coefficients, unstructured meshes) are rejected with GMG_ERR_UNSUPPORTED -- pass those whole.
 * A streamed P needs a streamed R (R = P^T is not formed from a stream); the coarsest matrix is passed whole unless the
 * coarse solver is iterative / a callback.  Several ranks: levels that are laid out like a single-GPU level -- the overlapping
 * layout (gmg_set_partition_overlap: one square local operator over the extended box) and the replicated levels -- keep the stream
 * form, transfers included; the matrix of an own | ghost level (gmg_set_partition, called BEFORE the first block) is streamed with
 * its local shape n_own x (n_own + n_ghost): every block is split into the own x own part, which goes to the stream, and the
 * entries in ghost columns, which become the small CSR the boundary fix-up applies once the halo has arrived. */
GMG_API int gmg_set_operator_rows(gmg_handle_t h, int lev, int op, int64_t nrows_total, int64_t ncols, int64_t row0,
                                  int64_t nrows_block, const void *ptr, const void *idx, const double *val,
                                  int index_base, int index_bytes);
/* Structured operators: append `count` further copies of the LAST nrows_block rows handed over, copy k with all its column
 * indices shifted by k * col_shift (uniform meshes: every interior node plane repeats the previous one).  No arrays cross the
 * boundary and nothing is hashed again; for level matrices col_shift must equal nrows_block.  No reference counterpart (the
 * reference assembles every row, GMGLinearSolvers.jl:342-353). */
GMG_API int gmg_set_operator_rows_repeat(gmg_handle_t h, int lev, int op, int64_t nrows_block, int64_t count, int64_t col_shift);
/* numerical_setup!(ns,A): same pattern, new values in the handle's 0-based CSR order (GMGLinearSolvers.jl:249-297;
 * JacobiLinearSolvers.jl:25-27), any level.  Requires gmg_setup to be called again; when nothing but values changed
 * since the last setup and the refreshed levels are stored with explicit values (SELL-64 / CSR-stream: what
 * variable-coefficient operators get), that gmg_setup keeps every layout, table and work vector and only rewrites the
 * value arrays, D^-1, the patch inverse blocks and the coarse inverse on the device -- bit-identical to a fresh setup.
 * Dictionary / row-pattern layouts depend on the values themselves: those levels trigger a full setup. */
GMG_API int gmg_update_values(gmg_handle_t h, int lev, const double *val);
/* The same for a level whose matrix was handed to gmg_set_matrix in GMG_CSC layout (Julia's SparseMatrixCSC, Gridap's default):
 * `val` = the new nzval in CSC order.  The first call on a level also takes the (unchanged) colptr / rowidx and records where
 * the transposition of gmg_set_matrix put every entry; later calls may pass NULL for both and cost one parallel scatter -- no
 * `sparse(transpose(A))` on the caller's side.  numerical_setup!(ns,A) of the weak-form variant refreshes EVERY level this way
 * (GMGLinearSolvers.jl:260-297: smatrices, smoothers and the coarsest solver are all recomputed). */
GMG_API int gmg_update_values_csc(gmg_handle_t h, int lev, const void *colptr, const void *rowidx, const double *val,
                                  int index_base, int index_bytes);
/* interp[lev+1] : level lev+1 -> lev, `mul!(dxh,interp,dxH)` GMGLinearSolvers.jl:491
 * (y = P x, GridTransferOperators.jl:391-401).  nrows = n(lev), ncols = n(lev+1). */
GMG_API int gmg_set_prolongation(gmg_handle_t h, int lev, int64_t nrows, int64_t ncols, int64_t nnz,
                                 const void *ptr, const void *idx, const double *val,
                                 int layout, int index_base, int index_bytes);
/* restrict[lev+1] : level lev -> lev+1, `mul!(rH,restrict,rh)` :484.  Optional:
 * when absent R = P^T is built (GridTransferOperators.jl:202-209,536-547). */
GMG_API int gmg_set_restriction(gmg_handle_t h, int lev, int64_t nrows, int64_t ncols, int64_t nnz,
                                const void *ptr, const void *idx, const double *val,
                                int layout, int index_base, int index_bytes);

/* ---- smoothers ------------------------------------------------------------- */
/* RichardsonSmoother(JacobiLinearSolver(),niter,omega): RichardsonSmoothers.jl:84-98,
 * JacobiLinearSolvers.jl:20-23,43-47. */
GMG_API int gmg_set_smoother_jacobi(gmg_handle_t h, int lev, int which, int niter, double omega);
/* RichardsonSmoother(PatchSolver|BlockJacobiSolver,niter,omega): PatchSolvers.jl:279-300,
 * BlockJacobiSolvers.jl:141-170.  patch_ptr has npatch+1 entries; patch_dofs lists
 * the rows(=cols) of each patch; blocks A[p,p] are extracted and factorised on the device. */
GMG_API int gmg_set_smoother_patch(gmg_handle_t h, int lev, int which, int niter, double omega,
                                   int kind, int64_t npatch, const void *patch_ptr,
                                   const void *patch_dofs, int index_base, int index_bytes);

/* The same smoother with the patch data the reference's numerical setup holds (PatchSolvers.jl:137-150,175-188):
 *   patch_rows / patch_cols  separate index tables (b[rows_p] is the local rhs, x[cols_p] += x_p; PatchSolvers.jl:237-240,296);
 *                            patch_cols = NULL means patch_cols = patch_rows;
 *   blocks                   NULL: blocks A[rows_p, cols_p] are gathered from the level matrix on the device (BlockJacobiSolvers.jl:160);
 *                            else the caller's own n_p x n_p patch matrices, COLUMN-major (Julia Matrix{Float64}), concatenated in
 *                            patch order -- assemble_matrix(biform, assem, trial, test) of the SOLVER'S weak form, which need not equal A[p,p];
 *   blocks_are_factors = 1   `blocks` hold the packed output of lu!(patch_mat) and `pivots` its 1-based LAPACK ipiv (patch_ptr layout;
 *                            NULL for NoPivot factors): the library forms F \ I exactly as ldiv! would and never re-factorises.
 * kind (GMG_PATCH_LU / GMG_PATCH_NOPIVOT) selects the pivoting of the device factorisation when matrices are given. */
GMG_API int gmg_set_smoother_patch_matrices(gmg_handle_t h, int lev, int which, int niter, double omega, int kind,
                                            int64_t npatch, const void *patch_ptr, const void *patch_rows,
                                            const void *patch_cols, int index_base, int index_bytes,
                                            const double *blocks, int blocks_are_factors, const int32_t *pivots);

/* Patch-corrected prolongation  y = P x - sum_p R_p^T A_pp^-1 R_p (A P x)  over the given patches
 * (PatchProlongationOperator, PatchBasedSmoothers/PatchTransferOperators.jl:153-172 with rhs = lhs = the
 * level operator; used by the reference for grad-div problems, test/Applications/StokesGMG.jl:125-133).
 * Same patch-table format as gmg_set_smoother_patch. */
GMG_API int gmg_set_prolongation_patch_correction(gmg_handle_t h, int lev, int kind, int64_t npatch,
                                                  const void *patch_ptr, const void *patch_dofs,
                                                  int index_base, int index_bytes);

/* rhs form of that correction when it differs from the level operator: PatchProlongationOperator(lev,sh,ptopo,lhs,rhs)
 * solves lhs(u_i,v_i) = rhs(uH,v_i) (PatchTransferOperators.jl:30-50); the Stokes application passes lhs = the velocity
 * form and rhs = the grad-div term only (test/Applications/StokesGMG.jl:125-127).  G = the assembled rhs form on level lev
 * (square, same format options as gmg_set_matrix); lhs patch blocks: A[p,p], or the caller's via the patch tables.   On a distributed level (gmg_set_partition first) the matrix holds this rank's rows
 * with [own | ghost] columns, and every correction patch must lie inside the owned dofs (coarse-cell interiors do). */
GMG_API int gmg_set_prolongation_patch_correction_rhs(gmg_handle_t h, int lev, int64_t n, int64_t nnz, const void *ptr,
                                                      const void *idx, const double *val, int layout, int index_base,
                                                      int index_bytes);

/* ---- coarsest solver ---------------------------------------------------------- */
/* kwarg `coarsest_solver` (GMGLinearSolvers.jl:54; cache :423-434; applied at :474).  The reference accepts any LinearSolver:
 * default LUSolver(); its MPI tests / applications pass iterative or PETSc solvers (joss_paper/scalability/src/stokes_gmg.jl:41-63). */
enum gmg_coarse_kind {
  GMG_COARSE_DENSE_INVERSE = 0,  /* LUSolver(): exact; dense inverse built at setup, one GEMV per cycle (default) */
  GMG_COARSE_CG_JACOBI = 1,      /* CGSolver(JacobiLinearSolver(); maxiter, atol, rtol) on the device, x0 = 0 (CGSolvers.jl:73-120) */
  GMG_COARSE_HOST_CALLBACK = 2   /* the host language's own solver: fn(ctx, n, r, x) on HOST vectors, x pre-zeroed; returns 0 on success */
};
typedef int (*gmg_coarse_solve_fn)(void *ctx, int64_t n, const double *r, double *x);
/* maxiter/atol/rtol: CG_JACOBI only.  fn/ctx: HOST_CALLBACK only (must stay valid while the handle is used; it is called
 * from the thread that calls the solve entry point, once per cycle visit of the coarsest level). */
GMG_API int gmg_set_coarse_solver(gmg_handle_t h, int kind, int maxiter, double atol, double rtol,
                                  gmg_coarse_solve_fn fn, void *ctx);
/* ConvergenceLog summary of the last iterative coarse solve (GMG_COARSE_CG_JACOBI). */
GMG_API int gmg_get_coarse_log(gmg_handle_t h, gmg_result *res);

/* ---- solver options --------------------------------------------------------- */
/* kwargs of GMGLinearSolver: mode, cycle_type, maxiter, atol, rtol (GMGLinearSolvers.jl:56-58). */
GMG_API int gmg_set_options(gmg_handle_t h, int mode, int cycle, int maxiter, double atol, double rtol);

/* Layout / schedule policy of ONE handle.  The reference configures a solver by constructor keywords only
 * (GMGLinearSolvers.jl:48-58); every switch this library used to read from the environment is therefore a per-handle option:
 *     gmg_set_option(h, "pat_tile", 2)        ("PAT_TILE", "GMG_PAT_TILE" name the same option; unknown keys -> GMG_ERR_INVALID)
 * Options take effect at the next gmg_setup (setting one after a setup invalidates it, like a new operator would), except the
 * "live" ones marked (*) which act at the next call.  The environment variable GMG_<KEY>, when set, still OVERRIDES the handle's
 * value -- a debugging / A-B device, not the configuration interface.  Keys (default):
 *   storage layout   pattern (1) pat_shared (1) opattern (1) sell (1) sell_maxpad (1.25) vdict (1) idx16 (1) force_ptr64 (0)
 *                    pat_coded_min_rows (500000) eager* (1) eager_min_rows* (20000) refresh* (1)
 *   measurement      prof_stride (7: HIP events on every n-th sweep launch of the profiled level)
 *   sweep kernels    pat_rsweep (1) pat_r2 (1: two rows per lane in the r-gather sweep) pat_r2_occ (2: its 64-register form, eight waves per SIMD, in workgroups of eight waves on big levels; 1: four waves; 0: off) pat_r2_wgs (0 = one round of workgroups) pat_tile (0: never -- levels >= pat_tile_rows (3500000) take the pair sweep with one slice per wave; 1: tile sweep on those levels; 2: wherever it applies) pat_tile_min (512)
 *                    pat_tile_t (48) pat_tile_lds (79872) pat_strict (1) pat_fma (0: products and sums rounded separately, as the
 *                    reference's mul!; 1: fused multiply-add taps in the row-pattern sweeps -- not bit-identical)
 *                    pat_defer (1) pat_dinv (1) pat_emit (1) pat_nb (0 = auto) pat_rb (3) pat_un (9) pat_wgs (2048) pat_batched (1)
 *                    pat_small_wpb (4) pat_small_wpb2 (2) pat_wide (1) pat_wide_lds (73728) pat_wide_rounds (1) one_gather (1)
 *                    sell_un (6) sell_block (0 = auto) sell_defer (1) nt (1) nt_rowwise (1) big_rows (4000000) xcd_remap (1)
 *                    xcd_remap_big (-1: chunks two gather reaches deep per XCD; 0 launch order; 1 contiguous eighths; n chunk of n workgroups)
 *                    lanes_log2 (-1 = auto)
 *                    pat_r2mv (1: y = A x, y -= A x, y = b - A x of row-pattern levels with two rows per lane) pat_r2mv_min (100000:
 *                    smallest level, in rows, that takes it and the pair prolongation) pat_r2mv_dot (1: dot(p, A p) of CG formed by
 *                    the mat-vec kernel) pat_pair_p (1: prolongation + correction with two rows per lane)
 *                    pat_zwalk (1: levels of >= pat_zwalk_rows (9000000: where the vectors of a sweep outgrow the 256 MB Infinity Cache) rows sweep as a walk up the grid planes -- an interval of a plane
 *                    per wave, three new windows per step, pat_zwalk_t (12) planes per chain; 2: every level; 0: off) pat_zwalk_mv (1: their
 *                    mat-vecs too) pat_zwalk_wide (1: the wide-row (Q2) operator applications of levels of >= pat_zwalk_wide_rows (1000000) rows in the same form, 25 windows in registers)
 *                    pat_fuse2 (0: opt-in.  1: on grid levels of >= pat_fuse2_rows (1000000) rows of one GPU a smoothing pass runs TWO sweeps per launch -- r_k in,
 *                    r_{k+2} out, r_{k+1} in LDS only, sells_z2sweep_kernel; 2: every level that qualifies.  Bit-identical to the single sweeps; measured slower
 *                    than them on MI355X, see profiles/r06_ab_fuse2.txt) pat_fuse2_w / pat_fuse2_t (0 = chosen from the size: grid lines per workgroup, planes
 *                    per block) pat_fuse2_box (1: constant-coefficient boxes read one coefficient set per wave from the kernel arguments)
 *                    pat_box (0: opt-in, measured slower -- profiles/r06_ab_box.txt.  1: levels of pat_box_min_rows (1000000) .. pat_box_max_rows (9000000) rows whose operator is a constant-coefficient stencil on a box --
 *                    verified row by row at setup -- sweep with one coefficient set per wave read from the kernel arguments: no pattern ids, no LDS,
 *                    sells_boxsweep_kernel; 2: every level that qualifies; 0: off) pat_box_t (0: planes per chain from the level's size)
 *   reductions       red_fused (1: inside CG the second stage of every dot is done by the kernel that consumes the scalar and the
 *                    norm is reduced + posted to the host by one launch; 0: one reduce launch per dot.  Same bits either way)
 *   one-launch pass  persist (1) persist_fenced (0) persist_max_slices (0 = one workgroup per CU) persist_shared (0)
 *                    persist_wpb (1: smallest workgroup, in waves)
 *   coarsest level   coarse_host_max (1500) coarse_host_fallback_max (6000) coarse_auto_cg_min (20000: a dense-inverse request on a
 *                    coarsest level of at least this many dofs is served by the device CG-Jacobi solver instead) gj_mfma (1) gj_wide_min (4096)
 *   patch smoother   patch_dedup (1) patch_source_dedup (1) patch_operator (1)
 *   distributed      overlap (1) halo_fuse_pack (1) host_async (0)
 *   host round trip  host_poll* (1: the residual norm of every Krylov iteration is posted into page-locked host memory and polled; 0: copy +
 *                    hipStreamSynchronize)
 *   host vectors     x0_zero* (0: the solve entry points read x as the initial guess, CGSolvers.jl:79; 1: x is taken as zero on
 *                    entry and never uploaded) host_chunk_bytes* (4194304: staging chunk of unregistered host vectors)
 *   diagnostics      prof_stride (8) setup_timing* (0) dbg_nogather (0) host_timeline* (0: 1 = host-side duration of every step of
 *                    the cycles on stderr when the handle is destroyed) persist_force_timeout* (0: test hook, the next n solves
 *                    behave as if a one-launch pass had timed out) */
GMG_API int gmg_set_option(gmg_handle_t h, const char *key, double value);
/* Effective value of an option (NaN: the built-in default applies); *source (may be NULL) = 0 default, 1 gmg_set_option, 2 environment. */
GMG_API int gmg_get_option(gmg_handle_t h, const char *key, double *value, int *source);

/* Host-memory callers (GMG_MEM_HOST: what the Julia binding passes for Vector{Float64}).  Registering the caller's arrays ONCE
 * page-locks and maps them, so every later solve moves b / x by DMA straight from / into the caller's pages at the PCIe link rate
 * instead of through the staging path for pageable memory -- the pattern of ext/GridapPETScExt/PETScCaches.jl:23-36, which pins
 * the exact x / b objects of a solve.  The range must stay allocated until gmg_host_unregister (or gmg_destroy, which unregisters
 * what is left; it never frees caller memory).  Vectors inside a registered range are recognised by address. */
GMG_API int gmg_host_register(gmg_handle_t h, const void *ptr, int64_t nbytes);
GMG_API int gmg_host_unregister(gmg_handle_t h, const void *ptr);
/* Bytes moved host -> device / device -> host for host-memory callers since gmg_create, and the number of registered ranges. */
GMG_API int gmg_get_host_io_stats(gmg_handle_t h, int64_t *bytes_up, int64_t *bytes_down, int64_t *nregistered);
/* One-launch smoothing passes: solves that were re-run per sweep after a pass timed out, and whether the passes are still on. */
GMG_API int gmg_get_persist_retries(gmg_handle_t h, int64_t *retries, int *persist_active);

/* kwarg `verbose` of GMGLinearSolver (GMGLinearSolvers.jl:58).  The library never prints; verbose > 0 makes the GMG's own
 * ConvergenceLog complete on every path: as a preconditioner with maxiter = 1 inside gmg_cg_solve / gmg_fgmres_solve the
 * post-cycle norm(rh) of GMGLinearSolvers.jl:639 only feeds that log (update! returns true at maxiter regardless), so
 * with verbose = 0 (default) it is not evaluated and residuals[2] of the GMG log reads NaN. */
/* The HIP stream (hipStream_t, passed as void*) the handle issues ALL its work on -- kernels, copies, the RCCL collectives.  By
 * default every handle owns a non-blocking stream (gmg_create).  A caller that produces b and consumes x with its own kernels
 * (a device-resident Krylov loop around gmg_apply; PyTorch / AMDGPU.jl arrays) passes ITS stream: the library's work is then
 * ordered with the caller's without any device synchronisation on either side.  stream = NULL returns to the handle's own
 * stream -- it does NOT name HIP's null stream.  The device's default ("null", legacy-synchronising) stream, whose hipStream_t
 * is 0 -- what `torch.cuda.current_stream().cuda_stream` and AMDGPU.jl's default stream report -- is named by
 * GMG_STREAM_LEGACY (= hipStreamLegacy), the per-thread default stream by GMG_STREAM_PER_THREAD (= hipStreamPerThread).
 * The call waits for the work already issued on the stream it leaves.  No reference counterpart (the reference runs on
 * the host); `gmg_get_stream` returns the current one (e.g. to record an event on it). */
#define GMG_STREAM_LEGACY ((void *)1)
#define GMG_STREAM_PER_THREAD ((void *)2)
GMG_API int gmg_set_stream(gmg_handle_t h, void *stream);
GMG_API int gmg_get_stream(gmg_handle_t h, void **stream);
GMG_API int gmg_set_verbose(gmg_handle_t h, int verbose);
/* ns.solver.log of the GMG after the last solve!/ldiv! (also when it ran as a preconditioner inside a Krylov call). */
GMG_API int gmg_get_log(gmg_handle_t h, gmg_result *res, double *hist, int hist_cap);

/* numerical_setup(ss,A): GMGLinearSolvers.jl:183-210 -- uploads operators, builds
 * D^-1, R = P^T, patch factors, work vectors and the coarse solver
 * (default coarsest_solver = LUSolver(), :54,423-434 -> dense inverse applied as one GEMV per cycle;
 * factorised on the host up to 1500 dofs, inverted on the device above; see gmg_set_coarse_solver). */
GMG_API int gmg_setup(gmg_handle_t h);

/* ---- hot path ---------------------------------------------------------------- */
/* solve!(x,ns::GMGNumericalSetup,b) / ldiv!: GMGLinearSolvers.jl:612-649.
 * hist (may be NULL) receives niters+1 residual norms; hist_cap = its capacity. */
GMG_API int gmg_apply(gmg_handle_t h, const double *b, double *x, int memspace,
                      gmg_result *res, double *hist, int hist_cap);
/* solve!(x,ns::CGNumericalSetup,b) with Pl = this GMG: Krylov/CGSolvers.jl:73-120.
 * use_precond: 0 = Pl nothing, 1 = this GMG, 2 = JacobiLinearSolver() on the finest
 * matrix (the reference's CGSolver(JacobiLinearSolver())), 3 = LinearSolverFromSmoother(finest
 * pre-smoother) (LinearSolverFromSmoothers.jl:44-50, test/LinearSolvers/SmoothersTests.jl).
 * x = initial guess on entry. */
GMG_API int gmg_cg_solve(gmg_handle_t h, const double *b, double *x, int memspace,
                         int maxiter, double atol, double rtol, int flexible, int use_precond,
                         gmg_result *res, double *hist, int hist_cap);
/* solve!(x,ns::FGMRESNumericalSetup,b) with Pr = this GMG, Pl = nothing:
 * Krylov/FGMRESSolvers.jl:130-199, KrylovUtils.jl:17-54. */
GMG_API int gmg_fgmres_solve(gmg_handle_t h, const double *b, double *x, int memspace,
                             int m, int restart, int m_add, int maxiter, double atol, double rtol,
                             int use_precond, gmg_result *res, double *hist, int hist_cap);
/* The same with a LEFT preconditioner as well (FGMRESSolver(m,Pr;Pl=...), FGMRESSolvers.jl:26-30; krylov_mul! / krylov_residual!
 * with Pl, KrylovUtils.jl:14-18,46-50): use_precond_left takes the values of use_precond; the handle's GMG can serve on one
 * side only. */
GMG_API int gmg_fgmres_solve_pl(gmg_handle_t h, const double *b, double *x, int memspace,
                                int m, int restart, int m_add, int maxiter, double atol, double rtol,
                                int use_precond, int use_precond_left, gmg_result *res, double *hist, int hist_cap);
/* solve!(x,ns::RichardsonLinearNumericalSetup,b): RichardsonLinearSolvers.jl:79-106, scalar omega;
 * use_precond as in gmg_cg_solve (Pl = nothing / this GMG / Jacobi / the finest pre-smoother). */
GMG_API int gmg_richardson_solve(gmg_handle_t h, const double *b, double *x, int memspace, double omega,
                                 int maxiter, double atol, double rtol, int use_precond,
                                 gmg_result *res, double *hist, int hist_cap);

/* ---- operator-level entry points (duck-typed `mul!` / smoother `solve!`) ------ */
/* mul!(y,op,x) for op in {A_lev, P_lev, R_lev}: RichardsonSmoothers.jl:94,
 * GMGLinearSolvers.jl:484,491,495. */
GMG_API int gmg_op_apply(gmg_handle_t h, int lev, int op, const double *x, double *y, int memspace);
/* solve!(x,ns::RichardsonSmootherNumericalSetup,r): updates x AND r in place
 * (RichardsonSmoothers.jl:84-98). which = GMG_PRE or GMG_POST. */
GMG_API int gmg_smooth(gmg_handle_t h, int lev, int which, double *x, double *r, int memspace);
/* solve!(dx,Mns,r) of the smoother's inner solver (Jacobi / patch), no relaxation:
 * JacobiLinearSolvers.jl:43-47, PatchSolvers.jl:238-241, BlockJacobiSolvers.jl:119-123. */
GMG_API int gmg_precond_apply(gmg_handle_t h, int lev, int which, const double *r, double *dx, int memspace);
/* coarsest solve!(xh,coarsest_solver_cache,rh): GMGLinearSolvers.jl:474. */
GMG_API int gmg_coarse_solve(gmg_handle_t h, const double *r, double *x, int memspace);
/* dot / norm as used by the Krylov solvers (CGSolvers.jl:85,95,105). */
GMG_API int gmg_dot(gmg_handle_t h, int64_t n, const double *a, const double *b, int memspace, double *out);

/* ---- multi-GPU: one handle per rank/GPU ---------------------------------
 * The reference row-partitions every level with PartitionedArrays (own-then-ghost local
 * numbering, JacobiLinearSolvers.jl:29-56) and communicates through `consistent!`
 * (owner->ghost, PatchSolvers.jl:231,256 and inside mul!(::PVector,::PSparseMatrix,::PVector))
 * and the reductions inside dot/norm.  Call order: gmg_create, gmg_comm_init_*,
 * gmg_set_partition + gmg_set_matrix/prolongation/restriction with LOCAL operators
 * (rows = owned dofs, columns = [own | ghost]), gmg_set_replication, gmg_setup.
 * Vectors passed to the solve calls hold the OWNED entries only. */
typedef void (*gmg_host_exchange_fn)(void *ctx, int nnbr, const int32_t *nbr_rank, const double *sendbuf,
                                     const int64_t *snd_ptr, double *recvbuf, const int64_t *rcv_ptr);
typedef void (*gmg_host_allreduce_fn)(void *ctx, double *vals, int n);
/* ncclGetUniqueId through the RCCL library at rccl_path (NULL: librccl.so.1); 128 bytes out.
 * Rank 0 calls it and the host language broadcasts the blob (MPI.bcast / torch.distributed). */
GMG_API int gmg_comm_unique_id(const char *rccl_path, char *id_out128);
/* RCCL over xGMI: grouped ncclSend/ncclRecv halos + ncclAllReduce on the handle's stream. */
GMG_API int gmg_comm_init_rccl(gmg_handle_t h, const char *rccl_path, const char *unique_id128, int rank, int nranks);
/* Diagnostic: all-reduce of 1.5 (-> out2[0] = 1.5*nranks) and a grouped self send/recv of
 * 42.0 (-> out2[1]) through the RCCL binding; valid on a 1-rank communicator. */
GMG_API int gmg_comm_selftest(gmg_handle_t h, double *out2);
/* Latency of the communication primitives exactly as the solver issues them (no reference counterpart: PartitionedArrays'
 * consistent! / dot over MPI, PatchSolvers.jl:231, CGSolvers.jl:85): out6[0] stream time in us of one halo exchange (event
 * hand-off to the communication stream, grouped ncclSend/ncclRecv of nmsg messages of `count` doubles, event hand-off back),
 * out6[1] host time to enqueue it, out6[2] / out6[3] the same for a 1-double ncclAllReduce + square root, out6[4] the
 * exchange issued in-stream, out6[5] an empty kernel.  Peers are the ring neighbours (the rank itself on one rank). */
GMG_API int gmg_comm_latency_probe(gmg_handle_t h, int nmsg, int64_t count, int reps, double *out6);
/* Loopback (testing / bring-up on one GPU; no reference counterpart -- the reference's analogue is running its MPI tests with one
 * process, test/LinearSolvers/mpi/GMGTests.jl:5-8): after gmg_comm_init_rccl(.., rank 0, nranks 1) or gmg_comm_init_host(.., 0, 1, ..)
 * declare that the partition handed over next was written for `virtual_nranks` ranks and FOLDED onto this one: every rank's owned rows
 * stacked, every rank's ghosts stacked behind them, every neighbour entry naming a rank other than 0 -- which then means "this rank".
 * Message k of a level's plan is sent to and received from this rank itself (RCCL matches the k-th send of a group with the k-th
 * receive), so pack -> ncclGroupStart/Send/Recv/GroupEnd on the communication stream -> event -> boundary fix-up -> ncclAllReduce
 * run exactly as on `virtual_nranks` GPUs, on the one that exists.  partition.fold_ranks() builds such a partition. */
GMG_API int gmg_comm_set_loopback(gmg_handle_t h, int virtual_nranks);
/* Host-staged transport through callbacks of the host language (MPI in Julia, gloo in the
 * Python tests); lets several ranks share one GPU.  Functional, not fast. */
GMG_API int gmg_comm_init_host(gmg_handle_t h, int rank, int nranks, gmg_host_exchange_fn exchange,
                               gmg_host_allreduce_fn allreduce, void *ctx);
/* Exchange plan of level lev: neighbours, local ids of owned entries to send (0-based),
 * and the ghost sub-ranges received from each neighbour (ghosts ordered by owner). */
GMG_API int gmg_set_partition(gmg_handle_t h, int lev, int64_t n_own, int64_t n_ghost, int nnbr,
                              const int32_t *nbr_rank, const int64_t *snd_ptr, const int64_t *snd_idx,
                              const int64_t *rcv_ptr);
/* The same for a level in the OVERLAPPING layout: owned and ghost entries share ONE local numbering of n_local entries chosen by the
 * caller (a structured box partition: the box extended by `depth` node layers, lexicographic), the local matrix is square over all of
 * them (rows of ghost entries are the true global rows restricted to the local columns -- exact except on the outermost layer), and
 * one exchange makes `depth` layers consistent.  A smoothing pass of Richardson(Jacobi) then communicates once per `depth`
 * sweeps instead of once per sweep: ghost layer j is recomputed redundantly and stays exact for depth - j sweeps, owned rows are
 * exact throughout and are summed in the order of a single-GPU run.  Transfers: P_lev has a row per local entry of level lev, R_lev's
 * rows of non-owned coarse entries are empty.  Levels >= 1; the finest level too when the Krylov operator is given separately (below).
 * snd_idx / rcv_idx: local ids (0-based) sent to / received from each neighbour, both sides enumerating in the same (global) order.
 * Reference analogue: consistent!(::PVector) once per mul! (RichardsonSmoothers.jl:94 through PartitionedArrays) -- here once per
 * `depth` applications.
 * Patch smoothers (round 4): gmg_set_smoother_patch with EVERY patch whose dofs all lie in the local box (local numbering), blocks
 * A[p,p] from the local matrix -- exact there --, patch_cols = patch_rows; `depth` then counts sweeps of Richardson(PatchSolver),
 * each of which consumes 3 order - 2 node layers on vertex stars (partition._OverlapGeom).  No assemble! and no consistent!(dx)
 * inside a block: ceil(niter / depth) exchanges per pass instead of 2 niter + niter reverse exchanges (PatchSolvers.jl:227-258). */
GMG_API int gmg_set_partition_overlap(gmg_handle_t h, int lev, int64_t n_local, int64_t n_ghost, int depth, int nnbr,
                                      const int32_t *nbr_rank, const int64_t *snd_ptr, const int64_t *snd_idx,
                                      const int64_t *rcv_ptr, const int64_t *rcv_idx);
/* Two of the exchanges a V-cycle makes on an overlapping level besides the smoothing blocks can be dropped when the halo geometry allows.
 * The caller states the GEOMETRY (it knows the halo in node layers; the library only knows `depth` in sweeps), the library decides:
 *   exact_node_layers, layers_per_sweep, restriction_reach -- the halo holds `exact_node_layers` node layers around the owned box, one
 *     sweep of this level's smoother makes `layers_per_sweep` more of them inexact (Richardson-Jacobi: `order`; vertex-star patches:
 *     3 order - 2), and the restriction of an owned coarse row reads fine nodes up to `restriction_reach` away (1 for Q1, 3 for Q2).
 *     After a smoothing pass of niter sweeps in blocks of `depth` the last block has niter - depth * ((niter - 1) / depth) sweeps;
 *     when exact_node_layers - that * layers_per_sweep >= restriction_reach, consistent!(r) before `rH = R rh`
 *     (GMGLinearSolvers.jl:484) is skipped -- evaluated per pass with the niter of the smoother that just ran (pre-smoother; the
 *     post-smoother on the second leg of a W / F cycle), so pre != post niter and later gmg_set_smoother_* calls cannot make it stale;
 *   correction_exact_near_owned -- P's local rows are complete (the coarse level is replicated, or overlapping with >= 1 layer) on every
 *     fine entry that `rh_own -= (A dxh)_own` reads: consistent!(dxh) before `rh -= A dxh` (:495) is skipped -- the ghost rows of r it
 *     leaves inexact are refreshed by the consistent!(r) that opens the post-smoothing pass.
 * Call AFTER gmg_set_partition_overlap and gmg_set_smoother_* of the level: both clear the hints (they were stated for the halo / the
 * smoother that was there before); zeros = no shortcut.  partition.overlap_hints() is the structured-grid rule; results on owned rows
 * are bit-identical with and without the hints. */
GMG_API int gmg_set_partition_overlap_hints(gmg_handle_t h, int lev, int exact_node_layers, int layers_per_sweep, int restriction_reach,
                                            int correction_exact_near_owned);
/* FINEST level in the overlapping layout (round 5).  The Krylov solver's vectors are the caller's -- own | ghost numbering, its dot
 * products run over owned entries -- so the solver then holds TWO finest operators: the preconditioner's level 0 in the overlapping
 * layout (gmg_set_partition_overlap(h, 0, ...), gmg_set_matrix(h, 0, square local matrix), P_0 / R_0 in that numbering) and the
 * Krylov operator in the own | ghost layout, passed through the same entry points with lev = GMG_LEVEL_KRYLOV:
 *   gmg_set_partition(h, GMG_LEVEL_KRYLOV, n_own, n_ghost, ...) ; gmg_set_matrix(h, GMG_LEVEL_KRYLOV, n_own x (n_own + n_ghost), ...)
 *   gmg_set_krylov_map(h, own_to_local, n_own)   -- local id, in level 0's overlapping numbering, of every owned entry.
 * gmg_cg_solve / gmg_fgmres_solve / gmg_apply take vectors of n_own entries; applying the preconditioner scatters r into level 0's
 * numbering (the first smoothing block's exchange fills the ghost layers), runs the cycle, and gathers the owned entries of the
 * correction: the finest smoothing passes then communicate once per `depth` sweeps like every other overlapping level, at the price
 * of (n_local - n_own) redundant rows and 32 bytes per owned row for the two index passes.  Worth it when the finest sweep is
 * shorter than a halo exchange (multigpu.plan_partition decides; at BASELINE config 4's 288^3 cells per GPU it is not).
 * The GMG runs as a preconditioner with maxiter = 1 in this form (its own residual norms would count ghost entries).
 * Reference analogue: none -- PartitionedArrays exchanges once per mul! (RichardsonSmoothers.jl:94). */
#define GMG_LEVEL_KRYLOV (-1)
GMG_API int gmg_set_krylov_map(gmg_handle_t h, const int64_t *own_to_local, int64_t n_own);
/* Halo exchanges and all-reduces this handle has issued since gmg_create (either may be NULL). */
GMG_API int gmg_get_comm_stats(gmg_handle_t h, int64_t *n_exchanges, int64_t *n_allreduces);
/* What the handle communicates through: *transport = 0 none / 1 RCCL / 2 host callbacks, its rank and rank count, what RCCL itself
 * reports for the communicator (ncclCommCount; -1 when not RCCL) and the HIP device it sits on (ncclCommCuDevice / the handle's
 * device).  Any pointer may be NULL.  bench.py prints these so that a multi-GPU line shows how many ranks RCCL really saw. */
GMG_API int gmg_get_comm_info(gmg_handle_t h, int *transport, int *rank, int *nranks, int *comm_count, int *comm_device);
/* Levels >= lev are REPLICATED: every rank passes the GLOBAL operators of those levels
 * (gmg_set_matrix / _prolongation / _restriction, no gmg_set_partition) and computes them
 * redundantly -- no halo traffic where the level is tiny.  Across the boundary, P_{lev-1} has
 * this rank's fine rows and GLOBAL coarse columns; R_{lev-1} yields the rows listed in
 * own_global_ids (global numbering of level lev), which are summed into the replicated
 * residual with one all-reduce.  At least the coarsest level must be replicated.
 * Reference analogue: coarse levels living on fewer ranks (np_per_level, ModelHierarchies.jl:80-148)
 * with redistribute! at the boundary (GridTransferOperators.jl:447-532). */
GMG_API int gmg_set_replication(gmg_handle_t h, int lev, const int64_t *own_global_ids, int64_t n_own);
/* Levels lev .. (first replicated level - 1) live on a RANK SUBSET (round 5; the reference's np_per_level with redistribute! at the
 * boundary, ModelHierarchies.jl:80-148, GridTransferOperators.jl:447-532 -- redistribute_cell_dofs! there, two p2p plans here).
 * Level lev then exists in two partitions: the GLUED one of all ranks (its dofs with the owners of the fine dofs they coincide with;
 * own | ghost numbering of n_glue_own + n_glue_ghost entries: P_{lev-1} has those columns, R_{lev-1} yields the n_glue_own owned rows)
 * and the subset's (gmg_set_partition / _overlap + operators on the member ranks only; the other ranks pass nothing for the levels
 * lev .. rep-1 and member = 0).  to_sub carries the restricted residual from the glued owners to the subset owners; from_sub carries
 * the correction from the subset owners to the glued own AND ghost entries (so P needs no consistent! of its own).  A plan lists
 * per neighbour the local ids sent (in the source space) and the local ids received (in the destination space), both sides in the
 * same order, plus the entries that stay on the rank.  Ranks outside the subset take part in the collectives below (the all-reduce
 * that assembles the replicated residual) with a zero contribution.  Every rank calls this, with its own plans, before gmg_setup. */
typedef struct {
  int nnbr;
  const int32_t *nbr_rank;
  const int64_t *snd_ptr, *snd_idx, *rcv_ptr, *rcv_idx;
  int64_t nself;
  const int64_t *self_src, *self_dst;
} gmg_redist_plan;
GMG_API int gmg_set_redistribution(gmg_handle_t h, int lev, int member, int64_t n_glue_own, int64_t n_glue_ghost,
                                   const gmg_redist_plan *to_sub, const gmg_redist_plan *from_sub);

/* ---- measurement --------------------------------------------------------------- */
/* Bracket launches of the fused Richardson-Jacobi sweep on `lev` with HIP events on the handle's
 * stream: every GMG_PROF_STRIDE-th launch (default 7 -- odd on purpose: sweeps alternate between the variant that leaves x
 * alone and the one that adds two increments to it, `x = (x + s_{k-1}) + s_k`, and an even stride would time one of them only;
 * a sampled launch costs ~11 us of stream time -- two ~5.7 us bubbles, one per event (profiles/r05_tuning.md section 7: at stride 7 that
 * was 5 % of a 128^3 solve) -- so a caller that times whole solves at the same time sets the option prof_stride (read at this call;
 * odd and coprime with the number of sweeps per solve, e.g. 61) to keep the samples rare); enable=0 stops.  Read with gmg_get_kernel_stats (launch-weighted totals) and
 * gmg_get_kernel_stats_by_variant (index 0: sweeps that update x every time -- the generic layouts' first sweep, one-launch
 * passes, patch sweeps; 1: x untouched; 2: x updated with two increments; layout_bytes[v] = the bytes ONE launch of form v moves with the
 * stored layout, every operand once -- gmg_kernel_stats.layout_bytes is their launch-weighted mean). */
GMG_API int gmg_profile_enable(gmg_handle_t h, int lev, int enable);
GMG_API int gmg_get_kernel_stats(gmg_handle_t h, gmg_kernel_stats *out);
GMG_API int gmg_get_kernel_stats_by_variant(gmg_handle_t h, double total_ms[3], int64_t launches[3], double layout_bytes[3]);
/* Algorithmic bytes (SURVEY 8d byte model) of one V-cycle / one CG iteration. */
GMG_API int gmg_model_bytes(gmg_handle_t h, double *vcycle_bytes, double *cg_iter_bytes);
/* Storage chosen for A_lev at setup: *sell = 0 CSR-stream, 1 SELL-64 / SELL-C, 2 SELL-P (row-pattern
 * dictionary), 3 SELL-O (SELL-64 value stream, column offsets from an offset-pattern table); 8-bit value dictionary, 16-bit column offsets, bytes of matrix stream per stored nonzero,
 * padding factor. */
GMG_API int gmg_level_format(gmg_handle_t h, int lev, int *sell, int *vdict, int *idx16,
                             double *stream_bytes_per_nnz, double *padding);
/* Kernel + template arguments + launch geometry of the fused sweep last launched on level lev ("" before the first sweep):
 * committed counter measurements (profiles/traffic_latest.json) are only attached to a bench line whose sweep has this signature. */
GMG_API int gmg_sweep_signature(gmg_handle_t h, int lev, char *buf, int cap);
/* Measured streaming ceiling: a 16 B/lane copy kernel over nbytes (read) + nbytes (write), reps launches timed with HIP
 * events on the handle's stream; *gbytes_per_s = moved bytes / time.  Reported by bench.py beside the 8 TB/s spec. */
GMG_API int gmg_stream_probe(gmg_handle_t h, int64_t nbytes, int reps, double *gbytes_per_s);
/* The same for a READ-ONLY stream of nbytes (the sweeps read ~9x more than they write, so their ceiling lies between the two). */
GMG_API int gmg_stream_probe_read(gmg_handle_t h, int64_t nbytes, int reps, double *gbytes_per_s);
/* Device memory held by the handle, bytes. */
GMG_API int gmg_device_bytes(gmg_handle_t h, int64_t *bytes);

/* ---- block preconditioners --------------------------------------------------
 * The glue that calls the GMG hot path once per outer FGMRES iteration in the reference's block
 * applications (test/Applications/StokesGMG.jl:142-153): BlockDiagonalSolver
 * (BlockSolvers/BlockDiagonalSolvers.jl:165-177) and BlockTriangularSolver
 * (BlockSolvers/BlockTriangularSolvers.jl:186-242) with the outer CG / FGMRES on the block system,
 * all on the device.  Block vectors are contiguous: block i occupies [off_i, off_i + size_i).
 * A gmg handle given to gmg_block_set_diag_gmg stays owned by the caller, must be set up
 * first, must outlive the block handle's use of it and issues its work on the block handle's stream
 * from gmg_block_setup until gmg_block_destroy. */
typedef struct gmg_block_solver *gmg_block_handle_t;
enum gmg_block_kind { GMG_BLOCK_DIAGONAL = 0, GMG_BLOCK_LOWER = 1, GMG_BLOCK_UPPER = 2 };
enum gmg_block_diag_kind {
  GMG_BLOCK_GMG = 1,        /* a GMGLinearSolver numerical setup (gmg handle) */
  GMG_BLOCK_CG_JACOBI = 2,  /* CGSolver(JacobiLinearSolver();maxiter,atol,rtol), CGSolvers.jl:73-120 */
  GMG_BLOCK_LU = 3,         /* LUSolver() on a small block: dense inverse on the device */
  GMG_BLOCK_JACOBI = 4      /* JacobiLinearSolver(), JacobiLinearSolvers.jl:43-47 */
};
/* BlockDiagonalSolver(solvers) / BlockTriangularSolver(blocks,solvers,coeffs,half): BlockTriangularSolvers.jl:55-85 */
GMG_API int gmg_block_create(gmg_block_handle_t *h, int nblocks, const int64_t *block_sizes, int kind, int device_id);
GMG_API int gmg_block_destroy(gmg_block_handle_t h);
GMG_API const char *gmg_block_last_error(gmg_block_handle_t h);
/* Distributed block systems (one handle per rank, like the GMG handles): the communicator of the block solver, and per block
 * the exchange plan of its vector space (same meaning as gmg_set_partition; blocks without ghost columns need none).  Call
 * before the blocks are set: block (i,j) then has bsize(i) rows and bsize(j) + n_ghost(j) columns ([own | ghost] numbering
 * of block j), block vectors hold the owned entries of every block, a GMG handle given to gmg_block_set_diag_gmg is itself
 * distributed over the same ranks.  Reference: BlockTriangularSolvers.jl:216-242 on BlockPVector / BlockPMatrix
 * (test/Applications/mpi/StokesGMG.jl).  LUSolver() diagonal blocks stay single-GPU. */
GMG_API int gmg_block_comm_init_rccl(gmg_block_handle_t h, const char *rccl_path, const char *unique_id128, int rank, int nranks);
GMG_API int gmg_block_comm_init_host(gmg_block_handle_t h, int rank, int nranks, gmg_host_exchange_fn exchange,
                                     gmg_host_allreduce_fn allreduce, void *ctx);
/* gmg_comm_set_loopback for a block handle: its own communicator of one rank serves a folded partition (see there). */
GMG_API int gmg_block_comm_set_loopback(gmg_block_handle_t h, int virtual_nranks);
GMG_API int gmg_block_set_partition(gmg_block_handle_t h, int j, int64_t n_own, int64_t n_ghost, int nnbr,
                                    const int32_t *nbr_rank, const int64_t *snd_ptr, const int64_t *snd_idx,
                                    const int64_t *rcv_ptr);
/* blocks(mat)[i,j] of the system matrix (numerical_setup(ss,mat::AbstractBlockMatrix), BlockTriangularSolvers.jl:132-152);
 * absent blocks are zero. */
GMG_API int gmg_block_set_system_block(gmg_block_handle_t h, int i, int j, int64_t nrows, int64_t ncols, int64_t nnz,
                                       const void *ptr, const void *idx, const double *val,
                                       int layout, int index_base, int index_bytes);
/* Off-diagonal block of the PRECONDITIONER when it is not the system's (MatrixBlock / BiformBlock,
 * BlockSolverInterfaces.jl); default = the system block (LinearSystemBlock). */
GMG_API int gmg_block_set_precond_block(gmg_block_handle_t h, int i, int j, int64_t nrows, int64_t ncols, int64_t nnz,
                                        const void *ptr, const void *idx, const double *val,
                                        int layout, int index_base, int index_bytes);
/* coeffs[i,j] (default 1.0; a zero coefficient drops the block, BlockTriangularSolvers.jl:194,223) */
GMG_API int gmg_block_set_coeff(gmg_block_handle_t h, int i, int j, double c);
/* solvers[i] */
GMG_API int gmg_block_set_diag_gmg(gmg_block_handle_t h, int i, gmg_handle_t g);
GMG_API int gmg_block_set_diag_solver(gmg_block_handle_t h, int i, int kind, int maxiter, double atol, double rtol);
/* matrix the solver of block i is set up on when it is not the system block (i,i) (e.g. the -1/alpha
 * pressure mass matrix, StokesGMG.jl:145) */
GMG_API int gmg_block_set_diag_matrix(gmg_block_handle_t h, int i, int64_t n, int64_t nnz, const void *ptr,
                                      const void *idx, const double *val, int layout, int index_base, int index_bytes);
/* numerical_setup */
GMG_API int gmg_block_setup(gmg_block_handle_t h);
/* solve!(x,ns::BlockDiagonalSolverNS|BlockTriangularSolverNS,b) */
GMG_API int gmg_block_precond_apply(gmg_block_handle_t h, const double *b, double *x, int memspace);
/* mul!(y,A,x) on the block system */
GMG_API int gmg_block_apply_system(gmg_block_handle_t h, const double *x, double *y, int memspace);
/* FGMRESSolver(m,P) / CGSolver(P) on the block system with P = this block preconditioner (use_precond=1)
 * or nothing (0): FGMRESSolvers.jl:130-199, CGSolvers.jl:73-120 */
GMG_API int gmg_block_fgmres_solve(gmg_block_handle_t h, const double *b, double *x, int memspace, int m, int restart,
                                   int m_add, int maxiter, double atol, double rtol, int use_precond,
                                   gmg_result *res, double *hist, int hist_cap);
GMG_API int gmg_block_cg_solve(gmg_block_handle_t h, const double *b, double *x, int memspace, int maxiter, double atol,
                               double rtol, int flexible, int use_precond, gmg_result *res, double *hist, int hist_cap);
/* ConvergenceLog of the last solve of diagonal block i (GMG / CG blocks) */
GMG_API int gmg_block_diag_log(gmg_block_handle_t h, int i, gmg_result *res);

#ifdef __cplusplus
}
#endif
#endif /* GMG_AMD_H */
