/*
 * gmg_oracle.c -- CPU restatement of the GridapSolvers.jl GMG hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke()
 * check in __graft_entry__.py and the cpu_baseline leg of bench.py may load
 * it.  The product (libgmgamd.so) never links, loads or calls anything here.
 *
 * Every function cites the reference file:line (relative to the
 * GridapSolvers.jl v0.7.1 source tree) whose operation order it follows.
 * The reference is pure Julia and its arithmetic primitives (mul!, dot, norm,
 * lu!, ldiv!) live in un-vendored third-party packages (SparseArrays,
 * SparseMatricesCSR, LinearAlgebra/BLAS/LAPACK, UMFPACK); those are restated
 * from their published algorithms:
 *   - mul!(y,A,x)      : y_i = sum_j a_ij x_j, CSR row gather, left-to-right
 *   - dot / norm       : plain left-to-right fp64 accumulation
 *   - lu!(A)           : LAPACK dgetf2 (partial pivoting, unblocked)
 *   - lu!(A,NoPivot()) : Doolittle elimination without pivoting
 *   - LUSolver()       : exact sparse direct solve; restated as banded LU with
 *                        partial pivoting (LAPACK dgbtf2 order)
 *   - givensAlgorithm  : Julia LinearAlgebra.givensAlgorithm (LAPACK dlartg)
 *
 * PARITY STATUS: "parity unpinned" for V-cycle vectors.  Julia is not
 * installed in the build container, the reference holds no golden vectors and
 * test/LinearSolvers/GMGTests.jl asserts nothing.  What IS pinned (see
 * tests/test_oracle.py): the reference's own known-answer criteria
 * test/LinearSolvers/SmoothersTests.jl:43 (E<1e-8), KrylovTests.jl:25
 * (E<1e-6), RichardsonLinearTests.jl:26 (E<1e-6), BlockDiagonalSolversTests.jl:45
 * (norm(x1-x)<1e-8), Applications/StokesGMG.jl:166 (residual<1e-7 criterion on a
 * synthetic saddle point) on the same problem definitions, plus operator identities.
 *
 * Index conventions: 0-based CSR, int64 row pointers, int32 column indices.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

typedef int64_t i64;
typedef int32_t i32;

#define ORC_API __attribute__((visibility("default")))

/* liboracle_omp.so (bench.py's cpu_baseline only) is this file compiled with -fopenmp -DORC_OMP: the row loops and the
 * reductions run on all host cores -- the analogue of the reference under MPI on P ranks.  The checker (liboracle.so)
 * is always the sequential build: its sums are left-to-right. */
#ifdef ORC_OMP
#include <omp.h>
#define ORC_PFOR _Pragma("omp parallel for schedule(static)")
#define ORC_PSUM(var) _Pragma("omp parallel for schedule(static) reduction(+:s)")
#else
#define ORC_PFOR
#define ORC_PSUM(var)
#endif

/* ------------------------------------------------------------------ */
/* L0 primitives (third-party in the reference)                        */
/* ------------------------------------------------------------------ */

/* mul!(y,A,x): RichardsonSmoothers.jl:94, GMGLinearSolvers.jl:495,623,
 * CGSolvers.jl:79,104, KrylovUtils.jl:19,24,27,31,47,52 */
ORC_API void orc_spmv(i64 n, const i64 *ptr, const i32 *idx, const double *val,
                      const double *x, double *y)
{
  ORC_PFOR
  for (i64 i = 0; i < n; ++i) {
    double s = 0.0;
    for (i64 k = ptr[i]; k < ptr[i + 1]; ++k) s += val[k] * x[idx[k]];
    y[i] = s;
  }
}

/* dot(a,b): CGSolvers.jl:95,105; FGMRESSolvers.jl:161 */
#ifndef ORC_BLAS_ORDER
ORC_API double orc_dot(i64 n, const double *a, const double *b)
{
  double s = 0.0;
  ORC_PSUM(s)
  for (i64 i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* norm(a): CGSolvers.jl:85,111; GMGLinearSolvers.jl:627,639; FGMRESSolvers.jl:141,164 */
ORC_API double orc_norm(i64 n, const double *a) { return sqrt(orc_dot(n, a, a)); }
#else
/* liboracle_blas.so (tests only): the summation orders Julia really uses on Vector{Float64} -- `dot` and `norm` go through
 * BLAS: ddot with several interleaved partial sums (SIMD lanes x unrolling; here 8 accumulators combined pairwise, the
 * shape of OpenBLAS' x86-64 kernels) and dnrm2 as a SCALED sum of squares (reference BLAS / LAPACK dlassq recurrence).
 * tests/test_oracle.py checks that iteration counts are identical and residual histories agree to 1e-8 between this
 * build and the left-to-right checker: the parity tolerances do not hinge on the summation order. */
ORC_API double orc_dot(i64 n, const double *a, const double *b)
{
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  i64 i = 0;
  for (; i + 8 <= n; i += 8)
    for (int k = 0; k < 8; ++k) acc[k] += a[i + k] * b[i + k];
  double tail = 0.0;
  for (; i < n; ++i) tail += a[i] * b[i];
  return (((acc[0] + acc[4]) + (acc[2] + acc[6])) + ((acc[1] + acc[5]) + (acc[3] + acc[7]))) + tail;
}
ORC_API double orc_norm(i64 n, const double *a)
{
  double scale = 0.0, ssq = 1.0;                     /* dnrm2 (reference BLAS): scale * sqrt(ssq) */
  for (i64 i = 0; i < n; ++i) {
    if (a[i] != 0.0) {
      const double ax = fabs(a[i]);
      if (scale < ax) { const double q = scale / ax; ssq = 1.0 + ssq * q * q; scale = ax; }
      else { const double q = ax / scale; ssq += q * q; }
    }
  }
  return scale * sqrt(ssq);
}
#endif

/* ------------------------------------------------------------------ */
/* A9: SolverTolerances.jl:117-128, ConvergenceLogs.jl:101-150         */
/* ------------------------------------------------------------------ */
typedef struct {
  int maxiter;
  double atol, rtol;
  int num_iters;
  double *residuals; /* length maxiter+1, ConvergenceLogs.jl:56 */
} orc_log;

enum { ORC_CONVERGED_ATOL = 0, ORC_CONVERGED_RTOL = 1, ORC_DIVERGED_MAXITER = 2, ORC_DIVERGED_BREAKDOWN = 3 };

/* finished(): SolverTolerances.jl:117-119 ; converged(): :126-128 (strict <) */
static int log_finished(const orc_log *l, int niter, double e_a, double e_r)
{
  return (niter >= l->maxiter) || (e_r < l->rtol) || (e_a < l->atol);
}
/* init!: ConvergenceLogs.jl:101-112 */
static int log_init(orc_log *l, double r0)
{
  l->num_iters = 0;
  for (int i = 0; i <= l->maxiter; ++i) l->residuals[i] = 0.0;
  l->residuals[0] = r0;
  return log_finished(l, l->num_iters, r0, 1.0);
}
/* update!: ConvergenceLogs.jl:119-129 */
static int log_update(orc_log *l, double r)
{
  l->num_iters += 1;
  l->residuals[l->num_iters] = r;
  double r_rel = r / l->residuals[0];
  return log_finished(l, l->num_iters, r, r_rel);
}
/* finalize!: ConvergenceLogs.jl:136-150 + finished_flag SolverTolerances.jl:97-110 */
static int log_finalize(const orc_log *l, double r)
{
  double r_rel = r / l->residuals[0];
  if (r_rel < l->rtol) return ORC_CONVERGED_RTOL;
  if (r < l->atol) return ORC_CONVERGED_ATOL;
  if (l->num_iters >= l->maxiter) return ORC_DIVERGED_MAXITER;
  return ORC_DIVERGED_BREAKDOWN;
}

/* ------------------------------------------------------------------ */
/* Dense LU kernels (third-party LAPACK in the reference)              */
/* ------------------------------------------------------------------ */

/* lu!(A): dgetf2, column-major n x n, partial pivoting. PatchSolvers.jl:176 */
static void dense_lu_pivot(int n, double *a, int *piv)
{
  for (int j = 0; j < n; ++j) {
    int p = j;
    double mx = fabs(a[j + (size_t)j * n]);
    for (int i = j + 1; i < n; ++i) {
      double v = fabs(a[i + (size_t)j * n]);
      if (v > mx) { mx = v; p = i; }
    }
    piv[j] = p;
    if (p != j)
      for (int k = 0; k < n; ++k) {
        double t = a[j + (size_t)k * n];
        a[j + (size_t)k * n] = a[p + (size_t)k * n];
        a[p + (size_t)k * n] = t;
      }
    double d = a[j + (size_t)j * n];
    if (d != 0.0) {
      double rd = 1.0 / d;
      for (int i = j + 1; i < n; ++i) a[i + (size_t)j * n] *= rd;
    }
    for (int k = j + 1; k < n; ++k) {
      double ajk = a[j + (size_t)k * n];
      for (int i = j + 1; i < n; ++i) a[i + (size_t)k * n] -= a[i + (size_t)j * n] * ajk;
    }
  }
}
/* lu!(A,NoPivot();check=false): BlockJacobiSolvers.jl:162 */
static void dense_lu_nopivot(int n, double *a)
{
  for (int j = 0; j < n; ++j) {
    double rd = 1.0 / a[j + (size_t)j * n];
    for (int i = j + 1; i < n; ++i) a[i + (size_t)j * n] *= rd;
    for (int k = j + 1; k < n; ++k) {
      double ajk = a[j + (size_t)k * n];
      for (int i = j + 1; i < n; ++i) a[i + (size_t)k * n] -= a[i + (size_t)j * n] * ajk;
    }
  }
}
/* ldiv!(LU,b): dgetrs = row swaps, unit-lower forward, upper backward */
static void dense_lu_solve(int n, const double *a, const int *piv, double *b)
{
  if (piv)
    for (int j = 0; j < n; ++j)
      if (piv[j] != j) { double t = b[j]; b[j] = b[piv[j]]; b[piv[j]] = t; }
  for (int j = 0; j < n; ++j) {
    double bj = b[j];
    for (int i = j + 1; i < n; ++i) b[i] -= a[i + (size_t)j * n] * bj;
  }
  for (int j = n - 1; j >= 0; --j) {
    b[j] /= a[j + (size_t)j * n];
    double bj = b[j];
    for (int i = 0; i < j; ++i) b[i] -= a[i + (size_t)j * n] * bj;
  }
}

/* ------------------------------------------------------------------ */
/* A6: coarsest solver, Gridap LUSolver() (GMGLinearSolvers.jl:54,     */
/* 423-434, 474): exact sparse direct solve.  Restated as banded LU    */
/* with partial pivoting (dgbtf2 / dgbtrs storage and order).          */
/* ------------------------------------------------------------------ */
typedef struct {
  int n, kl, ku, ldab;
  double *ab; /* ldab x n column-major, ldab = 2kl+ku+1 */
  int *piv;
} orc_band;

static orc_band *band_factor(i64 n64, const i64 *ptr, const i32 *idx, const double *val)
{
  int n = (int)n64, kl = 0, ku = 0;
  for (int i = 0; i < n; ++i)
    for (i64 k = ptr[i]; k < ptr[i + 1]; ++k) {
      int j = idx[k];
      if (i - j > kl) kl = i - j;
      if (j - i > ku) ku = j - i;
    }
  orc_band *b = (orc_band *)calloc(1, sizeof(orc_band));
  b->n = n; b->kl = kl; b->ku = ku; b->ldab = 2 * kl + ku + 1;
  b->ab = (double *)calloc((size_t)b->ldab * n, sizeof(double));
  b->piv = (int *)calloc(n, sizeof(int));
  int kv = kl + ku, ld = b->ldab;
  double *ab = b->ab;
#define AB(i, j) ab[(size_t)(kv + (i) - (j)) + (size_t)(j) * ld]
  for (int i = 0; i < n; ++i)
    for (i64 k = ptr[i]; k < ptr[i + 1]; ++k) AB(i, idx[k]) += val[k];
  int ju = 0;
  for (int j = 0; j < n; ++j) {
    int km = (kl < n - 1 - j) ? kl : n - 1 - j;
    int p = j;
    double mx = fabs(AB(j, j));
    for (int i = j + 1; i <= j + km; ++i) {
      double v = fabs(AB(i, j));
      if (v > mx) { mx = v; p = i; }
    }
    b->piv[j] = p;
    int jup = p + ku; if (jup > n - 1) jup = n - 1;
    if (jup > ju) ju = jup;
    if (ju < j) ju = j;
    if (p != j)
      for (int k = j; k <= ju; ++k) { double t = AB(j, k); AB(j, k) = AB(p, k); AB(p, k) = t; }
    double d = AB(j, j);
    if (d != 0.0 && km > 0) {
      double rd = 1.0 / d;
      for (int i = j + 1; i <= j + km; ++i) AB(i, j) *= rd;
      for (int k = j + 1; k <= ju; ++k) {
        double ajk = AB(j, k);
        if (ajk != 0.0)
          for (int i = j + 1; i <= j + km; ++i) AB(i, k) -= AB(i, j) * ajk;
      }
    }
  }
  return b;
}
static void band_solve(const orc_band *b, double *x)
{
  int n = b->n, kl = b->kl, ku = b->ku, kv = kl + ku, ld = b->ldab;
  const double *ab = b->ab;
  for (int j = 0; j < n; ++j) {
    int km = (kl < n - 1 - j) ? kl : n - 1 - j;
    int p = b->piv[j];
    if (p != j) { double t = x[j]; x[j] = x[p]; x[p] = t; }
    double xj = x[j];
    for (int i = j + 1; i <= j + km; ++i) x[i] -= AB(i, j) * xj;
  }
  for (int j = n - 1; j >= 0; --j) {
    x[j] /= AB(j, j);
    double xj = x[j];
    int lo = j - kv; if (lo < 0) lo = 0;
    for (int i = lo; i < j; ++i) x[i] -= AB(i, j) * xj;
  }
#undef AB
}
static void band_free(orc_band *b) { if (b) { free(b->ab); free(b->piv); free(b); } }

ORC_API void orc_direct_solve(i64 n, const i64 *ptr, const i32 *idx, const double *val,
                              const double *rhs, double *x)
{
  orc_band *b = band_factor(n, ptr, idx, val);
  memcpy(x, rhs, (size_t)n * sizeof(double));
  band_solve(b, x);
  band_free(b);
}

/* ------------------------------------------------------------------ */
/* Smoothers                                                           */
/* ------------------------------------------------------------------ */
enum { ORC_SM_JACOBI = 0, ORC_SM_PATCH = 1, ORC_SM_BLOCKJACOBI = 2 };

typedef struct {
  int kind, niter;
  double omega;
  /* Jacobi: JacobiLinearSolvers.jl:20-23 */
  double *inv_diag;
  /* Patch / block-Jacobi */
  i64 npatch;
  const i64 *patch_ptr;
  const i32 *patch_dofs;       /* patch_rows: b[rows_p] (PatchSolvers.jl:237-240) */
  const i32 *patch_cols;       /* patch_cols: x[cols_p] += x_p (:296); NULL = patch_rows */
  const double *user_mats;     /* caller-assembled patch matrices, column-major, concatenated (PatchSolvers.jl:137-150); NULL: A[rows_p,cols_p] */
  double **factors; /* per patch, column-major n_p x n_p LU (PatchSolvers.jl:176) */
  int **pivots;
  int max_np;
  double *xp, *Ak;
} orc_smoother;

typedef struct {
  i64 n, m; /* rows, cols */
  const i64 *ptr;
  const i32 *idx;
  const double *val;
} orc_csr;

static double csr_get(const orc_csr *A, i64 i, i32 j)
{
  double v = 0.0;
  for (i64 k = A->ptr[i]; k < A->ptr[i + 1]; ++k)
    if (A->idx[k] == j) v += A->val[k];
  return v;
}

/* numerical_setup(JacobiSymbolicSetup,A): JacobiLinearSolvers.jl:20-23 */
ORC_API void orc_jacobi_setup(i64 n, const i64 *ptr, const i32 *idx, const double *val,
                              double *inv_diag)
{
  orc_csr A = { n, n, ptr, idx, val };
  for (i64 i = 0; i < n; ++i) inv_diag[i] = 1.0 / csr_get(&A, i, (i32)i);
}

/* copyto!(Ak, view(A,rows,cols)): BlockJacobiSolvers.jl:160 */
static void extract_block2(const orc_csr *A, int np, const i32 *rows, const i32 *cols, double *Ak)
{
  for (int c = 0; c < np; ++c)
    for (int r = 0; r < np; ++r) Ak[r + (size_t)c * np] = csr_get(A, rows[r], cols[c]);
}
static void extract_block(const orc_csr *A, int np, const i32 *dofs, double *Ak) { extract_block2(A, np, dofs, dofs, Ak); }
static const i32 *sm_cols(const orc_smoother *s, i64 p) { return (s->patch_cols ? s->patch_cols : s->patch_dofs) + s->patch_ptr[p]; }

static void smoother_setup(orc_smoother *s, const orc_csr *A)
{
  if (s->kind == ORC_SM_JACOBI) {
    s->inv_diag = (double *)malloc((size_t)A->n * sizeof(double));
    orc_jacobi_setup(A->n, A->ptr, A->idx, A->val, s->inv_diag);
    return;
  }
  s->max_np = 0;
  for (i64 p = 0; p < s->npatch; ++p) {
    int np = (int)(s->patch_ptr[p + 1] - s->patch_ptr[p]);
    if (np > s->max_np) s->max_np = np;
  }
  s->xp = (double *)malloc((size_t)(s->max_np + 1) * sizeof(double));
  s->Ak = (double *)malloc((size_t)(s->max_np * s->max_np + 1) * sizeof(double));
  if (s->kind == ORC_SM_PATCH) {
    /* patch matrices of a :star patch = A[rows_p, cols_p]; factors = lu!(patch_mat)
     * PatchSolvers.jl:175-188 (collect_factorizations=true) */
    s->factors = (double **)calloc((size_t)s->npatch, sizeof(double *));
    s->pivots = (int **)calloc((size_t)s->npatch, sizeof(int *));
    size_t moff = 0;
    for (i64 p = 0; p < s->npatch; ++p) {
      int np = (int)(s->patch_ptr[p + 1] - s->patch_ptr[p]);
      if (np == 0) continue;
      s->factors[p] = (double *)malloc((size_t)np * np * sizeof(double));
      s->pivots[p] = (int *)malloc((size_t)np * sizeof(int));
      if (s->user_mats) { memcpy(s->factors[p], s->user_mats + moff, (size_t)np * np * sizeof(double)); moff += (size_t)np * np; }
      else extract_block2(A, np, s->patch_dofs + s->patch_ptr[p], sm_cols(s, p), s->factors[p]);
      dense_lu_pivot(np, s->factors[p], s->pivots[p]);
    }
  }
}

/* solve!(x,ns::JacobiNumericalSetup,b): JacobiLinearSolvers.jl:43-47 */
static void jacobi_apply(const orc_smoother *s, i64 n, double *x, const double *b)
{
  ORC_PFOR
  for (i64 i = 0; i < n; ++i) x[i] = s->inv_diag[i] * b[i];
}

/* solve_patch_overlapping!: PatchSolvers.jl:279-300 */
static void patch_apply(orc_smoother *s, i64 n, double *x, const double *b)
{
  for (i64 i = 0; i < n; ++i) x[i] = 0.0;              /* :287 fill!(x,0) */
  for (i64 p = 0; p < s->npatch; ++p) {                 /* :288 */
    int np = (int)(s->patch_ptr[p + 1] - s->patch_ptr[p]);
    if (np == 0) continue;                              /* :290 */
    const i32 *rows = s->patch_dofs + s->patch_ptr[p];
    const i32 *cols = sm_cols(s, p);
    for (int k = 0; k < np; ++k) s->xp[k] = b[rows[k]]; /* :237-240 Reindex(b) over patch_rows */
    dense_lu_solve(np, s->factors[p], s->pivots[p], s->xp); /* :295 ldiv! */
    for (int k = 0; k < np; ++k) x[cols[k]] += s->xp[k];    /* :296 */
  }
}

/* solve!(x,ns::BlockJacobiNS,b) -> solve_block_jacobi!: BlockJacobiSolvers.jl:119-123,141-170 */
static void blockjacobi_apply(orc_smoother *s, const orc_csr *A, i64 n, double *x, const double *b)
{
  for (i64 i = 0; i < n; ++i) x[i] = 0.0;              /* :120 */
  for (i64 p = 0; p < s->npatch; ++p) {                 /* :149 */
    int np = (int)(s->patch_ptr[p + 1] - s->patch_ptr[p]);
    if (np == 0) continue;
    const i32 *rows = s->patch_dofs + s->patch_ptr[p];
    const i32 *cols = sm_cols(s, p);
    extract_block2(A, np, rows, cols, s->Ak);           /* :160 */
    for (int k = 0; k < np; ++k) s->xp[k] = b[rows[k]]; /* :161 */
    dense_lu_nopivot(np, s->Ak);                        /* :162 */
    dense_lu_solve(np, s->Ak, NULL, s->xp);             /* :164 */
    for (int k = 0; k < np; ++k) x[cols[k]] += s->xp[k];/* :166 */
  }
}

static void precond_apply(orc_smoother *s, const orc_csr *A, double *dx, const double *r)
{
  if (s->kind == ORC_SM_JACOBI) jacobi_apply(s, A->n, dx, r);
  else if (s->kind == ORC_SM_PATCH) patch_apply(s, A->n, dx, r);
  else blockjacobi_apply(s, A, A->n, dx, r);
}

/* A3: solve!(x,ns::RichardsonSmootherNumericalSetup,r): RichardsonSmoothers.jl:84-98
 * Updates BOTH x and r in place. */
static void richardson_solve(orc_smoother *s, const orc_csr *A, double *x, double *r,
                             double *dx, double *Adx)
{
  i64 n = A->n;
  int iter = 1;
  for (i64 i = 0; i < n; ++i) dx[i] = 0.0;               /* :89 */
  while (iter <= s->niter) {                              /* :90 */
    precond_apply(s, A, dx, r);                           /* :91 solve!(dx,Mns,r) */
    ORC_PFOR
    for (i64 i = 0; i < n; ++i) dx[i] = s->omega * dx[i]; /* :92 */
    ORC_PFOR
    for (i64 i = 0; i < n; ++i) x[i] = x[i] + dx[i];      /* :93 */
    orc_spmv(n, A->ptr, A->idx, A->val, dx, Adx);         /* :94 */
    ORC_PFOR
    for (i64 i = 0; i < n; ++i) r[i] = r[i] - Adx[i];     /* :95 */
    iter += 1;
  }
}

/* ------------------------------------------------------------------ */
/* GMG object: GMGLinearSolvers.jl:172-210                             */
/* ------------------------------------------------------------------ */
enum { ORC_MODE_PRECONDITIONER = 0, ORC_MODE_SOLVER = 1 };
enum { ORC_V_CYCLE = 0, ORC_W_CYCLE = 1, ORC_F_CYCLE = 2 };

typedef struct {
  orc_csr A, P, R;
  int has_R;
  orc_smoother pre, post;
  int post_is_pre;
  orc_smoother pcorr;   /* patch tables + LU factors of the patch-corrected prolongation */
  int has_pcorr;
  orc_csr G;            /* rhs form of the correction (PatchTransferOperators.jl:30-50); has_G = 0: the level operator */
  int has_G;
  double *ptmp, *pcor;
  /* work vectors GMGLinearSolvers.jl:451-466 + smoother caches RichardsonSmoothers.jl:58-63 */
  double *dxh, *Adxh, *dxH, *rH, *sm_dx, *sm_Adx;
} orc_level;

typedef struct {
  int nlev;
  orc_level *lev;
  orc_band *coarse;
  /* coarsest_solver other than LUSolver(): CGSolver(JacobiLinearSolver();maxiter,atol,rtol) (GMGLinearSolvers.jl:54,423-434
   * accept any LinearSolver; the reference's MPI tests and applications pass iterative / PETSc solvers) */
  int coarse_cg, coarse_maxiter, coarse_niters;
  double coarse_atol, coarse_rtol;
  double *coarse_inv_diag;
  double *rh; /* finest_level_cache :391-396 */
  int mode, cycle;
  orc_log log;
} orc_gmg;

ORC_API orc_gmg *orc_gmg_create(int nlev)
{
  orc_gmg *g = (orc_gmg *)calloc(1, sizeof(orc_gmg));
  g->nlev = nlev;
  g->lev = (orc_level *)calloc((size_t)nlev, sizeof(orc_level));
  return g;
}
ORC_API void orc_gmg_set_matrix(orc_gmg *g, int l, i64 n, const i64 *ptr, const i32 *idx,
                                const double *val)
{
  g->lev[l].A = (orc_csr){ n, n, ptr, idx, val };
}
/* interp[l]: level l+1 -> l.  `mul!(dxh,interp,dxH)` GMGLinearSolvers.jl:491;
 * semantics y = P x, GridTransferOperators.jl:391-401 */
ORC_API void orc_gmg_set_prolongation(orc_gmg *g, int l, i64 n, i64 m, const i64 *ptr,
                                      const i32 *idx, const double *val)
{
  g->lev[l].P = (orc_csr){ n, m, ptr, idx, val };
}
/* restrict[l]: level l -> l+1. `mul!(rH,restrict,rh)` :484 ; = P^T in :residual mode
 * (GridTransferOperators.jl:202-209,536-547) */
ORC_API void orc_gmg_set_restriction(orc_gmg *g, int l, i64 n, i64 m, const i64 *ptr,
                                     const i32 *idx, const double *val)
{
  g->lev[l].R = (orc_csr){ n, m, ptr, idx, val };
  g->lev[l].has_R = 1;
}
/* patch_cols / user_mats: see orc_smoother; both optional (NULL) */
ORC_API void orc_gmg_set_smoother_ex(orc_gmg *g, int l, int which, int kind, int niter, double omega, i64 npatch,
                                     const i64 *patch_ptr, const i32 *patch_rows, const i32 *patch_cols, const double *user_mats);
ORC_API void orc_gmg_set_smoother(orc_gmg *g, int l, int which /*0 pre,1 post,2 both*/, int kind,
                                  int niter, double omega, i64 npatch, const i64 *patch_ptr,
                                  const i32 *patch_dofs)
{
  orc_gmg_set_smoother_ex(g, l, which, kind, niter, omega, npatch, patch_ptr, patch_dofs, NULL, NULL);
}
ORC_API void orc_gmg_set_smoother_ex(orc_gmg *g, int l, int which, int kind, int niter, double omega, i64 npatch,
                                     const i64 *patch_ptr, const i32 *patch_dofs, const i32 *patch_cols, const double *user_mats)
{
  orc_smoother s;
  memset(&s, 0, sizeof(s));
  s.kind = kind; s.niter = niter; s.omega = omega;
  s.npatch = npatch; s.patch_ptr = patch_ptr; s.patch_dofs = patch_dofs;
  s.patch_cols = patch_cols; s.user_mats = user_mats;
  if (which == 0 || which == 2) g->lev[l].pre = s;
  if (which == 1 || which == 2) g->lev[l].post = s;
  g->lev[l].post_is_pre = (which == 2);
}

/* PatchProlongationOperator(lev,sh,lhs,rhs,...) with rhs = lhs = level operator:
 * PatchTransferOperators.jl:2-60 ; mul! :153-172 */
ORC_API void orc_gmg_set_prolongation_correction(orc_gmg *g, int l, int kind, i64 npatch, const i64 *patch_ptr,
                                                 const i32 *patch_dofs)
{
  orc_smoother s;
  memset(&s, 0, sizeof(s));
  s.kind = kind; s.niter = 0; s.omega = 1.0;
  s.npatch = npatch; s.patch_ptr = patch_ptr; s.patch_dofs = patch_dofs;
  g->lev[l].pcorr = s;
  g->lev[l].has_pcorr = 1;
}

static double *dalloc(i64 n) { return (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double)); }

ORC_API void orc_gmg_set_prolongation_correction_rhs(orc_gmg *g, int l, i64 n, const i64 *ptr, const i32 *idx, const double *val)
{
  g->lev[l].G = (orc_csr){ n, n, ptr, idx, val };
  g->lev[l].has_G = 1;
}

ORC_API void orc_gmg_set_coarse_cg(orc_gmg *g, int maxiter, double atol, double rtol)
{
  g->coarse_cg = 1; g->coarse_maxiter = maxiter; g->coarse_atol = atol; g->coarse_rtol = rtol;
}
ORC_API int orc_gmg_coarse_niters(const orc_gmg *g) { return g->coarse_niters; }
ORC_API int orc_cg_solve(i64 n, const i64 *ptr, const i32 *idx, const double *val, int pc_kind, void *Pl,
                         double *x, const double *b, int maxiter, double atol, double rtol,
                         int flexible, int *niters, double *hist);
/* solve!(xh, coarsest_solver_cache, rh) GMGLinearSolvers.jl:474 ; xh holds fill!(dxH,0) (:487) = CG's initial guess */
static void coarse_solve(orc_gmg *g, double *xh, const double *rh)
{
  const orc_csr *A = &g->lev[g->nlev - 1].A;
  if (g->coarse_cg) {
    orc_cg_solve(A->n, A->ptr, A->idx, A->val, 2 /* ORC_PC_JACOBI */, g->coarse_inv_diag, xh, rh,
                 g->coarse_maxiter, g->coarse_atol, g->coarse_rtol, 0, &g->coarse_niters, NULL);
    return;
  }
  memcpy(xh, rh, (size_t)A->n * sizeof(double));
  band_solve(g->coarse, xh);
}

/* numerical_setup(GMGSymbolicSetup,mat): GMGLinearSolvers.jl:183-210 */
ORC_API void orc_gmg_setup(orc_gmg *g, int mode, int cycle, int maxiter, double atol, double rtol)
{
  g->mode = mode; g->cycle = cycle;
  g->log.maxiter = maxiter; g->log.atol = atol; g->log.rtol = rtol;
  g->log.residuals = (double *)calloc((size_t)(maxiter > 0 ? maxiter : 0) + 1, sizeof(double));
  g->rh = dalloc(g->lev[0].A.n);
  for (int l = 0; l < g->nlev - 1; ++l) {
    orc_level *L = &g->lev[l];
    i64 n = L->A.n, nH = g->lev[l + 1].A.n;
    L->dxh = dalloc(n); L->Adxh = dalloc(n); L->dxH = dalloc(nH); L->rH = dalloc(nH);
    L->sm_dx = dalloc(n); L->sm_Adx = dalloc(n);
    smoother_setup(&L->pre, &L->A);
    if (L->post_is_pre) L->post = L->pre; else smoother_setup(&L->post, &L->A);
    if (L->has_pcorr) { smoother_setup(&L->pcorr, &L->A); L->ptmp = dalloc(n); L->pcor = dalloc(n); }
  }
  const orc_csr *AL = &g->lev[g->nlev - 1].A;
  if (g->coarse_cg) {                                           /* JacobiLinearSolvers.jl:20-23 */
    g->coarse_inv_diag = dalloc(AL->n);
    for (i64 i = 0; i < AL->n; ++i) {
      double d = 0.0;
      for (i64 k = AL->ptr[i]; k < AL->ptr[i + 1]; ++k) if (AL->idx[k] == i) d += AL->val[k];
      g->coarse_inv_diag[i] = 1.0 / d;
    }
  } else
  g->coarse = band_factor(AL->n, AL->ptr, AL->idx, AL->val); /* :423-434 */
}

/* mul!(y,A::PatchProlongationOperator,x): PatchTransferOperators.jl:153-172
 *   interpolate!(uH,fv_h,Uh)                 -> dxh = P dxH
 *   patch_b = assemble_vector(liform,...)    -> (A dxh) restricted to the patch rows
 *   solve_patch_overlapping!(dx_h,...)       -> c = sum_p A_pp^-1 b_p
 *   fv_h .= fv_h .- dx_h                     -> dxh -= c */
static void apply_prolongation(orc_level *L, const double *dxH, double *dxh)
{
  orc_spmv(L->P.n, L->P.ptr, L->P.idx, L->P.val, dxH, dxh);
  if (!L->has_pcorr) return;
  if (L->has_G) orc_spmv(L->G.n, L->G.ptr, L->G.idx, L->G.val, dxh, L->ptmp);
  else orc_spmv(L->A.n, L->A.ptr, L->A.idx, L->A.val, dxh, L->ptmp);
  precond_apply(&L->pcorr, &L->A, L->pcor, L->ptmp);
  for (i64 i = 0; i < L->A.n; ++i) dxh[i] = dxh[i] - L->pcor[i];
}

static void apply_restriction(const orc_level *L, const double *rh, double *rH)
{
  if (L->has_R) { orc_spmv(L->R.n, L->R.ptr, L->R.idx, L->R.val, rh, rH); return; }
  /* R = P^T applied by column scatter of P (CSC of P^T == CSR of P) */
  i64 nH = L->P.m;
  for (i64 j = 0; j < nH; ++j) rH[j] = 0.0;
  for (i64 i = 0; i < L->P.n; ++i)
    for (i64 k = L->P.ptr[i]; k < L->P.ptr[i + 1]; ++k) rH[L->P.idx[k]] += L->P.val[k] * rh[i];
}

/* A1: gmg_v_cycle! :468-502, gmg_w_cycle! :504-555, gmg_f_cycle! :557-610.
 * ctype: which routine this level runs. */
static void gmg_cycle(orc_gmg *g, int lev, double *xh, double *rh, int ctype)
{
  if (lev == g->nlev - 1) {                                     /* :472-474 */
    coarse_solve(g, xh, rh);
    return;
  }
  orc_level *L = &g->lev[lev];
  i64 n = L->A.n, nH = g->lev[lev + 1].A.n;
  richardson_solve(&L->pre, &L->A, xh, rh, L->sm_dx, L->sm_Adx);   /* :481 */
  apply_restriction(L, rh, L->rH);                                 /* :484 */
  for (i64 i = 0; i < nH; ++i) L->dxH[i] = 0.0;                    /* :487 */
  gmg_cycle(g, lev + 1, L->dxH, L->rH, ctype);                     /* :488 */
  apply_prolongation(L, L->dxH, L->dxh);                           /* :491 */
  ORC_PFOR
  for (i64 i = 0; i < n; ++i) xh[i] = xh[i] + L->dxh[i];           /* :494 */
  orc_spmv(n, L->A.ptr, L->A.idx, L->A.val, L->dxh, L->Adxh);      /* :495 */
  ORC_PFOR
  for (i64 i = 0; i < n; ++i) rh[i] = rh[i] - L->Adxh[i];          /* :496 */
  if (ctype != ORC_V_CYCLE) {
    /* W: :531-547 ; F: :584-600 (second visit is a V-cycle in F) */
    richardson_solve(&L->post, &L->A, xh, rh, L->sm_dx, L->sm_Adx);
    apply_restriction(L, rh, L->rH);
    for (i64 i = 0; i < nH; ++i) L->dxH[i] = 0.0;
    gmg_cycle(g, lev + 1, L->dxH, L->rH, ctype == ORC_W_CYCLE ? ORC_W_CYCLE : ORC_V_CYCLE);
    apply_prolongation(L, L->dxH, L->dxh);
    for (i64 i = 0; i < n; ++i) xh[i] = xh[i] + L->dxh[i];
    orc_spmv(n, L->A.ptr, L->A.idx, L->A.val, L->dxh, L->Adxh);
    for (i64 i = 0; i < n; ++i) rh[i] = rh[i] - L->Adxh[i];
  }
  richardson_solve(&L->post, &L->A, xh, rh, L->sm_dx, L->sm_Adx);  /* :499 */
}

/* A2: solve!(x,ns::GMGNumericalSetup,b): GMGLinearSolvers.jl:612-645 */
ORC_API int orc_gmg_solve(orc_gmg *g, double *x, const double *b, int *niters, double *hist)
{
  const orc_csr *A = &g->lev[0].A;
  i64 n = A->n;
  double *rh = g->rh;
  if (g->mode == ORC_MODE_PRECONDITIONER) {
    for (i64 i = 0; i < n; ++i) x[i] = 0.0;                  /* :619 */
    memcpy(rh, b, (size_t)n * sizeof(double));               /* :620 */
  } else {
    orc_spmv(n, A->ptr, A->idx, A->val, x, rh);              /* :623 */
    for (i64 i = 0; i < n; ++i) rh[i] = b[i] - rh[i];        /* :624 */
  }
  double res = orc_norm(n, rh);                              /* :627 */
  int done = log_init(&g->log, res);                         /* :628 */
  while (!done) {
    gmg_cycle(g, 0, x, rh, g->cycle);                        /* :630-637 */
    res = orc_norm(n, rh);                                   /* :639 */
    done = log_update(&g->log, res);                         /* :640 */
  }
  int flag = log_finalize(&g->log, res);                     /* :643 */
  if (niters) *niters = g->log.num_iters;
  if (hist) memcpy(hist, g->log.residuals, (size_t)(g->log.num_iters + 1) * sizeof(double));
  return flag;
}

/* Standalone smoother application for per-kernel parity tests:
 * solve!(x, RichardsonSmootherNumericalSetup, r) on level l. */
ORC_API void orc_gmg_smooth(orc_gmg *g, int l, int post, double *x, double *r)
{
  orc_level *L = &g->lev[l];
  richardson_solve(post ? &L->post : &L->pre, &L->A, x, r, L->sm_dx, L->sm_Adx);
}
/* M^{-1} r for the level-l smoother's inner solver (Jacobi / patch / block-Jacobi) */
ORC_API void orc_gmg_precond(orc_gmg *g, int l, double *dx, const double *r)
{
  precond_apply(&g->lev[l].pre, &g->lev[l].A, dx, r);
}
ORC_API void orc_gmg_restrict(orc_gmg *g, int l, const double *rh, double *rH)
{
  apply_restriction(&g->lev[l], rh, rH);
}
ORC_API void orc_gmg_coarse_solve(orc_gmg *g, const double *rhs, double *x)
{
  for (i64 i = 0; i < g->lev[g->nlev - 1].A.n; ++i) x[i] = 0.0;
  coarse_solve(g, x, rhs);
}

static void smoother_free(orc_smoother *s)
{
  free(s->inv_diag); free(s->xp); free(s->Ak);
  if (s->factors) for (i64 p = 0; p < s->npatch; ++p) { free(s->factors[p]); free(s->pivots[p]); }
  free(s->factors); free(s->pivots);
}
ORC_API void orc_gmg_destroy(orc_gmg *g)
{
  if (!g) return;
  for (int l = 0; l < g->nlev - 1; ++l) {
    orc_level *L = &g->lev[l];
    free(L->dxh); free(L->Adxh); free(L->dxH); free(L->rH); free(L->sm_dx); free(L->sm_Adx);
    smoother_free(&L->pre);
    if (!L->post_is_pre) smoother_free(&L->post);
    if (L->has_pcorr) { smoother_free(&L->pcorr); free(L->ptmp); free(L->pcor); }
  }
  if (g->coarse) band_free(g->coarse);
  free(g->coarse_inv_diag);
  free(g->rh); free(g->log.residuals); free(g->lev); free(g);
}

/* preconditioner dispatch for the Krylov solvers.
 * kind 0: nothing ; 1: GMG (pc = orc_gmg*) ; 2: JacobiLinearSolver (pc = inv_diag,
 * JacobiLinearSolvers.jl:43-47) ; 3: block diagonal / triangular solver (pc = orc_block*) */
enum { ORC_PC_NONE = 0, ORC_PC_GMG = 1, ORC_PC_JACOBI = 2, ORC_PC_BLOCK = 3 };
typedef struct orc_block orc_block;
ORC_API void orc_block_apply(orc_block *B, double *x, const double *b);
static void pc_solve(int kind, void *pc, i64 n, double *z, const double *r)
{
  if (kind == ORC_PC_GMG) orc_gmg_solve((orc_gmg *)pc, z, r, NULL, NULL);
  else if (kind == ORC_PC_BLOCK) orc_block_apply((orc_block *)pc, z, r);
  else if (kind == ORC_PC_JACOBI) { const double *d = (const double *)pc; for (i64 i = 0; i < n; ++i) z[i] = d[i] * r[i]; }
  else memcpy(z, r, (size_t)n * sizeof(double));
}

/* ------------------------------------------------------------------ */
/* A7: solve!(x,ns::CGNumericalSetup,b): Krylov/CGSolvers.jl:73-120    */
/* ------------------------------------------------------------------ */
ORC_API int orc_cg_solve(i64 n, const i64 *ptr, const i32 *idx, const double *val, int pc_kind, void *Pl,
                         double *x, const double *b, int maxiter, double atol, double rtol,
                         int flexible, int *niters, double *hist)
{
  double *w = dalloc(n), *p = dalloc(n), *z = dalloc(n), *r = dalloc(n); /* :42-48 */
  orc_log log = { maxiter, atol, rtol, 0, (double *)calloc((size_t)(maxiter > 0 ? maxiter : 0) + 1, sizeof(double)) };
  orc_spmv(n, ptr, idx, val, x, w);                         /* :79 */
  for (i64 i = 0; i < n; ++i) r[i] = b[i] - w[i];
  for (i64 i = 0; i < n; ++i) p[i] = 0.0;                   /* :80 */
  for (i64 i = 0; i < n; ++i) z[i] = 0.0;                   /* :81 */
  double gamma = 1.0, beta, alpha;                          /* :82 */
  double res = orc_norm(n, r);                              /* :85 */
  int done = log_init(&log, res);                           /* :86 */
  while (!done) {
    if (pc_kind == ORC_PC_NONE) {                           /* :90-92 */
      for (i64 i = 0; i < n; ++i) z[i] = r[i];
      beta = gamma; gamma = orc_dot(n, r, r); beta = gamma / beta;
    } else if (!flexible) {                                 /* :93-95 */
      pc_solve(pc_kind, Pl, n, z, r);
      beta = gamma; gamma = orc_dot(n, z, r); beta = gamma / beta;
    } else {                                                /* :96-99 */
      double delta = orc_dot(n, z, r);
      pc_solve(pc_kind, Pl, n, z, r);
      beta = gamma; gamma = orc_dot(n, z, r); beta = (gamma - delta) / beta;
    }
    ORC_PFOR
    for (i64 i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];  /* :101 */
    orc_spmv(n, ptr, idx, val, p, w);                       /* :104 */
    alpha = gamma / orc_dot(n, p, w);                       /* :105 */
    ORC_PFOR
    for (i64 i = 0; i < n; ++i) x[i] += alpha * p[i];       /* :108 */
    ORC_PFOR
    for (i64 i = 0; i < n; ++i) r[i] -= alpha * w[i];       /* :109 */
    res = orc_norm(n, r);                                   /* :111 */
    done = log_update(&log, res);                           /* :112 */
  }
  int flag = log_finalize(&log, res);                       /* :118 */
  if (niters) *niters = log.num_iters;
  if (hist) memcpy(hist, log.residuals, (size_t)(log.num_iters + 1) * sizeof(double));
  free(w); free(p); free(z); free(r); free(log.residuals);
  return flag;
}

/* ------------------------------------------------------------------ */
/* LinearAlgebra.givensAlgorithm(f,g) (Julia stdlib, LAPACK dlartg)    */
/* used at FGMRESSolvers.jl:175                                        */
/* ------------------------------------------------------------------ */
static void givens_algorithm(double f, double g, double *cs, double *sn, double *r)
{
  /* floatmin2(Float64) = 2^(exponent(floatmin/eps) / 2 rounded) ; Julia:
   * floatmin2(::Type{Float64}) = reinterpret(Float64, 0x21a0000000000000) = 2^-485 */
  const double safmn2 = ldexp(1.0, -485);
  const double safmx2 = 1.0 / safmn2;
  if (g == 0.0) { *cs = 1.0; *sn = 0.0; *r = f; return; }
  if (f == 0.0) { *cs = 0.0; *sn = 1.0; *r = g; return; }
  double f1 = f, g1 = g, scale = fmax(fabs(f1), fabs(g1));
  if (scale >= safmx2) {
    int count = 0;
    do { count++; f1 *= safmn2; g1 *= safmn2; scale = fmax(fabs(f1), fabs(g1)); } while (scale >= safmx2);
    *r = sqrt(f1 * f1 + g1 * g1); *cs = f1 / *r; *sn = g1 / *r;
    for (int i = 0; i < count; ++i) *r *= safmx2;
  } else if (scale <= safmn2) {
    int count = 0;
    do { count++; f1 *= safmx2; g1 *= safmx2; scale = fmax(fabs(f1), fabs(g1)); } while (scale <= safmn2);
    *r = sqrt(f1 * f1 + g1 * g1); *cs = f1 / *r; *sn = g1 / *r;
    for (int i = 0; i < count; ++i) *r *= safmn2;
  } else {
    *r = sqrt(f1 * f1 + g1 * g1); *cs = f1 / *r; *sn = g1 / *r;
  }
  if (fabs(f) > fabs(g) && *cs < 0.0) { *cs = -*cs; *sn = -*sn; *r = -*r; }
}
ORC_API void orc_givens(double f, double g, double *out3) { givens_algorithm(f, g, out3, out3 + 1, out3 + 2); }

/* ------------------------------------------------------------------ */
/* A8: solve!(x,ns::FGMRESNumericalSetup,b): FGMRESSolvers.jl:130-199  */
/* Pl = nothing ; Pr = GMG (or NULL = identity: solve!(wr,Pr,x) ~ copy) */
/* ------------------------------------------------------------------ */
ORC_API int orc_fgmres_solve_pl(i64 n, const i64 *ptr, const i32 *idx, const double *val, int pc_kind, void *Pr,
                                int pl_kind, void *Pl, double *x, const double *b, int m0, int restart, int m_add,
                                int maxiter, double atol, double rtol, int *niters, double *hist);
ORC_API int orc_fgmres_solve(i64 n, const i64 *ptr, const i32 *idx, const double *val, int pc_kind, void *Pr,
                             double *x, const double *b, int m0, int restart, int m_add,
                             int maxiter, double atol, double rtol, int *niters, double *hist)
{
  return orc_fgmres_solve_pl(n, ptr, idx, val, pc_kind, Pr, ORC_PC_NONE, NULL, x, b, m0, restart, m_add, maxiter, atol, rtol, niters, hist);
}
/* Pl != nothing: krylov_residual!(r,x,A,b,Pl,w) KrylovUtils.jl:46-50 and krylov_mul!(y,A,x,Pr,Pl,wr,wl) :14-18 */
ORC_API int orc_fgmres_solve_pl(i64 n, const i64 *ptr, const i32 *idx, const double *val, int pc_kind, void *Pr,
                                int pl_kind, void *Pl, double *x, const double *b, int m0, int restart, int m_add,
                                int maxiter, double atol, double rtol, int *niters, double *hist)
{
  int m = m0;
  double *zl = dalloc(n);                                   /* :66 zl cache */
  /* caches :58-70 ; growable :77-94 */
  double **V = (double **)calloc((size_t)m + 1, sizeof(double *));
  double **Z = (double **)calloc((size_t)m, sizeof(double *));
  for (int i = 0; i <= m; ++i) V[i] = dalloc(n);
  for (int i = 0; i < m; ++i) Z[i] = dalloc(n);
  /* H stored with fixed leading dimension big enough for any growth (with restart the basis never exceeds m0) */
  int mcap = restart ? m0 + 2 : m + (maxiter + 1) * (m_add > 0 ? m_add : 1) + 1;
  int ldh = mcap + 1;
  double *H = (double *)calloc((size_t)ldh * mcap, sizeof(double));
  double *g = (double *)calloc((size_t)mcap + 1, sizeof(double));
  double *c = (double *)calloc((size_t)mcap, sizeof(double));
  double *s = (double *)calloc((size_t)mcap, sizeof(double));
#define HH(i, j) H[(size_t)((i)-1) + (size_t)((j)-1) * ldh] /* 1-based like the reference */
  orc_log log = { maxiter, atol, rtol, 0, (double *)calloc((size_t)(maxiter > 0 ? maxiter : 0) + 1, sizeof(double)) };

  for (i64 i = 0; i < n; ++i) V[0][i] = 0.0;                /* :136 */
  /* krylov_residual!(V[1],x,A,b,nothing,zl): KrylovUtils.jl:51-54 */
  if (pl_kind == ORC_PC_NONE) {
    orc_spmv(n, ptr, idx, val, x, V[0]);
    for (i64 i = 0; i < n; ++i) V[0][i] = b[i] - V[0][i];   /* :140 */
  } else {
    orc_spmv(n, ptr, idx, val, x, zl);
    for (i64 i = 0; i < n; ++i) zl[i] = b[i] - zl[i];
    pc_solve(pl_kind, Pl, n, V[0], zl);
  }
  double beta = orc_norm(n, V[0]);                          /* :141 */
  int done = log_init(&log, beta);                          /* :142 */
  while (!done) {
    int j = 1;                                              /* :145 */
    for (i64 i = 0; i < n; ++i) V[0][i] /= beta;            /* :146 */
    memset(H, 0, (size_t)ldh * mcap * sizeof(double));      /* :147 */
    memset(g, 0, ((size_t)mcap + 1) * sizeof(double)); g[0] = beta; /* :148 */
    while (!done && !(restart && j > m0)) {                 /* :149, restart() :32-38 */
      if (j > m) {                                          /* :151-154 expand */
        int m_new = m + m_add;
        V = (double **)realloc(V, ((size_t)m_new + 1) * sizeof(double *));
        Z = (double **)realloc(Z, (size_t)m_new * sizeof(double *));
        for (int i = m + 1; i <= m_new; ++i) V[i] = dalloc(n);
        for (int i = m; i < m_new; ++i) Z[i] = dalloc(n);
        m = m_new;
      }
      double *Vn = V[j], *Zj = Z[j - 1];
      for (i64 i = 0; i < n; ++i) Vn[i] = 0.0;              /* :157 */
      for (i64 i = 0; i < n; ++i) Zj[i] = 0.0;              /* :158 */
      /* krylov_mul!(V[j+1],A,V[j],Pr,nothing,Z[j],zl): KrylovUtils.jl:22-25 */
      pc_solve(pc_kind, Pr, n, Zj, V[j - 1]);
      if (pl_kind == ORC_PC_NONE) orc_spmv(n, ptr, idx, val, Zj, Vn);   /* :159 */
      else { orc_spmv(n, ptr, idx, val, Zj, zl); pc_solve(pl_kind, Pl, n, Vn, zl); }
      for (int i = 1; i <= j; ++i) {                        /* :160-163 MGS */
        double h = orc_dot(n, Vn, V[i - 1]);
        HH(i, j) = h;
        for (i64 k = 0; k < n; ++k) Vn[k] = Vn[k] - h * V[i - 1][k];
      }
      HH(j + 1, j) = orc_norm(n, Vn);                       /* :164 */
      { double hn = HH(j + 1, j); for (i64 k = 0; k < n; ++k) Vn[k] /= hn; } /* :165 */
      for (int i = 1; i <= j - 1; ++i) {                    /* :168-172 */
        double gm = c[i - 1] * HH(i, j) + s[i - 1] * HH(i + 1, j);
        HH(i + 1, j) = -s[i - 1] * HH(i, j) + c[i - 1] * HH(i + 1, j);
        HH(i, j) = gm;
      }
      double rr;                                            /* :175 */
      givens_algorithm(HH(j, j), HH(j + 1, j), &c[j - 1], &s[j - 1], &rr);
      HH(j, j) = c[j - 1] * HH(j, j) + s[j - 1] * HH(j + 1, j); HH(j + 1, j) = 0.0; /* :176 */
      g[j] = -s[j - 1] * g[j - 1]; g[j - 1] = c[j - 1] * g[j - 1];                 /* :177 */
      beta = fabs(g[j]);                                    /* :179 */
      j += 1;                                               /* :180 */
      done = log_update(&log, beta);                        /* :181 */
    }
    j = j - 1;                                              /* :183 */
    for (int i = j; i >= 1; --i) {                          /* :186-188 */
      double acc = 0.0;
      for (int k = i + 1; k <= j; ++k) acc += HH(i, k) * g[k - 1];
      g[i - 1] = (g[i - 1] - acc) / HH(i, i);
    }
    for (int i = 1; i <= j; ++i)                            /* :191-193 */
      for (i64 k = 0; k < n; ++k) x[k] += g[i - 1] * Z[i - 1][k];
    if (pl_kind == ORC_PC_NONE) {
      orc_spmv(n, ptr, idx, val, x, V[0]);                  /* :194 krylov_residual! */
      for (i64 i = 0; i < n; ++i) V[0][i] = b[i] - V[0][i];
    } else {
      orc_spmv(n, ptr, idx, val, x, zl);
      for (i64 i = 0; i < n; ++i) zl[i] = b[i] - zl[i];
      pc_solve(pl_kind, Pl, n, V[0], zl);
    }
  }
#undef HH
  int flag = log_finalize(&log, beta);                      /* :197 */
  if (niters) *niters = log.num_iters;
  if (hist) memcpy(hist, log.residuals, (size_t)(log.num_iters + 1) * sizeof(double));
  for (int i = 0; i <= m; ++i) free(V[i]);
  for (int i = 0; i < m; ++i) free(Z[i]);
  free(V); free(Z); free(H); free(g); free(c); free(s); free(log.residuals); free(zl);
  return flag;
}

/* ------------------------------------------------------------------ */
/* LinearSolverFromSmoother: LinearSolverFromSmoothers.jl:44-50        */
/* (x=0; r=copy(b); solve!(x,smoother,r)) -- used by the reference's   */
/* SmoothersTests.jl known-answer test that pins this oracle.          */
/* CG preconditioned by a Richardson(Jacobi) smoother.                 */
/* ------------------------------------------------------------------ */
ORC_API int orc_cg_smoother_solve(i64 n, const i64 *ptr, const i32 *idx, const double *val,
                                  int sm_niter, double sm_omega, double *x, const double *b,
                                  int maxiter, double atol, double rtol, int *niters, double *hist)
{
  orc_csr A = { n, n, ptr, idx, val };
  orc_smoother S; memset(&S, 0, sizeof(S));
  S.kind = ORC_SM_JACOBI; S.niter = sm_niter; S.omega = sm_omega;
  smoother_setup(&S, &A);
  double *w = dalloc(n), *p = dalloc(n), *z = dalloc(n), *r = dalloc(n);
  double *aux = dalloc(n), *dx = dalloc(n), *Adx = dalloc(n);
  orc_log log = { maxiter, atol, rtol, 0, (double *)calloc((size_t)(maxiter > 0 ? maxiter : 0) + 1, sizeof(double)) };
  orc_spmv(n, ptr, idx, val, x, w);
  for (i64 i = 0; i < n; ++i) r[i] = b[i] - w[i];
  double gamma = 1.0, beta, alpha;
  double res = orc_norm(n, r);
  int done = log_init(&log, res);
  while (!done) {
    for (i64 i = 0; i < n; ++i) z[i] = 0.0;                 /* LinearSolverFromSmoothers.jl:46 */
    memcpy(aux, r, (size_t)n * sizeof(double));             /* :47 */
    richardson_solve(&S, &A, z, aux, dx, Adx);              /* :48 */
    beta = gamma; gamma = orc_dot(n, z, r); beta = gamma / beta;
    for (i64 i = 0; i < n; ++i) p[i] = z[i] + beta * p[i];
    orc_spmv(n, ptr, idx, val, p, w);
    alpha = gamma / orc_dot(n, p, w);
    for (i64 i = 0; i < n; ++i) x[i] += alpha * p[i];
    for (i64 i = 0; i < n; ++i) r[i] -= alpha * w[i];
    res = orc_norm(n, r);
    done = log_update(&log, res);
  }
  int flag = log_finalize(&log, res);
  if (niters) *niters = log.num_iters;
  if (hist) memcpy(hist, log.residuals, (size_t)(log.num_iters + 1) * sizeof(double));
  free(w); free(p); free(z); free(r); free(aux); free(dx); free(Adx); free(log.residuals);
  smoother_free(&S);
  return flag;
}

/* ------------------------------------------------------------------ */
/* Block preconditioners: the glue that calls the GMG  */
/* hot path once per outer FGMRES iteration in the Stokes application   */
/* (test/Applications/StokesGMG.jl:142-153).                            */
/*   BlockDiagonalSolver   solve!: BlockDiagonalSolvers.jl:165-177      */
/*   BlockTriangularSolver solve!: BlockTriangularSolvers.jl:186-242    */
/* Block vectors are contiguous, block i at off[i]..off[i+1].           */
/* ------------------------------------------------------------------ */
enum { ORC_BLOCK_DIAGONAL = 0, ORC_BLOCK_LOWER = 1, ORC_BLOCK_UPPER = 2 };
enum { ORC_BD_GMG = 1, ORC_BD_CG_JACOBI = 2, ORC_BD_LU = 3, ORC_BD_JACOBI = 4 };

typedef struct {
  int kind;
  orc_gmg *gmg;                 /* ORC_BD_GMG */
  orc_csr M;                    /* matrix of the block solver (CG / LU / Jacobi) */
  double *inv_diag;             /* JacobiLinearSolvers.jl:20-23 */
  orc_band *lu;
  int maxiter; double atol, rtol;
} orc_bdiag;

struct orc_block {
  int nb, kind;
  i64 *off;
  orc_bdiag *diag;
  orc_csr *offd;                /* nb*nb, row-major; ptr == NULL: no block */
  double *coeff;                /* nb*nb */
  double *w, *y;                /* work caches, BlockTriangularSolvers.jl:145-150 */
};

ORC_API orc_block *orc_block_create(int nb, const i64 *sizes, int kind)
{
  orc_block *B = (orc_block *)calloc(1, sizeof(orc_block));
  B->nb = nb; B->kind = kind;
  B->off = (i64 *)calloc((size_t)nb + 1, sizeof(i64));
  for (int i = 0; i < nb; ++i) B->off[i + 1] = B->off[i] + sizes[i];
  B->diag = (orc_bdiag *)calloc((size_t)nb, sizeof(orc_bdiag));
  B->offd = (orc_csr *)calloc((size_t)nb * nb, sizeof(orc_csr));
  B->coeff = (double *)calloc((size_t)nb * nb, sizeof(double));
  for (int i = 0; i < nb * nb; ++i) B->coeff[i] = 1.0;       /* BlockTriangularSolvers.jl:66 coeffs=fill(1.0,...) */
  B->w = dalloc(B->off[nb]); B->y = dalloc(B->off[nb]);       /* zero-initialised like allocate_in_domain + fill! */
  return B;
}
ORC_API void orc_block_set_offdiag(orc_block *B, int i, int j, i64 nrows, i64 ncols, const i64 *ptr, const i32 *idx,
                                   const double *val, double coeff)
{
  orc_csr M = { nrows, ncols, ptr, idx, val };
  B->offd[i * B->nb + j] = M;
  B->coeff[i * B->nb + j] = coeff;
}
ORC_API void orc_block_set_diag_gmg(orc_block *B, int i, orc_gmg *g) { B->diag[i].kind = ORC_BD_GMG; B->diag[i].gmg = g; }
/* kind: ORC_BD_CG_JACOBI = CGSolver(JacobiLinearSolver();maxiter,atol,rtol), ORC_BD_LU = LUSolver(),
 * ORC_BD_JACOBI = JacobiLinearSolver() */
ORC_API void orc_block_set_diag_matrix(orc_block *B, int i, int kind, i64 n, const i64 *ptr, const i32 *idx,
                                       const double *val, int maxiter, double atol, double rtol)
{
  orc_bdiag *D = &B->diag[i];
  orc_csr M = { n, n, ptr, idx, val };
  D->kind = kind; D->M = M; D->maxiter = maxiter; D->atol = atol; D->rtol = rtol;
  if (kind == ORC_BD_CG_JACOBI || kind == ORC_BD_JACOBI) {
    D->inv_diag = dalloc(n);
    orc_jacobi_setup(n, ptr, idx, val, D->inv_diag);
  } else if (kind == ORC_BD_LU) {
    D->lu = band_factor(n, ptr, idx, val);
  }
}
static void bdiag_solve(orc_bdiag *D, i64 n, double *y, const double *w)
{
  switch (D->kind) {
  case ORC_BD_GMG: orc_gmg_solve(D->gmg, y, w, NULL, NULL); break;
  case ORC_BD_CG_JACOBI:                                     /* y keeps its previous content = initial guess */
    orc_cg_solve(n, D->M.ptr, D->M.idx, D->M.val, ORC_PC_JACOBI, D->inv_diag, y, w, D->maxiter, D->atol, D->rtol, 0, NULL, NULL);
    break;
  case ORC_BD_LU: memcpy(y, w, (size_t)n * sizeof(double)); band_solve(D->lu, y); break;
  case ORC_BD_JACOBI: for (i64 k = 0; k < n; ++k) y[k] = D->inv_diag[k] * w[k]; break;
  default: memcpy(y, w, (size_t)n * sizeof(double));
  }
}
/* eps(x) of Julia for a Float64 */
static double julia_eps(double x) { x = fabs(x); return nextafter(x, INFINITY) - x; }

/* solve!(x,ns,b) */
ORC_API void orc_block_apply(orc_block *B, double *x, const double *b)
{
  const int NB = B->nb;
  for (int step = 0; step < NB; ++step) {
    const int iB = (B->kind == ORC_BLOCK_UPPER) ? NB - 1 - step : step;   /* :218 NB:-1:1 / :190 1:NB */
    const i64 o = B->off[iB], n = B->off[iB + 1] - o;
    double *wi = B->w + o, *yi = B->y + o;
    memcpy(wi, b + o, (size_t)n * sizeof(double));           /* :192,221 copy!(wi,bi) */
    if (B->kind != ORC_BLOCK_DIAGONAL) {
      const int j0 = (B->kind == ORC_BLOCK_UPPER) ? iB + 1 : 0;
      const int j1 = (B->kind == ORC_BLOCK_UPPER) ? NB : iB;
      for (int jB = j0; jB < j1; ++jB) {
        const double cij = B->coeff[iB * NB + jB];
        const orc_csr *M = &B->offd[iB * NB + jB];
        if (fabs(cij) > julia_eps(cij) && M->ptr) {          /* :194-197,223-226 mul!(wi,M,xj,-cij,1.0) */
          const double *xj = x + B->off[jB];
          for (i64 r = 0; r < n; ++r) {
            double s = wi[r];
            for (i64 k = M->ptr[r]; k < M->ptr[r + 1]; ++k) s += M->val[k] * (xj[M->idx[k]] * (-cij));
            wi[r] = s;
          }
        }
      }
    }
    bdiag_solve(&B->diag[iB], n, yi, wi);                    /* :202-205,231-234 solve!(yi,nsi,wi) */
    memcpy(x + o, yi, (size_t)n * sizeof(double));           /* copy!(xi,yi) */
  }
}
ORC_API void orc_block_destroy(orc_block *B)
{
  if (!B) return;
  for (int i = 0; i < B->nb; ++i) { free(B->diag[i].inv_diag); band_free(B->diag[i].lu); }
  free(B->off); free(B->diag); free(B->offd); free(B->coeff); free(B->w); free(B->y); free(B);
}

/* ------------------------------------------------------------------ */
/* RichardsonLinearSolver(omega,maxiter;Pl): RichardsonLinearSolvers.jl:79-106 */
/* r .= b ; mul!(r,A,x,-1,1) accumulates entry by entry into r (SparseArrays 5-arg mul!). */
/* ------------------------------------------------------------------ */
static void resid_accumulate(i64 n, const i64 *ptr, const i32 *idx, const double *val, const double *x, const double *b, double *r)
{
  for (i64 i = 0; i < n; ++i) {
    double s = b[i];
    for (i64 k = ptr[i]; k < ptr[i + 1]; ++k) s += val[k] * (x[idx[k]] * -1.0);
    r[i] = s;
  }
}
ORC_API int orc_richardson_solve(i64 n, const i64 *ptr, const i32 *idx, const double *val, int pc_kind, void *Pl, double omega,
                                 double *x, const double *b, int maxiter, double atol, double rtol, int *niters, double *hist)
{
  double *z = dalloc(n), *r = dalloc(n);
  orc_log log = { maxiter, atol, rtol, 0, (double *)calloc((size_t)(maxiter > 0 ? maxiter : 0) + 1, sizeof(double)) };
  resid_accumulate(n, ptr, idx, val, x, b, r);               /* :84-85 */
  double res = orc_norm(n, r);
  int done = log_init(&log, res);                            /* :86 */
  while (!done) {
    if (pc_kind != ORC_PC_NONE) {                            /* :88-94 */
      pc_solve(pc_kind, Pl, n, z, r);
      for (i64 i = 0; i < n; ++i) x[i] += omega * z[i];
    } else {                                                 /* :97-102 */
      for (i64 i = 0; i < n; ++i) x[i] += omega * r[i];
    }
    resid_accumulate(n, ptr, idx, val, x, b, r);
    res = orc_norm(n, r);
    done = log_update(&log, res);
  }
  int flag = log_finalize(&log, res);
  if (niters) *niters = log.num_iters;
  if (hist) memcpy(hist, log.residuals, (size_t)(log.num_iters + 1) * sizeof(double));
  free(z); free(r); free(log.residuals);
  return flag;
}

ORC_API int orc_threads(void)
{
#ifdef ORC_OMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* cpu_baseline only: dst = src copied by the same static row-block schedule the OpenMP loops use, so that the pages of
 * the operator arrays are first touched (NUMA-placed) by the threads that will stream them.  `dst` must be untouched
 * memory (np.empty).  elem = bytes per element, n = elements. */
ORC_API void orc_parallel_copy(void *dst, const void *src, i64 n, int elem)
{
  ORC_PFOR
  for (i64 i = 0; i < n; ++i) memcpy((char *)dst + (size_t)i * elem, (const char *)src + (size_t)i * elem, (size_t)elem);
}
