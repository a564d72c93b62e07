/*
 * abi_smoke.c -- the C ABI used from plain C (no Python, no torch): a 1-D Poisson problem
 * (tridiagonal [-1 2 -1], Dirichlet ends eliminated), 3-level GMG V-cycle preconditioner inside CG,
 * solved through include/gmg_amd.h exactly as the Julia wrapper would drive it.
 *
 * Build:  gcc -std=c99 -O2 -I include tests/c/abi_smoke.c -o abi_smoke -L gridapsolvers.jl_amd -lgmgamd -lm
 * Run (needs a GPU): LD_LIBRARY_PATH=gridapsolvers.jl_amd ./abi_smoke
 * Exit code 0 and "OK" when the solution matches the exact one (the CPU check below is the tridiagonal
 * Thomas algorithm, written here -- nothing from oracle/ is linked).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "gmg_amd.h"

#define CHECK(call) do { int st_ = (call); if (st_ != GMG_OK) { \
  fprintf(stderr, "%s -> status %d: %s\n", #call, st_, gmg_last_error(h)); return 2; } } while (0)

static void laplace_1d(int n, int64_t *ptr, int64_t *idx, double *val)
{
  int64_t k = 0;
  for (int i = 0; i < n; ++i) {
    ptr[i] = k;
    if (i > 0) { idx[k] = i - 1; val[k++] = -1.0; }
    idx[k] = i; val[k++] = 2.0;
    if (i + 1 < n) { idx[k] = i + 1; val[k++] = -1.0; }
  }
  ptr[n] = k;
}

/* linear interpolation from nc = (n-1)/2 coarse points to n fine points (coarse point j sits at fine 2j+1) */
static void prolong_1d(int n, int nc, int64_t *ptr, int64_t *idx, double *val)
{
  int64_t k = 0;
  for (int i = 0; i < n; ++i) {
    ptr[i] = k;
    if (i % 2 == 1) { idx[k] = (i - 1) / 2; val[k++] = 1.0; }
    else {
      if (i / 2 - 1 >= 0) { idx[k] = i / 2 - 1; val[k++] = 0.5; }
      if (i / 2 < nc) { idx[k] = i / 2; val[k++] = 0.5; }
    }
  }
  ptr[n] = k;
}

int main(void)
{
  gmg_handle_t h = NULL;
  const int nlev = 3;
  int n[3] = {1023, 511, 255};
  CHECK(gmg_create(&h, nlev, 0));
  for (int l = 0; l < nlev; ++l) {
    int64_t *ptr = malloc(sizeof(int64_t) * (n[l] + 1)), *idx = malloc(sizeof(int64_t) * 3 * n[l]);
    double *val = malloc(sizeof(double) * 3 * n[l]);
    laplace_1d(n[l], ptr, idx, val);
    /* the Galerkin coarse operators of this P are (1/2) * [-1 2 -1] per coarsening: scale so that A_H = P^T A_h P */
    for (int64_t k = 0; k < ptr[n[l]]; ++k) val[k] *= pow(0.5, l);
    CHECK(gmg_set_matrix(h, l, n[l], n[l], ptr[n[l]], ptr, idx, val, GMG_CSR, 0, 8));
    free(ptr); free(idx); free(val);
    if (l + 1 < nlev) {
      int64_t *pp = malloc(sizeof(int64_t) * (n[l] + 1)), *pi = malloc(sizeof(int64_t) * 2 * n[l]);
      double *pv = malloc(sizeof(double) * 2 * n[l]);
      prolong_1d(n[l], n[l + 1], pp, pi, pv);
      CHECK(gmg_set_prolongation(h, l, n[l], n[l + 1], pp[n[l]], pp, pi, pv, GMG_CSR, 0, 8));   /* R = P^T is built by the library */
      CHECK(gmg_set_smoother_jacobi(h, l, GMG_PRE_AND_POST, 5, 2.0 / 3.0));
      free(pp); free(pi); free(pv);
    }
  }
  CHECK(gmg_set_options(h, GMG_MODE_PRECONDITIONER, GMG_V_CYCLE, 1, 1e-14, 1e-8));
  CHECK(gmg_setup(h));

  const int N = n[0];
  double *b = malloc(sizeof(double) * N), *x = calloc(N, sizeof(double)), *xe = malloc(sizeof(double) * N);
  for (int i = 0; i < N; ++i) b[i] = sin(0.01 * i) + 0.5;
  gmg_result res;
  double hist[64];
  CHECK(gmg_cg_solve(h, b, x, GMG_MEM_HOST, 50, 1e-14, 1e-10, 0, 1, &res, hist, 64));
  /* exact solution of tridiag(-1,2,-1) x = b by the Thomas algorithm */
  double *c = malloc(sizeof(double) * N), *d = malloc(sizeof(double) * N);
  c[0] = -1.0 / 2.0; d[0] = b[0] / 2.0;
  for (int i = 1; i < N; ++i) { const double m = 2.0 + c[i - 1]; c[i] = -1.0 / m; d[i] = (b[i] + d[i - 1]) / m; }
  xe[N - 1] = d[N - 1];
  for (int i = N - 2; i >= 0; --i) xe[i] = d[i] - c[i] * xe[i + 1];
  double err = 0.0, nrm = 0.0;
  for (int i = 0; i < N; ++i) { err += (x[i] - xe[i]) * (x[i] - xe[i]); nrm += xe[i] * xe[i]; }
  printf("CG+GMG: %d iterations, flag %d, residual %.3e -> %.3e, rel. error vs exact %.3e\n", res.niters, res.flag, res.res0, res.res,
         sqrt(err / nrm));
  const int ok = res.flag <= GMG_CONVERGED_RTOL && res.niters < 50 && sqrt(err / nrm) < 1e-8;
  gmg_destroy(h);
  free(b); free(x); free(xe); free(c); free(d);
  puts(ok ? "OK" : "FAILED");
  return ok ? 0 : 1;
}
