// tools/mb_generic.hip -- the generic (12 B/nnz SELL-64) and SELL-O (8 B/nnz) fused sweeps on a synthetic 27-point
// variable-coefficient operator, product kernels against the stream-in-flight variants (sell_pipe_kernel, sell_row_kernel,
// sello_pipe_kernel).  Diagnostic only (not part of the product): builds the SELL arrays on the device by formula, so 256^3
// (6.5 GB) needs no host CSR.  Every variant's output is compared bit for bit with the product kernel's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/mbg tools/mb_generic.hip && /tmp/mbg [n=255] [reps=20]
#include "../gridapsolvers.jl_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace gmg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// row i of the 27-point operator on n^3 free nodes (Dirichlet rows eliminated): entries in ascending column order
__global__ void build_kernel(int n, int64_t nrows, int32_t *scol, double *sval, int32_t *rowlen, uint16_t *rowpid, double *dinv)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t slice = i >> 6;
  const int lane = (int)(i & 63);
  const int64_t base = slice * 27 * 64 + lane;
  if (i >= nrows) {
    if (i < (slice + 1) * 64)
      for (int j = 0; j < 27; ++j) { scol[base + (int64_t)j * 64] = 0; sval[base + (int64_t)j * 64] = 0.0; }
    return;
  }
  const int x = (int)(i % n), y = (int)((i / n) % n), z = (int)(i / ((int64_t)n * n));
  int j = 0;
  double diag = 1.0;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = x + dx, yy = y + dy, zz = z + dz;
        if (xx < 0 || xx >= n || yy < 0 || yy >= n || zz < 0 || zz >= n) continue;
        const int64_t c = i + dx + (int64_t)dy * n + (int64_t)dz * n * n;
        const int nz = (dx != 0) + (dy != 0) + (dz != 0);
        const double kap = 1.0 + 0.25 * (double)((i * 2654435761u + c * 40503u) & 1023) / 1024.0;   // every value distinct
        const double v = (nz == 0 ? 8.0 / 3.0 : nz == 1 ? 0.0 - 1e-9 : nz == 2 ? -1.0 / 6.0 : -1.0 / 12.0) * kap;
        if (nz == 0) diag = v;
        scol[base + (int64_t)j * 64] = (int32_t)c;
        sval[base + (int64_t)j * 64] = v;
        ++j;
      }
  rowlen[i] = j;
  for (int q = j; q < 27; ++q) { scol[base + (int64_t)q * 64] = (int32_t)i; sval[base + (int64_t)q * 64] = 0.0; }
  const int bx = x == 0 ? 0 : x == n - 1 ? 2 : 1, by = y == 0 ? 0 : y == n - 1 ? 2 : 1, bz = z == 0 ? 0 : z == n - 1 ? 2 : 1;
  rowpid[i] = (uint16_t)(bx + 3 * by + 9 * bz);
  dinv[i] = 1.0 / diag;
}
__global__ void fill_kernel(int64_t n, double *r, double *s, double *x)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    r[i] = 1.0 + 1e-3 * (double)(i % 977); s[i] = 0.5 - 1e-3 * (double)(i % 613); x[i] = 0.0;
  }
}
__global__ void cmp_kernel(int64_t n, const double *a, const double *b, unsigned long long *ndiff)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    if (__double_as_longlong(a[i]) != __double_as_longlong(b[i])) atomicAdd(ndiff, 1ull);
}

template <typename F> float time_it(F f, int reps)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms / reps;
}

int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 255;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int64_t nrows = (int64_t)n * n * n;
  const int64_t nsl = (nrows + 63) / 64;
  const int64_t zpad = nsl * 27 * 64;
  int32_t *scol, *rowlen; double *sval, *dinv, *r, *s, *x, *r2, *s2, *ref; uint16_t *rowpid; int64_t *soff; int32_t *poff;
  unsigned long long *ndiff;
  CK(hipMalloc(&scol, zpad * 4)); CK(hipMalloc(&sval, zpad * 8)); CK(hipMalloc(&rowlen, nrows * 4)); CK(hipMalloc(&rowpid, nrows * 2));
  CK(hipMalloc(&dinv, nrows * 8)); CK(hipMalloc(&r, nrows * 8)); CK(hipMalloc(&s, nrows * 8)); CK(hipMalloc(&x, nrows * 8));
  CK(hipMalloc(&r2, nrows * 8)); CK(hipMalloc(&s2, nrows * 8)); CK(hipMalloc(&ref, nrows * 8)); CK(hipMalloc(&ndiff, 8));
  CK(hipMalloc(&soff, (nsl + 1) * 8));
  { std::vector<int64_t> h(nsl + 1); for (int64_t q = 0; q <= nsl; ++q) h[q] = q * 27 * 64; CK(hipMemcpy(soff, h.data(), (nsl + 1) * 8, hipMemcpyHostToDevice)); }
  // SELL-O offset table: 27 boundary types x 27 byte offsets (compacted, zero padded), + the empty last pattern
  { std::vector<int32_t> h(28 * 27, 0);
    for (int p = 0; p < 27; ++p) {
      const int bx = p % 3, by = (p / 3) % 3, bz = p / 9; int j = 0;
      for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
        if ((bx == 0 && dx < 0) || (bx == 2 && dx > 0) || (by == 0 && dy < 0) || (by == 2 && dy > 0) || (bz == 0 && dz < 0) || (bz == 2 && dz > 0)) continue;
        h[p * 27 + j++] = 8 * (dx + dy * n + dz * n * n);
      }
    }
    CK(hipMalloc(&poff, h.size() * 4)); CK(hipMemcpy(poff, h.data(), h.size() * 4, hipMemcpyHostToDevice)); }
  hipLaunchKernelGGL(build_kernel, dim3((unsigned)((nsl * 64 + 255) / 256)), dim3(256), 0, 0, n, nrows, scol, sval, rowlen, rowpid, dinv);
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, nrows, r, s, x);
  CK(hipDeviceSynchronize());
  const double B12 = 12.0 * 27.0 * nrows + 68.0 * nrows;   // SURVEY 8(d) with Z = 27 N (the padded boundary rows are < 1 %)
  const double B8 = 8.0 * 27.0 * nrows + 2.0 * nrows + 64.0 * nrows;
  printf("n = %d^3 = %lld rows, %lld slices; B_sweep(12 B/nnz) = %.1f MB, SELL-O layout bytes = %.1f MB\n", n, (long long)nrows, (long long)nsl, B12 / 1e6, B8 / 1e6);

  SellArgs a; memset(&a, 0, sizeof(a));
  a.soff = soff; a.scol = scol; a.sval = sval; a.rowlen = rowlen; a.nrows = nrows; a.nslices = (int)nsl; a.xcd_remap = 1;
  a.x = s; a.dinv = dinv; a.omega = 2.0 / 3.0; a.y = r2; a.b = r; a.x2 = x; a.s_out = s2; a.x_zero = 0;
  SellOArgs o; memset(&o, 0, sizeof(o));
  o.soff = soff; o.sval = sval; o.rowlen = rowlen; o.rowpid = rowpid; o.rowbase = nullptr; o.poff = poff; o.np = 28; o.W = 27;
  o.nrows = nrows; o.nslices = (int)nsl; o.xcd_remap = 1; o.x = s; o.dinv = dinv; o.omega = 2.0 / 3.0; o.y = r2; o.b = r; o.x2 = x; o.s_out = s2;
  const size_t lds = 28 * 27 * 4;

  auto check = [&](const char *name) {
    CK(hipMemset(ndiff, 0, 8));
    hipLaunchKernelGGL(cmp_kernel, dim3(2048), dim3(256), 0, 0, nrows, r2, ref, ndiff);
    unsigned long long h = 0; CK(hipMemcpy(&h, ndiff, 8, hipMemcpyDeviceToHost));
    if (h) printf("    !! %s: %llu rows differ from the product kernel\n", name, h);
  };
#define RUN(NAME, BYTES, WPB, REMAP, ...)                                                                            \
  { a.xcd_remap = REMAP; o.xcd_remap = REMAP; const dim3 g((unsigned)((nsl + (WPB) - 1) / (WPB))), b(64 * (WPB));    \
    CK(hipMemset(r2, 0, nrows * 8));                                                                                 \
    float t = time_it([&] { __VA_ARGS__; }, reps); CK(hipGetLastError());                                            \
    printf("  %-58s wpb %d remap %d : %8.1f us  %6.0f GB/s  frac %.3f\n", NAME, WPB, REMAP, t * 1e3, (BYTES) / t / 1e6, (BYTES) / t / 1e6 / 8000.0); \
    if (have_ref) check(NAME); }
  bool have_ref = false;
  // ---- generic 12 B/nnz ----
  RUN("sell_kernel<UN=6> (product)", B12, 4, 1, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, true>), g, b, 0, 0, a))
  CK(hipMemcpy(ref, r2, nrows * 8, hipMemcpyDeviceToDevice)); have_ref = true;
  RUN("sell_kernel<UN=6>", B12, 4, 0, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, true>), g, b, 0, 0, a))
  RUN("sell_kernel<UN=9>", B12, 4, 1, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 9, true>), g, b, 0, 0, a))
  RUN("sell_kernel<UN=27>", B12, 4, 1, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 27, true>), g, b, 0, 0, a))
  RUN("sell_kernel<UN=6> cached stream", B12, 4, 1, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, false>), g, b, 0, 0, a))
  RUN("sell_pipe<UN=3,PD=1>", B12, 4, 1, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 3, 1, true>), g, b, 0, 0, a))
  RUN("sell_pipe<UN=3,PD=2>", B12, 4, 1, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 3, 2, true>), g, b, 0, 0, a))
  RUN("sell_pipe<UN=6,PD=1>", B12, 4, 1, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 6, 1, true>), g, b, 0, 0, a))
  RUN("sell_pipe<UN=6,PD=2>", B12, 4, 1, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 6, 2, true>), g, b, 0, 0, a))
  RUN("sell_pipe<UN=9,PD=1>", B12, 4, 1, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 9, 1, true>), g, b, 0, 0, a))
  RUN("sell_pipe<UN=9,PD=2>", B12, 4, 1, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 9, 2, true>), g, b, 0, 0, a))
  RUN("sell_pipe<UN=9,PD=1>", B12, 4, 0, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 9, 1, true>), g, b, 0, 0, a))
  RUN("sell_pipe<UN=9,PD=1>", B12, 2, 1, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 9, 1, true>), g, b, 0, 0, a))
  RUN("sell_pipe<UN=9,PD=1>", B12, 1, 1, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 9, 1, true>), g, b, 0, 0, a))
  RUN("sell_pipe<UN=9,PD=1> cached stream", B12, 4, 1, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 9, 1, false>), g, b, 0, 0, a))
  RUN("sell_row<27>", B12, 4, 1, hipLaunchKernelGGL((sell_row_kernel<EPI_SWEEP, true, 27, true>), g, b, 0, 0, a))
  RUN("sell_row<27>", B12, 4, 0, hipLaunchKernelGGL((sell_row_kernel<EPI_SWEEP, true, 27, true>), g, b, 0, 0, a))
  RUN("sell_row<27>", B12, 2, 1, hipLaunchKernelGGL((sell_row_kernel<EPI_SWEEP, true, 27, true>), g, b, 0, 0, a))
  RUN("sell_row<27>", B12, 1, 1, hipLaunchKernelGGL((sell_row_kernel<EPI_SWEEP, true, 27, true>), g, b, 0, 0, a))
  RUN("sell_row<27> cached stream", B12, 4, 1, hipLaunchKernelGGL((sell_row_kernel<EPI_SWEEP, true, 27, false>), g, b, 0, 0, a))
  // ---- SELL-O 8 B/nnz ----
  RUN("sello_kernel<UN=3> (product)", B8, 4, 1, hipLaunchKernelGGL((sello_kernel<EPI_SWEEP, true, 3, true>), g, b, lds, 0, o))
  RUN("sello_kernel<UN=9>", B8, 4, 1, hipLaunchKernelGGL((sello_kernel<EPI_SWEEP, true, 9, true>), g, b, lds, 0, o))
  RUN("sello_pipe<UN=3>", B8, 4, 1, hipLaunchKernelGGL((sello_pipe_kernel<EPI_SWEEP, true, 3, 0, true>), g, b, lds, 0, o))
  RUN("sello_pipe<UN=9>", B8, 4, 1, hipLaunchKernelGGL((sello_pipe_kernel<EPI_SWEEP, true, 9, 0, true>), g, b, lds, 0, o))
  RUN("sello_pipe<UN=9>", B8, 4, 0, hipLaunchKernelGGL((sello_pipe_kernel<EPI_SWEEP, true, 9, 0, true>), g, b, lds, 0, o))
  RUN("sello_pipe<row 27>", B8, 4, 1, hipLaunchKernelGGL((sello_pipe_kernel<EPI_SWEEP, true, 9, 27, true>), g, b, lds, 0, o))
  RUN("sello_pipe<row 27>", B8, 2, 1, hipLaunchKernelGGL((sello_pipe_kernel<EPI_SWEEP, true, 9, 27, true>), g, b, lds, 0, o))
  RUN("sello_pipe<row 27>", B8, 8, 1, hipLaunchKernelGGL((sello_pipe_kernel<EPI_SWEEP, true, 9, 27, true>), g, b, lds, 0, o))
  RUN("sello_pipe<row 27> cached stream", B8, 4, 1, hipLaunchKernelGGL((sello_pipe_kernel<EPI_SWEEP, true, 9, 27, false>), g, b, lds, 0, o))
  return 0;
}
