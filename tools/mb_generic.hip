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

// ---- ablations of the generic sweep (wrong results by design) ----
// MODE 0: stream only (col,val in the SELL layout, 4 B + 8 B per lane) ; 1: + gathers that always hit (col & 1023)
// 2: + row-wise operand loads and the three stores ; 3: loads only ; 4: stores only ; 5: as 2, non-temporal stores ;
// 6: as 2, non-temporal loads and stores ; 7: as 2 but one store (r) ; 8: as 2, stores of a slice issued by ... (unused)
template <int MODE, int UN>
__global__ __launch_bounds__(256) void abl_kernel(SellArgs a)
{
  const int lane = threadIdx.x & 63;
  const int slice = __builtin_amdgcn_readfirstlane((int)(remap_block(blockIdx.x, gridDim.x, a.xcd_remap) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  if (slice >= a.nslices) return;
  const int64_t base = a.soff[slice];
  const int w = (int)((a.soff[slice + 1] - base) >> 6);
  const int64_t row = min((int64_t)slice * 64 + lane, a.nrows - 1);
  const int32_t *cp = a.scol + base + lane;
  const double *vp = a.sval + base + lane;
  constexpr bool LD = MODE == 2 || MODE == 3 || MODE == 5 || MODE == 6 || MODE == 7;
  constexpr bool ST = MODE == 2 || MODE == 4 || MODE == 5 || MODE == 6 || MODE == 7;
  constexpr bool NTL = MODE == 6, NTS = MODE == 5 || MODE == 6;
  double e0 = 0.0, e1 = 0.0, e2 = 0.0, dr = 0.0;
  if (LD) {
    if (NTL) { e0 = __builtin_nontemporal_load(a.b + row); e1 = a.x[row]; e2 = __builtin_nontemporal_load(a.x2 + row); dr = __builtin_nontemporal_load(a.dinv + row); }
    else { e0 = a.b[row]; e1 = a.x[row]; e2 = a.x2[row]; dr = a.dinv[row]; }
  }
  double s = 0.0;
  for (int j = 0; j + UN <= w; j += UN) {
    int32_t c[UN]; double v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) { c[u] = __builtin_nontemporal_load(cp + (int64_t)(j + u) * 64); v[u] = __builtin_nontemporal_load(vp + (int64_t)(j + u) * 64); }
#pragma unroll
    for (int u = 0; u < UN; ++u) s += v[u] * (MODE >= 1 ? a.x[c[u] & 1023] : (double)c[u]);
  }
  if (ST) {
    const double rn = e0 - s;
    if (NTS) { __builtin_nontemporal_store(e2 + e1, a.x2 + row); __builtin_nontemporal_store(rn, a.y + row); a.s_out[row] = a.omega * (dr * rn); }
    else if (MODE == 7) a.y[row] = rn + e2 + e1 + dr;
    else { a.x2[row] = e2 + e1; a.y[row] = rn; a.s_out[row] = a.omega * (dr * rn); }
  } else if (s + e0 + e1 + e2 + dr == 1.2345e300) a.y[row] = s;
}
// ---- candidate: NB consecutive slices per wave, all stores after the last slice's loads (bit-identical to sell_kernel) ----
template <int NB, int UN>
__global__ __launch_bounds__(256) void sell_multi_kernel(SellArgs a)
{
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(remap_block(blockIdx.x, gridDim.x, a.xcd_remap) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
  const int slice0 = wv * NB;
  if (slice0 >= a.nslices) return;
  const double *__restrict__ xg = a.x;
  const double omega = a.omega;
  double xn[NB], rn[NB], sn[NB];
  int64_t rows[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int slice = min(slice0 + i, a.nslices - 1);
    const int64_t base = a.soff[slice];
    const int w = (int)((a.soff[slice + 1] - base) >> 6);
    const int64_t row = (int64_t)slice * 64 + lane;
    const int64_t rc = min(row, a.nrows - 1);
    rows[i] = (slice0 + i < a.nslices && row < a.nrows) ? row : -1;
    int len = a.rowlen[rc];
    if (rows[i] < 0) len = 0;
    const double e0 = a.b[rc], e1 = xg[rc], dinv_row = a.dinv[rc];
    double e2 = a.x2[rc];
    if (a.x_zero) e2 = 0.0;
    const int32_t *cp = a.scol + base + lane;
    const double *vp = a.sval + base + lane;
    double s = 0.0;
    int j = 0;
    for (; j + UN <= w; j += UN) {
      int32_t c[UN]; double v[UN], g[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) { c[u] = __builtin_nontemporal_load(cp + (int64_t)(j + u) * 64); v[u] = __builtin_nontemporal_load(vp + (int64_t)(j + u) * 64); }
#pragma unroll
      for (int u = 0; u < UN; ++u) g[u] = xg[c[u]];
#pragma unroll
      for (int u = 0; u < UN; ++u) { const double pr = v[u] * g[u]; s = (j + u < len) ? s + pr : s; }
    }
    for (; j < w; ++j) {
      const int32_t c = cp[(int64_t)j * 64]; const double v = vp[(int64_t)j * 64];
      const double pr = v * xg[c]; s = (j < len) ? s + pr : s;
    }
    xn[i] = e2 + e1;
    rn[i] = e0 - s;
    sn[i] = omega * (dinv_row * rn[i]);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i)
    if (rows[i] >= 0) { a.x2[rows[i]] = xn[i]; a.y[rows[i]] = rn[i]; a.s_out[rows[i]] = sn[i]; }
}

// flat 16 B/lane read of nbytes (the same arrays as one stream)
__global__ __launch_bounds__(256) void flat_read_kernel(int64_t n2, const double *__restrict__ src, double *__restrict__ sink)
{
  typedef double d2 __attribute__((ext_vector_type(2)));
  const d2 *s2 = reinterpret_cast<const d2 *>(src);
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (int64_t)gridDim.x * blockDim.x) {
    const d2 v = __builtin_nontemporal_load(s2 + i);
    acc += v.x + v.y;
  }
  if (acc == 1.2345e300) sink[0] = acc;
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
  RUN("sello_pipe<row 27> cached stream", B8, 4, 1, hipLaunchKernelGGL((sello_pipe_kernel<EPI_SWEEP, true, 9, 27, false>), g, b, lds, 0, o))
  // ---- ceilings and ablations over the same footprint ----
  have_ref = false;
  const double Bs = 12.0 * 27.0 * 64.0 * nsl;
  double *cpy; CK(hipMalloc(&cpy, zpad * 4));
  { float t = time_it([&] { hipLaunchKernelGGL(flat_read_kernel, dim3(256 * 16), dim3(256), 0, 0, zpad / 2, sval, ref); }, reps);
    printf("  flat 16 B/lane read of sval (%.0f MB), 4096 wgs              : %8.1f us  %6.0f GB/s\n", 8.0 * zpad / 1e6, t * 1e3, 8.0 * zpad / t / 1e6); }
  { float t = time_it([&] { hipLaunchKernelGGL(flat_read_kernel, dim3(256 * 64), dim3(256), 0, 0, zpad / 2, sval, ref); }, reps);
    printf("  flat 16 B/lane read of sval (%.0f MB), 16384 wgs             : %8.1f us  %6.0f GB/s\n", 8.0 * zpad / 1e6, t * 1e3, 8.0 * zpad / t / 1e6); }
  { float t = time_it([&] { hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 16), dim3(256), 0, 0, zpad / 4, sval, cpy); }, reps);
    printf("  16 B/lane copy of half of sval (%.0f MB read + written)      : %8.1f us  %6.0f GB/s\n", 8.0 * zpad / 1e6, t * 1e3, 8.0 * zpad / t / 1e6); }
  RUN("ablation: stream only (SELL layout, UN=6)", Bs, 4, 1, hipLaunchKernelGGL((abl_kernel<0, 6>), g, b, 0, 0, a))
  RUN("ablation: stream only (SELL layout, UN=6)", Bs, 4, 0, hipLaunchKernelGGL((abl_kernel<0, 6>), g, b, 0, 0, a))
  RUN("ablation: stream only (SELL layout, UN=9)", Bs, 4, 0, hipLaunchKernelGGL((abl_kernel<0, 9>), g, b, 0, 0, a))
  RUN("ablation: stream + always-hit gathers", Bs, 4, 1, hipLaunchKernelGGL((abl_kernel<1, 6>), g, b, 0, 0, a))
  RUN("ablation: stream + always-hit gathers", Bs, 4, 0, hipLaunchKernelGGL((abl_kernel<1, 6>), g, b, 0, 0, a))
  RUN("ablation 2: stream + hit gathers + row loads + 3 stores", B12, 4, 0, hipLaunchKernelGGL((abl_kernel<2, 6>), g, b, 0, 0, a))
  RUN("ablation 3: ... row loads only", B12, 4, 0, hipLaunchKernelGGL((abl_kernel<3, 6>), g, b, 0, 0, a))
  RUN("ablation 4: ... stores only", B12, 4, 0, hipLaunchKernelGGL((abl_kernel<4, 6>), g, b, 0, 0, a))
  RUN("ablation 5: ... loads + nt stores (x, r)", B12, 4, 0, hipLaunchKernelGGL((abl_kernel<5, 6>), g, b, 0, 0, a))
  RUN("ablation 6: ... nt loads + nt stores", B12, 4, 0, hipLaunchKernelGGL((abl_kernel<6, 6>), g, b, 0, 0, a))
  RUN("ablation 7: ... loads + ONE store", B12, 4, 0, hipLaunchKernelGGL((abl_kernel<7, 6>), g, b, 0, 0, a))
  have_ref = true;
#define RUNM(NBV, UNV, WPB, REMAP)                                                                                     \
  { a.xcd_remap = REMAP; const int64_t nw = (nsl + (NBV) - 1) / (NBV); const dim3 g((unsigned)((nw + (WPB) - 1) / (WPB))), b(64 * (WPB)); \
    CK(hipMemset(r2, 0, nrows * 8));                                                                                   \
    float t = time_it([&] { hipLaunchKernelGGL((sell_multi_kernel<NBV, UNV>), g, b, 0, 0, a); }, reps); CK(hipGetLastError()); \
    printf("  sell_multi<NB=%d,UN=%d>                                     wpb %d remap %d : %8.1f us  %6.0f GB/s  frac %.3f\n", NBV, UNV, WPB, REMAP, t * 1e3, B12 / t / 1e6, B12 / t / 1e6 / 8000.0); \
    check("sell_multi"); }
  RUNM(1, 6, 4, 0) RUNM(2, 6, 4, 0) RUNM(4, 6, 4, 0) RUNM(8, 6, 4, 0) RUNM(16, 6, 4, 0)
  RUNM(2, 9, 4, 0) RUNM(4, 9, 4, 0) RUNM(4, 6, 4, 1) RUNM(4, 6, 2, 0) RUNM(4, 6, 1, 0) RUNM(8, 6, 1, 0)
  have_ref = false;   // (two-gather form: r ping-pong, different operands -- timing only)
  RUN("sell_kernel<ONEG=false (gather r and 1/diag), UN=6, NT=1>", B12, 4, 0, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, false, 6, 1>), g, b, 0, 0, a))
  RUN("sell_kernel<ONEG=false (gather r and 1/diag), UN=6, NT=2>", B12, 4, 0, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, false, 6, 2>), g, b, 0, 0, a))
  have_ref = true;
  RUN("sell_kernel<UN=6,NT=2,XM=1> (x untouched)", B12, 4, 0, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, 2, 1>), g, b, 0, 0, a))
  RUN("sell_kernel<UN=6,NT=1>", B12, 4, 0, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, 1>), g, b, 0, 0, a))
  RUN("sell_kernel<UN=6,NT=2> (nt row operands + stores)", B12, 4, 0, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, 2>), g, b, 0, 0, a))
  RUN("sell_kernel<UN=6,NT=2> (nt row operands + stores)", B12, 4, 1, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, 2>), g, b, 0, 0, a))
  // ---- XCD chunk size (blocks of 256 rows per XCD inside a window of 8 chunks) ----
  have_ref = true;
  for (int C : {0, 256, 1}) {
    char nm[96]; snprintf(nm, sizeof nm, "sell_kernel<UN=6>, XCD chunk %d blocks", C);
    RUN(nm, B12, 4, C, hipLaunchKernelGGL((sell_kernel<EPI_SWEEP, true, 6, true>), g, b, 0, 0, a))
  }
  for (int C : {0, 1}) {
    char nm[96]; snprintf(nm, sizeof nm, "sell_pipe<UN=9,PD=1>, XCD chunk %d blocks", C);
    RUN(nm, B12, 4, C, hipLaunchKernelGGL((sell_pipe_kernel<EPI_SWEEP, true, 9, 1, true>), g, b, 0, 0, a))
  }
  for (int C : {0, 1}) {
    char nm[96]; snprintf(nm, sizeof nm, "sello_kernel<UN=9>, XCD chunk %d blocks", C);
    RUN(nm, B8, 4, C, hipLaunchKernelGGL((sello_kernel<EPI_SWEEP, true, 9, true>), g, b, lds, 0, o))
  }
  return 0;
}
