// tools/microbench.hip -- ablation microbenchmarks for the CSR-stream sweep kernel.
// Diagnostic only (not part of the product).  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mb tools/microbench.hip && /tmp/mb [ncells]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int kBlock = 256;
constexpr int TILE = 2048;
constexpr int U = TILE / kBlock;

__global__ void copy_kernel(const double4 *__restrict__ a, double4 *__restrict__ b, size_t n4)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void read_kernel(const double4 *__restrict__ a, double *__restrict__ out, size_t n4)
{
  double s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 v = __builtin_nontemporal_load(reinterpret_cast<const d4 *>(a) + i);
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 12345.678) out[0] = s;
}

// MODE bit0: gather x[col]; bit1: LDS products + row reduce; bit2: non-temporal loads; bit3: second gather
template <int MODE>
__global__ __launch_bounds__(kBlock) void stream_kernel(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                        const double *__restrict__ val, const int *__restrict__ blk_row,
                                                        const double *__restrict__ x, const double *__restrict__ x2,
                                                        double *__restrict__ y, int lg)
{
  __shared__ double prod[TILE];
  const int tid = threadIdx.x;
  const int blk = blockIdx.x;
  const int r0 = blk_row[blk], r1 = blk_row[blk + 1];
  const int nz0 = rowptr[r0], cnt = rowptr[r1] - nz0;
  const int G = 1 << lg, sub = tid & (G - 1), row = r0 + (tid >> lg);
  int k0 = 0, k1 = 0;
  if (row < r1) { k0 = rowptr[row] - nz0; k1 = rowptr[row + 1] - nz0; }
  int c[U]; double v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = tid + u * kBlock;
    const bool ok = k < cnt;
    if (MODE & 4) { c[u] = ok ? __builtin_nontemporal_load(col + nz0 + k) : -1; v[u] = ok ? __builtin_nontemporal_load(val + nz0 + k) : 0.0; }
    else { c[u] = ok ? col[nz0 + k] : -1; v[u] = ok ? val[nz0 + k] : 0.0; }
  }
  double g[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int cc = c[u] >= 0 ? c[u] : 0;
    g[u] = (MODE & 1) ? x[cc] : (double)cc;
    if (MODE & 8) g[u] *= x2[cc];
  }
  if (MODE & 2) {
#pragma unroll
    for (int u = 0; u < U; ++u) { const int k = tid + u * kBlock; if (c[u] >= 0) prod[k] = v[u] * g[u]; }
    __syncthreads();
    double s = 0.0;
    for (int k = k0 + sub; k < k1; k += G) s += prod[k];
    for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (row < r1 && sub == 0) y[row] = s;
  } else {
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u] * g[u];
    if (s == 12345.678) y[blk] = s;
  }
}


// ---- LDS-window variant: 16-bit tile-local column indices + gathered vector staged in LDS ----
// per tile: segments (start,len) of the distinct columns; window = concatenation of the segments.
template <int ALIAS>
__global__ __launch_bounds__(kBlock) void stream_win_kernel(const int *__restrict__ rowptr, const unsigned short *__restrict__ idx16,
                                                            const double *__restrict__ val, const int *__restrict__ blk_row,
                                                            const int *__restrict__ seg_ptr, const int *__restrict__ seg_start,
                                                            const int *__restrict__ seg_woff,
                                                            const double *__restrict__ x, double *__restrict__ y, int lg)
{
  __shared__ double prod[TILE];
  __shared__ double win[ALIAS ? 1 : 1536];
  double *w = ALIAS ? prod : win;
  const int tid = threadIdx.x;
  const int blk = blockIdx.x;
  const int r0 = blk_row[blk], r1 = blk_row[blk + 1];
  const int nz0 = rowptr[r0], cnt = rowptr[r1] - nz0;
  const int G = 1 << lg, sub = tid & (G - 1), row = r0 + (tid >> lg);
  int k0 = 0, k1 = 0;
  if (row < r1) { k0 = rowptr[row] - nz0; k1 = rowptr[row + 1] - nz0; }
  // stream loads
  unsigned short c[U]; double v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = tid + u * kBlock;
    const bool ok = k < cnt;
    c[u] = ok ? __builtin_nontemporal_load(idx16 + nz0 + k) : (unsigned short)0;
    v[u] = ok ? __builtin_nontemporal_load(val + nz0 + k) : 0.0;
  }
  // window fill: wave j takes segments j, j+4, ...
  const int s0 = seg_ptr[blk], s1 = seg_ptr[blk + 1] - 1; // last entry = sentinel (total window)
  const int wave = tid >> 6, lane = tid & 63;
  for (int sgi = s0 + wave; sgi < s1; sgi += 4) {
    const int st = seg_start[sgi], wo = seg_woff[sgi], len = seg_woff[sgi + 1] - wo;
    for (int i = lane; i < len; i += 64) w[wo + i] = x[st + i];
  }
  __syncthreads();
  double g[U];
#pragma unroll
  for (int u = 0; u < U; ++u) g[u] = w[c[u]];
  if (ALIAS) __syncthreads();
#pragma unroll
  for (int u = 0; u < U; ++u) { const int k = tid + u * kBlock; if (k < cnt) prod[k] = v[u] * g[u]; }
  __syncthreads();
  double s = 0.0;
  for (int k = k0 + sub; k < k1; k += G) s += prod[k];
  for (int off = G >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (row < r1 && sub == 0) y[row] = s;
}

// SELL-64 (sliced ELLPACK, slice = one wave of 64 rows, column-major inside the slice):
// lane == row, every load coalesced, no LDS, sequential left-to-right row sums.
// MODE bit0: gather x[col]; bit3: second gather; bit2: nt loads; bit4: fused sweep epilogue (reads s,r,x,dinv; writes x,r,s)
template <int MODE, int UN>
__global__ __launch_bounds__(kBlock) void sell_kernel(int nslices, const long *__restrict__ soff, const int *__restrict__ rowlen,
                                                      const int *__restrict__ col, const double *__restrict__ val,
                                                      const double *__restrict__ x, const double *__restrict__ x2,
                                                      double *__restrict__ y, double *__restrict__ y2, double *__restrict__ y3, long N)
{
  const int lane = threadIdx.x & 63;
  const int slice = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (slice >= nslices) return;
  const long base = soff[slice];
  const int w = (int)((soff[slice + 1] - base) >> 6);
  const long row = (long)slice * 64 + lane;
  const int len = row < N ? rowlen[row] : 0;
  double e_s = 0, e_r = 0, e_x = 0, e_d = 0;
  if ((MODE & 16) && row < N) { e_s = x[row]; e_r = x2[row]; e_x = y2[row]; e_d = y3[row]; }
  const int *cp = col + base + lane;
  const double *vp = val + base + lane;
  double s = 0.0;
  int j = 0;
  for (; j + UN <= w; j += UN) {
    int c[UN]; double v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (MODE & 4) { c[u] = __builtin_nontemporal_load(cp + (j + u) * 64); v[u] = __builtin_nontemporal_load(vp + (j + u) * 64); }
      else { c[u] = cp[(j + u) * 64]; v[u] = vp[(j + u) * 64]; }
    }
    double g[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {   // unconditional gathers (padded entries point at a valid column)
      g[u] = (MODE & 1) ? x[c[u]] : (double)c[u];
      if (MODE & 8) g[u] *= x2[c[u]];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) { const double pr = v[u] * g[u]; s = (j + u < len) ? s + pr : s; }
  }
  for (; j < w; ++j) {
    const int c = cp[j * 64]; const double v = vp[j * 64];
    if (j < len) { double g = (MODE & 1) ? x[c] : (double)c; if (MODE & 8) g *= x2[c]; s += v * g; }
  }
  if (row < N) {
    if (MODE & 16) { y2[row] = e_x + e_s; const double rn = e_r - s; y[row] = rn; y3[row] = 0.66 * (e_d * rn); }
    else y[row] = s;
  }
}

template <typename F>
float time_it(F f, int reps = 20)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv)
{
  const int nc = argc > 1 ? atoi(argv[1]) : 128;
  const int lg = argc > 2 ? atoi(argv[2]) : 2;
  const int m = nc - 1;
  const int64_t N = (int64_t)m * m * m;
  std::vector<int> ptr(N + 1, 0), col; std::vector<double> val;
  col.reserve(27 * N); val.reserve(27 * N);
  for (int z = 0; z < m; ++z) for (int y = 0; y < m; ++y) for (int x = 0; x < m; ++x) {
    const int64_t i = x + (int64_t)m * (y + (int64_t)m * z);
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      const int X = x + dx, Y = y + dy, Z = z + dz;
      if (X < 0 || Y < 0 || Z < 0 || X >= m || Y >= m || Z >= m) continue;
      col.push_back(X + m * (Y + m * Z)); val.push_back((dx || dy || dz) ? -1.0 : 26.0);
    }
    ptr[i + 1] = (int)col.size();
  }
  const int64_t Z = col.size();
  std::vector<int> blk; blk.push_back(0);
  const int max_rows = kBlock >> lg;
  for (int64_t r = 0; r < N;) { int64_t e = r; while (e < N && ptr[e + 1] - ptr[r] <= TILE && e - r < max_rows) ++e; blk.push_back((int)e); r = e; }
  const int nb = (int)blk.size() - 1;
  printf("N=%ld Z=%ld tiles=%d lanes=%d\n", (long)N, (long)Z, nb, 1 << lg);
  int *d_ptr, *d_col, *d_blk; double *d_val, *d_x, *d_x2, *d_y, *d_big, *d_big2;
  CK(hipMalloc(&d_ptr, (N + 1) * 4)); CK(hipMalloc(&d_col, Z * 4)); CK(hipMalloc(&d_val, Z * 8)); CK(hipMalloc(&d_blk, (nb + 1) * 4));
  CK(hipMalloc(&d_x, N * 8)); CK(hipMalloc(&d_x2, N * 8)); CK(hipMalloc(&d_y, N * 8));
  const size_t big = (size_t)Z * 12 / 32 * 32;
  CK(hipMalloc(&d_big, big)); CK(hipMalloc(&d_big2, big));
  CK(hipMemcpy(d_ptr, ptr.data(), (N + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_col, col.data(), Z * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_val, val.data(), Z * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_blk, blk.data(), (nb + 1) * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_x, 0, N * 8)); CK(hipMemset(d_x2, 0, N * 8)); CK(hipMemset(d_big, 0, big));
  const double mat_bytes = 12.0 * Z + 4.0 * N;
  float t;
  t = time_it([&] { hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const double4 *)d_big, (double4 *)d_big2, big / 32); });
  printf("copy  %8.1f us  %7.1f GB/s (read+write %.0f MB)\n", t * 1e3, 2.0 * big / t / 1e6, 2.0 * big / 1e6);
  t = time_it([&] { hipLaunchKernelGGL(read_kernel, dim3(2048), dim3(256), 0, 0, (const double4 *)d_big, d_y, big / 32); });
  printf("read  %8.1f us  %7.1f GB/s (read %.0f MB, 32 B/lane nt)\n", t * 1e3, big / t / 1e6, big / 1e6);
#define RUN(MODE, label) \
  t = time_it([&] { hipLaunchKernelGGL((stream_kernel<MODE>), dim3(nb), dim3(kBlock), 0, 0, d_ptr, d_col, d_val, d_blk, d_x, d_x2, d_y, lg); }); \
  printf("%-34s %8.1f us  %7.1f GB/s of matrix bytes\n", label, t * 1e3, mat_bytes / t / 1e6);
  RUN(0, "stream only");
  RUN(4, "stream only, nt");
  RUN(5, "stream nt + 1 gather");
  RUN(13, "stream nt + 2 gathers");
  RUN(6, "stream nt + LDS reduce");
  RUN(7, "stream nt + 1 gather + LDS reduce");
  RUN(15, "stream nt + 2 gathers + LDS reduce");
  RUN(3, "stream + 1 gather + LDS reduce");


  // ---- LDS window ----
  {
    std::vector<unsigned short> idx16(Z);
    std::vector<int> seg_ptr(nb + 1, 0), seg_start, seg_woff;
    int maxW = 0, maxseg = 0; long totW = 0;
    for (int b = 0; b < nb; ++b) {
      const int a0 = ptr[blk[b]], a1 = ptr[blk[b + 1]];
      std::vector<int> u(col.begin() + a0, col.begin() + a1);
      std::sort(u.begin(), u.end()); u.erase(std::unique(u.begin(), u.end()), u.end());
      // segments with gap <= 4
      std::vector<int> st, en;
      for (size_t i = 0; i < u.size(); ++i) {
        if (st.empty() || u[i] - en.back() > 4) { st.push_back(u[i]); en.push_back(u[i] + 1); } else en.back() = u[i] + 1;
      }
      int wo = 0;
      for (size_t k = 0; k < st.size(); ++k) { seg_start.push_back(st[k]); seg_woff.push_back(wo); wo += en[k] - st[k]; }
      seg_ptr[b + 1] = (int)seg_start.size();
      maxW = std::max(maxW, wo); maxseg = std::max<int>(maxseg, st.size()); totW += wo;
      // local indices
      for (int k = a0; k < a1; ++k) {
        const int cc = col[k];
        size_t j = std::upper_bound(st.begin(), st.end(), cc) - st.begin() - 1;
        int base = 0; for (size_t q = 0; q < j; ++q) base += en[q] - st[q];
        idx16[k] = (unsigned short)(base + cc - st[j]);
      }
    }
    seg_woff.push_back(0);
    // make seg_woff[sgi+1]-seg_woff[sgi] valid at tile ends: store lengths via an extra array trick -> rebuild as absolute with sentinel per tile
    // simpler: recompute woff array with per-tile sentinel
    std::vector<int> woff2; std::vector<int> start2; std::vector<int> ptr2(nb + 1, 0);
    for (int b = 0; b < nb; ++b) {
      int wo = 0;
      for (int sgi = seg_ptr[b]; sgi < seg_ptr[b + 1]; ++sgi) {
        start2.push_back(seg_start[sgi]); woff2.push_back(seg_woff[sgi]);
        (void)wo;
      }
      // sentinel: total window of the tile
      int a0 = seg_ptr[b], a1 = seg_ptr[b + 1];
      int tot = 0; { const int q0 = ptr[blk[b]], q1 = ptr[blk[b + 1]]; for (int k = q0; k < q1; ++k) tot = std::max<int>(tot, idx16[k] + 1); }
      start2.push_back(0); woff2.push_back(tot);
      ptr2[b + 1] = (int)start2.size();
      (void)a0; (void)a1;
    }
    printf("LDS window: max W %d doubles, max segments %d, avg W %.1f per tile (x%.2f of tile nnz)\n", maxW, maxseg, (double)totW / nb, (double)totW / Z);
    unsigned short *d_i16; int *d_sp, *d_ss, *d_sw;
    CK(hipMalloc(&d_i16, Z * 2 + 64)); CK(hipMalloc(&d_sp, (nb + 1) * 4)); CK(hipMalloc(&d_ss, start2.size() * 4)); CK(hipMalloc(&d_sw, woff2.size() * 4));
    // seg_ptr for kernel: segments of tile b are ptr2[b] .. ptr2[b+1]-1 (excluding sentinel)
    std::vector<int> kptr(nb + 1); for (int b = 0; b <= nb; ++b) kptr[b] = ptr2[b];
    CK(hipMemcpy(d_i16, idx16.data(), Z * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ss, start2.data(), start2.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sw, woff2.data(), woff2.size() * 4, hipMemcpyHostToDevice));
    // kernel loops sgi in [s0, s1): use s1 = ptr2[b+1]-1 -> pass adjusted array
    std::vector<int> kend(nb + 1); // trick: kernel reads seg_ptr[blk], seg_ptr[blk+1]; we need end excluding sentinel, so build interleaved array
    std::vector<int> sp2(nb + 1);
    for (int b = 0; b < nb; ++b) sp2[b] = ptr2[b];
    sp2[nb] = ptr2[nb];
    CK(hipMemcpy(d_sp, sp2.data(), (nb + 1) * 4, hipMemcpyHostToDevice));
    if (maxW <= 1536) {
      t = time_it([&] { hipLaunchKernelGGL((stream_win_kernel<0>), dim3(nb), dim3(kBlock), 0, 0, d_ptr, d_i16, d_val, d_blk, d_sp, d_ss, d_sw, d_x, d_y, lg); });
      printf("%-44s %8.1f us  %7.1f GB/s of CSR(12B) matrix bytes\n", "win: val+idx16, LDS window, separate prod", t * 1e3, mat_bytes / t / 1e6);
    }
    t = time_it([&] { hipLaunchKernelGGL((stream_win_kernel<1>), dim3(nb), dim3(kBlock), 0, 0, d_ptr, d_i16, d_val, d_blk, d_sp, d_ss, d_sw, d_x, d_y, lg); });
    printf("%-44s %8.1f us  %7.1f GB/s of CSR(12B) matrix bytes\n", "win: val+idx16, LDS window aliased w/ prod", t * 1e3, mat_bytes / t / 1e6);
  }
  // ---- SELL-64 ----
  {
    const int nsl = (int)((N + 63) / 64);
    std::vector<long> soff(nsl + 1, 0); std::vector<int> rowlen(N);
    for (int64_t i = 0; i < N; ++i) rowlen[i] = ptr[i + 1] - ptr[i];
    for (int sidx = 0; sidx < nsl; ++sidx) { int w = 0; for (int64_t i = (int64_t)sidx * 64; i < std::min<int64_t>(N, (int64_t)sidx * 64 + 64); ++i) w = std::max(w, rowlen[i]); soff[sidx + 1] = soff[sidx] + (long)w * 64; }
    const long ZP = soff[nsl];
    std::vector<int> scol(ZP); std::vector<double> sval(ZP, 0.0);
    for (int sidx = 0; sidx < nsl; ++sidx) { const int w = (int)((soff[sidx + 1] - soff[sidx]) / 64);
      for (int l = 0; l < 64; ++l) { const int64_t i = (int64_t)sidx * 64 + l;
        for (int j = 0; j < w; ++j) { const long q = soff[sidx] + (long)j * 64 + l;
          if (i < N && j < rowlen[i]) { scol[q] = col[ptr[i] + j]; sval[q] = val[ptr[i] + j]; } else { scol[q] = i < N ? (int)i : 0; sval[q] = 0.0; } } } }
    printf("SELL-64: padded nnz %ld (%.2f%% padding)\n", ZP, 100.0 * (ZP - Z) / Z);
    long *d_soff; int *d_rl, *d_sc; double *d_sv, *d_y2, *d_y3;
    CK(hipMalloc(&d_soff, (nsl + 1) * 8)); CK(hipMalloc(&d_rl, N * 4)); CK(hipMalloc(&d_sc, ZP * 4)); CK(hipMalloc(&d_sv, ZP * 8)); CK(hipMalloc(&d_y2, N * 8)); CK(hipMalloc(&d_y3, N * 8));
    CK(hipMemcpy(d_soff, soff.data(), (nsl + 1) * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_rl, rowlen.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_sc, scol.data(), ZP * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_sv, sval.data(), ZP * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_y2, 0, N * 8)); CK(hipMemset(d_y3, 0, N * 8));
    const int grid = (nsl + 3) / 4;
#define RUNS(MODE, UN, label) \
    t = time_it([&] { hipLaunchKernelGGL((sell_kernel<MODE, UN>), dim3(grid), dim3(kBlock), 0, 0, nsl, d_soff, d_rl, d_sc, d_sv, d_x, d_x2, d_y, d_y2, d_y3, (long)N); }); \
    printf("%-44s %8.1f us  %7.1f GB/s of CSR matrix bytes\n", label, t * 1e3, mat_bytes / t / 1e6);
    RUNS(4, 9, "sell stream only nt (un9)");
    RUNS(4, 27, "sell stream only nt (un27)");
    RUNS(5, 4, "sell nt + 1 gather (un4)");
    RUNS(5, 9, "sell nt + 1 gather (un9)");
    RUNS(5, 14, "sell nt + 1 gather (un14)");
    RUNS(5, 27, "sell nt + 1 gather (un27)");
    RUNS(13, 9, "sell nt + 2 gathers (un9)");
    RUNS(13, 27, "sell nt + 2 gathers (un27)");
    RUNS(21, 9, "sell nt + 1 gather + sweep epilogue (un9)");
    RUNS(21, 14, "sell nt + 1 gather + sweep epilogue (un14)");
    RUNS(21, 27, "sell nt + 1 gather + sweep epilogue (un27)");
    RUNS(29, 27, "sell nt + 2 gathers + sweep epilogue (un27)");
  }
  return 0;
}
