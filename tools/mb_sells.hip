// tools/mb_sells.hip -- where does the time of the shared-offset row-pattern sweep (sells_kernel<EPI_SWEEP>) go?
// Diagnostic only (not part of the product): runs the PRODUCT kernel on a synthetic 27-point operator of (n-1)^3 rows and its
// DBG ablations (one ingredient dropped each: mask, LDS coefficient reads, DPP shifts, gathers), with the product's launch
// geometry (2048 workgroups x 4 waves, deferred-x pairs).  Prints us per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/mbsells tools/mb_sells.hip && /tmp/mbsells [cells]
#include "../gridapsolvers.jl_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace gmg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename F> float time_it(F f, int reps = 50)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 6; ++i) f();
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv)
{
  const int nc = argc > 1 ? atoi(argv[1]) : 128;
  const int m = nc - 1;
  const int64_t N = (int64_t)m * m * m;
  // patterns: 27 boundary types (lo / interior / hi per axis); entries = the 27 offsets, absent ones masked
  const int nruns = 9, K = 3, nu = 27, np = 28;      // + trailing empty pattern
  std::vector<int32_t> runs;
  for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) runs.push_back(dz * m * m + dy * m - 1);
  std::vector<PatEntry> tab((size_t)np * nu);
  std::memset(tab.data(), 0, tab.size() * sizeof(PatEntry));
  std::vector<double> pdinv(np, 0.0);
  for (int tz = 0; tz < 3; ++tz) for (int ty = 0; ty < 3; ++ty) for (int tx = 0; tx < 3; ++tx) {
    const int p = (tz * 3 + ty) * 3 + tx;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      const bool ok = !((tx == 0 && dx < 0) || (tx == 2 && dx > 0) || (ty == 0 && dy < 0) || (ty == 2 && dy > 0) || (tz == 0 && dz < 0) || (tz == 2 && dz > 0));
      const int e = ((dz + 1) * 3 + (dy + 1)) * 3 + (dx + 1);
      if (ok) { tab[(size_t)p * nu + e].v = (dx || dy || dz) ? -1.0 / 12.0 : 8.0 / 3.0; tab[(size_t)p * nu + e].m = 0xffffffffu; }
    }
    pdinv[p] = 3.0 / 8.0;
  }
  std::vector<uint16_t> rowpid((size_t)N + 64, 27);
  for (int z = 0; z < m; ++z) for (int y = 0; y < m; ++y) for (int x = 0; x < m; ++x) {
    auto t = [&](int c) { return c == 0 ? 0 : (c == m - 1 ? 2 : 1); };
    rowpid[(size_t)x + (size_t)m * (y + (size_t)m * z)] = (uint16_t)((t(z) * 3 + t(y)) * 3 + t(x));
  }
  uint16_t *d_pid; PatEntry *d_tab; int32_t *d_run; double *d_pd, *r, *s0, *s1, *x;
  CK(hipMalloc(&d_pid, rowpid.size() * 2)); CK(hipMalloc(&d_tab, tab.size() * sizeof(PatEntry))); CK(hipMalloc(&d_run, runs.size() * 4));
  CK(hipMalloc(&d_pd, np * 8)); CK(hipMalloc(&r, (N + 64) * 8)); CK(hipMalloc(&s0, (N + 64) * 8)); CK(hipMalloc(&s1, (N + 64) * 8)); CK(hipMalloc(&x, (N + 64) * 8));
  CK(hipMemcpy(d_pid, rowpid.data(), rowpid.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(PatEntry), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_run, runs.data(), runs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pd, pdinv.data(), np * 8, hipMemcpyHostToDevice));
  CK(hipMemset(r, 0, (N + 64) * 8)); CK(hipMemset(s0, 0, (N + 64) * 8)); CK(hipMemset(s1, 0, (N + 64) * 8)); CK(hipMemset(x, 0, (N + 64) * 8));
  SellSArgs a;
  std::memset(&a, 0, sizeof(a));
  a.rowpid = d_pid; a.tab = d_tab; a.run_off = d_run; a.np = np; a.nruns = nruns; a.minoff = runs.front(); a.maxoff = runs.back();
  a.nrows = N; a.ncols = N; a.nslices = (int)((N + 61) / 62); a.xcd_remap = 1; a.pdinv = d_pd; a.omega = 2.0 / 3.0; a.dinv = nullptr;
  const int nwg = std::min((a.nslices + 3) / 4, 2048);
  const size_t lds = (size_t)np * nu * 12 + 8 + (size_t)np * 8;
  const size_t lds16 = (size_t)np * nu * 16 + (size_t)np * 8 + 16;
  printf("rows %ld slices %d workgroups %d lds %zu B\n", (long)N, a.nslices, nwg, lds);
#define RUN(DBGV, label) { \
    float t = time_it([&] { \
      SellSArgs b = a; b.x = s0; b.s_out = s1; b.b = r; b.y = r; b.x2 = x; b.xmode = 1; \
      hipLaunchKernelGGL((sells_kernel<EPI_SWEEP, true, 3, 3, false, DBGV>), dim3(nwg), dim3(256), lds, 0, b); \
      b.x = s1; b.s_out = s0; b.xmode = 2; \
      hipLaunchKernelGGL((sells_kernel<EPI_SWEEP, true, 3, 3, false, DBGV>), dim3(nwg), dim3(256), lds, 0, b); }); \
    printf("%-52s %7.2f us per sweep\n", label, t * 1e3 / 2); }
  // launch-geometry sensitivity of the product kernel and of its bare skeleton
  for (int wpbv : {4}) for (int wg : {2048}) {
    const int g2 = std::min((a.nslices + wpbv - 1) / wpbv, wg);
    float t0 = time_it([&] { SellSArgs b = a; b.x = s0; b.s_out = s1; b.b = r; b.y = r; b.x2 = x; b.xmode = 1;
      hipLaunchKernelGGL((sells_kernel<EPI_SWEEP, true, 3, 3, false, 0>), dim3(g2), dim3(64 * wpbv), lds, 0, b); b.x = s1; b.s_out = s0; b.xmode = 2;
      hipLaunchKernelGGL((sells_kernel<EPI_SWEEP, true, 3, 3, false, 0>), dim3(g2), dim3(64 * wpbv), lds, 0, b); }, 30);
    float t1 = time_it([&] { SellSArgs b = a; b.x = s0; b.s_out = s1; b.b = r; b.y = r; b.x2 = x; b.xmode = 1;
      hipLaunchKernelGGL((sells_kernel<EPI_SWEEP, true, 3, 3, false, 15>), dim3(g2), dim3(64 * wpbv), lds, 0, b); b.x = s1; b.s_out = s0; b.xmode = 2;
      hipLaunchKernelGGL((sells_kernel<EPI_SWEEP, true, 3, 3, false, 15>), dim3(g2), dim3(64 * wpbv), lds, 0, b); }, 30);
    printf("waves/wg %d workgroups %6d : product %7.2f us   skeleton %7.2f us\n", wpbv, g2, t0 * 1e3 / 2, t1 * 1e3 / 2);
  }
#define RUNB(NBV, wpbv, wg, NTV, REMAP) { const int g2 = std::min((a.nslices + wpbv - 1) / wpbv, wg); \
    SellSArgs b = a; b.xcd_remap = REMAP; b.b = r; b.y = r; b.x2 = x; float t = time_it([&] { b.x = s0; b.s_out = s1; b.xmode = 1; \
      hipLaunchKernelGGL((sells_sweep_kernel<1, NBV, true, true, NTV>), dim3(g2), dim3(64 * wpbv), lds16, 0, b); b.x = s1; b.s_out = s0; b.xmode = 2; \
      hipLaunchKernelGGL((sells_sweep_kernel<2, NBV, true, true, NTV>), dim3(g2), dim3(64 * wpbv), lds16, 0, b); }, 30); \
    printf("sells_sweep_kernel NB=%d wpb=%d wgs=%d nt=%d remap=%d : %8.2f us per sweep (pair average)\n", NBV, wpbv, g2, NTV, REMAP, t * 1e3 / 2); }
  for (int wg : {1024, 2048, 4096}) { RUNB(1, 4, wg, 0, 1) RUNB(2, 4, wg, 0, 1) RUNB(2, 4, wg, 1, 1) RUNB(2, 4, wg, 0, 0) RUNB(2, 4, wg, 1, 0) RUNB(1, 4, wg, 1, 0) }
  RUN(0, "product kernel")
  RUN(1, "- high-word mask")
  RUN(2, "- LDS coefficient reads")
  RUN(3, "- mask - coefficient reads (no LDS in the tap loop)")
  RUN(4, "- DPP shifts")
  RUN(8, "- gathers")
  RUN(12, "- gathers - DPP")
  RUN(15, "- everything (head/tail streams + 27 FMAs only)")
  return 0;
}
