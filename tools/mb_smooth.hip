// tools/mb_smooth.hip -- where does the time of the one-launch smoothing pass (sells_smooth_kernel) go?
// Diagnostic only (not part of the product): runs the PRODUCT kernel on a synthetic 27-point operator of (n-1)^3 rows with the
// launcher's geometry and its DBG ablations (results of the ablated variants are meaningless; only their time is read).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/mbsmooth tools/mb_smooth.hip && /tmp/mbsmooth [cells] [niter]
#include "../gridapsolvers.jl_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace gmg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename F> float time_it(F f, int reps = 40)
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
  const int nc = argc > 1 ? atoi(argv[1]) : 64;
  const int niter = argc > 2 ? atoi(argv[2]) : 10;
  const int m = nc - 1;
  const int64_t N = (int64_t)m * m * m;
  const int nruns = 9, nu = 27, np = 28;
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
  std::vector<double> tab8(tab.size());
  for (size_t q = 0; q < tab.size(); ++q) tab8[q] = tab[q].v;
  std::vector<uint16_t> rowpid((size_t)N + 64, 27);
  for (int z = 0; z < m; ++z) for (int y = 0; y < m; ++y) for (int x = 0; x < m; ++x) {
    auto t = [&](int c) { return c == 0 ? 0 : (c == m - 1 ? 2 : 1); };
    rowpid[(size_t)x + (size_t)m * (y + (size_t)m * z)] = (uint16_t)((t(z) * 3 + t(y)) * 3 + t(x));
  }
  uint16_t *d_pid; PatEntry *d_tab; int32_t *d_run; double *d_pd, *d_t8, *r, *s0, *s1, *x; uint32_t *flags, *err;
  CK(hipMalloc(&d_pid, rowpid.size() * 2)); CK(hipMalloc(&d_tab, tab.size() * sizeof(PatEntry))); CK(hipMalloc(&d_run, runs.size() * 4)); CK(hipMalloc(&d_t8, tab8.size() * 8));
  CK(hipMalloc(&d_pd, np * 8)); CK(hipMalloc(&r, (N + 64) * 8)); CK(hipMalloc(&s0, (N + 64) * 8)); CK(hipMalloc(&s1, (N + 64) * 8)); CK(hipMalloc(&x, (N + 64) * 8));
  CK(hipMalloc(&flags, 256 * 64)); CK(hipMalloc(&err, 64)); CK(hipMemset(flags, 0, 256 * 64)); CK(hipMemset(err, 0, 64));
  CK(hipMemcpy(d_pid, rowpid.data(), rowpid.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(PatEntry), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_t8, tab8.data(), tab8.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_run, runs.data(), runs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pd, pdinv.data(), np * 8, hipMemcpyHostToDevice));
  CK(hipMemset(r, 0, (N + 64) * 8)); CK(hipMemset(s0, 0, (N + 64) * 8)); CK(hipMemset(s1, 0, (N + 64) * 8)); CK(hipMemset(x, 0, (N + 64) * 8));
  // geometry as gmg_solver::smooth_persistent picks it
  const int n_cus = 256, rows = 62;
  const int nsl = (int)((N + rows - 1) / rows);
  int wpb = 1;
  while (wpb < 16 && (nsl + wpb - 1) / wpb > n_cus) wpb *= 2;
  int ns = 1;
  while ((nsl + wpb * ns - 1) / (wpb * ns) > n_cus) ++ns;   // (the library stops at 2)
  const int64_t reach = std::max<int64_t>(-(int64_t)runs.front(), (int64_t)runs.back() + 2);
  const int halo = (int)((reach + (int64_t)wpb * ns * rows - 1) / ((int64_t)wpb * ns * rows));
  const int nwg = (nsl + wpb * ns - 1) / (wpb * ns);
  printf("rows %lld slices %d : %d workgroups x %d waves, %d slice(s) per wave, neighbours +-%d, %d sweeps per launch\n", (long long)N, nsl, nwg, wpb, ns, halo, niter);
  if (ns != 1 && ns != 2 && ns != 4 && ns != 9) { printf("(NS = %d is not instantiated here)\n", ns); return 0; }
  SellSmoothArgs a;
  std::memset(&a, 0, sizeof(a));
  a.rowpid = d_pid; a.tab = d_tab; a.tab8 = d_t8; a.run_off = d_run; a.np = np; a.nruns = nruns; a.nrows = N; a.ncols = N; a.nslices = nsl;
  a.pdinv = d_pd; a.omega = 2.0 / 3.0; a.niter = niter; a.x_zero = 0; a.r_in = r; a.r_out = r; a.x = x; a.s_a = s0; a.s_b = s1;
  a.flags = flags; a.err = err; a.halo_wg = halo;
  uint32_t epoch = 1;
  const size_t lds = (size_t)np * nu * 16 + (size_t)np * 8 + 16;
  const dim3 g(nwg), b(64 * wpb);
#define RUN(DBGV, label)                                                                                         \
  do {                                                                                                           \
    const float ms = time_it([&] { a.epoch = epoch; epoch += niter;                                              \
      if (ns == 1) hipLaunchKernelGGL((sells_smooth_kernel<1, true, true, DBGV>), g, b, lds, 0, a);              \
      else if (ns == 2) hipLaunchKernelGGL((sells_smooth_kernel<2, true, true, DBGV>), g, b, lds, 0, a);         \
      else if (ns == 4) hipLaunchKernelGGL((sells_smooth_kernel<4, true, true, DBGV>), g, b, lds, 0, a);         \
      else hipLaunchKernelGGL((sells_smooth_kernel<9, true, true, DBGV>), g, b, lds, 0, a); });                  \
    uint32_t e = 0; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));                                            \
    printf("%-58s %8.2f us per launch  %6.2f us per sweep%s\n", label, ms * 1e3, ms * 1e3 / niter, e ? "  (TIMEOUT FLAG SET)" : ""); \
  } while (0)
  RUN(0, "product kernel (relaxed progress words)");
  a.fenced = 1;
  RUN(0, "product kernel, release / acquire progress words");
  a.fenced = 0;
  RUN(1, "- neighbour waits");
  RUN(1 | 16, "- waits - store drain / flag publish");
  RUN(2, "plain gather loads (stale L1 allowed)");
  RUN(4, "plain s stores");
  RUN(2 | 4, "plain loads + plain stores");
  RUN(8, "- taps (3 of 27)");
  RUN(1 | 16 | 2 | 4, "no sync, plain loads and stores");
  RUN(1 | 16 | 2 | 4 | 8, "no sync, plain memory ops, no taps (skeleton)");
  {
    const float ms = time_it([&] { a.epoch = epoch; epoch += niter;
      if (ns == 1) hipLaunchKernelGGL((sells_smooth_kernel<1, true, false, 0>), g, b, lds, 0, a);
      else if (ns == 2) hipLaunchKernelGGL((sells_smooth_kernel<2, true, false, 0>), g, b, lds, 0, a);
      else if (ns == 4) hipLaunchKernelGGL((sells_smooth_kernel<4, true, false, 0>), g, b, lds, 0, a);
      else hipLaunchKernelGGL((sells_smooth_kernel<9, true, false, 0>), g, b, lds, 0, a); });
    printf("%-58s %8.2f us per launch  %6.2f us per sweep\n", "product kernel, 8-byte table entries", ms * 1e3, ms * 1e3 / niter);
  }
  return 0;
}
