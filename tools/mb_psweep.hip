// tools/mb_psweep.hip -- where does the time of the r-gather sweep go, and what do two rows per lane / a software pipeline change?
// Diagnostic only (not part of the product).  On a synthetic 27-point operator of (n-1)^3 rows, with the product's launch geometry:
// sells_rsweep_kernel (round 3) and sells_r2sweep_kernel (round 4, the product) from kernels.hpp, checked against each other bit for
// bit, and sells_psweep_kernel -- an experiment that lives in THIS file: the gather sweep as an inline-asm software pipeline (no gain,
// profiles/r04_tuning.md) which carries the DBG ablations (no taps / no stores / no gathers / no table / no requests).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o build_tools/mbps tools/mb_psweep.hip && build_tools/mbps [cells] [wgs]
#include "../gridapsolvers.jl_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
using namespace gmg;
namespace gmg {
// ---------------------------------------------------------------------------
// The r-gather sweep as a SOFTWARE PIPELINE ("pipelined gather sweep").  sells_rsweep_kernel's waves are phase-locked: all eight
// waves of a SIMD start together, request the operands of their slice, wait out the same ~2 us of memory latency, multiply at the
// same time (each at 1/8 of the SIMD), store, and request the next slice -- memory latency and arithmetic never overlap (128^3: four
// rounds of ~2 us latency + ~2.3 us of taps = the 21 us the kernel takes whatever its instruction count or launch shape).  Letting a
// wave request slice k+1 before it multiplies slice k does not work with compiler-generated loads on gfx9: loads and stores share
// vmcnt and complete out of order with respect to each other, so with a store pending the compiler waits for a load with vmcnt(0) --
// which also waits for the requests just issued.  Here every operand of the pipeline is requested with an inline-asm global_load
// (the compiler does not track them) and waited for by hand, and the stores of slice k are issued one iteration LATE, right after the
// wait of iteration k+1:
//     request(k+1) ; s_waitcnt vmcnt(#requests) ; store(k-1) ; taps(k)
// At the wait, everything older than the newest requests -- operands of slice k and the stores of slice k-2 -- has had a whole
// taps phase to complete, so vmcnt(#requests) holds as soon as the operands of k are there: one latency up front, then arithmetic-bound.
// Two explicit register sets (the loop is unrolled by two) so that no compiler copy ever touches a register whose load is in flight.
// Same taps, order and roundings as sells_rsweep_kernel: bit-identical.  FM: fused multiply-add taps (one rounding per tap instead of
// two: NOT the reference's mul! arithmetic; option pat_fma).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void gl_req_f64(double &dst, uint32_t byteoff, const double *base)
{
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(byteoff), "s"(base));
}
__device__ __forceinline__ void gl_req_u16(int &dst, uint32_t byteoff, const uint16_t *base)
{
  asm volatile("global_load_ushort %0, %1, %2" : "=v"(dst) : "v"(byteoff), "s"(base));
}

template <int NR>
struct PipeRegs {
  double g[NR];      // r_k at row + run_off[q] (clamped)
  double e0, x, rp;  // the row's own r_k, x, r_{k-1}
  int pid;
};

// DBG (tools/mb_psweep.hip only; the library instantiates DBG = 0): timing ablations, WRONG results by design -- bit0 no taps, bit1 no
// stores, bit2 no gather requests (row-wise operands only), bit3 no pattern table in LDS, bit4 no requests at all.
template <int XM, bool MK, bool FM, int NR, int DBG = 0>
__global__ __launch_bounds__(kBlock) void sells_psweep_kernel(SellSArgs a)
{
  constexpr int K = 3, ROWS = 65 - K;
  constexpr int NL = NR + 2 + (XM != 1 ? 1 : 0) + (XM == 2 ? 1 : 0);      // requests per slice
  extern __shared__ double sp_smem[];
  const int nu = K * NR;
  const int tot = a.np * nu;
  double *s_tab8 = sp_smem;                                   // [np*nu] coefficients, dense | MK: [np*nu] high-word masks
  uint32_t *s_msk = reinterpret_cast<uint32_t *>(sp_smem + tot);
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: the pipeline's branches are uniform
  const int nwg = gridDim.x;
  const int blk = remap_block(blockIdx.x, nwg, a.xcd_remap);
  const int chunk_lo = a.nslices / nwg, chunk_rem = a.nslices % nwg;
  const int s_begin = blk * chunk_lo + min(blk, chunk_rem);
  const int s_end = s_begin + chunk_lo + (blk < chunk_rem ? 1 : 0);
  const double *__restrict__ rg = a.x;
  const double omega = a.omega;
  const double du = a.pdinv[0];
  const int last8 = 8 * ((int)a.ncols - 1);
  const int lastrow = (int)a.nrows - 1;
  const bool xz = a.x_zero != 0;
  int roff8[NR];
#pragma unroll
  for (int q = 0; q < NR; ++q) roff8[q] = 8 * a.run_off[q];                // scalar registers

  auto request = [&](PipeRegs<NR> &R, int slice) {
    const int row = slice * ROWS + lane;
    const int rc = min(row, lastrow);
    if (DBG & 16) { R.pid = 0; R.e0 = 1.0; R.x = 1.0; R.rp = 1.0; for (int q = 0; q < NR; ++q) R.g[q] = (double)(row + q); return; }
    gl_req_u16(R.pid, 2u * (uint32_t)rc, a.rowpid);
    gl_req_f64(R.e0, 8u * (uint32_t)rc, rg);
    if (XM != 1) gl_req_f64(R.x, 8u * (uint32_t)rc, a.x2);
    if (XM == 2) gl_req_f64(R.rp, 8u * (uint32_t)rc, a.s_out);
    if (DBG & 4) { for (int q = 0; q < NR; ++q) R.g[q] = (double)(row + q); return; }
#pragma unroll
    for (int q = 0; q < NR; ++q) gl_req_f64(R.g[q], (uint32_t)clamp0_med3(8 * row + roff8[q], last8), rg);
  };
  // wait until at most `n` vector-memory operations are outstanding; the registers of R become valid here (tied operands: the
  // compiler may not move their uses above this point)
  auto arrive = [&](PipeRegs<NR> &R, bool more) {
    if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DBG & 16) ? 0 : ((DBG & 4) ? NL - NR : NL)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(R.pid), "+v"(R.e0));
    if (XM != 1) asm volatile("" : "+v"(R.x));
    if (XM == 2) asm volatile("" : "+v"(R.rp));
#pragma unroll
    for (int q = 0; q < NR; ++q) asm volatile("" : "+v"(R.g[q]));
  };
  double p_rn = 0.0, p_xn = 0.0;
  int p_row = 0;
  bool p_live = false;                                          // results of the previous slice, stored one iteration late
  auto store_prev = [&]() {
    if ((DBG & 2) && p_rn != 1.2345e300) { asm volatile("" ::: "memory"); return; }
    if (p_live && lane < ROWS && p_row <= lastrow) {
      if (XM != 1) a.x2[p_row] = p_xn;
      a.y[p_row] = p_rn;
    }
    asm volatile("" ::: "memory");
  };
  auto taps = [&](PipeRegs<NR> &R, int slice) {
    const uint32_t *tm = s_msk + R.pid * nu;
    const double *tv = s_tab8 + R.pid * nu;
    if (DBG & 1) {
      double s = 0.0;
      for (int q = 0; q < NR; ++q) s += R.g[q];
      p_rn = R.e0 - s; p_xn = R.x + R.rp; p_row = slice * ROWS + lane; p_live = true;
      return;
    }
    double cur[NR];
    bool fin = true;
#pragma unroll
    for (int q = 0; q < NR; ++q) {
      cur[q] = omega * (du * R.g[q]);                           // s = omega*(Dinv*r): once per loaded value
      if (MK) fin = fin && __builtin_isfinite(cur[q]);
    }
    if (MK) fin = __all(fin);
    double s = 0.0;
    if (MK && !fin) {
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        double c = cur[q];
#pragma unroll
        for (int t = 0; t < K; ++t) {
          if (t > 0) c = wave_shl1(c);
          const double g = __hiloint2double(__double2hiint(c) & (int)tm[q * K + t], __double2loint(c));
          s = FM ? __builtin_fma(tv[q * K + t], g, s) : s + tv[q * K + t] * g;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        double c = cur[q];
#pragma unroll
        for (int t = 0; t < K; ++t) {
          if (t > 0) c = wave_shl1(c);
          const double cf = tv[q * K + t];
          s = FM ? __builtin_fma(cf, c, s) : s + cf * c;
        }
      }
    }
    const double sk = omega * (du * R.e0);                      // the row's own s_k
    p_rn = R.e0 - s;
    if (XM == 0) p_xn = (xz ? 0.0 : R.x) + sk;
    else if (XM == 2) p_xn = ((xz ? 0.0 : R.x) + omega * (du * R.rp)) + sk;
    p_row = slice * ROWS + lane;
    p_live = true;
  };

  PipeRegs<NR> A, B;
  int sl = s_begin + wave;
  const bool any = sl < s_end;
  if (any) request(A, sl);                                      // in flight while the table is staged
  if (!(DBG & 8)) {
    if (MK) { for (int i = threadIdx.x; i < tot; i += blockDim.x) { const PatEntry en = a.tab[i]; s_tab8[i] = en.v; s_msk[i] = en.m; } }
    else { for (int i = threadIdx.x; i < tot; i += blockDim.x) s_tab8[i] = a.tab8[i]; }
  }
  __syncthreads();
  if (!any) return;
  while (true) {
    int sn = sl + wpb;
    bool more = sn < s_end;
    if (more) request(B, sn);
    arrive(A, more);
    store_prev();
    taps(A, sl);
    if (!more) break;
    sl = sn; sn = sl + wpb; more = sn < s_end;
    if (more) request(A, sn);
    arrive(B, more);
    store_prev();
    taps(B, sl);
    if (!more) break;
    sl = sn;
  }
  store_prev();
}

} // namespace gmg
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename F> float time_it(F f, int reps = 60)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 6; ++i) f();
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}
__global__ void empty_kernel(int) {}

int main(int argc, char **argv)
{
  const int nc = argc > 1 ? atoi(argv[1]) : 128;
  const int wgs_arg = argc > 2 ? atoi(argv[2]) : 2048;
  const int m = nc - 1;
  const int64_t N = (int64_t)m * m * m;
  const int nruns = 9, nu = 27, np = 28;
  std::vector<int32_t> runs;
  // argv[3] = 1: timing experiment only (wrong results) -- the three runs of a plane all read the SAME line, so two of three gathers hit
  // in L1: what would the sweep gain if the lines of a plane were fetched once from L2 instead of three times?
  const int same_line = argc > 3 ? atoi(argv[3]) : 0;
  for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) runs.push_back(dz * m * m + (same_line ? 0 : dy) * m - 1);
  std::vector<PatEntry> tab((size_t)np * nu);
  std::memset(tab.data(), 0, tab.size() * sizeof(PatEntry));
  std::vector<double> pdinv(np, 3.0 / 8.0);
  for (int tz = 0; tz < 3; ++tz) for (int ty = 0; ty < 3; ++ty) for (int tx = 0; tx < 3; ++tx) {
    const int p = (tz * 3 + ty) * 3 + tx;
    for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) {
      const bool ok = !((tx == 0 && dx < 0) || (tx == 2 && dx > 0) || (ty == 0 && dy < 0) || (ty == 2 && dy > 0) || (tz == 0 && dz < 0) || (tz == 2 && dz > 0));
      const int e = ((dz + 1) * 3 + (dy + 1)) * 3 + (dx + 1);
      if (ok) { tab[(size_t)p * nu + e].v = (dx || dy || dz) ? -1.0 / 12.0 : 8.0 / 3.0; tab[(size_t)p * nu + e].m = 0xffffffffu; }
    }
  }
  std::vector<double> tab8(tab.size());
  for (size_t i = 0; i < tab.size(); ++i) tab8[i] = tab[i].v;
  std::vector<uint16_t> rowpid((size_t)N + 64, 27);
  for (int z = 0; z < m; ++z) for (int y = 0; y < m; ++y) for (int x = 0; x < m; ++x) {
    auto t = [&](int c) { return c == 0 ? 0 : (c == m - 1 ? 2 : 1); };
    rowpid[(size_t)x + (size_t)m * (y + (size_t)m * z)] = (uint16_t)((t(z) * 3 + t(y)) * 3 + t(x));
  }
  uint16_t *d_pid; PatEntry *d_tab; double *d_tab8; int32_t *d_run; double *d_pd, *r0, *r1, *x;
  CK(hipMalloc(&d_pid, rowpid.size() * 2)); CK(hipMalloc(&d_tab, tab.size() * sizeof(PatEntry))); CK(hipMalloc(&d_tab8, tab8.size() * 8)); CK(hipMalloc(&d_run, runs.size() * 4));
  CK(hipMalloc(&d_pd, np * 8)); CK(hipMalloc(&r0, (N + 64) * 8)); CK(hipMalloc(&r1, (N + 64) * 8)); CK(hipMalloc(&x, (N + 64) * 8));
  CK(hipMemcpy(d_pid, rowpid.data(), rowpid.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tab, tab.data(), tab.size() * sizeof(PatEntry), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tab8, tab8.data(), tab8.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_run, runs.data(), runs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_pd, pdinv.data(), np * 8, hipMemcpyHostToDevice));
  CK(hipMemset(r0, 0, (N + 64) * 8)); CK(hipMemset(r1, 0, (N + 64) * 8)); CK(hipMemset(x, 0, (N + 64) * 8));
  SellSArgs a;
  std::memset(&a, 0, sizeof(a));
  a.rowpid = d_pid; a.tab = d_tab; a.tab8 = d_tab8; a.run_off = d_run; a.np = np; a.nruns = nruns; a.minoff = runs.front(); a.maxoff = runs.back();
  a.nrows = N; a.ncols = N; a.nslices = (int)((N + 61) / 62); a.xcd_remap = 1; a.pdinv = d_pd; a.omega = 2.0 / 3.0;
  const size_t lds = (size_t)np * nu * 16 + 16;
  printf("rows %ld slices %d\n", (long)N, a.nslices);
  // correctness first: one XM = 1 + XM = 2 pair on random data, two-rows-per-lane and pipelined forms against sells_rsweep_kernel, bit for bit
  {
    std::vector<double> h((size_t)N), hx((size_t)N);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
    for (auto &v : h) v = rnd();
    for (auto &v : hx) v = rnd();
    std::vector<double> ref_r((size_t)N), ref_x((size_t)N), got_r((size_t)N), got_x((size_t)N);
    auto reset = [&]() { CK(hipMemcpy(r0, h.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemset(r1, 0, N * 8)); CK(hipMemcpy(x, hx.data(), N * 8, hipMemcpyHostToDevice)); };
    auto fetch = [&](std::vector<double> &rr, std::vector<double> &xx) { CK(hipDeviceSynchronize()); CK(hipMemcpy(rr.data(), r0, N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(xx.data(), x, N * 8, hipMemcpyDeviceToHost)); };
    const int g1 = std::min((a.nslices + 3) / 4, 2048);
    reset();
    { SellSArgs b = a; b.x = r0; b.b = r0; b.y = r1; b.x2 = x; b.s_out = nullptr; b.xmode = 1;
      hipLaunchKernelGGL((sells_rsweep_kernel<1, 1, true>), dim3(g1), dim3(256), lds, 0, b);
      b.x = r1; b.b = r1; b.y = r0; b.s_out = r0; b.xmode = 2;
      hipLaunchKernelGGL((sells_rsweep_kernel<2, 1, true>), dim3(g1), dim3(256), lds, 0, b); }
    fetch(ref_r, ref_x);
    SellSArgs a2 = a; a2.nslices = (int)((N + 125) / 126);
    const int g2 = std::min((a2.nslices + 3) / 4, 2048);
    reset();
    { SellSArgs b = a2; b.x = r0; b.b = r0; b.y = r1; b.x2 = x; b.s_out = nullptr; b.xmode = 1;
      hipLaunchKernelGGL((sells_r2sweep_kernel<1, true, false, 9>), dim3(g2), dim3(256), lds, 0, b);
      b.x = r1; b.b = r1; b.y = r0; b.s_out = r0; b.xmode = 2;
      hipLaunchKernelGGL((sells_r2sweep_kernel<2, true, false, 9>), dim3(g2), dim3(256), lds, 0, b); }
    fetch(got_r, got_x);
    size_t bad = 0;
    for (size_t i = 0; i < (size_t)N; ++i) bad += (std::memcmp(&got_r[i], &ref_r[i], 8) != 0) + (std::memcmp(&got_x[i], &ref_x[i], 8) != 0);
    printf("two rows per lane vs sells_rsweep_kernel: %zu differing values of %ld\n", bad, (long)(2 * N));
    reset();
    { SellSArgs b = a; b.x = r0; b.b = r0; b.y = r1; b.x2 = x; b.s_out = nullptr; b.xmode = 1;
      hipLaunchKernelGGL((sells_psweep_kernel<1, true, false, 9>), dim3(g1), dim3(256), lds, 0, b);
      b.x = r1; b.b = r1; b.y = r0; b.s_out = r0; b.xmode = 2;
      hipLaunchKernelGGL((sells_psweep_kernel<2, true, false, 9>), dim3(g1), dim3(256), lds, 0, b); }
    fetch(got_r, got_x);
    bad = 0;
    for (size_t i = 0; i < (size_t)N; ++i) bad += (std::memcmp(&got_r[i], &ref_r[i], 8) != 0) + (std::memcmp(&got_x[i], &ref_x[i], 8) != 0);
    printf("pipelined form vs sells_rsweep_kernel:    %zu differing values of %ld\n", bad, (long)(2 * N));
    reset();
    { SellSArgs b = a2; b.x = r0; b.b = r0; b.y = r1; b.x2 = x; b.s_out = nullptr; b.xmode = 1;
      hipLaunchKernelGGL((sells_r2sweep_kernel<1, true, true, 9>), dim3(g2), dim3(256), lds, 0, b);
      b.x = r1; b.b = r1; b.y = r0; b.s_out = r0; b.xmode = 2;
      hipLaunchKernelGGL((sells_r2sweep_kernel<2, true, true, 9>), dim3(g2), dim3(256), lds, 0, b); }
    fetch(got_r, got_x);
    double mx = 0.0, mr = 0.0;
    for (size_t i = 0; i < (size_t)N; ++i) { mx = std::max(mx, std::fabs(got_r[i] - ref_r[i])); mr = std::max(mr, std::fabs(ref_r[i])); }
    printf("fused multiply-add taps vs exact: max |dr| / max |r| = %.3e\n", mx / mr);
    CK(hipMemset(r0, 0, (N + 64) * 8)); CK(hipMemset(r1, 0, (N + 64) * 8)); CK(hipMemset(x, 0, (N + 64) * 8));
  }
  { float t = time_it([&] { hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(256), 0, 0, 0); hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(256), 0, 0, 0); });
    printf("%-64s %7.2f us per launch\n", "empty kernel, 2048 x 256 (back-to-back launch floor)", t * 1e3 / 2); }
  // pair of sweeps as the product issues them: XM = 1 (r_cur -> r_next) then XM = 2 (r_next -> r_cur, x updated with both increments)
#define RUNP(MKV, FMV, DBGV, wpbv, wg, label) { const int g2 = std::min((a.nslices + wpbv - 1) / wpbv, wg); \
    float t = time_it([&] { SellSArgs b = a; b.x = r0; b.b = r0; b.y = r1; b.x2 = x; b.s_out = nullptr; b.xmode = 1; \
      hipLaunchKernelGGL((sells_psweep_kernel<1, MKV, FMV, 9, DBGV>), dim3(g2), dim3(64 * wpbv), lds, 0, b); \
      b.x = r1; b.b = r1; b.y = r0; b.s_out = r0; b.xmode = 2; \
      hipLaunchKernelGGL((sells_psweep_kernel<2, MKV, FMV, 9, DBGV>), dim3(g2), dim3(64 * wpbv), lds, 0, b); }); \
    printf("%-64s %7.2f us per sweep (wpb %d, wgs %d)\n", label, t * 1e3 / 2, wpbv, g2); }
#define RUNR(MKV, wpbv, wg, label) { const int g2 = std::min((a.nslices + wpbv - 1) / wpbv, wg); \
    float t = time_it([&] { SellSArgs b = a; b.x = r0; b.b = r0; b.y = r1; b.x2 = x; b.s_out = nullptr; b.xmode = 1; \
      hipLaunchKernelGGL((sells_rsweep_kernel<1, 1, MKV>), dim3(g2), dim3(64 * wpbv), lds, 0, b); \
      b.x = r1; b.b = r1; b.y = r0; b.s_out = r0; b.xmode = 2; \
      hipLaunchKernelGGL((sells_rsweep_kernel<2, 1, MKV>), dim3(g2), dim3(64 * wpbv), lds, 0, b); }); \
    printf("%-64s %7.2f us per sweep (wpb %d, wgs %d)\n", label, t * 1e3 / 2, wpbv, g2); }
  RUNR(true, 4, wgs_arg, "sells_rsweep_kernel (round-3 product)")
  // two rows per lane: slices of 126 rows
  { SellSArgs a2 = a; a2.nslices = (int)((N + 125) / 126);
#define RUN2(MKV, FMV, wpbv, wg, label) { const int g2 = std::min((a2.nslices + wpbv - 1) / wpbv, wg); \
    float t = time_it([&] { SellSArgs b = a2; b.x = r0; b.b = r0; b.y = r1; b.x2 = x; b.s_out = nullptr; b.xmode = 1; \
      hipLaunchKernelGGL((sells_r2sweep_kernel<1, MKV, FMV, 9>), dim3(g2), dim3(64 * wpbv), lds, 0, b); \
      b.x = r1; b.b = r1; b.y = r0; b.s_out = r0; b.xmode = 2; \
      hipLaunchKernelGGL((sells_r2sweep_kernel<2, MKV, FMV, 9>), dim3(g2), dim3(64 * wpbv), lds, 0, b); }); \
    printf("%-64s %7.2f us per sweep (wpb %d, wgs %d)\n", label, t * 1e3 / 2, wpbv, g2); }
    for (int wg : {512, 1024, 1355, 2048, 4096}) RUN2(true, false, 4, wg, "sells_r2sweep_kernel (two rows per lane)")
#define RUN2O(MKV, FMV, wpbv, wg, label) { const int g2 = std::min((a2.nslices + wpbv - 1) / wpbv, wg); \
    float t = time_it([&] { SellSArgs b = a2; b.x = r0; b.b = r0; b.y = r1; b.x2 = x; b.s_out = nullptr; b.xmode = 1; \
      hipLaunchKernelGGL((sells_r2sweep_kernel<1, MKV, FMV, 9, 1>), dim3(g2), dim3(64 * wpbv), lds, 0, b); \
      b.x = r1; b.b = r1; b.y = r0; b.s_out = r0; b.xmode = 2; \
      hipLaunchKernelGGL((sells_r2sweep_kernel<2, MKV, FMV, 9, 1>), dim3(g2), dim3(64 * wpbv), lds, 0, b); }); \
    printf("%-64s %7.2f us per sweep (wpb %d, wgs %d)\n", label, t * 1e3 / 2, wpbv, g2); }
    for (int wg : {1024, 1536, 2048, 2560, 3072, 4096}) RUN2O(true, false, 4, wg, "sells_r2sweep_kernel, rolled run loop, <= 64 registers")
    for (int wg : {512, 1024, 2048}) RUN2O(true, false, 8, wg, "sells_r2sweep_kernel, rolled run loop, <= 64 registers")
    for (int wg : {2048, 4096}) RUN2O(true, false, 2, wg, "sells_r2sweep_kernel, rolled run loop, <= 64 registers")
    for (int wg : {2048}) RUN2O(true, true, 4, wg, "sells_r2sweep_kernel, rolled, fused multiply-add")
    for (int wg : {1024, 2048}) RUN2(true, true, 4, wg, "sells_r2sweep_kernel, fused multiply-add taps")
    for (int wg : {1024, 2048}) RUN2(false, false, 4, wg, "sells_r2sweep_kernel, no mask array at all")
    for (int wg : {2048}) RUN2(false, true, 4, wg, "sells_r2sweep_kernel, no masks, fused multiply-add")
    for (int wg : {2048, 4096}) RUN2(true, false, 2, wg, "sells_r2sweep_kernel, 2 waves per workgroup")
  }
  RUNP(true, false, 0, 4, wgs_arg, "sells_psweep_kernel (pipelined)")
  RUNP(true, true, 0, 4, wgs_arg, "  fused multiply-add taps")
  RUNP(true, false, 1, 4, wgs_arg, "  - taps (sum of the gathers instead)")
  RUNP(true, false, 2, 4, wgs_arg, "  - stores")
  RUNP(true, false, 3, 4, wgs_arg, "  - taps - stores")
  RUNP(true, false, 4, 4, wgs_arg, "  - gather requests")
  RUNP(true, false, 8, 4, wgs_arg, "  - pattern table staging")
  RUNP(true, false, 9, 4, wgs_arg, "  - taps - table")
  RUNP(true, false, 5, 4, wgs_arg, "  - taps - gathers (row-wise operands in, results out)")
  RUNP(true, false, 13, 4, wgs_arg, "  - taps - gathers - table")
  RUNP(true, false, 15, 4, wgs_arg, "  - taps - gathers - table - stores")
  RUNP(true, false, 16 + 8 + 2, 4, wgs_arg, "  taps only (no requests, no table, no stores)")
  RUNP(true, false, 16 + 8 + 2 + 1, 4, wgs_arg, "  nothing (launch + loop skeleton)")
  for (int wpbv : {4}) for (int wg : {512, 1024, 4096, 8192}) RUNP(true, false, 0, 4, wg, "sells_psweep_kernel, other grids")
  RUNP(true, false, 0, 2, 4096, "sells_psweep_kernel, 2 waves per workgroup")
  return 0;
}
