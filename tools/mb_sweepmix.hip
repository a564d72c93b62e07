// tools/mb_sweepmix.hip -- what can the vector traffic of the row-pattern sweep reach with NO operator work at all?
// Diagnostic only (not part of the product).  The fused SELL-P sweep moves, per row: rowpid (2 B) + r in/out + s in/out
// (+ x in/out and s_prev in every second sweep).  This program streams exactly those arrays with (a) 8 B per lane -- the
// sweep's own access width (lane = row) -- and (b) 16 B per lane, persistent launch of 2048 x 256 like the product kernel,
// and prints the achieved GB/s: the ceiling the sweep's 3.3-3.5 TB/s has to be read against.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mbs tools/mb_sweepmix.hip && /tmp/mbs [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// XMODE 0: x untouched (odd sweeps of the deferred-x scheme) ; 1: x read+write and s_prev read (even sweeps)
template <int XMODE, int UN>
__global__ __launch_bounds__(256) void mix8(int64_t n, const uint16_t *__restrict__ pid, const double *__restrict__ r, const double *__restrict__ s,
                                            double *__restrict__ r2, double *__restrict__ s2, double *__restrict__ x)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += stride * UN) {
    double rv[UN], sv[UN], xv[UN], sp[UN]; int pv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n) { pv[u] = pid[i]; rv[u] = r[i]; sv[u] = s[i]; if (XMODE) { xv[u] = x[i]; sp[u] = s2[i]; } }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n) {
        const double rn = rv[u] - 0.5 * sv[u] * (double)(pv[u] & 1);
        r2[i] = rn; s2[i] = 0.25 * rn;
        if (XMODE) x[i] = (xv[u] + sp[u]) + sv[u];
      }
    }
  }
}
template <int XMODE, int UN>
__global__ __launch_bounds__(256) void mix16(int64_t n2, const uint32_t *__restrict__ pid2, const double2 *__restrict__ r, const double2 *__restrict__ s,
                                             double2 *__restrict__ r2, double2 *__restrict__ s2, double2 *__restrict__ x)
{
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n2; i0 += stride * UN) {
    double2 rv[UN], sv[UN], xv[UN], sp[UN]; uint32_t pv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n2) { pv[u] = pid2[i]; rv[u] = r[i]; sv[u] = s[i]; if (XMODE) { xv[u] = x[i]; sp[u] = s2[i]; } }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t i = i0 + u * stride;
      if (i < n2) {
        double2 rn; rn.x = rv[u].x - 0.5 * sv[u].x * (double)(pv[u] & 1); rn.y = rv[u].y - 0.5 * sv[u].y * (double)((pv[u] >> 16) & 1);
        r2[i] = rn; double2 sn; sn.x = 0.25 * rn.x; sn.y = 0.25 * rn.y; s2[i] = sn;
        if (XMODE) { double2 xn; xn.x = (xv[u].x + sp[u].x) + sv[u].x; xn.y = (xv[u].y + sp[u].y) + sv[u].y; x[i] = xn; }
      }
    }
  }
}

template <typename F> float time_it(F f, int reps = 40)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 5; ++i) f();
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char **argv)
{
  const int64_t n = argc > 1 ? atoll(argv[1]) : 2048383;
  const int64_t npad = (n + 1023) / 1024 * 1024;
  uint16_t *pid; double *r, *s, *r2, *s2, *x;
  CK(hipMalloc(&pid, npad * 2)); CK(hipMalloc(&r, npad * 8)); CK(hipMalloc(&s, npad * 8)); CK(hipMalloc(&r2, npad * 8)); CK(hipMalloc(&s2, npad * 8)); CK(hipMalloc(&x, npad * 8));
  CK(hipMemset(pid, 0, npad * 2)); CK(hipMemset(r, 0, npad * 8)); CK(hipMemset(s, 0, npad * 8)); CK(hipMemset(r2, 0, npad * 8)); CK(hipMemset(s2, 0, npad * 8)); CK(hipMemset(x, 0, npad * 8));
  const double b0 = (2.0 + 32.0) * n, b1 = (2.0 + 56.0) * n;       // bytes per launch: x untouched / x+s_prev touched
  printf("rows %ld ; per-launch bytes: %.1f MB (x untouched), %.1f MB (x pass); average of the pair %.1f MB\n", (long)n, b0 / 1e6, b1 / 1e6, 0.5 * (b0 + b1) / 1e6);
  const dim3 g(2048), b(256);
#define RUN8(XM, UN) { float t = time_it([&] { hipLaunchKernelGGL((mix8<XM, UN>), g, b, 0, 0, n, pid, r, s, r2, s2, x); }); \
    printf("  8 B/lane  x%s unroll %d : %7.2f us  %7.0f GB/s\n", XM ? "+" : "-", UN, t * 1e3, (XM ? b1 : b0) / t / 1e6); }
#define RUN16(XM, UN) { float t = time_it([&] { hipLaunchKernelGGL((mix16<XM, UN>), g, b, 0, 0, n / 2, (const uint32_t *)pid, (const double2 *)r, (const double2 *)s, (double2 *)r2, (double2 *)s2, (double2 *)x); }); \
    printf("  16 B/lane x%s unroll %d : %7.2f us  %7.0f GB/s\n", XM ? "+" : "-", UN, t * 1e3, (XM ? b1 : b0) / t / 1e6); }
  RUN8(0, 1) RUN8(0, 2) RUN8(0, 4) RUN8(1, 1) RUN8(1, 2) RUN8(1, 4)
  RUN16(0, 1) RUN16(0, 2) RUN16(0, 4) RUN16(1, 1) RUN16(1, 2) RUN16(1, 4)
  // alternating pair, as the smoother issues them (x every second sweep)
  { float t = time_it([&] { hipLaunchKernelGGL((mix8<0, 2>), g, b, 0, 0, n, pid, r, s, r2, s2, x); hipLaunchKernelGGL((mix8<1, 2>), g, b, 0, 0, n, pid, r2, s2, r, s, x); }, 20);
    printf("  pair (8 B/lane, unroll 2): %7.2f us per sweep  %7.0f GB/s\n", t * 1e3 / 2, (b0 + b1) / t / 1e6); }
  { float t = time_it([&] { hipLaunchKernelGGL((mix16<0, 2>), g, b, 0, 0, n / 2, (const uint32_t *)pid, (const double2 *)r, (const double2 *)s, (double2 *)r2, (double2 *)s2, (double2 *)x);
                            hipLaunchKernelGGL((mix16<1, 2>), g, b, 0, 0, n / 2, (const uint32_t *)pid, (const double2 *)r2, (const double2 *)s2, (double2 *)r, (double2 *)s, (double2 *)x); }, 20);
    printf("  pair (16 B/lane, unroll 2): %7.2f us per sweep  %7.0f GB/s\n", t * 1e3 / 2, (b0 + b1) / t / 1e6); }
  return 0;
}
