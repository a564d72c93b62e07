// Dispatch-gap probe (profiles/r05_tuning.md section 7): does a kernel with 1024-thread workgroups cost extra idle time before / after it in an in-order stream?
// Sequence per round: A (256 threads x 1024 workgroups, short) ; B (bs threads x nb workgroups, spins ~us microseconds) ; A.
// Prints the wall time per round for several shapes of B with the same number of waves; the differences are dispatch gaps.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
__global__ void short_kernel(double *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
__global__ void spin_kernel(double *p, long long ticks)
{
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0;
}
// MODE 0: plain stores of `per` doubles per thread, 1: agent-scope (sc1) atomic stores, 2: agent-scope loads + plain stores; then the spin
template <int MODE>
__global__ void work_kernel(double *q, int per, long long ticks)
{
  extern __shared__ double sh[];
  const long long t0 = wall_clock64();
  const size_t base = ((size_t)blockIdx.x * blockDim.x + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  double acc = 0.0;
  for (int k = 0; k < per; ++k) {
    double *a = q + base + (size_t)k * stride;
    if (MODE == 1) __hip_atomic_store(a, (double)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 2) { acc += __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *a = acc; }
    else *a = (double)k;
  }
  if (threadIdx.x == 0) sh[0] = acc;
  while (wall_clock64() - t0 < ticks) { }
}
int main()
{
  double *p; const int n = 1 << 18;
  CK(hipMalloc(&p, sizeof(double) * n)); CK(hipMemset(p, 0, sizeof(double) * n));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const long long ticks = 40 * 100;   // wall_clock64 runs at 100 MHz: 40 us
  const int shapes[][2] = {{1024, 253}, {512, 506}, {256, 1012}, {128, 2024}, {1024, 64}, {64, 4048}};
  for (auto &sh : shapes) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipStreamSynchronize(s));
      const auto t0 = std::chrono::steady_clock::now();
      const int rounds = 200;
      for (int r = 0; r < rounds; ++r) {
        hipLaunchKernelGGL(short_kernel, dim3(1024), dim3(256), 0, s, p, n);
        hipLaunchKernelGGL(spin_kernel, dim3(sh[1]), dim3(sh[0]), 0, s, p, ticks);
        hipLaunchKernelGGL(short_kernel, dim3(1024), dim3(256), 0, s, p, n);
      }
      CK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
      if (rep == 1) std::printf("B = %4d threads x %4d workgroups: %.2f us per round (A ; B(40 us spin) ; A)\n", sh[0], sh[1], us);
    }
  }
  // what timing B costs: nothing / hipEventRecord before and after / hipExtLaunchKernel with start and stop events (the kernel's own
  // dispatch timestamps, no marker packets)
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
      double us = 0.0, ev_us = 0.0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        const int rounds = 200;
        ev_us = 0.0;
        for (int r = 0; r < rounds; ++r) {
          hipLaunchKernelGGL(short_kernel, dim3(1024), dim3(256), 0, s, p, n);
          if (mode == 1) CK(hipEventRecord(e0, s));
          if (mode == 2) { long long tk = ticks; void *args[2] = {&p, &tk}; CK(hipExtLaunchKernel((const void *)spin_kernel, dim3(253), dim3(1024), args, 0, s, e0, e1, 0)); }
          else hipLaunchKernelGGL(spin_kernel, dim3(253), dim3(1024), 0, s, p, ticks);
          if (mode == 1) CK(hipEventRecord(e1, s));
          hipLaunchKernelGGL(short_kernel, dim3(1024), dim3(256), 0, s, p, n);
          if (mode && (r % 50) == 49) { CK(hipStreamSynchronize(s)); float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ev_us = ms * 1e3; }
        }
        CK(hipStreamSynchronize(s));
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
      }
      std::printf("timing mode %d (0 none, 1 hipEventRecord pair, 2 hipExtLaunchKernel events): %.2f us per round, events say %.2f us for the 40 us kernel\n", mode, us, ev_us);
    }
  }
  // the same with a B that touches memory like a one-launch smoothing pass: 4 vectors of 2.6e5 doubles (8 MB), 100 KB of dynamic LDS
  double *q; const size_t nq = (size_t)253 * 1024 * 32;
  CK(hipMalloc(&q, sizeof(double) * nq)); CK(hipMemset(q, 0, sizeof(double) * nq));
  CK(hipFuncSetAttribute((const void *)work_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  CK(hipFuncSetAttribute((const void *)work_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  CK(hipFuncSetAttribute((const void *)work_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  for (int mode = 0; mode < 3; ++mode)
    for (int per : {0, 4, 32})
      for (int lds : {0, 100 * 1024}) {
        double us = 0.0;
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipStreamSynchronize(s));
          const auto t0 = std::chrono::steady_clock::now();
          const int rounds = 200;
          for (int r = 0; r < rounds; ++r) {
            hipLaunchKernelGGL(short_kernel, dim3(1024), dim3(256), 0, s, p, n);
            if (mode == 0) hipLaunchKernelGGL(work_kernel<0>, dim3(253), dim3(1024), lds, s, q, per, ticks);
            else if (mode == 1) hipLaunchKernelGGL(work_kernel<1>, dim3(253), dim3(1024), lds, s, q, per, ticks);
            else hipLaunchKernelGGL(work_kernel<2>, dim3(253), dim3(1024), lds, s, q, per, ticks);
            hipLaunchKernelGGL(short_kernel, dim3(1024), dim3(256), 0, s, p, n);
          }
          CK(hipStreamSynchronize(s));
          us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
        }
        std::printf("B = 1024 x 253, mode %d (0 plain stores, 1 sc1 stores, 2 sc1 loads + stores), %2d doubles per thread (%.1f MB), LDS %3d KB: %.2f us per round\n",
                    mode, per, per * 253.0 * 1024 * 8 / 1e6, lds / 1024, us);
      }
  return 0;
}
