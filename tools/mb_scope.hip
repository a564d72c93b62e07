// tools/mb_scope.hip -- what do memory scopes cost on the gathers of a sweep?
// Diagnostic only (not part of the product).  A vector of (n-1)^3 doubles is written by one kernel (plain or agent-scope stores) and
// gathered by the next with the nine run offsets of a 27-point operator, two rows per lane (16-byte loads), with
//   L = 0 plain loads ; 1 sc0 (workgroup scope) ; 2 sc1 (agent scope) ; 3 sc0 sc1 (system scope) ; 4 nt
// The question behind it: do agent-scope loads of data written by OTHER XCDs still hit in the L2 (a persistent multi-sweep kernel on a
// big level needs coherent gathers every sweep), or does every one of them go to the memory side?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mbscope tools/mb_scope.hip && /tmp/mbscope [cells]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int L>
__device__ __forceinline__ d2 ld2(const double *p)
{
  d2 r;
  if (L == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  else if (L == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(r) : "v"(p) : "memory");
  else if (L == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(r) : "v"(p) : "memory");
  else if (L == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(r) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r) : "v"(p) : "memory");
  return r;
}

template <int S>
__global__ __launch_bounds__(256) void write_kernel(int64_t n, double *v, double f)
{
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2; i + 1 < n; i += (int64_t)gridDim.x * 512) {
    d2 w = {f * (double)i, f * (double)(i + 1)};
    if (S == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(v + i), "v"(w) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(v + i), "v"(w) : "memory");
  }
}

struct Offs { int o[9]; };

// chunked like the product's pair sweep: workgroup -> contiguous slices, XCD-contiguous eighths (blockIdx & 7 = XCD, round-robin dispatch)
template <int L>
__global__ __launch_bounds__(256) void gather_kernel(int64_t n, const double *__restrict__ v, double *__restrict__ out, Offs offs, int nslices)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3, per = nwg >> 3;
  const int blk = xcd * per + q;                             // contiguous eighth per XCD
  const int lo = (int)((int64_t)nslices * blk / nwg), hi = (int)((int64_t)nslices * (blk + 1) / nwg);
  for (int s = lo + wave; s < hi; s += 4) {
    const int64_t row = (int64_t)s * 126 + 2 * lane;
    d2 acc = {0.0, 0.0};
    d2 g[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) {
      int64_t c = row + offs.o[r];
      c = c < 0 ? 0 : (c > n - 2 ? n - 2 : c);
      g[r] = ld2<L>(v + c);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < 9; ++r) { acc.x += g[r].x; acc.y += g[r].y; }
    if (row + 1 < n && lane < 63) { out[row] = acc.x; out[row + 1] = acc.y; }
  }
}

template <typename F> float time_it(F f, int reps = 30)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 4; ++i) f();
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
  Offs offs;
  int k = 0;
  for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) offs.o[k++] = dz * m * m + dy * m - 1;
  double *v, *out;
  CK(hipMalloc(&v, sizeof(double) * (N + 64)));
  CK(hipMalloc(&out, sizeof(double) * (N + 64)));
  CK(hipMemset(v, 0, sizeof(double) * (N + 64)));
  const int nslices = (int)((N + 125) / 126);
  const int nwg = 1024;
  printf("cells %d rows %lld (%.1f MB per vector), %d slices of 126 rows, %d workgroups\n", nc, (long long)N, N * 8e-6, nslices, nwg);
  const char *names[5] = {"plain", "sc0", "sc1", "sc0 sc1", "nt"};
  for (int S = 0; S < 2; ++S) {
    // time of the writer alone
    float tw = time_it([&] { if (S == 0) hipLaunchKernelGGL(write_kernel<0>, dim3(1024), dim3(256), 0, 0, N, v, 1.0); else hipLaunchKernelGGL(write_kernel<1>, dim3(1024), dim3(256), 0, 0, N, v, 1.0); });
    printf("writer (%s stores): %.2f us\n", S == 0 ? "plain" : "sc1", tw * 1e3);
    for (int L = 0; L < 5; ++L) {
      auto both = [&] {
        if (S == 0) hipLaunchKernelGGL(write_kernel<0>, dim3(1024), dim3(256), 0, 0, N, v, 1.0); else hipLaunchKernelGGL(write_kernel<1>, dim3(1024), dim3(256), 0, 0, N, v, 1.0);
        switch (L) {
        case 0: hipLaunchKernelGGL(gather_kernel<0>, dim3(nwg), dim3(256), 0, 0, N, v, out, offs, nslices); break;
        case 1: hipLaunchKernelGGL(gather_kernel<1>, dim3(nwg), dim3(256), 0, 0, N, v, out, offs, nslices); break;
        case 2: hipLaunchKernelGGL(gather_kernel<2>, dim3(nwg), dim3(256), 0, 0, N, v, out, offs, nslices); break;
        case 3: hipLaunchKernelGGL(gather_kernel<3>, dim3(nwg), dim3(256), 0, 0, N, v, out, offs, nslices); break;
        default: hipLaunchKernelGGL(gather_kernel<4>, dim3(nwg), dim3(256), 0, 0, N, v, out, offs, nslices); break;
        }
      };
      const float t = time_it(both);
      printf("  writer + gather with %-8s loads: %.2f us  => gather %.2f us (9 x 16 B per lane pair: %.0f MB requested)\n", names[L], t * 1e3, (t - tw) * 1e3, 9.0 * N * 8e-6);
    }
  }
  return 0;
}
