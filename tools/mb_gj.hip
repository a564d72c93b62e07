// tools/mb_gj.hip -- where does a pass of the blocked Gauss-Jordan trailing update go?  Diagnostic only (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o build_tools/mbgj tools/mb_gj.hip && build_tools/mbgj [n]
// Times the product kernels on an n x n matrix (panel data arbitrary: timing only) and access-pattern ablations of the update.
#include "../gridapsolvers.jl_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gmg;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename F> float time_it(F f, int reps = 5)
{
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

// A[i][j] = A[i][j] * s in the MFMA output pattern: wave = RT x 16 rows, 64 columns; register a of lane (lk, lr) is row lk + 4 a, column lr
template <int RT, bool WR>
__global__ __launch_bounds__(256) void tile_rw_kernel(int n, int64_t lda, double *__restrict__ A, double s, double *__restrict__ sink)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ti = blockIdx.y * (RT * 32) + (wave >> 1) * (RT * 16);
  const int tj = blockIdx.x * 128 + (wave & 1) * 64;
  const int lr = lane & 15, lk = lane >> 4;
  double keep = 0.0;
  for (int c = 0; c < 4; ++c) {
    const int j = tj + 16 * c + lr;
    if (j >= n) continue;
    for (int rt = 0; rt < RT; ++rt)
      for (int a = 0; a < 4; ++a) {
        const int i = ti + 16 * rt + lk + 4 * a;
        if (i >= n) continue;
        const double v = A[(size_t)i * lda + j] * s;
        if (WR) A[(size_t)i * lda + j] = v; else keep += v;
      }
  }
  if (!WR && keep == 12345.678) sink[0] = keep;
}
// the same bytes, row-contiguous: a wave owns 64 consecutive columns x ROWS rows, lane = column (512 B per row and instruction)
template <bool WR>
__global__ __launch_bounds__(256) void row_rw_kernel(int n, int64_t lda, double *__restrict__ A, double s, double *__restrict__ sink)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ti = blockIdx.y * 64 + (wave >> 1) * 32;
  const int j = blockIdx.x * 128 + (wave & 1) * 64 + lane;
  if (j >= n) return;
  double keep = 0.0;
  for (int q = 0; q < 32; ++q) {
    const int i = ti + q;
    if (i >= n) break;
    const double v = A[(size_t)i * lda + j] * s;
    if (WR) A[(size_t)i * lda + j] = v; else keep += v;
  }
  if (!WR && keep == 12345.678) sink[0] = keep;
}
// flat stream over the same buffer
__global__ void flat_rw_kernel(int64_t total, double *__restrict__ A, double s)
{
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) A[i] *= s;
}


// f64 MFMA rate: NCH independent accumulator chains, 128 MFMAs per wave and outer iteration, operands in registers
template <int NCH>
__global__ __launch_bounds__(256) void mfma_rate_kernel(int iters, double *__restrict__ sink)
{
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc[NCH];
  for (int q = 0; q < NCH; ++q) acc[q] = d4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 128 / NCH; ++s)
#pragma unroll
      for (int q = 0; q < NCH; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  double t = 0.0;
  for (int q = 0; q < NCH; ++q) t += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  if (t == 12345.678) sink[0] = t;
}
// f64 FMA rate on the vector ALU, same flop count per wave and iteration (128 x 2048 flops = 128 x 16 fused multiply-adds per lane)
__global__ __launch_bounds__(256) void fma_rate_kernel(int iters, double *__restrict__ sink)
{
  double acc[16];
  for (int q = 0; q < 16; ++q) acc[q] = q;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int s = 0; s < 128; ++s)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = __builtin_fma(acc[q], a, b);
  double t = 0.0;
  for (int q = 0; q < 16; ++q) t += acc[q];
  if (t == 12345.678) sink[0] = t;
}

// v_mul_f64 / v_add_f64 rates (the sparse kernels multiply and add separately: -ffp-contract=off, the reference's roundings)
template <int MODE>   // 0: 16 independent mul chains, 1: 16 independent add chains, 2: mul feeding ONE dependent add chain (a row sum)
__global__ __launch_bounds__(256) void muladd_rate_kernel(int iters, double *__restrict__ sink)
{
  double acc[16];
  for (int q = 0; q < 16; ++q) acc[q] = 1.0 + q * 1e-3;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  double sum = 0.0;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int s = 0; s < 128; ++s)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        if (MODE == 0) acc[q] = acc[q] * a;
        else if (MODE == 1) acc[q] = acc[q] + b;
        else { sum = sum + acc[q] * a; }
      }
  double t = sum;
  for (int q = 0; q < 16; ++q) t += acc[q];
  if (t == 12345.678) sink[0] = t;
}

int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 29791;
  const int64_t lda = (((int64_t)n + 127) / 128) * 128, nr = (((int64_t)n + 127) / 128) * 128;
  double *A, *R, *C, *Cp, *Pinv, *sink; int *bad;
  CK(hipMalloc(&A, sizeof(double) * (size_t)nr * lda)); CK(hipMemset(A, 0, sizeof(double) * (size_t)nr * lda));
  CK(hipMalloc(&R, sizeof(double) * 64 * lda)); CK(hipMemset(R, 0, sizeof(double) * 64 * lda));
  CK(hipMalloc(&C, sizeof(double) * 64 * (size_t)nr)); CK(hipMemset(C, 0, sizeof(double) * 64 * (size_t)nr));
  CK(hipMalloc(&Cp, sizeof(double) * 64 * (size_t)nr)); CK(hipMemset(Cp, 0, sizeof(double) * 64 * (size_t)nr));
  CK(hipMalloc(&Pinv, sizeof(double) * 64 * 64)); CK(hipMemset(Pinv, 0, sizeof(double) * 64 * 64));
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&bad, 64)); CK(hipMemset(bad, 0, 64));
  const double gb = 16.0 * (double)n * (double)n / 1e9;
  printf("n %d lda %lld: read + write of the matrix = %.2f GB\n", n, (long long)lda, gb);
  const dim3 g64((n + 127) / 128, (n + 63) / 64), g32((n + 127) / 128, (n + 31) / 32);
  float t;
  t = time_it([&] { hipLaunchKernelGGL(flat_rw_kernel, dim3(256 * 16), dim3(256), 0, 0, (int64_t)n * lda, A, 1.0); });
  printf("flat read-modify-write stream          %8.3f ms  %6.2f TB/s\n", t, gb / t);
  t = time_it([&] { hipLaunchKernelGGL((row_rw_kernel<true>), g64, dim3(256), 0, 0, n, lda, A, 1.0, sink); });
  printf("row-contiguous tiles (32 x 64 / wave)  %8.3f ms  %6.2f TB/s\n", t, gb / t);
  t = time_it([&] { hipLaunchKernelGGL((row_rw_kernel<false>), g64, dim3(256), 0, 0, n, lda, A, 1.0, sink); });
  printf("  ... read only                        %8.3f ms  %6.2f TB/s (of the read half)\n", t, 0.5 * gb / t);
  t = time_it([&] { hipLaunchKernelGGL((tile_rw_kernel<2, true>), g64, dim3(256), 0, 0, n, lda, A, 1.0, sink); });
  printf("MFMA-pattern tiles, 32 x 64 per wave   %8.3f ms  %6.2f TB/s\n", t, gb / t);
  t = time_it([&] { hipLaunchKernelGGL((tile_rw_kernel<2, false>), g64, dim3(256), 0, 0, n, lda, A, 1.0, sink); });
  printf("  ... read only                        %8.3f ms  %6.2f TB/s (of the read half)\n", t, 0.5 * gb / t);
  t = time_it([&] { hipLaunchKernelGGL((tile_rw_kernel<1, true>), g32, dim3(256), 0, 0, n, lda, A, 1.0, sink); });
  printf("MFMA-pattern tiles, 16 x 64 per wave   %8.3f ms  %6.2f TB/s\n", t, gb / t);
  {
    const int64_t nst = (int64_t)(((n + 127) / 128 + GJ_SX - 1) / GJ_SX) * ((n + 1023) / 1024);
    const dim3 t2((unsigned)(((nst + 7) / 8) * 8 * GJ_SX * 16)), t4((unsigned)(((nst + 7) / 8) * 8 * GJ_SX * 8));
    t = time_it([&] { hipLaunchKernelGGL((gj_update64_kernel<2>), t2, dim3(256), 0, 0, n, lda, 64, 64, A, Pinv, R, C, Cp); });
    printf("gj_update64_kernel<2> (32 x 64 / wave) %8.3f ms  %6.2f TB/s  %6.1f TFLOP/s\n", t, gb / t, 2.0 * n * (double)n * 64 / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((gj_update64_kernel<4>), t4, dim3(256), 0, 0, n, lda, 64, 64, A, Pinv, R, C, Cp); });
    printf("gj_update64_kernel<4> (64 x 64 / wave) %8.3f ms  %6.2f TB/s  %6.1f TFLOP/s\n", t, gb / t, 2.0 * n * (double)n * 64 / t / 1e9);
  }
  {
    const dim3 tiles((n + 63) / 64, (n + 63) / 64);
    t = time_it([&] { hipLaunchKernelGGL(gj_update_mfma_kernel, tiles, dim3(256), 0, 0, n, 64, 32, A, Pinv, R, C, Cp); });
    printf("gj_update_mfma_kernel (32-wide, lda=n) %8.3f ms  %6.2f TB/s  %6.1f TFLOP/s\n", t, gb / t, 2.0 * n * (double)n * 32 / t / 1e9);
  }
  {
    const int iters = 64;
    const double fl = 256.0 * 16 * 4 * iters * 128 * 2048.0;   // 256 CUs x 16 workgroups x 4 waves
    t = time_it([&] { hipLaunchKernelGGL((mfma_rate_kernel<1>), dim3(256 * 16), dim3(256), 0, 0, iters, sink); });
    printf("v_mfma_f64_16x16x4, 1 chain / wave     %8.3f ms  %6.1f TFLOP/s\n", t, fl / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((mfma_rate_kernel<2>), dim3(256 * 16), dim3(256), 0, 0, iters, sink); });
    printf("v_mfma_f64_16x16x4, 2 chains / wave    %8.3f ms  %6.1f TFLOP/s\n", t, fl / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((mfma_rate_kernel<8>), dim3(256 * 16), dim3(256), 0, 0, iters, sink); });
    printf("v_mfma_f64_16x16x4, 8 chains / wave    %8.3f ms  %6.1f TFLOP/s\n", t, fl / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL(fma_rate_kernel, dim3(256 * 16), dim3(256), 0, 0, iters, sink); });
    printf("v_fma_f64, 16 chains / lane            %8.3f ms  %6.1f TFLOP/s\n", t, fl / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((muladd_rate_kernel<0>), dim3(256 * 16), dim3(256), 0, 0, iters, sink); });
    printf("v_mul_f64, 16 chains / lane            %8.3f ms  %6.1f Tinstr-lanes/s (x 1e12)\n", t, fl / 2 / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((muladd_rate_kernel<1>), dim3(256 * 16), dim3(256), 0, 0, iters, sink); });
    printf("v_add_f64, 16 chains / lane            %8.3f ms  %6.1f\n", t, fl / 2 / t / 1e9);
    t = time_it([&] { hipLaunchKernelGGL((muladd_rate_kernel<2>), dim3(256 * 16), dim3(256), 0, 0, iters, sink); });
    printf("mul + dependent add (one row sum)      %8.3f ms  %6.1f pairs\n", t, fl / 2 / t / 1e9);
  }
  return 0;
}
