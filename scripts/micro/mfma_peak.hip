// Micro-benchmark (diagnostic, not part of the product library): what dense fp16 MFMA rate does THIS device sustain when nothing but
// the matrix pipe works?  Every wave issues back-to-back v_mfma_f32_16x16x32_f16 on NACC independent accumulators (operands in
// registers, no LDS, no memory), for launches of different lengths and 1 / 2 / 4 waves per SIMD.  Next to the rate the kernel reports
// the shader clock it saw (s_memtime cycles of wave 0 of workgroup 0 / event time): the roofline peak bench.py quotes (2.5 PFLOP/s) is
// the 2.4 GHz figure, and the conv / GEMM kernels run at whatever clock the power management leaves under a matrix load.
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -o scripts/micro/mfma_peak scripts/micro/mfma_peak.hip ; run: ./scripts/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// acc[NA][NB] = mfma(A[a], B[b], acc[a][b]) in the order of a conv / GEMM wave tile; after every MFMA: NOP x `s_nop 0`, NVALU independent
// v_fma_f32.  ORDER 0: a outer, b inner (B changes every instruction); 1: every accumulator twice in a row (dependent pairs).
template <int NA, int NB, int NOP, int NVALU, int ORDER>
__global__ __launch_bounds__(256) void mfma_kernel(int iters, float* sink, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  f16x8 A[NA], B[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) A[i][j] = (_Float16)(0.001f * ((lane * 7 + j * 3 + i * 11) % 97) - 0.05f);
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) B[i][j] = (_Float16)(0.002f * ((lane * 5 + j * 13 + i * 17) % 89) - 0.09f);
  f32x4 acc[NA][NB];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float v[4] = {0.001f * lane, 1.0f, 0.5f, 0.25f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    auto one = [&](int a, int b) __attribute__((always_inline)) {
      acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[a], B[b], acc[a][b], 0, 0, 0);
#pragma unroll
      for (int n = 0; n < NOP; ++n) asm volatile("s_nop 0");
#pragma unroll
      for (int n = 0; n < NVALU; ++n) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n & 3]) : "v"(v[(n + 1) & 3]));
    };
    if (ORDER == 2) {   // a row of NB accumulators, then the same row again (dependent at distance NB: the two k-halves of a step)
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
#pragma unroll
          for (int b = 0; b < NB; ++b) one(a, b);
    } else if (ORDER == 3) {   // the whole tile, then the whole tile again (distance NA*NB: what the kernels do today)
#pragma unroll
      for (int rep = 0; rep < 2; ++rep)
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b) one(a, b);
    } else {
#pragma unroll
      for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
          for (int rep = 0; rep < (ORDER == 1 ? 2 : 1); ++rep) one(a, b);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x < 8) clk[blockIdx.x] = t1 - t0;   // workgroups 0..7 land on the 8 XCDs
}

template <int NA, int NB, int NOP, int NVALU, int ORDER>
void run(int wg_per_cu, int iters, float* sink, unsigned long long* clk, int cus) {
  auto k = mfma_kernel<NA, NB, NOP, NVALU, ORDER>;
  const int grid = cus * wg_per_cu;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, 64, sink, clk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, iters, sink, clk);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c[8];
  CK(hipMemcpy(c, clk, sizeof(c), hipMemcpyDeviceToHost));
  unsigned long long cmin = c[0], cmax = c[0];
  for (int i = 1; i < 8; ++i) { cmin = c[i] < cmin ? c[i] : cmin; cmax = c[i] > cmax ? c[i] : cmax; }
  const double per_wave = (double)iters * NA * NB * (ORDER >= 1 ? 2 : 1);
  const double flops = (double)grid * 4 * per_wave * 16.0 * 16.0 * 32.0 * 2.0;
  // one 16x16x32 f16 MFMA occupies a SIMD's matrix pipe for 16 cycles at the data-sheet rate (2.5 PFLOP/s at 2.4 GHz)
  const double ghz_if_saturated = (double)wg_per_cu * per_wave * 16.0 / (ms * 1e-3) * 1e-9;
  printf("tile %dx%d order %d nop %d valu %d  wg/CU=%d  %8.3f ms  %7.1f TFLOP/s  (saturated pipe at %.2f GHz; s_memtime/time on the 8 XCDs %.2f..%.2f GHz)\n",
         NA, NB, ORDER, NOP, NVALU, wg_per_cu, ms, flops / (ms * 1e-3) * 1e-12, ghz_if_saturated, (double)cmin / (ms * 1e-3) * 1e-9, (double)cmax / (ms * 1e-3) * 1e-9);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  int mhz = 0;
  CK(hipDeviceGetAttribute(&mhz, hipDeviceAttributeClockRate, dev));
  printf("device %d: %d CUs, reported peak clock %.2f GHz -> dense fp16 16x16x32 peak %.0f TFLOP/s at that clock\n", dev, cus, mhz * 1e-6,
         cus * 4 * (16.0 * 16 * 32 * 2 / 16.0) * mhz * 1e3 * 1e-12);
  float* sink; unsigned long long* clk;
  CK(hipMalloc(&sink, 64)); CK(hipMalloc(&clk, 64));
  const int N = 1600000;   // MFMAs per wave and launch (~25-50 ms)
  for (int wg = 2; wg >= 1; --wg) {
    printf("-- %d wave(s) per SIMD\n", wg);
    run<4, 4, 0, 0, 3>(wg, N / 32, sink, clk, cus);   // tile, tile (today's order: k-half 0 of all accumulators, then k-half 1)
    run<4, 4, 0, 0, 2>(wg, N / 32, sink, clk, cus);   // row, same row
    run<4, 4, 0, 0, 1>(wg, N / 32, sink, clk, cus);   // pairs
    run<4, 4, 1, 0, 3>(wg, N / 32, sink, clk, cus);
    run<4, 4, 2, 0, 3>(wg, N / 32, sink, clk, cus);
    run<4, 4, 3, 0, 3>(wg, N / 32, sink, clk, cus);
    run<4, 4, 4, 0, 3>(wg, N / 32, sink, clk, cus);
    run<4, 4, 0, 1, 3>(wg, N / 32, sink, clk, cus);
    run<4, 4, 0, 2, 3>(wg, N / 32, sink, clk, cus);
    run<4, 4, 2, 0, 2>(wg, N / 32, sink, clk, cus);
    run<4, 4, 2, 0, 1>(wg, N / 32, sink, clk, cus);
    run<4, 4, 0, 0, 3>(wg, N / 32, sink, clk, cus);   // repeat of the first line (drift within the run)
  }
  run<4, 4, 0, 0, 3>(2, N / 32 * 4, sink, clk, cus);   // 4x longer launches
  run<4, 4, 0, 0, 1>(2, N / 32 * 4, sink, clk, cus);
  return 0;
}
