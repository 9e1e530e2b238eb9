// Micro-benchmark (diagnostic, not part of the product library): issue rates of the VALU instructions the GroupNorm + SiLU transform and
// the softmax loops are made of, on THIS device: independent chains of one instruction kind per wave, 1 and 2 waves per SIMD.
// Reported: wave-instructions per SIMD per microsecond and the cycles per wave-instruction at the clock a v_fma_f32 stream implies
// (a 64-lane v_fma_f32 issues in 4 cycles on a 16-lane SIMD).
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -o scripts/micro/valu_rates scripts/micro/valu_rates.hip ; run: ./scripts/micro/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(int iters, float* sink) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.001f * (threadIdx.x + i);
  unsigned h[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = 0x38003800u + threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {   // eight independent chains: latency never limits
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[i]));
        if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        if (KIND == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
        if (KIND == 3) asm volatile("v_exp_f16 %0, %0" : "+v"(h[i]));
        if (KIND == 4) asm volatile("v_pk_fma_f16 %0, %0, %0, %0" : "+v"(h[i]));
        if (KIND == 5) asm volatile("v_fma_mix_f32 %0, %1, %0, %0 op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(h[i]));
        if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&v[i & 6])));
        if (KIND == 7) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h[i]) : "v"(v[i]));
        if (KIND == 8) asm volatile("v_rsq_f32 %0, %0" : "+v"(v[i]));
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i] + (float)h[i];
  if (s == 12345.678f) sink[0] = s;
}

template <int KIND>
double run(int wgs_per_cu, int cus, float* sink) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000;
  hipLaunchKernelGGL(k<KIND>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, 10, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<KIND>, dim3(cus * wgs_per_cu), dim3(256), 0, 0, iters, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // wave-instructions per SIMD: iters * 64 per wave, wgs_per_cu waves per SIMD (4 waves per workgroup, one per SIMD)
  return (double)iters * 64.0 * wgs_per_cu / (ms * 1e3);   // per SIMD per microsecond
}

int main() {
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  float* sink;
  CK(hipMalloc(&sink, 4));
  const char* names[9] = {"v_fma_f32", "v_exp_f32", "v_rcp_f32", "v_exp_f16", "v_pk_fma_f16", "v_fma_mix_f32", "v_pk_fma_f32", "v_cvt_f16_f32", "v_rsq_f32"};
  for (int w = 1; w <= 2; ++w) {
    double r[9];
    r[0] = run<0>(w, cus, sink); r[1] = run<1>(w, cus, sink); r[2] = run<2>(w, cus, sink); r[3] = run<3>(w, cus, sink); r[4] = run<4>(w, cus, sink);
    r[5] = run<5>(w, cus, sink); r[6] = run<6>(w, cus, sink); r[7] = run<7>(w, cus, sink); r[8] = run<8>(w, cus, sink);
    for (int i = 0; i < 9; ++i)
      printf("%d wave(s)/SIMD  %-14s %8.1f wave-instr / SIMD / us   = %5.2f cycles each if v_fma_f32 takes 4\n", w, names[i], r[i], 4.0 * r[0] / r[i]);
  }
  return 0;
}
