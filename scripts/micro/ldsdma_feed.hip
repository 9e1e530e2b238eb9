// Micro-benchmark (diagnostic, not part of the product library): how fast can a CU pull operand slices from L2 into LDS with
// `buffer_load_dwordx4 ... lds` in the access shape of the GEMM / conv kernels (one wave-instruction = 1 KiB = SEG-byte segments
// of 1024/SEG different rows), as a function of the bytes in flight (ring depth) and the workgroups per CU?
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -o ldsdma_feed scripts/micro/ldsdma_feed.hip ; run: ./ldsdma_feed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(3))) void lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// Each workgroup (256 threads) streams `steps` slices of PIECES KiB per wave (4*PIECES KiB per slice) from a matrix of `rows` rows of
// `pitch` bytes that every workgroup shares (L2 resident), through a ring of STAGES slots; per step it waits (counted vmcnt) for
// the oldest slice, passes a barrier and issues the next one.  No compute: this is the feed ceiling of that loop shape.
template <int PIECES, int STAGES, int SEG, int NWV = 4>
__global__ __launch_bounds__(NWV * 64) void feed_kernel(const char* base, int rows, int pitch, int steps, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int LPS = SEG / 16;              // lanes per segment
  constexpr int RPI = 64 / LPS;              // rows per wave-instruction
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, rows * pitch, 0x00020000);
  int voff[PIECES];
  const int row0 = (blockIdx.x * 37) % (rows - NWV * PIECES * RPI);   // workgroups start at different rows of the shared matrix
#pragma unroll
  for (int i = 0; i < PIECES; ++i) voff[i] = (row0 + (wave * PIECES + i) * RPI + lane / LPS) * pitch + (lane % LPS) * 16;
  constexpr unsigned SLICE = NWV * PIECES * 1024;
  auto issue = [&](int step, int slot) {
    const int soff = (step * SEG) % (pitch - SEG + 1) & ~15;        // walk along the rows like a K loop
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass rejects this builtin and then silently drops the kernel stub
#pragma unroll
    for (int i = 0; i < PIECES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t*)(smem + slot * SLICE + (wave * PIECES + i) * 1024), 16, voff[i], soff, 0, 0);
#endif
  };
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue(s, s);
  unsigned acc = 0;
  for (int step = 0; step < steps; ++step) {
    if (step + STAGES - 1 < steps) {
      issue(step + STAGES - 1, (step + STAGES - 1) % STAGES);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 1) * PIECES) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc += *reinterpret_cast<volatile unsigned*>(smem + (step % STAGES) * SLICE + threadIdx.x * 4);   // touch the landed slice
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int PIECES, int STAGES, int SEG, int NWV = 4>
void run(const char* d, int rows, int pitch, unsigned* sink, int wg_per_cu) {
  const int steps = 4000, grid = 256 * wg_per_cu;
  const size_t smem = (size_t)STAGES * NWV * PIECES * 1024;
  auto k = feed_kernel<PIECES, STAGES, SEG, NWV>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k, dim3(grid), dim3(NWV * 64), smem, 0, d, rows, pitch, 200, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(grid), dim3(NWV * 64), smem, 0, d, rows, pitch, steps, sink);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)grid * steps * NWV * PIECES * 1024;
  printf("seg %4d B  slice %3d KiB  stages %d  waves/wg %d  wg/CU %d  in flight/CU %4zu KiB : %7.1f GB/s per CU  (%6.2f TB/s chip)  %.2f us/step\n", SEG, NWV * PIECES,
         STAGES, NWV, wg_per_cu, (size_t)(STAGES - 1) * NWV * PIECES * wg_per_cu, bytes / (ms * 1e-3) / 256 / 1e9, bytes / (ms * 1e-3) / 1e12, ms * 1e3 / steps);
}

int main() {
  const int rows = 4096, pitch = 2560;   // a 10 MB operand matrix shared by all workgroups (L2 / MALL resident), rows of K = 1280 fp16
  char* d; unsigned* sink;
  CK(hipMalloc(&d, (size_t)rows * pitch)); CK(hipMemset(d, 1, (size_t)rows * pitch)); CK(hipMalloc(&sink, 64));
  // GEMM 128x128 shape: 32 KiB slice (8 pieces per wave), 128-byte segments
  run<8, 2, 128>(d, rows, pitch, sink, 1); run<8, 2, 128>(d, rows, pitch, sink, 2);
  run<8, 3, 128>(d, rows, pitch, sink, 1); run<8, 4, 128>(d, rows, pitch, sink, 1);
  // GEMM 128x64 shape: 24 KiB slice
  run<6, 2, 128>(d, rows, pitch, sink, 2); run<6, 2, 128>(d, rows, pitch, sink, 3); run<6, 3, 128>(d, rows, pitch, sink, 2);
  // conv weight slice: 16 KiB
  run<4, 2, 128>(d, rows, pitch, sink, 2); run<4, 3, 128>(d, rows, pitch, sink, 2); run<4, 4, 128>(d, rows, pitch, sink, 2); run<4, 2, 128>(d, rows, pitch, sink, 4);
  // wider segments (BK = 128 / 512 per row piece)
  run<8, 2, 256>(d, rows, pitch, sink, 2); run<8, 2, 1024>(d, rows, pitch, sink, 2); run<8, 3, 1024>(d, rows, pitch, sink, 1);
  run<4, 2, 1024>(d, rows, pitch, sink, 4);
  // round 4: one 8-wave workgroup per CU (a 256 x 128 GEMM step = 48 KiB, 6 pieces per wave) against two 4-wave workgroups
  run<6, 2, 128, 8>(d, rows, pitch, sink, 1); run<6, 3, 128, 8>(d, rows, pitch, sink, 1); run<4, 2, 128, 8>(d, rows, pitch, sink, 1); run<8, 2, 128, 8>(d, rows, pitch, sink, 1);
  run<6, 2, 1024, 8>(d, rows, pitch, sink, 1);
  return 0;
}
