// Micro-benchmark (diagnostic, not part of the product library): a conv / GEMM weight stream that comes from HBM -- not from L2 as in
// ldsdma_feed.hip -- through an LDS ring of `STAGES` slots filled by `buffer_load_dwordx4 ... lds`, as a function of the ring depth and of
// the number of workgroups.  This is the access shape of the UNet's small-map layers (8 x 8 / 16 x 16 maps, and every layer at batch 1):
// a workgroup walks `steps` K-steps, each needing one 16-KiB weight slice (128 rows x 128 B = a [BN = 128][64] fp16 slice) that nobody
// else has touched, waits (counted vmcnt) for the oldest slice, passes a barrier, "computes" for `work` s_sleep units, passes a barrier.
// With two slots (the product kernels before round 6) a step costs one HBM round trip whatever the chip could stream.
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -o hbm_ring scripts/micro/hbm_ring.hip ; run: ./hbm_ring
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// TILED: the slice of a step is one contiguous 16 KiB block (a wave-instruction fetches one contiguous KiB) instead of 128 row segments of 128 B
template <int STAGES, int NT, int TILED = 0>
__global__ __launch_bounds__(256) void ring_kernel(const char* base, long long wg_stride, int pitch, int steps, int work, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int PIECES = 4;   // 4 KiB per wave and slice, 16 KiB per workgroup
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* mine = base + (long long)blockIdx.x * wg_stride;     // this workgroup's 128 weight rows (pitch bytes each)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, 128 * pitch, 0x00020000);
  int voff[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) voff[i] = TILED ? (wave * PIECES + i) * 1024 + lane * 16 : ((wave * PIECES + i) * 8 + (lane >> 3)) * pitch + (lane & 7) * 16;
  auto issue = [&](int step, int slot) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < PIECES; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t*)(smem + slot * 16384 + (wave * PIECES + i) * 1024), 16, voff[i], TILED ? step * 16384 : step * 128, 0, NT ? 2 : 0);
#endif
  };
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) if (s < steps) issue(s, s);
  unsigned acc = 0;
  for (int step = 0; step < steps; ++step) {
    if (step + STAGES - 1 < steps) {
      issue(step + STAGES - 1, (step + STAGES - 1) % STAGES);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 1) * PIECES) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (tail: conservative)
    __builtin_amdgcn_s_barrier();
    acc += *reinterpret_cast<volatile unsigned*>(smem + (step % STAGES) * 16384 + threadIdx.x * 4);
    for (int w = 0; w < work; ++w) __builtin_amdgcn_s_sleep(2);   // ~128 cycles each: stands in for the MFMAs of the step (a 64 x 128 x 64 step is ~256)
    __builtin_amdgcn_s_barrier();
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// The same bytes by plain global_load_dwordx4 into registers, UNR loads in flight per thread (no LDS, no barrier): what a small GEMM that takes its
// MFMA fragments straight from memory would see.  A workgroup reads `bytes` contiguous bytes of its own.
template <int UNR>
__global__ __launch_bounds__(256) void direct_kernel(const char* base, long long wg_stride, int bytes, unsigned* sink) {
  const uint4* p = reinterpret_cast<const uint4*>(base + (long long)blockIdx.x * wg_stride) + threadIdx.x;
  const int n = bytes / 4096;   // loads per thread
  unsigned acc = 0;
  for (int i = 0; i < n; i += UNR) {
    uint4 v[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) v[j] = p[(i + j) * 256];
#pragma unroll
    for (int j = 0; j < UNR; ++j) acc += v[j].x ^ v[j].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

static char* g_buf;
static size_t g_bytes;
static unsigned* g_sink;
static size_t g_cursor = 0;

static bool g_warm = false;   // true: every repetition reads the SAME region (<= 60 MB: it stays in the 256 MiB Infinity Cache between launches)
template <int STAGES, int NT, int TILED = 0>
void run(int grid, int steps, int work) {
  const int pitch = steps * 128;                        // a [128][K] weight slice per workgroup, K = 64 * steps
  const long long wg_stride = (long long)128 * pitch;
  const size_t launch_bytes = (size_t)grid * wg_stride;
  auto k = ring_kernel<STAGES, NT, TILED>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, STAGES * 16384));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f, sum = 0.f;
  const int reps = 6;
  for (int r = 0; r < reps; ++r) {
    if (g_cursor + launch_bytes > g_bytes) g_cursor = 0;   // every launch reads a region no cache holds (the buffer is 3 GiB, the MALL 256 MiB)
    const char* src = g_buf + g_cursor;
    if (!g_warm) g_cursor += launch_bytes;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), STAGES * 16384, 0, src, wg_stride, pitch, steps, work, g_sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) { sum += ms; if (ms < best) best = ms; }
  }
  const double us = sum / (reps - 1) * 1e3;
  printf("%s%s wgs %4d  steps %3d  work %d  stages %d  nt %d : %7.1f us per launch  (best %6.1f)  %6.2f TB/s  %5.2f us per step\n", g_warm ? "warm " : "", TILED ? "tiled" : "rows ", grid, steps, work, STAGES, NT, us,
         best * 1e3, launch_bytes / (us * 1e-6) / 1e12, us / steps);
}

template <int UNR>
void run_direct(int grid, int kib) {
  const long long wg_stride = (long long)kib * 1024;
  const size_t launch_bytes = (size_t)grid * wg_stride;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float sum = 0.f;
  const int reps = 6;
  for (int r = 0; r < reps; ++r) {
    if (g_cursor + launch_bytes > g_bytes) g_cursor = 0;
    const char* src = g_buf + g_cursor;
    if (!g_warm) g_cursor += launch_bytes;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(direct_kernel<UNR>, dim3(grid), dim3(256), 0, 0, src, wg_stride, kib * 1024, g_sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r) sum += ms;
  }
  const double us = sum / (reps - 1) * 1e3;
  printf("%sdirect wgs %4d  %4d KiB per workgroup  %2d loads in flight per thread : %7.1f us per launch  %6.2f TB/s  %6.1f GB/s per workgroup\n", g_warm ? "warm " : "", grid, kib, UNR, us,
         launch_bytes / (us * 1e-6) / 1e12, wg_stride / (us * 1e-6) / 1e9);
}

template <int NT>
void sweep(int grid, int steps, int work) {
  run<2, NT>(grid, steps, work); run<3, NT>(grid, steps, work); run<4, NT>(grid, steps, work); run<6, NT>(grid, steps, work); run<8, NT>(grid, steps, work);
}

int main() {
  g_bytes = (size_t)3 << 30;
  CK(hipMalloc(&g_buf, g_bytes)); CK(hipMemset(g_buf, 1, g_bytes)); CK(hipMalloc(&g_sink, 64));
  CK(hipDeviceSynchronize());
  // conv3x3 1280 -> 1280 at 8 x 8, batch 1: 10 channel tiles x split-K 8 = 80 workgroups of ~23 steps; batch 8: 400 workgroups of 36 steps
  for (int work : {0, 4}) {
    sweep<0>(80, 24, work);
    sweep<0>(256, 36, work);
    sweep<0>(400, 36, work);
  }
  sweep<1>(80, 24, 4);
  sweep<1>(400, 36, 4);
  // GEMM at batch 1 (64 x 64 tiles: here 16-KiB slices too): 20 tiles x 20 steps
  sweep<0>(20, 20, 2);
  // the same 30 MB (1920 slices) cut finer: more workgroups, fewer steps each
  for (int work : {0, 1}) {
    run<2, 0>(80, 24, work); run<4, 0>(80, 24, work);
    run<2, 0>(160, 12, work); run<4, 0>(160, 12, work);
    run<2, 0>(240, 8, work); run<4, 0>(240, 8, work);
    run<2, 0>(480, 4, work); run<4, 0>(480, 4, work);
    run<2, 0>(960, 2, work);
    // contiguous 16-KiB slices (tiled weights)
    run<2, 0, 1>(80, 24, work); run<4, 0, 1>(80, 24, work); run<8, 0, 1>(80, 24, work);
    run<2, 0, 1>(240, 8, work); run<4, 0, 1>(240, 8, work);
    run<2, 0, 1>(480, 4, work); run<4, 0, 1>(480, 4, work);
    run<2, 0, 1>(400, 36, work); run<4, 0, 1>(400, 36, work);
    run<2, 1, 1>(80, 24, work); run<4, 1, 1>(240, 8, work);
  }
  // the same launches with the region resident in the Infinity Cache (what a weight prefetcher running ahead of the layers would buy)
  g_warm = true;
  for (int work : {0, 2}) {
    run<2, 0>(80, 24, work); run<4, 0>(80, 24, work); run<2, 0>(160, 12, work); run<2, 0>(240, 8, work); run<2, 0>(20, 20, work); run<2, 0>(320, 5, work);
  }
  g_warm = false;
  for (int work : {0, 2}) { run<2, 0>(20, 20, work); run<2, 0>(320, 5, work); }
  // plain register loads, all of a workgroup's bytes as deep in flight as the registers allow: 80 KiB = a 64 x 64 tile at K = 320, 320 KiB at K = 1280
  for (int w = 0; w < 2; ++w) {
    g_warm = w != 0;
    run_direct<4>(320, 80); run_direct<8>(320, 80); run_direct<16>(320, 80);
    run_direct<8>(80, 320); run_direct<16>(80, 320); run_direct<16>(20, 320); run_direct<16>(160, 160); run_direct<16>(640, 40);
  }
  return 0;
}
