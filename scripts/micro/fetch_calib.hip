// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access forms the dataflow conv uses (diagnostic, not part of the library):
// each kernel reads (or writes) exactly 1 GiB once, in one dispatch, so the per-dispatch counter value over 2^30 is the factor to apply.
//   1 read_plain      global_load_dwordx4, default policy            5 read_buf_nt     raw_buffer_load_b128 aux = 2 (nt)
//   2 read_nt         global_load_dwordx4 nt                         6 write_plain     global_store_dwordx4
//   3 read_lds        buffer_load_dwordx4 ... lds, default policy    7 write_buf_nt    raw_buffer_store_b128 aux = 2 (nt)
//   4 read_lds_nt     buffer_load_dwordx4 ... lds, aux = 2 (nt)
//   build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -o fetch_calib scripts/micro/fetch_calib.hip
//   run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- ./fetch_calib   (and the same with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void lptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr size_t BYTES = (size_t)1 << 30;
constexpr int GRID = 2048, PER_WG = (int)(BYTES / GRID);   // 512 KiB per workgroup, 256 threads x 16 B = 4 KiB per sweep, 128 sweeps

__global__ __launch_bounds__(256) void read_plain(const uint4* __restrict__ x, unsigned* sink) {
  const uint4* p = x + (size_t)blockIdx.x * (PER_WG / 16) + threadIdx.x;
  unsigned a = 0;
#pragma unroll 8
  for (int i = 0; i < PER_WG / 4096; ++i) { const uint4 v = p[i * 256]; a += v.x ^ v.y ^ v.z ^ v.w; }
  if (a == 0x12345678u) sink[0] = a;
}
__global__ __launch_bounds__(256) void read_nt(const uint4* __restrict__ x, unsigned* sink) {
  const uint4* p = x + (size_t)blockIdx.x * (PER_WG / 16) + threadIdx.x;
  unsigned a = 0;
#pragma unroll 8
  for (int i = 0; i < PER_WG / 4096; ++i) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4*>(p + i * 256));
    a += v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (a == 0x12345678u) sink[0] = a;
}
template <int AUX>
__global__ __launch_bounds__(256) void read_lds(const char* __restrict__ x, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 4096];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const char* mine = x + (size_t)blockIdx.x * PER_WG;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, PER_WG, 0x00020000);
  unsigned a = 0;
  for (int i = 0; i < PER_WG / 4096; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t*)(smem + (i & 3) * 4096 + wave * 1024), 16, wave * 1024 + lane * 16, i * 4096, 0, AUX);
#endif
    if ((i & 3) == 3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); a += *reinterpret_cast<volatile unsigned*>(smem + threadIdx.x * 4); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (a == 0x12345678u) sink[0] = a;
}
__global__ __launch_bounds__(256) void read_buf_nt(const char* __restrict__ x, unsigned* sink) {
  const char* mine = x + (size_t)blockIdx.x * PER_WG;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, PER_WG, 0x00020000);
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  unsigned a = 0;
#pragma unroll 8
  for (int i = 0; i < PER_WG / 4096; ++i) { const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, threadIdx.x * 16, i * 4096, 2); a += v[0] ^ v[1] ^ v[2] ^ v[3]; }
  if (a == 0x12345678u) sink[0] = a;
}
__global__ __launch_bounds__(256) void write_plain(uint4* __restrict__ x) {
  uint4* p = x + (size_t)blockIdx.x * (PER_WG / 16) + threadIdx.x;
#pragma unroll 8
  for (int i = 0; i < PER_WG / 4096; ++i) p[i * 256] = make_uint4(i, 1, 2, 3);
}
__global__ __launch_bounds__(256) void write_buf_nt(char* __restrict__ x) {
  char* mine = x + (size_t)blockIdx.x * PER_WG;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)mine, 0, PER_WG, 0x00020000);
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll 8
  for (int i = 0; i < PER_WG / 4096; ++i) __builtin_amdgcn_raw_buffer_store_b128((u4){(unsigned)i, 1u, 2u, 3u}, rs, threadIdx.x * 16, i * 4096, 2);
}

int main() {
  char *a, *flush; unsigned* sink;
  CK(hipMalloc(&a, BYTES)); CK(hipMalloc(&flush, BYTES)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(a, 1, BYTES));
  auto cold = [&] { CK(hipMemset(flush, 2, BYTES)); CK(hipDeviceSynchronize()); };   // 1 GiB through the caches: nothing of `a` stays in L2 / MALL
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timed = [&](const char* name, auto launch) {
    cold();
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-14s 1 GiB in %7.1f us = %5.2f TB/s\n", name, ms * 1e3, BYTES / (ms * 1e-3) / 1e12);
  };
  timed("read_plain", [&] { hipLaunchKernelGGL(read_plain, dim3(GRID), dim3(256), 0, 0, (const uint4*)a, sink); });
  timed("read_nt", [&] { hipLaunchKernelGGL(read_nt, dim3(GRID), dim3(256), 0, 0, (const uint4*)a, sink); });
  timed("read_lds", [&] { hipLaunchKernelGGL(read_lds<0>, dim3(GRID), dim3(256), 0, 0, (const char*)a, sink); });
  timed("read_lds_nt", [&] { hipLaunchKernelGGL(read_lds<2>, dim3(GRID), dim3(256), 0, 0, (const char*)a, sink); });
  timed("read_buf_nt", [&] { hipLaunchKernelGGL(read_buf_nt, dim3(GRID), dim3(256), 0, 0, (const char*)a, sink); });
  timed("write_plain", [&] { hipLaunchKernelGGL(write_plain, dim3(GRID), dim3(256), 0, 0, (uint4*)a); });
  timed("write_buf_nt", [&] { hipLaunchKernelGGL(write_buf_nt, dim3(GRID), dim3(256), 0, 0, a); });
  return 0;
}
