// 3x3 stride-1 convolution with a GroupNorm(+SiLU) prologue on the large feature maps (the VAE decoder's resnet convs: 71 % of a patch's
// MACs, /root/reference/pixel_latent_vector.py:81 -> pipeline.decode_latents -> AutoencoderKL.decode; segmentor.py:106):
// producer / consumer wave specialisation inside one persistent 512-thread workgroup per CU ("dataflow" kernel).
//
// Why another conv3x3 kernel: in the 8x16 halo-tile kernel (kernels_conv3x3.hip) every wave does everything -- stages the halo through
// registers, normalises it, issues the weight DMA, runs the MFMAs, and meets the other three waves at a workgroup barrier once per tap.
// Its matrix pipe is busy 36-40 % of the time; the serial phases of a tile (first slab, epilogue) and the per-tap barrier are what is
// left (profiles/r02_conv3x3_pingpong.md).  Here the roles are split:
//   * waves 0-3, one per SIMD: CONSUMERS.  Nothing but operand reads (ds_read_b128, counted lgkmcnt) and MFMAs over a 16x16-pixel x
//     128-channel tile: each wave 8 pixel rows x 64 channels = 8x4 accumulator tiles (128 VGPRs), weights of a tap held in registers
//     for the whole step (both k-halves of all four channel tiles), pixel rows streamed in pairs one pair ahead, both k-halves of an
//     accumulator back to back.  No workgroup barrier anywhere in the loop: a step starts when a progress word in LDS says its
//     weight slice and halo image have landed.
//   * waves 4-7, one per SIMD: PRODUCERS.  Per consumer step: the next-but-two weight slice [128][64] by LDS-DMA into a 4-slot ring;
//     1/6 of the next slab's halo image global -> registers (two steps ahead of its use) -> GroupNorm-apply + SiLU -> LDS, with the
//     transform as v_fma_mix pair blocks (common.h gn_pair): beside a matrix stream on the same SIMD this form costs ~3 % of the
//     matrix rate, hipcc's packed-fp32 form 22 % (scripts/micro/conv_consumer.hip).  All vector-memory traffic of a producer wave is
//     inline asm behind ONE counted vmcnt per step (loads, LDS-DMA and their order are known statically per tap).
//   * progress words (LDS): producer wave w publishes "iterations completed + 1" after its LDS writes have landed, consumer wave w
//     publishes "steps whose weights are in registers"; a producer iteration i needs min(consumers) >= i (ring slot and halo buffer
//     free), a consumer step s needs min(producers) >= s + 2 before it prefetches step s + 1's operands.
// A workgroup walks a contiguous run of (pixel tile, channel tile) units (XCD-aware: the runs of one XCD's workgroups are adjacent), the
// step sequence runs through unit boundaries on the producer side; the consumer's epilogue (residual, fp16 rounding, fused GroupNorm
// statistics, 16-byte stores) is the only phase in which a SIMD's matrix pipe idles.
//
// Scope (conv3x3d_selected): one source, Cin % 64 == 0, N % 128 == 0, H, W % 16 == 0, GroupNorm prologue, plain fp16 output and residual.
// Everything else stays on the kernels of kernels_conv3x3.hip / kernels_conv3x3p.hip.
#include "common.h"
#include <map>
#include <mutex>

namespace {

typedef __attribute__((address_space(3))) void lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int V> using ic_t = std::integral_constant<int, V>;

constexpr int D_HWD = 18, D_HPX = 18 * 18, D_ROWB = D_HWD * 128;       // halo image of a 16x16 tile: 18 x 18 pixels of 128 B (64 channels)
constexpr unsigned D_HB = 41 * 1024;                                   // one halo image = 41 DMA pieces of 8 pixels (324 pixels + 4 pad), two of them
constexpr int D_NSLOT = 4;                                             // weight ring
constexpr unsigned D_WSLOT = 128 * 128, D_WOFF = 2 * D_HB;
constexpr unsigned D_FLAGS = D_WOFF + D_NSLOT * D_WSLOT;               // [producer progress x4][consumer progress x4] | per-wave dump rows
constexpr unsigned D_DUMP = D_FLAGS + 64;                              // 8 waves x 256 B: where the lanes other than 0 put their copy of a progress word
constexpr unsigned D_DMADUMP = D_DUMP + 8 * 256;                       // 4 KiB: target of the DMA instructions that exist only to keep the counted waits uniform
constexpr unsigned D_AFF = D_DMADUMP + 4096;                           // [producer wave 4][table 2] x 512 B: scale (256 B) | shift (256 B) of a slab's 64 channels
constexpr unsigned D_BT = D_AFF + 8 * 512;                               // [unit parity 2] x (bias 128 floats | time embedding 128 floats) of a unit's channel tile
constexpr unsigned D_LDS = D_BT + 2 * 1024;
static_assert(D_LDS <= 160 * 1024, "LDS budget of one workgroup per CU");
constexpr unsigned D_OOR = 0x80000000u;                                // beyond num_records of every eligible tensor: the load returns zeros
constexpr int D_NROUND = 11;                                           // 324 pixels x 8 chunks = 2592 = 10 x 256 + 32 lane-chunks
enum { D_RES = 1, D_STATS = 2 };

__device__ __forceinline__ int d_swzx(int hx) { return (0xcb5888 >> (3 * (hx >> 1))) & 7; }   // column swizzle of the halo image (kernels_conv3x3.hip)

// ---- LDS / memory primitives the compiler must not fence or wait for (counted by hand) ----
template <int OFF>
__device__ __forceinline__ void lds_read128(f16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT>
__device__ __forceinline__ void lds_wait2(f16x8& a, f16x8& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void lds_wait4(f16x8& a, f16x8& b, f16x8& c, f16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(CNT));
}
template <int CNT>
__device__ __forceinline__ void lds_wait6(f16x8& a, f16x8& b, f16x8& c, f16x8& d, f16x8& e, f16x8& f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(CNT));
}
__device__ __forceinline__ void lds_wait_flags(f16x8& a, f16x8& b, f16x8& c, f16x8& d, u32x4& fl) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(fl));
}
__device__ __forceinline__ void lds_read_flags(u32x4& d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ unsigned flags_min_now(unsigned addr) {   // slow path: read the four progress words and wait for them
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  const unsigned m = min(min(v[0], v[1]), min(v[2], v[3]));
  return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}
__device__ __forceinline__ void lds_write32(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write128(unsigned addr, const u32x4& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// (a __builtin_bit_cast applied directly to a vector ELEMENT expression reads element 0 whatever the index: hipcc 7.2; by value it is fine)
__device__ __forceinline__ float u2f(unsigned v) { return __builtin_bit_cast(float, v); }

struct UnitC { int b, oy0, ox0, n0; };

#ifdef C3D_STAMPS   // diagnostic build only (scripts/conv_stamps_d.py): cycle sums / poll counts of waves 0 and 4 of workgroup 0
__device__ unsigned long long c3d_dbg[48];
__device__ __forceinline__ unsigned long long d_stamp() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define DSTAMP(v) const unsigned long long v = d_stamp()
#define DACC(i, expr) dbg[i] += (expr)
#else
#define DSTAMP(v)
#define DACC(i, expr)
#endif

template <int FLAGS>
__global__ __launch_bounds__(512, 2) void conv3x3d_kernel(const ConvParams p, const int units) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(lptr_t*)smem_raw;
  const int Cin = p.C1, nslab = Cin >> 6, ntn = p.N >> 7;
  const int H = p.Hout, W = p.Wout, tiles_x = W >> 4, tiles_y = H >> 4;

  // this workgroup's run of units (n-tile fastest, then x, y, image); the runs of the workgroups that share an XCD are adjacent
  const int G = gridDim.x, id = blockIdx.x;
  const int q8 = G >> 3, r8 = G & 7, xcd = id & 7, idx = id >> 3;
  const int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int u0 = (int)((long long)sw * units / G), u1 = (int)((long long)(sw + 1) * units / G);
  const int n_u = u1 - u0;
  const int S = n_u * nslab * 9, total_slabs = n_u * nslab;
  auto decode = [&](int u) __attribute__((always_inline)) -> UnitC {   // (readfirstlane: descriptors and LDS-DMA bases built from these must be provably uniform)
    UnitC c;
    int t = u / ntn;
    c.n0 = __builtin_amdgcn_readfirstlane((u - t * ntn) * 128);
    const int t2 = t / tiles_x;
    c.ox0 = __builtin_amdgcn_readfirstlane((t - t2 * tiles_x) * 16);
    const int t3 = t2 / tiles_y;
    c.oy0 = __builtin_amdgcn_readfirstlane((t2 - t3 * tiles_y) * 16);
    c.b = __builtin_amdgcn_readfirstlane(t3);
    return c;
  };

  // Stagger: workgroups that start together stay in lock-step (equal work per unit), and then all 256 of them store their 64 KB output
  // tiles in the same few microseconds: the epilogue, the one phase in which the matrix pipes idle, is stretched 4x by a chip-wide write
  // burst (scripts/conv_stamps_d.py).  One sixteenth of a unit's duration per phase step spreads the epilogues evenly over time.
  if (n_u >= 4) {
    const int nsleep = ((id >> 3) & 15) * nslab;
    for (int k = 0; k < nsleep; ++k) __builtin_amdgcn_s_sleep(10);
  }
  if (tid < 8) *reinterpret_cast<volatile unsigned*>(smem_raw + D_FLAGS + tid * 4) = 0u;
  __syncthreads();
  if (n_u <= 0) return;

  if (wave >= 4) {
    // ================================================= PRODUCERS =================================================
    // Nothing asynchronous ever targets a VGPR here (an asm load's destination may be COPIED by the register allocator before the data has
    // landed wherever the value lives across control flow: measured, wrong halo rows).  Raw halo chunks go by LDS-DMA straight into
    // their final place in the halo image (piece j = halo pixels [8j, 8j + 8) = 1 KiB; the column swizzle of the image is applied on the
    // SOURCE address: lane l of a piece fetches channel chunk (l & 7) ^ swzx(hx)) and are normalised IN PLACE two iterations later:
    // ds_read own chunk + this wave's private scale / shift table (also by DMA) -> gn_pair -> mask -> ds_write, all inside one basic block.
    const int pw = wave - 4;
    const int ld1 = p.ld1 ? p.ld1 : p.C1;
    const long long Kw = 9LL * Cin;
    const __amdgpu_buffer_rsrc_t scrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.gn_scale, 0, (int)((long long)p.B * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t shrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.gn_shift, 0, (int)((long long)p.B * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long long)p.Nrows * Kw * 2), 0x00020000);
    const unsigned pflag = lds0 + D_FLAGS + (unsigned)pw * 4u, cflags = lds0 + D_FLAGS + 16u;

    // per-round constants of this lane (round r = piece pw + 4 r, lane l -> halo pixel px = 8 (pw + 4 r) + l / 8):
    //   DMA side: the lane fetches channel chunk (l & 7) ^ swzx(hx) into position l & 7 of the pixel's 128-byte row (LDS-DMA writes lane-linear);
    //   transform side: the lane normalises channel chunk l & 7 -- a FIXED chunk, so its scale / shift live in registers for a whole slab --
    //   which sits at position (l & 7) ^ swzx(hx) of the same row.
    unsigned rc_yx[D_NROUND];   // hy << 8 | hx, 0xffff: no such pixel (piece 40's pad pixels; pieces that do not exist)
    unsigned rc_rel[D_NROUND];  // DMA source offset relative to halo pixel (0, 0) of the unit: ((hy W + hx) pitch + 8 chunk) * 2 bytes
    unsigned rc_lds[D_NROUND];  // transform address inside a halo image
#pragma unroll
    for (int r = 0; r < D_NROUND; ++r) {
      const int px = (pw + 4 * r) * 8 + (lane >> 3);
      const int hy = px / D_HWD, hx = px - hy * D_HWD, sz = d_swzx(hx);
      rc_yx[r] = px < D_HPX ? (unsigned)(hy << 8 | hx) : 0xffffu;
      rc_rel[r] = (unsigned)(((hy * W + hx) * ld1 + ((lane & 7) ^ sz) * 8) * 2);
      rc_lds[r] = lds0 + (unsigned)(px * 128 + (((lane & 7) ^ sz) << 4));
    }
    // ---- cursors ----
    int w_step = 0, w_tap = 0, w_c = 0, w_u = u0;   // weights: next step whose slice is issued
    int w_voff[4];
    auto set_wvoff = [&](int n0) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = pw * 32 + i * 8 + (lane >> 3), pos = lane & 7;
        w_voff[i] = (int)((unsigned)(n0 + r) * (unsigned)(Kw * 2) + (unsigned)((pos ^ ((r >> 1) & 7)) * 16)) - i * 1024;   // the instruction offset is added to BOTH addresses
      }
    };
    set_wvoff(decode(u0).n0);
    // slice of step w_step -> ring slot w_step % D_NSLOT in four 1-KiB pieces (issued one by one between the transform's arithmetic: a wave
    // that issues its vector-memory instructions back to back waits ~100 cycles on each), then advance
    unsigned char* w_dst = nullptr; int w_soff = 0; bool w_live = false;
    auto w_begin = [&]() __attribute__((always_inline)) {
      w_live = w_step < S;
      w_dst = w_live ? smem_raw + D_WOFF + (unsigned)(w_step & (D_NSLOT - 1)) * D_WSLOT + pw * 4096 : smem_raw + D_DMADUMP;   // past the end: the same
      w_soff = w_live ? (w_tap * Cin + w_c * 64) * 2 : 0;                                                                      // instructions, data nobody reads
    };
    auto w_piece = [&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass rejects this builtin (target feature) and then silently drops the kernel stub
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t*)w_dst, 16, w_live ? w_voff[i] : (int)D_OOR, w_soff, i * 1024, 0);
#endif
    };
    // bias and time embedding of the unit the weight cursor enters -> LDS table (unit parity): the consumers start that unit's sums from it.
    // Issued behind the iteration's other pieces by producer wave 0 only: the next counted wait then also covers some halo pieces (stricter, never weaker).
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : p.gn_scale), 0, p.bias ? p.Nrows * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.temb ? p.temb : p.gn_scale), 0, p.temb ? (int)((long long)p.B * p.ld_temb * 4) : 0, 0x00020000);
    auto dma_bt = [&](const UnitC& un, int parity) __attribute__((always_inline)) {
      if (pw == 0 && lane < 32) {
        unsigned char* dst = smem_raw + D_BT + (unsigned)parity * 1024u;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lptr_t*)dst, 16, (un.n0 + lane * 4) * 4, 0, 0, 0);            // no bias: zero records, the DMA writes zeros
        __builtin_amdgcn_raw_ptr_buffer_load_lds(trsrc, (lptr_t*)(dst + 512), 16, (un.b * p.ld_temb + un.n0 + lane * 4) * 4, 0, 0, 0);
#endif
      }
    };
    auto w_end = [&]() __attribute__((always_inline)) {
      if (w_live) {
        ++w_step;
        if (++w_tap == 9) {
          w_tap = 0;
          if (++w_c == nslab) {
            w_c = 0; ++w_u;
            if (w_u < u1) { const UnitC un = decode(w_u); set_wvoff(un.n0); dma_bt(un, (w_u - u0) & 1); }
          }
        }
      }
    };
    auto issue_w = [&]() __attribute__((always_inline)) { w_begin(); static_for<0, 4>([&](auto ic) { w_piece(ic); }); w_end(); };
    // halo: the slab being built (global slab index l_k, unit coordinates l_un, slab of the unit l_c); l_k >= total_slabs: nothing to build
    int l_k = 0, l_c = 0, l_u = u0;
    UnitC l_un = decode(u0);
    // per unit: which of this lane's halo pixels lie inside the image (bit r of vbits), and a descriptor whose base is halo pixel (0, 0) of
    // the unit (it may lie in front of the tensor for border tiles: only lanes whose pixel is inside are ever given a real offset)
    unsigned vbits = 0;
    __amdgpu_buffer_rsrc_t xrsrc;
    auto set_unit = [&]() __attribute__((always_inline)) {
      vbits = 0;
#pragma unroll
      for (int r = 0; r < D_NROUND; ++r) {
        const int hy = (int)(rc_yx[r] >> 8), hx = (int)(rc_yx[r] & 0xff);
        const bool inb = rc_yx[r] != 0xffffu && (unsigned)(l_un.oy0 + hy - 1) < (unsigned)H && (unsigned)(l_un.ox0 + hx - 1) < (unsigned)W;
        vbits |= inb ? 1u << r : 0u;
      }
      const long long org = ((long long)(l_un.b * H + l_un.oy0 - 1) * W + l_un.ox0 - 1) * ld1 * 2;
      const unsigned long long xa = (unsigned long long)reinterpret_cast<const char*>(p.x) + (unsigned long long)org;
      const unsigned xlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xa), xhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xa >> 32));
      xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)xhi << 32) | xlo), 0, 0x7fffffff, 0x00020000);
    };
    set_unit();
    auto next_slab = [&]() __attribute__((always_inline)) {
      ++l_k;
      if (++l_c == nslab) { l_c = 0; ++l_u; if (l_u < u1) { l_un = decode(l_u); set_unit(); } }
    };
    auto dma_round = [&](auto rc_, unsigned hbuf) __attribute__((always_inline)) {   // raw piece pw + 4 r of the slab at the cursor -> its place in image hbuf
      constexpr int r = decltype(rc_)::value;
      const bool live = l_k < total_slabs && (r < 10 || pw == 0);
      const int voff = live && ((vbits >> r) & 1u) ? (int)rc_rel[r] : (int)D_OOR;
      unsigned char* dst = live ? smem_raw + hbuf + (unsigned)(pw + 4 * r) * 1024u : smem_raw + D_DMADUMP;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (lptr_t*)dst, 16, voff, l_c * 128, 0, 0);
#endif
    };
    auto dma_affine = [&](int tab) __attribute__((always_inline)) {   // scale | shift of the slab's 64 channels -> this wave's table `tab` (512 B): lanes 0-15 | 16-31
      const bool live = l_k < total_slabs;
      const int voff = live && lane < 32 ? (l_un.b * Cin + l_c * 64) * 4 + (lane & 15) * 16 : (int)D_OOR;
      unsigned char* dst = live ? smem_raw + D_AFF + (unsigned)(pw * 2 + tab) * 512u : smem_raw + D_DMADUMP;
#if defined(__HIP_DEVICE_COMPILE__)
      if (lane < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(scrsrc, (lptr_t*)dst, 16, voff, 0, 0, 0);
      else if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(shrsrc, (lptr_t*)dst, 16, voff, 0, 0, 0);
#endif
    };
    float gs[8], gt[8];   // scale / shift of channels [8 (l & 7), + 8) of the slab being normalised
    auto read_affine = [&](int tab) __attribute__((always_inline)) {
      const unsigned taddr = lds0 + D_AFF + (unsigned)(pw * 2 + tab) * 512u + (unsigned)(lane & 7) * 32u;
      u32x4 sc0, sc1, sh0, sh1;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:256\n\tds_read_b128 %3, %4 offset:272\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(sc0), "=&v"(sc1), "=&v"(sh0), "=&v"(sh1) : "v"(taddr) : "memory");
      gs[0] = u2f(sc0[0]); gs[1] = u2f(sc0[1]); gs[2] = u2f(sc0[2]); gs[3] = u2f(sc0[3]); gs[4] = u2f(sc1[0]); gs[5] = u2f(sc1[1]); gs[6] = u2f(sc1[2]); gs[7] = u2f(sc1[3]);
      gt[0] = u2f(sh0[0]); gt[1] = u2f(sh0[1]); gt[2] = u2f(sh0[2]); gt[3] = u2f(sh0[3]); gt[4] = u2f(sh1[0]); gt[5] = u2f(sh1[1]); gt[6] = u2f(sh1[2]); gt[7] = u2f(sh1[3]);
    };
    auto quad = [&](auto hc, const u32x4& x, u32x4& v) __attribute__((always_inline)) {   // elements 4h .. 4h+3 of a chunk
      constexpr int h = decltype(hc)::value;
      unsigned oa, ob;
      gn_quad<true>(x[2 * h], x[2 * h + 1], gs[4 * h], gt[4 * h], gs[4 * h + 1], gt[4 * h + 1], gs[4 * h + 2], gt[4 * h + 2], gs[4 * h + 3], gt[4 * h + 3], oa, ob);   // SiLU always (conv3x3d_selected)
      v[2 * h] = oa; v[2 * h + 1] = ob;
    };
    // group g of a slab = rounds 2g, 2g+1 (g < 5); group 5 = round 10 (piece 40: wave 4 only, its last four pixels are padding)
    // One producer iteration's work: the step's four weight pieces, the halo pieces of group GD (-1: none; with group 0 the scale / shift
    // table), and the in-place transform of group GX (-1: none), with the DMA instructions spread between the transform's four quads.
    auto iteration_work = [&](auto gdc, auto gxc, auto wc_, unsigned hbuf, int tab) __attribute__((always_inline)) {
      constexpr int GD = decltype(gdc)::value, GX = decltype(gxc)::value;
      constexpr bool WITH_W = decltype(wc_)::value != 0;   // 0: the prologue's transform of the first image (nothing to issue)
      if constexpr (WITH_W) w_begin();
      auto dma_slot = [&](auto kc) __attribute__((always_inline)) {   // slot k of 4.  ALL weight pieces before the halo pieces: the next iteration's
        constexpr int k = decltype(kc)::value;                          // wait then retires the weights while the (HBM-latency) halo pieces fly on
        if constexpr (WITH_W) { if constexpr (k == 0) { w_piece(ic_t<0>{}); w_piece(ic_t<1>{}); } if constexpr (k == 1) w_piece(ic_t<2>{}); if constexpr (k == 2) w_piece(ic_t<3>{}); }
        if constexpr (GD >= 0) {
          if constexpr (k == 3) {
            if constexpr (GD == 0) dma_affine(tab);
            dma_round(ic_t<2 * GD>{}, hbuf);
            if constexpr (GD < 5) dma_round(ic_t<2 * GD + 1>{}, hbuf);
          }
        }
      };
      const bool xf = GX >= 0 && l_k < total_slabs && (GX < 5 || pw == 0);
      if (xf) {
        if constexpr (GX == 0) read_affine(tab);
        if constexpr (GX >= 0 && GX < 5) {
          const unsigned a0 = rc_lds[2 * (GX < 0 ? 0 : GX)] + hbuf, a1 = rc_lds[2 * (GX < 0 ? 0 : GX) + 1] + hbuf;
          u32x4 x0, x1, v0, v1;
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0), "=&v"(x1) : "v"(a0), "v"(a1) : "memory");
          const unsigned k0 = (unsigned)__builtin_amdgcn_sbfe((int)vbits, 2 * (GX < 0 ? 0 : GX), 1), k1 = (unsigned)__builtin_amdgcn_sbfe((int)vbits, 2 * (GX < 0 ? 0 : GX) + 1, 1);
          quad(ic_t<0>{}, x0, v0); dma_slot(ic_t<0>{});
          quad(ic_t<1>{}, x0, v0); dma_slot(ic_t<1>{});
          v0[0] &= k0; v0[1] &= k0; v0[2] &= k0; v0[3] &= k0;   // zero padding applies to the NORMALISED tensor
          lds_write128(a0, v0);
          quad(ic_t<0>{}, x1, v1); dma_slot(ic_t<2>{});
          quad(ic_t<1>{}, x1, v1); dma_slot(ic_t<3>{});
          v1[0] &= k1; v1[1] &= k1; v1[2] &= k1; v1[3] &= k1;
          lds_write128(a1, v1);
        } else if constexpr (GX == 5) {
          const unsigned a0 = rc_lds[10] + hbuf;
          u32x4 x0, v0;
          asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0) : "v"(a0) : "memory");
          const unsigned k0 = (unsigned)__builtin_amdgcn_sbfe((int)vbits, 10, 1);
          quad(ic_t<0>{}, x0, v0); dma_slot(ic_t<0>{}); dma_slot(ic_t<1>{});
          quad(ic_t<1>{}, x0, v0); dma_slot(ic_t<2>{}); dma_slot(ic_t<3>{});
          v0[0] &= k0; v0[1] &= k0; v0[2] &= k0; v0[3] &= k0;
          lds_write128(a0, v0);
        }
      } else {
        static_for<0, 4>([&](auto kc) { dma_slot(kc); });
      }
      if constexpr (WITH_W) w_end();
    };
    // progress word by ONE unmasked ds_write_b32: lane 0 hits the word, the other lanes a dump row of their own
    const unsigned pflag_addr = lane == 0 ? pflag : lds0 + D_DUMP + (unsigned)(4 + pw) * 256u + (unsigned)lane * 4u;
    auto publish = [&](unsigned v) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      lds_write32(pflag_addr, v);
    };

#ifdef C3D_STAMPS
    unsigned long long dbg[32] = {0};
#endif
    DSTAMP(p_t0);
    // ---- prologue: weight slices of steps 0..2, the whole first halo image ----
    issue_w(); issue_w(); issue_w();
    dma_affine(0);
    static_for<0, D_NROUND>([&](auto rc_) { dma_round(rc_, 0u); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    static_for<0, 6>([&](auto gc) { iteration_work(ic_t<-1>{}, gc, ic_t<0>{}, 0u, 0); });
    next_slab();
    publish(1u);
    DSTAMP(p_t1);
    DACC(0, p_t1 - p_t0);

    // ---- iterations: i = 9 blk + ph mirrors consumer step i; block blk builds the image of slab blk + 1 (cursor l_k) ----
    // ph:              0    1    2    3    4    5    6    7    8
    // DMA (group)      -    0    1    2    3    4    5    -    -      (+ the scale / shift table with group 0, + 4 weight pieces every phase)
    // transform        -    -    -    0    1    2    3    4    5      (two iterations behind its DMA)
    // The halo buffer is free from iteration 9 blk + 1 on (gate: every consumer past the first weights of step 9 blk).
    // The iteration opens with ONE counted wait that leaves only iteration i - 1's halo / table pieces in flight (0 4 2 2 2 2 1 0 0 by phase;
    // they are issued behind its weight pieces): the halo pieces about to be normalised (iteration i - 2) and the weight slice of step
    // i + 2 (iteration i - 1) have landed.  Progress i + 2 at the end of the iteration therefore means: weights up to step i + 2, halo groups
    // up to this iteration's.
    // Past the end of the unit list the same instructions are issued with out-of-range sources: the counts stay valid.
    const int nblk = n_u * nslab;
    for (int blk = 0; blk < nblk; ++blk) {
      const unsigned hbuf = (unsigned)((blk + 1) & 1) * D_HB;
      const int tab = (blk + 1) & 1;
      static_for<0, 9>([&](auto phc) {
        constexpr int ph = decltype(phc)::value;
        constexpr int HP[9] = {0, 4, 2, 2, 2, 2, 1, 0, 0};
        constexpr int NWAIT = HP[(ph + 8) % 9];
        const int i = blk * 9 + ph;
        DSTAMP(q0);
        while (flags_min_now(cflags) < (unsigned)i) { __builtin_amdgcn_s_sleep(1); DACC(5, 1); }
        DSTAMP(q1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NWAIT) : "memory");
        DSTAMP(q2);
        iteration_work(ic_t<(ph >= 1 && ph <= 6) ? ph - 1 : -1>{}, ic_t<ph >= 3 ? ph - 3 : -1>{}, ic_t<1>{}, hbuf, tab);
        DSTAMP(q3);
        publish((unsigned)i + 2u);
        if constexpr (ph == 8) next_slab();   // (behind the publication: a unit change is ~250 instructions of tile arithmetic)
        DSTAMP(q4);
        DACC(1, q1 - q0); DACC(2, q2 - q1); DACC(3, q3 - q2); DACC(4, q4 - q3); DACC(6, 1); DACC(8 + ph, q3 - q2); DACC(17 + ph, q1 - q0);
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef C3D_STAMPS
    { DSTAMP(p_t2); dbg[7] = p_t2 - p_t0; if (blockIdx.x == 0 && pw == 0 && lane == 0) for (int i = 0; i < 32; ++i) c3d_dbg[16 + i] = dbg[i]; }
#endif
    return;
  }

  // ================================================= CONSUMERS =================================================
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int g = lane >> 4, l15 = lane & 15;
  unsigned xb[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int hx = l15 + j;
    xb[j] = lds0 + (unsigned)(((wave_m * 8) * D_HWD + hx) * 128 + ((g ^ d_swzx(hx)) << 4));
  }
  const int wrow = wave_n * 64 + l15;
  const unsigned w_lane = lds0 + D_WOFF + (unsigned)(wrow * 128 + ((g ^ ((wrow >> 1) & 7)) << 4));
  const unsigned pflags = lds0 + D_FLAGS;
  // progress word of this wave by ONE unmasked ds_write_b32: lane 0 hits the word, the other lanes a dump row of their own
  const unsigned cflag_addr = lane == 0 ? lds0 + D_FLAGS + 16u + (unsigned)wave * 4u : lds0 + D_DUMP + (unsigned)wave * 256u + (unsigned)lane * 4u;

#ifdef C3D_STAMPS
  unsigned long long dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  f32x4 acc[4][8];
  f16x8 Wf[2][4], X[2][2][2];   // Wf[k-half][channel tile]; X[buffer][row of the pair][k-half]
  u32x4 fl;                     // producers' progress words, read in the shadow of pair 2

  auto issue_x = [&](auto pc, auto kyc, auto kxc, unsigned hb, f16x8 (&dst)[2][2]) __attribute__((always_inline)) {
    constexpr int pr = decltype(pc)::value, ky = decltype(kyc)::value, kx = decltype(kxc)::value;
    const unsigned b0 = xb[kx] + hb, b1 = b0 ^ 64u;   // chunk bit 2 = k-half: XOR commutes with the swizzle
    lds_read128<(2 * pr + ky) * D_ROWB>(dst[0][0], b0);
    lds_read128<(2 * pr + ky) * D_ROWB>(dst[0][1], b1);
    lds_read128<(2 * pr + 1 + ky) * D_ROWB>(dst[1][0], b0);
    lds_read128<(2 * pr + 1 + ky) * D_ROWB>(dst[1][1], b1);
  };
  auto issue_w1 = [&](auto ac, unsigned wc) __attribute__((always_inline)) {
    constexpr int a = decltype(ac)::value;
    lds_read128<a * 2048>(Wf[0][a], wc);
    lds_read128<a * 2048>(Wf[1][a], wc ^ 64u);
  };
  auto mfma4 = [&](auto ac, auto pc, f16x8 (&x)[2][2]) __attribute__((always_inline)) {
    constexpr int a = decltype(ac)::value, pr = decltype(pc)::value;
    acc[a][2 * pr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][a], x[0][0], acc[a][2 * pr], 0, 0, 0);
    acc[a][2 * pr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[1][a], x[0][1], acc[a][2 * pr], 0, 0, 0);
    acc[a][2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][a], x[1][0], acc[a][2 * pr + 1], 0, 0, 0);
    acc[a][2 * pr + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[1][a], x[1][1], acc[a][2 * pr + 1], 0, 0, 0);
  };
  auto wait_producers = [&](unsigned need) __attribute__((always_inline)) {
    while (flags_min_now(pflags) < need) {}
  };

  // bias + time embedding of a unit -> starting value of its sums (4 channels per lane and channel tile)
  f32x4 bt[4];
  auto load_bt = [&](const UnitC& u) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int n = u.n0 + wave_n * 64 + a * 16 + g * 4;
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), tt = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + n);
      if (p.temb) tt = *reinterpret_cast<const float4*>(p.temb + (long long)u.b * p.ld_temb + n);
      bt[a] = (f32x4){bb.x + tt.x, bb.y + tt.y, bb.z + tt.z, bb.w + tt.w};
    }
  };
  auto init_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[a][m] = bt[a];
  };

  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((long long)p.M * p.ldy * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : (const f16*)p.y), 0, (int)((long long)p.M * (p.res ? p.ld_res : p.ldy) * 2), 0x00020000);

  // Epilogue of a unit.  Lane holds y[pixel = (row m, column l15)][channels 4g .. 4g+3 of channel tile a].  One v_permlane16_swap per dword
  // between the packed values of two rows P = 2pr, Q = 2pr + 1 leaves lanes with even g holding channels 4g .. 4g+7 of row P and lanes
  // with odd g channels 4(g-1) .. 4(g-1)+7 of row Q: 16-byte stores; the residual is read in the same shape and un-swapped the same way
  // (the swap is its own inverse).
  auto epilogue = [&](const UnitC& u, int next_parity, bool has_next) __attribute__((always_inline)) {
    const int mrow = wave_m * 8 + (g & 1);
    const unsigned pix = (unsigned)((u.b * H + u.oy0 + mrow) * W + u.ox0 + l15);
    const unsigned chb = (unsigned)(u.n0 + wave_n * 64 + (g & ~1) * 4);
    const unsigned yoff = (pix * (unsigned)p.ldy + chb) * 2u, ystep = (unsigned)(2 * W * p.ldy) * 2u;
    const unsigned roff = (pix * (unsigned)p.ld_res + chb) * 2u, rstep = (unsigned)(2 * W * p.ld_res) * 2u;
    constexpr bool RES = (FLAGS & D_RES) != 0, ST = (FLAGS & D_STATS) != 0;
    DSTAMP(ep0);
    u32x4 R[2][4];
    if constexpr (RES) {
#pragma unroll
      for (int a = 0; a < 4; ++a) R[0][a] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (int)(roff + a * 32), 0, 0);
    }
    // fused GroupNorm statistics: per (channel tile, row pair) the 4 sums and 4 sums of squares of the lane's channels are reduced over the
    // 16 pixel lanes at once (row16_reduce_spread<8>: lane keeps ONE of the eight totals, value index jv) and added up over the row pairs:
    // four live registers instead of thirty-two (the epilogue shares the register file with 128 accumulators)
    float tot[4] = {0.f, 0.f, 0.f, 0.f};
    static_for<0, 4>([&](auto pc) {
      constexpr int pr = decltype(pc)::value;
      DSTAMP(ep1);
      if constexpr (RES && pr < 3) {
#pragma unroll
        for (int a = 0; a < 4; ++a) R[(pr + 1) & 1][a] = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, (int)(roff + (pr + 1) * rstep + a * 32), 0, 0);
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        f32x4 v0 = acc[a][2 * pr], v1 = acc[a][2 * pr + 1];
        if constexpr (RES) {
          const u32x4 r = R[pr & 1][a];
          auto s0 = __builtin_amdgcn_permlane16_swap(r[0], r[2], false, false);
          auto s1 = __builtin_amdgcn_permlane16_swap(r[1], r[3], false, false);
          v0 += up4(__builtin_bit_cast(f16x4, make_uint2(s0[0], s1[0])));
          v1 += up4(__builtin_bit_cast(f16x4, make_uint2(s0[1], s1[1])));
        }
        const f16x4 o0 = cvt4(v0), o1 = cvt4(v1);
        const uint2 q0 = __builtin_bit_cast(uint2, o0), q1 = __builtin_bit_cast(uint2, o1);
        auto r0 = __builtin_amdgcn_permlane16_swap(q0.x, q1.x, false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(q0.y, q1.y, false, false);
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){r0[0], r1[0], r0[1], r1[1]}, yrsrc, (int)(yoff + pr * ystep + a * 32), 0, 0);
        asm volatile("s_nop 1" ::: "memory");   // the next VALU instruction may overwrite the store's data registers (profiles/r02_conv3x3_pingpong.md)
        if constexpr (ST) {   // statistics of what the consumer of this tensor will read: the fp16-rounded values
          const f32x4 f0 = up4(o0), f1 = up4(o1);
          float x8[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) { x8[r] = f0[r] + f1[r]; x8[4 + r] = f0[r] * f0[r] + f1[r] * f1[r]; }
          tot[a] += row16_reduce_spread<8>(x8, l15);
        }
      }
      DSTAMP(ep2);
      DACC(8 + pr, ep2 - ep1);
    });
    DSTAMP(ep3);
    if constexpr (ST) {   // one row block per consumer wave (8 x 16 pixels); lanes l15 and l15 ^ 1 hold the same total: the even one stores it
      const int jv = ((l15 >> 3) & 1) | ((l15 >> 1) & 2) | ((l15 << 1) & 4);   // bits 0-1: channel of the lane's four, bit 2: sum / sum of squares
      const long long rblk = ((long long)(u.oy0 >> 4) * tiles_x + (u.ox0 >> 4)) * 2 + wave_m;
      if ((l15 & 1) == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int n = u.n0 + wave_n * 64 + a * 16 + g * 4 + (jv & 3);
          p.stats[(((long long)u.b * p.N + n) * p.stats_R + rblk) * 2 + (jv >> 2)] = tot[a];
        }
      }
    }
    if (has_next) {   // the next unit's sums start at bias + time embedding: table written by producer wave 0 when the weight cursor entered that unit
      const unsigned taddr = lds0 + D_BT + (unsigned)next_parity * 1024u + (unsigned)(wave_n * 64 + g * 4) * 4u;
      static_for<0, 4>([&](auto ac) {
        constexpr int a = decltype(ac)::value;
        f32x4 bb, tt;
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)" : "=&v"(bb), "=&v"(tt) : "v"(taddr), "n"(a * 64), "n"(512 + a * 64) : "memory");
        const f32x4 b = bb + tt;
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[a][m] = b;
      });
    }
    DSTAMP(ep4);
    DACC(12, ep4 - ep3); DACC(13, ep3 - ep0);
  };

  DSTAMP(c_t0);
  // ---- prologue ----
  UnitC cur = decode(u0), nxt = cur;
  load_bt(cur);
  init_acc();
  wait_producers(1u);
  DSTAMP(c_t1);
  DACC(0, c_t1 - c_t0);
  issue_x(ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, 0u, X[0]);
  static_for<0, 4>([&](auto ac) { issue_w1(ac, w_lane); });

  int s = 0;   // global step of this workgroup
  for (int u = u0; u < u1; ++u) {
    const bool has_next = u + 1 < u1;
    if (has_next) nxt = decode(u + 1);
    for (int c = 0; c < nslab; ++c) {
      const bool last_slab = c == nslab - 1;
      const unsigned hb = (unsigned)((s / 9) & 1) * D_HB;
      static_for<0, 9>([&](auto tc) {
        constexpr int T = decltype(tc)::value, ky = T / 3, kx = T % 3;
        constexpr int nky = ((T + 1) % 9) / 3, nkx = (T + 1) % 3;
        const unsigned wc_next = w_lane + (unsigned)((s + 1) & (D_NSLOT - 1)) * D_WSLOT;
        const unsigned hb_next = T == 8 ? D_HB - hb : hb;
        // entry: outstanding LDS reads = X pair 0 (4), W (8: a0 k0, a0 k1, a1 k0, ...)
        lds_wait6<6>(X[0][0][0], X[0][0][1], X[0][1][0], X[0][1][1], Wf[0][0], Wf[1][0]);
        mfma4(ic_t<0>{}, ic_t<0>{}, X[0]);
        issue_x(ic_t<1>{}, ic_t<ky>{}, ic_t<kx>{}, hb, X[1]);
        __builtin_amdgcn_sched_barrier(0);
        lds_wait2<8>(Wf[0][1], Wf[1][1]);
        mfma4(ic_t<1>{}, ic_t<0>{}, X[0]);
        __builtin_amdgcn_sched_barrier(0);
        lds_wait2<6>(Wf[0][2], Wf[1][2]);
        mfma4(ic_t<2>{}, ic_t<0>{}, X[0]);
        __builtin_amdgcn_sched_barrier(0);
        lds_wait2<4>(Wf[0][3], Wf[1][3]);
        // the step's weights are in registers: its ring slot is free, and so is everything older (outstanding: X pair 1, then this write)
        lds_write32(cflag_addr, (unsigned)(s + 1));
        mfma4(ic_t<3>{}, ic_t<0>{}, X[0]);
        __builtin_amdgcn_sched_barrier(0);
        // pair 1 (outstanding: X1 x4, flag write, X2 x4)
        issue_x(ic_t<2>{}, ic_t<ky>{}, ic_t<kx>{}, hb, X[0]);
        lds_wait4<4>(X[1][0][0], X[1][0][1], X[1][1][0], X[1][1][1]);
        static_for<0, 4>([&](auto ac) { mfma4(ac, ic_t<1>{}, X[1]); __builtin_amdgcn_sched_barrier(0); });
        // pair 2: the producers' progress words are read behind pair 3's operands
        issue_x(ic_t<3>{}, ic_t<ky>{}, ic_t<kx>{}, hb, X[1]);
        lds_read_flags(fl, pflags);
        lds_wait4<5>(X[0][0][0], X[0][0][1], X[0][1][0], X[0][1][1]);
        static_for<0, 4>([&](auto ac) { mfma4(ac, ic_t<2>{}, X[0]); __builtin_amdgcn_sched_barrier(0); });
        // pair 3
        lds_wait_flags(X[1][0][0], X[1][0][1], X[1][1][0], X[1][1][1], fl);
        // step s + 1 needs its weight slice (producer iteration s - 1 complete: progress >= s + 1) and, if it opens a slab, that slab's halo image
        // (iteration s complete: progress >= s + 2).
        // At the last step of a unit the same reads go out unchecked and unused (one code path, no join for the register allocator: a
        // second variant of this pair made hipcc spill the weight fragments around every slab); the next unit's first operands are
        // issued again behind the epilogue, whose registers these are.
        if (!(T == 8 && last_slab)) {
          const unsigned need = (unsigned)s + (T == 8 ? 2u : 1u);   // weights of step s + 1: iteration s - 1; a new slab's halo image: iteration s
          unsigned have = (unsigned)__builtin_amdgcn_readfirstlane((int)min(min(fl[0], fl[1]), min(fl[2], fl[3])));
          while (have < need) { have = flags_min_now(pflags); DACC(1, 1); }
          DACC(2, 1);
        }
        issue_x(ic_t<0>{}, ic_t<nky>{}, ic_t<nkx>{}, hb_next, X[0]);
        static_for<0, 4>([&](auto ac) {
          mfma4(ac, ic_t<3>{}, X[1]);
          issue_w1(ac, wc_next);
          __builtin_amdgcn_sched_barrier(0);
        });
        ++s;
      });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the unused reads of the unit's last step
    DSTAMP(e0);
    epilogue(cur, (u + 1 - u0) & 1, has_next);
    DSTAMP(e1);
    DACC(3, e1 - e0); DACC(4, 1);
    if (has_next) {
      wait_producers((unsigned)s + 1u);
      DSTAMP(e2);
      DACC(5, e2 - e1);   // step s (first of the next unit) needs producer iteration s - 1
      const unsigned hb0 = (unsigned)((s / 9) & 1) * D_HB;
      issue_x(ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, hb0, X[0]);
      const unsigned wc0 = w_lane + (unsigned)(s & (D_NSLOT - 1)) * D_WSLOT;
      static_for<0, 4>([&](auto ac) { issue_w1(ac, wc0); });
      cur = nxt;
    }
  }
#ifdef C3D_STAMPS
  { DSTAMP(c_t2); dbg[6] = c_t2 - c_t0; if (blockIdx.x == 0 && wave == 0 && lane == 0) for (int i = 0; i < 16; ++i) c3d_dbg[i] = dbg[i]; }
#endif
}

int d_num_cus() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  int n = 0;
  HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
  return cus[dev] = n > 0 ? n : 256;
}

template <int FLAGS>
void launch_c3d(const ConvParams& p, hipStream_t s) {
  auto kern = conv3x3d_kernel<FLAGS>;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)D_LDS);
  const int units = p.B * (p.Hout >> 4) * (p.Wout >> 4) * (p.N >> 7);
  const int grid = units < d_num_cus() ? units : d_num_cus();
  const double bytes = (double)p.B * p.Hin * p.Win * p.C1 * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * 2.0 + (p.res ? (double)p.M * p.N * 2.0 : 0.0);
  ProfScope prof("conv3x3<16x16d,128,gn>", 2.0 * p.M * (double)p.N * p.K, bytes, s);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), D_LDS, s, p, units);
  HIP_CHECK(hipGetLastError());
}

}  // namespace

// LDIFF_CONV3X3_DATAFLOW: 0 = off, 1 (default) = where the unit list fills the chip, 2 = every eligible launch (tests, A/B timing)
bool conv3x3d_selected(const ConvParams& p) {
  static const int mode = [] { const char* e = getenv("LDIFF_CONV3X3_DATAFLOW"); return e ? atoi(e) : 1; }();
  if (mode == 0) return false;
  if (p.ks != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.ups != 0 || p.w_par || p.splitk > 1) return false;
  if (!p.gn_scale || !p.silu_in || p.x2 || p.C2 != 0 || p.C1 % 64 != 0 || p.N % 128 != 0 || p.Nrows < p.N) return false;
  if (p.Hout % 16 != 0 || p.Wout % 16 != 0 || p.Hin != p.Hout || p.Win != p.Wout) return false;
  if (p.out_f32 || p.y_lo || p.res_lo || (p.ldy & 7) || (p.res && (p.ld_res & 7))) return false;
  const long long px = (long long)p.B * p.Hin * p.Win;
  if (px * (p.ld1 ? p.ld1 : p.C1) * 2 >= (1LL << 31) || (long long)p.M * p.ldy * 2 >= (1LL << 31) || (p.res && (long long)p.M * p.ld_res * 2 >= (1LL << 31))) return false;
  if ((long long)p.Nrows * 9 * p.C1 * 2 >= (1LL << 31)) return false;
  const long long units = (long long)p.B * (p.Hout >> 4) * (p.Wout >> 4) * (p.N >> 7);
  if (mode == 2) return true;
  const int cus = d_num_cus();
  const long long rounds = (units + cus - 1) / cus;
  return units >= cus && units * 100 >= rounds * cus * 88;   // the runs must split evenly over the CUs
}
int conv3x3d_stats_blocks(const ConvParams& p) { return (p.Hout >> 4) * (p.Wout >> 4) * 2; }   // one row block per consumer wave pair (8 x 16 pixels)
#ifdef C3D_STAMPS
extern "C" int ldiff_debug_c3d_stamps(unsigned long long* out) {   // diagnostic build only
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c3d_dbg), sizeof(unsigned long long) * 48);
}
#endif
void launch_conv3x3d(const ConvParams& p, hipStream_t s) {
  const int f = (p.res ? D_RES : 0) | (p.stats ? D_STATS : 0);
  if (f == 0) launch_c3d<0>(p, s);
  else if (f == 1) launch_c3d<1>(p, s);
  else if (f == 2) launch_c3d<2>(p, s);
  else launch_c3d<3>(p, s);
}
