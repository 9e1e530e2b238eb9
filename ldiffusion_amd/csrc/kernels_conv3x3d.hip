// 3x3 stride-1 convolution with a GroupNorm + SiLU prologue on the large feature maps (the VAE decoder's resnet convs: 71 % of a patch's
// MACs, /root/reference/pixel_latent_vector.py:81 -> pipeline.decode_latents -> AutoencoderKL.decode; segmentor.py:106):
// producer / consumer wave specialisation inside one persistent 512-thread workgroup per CU ("dataflow" kernel).
//
// Why another conv3x3 kernel: in the 8x16 halo-tile kernel (kernels_conv3x3.hip) every wave does everything -- stages the halo through
// registers, normalises it, issues the weight DMA, runs the MFMAs, and meets the other three waves at a workgroup barrier once per tap.
// Its matrix pipe is busy 36-40 % of the time; the serial phases of a tile (first slab, epilogue) and the per-tap barrier are what is
// left (profiles/r02_conv3x3_pingpong.md).  Here the roles are split and nothing in the tap loop is synchronised:
//   * waves 0-3, one per SIMD: CONSUMERS.  Operand reads and MFMAs over a 16x16-pixel x 128-channel tile: each wave 8 pixel rows x 64
//     channels = 8x4 accumulator tiles (128 VGPRs).  A step = one tap of one 64-channel slab = 64 MFMAs per wave.  The step's WEIGHTS come
//     straight from global memory (L2) into registers: the matrix is stored fragment-packed per layer (launch_pack_frag_weights: every
//     v_mfma_f32_16x16x32_f16 A fragment of a wave is one contiguous KiB), eight global_load_dwordx4 per step issued one step ahead
//     between the MFMA groups, counted vmcnt -- no LDS ring, no DMA, no hand-over per step (scripts/micro/conv_consumer.hip: the matrix
//     rate is the same as with an LDS ring, and the producers lose three quarters of their vector-memory instructions).  The PIXELS come
//     from a swizzled halo image in LDS (ds_read_b128, counted lgkmcnt), rows in pairs one pair ahead, both k-halves of an accumulator
//     back to back.  The only synchronisation is one look at the producers' progress words per slab.
//   * waves 4-7, one per SIMD: PRODUCERS.  Free-running, up to two slabs ahead (three halo images): per slab, every raw halo piece
//     (8 pixels x 128 B) goes by LDS-DMA straight into its place in the image, one slab ahead of its transform; then GroupNorm-apply +
//     SiLU in place (ds_read -> common.h gn_quad: the v_fma_mix form, no packed-fp32 VALU beside the matrix stream -> mask -> ds_write).
//     Nothing asynchronous ever targets a VGPR on this side (the register allocator may copy an asm load's destination before the data
//     has landed wherever the value lives across control flow: measured).
//     The same waves run the EPILOGUE: a consumer only rounds its sums to fp16 and writes them to LDS (a staging area + the halo image its
//     unit's last slab has just released); producer wave w then stores what consumer wave w computed as full 128-byte lines -- residual
//     (read in the same shape, added in fp16) and the fused GroupNorm statistics of the stored values on its side -- while the consumers
//     are already in the next unit.  (The memory pipe works in lines: stores in the accumulator shape, 32 bytes per pixel and
//     instruction, cost 6 % of a plain layer and twice that with a residual.)  The fp16 hand-over rounds a residual layer's output twice.
//   * progress words (LDS): producer wave w publishes "slabs complete" and "tiles read from the staging area", consumer wave w "slabs
//     whose pixels are all in registers" and "tiles staged"; a producer may fill image k % 3 once every consumer has finished slab k - 3
//     (and, where that image carried a staged tile, once every producer has it in registers), a consumer starts slab k when every
//     producer has it.
// A workgroup walks a contiguous run of (pixel tile, channel tile) units (XCD-aware: the runs of one XCD's workgroups are adjacent);
// the slab sequence runs through unit boundaries on both sides.  Workgroups start staggered: in lock-step all 256 of them would hand
// over their output tiles in the same few microseconds.
//
// Scope (conv3x3d_selected): one source, Cin % 64 == 0, N % 128 == 0, H, W % 16 == 0, GroupNorm + SiLU prologue, plain fp16 output and
// residual (no split hi|lo tensors: the staged tile is fp16).  Everything else stays on the kernels of kernels_conv3x3.hip / kernels_conv3x3p.hip.
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
constexpr unsigned D_HB = 41 * 1024;                                   // one halo image = 41 DMA pieces of 8 pixels (324 pixels + 4 pad)
constexpr int D_NBUF = 3;                                              // halo images: one being read, two being built
constexpr unsigned D_FLAGS = D_NBUF * D_HB;                            // [producer slabs x4][consumer slabs x4][consumer tiles staged x4][producer tiles stored x4]
constexpr unsigned D_DUMP = D_FLAGS + 64;                              // 8 waves x 256 B: where the lanes other than 0 put their copy of a progress word
constexpr unsigned D_AFF = D_DUMP + 8 * 256;                           // [producer wave 4][image 3] x 512 B: scale (256 B) | shift (256 B) of a slab's 64 channels
constexpr unsigned D_BT = D_AFF + 12 * 512;                            // [unit parity 2] x (bias 128 floats | time embedding 128 floats) of a unit's channel tile
// Output staging: a unit's fp16 tile (16 chunks of 4 KiB: [channel tile a][consumer wave w][row pair][lane][16 B]) goes from the consumers
// to the producers through LDS: chunks 0-5 live here, chunks 6-15 in the halo image the unit's last slab has just released.
constexpr unsigned D_STG = D_BT + 2 * 1024;
constexpr int D_STG_CHUNKS = 6;
constexpr unsigned D_LDS = D_STG + D_STG_CHUNKS * 4096;
static_assert((16 - D_STG_CHUNKS) * 4096 <= D_HB, "the rest of a staged tile must fit one halo image");
static_assert(D_LDS <= 160 * 1024, "LDS budget of one workgroup per CU");
constexpr unsigned D_OOR = 0x80000000u;                                // beyond num_records of every descriptor used here: the load returns zeros
constexpr int D_NROUND = 11;                                           // pieces per producer wave and slab: 41 = 4 x 10 + 1
enum { D_RES = 1, D_STATS = 2, D_SC = 4, D_UPS = 8 };   // D_UPS: nearest-2x upsample folded into the conv (ConvParams::w_par): units per output parity, four taps, raw operand   // D_SC: the folded 1x1 shortcut (ConvParams::xs) -- a flag of its own so that the other instantiations carry none of its code

__device__ __forceinline__ int d_swzx(int hx) { return (0xcb5888 >> (3 * (hx >> 1))) & 7; }   // column swizzle of the halo image (kernels_conv3x3.hip)

// ---- LDS / memory primitives the compiler must not fence or wait for (counted by hand) ----
template <int OFF>
__device__ __forceinline__ void lds_read128(f16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT>
__device__ __forceinline__ void lds_wait4(f16x8& a, f16x8& b, f16x8& c, f16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(CNT));
}
template <int CNT>
__device__ __forceinline__ void vm_wait4(f16x8& a, f16x8& b, f16x8& c, f16x8& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(CNT));
}
template <int OFF>
__device__ __forceinline__ void glb_read128(f16x8& d, const char* addr) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ unsigned flags_min_now(unsigned addr) {   // read four progress words and wait for them
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  const unsigned m = min(min(v[0], v[1]), min(v[2], v[3]));
  return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}
__device__ __forceinline__ void lds_write32(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write128(unsigned addr, const u32x4& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// (a __builtin_bit_cast applied directly to a vector ELEMENT expression reads element 0 whatever the index: hipcc 7.2; by value it is fine)
__device__ __forceinline__ float u2f(unsigned v) { return __builtin_bit_cast(float, v); }

struct UnitC { int b, oy0, ox0, n0, q; };   // q: output parity 2 py + px of a D_UPS unit (else 0)

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
  const int nsx = (FLAGS & D_SC) ? p.Cs >> 6 : 0, nsl = nslab + nsx;   // slabs of the folded 1x1 shortcut (centre tap only), slabs of a unit in all
  constexpr bool UPS = (FLAGS & D_UPS) != 0;
  const int H = UPS ? p.Hin : p.Hout, W = UPS ? p.Win : p.Wout, tiles_x = W >> 4, tiles_y = H >> 4;   // the map the tiles and halos live on (D_UPS: the source; the output is 2H x 2W)

  // this workgroup's run of units (n-tile fastest, then x, y, image); the runs of the workgroups that share an XCD are adjacent
  const int G = gridDim.x, id = blockIdx.x;
  const int q8 = G >> 3, r8 = G & 7, xcd = id & 7, idx = id >> 3;
  const int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int u0 = (int)((long long)sw * units / G), u1 = (int)((long long)(sw + 1) * units / G);
  const int n_u = u1 - u0;
  const int total_slabs = n_u * nsl;
  auto decode = [&](int u) __attribute__((always_inline)) -> UnitC {   // (readfirstlane: descriptors and LDS-DMA bases built from these must be provably uniform)
    UnitC c;
    int t = u / ntn;
    c.n0 = __builtin_amdgcn_readfirstlane((u - t * ntn) * 128);
    c.q = 0;
    if constexpr (UPS) {   // the four parities of a pixel tile are neighbours in the unit list: their halos meet in L2
      const int tq = t >> 2;
      c.q = __builtin_amdgcn_readfirstlane(t - tq * 4);
      t = tq;
    }
    const int t2 = t / tiles_x;
    c.ox0 = __builtin_amdgcn_readfirstlane((t - t2 * tiles_x) * 16);
    const int t3 = t2 / tiles_y;
    c.oy0 = __builtin_amdgcn_readfirstlane((t2 - t3 * tiles_y) * 16);
    c.b = __builtin_amdgcn_readfirstlane(t3);
    return c;
  };

  // Stagger: workgroups that start together stay in lock-step (equal work per unit), and then all 256 of them run their epilogues -- the one
  // phase in which the matrix pipes idle -- and store their 64 KB output tiles at the same time.  One sixteenth of a unit's duration per
  // phase step spreads them evenly.
  if (n_u >= 4) {
    const int nsleep = ((id >> 3) & 15) * nslab;
    for (int k = 0; k < nsleep; ++k) __builtin_amdgcn_s_sleep(10);
  }
  if (tid < 16) *reinterpret_cast<volatile unsigned*>(smem_raw + D_FLAGS + tid * 4) = 0u;
  __syncthreads();
  if (n_u <= 0) return;

  if (wave >= 4) {
    // ================================================= PRODUCERS =================================================
    const int pw = wave - 4;
    // tensors that no cache level holds until their next use stream past L2 (non-temporal halo DMA, residual reads and stores: launch_c3d decides, ConvParams::nt_hint):
    // the weight fragments every unit re-reads stay resident instead
    const bool nt = p.nt_hint != 0;
    const int ld1 = p.ld1 ? p.ld1 : p.C1;
    const __amdgpu_buffer_rsrc_t scrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.gn_scale, 0, (int)((long long)p.B * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t shrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.gn_shift, 0, (int)((long long)p.B * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : p.gn_scale), 0, p.bias ? p.Nrows * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.temb ? p.temb : p.gn_scale), 0, p.temb ? (int)((long long)p.B * p.ld_temb * 4) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((long long)p.M * p.ldy * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : (const f16*)p.y), 0, (int)((long long)p.M * (p.res ? p.ld_res : p.ldy) * 2), 0x00020000);
    const unsigned cflags = lds0 + D_FLAGS + 16u, eflags = lds0 + D_FLAGS + 32u, dflags = lds0 + D_FLAGS + 48u;
    const unsigned dflag_addr = lane == 0 ? lds0 + D_FLAGS + 48u + (unsigned)pw * 4u : lds0 + D_DUMP + (unsigned)(4 + pw) * 256u + (unsigned)lane * 4u;
    // progress word by ONE unmasked ds_write_b32: lane 0 hits the word, the other lanes a dump row of their own
    const unsigned pflag_addr = lane == 0 ? lds0 + D_FLAGS + (unsigned)pw * 4u : lds0 + D_DUMP + (unsigned)(4 + pw) * 256u + (unsigned)lane * 4u;

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
    // A slab's coordinates: which of this lane's halo pixels lie inside the image (bit r), and a descriptor whose base is halo pixel (0, 0) of
    // the unit (it may lie in front of the tensor for border tiles: only lanes whose pixel is inside are ever given a real offset).
    struct SlabC { int k, c, u; UnitC un; unsigned vbits; __amdgpu_buffer_rsrc_t xrsrc, xrsrc2; };   // (xrsrc2: D_SC only)
    const int lds2 = p.lds ? p.lds : p.Cs;
    auto set_unit = [&](SlabC& sc) __attribute__((always_inline)) {
      sc.un = decode(sc.u);
      unsigned vb = 0;
#pragma unroll
      for (int r = 0; r < D_NROUND; ++r) {
        const int hy = (int)(rc_yx[r] >> 8), hx = (int)(rc_yx[r] & 0xff);
        const bool inb = rc_yx[r] != 0xffffu && (unsigned)(sc.un.oy0 + hy - 1) < (unsigned)H && (unsigned)(sc.un.ox0 + hx - 1) < (unsigned)W;
        vb |= inb ? 1u << r : 0u;
      }
      sc.vbits = vb;
      const long long org = ((long long)(sc.un.b * H + sc.un.oy0 - 1) * W + sc.un.ox0 - 1) * ld1 * 2;
      const unsigned long long xa = (unsigned long long)reinterpret_cast<const char*>(p.x) + (unsigned long long)org;
      const unsigned xlo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xa), xhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xa >> 32));
      sc.xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)xhi << 32) | xlo), 0, 0x7fffffff, 0x00020000);
      if constexpr ((FLAGS & D_SC) != 0) {   // the folded shortcut's source: same pixels, its own pitch
        const long long org2 = ((long long)(sc.un.b * H + sc.un.oy0 - 1) * W + sc.un.ox0 - 1) * lds2 * 2;
        const unsigned long long xa2 = (unsigned long long)reinterpret_cast<const char*>(p.xs) + (unsigned long long)org2;
        const unsigned xlo2 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xa2), xhi2 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xa2 >> 32));
        sc.xrsrc2 = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)xhi2 << 32) | xlo2), 0, 0x7fffffff, 0x00020000);
      }
    };
    auto advance = [&](SlabC& sc) __attribute__((always_inline)) {   // next slab of the run; the unit's constants are rebuilt when it changes
      ++sc.k;
      if (++sc.c == nsl) { sc.c = 0; ++sc.u; if (sc.u < u1) set_unit(sc); }
    };
    // everything slab `sc` needs from memory, as LDS-DMA: the scale / shift table (2 instructions), the bias / time-embedding table when the
    // slab opens a unit (2, producer wave 0), the raw halo pieces (10, or 11 for producer wave 0).  Returns the number of instructions issued.
    auto issue_slab = [&](const SlabC& sc) __attribute__((always_inline)) -> int {
      const unsigned hbuf = (unsigned)(sc.k % D_NBUF) * D_HB;
      int n = 2;
      {
        const int voff = (!UPS && lane < 32 && ((FLAGS & D_SC) == 0 || sc.c < nslab)) ? (sc.un.b * Cin + sc.c * 64) * 4 + (lane & 15) * 16 : (int)D_OOR;   // (a shortcut slab: zeros, unused -- the instruction count of a slab stays the same)
        unsigned char* dst = smem_raw + D_AFF + (unsigned)(pw * 3 + sc.k % D_NBUF) * 512u;
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass rejects this builtin (target feature) and then silently drops the kernel stub
        if (lane < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(scrsrc, (lptr_t*)dst, 16, voff, 0, 0, 0);
        else if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(shrsrc, (lptr_t*)dst, 16, voff, 0, 0, 0);
#endif
      }
      if (pw == 0 && sc.c == 0 && sc.u > u0) {   // the consumers start this unit's sums from the table (the first unit's they load themselves)
        unsigned char* dst = smem_raw + D_BT + (unsigned)((sc.u - u0) & 1) * 1024u;
#if defined(__HIP_DEVICE_COMPILE__)
        if (lane < 32) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(brsrc, (lptr_t*)dst, 16, (sc.un.n0 + lane * 4) * 4, 0, 0, 0);            // no bias: zero records, the DMA writes zeros
          __builtin_amdgcn_raw_ptr_buffer_load_lds(trsrc, (lptr_t*)(dst + 512), 16, (sc.un.b * p.ld_temb + sc.un.n0 + lane * 4) * 4, 0, 0, 0);
        }
#endif
        n += 2;
      }
      const bool extra = (FLAGS & D_SC) != 0 && sc.c >= nslab;   // a slab of the folded shortcut: the other source, its own pitch (offsets rebuilt from the pixel coordinates)
      static_for<0, D_NROUND>([&](auto rc_) {
        constexpr int r = decltype(rc_)::value;
#ifdef C3D_ABL_NODMA   // ablation (timing only, results wrong): only the first three halo images are fetched
        if (sc.k >= 3) return;
#endif
        if (r < 10 || pw == 0) {
          int voff = ((sc.vbits >> r) & 1u) ? (int)rc_rel[r] : (int)D_OOR;
          unsigned char* dst = smem_raw + hbuf + (unsigned)(pw + 4 * r) * 1024u;
#if defined(__HIP_DEVICE_COMPILE__)
          if (extra) {
            const int hy = (int)(rc_yx[r] >> 8), hx = (int)(rc_yx[r] & 0xff);
            if ((sc.vbits >> r) & 1u) voff = ((hy * W + hx) * lds2 + (((lane & 7) ^ d_swzx(hx)) << 3)) * 2;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(sc.xrsrc2, (lptr_t*)dst, 16, voff, (sc.c - nslab) * 128, 0, 0);
          } else
            { if (nt) __builtin_amdgcn_raw_ptr_buffer_load_lds(sc.xrsrc, (lptr_t*)dst, 16, voff, sc.c * 128, 0, 2); else __builtin_amdgcn_raw_ptr_buffer_load_lds(sc.xrsrc, (lptr_t*)dst, 16, voff, sc.c * 128, 0, 0); }
#endif
          ++n;
        }
      });
      return n;
    };
    // normalise slab `sc`'s image in place (its pieces have landed)
    auto xform_slab = [&](const SlabC& sc) __attribute__((always_inline)) {
      const unsigned hbuf = (unsigned)(sc.k % D_NBUF) * D_HB;
      float gs[8], gt[8];   // scale / shift of channels [8 (l & 7), + 8) of the slab
      {
        const unsigned taddr = lds0 + D_AFF + (unsigned)(pw * 3 + sc.k % D_NBUF) * 512u + (unsigned)(lane & 7) * 32u;
        u32x4 sc0, sc1, sh0, sh1;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:256\n\tds_read_b128 %3, %4 offset:272\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(sc0), "=&v"(sc1), "=&v"(sh0), "=&v"(sh1) : "v"(taddr) : "memory");
        gs[0] = u2f(sc0[0]); gs[1] = u2f(sc0[1]); gs[2] = u2f(sc0[2]); gs[3] = u2f(sc0[3]); gs[4] = u2f(sc1[0]); gs[5] = u2f(sc1[1]); gs[6] = u2f(sc1[2]); gs[7] = u2f(sc1[3]);
        gt[0] = u2f(sh0[0]); gt[1] = u2f(sh0[1]); gt[2] = u2f(sh0[2]); gt[3] = u2f(sh0[3]); gt[4] = u2f(sh1[0]); gt[5] = u2f(sh1[1]); gt[6] = u2f(sh1[2]); gt[7] = u2f(sh1[3]);
      }
      auto one = [&](const u32x4& x, unsigned keep) __attribute__((always_inline)) -> u32x4 {
        u32x4 v;
        unsigned oa, ob;
        gn_quad<true>(x[0], x[1], gs[0], gt[0], gs[1], gt[1], gs[2], gt[2], gs[3], gt[3], oa, ob);
        v[0] = oa & keep; v[1] = ob & keep;   // zero padding applies to the NORMALISED tensor
        gn_quad<true>(x[2], x[3], gs[4], gt[4], gs[5], gt[5], gs[6], gt[6], gs[7], gt[7], oa, ob);
        v[2] = oa & keep; v[3] = ob & keep;
        return v;
      };
      static_for<0, 5>([&](auto gc) {   // two rounds at a time: both reads, then the arithmetic (one wave alone cannot hide an LDS round trip per round)
        constexpr int r0 = 2 * decltype(gc)::value, r1 = r0 + 1;
        const unsigned a0 = rc_lds[r0] + hbuf, a1 = rc_lds[r1] + hbuf;
        u32x4 x0, x1;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0), "=&v"(x1) : "v"(a0), "v"(a1) : "memory");
        lds_write128(a0, one(x0, (unsigned)__builtin_amdgcn_sbfe((int)sc.vbits, r0, 1)));
        lds_write128(a1, one(x1, (unsigned)__builtin_amdgcn_sbfe((int)sc.vbits, r1, 1)));
      });
      if (pw == 0) {
        const unsigned a0 = rc_lds[10] + hbuf;
        u32x4 x0;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0) : "v"(a0) : "memory");
        lds_write128(a0, one(x0, (unsigned)__builtin_amdgcn_sbfe((int)sc.vbits, 10, 1)));
      }
    };

#ifdef C3D_STAMPS
    unsigned long long dbg[32] = {0};
#endif
    // Epilogue of local unit i, from the tile the consumers staged (D_STG + the halo image `hb_last` of the unit's last slab): producer wave
    // pw stores the 8 rows x 16 columns x 64 channels consumer wave pw computed, in FULL 128-byte lines: a staged 16-byte item = 8
    // consecutive channels ("octet" o = 2 a + gh of the wave's 64) of one pixel, and lane (px = lane >> 3, o = lane & 7) of store j takes
    // octet o of pixel (row j >> 1, column 8 (j & 1) + px): eight lanes = one line, one instruction = eight lines.  (Storing the items
    // in the accumulator shape -- 32 bytes per pixel and instruction -- costs 6 % of the layer: the memory pipe works in lines.)
    // The consumers rotate an item's 16-byte slot inside its 256-byte staging row by 2 o, so that the sixteen lanes of a read phase
    // (2 pixels x 8 octets) hit sixteen different slots.  Residual: read in the same shape, added in fp16 (v_pk_add_f16 = the exactly
    // rounded sum of the two fp16 values).  Fused GroupNorm statistics of the stored values: a lane's octet is the same in all sixteen
    // stores, so 8 sums + 8 sums of squares accumulate per lane and are reduced over the eight pixel lanes in three halving steps
    // (row_ror:8, v_permlane16_swap, v_permlane32_swap): every lane ends with (sum, sum of squares) of ONE channel: one 8-byte store.
    const int e_wm = pw >> 1, e_wn = pw & 1, e_px = lane >> 3, e_o = lane & 7;
    auto store_unit = [&](int i, unsigned hb_last) __attribute__((always_inline)) {
      constexpr bool RES = (FLAGS & D_RES) != 0, ST = (FLAGS & D_STATS) != 0;
      const UnitC un = decode(u0 + i);
      // first pixel of the wave's 8 x 16 block; D_UPS: source pixel (Y, X) of parity (py, px) is output pixel (2 Y + py, 2 X + px) of the 2H x 2W map
      const unsigned pix0 = UPS ? (unsigned)((un.b * 2 * H + 2 * (un.oy0 + e_wm * 8) + (un.q >> 1)) * 2 * W + 2 * un.ox0 + (un.q & 1))
                                : (unsigned)((un.b * H + un.oy0 + e_wm * 8) * W + un.ox0);
      const unsigned ch0 = (unsigned)(un.n0 + e_wn * 64);
      const int ypx = UPS ? 2 : 1, yrow = UPS ? 4 * W : W;   // output pixels per tile column / per tile row
      const int ylane = (e_px * ypx * p.ldy + e_o * 8) * 2, rlane = (e_px * p.ld_res + e_o * 8) * 2;
      const unsigned ybase = (pix0 * (unsigned)p.ldy + ch0) * 2u, rbase = (pix0 * (unsigned)p.ld_res + ch0) * 2u;
      u32x4 R[16];
      if constexpr (RES) {   // (nothing of this wave's is in flight here: the compiler's own waits for these loads are exact)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int ro = rlane + (int)(rbase + (unsigned)(((j >> 1) * W + (j & 1) * 8) * p.ld_res) * 2u);
          R[j] = nt ? __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ro, 0, 2) : __builtin_amdgcn_raw_buffer_load_b128(rrsrc, ro, 0, 0);
        }
      }
      while ((int)flags_min_now(eflags) < i + 1) { __builtin_amdgcn_s_sleep(2); DACC(8, 1); }
      u32x4 U[16];
      {
        const int idx = (e_o >> 1) * 4 + pw;   // chunk of this lane's channel tile
        const unsigned cb = (idx < D_STG_CHUNKS ? lds0 + D_STG + (unsigned)idx * 4096u : lds0 + hb_last + (unsigned)(idx - D_STG_CHUNKS) * 4096u) + (unsigned)(e_o & 1) * 512u;
        const unsigned a0 = cb + (unsigned)((e_px + 2 * e_o) & 15) * 16u, a8 = cb + (unsigned)((8 + e_px + 2 * e_o) & 15) * 16u;
        // row rr = j >> 1 of the wave's eight = row pair rr >> 1 (1 KiB), row of the pair rr & 1 (256 B)
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\tds_read_b128 %2, %8 offset:256\n\tds_read_b128 %3, %9 offset:256\n\t"
                     "ds_read_b128 %4, %8 offset:1024\n\tds_read_b128 %5, %9 offset:1024\n\tds_read_b128 %6, %8 offset:1280\n\tds_read_b128 %7, %9 offset:1280"
                     : "=&v"(U[0]), "=&v"(U[1]), "=&v"(U[2]), "=&v"(U[3]), "=&v"(U[4]), "=&v"(U[5]), "=&v"(U[6]), "=&v"(U[7]) : "v"(a0), "v"(a8) : "memory");
        asm volatile("ds_read_b128 %0, %8 offset:2048\n\tds_read_b128 %1, %9 offset:2048\n\tds_read_b128 %2, %8 offset:2304\n\tds_read_b128 %3, %9 offset:2304\n\t"
                     "ds_read_b128 %4, %8 offset:3072\n\tds_read_b128 %5, %9 offset:3072\n\tds_read_b128 %6, %8 offset:3328\n\tds_read_b128 %7, %9 offset:3328"
                     : "=&v"(U[8]), "=&v"(U[9]), "=&v"(U[10]), "=&v"(U[11]), "=&v"(U[12]), "=&v"(U[13]), "=&v"(U[14]), "=&v"(U[15]) : "v"(a0), "v"(a8) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(U[0]), "+v"(U[1]), "+v"(U[2]), "+v"(U[3]), "+v"(U[4]), "+v"(U[5]), "+v"(U[6]), "+v"(U[7]),
                                            "+v"(U[8]), "+v"(U[9]), "+v"(U[10]), "+v"(U[11]), "+v"(U[12]), "+v"(U[13]), "+v"(U[14]), "+v"(U[15]));
      lds_write32(dflag_addr, (unsigned)i + 1u);   // the staged tile is in registers: its LDS is free (the halo image once all four producers say so)
      float x16[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) x16[j] = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        u32x4 o = U[j];
        if constexpr (RES) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const unsigned ud = o[d], rd = R[j][d];
            const f16x2 sum = __builtin_bit_cast(f16x2, ud) + __builtin_bit_cast(f16x2, rd);
            o[d] = __builtin_bit_cast(unsigned, sum);
          }
        }
        // (offset in the VGPR and a wait state behind the store: with an SGPR soffset a VALU write of the data registers right after a
        // 16-byte store is seen by the store -- profiles/r02_conv3x3_pingpong.md)
        {
          const int so = ylane + (int)(ybase + (unsigned)(((j >> 1) * yrow + (j & 1) * 8 * ypx) * p.ldy) * 2u);
          if (nt) __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, so, 0, 2); else __builtin_amdgcn_raw_buffer_store_b128(o, yrsrc, so, 0, 0);
        }
        asm volatile("s_nop 1" ::: "memory");
        if constexpr (ST) {
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const unsigned od = o[d];
            const f16x2 h = __builtin_bit_cast(f16x2, od);
            const float f0 = (float)h[0], f1 = (float)h[1];
            x16[2 * d] += f0; x16[2 * d + 1] += f1;
            x16[8 + 2 * d] += f0 * f0; x16[8 + 2 * d + 1] += f1 * f1;
          }
        }
      }
      if constexpr (ST) {
        // halving steps over the pixel lanes (lane bits 3, 4, 5); value index bit b is decided at step b, bit 3 (sum / sum of squares) stays
        const bool b0 = (lane & 8) != 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float keep = b0 ? x16[2 * q + 1] : x16[2 * q], send = b0 ? x16[2 * q] : x16[2 * q + 1];
          x16[q] = keep + dpp_f<0x128>(send);   // row_ror:8 = lane ^ 8
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {   // swap(odd, even): rows 0, 2 end with both copies of the ODD value, rows 1, 3 with both of the EVEN one
          auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x16[2 * q + 1]), __builtin_bit_cast(unsigned, x16[2 * q]), false, false);
          x16[q] = u2f(sw[0]) + u2f(sw[1]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {   // the same between the wave's halves
          auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x16[2 * q + 1]), __builtin_bit_cast(unsigned, x16[2 * q]), false, false);
          x16[q] = u2f(sw[0]) + u2f(sw[1]);
        }
        const int chn = ((lane >> 3) & 1) | ((((lane >> 4) & 1) ^ 1) << 1) | ((((lane >> 5) & 1) ^ 1) << 2);
        const long long rblk = (((long long)(un.oy0 >> 4) * tiles_x + (un.ox0 >> 4)) * 2 + e_wm) * (UPS ? 4 : 1) + un.q;
        const int n = (int)ch0 + e_o * 8 + chn;
        *reinterpret_cast<float2*>(p.stats + (((long long)un.b * p.N + n) * p.stats_R + rblk) * 2) = make_float2(x16[0], x16[1]);
      }
    };

    DSTAMP(p_t0);
    // Software pipeline over the run's slabs: the pieces of slab k + 1 are in flight while slab k is normalised.  Iteration k is also where
    // the tile of the unit whose LAST slab was k - 2 is stored (nslab >= 2: at most every other iteration): its staged rows occupy image
    // (k + 1) % 3, the one slab k + 1 goes to, so that slab's pieces are issued behind the stores instead of in front of the transform --
    // the consumers are a whole slab (nine steps) away from needing it.
    SlabC cur, nxt;
    cur.k = 0; cur.c = 0; cur.u = u0; set_unit(cur);
    issue_slab(cur);
    nxt = cur;
    int stored = 0;   // units whose tile has been stored
    for (int k = 0; k < total_slabs; ++k) {
      advance(nxt);
      const bool store_here = k >= 2 && (k - 1) % nsl == 0;
      int n_next = 0;
      DSTAMP(q0);
      if (!store_here && nxt.k < total_slabs) {
        // image (k + 1) % 3 held slab k - 2: free once every consumer has all of that slab's pixels in registers
        while ((int)flags_min_now(cflags) < k - 1) { __builtin_amdgcn_s_sleep(2); DACC(5, 1); }
        n_next = issue_slab(nxt);
      }
      DSTAMP(q1);
      // slab k's pieces are older than everything just issued: leave exactly those in flight
      if (n_next >= 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
      else if (n_next >= 14) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      else if (n_next >= 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
      else if (n_next >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      DSTAMP(q2);
#ifndef C3D_ABL_NOXF
      if constexpr (!UPS)   // (D_UPS: the conv behind an Upsample2D has no GroupNorm in front of it -- raw operand; zero padding = the DMA's zeros)
        if ((FLAGS & D_SC) == 0 || cur.c < nslab) xform_slab(cur);   // (a slab of the folded shortcut is a raw operand)
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      lds_write32(pflag_addr, (unsigned)k + 1u);
      DSTAMP(q3);
      if (store_here) {
        const int i = (k - 1) / nsl - 1;
#ifndef C3D_ABL_NOSTORE
        store_unit(i, (unsigned)((k + 1) % D_NBUF) * D_HB);
#else
        lds_write32(dflag_addr, (unsigned)i + 1u);
#endif
        stored = i + 1;
        if (nxt.k < total_slabs) {
          while ((int)flags_min_now(dflags) < i + 1) { __builtin_amdgcn_s_sleep(1); DACC(9, 1); }   // every producer has its part of the staged tile in registers
          issue_slab(nxt);
        }
      }
      DSTAMP(q4);
      DACC(1, q1 - q0); DACC(2, q2 - q1); DACC(3, q3 - q2); DACC(4, q4 - q3); DACC(6, 1);
      cur = nxt;
    }
#ifndef C3D_ABL_NOSTORE
    for (int i = stored; i < n_u; ++i) store_unit(i, (unsigned)(((i + 1) * nsl - 1) % D_NBUF) * D_HB);   // the last unit (one-slab units: the last two)
#endif
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
  const unsigned pflags = lds0 + D_FLAGS;
  const unsigned cflag_addr = lane == 0 ? lds0 + D_FLAGS + 16u + (unsigned)wave * 4u : lds0 + D_DUMP + (unsigned)wave * 256u + (unsigned)lane * 4u;
  // fragment-packed weights: [tap][slab][channel tile of 128][wave_n][a][k-half][lane][16 B]: a step's eight fragments of this wave are 8 KiB in a row
  const char* wfrag = reinterpret_cast<const char*>(p.w_frag) + wave_n * 8192 + lane * 16;
#ifdef C3D_ABL_NOW
  bool abl_w_loaded[2] = {false, false};
#endif
#ifdef C3D_ABL_NOX
  int abl_nx = 0;
#endif
  const long long w_step_bytes = (long long)ntn * 16384;   // from slab c to slab c + 1 of a tap; a tap is nslab of these

#ifdef C3D_STAMPS
  unsigned long long dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  f32x4 acc[4][8];
  f16x8 Wf[2][4], X[2][4];   // Wf[k-half][channel tile]; X[buffer][row of the group of four]

  // A step (tap of a slab) runs as two HALF-steps, one per k-half, each over two groups of four pixel rows.  The weights of a half-step are
  // loaded one half-step ahead (its registers are the ones the half-step before it has just released: ~512 matrix cycles of lead for an L2
  // round trip), the pixels of a group one group ahead (~256 cycles for an LDS round trip).
  auto issue_x = [&](auto gc, auto kyc, auto kxc, auto khc, unsigned hb, f16x8 (&dst)[4]) __attribute__((always_inline)) {   // rows 4 gr .. 4 gr + 3 of k-half kh at tap (ky, kx)
    constexpr int gr = decltype(gc)::value, ky = decltype(kyc)::value, kx = decltype(kxc)::value, kh = decltype(khc)::value;
    const unsigned b0 = (xb[kx] + hb) ^ (kh ? 64u : 0u);   // chunk bit 2 = k-half: XOR commutes with the swizzle
#ifdef C3D_ABL_NOX   // ablation (timing only, results wrong): only the first eight groups of pixel rows are read
    if (abl_nx >= 8) return;
    ++abl_nx;
#endif
    lds_read128<(4 * gr + ky) * D_ROWB>(dst[0], b0);
    lds_read128<(4 * gr + 1 + ky) * D_ROWB>(dst[1], b0);
    lds_read128<(4 * gr + 2 + ky) * D_ROWB>(dst[2], b0);
    lds_read128<(4 * gr + 3 + ky) * D_ROWB>(dst[3], b0);
  };
  // D_UPS: the same four rows at tap row ty of the column base `base` (= xb[column tap] + parity row + image; chosen per unit, not per instantiation)
  auto issue_xb = [&](auto gc, auto tyc, auto khc, unsigned base, f16x8 (&dst)[4]) __attribute__((always_inline)) {
    constexpr int gr = decltype(gc)::value, ty = decltype(tyc)::value, kh = decltype(khc)::value;
    const unsigned b0 = base ^ (kh ? 64u : 0u);
    lds_read128<(4 * gr + ty) * D_ROWB>(dst[0], b0);
    lds_read128<(4 * gr + 1 + ty) * D_ROWB>(dst[1], b0);
    lds_read128<(4 * gr + 2 + ty) * D_ROWB>(dst[2], b0);
    lds_read128<(4 * gr + 3 + ty) * D_ROWB>(dst[3], b0);
  };
  // D_UPS: output parity (py, px) reads source rows Y - 1 + py + ty and columns X - 1 + px + tx, ty, tx in {0, 1} = halo rows py + ty, halo columns px + tx;
  // fragment-packed weights [parity][tap 2 ty + tx][slab][channel tile] (launch_pack_frag_weights_par)
  unsigned ub[2] = {xb[0], xb[1]};
  auto set_parity = [&](const UnitC& un) __attribute__((always_inline)) {
    const unsigned ro = (unsigned)((un.q >> 1) * D_ROWB);
    ub[0] = ((un.q & 1) ? xb[1] : xb[0]) + ro;
    ub[1] = ((un.q & 1) ? xb[2] : xb[1]) + ro;
  };
  auto wbase = [&](const UnitC& un) __attribute__((always_inline)) -> const char* {
    return wfrag + ((long long)(UPS ? un.q * 4 * nslab * ntn : 0) + (un.n0 >> 7)) * 16384;
  };
  auto issue_w = [&](auto khc, const char* wq) __attribute__((always_inline)) {   // the four channel tiles of k-half kh of the step at wq
    constexpr int kh = decltype(khc)::value;
#ifdef C3D_ABL_NOW   // ablation (timing only, results wrong): each k-half's weight fragments are loaded ONCE
    if (abl_w_loaded[kh]) return;
    abl_w_loaded[kh] = true;
#endif
    glb_read128<kh * 1024>(Wf[kh][0], wq);          // (the instruction offset is 13 bits signed: channel tiles 2, 3 through a second base)
    glb_read128<2048 + kh * 1024>(Wf[kh][1], wq);
    glb_read128<kh * 1024>(Wf[kh][2], wq + 4096);
    glb_read128<2048 + kh * 1024>(Wf[kh][3], wq + 4096);
  };
  auto mfma16 = [&](auto khc, auto gc, f16x8 (&x)[4]) __attribute__((always_inline)) {
    constexpr int kh = decltype(khc)::value, gr = decltype(gc)::value;
#ifdef C3D_ABL_PAIRS   // issue-order probe (timing only, results wrong): the sixteen MFMAs of a half-step as eight DEPENDENT pairs (profiles/r02_mfma_peak.md)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int a = 2 * kh; a < 2 * kh + 2; ++a) {
        acc[a][4 * gr + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[0][a], x[r], acc[a][4 * gr + r], 0, 0, 0);
        acc[a][4 * gr + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[1][a], x[r], acc[a][4 * gr + r], 0, 0, 0);
      }
    return;
#endif
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        acc[a][4 * gr + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[kh][a], x[r], acc[a][4 * gr + r], 0, 0, 0);
#ifdef C3D_MFMA_GAP   // experiment: issue slots between matrix instructions (profiles/r02_mfma_peak.md)
        __builtin_amdgcn_sched_barrier(0);
        for (int q = 0; q < C3D_MFMA_GAP; ++q) asm volatile("s_nop 0");
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
  };
  auto wait_producers = [&](unsigned need) __attribute__((always_inline)) {
    while (flags_min_now(pflags) < need) { DACC(1, 1); }
  };

  // End of a unit: the consumer only ROUNDS its sums to fp16 and hands them to the producers through LDS (store_unit above) -- 16 ds_write_b128
  // per lane instead of a residual read, the statistics and 64 KB of stores in front of the next unit's first MFMA.  Lane holds
  // y[pixel = (row m, column l15)][channels 4g .. 4g+3 of channel tile a]; one v_permlane16_swap per dword between the packed values of two
  // rows P = 2pr, Q = 2pr + 1 leaves lanes with even g holding channels 4g .. 4g+7 of row P and lanes with odd g channels 4(g-1) .. 4(g-1)+7
  // of row Q: the 16-byte items the producers store as they are.  Chunk a * 4 + wave: the first D_STG_CHUNKS in the staging area (free once
  // the producers hold the previous tile in registers), the rest in the halo image of the unit's last slab (free once every consumer has
  // read its last pixels).
  const unsigned cflags_all = lds0 + D_FLAGS + 16u, dflags = lds0 + D_FLAGS + 48u;
  const unsigned eflag_addr = lane == 0 ? lds0 + D_FLAGS + 32u + (unsigned)wave * 4u : lds0 + D_DUMP + (unsigned)wave * 256u + (unsigned)lane * 4u;
  auto stage_ready = [&](int i, unsigned slabs_done) __attribute__((always_inline)) {
    while ((int)flags_min_now(dflags) < i) { DACC(7, 1); }
    while (flags_min_now(cflags_all) < slabs_done) { DACC(8, 1); }
  };
  auto stage_tile = [&](int i, unsigned hb_last) __attribute__((always_inline)) {   // straight-line: the next unit's weights are in flight across it
    static_for<0, 4>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      const int idx = a * 4 + wave;
      // 256-byte row g of the row pair; the item's slot in it rotated by twice its octet (store_unit reads two pixels x eight octets at a time)
      const unsigned base = (idx < D_STG_CHUNKS ? lds0 + D_STG + (unsigned)idx * 4096u : lds0 + hb_last + (unsigned)(idx - D_STG_CHUNKS) * 4096u) +
                            (unsigned)g * 256u + (unsigned)((l15 + 4 * a + 2 * (g >> 1)) & 15) * 16u;
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const f16x4 o0 = cvt4(acc[a][2 * pr]), o1 = cvt4(acc[a][2 * pr + 1]);
        const uint2 q0 = __builtin_bit_cast(uint2, o0), q1 = __builtin_bit_cast(uint2, o1);
        auto r0 = __builtin_amdgcn_permlane16_swap(q0.x, q1.x, false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(q0.y, q1.y, false, false);
        lds_write128(base + (unsigned)pr * 1024u, (u32x4){r0[0], r1[0], r0[1], r1[1]});
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    lds_write32(eflag_addr, (unsigned)i + 1u);
  };
  // a unit's sums start at bias + time embedding: from the table producer wave 0 wrote with the unit's first slab
  auto init_from_table = [&](int parity) __attribute__((always_inline)) {
    const unsigned taddr = lds0 + D_BT + (unsigned)parity * 1024u + (unsigned)(wave_n * 64 + g * 4) * 4u;
    f32x4 bb[4], tt[4];
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:64\n\tds_read_b128 %2, %8 offset:128\n\tds_read_b128 %3, %8 offset:192\n\t"
                 "ds_read_b128 %4, %8 offset:512\n\tds_read_b128 %5, %8 offset:576\n\tds_read_b128 %6, %8 offset:640\n\tds_read_b128 %7, %8 offset:704\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(bb[0]), "=&v"(bb[1]), "=&v"(bb[2]), "=&v"(bb[3]), "=&v"(tt[0]), "=&v"(tt[1]), "=&v"(tt[2]), "=&v"(tt[3]) : "v"(taddr) : "memory");
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const f32x4 b = bb[a] + tt[a];
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[a][m] = b;
    }
  };

  DSTAMP(c_t0);
  // ---- prologue ----
  UnitC cur = decode(u0);
  {   // the first unit's bias + time embedding straight from memory
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int n = cur.n0 + wave_n * 64 + a * 16 + g * 4;
      float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), tt = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + n);
      if (p.temb) tt = *reinterpret_cast<const float4*>(p.temb + (long long)cur.b * p.ld_temb + n);
      const f32x4 b = (f32x4){bb.x + tt.x, bb.y + tt.y, bb.z + tt.z, bb.w + tt.w};
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[a][m] = b;
    }
  }
  const char* wq = wbase(cur);   // weights of the current step (tap 0, slab 0 of the unit's channel tile)
  if constexpr (UPS) set_parity(cur);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (nothing of the compiler's may be younger than the hand-counted loads below)
  wait_producers(1u);
  DSTAMP(c_t1);
  DACC(0, c_t1 - c_t0);
  if constexpr (UPS) issue_xb(ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, ub[0], X[0]);
  else issue_x(ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, 0u, X[0]);
  issue_w(ic_t<0>{}, wq);

  int k = 0;   // global slab of this workgroup
  for (int u = u0; u < u1; ++u) {
    const bool has_next = u + 1 < u1;
    // the folded shortcut's weights of this unit's channel tile: behind the nine taps, one 16-KiB block per (slab, channel tile)
    const char* wx0 = wfrag + (long long)9 * nslab * w_step_bytes + (long long)(cur.n0 >> 7) * 16384;
    for (int c = 0; c < nslab; ++c, ++k) {
      const bool last_slab = c == nsl - 1;
      const bool to_extra = (FLAGS & D_SC) != 0 && nsx > 0 && c == nslab - 1;   // the next slab is the shortcut's first: its one step is the centre tap
      const unsigned hb = (unsigned)(k % D_NBUF) * D_HB;
      const unsigned hbn = (unsigned)((k + 1) % D_NBUF) * D_HB;
      if constexpr (UPS) {
        static_for<0, 4>([&](auto tc) {   // the four taps of the unit's parity: the step of the nine-tap loop below with a column base chosen per unit
          constexpr int T = decltype(tc)::value, ty = T >> 1, tx = T & 1;
          constexpr int nty = ((T + 1) & 3) >> 1, ntx = (T + 1) & 1;
          const char* wn = T == 3 ? wq - (3 * nslab - 1) * w_step_bytes : wq + nslab * w_step_bytes;
          const unsigned xc = ub[tx] + hb;
          issue_w(ic_t<1>{}, wq);
          issue_xb(ic_t<1>{}, ic_t<ty>{}, ic_t<0>{}, xc, X[1]);
          lds_wait4<4>(X[0][0], X[0][1], X[0][2], X[0][3]);
          vm_wait4<4>(Wf[0][0], Wf[0][1], Wf[0][2], Wf[0][3]);
          mfma16(ic_t<0>{}, ic_t<0>{}, X[0]);
          __builtin_amdgcn_sched_barrier(0);
          issue_xb(ic_t<0>{}, ic_t<ty>{}, ic_t<1>{}, xc, X[0]);
          lds_wait4<4>(X[1][0], X[1][1], X[1][2], X[1][3]);
          mfma16(ic_t<0>{}, ic_t<1>{}, X[1]);
          __builtin_amdgcn_sched_barrier(0);
          issue_w(ic_t<0>{}, wn);
          issue_xb(ic_t<1>{}, ic_t<ty>{}, ic_t<1>{}, xc, X[1]);
          lds_wait4<4>(X[0][0], X[0][1], X[0][2], X[0][3]);
          vm_wait4<4>(Wf[1][0], Wf[1][1], Wf[1][2], Wf[1][3]);
          mfma16(ic_t<1>{}, ic_t<0>{}, X[0]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (T == 3) {
            lds_wait4<0>(X[1][0], X[1][1], X[1][2], X[1][3]);
            lds_write32(cflag_addr, (unsigned)k + 1u);
            if (!last_slab) { DACC(2, 1); wait_producers((unsigned)k + 2u); }
            issue_xb(ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, ub[0] + hbn, X[0]);
          } else {
            issue_xb(ic_t<0>{}, ic_t<nty>{}, ic_t<0>{}, ub[ntx] + hb, X[0]);
            lds_wait4<4>(X[1][0], X[1][1], X[1][2], X[1][3]);
          }
          mfma16(ic_t<1>{}, ic_t<1>{}, X[1]);
          __builtin_amdgcn_sched_barrier(0);
          wq = wn;
        });
      } else
      static_for<0, 9>([&](auto tc) {
        constexpr int T = decltype(tc)::value, ky = T / 3, kx = T % 3;
        constexpr int nky = ((T + 1) % 9) / 3, nkx = (T + 1) % 3;
        // weights of the next step: the next tap of this slab (a tap = nslab slab blocks), or tap 0 of the next slab
        const char* wn = T == 8 ? (to_extra ? wx0 : wq - (8 * nslab - 1) * w_step_bytes) : wq + nslab * w_step_bytes;
        // entry: in flight are group 0 of k-half 0 (4 LDS reads -> X[0]) and the weights of k-half 0 (4 global loads -> Wf[0])
        // ---- k-half 0 ----
        issue_w(ic_t<1>{}, wq);
        issue_x(ic_t<1>{}, ic_t<ky>{}, ic_t<kx>{}, ic_t<0>{}, hb, X[1]);
        lds_wait4<4>(X[0][0], X[0][1], X[0][2], X[0][3]);
        vm_wait4<4>(Wf[0][0], Wf[0][1], Wf[0][2], Wf[0][3]);
        mfma16(ic_t<0>{}, ic_t<0>{}, X[0]);
        __builtin_amdgcn_sched_barrier(0);
        issue_x(ic_t<0>{}, ic_t<ky>{}, ic_t<kx>{}, ic_t<1>{}, hb, X[0]);
        lds_wait4<4>(X[1][0], X[1][1], X[1][2], X[1][3]);
        mfma16(ic_t<0>{}, ic_t<1>{}, X[1]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- k-half 1 ----
        issue_w(ic_t<0>{}, wn);
        issue_x(ic_t<1>{}, ic_t<ky>{}, ic_t<kx>{}, ic_t<1>{}, hb, X[1]);
        lds_wait4<4>(X[0][0], X[0][1], X[0][2], X[0][3]);
        vm_wait4<4>(Wf[1][0], Wf[1][1], Wf[1][2], Wf[1][3]);
        mfma16(ic_t<1>{}, ic_t<0>{}, X[0]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (T == 8) {
          // every pixel of slab k is in registers once the last group has landed: its image is free.  The next slab (if it is this unit's)
          // must be complete before its first rows are read; at a unit's last step the same reads go out unchecked and unused (one code
          // path, no join for the register allocator), and the next unit's first operands are issued again behind the epilogue.
          lds_wait4<0>(X[1][0], X[1][1], X[1][2], X[1][3]);
          lds_write32(cflag_addr, (unsigned)k + 1u);
          if (!last_slab) { DACC(2, 1); wait_producers((unsigned)k + 2u); }
          if (to_extra) issue_x(ic_t<0>{}, ic_t<1>{}, ic_t<1>{}, ic_t<0>{}, hbn, X[0]);
          else issue_x(ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, hbn, X[0]);
        } else {
          issue_x(ic_t<0>{}, ic_t<nky>{}, ic_t<nkx>{}, ic_t<0>{}, hb, X[0]);
          lds_wait4<4>(X[1][0], X[1][1], X[1][2], X[1][3]);
        }
        mfma16(ic_t<1>{}, ic_t<1>{}, X[1]);
        __builtin_amdgcn_sched_barrier(0);
        wq = wn;
      });
    }
    // ---- the folded 1x1 shortcut: one step (the centre tap) per 64-channel slab of its source ----
    for (int e = 0; e < nsx; ++e, ++k) {
      const bool last_slab = e == nsx - 1;
      const unsigned hb = (unsigned)(k % D_NBUF) * D_HB;
      const unsigned hbn = (unsigned)((k + 1) % D_NBUF) * D_HB;
      const char* wn = wq + w_step_bytes;   // the next slab's block (behind the last one: padding, loaded and unused)
      issue_w(ic_t<1>{}, wq);
      issue_x(ic_t<1>{}, ic_t<1>{}, ic_t<1>{}, ic_t<0>{}, hb, X[1]);
      lds_wait4<4>(X[0][0], X[0][1], X[0][2], X[0][3]);
      vm_wait4<4>(Wf[0][0], Wf[0][1], Wf[0][2], Wf[0][3]);
      mfma16(ic_t<0>{}, ic_t<0>{}, X[0]);
      __builtin_amdgcn_sched_barrier(0);
      issue_x(ic_t<0>{}, ic_t<1>{}, ic_t<1>{}, ic_t<1>{}, hb, X[0]);
      lds_wait4<4>(X[1][0], X[1][1], X[1][2], X[1][3]);
      mfma16(ic_t<0>{}, ic_t<1>{}, X[1]);
      __builtin_amdgcn_sched_barrier(0);
      issue_w(ic_t<0>{}, wn);
      issue_x(ic_t<1>{}, ic_t<1>{}, ic_t<1>{}, ic_t<1>{}, hb, X[1]);
      lds_wait4<4>(X[0][0], X[0][1], X[0][2], X[0][3]);
      vm_wait4<4>(Wf[1][0], Wf[1][1], Wf[1][2], Wf[1][3]);
      mfma16(ic_t<1>{}, ic_t<0>{}, X[0]);
      __builtin_amdgcn_sched_barrier(0);
      lds_wait4<0>(X[1][0], X[1][1], X[1][2], X[1][3]);
      lds_write32(cflag_addr, (unsigned)k + 1u);
      if (!last_slab) { DACC(2, 1); wait_producers((unsigned)k + 2u); }
      issue_x(ic_t<0>{}, ic_t<1>{}, ic_t<1>{}, ic_t<0>{}, hbn, X[0]);
      mfma16(ic_t<1>{}, ic_t<1>{}, X[1]);
      __builtin_amdgcn_sched_barrier(0);
      wq = wn;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the unused operands of the unit's last step
    DSTAMP(e0);
    UnitC nxt = cur;
    if (has_next) nxt = decode(u + 1);
    wq = wbase(nxt);
    stage_ready(u - u0, (unsigned)k);
    issue_w(ic_t<0>{}, wq);   // the next unit's first weights travel while the tile is staged (after the last unit: loaded again, unused)
#ifndef C3D_ABL_NOSTAGE
    stage_tile(u - u0, (unsigned)((k - 1) % D_NBUF) * D_HB);
#else
    lds_write32(eflag_addr, (unsigned)(u - u0) + 1u);
#endif
    vm_wait4<0>(Wf[0][0], Wf[0][1], Wf[0][2], Wf[0][3]);   // landed long ago; nothing asynchronous is live across the polls below
    DSTAMP(e1);
    DACC(3, e1 - e0); DACC(4, 1);
    if (has_next) {
      wait_producers((unsigned)k + 1u);                 // the next unit's first slab (and with it the bias table producer wave 0 issued in front of it)
      init_from_table((u + 1 - u0) & 1);
      DSTAMP(e2);
      DACC(5, e2 - e1);
      if constexpr (UPS) { set_parity(nxt); issue_xb(ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, ub[0] + (unsigned)(k % D_NBUF) * D_HB, X[0]); }
      else issue_x(ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, ic_t<0>{}, (unsigned)(k % D_NBUF) * D_HB, X[0]);
      cur = nxt;
    }
  }
#ifdef C3D_STAMPS
  { DSTAMP(c_t2); dbg[6] = c_t2 - c_t0; if (blockIdx.x == 0 && wave == 0 && lane == 0) for (int i = 0; i < 16; ++i) c3d_dbg[i] = dbg[i]; }
#endif
}

// fragment-packed copy of a conv3x3 weight matrix [N][9 Cin] (K-major, k = tap Cin + c) for the consumers of conv3x3d_kernel:
// wf[(((((tap nslab + slab) ntn + nt) 2 + wave_n) 4 + a) 2 + kh) * 1024 + lane * 16 ..] = w[nt 128 + wave_n 64 + a 16 + (lane & 15)][tap Cin + slab 64 + kh 32 + (lane >> 4) 8 .. + 8]
__global__ void pack_frag_weights_kernel(const f16* __restrict__ w, f16* __restrict__ wf, int N, int Cin) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nslab = Cin >> 6, ntn = N >> 7;
  const long long total = (long long)9 * nslab * ntn * 16 * 64;
  if (t >= total) return;
  const int lane = (int)(t & 63);
  long long f = t >> 6;
  const int kh = (int)(f & 1); f >>= 1;
  const int a = (int)(f & 3); f >>= 2;
  const int wn = (int)(f & 1); f >>= 1;
  const int nt = (int)(f % ntn); f /= ntn;
  const int slab = (int)(f % nslab);
  const int tap = (int)(f / nslab);
  const int n = nt * 128 + wn * 64 + a * 16 + (lane & 15);
  const long long k = (long long)tap * Cin + slab * 64 + kh * 32 + (lane >> 4) * 8;
  *reinterpret_cast<uint4*>(wf + t * 8) = *reinterpret_cast<const uint4*>(w + (long long)n * 9 * Cin + k);
}

// the folded shortcut's [N][Cs] (row pitch ld) behind the nine taps: block (slab e, channel tile nt) at 9 nslab ntn + e ntn + nt, same fragment order
__global__ void pack_frag_weights_sc_kernel(const f16* __restrict__ wsc, f16* __restrict__ wf, int N, int Cin, int Cs, int ld) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nslab = Cin >> 6, nsx = Cs >> 6, ntn = N >> 7;
  const long long total = (long long)nsx * ntn * 16 * 64;
  if (t >= total) return;
  const int lane = (int)(t & 63);
  long long f = t >> 6;
  const int kh = (int)(f & 1); f >>= 1;
  const int a = (int)(f & 3); f >>= 2;
  const int wn = (int)(f & 1); f >>= 1;
  const int nt = (int)(f % ntn);
  const int e = (int)(f / ntn);
  const int n = nt * 128 + wn * 64 + a * 16 + (lane & 15);
  const int k = e * 64 + kh * 32 + (lane >> 4) * 8;
  *reinterpret_cast<uint4*>(wf + ((long long)9 * nslab * ntn * 1024 + t) * 8) = *reinterpret_cast<const uint4*>(wsc + (long long)n * ld + k);
}

// D_UPS: the parity-folded weights w_par[q][Nrows][t][Cin] (launch_make_parity_weights) in the same fragment order, block (q, t, slab, channel tile) at
// ((q 4 + t) nslab + slab) ntn + nt
__global__ void pack_frag_weights_par_kernel(const f16* __restrict__ wpar, f16* __restrict__ wf, int N, int Nrows, int Cin) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int nslab = Cin >> 6, ntn = N >> 7;
  const long long total = (long long)16 * nslab * ntn * 16 * 64;
  if (t >= total) return;
  const int lane = (int)(t & 63);
  long long f = t >> 6;
  const int kh = (int)(f & 1); f >>= 1;
  const int a = (int)(f & 3); f >>= 2;
  const int wn = (int)(f & 1); f >>= 1;
  const int nt = (int)(f % ntn); f /= ntn;
  const int slab = (int)(f % nslab); f /= nslab;
  const int tap = (int)(f & 3), q = (int)(f >> 2);
  const int n = nt * 128 + wn * 64 + a * 16 + (lane & 15);
  const int k = slab * 64 + kh * 32 + (lane >> 4) * 8;
  *reinterpret_cast<uint4*>(wf + t * 8) = *reinterpret_cast<const uint4*>(wpar + (((long long)q * Nrows + n) * 4 + tap) * Cin + k);
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
  constexpr bool UPS = (FLAGS & D_UPS) != 0;
  const int units = UPS ? p.B * (p.Hin >> 4) * (p.Win >> 4) * 4 * (p.N >> 7) : p.B * (p.Hout >> 4) * (p.Wout >> 4) * (p.N >> 7);
  // Run length (units per workgroup): 1 where the executor says its graph shares the chip (the sampler's decode beside the next UNet pass),
  // else 0 = one workgroup per CU walking its whole share; LDIFF_C3D_RUN overrides.  The kernel alone is
  // fastest fully persistent (0.43 of the MFMA peak against 0.42), but the sampler decodes on a side stream beside the next UNet pass, and a
  // workgroup that holds a CU for half a millisecond keeps that pass's kernels out: measured on the whole step (8 patches, 5 passes)
  // 175.7 ms fully persistent, 175.1 / 174.7 / 171.6 ms at 8 / 4 / 2 units per workgroup, 178.2 ms with the 8x16 kernel.
  static const int run_env = [] { const char* e = getenv("LDIFF_C3D_RUN"); return e ? atoi(e) : -1; }();   // -1: 1 unit where the graph shares the chip (ConvParams::short_runs), else persistent
  const int run_cap = run_env >= 0 ? run_env : (p.short_runs ? 1 : 0);   // (whole step, same box: 171.1 ms at 2 units, 169.6 at 1, 172.0 at 3)
  int grid = units < d_num_cus() ? units : d_num_cus();
  if (run_cap > 0 && units > grid * run_cap) grid = (units + run_cap - 1) / run_cap;
  const double bytes = (double)p.B * p.Hin * p.Win * p.C1 * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * 2.0 + (p.res ? (double)p.M * p.N * 2.0 : 0.0);
  // (D_UPS: executed flops -- four taps per output pixel -- as the parity-folded 16 x 16 kernel counts them)
  ProfScope prof(UPS ? "conv3x3<16x16d,128,ups>" : "conv3x3<16x16d,128,gn>", UPS ? 2.0 * p.M * (double)p.N * 4.0 * p.C1 : 2.0 * p.M * (double)p.N * (p.K + p.Cs),
                 bytes + (double)p.M * p.Cs * 2.0 + (double)p.N * p.Cs * 2.0, s);
  // Non-temporal streaming where source AND output are beyond what L2 / the MALL keep between layers (>= 192 MB each: the 256^2 and 512^2 maps of the decoder at B = 8):
  // same box, one unit per workgroup, residual + statistics: 128 -> 128 @512^2 889 -> 849 us, 256 -> 128 @512^2 1249 -> 1210, 256 -> 256 @256^2 628 -> 602; on the 128^2 /
  // 64^2 maps (the four channel tiles of a pixel tile re-read its halo from L2) it costs 0 ... 4 % (profiles/r05_conv3x3d_cache_hints.txt).  LDIFF_C3D_NT=0 / 1: never / always.
  static const int nt_env = [] { const char* e = getenv("LDIFF_C3D_NT"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  ConvParams q = p;
  const double in_b = (double)p.B * p.Hin * p.Win * p.C1 * 2.0, out_b = (double)p.M * p.N * 2.0;
  q.nt_hint = nt_env >= 0 ? nt_env : ((p.short_runs || (in_b >= 192e6 && out_b >= 192e6)) ? 1 : 0);   // beside the UNet stream: always (whole step, same box: 155.5 / 155.0 ms without, 154.4 / 153.1 with the size rule, 152.8 / 153.6 always)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), D_LDS, s, q, units);
  HIP_CHECK(hipGetLastError());
}

}  // namespace

void launch_pack_frag_weights_sc(const f16* wsc, f16* wf, int N, int Cin, int Cs, int ld_wsc, hipStream_t s) {
  const long long total = (long long)(Cs >> 6) * (N >> 7) * 16 * 64;
  hipLaunchKernelGGL(pack_frag_weights_sc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wsc, wf, N, Cin, Cs, ld_wsc);
  HIP_CHECK(hipGetLastError());
}
void launch_pack_frag_weights_par(const f16* wpar, f16* wf, int N, int Nrows, int Cin, hipStream_t s) {
  const long long total = (long long)16 * (Cin >> 6) * (N >> 7) * 16 * 64;
  hipLaunchKernelGGL(pack_frag_weights_par_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wpar, wf, N, Nrows, Cin);
  HIP_CHECK(hipGetLastError());
}
void launch_pack_frag_weights(const f16* w, f16* wf, int N, int Cin, hipStream_t s) {
  const long long total = (long long)9 * (Cin >> 6) * (N >> 7) * 16 * 64;
  hipLaunchKernelGGL(pack_frag_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, wf, N, Cin);
  HIP_CHECK(hipGetLastError());
}

// LDIFF_CONV3X3_DATAFLOW: 0 = off, 1 (default) = where the unit list fills the chip, 2 = every eligible launch (tests, A/B timing)
bool conv3x3d_selected(const ConvParams& p) {
  static const int mode = [] { const char* e = getenv("LDIFF_CONV3X3_DATAFLOW"); return e ? atoi(e) : 1; }();
  if (mode == 0) return false;
  if (p.ks != 3 || p.stride != 1 || p.pad_t != 1 || p.pad_l != 1 || p.splitk > 1) return false;
  if (p.x2 || p.C2 != 0 || p.C1 % 64 != 0 || p.N % 128 != 0 || p.Nrows < p.N) return false;
  const bool ups = p.ups != 0;
  if (ups) {   // nearest-2x upsample folded into the conv (ConvParams::w_par): no prologue, no residual, no time embedding; LDIFF_C3D_UPS=0: the 16 x 16 ping-pong kernel
    // Where it is chosen: measured against the 16 x 16 ping-pong kernel on one box (profiles/r05_conv3x3d_upsample.txt) it wins 11 % on the 256^2 -> 512^2 conv with one
    // persistent workgroup per CU and nothing (0 ... +4 % time) on the smaller maps or with one-unit runs beside the UNet stream -- a parity unit streams a whole halo
    // image for four taps instead of nine, so the producers' DMA per MFMA is 2.25 times the plain conv's; the whole step, where the decode runs one-unit workgroups, gets
    // 0.3-0.9 % SLOWER.  So it is OFF unless asked for: ConvParams::c3d_ups = 1 (tests, timing) or LDIFF_C3D_UPS=1 for the whole process.  (A choice by short_runs would
    // also give the pipelined and the serial sampler different kernels for one layer, and bench.py checks the two bit for bit.)
    static const int env = [] { const char* e = getenv("LDIFF_C3D_UPS"); return e ? (atoi(e) != 0 ? 1 : -1) : 0; }();
    const int want = p.c3d_ups ? p.c3d_ups : env;
    if (want <= 0) return false;
    if (!p.w_par || p.gn_scale || p.res || p.temb || p.xs || p.Hin % 16 != 0 || p.Win % 16 != 0 || p.Hout != 2 * p.Hin || p.Wout != 2 * p.Win) return false;
  } else {
    if (p.w_par || !p.gn_scale || !p.silu_in) return false;
    if (p.Hout % 16 != 0 || p.Wout % 16 != 0 || p.Hin != p.Hout || p.Win != p.Wout) return false;
  }
  if (p.out_f32 || p.y_lo || p.res_lo || (p.ldy & 7) || (p.res && (p.ld_res & 7))) return false;
  if (p.xs && (p.res || p.Cs % 64 != 0 || p.Cs <= 0 || ((p.lds ? p.lds : p.Cs) & 7) || (long long)p.B * p.Hin * p.Win * (p.lds ? p.lds : p.Cs) * 2 >= (1LL << 31))) return false;
  const long long px = (long long)p.B * p.Hin * p.Win;
  if (px * (p.ld1 ? p.ld1 : p.C1) * 2 >= (1LL << 31) || (long long)p.M * p.ldy * 2 >= (1LL << 31) || (p.res && (long long)p.M * p.ld_res * 2 >= (1LL << 31))) return false;
  if ((long long)p.Nrows * 9 * p.C1 * 2 >= (1LL << 31)) return false;
  const long long units = (long long)p.B * (p.Hin >> 4) * (p.Win >> 4) * (p.N >> 7) * (ups ? 4 : 1);
  if (mode == 2) return true;
  const int cus = d_num_cus();
  const long long rounds = (units + cus - 1) / cus;
  return units >= cus && units * 100 >= rounds * cus * 88;   // the runs must split evenly over the CUs
}
// (+ the folded shortcut's blocks and, behind them, one unread step of padding: the consumers load one step ahead)
size_t conv3x3d_frag_bytes(const ConvParams& p) {
  if (p.ups) return (size_t)p.N * 16 * p.C1 * sizeof(f16) + (size_t)(p.N >> 7) * 16384 + 16384;   // 4 parities x 4 taps (+ one unread step behind the last)
  return (size_t)p.N * (9 * p.C1 + p.Cs) * sizeof(f16) + (p.Cs ? (size_t)(p.N >> 7) * 16384 + 16384 : 0);
}
int conv3x3d_stats_blocks(const ConvParams& p) { return p.ups ? (p.Hin >> 4) * (p.Win >> 4) * 8 : (p.Hout >> 4) * (p.Wout >> 4) * 2; }   // one row block per consumer wave pair (8 x 16 pixels)
#ifdef C3D_STAMPS
extern "C" int ldiff_debug_c3d_stamps(unsigned long long* out) {   // diagnostic build only
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c3d_dbg), sizeof(unsigned long long) * 48);
}
#endif
void launch_conv3x3d(const ConvParams& p, hipStream_t s) {
  LDIFF_CHECK(p.w_frag != nullptr, LDIFF_ERR_INVALID, "conv3x3 (dataflow): the caller did not provide the fragment-packed weights (launch_pack_frag_weights)");
  const int f = (p.res ? D_RES : 0) | (p.stats ? D_STATS : 0) | (p.xs ? D_SC : 0) | (p.ups ? D_UPS : 0);
  if (f == D_UPS) launch_c3d<D_UPS>(p, s);
  else if (f == (D_UPS | D_STATS)) launch_c3d<D_UPS | D_STATS>(p, s);
  else if (f == 0) launch_c3d<0>(p, s);
  else if (f == 1) launch_c3d<1>(p, s);
  else if (f == 2) launch_c3d<2>(p, s);
  else if (f == 3) launch_c3d<3>(p, s);
  else if (f == D_SC) launch_c3d<D_SC>(p, s);
  else if (f == (D_SC | D_STATS)) launch_c3d<D_SC | D_STATS>(p, s);
  else LDIFF_CHECK(false, LDIFF_ERR_INVALID, "conv3x3 (dataflow): a folded shortcut replaces the residual");
}
