// 3x3 stride-1 convolution for gfx950, persistent "ping-pong" variant for the large feature maps WITHOUT a GroupNorm prologue (the
// nearest-2x upsampling convs of the VAE decoder, the PREC_FULL convs of the VAE encoder on normalised operands): 35 of the 285
// conv3x3 launches of a bench step; the others stay on the 8 x 16 kernel (kernels_conv3x3.hip), see conv3x3p_selected.
//
// Same algorithm and operand images as conv3x3w_kernel (kernels_conv3x3.hip: one halo image per 64-channel slab, one MFMA step per tap
// on the shifted halo, weight slices by LDS-DMA), but ONE persistent 512-thread workgroup per CU that walks a list of 16 x 16-pixel
// tiles, and whose two wave groups (waves 0-3 = A, 4-7 = B; wave w and w+4 share a SIMD) run the step loop half a step apart:
//
//      phase        2k            2k+1           2k+2           2k+3
//      group A   compute(k)   | load(k)      | compute(k+1) | load(k+1)    |        `|` = s_barrier of the whole workgroup
//      group B   load(k-1)    | compute(k)   | load(k)      | compute(k+1) |
//
// so on every SIMD one wave is in its matrix segment (32 MFMA + the 8 operand reads of the second k-half) while its partner is in its load
// segment.  Measured before this kernel (profiles/r01_conv3x3_issue_profile.md, MI355X_MICROARCH.md "Two waves per SIMD"): with two
// independent 256-thread workgroups per CU the matrix segments of co-resident waves coincide (MFMA busy 41 % of the cycles), and a
// non-persistent 16 x 16 tile pays 8-13 us of un-overlapped prologue + epilogue per tile (as much as the 18 steps of a 128-channel layer).
//   * tile = 16 x 16 pixels of one image x BN output channels; group g owns pixel rows [8g, 8g+8), its waves 2 x 2 (64 pixels x BN/2
//     channels per wave); both groups read ONE 18 x 18 halo image and ONE weight slice per step: half the weight LDS-DMA bytes per MFMA
//     of the 8 x 16 kernel, halo redundancy 1.27 instead of 1.41;
//   * the step sequence runs THROUGH tile boundaries: the halo of the next tile's first slab is staged during the last slab of the
//     current tile like any other slab, the weight ring keeps turning, and a group writes its tile out in the load segment of the
//     tile's last step while the other group computes;
//   * halo staging by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB = 8 halo pixels per wave-instruction): no staging registers
//     live across matrix segments, zero padding from the buffer descriptor's range check, the column swizzle applied on the SOURCE
//     address (lane at position q of pixel (hy,hx) fetches channel chunk q ^ swzx(hx)).  There is NO GroupNorm prologue here: next to
//     a partner wave that issues MFMAs the transform of one 16-byte chunk per lane (8 x cvt, fma, exp, rcp, mul) takes ~1000 cycles of
//     a load segment against 650 for the matrix segment it should hide behind, +190..250 us on a 128-channel 512 x 512 layer, while one
//     norm_apply pass over the tensor (kernels_norm.hip, HBM-bound) would cost ~110 us: with this kernel ~10 % faster than the 8 x 16
//     kernel on the same layer without GroupNorm, "normalise first, then this kernel" came out 4.5 ms per bench step SLOWER than the
//     fused prologue of the 8 x 16 kernel, so convs with a GroupNorm prologue stay there (conv3x3p_selected);
//   * weight ring of 4 slots: load(k) issues this wave's rows of step k+3 into slot (k+3) % 4, last read in compute(k-1) of both
//     groups; a wave drains its DMAs (vmcnt(0)) only at the END of its next matrix segment, one full segment after issue, so the
//     slice of step k+1 is complete and published when load(k) ends;
//   * the barrier that ends a load segment waits for LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier): the DMAs stay in flight;
//   * the accumulators start at bias + time embedding, the residual is added in the load segments of the tile's last taps, and the
//     epilogue proper only converts, swaps (v_permlane16_swap -> 16-byte stores) and stores: a monolithic epilogue inside the
//     persistent loop costs ~70 registers on top of the accumulators and exposes the residual's memory round trip once per tile.
// Replaces the conv2d of diffusers' ResnetBlock2D / Upsample2D (/root/reference/segmentor.py:103,106,519, pixel_latent_vector.py:73-81).
#include <cstdlib>
#include <map>
#include <mutex>

#include "common.h"

namespace {

__device__ __forceinline__ int swz8(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
typedef __attribute__((address_space(3))) void lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int OFF>
__device__ __forceinline__ void lds_read128(f16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT>
__device__ __forceinline__ void lds_wait(f16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void lds_wait(f16x8& a, f16x8& b, f16x8& c, f16x8& d, f16x8& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(CNT));
}
// LDS accesses the compiler must not see as such: it orders every LDS access it knows of behind ALL pending vector-memory operations
// when an LDS-DMA may be among them (s_waitcnt vmcnt(0)), which after the epilogue's stores means a full write round trip
template <int OFF>
__device__ __forceinline__ void lds_read128f(f32x4& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wait0(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void lds_write64(unsigned addr, float2 v) {
  asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
// end of a load segment: the LDS accesses of this wave are done, its LDS-DMAs and global loads stay in flight
__device__ __forceinline__ void lgkm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// end of a matrix segment: everything this wave issued one segment ago (LDS-DMA weight rows and halo pieces) has landed.  Explicit:
// __syncthreads() alone does not drain LDS-DMA (a workgroup-scope fence needs no vmcnt wait on this target).
__device__ __forceinline__ void full_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// column swizzle of the halo image (kernels_conv3x3.hip): conflict-free 16x16x32 operand fetch at all three column shifts
__device__ __forceinline__ int swzx(int hx) { return (0xcb5888 >> (3 * (hx >> 1))) & 7; }
// lane-invariant values recomputed per tile: keep the optimiser from hoisting them out of the persistent loop (each one would
// occupy a register for the whole kernel)
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

#ifdef C3P_STAMPS   // diagnostic build only (scripts/conv_stamps.py): per-segment cycle sums of waves 0 and 4 of workgroup 0
__device__ unsigned long long c3p_dbg[2][16];
__device__ __forceinline__ unsigned long long stamp() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define STAMP(v) const unsigned long long v = stamp()
#define ACCUM(i, a, b) dbg[i] += (b) - (a)
#else
#define STAMP(v)
#define ACCUM(i, a, b)
#endif
struct TileC { int b, oy0, ox0, n0, q; };   // image, pixel origin (tile space), first output channel, output parity (parity mode)

// FLAGS: what the epilogue has to do.  Compiled in only where needed: every optional block inside the persistent loop costs the other
// configurations time as well (the same kernel without the residual / split-output / statistics code ran the epilogue segment in
// 1290 instead of 3840 cycles and every load segment ~35 % faster: instruction fetch of a 50 KB loop body with cold blocks in it)
enum { C3P_RES = 1, C3P_LO = 2, C3P_STATS = 4, C3P_LO8 = 8 };   // LO8: the input is a split operand with an fp8 lo half (ConvParams::lo8_slab0)
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v8i_t __attribute__((ext_vector_type(8)));
// the two 16-byte k-half fragments of a lane side by side = its 32 bytes of a 16x16x128 fp8 operand (any lane -> channel mapping is valid
// as long as A and B use the same one: chunk g and chunk 4 + g of the 128-byte row, exactly what the fp16 path reads)
__device__ __forceinline__ v8i_t cat8(const f16x8& a, const f16x8& b) {
  const v4i_t x = __builtin_bit_cast(v4i_t, a), y = __builtin_bit_cast(v4i_t, b);
  return (v8i_t){x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
}

template <int BN, bool PAR, int FLAGS>
__global__ __launch_bounds__(512, 2) void conv3x3p_kernel(const ConvParams p, const int total_tiles) {
  constexpr int TH = 16, TW = 16, HWD = 18, HP = 18 * 18, MT = 4, NT = BN / 32, A_IT = 6, NP = BN / 64, NSLOT = 4;
  constexpr int NTAPS = PAR ? 4 : 9;
  constexpr int ROWB = HWD * 128;                                    // bytes per halo row
  constexpr unsigned HBUF = 41 * 1024;                               // one halo image: 41 DMA pieces of 8 pixels (324 pixels + 4 pad)
  constexpr unsigned W_OFF = 2 * HBUF, W_BYTES = BN * 128;           // LDS map: halo[2] | weights[4] | bias table | stats exchange
  constexpr unsigned BTAB = W_OFF + NSLOT * W_BYTES;                             // [bias BN | temb BN] float of the workgroup's next tile
  constexpr unsigned XCH = BTAB + 1024;                              // [grp 2][wave_m 2][wave_n 2][g 4][16 values] float2
  constexpr int NF = NT + MT;
  constexpr bool LO8 = (FLAGS & C3P_LO8) != 0;
  static_assert(BN == 64 || BN == 128, "a wave owns BN/8 weight rows = whole 1 KiB DMA pieces");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wave_m = (wave >> 1) & 1, wave_n = wave & 1;
  const int wm4 = grp * 2 + wave_m;                                  // 64-pixel row block (4 tile rows) of this wave
  const int g = lane >> 4, l15 = lane & 15;
  const int Cin = p.C1 + p.C2;
  const int Ht = PAR ? p.Hin : p.Hout, Wt = PAR ? p.Win : p.Wout;
  const int tiles_x = (Wt + TW - 1) / TW, tiles_y = (Ht + TH - 1) / TH;
  const int ntn = (p.N + BN - 1) / BN;
  const int sh = PAR ? 0 : p.ups;
  const int He = p.Hin << sh, We = p.Win << sh;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t*)smem_raw;
  const long long Kw = (long long)NTAPS * Cin;
  const int nslab = Cin / 64;
  // fp8 lo slabs (LO8): slab index from which the operands are e4m3, and the two E8M0 scale operands of the block-scaled MFMA
  const int lo8_c0 = LO8 ? p.lo8_slab0 : 0x7fffffff, lo8_sb = p.lo8_sb;
  const int lo8_sa = LO8 ? __builtin_amdgcn_readfirstlane(*p.lo8_sa) : 0;

  // Persistent workgroups: workgroup w walks tiles w, w + gridDim.x, ...  XCD-aware order: the workgroups of one XCD (id % 8) walk one
  // contiguous range of the tile list (n-tile fastest, then x, y, image, parity), so halos and weight slices are shared through its L2.
  auto decode = [&](int v) __attribute__((always_inline)) -> TileC {
    const int q8 = total_tiles >> 3, r8 = total_tiles & 7, xcd = v & 7, idx = v >> 3;
    int tm = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    TileC t;
    t.n0 = __builtin_amdgcn_readfirstlane((tm % ntn) * BN); tm /= ntn;
    t.ox0 = __builtin_amdgcn_readfirstlane((tm % tiles_x) * TW); tm /= tiles_x;
    t.oy0 = __builtin_amdgcn_readfirstlane((tm % tiles_y) * TH); tm /= tiles_y;
    t.b = __builtin_amdgcn_readfirstlane(tm % p.B);
    t.q = __builtin_amdgcn_readfirstlane(tm / p.B);
    return t;
  };
  // The workgroup's next tile is gridDim.x / 8 places further in the list (gridDim.x is a multiple of 8 whenever a workgroup has a
  // next tile: the launcher rounds the persistent grid): one mixed-radix addition of a constant instead of five divisions per tile.
  TileC step_d;
  {
    int d = (int)gridDim.x >> 3;
    step_d.n0 = (d % ntn) * BN; d /= ntn;
    step_d.ox0 = (d % tiles_x) * TW; d /= tiles_x;
    step_d.oy0 = (d % tiles_y) * TH; d /= tiles_y;
    step_d.b = d % p.B;
    step_d.q = d / p.B;
  }
  auto next_tile = [&](const TileC& t) __attribute__((always_inline)) -> TileC {
    TileC r;
    int cy;
    r.n0 = t.n0 + step_d.n0; cy = r.n0 >= ntn * BN; r.n0 -= cy ? ntn * BN : 0;
    r.ox0 = t.ox0 + step_d.ox0 + (cy ? TW : 0); cy = r.ox0 >= tiles_x * TW; r.ox0 -= cy ? tiles_x * TW : 0;
    r.oy0 = t.oy0 + step_d.oy0 + (cy ? TH : 0); cy = r.oy0 >= tiles_y * TH; r.oy0 -= cy ? tiles_y * TH : 0;
    r.b = t.b + step_d.b + cy; cy = r.b >= p.B; r.b -= cy ? p.B : 0;
    r.q = t.q + step_d.q + cy;
    return r;
  };

  // ---- halo staging by LDS-DMA: piece j = halo pixels [8j, 8j+8) = 1 KiB of the image; wave w issues pieces w, w+8, ... (piece 40, the
  // last 4 pixels + 4 pad pixels, is wave 0's sixth).  Lane l of piece j: pixel 8j + l/8, position l%8, channel chunk (l%8) ^ swzx(hx).
  // s_src[i]: element offset pix * pitch is formed at issue time from the source pixel index, or the out-of-range sentinel for padding.
  constexpr unsigned OOR = 0x80000000u;        // >= num_records of every eligible tensor: the DMA writes zeros
  unsigned s_pix[A_IT];                        // source pixel index, OOR for padding / pad pixels
  unsigned s_kc = 0;                           // 3 bits per piece: channel chunk of this lane
  unsigned s_yx[A_IT / 2];                     // (hy << 8 | hx) of this lane's halo pixel, two pieces per register; hy = 255: no such pixel
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int hp = wave * 8 + i * 64 + (lane >> 3);
    const int hy = hp / HWD, hx = hp - hy * HWD;
    const unsigned e = hp < HP ? (unsigned)(hy << 8 | hx) : 0xff00u;
    if (i & 1) s_yx[i >> 1] |= e << 16; else s_yx[i >> 1] = e;
    s_kc |= (unsigned)((lane & 7) ^ swzx(hx)) << (3 * i);
  }
  auto set_stage_coords = [&](const TileC& t) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const unsigned e = opaque((int)s_yx[i >> 1]) >> ((i & 1) * 16);
      const int hy = (e >> 8) & 0xff, hx = e & 0xff;
      const int iy = t.oy0 + hy - 1, ix = t.ox0 + hx - 1;
      const bool inb = hy != 0xff && (unsigned)iy < (unsigned)He && (unsigned)ix < (unsigned)We;
      s_pix[i] = inb ? (unsigned)(((t.b * p.Hin + (iy >> sh)) * p.Win) + (ix >> sh)) : OOR;
    }
  };
  const bool sixth = wave == 0;                // pieces 40..47 do not exist for the other waves
  const __amdgpu_buffer_rsrc_t xrsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((long long)p.B * p.Hin * p.Win * (p.ld1 ? p.ld1 : p.C1) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t xrsrc2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x), 0,
                                                                           (int)((long long)p.B * p.Hin * p.Win * (p.x2 ? (p.ld2 ? p.ld2 : p.C2) : 0) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((long long)p.M * p.ldy * 2), 0x00020000);
  auto issue_halo = [&](auto ic, int c, unsigned buf) __attribute__((always_inline)) {   // piece i of slab c of the tile in s_pix -> halo image buf
    constexpr int i = decltype(ic)::value;
    if (i == A_IT - 1 && !sixth) return;
    const int cb = c * 64;
    const bool first = cb < p.C1;
    const int Cs = first ? (p.ld1 ? p.ld1 : p.C1) : (p.ld2 ? p.ld2 : p.C2);
    const int soff = (first ? cb : cb - p.C1) * 2;
    unsigned char* dst = smem_raw + buf * HBUF + wave * 1024 + i * 8192;
    const unsigned kc8 = ((s_kc >> (3 * i)) & 7u) * 8u;
    const int voff = s_pix[i] == OOR ? (int)OOR : (int)((s_pix[i] * (unsigned)Cs + kc8) * 2u);
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass rejects this builtin (target feature) and then silently drops the kernel stub
    if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc1, (lptr_t*)dst, 16, voff, soff, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc2, (lptr_t*)dst, 16, voff, soff, 0, 0);
#endif
  };
  // ---- weight slices: wave w owns rows [w*BN/8, (w+1)*BN/8) of the [BN][64] slice = NP pieces of 8 rows (1 KiB) ----
  struct WSrc { __amdgpu_buffer_rsrc_t rsrc; int voff[NP]; };
  auto weights_of = [&](const TileC& t) __attribute__((always_inline)) -> WSrc {
    WSrc w;
    const f16* wsrc = PAR ? p.w_par + (long long)t.q * p.Nrows * Kw : p.w;
    w.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, (int)((long long)p.Nrows * Kw * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int r = wave * (BN / 8) + i * 8 + (opaque(lane) >> 3), pos = lane & 7;
      int n = t.n0 + r;
      n = n < p.Nrows ? n : p.Nrows - 1;   // rows beyond the matrix are never stored; keep the address in range
      w.voff[i] = (int)((unsigned)n * (unsigned)(Kw * 2) + (unsigned)(swz8(r, pos) * 16)) - i * 1024;   // the instruction offset is added to BOTH addresses
    }
    return w;
  };
  auto issue_w = [&](const WSrc& w, int soff, unsigned slot) __attribute__((always_inline)) {   // soff: byte offset of (tap, slab) inside a weight row
    unsigned char* dst = smem_raw + W_OFF + slot * W_BYTES + wave * (BN * 16);
    static_for<0, NP>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w.rsrc, (lptr_t*)dst, 16, w.voff[i], soff, i * 1024, 0);
#endif
    });
  };

  f32x4 acc[NT][MT];   // start at bias + time embedding (see the header)

  // per-lane operand addresses (k-half 0): X base of column shift j (kx = j, or px + j in parity mode), W row of slot 0
  unsigned xb[3];
  auto set_xb = [&](const TileC& t) __attribute__((always_inline)) {
    const int py = t.q >> 1, px = t.q & 1;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int hx = l15 + (PAR ? px + (j & 1) : j);
      xb[j] = lds0 + (unsigned)(((wm4 * 4 + (PAR ? py : 0)) * HWD + hx) * 128 + ((g ^ swzx(hx)) << 4));
    }
  };
  const int wrow = wave_n * (BN / 2) + l15;   // + a*16: same swizzle phase, +2048 B per a
  const unsigned w_lane = lds0 + W_OFF + (unsigned)(wrow * 128 + ((g ^ ((wrow >> 1) & 7)) << 4));

  // ---- epilogue pieces ----
  const int R_st = p.stats_R;
  long long st_prev = -1;   // statistics row block of the group's previous tile, still to be combined (see below): float index of channel 0
  int st_ncol = 0;          // ... and this lane's first channel in it
  auto out_row = [&](const TileC& t, int m) __attribute__((always_inline)) -> long long {   // output row of m-tile m's pixel of this lane, -1 outside
    const int ml = wm4 * 64 + m * 16 + opaque(l15);
    const int ty_ = t.oy0 + ml / TW, tx_ = t.ox0 + ml % TW;
    const int oy = PAR ? 2 * ty_ + (t.q >> 1) : ty_, ox = PAR ? 2 * tx_ + (t.q & 1) : tx_;
    return (ty_ < Ht && tx_ < Wt) ? ((long long)t.b * p.Hout + oy) * p.Wout + ox : -1;
  };
  // bias + time embedding of a tile: waves 0, 1 fetch bias[n0 .. n0 + BN), waves 2, 3 temb[b][n0 .. n0 + BN) (one float per lane, BN = 64:
  // waves 0 and 2 only), park them in LDS one load segment later, and both groups start the tile's sums from the table
  float btv = 0.f;
  auto load_bt = [&](const TileC& t) __attribute__((always_inline)) {
    btv = 0.f;
    if (wave < 4) {
      const int n = t.n0 + (wave & 1) * 64 + lane;
      if ((BN == 128 || (wave & 1) == 0) && n < p.N) {
        if (wave < 2) { if (p.bias) btv = p.bias[n]; }
        else if (p.temb) btv = p.temb[(long long)t.b * p.ld_temb + n];
      }
    }
  };
  auto store_bt = [&]() __attribute__((always_inline)) {
    if (wave < 4 && (BN == 128 || (wave & 1) == 0))
      *reinterpret_cast<float*>(smem_raw + BTAB + (wave >> 1) * (BN * 4) + (wave & 1) * 256 + lane * 4) = btv;
  };
  auto init_acc = [&]() __attribute__((always_inline)) {
    const unsigned tb = lds0 + BTAB + (unsigned)(wave_n * (BN / 2) + opaque(g) * 4) * 4u;
    static_for<0, NT / 2>([&](auto hc) {
      constexpr int a0 = decltype(hc)::value * 2;
      f32x4 b0, b1, t0, t1;
      lds_read128f<a0 * 64>(b0, tb); lds_read128f<a0 * 64 + 64>(b1, tb);
      lds_read128f<BN * 4 + a0 * 64>(t0, tb); lds_read128f<BN * 4 + a0 * 64 + 64>(t1, tb);
      lds_wait0(b0, b1, t0, t1);
#pragma unroll
      for (int m = 0; m < MT; ++m) { acc[a0][m] = b0 + t0; acc[a0 + 1][m] = b1 + t1; }
    });
  };
  f16x4 rr[MT / 2][NT];     // residual operand in flight: two of the wave's four pixel rows, hi or lo half
  auto load_res = [&](const TileC& t, int mh, int lo_off) __attribute__((always_inline)) {
    const int ncol = t.n0 + wave_n * (BN / 2) + opaque(g) * 4;
#pragma unroll
    for (int m = 0; m < MT / 2; ++m) {
      const long long row = out_row(t, mh * 2 + m);
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        rr[m][a] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        if (row >= 0 && ncol + a * 16 < p.N) rr[m][a] = *reinterpret_cast<const f16x4*>(p.res + row * p.ld_res + lo_off + ncol + a * 16);
      }
    }
  };
  auto add_res = [&](auto mhc) __attribute__((always_inline)) {
    constexpr int mh = decltype(mhc)::value;
#pragma unroll
    for (int m = 0; m < MT / 2; ++m)
#pragma unroll
      for (int a = 0; a < NT; ++a) acc[a][mh * 2 + m] += up4(rr[m][a]);
  };
  // lane holds y[pixel = column][n = 4g + r], complete in acc.  16-byte stores after one v_permlane16_swap per dword (conv3x3w_kernel);
  // the launcher guarantees their alignment (N, ldy, y_lo multiples of 8, fp16 output).
  auto epilogue = [&](const TileC& t) __attribute__((always_inline)) {
    const int py = t.q >> 1, px = t.q & 1;
    const int gq = opaque(g), lq = opaque(l15);
    constexpr bool has_lo = (FLAGS & C3P_LO) != 0, has_st = (FLAGS & C3P_STATS) != 0;
    // One v_permlane16_swap per dword between the packed values of two ADJACENT CHANNEL tiles (a, a+1) of one pixel tile leaves lane
    // row g holding channels [8 (g/2), 8 (g/2) + 8) of tile a + (g & 1): the four lanes of a pixel write 64 contiguous bytes.
    // Buffer stores: one 32-bit byte offset per lane for the wave's first pixel row, a scalar stride from row to row, rows / columns /
    // channels outside the tensor by the out-of-range sentinel (no exec masking, no 64-bit address arithmetic: the epilogue's VALU work
    // shares the SIMD with the partner wave's MFMAs).  Measured and dropped: a transposition through LDS so that neighbouring lanes
    // write neighbouring addresses (the segment got 45 % longer: it is bound by instruction issue next to the partner's MFMAs, not by
    // the 64 scattered 16-byte writes of a store).
    const int nb0 = t.n0 + wave_n * (BN / 2) + (gq & 1) * 16 + (gq >> 1) * 8;   // this lane's 8 channels of the pair (a, a+1) start at nb0 + 16 a
    const int nrem = p.N - nb0;                                                   // ... and exist while 16 a < nrem
    const int ty0 = t.oy0 + wm4 * 4, txs = t.ox0 + lq;              // tile-space row of m-tile 0 (wave-uniform) and column of this lane
    const unsigned pix0 = (unsigned)((t.b * p.Hout + (PAR ? 2 * ty0 + py : ty0)) * p.Wout + (PAR ? 2 * txs + px : txs));
    const unsigned off0 = (pix0 * (unsigned)p.ldy + (unsigned)nb0) * 2u;
    const unsigned rstride = (unsigned)((PAR ? 2 : 1) * p.Wout * p.ldy) * 2u;
    const bool okx = txs < Wt;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const bool ok = okx && ty0 + m < Ht;
      const unsigned offm = off0 + (unsigned)m * rstride;
#pragma unroll
      for (int a = 0; a < NT; a += 2) {
        f16x4 o[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) o[h] = cvt4(acc[a + h][m]);
        const uint2 q0 = __builtin_bit_cast(uint2, o[0]), q1 = __builtin_bit_cast(uint2, o[1]);
        auto r0 = __builtin_amdgcn_permlane16_swap(q0.x, q1.x, false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(q0.y, q1.y, false, false);
        const int voff = ok && a * 16 < nrem ? (int)offm : (int)OOR;
        // s_nop after every store: the very next VALU instruction may overwrite the store's data registers, and on this part that needs
        // wait states the compiler does not insert (measured: with the scalar lo offset in soffset, the first data dword of a lo store
        // came out as the address computation that followed it)
        __builtin_amdgcn_raw_buffer_store_b128((u32x4){r0[0], r1[0], r0[1], r1[1]}, yrsrc, voff + a * 32, 0, 0);
        asm volatile("s_nop 1" ::: "memory");
        if constexpr (has_lo) {   // lo halves of a split output, same swap
          const uint2 e0 = __builtin_bit_cast(uint2, cvt4(acc[a][m] - up4(o[0]))), e1 = __builtin_bit_cast(uint2, cvt4(acc[a + 1][m] - up4(o[1])));
          auto l0 = __builtin_amdgcn_permlane16_swap(e0.x, e1.x, false, false);
          auto l1 = __builtin_amdgcn_permlane16_swap(e0.y, e1.y, false, false);
          __builtin_amdgcn_raw_buffer_store_b128((u32x4){l0[0], l1[0], l0[1], l1[1]}, yrsrc, voff + (a * 32 + p.y_lo * 2), 0, 0);
          asm volatile("s_nop 1" ::: "memory");
        } else if constexpr (has_st) {   // statistics: of what the consumer will read (hi + lo ~ the fp32 sums for a split tensor)
          acc[a][m] = up4(o[0]); acc[a + 1][m] = up4(o[1]);
        }
      }
    }
    if constexpr (has_st) {
      // fused GroupNorm statistics (common.h): one row block per wave GROUP (128 pixels).  Per wave: sums over its 64 pixels (rows
      // outside the map are wave-uniform, columns outside it are masked once), reduced over the 16 pixel lanes with the totals spread
      // over the lanes (row16_reduce_spread), parked in LDS by one 8-byte write per lane; the two waves of a group that share a channel
      // half are combined in the group's NEXT load segment (two workgroup barriers later: a barrier of its own would break the pairing
      // with the other group's phases), see combine_stats.
      constexpr int NV = NT * 4;
      const int ty0s = t.oy0 + wm4 * 4;
      const bool okxs = t.ox0 + lq < Wt;
      float sv[NV], qv[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) { sv[j] = 0.f; qv[j] = 0.f; }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (ty0s + m >= Ht) continue;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = acc[a][m][r]; sv[a * 4 + r] += v; qv[a * 4 + r] += v * v; }
      }
      if (!okxs) {
#pragma unroll
        for (int j = 0; j < NV; ++j) { sv[j] = 0.f; qv[j] = 0.f; }
      }
      const float st_s = row16_reduce_spread<NV>(sv, lq), st_q = row16_reduce_spread<NV>(qv, lq);
      const int jv = ((lq >> 3) & 1) | ((lq >> 1) & 2) | ((lq << 1) & 4) | ((lq << 3) & 8);   // value index (a * 4 + r) this lane ends up with
      lds_write64(lds0 + XCH + (unsigned)((((grp * 2 + wave_m) * 2 + wave_n) * 4 + gq) * 16 + jv) * 8u, make_float2(st_s, st_q));
      const long long rblk = (((long long)t.q * tiles_y + t.oy0 / TH) * tiles_x + t.ox0 / TW) * 2 + grp;
      st_prev = ((long long)t.b * p.N * R_st + rblk) * 2;
      st_ncol = t.n0 + wave_n * (BN / 2) + gq * 4 + (jv >> 2) * 16 + (jv & 3);   // the channel this lane's totals belong to (NV = 8: j < 8)
    }
  };
  auto combine_stats = [&]() __attribute__((always_inline)) {   // the group's statistics of its previous tile: wave_m 0 adds its partner's totals, one store per lane
    if (wave_m == 0) {
      constexpr int NV = NT * 4;
      const int jv = ((l15 >> 3) & 1) | ((l15 >> 1) & 2) | ((l15 << 1) & 4) | ((l15 << 3) & 8);
      const float2* x0 = reinterpret_cast<const float2*>(smem_raw + XCH) + (((grp * 2 + 0) * 2 + wave_n) * 4 + g) * 16 + jv;
      const float2* x1 = reinterpret_cast<const float2*>(smem_raw + XCH) + (((grp * 2 + 1) * 2 + wave_n) * 4 + g) * 16 + jv;
      const float2 u = *x0, v = *x1;
      if ((NV == 16 || jv < NV) && st_ncol < p.N) *reinterpret_cast<float2*>(p.stats + st_prev + (long long)st_ncol * R_st * 2) = make_float2(u.x + v.x, u.y + v.y);
    }
    st_prev = -1;
  };

  int t_id = blockIdx.x;
  TileC cur = decode(t_id), nxt = cur;
  WSrc wcur = weights_of(cur), wnxt = wcur;
  set_stage_coords(cur);
  set_xb(cur);
  // prologue: halo of the first slab (normalised in place), weight slices of steps 0, 1 and 2, bias + time embedding
  static_for<0, A_IT>([&](auto ic) { issue_halo(ic, 0, 0); });
  issue_w(wcur, 0, 0);
  issue_w(wcur, Cin * 2, 1);
  issue_w(wcur, 2 * Cin * 2, 2);
  load_bt(cur);
  store_bt();
  full_barrier();
  init_acc();
  f16x8 xf0[MT], wf0[NT];   // k-half 0 operands of the next matrix segment
  f16x8 xf1p[MT], wf1p[NT]; // LO8 only: k-half 1 as well (an fp8 step needs both halves of an operand in front of its first MFMA)
  {
    constexpr int kx0 = 0;   // tap 0: ky = 0, column shift 0 (parity mode: px)
    static_for<0, MT>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<m * ROWB>(xf0[m], xb[kx0]); });
    static_for<0, NT>([&](auto ac) { constexpr int a = decltype(ac)::value; lds_read128<a * 2048>(wf0[a], w_lane); });
    if constexpr (LO8) {
      static_for<0, MT>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<m * ROWB>(xf1p[m], xb[kx0] ^ 64u); });
      static_for<0, NT>([&](auto ac) { constexpr int a = decltype(ac)::value; lds_read128<a * 2048>(wf1p[a], w_lane ^ 64u); });
    }
  }
  lgkm_barrier();
  if (grp) lgkm_barrier();   // group B runs one phase behind

  unsigned sp = 0, hb = 0;   // ring slot of the slab's tap 0; halo image of the slab
#ifdef C3P_STAMPS
  unsigned long long dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  // The step loop.  ONE code block for every slab (tile-first / tile-last / last of the workgroup are uniform run-time flags): three
  // specialised copies of the slab made the register allocator keep two sets of accumulators (the sums flow through all of them)
  // and tripled the code the instruction cache has to hold.
  int c = 0;
  bool more = t_id + (int)gridDim.x < total_tiles;
  // One slab (9 or 4 steps).  F8C (LO8 kernels only): the slab's operands are e4m3 -- a compile-time property of the call site, because a
  // run-time branch between the two MFMA forms inside ONE loop body makes the accumulators of the two paths distinct registers (the CFG is
  // structurised: 64 more registers, 130 spills); the LO8 kernels therefore run two loops per tile, fp16 slabs then fp8 slabs, and the
  // fp16 copy knows at compile time that it never holds the tile's last slab (no epilogue code in it).
  auto slab = [&](auto f8c) __attribute__((always_inline)) -> bool {
    constexpr bool F8 = decltype(f8c)::value;
    const bool tlast = (LO8 && !F8) ? false : c == nslab - 1;   // last slab of the tile
    const bool stage = !tlast || more;          // another slab follows (of this tile, or slab 0 of the workgroup's next tile)
    const unsigned hbuf = hb * HBUF;
    unsigned xc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) xc[j] = xb[j] + hbuf;
    auto wslot = [&](int j) __attribute__((always_inline)) -> unsigned { return w_lane + ((sp + j) & (NSLOT - 1)) * W_BYTES; };   // LDS address of this lane's W row of step j of the slab
    static_for<0, NTAPS>([&](auto tc) {
      constexpr int T = decltype(tc)::value;
      constexpr int kyi = PAR ? (T >> 1) : T / 3, kxi = PAR ? (T & 1) : T % 3;
      constexpr bool LAST = T == NTAPS - 1;
      const bool final = LAST && !stage;        // the workgroup's very last step
      // ---------------- matrix segment ----------------
      STAMP(t0);
      if constexpr (LO8) {
        // both k-halves were prefetched in the previous load segment.  fp16 slab: 2 x 16 MFMAs of K = 32; fp8 slab (128 channels in the same
        // 128 bytes per pixel / weight row): 16 block-scaled MFMAs of K = 128 over the two halves side by side -- the same matrix-pipe time
        if constexpr (F8) {
          static_for<0, NT>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            const v8i_t wv = cat8(wf0[a], wf1p[a]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
              acc[a][m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, cat8(xf0[m], xf1p[m]), acc[a][m], 0, 0, 0, lo8_sa, 0, lo8_sb);
            __builtin_amdgcn_sched_barrier(0);
          });
        } else {
          static_for<0, 2 * NT>([&](auto ic) {
            constexpr int kk = decltype(ic)::value / NT, a = decltype(ic)::value % NT;
#pragma unroll
            for (int m = 0; m < MT; ++m)
              acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kk ? wf1p[a] : wf0[a], kk ? xf1p[m] : xf0[m], acc[a][m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          });
        }
      } else {
        // k-half 0 is in registers already (prefetched at the end of the previous load segment); k-half 1 flies under its MFMAs
        const unsigned wc1 = wslot(T) ^ 64u, xk1 = xc[kxi] ^ 64u;   // chunk bit 2 = k-half: XOR commutes with the swizzle
        f16x8 wf1[NT], xf1[MT];
        // the k-half-1 reads go out BETWEEN the first MFMA groups (an LDS instruction issues while the matrix pipe works; eight of
        // them in front of the first MFMA cost ~100 cycles of every matrix segment)
        static_for<0, NT>([&](auto ac) {
          constexpr int a = decltype(ac)::value;
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[a], xf0[m], acc[a][m], 0, 0, 0);
          if constexpr (a == 0) static_for<0, MT>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<(m + kyi) * ROWB>(xf1[m], xk1); });
          if constexpr (a == 1) static_for<0, NT>([&](auto bc) { constexpr int b2 = decltype(bc)::value; lds_read128<b2 * 2048>(wf1[b2], wc1); });
          __builtin_amdgcn_sched_barrier(0);
        });
        static_for<0, NT>([&](auto ac) {
          constexpr int a = decltype(ac)::value;
          constexpr int pending = NT - 1 - a;   // reads issued after W_a
          if constexpr (a == 0) lds_wait<pending>(xf1[0], xf1[1], xf1[2], xf1[3], wf1[0]);
          else lds_wait<pending>(wf1[a]);
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[a], xf1[m], acc[a][m], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      // end of the matrix segment: this wave's DMAs issued one segment ago have landed (vmcnt(0)) and are published.
      // Group B's very last matrix segment needs neither (nothing follows it but its epilogue).
      STAMP(t1);
#ifdef C3P_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(t1b);
      dbg[14] += t1b - t1;
#endif
      if (!final || grp == 0) full_barrier();
      else asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the epilogue's first VALU reads the last MFMAs' results: the hazard
                                                                 // recogniser does not look across this branch (measured: stale sums)
      STAMP(t2);
      __builtin_amdgcn_sched_barrier(0);   // the load segment's work must not be hoisted into the matrix segment
      // ---------------- load segment ----------------
      // LDS accesses first, DMAs last: the compiler orders an LDS access behind every pending LDS-DMA with vmcnt(0).
      if constexpr (T == 0 && (FLAGS & C3P_STATS) != 0) { if (st_prev >= 0) combine_stats(); }
      if constexpr (!PAR && T >= 3 && (FLAGS & C3P_RES) != 0) {
        // residual (never in front of an upsampling conv): added to the sums in the load segments of the tile's last slab, 16 registers
        // at a time: rows 0-1 / 2-3 of the hi half fly over the matrix segments of taps 4 / 5, those of a split residual's lo half over
        // taps 6 / 7; the last of it is in before tap 8, so at most a few MFMA steps round on top of it
        if (tlast && (!LO8 || p.res)) {   // (the LO8 build serves convs with and without a residual: its residual-free variant does not fit the registers)
          constexpr std::integral_constant<int, 0> h0{}; constexpr std::integral_constant<int, 1> h1{};
          if constexpr (T == 3) load_res(cur, 0, 0);
          if constexpr (T == 4) { add_res(h0); __builtin_amdgcn_sched_barrier(0); load_res(cur, 1, 0); }
          if constexpr (T == 5) { add_res(h1); __builtin_amdgcn_sched_barrier(0); if (p.res_lo) load_res(cur, 0, p.res_lo); }
          if constexpr (T == 6) { if (p.res_lo) { add_res(h0); __builtin_amdgcn_sched_barrier(0); load_res(cur, 1, p.res_lo); } }
          if constexpr (T == 7) { if (p.res_lo) add_res(h1); }
        }
      }
      if constexpr (LAST) {
        if (tlast) {   // the tile is complete: write it out, then this group moves on to the next tile
          STAMP(e0);
          epilogue(cur);
          __builtin_amdgcn_sched_barrier(0);   // the new sums must not become live before the old ones are stored (2 x 64 registers)
          STAMP(e1);
          if (stage) { cur = nxt; if (PAR) set_xb(cur); init_acc(); }
          STAMP(e2);
          ACCUM(8, e0, e1); ACCUM(9, e1, e2); ACCUM(11, t2, e0);
#ifdef C3P_STAMPS
          dbg[12] = e2;
#endif
        }
      }
      // the next slab is slab 0 of the workgroup's next tile: its coordinates now (the first halo piece goes out in this segment), its
      // weight rows and bias table in the following segments (a tile switch in one piece made this segment 4x the matrix segment)
      if constexpr (T == 0) { if (stage && tlast) { nxt = next_tile(cur); set_stage_coords(nxt); } }
      if constexpr (T == (PAR ? 0 : 1)) { if (stage && tlast) wnxt = weights_of(nxt); }
      if constexpr (T == (PAR ? 1 : 2)) { if (stage && tlast) load_bt(nxt); }
      if constexpr (T == (PAR ? 2 : 3)) { if (stage && tlast) store_bt(); }
      // the next slab's halo: HPT pieces per load segment (a piece costs ~400 cycles of issue with 8 waves at it)
      constexpr int HPT = PAR ? 2 : 1;
      if (stage) static_for<T * HPT, (T * HPT + HPT < A_IT ? T * HPT + HPT : A_IT)>([&](auto ic) { issue_halo(ic, tlast ? 0 : c + 1, hb ^ 1u); });
      // this wave's rows of the weight slice three steps ahead
      if constexpr (T + 3 < NTAPS) issue_w(wcur, ((T + 3) * Cin + c * 64) * 2, (sp + T + 3) & (NSLOT - 1));
      else {
        if (stage) {
          if (tlast) issue_w(wnxt, (T + 3 - NTAPS) * Cin * 2, (sp + T + 3) & (NSLOT - 1));
          else issue_w(wcur, ((T + 3 - NTAPS) * Cin + (c + 1) * 64) * 2, (sp + T + 3) & (NSLOT - 1));
        }
      }
      if constexpr (LAST) { if (tlast && stage) wcur = wnxt; }
      // k-half 0 operands of this group's NEXT step (its weight slice was complete and published one phase ago; the halo image of a
      // next slab since its tap 7), so that the matrix segment opens with MFMAs instead of an LDS round trip
      __builtin_amdgcn_sched_barrier(0);
      if (!final) {
        constexpr int Tn = LAST ? 0 : T + 1;
        constexpr int kyn = PAR ? (Tn >> 1) : Tn / 3, kxn = PAR ? (Tn & 1) : Tn % 3;
        const unsigned xn = LAST ? xb[kxn] + (hb ^ 1u) * HBUF : xc[kxn];
        const unsigned wn = wslot(T + 1);
        static_for<0, MT>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<(m + kyn) * ROWB>(xf0[m], xn); });
        static_for<0, NT>([&](auto ac) { constexpr int a = decltype(ac)::value; lds_read128<a * 2048>(wf0[a], wn); });
        if constexpr (LO8) {
          static_for<0, MT>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<(m + kyn) * ROWB>(xf1p[m], xn ^ 64u); });
          static_for<0, NT>([&](auto ac) { constexpr int a = decltype(ac)::value; lds_read128<a * 2048>(wf1p[a], wn ^ 64u); });
        }
      }
      // the very last step's load segment is followed by nothing: no barrier
      __builtin_amdgcn_sched_barrier(0);
      STAMP(t3);
      if (!final) lgkm_barrier();
      __builtin_amdgcn_sched_barrier(0);
      STAMP(t4);
      ACCUM(0, t0, t1); ACCUM(1, t1, t2); ACCUM(2, t2, t3); ACCUM(3, t3, t4);
#ifdef C3P_STAMPS
      dbg[4] += 1; if (T == 0) { dbg[5] += t3 - t2; } if (LAST && tlast) { dbg[6] += t3 - t2; dbg[7] += 1; dbg[10] += t3 - dbg[12]; }
      if (T == 3) dbg[13] += t3 - t2;
#endif
    });
    sp = (sp + NTAPS) & (NSLOT - 1);
    hb ^= 1u;
    return tlast;
  };
  auto next_tile_or_done = [&]() __attribute__((always_inline)) -> bool {
    if (!more) return true;
    t_id += gridDim.x;
    more = t_id + (int)gridDim.x < total_tiles;
    c = 0;
    return false;
  };
  if constexpr (!LO8) {
    for (;;) {
      if (slab(std::false_type{})) { if (next_tile_or_done()) break; }
      else ++c;
    }
  } else {
    for (;;) {
      for (; c < lo8_c0; ++c) slab(std::false_type{});
      for (;; ++c) if (slab(std::true_type{})) break;
      if (next_tile_or_done()) break;
    }
  }

#ifdef C3P_STAMPS
  if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0)
    for (int i = 0; i < 16; ++i) c3p_dbg[grp][i] = dbg[i];
#endif
  if constexpr ((FLAGS & C3P_STATS) != 0) {   // the last tile's statistics: one more rendezvous of the whole workgroup, then the deferred combine
    full_barrier();
    if (st_prev >= 0) combine_stats();
  }
}

template <int BN>
constexpr size_t c3p_smem() { return (size_t)2 * 41 * 1024 + 4 * BN * 128 + 1024 + 8 * 4 * 16 * 8; }

int c3p_num_cus() {   // per device: the persistent grid is one workgroup per CU
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

template <int BN, bool PAR, int FLAGS>
void launch_c3p(const ConvParams& p, hipStream_t s) {
  const size_t smem = c3p_smem<BN>();
  auto kern = conv3x3p_kernel<BN, PAR, FLAGS>;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)smem);
  const int Ht = PAR ? p.Hin : p.Hout, Wt = PAR ? p.Win : p.Wout;
  const int ntn = (p.N + BN - 1) / BN;
  const int total = p.B * ((Ht + 15) / 16) * ((Wt + 15) / 16) * ntn * (PAR ? 4 : 1);
  // persistent: one workgroup per CU walks the tile list (stride % 8 == 0); LDIFF_C3P_RUN = n > 0 caps a workgroup's walk at n tiles (more,
  // shorter workgroups: the decode side stream then leaves CUs to the UNet stream more often, as LDIFF_C3D_RUN does for the dataflow kernel)
  static const int run_env = [] { const char* e = getenv("LDIFF_C3P_RUN"); return e ? atoi(e) : -1; }();   // -1: 1 tile where the graph shares the chip, else persistent
  const int run_cap = run_env >= 0 ? run_env : (p.short_runs ? 1 : 0);   // (whole step, same box: 176.5 ms persistent, 175.0 / 175.5 / 176.1 at 2 / 4 / 8 tiles)
  int grid = total <= c3p_num_cus() ? total : (c3p_num_cus() & ~7);
  if (run_cap > 0 && total > grid * run_cap) grid = ((total + run_cap - 1) / run_cap + 7) & ~7;
  static const std::string pname = std::string("conv3x3<16x16,") + std::to_string(BN) + ">";
  const double bytes = (double)p.B * p.Hin * p.Win * (p.C1 + p.C2) * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * (p.y_lo ? 4.0 : 2.0) +
                       (p.res ? (double)p.M * p.N * (p.res_lo ? 4.0 : 2.0) : 0.0);
  // flops = MFMA work actually executed: parity mode (nearest-2x folded into 4 taps) runs 16/36 of the 9-tap MACs
  ProfScope prof(pname.c_str(), 2.0 * p.M * (double)p.N * p.K * (PAR ? 16.0 / 36.0 : 1.0), bytes, s);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, p, total);
  HIP_CHECK(hipGetLastError());
}
int c3p_bn(const ConvParams& p) { return p.N <= 64 ? 64 : 128; }

}  // namespace

// The 16x16 ping-pong kernel serves the maps that fill the chip with one 512-thread workgroup per CU and whose outputs take the
// 16-byte store path; everything else (small maps, split-K levels, narrow or fp32 outputs, GroupNorm in front of an upsampling conv)
// stays on the 8x16 / 8x8 kernels.  LDIFF_CONV3X3_PINGPONG=0 switches it off, =2 also takes grids smaller than the chip (A/B timing and tests only).
static bool c3p_instantiated(const ConvParams& p);
bool conv3x3p_selected(const ConvParams& p) {
  static const int mode = [] { const char* e = getenv("LDIFF_CONV3X3_PINGPONG"); return e ? atoi(e) : 1; }();   // 0: off, 2: also for grids smaller than the chip
  if (mode == 0 || p.splitk > 1 || p.out_f32) return false;
  const bool par = p.w_par != nullptr;
  if (p.gn_scale || !c3p_instantiated(p)) return false;   // GroupNorm prologue: the 8x16 kernel (see the header)
  if (p.lo8_slab0 && (par || p.x2 || p.ups || p.ld1 || p.C1 != 96 * p.lo8_slab0 || (p.lo8_slab0 & 1) || !p.lo8_sa)) return false;   // fp8 lo half: one source of 3C/2 "elements", C % 128 == 0
  const int Ht = par ? p.Hin : p.Hout, Wt = par ? p.Win : p.Wout;
  if (Ht < 16 || Wt < 16 || p.N < 48 || (p.N & 7) || (p.ldy & 7) || (p.y_lo & 7)) return false;
  // halo staging addresses the sources through buffer descriptors with a 2^31 out-of-range sentinel
  const long long px = (long long)p.B * p.Hin * p.Win;
  if (px * (p.ld1 ? p.ld1 : p.C1) * 2 >= (1LL << 31) || (p.x2 && px * (p.ld2 ? p.ld2 : p.C2) * 2 >= (1LL << 31))) return false;
  if ((long long)p.M * p.ldy * 2 >= (1LL << 31)) return false;   // the epilogue's buffer stores use the same sentinel
  const int bn = c3p_bn(p);
  const long long wgs = (long long)p.B * ((Ht + 15) / 16) * ((Wt + 15) / 16) * ((p.N + bn - 1) / bn) * (par ? 4 : 1);
  // persistent workgroups, one per CU: the tile list must split evenly (384 tiles on 256 CUs would leave half the chip idle for the
  // second round)
  const int cus = c3p_num_cus();
  const long long rounds = (wgs + cus - 1) / cus;
  return mode == 2 || wgs * 100 >= rounds * cus * 88;
}
int conv3x3p_stats_blocks(const ConvParams& p) {
  const bool par = p.w_par != nullptr;
  const int Ht = par ? p.Hin : p.Hout, Wt = par ? p.Win : p.Wout;
  return ((Ht + 15) / 16) * ((Wt + 15) / 16) * 2 * (par ? 4 : 1);   // one row block per wave group (8 x 16 pixels)
}
// instantiated configurations: 128 channels per tile: every combination (the VAE / UNet resnet convs), 64: no split outputs,
// parity mode (upsampling convs): statistics or nothing
static int c3p_flags(const ConvParams& p) { return (p.res || p.lo8_slab0 ? C3P_RES : 0) | (p.y_lo ? C3P_LO : 0) | (p.stats ? C3P_STATS : 0) | (p.lo8_slab0 ? C3P_LO8 : 0); }
static bool c3p_instantiated(const ConvParams& p) {
  const int f = c3p_flags(p);
  if (p.w_par) return (f & ~C3P_STATS) == 0;
  if (f & C3P_LO8) return c3p_bn(p) == 128 && (f & ~C3P_RES) == (C3P_LO8 | C3P_LO | C3P_STATS);   // the encoder's resnet convs: split output with statistics
  return c3p_bn(p) == 128 || (f & C3P_LO) == 0;
}
template <int BN, bool PAR, int... FS>
static void c3p_dispatch(const ConvParams& p, hipStream_t s, std::integer_sequence<int, FS...>) {
  const int f = c3p_flags(p);
  bool done = false;
  ((f == FS ? (launch_c3p<BN, PAR, FS>(p, s), done = true) : false), ...);
  LDIFF_CHECK(done, LDIFF_ERR_RUNTIME, "conv3x3 (16x16): epilogue configuration %d is not built", f);
}
void launch_conv3x3p(const ConvParams& p, hipStream_t s) {
  if (p.w_par) {
    if (c3p_bn(p) == 64) c3p_dispatch<64, true>(p, s, std::integer_sequence<int, 0, 4>{});
    else c3p_dispatch<128, true>(p, s, std::integer_sequence<int, 0, 4>{});
  } else if (c3p_bn(p) == 64) c3p_dispatch<64, false>(p, s, std::integer_sequence<int, 0, 1, 4, 5>{});
  else c3p_dispatch<128, false>(p, s, std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7, 15>{});
}

#ifdef C3P_STAMPS
extern "C" int ldiff_debug_c3p_stamps(unsigned long long* out) {   // diagnostic build only
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c3p_dbg), sizeof(unsigned long long) * 32);
}
#endif
