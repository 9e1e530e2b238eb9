// 3x3 stride-1 convolution for gfx950 (MI355X): halo-tile implicit GEMM, fp16 operands, fp32 MFMA accumulate.
//
// This is the dominant kernel of the sampling path (97 % of the VAE MACs, 50 % of the UNet MACs).
// It replaces the conv2d inside diffusers' ResnetBlock2D / Upsample2D / conv_in / conv_out
// (reached from /root/reference/segmentor.py:103,106,519 and pixel_latent_vector.py:73,78,81).
//
// Per workgroup (256 threads = 4 waves as 2x2): an output tile of TH x TW pixels of ONE image and BN output channels.
// For every 64-channel slab of the input:
//   * the (TH+2) x (TW+2) halo tile of that slab is loaded ONCE global -> registers -> LDS, with GroupNorm-apply
//     (+SiLU), the nearest-2x upsample gather and the skip-concat source select applied on the way (so the
//     normalisation runs once per element per workgroup instead of once per tap), double-buffered across slabs;
//   * the 9 taps are 9 MFMA steps that read the SAME halo image at a shifted pixel offset, while the [BN][64]
//     weight slice of each tap streams global -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR/ds_write
//     traffic), double-buffered one tap ahead.
// LDS rows are 128 B; 16-byte chunk c of row r lives at position c ^ ((r>>1)&7) (conflict-free ds_read_b128 for
// the 16x16x32 operand fetch).  The DMA writes LDS linearly, so the swizzle is applied on the weight SOURCE address.
#include "common.h"

namespace {

__device__ __forceinline__ int swz8(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int TH, int TW, int BN, bool GN>
__global__ __launch_bounds__(256, 2) void conv3x3_kernel(const ConvParams p) {   // 2 waves/SIMD = 2 workgroups per CU
  constexpr int BM = TH * TW, HWD = TW + 2, HP = (TH + 2) * (TW + 2);
  constexpr int MT = BM / 32, NT = BN / 32;
  constexpr int A_IT = (HP * 8 + 255) / 256, W_IT = BN * 8 / 256;
  static_assert(BN % 32 == 0 && (BN * 8) % 256 == 0, "BN must be a multiple of 32");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* sA = reinterpret_cast<uint4*>(smem_raw);   // [2][HP*8]
  uint4* sW = sA + 2 * HP * 8;                      // [2][BN*8]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (LDS-DMA base, M0)
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int g = lane >> 4, l15 = lane & 15;
  const int Cin = p.C1 + p.C2;
  // Parity mode (nearest-2x upsample folded algebraically): tiles walk the SOURCE grid; workgroup parity q = (py,px)
  // produces output pixels (2y+py, 2x+px) from the 2x2 source neighbourhood {y+py-1, y+py} x {x+px-1, x+px} with the
  // 3x3 taps that alias onto the same source pixel pre-summed (w_par).  4 taps per slab instead of 9.
  const bool par = p.w_par != nullptr;
  const int NTAPS = par ? 4 : 9;
  const int Ht = par ? p.Hin : p.Hout, Wt = par ? p.Win : p.Wout;   // tile-space extent
  const int tiles_x = (Wt + TW - 1) / TW, tiles_y = (Ht + TH - 1) / TH;
  const int ntn = (p.N + BN - 1) / BN;

  // XCD-aware order: blocks sharing an XCD (blockIdx % 8) walk consecutive tiles -> the n-tiles of one pixel tile
  // (same halo) and neighbouring pixel tiles (overlapping halos) share a 4 MiB L2.
  int nwg = gridDim.x, id = blockIdx.x;
  int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7, idx = id >> 3;
  int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  // small maps (conv3x3_img_fast): an XCD's consecutive workgroups share the n-tile (its weight slice stays in that XCD's L2), not the halo
  const int tile_n = p.img_fast ? sw / p.tiles_m : sw % ntn;
  int tm = p.img_fast ? sw - tile_n * p.tiles_m : sw / ntn;
  // parity mode: the four output parities of a source tile are neighbours in the list (they read the same halo: as the slowest grid
  // dimension every parity streamed the whole input from HBM again)
  const int q_par = par ? tm & 3 : 0, py = q_par >> 1, px = q_par & 1;
  tm = par ? tm >> 2 : tm;
  const int tx = tm % tiles_x; tm /= tiles_x;
  const int ty = tm % tiles_y;
  const int b = tm / tiles_y;
  const int n0 = tile_n * BN, oy0 = ty * TH, ox0 = tx * TW;
  const int sh = par ? 0 : p.ups;   // halo coordinates -> source pixel shift
  const int He = p.Hin << sh, We = p.Win << sh;

  // ---- halo staging: thread owns chunk column kc of halo pixels hp = tid/8 + 32*i ----
  const int kc = tid & 7;
  long long a_off[A_IT];   // source pixel offset (in pixels) or -1 when the halo pixel is padding / outside
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int hp = (tid >> 3) + i * 32;
    a_off[i] = -1;
    if (hp < HP) {
      const int hy = hp / HWD, hx = hp - hy * HWD;
      const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
      if (iy >= 0 && iy < He && ix >= 0 && ix < We) a_off[i] = ((long long)b * p.Hin + (iy >> sh)) * p.Win + (ix >> sh);
    }
  }
  uint4 ra[A_IT];
  float4 gs0, gs1, gt0, gt1;
  auto load_halo = [&](int c) {
    const int cb = c * 64;
    const f16* src; int cs, Cs;
    if (cb < p.C1) { src = p.x; cs = cb; Cs = p.ld1 ? p.ld1 : p.C1; } else { src = p.x2; cs = cb - p.C1; Cs = p.ld2 ? p.ld2 : p.C2; }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a_off[i] >= 0) v = *reinterpret_cast<const uint4*>(src + a_off[i] * Cs + cs + kc * 8);
      ra[i] = v;
    }
    if (GN) {
      const float* sc = p.gn_scale + (long long)b * Cin + cb + kc * 8;
      const float* sh = p.gn_shift + (long long)b * Cin + cb + kc * 8;
      gs0 = *reinterpret_cast<const float4*>(sc); gs1 = *reinterpret_cast<const float4*>(sc + 4);
      gt0 = *reinterpret_cast<const float4*>(sh); gt1 = *reinterpret_cast<const float4*>(sh + 4);
    }
  };
  const bool silu = p.silu_in != 0;
  auto store_halo = [&](int buf) {
    float sv[8] = {gs0.x, gs0.y, gs0.z, gs0.w, gs1.x, gs1.y, gs1.z, gs1.w};
    float tv[8] = {gt0.x, gt0.y, gt0.z, gt0.w, gt1.x, gt1.y, gt1.z, gt1.w};
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int hp = (tid >> 3) + i * 32;
      if (hp >= HP) continue;
      uint4 v = ra[i];
      if (GN && a_off[i] >= 0) {   // zero padding applies to the normalised tensor: padding chunks stay exactly 0
        const uint4 r = v;
        if (silu) {
          v.x = gn_pair<true>(r.x, sv[0], tv[0], sv[1], tv[1]); v.y = gn_pair<true>(r.y, sv[2], tv[2], sv[3], tv[3]);
          v.z = gn_pair<true>(r.z, sv[4], tv[4], sv[5], tv[5]); v.w = gn_pair<true>(r.w, sv[6], tv[6], sv[7], tv[7]);
        } else {
          v.x = gn_pair<false>(r.x, sv[0], tv[0], sv[1], tv[1]); v.y = gn_pair<false>(r.y, sv[2], tv[2], sv[3], tv[3]);
          v.z = gn_pair<false>(r.z, sv[4], tv[4], sv[5], tv[5]); v.w = gn_pair<false>(r.w, sv[6], tv[6], sv[7], tv[7]);
        }
      }
      sA[buf * HP * 8 + hp * 8 + swz8(hp, kc)] = v;
    }
  };
  // ---- weight slice of step s = (slab c, tap) by LDS-DMA: LDS position q (linear) <- global chunk (row q/8, (q%8)^swz) ----
  // Per-lane part of the source address is fixed for the whole kernel (32-bit byte offset); per step only a uniform base
  // moves: one DMA = one scalar base + one VGPR offset, no 64-bit vector address arithmetic inside the loop.
  const long long Kw = (long long)NTAPS * Cin;
  const char* wsrc = reinterpret_cast<const char*>(par ? p.w_par + (long long)q_par * p.Nrows * Kw : p.w);
  unsigned w_voff[W_IT];
#pragma unroll
  for (int i = 0; i < W_IT; ++i) {
    const int q = tid + i * 256, r = q >> 3, pos = q & 7;
    int n = n0 + r;
    n = n < p.Nrows ? n : p.Nrows - 1;   // rows beyond the matrix are never stored; keep the address valid
    w_voff[i] = (unsigned)(((long long)n * Kw + swz8(r, pos) * 8) * 2);
  }
  auto issue_w = [&](int c, int tap, int buf) {
    const char* wbase = wsrc + ((long long)tap * Cin + c * 64) * 2;   // uniform
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      uint4* ldst = sW + buf * BN * 8 + i * 256 + wave * 64;   // wave-uniform base; hardware adds lane*16 B
      __builtin_amdgcn_global_load_lds((gptr_t*)(wbase + w_voff[i]), (lptr_t*)ldst, 16, 0, 0);
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[a][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane halo pixel of output row (m-tile, l15) at tap (0,0)
  int hp0[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int ml = wave_m * (BM / 2) + m * 16 + l15;
    hp0[m] = (ml / TW) * HWD + (ml % TW);
  }

  // split-K (small grids, huge K: the 8x8 / 16x16 UNet levels): blockIdx.y owns a contiguous range of input slabs
  const int nslab_all = Cin / 64, S = p.splitk > 1 ? p.splitk : 1, ksplit = blockIdx.y;
  const int c_begin = ksplit * nslab_all / S, nslab = (ksplit + 1) * nslab_all / S;
  load_halo(c_begin);
  store_halo(0);
  issue_w(c_begin, 0, 0);
  __syncthreads();

  int step = 0;
  for (int c = c_begin; c < nslab; ++c) {
    const uint4* cA = sA + ((c - c_begin) & 1) * HP * 8;
#pragma unroll 1
    for (int tap = 0; tap < NTAPS; ++tap, ++step) {
      const bool more = !(c == nslab - 1 && tap == NTAPS - 1);
      if (more) { if (tap == NTAPS - 1) issue_w(c + 1, 0, (step + 1) & 1); else issue_w(c, tap + 1, (step + 1) & 1); }
      const bool stage = tap == 0 && c + 1 < nslab;
      if (stage) load_halo(c + 1);
      const uint4* cW = sW + (step & 1) * BN * 8;
      const int ky = par ? py + (tap >> 1) : tap / 3, kx = par ? px + (tap & 1) : tap - (tap / 3) * 3;
      const int hoff = ky * HWD + kx;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        f16x8 wf[NT], xf[MT];
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          const int row = wave_n * (BN / 2) + a * 16 + l15;
          wf[a] = __builtin_bit_cast(f16x8, cW[row * 8 + swz8(row, kk * 4 + g)]);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int hp = hp0[m] + hoff;
          xf[m] = __builtin_bit_cast(f16x8, cA[hp * 8 + swz8(hp, kk * 4 + g)]);
        }
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[m], acc[a][m], 0, 0, 0);
      }
      if (stage) store_halo((c + 1 - c_begin) & 1);
      __syncthreads();   // also drains the weight DMA issued at the top of this step (vmcnt(0) before s_barrier)
    }
  }

  if (S > 1) {   // raw fp32 partial sums; bias / time embedding / residual are applied by splitk_reduce_kernel
    const int ncol_s = n0 + wave_n * (BN / 2) + g * 4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int ml = wave_m * (BM / 2) + m * 16 + l15;
      const int ty_ = oy0 + ml / TW, tx_ = ox0 + ml % TW;
      const int oy = par ? 2 * ty_ + py : ty_, ox = par ? 2 * tx_ + px : tx_;
      if (ty_ >= Ht || tx_ >= Wt) continue;
      const long long mrow_s = ((long long)b * p.Hout + oy) * p.Wout + ox;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int n = ncol_s + a * 16;
        if (n < p.N) *reinterpret_cast<f32x4*>(p.splitk_ws + ((long long)ksplit * p.M + mrow_s) * p.N + n) = acc[a][m];
      }
    }
    return;
  }
  // ---- epilogue: lane holds y[pixel = column][n = 4g + r] ----
  // All global loads of the epilogue (bias, time embedding, residual) are issued back to back BEFORE any use; a
  // load -> wait -> store chain per 16x16 tile costs a full memory round trip per tile (16-48 us per workgroup).
  const int ncol = n0 + wave_n * (BN / 2) + g * 4;
  f32x4 bt[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    const int n = ncol + a * 16;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), tt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < p.N) {
      if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + n);
      if (p.temb) tt = *reinterpret_cast<const float4*>(p.temb + (long long)b * p.ld_temb + n);
    }
    bt[a] = (f32x4){bb.x + tt.x, bb.y + tt.y, bb.z + tt.z, bb.w + tt.w};
  }
  long long mrow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int ml = wave_m * (BM / 2) + m * 16 + l15;
    const int ty_ = oy0 + ml / TW, tx_ = ox0 + ml % TW;
    const int oy = par ? 2 * ty_ + py : ty_, ox = par ? 2 * tx_ + px : tx_;
    mrow[m] = (ty_ < Ht && tx_ < Wt) ? ((long long)b * p.Hout + oy) * p.Wout + ox : -1;
  }
  f16x4 rr[MT][NT], rl[MT][NT];
  if (p.res) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        rr[m][a] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        rl[m][a] = rr[m][a];
        if (mrow[m] >= 0 && ncol + a * 16 < p.N) {
          rr[m][a] = *reinterpret_cast<const f16x4*>(p.res + mrow[m] * p.ld_res + ncol + a * 16);
          if (p.res_lo) rl[m][a] = *reinterpret_cast<const f16x4*>(p.res + mrow[m] * p.ld_res + p.res_lo + ncol + a * 16);
        }
      }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (mrow[m] < 0) continue;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int n = ncol + a * 16;
      if (n >= p.N) continue;
      f32x4 v = acc[a][m] + bt[a];
      if (p.res) { v += up4(rr[m][a]); if (p.res_lo) v += up4(rl[m][a]); }
      if (p.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + mrow[m] * p.ldy + n) = v;
      } else {
        const f16x4 o = cvt4(v);
        *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + mrow[m] * p.ldy + n) = o;
        if (p.y_lo) *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + mrow[m] * p.ldy + p.y_lo + n) = cvt4(v - up4(o));
        if (p.stats) acc[a][m] = p.y_lo ? v : up4(o);   // what the consumer will read (hi + lo ~ v for a split tensor)
      }
    }
  }
  if (p.stats) {   // fused GroupNorm statistics of this half tile (common.h)
    bool ok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ok[m] = mrow[m] >= 0;
    const long long R = p.stats_R, rblk = (((long long)q_par * tiles_y + ty) * tiles_x + tx) * 2 + wave_m;
    wave_stats_store<MT, NT>(acc, ok, 0, MT, p.stats + ((long long)b * p.N * R + rblk) * 2, R, p.N, ncol, l15);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Wide-tile kernel (8 x 16 pixels x BN channels): the algorithm of conv3x3_kernel with the per-step instruction stream
// cut down.  Counters of the generic loop on MI355X (profiles/r01_conv3x3_issue_profile.md): a wave spends 41 % of its
// lifetime ISSUING (VALU 23 %, SALU 12 %), 22 % issue-stalled and 36 % parked, and one workgroup alone on a CU still
// needs ~1900 cycles per tap for 512 cycles of MFMA: the loop is bound by its own serial instruction stream.  So:
//   * the taps are fully unrolled (template on the parity mode): ky, kx and every tap offset are immediates;
//   * the halo image is swizzled by the COLUMN hx of the halo pixel (not its linear index), so the X fragment address
//     of tap (ky,kx), m-tile m is base[kx] + (m+ky)*18*128: three per-lane bases per slab, m and ky in the ds_read
//     offset field (conflict-free because a 16x16x32 operand row block is 16 consecutive pixels of ONE halo row);
//   * weight slices by `buffer_load_dwordx4 ... lds`: one descriptor, per-lane voffsets fixed for the kernel, a step
//     moves only the scalar soffset; the pieces of a wave are contiguous in LDS, so one M0 serves them through the
//     instruction offset (the hardware adds it to BOTH the memory and the LDS address: the voffset is pre-compensated);
//   * operand reads as inline asm with counted lgkmcnt waits (hipcc only ever emits lgkmcnt(0) there);
//   * halo staging without exec masking (clamped source address, AND mask, dump slot for the lanes beyond the halo),
//     GroupNorm+SiLU of chunk i spread over taps 1.., SiLU chosen by one uniform branch per chunk.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int OFF>
__device__ __forceinline__ void lds_read128(f16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// s_waitcnt lgkmcnt(CNT); the fragments it covers are in/out operands so no MFMA consuming them can move above it
template <int CNT>
__device__ __forceinline__ void lds_wait(f16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void lds_wait(f16x8& a, f16x8& b, f16x8& c, f16x8& d, f16x8& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(CNT));
}

// Column swizzle of the halo image: 16-byte chunk c of halo pixel (hy, hx) sits at position c ^ swzx(hx) of its 128-B row.
// A 16x16x32 operand fetch reads 16 consecutive pixels of one halo row starting at column kx = 0, 1 or 2; its four
// ds_read_b128 lane groups mix chunks c (8 lanes) and c^1 (8 lanes).  The table (one 3-bit entry per column PAIR, found by
// exhaustive search against the bank model of MI355X_MICROARCH.md, LDS) makes all three starts conflict-free; the plain
// (hx>>1)&7 is conflict-free only for kx = 0 and costs 2x on the other two thirds of the taps (SQ_LDS_BANK_CONFLICT = 26 %
// of the LDS-active cycles).  The position is an XOR of the chunk index, so k-half 1 is still byte address ^ 64.
__device__ __forceinline__ int swzx(int hx) { return (0xcb5888 >> (3 * (hx >> 1))) & 7; }

#ifdef C3W_STAMPS   // diagnostic build only (scripts/build_c3w_stamps.sh, scripts/conv_stamps_w.py): where a (slab, tap) step of wave 0 of workgroup 0 goes
__device__ unsigned long long c3w_dbg[32];   // [0..4] totals, [8 + T] MFMA segment of tap T, [20 + T] issue segment of tap T
__device__ __forceinline__ unsigned long long w_stamp() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define WSTAMP(v) const unsigned long long v = w_stamp()
#define WACC(i, expr) wdbg[i] += (expr)
#else
#define WSTAMP(v)
#define WACC(i, expr)
#endif

template <int BN, bool GN>
__global__ __launch_bounds__(256, BN > 128 ? 1 : 2) void conv3x3w_kernel(const ConvParams p) {
  constexpr int TH = 8, TW = 16, BM = 128, HWD = 18, HP = 180, MT = 4, NT = BN / 32, A_IT = 6, NP = BN / 32;
  constexpr int ROWB = HWD * 128;                                  // bytes per halo row
  // R3 (round 6; the 160-column kernel, which runs ONE workgroup per CU: nothing else covers a wave's issue time): THREE weight slots, the slice of step
  // s + 2 fetched during step s, its DMA pieces issued BETWEEN the step's first MFMA groups instead of in front of them, and a counted vmcnt at the
  // step's barrier that leaves those pieces in flight.  In-kernel stamps (scripts/conv_stamps_w.py, profiles/r06_conv3x3w_stamps.txt): a step of wave 0 was
  // 28 % issue (mostly the five LDS-DMA pieces, 60+ cycles each), 67 % MFMA groups, 5 % barrier.  Nine taps per slab = a multiple of three: the slot of tap T
  // is T % 3 whatever the slab.  (Parity mode, four taps, keeps the two-slot scheme.)
#ifdef C3W_RING2   // (A/B build: the two-slot scheme everywhere)
  constexpr bool R3 = false;
#else
  constexpr bool R3 = BN == 160;
#endif
  constexpr int NSLOT = R3 ? 3 : 2;
  constexpr unsigned W_OFF = 2 * HP * 128, W_BYTES = BN * 128;     // LDS map: halo[2] | weights[NSLOT] | dump
  constexpr unsigned DUMP = W_OFF + NSLOT * W_BYTES;               // 96 lanes x 16 B: stores of lanes beyond the halo
  constexpr int NF = NT + MT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int g = lane >> 4, l15 = lane & 15;
  const int Cin = p.C1 + p.C2;
  const bool par = p.w_par != nullptr;
  const int Ht = par ? p.Hin : p.Hout, Wt = par ? p.Win : p.Wout;
  const int tiles_x = (Wt + TW - 1) / TW, tiles_y = (Ht + TH - 1) / TH;
  const int ntn = (p.N + BN - 1) / BN;
  int nwg = gridDim.x, id = blockIdx.x;
  int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7, idx = id >> 3;
  int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  // tile decode with the launcher's reciprocals (q = umulhi(n, floor(2^32 / d) + 1), exact while n * d < 2^32): a run-time integer
  // division is ~25 dependent scalar / vector instructions with a VALU -> SALU round trip, and three of them stood in front of the
  // first halo load of every tile
  auto fdiv = [](int n, unsigned m) { return m ? (int)__umulhi((unsigned)n, m) : n; };   // m = 0 encodes a divisor of 1
  int tm, tile_n;
  if (p.img_fast) { tile_n = fdiv(sw, p.div_tm); tm = sw - tile_n * p.tiles_m; }   // (see conv3x3_kernel)
  else { tm = fdiv(sw, p.div_ntn); tile_n = sw - tm * ntn; }
  const int q_par = par ? tm & 3 : 0, py = q_par >> 1, px = q_par & 1;   // parity next to the n-tile (see conv3x3_kernel)
  tm = par ? tm >> 2 : tm;
  const int tq = fdiv(tm, p.div_tx);
  const int tx = tm - tq * tiles_x;
  const int b = fdiv(tq, p.div_ty);
  const int ty = tq - b * tiles_y;
  const int n0 = tile_n * BN, oy0 = ty * TH, ox0 = tx * TW;
  const int sh = par ? 0 : p.ups;
  const int He = p.Hin << sh, We = p.Win << sh;
  const unsigned lds0 = (unsigned)(size_t)(lptr_t*)smem_raw;
#ifdef C3W_STAMPS
  unsigned long long wdbg[32] = {0};
  WSTAMP(w_t0);
#endif

  // ---- halo staging: thread owns chunk column kc of halo pixels hp = tid/8 + 32*i ----
  const int kc = tid & 7;
  unsigned a_pix[A_IT], a_msk[A_IT], a_lds[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int hp = (tid >> 3) + i * 32;
    const int hy = hp / HWD, hx = hp - hy * HWD;
    const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
    const bool inb = hp < HP && iy >= 0 && iy < He && ix >= 0 && ix < We;
    a_pix[i] = inb ? (unsigned)(((b * p.Hin + (iy >> sh)) * p.Win) + (ix >> sh)) : 0u;
    a_msk[i] = inb ? 0xffffffffu : 0u;   // zero padding applies to the NORMALISED tensor: padding chunks are exactly 0
    a_lds[i] = hp < HP ? (unsigned)(hp * 128 + ((kc ^ swzx(hx)) << 4)) : DUMP + (unsigned)((tid - 160) * 16);
  }
  uint4 ra[A_IT];
  float4 gs0, gs1, gt0, gt1;
  auto load_halo = [&](int c) {
    const int cb = c * 64;
    const f16* src; int cs, Cs;
    if (cb < p.C1) { src = p.x; cs = cb; Cs = p.ld1 ? p.ld1 : p.C1; } else { src = p.x2; cs = cb - p.C1; Cs = p.ld2 ? p.ld2 : p.C2; }
    src += cs + kc * 8;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) ra[i] = *reinterpret_cast<const uint4*>(src + (size_t)a_pix[i] * (unsigned)Cs);
    if (GN) {
      const float* sc = p.gn_scale + (long long)b * Cin + cb + kc * 8;
      const float* st = p.gn_shift + (long long)b * Cin + cb + kc * 8;
      gs0 = *reinterpret_cast<const float4*>(sc); gs1 = *reinterpret_cast<const float4*>(sc + 4);
      gt0 = *reinterpret_cast<const float4*>(st); gt1 = *reinterpret_cast<const float4*>(st + 4);
    }
  };
  // the first slab's halo goes out before the rest of the set-up (weight descriptor, operand addresses, accumulators), which then runs
  // under the global round trip
  const int nslab_all = Cin >> 6, S = p.splitk > 1 ? p.splitk : 1, ksplit = blockIdx.y;
  int c_begin = 0, nslab = nslab_all;
  if (S > 1) { c_begin = ksplit * nslab_all / S; nslab = (ksplit + 1) * nslab_all / S; }
  load_halo(c_begin);
  __builtin_amdgcn_sched_barrier(0);
  const bool silu = p.silu_in != 0;
  auto xform_store = [&](auto ic, unsigned bufoff) {   // chunk i of the staged slab -> LDS (normalised, activated, masked)
    constexpr int i = decltype(ic)::value;
    uint4 v = ra[i];
    if (GN) {
      if (silu) {
        v.x = gn_pair<true>(ra[i].x, gs0.x, gt0.x, gs0.y, gt0.y); v.y = gn_pair<true>(ra[i].y, gs0.z, gt0.z, gs0.w, gt0.w);
        v.z = gn_pair<true>(ra[i].z, gs1.x, gt1.x, gs1.y, gt1.y); v.w = gn_pair<true>(ra[i].w, gs1.z, gt1.z, gs1.w, gt1.w);
      } else {
        v.x = gn_pair<false>(ra[i].x, gs0.x, gt0.x, gs0.y, gt0.y); v.y = gn_pair<false>(ra[i].y, gs0.z, gt0.z, gs0.w, gt0.w);
        v.z = gn_pair<false>(ra[i].z, gs1.x, gt1.x, gs1.y, gt1.y); v.w = gn_pair<false>(ra[i].w, gs1.z, gt1.z, gs1.w, gt1.w);
      }
    }
    v.x &= a_msk[i]; v.y &= a_msk[i]; v.z &= a_msk[i]; v.w &= a_msk[i];
    *reinterpret_cast<uint4*>(smem_raw + bufoff + a_lds[i]) = v;
  };

  // ---- weight slices: wave w owns rows [w*BN/4, (w+1)*BN/4) of the [BN][64] slice = NP pieces of 8 rows (1 KiB) ----
  const long long Kw = (long long)(par ? 4 : 9) * Cin;
  const f16* wsrc = par ? p.w_par + (long long)q_par * p.Nrows * Kw : p.w;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, 0, (int)((long long)p.Nrows * Kw * 2), 0x00020000);
  int w_voff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int r = wave * (BN / 4) + i * 8 + (lane >> 3), pos = lane & 7;
    int n = n0 + r;
    n = n < p.Nrows ? n : p.Nrows - 1;   // rows beyond the matrix are never stored; keep the address in range
    w_voff[i] = (int)(((long long)n * Kw + swz8(r, pos) * 8) * 2) - (i & 3) * 1024;
  }
  auto issue_w_piece = [&](auto ic, int soff, unsigned slot) {   // one 1-KiB piece of the slice (R3: issued between MFMA groups)
    constexpr int i = decltype(ic)::value;
    unsigned char* dst = smem_raw + W_OFF + slot * W_BYTES + wave * (BN * 32);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t*)(dst + (i >> 2) * 4096), 16, w_voff[i], soff, (i & 3) * 1024, 0);
#endif
  };
  auto issue_w = [&](int soff, unsigned slot) {   // soff: byte offset of (tap, slab) inside a weight row
    unsigned char* dst = smem_raw + W_OFF + slot * W_BYTES + wave * (BN * 32);
    static_for<0, NP>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass rejects this builtin (target feature) and then silently drops the kernel stub
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lptr_t*)(dst + (i >> 2) * 4096), 16, w_voff[i], soff, (i & 3) * 1024, 0);
#endif
    });
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[a][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane operand addresses (k-half 0): X base of column shift j (kx = j, or px + j in parity mode), W row of slot 0
  unsigned xb[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int hx = l15 + (par ? px + (j & 1) : j);
    xb[j] = lds0 + (unsigned)(((wave_m * 4 + (par ? py : 0)) * HWD + hx) * 128 + ((g ^ swzx(hx)) << 4));
  }
  const int wrow = wave_n * (BN / 2) + l15;   // + a*16: same swizzle phase, +2048 B per a
  const unsigned w_lane = lds0 + W_OFF + (unsigned)(wrow * 128 + ((g ^ ((wrow >> 1) & 7)) << 4));

  issue_w(c_begin * 128, 0);
  if (R3 && !par) issue_w((Cin + c_begin * 64) * 2, 1);   // three slots: step 1's slice too (a slab always has nine steps)
  static_for<0, A_IT>([&](auto ic) { xform_store(ic, 0u); });
  __syncthreads();

  // GroupNorm(+SiLU) of ONE element of a staged chunk; issued between the MFMAs of the k-half-1 groups so that its
  // VALU/transcendental issue slots fall into the shadow of the matrix pipe (an MFMA holds the vector issue port for 8 of
  // its 16 cycles) and the k-half-0 fragments' registers are free for its temporaries
  auto xform_pair = [&](auto ic, auto dc, auto siluc) -> unsigned {   // dword d (elements 2d, 2d+1) of staged chunk i
    constexpr int i = decltype(ic)::value, d = decltype(dc)::value;
    return gn_pair<decltype(siluc)::value>(dword4<d>(ra[i]), comp8<2 * d>(gs0, gs1), comp8<2 * d>(gt0, gt1), comp8<2 * d + 1>(gs0, gs1), comp8<2 * d + 1>(gt0, gt1));
  };
  auto mask_store = [&](auto ic, uint4 v, unsigned bufoff) {
    constexpr int i = decltype(ic)::value;
    v.x &= a_msk[i]; v.y &= a_msk[i]; v.z &= a_msk[i]; v.w &= a_msk[i];
    *reinterpret_cast<uint4*>(smem_raw + bufoff + a_lds[i]) = v;
  };

  // the (slab, tap) loop; PAR (parity mode) and SILU static.  sp = LDS weight slot of the slab's tap 0.
  auto run = [&](auto parc, auto siluc) {
    constexpr bool PAR = decltype(parc)::value;
    constexpr int NTAPS = PAR ? 4 : 9, CPT = PAR ? 2 : 1;   // chunks of the next slab transformed per tap
    unsigned sp = 0;
    // one slab; STAGE = another slab follows (its halo is fetched, normalised and stored during this one)
    auto slab = [&](int c, auto stagec) {
      constexpr bool STAGE = decltype(stagec)::value, stage = STAGE;
      const unsigned hbuf = (unsigned)((c - c_begin) & 1) * (HP * 128);
      unsigned xc[3], xc1[3], wc[2], wc1[2];
#pragma unroll
      for (int j = 0; j < 3; ++j) { xc[j] = xb[j] + hbuf; xc1[j] = xc[j] ^ 64u; }   // chunk bit 2 = k-half: XOR commutes with the swizzle
      wc[0] = w_lane + sp * W_BYTES; wc[1] = w_lane + (sp ^ 1u) * W_BYTES;
      wc1[0] = wc[0] ^ 64u; wc1[1] = wc[1] ^ 64u;
      constexpr bool RING3 = R3 && !PAR;
      static_for<0, NTAPS>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        constexpr int kyi = PAR ? (T >> 1) : T / 3, kxi = PAR ? (T & 1) : T % 3;
        constexpr bool LAST = T == NTAPS - 1;
        // chunks [C0, C1) of the NEXT slab are normalised during this tap
        constexpr int C0 = T >= 1 ? (T - 1) * CPT : A_IT, C1 = T >= 1 ? (T * CPT < A_IT ? T * CPT : A_IT) : A_IT;
        constexpr int NE = GN && STAGE && C0 < C1 ? (C1 - C0) * 4 : 0, EPG = (NE + NT - 1) / NT;   // NE element PAIRS (dwords), EPG per MFMA group
        unsigned pk[NE > 0 ? NE : 1];         // normalised elements, packed f16 pairs
        f16x8 wf[2][NT], xf[2][MT];
        WSTAMP(s0);
        // k-half 0 up front; the k-half-1 reads go out between the first MFMA groups (an LDS instruction issues while the matrix pipe
        // works: all sixteen in front of the first MFMA cost ~100 cycles of every step)
        // RING3: slot T % 3 (static); the slice of step + 2 goes out piece by piece between the first MFMA groups below
        const unsigned wcur = RING3 ? w_lane + (unsigned)(T % 3) * W_BYTES : wc[T & 1], wcur1 = wcur ^ 64u;
        constexpr int T2 = T + 2;                                  // the step whose slice is fetched during this one (RING3)
        constexpr bool DMA_NOW = RING3 && (T2 < NTAPS || STAGE);   // ... if it exists: the same slab's tap T2, or tap T2 - 9 of the next slab
        const int soff2 = T2 < NTAPS ? (T2 * Cin + c * 64) * 2 : ((T2 - NTAPS) * Cin + (c + 1) * 64) * 2;
        static_for<0, MT>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<(m + kyi) * ROWB>(xf[0][m], xc[kxi]); });
        static_for<0, NT>([&](auto ac) { constexpr int a = decltype(ac)::value; lds_read128<a * 2048>(wf[0][a], wcur); });
        __builtin_amdgcn_sched_barrier(0);
        // next step's weight slice -> the other slot (read last in the previous step); issued while the operand reads fly
        if constexpr (!RING3) {
          if (!LAST) issue_w(((T + 1) * Cin + c * 64) * 2, sp ^ (unsigned)((T + 1) & 1));
          else if (stage) issue_w((c + 1) * 128, sp ^ (unsigned)(NTAPS & 1));
        }
        if (T == 0 && stage) load_halo(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        WSTAMP(s1);   // (the stamp drains the LDS queue: the first MFMA group no longer waits for its operands inside the next segment)
        static_for<0, 2 * NT>([&](auto ic) {
          constexpr int kk = decltype(ic)::value / NT, a = decltype(ic)::value % NT;
          constexpr int WI = NT > 1 ? 1 : 0;   // the k-half-1 X reads follow group 0 of k-half 0, the W reads group WI
          // reads issued after W_a of this k-half at the time of the wait
          constexpr int pending = kk == 1 ? NT - 1 - a : (NT - 1 - a) + (a >= 1 ? MT : 0) + (a > WI ? NT : 0);
          if constexpr (a == 0) lds_wait<pending>(xf[kk][0], xf[kk][1], xf[kk][2], xf[kk][3], wf[kk][0]);
          else lds_wait<pending>(wf[kk][a]);
#pragma unroll
          for (int m = 0; m < MT; ++m)
            acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][a], xf[kk][m], acc[a][m], 0, 0, 0);
          if constexpr (kk == 0 && a == 0) static_for<0, MT>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<(m + kyi) * ROWB>(xf[1][m], xc1[kxi]); });
          if constexpr (kk == 0 && a == WI) static_for<0, NT>([&](auto bc) { constexpr int b2 = decltype(bc)::value; lds_read128<b2 * 2048>(wf[1][b2], wcur1); });
          if constexpr (DMA_NOW && kk == 0 && a < NP) issue_w_piece(std::integral_constant<int, a>{}, soff2, (unsigned)(T2 % 3));   // (NP == NT: one piece behind each k-half-0 group)
          static_for<(kk == 1 ? a * EPG : NE), (kk == 1 ? ((a + 1) * EPG < NE ? (a + 1) * EPG : NE) : NE)>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            pk[e] = xform_pair(std::integral_constant<int, C0 + e / 4>{}, std::integral_constant<int, e % 4>{}, siluc);
          });
          __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (STAGE) static_for<C0, C1>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          uint4 v = ra[i];
          if constexpr (GN) v = make_uint4(pk[(i - C0) * 4], pk[(i - C0) * 4 + 1], pk[(i - C0) * 4 + 2], pk[(i - C0) * 4 + 3]);
          mask_store(ic, v, (unsigned)(HP * 128) - hbuf);
        });
        WSTAMP(s2);
        if constexpr (RING3) {
          // the slice the NEXT step reads was issued a step ago: everything older than this step's NP pieces must have landed (the halo loads of tap 0,
          // issued in front of them, included); the staged halo chunk's LDS stores must be out before the barrier publishes them
          if constexpr (DMA_NOW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NP) : "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        } else __syncthreads();   // drains this step's weight DMA (vmcnt(0)) and publishes it
        WSTAMP(s3);
        WACC(1, s1 - s0); WACC(2, s2 - s1); WACC(3, s3 - s2); WACC(4, 1); WACC(8 + T, s2 - s1); WACC(20 + T, s1 - s0);
      });
      sp ^= (unsigned)(NTAPS & 1);
    };
    for (int c = c_begin; c + 1 < nslab; ++c) slab(c, std::true_type{});
    slab(nslab - 1, std::false_type{});
  };
  if (GN && silu) { if (par) run(std::true_type{}, std::true_type{}); else run(std::false_type{}, std::true_type{}); }
  else { if (par) run(std::true_type{}, std::false_type{}); else run(std::false_type{}, std::false_type{}); }
#ifdef C3W_STAMPS
  { WSTAMP(w_t1); wdbg[0] = w_t1 - w_t0; }   // kernel start -> end of the main loop (set-up and the first halo / weight round trip included; the epilogue is not)
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) for (int i = 0; i < 32; ++i) c3w_dbg[i] = wdbg[i];
#endif

  if (S > 1) {   // raw fp32 partial sums; bias / time embedding / residual are applied by splitk_reduce_kernel
    const int ncol_s = n0 + wave_n * (BN / 2) + g * 4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int ml = wave_m * (BM / 2) + m * 16 + l15;
      const int ty_ = oy0 + ml / TW, tx_ = ox0 + ml % TW;
      const int oy = par ? 2 * ty_ + py : ty_, ox = par ? 2 * tx_ + px : tx_;
      if (ty_ >= Ht || tx_ >= Wt) continue;
      const long long mrow_s = ((long long)b * p.Hout + oy) * p.Wout + ox;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int n = ncol_s + a * 16;
        if (n < p.N) *reinterpret_cast<f32x4*>(p.splitk_ws + ((long long)ksplit * p.M + mrow_s) * p.N + n) = acc[a][m];
      }
    }
    return;
  }
  // ---- epilogue: lane holds y[pixel = column][n = 4g + r] ----
  // All global loads of the epilogue (bias, time embedding, residual) are issued back to back BEFORE any use; a
  // load -> wait -> store chain per 16x16 tile costs a full memory round trip per tile (16-48 us per workgroup).
  const int ncol = n0 + wave_n * (BN / 2) + g * 4;
  f32x4 bt[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    const int n = ncol + a * 16;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f), tt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < p.N) {
      if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias + n);
      if (p.temb) tt = *reinterpret_cast<const float4*>(p.temb + (long long)b * p.ld_temb + n);
    }
    bt[a] = (f32x4){bb.x + tt.x, bb.y + tt.y, bb.z + tt.z, bb.w + tt.w};
  }
  long long mrow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int ml = wave_m * (BM / 2) + m * 16 + l15;
    const int ty_ = oy0 + ml / TW, tx_ = ox0 + ml % TW;
    const int oy = par ? 2 * ty_ + py : ty_, ox = par ? 2 * tx_ + px : tx_;
    mrow[m] = (ty_ < Ht && tx_ < Wt) ? ((long long)b * p.Hout + oy) * p.Wout + ox : -1;
  }
  f16x4 rr[MT][NT], rl[MT][NT];
  if (p.res) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        rr[m][a] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        rl[m][a] = rr[m][a];
        if (mrow[m] >= 0 && ncol + a * 16 < p.N) {
          rr[m][a] = *reinterpret_cast<const f16x4*>(p.res + mrow[m] * p.ld_res + ncol + a * 16);
          if (p.res_lo) rl[m][a] = *reinterpret_cast<const f16x4*>(p.res + mrow[m] * p.ld_res + p.res_lo + ncol + a * 16);
        }
      }
  }
  // 16-byte stores: lanes g and g+1 hold adjacent 4-channel groups of the SAME pixel.  One v_permlane16_swap per dword
  // between the packed values of two m-tiles P, Q leaves lanes with even g holding channels 4g..4g+7 of pixel P and lanes
  // with odd g channels 4(g-1)..4(g-1)+7 of pixel Q: half as many store instructions for the same bytes (the store tail of an
  // MFMA epilogue is issue-bound per instruction).
  if (!p.out_f32 && (p.N & 7) == 0 && (p.ldy & 7) == 0 && (p.y_lo & 7) == 0) {
#pragma unroll
    for (int mp = 0; mp < MT; mp += 2) {
      // this lane's pixel after the swap, from arithmetic (a select between mrow[] entries becomes a scratch-indexed load)
      const int mls = wave_m * (BM / 2) + (mp + (g & 1)) * 16 + l15;
      const int tys = oy0 + mls / TW, txs = ox0 + mls % TW;
      const long long row = (tys < Ht && txs < Wt) ? ((long long)b * p.Hout + (par ? 2 * tys + py : tys)) * p.Wout + (par ? 2 * txs + px : txs) : -1;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        uint2 pq[2], pl[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = mp + h;
          f32x4 v = acc[a][m] + bt[a];
          if (p.res) { v += up4(rr[m][a]); if (p.res_lo) v += up4(rl[m][a]); }
          const f16x4 o = cvt4(v);
          if (p.stats) acc[a][m] = p.y_lo ? v : up4(o);   // what the consumer will read (hi + lo ~ v for a split tensor)
          pq[h] = __builtin_bit_cast(uint2, o);
          pl[h] = __builtin_bit_cast(uint2, cvt4(v - up4(o)));
        }
        auto r0 = __builtin_amdgcn_permlane16_swap(pq[0].x, pq[1].x, false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(pq[0].y, pq[1].y, false, false);
        const int nb = n0 + wave_n * (BN / 2) + a * 16 + (g & ~1) * 4;
        if (row >= 0 && nb < p.N) *reinterpret_cast<uint4*>(reinterpret_cast<f16*>(p.y) + row * p.ldy + nb) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        if (p.y_lo) {   // lo halves of the split output, same swap
          auto l0 = __builtin_amdgcn_permlane16_swap(pl[0].x, pl[1].x, false, false);
          auto l1 = __builtin_amdgcn_permlane16_swap(pl[0].y, pl[1].y, false, false);
          if (row >= 0 && nb < p.N) *reinterpret_cast<uint4*>(reinterpret_cast<f16*>(p.y) + row * p.ldy + p.y_lo + nb) = make_uint4(l0[0], l1[0], l0[1], l1[1]);
        }
      }
    }
  } else
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (mrow[m] < 0) continue;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int n = ncol + a * 16;
      if (n >= p.N) continue;
      f32x4 v = acc[a][m] + bt[a];
      if (p.res) { v += up4(rr[m][a]); if (p.res_lo) v += up4(rl[m][a]); }
      if (p.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + mrow[m] * p.ldy + n) = v;
      } else {
        const f16x4 o = cvt4(v);
        *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + mrow[m] * p.ldy + n) = o;
        if (p.y_lo) *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + mrow[m] * p.ldy + p.y_lo + n) = cvt4(v - up4(o));
        if (p.stats) acc[a][m] = p.y_lo ? v : up4(o);   // what the consumer will read
      }
    }
  }
  if (p.stats) {   // fused GroupNorm statistics of this pixel tile (common.h): one row block per workgroup
    // per wave: sums over its 64 pixels (in-lane over the m-tiles, DPP over the 16 pixels of a row group); the two waves that
    // share a channel half (wave_m = 0, 1) are combined through LDS (free after the main loop's last barrier), so the
    // finalize kernel reads half as many partials as with one block per wave
    if constexpr (NT == 2 || NT == 4) {
      // all NT*4 channel sums of the wave reduced over the 16 pixel lanes at once, totals spread over the lanes (common.h,
      // row16_reduce_spread): a quarter of the DPP adds, and one LDS write / one read / one store per lane instead of 16 by one lane in 16
      constexpr int NV = NT * 4;
      float sv[NV], qv[NV];
#pragma unroll
      for (int j = 0; j < NV; ++j) { sv[j] = 0.f; qv[j] = 0.f; }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = mrow[m] < 0 ? 0.f : acc[a][m][r]; sv[a * 4 + r] += v; qv[a * 4 + r] += v * v; }
      }
      const float st_s = row16_reduce_spread<NV>(sv, l15), st_q = row16_reduce_spread<NV>(qv, l15);
      const int jv = ((l15 >> 3) & 1) | ((l15 >> 1) & 2) | ((l15 << 1) & 4) | ((l15 << 3) & 8);   // value index a*4 + r of this lane's totals
      float2* xch = reinterpret_cast<float2*>(smem_raw) + (wave_n * 4 + g) * 16 + jv;        // [wave_n][g][16]
      if (wave_m == 1) *xch = make_float2(st_s, st_q);
      __syncthreads();
      const int n = ncol + (jv >> 2) * 16 + (jv & 3);
      if (wave_m == 0 && (NV == 16 || jv < NV) && n < p.N) {
        const long long R = p.stats_R, rblk = ((long long)q_par * tiles_y + ty) * tiles_x + tx;
        const float2 t = *xch;
        *reinterpret_cast<float2*>(p.stats + ((long long)b * p.N * R + rblk) * 2 + (long long)n * R * 2) = make_float2(st_s + t.x, st_q + t.y);
      }
      return;
    }
    float2 sq[NT][4];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      float sv[4] = {0.f, 0.f, 0.f, 0.f}, qv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (mrow[m] < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float v = acc[a][m][r]; sv[r] += v; qv[r] += v * v; }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) sq[a][r] = make_float2(row16_sum(sv[r]), row16_sum(qv[r]));
    }
    float2* xch = reinterpret_cast<float2*>(smem_raw);   // [wave_n][a][g][r]
    if (wave_m == 1 && l15 == 0) {
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[((wave_n * NT + a) * 4 + g) * 4 + r] = sq[a][r];
    }
    __syncthreads();
    if (wave_m == 0 && l15 == 0) {
      const long long R = p.stats_R, rblk = ((long long)q_par * tiles_y + ty) * tiles_x + tx;
      float* dst = p.stats + ((long long)b * p.N * R + rblk) * 2;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int n = ncol + a * 16;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float2 t = xch[((wave_n * NT + a) * 4 + g) * 4 + r];
          *reinterpret_cast<float2*>(dst + (long long)(n + r) * R * 2) = make_float2(sq[a][r].x + t.x, sq[a][r].y + t.y);
        }
      }
    }
  }
}


// split-K reduction + epilogue: y[m, n..n+3] = sum_s ws[s][m][n..] + bias + temb + res
// One row's four columns: every partial (and the epilogue's operands) is loaded BEFORE the first add -- as a plain `for s: v += load` the compiler waits
// for each load in turn, S dependent memory round trips per thread (14.6 us per launch on 40 launches of a UNet pass, round 5); the sum order is s = 0, 1, ...
constexpr int SPLITK_MAX = 16;
template <int SM>   // SM = 8: plans of up to eight splits (every launch before round 6); 16: the finer cuts of the small grids
__device__ __forceinline__ f32x4 splitk_row(const ConvParams& p, long long m, int n) {
  const float* wsp = p.splitk_ws + m * p.N + n;
  const long long sstride = (long long)p.M * p.N;
  f32x4 part[SM];
  const int S = p.splitk;
#pragma unroll
  for (int s = 0; s < SM; ++s) part[s] = *reinterpret_cast<const f32x4*>(wsp + (s < S ? s : S - 1) * sstride);   // (unconditional: a slot beyond S re-reads the last one, unused)
  // (null operands read a valid dummy address -- the partials -- and are masked out: no branch, hence no wait, between the loads)
  const float4 tb = *reinterpret_cast<const float4*>(p.bias ? p.bias + n : wsp);
  const float4 tt = *reinterpret_cast<const float4*>(p.temb ? p.temb + (m / ((long long)p.Hout * p.Wout)) * p.ld_temb + n : wsp);
  const f16x4 rh = *reinterpret_cast<const f16x4*>(p.res ? p.res + m * p.ld_res + n : reinterpret_cast<const f16*>(wsp));
  const f16x4 rl = *reinterpret_cast<const f16x4*>((p.res && p.res_lo) ? p.res + m * p.ld_res + p.res_lo + n : reinterpret_cast<const f16*>(wsp));
  f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < SM; ++s) if (s < S) v += part[s];
  if (p.bias) { v[0] += tb.x; v[1] += tb.y; v[2] += tb.z; v[3] += tb.w; }
  if (p.temb) { v[0] += tt.x; v[1] += tt.y; v[2] += tt.z; v[3] += tt.w; }
  if (p.res) {
    v += up4(rh);
    if (p.res_lo) v += up4(rl);
  }
  return v;
}
// stores the row's four columns; returns what the consumer will read (the statistics are of THAT: hi + lo ~ v for a split tensor, else the fp16 value)
__device__ __forceinline__ f32x4 splitk_store(const ConvParams& p, long long m, int n, const f32x4& v) {
  if (p.out_f32) { *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + m * p.ldy + n) = v; return v; }
  const f16x4 o = cvt4(v);
  *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + m * p.ldy + n) = o;
  if (p.y_lo) { *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + m * p.ldy + p.y_lo + n) = cvt4(v - up4(o)); return v; }
  return up4(o);
}
template <int SM>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvParams p) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int n4 = p.N >> 2;
  if (i >= (long long)p.M * n4) return;
  const long long m = i / n4;
  const int n = (int)(i - m * n4) * 4;
  splitk_store(p, m, n, splitk_row<SM>(p, m, n));
}
// The same with the fused GroupNorm partial statistics of the output (common.h: stats[((b N + n) R + r) 2 + {0, 1}], r = (m % HW) / 32, R = HW / 32 --
// the layout of the GEMM kernels' epilogues): a split-K launch's own epilogue never sees final values, so before round 6 every such tensor paid a
// separate statistics pass (gn_stats: 27 launches per UNet pass at B = 8, 43 at B = 1 where nearly every conv is split).  grid (M / 32, ceil(N / 64)),
// thread (row pair rr, column quad cq): rows rr and rr + 16 of the block, four columns; the 16 row-threads of a column combine through LDS in fixed order.
template <int SM>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const ConvParams p) {
  __shared__ float2 red[16][64];
  const int tid = threadIdx.x, cq = tid & 15, rr = tid >> 4;
  const int n = blockIdx.y * 64 + cq * 4;
  const long long m0 = (long long)blockIdx.x * 32;
  float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < p.N) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long m = m0 + rr + h * 16;
      const f32x4 o = splitk_store(p, m, n, splitk_row<SM>(p, m, n));
#pragma unroll
      for (int r = 0; r < 4; ++r) { sm[r] += o[r]; sq[r] = __builtin_fmaf(o[r], o[r], sq[r]); }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[rr][cq * 4 + r] = make_float2(sm[r], sq[r]);
  __syncthreads();
  if (tid < 64 && blockIdx.y * 64 + tid < p.N) {
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { a += red[k][tid].x; q += red[k][tid].y; }
    const int hw = p.Hout * p.Wout;
    const long long b = m0 / hw;
    const int rblk = (int)(m0 - b * hw) >> 5;
    *reinterpret_cast<float2*>(p.stats + ((b * p.N + blockIdx.y * 64 + tid) * (long long)p.stats_R + rblk) * 2) = make_float2(a, q);
  }
}

void launch_splitk_reduce_impl(const ConvParams& p, hipStream_t s) {
  LDIFF_CHECK(p.splitk >= 2 && p.splitk <= SPLITK_MAX && p.splitk_ws && (p.N & 3) == 0, LDIFF_ERR_INVALID, "split-K reduce: %d splits (at most %d)", p.splitk, SPLITK_MAX);
  if (p.stats) {
    const int hw = p.Hout * p.Wout;
    LDIFF_CHECK(!p.out_f32 && hw % 32 == 0 && p.stats_R == hw / 32 && p.M % 32 == 0, LDIFF_ERR_INVALID, "split-K reduce: fused statistics need 32 | H W and R = H W / 32 (R = %d)", p.stats_R);
    const dim3 grid((unsigned)(p.M / 32), (unsigned)((p.N + 63) / 64));
    if (p.splitk <= 8) hipLaunchKernelGGL(splitk_reduce_stats_kernel<8>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(splitk_reduce_stats_kernel<16>, grid, dim3(256), 0, s, p);
  } else {
    const long long n = (long long)p.M * (p.N >> 2);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (p.splitk <= 8) hipLaunchKernelGGL(splitk_reduce_kernel<8>, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(splitk_reduce_kernel<16>, grid, dim3(256), 0, s, p);
  }
  HIP_CHECK(hipGetLastError());
}

// Tile order inside an XCD's run of workgroups.  Default: the n-tiles of one pixel tile are neighbours (they share the halo).  On the small
// maps of the UNet's lower levels the halo is a few hundred KB and the WEIGHTS are what every pixel tile re-reads (1280 -> 1280 at 8 x 8:
// 29.5 MB against 164 KB of input per image; with the n-tiles of one image on an XCD every XCD streamed the whole matrix: 7-12x its size
// in HBM traffic, 50 us per launch = 4.7 TB/s).  There the pixel tiles of one n-tile are neighbours: its 128 x K weight slice (<= 2.9 MB)
// is fetched once per XCD and hit in that XCD's L2 (4 MB) by the other images.  LDIFF_CONV3X3_IMGFAST: -1 (default) automatic, 0 off, n > 0 =
// at most n pixel tiles per image.
static bool conv3x3_img_fast(const ConvParams& p, int tiles_per_image, int ntn, int bn) {
  static const int mode = [] { const char* e = getenv("LDIFF_CONV3X3_IMGFAST"); return e ? atoi(e) : -1; }();
  if (mode == 0 || ntn <= 1 || p.B <= 1) return false;
  if (mode > 0) return tiles_per_image <= mode;
  (void)bn;
  return tiles_per_image <= 2;   // 8 x 8 and 16 x 16 maps (measured, same box: 1280 -> 1280 at 8 x 8 45.6 -> 39.5 us, the 2560-channel concat
                                 // layers 77.4 -> 58.1 us; 16 x 16: -2 ... -8 %; 32 x 32 and larger: no change)
}

template <int TH, int TW, int BN, bool GN>
void launch_c3(const ConvParams& p, hipStream_t s) {
  constexpr int HP = (TH + 2) * (TW + 2);
  const size_t smem = (size_t)(2 * HP * 8 + 2 * BN * 8) * 16;
  auto kern = conv3x3_kernel<TH, TW, BN, GN>;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)smem);
  const bool par = p.w_par != nullptr;
  const int Ht = par ? p.Hin : p.Hout, Wt = par ? p.Win : p.Wout;
  const int tiles = p.B * ((Ht + TH - 1) / TH) * ((Wt + TW - 1) / TW);
  const int ntn = (p.N + BN - 1) / BN;
  static const std::string pname = std::string("conv3x3<") + std::to_string(TH) + "x" + std::to_string(TW) + "," + std::to_string(BN) + (GN ? ",gn>" : ">");
  const double bytes = (double)p.B * p.Hin * p.Win * (p.C1 + p.C2) * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * (p.out_f32 ? 4.0 : (p.y_lo ? 4.0 : 2.0)) +
                       (p.res ? (double)p.M * p.N * (p.res_lo ? 4.0 : 2.0) : 0.0);
  // flops = MFMA work actually executed: parity mode (nearest-2x folded into 4 taps) runs 16/36 of the 9-tap MACs
  ProfScope prof(pname.c_str(), 2.0 * p.M * (double)p.N * p.K * (par ? 16.0 / 36.0 : 1.0), bytes, s);
  const int S = p.splitk > 1 ? p.splitk : 1;
  ConvParams q = p;
  const int npar = par ? 4 : 1;
  q.tiles_m = tiles * npar; q.img_fast = conv3x3_img_fast(p, tiles / p.B, ntn, BN) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(tiles * ntn * npar, S, 1), dim3(256), smem, s, q);
  HIP_CHECK(hipGetLastError());
  if (S > 1) launch_splitk_reduce_impl(p, s);
}

template <int BN, bool GN>
void launch_c3w(const ConvParams& p, hipStream_t s) {
  constexpr int TH = 8, TW = 16, HP = 180;
  const size_t smem = (size_t)2 * HP * 128 + (BN == 160 ? 3 : 2) * BN * 128 + 1536;   // (the 160-column kernel's weight ring has three slots)
  auto kern = conv3x3w_kernel<BN, GN>;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)smem);
  const bool par = p.w_par != nullptr;
  const int Ht = par ? p.Hin : p.Hout, Wt = par ? p.Win : p.Wout;
  const int tiles = p.B * ((Ht + TH - 1) / TH) * ((Wt + TW - 1) / TW);
  const int ntn = (p.N + BN - 1) / BN;
  static const std::string pname = std::string("conv3x3<8x16,") + std::to_string(BN) + (GN ? ",gn>" : ">");
  const double bytes = (double)p.B * p.Hin * p.Win * (p.C1 + p.C2) * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * (p.out_f32 ? 4.0 : (p.y_lo ? 4.0 : 2.0)) +
                       (p.res ? (double)p.M * p.N * (p.res_lo ? 4.0 : 2.0) : 0.0);
  // flops = MFMA work actually executed: parity mode (nearest-2x folded into 4 taps) runs 16/36 of the 9-tap MACs
  ProfScope prof(pname.c_str(), 2.0 * p.M * (double)p.N * p.K * (par ? 16.0 / 36.0 : 1.0), bytes, s);
  const int S = p.splitk > 1 ? p.splitk : 1;
  const int tiles_x = (Wt + TW - 1) / TW, tiles_y = (Ht + TH - 1) / TH;
  LDIFF_CHECK((long long)tiles * (par ? 4 : 1) * ntn * std::max(std::max(ntn, tiles * (par ? 4 : 1)), std::max(tiles_x, tiles_y)) < (1LL << 32), LDIFF_ERR_INVALID, "conv3x3: %d tiles x %d channel tiles exceed the tile decode's range", tiles, ntn);
  ConvParams q = p;
  auto recip = [](int d) { return d == 1 ? 0u : (unsigned)((1ULL << 32) / (unsigned)d + 1ULL); };   // 0 encodes a divisor of 1
  q.div_ntn = recip(ntn); q.div_tx = recip(tiles_x); q.div_ty = recip(tiles_y);
  const int npar = par ? 4 : 1;
  q.tiles_m = tiles * npar; q.div_tm = recip(tiles * npar); q.img_fast = conv3x3_img_fast(p, tiles / p.B, ntn, BN) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(tiles * ntn * npar, S, 1), dim3(256), smem, s, q);
  HIP_CHECK(hipGetLastError());
  if (S > 1) launch_splitk_reduce_impl(p, s);
}
template <int BN>
void launch_c3w_gn(const ConvParams& p, hipStream_t s) {
  if (p.gn_scale) launch_c3w<BN, true>(p, s); else launch_c3w<BN, false>(p, s);
}

template <int TH, int TW, int BN>
void launch_c3_gn(const ConvParams& p, hipStream_t s) {
  if (p.gn_scale) launch_c3<TH, TW, BN, true>(p, s); else launch_c3<TH, TW, BN, false>(p, s);
}

}  // namespace

static inline int c3_tile_w(const ConvParams& p) { return (p.w_par ? p.Win : p.Wout) >= 16 ? 16 : 8; }

int conv3x3_stats_blocks(const ConvParams& p) {
  if (p.lo8_slab0) return conv3x3p_stats_blocks(p);
  if (conv3x3d_selected(p)) return conv3x3d_stats_blocks(p);
  if (conv3x3p_selected(p)) return conv3x3p_stats_blocks(p);
  const int TW = c3_tile_w(p);
  const int Ht = p.w_par ? p.Hin : p.Hout, Wt = p.w_par ? p.Win : p.Wout;
  // 8x16 tiles (wide kernel): one block per workgroup; 8x8 tiles: one per wave half
  return ((Ht + 7) / 8) * ((Wt + TW - 1) / TW) * (TW == 16 ? 1 : 2) * (p.w_par ? 4 : 1);
}

// w_par[q][n][t][c] = sum of the 3x3 taps of w[n][ky][kx][c] that read the same source pixel for output parity q=(py,px):
//   rows:  py=0: t_y=0 <- {ky=0},   t_y=1 <- {ky=1,2};   py=1: t_y=0 <- {ky=0,1}, t_y=1 <- {ky=2}   (same for columns)
__global__ void make_parity_weights_kernel(const f16* __restrict__ w, f16* __restrict__ wp, int Nrows, int Cin) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = 4LL * Nrows * 4 * Cin;
  if (i >= total) return;
  const int c = (int)(i % Cin);
  const int t = (int)((i / Cin) % 4);
  const int n = (int)((i / (4LL * Cin)) % Nrows);
  const int q = (int)(i / (4LL * Cin * Nrows));
  const int py = q >> 1, px = q & 1, ty = t >> 1, tx = t & 1;
  const int ky0 = py == 0 ? (ty == 0 ? 0 : 1) : (ty == 0 ? 0 : 2), ky1 = py == 0 ? (ty == 0 ? 0 : 2) : (ty == 0 ? 1 : 2);
  const int kx0 = px == 0 ? (tx == 0 ? 0 : 1) : (tx == 0 ? 0 : 2), kx1 = px == 0 ? (tx == 0 ? 0 : 2) : (tx == 0 ? 1 : 2);
  float acc = 0.f;
  for (int ky = ky0; ky <= ky1; ++ky)
    for (int kx = kx0; kx <= kx1; ++kx) acc += (float)w[((long long)n * 9 + ky * 3 + kx) * Cin + c];
  wp[i] = (f16)acc;
}
void launch_make_parity_weights(const f16* w, f16* w_par, int Nrows, int Cin, hipStream_t s) {
  const long long total = 4LL * Nrows * 4 * Cin;
  hipLaunchKernelGGL(make_parity_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, w_par, Nrows, Cin);
  HIP_CHECK(hipGetLastError());
}

bool conv3x3_eligible(const ConvParams& p) {
  const int Cin = p.C1 + p.C2;
  return p.ks == 3 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 && Cin % 64 == 0 && p.C1 % 64 == 0 && (long long)p.Nrows * 9 * Cin * 2 < (1LL << 32) &&
         p.Hout == (p.Hin << p.ups) && p.Wout == (p.Win << p.ups);
}

void launch_splitk_reduce(const ConvParams& p, hipStream_t s) { launch_splitk_reduce_impl(p, s); }

int conv3x3_splitk_plan(const ConvParams& p) {
  if (p.w_par) return 1;
  // few workgroups and a long K loop (UNet 8x8 / 16x16 levels, K = 9*1280..9*2560): split the slabs so the grid fills the chip
  const int TW = p.Wout >= 16 ? 16 : 8;
  const int bn = (p.N % 128 != 0 && p.N % 160 == 0) ? 160 : (p.N <= 32 ? 32 : (p.N <= 64 ? 64 : 128));
  const int wgs = p.B * ((p.Hout + 7) / 8) * ((p.Wout + TW - 1) / TW) * ((p.N + bn - 1) / bn);
  const int nslab = (p.C1 + p.C2) / 64;
  int S = 512 / (wgs > 0 ? wgs : 1);
  if (S > nslab / 4) S = nslab / 4;
  if (S > 8) S = 8;
  if (S < 1) S = 1;
  // small grids (batch 1 / 2, and the 8 x 8 level at any batch): a finer cut where the model of common.h sees it (whole slabs: a split starts at a slab)
  static const bool fine = [] { const char* e = getenv("LDIFF_SPLITK_FINE"); return !e || atoi(e) != 0; }();
  if (fine && wgs > 0 && wgs * S < 256) {
    int Sm = splitk_by_model(wgs, nslab * 9, 9, (double)p.M * p.N * 4.0, S);
    while (Sm > S && nslab / Sm < 1) --Sm;
    S = Sm;
  }
  return S >= 2 ? S : 1;
}

void launch_conv3x3(const ConvParams& p, hipStream_t s) {
  LDIFF_CHECK(p.splitk <= 1 || p.splitk_ws, LDIFF_ERR_INVALID, "conv3x3: split-K needs a workspace");   // (fused statistics of a split launch: by the reduce kernel)
  LDIFF_CHECK(!p.w_par || (p.ups == 1 && p.splitk <= 1), LDIFF_ERR_INVALID, "conv3x3: parity weights need ups=1 and no split-K");
  if (p.lo8_slab0) {   // split operand with an fp8 lo half: only the 16 x 16 ping-pong kernel reads that layout
    LDIFF_CHECK(p.splitk <= 1 && conv3x3p_selected(p), LDIFF_ERR_INVALID, "conv3x3: an fp8 lo half needs the 16 x 16 ping-pong kernel (C1=%d N=%d %dx%d)", p.C1, p.N, p.Hout, p.Wout);
    launch_conv3x3p(p, s);
    return;
  }
  if (conv3x3n_selected(p)) { LDIFF_CHECK(!p.xs, LDIFF_ERR_INVALID, "conv3x3: a folded shortcut (xs) needs the dataflow kernel"); launch_conv3x3n(p, s); return; }
  if (conv3x3d_selected(p)) { launch_conv3x3d(p, s); return; }
  // every kernel below ignores ConvParams::xs while the caller has already summed the shortcut's bias into p.bias: refuse instead of a silently wrong sum
  LDIFF_CHECK(!p.xs, LDIFF_ERR_INVALID, "conv3x3: a folded shortcut (xs) needs the dataflow kernel, which does not take this launch (split-K %d)", p.splitk);
  if (conv3x3p_selected(p)) { launch_conv3x3p(p, s); return; }
  const bool wide = c3_tile_w(p) == 16;
  const int bn = (p.N % 128 != 0 && p.N % 160 == 0) ? 160 : (p.N <= 32 ? 32 : (p.N <= 64 ? 64 : 128));
  if (wide) {
    if (bn == 160) launch_c3w_gn<160>(p, s);
    else if (bn == 64) launch_c3w_gn<64>(p, s);
    else if (bn == 32) launch_c3w_gn<32>(p, s);
    else launch_c3w_gn<128>(p, s);
  } else {
    if (bn == 160) launch_c3_gn<8, 8, 160>(p, s);
    else if (bn == 64) launch_c3_gn<8, 8, 64>(p, s);
    else if (bn == 32) launch_c3_gn<8, 8, 32>(p, s);
    else launch_c3_gn<8, 8, 128>(p, s);
  }
}

#ifdef C3W_STAMPS
extern "C" int ldiff_debug_c3w_stamps(unsigned long long* out) {   // diagnostic build only
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(c3w_dbg), sizeof(unsigned long long) * 32);
}
#endif
