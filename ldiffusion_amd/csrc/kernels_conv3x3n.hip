// 3x3 stride-1 convolution with a NARROW output (N <= 4: the VAE's conv_out 128 -> 3 at 512x512, the UNet's conv_out 320 -> 4) for
// gfx950 (MI355X).  Replaces the last conv of AutoencoderKL.decode / UNet2DConditionModel.forward (reached from
// /root/reference/pixel_latent_vector.py:78,81 and segmentor.py:103,106).
//
// Why its own kernel: in the halo-tile kernels (kernels_conv3x3.hip) every tap step streams a [BN][64] weight slice by LDS-DMA and ends
// with a workgroup barrier that drains it.  With 3 output channels a step is 8 MFMAs per wave, so the step time IS the DMA round trip
// (conv3x3<8x16,32,gn>: 348 us per VAE decode for 570 MB of traffic, 10.9 us per tile, and its 80 KB workgroups hold every CU of the chip
// meanwhile).  Here the REAL weight rows (4 of them; an MFMA tile has 16, rows 4..15 all read one zero row) of every (slab, tap) are
// loaded once per workgroup and stay in LDS: Cin/64 x 9 x 512 B = 9 KB at 128 channels.  No DMA, no per-tap barrier; the tap loop is LDS
// reads + 4 MFMAs per wave.  What is left is a streaming kernel (one HBM pass over the input, GroupNorm+SiLU of the halo image on the way),
// and a streaming kernel lives on occupancy: ONE halo buffer (23 KB + weights) and <= 168 VGPRs put three workgroups on a CU.
#include "common.h"
#include <algorithm>

namespace {

__device__ __forceinline__ int swz8(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }

// grid: workgroups that each walk tiles blockIdx.x, blockIdx.x + gridDim.x, ... (8 x 16 output pixels of one image; neighbours in the grid work
// on neighbouring tiles at the same time: their halos meet in L2); block 256 = 4 waves, wave w owns pixels [32w, 32w + 32) of the tile (two
// 16-pixel MFMA tiles) and all (<= 4) output channels.  The weights are loaded once per workgroup, and the first slab of the NEXT tile travels
// under the last slab's taps and the epilogue of the current one -- a tile no longer starts with an exposed HBM round trip.
template <bool GN>
__global__ __launch_bounds__(256, 3) void conv3x3n_kernel(const ConvParams p, const int tiles) {   // <= 168 VGPRs: three workgroups per CU (four: the GroupNorm variant spills)
  constexpr int TH = 8, TW = 16, HWD = TW + 2, HP = (TH + 2) * HWD, MT = 2;
  constexpr int A_IT = (HP * 8 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* sA = reinterpret_cast<uint4*>(smem_raw);     // [HP * 8] halo image of a 64-channel slab, chunk-swizzled (ONE buffer: LDS decides
                                                      // how many workgroups hide each other's global round trips, see below)
  uint4* sW = sA + HP * 8;                        // [nslab * 9][4 rows][8 chunks] + one zero row

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int Cin = p.C1 + p.C2, nslab = Cin >> 6;
  const int tiles_x = (p.Wout + TW - 1) / TW, tiles_y = (p.Hout + TH - 1) / TH;
  struct TileC { int b, oy0, ox0; };
  auto coords = [&](int t) -> TileC {
    TileC c;
    c.ox0 = (t % tiles_x) * TW; t /= tiles_x;
    c.oy0 = (t % tiles_y) * TH;
    c.b = t / tiles_y;
    return c;
  };

  // ---- halo staging (as in conv3x3_kernel): thread owns chunk column kc of halo pixels hp = tid/8 + 32*i ----
  const int kc = tid & 7;
  long long a_off[A_IT];   // of the tile whose slab is in registers / being loaded
  auto set_offsets = [&](const TileC& tc) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int hp = (tid >> 3) + i * 32;
      a_off[i] = -1;
      if (hp < HP) {
        const int hy = hp / HWD, hx = hp - hy * HWD;
        const int iy = tc.oy0 + hy - 1, ix = tc.ox0 + hx - 1;
        if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) a_off[i] = ((long long)tc.b * p.Hin + iy) * p.Win + ix;
      }
    }
  };
  uint4 ra[A_IT];
  float4 gs0, gs1, gt0, gt1;
  auto load_halo = [&](int c, int b) {
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
  auto store_halo = [&]() {
    const float sv[8] = {gs0.x, gs0.y, gs0.z, gs0.w, gs1.x, gs1.y, gs1.z, gs1.w};
    const float tv[8] = {gt0.x, gt0.y, gt0.z, gt0.w, gt1.x, gt1.y, gt1.z, gt1.w};
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int hp = (tid >> 3) + i * 32;
      if (hp >= HP) continue;
      uint4 v = ra[i];
      if (GN && a_off[i] >= 0) {   // zero padding applies to the normalised tensor: padding chunks stay exactly 0
        f16x8 h = __builtin_bit_cast(f16x8, v), o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float f = (float)h[j] * sv[j] + tv[j];
          const float fs = silu_f(f);
          f = silu ? fs : f;
          o[j] = (f16)f;
        }
        v = __builtin_bit_cast(uint4, o);
      }
      sA[hp * 8 + swz8(hp, kc)] = v;
    }
  };

  int t = blockIdx.x;
  TileC cur = coords(t);
  set_offsets(cur);
  load_halo(0, cur.b);
  // ---- weights: rows 0..3 of every (slab, tap) slice, [step = c*9 + tap][row][chunk]; weight rows are K-major, k = tap*Cin + channel ----
  const int nchunks = nslab * 9 * 4 * 8;   // 16-byte chunks
  for (int q = tid; q < nchunks; q += 256) {
    const int ch = q & 7, row = (q >> 3) & 3, st = q >> 5, c = st / 9, tap = st - c * 9;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < p.Nrows) v = *reinterpret_cast<const uint4*>(p.w + (long long)row * 9 * Cin + (long long)tap * Cin + c * 64 + ch * 8);
    sW[q] = v;
  }
  if (tid < 8) sW[nchunks + tid] = make_uint4(0, 0, 0, 0);   // the zero row every lane with l15 >= 4 reads
  store_halo();
  __syncthreads();

  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int hp0[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int ml = wave * 32 + m * 16 + l15;
    hp0[m] = (ml / TW) * HWD + (ml % TW);
  }
  const int wrow = l15 < 4 ? l15 * 8 : -1;   // chunk offset of this lane's weight row inside a slice, or the zero row
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias);

  for (int c = 0;;) {
    const uint4* cA = sA;
    // the next slab into registers, under this slab's taps: this tile's, or slab 0 of the workgroup's next tile
    const bool tile_done = c + 1 == nslab;
    const int tn = tile_done ? t + (int)gridDim.x : t;
    const bool has_next = tn < tiles;
    TileC nxt = cur;
    if (has_next) {
      if (tile_done) { nxt = coords(tn); set_offsets(nxt); }
      load_halo(tile_done ? 0 : c + 1, nxt.b);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3, hoff = ky * HWD + kx;
      const uint4* cW = sW + (c * 9 + tap) * 32;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const f16x8 wf = __builtin_bit_cast(f16x8, wrow >= 0 ? cW[wrow + kk * 4 + g] : sW[nchunks + kk * 4 + g]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int hp = hp0[m] + hoff;
          const f16x8 xf = __builtin_bit_cast(f16x8, cA[hp * 8 + swz8(hp, kk * 4 + g)]);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf, acc[m], 0, 0, 0);
        }
      }
    }
    if (tile_done) {
      // ---- epilogue: lane (g, l15) holds y[pixel l15 of m-tile][n = 4g + r]; only g == 0 carries real channels ----
      if (g == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const int ml = wave * 32 + m * 16 + l15;
          const int oy = cur.oy0 + ml / TW, ox = cur.ox0 + ml % TW;
          if (oy >= p.Hout || ox >= p.Wout) continue;
          const long long row = ((long long)cur.b * p.Hout + oy) * p.Wout + ox;
          const f32x4 v = acc[m] + (f32x4){bb.x, bb.y, bb.z, bb.w};
          if (!p.post_only) {
            if (p.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + row * p.ldy) = v;
            else *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + row * p.ldy) = cvt4(v);
          }
          if (p.post_img || p.post_rgb || p.post_luma) {   // the decode_latents tail on the fp32 sums, operation for operation as decode_post_kernel
            unsigned q[3];
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
              float u = __fadd_rn(__fmul_rn(v[ch], 0.5f), 0.5f);
              u = fminf(fmaxf(u, 0.f), 1.f);
              if (v[ch] != v[ch]) u = v[ch];   // clamp propagates NaN in torch
              if (p.post_img) p.post_img[row * 3 + ch] = u;
              q[ch] = (unsigned)(int)rintf(__fmul_rn(u, 255.0f));
            }
            if (p.post_rgb) { p.post_rgb[row * 3 + 0] = (uint8_t)q[0]; p.post_rgb[row * 3 + 1] = (uint8_t)q[1]; p.post_rgb[row * 3 + 2] = (uint8_t)q[2]; }
            if (p.post_luma) {
              const long long hw = (long long)p.Hout * p.Wout;
              p.post_luma[((long long)cur.b * p.post_slots + p.post_slot) * hw + (long long)oy * p.Wout + ox] =
                  (uint8_t)((19595u * q[0] + 38470u * q[1] + 7471u * q[2] + 0x8000u) >> 16);
            }
          }
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (!has_next) break;
    __syncthreads();   // every wave is done with the slab's image
    store_halo();
    __syncthreads();
    c = tile_done ? 0 : c + 1;
    t = tn;
    cur = nxt;
  }
}


// ---- tap-folded form for <= 3 output channels over <= 128 input channels (the VAE's conv_out) ----
// The kernel above spends its time on operand reads: every MFMA needs a 1-KiB pixel fragment from LDS and keeps 3 of its 16 output columns.
// Here the nine taps become output COLUMNS: P[pixel][tap * 3 + n] = sum_k X[pixel][k] W[n][tap][k] is one narrow GEMM over the HALO pixels
// (27 of 32 columns used, every pixel fragment read once, straight from global memory into the MFMA operand registers -- no LDS image, no
// barrier before the matrix work), and y[o][n] = bias[n] + sum_tap P[o + tap][tap * 3 + n] is a gather of fp32 values through LDS.
// 4.5x fewer MFMAs and ~5x less LDS traffic per tile; what is left is the GroupNorm + SiLU arithmetic on the way in and the HBM pass itself.
//   * a wave owns halo pixels 16 f + l15, f = w, w + 4, w + 8 (12 fragments cover the 180 halo pixels of an 8 x 16 tile); lane (l15, g) loads
//     channels 32 s + 8 g .. + 8 of its pixel for the NS = Cin / 32 k-steps -- the B operand of v_mfma_f32_16x16x32_f16 as it lies in memory;
//   * the weights (A operand: row j = tap * 3 + n) stay in registers for the workgroup's whole run of tiles (2 NS fragments);
//   * the next tile's fragments are requested as soon as the current tile's have been consumed (one fragment at a time);
//   * P goes to LDS as [pixel][32 floats] with the 16-byte chunks rotated by the pixel index, two buffers: one barrier per tile.
constexpr int NT_HP = 180, NT_HWD = 18;
template <bool GN, int NS>
__global__ __launch_bounds__(256, 2) void conv3x3nt_kernel(const ConvParams p, const int tiles) {
  constexpr int TH = 8, TW = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sP = reinterpret_cast<float*>(smem_raw);   // [2][192][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int Cin = p.C1, ld1 = p.ld1 ? p.ld1 : p.C1;
  const int tiles_x = (p.Wout + TW - 1) / TW, tiles_y = (p.Hout + TH - 1) / TH;
  struct TileC { int b, oy0, ox0; };
  auto coords = [&](int t) -> TileC {
    TileC c;
    c.ox0 = (t % tiles_x) * TW; t /= tiles_x;
    c.oy0 = (t % tiles_y) * TH;
    c.b = t / tiles_y;
    return c;
  };

  // weights: A fragment (n-tile nt, k-step s): row j = 16 nt + l15 = tap * 3 + n, k = 32 s + 8 g .. + 8 of that tap; rows >= 27 are zero
  f16x8 Wf[NS][2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const int j = nt * 16 + l15, tap = j / 3, n = j - tap * 3;
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (j < 27) v = *reinterpret_cast<const uint4*>(p.w + (long long)n * 9 * Cin + (long long)tap * Cin + s_ * 32 + g * 8);
      Wf[s_][nt] = __builtin_bit_cast(f16x8, v);
    }
  }
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias);

  // this lane's halo pixels (fragment slot i -> f = wave + 4 i): position inside the halo, and per tile its address / validity
  int hy_[3], hx_[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int hp = (wave + 4 * i) * 16 + l15;
    hy_[i] = hp / NT_HWD; hx_[i] = hp - hy_[i] * NT_HWD;
  }
  uint4 xr[3][NS];
  bool inb[3];
  auto request = [&](int i, const TileC& tc) {   // the NS chunks of fragment slot i of tile tc
    const int hp = (wave + 4 * i) * 16 + l15;
    const int iy = tc.oy0 + hy_[i] - 1, ix = tc.ox0 + hx_[i] - 1;
    inb[i] = hp < NT_HP && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
    const f16* src = p.x + (((long long)tc.b * p.Hin + (inb[i] ? iy : 0)) * p.Win + (inb[i] ? ix : 0)) * ld1 + g * 8;
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) xr[i][s_] = inb[i] ? *reinterpret_cast<const uint4*>(src + s_ * 32) : make_uint4(0, 0, 0, 0);
  };
  // GroupNorm scale / shift of this lane's channels (32 s + 8 g .. + 8) of image b
  float gs[NS][8], gt[NS][8];
  int gb = -1;
  auto load_affine = [&](int b) {
    if (!GN || b == gb) return;
    gb = b;
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
      const float4 a0 = *reinterpret_cast<const float4*>(p.gn_scale + (long long)b * Cin + s_ * 32 + g * 8), a1 = *reinterpret_cast<const float4*>(p.gn_scale + (long long)b * Cin + s_ * 32 + g * 8 + 4);
      const float4 c0 = *reinterpret_cast<const float4*>(p.gn_shift + (long long)b * Cin + s_ * 32 + g * 8), c1 = *reinterpret_cast<const float4*>(p.gn_shift + (long long)b * Cin + s_ * 32 + g * 8 + 4);
      gs[s_][0] = a0.x; gs[s_][1] = a0.y; gs[s_][2] = a0.z; gs[s_][3] = a0.w; gs[s_][4] = a1.x; gs[s_][5] = a1.y; gs[s_][6] = a1.z; gs[s_][7] = a1.w;
      gt[s_][0] = c0.x; gt[s_][1] = c0.y; gt[s_][2] = c0.z; gt[s_][3] = c0.w; gt[s_][4] = c1.x; gt[s_][5] = c1.y; gt[s_][6] = c1.z; gt[s_][7] = c1.w;
    }
  };
  const bool silu = p.silu_in != 0;
  auto operand = [&](int i, int s_) -> f16x8 {   // normalised fragment; zero padding applies to the NORMALISED tensor
    uint4 v = xr[i][s_];
    if (GN) {
      unsigned oa, ob, oc, od;
      if (silu) {
        gn_quad<true>(v.x, v.y, gs[s_][0], gt[s_][0], gs[s_][1], gt[s_][1], gs[s_][2], gt[s_][2], gs[s_][3], gt[s_][3], oa, ob);
        gn_quad<true>(v.z, v.w, gs[s_][4], gt[s_][4], gs[s_][5], gt[s_][5], gs[s_][6], gt[s_][6], gs[s_][7], gt[s_][7], oc, od);
      } else {
        gn_quad<false>(v.x, v.y, gs[s_][0], gt[s_][0], gs[s_][1], gt[s_][1], gs[s_][2], gt[s_][2], gs[s_][3], gt[s_][3], oa, ob);
        gn_quad<false>(v.z, v.w, gs[s_][4], gt[s_][4], gs[s_][5], gt[s_][5], gs[s_][6], gt[s_][6], gs[s_][7], gt[s_][7], oc, od);
      }
      const unsigned keep = inb[i] ? 0xffffffffu : 0u;
      v = make_uint4(oa & keep, ob & keep, oc & keep, od & keep);
    }
    return __builtin_bit_cast(f16x8, v);
  };

  // output side: thread (o = tid >> 1, h = tid & 1) sums taps 0..4 (h = 0) or 5..8 (h = 1) of output pixel o; the pair meets by one lane exchange
  const int o = tid >> 1, h = tid & 1, oyl = o >> 4, oxl = o & 15;

  int t = blockIdx.x;
  TileC cur = coords(t);
  load_affine(cur.b);
#pragma unroll
  for (int i = 0; i < 3; ++i) request(i, cur);
  for (int it = 0;; ++it) {
    const int tn = t + (int)gridDim.x;
    const bool has_next = tn < tiles;
    TileC nxt = cur;
    if (has_next) nxt = coords(tn);
    float* P = sP + (it & 1) * (192 * 32);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int s_ = 0; s_ < NS; ++s_) {
        const f16x8 xf = operand(i, s_);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[s_][0], xf, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[s_][1], xf, acc1, 0, 0, 0);
      }
      if (has_next) request(i, nxt);   // (affine of the next image: loaded below, before its first use)
      // lane (pixel l15, g) holds columns 16 nt + 4 g .. + 4 of its pixel: chunk c = 4 nt + g, stored at chunk (c + pixel) & 7
      const int hp = (wave + 4 * i) * 16 + l15;
      *reinterpret_cast<f32x4*>(P + hp * 32 + ((g + hp) & 7) * 4) = acc0;
      *reinterpret_cast<f32x4*>(P + hp * 32 + ((4 + g + hp) & 7) * 4) = acc1;
    }
    __syncthreads();
    {
      float sum[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 5; ++q) {
        const int tap = h * 5 + q;
        if (tap < 9) {
          const int ky = tap / 3, kx = tap - ky * 3;
          const int hp = (oyl + ky) * NT_HWD + oxl + kx;
#pragma unroll
          for (int n = 0; n < 3; ++n) {
            const int j = tap * 3 + n;
            sum[n] += P[hp * 32 + (((j >> 2) + hp) & 7) * 4 + (j & 3)];
          }
        }
      }
#pragma unroll
      for (int n = 0; n < 3; ++n) sum[n] += __shfl_xor(sum[n], 1);
      const int oy = cur.oy0 + oyl, ox = cur.ox0 + oxl;
      if (h == 0 && oy < p.Hout && ox < p.Wout) {
        const long long row = ((long long)cur.b * p.Hout + oy) * p.Wout + ox;
        const f32x4 v = (f32x4){sum[0] + bb.x, sum[1] + bb.y, sum[2] + bb.z, bb.w};
        if (!p.post_only) {
          if (p.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + row * p.ldy) = v;
          else *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + row * p.ldy) = cvt4(v);
        }
        if (p.post_img || p.post_rgb || p.post_luma) {   // the decode_latents tail on the fp32 sums, operation for operation as decode_post_kernel
          unsigned q[3];
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) {
            float u = __fadd_rn(__fmul_rn(v[ch], 0.5f), 0.5f);
            u = fminf(fmaxf(u, 0.f), 1.f);
            if (v[ch] != v[ch]) u = v[ch];   // clamp propagates NaN in torch
            if (p.post_img) p.post_img[row * 3 + ch] = u;
            q[ch] = (unsigned)(int)rintf(__fmul_rn(u, 255.0f));
          }
          if (p.post_rgb) { p.post_rgb[row * 3 + 0] = (uint8_t)q[0]; p.post_rgb[row * 3 + 1] = (uint8_t)q[1]; p.post_rgb[row * 3 + 2] = (uint8_t)q[2]; }
          if (p.post_luma) {
            const long long hw = (long long)p.Hout * p.Wout;
            p.post_luma[((long long)cur.b * p.post_slots + p.post_slot) * hw + (long long)oy * p.Wout + ox] =
                (uint8_t)((19595u * q[0] + 38470u * q[1] + 7471u * q[2] + 0x8000u) >> 16);
          }
        }
      }
    }
    if (!has_next) break;
    t = tn;
    cur = nxt;
    load_affine(cur.b);
  }
}

}  // namespace

// plain epilogue only (bias; fp32 or fp16 output of 4 stored columns), one or two sources, <= 512 input channels (weights resident in LDS)
bool conv3x3n_selected(const ConvParams& p) {
  static const bool off = [] { const char* e = getenv("LDIFF_CONV3X3_NARROW"); return e && atoi(e) == 0; }();   // =0: A/B timing and tests
  const int Cin = p.C1 + p.C2;
  return !off && p.N == 4 && p.Nrows >= 4 && p.ups == 0 && !p.w_par && !p.res && !p.temb && !p.stats && p.splitk <= 1 && !p.y_lo && !p.geglu &&
         p.w_bstride == 0 && Cin <= 512 && p.ldy % 4 == 0 && p.Wout >= 16 && p.Hout >= 8;
}

// the tap-folded form: at most three real output channels (27 of the 32 columns), one source of 128 channels (weights in registers), no zero
// row needed behind the real ones.  LDIFF_CONV3X3_NARROW_FOLD=0: the LDS-image kernel (A/B timing, tests)
static bool conv3x3nt_selected(const ConvParams& p) {
  static const bool off = [] { const char* e = getenv("LDIFF_CONV3X3_NARROW_FOLD"); return e && atoi(e) == 0; }();
  return !off && p.n_real > 0 && p.n_real <= 3 && p.C2 == 0 && p.C1 == 128 && ((p.ld1 ? p.ld1 : p.C1) & 7) == 0;
}

void launch_conv3x3n(const ConvParams& p, hipStream_t s) {
  if (conv3x3nt_selected(p)) {
    const int tiles = p.B * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16);
    static const int run = [] { const char* e = getenv("LDIFF_C3N_RUN"); const int v = e ? atoi(e) : 16; return v > 0 ? v : 16; }();   // tiles per workgroup (same box: 242 / 226 / 222 / 215 us at 2 / 4 / 8 / 16)
    const int grid = tiles <= 512 ? tiles : std::max(512, (tiles + run - 1) / run);
    const bool gn = p.gn_scale != nullptr;
    const size_t smem = (size_t)2 * 192 * 32 * sizeof(float);
    const void* kern = gn ? reinterpret_cast<const void*>(conv3x3nt_kernel<true, 4>) : reinterpret_cast<const void*>(conv3x3nt_kernel<false, 4>);
    ensure_dyn_smem(kern, (int)smem);
    const double bytes = (double)p.B * p.Hin * p.Win * p.C1 * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * (p.out_f32 ? 4.0 : 2.0);
    ProfScope prof(gn ? "conv3x3<8x16,n3fold,gn>" : "conv3x3<8x16,n3fold>", 2.0 * p.M * (double)p.N * p.K, bytes, s);
    if (gn) hipLaunchKernelGGL((conv3x3nt_kernel<true, 4>), dim3(grid), dim3(256), smem, s, p, tiles);
    else hipLaunchKernelGGL((conv3x3nt_kernel<false, 4>), dim3(grid), dim3(256), smem, s, p, tiles);
    HIP_CHECK(hipGetLastError());
    return;
  }
  const int Cin = p.C1 + p.C2, nslab = Cin / 64;
  const size_t smem = (size_t)180 * 128 + (size_t)nslab * 9 * 512 + 128;
  const int tiles = p.B * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16);
  // tiles per workgroup: LDIFF_C3N_RUN (default 8; 1 = one tile per workgroup, the round-4 launch shape)
  static const int run = [] { const char* e = getenv("LDIFF_C3N_RUN"); const int v = e ? atoi(e) : 8; return v > 0 ? v : 8; }();
  const int grid = tiles <= 768 ? tiles : std::max(768, (tiles + run - 1) / run);
  const bool gn = p.gn_scale != nullptr;
  const void* kern = gn ? reinterpret_cast<const void*>(conv3x3n_kernel<true>) : reinterpret_cast<const void*>(conv3x3n_kernel<false>);
  ensure_dyn_smem(kern, (int)smem);
  const double bytes = (double)p.B * p.Hin * p.Win * Cin * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * (p.out_f32 ? 4.0 : 2.0);
  ProfScope prof(gn ? "conv3x3<8x16,n4,gn>" : "conv3x3<8x16,n4>", 2.0 * p.M * (double)p.N * p.K, bytes, s);
  if (gn) hipLaunchKernelGGL(conv3x3n_kernel<true>, dim3(grid), dim3(256), smem, s, p, tiles);
  else hipLaunchKernelGGL(conv3x3n_kernel<false>, dim3(grid), dim3(256), smem, s, p, tiles);
  HIP_CHECK(hipGetLastError());
}
