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

namespace {

__device__ __forceinline__ int swz8(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }

// grid: tiles (8 x 16 output pixels of one image); block 256 = 4 waves, wave w owns pixels [32w, 32w + 32) of the tile (two 16-pixel
// MFMA tiles) and all (<= 4) output channels.
template <bool GN>
__global__ __launch_bounds__(256, 3) void conv3x3n_kernel(const ConvParams p) {   // <= 168 VGPRs: three workgroups per CU (four: the GroupNorm variant spills)
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
  int tm = blockIdx.x;
  const int tx = tm % tiles_x; tm /= tiles_x;
  const int ty = tm % tiles_y;
  const int b = tm / tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;

  // ---- halo staging (as in conv3x3_kernel): thread owns chunk column kc of halo pixels hp = tid/8 + 32*i ----
  const int kc = tid & 7;
  long long a_off[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int hp = (tid >> 3) + i * 32;
    a_off[i] = -1;
    if (hp < HP) {
      const int hy = hp / HWD, hx = hp - hy * HWD;
      const int iy = oy0 + hy - 1, ix = ox0 + hx - 1;
      if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) a_off[i] = ((long long)b * p.Hin + iy) * p.Win + ix;
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

  load_halo(0);
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

  for (int c = 0; c < nslab; ++c) {
    const uint4* cA = sA;
    if (c + 1 < nslab) load_halo(c + 1);   // into registers, under this slab's taps
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
    if (c + 1 < nslab) {
      __syncthreads();   // every wave is done with the slab's image
      store_halo();
      __syncthreads();
    }
  }

  // ---- epilogue: lane (g, l15) holds y[pixel l15 of m-tile][n = 4g + r]; only g == 0 carries real channels ----
  if (g != 0) return;
  float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bb = *reinterpret_cast<const float4*>(p.bias);
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int ml = wave * 32 + m * 16 + l15;
    const int oy = oy0 + ml / TW, ox = ox0 + ml % TW;
    if (oy >= p.Hout || ox >= p.Wout) continue;
    const long long row = ((long long)b * p.Hout + oy) * p.Wout + ox;
    const f32x4 v = acc[m] + (f32x4){bb.x, bb.y, bb.z, bb.w};
    if (!p.post_only) {
      if (p.out_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + row * p.ldy) = v;
      else *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + row * p.ldy) = cvt4(v);
    }
    if (p.post_img || p.post_rgb || p.post_luma) {   // the decode_latents tail on the fp32 sums, operation for operation as decode_post_kernel
      unsigned q[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float u = __fadd_rn(__fmul_rn(v[c], 0.5f), 0.5f);
        u = fminf(fmaxf(u, 0.f), 1.f);
        if (v[c] != v[c]) u = v[c];   // clamp propagates NaN in torch
        if (p.post_img) p.post_img[row * 3 + c] = u;
        q[c] = (unsigned)(int)rintf(__fmul_rn(u, 255.0f));
      }
      if (p.post_rgb) { p.post_rgb[row * 3 + 0] = (uint8_t)q[0]; p.post_rgb[row * 3 + 1] = (uint8_t)q[1]; p.post_rgb[row * 3 + 2] = (uint8_t)q[2]; }
      if (p.post_luma) {
        const long long hw = (long long)p.Hout * p.Wout;
        p.post_luma[((long long)b * p.post_slots + p.post_slot) * hw + (long long)oy * p.Wout + ox] =
            (uint8_t)((19595u * q[0] + 38470u * q[1] + 7471u * q[2] + 0x8000u) >> 16);
      }
    }
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

void launch_conv3x3n(const ConvParams& p, hipStream_t s) {
  const int Cin = p.C1 + p.C2, nslab = Cin / 64;
  const size_t smem = (size_t)180 * 128 + (size_t)nslab * 9 * 512 + 128;
  const int tiles = p.B * ((p.Hout + 7) / 8) * ((p.Wout + 15) / 16);
  const bool gn = p.gn_scale != nullptr;
  const void* kern = gn ? reinterpret_cast<const void*>(conv3x3n_kernel<true>) : reinterpret_cast<const void*>(conv3x3n_kernel<false>);
  ensure_dyn_smem(kern, (int)smem);
  const double bytes = (double)p.B * p.Hin * p.Win * Cin * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * (p.out_f32 ? 4.0 : 2.0);
  ProfScope prof(gn ? "conv3x3<8x16,n4,gn>" : "conv3x3<8x16,n4>", 2.0 * p.M * (double)p.N * p.K, bytes, s);
  if (gn) hipLaunchKernelGGL(conv3x3n_kernel<true>, dim3(tiles), dim3(256), smem, s, p);
  else hipLaunchKernelGGL(conv3x3n_kernel<false>, dim3(tiles), dim3(256), smem, s, p);
  HIP_CHECK(hipGetLastError());
}
