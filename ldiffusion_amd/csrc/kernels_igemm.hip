// Implicit-GEMM convolution / linear for gfx950 (MI355X): fp16 operands, fp32 accumulate on MFMA 16x16x32.
//
// One kernel family covers every contraction of the sampling path except attention:
//   conv3x3 (stride 1/2, symmetric or VAE-asymmetric padding, optional nearest-2x upsample gather,
//   optional two-source channel concat), conv1x1 and nn.Linear (ks=1), with
//   - GroupNorm-apply (+SiLU) folded into the A-tile load (statistics come from kernels_norm.hip),
//   - bias / per-(batch,channel) time-embedding / residual folded into the epilogue.
// Replaces the ATen conv2d / linear calls made inside diffusers' ResnetBlock2D, Transformer2DModel,
// Downsample2D, Upsample2D (reached from /root/reference/segmentor.py:103,526 and pixel_latent_vector.py:78,81).
//
// Layout: activations NHWC fp16 (channels % 8 == 0), weights [N][ky][kx][Cin] fp16 (K-major rows).
// Tiling: BM x BN x 64 per 256-thread workgroup (4 waves as 2x2), double-buffered LDS, register-staged
// loads issued before the MFMA phase and written to LDS after it (one barrier per K-step).
// The MFMA computes C^T (A-operand = weight rows, B-operand = activation rows) so that each lane ends up
// holding 4 consecutive output channels of one pixel -> 8-byte epilogue loads/stores.
// LDS rows are 128 B; 16-byte chunks are XOR-swizzled with (row>>1)&7, which makes every ds_read_b128
// lane group of the 16x16x32 operand fetch conflict-free (bank math in DESIGN.md).
#include "common.h"

#define BK 64
#define CPR (BK / 8)  // 16-byte chunks per tile row

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// x * sigmoid(x) with the two hardware transcendentals only (v_exp_f32, v_rcp_f32: 1 ulp); __frcp_rn would expand to a
// 11-instruction IEEE division per element
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }

__device__ __forceinline__ uint4 gn_apply8(uint4 raw, const float* __restrict__ sc, const float* __restrict__ sh, int silu) {
  f16x8 h = __builtin_bit_cast(f16x8, raw);
  float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 4);
  float4 t0 = *reinterpret_cast<const float4*>(sh), t1 = *reinterpret_cast<const float4*>(sh + 4);
  float sv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
  float tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = (float)h[j] * sv[j] + tv[j];
    const float vs = silu_f(v);
    v = silu ? vs : v;   // select, not a branch per element
    o[j] = (f16)v;
  }
  return __builtin_bit_cast(uint4, o);
}

template <int BM, int BN, bool FAST, bool GN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const ConvParams p) {   // 2 waves/SIMD: accumulators stay in VGPRs
  constexpr int MT = BM / 32, NT = BN / 32;       // 16x16 tiles per wave along m / n (wave tile = BM/2 x BN/2)
  constexpr int A_IT = BM * CPR / 256, B_IT = BN * CPR / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* sA = reinterpret_cast<uint4*>(smem_raw);              // [2][BM*CPR]
  uint4* sB = sA + 2 * BM * CPR;                               // [2][BN*CPR]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int Cin = p.C1 + p.C2;
  const int ntn = (p.N + BN - 1) / BN;

  // XCD-aware tile order: workgroups that share an XCD (blockIdx % 8) walk consecutive tiles, so the
  // n-tiles of one m-tile (same activation rows, all 9 taps) hit the same 4 MiB L2.
  int nwg = gridDim.x, id = blockIdx.x;
  int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7, idx = id >> 3;
  int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int m0 = (sw / ntn) * BM, n0 = (sw % ntn) * BN;

  // ---- per-thread A rows (fixed over the K loop) ----
  const int kc = tid & (CPR - 1);
  int rb[A_IT], ry[A_IT], rx[A_IT];
  const int HWo = p.Hout * p.Wout;
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    int m = m0 + (tid >> 3) + i * 32;
    if (m < p.M) {
      int b = m / HWo, rem = m - b * HWo;
      int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      rb[i] = b; ry[i] = oy * p.stride - p.pad_t; rx[i] = ox * p.stride - p.pad_l;
    } else { rb[i] = -1; ry[i] = 0; rx[i] = 0; }
  }
  const int He = p.Hin << p.ups, We = p.Win << p.ups;

  uint4 ra[A_IT], rw[B_IT];
  int gidx[A_IT];  // b*Cin + c of the chunk (GN path), -1 when the chunk is padding

  auto load_tiles = [&](int kt) {
    const int kbase = kt * BK;
    int tap_u = 0, cb_u = 0;
    if (FAST) { tap_u = kbase / Cin; cb_u = kbase - tap_u * Cin; }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int tap, c;
      bool kval = true;
      if (FAST) { tap = tap_u; c = cb_u + kc * 8; }
      else {
        int k0 = kbase + kc * 8;
        kval = k0 < p.K;
        tap = k0 / Cin; c = k0 - tap * Cin;
      }
      int ky = tap / p.ks, kx = tap - ky * p.ks;
      int iy = ry[i] + ky, ix = rx[i] + kx;
      bool inb = kval && rb[i] >= 0 && iy >= 0 && iy < He && ix >= 0 && ix < We;
      uint4 v = make_uint4(0, 0, 0, 0);
      gidx[i] = -1;
      if (inb) {
        const f16* src; int cs, Cs;
        if (c < p.C1) { src = p.x; cs = c; Cs = p.ld1 ? p.ld1 : p.C1; } else { src = p.x2; cs = c - p.C1; Cs = p.ld2 ? p.ld2 : p.C2; }
        long long pix = ((long long)rb[i] * p.Hin + (iy >> p.ups)) * p.Win + (ix >> p.ups);
        v = *reinterpret_cast<const uint4*>(src + pix * Cs + cs);
        gidx[i] = rb[i] * Cin + c;
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int n = n0 + (tid >> 3) + i * 32;
      int k0 = kbase + kc * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (n < p.Nrows && k0 < p.K) v = *reinterpret_cast<const uint4*>(p.w + (long long)n * p.K + k0);
      rw[i] = v;
    }
  };

  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      int row = (tid >> 3) + i * 32;
      uint4 v = ra[i];
      if (GN) { if (gidx[i] >= 0) v = gn_apply8(v, p.gn_scale + gidx[i], p.gn_shift + gidx[i], p.silu_in); }
      sA[buf * BM * CPR + row * CPR + swz(row, kc)] = v;
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      int row = (tid >> 3) + i * 32;
      sB[buf * BN * CPR + row * CPR + swz(row, kc)] = rw[i];
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // split-K (few tiles, long K: the stride-2 convs of the deep UNet levels): blockIdx.y owns a contiguous range of K-steps and writes
  // raw fp32 partials; splitk_reduce_kernel (kernels_conv3x3.hip) sums them and applies the epilogue
  const int nk_all = (p.K + BK - 1) / BK, S = p.splitk > 1 ? p.splitk : 1, ksplit = blockIdx.y;
  const int kt0 = S > 1 ? ksplit * nk_all / S : 0, nk = S > 1 ? (ksplit + 1) * nk_all / S : nk_all;
  const int g = lane >> 4, l15 = lane & 15;

  load_tiles(kt0);
  store_tiles(kt0 & 1);
  __syncthreads();

  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tiles(kt + 1);
    const uint4* cA = sA + cur * BM * CPR;
    const uint4* cB = sB + cur * BN * CPR;
#pragma unroll
    for (int kk = 0; kk < BK / 32; ++kk) {
      f16x8 wf[NT], xf[MT];
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        int row = wave_n * (BN / 2) + a * 16 + l15;
        wf[a] = __builtin_bit_cast(f16x8, cB[row * CPR + swz(row, kk * 4 + g)]);
      }
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        int row = wave_m * (BM / 2) + b * 16 + l15;
        xf[b] = __builtin_bit_cast(f16x8, cA[row * CPR + swz(row, kk * 4 + g)]);
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[b], acc[a][b], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds y[m = col][n = 4g + r], r = 0..3 ----
  // Loads (bias, time embedding, residual) are issued back to back before any use: a load -> wait -> store chain per
  // 16x16 tile costs one memory round trip per tile.
  const int ncol = n0 + wave_n * (BN / 2) + g * 4;
  if (S > 1) {
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      const int m = m0 + wave_m * (BM / 2) + b * 16 + l15;
      if (m >= p.M) continue;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int n = ncol + a * 16;
        if (n < p.N) *reinterpret_cast<f32x4*>(p.splitk_ws + ((long long)ksplit * p.M + m) * p.N + n) = acc[a][b];
      }
    }
    return;
  }
  f32x4 bb[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && ncol + a * 16 < p.N) t = *reinterpret_cast<const float4*>(p.bias + ncol + a * 16);
    bb[a] = (f32x4){t.x, t.y, t.z, t.w};
  }
  int mrow[MT];
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int m = m0 + wave_m * (BM / 2) + b * 16 + l15;
    mrow[b] = m < p.M ? m : -1;
  }
  f16x4 rr[MT][NT], rl[MT][NT];
  if (p.res) {
#pragma unroll
    for (int b = 0; b < MT; ++b)
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        rr[b][a] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        rl[b][a] = rr[b][a];
        if (mrow[b] >= 0 && ncol + a * 16 < p.N) {
          rr[b][a] = *reinterpret_cast<const f16x4*>(p.res + (long long)mrow[b] * p.ld_res + ncol + a * 16);
          if (p.res_lo) rl[b][a] = *reinterpret_cast<const f16x4*>(p.res + (long long)mrow[b] * p.ld_res + p.res_lo + ncol + a * 16);
        }
      }
  }
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    if (mrow[b] < 0) continue;
    const long long m = mrow[b];
    f32x4 tt[NT];
    if (p.temb) {
      const int bi = mrow[b] / HWo;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ncol + a * 16 < p.N) t = *reinterpret_cast<const float4*>(p.temb + (long long)bi * p.ld_temb + ncol + a * 16);
        tt[a] = (f32x4){t.x, t.y, t.z, t.w};
      }
    }
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int n = ncol + a * 16;
      if (n >= p.N) continue;
      f32x4 v = acc[a][b] + bb[a];
      if (p.temb) v += tt[a];
      if (p.res) { v += up4(rr[b][a]); if (p.res_lo) v += up4(rl[b][a]); }
      if (p.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + m * p.ldy + n) = v;
      } else {
        const f16x4 o = cvt4(v);
        *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + m * p.ldy + n) = o;
        if (p.y_lo) *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + m * p.ldy + p.y_lo + n) = cvt4(v - up4(o));
        if (p.stats) acc[a][b] = p.y_lo ? v : up4(o);
      }
    }
  }
  if (p.stats) {   // fused GroupNorm statistics per 32-row block (common.h)
    bool ok[MT];
#pragma unroll
    for (int b = 0; b < MT; ++b) ok[b] = mrow[b] >= 0;
#pragma unroll
    for (int sb = 0; sb < MT / 2; ++sb) {
      const long long blk = (m0 + wave_m * (BM / 2)) / 32 + sb, R = p.stats_R;   // global 32-row block -> (image, block in image)
      if (blk * 32 < p.M) wave_stats_store<MT, NT>(acc, ok, 2 * sb, 2 * sb + 2, p.stats + ((blk / R) * p.N * R + blk % R) * 2, R, p.N, ncol, l15);
    }
  }
}

template <int BM, int BN, bool FAST, bool GN>
static void launch_cfg(const ConvParams& p, hipStream_t s) {
  const size_t smem = 2 * (BM + BN) * BK * sizeof(f16);
  auto kern = igemm_kernel<BM, BN, FAST, GN>;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)smem);
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  static const std::string pname = std::string("igemm<") + std::to_string(BM) + "," + std::to_string(BN) + (FAST ? ",fast" : ",gen") +
                                   (GN ? ",gn>" : ">");
  // algorithmic work: 2*M*N*K flops; each source tensor, the weights and the residual read once, the output written once
  const double esz = 2.0;
  const double in_bytes = (double)p.B * p.Hin * p.Win * (p.C1 + p.C2) * esz;
  const double bytes = in_bytes + (double)p.N * p.K * esz + (double)p.M * p.N * (p.out_f32 || p.y_lo ? 4.0 : esz) + (p.res ? (double)p.M * p.N * (p.res_lo ? 4.0 : esz) : 0.0);
  ProfScope prof(pname.c_str(), 2.0 * p.M * (double)p.N * p.K, bytes, s);
  const int S = p.splitk > 1 ? p.splitk : 1;
  LDIFF_CHECK(S == 1 || (p.splitk_ws && !p.geglu), LDIFF_ERR_INVALID, "igemm: split-K needs a workspace");   // (fused statistics of a split launch: by the reduce kernel)
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, S), dim3(256), smem, s, p);
  HIP_CHECK(hipGetLastError());
  if (S > 1) launch_splitk_reduce(p, s);
}

template <int BM, int BN>
static void launch_bmn(const ConvParams& p, bool fast, hipStream_t s) {
  const bool gn = p.gn_scale != nullptr;
  if (fast) { if (gn) launch_cfg<BM, BN, true, true>(p, s); else launch_cfg<BM, BN, true, false>(p, s); }
  else      { if (gn) launch_cfg<BM, BN, false, true>(p, s); else launch_cfg<BM, BN, false, false>(p, s); }
}

int conv3x3_stats_blocks(const ConvParams& p);

int conv_stats_blocks_per_image(const ConvParams& p) {
  if (p.out_f32) return 0;
  const int hw = p.Hout * p.Wout;
  if (p.splitk > 1) return hw % 32 == 0 ? hw / 32 : 0;   // a split launch: the reduce kernel's 32-row blocks, whichever kernel wrote the partials
  if (conv3x3_eligible(p)) return conv3x3_stats_blocks(p);
  return hw % 32 == 0 ? hw / 32 : 0;
}

// Split-K for the register-staged kernel: only where its launcher picks 64x64 tiles, the tiles leave most workgroup slots empty and the
// K loop is long (stride-2 3x3 convs of the UNet's 16x16 -> 8x8 level: 160 tiles x 180-360 K-steps)
int igemm_splitk_plan(const ConvParams& p) {
  if (p.out_f32 || p.geglu || p.M <= 0) return 1;
  if (p.stats && (p.Hout * p.Wout) % 32 != 0) return 1;   // the reduce kernel emits them in 32-row blocks
  if (conv3x3_eligible(p) || gemm_dma_eligible(p)) return 1;
  auto tiles = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  if (tiles(128, 64) >= 384) return 1;
  const int nk = (p.K + BK - 1) / BK;
  auto splits = [&](long long t) { int S = (int)(512 / t); S = S > nk / 16 ? nk / 16 : S; return S > 8 ? 8 : S; };
  // 128x64 tiles (half the operand bytes per flop of 64x64) when their splits fill the chip; launch_igemm picks the tile by tiles x S
  const int S128 = splits(tiles(128, 64));
  if (S128 >= 2 && tiles(128, 64) * S128 >= 384) return S128;
  const int S64 = splits(tiles(64, 64));
  return S64 >= 2 ? S64 : 1;
}

void launch_igemm(const ConvParams& p, hipStream_t s) {
  const int Cin = p.C1 + p.C2;
  LDIFF_CHECK(p.C1 % 8 == 0 && p.C2 % 8 == 0 && Cin > 0, LDIFF_ERR_INVALID, "igemm: channel counts must be multiples of 8 (C1=%d C2=%d)", p.C1, p.C2);
  LDIFF_CHECK(p.K == p.ks * p.ks * Cin, LDIFF_ERR_INVALID, "igemm: K=%d != ks*ks*Cin=%d", p.K, p.ks * p.ks * Cin);
  LDIFF_CHECK(p.N % 4 == 0 && p.N <= p.Nrows && p.ldy % 4 == 0 && (p.geglu ? p.N / 2 : p.N) <= p.ldy, LDIFF_ERR_INVALID, "igemm: bad N=%d Nrows=%d ldy=%d", p.N, p.Nrows, p.ldy);
  LDIFF_CHECK((p.C2 == 0) == (p.x2 == nullptr), LDIFF_ERR_INVALID, "igemm: x2/C2 mismatch");
  LDIFF_CHECK(!p.res || (p.ld_res % 4 == 0 && p.res_lo % 4 == 0), LDIFF_ERR_INVALID, "igemm: ld_res / res_lo must be multiples of 4");
  LDIFF_CHECK(p.y_lo % 4 == 0 && (p.y_lo == 0 || (!p.out_f32 && !p.geglu && p.y_lo >= p.N && p.y_lo + p.N <= p.ldy)), LDIFF_ERR_INVALID,
              "igemm: split output needs fp16 y with N <= y_lo and y_lo + N <= ldy (N=%d y_lo=%d ldy=%d)", p.N, p.y_lo, p.ldy);
  LDIFF_CHECK((p.ld1 == 0 || (p.ld1 >= p.C1 && p.ld1 % 8 == 0)) && (p.ld2 == 0 || (p.ld2 >= p.C2 && p.ld2 % 8 == 0)), LDIFF_ERR_INVALID,
              "igemm: row pitches must be multiples of 8 and >= the channel counts");
  // per-image weights (GroupNorm folded into the layer) exist only in the LDS-DMA GEMM: any other kernel would silently use image 0's
  LDIFF_CHECK(p.w_bstride == 0 || (!conv3x3_eligible(p) && gemm_dma_eligible(p)), LDIFF_ERR_INVALID, "igemm: per-image weights need the DMA GEMM path");
  LDIFF_CHECK(!p.temb || p.ld_temb % 4 == 0, LDIFF_ERR_INVALID, "igemm: ld_temb must be a multiple of 4");
  if (p.M <= 0) return;
  if (conv3x3_eligible(p)) { launch_conv3x3(p, s); return; }
  if (gemm_dma_eligible(p)) {
    if (p.w_frag && gemm_df_selected(p)) launch_gemm_df(p, s);   // the caller packed the weights for the dataflow kernel
    else launch_gemm_dma(p, s);
    return;
  }
  LDIFF_CHECK(!p.geglu, LDIFF_ERR_INVALID, "GEGLU epilogue: only on 1x1 / linear layers with K %% 64 == 0, N %% 32 == 0, fp16 output, no residual");
  const bool fast = (Cin % BK == 0) && (p.C1 % BK == 0);
  // Tile choice: largest tile that still yields >= ~2 workgroups per CU worth of tiles; narrow N gets BN=64.
  const int S = p.splitk > 1 ? p.splitk : 1;   // split-K multiplies the workgroups of a tile shape (igemm_splitk_plan)
  auto tiles = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  const bool n_small = p.N <= 64 || (p.N % 128 != 0 && p.N % 128 <= 64 && p.N < 512);
  if (!n_small && S == 1 && tiles(128, 128) >= 384) launch_bmn<128, 128>(p, fast, s);
  else if (tiles(128, 64) * S >= 384) launch_bmn<128, 64>(p, fast, s);
  else launch_bmn<64, 64>(p, fast, s);
}
