// Plain GEMM for gfx950 (MI355X): y[M,N] = A[M,K] W[N,K]^T (+bias +residual), fp16 operands, fp32 MFMA accumulate.
// Both operands stream global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR staging, no ds_write traffic.
//
// Serves every nn.Linear of diffusers' BasicTransformerBlock (to_q/k/v fused, to_out, GEGLU proj, ff out, cross-attn q),
// the 1x1 conv_shortcut of ResnetBlock2D (two-source channel concat = two row pointers) and proj_out of
// Transformer2DModel (reached from /root/reference/segmentor.py:103,526 and pixel_latent_vector.py:78).
// Contractions that need a transform on the load path (GroupNorm-apply) or an im2col gather (3x3 stride 2, tiny Cin)
// stay on the register-staged kernel in kernels_igemm.hip.
//
// Tile BM x BN x 64, 256 threads (4 waves as 2x2), two LDS stages, one barrier per K-step.  LDS rows are 128 B with
// 16-byte chunk c of row r at position c ^ ((r>>1)&7); the DMA writes LDS linearly, so the XOR is applied to the SOURCE
// chunk each lane fetches.  Per-lane source offsets are fixed 32-bit values; per step only a uniform base moves.
#include "common.h"

namespace {

__device__ __forceinline__ int swz8(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_dma_kernel(const ConvParams p) {
  constexpr int MT = BM / 32, NT = BN / 32;
  constexpr int A_IT = BM * 8 / 256, W_IT = BN * 8 / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* sA = reinterpret_cast<uint4*>(smem_raw);   // [2][BM*8]
  uint4* sW = sA + 2 * BM * 8;                      // [2][BN*8]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int g = lane >> 4, l15 = lane & 15;
  const int ntn = (p.N + BN - 1) / BN;

  int nwg = gridDim.x, id = blockIdx.x;
  int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7, idx = id >> 3;
  int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int m0 = (sw / ntn) * BM, n0 = (sw % ntn) * BN;

  // per-lane source offsets (bytes); rows beyond M / N are clamped (their results are never stored)
  unsigned a_row[A_IT], a_sw[A_IT], w_voff[W_IT];   // A offset = row * (bytes per row of the current concat source) + swizzled chunk
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int q = tid + i * 256, r = q >> 3, pos = q & 7;
    int m = m0 + r;
    m = m < p.M ? m : p.M - 1;
    a_row[i] = (unsigned)m;
    a_sw[i] = (unsigned)(swz8(r, pos) * 16);
  }
#pragma unroll
  for (int i = 0; i < W_IT; ++i) {
    const int q = tid + i * 256, r = q >> 3, pos = q & 7;
    int n = n0 + r;
    n = n < p.Nrows ? n : p.Nrows - 1;
    w_voff[i] = (unsigned)(((long long)n * p.K + swz8(r, pos) * 8) * 2);
  }
  auto issue = [&](int kt, int buf) {
    const int kbase = kt * 64;
    const bool second = kbase >= p.C1;                                   // uniform: which concat source this slab is in
    const char* abase = second ? reinterpret_cast<const char*>(p.x2) + (long long)(kbase - p.C1) * 2
                               : reinterpret_cast<const char*>(p.x) + (long long)kbase * 2;
    const char* wbase = reinterpret_cast<const char*>(p.w) + (long long)kbase * 2;
    const unsigned row_bytes = (unsigned)(second ? p.C2 : p.C1) * 2u;   // uniform; one v_mad_u32_u24 per DMA (rows < 2^24)
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      uint4* ldst = sA + buf * BM * 8 + i * 256 + wave * 64;
      __builtin_amdgcn_global_load_lds((gptr_t*)(abase + (__umul24(a_row[i], row_bytes) + a_sw[i])), (lptr_t*)ldst, 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < W_IT; ++i) {
      uint4* ldst = sW + buf * BN * 8 + i * 256 + wave * 64;
      __builtin_amdgcn_global_load_lds((gptr_t*)(wbase + w_voff[i]), (lptr_t*)ldst, 16, 0, 0);
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[a][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int w_addr[NT], x_addr[MT];
#pragma unroll
  for (int a = 0; a < NT; ++a) { const int row = wave_n * (BN / 2) + a * 16 + l15; w_addr[a] = row * 8 + swz8(row, g); }
#pragma unroll
  for (int m = 0; m < MT; ++m) { const int row = wave_m * (BM / 2) + m * 16 + l15; x_addr[m] = row * 8 + swz8(row, g); }

  const int nk = p.K / 64;
  issue(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) issue(kt + 1, cur ^ 1);
    const uint4* cA = sA + cur * BM * 8;
    const uint4* cW = sW + cur * BN * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      f16x8 wf[NT], xf[MT];
#pragma unroll
      for (int a = 0; a < NT; ++a) wf[a] = __builtin_bit_cast(f16x8, cW[w_addr[a] ^ (kk * 4)]);
#pragma unroll
      for (int m = 0; m < MT; ++m) xf[m] = __builtin_bit_cast(f16x8, cA[x_addr[m] ^ (kk * 4)]);
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[a], xf[m], acc[a][m], 0, 0, 0);
    }
    __syncthreads();   // drains the DMA issued at the top of this step (vmcnt(0)) and protects the buffer swap
  }

  // ---- epilogue (all loads issued before any use) ----
  const int ncol = n0 + wave_n * (BN / 2) + g * 4;
  f32x4 bb[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && ncol + a * 16 < p.N) t = *reinterpret_cast<const float4*>(p.bias + ncol + a * 16);
    bb[a] = (f32x4){t.x, t.y, t.z, t.w};
  }
  int mrow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mm = m0 + wave_m * (BM / 2) + m * 16 + l15;
    mrow[m] = mm < p.M ? mm : -1;
  }
  f16x4 rr[MT][NT];
  if (p.res) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        rr[m][a] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        if (mrow[m] >= 0 && ncol + a * 16 < p.N) rr[m][a] = *reinterpret_cast<const f16x4*>(p.res + (long long)mrow[m] * p.ld_res + ncol + a * 16);
      }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (mrow[m] < 0) continue;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int n = ncol + a * 16;
      if (n >= p.N) continue;
      f32x4 v = acc[a][m] + bb[a];
      if (p.res) { v[0] += (float)rr[m][a][0]; v[1] += (float)rr[m][a][1]; v[2] += (float)rr[m][a][2]; v[3] += (float)rr[m][a][3]; }
      if (p.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + (long long)mrow[m] * p.ldy + n) = v;
      } else {
        f16x4 o = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + (long long)mrow[m] * p.ldy + n) = o;
        if (p.stats) acc[a][m] = (f32x4){(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
      }
    }
  }
  if (p.stats) {   // fused GroupNorm statistics per 32-row block (common.h)
    bool ok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ok[m] = mrow[m] >= 0;
#pragma unroll
    for (int sb = 0; sb < MT / 2; ++sb) {
      const long long blk = (m0 + wave_m * (BM / 2)) / 32 + sb, R = p.stats_R;   // global 32-row block -> (image, block in image)
      if (blk * 32 < p.M) wave_stats_store<MT, NT>(acc, ok, 2 * sb, 2 * sb + 2, p.stats + ((blk / R) * p.N * R + blk % R) * 2, R, p.N, ncol, l15);
    }
  }
}

template <int BM, int BN>
void launch_g(const ConvParams& p, hipStream_t s) {
  static bool attr_set = false;
  const size_t smem = (size_t)2 * (BM + BN) * 8 * 16;
  auto kern = gemm_dma_kernel<BM, BN>;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  static const std::string pname = std::string("gemm_dma<") + std::to_string(BM) + "," + std::to_string(BN) + ">";
  const double bytes = (double)p.M * p.K * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * (p.out_f32 ? 4.0 : 2.0) + (p.res ? (double)p.M * p.N * 2.0 : 0.0);
  ProfScope prof(pname.c_str(), 2.0 * p.M * (double)p.N * p.K, bytes, s);
  hipLaunchKernelGGL(kern, dim3(ntm * ntn), dim3(256), smem, s, p);
  HIP_CHECK(hipGetLastError());
}

}  // namespace

bool gemm_dma_eligible(const ConvParams& p) {
  return p.ks == 1 && p.stride == 1 && p.ups == 0 && p.pad_t == 0 && p.pad_l == 0 && !p.gn_scale && !p.temb && p.K % 64 == 0 && p.C1 % 64 == 0 &&
         p.Hout == p.Hin && p.Wout == p.Win && p.M < (1 << 24) && (p.C1 > p.C2 ? p.C1 : p.C2) * 2 < (1 << 24) && (long long)p.M * (p.C1 > p.C2 ? p.C1 : p.C2) * 2 < (1LL << 32) && (long long)p.Nrows * p.K * 2 < (1LL << 32);
}

void launch_gemm_dma(const ConvParams& p, hipStream_t s) {
  auto tiles = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  const bool n_small = p.N <= 64 || (p.N % 128 != 0 && p.N % 128 <= 64 && p.N < 512);
  if (!n_small && tiles(128, 128) >= 384) launch_g<128, 128>(p, s);
  else if (tiles(128, 64) >= 384) launch_g<128, 64>(p, s);
  else launch_g<64, 64>(p, s);
}
