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

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
// operand reads as inline asm + counted lgkmcnt waits (hipcc only emits lgkmcnt(0)); see kernels_conv3x3.hip
template <int OFF>
__device__ __forceinline__ void lds_read128(f16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT>
__device__ __forceinline__ void lds_wait(f16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void lds_wait(f16x8& a, f16x8& b, f16x8& c) { asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void lds_wait(f16x8& a, f16x8& b, f16x8& c, f16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(CNT));
}
template <int CNT>
__device__ __forceinline__ void lds_wait(f16x8& a, f16x8& b, f16x8& c, f16x8& d, f16x8& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(CNT));
}

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_dma_kernel(const ConvParams p) {
  constexpr int MT = BM / 32, NT = BN / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave >> 1, wave_n = wave & 1;
  const int g = lane >> 4, l15 = lane & 15;
  const int ntn = (p.N + BN - 1) / BN;

  int nwg = gridDim.x, id = blockIdx.x;
  int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7, idx = id >> 3;
  int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  // tile order inside an XCD's run: n-tiles of one m-tile are neighbours (they share the activation rows), or -- where the weights are the
  // larger operand (gemm_m_fast) -- the m-tiles of one n-tile (its weight rows stay in that XCD's L2)
  const int m0 = (p.img_fast ? sw % p.tiles_m : sw / ntn) * BM, n0 = (p.img_fast ? sw / p.tiles_m : sw % ntn) * BN;

  // ---- operand slices by `buffer_load_dwordx4 ... lds` (see the wide conv3x3 kernel): wave w owns rows [w*B/4, (w+1)*B/4)
  // of each [B][64] slice = B/32 pieces of 8 rows (1 KiB); per-lane voffsets are fixed, a step moves only the scalar soffset;
  // the pieces of a wave are contiguous in LDS, so one M0 per operand serves them through the instruction offset, which the
  // hardware adds to BOTH the memory and the LDS address (voffset pre-compensated).  Rows beyond M / N are clamped (their
  // results are never stored).
  constexpr int A_NP = BM / 32, W_NP = BN / 32;
  const int pitch1 = p.ld1 ? p.ld1 : p.C1, pitch2 = p.ld2 ? p.ld2 : p.C2;   // row pitch of the two sources (elements)
  constexpr unsigned A_BYTES = BM * 128, W_BYTES = BN * 128, W_OFF = 2 * A_BYTES;   // LDS map: A[2] | W[2]
  int a_row[A_NP], a_sw[A_NP], a_voff[A_NP], w_voff[W_NP];
#pragma unroll
  for (int i = 0; i < A_NP; ++i) {
    const int r = wave * (BM / 4) + i * 8 + (lane >> 3), pos = lane & 7;
    int m = m0 + r;
    a_row[i] = m < p.M ? m : p.M - 1;
    a_sw[i] = swz8(r, pos) * 16 - (i & 3) * 1024;
    a_voff[i] = a_row[i] * (pitch1 * 2) + a_sw[i];
  }
#pragma unroll
  for (int i = 0; i < W_NP; ++i) {
    const int r = wave * (BN / 4) + i * 8 + (lane >> 3), pos = lane & 7;
    int n = n0 + r;
    n = n < p.Nrows ? n : p.Nrows - 1;
    w_voff[i] = (int)(((long long)n * p.K + swz8(r, pos) * 8) * 2) - (i & 3) * 1024;
  }
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((long long)p.M * pitch1 * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x), 0, (int)((long long)p.M * (p.x2 ? pitch2 : pitch1) * 2), 0x00020000);
  // per-image weights (GroupNorm folded into the layer): tiles never straddle images, so the image of this tile is uniform
  const int img = p.w_bstride > 0 ? m0 / (p.Hout * p.Wout) : 0;
  const f16* wmat = p.w + (long long)img * p.w_bstride;
  const float* bvec = p.bias ? p.bias + (long long)img * p.bias_bstride : nullptr;
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wmat, 0, (int)((long long)p.Nrows * p.K * 2), 0x00020000);
  const int kt2 = p.C1 / 64;   // first K-step of the second concat source
  auto issue = [&](int kt, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const bool second = kt >= kt2;
    if (kt == kt2) {   // the row pitch changes with the source: new voffsets, once
#pragma unroll
      for (int i = 0; i < A_NP; ++i) a_voff[i] = a_row[i] * (pitch2 * 2) + a_sw[i];
    }
    unsigned char* adst = smem_raw + buf * A_BYTES + wave * (BM * 32);
    unsigned char* wdst = smem_raw + W_OFF + buf * W_BYTES + wave * (BN * 32);
    const int asoff = (second ? kt - kt2 : kt) * 128, wsoff = kt * 128;
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass rejects this builtin and then silently drops the kernel stub
    static_for<0, A_NP>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lptr_t*)(adst + (i >> 2) * 4096), 16, a_voff[i], asoff, (i & 3) * 1024, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t*)(adst + (i >> 2) * 4096), 16, a_voff[i], asoff, (i & 3) * 1024, 0);
    });
    static_for<0, W_NP>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lptr_t*)(wdst + (i >> 2) * 4096), 16, w_voff[i], wsoff, (i & 3) * 1024, 0);
    });
#endif
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[a][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // per-lane operand addresses: stage, k-half, m / a all go into the ds_read offset field or an XOR of 64
  const unsigned lds0 = (unsigned)(size_t)(lptr_t*)smem_raw;
  const int xrow = wave_m * (BM / 2) + l15, wrow = wave_n * (BN / 2) + l15;
  const unsigned xa = lds0 + (unsigned)(xrow * 128 + ((g ^ ((xrow >> 1) & 7)) << 4)), xa1 = xa ^ 64u;
  const unsigned wa = lds0 + W_OFF + (unsigned)(wrow * 128 + ((g ^ ((wrow >> 1) & 7)) << 4)), wa1 = wa ^ 64u;
  constexpr int NF = NT + MT;

  // split-K (few tiles, long K: the concat shortcut convs of the 8x8 level): blockIdx.y owns a contiguous range of K-steps and writes raw
  // fp32 partials; splitk_reduce_kernel sums them and applies the epilogue
  const int nk_all = p.K / 64, S = p.splitk > 1 ? p.splitk : 1, ksplit = blockIdx.y;
  const int kt0 = S > 1 ? ksplit * nk_all / S : 0, nk = S > 1 ? (ksplit + 1) * nk_all / S : nk_all;
  if (kt0 > kt2) {   // a split that starts INSIDE the second concat source never passes the step at which the voffsets change pitch
#pragma unroll
    for (int i = 0; i < A_NP; ++i) a_voff[i] = a_row[i] * (pitch2 * 2) + a_sw[i];
  }
  issue(kt0, std::integral_constant<int, 0>{});
  // 64 x 64 tiles (the small launches: 5-40 K steps of ~1 us): the epilogue's operands -- bias, residual hi / lo -- are fetched HERE, under the first
  // slice's round trip, instead of as one more dependent round trip behind the K loop (8 registers at this tile size; the larger tiles have none to spare)
  constexpr bool EARLY = BM * BN <= 64 * 64;
  const int ncol_e = n0 + wave_n * (BN / 2) + g * 4;
  f32x4 bb_e[EARLY ? NT : 1];
  f16x4 rr_e[EARLY ? MT : 1][EARLY ? NT : 1], rl_e[EARLY ? MT : 1][EARLY ? NT : 1];
  if constexpr (EARLY) {
    if (S == 1) {
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bvec && ncol_e + a * 16 < p.N) t = *reinterpret_cast<const float4*>(bvec + ncol_e + a * 16);
        bb_e[a] = (f32x4){t.x, t.y, t.z, t.w};
      }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int mm = m0 + wave_m * (BM / 2) + m * 16 + l15;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          rr_e[m][a] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
          rl_e[m][a] = rr_e[m][a];
          if (p.res && mm < p.M && ncol_e + a * 16 < p.N) {
            rr_e[m][a] = *reinterpret_cast<const f16x4*>(p.res + (long long)mm * p.ld_res + ncol_e + a * 16);
            if (p.res_lo) rl_e[m][a] = *reinterpret_cast<const f16x4*>(p.res + (long long)mm * p.ld_res + p.res_lo + ncol_e + a * 16);
          }
        }
      }
    }
  }
  __syncthreads();
  auto step = [&](int kt, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    f16x8 wf[2][NT], xf[2][MT];
    constexpr bool DMA_FIRST = BM * BN < 128 * 128;   // short steps (8-16 MFMA per wave): the DMA needs the whole step to land
    if (DMA_FIRST && kt + 1 < nk) issue(kt + 1, std::integral_constant<int, buf ^ 1>{});
    if (DMA_FIRST) __builtin_amdgcn_sched_barrier(0);
    static_for<0, MT>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<buf * (int)A_BYTES + m * 2048>(xf[0][m], xa); });
    static_for<0, NT>([&](auto ac) { constexpr int a = decltype(ac)::value; lds_read128<buf * (int)W_BYTES + a * 2048>(wf[0][a], wa); });
    static_for<0, MT>([&](auto mc) { constexpr int m = decltype(mc)::value; lds_read128<buf * (int)A_BYTES + m * 2048>(xf[1][m], xa1); });
    static_for<0, NT>([&](auto ac) { constexpr int a = decltype(ac)::value; lds_read128<buf * (int)W_BYTES + a * 2048>(wf[1][a], wa1); });
    __builtin_amdgcn_sched_barrier(0);
    if (!DMA_FIRST && kt + 1 < nk) issue(kt + 1, std::integral_constant<int, buf ^ 1>{});   // into the stage read last in the previous step
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 2 * NT>([&](auto ic) {
      constexpr int kk = decltype(ic)::value / NT, a = decltype(ic)::value % NT;
      constexpr int pending = (1 - kk) * NF + (NT - 1 - a);   // LDS reads issued after W_a of this k-half
      if constexpr (MT == 4) {
        if constexpr (a == 0) lds_wait<pending>(xf[kk][0], xf[kk][1], xf[kk][2], xf[kk][3], wf[kk][0]);
        else lds_wait<pending>(wf[kk][a]);
      } else if constexpr (a == 0) {   // 64 x 64 tile: groups of 2 MFMA are too short for a wait each: one per k-half
        static_assert(NT == 2, "64-row tiles come with 64 columns");
        lds_wait<(1 - kk) * NF>(xf[kk][0], xf[kk][1], wf[kk][0], wf[kk][1]);
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
        acc[a][m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][a], xf[kk][m], acc[a][m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();   // drains the DMA issued in this step (vmcnt(0)) and protects the stage swap
  };
  for (int kt = kt0; kt < nk; kt += 2) {
    step(kt, std::integral_constant<int, 0>{});
    if (kt + 1 < nk) step(kt + 1, std::integral_constant<int, 1>{});
  }

  // ---- epilogue (all loads issued before any use) ----
  const int ncol = n0 + wave_n * (BN / 2) + g * 4;
  if (S > 1) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int mm = m0 + wave_m * (BM / 2) + m * 16 + l15;
      if (mm >= p.M) continue;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int n = ncol + a * 16;
        if (n < p.N) *reinterpret_cast<f32x4*>(p.splitk_ws + ((long long)ksplit * p.M + mm) * p.N + n) = acc[a][m];
      }
    }
    return;
  }
  f32x4 bb[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    if constexpr (EARLY) bb[a] = bb_e[a];
    else {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bvec && ncol + a * 16 < p.N) t = *reinterpret_cast<const float4*>(bvec + ncol + a * 16);
      bb[a] = (f32x4){t.x, t.y, t.z, t.w};
    }
  }
  int mrow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mm = m0 + wave_m * (BM / 2) + m * 16 + l15;
    mrow[m] = mm < p.M ? mm : -1;
  }
  f16x4 rr[MT][NT], rl[MT][NT];
  if constexpr (EARLY) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int a = 0; a < NT; ++a) { rr[m][a] = rr_e[m][a]; rl[m][a] = rl_e[m][a]; }
  } else if (p.res) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        rr[m][a] = (f16x4){(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        rl[m][a] = rr[m][a];
        if (mrow[m] >= 0 && ncol + a * 16 < p.N) {
          rr[m][a] = *reinterpret_cast<const f16x4*>(p.res + (long long)mrow[m] * p.ld_res + ncol + a * 16);
          if (p.res_lo) rl[m][a] = *reinterpret_cast<const f16x4*>(p.res + (long long)mrow[m] * p.ld_res + p.res_lo + ncol + a * 16);
        }
      }
  }
  // GEGLU epilogue (BasicTransformerBlock.ff.net.0: Linear(C, 8C) -> x * gelu_erf(gate)): the weight rows were interleaved at
  // load time so that column tile 2j of a wave is x[16 channels] and tile 2j+1 the matching gate; the [M, 8C] intermediate (168
  // MB per call at SD-v1.5 sizes) and the separate activation kernel disappear.  Stores: 16 bytes after a permlane swap, as below.
  if (p.geglu) {
    static_assert(NT % 2 == 0, "GEGLU pairs column tiles");
#pragma unroll
    for (int mp = 0; mp < MT; mp += 2) {
      const int mms = m0 + wave_m * (BM / 2) + (mp + (g & 1)) * 16 + l15;   // this lane's row after the swap
#pragma unroll
      for (int j = 0; j < NT / 2; ++j) {
        uint2 pq[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x4 xv = acc[2 * j][mp + h] + bb[2 * j], gv = acc[2 * j + 1][mp + h] + bb[2 * j + 1];
          f16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (f16)(xv[r] * gelu_erf(gv[r]));
          pq[h] = __builtin_bit_cast(uint2, o);
        }
        auto r0 = __builtin_amdgcn_permlane16_swap(pq[0].x, pq[1].x, false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(pq[0].y, pq[1].y, false, false);
        const int nb = (n0 + wave_n * (BN / 2)) / 2 + j * 16 + (g & ~1) * 4;   // output channel (of N/2)
        if (mms < p.M && nb < p.N / 2) *reinterpret_cast<uint4*>(reinterpret_cast<f16*>(p.y) + (long long)mms * p.ldy + nb) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
      }
    }
    return;
  }
  // 16-byte stores (see kernels_conv3x3.hip): one v_permlane16_swap per dword between the packed values of two m-tiles
  // leaves even-g lanes with channels 4g..4g+7 of the first tile's row and odd-g lanes with 4(g-1)..4(g-1)+7 of the second's
  if (!p.out_f32 && (p.N & 7) == 0 && (p.ldy & 7) == 0 && (p.y_lo & 7) == 0) {
#pragma unroll
    for (int mp = 0; mp < MT; mp += 2) {
      const int mms = m0 + wave_m * (BM / 2) + (mp + (g & 1)) * 16 + l15;   // this lane's row after the swap
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        uint2 pq[2], pl[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = mp + h;
          f32x4 v = acc[a][m] + bb[a];
          if (p.res) { v += up4(rr[m][a]); if (p.res_lo) v += up4(rl[m][a]); }
          const f16x4 o = cvt4(v);
          if (p.stats) acc[a][m] = p.y_lo ? v : up4(o);
          pq[h] = __builtin_bit_cast(uint2, o);
          pl[h] = __builtin_bit_cast(uint2, cvt4(v - up4(o)));
        }
        auto r0 = __builtin_amdgcn_permlane16_swap(pq[0].x, pq[1].x, false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(pq[0].y, pq[1].y, false, false);
        const int nb = n0 + wave_n * (BN / 2) + a * 16 + (g & ~1) * 4;
        if (mms < p.M && nb < p.N) *reinterpret_cast<uint4*>(reinterpret_cast<f16*>(p.y) + (long long)mms * p.ldy + nb) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        if (p.y_lo) {   // lo halves of the split output
          auto l0 = __builtin_amdgcn_permlane16_swap(pl[0].x, pl[1].x, false, false);
          auto l1 = __builtin_amdgcn_permlane16_swap(pl[0].y, pl[1].y, false, false);
          if (mms < p.M && nb < p.N) *reinterpret_cast<uint4*>(reinterpret_cast<f16*>(p.y) + (long long)mms * p.ldy + p.y_lo + nb) = make_uint4(l0[0], l1[0], l0[1], l1[1]);
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
      f32x4 v = acc[a][m] + bb[a];
      if (p.res) { v += up4(rr[m][a]); if (p.res_lo) v += up4(rl[m][a]); }
      if (p.out_f32) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.y) + (long long)mrow[m] * p.ldy + n) = v;
      } else {
        const f16x4 o = cvt4(v);
        *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + (long long)mrow[m] * p.ldy + n) = o;
        if (p.y_lo) *reinterpret_cast<f16x4*>(reinterpret_cast<f16*>(p.y) + (long long)mrow[m] * p.ldy + p.y_lo + n) = cvt4(v - up4(o));
        if (p.stats) acc[a][m] = p.y_lo ? v : up4(o);
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

// LDIFF_GEMM_MFAST: -1 (default) automatic, 0 off, 1 always.  Automatic: the weight matrix is the larger operand and does not fit an XCD's L2
// beside the activations (the UNet's 16 x 16 and 8 x 8 levels: M = 2048 / 512 rows against N = 1280 ... 10240).
static bool gemm_m_fast(const ConvParams& p, int ntm, int ntn) {
  static const int mode = [] { const char* e = getenv("LDIFF_GEMM_MFAST"); return e ? atoi(e) : -1; }();
  if (mode == 0 || ntm <= 1 || ntn <= 1 || p.w_bstride != 0) return false;
  if (mode > 0) return true;
  // measured, same box: 1280 -> 10240 at 2048 rows 78.0 -> 64.3 us, 1280 -> 3840 at 512 rows 15.5 -> 10.9 us; the 64 x 64 / 32 x 32 levels lose 5-50 %
  return 2LL * p.N > 3LL * p.M && (long long)p.N * p.K * 2 > (2LL << 20);
}

template <int BM, int BN>
void launch_g(const ConvParams& p, hipStream_t s) {
  const size_t smem = (size_t)2 * (BM + BN) * 8 * 16;
  auto kern = gemm_dma_kernel<BM, BN>;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)smem);
  const int ntm = (p.M + BM - 1) / BM, ntn = (p.N + BN - 1) / BN;
  static const std::string pname = std::string("gemm_dma<") + std::to_string(BM) + "," + std::to_string(BN) + ">";
  const double bytes = (double)p.M * p.K * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * p.N * (p.out_f32 || p.y_lo ? 4.0 : 2.0) + (p.res ? (double)p.M * p.N * (p.res_lo ? 4.0 : 2.0) : 0.0);
  ProfScope prof(pname.c_str(), 2.0 * p.M * (double)p.N * p.K, bytes, s);
  const int S = p.splitk > 1 ? p.splitk : 1;
  LDIFF_CHECK(S == 1 || (p.splitk_ws && !p.geglu && !p.out_f32 && p.w_bstride == 0), LDIFF_ERR_INVALID, "gemm: split-K needs a workspace and a plain fp16 epilogue");   // (fused statistics of a split launch: by the reduce kernel)
  ConvParams q = p;
  q.tiles_m = ntm; q.img_fast = gemm_m_fast(p, ntm, ntn) ? 1 : 0;
  hipLaunchKernelGGL(kern, dim3(ntm * ntn, S), dim3(256), smem, s, q);
  if (S > 1) { HIP_CHECK(hipGetLastError()); launch_splitk_reduce(p, s); }
  HIP_CHECK(hipGetLastError());
}

}  // namespace

bool gemm_dma_eligible(const ConvParams& p) {
  if (p.w_bstride > 0 && ((p.Hout * p.Wout) % 64 != 0 || p.stats)) return false;   // a tile must lie inside one image
  if (p.geglu && (p.N % 32 != 0 || (p.ldy & 7) != 0 || p.res || p.out_f32 || p.stats || p.y_lo)) return false;
  const int pitch1 = p.ld1 ? p.ld1 : p.C1, pitch2 = p.ld2 ? p.ld2 : p.C2, pmax = pitch1 > pitch2 ? pitch1 : pitch2;
  return p.ks == 1 && p.stride == 1 && p.ups == 0 && p.pad_t == 0 && p.pad_l == 0 && !p.gn_scale && !p.temb && p.K % 64 == 0 && p.C1 % 64 == 0 &&
         pitch1 % 8 == 0 && pitch2 % 8 == 0 &&
         p.Hout == p.Hin && p.Wout == p.Win && p.M < (1 << 24) && pmax * 2 < (1 << 24) && (long long)p.M * pmax * 2 < (1LL << 31) && (long long)p.Nrows * p.K * 2 < (1LL << 31);
}

// Split-K for the LDS-DMA GEMM, same rule as igemm_splitk_plan: only where the tiles leave most workgroup slots empty and K is long
// (1x1 shortcut convs over the concat input at the 8x8 level: M = 512, K = 2560 or 5120 on split operands)
int gemm_dma_splitk_plan(const ConvParams& p) {
  if (p.out_f32 || p.geglu || p.w_bstride != 0 || p.M <= 0 || !gemm_dma_eligible(p)) return 1;
  if (p.stats && (p.Hout * p.Wout) % 32 != 0) return 1;   // the reduce kernel emits them in 32-row blocks
  auto tiles = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  if (tiles(128, 64) >= 384) return 1;
  const int nk = p.K / 64;
  auto splits = [&](long long t) { int S = (int)(512 / t); S = S > nk / 16 ? nk / 16 : S; return S > 8 ? 8 : S; };
  // the largest tile whose splits fill the chip (128x128: 0.0156 operand bytes per flop, 128x64: 0.023, 64x64: 0.031); launch_gemm_dma picks
  // the tile by tiles x S with the same thresholds
  const bool n_small = p.N <= 64 || (p.N % 128 != 0 && p.N % 128 <= 64 && p.N < 512);
  const int S256 = n_small ? 1 : splits(tiles(128, 128));
  if (S256 >= 2 && tiles(128, 128) * S256 >= 384) return S256;
  const int S128 = splits(tiles(128, 64));
  if (S128 >= 2 && tiles(128, 64) * S128 >= 384) return S128;
  int S64 = splits(tiles(64, 64));
  if (S64 < 1) S64 = 1;
  // small grids (batch 1 / 2; the 8 x 8 level): a finer cut where the model of common.h sees it
  static const bool fine = [] { const char* e = getenv("LDIFF_SPLITK_FINE"); return !e || atoi(e) != 0; }();
  if (fine && tiles(64, 64) * S64 < 256) S64 = splitk_by_model(tiles(64, 64), nk, 4, (double)p.M * p.N * 4.0, S64);
  return S64 >= 2 ? S64 : 1;
}

void launch_gemm_dma(const ConvParams& p, hipStream_t s) {
  static const int force = [] { const char* e = getenv("LDIFF_GEMM_TILE"); return e ? atoi(e) : 0; }();   // diagnostic: 1 = 128x128, 2 = 128x64, 3 = 64x64
  if (force == 1 && (p.w_bstride == 0 || (p.Hout * p.Wout) % 128 == 0)) return launch_g<128, 128>(p, s);
  if (force == 2 && (p.w_bstride == 0 || (p.Hout * p.Wout) % 128 == 0)) return launch_g<128, 64>(p, s);
  if (force == 3) return launch_g<64, 64>(p, s);
  const int S = p.splitk > 1 ? p.splitk : 1;   // split-K multiplies the workgroups of a tile shape
  auto tiles = [&](int bm, int bn) { return (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn); };
  const bool n_small = p.N <= 64 || (p.N % 128 != 0 && p.N % 128 <= 64 && p.N < 512);
  const bool bm128_ok = p.w_bstride == 0 || (p.Hout * p.Wout) % 128 == 0;   // per-image weights: 128-row tiles only if they divide an image
  if (!n_small && bm128_ok && tiles(128, 128) * S >= 384) launch_g<128, 128>(p, s);
  else if (bm128_ok && tiles(128, 64) * S >= 384) launch_g<128, 64>(p, s);
  else launch_g<64, 64>(p, s);
}
