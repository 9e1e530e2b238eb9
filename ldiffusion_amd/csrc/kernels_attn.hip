// Flash-style attention for gfx950 (MI355X): fp16 operands, fp32 online softmax, MFMA 16x16x32.
//
// Replaces the scaled_dot_product_attention calls inside diffusers' BasicTransformerBlock (self- and
// cross-attention, 8 heads, head dims 40/80/160) and the single-head 512-dim attention of the VAE
// mid block (reached from /root/reference/segmentor.py:103,519,529 and pixel_latent_vector.py:73,78,81).
//
// Per workgroup: one (batch, head) and 64*QT query rows; 4 waves, each wave owns 16*QT query rows.
//   S^T = K Q^T   : A-operand = K tile rows from LDS (ds_read_b128, XOR-swizzled), B-operand = Q in registers.
//                   The accumulator then holds, per lane, one query column and 4 keys per 16-key tile, so the
//                   softmax statistics are lane-local up to two cross-lane steps (xor 16, xor 32).
//   O^T += V^T P^T: A-operand = V^T read straight from the row-major V tile with ds_read_b64_tr_b16 (hardware
//                   transpose), B-operand = the probabilities converted in place (no LDS round trip).
// The key order inside one 32-key MFMA step is the accumulator's native order (4g+r of tile 2s, then of
// tile 2s+1); the V^T read uses the same order, so no permutation is ever materialised.
#include <type_traits>

#include "common.h"

template <int DQK>
struct KLayout {
  static constexpr int STR = ((DQK * 2 + 127) / 128) * 8;  // row stride in 16-byte chunks (multiple of 128 B)
  static constexpr bool EVEN = ((STR / 8) % 2) == 0;       // 256-B-multiple rows: swizzle over 16 chunks
  __device__ static __forceinline__ int off(int row, int chunk) {
    return row * STR + (EVEN ? (chunk ^ (row & 15)) : (chunk ^ ((row >> 1) & 7)));
  }
};

template <int DV>
struct VLayout {
  static constexpr int STR_DW = ((DV / 16) % 2 == 1) ? DV / 2 : DV / 2 + 8;  // row stride in dwords, == 8*odd
};

// MINW = waves per SIMD the register allocation must allow: 2 keeps the whole accumulator file in VGPRs (no
// v_accvgpr_read/write traffic around the softmax / rescale VALU work); the large-head variants need 1.
template <int DQK, int DV, int BKV, int QT, int MINW>
__global__ __launch_bounds__(256, MINW) void attn_kernel(const AttnParams p) {
  using KL = KLayout<DQK>;
  constexpr int KS = DQK / 32;        // MFMA k-steps for Q K^T
  constexpr int DT = DV / 16;         // output d tiles
  constexpr int NT = BKV / 16;        // key tiles per step
  constexpr int VSTR = VLayout<DV>::STR_DW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* sK = reinterpret_cast<uint4*>(smem_raw);                       // [BKV][KL::STR] chunks
  unsigned* sV = reinterpret_cast<unsigned*>(sK + BKV * KL::STR);       // [BKV][VSTR] dwords

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int h = blockIdx.y, b = blockIdx.z;
  const int d = p.d;
  const int qbase = blockIdx.x * (64 * QT) + wave * (16 * QT);

  const f16* Qp = p.q + (long long)b * p.q_bstride + h * d;
  const f16* Kp = p.k + (long long)b * p.kv_bstride + h * d;
  const f16* Vp = p.v + (long long)b * p.kv_bstride + h * d;

  // ---- Q fragments (B operand): lane holds Q[q = l15][dd = ks*32 + 8g .. +7] ----
  f16x8 qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = qbase + qt * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int dd = ks * 32 + g * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qi < p.Lq && dd < d) v = *reinterpret_cast<const uint4*>(Qp + (long long)qi * p.ldq + dd);
      qf[qt][ks] = __builtin_bit_cast(f16x8, v);
    }
  }

  f32x4 oacc[QT][DT];
  float mrun[QT], lrun[QT];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrun[qt] = -1e30f; lrun[qt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float sl2 = p.scale * 1.4426950408889634f;

  // K/V tiles are prefetched global -> registers one tile ahead (issued right after the previous tile was written to
  // LDS, so the loads fly under the MFMA/softmax work of the current tile) and written to LDS after the next barrier.
  constexpr int KCH = BKV * (DQK / 8), VCH = BKV * (DV / 8);
  constexpr int KIT = (KCH + 255) / 256, VIT = (VCH + 255) / 256;
  constexpr bool PFV = DV <= 160;   // d=512: the register file is full (O^T alone is 128 registers) -> V is staged without prefetch
  uint4 rk[KIT], rv[PFV ? VIT : 1];
  auto load_kv = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
      const int c = tid + i * 256, row = c / (DQK / 8), ch = c - row * (DQK / 8);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (c < KCH && kv0 + row < p.Lk && ch * 8 < d) v = *reinterpret_cast<const uint4*>(Kp + (long long)(kv0 + row) * p.ldk + ch * 8);
      rk[i] = v;
    }
    if (PFV) {
#pragma unroll
      for (int i = 0; i < VIT; ++i) {
        const int c = tid + i * 256, row = c / (DV / 8), ch = c - row * (DV / 8);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < VCH && kv0 + row < p.Lk && ch * 8 < d) v = *reinterpret_cast<const uint4*>(Vp + (long long)(kv0 + row) * p.ldv + ch * 8);
        rv[i] = v;
      }
    }
  };
  auto store_kv = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
      const int c = tid + i * 256, row = c / (DQK / 8), ch = c - row * (DQK / 8);
      if (c < KCH) sK[KL::off(row, ch)] = rk[i];
    }
#pragma unroll
    for (int i = 0; i < VIT; ++i) {
      const int c = tid + i * 256, row = c / (DV / 8), ch = c - row * (DV / 8);
      uint4 v = make_uint4(0, 0, 0, 0);
      if (PFV) v = rv[i];
      else if (c < VCH && kv0 + row < p.Lk && ch * 8 < d) v = *reinterpret_cast<const uint4*>(Vp + (long long)(kv0 + row) * p.ldv + ch * 8);
      if (c < VCH) *reinterpret_cast<uint4*>(sV + row * VSTR + ch * 4) = v;
    }
  };
  load_kv(0);

  // One key tile.  TAIL is static: the key mask (compares + selects on every score) exists only in the instantiation
  // used for the last, partial tile.
  auto tile = [&](int kv0, auto TAILC) {
    constexpr bool TAIL = decltype(TAILC)::value;
    __syncthreads();  // previous tile fully consumed
    store_kv(kv0);
    __syncthreads();
    if (kv0 + BKV < p.Lk) load_kv(kv0 + BKV);

    // ---- S^T = K Q^T ----
    f32x4 sacc[QT][NT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int t = 0; t < NT; ++t) sacc[qt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int row = t * 16 + l15;
        f16x8 kf = __builtin_bit_cast(f16x8, sK[KL::off(row, ks * 4 + g)]);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          sacc[qt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[qt][ks], sacc[qt][t], 0, 0, 0);
      }
    }

    // ---- online softmax (lane: query l15, keys 16t + 4g + r) ----
    // VALU-bound for small head dims, so the per-score work is kept to max, one FMA, one bare v_exp_f32, one add and
    // the fp16 convert: the running max is tracked on the RAW scores (scale > 0 keeps the order), the softmax scale
    // and log2(e) are folded into the exponent FMA, and the key mask is applied only on the (uniform) tail tile.
    f16x8 pf[QT][NT / 2];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = -1e30f;
      if (TAIL) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (kv0 + t * 16 + g * 4 + r >= p.Lk) sacc[qt][t][r] = -1e30f;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[qt][t][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mnew = fmaxf(mrun[qt], mx);
      const float alpha = __builtin_amdgcn_exp2f((mrun[qt] - mnew) * sl2);
      const float moff = -mnew * sl2;
      float rs = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(fmaf(sacc[qt][t][r], sl2, moff));
          rs += pv;
          pf[qt][t >> 1][(t & 1) * 4 + r] = (f16)pv;
        }
      rs += __shfl_xor(rs, 16);
      rs += __shfl_xor(rs, 32);
      lrun[qt] = lrun[qt] * alpha + rs;
      mrun[qt] = mnew;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        oacc[qt][dt][0] *= alpha; oacc[qt][dt][1] *= alpha; oacc[qt][dt][2] *= alpha; oacc[qt][dt][3] *= alpha;
      }
    }

    // ---- O^T += V^T P^T ----
    const int tq = l15 >> 2, tp = l15 & 3;  // lane 4q+p of its 16-lane group addresses row q, columns 4p..4p+3
#pragma unroll
    for (int s2 = 0; s2 < NT / 2; ++s2) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int r0 = (2 * s2) * 16 + g * 4 + tq, r1 = r0 + 16;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sV + r0 * VSTR + dt * 8 + tp * 2));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sV + r1 * VSTR + dt * 8 + tp * 2));
        f16x4 lo_h = __builtin_bit_cast(f16x4, lo), hi_h = __builtin_bit_cast(f16x4, hi);
        f16x8 vf = {lo_h[0], lo_h[1], lo_h[2], lo_h[3], hi_h[0], hi_h[1], hi_h[2], hi_h[3]};
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
          oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt][s2], oacc[qt][dt], 0, 0, 0);
      }
    }
  };
  const int full = p.Lk / BKV * BKV;
  for (int kv0 = 0; kv0 < full; kv0 += BKV) tile(kv0, std::false_type());
  if (full < p.Lk) tile(full, std::true_type());

  // ---- normalise and store: lane holds O[q = l15][dd = 16dt + 4g + r] ----
  f16* Op = p.o + (long long)b * p.o_bstride + h * d;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = qbase + qt * 16 + l15;
    const float inv = 1.0f / lrun[qt];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int dd = dt * 16 + g * 4;
      if (qi < p.Lq && dd < d) {
        f16x4 o = {(f16)(oacc[qt][dt][0] * inv), (f16)(oacc[qt][dt][1] * inv), (f16)(oacc[qt][dt][2] * inv), (f16)(oacc[qt][dt][3] * inv)};
        *reinterpret_cast<f16x4*>(Op + (long long)qi * p.ldo + dd) = o;
      }
    }
  }
}

template <int DQK, int DV, int BKV, int QT, int MINW = 2>
static void launch_attn_cfg(const AttnParams& p, hipStream_t s) {
  static bool attr_set = false;
  const size_t smem = (size_t)BKV * KLayout<DQK>::STR * 16 + (size_t)BKV * VLayout<DV>::STR_DW * 4;
  auto kern = attn_kernel<DQK, DV, BKV, QT, MINW>;
  if (!attr_set) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  dim3 grid((p.Lq + 64 * QT - 1) / (64 * QT), p.heads, p.B);
  static const std::string pname = std::string("attn<") + std::to_string(DQK) + "," + std::to_string(DV) + ">";
  const double bh = (double)p.B * p.heads;
  ProfScope prof(pname.c_str(), 4.0 * bh * p.Lq * p.Lk * p.d,
                 2.0 * bh * p.d * (2.0 * p.Lq + 2.0 * p.Lk * (p.kv_bstride ? 1.0 : 1.0 / p.B)), s);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
  HIP_CHECK(hipGetLastError());
}

void launch_attention(const AttnParams& p, hipStream_t s) {
  LDIFF_CHECK(p.d % 8 == 0 && p.d > 0 && p.d <= 512, LDIFF_ERR_INVALID, "attention: head dim %d must be a multiple of 8 and <= 512", p.d);
  LDIFF_CHECK(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0, LDIFF_ERR_INVALID, "attention: row strides must be multiples of 8");
  LDIFF_CHECK(p.Lk > 0 && p.Lq > 0, LDIFF_ERR_INVALID, "attention: empty sequence (Lq=%d Lk=%d)", p.Lq, p.Lk);
  const int d = p.d;
  if (d <= 16) launch_attn_cfg<32, 16, 64, 2>(p, s);
  else if (d <= 32) launch_attn_cfg<32, 32, 64, 2>(p, s);
  else if (d <= 48) launch_attn_cfg<64, 48, 64, 2>(p, s);
  else if (d <= 64) launch_attn_cfg<64, 64, 64, 2>(p, s);
  else if (d <= 80) launch_attn_cfg<96, 80, 64, 2>(p, s);
  else if (d <= 96) launch_attn_cfg<96, 96, 64, 2>(p, s);
  else if (d <= 128) launch_attn_cfg<128, 128, 64, 2, 1>(p, s);
  else if (d <= 160) launch_attn_cfg<160, 160, 64, 2, 1>(p, s);
  else launch_attn_cfg<512, 512, 32, 1, 1>(p, s);
}
