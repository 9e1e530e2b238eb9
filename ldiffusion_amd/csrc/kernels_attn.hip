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

// max over lane pairs (l, l ^ 16) / (l, l ^ 32): v_permlane16_swap / v_permlane32_swap of the value with itself leave the two partners' values
// side by side in the two results
__device__ __forceinline__ float xmax16(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float xmax32(float v) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}

// MINW = waves per SIMD the register allocation must allow: 2 keeps the whole accumulator file in VGPRs (no
// v_accvgpr_read/write traffic around the softmax / rescale VALU work); the large-head variants need 1.
// ONES: the head dim leaves padding columns in the V tile (d < DV, e.g. 40 of 48): column d of V is set to 1, so row d of O^T
// accumulates the softmax row sums inside the P.V MFMAs and the per-score adds and cross-lane sums of the (VALU-bound) softmax go.
template <int DQK, int DV, int BKV, int QT, int MINW, bool ONES, bool PRE>
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
  // XCD-aware order: the hardware deals workgroups round-robin over the 8 XCDs, so the query tiles of one (image, head) pair -- which all
  // stream the same K and V -- would land on eight different L2s.  Remapped, the workgroups an XCD receives form a contiguous run of
  // the (pair, query tile) list: a pair's K / V are fetched into ONE L2.
  int qtile = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  if (p.xcd_order) {
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
    const int id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7, idx = id >> 3;
    int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    qtile = sw % gx; sw /= gx;
    h = sw % gy; b = sw / gy;
  }
  const int d = p.d;
  const int qbase = qtile * (64 * QT) + wave * (16 * QT);

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
  float mrun[QT], lrun[QT], mref[QT];   // mref: PRESCALED form, the running reference (an fp16 value, log2 units)
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    mrun[qt] = -1e30f; lrun[qt] = 0.f; mref[qt] = 0.f;
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
      if (PRE && c < KCH && kv0 + row < p.Lk && ch * 8 == d) v.x = 0x3C00u;   // K[key][d] = 1.0: the MFMA then adds Q[q][d] = -reference to every score
      rk[i] = v;
    }
    if (PFV) {
#pragma unroll
      for (int i = 0; i < VIT; ++i) {
        const int c = tid + i * 256, row = c / (DV / 8), ch = c - row * (DV / 8);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < VCH && kv0 + row < p.Lk && ch * 8 < d) v = *reinterpret_cast<const uint4*>(Vp + (long long)(kv0 + row) * p.ldv + ch * 8);
        if (ONES && c < VCH && kv0 + row < p.Lk && ch * 8 == d) v.x = 0x3C00u;   // V[key][d] = 1.0 (fp16), the rest of the padding stays 0
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
      else if (ONES && c < VCH && kv0 + row < p.Lk && ch * 8 == d) v.x = 0x3C00u;
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
      for (int t = 0; t < NT; ++t) {   // (a max b) max c: one v_max3_f32 per two scores
        mx = fmaxf(fmaxf(mx, sacc[qt][t][0]), sacc[qt][t][1]);
        mx = fmaxf(fmaxf(mx, sacc[qt][t][2]), sacc[qt][t][3]);
      }
      mx = xmax16(mx);   // over the four lanes l15 + 16 g of the query (VALU lane swaps: no LDS crossbar round trip in the chain)
      mx = xmax32(mx);
      float alpha = 1.0f;
      float rs = 0.f;
      if constexpr (PRE) {
        // PRESCALED form (AttnParams::prescaled): Q arrives multiplied by scale * log2(e) (in the producing GEMM's fp32 epilogue: same number
        // of roundings), K carries 1.0 in padding column d and Q the NEGATED running reference there, so the MFMAs deliver S' - reference and
        // the per-score FMA disappears: p = exp2(sacc).  The reference is an fp16 value (it lives in a Q fragment); softmax does not care which
        // reference is used as long as every probability of the row uses the same one, which the fix-up below guarantees: on a tile where a
        // row's maximum exceeds its reference (or on the first tile) the new reference is f16(reference + tile maximum), the difference is
        // exact in fp32, this tile's scores are shifted by it, O^T (with the row sums in its row d or in lrun) is scaled by 2^-difference and
        // the Q fragment is updated for the tiles to come.  Wave-uniform branch: after the first tiles it is rarely taken.
        const bool need = kv0 == 0 || mx > p.rescale_log2;
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
          const float rnew = need ? (float)(f16)(mref[qt] + mx) : mref[qt];
          const float delta = rnew - mref[qt];
          alpha = kv0 == 0 ? 0.f : __builtin_amdgcn_exp2f(-delta);
          mref[qt] = rnew;
          if (g == ((d & 31) >> 3)) qf[qt][DQK / 32 - 1][0] = (f16)(-rnew);   // (d % 8 == 0 and d >= DQK - 32: the padding column sits in the last k-step)
#pragma unroll
          for (int t = 0; t < NT; ++t) { sacc[qt][t][0] -= delta; sacc[qt][t][1] -= delta; sacc[qt][t][2] -= delta; sacc[qt][t][3] -= delta; }
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            oacc[qt][dt][0] *= alpha; oacc[qt][dt][1] *= alpha; oacc[qt][dt][2] *= alpha; oacc[qt][dt][3] *= alpha;
          }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pv = __builtin_amdgcn_exp2f(sacc[qt][t][r]);
            if (!ONES) rs += pv;
            pf[qt][t >> 1][(t & 1) * 4 + r] = (f16)pv;
          }
      } else {
      // Deferred rescale: the running maximum only moves when the tile's maximum exceeds it by more than p.rescale_log2 in the exponent's
      // (log2) units -- until then the probabilities are taken against the OLD maximum and may reach 2^threshold instead of 1.  Threshold 0 (the default) is
      // the classic rule with the rescale skipped, wave-uniformly, on the tiles where no row's maximum grew.  Everything at the old scale (O^T, with
      // the row sums in its row d or in lrun) is scaled exactly once, before this tile's P exists.
      const bool need = (mx - mrun[qt]) * sl2 > p.rescale_log2;
      if (__builtin_amdgcn_ballot_w64(need) != 0) {
        const float mnew = need ? mx : mrun[qt];
        alpha = __builtin_amdgcn_exp2f((mrun[qt] - mnew) * sl2);   // 1 where the row keeps its maximum
        mrun[qt] = mnew;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          oacc[qt][dt][0] *= alpha; oacc[qt][dt][1] *= alpha; oacc[qt][dt][2] *= alpha; oacc[qt][dt][3] *= alpha;
        }
      }
      const float moff = -mrun[qt] * sl2;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pv = __builtin_amdgcn_exp2f(fmaf(sacc[qt][t][r], sl2, moff));
          if (!ONES) rs += pv;
          pf[qt][t >> 1][(t & 1) * 4 + r] = (f16)pv;
        }
      }
      if (!ONES) {
        rs += __shfl_xor(rs, 16);
        rs += __shfl_xor(rs, 32);
        lrun[qt] = lrun[qt] * alpha + rs;
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
    float lsum = lrun[qt];
    if (ONES) {   // the row sum sits in O^T row d: lane (g = (d%16)/4, same l15), register d%4 of d tile d/16
      float cand = 0.f;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (dt * 16 + g * 4 + r == d) cand = oacc[qt][dt][r];
      lsum = __shfl(cand, ((d & 15) >> 2) * 16 + l15);
    }
    const float inv = 1.0f / lsum;
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

template <int DQK, int DV, int BKV, int QT, int MINW, bool ONES, bool PRE>
static void launch_attn_cfg2(const AttnParams& p, hipStream_t s);
template <int DQK, int DV, int BKV, int QT, int MINW = 2>
static void launch_attn_cfg(const AttnParams& p, hipStream_t s) {
  // PRESCALED needs a free K / Q column in the LAST k-step: instantiated for the two shapes the UNet has (d = 40 of 64, d = 80 of 96)
  constexpr bool HAS_PRE = (DQK == 64 && DV == 48) || (DQK == 96 && DV == 80);
  if constexpr (HAS_PRE) {
    if (p.prescaled) {
      LDIFF_CHECK(p.d % 8 == 0 && p.d < DQK && p.d >= DQK - 32, LDIFF_ERR_INVALID, "attention: the prescaled form needs a padding column in the last k-step (d=%d)", p.d);
      if (p.d < DV) launch_attn_cfg2<DQK, DV, BKV, QT, MINW, true, true>(p, s);
      else launch_attn_cfg2<DQK, DV, BKV, QT, MINW, false, true>(p, s);
      return;
    }
  }
  LDIFF_CHECK(!p.prescaled, LDIFF_ERR_INVALID, "attention: the prescaled form is not built for head dim %d", p.d);
  if (p.d < DV) launch_attn_cfg2<DQK, DV, BKV, QT, MINW, true, false>(p, s);    // a padding column of V is free for the row sums
  else launch_attn_cfg2<DQK, DV, BKV, QT, MINW, false, false>(p, s);
}
template <int DQK, int DV, int BKV, int QT, int MINW, bool ONES, bool PRE>
static void launch_attn_cfg2(const AttnParams& p, hipStream_t s) {
  const size_t smem = (size_t)BKV * KLayout<DQK>::STR * 16 + (size_t)BKV * VLayout<DV>::STR_DW * 4;
  auto kern = attn_kernel<DQK, DV, BKV, QT, MINW, ONES, PRE>;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)smem);
  dim3 grid((p.Lq + 64 * QT - 1) / (64 * QT), p.heads, p.B);
  static const std::string pname = std::string("attn<") + std::to_string(DQK) + "," + std::to_string(DV) + ">";
  const double bh = (double)p.B * p.heads;
  ProfScope prof(pname.c_str(), 4.0 * bh * p.Lq * p.Lk * p.d,
                 2.0 * bh * p.d * (2.0 * p.Lq + 2.0 * p.Lk * (p.kv_bstride ? 1.0 : 1.0 / p.B)), s);
  static const int xcd_mode = [] { const char* e = getenv("LDIFF_ATTN_XCD"); return e ? atoi(e) : 1; }();   // 0: the grid's own order (A/B timing)
  static const float rescale = [] { const char* e = getenv("LDIFF_ATTN_RESCALE"); return e ? (float)atof(e) : 0.0f; }();   // 0 (default): the maximum moves on every growth; 8: deferred (measured: -0.3 % of a UNet pass, but the
                                                                                                                   // reduced-width UNet error at precision 2 rose 3.6e-4 -> 6.0e-4: profiles/r04_attention_deferred_rescale.txt)
  AttnParams q = p;
  q.rescale_log2 = rescale;
  q.xcd_order = xcd_mode && grid.x > 1 && p.Lk >= 256 ? 1 : 0;   // (short K / V, the cross-attention: nothing to share, the remap only costs; measured
                                                                    //  same box: 4096 x 4096, d = 40: 388 -> 378 us, 1024 x 1024, d = 80: 45.9 -> 43.0 us)
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, q);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------
// Large head dims (the VAE mid block: one head of 512): "d-split" variant.  The generic kernel gives every wave 16 queries
// and the whole head dim, so each wave re-reads the complete K and V tiles from LDS (1 KiB of LDS per MFMA: the LDS array,
// not the matrix pipe, bounds it at ~215 TFLOP/s).  Here a workgroup still owns 64 queries, but
//   * S^T = K Q^T and the online softmax stay per wave on 16 queries (all 32 keys of the step);
//   * the probabilities (fp16) and the rescale factors go through LDS once;
//   * O^T += V^T P^T is split over the HEAD DIM: wave w accumulates d in [128w, 128w+128) for all 64 queries, so every
//     V^T fragment read from LDS feeds four MFMAs (one per query tile) and a wave reads a quarter of the V tile.
// LDS per wave and step: 32 KiB of K + 8 KiB of V + 4 KiB of P for 64 MFMAs (0.69 KiB per MFMA).

// Hand-scheduled LDS fetches for the d-split kernel (one wave per SIMD: nothing else hides an LDS round trip).  hipcc emits
// "ds_read; s_waitcnt lgkmcnt(0); v_mfma" per fragment, i.e. 32 exposed round trips per tile in S^T = K Q^T alone; the reads
// are therefore issued in batches as inline asm and waited for with counted lgkmcnt (the wait names its fragment as an
// in/out operand so the consuming MFMA cannot move above it).
template <int I, int N, class F>
__device__ __forceinline__ void attn_static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); attn_static_for<I + 1, N>(f); }
}
template <int OFF>
__device__ __forceinline__ void attn_lds_read128(f16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void attn_lds_read_tr(f16x4& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT>
__device__ __forceinline__ void attn_lds_wait(f16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void attn_lds_wait(f16x4& a, f16x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(CNT)); }

template <int BKV>
__global__ __launch_bounds__(256, 1) void attn_dsplit_kernel(const AttnParams p) {
  constexpr int D = 512, KS = D / 32, NT = BKV / 16, DTW = 8, PSTR = BKV * 2 + 16;   // P row stride in bytes (16-B aligned, odd multiple of 16)
  static_assert(BKV == 32, "one 32-key MFMA step per tile");
  using KL = KLayout<D>;
  constexpr int VSTR = VLayout<D>::STR_DW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* sK = reinterpret_cast<uint4*>(smem_raw);                        // [BKV][KL::STR] chunks
  unsigned* sV = reinterpret_cast<unsigned*>(sK + BKV * KL::STR);        // [BKV][VSTR] dwords
  unsigned char* sP = reinterpret_cast<unsigned char*>(sV + BKV * VSTR); // [64][PSTR] bytes: P[q][key] fp16
  float* sAl = reinterpret_cast<float*>(sP + 64 * PSTR);                 // [64] rescale factor of the step / final row sums

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  // XCD-aware order over the linearised (query block, head, batch) grid: the workgroups sharing an XCD (blockIdx % 8) take a
  // contiguous range, i.e. the query blocks of ONE (batch, head) at a time, and stream its K/V (8 MB at 4,096 tokens: twice
  // the 4 MB L2 of an XCD) roughly in step, so a tile is fetched from HBM once per XCD instead of once per workgroup
  // (measured before: 4.7 TB/s of K/V re-reads, the kernel was HBM-bound).
  const int nqb = (p.Lq + 63) / 64, nwg = gridDim.x, id = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7, idx = id >> 3;
  int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int qb = sw % nqb; sw /= nqb;
  const int h = sw % p.heads, b = sw / p.heads;
  const int d = p.d;
  const int q0 = qb * 64, qown = q0 + wave * 16 + l15;   // this lane's query in the softmax phase

  const f16* Qp = p.q + (long long)b * p.q_bstride + h * d;
  const f16* Kp = p.k + (long long)b * p.kv_bstride + h * d;
  const f16* Vp = p.v + (long long)b * p.kv_bstride + h * d;

  f16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int dd = ks * 32 + g * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (qown < p.Lq && dd < d) v = *reinterpret_cast<const uint4*>(Qp + (long long)qown * p.ldq + dd);
    qf[ks] = __builtin_bit_cast(f16x8, v);
  }
  f32x4 oacc[4][DTW];   // [query tile][d tile of this wave's slice]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int dt = 0; dt < DTW; ++dt) oacc[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -1e30f, lrun = 0.f;
  const float sl2 = p.scale * 1.4426950408889634f;
  typedef __attribute__((address_space(3))) void lds_void;
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem_raw;
  const unsigned kbase = lds0 + (unsigned)(l15 * 1024 + ((g ^ l15) << 4));                       // K fragment, tile 0, k-step 0
  const unsigned vbase = lds0 + (unsigned)(BKV * KL::STR * 16 + ((g * 8 + (l15 >> 2)) * VSTR + wave * DTW * 8 + (l15 & 3) * 2) * 4);   // V^T fragment, d tile 0

  constexpr int KCH = BKV * (D / 8), KIT = KCH / 256;
  // K and V tiles are prefetched global -> registers one tile ahead (issued right after the previous tile went to LDS, so
  // the loads fly under this tile's MFMA / softmax work) and written to LDS after the next barrier
  uint4 rk[KIT], rv[KIT];
  auto load_k = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
      const int c = tid + i * 256, row = c / (D / 8), ch = c - row * (D / 8);
      uint4 v = make_uint4(0, 0, 0, 0), u = make_uint4(0, 0, 0, 0);
      if (kv0 + row < p.Lk && ch * 8 < d) {
        v = *reinterpret_cast<const uint4*>(Kp + (long long)(kv0 + row) * p.ldk + ch * 8);
        u = *reinterpret_cast<const uint4*>(Vp + (long long)(kv0 + row) * p.ldv + ch * 8);
      }
      rk[i] = v; rv[i] = u;
    }
  };
  auto store_kv = [&](int kv0) {
#pragma unroll
    for (int i = 0; i < KIT; ++i) {
      const int c = tid + i * 256, row = c / (D / 8), ch = c - row * (D / 8);
      sK[KL::off(row, ch)] = rk[i];
      *reinterpret_cast<uint4*>(sV + row * VSTR + ch * 4) = rv[i];
    }
  };
  load_k(0);

  auto tile = [&](int kv0, auto TAILC) {
    constexpr bool TAIL = decltype(TAILC)::value;
    __syncthreads();   // previous tile fully consumed (K, V, P, alpha)
    store_kv(kv0);
    __syncthreads();
    if (kv0 + BKV < p.Lk) load_k(kv0 + BKV);

    // ---- S^T = K Q^T for this wave's 16 queries: K fragments in batches of 8 reads, counted waits ----
    // K[row = 16t + l15][chunk 4ks + g] sits at chunk position (4ks + g) ^ l15 = (g ^ l15) ^ 4ks: byte address = kbase ^ 64ks
    f32x4 sacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sacc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    attn_static_for<0, NT * KS / 8>([&](auto bc) {
      constexpr int t = decltype(bc)::value / (KS / 8), k0 = (decltype(bc)::value % (KS / 8)) * 8;
      f16x8 kf[8];
      attn_static_for<0, 8>([&](auto ic) { constexpr int i = decltype(ic)::value; attn_lds_read128<t * 16 * 1024>(kf[i], kbase ^ (unsigned)((k0 + i) * 64)); });
      attn_static_for<0, 8>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        attn_lds_wait<7 - i>(kf[i]);
        sacc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[i], qf[k0 + i], sacc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    // ---- online softmax (lane: query l15, keys 16t + 4g + r); P and alpha to LDS ----
    if (TAIL) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kv0 + t * 16 + g * 4 + r >= p.Lk) sacc[t][r] = -1e30f;
    }
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mnew = fmaxf(mrun, mx);
    const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * sl2);
    const float moff = -mnew * sl2;
    float rs = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      f16x4 ph;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[t][r], sl2, moff));
        rs += pv;
        ph[r] = (f16)pv;
      }
      *reinterpret_cast<f16x4*>(sP + (wave * 16 + l15) * PSTR + (t * 16 + g * 4) * 2) = ph;   // P[q][key], keys in natural order
    }
    rs += __shfl_xor(rs, 16);
    rs += __shfl_xor(rs, 32);
    lrun = lrun * alpha + rs;
    mrun = mnew;
    if (g == 0) sAl[wave * 16 + l15] = alpha;
    __syncthreads();

    // ---- O^T[d slice of this wave][64 queries] = alpha * O^T + V^T P^T ----
    // The rescale is skipped (exactly: alpha == 1 means the running maximum did not move) unless some query of the workgroup
    // raised its maximum in this step; after the first tiles that is rare.
    f16x8 pf[4];
    float al[4];
    bool moved = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      al[j] = sAl[j * 16 + l15];
      moved |= al[j] != 1.0f;
      pf[j] = *reinterpret_cast<const f16x8*>(sP + (j * 16 + l15) * PSTR + g * 16);   // B operand: query 16j+l15, keys 8g..8g+7
    }
    if (__builtin_amdgcn_ballot_w64(moved) != 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int dt = 0; dt < DTW; ++dt) { oacc[j][dt][0] *= al[j]; oacc[j][dt][1] *= al[j]; oacc[j][dt][2] *= al[j]; oacc[j][dt][3] *= al[j]; }
    }
    // V^T fragments: all 16 transposed reads of the step are issued first (lane 4q+p of its 16-lane group addresses key row
    // 8g+q (+4), d columns 4p..4p+3 of d tile dt), then 8 x 4 MFMAs with counted waits
    f16x4 vlo[DTW], vhi[DTW];
    attn_static_for<0, DTW>([&](auto dc) {
      constexpr int dt = decltype(dc)::value;
      attn_lds_read_tr<dt * 32>(vlo[dt], vbase);                    // keys 8g .. 8g+3
      attn_lds_read_tr<dt * 32 + 4 * VSTR * 4>(vhi[dt], vbase);     // keys 8g+4 .. 8g+7
    });
    attn_static_for<0, DTW>([&](auto dc) {
      constexpr int dt = decltype(dc)::value;
      attn_lds_wait<2 * (DTW - 1 - dt)>(vlo[dt], vhi[dt]);
      const f16x8 vf = {vlo[dt][0], vlo[dt][1], vlo[dt][2], vlo[dt][3], vhi[dt][0], vhi[dt][1], vhi[dt][2], vhi[dt][3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) oacc[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[j], oacc[j][dt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  const int full = p.Lk / BKV * BKV;
  for (int kv0 = 0; kv0 < full; kv0 += BKV) tile(kv0, std::false_type());
  if (full < p.Lk) tile(full, std::true_type());

  // ---- normalise and store: lane holds O[q = 16j + l15][dd = 128 wave + 16 dt + 4g + r] ----
  __syncthreads();
  if (g == 0) sAl[wave * 16 + l15] = lrun;
  __syncthreads();
  f16* Op = p.o + (long long)b * p.o_bstride + h * d;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int qi = q0 + j * 16 + l15;
    const float inv = 1.0f / sAl[j * 16 + l15];
#pragma unroll
    for (int dt = 0; dt < DTW; ++dt) {
      const int dd = (wave * DTW + dt) * 16 + g * 4;
      if (qi < p.Lq && dd < d) {
        const f16x4 o = {(f16)(oacc[j][dt][0] * inv), (f16)(oacc[j][dt][1] * inv), (f16)(oacc[j][dt][2] * inv), (f16)(oacc[j][dt][3] * inv)};
        *reinterpret_cast<f16x4*>(Op + (long long)qi * p.ldo + dd) = o;
      }
    }
  }
}

static void launch_attn_dsplit(const AttnParams& p, hipStream_t s) {
  constexpr int BKV = 32;
  const size_t smem = (size_t)BKV * KLayout<512>::STR * 16 + (size_t)BKV * VLayout<512>::STR_DW * 4 + 64 * (BKV * 2 + 16) + 64 * 4;
  auto kern = attn_dsplit_kernel<BKV>;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)smem);
  dim3 grid(((p.Lq + 63) / 64) * p.heads * p.B);
  const double bh = (double)p.B * p.heads;
  ProfScope prof("attn<512,512>", 4.0 * bh * p.Lq * p.Lk * p.d, 2.0 * bh * p.d * (2.0 * p.Lq + 2.0 * p.Lk * (p.kv_bstride ? 1.0 : 1.0 / p.B)), s);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------
// d = 512, one head, long sequences (the VAE mid-block attention: 4,096 tokens per image at 512^2, 16,384 at 1024^2).  Round 4: the kernel above
// moves 64 KiB of K / V per 64-query step through registers and runs at 0.13 of the MFMA peak.  This one
//   * takes 128 queries per workgroup (wave w: 32 queries in the softmax phase, d slice [128 w, 128 w + 128) of all 128 queries in the P V phase),
//     so a 64 KiB K / V tile feeds 128 MFMAs per wave instead of 64 and every K fragment read from LDS feeds two MFMAs;
//   * brings K / V in by LDS-DMA (one piece = one 1 KiB row; the K swizzle is applied by which 16-byte chunk a lane fetches), double-buffered:
//     tile t + 1 travels during the whole of tile t; two raw barriers per tile with hand-counted waits (every LDS access between them is inline
//     asm, so the compiler cannot put a vmcnt(0) for the travelling tile in front of it);
//   * keeps the 256 output accumulators of a wave in AGPRs and NEVER rescales them (vector instructions cannot address AGPRs: an online-softmax
//     rescale would shuttle all of them through VGPRs, and the register allocator answers that with 700 spills).  Instead every query gets a
//     FIXED reference: m_ref = (maximum over the first key tile) + 4 binades, p = 2^(s - m_ref) -- exact softmax algebra for any reference; fp16
//     P holds up to 2^15, so the pass is valid while no score exceeds m_ref by more than 15 binades (the true maximum may lie up to 19 binades
//     = 13 nats above the first tile's).  A wave tracks the true maximum on the side; if some query left the window, the whole workgroup runs
//     the pass once more with m_ref = the true maxima, which cannot fail.  (The softmax is also cheaper: no per-tile lane reductions, no alpha.)
// Registers: Q fragments 128 + O accumulators 256 (AGPRs) + ~90: one wave per SIMD.
constexpr int D5_QB = 128, D5_BKV = 32;
constexpr unsigned D5_KB = 32 * 1024, D5_VROW = VLayout<512>::STR_DW * 4, D5_VB = 32 * D5_VROW, D5_PSTR = D5_BKV * 2 + 16;
constexpr unsigned D5_K0 = 0, D5_V0 = 2 * D5_KB, D5_P0 = D5_V0 + 2 * D5_VB, D5_AL = D5_P0 + D5_QB * D5_PSTR, D5_FL = D5_AL + D5_QB * 4, D5_LDS = D5_FL + 16;
static_assert(D5_LDS <= 160 * 1024, "LDS budget");
constexpr float D5_LEAD = 4.0f, D5_WINDOW = 15.0f;   // binades: reference above the first tile's maximum; headroom of fp16 P above the reference

__device__ __forceinline__ void d5_lds_write64(unsigned addr, uint2 v) { asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void d5_lds_write32(unsigned addr, float v) { asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void d5_lds_read32(float& d, unsigned addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }

__global__ __launch_bounds__(256, 1) void attn_d512_kernel(const AttnParams p) {
  typedef __attribute__((address_space(3))) void lds_void;
  constexpr int D = 512, KS = D / 32, NT = D5_BKV / 16, DTW = 8, NJ = D5_QB / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  // XCD-aware order (as above): the workgroups of one XCD take a contiguous range of (query block, batch), so an image's K / V tiles are
  // fetched from HBM once per XCD; at 8 images x 32 query blocks an XCD owns exactly one image
  const int nqb = (p.Lq + D5_QB - 1) / D5_QB, nwg = gridDim.x, id = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7, idx = id >> 3;
  int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int qb = sw % nqb; sw /= nqb;
  const int h = sw % p.heads, b = sw / p.heads;
  const int q0 = qb * D5_QB;

  const f16* Qp = p.q + (long long)b * p.q_bstride + h * D;
  const f16* Kp = p.k + (long long)b * p.kv_bstride + h * D;
  const f16* Vp = p.v + (long long)b * p.kv_bstride + h * D;
  const int ldk2 = p.ldk * 2, ldv2 = p.ldv * 2;
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)((long long)(p.Lk - 1) * ldk2 + D * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)((long long)(p.Lk - 1) * ldv2 + D * 2), 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem_raw;

  // rows 8 wave .. 8 wave + 7 of the K and the V tile: 16 pieces of 1 KiB per wave and tile; rows beyond Lk read zeros (num_records)
  auto issue_row = [&](int kv0, int buf, auto pc) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int pp = decltype(pc)::value;
    const int r = wave * 8 + pp;
    const int kvo = (lane ^ (r & 15)) << 4;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk, (lds_void*)(smem_raw + D5_K0 + buf * D5_KB + r * 1024), 16, kvo, (kv0 + r) * ldk2, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (lds_void*)(smem_raw + D5_V0 + buf * D5_VB + r * D5_VROW), 16, lane << 4, (kv0 + r) * ldv2, 0, 0);
#endif
  };
  auto issue_tile = [&](int kv0, int buf) __attribute__((always_inline)) {
    attn_static_for<0, 8>([&](auto pc) { issue_row(kv0, buf, pc); });
  };
  issue_tile(0, 0);

  // Q fragments of the wave's two query tiles (softmax-phase rows q0 + 32 wave + 16 j + l15)
  f16x8 qf[2][KS];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int qi = q0 + wave * 32 + j * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qi < p.Lq) v = *reinterpret_cast<const uint4*>(Qp + (long long)qi * p.ldq + ks * 32 + g * 8);
      qf[j][ks] = __builtin_bit_cast(f16x8, v);
    }
  }
  const float sl2 = p.scale * 1.4426950408889634f;
  const unsigned kbase = lds0 + D5_K0 + (unsigned)(l15 * 1024 + ((g ^ l15) << 4));                      // K fragment, key tile 0, k-step 0
  const unsigned vbase = lds0 + D5_V0 + (unsigned)(((g * 8 + (l15 >> 2)) * (int)VLayout<512>::STR_DW + wave * DTW * 8 + (l15 & 3) * 2) * 4);   // V^T fragment, d tile 0
  const unsigned pw_base = lds0 + D5_P0 + (unsigned)((wave * 32 + l15) * D5_PSTR + g * 8);              // P write: row 32 wave + 16 j + l15, keys 16 t + 4 g ..
  const unsigned pr_base = lds0 + D5_P0 + (unsigned)(l15 * D5_PSTR + g * 16);                           // P read: row 16 j + l15, keys 8 g ..
  const unsigned al_w = lds0 + D5_AL + (unsigned)((wave * 32 + l15) * 4), al_r = lds0 + D5_AL + (unsigned)(l15 * 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the Q loads and tile 0: nothing of the compiler's is younger than the hand-counted pieces below

  // S^T = K Q^T of one tile: 32 keys x 32 queries of this wave, scaled to binades (s * scale * log2 e); a K fragment feeds both query tiles.
  // Eight batches of four fragments; batch b + 1 is read from LDS before the MFMAs of batch b are issued, and (PREFETCH) two of the next tile's
  // sixteen LDS-DMA pieces go out behind every batch instead of all in front of the tile's first MFMA.
  auto scores = [&](int kv0, int buf, bool tail, f32x4 (&sacc)[2][NT], auto prefetch) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t) sacc[j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned kb = kbase + (unsigned)buf * D5_KB;
    constexpr int NB = NT * KS / 4;
    f16x8 kf[2][4];
    auto issue_k = [&](auto bc) __attribute__((always_inline)) {
      constexpr int bb = decltype(bc)::value, t = bb / (KS / 4), k0 = (bb % (KS / 4)) * 4;
      attn_static_for<0, 4>([&](auto ic) { constexpr int i = decltype(ic)::value; attn_lds_read128<t * 16 * 1024>(kf[bb & 1][i], kb ^ (unsigned)((k0 + i) * 64)); });
    };
    issue_k(std::integral_constant<int, 0>{});
    attn_static_for<0, NB>([&](auto bc) {
      constexpr int bb = decltype(bc)::value, t = bb / (KS / 4), k0 = (bb % (KS / 4)) * 4;
      if constexpr (bb + 1 < NB) issue_k(std::integral_constant<int, bb + 1>{});
      attn_static_for<0, 4>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        attn_lds_wait<(bb + 1 < NB ? 4 : 0) + 3 - i>(kf[bb & 1][i]);
        sacc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[bb & 1][i], qf[0][k0 + i], sacc[0][t], 0, 0, 0);
        sacc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[bb & 1][i], qf[1][k0 + i], sacc[1][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
      prefetch(bc);
      __builtin_amdgcn_sched_barrier(0);
    });
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sacc[j][t][r] *= sl2;
          if (tail && kv0 + t * 16 + g * 4 + r >= p.Lk) sacc[j][t][r] = -1e30f;
        }
  };
  auto rowmax = [&](const f32x4 (&sa)[NT]) __attribute__((always_inline)) {   // over this lane's 8 keys of the tile
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < NT; ++t) mx = fmaxf(fmaxf(mx, fmaxf(sa[t][0], sa[t][1])), fmaxf(sa[t][2], sa[t][3]));
    return mx;
  };

  // ---- references: maximum over the first key tile + D5_LEAD ----
  float mref[2];
  {
    __builtin_amdgcn_s_barrier();   // (tile 0 landed: every wave waited for its pieces above)
    f32x4 sacc[2][NT];
    scores(0, 0, p.Lk < D5_BKV, sacc, [](auto) {});
#pragma unroll
    for (int j = 0; j < 2; ++j) mref[j] = xmax32(xmax16(rowmax(sacc[j]))) + D5_LEAD;
  }

  f32x4 oacc[NJ][DTW];   // [query tile][d tile of this wave's slice]
  float mobs[2], lsum[2];
  const int full = p.Lk / D5_BKV * D5_BKV;
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int dt = 0; dt < DTW; ++dt) oacc[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2; ++j) { mobs[j] = -1e30f; lsum[j] = 0.f; }

    auto tile = [&](int kv0, int buf, auto TAILC) __attribute__((always_inline)) {
      constexpr bool TAIL = decltype(TAILC)::value;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the tile
      __builtin_amdgcn_s_barrier();                      // every wave's pieces landed; every wave is done with the other buffer and with P
      const bool more = kv0 + D5_BKV < p.Lk;
      f32x4 sacc[2][NT];
      scores(kv0, buf, TAIL, sacc, [&](auto bc) __attribute__((always_inline)) { if (more) issue_row(kv0 + D5_BKV, buf ^ 1, bc); });
      // ---- p = 2^(s - m_ref) (lane: query 16 j + l15 of the wave, keys 16 t + 4 g + r), row sums per lane, P to LDS ----
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        mobs[j] = fmaxf(mobs[j], rowmax(sacc[j]));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f16x4 ph;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = __builtin_amdgcn_exp2f(sacc[j][t][r] - mref[j]);
            lsum[j] += pv;
            ph[r] = (f16)pv;
          }
          d5_lds_write64(pw_base + (unsigned)(j * 16 * D5_PSTR + t * 32), __builtin_bit_cast(uint2, ph));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();

      // ---- O^T[d slice of this wave][128 queries] += V^T P^T ----
      f16x8 pf[NJ];
      attn_static_for<0, NJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        attn_lds_read128<j * 16 * (int)D5_PSTR>(pf[j], pr_base);
      });
      const unsigned vb = vbase + (unsigned)buf * D5_VB;
      f16x4 vlo[2][2], vhi[2][2];   // [parity of the batch][d tile of the batch]: the next batch's reads are issued before this batch's MFMAs
      auto issue_v = [&](auto hc) __attribute__((always_inline)) {
        constexpr int hb = decltype(hc)::value;
        attn_static_for<0, 2>([&](auto dc) {
          constexpr int i = decltype(dc)::value, dt = hb * 2 + i;
          attn_lds_read_tr<dt * 32>(vlo[hb & 1][i], vb);                                          // keys 8g .. 8g+3
          attn_lds_read_tr<dt * 32 + 4 * (int)D5_VROW>(vhi[hb & 1][i], vb);                       // keys 8g+4 .. 8g+7
        });
      };
      issue_v(std::integral_constant<int, 0>{});
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(pf[0]), "+v"(pf[1]), "+v"(pf[2]), "+v"(pf[3]), "+v"(pf[4]), "+v"(pf[5]), "+v"(pf[6]), "+v"(pf[7]));
      attn_static_for<0, 4>([&](auto hc) {
        constexpr int hb = decltype(hc)::value;
        if constexpr (hb < 3) issue_v(std::integral_constant<int, hb + 1>{});
        attn_static_for<0, 2>([&](auto dc) {
          constexpr int i = decltype(dc)::value, dt = hb * 2 + i;
          attn_lds_wait<(hb < 3 ? 4 : 0) + 2 * (1 - i)>(vlo[hb & 1][i], vhi[hb & 1][i]);
          const f16x8 vf = {vlo[hb & 1][i][0], vlo[hb & 1][i][1], vlo[hb & 1][i][2], vlo[hb & 1][i][3], vhi[hb & 1][i][0], vhi[hb & 1][i][1], vhi[hb & 1][i][2], vhi[hb & 1][i][3]};
#pragma unroll
          for (int j = 0; j < NJ; ++j) oacc[j][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[j], oacc[j][dt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    };
    int buf = 0;
    for (int kv0 = 0; kv0 < full; kv0 += D5_BKV, buf ^= 1) tile(kv0, buf, std::false_type());
    if (full < p.Lk) tile(full, buf, std::true_type());

    // ---- did every score stay inside the window of its reference?  One word per wave, read by all ----
#pragma unroll
    for (int j = 0; j < 2; ++j) mobs[j] = xmax32(xmax16(mobs[j]));
    const bool left = mobs[0] - mref[0] > D5_WINDOW || mobs[1] - mref[1] > D5_WINDOW;
    __builtin_amdgcn_s_barrier();   // the last tile's P has been read
    const float flag = __builtin_amdgcn_ballot_w64(left) != 0 ? 1.0f : 0.0f;   // (all lanes vote: outside the branch below)
    if (lane == 0) d5_lds_write32(lds0 + D5_FL + (unsigned)(wave * 4), flag);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 fl;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(fl) : "v"(lds0 + D5_FL) : "memory");
    if (__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fl[0] + fl[1] + fl[2] + fl[3])) == 0) break;
    // second pass (never a third: the references are now the true maxima)
    mref[0] = mobs[0]; mref[1] = mobs[1];
    __builtin_amdgcn_s_barrier();   // (the flag words are rewritten at the end of the next pass)
    issue_tile(0, 0);
  }

  // ---- normalise and store: lane holds O[q = 16 j + l15][dd = 128 wave + 16 dt + 4 g + r] ----
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float l = lsum[j];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    d5_lds_write32(al_w + (unsigned)(j * 64), l);   // (the four g lanes of a query write the same value)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f16* Op = p.o + (long long)b * p.o_bstride + h * D;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int qi = q0 + j * 16 + l15;
    float sum;
    d5_lds_read32(sum, al_r + (unsigned)(j * 64));
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sum));
    const float inv = 1.0f / sum;
    if (qi < p.Lq) {
#pragma unroll
      for (int dt = 0; dt < DTW; ++dt) {
        const int dd = (wave * DTW + dt) * 16 + g * 4;
        const f16x4 o = {(f16)(oacc[j][dt][0] * inv), (f16)(oacc[j][dt][1] * inv), (f16)(oacc[j][dt][2] * inv), (f16)(oacc[j][dt][3] * inv)};
        *reinterpret_cast<f16x4*>(Op + (long long)qi * p.ldo + dd) = o;
      }
    }
  }
}

// LDIFF_ATTN_D512: 1 (default) = the 128-query kernel where d == 512 and a workgroup's queries are mostly real, 0 = the d-split kernel above
static bool attn_d512_selected(const AttnParams& p) {
  static const int mode = [] { const char* e = getenv("LDIFF_ATTN_D512"); return e ? atoi(e) : 1; }();
  if (!mode || p.d != 512 || p.prescaled) return false;
  if ((long long)(p.Lk - 1) * p.ldk * 2 + 1024 >= (1LL << 31) || (long long)(p.Lk - 1) * p.ldv * 2 + 1024 >= (1LL << 31)) return false;   // buffer descriptors
  return mode == 2 || p.Lq > 64;
}
static void launch_attn_d512(const AttnParams& p, hipStream_t s) {
  auto kern = attn_d512_kernel;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)D5_LDS);
  dim3 grid(((p.Lq + D5_QB - 1) / D5_QB) * p.heads * p.B);
  const double bh = (double)p.B * p.heads;
  ProfScope prof("attn<512,128q>", 4.0 * bh * p.Lq * p.Lk * p.d, 2.0 * bh * p.d * (2.0 * p.Lq + 2.0 * p.Lk * (p.kv_bstride ? 1.0 : 1.0 / p.B)), s);
  hipLaunchKernelGGL(kern, grid, dim3(256), D5_LDS, s, p);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------
// d = 40 self-attention over long sequences (UNet level 0: 4,096 tokens x 8 heads, the largest kernel of a UNet pass; 16,384 tokens at 1024^2).
// The generic kernel spends ~5.5 vector issue slots per score (FMA, exp, max chain, two converts and a pack, lane swaps, alpha) against 0.9 matrix
// cycles and is bound by them.  This one keeps the same tiling (128 queries per workgroup, 64-key tiles, S^T = K Q^T, P in registers) and
//   * fixes every query's softmax reference after the first key tile (its maximum + 4 binades, as attn_d512_kernel): no max chain, no lane
//     swaps, no alpha, no rescale in the loop.  The row sums come out of the P V MFMAs (V^T row 40 = 1: the lanes that hold channel 40 of a
//     V^T fragment substitute 1.0).  v_cvt_pkrtz CLAMPS an overflowing probability to the largest finite fp16 (round toward zero never
//     produces inf), so one clamped probability puts >= 65504 into its fp32 row sum: the end of the pass takes "row sum >= 65504" (or not
//     finite: exp2 itself overflowed) as the overflow signal -- conservative (a row whose in-window probabilities add up to that much also
//     repeats; both cases need > 2^20 times the weight of the first tile's best key in later tiles) -- and the (rare) workgroup that saw it
//     takes the true row maxima in a scores-only pass and repeats with those;
//   * packs P with v_cvt_pkrtz_f16_f32 (one instruction per two scores; numerator and row sum see the same rounded P);
//   * brings K / V in by LDS-DMA, double-buffered, ONE raw barrier per tile (zero padding of the 128-byte K rows and the channel-40..47 chunk of
//     V by the out-of-range sentinel), operand reads issued in batches with counted waits.
// ~3.5 issue slots per score.  Two workgroups per CU: one's softmax runs beside the other's MFMAs.
constexpr int FR_BKV = 64, FR_D = 40;
constexpr unsigned FR_KB = FR_BKV * 128, FR_VROW = 96, FR_VB = FR_BKV * FR_VROW, FR_K0 = 0, FR_V0 = 2 * FR_KB, FR_FL = FR_V0 + 2 * FR_VB, FR_LDS = FR_FL + 16;
constexpr float FR_LEAD = 4.0f;

__global__ __launch_bounds__(256, 2) void attn_fr40_kernel(const AttnParams p) {
  typedef __attribute__((address_space(3))) void lds_void;
  constexpr int QT = 2, KS = 2, NT = FR_BKV / 16, DT = 3;
  constexpr unsigned OOR = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l15 = lane & 15;
  int qtile = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  if (p.xcd_order) {   // (see attn_kernel)
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
    const int id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7, idx = id >> 3;
    int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    qtile = sw % gx; sw /= gx;
    h = sw % gy; b = sw / gy;
  }
  const int qbase = qtile * (64 * QT) + wave * (16 * QT);
  const f16* Qp = p.q + (long long)b * p.q_bstride + h * FR_D;
  const f16* Kp = p.k + (long long)b * p.kv_bstride + h * FR_D;
  const f16* Vp = p.v + (long long)b * p.kv_bstride + h * FR_D;
  const int ldk2 = p.ldk * 2, ldv2 = p.ldv * 2;
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)((long long)(p.Lk - 1) * ldk2 + FR_D * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)((long long)(p.Lk - 1) * ldv2 + FR_D * 2), 0x00020000);
  const unsigned lds0 = (unsigned)(size_t)(lds_void*)smem_raw;

  // K tile: 64 rows of 128 B (chunk c of row r at position c ^ ((r >> 1) & 7); chunks 5..7 zero) = 8 pieces of 8 rows, wave w: pieces 2w, 2w + 1.
  // V tile: 64 rows of 96 B (chunk 5 zero) = 6 pieces of 1 KiB, wave w: piece w and, for w < 2, piece 4 + w.  Per-lane byte offsets inside
  // the tile are fixed; the tile's first row goes into the scalar offset.
  int kvo[2], vvo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 2 + i) * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
    kvo[i] = c < 5 ? row * ldk2 + c * 16 : (int)OOR;
    const int idx = (i == 0 ? wave : 4 + wave) * 64 + lane, vr = idx / 6, vc = idx - vr * 6;
    vvo[i] = vc < 5 ? vr * ldv2 + vc * 16 : (int)OOR;
  }
  auto issue_tile = [&](int kv0, int buf) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk, (lds_void*)(smem_raw + FR_K0 + buf * FR_KB + (wave * 2) * 1024), 16, kvo[0], kv0 * ldk2, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsk, (lds_void*)(smem_raw + FR_K0 + buf * FR_KB + (wave * 2 + 1) * 1024), 16, kvo[1], kv0 * ldk2, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (lds_void*)(smem_raw + FR_V0 + buf * FR_VB + wave * 1024), 16, vvo[0], kv0 * ldv2, 0, 0);
    if (wave < 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsv, (lds_void*)(smem_raw + FR_V0 + buf * FR_VB + (4 + wave) * 1024), 16, vvo[1], kv0 * ldv2, 0, 0);
#endif
  };
  issue_tile(0, 0);

  f16x8 qf[QT][KS];
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = qbase + qt * 16 + l15;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int dd = ks * 32 + g * 8;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (qi < p.Lq && dd < FR_D) v = *reinterpret_cast<const uint4*>(Qp + (long long)qi * p.ldq + dd);
      qf[qt][ks] = __builtin_bit_cast(f16x8, v);
    }
  }
  const float sl2 = p.scale * 1.4426950408889634f;
  // K fragment of key tile t, k-step ks: row 16 t + l15, chunk 4 ks + g at position (4 ks + g) ^ ((row >> 1) & 7); (row >> 1) & 7 = (l15 >> 1) for every t
  const unsigned kb0 = lds0 + FR_K0 + (unsigned)(l15 * 128 + ((g ^ (l15 >> 1)) << 4));
  // V^T fragment (transposed read): lane 4 q + pp of its 16-lane group addresses key row 8 g' .. (see attn_kernel): rows 16 (2 s2) + 4 g + (l15 >> 2) (+ 16), columns 16 dt + 4 (l15 & 3)
  const unsigned vb0 = lds0 + FR_V0 + (unsigned)((g * 4 + (l15 >> 2)) * FR_VROW + (l15 & 3) * 8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  auto scores = [&](int buf, f32x4 (&sacc)[QT][NT]) __attribute__((always_inline)) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int t = 0; t < NT; ++t) sacc[qt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned kb = kb0 + (unsigned)buf * FR_KB;
    f16x8 kf[NT][KS];
    attn_static_for<0, NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      attn_lds_read128<t * 16 * 128>(kf[t][0], kb);
      attn_lds_read128<t * 16 * 128>(kf[t][1], kb ^ 64u);
    });
    attn_static_for<0, NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      attn_lds_wait<2 * (NT - 1 - t) + 1>(kf[t][0]);
      sacc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t][0], qf[0][0], sacc[0][t], 0, 0, 0);
      sacc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t][0], qf[1][0], sacc[1][t], 0, 0, 0);
      attn_lds_wait<2 * (NT - 1 - t)>(kf[t][1]);
      sacc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t][1], qf[0][1], sacc[0][t], 0, 0, 0);
      sacc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t][1], qf[1][1], sacc[1][t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto tile_max = [&](int kv0, bool tail, const f32x4 (&sa)[NT]) __attribute__((always_inline)) {   // over this lane's 16 keys of the tile, raw scores
    float mx = -1e30f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, (tail && kv0 + t * 16 + g * 4 + r >= p.Lk) ? -1e30f : sa[t][r]);
    return mx;
  };

  // ---- references (binades): maximum over the first key tile + FR_LEAD ----
  float mref[QT];
  {
    __builtin_amdgcn_s_barrier();
    f32x4 sacc[QT][NT];
    scores(0, sacc);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) mref[qt] = xmax32(xmax16(tile_max(0, p.Lk < FR_BKV, sacc[qt]))) * sl2 + FR_LEAD;
  }

  f32x4 oacc[QT][DT];
  const int full = p.Lk / FR_BKV * FR_BKV;
  const f16 one = (f16)1.0f;
  const f16x8 ones8 = {one, one, one, one, one, one, one, one};
  for (int pass = 0;; ++pass) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) oacc[qt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto tile = [&](int kv0, int buf, auto TAILC) __attribute__((always_inline)) {
      constexpr bool TAIL = decltype(TAILC)::value;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the tile
      __builtin_amdgcn_s_barrier();                      // every wave's pieces landed; every wave is done with the other buffer
      if (kv0 + FR_BKV < p.Lk) issue_tile(kv0 + FR_BKV, buf ^ 1);
      f32x4 sacc[QT][NT];
      scores(buf, sacc);
      // ---- p = 2^(s * scale * log2 e - reference); P^T fragments for the P V products straight from the score registers ----
      f16x8 pf[QT][NT / 2];
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) {
        const float moff = -mref[qt];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float e[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            e[r] = __builtin_amdgcn_exp2f(fmaf(sacc[qt][t][r], sl2, moff));
            if (TAIL && kv0 + t * 16 + g * 4 + r >= p.Lk) e[r] = 0.f;
          }
          typedef __fp16 h2_t __attribute__((ext_vector_type(2)));
          const h2_t a = __builtin_amdgcn_cvt_pkrtz(e[0], e[1]), c = __builtin_amdgcn_cvt_pkrtz(e[2], e[3]);
          pf[qt][t >> 1][(t & 1) * 4 + 0] = (f16)a[0]; pf[qt][t >> 1][(t & 1) * 4 + 1] = (f16)a[1];
          pf[qt][t >> 1][(t & 1) * 4 + 2] = (f16)c[0]; pf[qt][t >> 1][(t & 1) * 4 + 3] = (f16)c[1];
        }
      }
      // ---- O^T += V^T P^T; V^T row 40 (d tile 2, lane l15 = 8) is all ones: row 40 of O^T accumulates the row sums ----
      const unsigned vb = vb0 + (unsigned)buf * FR_VB;
      f16x4 vlo[NT / 2][DT], vhi[NT / 2][DT];
      attn_static_for<0, NT / 2>([&](auto sc) {
        constexpr int s2 = decltype(sc)::value;
        attn_static_for<0, DT>([&](auto dc) {
          constexpr int dt = decltype(dc)::value;
          attn_lds_read_tr<(2 * s2) * 16 * (int)FR_VROW + dt * 32>(vlo[s2][dt], vb);
          attn_lds_read_tr<(2 * s2 + 1) * 16 * (int)FR_VROW + dt * 32>(vhi[s2][dt], vb);
        });
      });
      attn_static_for<0, NT / 2>([&](auto sc) {
        constexpr int s2 = decltype(sc)::value;
        attn_static_for<0, DT>([&](auto dc) {
          constexpr int dt = decltype(dc)::value;
          attn_lds_wait<2 * ((NT / 2 - 1 - s2) * DT + (DT - 1 - dt))>(vlo[s2][dt], vhi[s2][dt]);
          f16x8 vf = {vlo[s2][dt][0], vlo[s2][dt][1], vlo[s2][dt][2], vlo[s2][dt][3], vhi[s2][dt][0], vhi[s2][dt][1], vhi[s2][dt][2], vhi[s2][dt][3]};
          if constexpr (dt == 2) vf = l15 == 8 ? ones8 : vf;
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) oacc[qt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[qt][s2], oacc[qt][dt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    };
    int buf = 0;
    for (int kv0 = 0; kv0 < full; kv0 += FR_BKV, buf ^= 1) tile(kv0, buf, std::false_type());
    if (full < p.Lk) tile(full, buf, std::true_type());

    // ---- no probability left fp16's range?  (row 40 of O^T: lane g = 2, register 0 of d tile 2; a clamped P alone contributes 65504) ----
    bool bad = false;
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) bad |= g == 2 && !(oacc[qt][2][0] < 65504.0f);
    const float flag = __builtin_amdgcn_ballot_w64(bad) != 0 ? 1.0f : 0.0f;
    __builtin_amdgcn_s_barrier();   // (all K / V reads of the pass are done)
    if (lane == 0) d5_lds_write32(lds0 + FR_FL + (unsigned)(wave * 4), flag);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    f32x4 fl;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(fl) : "v"(lds0 + FR_FL) : "memory");
    if (pass == 1 || __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, fl[0] + fl[1] + fl[2] + fl[3])) == 0) break;
    // ---- some probability overflowed fp16: the true row maxima in a scores-only pass, then once more (cannot fail) ----
    float mobs[QT] = {-1e30f, -1e30f};
    __builtin_amdgcn_s_barrier();
    issue_tile(0, 0);
    buf = 0;
    for (int kv0 = 0; kv0 < p.Lk; kv0 += FR_BKV, buf ^= 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kv0 + FR_BKV < p.Lk) issue_tile(kv0 + FR_BKV, buf ^ 1);
      f32x4 sacc[QT][NT];
      scores(buf, sacc);
#pragma unroll
      for (int qt = 0; qt < QT; ++qt) mobs[qt] = fmaxf(mobs[qt], tile_max(kv0, kv0 + FR_BKV > p.Lk, sacc[qt]));
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) mref[qt] = xmax32(xmax16(mobs[qt])) * sl2;
    __builtin_amdgcn_s_barrier();
    issue_tile(0, 0);
  }

  // ---- normalise and store: lane holds O[q = l15][dd = 16 dt + 4 g + r]; the row sum sits in row 40 = lane (g = 2, l15), register 0 of d tile 2 ----
  f16* Op = p.o + (long long)b * p.o_bstride + h * FR_D;
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const int qi = qbase + qt * 16 + l15;
    const float lsum = __shfl(oacc[qt][2][0], 32 + l15);
    const float inv = 1.0f / lsum;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int dd = dt * 16 + g * 4;
      if (qi < p.Lq && dd < FR_D) {
        const f16x4 o = {(f16)(oacc[qt][dt][0] * inv), (f16)(oacc[qt][dt][1] * inv), (f16)(oacc[qt][dt][2] * inv), (f16)(oacc[qt][dt][3] * inv)};
        *reinterpret_cast<f16x4*>(Op + (long long)qi * p.ldo + dd) = o;
      }
    }
  }
}

// LDIFF_ATTN_FIXREF: 1 (default) = d = 40 self-attention over at least two key tiles on the fixed-reference kernel, 0 = the generic kernel
static bool attn_fr40_selected(const AttnParams& p) {
  static const int mode = [] { const char* e = getenv("LDIFF_ATTN_FIXREF"); return e ? atoi(e) : 1; }();
  if (!mode || p.d != FR_D || p.prescaled || p.Lk < 2 * FR_BKV) return false;
  if ((long long)(p.Lk - 1) * p.ldk * 2 + 128 >= (1LL << 31) || (long long)(p.Lk - 1) * p.ldv * 2 + 128 >= (1LL << 31)) return false;
  return true;
}
static void launch_attn_fr40(const AttnParams& p, hipStream_t s) {
  auto kern = attn_fr40_kernel;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)FR_LDS);
  dim3 grid((p.Lq + 127) / 128, p.heads, p.B);
  const double bh = (double)p.B * p.heads;
  ProfScope prof("attn<40,fixref>", 4.0 * bh * p.Lq * p.Lk * p.d, 2.0 * bh * p.d * (2.0 * p.Lq + 2.0 * p.Lk * (p.kv_bstride ? 1.0 : 1.0 / p.B)), s);
  static const int xcd_mode = [] { const char* e = getenv("LDIFF_ATTN_XCD"); return e ? atoi(e) : 1; }();
  AttnParams q = p;
  q.xcd_order = xcd_mode && grid.x > 1 && p.Lk >= 256 ? 1 : 0;
  hipLaunchKernelGGL(kern, grid, dim3(256), FR_LDS, s, q);
  HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------------------------
// Short-K/V cross-attention: L_ctx <= 16 keys (the unpadded prompt "A pathological slide" is 5-6 tokens), all heads of a query row in
// ONE wave.  The launch is pure streaming of Q and O (K / V are a few KB); one workgroup per (image, head, query tile) -- the generic kernel --
// reads 80-byte pieces of the 640-byte query rows and pays a K / V tile staging plus two barriers per 64 queries.  Here
//   * K and V of ALL heads ([L_ctx][C] each, zero rows up to 16) sit in LDS once per workgroup;
//   * a wave owns 16 query rows and walks the heads: every Q fragment load of the row block is issued up front (whole rows end up in the wave's
//     hands back to back: each 128-byte line is fetched once), per head 2-5 MFMAs for S^T = K Q^T, a one-tile softmax (no running maximum), DT MFMAs
//     for O^T = V^T P^T with the key dimension zero-padded to 32;
//   * no barrier after the prologue.
// Head dims 40 / 80 / 160 (template D), heads * D = C.
constexpr int XRB = 1;   // 64-row blocks per workgroup of the short-K/V kernel (2: 15.4 us against 13.1 us at 8 x 4096 rows: more, smaller workgroups win)
template <int D>
__global__ __launch_bounds__(256, 2) void xattn_kernel(const AttnParams p) {
  constexpr int KS = (D + 31) / 32, DT = (D + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int C = p.heads * D;
  const int KSTR = C + 64;                     // row pitch (elements) of the K / V images: the last head's last k-step reads 24 columns past C (zeros)
  f16* sK = reinterpret_cast<f16*>(smem_raw);  // [16][KSTR]
  f16* sV = sK + 16 * KSTR;                    // [16][KSTR]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.y;
  const f16* Kp = p.k + (long long)b * p.kv_bstride;
  const f16* Vp = p.v + (long long)b * p.kv_bstride;
  const float sl2 = p.scale * 1.4426950408889634f;

  // A workgroup takes XRB blocks of 64 query rows; a wave's next (row block, head group) of Q fragments is in flight while it works on the current one.
  constexpr int HB = D <= 80 ? 8 : 4;          // heads whose Q fragments are loaded together (register budget)
  auto load_q = [&](uint4 (&qr)[HB][KS], int qrow, int h0) {
    const f16* Qp = p.q + (long long)b * p.q_bstride + (long long)(qrow < p.Lq ? qrow : p.Lq - 1) * p.ldq;
#pragma unroll
    for (int hh = 0; hh < HB; ++hh)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int dd = ks * 32 + g * 8;
        qr[hh][ks] = (h0 + hh < p.heads && dd < D) ? *reinterpret_cast<const uint4*>(Qp + (h0 + hh) * D + dd) : make_uint4(0, 0, 0, 0);
      }
  };
  const int nstep = XRB * ((p.heads + HB - 1) / HB);   // (row block, head group) pairs, row block outer
  auto step_rows = [&](int st) { return (blockIdx.x * XRB + st / ((p.heads + HB - 1) / HB)) * 64 + wave * 16 + l15; };
  auto step_h0 = [&](int st) { return (st % ((p.heads + HB - 1) / HB)) * HB; };
  uint4 qcur[HB][KS], qnxt[HB][KS];
  load_q(qcur, step_rows(0), step_h0(0));   // in flight under the K / V staging
  for (int i = tid; i < 16 * (KSTR / 8); i += 256) {   // 16-byte chunks; rows >= Lk and columns >= C are zero
    const int row = i / (KSTR / 8), ch = i - row * (KSTR / 8);
    uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
    if (row < p.Lk && ch * 8 < C) {
      kv = *reinterpret_cast<const uint4*>(Kp + (long long)row * p.ldk + ch * 8);
      vv = *reinterpret_cast<const uint4*>(Vp + (long long)row * p.ldv + ch * 8);
    }
    *reinterpret_cast<uint4*>(sK + row * KSTR + ch * 8) = kv;
    *reinterpret_cast<uint4*>(sV + row * KSTR + ch * 8) = vv;
  }
  __syncthreads();
  for (int st = 0; st < nstep; ++st) {
    const int qi = step_rows(st), h0 = step_h0(st);
    const bool qok = qi < p.Lq;
    if (st + 1 < nstep) load_q(qnxt, step_rows(st + 1), step_h0(st + 1));
    f16* Op = p.o + (long long)b * p.o_bstride + (long long)qi * p.ldo;
#pragma unroll
    for (int hh = 0; hh < HB; ++hh) {
      const int h = h0 + hh;
      if (h >= p.heads) break;
      // ---- S^T[key = l15][query] over the head's D channels: A = K rows from LDS, B = Q fragments ----
      f32x4 sacc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f16x8 kf = *reinterpret_cast<const f16x8*>(sK + l15 * KSTR + h * D + ks * 32 + g * 8);   // columns beyond the head meet zero Q elements
        sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, __builtin_bit_cast(f16x8, qcur[hh][ks]), sacc, 0, 0, 0);
      }
      // ---- softmax over the <= 16 keys of this lane's query (keys 4g + r; the four lanes l15 + 16 g hold one query) ----
      float mx = -1e30f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (g * 4 + r >= p.Lk) sacc[r] = -1e30f;
        mx = fmaxf(mx, sacc[r]);
      }
      mx = xmax32(xmax16(mx));
      const float moff = -mx * sl2;
      float rs = 0.f;
      f16x8 pf = {(f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};   // keys 16 .. 31 of the MFMA step: zero
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __builtin_amdgcn_exp2f(fmaf(sacc[r], sl2, moff));
        rs += pv;
        pf[r] = (f16)pv;
      }
      rs += __shfl_xor(rs, 16);
      rs += __shfl_xor(rs, 32);
      const float inv = 1.0f / rs;
      // ---- O^T[d][query] = V^T P^T: V^T fragments by transposed LDS reads (keys 4g .. 4g+3 of tile 0; the second half of the k-step is zero) ----
      const int tq = l15 >> 2, tp = l15 & 3;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sV + (g * 4 + tq) * KSTR + h * D + dt * 16 + tp * 4));
        const f16x4 lo_h = __builtin_bit_cast(f16x4, lo);
        const f16x8 vf = {lo_h[0], lo_h[1], lo_h[2], lo_h[3], (f16)0.f, (f16)0.f, (f16)0.f, (f16)0.f};
        f32x4 o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const int dd = dt * 16 + g * 4;
        if (qok && dd < D) {
          const f16x4 ov = {(f16)(o[0] * inv), (f16)(o[1] * inv), (f16)(o[2] * inv), (f16)(o[3] * inv)};
          *reinterpret_cast<f16x4*>(Op + h * D + dd) = ov;
        }
      }
    }
#pragma unroll
    for (int hh = 0; hh < HB; ++hh)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) qcur[hh][ks] = qnxt[hh][ks];
  }
}

template <int D>
static void launch_xattn(const AttnParams& p, hipStream_t s) {
  const int C = p.heads * D;
  const size_t smem = (size_t)2 * 16 * (C + 64) * sizeof(f16);
  auto kern = xattn_kernel<D>;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)smem);
  const double bh = (double)p.B * p.heads;
  ProfScope prof("xattn<short-kv>", 4.0 * bh * p.Lq * p.Lk * p.d, 2.0 * bh * p.d * (2.0 * p.Lq + 2.0 * p.Lk * (p.kv_bstride ? 1.0 : 1.0 / p.B)), s);
  hipLaunchKernelGGL(kern, dim3((p.Lq + 64 * XRB - 1) / (64 * XRB), p.B), dim3(256), smem, s, p);
  HIP_CHECK(hipGetLastError());
}
// LDIFF_XATTN: 1 (default) = cross-attention with L_ctx <= 16 on the short-K/V kernel, 0 = the generic kernel (A/B timing)
static bool xattn_selected(const AttnParams& p) {
  static const int mode = [] { const char* e = getenv("LDIFF_XATTN"); return e ? atoi(e) : 1; }();
  if (!mode || p.Lk > 16 || p.prescaled) return false;
  if (!(p.d == 40 || p.d == 80 || p.d == 160) || p.heads < 1 || p.heads > 16) return false;
  const long long C = (long long)p.heads * p.d;
  if (p.ldq < C || p.ldk < C || p.ldv < C || p.ldo < C) return false;           // the heads of a row must lie side by side
  if ((size_t)2 * 16 * (C + 64) * sizeof(f16) > 150 * 1024) return false;
  // enough workgroups to fill the chip twice: measured same box at B = 8 -- 4096 x 6, d = 40 (512 workgroups): 15.9 -> 13.1 us; 1024 x 6, d = 80 (128):
  // 8.1 -> 11.8 us and 256 x 6, d = 160 (32): 8.1 -> 18.0 us (too few workgroups for the K / V staging): the generic kernel keeps those (mode 2: always)
  return mode == 2 || (long long)p.B * ((p.Lq + 64 * XRB - 1) / (64 * XRB)) >= 384;
}

// head dims whose self-attention can take Q pre-multiplied by scale * log2(e) (AttnParams::prescaled)
bool attention_prescale_supported(int d) { return d == 40 || d == 80; }

void launch_attention(const AttnParams& p, hipStream_t s) {
  LDIFF_CHECK(p.d % 8 == 0 && p.d > 0 && p.d <= 512, LDIFF_ERR_INVALID, "attention: head dim %d must be a multiple of 8 and <= 512", p.d);
  LDIFF_CHECK(p.ldq % 8 == 0 && p.ldk % 8 == 0 && p.ldv % 8 == 0 && p.ldo % 4 == 0, LDIFF_ERR_INVALID, "attention: row strides must be multiples of 8");
  LDIFF_CHECK(p.Lk > 0 && p.Lq > 0, LDIFF_ERR_INVALID, "attention: empty sequence (Lq=%d Lk=%d)", p.Lq, p.Lk);
  const int d = p.d;
  if (xattn_selected(p)) {
    if (d == 40) launch_xattn<40>(p, s); else if (d == 80) launch_xattn<80>(p, s); else launch_xattn<160>(p, s);
    return;
  }
  if (attn_fr40_selected(p)) { launch_attn_fr40(p, s); return; }
  if (d <= 16) launch_attn_cfg<32, 16, 64, 2>(p, s);
  else if (d <= 32) launch_attn_cfg<32, 32, 64, 2>(p, s);
  else if (d <= 48) launch_attn_cfg<64, 48, 64, 2>(p, s);
  else if (d <= 64) launch_attn_cfg<64, 64, 64, 2>(p, s);
  else if (d <= 80) launch_attn_cfg<96, 80, 64, 2>(p, s);
  else if (d <= 96) launch_attn_cfg<96, 96, 64, 2>(p, s);
  else if (d <= 128) launch_attn_cfg<128, 128, 64, 2, 1>(p, s);
  else if (d <= 160) launch_attn_cfg<160, 160, 64, 2, 1>(p, s);
  else if (attn_d512_selected(p)) launch_attn_d512(p, s);
  else launch_attn_dsplit(p, s);
}
