// Shared declarations for the gfx950 (MI355X / CDNA4) kernels and the host-side executors.
// Everything here is internal to libldiff_hip.so; the public boundary is include/ldiff.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <vector>

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ---- status / error reporting (thread-local message; status codes from include/ldiff.h) ----------
#include "../../include/ldiff.h"
void ldiff_set_error(const char* fmt, ...);

struct LdiffError {
  int code;
};
#define LDIFF_CHECK(cond, code, ...)     \
  do {                                   \
    if (!(cond)) {                       \
      ldiff_set_error(__VA_ARGS__);      \
      throw LdiffError{code};            \
    }                                    \
  } while (0)
#define HIP_CHECK(expr)                                                                     \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      ldiff_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      throw LdiffError{LDIFF_ERR_RUNTIME};                                                  \
    }                                                                                       \
  } while (0)

// ---- implicit-GEMM convolution / linear (kernels_igemm.hip) ---------------------------------
// y[m, n] = sum_k A[m, k] * W[n, k]  (+ bias[n] + temb[b(m), n] + res[m, n])
//   A is the im2col view of one or two NHWC fp16 tensors (channel concat), optionally nearest-2x
//   upsampled, optionally with GroupNorm-apply (+SiLU) folded into the load:  a = silu(x*scale[b,c]+shift[b,c]).
//   k = (ky*ks + kx) * Cin + c,  Cin = C1 + C2.
struct ConvParams {
  const f16* x;      // [B, Hin, Win, C1]  (row pitch ld1 elements)
  const f16* x2;     // [B, Hin, Win, C2] or nullptr  (row pitch ld2)
  int C1, C2;        // multiples of 8
  int ld1, ld2;      // row pitch in elements of x / x2; 0 => C1 / C2.  A split tensor (see below) read as a plain operand has
                     // C1 = channels, ld1 = 2*channels (the kernel sees only the hi halves); read as a split operand it is a
                     // plain tensor of 2*channels whose weights were duplicated along K (Exec::conv, MatW::dup)
  int B, Hin, Win;   // source spatial size (before the optional 2x upsample)
  int Hout, Wout;
  int ks, stride, pad_t, pad_l, ups;
  const f16* w;      // [Nrows, K] K-major, Nrows = roundup(N,16), rows >= N zero
  int N, Nrows, K;   // N = columns stored (multiple of 4)
  int nt_hint = 0;   // dataflow conv: stream source, residual and output past L2 (set by its launcher)
  int c3d_ups = 0;   // upsampling conv on the dataflow kernel: 1 = wherever eligible (tests, timing); 0 / -1 = no (conv3x3d_selected: it does not pay in the step)
  int n_real = 0;    // output channels of the layer before that rounding (0: not stated; kernels that need it decline)
  const float* gn_scale;  // [B, Cin] or nullptr
  const float* gn_shift;  // [B, Cin]
  int silu_in;            // apply SiLU after the affine (only with gn_scale)
  const float* bias;      // [Nrows] or nullptr
  const float* temb;      // [B, ld_temb] (+ column offset applied by caller) or nullptr
  int ld_temb;
  const f16* res;         // [M, ld_res] or nullptr
  int ld_res;
  // Split tensors (fp16 pair hi | lo per row, value = hi + lo to ~22 bits): the residual stream of the UNet / VAE is kept this
  // way so that the reference's fp32 residual adds are reproduced to fp32 round-off while every MFMA operand stays fp16.
  int res_lo;             // > 0: the residual is split, its lo half starts res_lo elements after the hi half in each row
  int y_lo;               // > 0: write y split: hi = f16(v) at column n, lo = f16(v - hi) at column y_lo + n (fp16 output only)
  void* y;                // [M, ldy] fp16 or fp32
  int ldy;
  int out_f32;
  int M;                  // B*Hout*Wout
  float* stats;           // optional fused GroupNorm partial statistics of the output (see below)
  int stats_R;            // row blocks per image of `stats`
  long long w_bstride;    // gemm_dma only: > 0 => one weight matrix (and bias vector, bias_bstride) PER IMAGE: image b = row / (Hout*Wout)
  int bias_bstride;       //   (GroupNorm folded into a 1x1 conv: W_b = W diag(scale_b), bias_b = bias + W shift_b); tiles never straddle images
  int geglu;              // gemm_dma only: weight rows are x/gate-interleaved by 16 (MatW::geglu); y[m, n/2..] = x * gelu_erf(gate), ldy counts the N/2 outputs
  int splitk;             // conv3x3 / igemm: >1 => K (input-channel slabs) split over blockIdx.y, fp32 partials to splitk_ws
  float* splitk_ws;       // [splitk][M][N] fp32 workspace (then reduced + epilogue by splitk_reduce)
  const f16* w_par;       // conv3x3 with ups=1 only: parity weights [4][Nrows][4*Cin] (see kernels_conv3x3.hip); nullptr => 9-tap gather
  unsigned div_ntn, div_tx, div_ty;   // conv3x3 8x16 kernel only, set by its launcher: reciprocals of its tile decode (0 = divisor 1)
  // decode_latents tail fused into the epilogue (narrow-output kernel only, conv3x3n_selected): from the fp32 sums (x/2+0.5).clamp(0,1) -> post_img
  // [M,3] f32, (.*255).round() half-even -> post_rgb [M,3] u8, ITU-601 integer luma -> post_luma[b, post_slot, pixel] (kernels_elem.hip decode_post)
  float* post_img = nullptr; uint8_t* post_rgb = nullptr; uint8_t* post_luma = nullptr; int post_slots = 0, post_slot = 0, post_only = 0;   // post_only: y is not written
  int short_runs = 0;   // set by the executor of a graph that shares the chip with another stream (the VAE decoder beside the UNet): persistent kernels cap their run length
  unsigned div_tm; int tiles_m, img_fast;   // conv3x3 halo-tile kernels, set by their launchers: pixel tiles of the launch; tile order (see conv3x3_img_fast)
  const f16* w_frag;      // dataflow kernels only: the weights fragment-packed by launch_pack_frag_weights (conv3x3d_selected) / launch_pack_gemm_frag (gemm_df_selected)
  // Split operand with an fp8 lo half (16 x 16 ping-pong kernel only, conv3x3p_selected): a row of x is [C fp16 hi | C e4m3 lo * 2^LO8_SHIFT] = 3C bytes,
  // a row of w per tap [C fp16 | C e4m3 of w * 2^sw] (launch_lo8_weights); in 128-byte slabs: C / 64 fp16 slabs, then C / 128 fp8 slabs, so
  // C1 = 3C / 2 "elements" and K = taps * 3C / 2.  lo8_slab0 = C / 64 (first fp8 slab; 0 = no such operand), lo8_sa -> the E8M0 scale operand
  // 127 - sw written by launch_lo8_weights, lo8_sb = 127 - LO8_SHIFT.  The fp8 slabs go through v_mfma_scale_f32_16x16x128_f8f6f4.
  int lo8_slab0 = 0, lo8_sb = 0;
  const int* lo8_sa = nullptr;
  // conv3x3 dataflow kernel only: the 1x1 conv_shortcut of a ResnetBlock2D that changes width, folded into the block's second conv -- y = conv3x3(gn(x)) +
  // W_sc xs + (b + b_sc): the shortcut's Cs input channels are Cs / 64 extra slabs taken at the CENTRE tap only (raw operand, no GroupNorm), their
  // weights behind the nine taps in w_frag (launch_pack_frag_weights_sc), the two biases summed by the caller.  No residual then.
  const f16* xs = nullptr; int Cs = 0, lds = 0;
  int df_force = 0;   // dataflow GEMM (gemm_df_selected): 0 = by the unit list, -1 = never, 1 = wherever eligible, 16 mt + ntw = with that unit shape (ldiff_conv_args.gemm_df)
};
constexpr int LO8_SHIFT = 15;   // lo = x - fp16(x) of a GroupNorm + SiLU output: |lo| <= half an fp16 ulp = 2^-7 for |x| < 32, so lo * 2^15 <= 256 stays inside e4m3's 448;
                                // for |x| in [32, 64) it reaches 512 and saturates at 448 (the correction term is clamped, harmless), as for everything beyond
// [Nrows][taps][Cin] fp16 -> [Nrows][taps][Cin fp16 | Cin e4m3 of w * 2^sw], sw = floor(log2(448 / max |w|)); scale_out[0] = 127 - sw (one int)
void launch_lo8_weights(const f16* w, void* wd, int* scale_out, int Nrows, int taps, int Cin, hipStream_t s);
void launch_igemm(const ConvParams& p, hipStream_t s);   // dispatches to the halo-tile 3x3 kernel when eligible
bool conv3x3_eligible(const ConvParams& p);
int conv3x3_splitk_plan(const ConvParams& p);
int gemm_dma_splitk_plan(const ConvParams& p);           // ... and for the LDS-DMA GEMM (1x1 convs / linears with few tiles and a long K)
int igemm_splitk_plan(const ConvParams& p);              // same contract for the register-staged implicit GEMM (stride-2 convs with few tiles)
void launch_splitk_reduce(const ConvParams& p, hipStream_t s);   // sums p.splitk fp32 partials of splitk_ws and applies the epilogue (+ the fused GroupNorm statistics: R = H W / 32)
// Split count of a launch whose tiles do not fill the chip (round 6).  What bounds such a launch is not bytes: a workgroup takes its operand slices
// (16 KiB per K-step) at ~0.95 us per step whatever the ring depth or the slice layout, from HBM and L2 alike (scripts/micro/hbm_ring.hip,
// profiles/r06_hbm_ring.txt: the same 30 MB take 22 us on 80 workgroups, 13 on 160, 10.6 on 240, 8.9 on 480), so the lever is the NUMBER of workgroups --
// against the fp32 partials a finer cut writes and the reduce launch reads.  Returns the S in [s_old, s_cap] with the shortest modelled time
//     T(S) = 4 us + ceil(steps / S) * 0.95 us * max(1, tiles S / 330)  +  [S > 1] * (4 us + 2 * S * partial_bytes / 3 TB/s)
// and s_old (the rule of rounds 2-5, which the B = 8 shapes were tuned with) unless the model sees at least 15 % less.
inline int splitk_by_model(long long tiles, int steps, int min_steps, double partial_bytes, int s_old, int s_cap = 16) {
  auto T = [&](int S) {
    const double wgs = (double)tiles * S, per = (steps + S - 1) / S;
    return 4.0 + per * 0.95 * (wgs > 330.0 ? wgs / 330.0 : 1.0) + (S > 1 ? 4.0 + 2.0 * S * partial_bytes / 3.0e6 : 0.0);
  };
  if (s_old < 1) s_old = 1;
  int best = s_old;
  double tb = T(s_old);
  for (int S = s_old + 1; S <= s_cap && steps / S >= min_steps; ++S)
    if (T(S) < tb) { tb = T(S); best = S; }
  return tb <= 0.85 * T(s_old) ? best : s_old;
}
// nearest-2x upsample + conv3x3 == four 2x2 convs on the source grid (one per output parity) with pre-summed taps:
// w_par[q][n][t][c] from w[n][ky][kx][c]  (2.25x fewer MACs than gathering 9 taps from the upsampled image)
void launch_make_parity_weights(const f16* w, f16* w_par, int Nrows, int Cin, hipStream_t s);                  // 1 = no split; >1 needs splitk_ws of splitk*M*N floats
void launch_conv3x3(const ConvParams& p, hipStream_t s);       // kernels_conv3x3.hip
// 16x16-tile ping-pong variant for the maps that fill the chip (kernels_conv3x3p.hip); launch_conv3x3 dispatches to it
bool conv3x3n_selected(const ConvParams& p);   // narrow-output kernel (N == 4, plain epilogue, weights resident in LDS): kernels_conv3x3n.hip
void launch_conv3x3n(const ConvParams& p, hipStream_t s);
bool conv3x3p_selected(const ConvParams& p);
// producer / consumer ("dataflow") kernel for GroupNorm-prologue convs on the large maps: kernels_conv3x3d.hip
bool conv3x3d_selected(const ConvParams& p);
int conv3x3d_stats_blocks(const ConvParams& p);
size_t conv3x3d_frag_bytes(const ConvParams& p);
void launch_pack_frag_weights(const f16* w, f16* wf, int N, int Cin, hipStream_t s);   // [N][9 Cin] K-major -> MFMA A fragments, one KiB each
void launch_pack_frag_weights_par(const f16* wpar, f16* wf, int N, int Nrows, int Cin, hipStream_t s);   // ... of the parity-folded weights (ups = 1)
void launch_pack_frag_weights_sc(const f16* wsc, f16* wf, int N, int Cin, int Cs, int ld_wsc, hipStream_t s);   // the folded shortcut's [N][Cs] behind them (ConvParams::xs)
void launch_add_vectors(const float* a, const float* b, float* out, int n, hipStream_t s);   // out = a + b (either may be null = 0)
void launch_conv3x3d(const ConvParams& p, hipStream_t s);
int conv3x3p_stats_blocks(const ConvParams& p);
void launch_conv3x3p(const ConvParams& p, hipStream_t s);
// LayerNorm folded into the consuming GEMM, activations stationary in registers (kernels_gemm_ast.hip): y = LN(x) W^T + bias (optional GEGLU epilogue)
bool lngemm_eligible(int C, int N, int ldx, int x_lo, int ldy, bool geglu);
void launch_lngemm_tile_weights(const f16* w, f16* wt, int N, int C, hipStream_t s);   // [N][C] -> the panel images the kernel streams (N*C fp16)
void launch_lngemm(const f16* x, int ldx, int x_lo, int M, int C, const float* gamma, const float* beta, float eps, const f16* w_tiled, int N,
                   const float* bias, bool geglu, f16* y, int ldy, hipStream_t s, int qcols = 0, float qscale = 1.0f);   // columns [0, qcols) *= qscale before rounding
bool gemm_dma_eligible(const ConvParams& p);
// producer / consumer ("dataflow") GEMM for 1x1 convs / linears whose unit list fills the chip: kernels_gemm_df.hip
bool gemm_df_selected(const ConvParams& p);
size_t gemm_df_frag_bytes(const ConvParams& p);
void launch_pack_gemm_frag(const f16* w, f16* wf, int Nrows, int K, hipStream_t s);   // [Nrows][K] K-major -> MFMA A fragments, one KiB each
void launch_gemm_df(const ConvParams& p, hipStream_t s);   // needs ConvParams::w_frag
void launch_gemm_dma(const ConvParams& p, hipStream_t s);      // kernels_gemm.hip

// ---- attention (kernels_attn.hip) ------------------------------------------------------------
// O[b, q, h*d + :] = softmax(Q K^T / sqrt(d)) V   per (b, head).   All fp16, row strides in elements.
struct AttnParams {
  const f16* q; int ldq;      // [B, Lq, ldq], head h at column h*d
  const f16* k; int ldk;      // [Bk, Lk, ldk]
  const f16* v; int ldv;
  f16* o; int ldo;            // [B, Lq, ldo]
  int B, heads, Lq, Lk, d;
  long long q_bstride, kv_bstride, o_bstride;  // batch strides in elements (kv_bstride 0 => broadcast)
  float scale;
  int prescaled = 0;          // q already holds Q * scale * log2(e) (rounded once, by its producer): kernels_attn.hip PRE; `scale` is then unused
  float rescale_log2 = 0.0f;  // set by the launcher: online-softmax rescale threshold in log2 units (kernels_attn.hip; 0 = the maximum moves on every growth)
  int xcd_order = 0;          // set by the launcher: workgroup -> (query tile, head, image) through the XCD-aware remap (kernels_attn.hip)
};
void launch_attention(const AttnParams& p, hipStream_t s);
bool attention_prescale_supported(int d);

// ---- normalisation (kernels_norm.hip) --------------------------------------------------------
// GroupNorm statistics over one or two NHWC sources (channel concat) -> per-(b,c) scale/shift (fp32):
//   scale = rstd*gamma, shift = beta - mean*rstd*gamma   so that  gn(x) = x*scale + shift.
// A source is described by (pointer, channels C, row pitch ld, lo offset): lo > 0 => split tensor, value = x[c] + x[lo + c].
struct SrcView { const f16* p; int C, ld, lo; };
void launch_gn_stats(SrcView x1, SrcView x2 /* p == nullptr: none */, int B, int HW, int groups, float eps,
                     const float* gamma, const float* beta, float* partial /*workspace*/, size_t partial_bytes,
                     float* scale, float* shift, hipStream_t s, int* nonfinite = nullptr /* sticky flag of the owning handle: set when a total is not finite */);
size_t gn_partial_bytes(int B, int HW, int C);
void launch_layernorm(SrcView x, f16* y, int rows, const float* gamma, const float* beta, float eps, hipStream_t s);
// y[m, c] = act(x[m, c] * scale[b, c] + shift[b, c]) over the channel concat of one or two sources, written plain (y_lo = 0,
// ONE fp16 rounding of the fp32 result) or split (hi | lo).  GroupNorm-apply(+SiLU) as its own pass: used where the consumer is
// a split-operand contraction (the GEMM kernels take their operands by LDS-DMA and cannot transform them on the way).
void launch_norm_apply(SrcView x1, SrcView x2, int B, int HW, const float* scale, const float* shift, int silu, f16* y, int ldy, int y_lo, int lo8,
                       hipStream_t s);
// wd[n][tap][...] = { a(Ca), a(Ca), b(Cb), b(Cb), 0... } from w[n][tap][a(Ca) b(Cb) ...]: the weights of a contraction whose
// operand is a split tensor [hi | lo] (K doubled, same weights for both halves)
void launch_dup_weights(const f16* w, f16* wd, int Nrows, int taps, int src_tap_stride, int Ca, int Cb, int dst_tap_stride, hipStream_t s);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel): the attribute is per device
void ensure_dyn_smem(const void* kernel, int bytes);

// ---- elementwise / layout / sampler arithmetic (kernels_elem.hip) ----------------------------
// lo_off > 0: also write the rounding remainder f16(x - f16(x)) at channel lo_off + c (split input of the first conv)
void launch_nchw_f32_to_nhwc_f16(const float* x, f16* y, int B, int C, int H, int W, int Cpad, hipStream_t s, int lo_off = 0);
void launch_nhwc_f32_to_nchw_f32(const float* x, float* y, int B, int C, int H, int W, int ldx, hipStream_t s);
void launch_geglu(const f16* x, f16* y, long long M, int C4, hipStream_t s);  // x [M, 2*C4] -> y [M, C4]
void launch_timestep_embed(float t, const float* t_dev /* overrides t when not null */, f16* y, int B, int dim, int flip_sin_to_cos, float freq_shift,
                           hipStream_t s);
void launch_set_scalar(float* p, float v, hipStream_t s);
void launch_add_nchw_residual(f16* act, int ld, int lo, const float* res_nchw_f32, int B, int C, int HW, hipStream_t s);
bool prof_enabled();   // any per-launch profiling active (graph replay would hide the launches from it)
void launch_silu_f32_to_f16(const float* x, f16* y, long long n, hipStream_t s);
void launch_lincomb(const float* coef, const void* const* ops, int nops, float* out, long long n, hipStream_t s);
void launch_laplace_add(const float* z0, float scale, const float* u, unsigned long long seed, unsigned long long offset,
                        float* out, long long n, hipStream_t s);
void launch_scale_f32(const float* x, float* y, float a, long long n, hipStream_t s);
void launch_decode_post(const float* x, int ldx, int B, int H, int W, float* img_f32 /*[B,H,W,3] or null*/,
                        uint8_t* rgb_u8 /*[B,H,W,3] or null*/, uint8_t* luma /*base of [B,N,H,W] or null*/, int n_slots, int slot,
                        hipStream_t s);
void launch_argmax_u8(const float* logits, int B, int C, int H, int W, uint8_t* mask, hipStream_t s);
void launch_probe_argmax_u8(const uint8_t* feat, int B, int N, int H, int W, const float* w, const float* bias, float scale, int C, uint8_t* mask, hipStream_t s);
void launch_fold_gn_weights(const f16* w, const float* bias, const float* scale, const float* shift, f16* wb, float* biasb, int B, int Nrows, int C,
                            hipStream_t s);
void launch_bilinear_resize(const float* x, float* y, int B, int C, int H, int W, int oh, int ow, hipStream_t s);
void launch_luma_float(const float* rgb_nchw, float* gray, int B, int H, int W, hipStream_t s);
void launch_zero_bytes(void* p, size_t bytes, hipStream_t s);   // zero fill by a kernel (capturable; see kernels_elem.hip)
void launch_window_accumulate(void* acc, void* cnt, const void* pred, const void* g, int C, int H, int W, int th, int tw, int y0, int x0, int dtypes, hipStream_t s);

// ---- backward-pass primitives of the fine-tuning step (kernels_bwd.hip) -------------------------------------
void launch_im2col_t(const f16* x, f16* out, int B, int H, int W, int C, int ks, int stride, int pad, int ups, int Ho, int Wo, int Mpad, hipStream_t s);
void launch_transpose_rows(const f16* x, f16* out, int M, int N, int ldx, int Mpad, hipStream_t s);
void launch_colsum(const f16* dy, float* db, int M, int N, int ld, hipStream_t s);
void launch_gn_train_fwd(const f16* x, f16* y, const float* gamma, const float* beta, float* mean, float* rstd, int B, int HW, int C, int G, float eps,
                         int silu, hipStream_t s);
void launch_gn_train_bwd(const f16* x, const f16* dy, const float* gamma, const float* beta, const float* mean, const float* rstd, f16* dx, float* dgamma,
                         float* dbeta, int B, int HW, int C, int G, int silu, hipStream_t s);
void launch_ln_bwd(const f16* x, const f16* dy, const float* gamma, f16* dx, float* dgamma, float* dbeta, int rows, int C, float eps, hipStream_t s);
void launch_geglu_bwd(const f16* x, const f16* dy, f16* dx, long long M, int C4, hipStream_t s);
void launch_silu_f16(const f16* x, f16* y, long long n, hipStream_t s);                        // y = x sigma(x) on fp16 elements
void launch_silu_bwd_f16(const f16* x, const f16* dy, f16* dx, long long n, hipStream_t s);    // dx = dy sigma(x) (1 + x (1 - sigma(x)))
void launch_attn_bwd(const AttnParams& p, const f16* dO, f16* dq, f16* dk, f16* dv, hipStream_t s);
void launch_pack_weight(const float* w, f16* dst, int Cout, int Cin, int k, int R, int Cp, int mode, hipStream_t s);
void launch_unpack_wgrad(const float* g, float* dw, int Cout, int Cin, int k, int Cx, int ldg, hipStream_t s);
// multi-tensor AdamW (kernels_bwd.hip): device tables built by the host side (ldiffusion_amd/autograd.py)
#define ADAMW_CHUNK 16384
// one (weight tensor, layout) of ldiff_op_pack_weight_multi: mode 0 forward / 1 dgrad (kernels_bwd.hip), tiles_x = workgroup tiles per row of tiles
struct PackEntry { const float* w; f16* dst; int Cout, Cin, kk, R, Cp, mode, tiles_x, pad; };
void launch_pack_weight_multi(const PackEntry* entries, const int* tile_prefix, int n_entries, int n_tiles, hipStream_t s);
struct AdamTensor { float* p; float* m; float* v; long long n; };
struct AdamChunk { int tensor; int pad; long long first; };
void launch_adamw_multi(const AdamTensor* tensors, const float* const* grads, const AdamChunk* chunks, long long nchunks, float lr, float b1, float b2, float eps,
                        float wd, int step, hipStream_t s);
void launch_infonce(const float* feat, int B, int n, long long HW, const int* bi, const int* ai, const int* pi, const int* ni, int T, const int* t_dev, int K,
                    float temperature, float* loss, float* dfeat, hipStream_t s);   // contrastive loss over given sample triples + its gradient (kernels_bwd.hip)
void launch_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd, int step, hipStream_t s);

// ---- device arena: bump/free-list allocator over one hipMalloc'd slab ------------------------
// No hipMalloc/hipFree in a forward pass (graph-capturable, no implicit syncs).  Stream-ordered reuse:
// all kernels of one handle run on one stream, so a block may be reused as soon as it is released on the host.
class Arena {
 public:
  ~Arena();
  void reserve(size_t bytes);       // (re)allocate the slab if smaller; only outside a forward
  void* alloc(size_t bytes);
  void free(void* p);
  void reset();
  size_t capacity() const { return cap_; }
  size_t high_water() const { return high_; }

 private:
  struct Block { size_t off, size; bool used; };
  char* base_ = nullptr;
  size_t cap_ = 0, high_ = 0;
  std::vector<Block> blocks_;
};

// ---- optional per-launch profiling with HIP events (prof.hip) ---------------------------------
// When enabled (ldiff_prof_enable), every wrapped launch is bracketed by two events recorded on the
// launch stream; ldiff_prof_collect sums elapsed time / algorithmic flops / algorithmic bytes per kernel name.
bool prof_on(const char* name);
void prof_begin(const char* name, double flops, double bytes, hipStream_t s);
void prof_end(hipStream_t s);
struct ProfScope {
  hipStream_t s; bool on;
  ProfScope(const char* name, double flops, double bytes, hipStream_t st) : s(st), on(prof_on(name)) { if (on) prof_begin(name, flops, bytes, s); }
  ~ProfScope() { if (on) prof_end(s); }
};

// erf to 1.5e-7 absolute (Abramowitz & Stegun 7.1.26) with the two hardware transcendentals: the libm erff is ~30 instructions with
// branches, and the GEGLU epilogue evaluates it 32 times per thread and tile.  gelu_erf(g) = 0.5 g (1 + erf(g / sqrt 2)).
__device__ __forceinline__ float erf_as(float x) {
  const float ax = __builtin_fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  poly = __builtin_fmaf(poly, t, 1.421413741f);
  poly = __builtin_fmaf(poly, t, -0.284496736f);
  poly = __builtin_fmaf(poly, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(ax * ax * -1.4426950408889634f);
  const float y = __builtin_fmaf(-poly * t, e, 1.0f);
  return __builtin_copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float g) { return 0.5f * g * (1.0f + erf_as(g * 0.70710678118654752f)); }

// ---- GroupNorm statistics fused into the producing kernel's epilogue ------------------------------------------
// ConvParams::stats (optional): per row-block partial sums of the fp16-rounded outputs, CHANNEL-major so that the
// finalize kernel reads each (image, group) as one contiguous run:
//     stats[((b * N + n) * R + r) * 2 + {0,1}] = {sum, sum of squares} of channel n over row block r of image b
// Row blocks never straddle two images (R = ConvParams::stats_R blocks per image):
//   conv3x3   : r = ty*tiles_x + tx (8x16 tiles, both wave halves combined through LDS) or (ty*tiles_x + tx)*2 + wave_m (8x8 tiles)
//   gemm/igemm: r = (m % HW) / 32                      (32 consecutive rows; needs HW % 32 == 0)
// conv_stats_blocks_per_image() returns R for a launch (0 = not supported -> the caller falls back to the separate
// statistics pass of kernels_norm.hip).
int conv_stats_blocks_per_image(const ConvParams& p);
void launch_gn_finalize(const float* part1, int R1, int C1, const float* part2, int R2, int C2, int B, int HW, int groups, float eps,
                        const float* gamma, const float* beta, float* scale, float* shift, hipStream_t s, int* nonfinite = nullptr);

#ifdef __HIPCC__
// GroupNorm-apply (+SiLU) of TWO fp16 elements (one dword) -> one packed fp16 dword, six single-issue VALU instructions per element:
//   y = fma(x, scale, shift)  (v_fma_mix_f32 reads the fp16 half directly: no v_cvt),  r = 1 / (1 + 2^(-y log2 e)),  out = f16(y * r)
// (v_fma_mixlo/hi_f16: multiply and ONE rounding to fp16 in one instruction).  hipcc's own code for the same arithmetic packs the
// fp32 multiplies / fmas of neighbouring elements into v_pk_fma_f32 / v_pk_mul_f32, which cost ~5x a plain VALU instruction beside
// MFMAs (MI355X_MICROARCH.md, cycle constants: "an anti-lever beside MFMAs"; scripts/micro/conv_consumer.hip measures it: the same
// transform next to a matrix stream costs 22 % of the matrix rate in the compiler's form and 3 % in this one).  The two elements'
// chains are interleaved so that every transcendental result has one independent instruction before its first reader (gfx940+
// trans-forwarding hazard: nothing pads it inside an asm statement).
template <bool SILU>
__device__ __forceinline__ unsigned gn_pair(unsigned xw, float s0, float t0, float s1, float t1) {
  unsigned o;
  if constexpr (SILU) {
    float y0, y1, e0, e1;
    asm("v_fma_mix_f32 %1, %5, %6, %7 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %5, %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_mul_f32 %3, 0xbfb8aa3b, %1\n\t"
        "v_mul_f32 %4, 0xbfb8aa3b, %2\n\t"
        "v_exp_f32 %3, %3\n\t"
        "v_exp_f32 %4, %4\n\t"
        "v_add_f32 %3, 1.0, %3\n\t"
        "v_add_f32 %4, 1.0, %4\n\t"
        "v_rcp_f32 %3, %3\n\t"
        "v_rcp_f32 %4, %4\n\t"
        "v_fma_mixlo_f16 %0, %1, %3, 0\n\t"
        "v_fma_mixhi_f16 %0, %2, %4, 0"
        : "=&v"(o), "=&v"(y0), "=&v"(y1), "=&v"(e0), "=&v"(e1)
        : "v"(xw), "v"(s0), "v"(t0), "v"(s1), "v"(t1));
  } else {
    float y0, y1;
    asm("v_fma_mix_f32 %1, %3, %4, %5 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %2, %3, %6, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixlo_f16 %0, %1, 1.0, 0\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, 0"
        : "=&v"(o), "=&v"(y0), "=&v"(y1)
        : "v"(xw), "v"(s0), "v"(t0), "v"(s1), "v"(t1));
  }
  return o;
}
// The same for FOUR elements (two dwords) with the four chains interleaved: one wave alone issues a dependent chain at its latency, not its
// rate, and a producer wave of the dataflow conv kernel (kernels_conv3x3d.hip) has no second wave to fill the gaps.
template <bool SILU>
__device__ __forceinline__ void gn_quad(unsigned xa, unsigned xb, float s0, float t0, float s1, float t1, float s2, float t2, float s3, float t3, unsigned& oa, unsigned& ob) {
  if constexpr (SILU) {
    float y0, y1, y2, y3, e0, e1, e2, e3;
    asm("v_fma_mix_f32 %2, %10, %12, %13 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %3, %10, %14, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %4, %11, %16, %17 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mix_f32 %5, %11, %18, %19 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
        "v_mul_f32 %6, 0xbfb8aa3b, %2\n\t"
        "v_mul_f32 %7, 0xbfb8aa3b, %3\n\t"
        "v_mul_f32 %8, 0xbfb8aa3b, %4\n\t"
        "v_mul_f32 %9, 0xbfb8aa3b, %5\n\t"
        "v_exp_f32 %6, %6\n\t"
        "v_exp_f32 %7, %7\n\t"
        "v_exp_f32 %8, %8\n\t"
        "v_exp_f32 %9, %9\n\t"
        "v_add_f32 %6, 1.0, %6\n\t"
        "v_add_f32 %7, 1.0, %7\n\t"
        "v_add_f32 %8, 1.0, %8\n\t"
        "v_add_f32 %9, 1.0, %9\n\t"
        "v_rcp_f32 %6, %6\n\t"
        "v_rcp_f32 %7, %7\n\t"
        "v_rcp_f32 %8, %8\n\t"
        "v_rcp_f32 %9, %9\n\t"
        "v_fma_mixlo_f16 %0, %2, %6, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %7, 0\n\t"
        "v_fma_mixlo_f16 %1, %4, %8, 0\n\t"
        "v_fma_mixhi_f16 %1, %5, %9, 0"
        : "=&v"(oa), "=&v"(ob), "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3)
        : "v"(xa), "v"(xb), "v"(s0), "v"(t0), "v"(s1), "v"(t1), "v"(s2), "v"(t2), "v"(s3), "v"(t3));
  } else {
    oa = gn_pair<false>(xa, s0, t0, s1, t1);
    ob = gn_pair<false>(xb, s2, t2, s3, t3);
  }
}
template <int K>
__device__ __forceinline__ float comp8(const float4& a, const float4& b) {
  if constexpr (K == 0) return a.x; else if constexpr (K == 1) return a.y; else if constexpr (K == 2) return a.z; else if constexpr (K == 3) return a.w;
  else if constexpr (K == 4) return b.x; else if constexpr (K == 5) return b.y; else if constexpr (K == 6) return b.z; else return b.w;
}
template <int D>
__device__ __forceinline__ unsigned dword4(const uint4& v) {
  if constexpr (D == 0) return v.x; else if constexpr (D == 1) return v.y; else if constexpr (D == 2) return v.z; else return v.w;
}

// ---- epilogue helpers for split tensors (ConvParams::res_lo / y_lo) ----
__device__ __forceinline__ f16x4 cvt4(const f32x4& v) { return (f16x4){(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]}; }
__device__ __forceinline__ f32x4 up4(const f16x4& h) { return (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]}; }
// sum over the 16 lanes that share lane>>4 (one MFMA accumulator row group), by DPP (no LDS crossbar traffic)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}
// Sum NV values per lane over the 16 lanes of an MFMA row group, all at once: at every step a lane keeps half of its values, adds the
// partner lane's copies of them and hands the other half over (row_mirror, row_half_mirror, quad xor 2, quad xor 1: the partner
// differs in exactly the lane bit that selects).  NV - 1 adds instead of 4 NV, and the totals end up spread over the lanes (value
// index j in lane bits: bit0 = l15 & 8, bit1 = l15 & 4, bit2 = l15 & 2, bit3 = l15 & 1), so ONE store per lane writes them all.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true)); }
template <int NV>
__device__ __forceinline__ float row16_reduce_spread(float (&x)[NV], int l15) {
  static_assert(NV == 8 || NV == 16, "values per lane");
  int n = NV;
  auto step = [&](auto ctrlc, bool bit) __attribute__((always_inline)) {
    constexpr int CTRL = decltype(ctrlc)::value;
    if (n > 1) {
      n >>= 1;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (i < n) { const float keep = bit ? x[2 * i + 1] : x[2 * i], send = bit ? x[2 * i] : x[2 * i + 1]; x[i] = keep + dpp_f<CTRL>(send); }
    } else x[0] += dpp_f<CTRL>(x[0]);
  };
  step(std::integral_constant<int, 0x140>{}, (l15 & 8) != 0);
  step(std::integral_constant<int, 0x141>{}, (l15 & 4) != 0);
  step(std::integral_constant<int, 0x4E>{}, (l15 & 2) != 0);
  step(std::integral_constant<int, 0xB1>{}, (l15 & 1) != 0);
  return x[0];
}
// Per-wave reduction used by the epilogues: `o[a][m][r]` = final (fp16-rounded) value of pixel (m, l15), channel
// ncol + 16a + r; `ok[m]` = pixel valid.  Tiles m in [M0, M1) form one row block.  Lanes with l15 == 0 write 4 channels.
// dst = &stats[(b*N*R + rblk) * 2]; channel n lives at dst + n*R*2.
template <int MT, int NT>
__device__ __forceinline__ void wave_stats_store(const f32x4 (&o)[NT][MT], const bool (&ok)[MT], int M0, int M1, float* dst, long long R, int N,
                                                 int ncol, int l15) {
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m < M0 || m >= M1 || !ok[m]) continue;
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float v = o[a][m][r]; s[r] += v; q[r] = __builtin_fmaf(v, v, q[r]); }   // (explicit: kernels_gemm_df.hip sums in this association, bit for bit)
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { s[r] = row16_sum(s[r]); q[r] = row16_sum(q[r]); }
    const int n = ncol + a * 16;
    if (l15 == 0 && n < N) {
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<float2*>(dst + (long long)(n + r) * R * 2) = make_float2(s[r], q[r]);
    }
  }
}
#endif
