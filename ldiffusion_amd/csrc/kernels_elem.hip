// Elementwise / layout / sampler-arithmetic kernels for gfx950 (MI355X).  All HBM- or latency-bound.
// Each kernel names the reference expression it replaces (file:line into /root/reference).
#include "common.h"

static inline int nblocks(long long n, int per = 256) { return (int)((n + per - 1) / per); }

// ---- boundary layout: torch NCHW fp32 <-> internal NHWC fp16 ------------------------------------
__global__ void nchw_f32_to_nhwc_f16_kernel(const float* __restrict__ x, f16* __restrict__ y, int B, int C, int HW, int Cpad, int lo_off) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * HW) return;
  int b = (int)(i / HW), pix = (int)(i - (long long)b * HW);
  const float* src = x + (long long)b * C * HW + pix;
  f16* dst = y + i * Cpad;
  for (int c0 = 0; c0 < Cpad; c0 += 8) {
    f16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      float v = 0.f;
      if (c < C) v = src[(long long)c * HW];
      f16 r = (f16)v;
      if (lo_off > 0 && c >= lo_off && c - lo_off < C) {   // remainder of channel c - lo_off: x = hi + lo to ~22 bits
        const float f = src[(long long)(c - lo_off) * HW];
        r = (f16)(f - (float)(f16)f);
      }
      o[j] = r;
    }
    *reinterpret_cast<uint4*>(dst + c0) = __builtin_bit_cast(uint4, o);
  }
}
void launch_nchw_f32_to_nhwc_f16(const float* x, f16* y, int B, int C, int H, int W, int Cpad, hipStream_t s, int lo_off) {
  LDIFF_CHECK(Cpad % 8 == 0 && Cpad >= C && (lo_off == 0 || (lo_off >= C && lo_off + C <= Cpad)), LDIFF_ERR_INVALID,
              "layout: Cpad=%d must be a multiple of 8 and >= C=%d (split: 2C <= Cpad)", Cpad, C);
  long long n = (long long)B * H * W;
  if (n == 0) return;
  hipLaunchKernelGGL(nchw_f32_to_nhwc_f16_kernel, dim3(nblocks(n)), dim3(256), 0, s, x, y, B, C, H * W, Cpad, lo_off);
  HIP_CHECK(hipGetLastError());
}

// ---- act[b, pix, c] += r[b, c, pix]: ControlNet residual (fp32 NCHW, segmentor.py:366-372) added to an NHWC activation, plain or split ----
__global__ void add_nchw_residual_kernel(f16* __restrict__ a, int ld, int lo, const float* __restrict__ r, int B, int C, int HW) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * HW * C) return;
  const int c = (int)(i % C);
  const long long m = i / C;
  const int b = (int)(m / HW), pix = (int)(m - (long long)b * HW);
  f16* row = a + m * ld;
  float v = (float)row[c] + (lo ? (float)row[lo + c] : 0.f) + r[((long long)b * C + c) * HW + pix];
  const f16 hi = (f16)v;
  row[c] = hi;
  if (lo) row[lo + c] = (f16)(v - (float)hi);
}
void launch_add_nchw_residual(f16* act, int ld, int lo, const float* res_nchw, int B, int C, int HW, hipStream_t s) {
  const long long n = (long long)B * HW * C;
  if (n == 0) return;
  hipLaunchKernelGGL(add_nchw_residual_kernel, dim3(nblocks(n)), dim3(256), 0, s, act, ld, lo, res_nchw, B, C, HW);
  HIP_CHECK(hipGetLastError());
}

// ---- weights of a contraction over a split operand: K doubled, the same weights against the hi and the lo halves ----
__global__ void dup_weights_kernel(const f16* __restrict__ w, f16* __restrict__ wd, long long rows /* Nrows*taps */, int sstride, int Ca, int Cb,
                                   int dstride) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * dstride) return;
  const long long r = i / dstride;
  const int j = (int)(i - r * dstride);
  f16 v = (f16)0.f;
  if (j < 2 * Ca) v = w[r * sstride + (j < Ca ? j : j - Ca)];
  else if (j < 2 * Ca + 2 * Cb) { const int k = j - 2 * Ca; v = w[r * sstride + Ca + (k < Cb ? k : k - Cb)]; }
  wd[i] = v;
}
void launch_dup_weights(const f16* w, f16* wd, int Nrows, int taps, int src_tap_stride, int Ca, int Cb, int dst_tap_stride, hipStream_t s) {
  LDIFF_CHECK(Ca + Cb <= src_tap_stride && 2 * (Ca + Cb) <= dst_tap_stride, LDIFF_ERR_INVALID, "dup_weights: bad strides");
  const long long n = (long long)Nrows * taps * dst_tap_stride;
  hipLaunchKernelGGL(dup_weights_kernel, dim3(nblocks(n)), dim3(256), 0, s, w, wd, (long long)Nrows * taps, src_tap_stride, Ca, Cb, dst_tap_stride);
  HIP_CHECK(hipGetLastError());
}

// ---- weights of a split operand whose lo half is fp8 (ConvParams::lo8_slab0): per (row, tap) [Cin fp16 | Cin e4m3 of w * 2^sw] ----
__global__ __launch_bounds__(1024) void lo8_absmax_kernel(const f16* __restrict__ w, long long n, int* __restrict__ scale_out) {
  __shared__ float red[16];
  float m = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf((float)w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    int sw = 0;
    if (m > 0.f) {
      int e;
      frexpf(448.0f / m, &e);   // 448 / m = f * 2^e with f in [0.5, 1): floor(log2) = e - 1
      sw = e - 1;
      sw = sw < -100 ? -100 : (sw > 100 ? 100 : sw);
    }
    scale_out[0] = 127 - sw;
  }
}
__global__ void lo8_weights_kernel(const f16* __restrict__ w, unsigned char* __restrict__ wd, long long rows /* Nrows*taps */, int Cin, const int* __restrict__ scale) {
  const int cpr = Cin >> 3;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cpr) return;
  const long long r = i / cpr;
  const int c = (int)(i - r * cpr) * 8;
  const float sc = __builtin_ldexpf(1.0f, 127 - scale[0]);
  const uint4 v = *reinterpret_cast<const uint4*>(w + r * Cin + c);
  const f16x8 h = __builtin_bit_cast(f16x8, v);
  unsigned char* row = wd + r * (3LL * Cin);
  *reinterpret_cast<uint4*>(row + 2 * c) = v;
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = __builtin_amdgcn_fmed3f((float)h[j] * sc, -448.f, 448.f);
  int q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false), q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], 0, false);
  q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], q0, true); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], q1, true);
  *reinterpret_cast<int2*>(row + 2 * Cin + c) = make_int2(q0, q1);
}
__global__ void add_vectors_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (a ? a[i] : 0.f) + (b ? b[i] : 0.f);
}
void launch_add_vectors(const float* a, const float* b, float* out, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(add_vectors_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, out, n);
  HIP_CHECK(hipGetLastError());
}
void launch_lo8_weights(const f16* w, void* wd, int* scale_out, int Nrows, int taps, int Cin, hipStream_t s) {
  LDIFF_CHECK(w && wd && scale_out && Cin % 128 == 0 && Nrows > 0 && taps > 0, LDIFF_ERR_INVALID, "lo8_weights: Cin %d must be a multiple of 128", Cin);
  const long long rows = (long long)Nrows * taps, n = rows * (Cin >> 3);
  hipLaunchKernelGGL(lo8_absmax_kernel, dim3(1), dim3(1024), 0, s, w, rows * Cin, scale_out);
  hipLaunchKernelGGL(lo8_weights_kernel, dim3(nblocks(n)), dim3(256), 0, s, w, (unsigned char*)wd, rows, Cin, scale_out);
  HIP_CHECK(hipGetLastError());
}

__global__ void nhwc_f32_to_nchw_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW, int ldx) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * C * HW) return;
  int pix = (int)(i % HW);
  long long bc = i / HW;
  int c = (int)(bc % C), b = (int)(bc / C);
  y[i] = x[((long long)b * HW + pix) * ldx + c];
}
void launch_nhwc_f32_to_nchw_f32(const float* x, float* y, int B, int C, int H, int W, int ldx, hipStream_t s) {
  long long n = (long long)B * C * H * W;
  if (n == 0) return;
  hipLaunchKernelGGL(nhwc_f32_to_nchw_f32_kernel, dim3(nblocks(n)), dim3(256), 0, s, x, y, B, C, H * W, ldx);
  HIP_CHECK(hipGetLastError());
}

// ---- GEGLU: h * gelu_erf(gate)   (diffusers GEGLU inside FeedForward; BasicTransformerBlock.ff) ----
__global__ void geglu_kernel(const f16* __restrict__ x, f16* __restrict__ y, long long M, int C4) {
  const int cpr = C4 >> 3;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * cpr) return;
  long long m = i / cpr;
  int cc = (int)(i - m * cpr);
  const f16* row = x + m * 2 * C4;
  f16x8 hv = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(row + cc * 8));
  f16x8 gv = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(row + C4 + cc * 8));
  f16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    o[j] = (f16)((float)hv[j] * gelu_erf((float)gv[j]));
  }
  *reinterpret_cast<uint4*>(y + m * C4 + cc * 8) = __builtin_bit_cast(uint4, o);
}
void launch_geglu(const f16* x, f16* y, long long M, int C4, hipStream_t s) {
  LDIFF_CHECK(C4 % 8 == 0, LDIFF_ERR_INVALID, "geglu: inner dim %d must be a multiple of 8", C4);
  long long n = M * (C4 >> 3);
  if (n == 0) return;
  hipLaunchKernelGGL(geglu_kernel, dim3(nblocks(n)), dim3(256), 0, s, x, y, M, C4);
  HIP_CHECK(hipGetLastError());
}

// ---- sinusoidal timestep embedding (diffusers get_timestep_embedding) ------------------
__global__ void timestep_embed_kernel(float tval, const float* __restrict__ t_dev, f16* __restrict__ y, int B, int dim, int flip, float freq_shift) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= B * half) return;
  if (t_dev) tval = *t_dev;   // timestep from device memory: lets a captured graph of the forward be replayed for any t
  int b = i / half, k = i - b * half;
  float exponent = (-9.210340371976184f * (float)k) / ((float)half - freq_shift);  // -ln(10000) * k / (half - shift)
  float arg = tval * expf(exponent);
  float sn = sinf(arg), cs = cosf(arg);
  f16* row = y + (long long)b * dim;
  if (flip) { row[k] = (f16)cs; row[half + k] = (f16)sn; }
  else { row[k] = (f16)sn; row[half + k] = (f16)cs; }
}
void launch_timestep_embed(float t, const float* t_dev, f16* y, int B, int dim, int flip, float freq_shift, hipStream_t s) {
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(nblocks((long long)B * dim / 2)), dim3(256), 0, s, t, t_dev, y, B, dim, flip, freq_shift);
  HIP_CHECK(hipGetLastError());
}
__global__ void set_scalar_kernel(float* p, float v) { *p = v; }
void launch_set_scalar(float* p, float v, hipStream_t s) {
  hipLaunchKernelGGL(set_scalar_kernel, dim3(1), dim3(1), 0, s, p, v);
  HIP_CHECK(hipGetLastError());
}

__global__ void silu_f32_to_f16_kernel(const float* __restrict__ x, f16* __restrict__ y, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  y[i] = (f16)(v / (1.0f + expf(-v)));
}
void launch_silu_f32_to_f16(const float* x, f16* y, long long n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(silu_f32_to_f16_kernel, dim3(nblocks(n)), dim3(256), 0, s, x, y, n);
  HIP_CHECK(hipGetLastError());
}

// ---- PNDM/PLMS update as one linear combination (scheduler.step; segmentor.py:104, pixel_latent_vector.py:79) ----
struct LinComb { float c[6]; const float* p[6]; int n; };
__global__ void lincomb_kernel(LinComb a, float* __restrict__ out, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (k < a.n) v += a.c[k] * a.p[k][i];
  out[i] = v;
}
void launch_lincomb(const float* coef, const void* const* ops, int nops, float* out, long long n, hipStream_t s) {
  LDIFF_CHECK(nops >= 1 && nops <= 6, LDIFF_ERR_INVALID, "lincomb: nops=%d out of range [1,6]", nops);
  if (n == 0) return;
  LinComb a;
  a.n = nops;
  for (int k = 0; k < 6; ++k) { a.c[k] = k < nops ? coef[k] : 0.f; a.p[k] = k < nops ? (const float*)ops[k] : nullptr; }
  hipLaunchKernelGGL(lincomb_kernel, dim3(nblocks(n)), dim3(256), 0, s, a, out, n);
  HIP_CHECK(hipGetLastError());
}

// ---- Laplace forward noise (ldiffusion.py:234-237; torch.distributions.Laplace.rsample) ----
// x = z0 - scale * sign(u) * log1p(-|u|),  u ~ U(eps_f32 - 1, 1).  `u` may be supplied (parity is defined
// given u) or drawn from a counter-based Philox4x32-10 stream keyed by (seed, offset + element index / 4).
__device__ __forceinline__ void philox4x32_10(unsigned long long ctr, unsigned long long key, unsigned out[4]) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0, c3 = 0;
  unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ void laplace_add_kernel(const float* __restrict__ z0, float scale, const float* __restrict__ u, unsigned long long seed,
                                   unsigned long long offset, float* __restrict__ out, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float uv;
  if (u) uv = u[i];
  else {
    unsigned r[4];
    philox4x32_10(offset + (unsigned long long)(i >> 2), seed, r);
    float f = (float)(r[i & 3] >> 8) * (1.0f / 16777216.0f);           // [0,1) with 24 bits
    const float lo = 1.1920928955078125e-07f - 1.0f;                   // finfo(float32).eps - 1
    uv = lo + f * (1.0f - lo);
  }
  float sgn = (uv > 0.f) ? 1.f : ((uv < 0.f) ? -1.f : 0.f);
  out[i] = z0[i] + (0.f - scale * sgn * log1pf(-fabsf(uv)));
}
void launch_laplace_add(const float* z0, float scale, const float* u, unsigned long long seed, unsigned long long offset, float* out,
                        long long n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(laplace_add_kernel, dim3(nblocks(n)), dim3(256), 0, s, z0, scale, u, seed, offset, out, n);
  HIP_CHECK(hipGetLastError());
}

__global__ void scale_f32_kernel(const float* __restrict__ x, float* __restrict__ y, float a, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i];
}
void launch_scale_f32(const float* x, float* y, float a, long long n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(scale_f32_kernel, dim3(nblocks(n)), dim3(256), 0, s, x, y, a, n);
  HIP_CHECK(hipGetLastError());
}

// ---- decode post-process + uint8 + luma feature slot --------------------------------------------
// decode_latents tail  (x/2+0.5).clamp(0,1)                       -> img_f32 [B,H,W,3]   (segmentor.py:106)
// numpy_to_pil         (img*255).round().astype(uint8), half-even -> rgb_u8  [B,H,W,3]   (segmentor.py:107)
// PIL convert("L")     (19595R + 38470G + 7471B + 0x8000) >> 16   -> luma[b, slot, h, w] (pixel_latent_vector.py:85)
__global__ void decode_post_kernel(const float* __restrict__ x, int ldx, long long npix, int HW, float* __restrict__ img,
                                   uint8_t* __restrict__ rgb, uint8_t* __restrict__ luma, int n_slots, int slot) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const float* px = x + i * ldx;
  unsigned q[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = __fadd_rn(__fmul_rn(px[c], 0.5f), 0.5f);
    v = fminf(fmaxf(v, 0.f), 1.f);
    if (px[c] != px[c]) v = px[c];  // clamp propagates NaN in torch
    if (img) img[i * 3 + c] = v;
    q[c] = (unsigned)(int)rintf(__fmul_rn(v, 255.0f));
  }
  if (rgb) { rgb[i * 3 + 0] = (uint8_t)q[0]; rgb[i * 3 + 1] = (uint8_t)q[1]; rgb[i * 3 + 2] = (uint8_t)q[2]; }
  if (luma) {
    long long b = i / HW, pix = i - b * HW;
    luma[(b * n_slots + slot) * HW + pix] = (uint8_t)((19595u * q[0] + 38470u * q[1] + 7471u * q[2] + 0x8000u) >> 16);
  }
}
void launch_decode_post(const float* x, int ldx, int B, int H, int W, float* img_f32, uint8_t* rgb_u8, uint8_t* luma, int n_slots, int slot,
                        hipStream_t s) {
  LDIFF_CHECK(!luma || (slot >= 0 && slot < n_slots), LDIFF_ERR_INVALID, "decode_post: slot %d out of range [0,%d)", slot, n_slots);
  long long n = (long long)B * H * W;
  if (n == 0) return;
  hipLaunchKernelGGL(decode_post_kernel, dim3(nblocks(n)), dim3(256), 0, s, x, ldx, n, H * W, img_f32, rgb_u8, luma, n_slots, slot);
  HIP_CHECK(hipGetLastError());
}

// ---- zero fill as a KERNEL: hipMemsetAsync nodes inside a captured graph were not ordered with the kernels behind them on this runtime
// (seen in train.GraphedStep: garbage on every second of two back-to-back replays), so nothing that may be captured uses memset ----
__global__ void zero_bytes_kernel(unsigned* __restrict__ p, long long words) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < words) p[i] = 0u;
}
void launch_zero_bytes(void* p, size_t bytes, hipStream_t s) {
  LDIFF_CHECK(bytes % 4 == 0 && ((size_t)p & 3) == 0, LDIFF_ERR_INVALID, "zero fill: 4-byte granularity");
  if (bytes == 0) return;
  hipLaunchKernelGGL(zero_bytes_kernel, dim3(nblocks((long long)(bytes / 4))), dim3(256), 0, s, (unsigned*)p, (long long)(bytes / 4));
  HIP_CHECK(hipGetLastError());
}

// ---- diagnostic: max |value| of an activation (Exec::trace, LDIFF_TRACE_ABSMAX=1); split tensors as hi + lo ----
__global__ __launch_bounds__(256) void absmax_kernel(const f16* __restrict__ x, long long rows, int C, int ld, int lo, float* __restrict__ out) {
  const long long n = rows * C;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / C;
    const int c = (int)(i - r * C);
    float v = (float)x[r * ld + c];
    if (lo) v += (float)x[r * ld + lo + c];
    v = __builtin_fabsf(v);
    m = (v > m || v != v) ? v : m;   // NaN sticks
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(m, o); m = (t > m || t != t) ? t : m; }
  if ((threadIdx.x & 63) == 0) {
    if (m != m) atomicExch(reinterpret_cast<unsigned*>(out), 0x7fc00000u);
    else atomicMax(reinterpret_cast<unsigned*>(out), __builtin_bit_cast(unsigned, m));   // non-negative floats order like their bit patterns
  }
}
void launch_absmax(const f16* x, long long rows, int C, int ld, int lo, float* out, hipStream_t s) {
  launch_zero_bytes(out, 4, s);
  if (rows * C == 0) return;
  hipLaunchKernelGGL(absmax_kernel, dim3(1024), dim3(256), 0, s, x, rows, C, ld ? ld : C, lo, out);
  HIP_CHECK(hipGetLastError());
}

// ---- mask tail: argmax over classes (segmentor.py:536; softmax is monotone) ----------------------
__global__ void argmax_u8_kernel(const float* __restrict__ logits, int B, int C, int HW, uint8_t* __restrict__ mask) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * HW) return;
  int b = (int)(i / HW), pix = (int)(i - (long long)b * HW);
  const float* p = logits + (long long)b * C * HW + pix;
  float best = p[0];
  int bi = 0;
  bool any_nan = best != best || best == INFINITY;
  for (int c = 1; c < C; ++c) {
    float v = p[(long long)c * HW];
    any_nan |= v != v || v == INFINITY;
    if (v > best) { best = v; bi = c; }  // first maximal element, like torch.argmax
  }
  // softmax turns a pixel with any NaN or +inf logit into all-NaN; torch.argmax of an all-NaN row is index 0
  mask[i] = any_nan ? (uint8_t)0 : (uint8_t)bi;
}
void launch_argmax_u8(const float* logits, int B, int C, int H, int W, uint8_t* mask, hipStream_t s) {
  LDIFF_CHECK(C >= 1 && C <= 256, LDIFF_ERR_INVALID, "argmax: class count %d out of range [1,256]", C);
  long long n = (long long)B * H * W;
  if (n == 0) return;
  hipLaunchKernelGGL(argmax_u8_kernel, dim3(nblocks(n)), dim3(256), 0, s, logits, B, C, H * W, mask);
  HIP_CHECK(hipGetLastError());
}

// ---- mask tail over the per-pixel latent vectors: linear probe + argmax in ONE launch ------------------------------------------------
// features u8 [B,N,H,W] (pixel_latent_vector.py:85-93: the N luma planes of a pixel are its latent vector) -> logits_c = bias_c +
// sum_n w[c,n] * (f_n * scale) -> first maximal class (segmentor.py:536-537).  The arithmetic is fixed so that a host statement can
// reproduce it bit for bit: x_n = fl(float(f_n) * scale); acc = bias_c; acc = fl(acc + fl(w[c,n] * x_n)) for n = 0..N-1 (no fused
// multiply-add).  Four pixels per thread (one dword of every plane), weights in LDS.
constexpr int PROBE_MAX_C = 32, PROBE_MAX_N = 64;
__global__ __launch_bounds__(256) void probe_argmax_u8_kernel(const uint8_t* __restrict__ feat, int B, int N, long long HW, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float scale, int C, uint8_t* __restrict__ mask) {
  __shared__ float sw[PROBE_MAX_C * PROBE_MAX_N + PROBE_MAX_C];
  for (int i = threadIdx.x; i < C * N; i += blockDim.x) sw[i] = w[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) sw[PROBE_MAX_C * PROBE_MAX_N + i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x, quads = HW / 4;   // HW % 4 == 0 (checked by the launcher)
  if (q >= (long long)B * quads) return;
  const int b = (int)(q / quads);
  const long long pix = (q - (long long)b * quads) * 4;
  const uint8_t* f = feat + (long long)b * N * HW + pix;
  float best[4];
  int bi[4] = {0, 0, 0, 0};
  for (int c = 0; c < C; ++c) {
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = sw[PROBE_MAX_C * PROBE_MAX_N + c];
    for (int n = 0; n < N; ++n) {
      const unsigned v = *reinterpret_cast<const unsigned*>(f + (long long)n * HW);   // L1/L2 resident after class 0
      const float wv = sw[c * N + n];
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __fadd_rn(acc[j], __fmul_rn(wv, __fmul_rn((float)((v >> (8 * j)) & 255u), scale)));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c == 0 || acc[j] > best[j]) { best[j] = acc[j]; bi[j] = c; }   // first maximal class, like torch.argmax (finite logits by construction)
  }
  *reinterpret_cast<unsigned*>(mask + (long long)b * HW + pix) = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
}
void launch_probe_argmax_u8(const uint8_t* feat, int B, int N, int H, int W, const float* w, const float* bias, float scale, int C, uint8_t* mask, hipStream_t s) {
  LDIFF_CHECK(C >= 1 && C <= PROBE_MAX_C && N >= 1 && N <= PROBE_MAX_N, LDIFF_ERR_INVALID, "probe_argmax: %d classes x %d planes out of range [1,%d] x [1,%d]", C, N,
              PROBE_MAX_C, PROBE_MAX_N);
  const long long HW = (long long)H * W;
  LDIFF_CHECK(HW % 4 == 0 && ((size_t)feat & 3) == 0 && ((size_t)mask & 3) == 0, LDIFF_ERR_INVALID, "probe_argmax: H*W must be a multiple of 4 and the buffers 4-byte aligned");
  const long long n = (long long)B * (HW / 4);
  if (n == 0) return;
  hipLaunchKernelGGL(probe_argmax_u8_kernel, dim3(nblocks(n)), dim3(256), 0, s, feat, B, N, HW, w, bias, scale, C, mask);
  HIP_CHECK(hipGetLastError());
}

// ---- sliding-window / tile merge: logits[:, window] += pred * g;  n[window] += g  --------------------------------------------
// (nnU-Net's predictor, model/nnunetv2/inference/predict_from_raw_data.py:563-570 as called from /root/reference/segmentor.py:388-488,
// and the sampler's own tile merge of BASELINE configs[3]).  The arithmetic is the tensor formulation's, rounding for rounding: the
// product is rounded to the storage type, then the sum (no fused multiply-add), so that float16 accumulators reproduce the reference's.
template <typename T, typename P>
__global__ void window_accumulate_kernel(T* __restrict__ acc, T* __restrict__ cnt, const P* __restrict__ pred, const T* __restrict__ g, int C, int H, int W,
                                         int th, int tw, int y0, int x0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= th * tw) return;
  const int ty = i / tw, tx = i - ty * tw;
  const long long o = (long long)(y0 + ty) * W + x0 + tx;
  const float gv = g ? (float)g[i] : 1.0f;
  for (int c = 0; c < C; ++c) {
    const float pv = (float)pred[(long long)c * th * tw + i];
    // the product has the promoted type of (prediction, weight): rounded to float16 only when both are float16
    const float prod = g ? (float)(P)__fmul_rn(pv, gv) : pv;
    acc[(long long)c * H * W + o] = (T)__fadd_rn((float)acc[(long long)c * H * W + o], prod);
  }
  cnt[o] = (T)__fadd_rn((float)cnt[o], gv);
}
void launch_window_accumulate(void* acc, void* cnt, const void* pred, const void* g, int C, int H, int W, int th, int tw, int y0, int x0, int dtypes, hipStream_t s) {
  LDIFF_CHECK(C >= 1 && th >= 1 && tw >= 1 && y0 >= 0 && x0 >= 0 && y0 + th <= H && x0 + tw <= W, LDIFF_ERR_INVALID, "window_accumulate: the window must lie inside the image");
  LDIFF_CHECK(dtypes == 0 || dtypes == 1 || dtypes == 3, LDIFF_ERR_INVALID, "window_accumulate: dtypes 0 (all float32), 1 (all float16) or 3 (float16 accumulators, float32 prediction)");
  const dim3 grid(nblocks((long long)th * tw)), block(256);
  if (dtypes == 1) hipLaunchKernelGGL((window_accumulate_kernel<f16, f16>), grid, block, 0, s, (f16*)acc, (f16*)cnt, (const f16*)pred, (const f16*)g, C, H, W, th, tw, y0, x0);
  else if (dtypes == 3) hipLaunchKernelGGL((window_accumulate_kernel<f16, float>), grid, block, 0, s, (f16*)acc, (f16*)cnt, (const float*)pred, (const f16*)g, C, H, W, th, tw, y0, x0);
  else hipLaunchKernelGGL((window_accumulate_kernel<float, float>), grid, block, 0, s, (float*)acc, (float*)cnt, (const float*)pred, (const float*)g, C, H, W, th, tw, y0, x0);
  HIP_CHECK(hipGetLastError());
}

// ---- float luma of the training-time features (ldiffusion.py:241-242) ---------------------------
__global__ void luma_float_kernel(const float* __restrict__ rgb, float* __restrict__ gray, int B, int HW) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * HW) return;
  int b = (int)(i / HW), pix = (int)(i - (long long)b * HW);
  const float* p = rgb + (long long)b * 3 * HW + pix;
  float acc = __fmul_rn(p[0], 0.2989f);
  acc = __fadd_rn(acc, __fmul_rn(p[HW], 0.5870f));
  acc = __fadd_rn(acc, __fmul_rn(p[2LL * HW], 0.1140f));
  gray[i] = acc;
}
void launch_luma_float(const float* rgb_nchw, float* gray, int B, int H, int W, hipStream_t s) {
  long long n = (long long)B * H * W;
  if (n == 0) return;
  hipLaunchKernelGGL(luma_float_kernel, dim3(nblocks(n)), dim3(256), 0, s, rgb_nchw, gray, B, H * W);
  HIP_CHECK(hipGetLastError());
}


// ---- bilinear resize, align_corners = False (F.interpolate(mode="bilinear"), /root/reference/ldiffusion.py:240,250) ----
// fp32 NCHW -> fp32 NCHW; source index = (dst + 0.5) * in/out - 0.5 clamped at 0, neighbours clamped at in-1, no antialiasing
// (exactly ATen's upsample_bilinear2d: at 512 -> 64 it samples a 2x2 neighbourhood every 8 pixels).
__global__ void bilinear_resize_kernel(const float* __restrict__ x, float* __restrict__ y, long long planes, int H, int W, int oh, int ow,
                                       float sy, float sx) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = planes * oh * ow;
  if (i >= total) return;
  const int ox = (int)(i % ow), oy = (int)((i / ow) % oh);
  const long long pl = i / ((long long)ow * oh);
  float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
  fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
  const float* p = x + pl * H * W;
  y[i] = hy * (hx * p[(long long)y0 * W + x0] + lx * p[(long long)y0 * W + x1]) + ly * (hx * p[(long long)y1 * W + x0] + lx * p[(long long)y1 * W + x1]);
}
void launch_bilinear_resize(const float* x, float* y, int B, int C, int H, int W, int oh, int ow, hipStream_t s) {
  const long long total = (long long)B * C * oh * ow;
  if (total == 0) return;
  hipLaunchKernelGGL(bilinear_resize_kernel, dim3(nblocks(total)), dim3(256), 0, s, x, y, (long long)B * C, H, W, oh, ow, (float)H / (float)oh,
                     (float)W / (float)ow);
  HIP_CHECK(hipGetLastError());
}
