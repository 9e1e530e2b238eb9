// Backward-pass primitives for the fine-tuning step of the reference (gfx950, MI355X).
//
// The reference fine-tunes the UNet + the 768->768 text projection on 64 x 64 images = 8 x 8 latents
// (/root/reference/ldiffusion.py:198-255: Resize(64), V5 loop, `engine.backward(loss)`, `engine.step()`), so a training step is
// bound by the 859 M weights it reads (forward, dgrad) and writes (wgrad, AdamW), not by MFMA rate: M = B*h*w is 512 rows at the top
// UNet level and 8 at the bottom.  The contractions of the backward pass therefore REUSE the forward kernels of this library:
//   dgrad  dx = conv(dy, flip/transpose(W))                       -> ldiff_op_conv on rearranged weights (python side, a layout cast)
//   wgrad  dW[n][tap*C + c] = sum_m dy[m, n] * xcol[m, tap*C + c]  -> ldiff_op_conv as a plain GEMM over K = M on the two transposed
//                                                                    operands this file produces (im2col_t, transpose_rows)
// and this file adds what has no forward counterpart: the transposed im2col / transpose staging, bias-gradient column sums,
// GroupNorm(+SiLU) and LayerNorm forward-with-statistics / backward, GEGLU backward, attention backward for short sequences, AdamW.
// Activations and their gradients are NHWC fp16, reductions and parameter gradients fp32.
#include "common.h"

static inline int nblk(long long n, int per = 256) { return (int)((n + per - 1) / per); }
__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + __expf(-v)); }

// ---- xcolT[(tap*C + c)][m] = x[b, (oy*stride - pad + ky) >> ups, (ox*stride - pad + kx) >> ups, c]  (0 outside), m = (b, oy, ox) ----
// rows of Mpad (multiple of 8) columns, columns >= M zero: the K-major "weight" operand of the wgrad GEMM
__global__ void im2col_t_kernel(const f16* __restrict__ x, f16* __restrict__ out, int B, int H, int W, int C, int ks, int stride, int pad, int ups,
                                int Ho, int Wo, int Mpad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long rows = (long long)ks * ks * C;
  if (i >= rows * Mpad) return;
  const int m = (int)(i % Mpad);
  const long long r = i / Mpad;
  const int c = (int)(r % C), tap = (int)(r / C);
  f16 v = (f16)0.f;
  const int M = B * Ho * Wo;
  if (m < M) {
    const int b = m / (Ho * Wo), rem = m - b * (Ho * Wo), oy = rem / Wo, ox = rem - oy * Wo;
    const int iy = oy * stride - pad + tap / ks, ix = ox * stride - pad + tap % ks;
    if (iy >= 0 && iy < (H << ups) && ix >= 0 && ix < (W << ups)) v = x[(((long long)b * H + (iy >> ups)) * W + (ix >> ups)) * C + c];
  }
  out[i] = v;
}
void launch_im2col_t(const f16* x, f16* out, int B, int H, int W, int C, int ks, int stride, int pad, int ups, int Ho, int Wo, int Mpad, hipStream_t s) {
  LDIFF_CHECK(Mpad % 8 == 0 && Mpad >= B * Ho * Wo, LDIFF_ERR_INVALID, "im2col_t: Mpad=%d must be a multiple of 8 and >= M", Mpad);
  const long long n = (long long)ks * ks * C * Mpad;
  if (n == 0) return;
  hipLaunchKernelGGL(im2col_t_kernel, dim3(nblk(n)), dim3(256), 0, s, x, out, B, H, W, C, ks, stride, pad, ups, Ho, Wo, Mpad);
  HIP_CHECK(hipGetLastError());
}

// ---- out[n][m] = x[m][n] (rows of Mpad columns, zero padded): dy^T, the "activation" operand of the wgrad GEMM ----
__global__ void transpose_rows_kernel(const f16* __restrict__ x, f16* __restrict__ out, int M, int N, int ldx, int Mpad) {
  __shared__ f16 tile[32][33];
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int m = m0 + j, n = n0 + tx;
    tile[j][tx] = (m < M && n < N) ? x[(long long)m * ldx + n] : (f16)0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int n = n0 + j, m = m0 + tx;
    if (n < N && m < Mpad) out[(long long)n * Mpad + m] = tile[tx][j];
  }
}
void launch_transpose_rows(const f16* x, f16* out, int M, int N, int ldx, int Mpad, hipStream_t s) {
  LDIFF_CHECK(Mpad % 8 == 0 && Mpad >= M, LDIFF_ERR_INVALID, "transpose: Mpad=%d must be a multiple of 8 and >= M", Mpad);
  if (M == 0 || N == 0) return;
  hipLaunchKernelGGL(transpose_rows_kernel, dim3((Mpad + 31) / 32, (N + 31) / 32), dim3(256), 0, s, x, out, M, N, ldx, Mpad);
  HIP_CHECK(hipGetLastError());
}

// ---- db[n] = sum_m dy[m, n]  (bias gradient): one workgroup per 64 columns, fp32 ----
__global__ void colsum_kernel(const f16* __restrict__ dy, float* __restrict__ db, int M, int N, int ld) {
  __shared__ float red[4][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float a = 0.f;
  if (n < N)
    for (int m = part; m < M; m += 4) a += (float)dy[(long long)m * ld + n];
  red[part][threadIdx.x & 63] = a;
  __syncthreads();
  if (part == 0 && n < N) db[n] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
void launch_colsum(const f16* dy, float* db, int M, int N, int ld, hipStream_t s) {
  if (N == 0) return;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, s, dy, db, M, N, ld);
  HIP_CHECK(hipGetLastError());
}

// ---- block reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float block_sum(float v, float* red /* [4] */) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---- GroupNorm (+SiLU) forward that keeps the statistics: y = act(xhat*gamma + beta), mean / rstd per (b, group).  One workgroup
// per (b, group); x [B, HW, C] NHWC fp16.  (diffusers ResnetBlock2D.norm1/2 + nonlinearity, Transformer2DModel.norm) ----
__global__ __launch_bounds__(256) void gn_train_fwd_kernel(const f16* __restrict__ x, f16* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ rstd, int HW, int C,
                                                           int G, float eps, int silu) {
  __shared__ float red[4];
  const int b = blockIdx.y, g = blockIdx.x, Cg = C / G, c0 = g * Cg;
  const long long base = (long long)b * HW * C;
  const int n = HW * Cg;
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += (float)x[base + (long long)(i / Cg) * C + c0 + i % Cg];
  const float mu = block_sum(s, red) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float d = (float)x[base + (long long)(i / Cg) * C + c0 + i % Cg] - mu; q += d * d; }
  const float r = rsqrtf(block_sum(q, red) / (float)n + eps);
  if (threadIdx.x == 0) { mean[b * G + g] = mu; rstd[b * G + g] = r; }
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = c0 + i % Cg;
    const long long at = base + (long long)(i / Cg) * C + c;
    float a = ((float)x[at] - mu) * r * gamma[c] + beta[c];
    if (silu) a = a * sigmoid_f(a);
    y[at] = (f16)a;
  }
}
void launch_gn_train_fwd(const f16* x, f16* y, const float* gamma, const float* beta, float* mean, float* rstd, int B, int HW, int C, int G, float eps,
                         int silu, hipStream_t s) {
  LDIFF_CHECK(G > 0 && C % G == 0, LDIFF_ERR_INVALID, "gn_train_fwd: C=%d not divisible by groups=%d", C, G);
  if (B == 0 || HW == 0) return;
  hipLaunchKernelGGL(gn_train_fwd_kernel, dim3(G, B), dim3(256), 0, s, x, y, gamma, beta, mean, rstd, HW, C, G, eps, silu);
  HIP_CHECK(hipGetLastError());
}

// ---- GroupNorm (+SiLU) backward.  With a = xhat*gamma + beta, y = act(a), da = dy * act'(a):
//   dgamma[c] += sum da*xhat,  dbeta[c] += sum da,
//   dx = rstd * (da*gamma - mean_g(da*gamma) - xhat * mean_g(da*gamma*xhat))          (means over the HW*Cg elements of the group)
// One workgroup per (b, group); dgamma / dbeta accumulate over b with fp32 atomics (pre-zeroed by the caller). ----
__global__ __launch_bounds__(256) void gn_train_bwd_kernel(const f16* __restrict__ x, const f16* __restrict__ dy, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           f16* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int HW, int C, int G,
                                                           int silu) {
  __shared__ float red[4];
  extern __shared__ float cacc[];   // [2][Cg]: per-channel sums of da*xhat and da
  const int b = blockIdx.y, g = blockIdx.x, Cg = C / G, c0 = g * Cg;
  const long long base = (long long)b * HW * C;
  const int n = HW * Cg;
  const float mu = mean[b * G + g], r = rstd[b * G + g];
  for (int i = threadIdx.x; i < 2 * Cg; i += 256) cacc[i] = 0.f;
  __syncthreads();
  auto da_of = [&](long long at, int c, float& xh) {
    xh = ((float)x[at] - mu) * r;
    float d = (float)dy[at];
    if (silu) {
      const float a = xh * gamma[c] + beta[c], sg = sigmoid_f(a);
      d *= sg * (1.0f + a * (1.0f - sg));
    }
    return d;
  };
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = c0 + i % Cg;
    float xh;
    const float d = da_of(base + (long long)(i / Cg) * C + c, c, xh);
    s1 += d * gamma[c];
    s2 += d * gamma[c] * xh;
    atomicAdd(&cacc[i % Cg], d * xh);
    atomicAdd(&cacc[Cg + i % Cg], d);
  }
  const float m1 = block_sum(s1, red) / (float)n;
  const float m2 = block_sum(s2, red) / (float)n;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = c0 + i % Cg;
    const long long at = base + (long long)(i / Cg) * C + c;
    float xh;
    const float d = da_of(at, c, xh);
    dx[at] = (f16)(r * (d * gamma[c] - m1 - xh * m2));
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cg; i += 256) { atomicAdd(&dgamma[c0 + i], cacc[i]); atomicAdd(&dbeta[c0 + i], cacc[Cg + i]); }
}
void launch_gn_train_bwd(const f16* x, const f16* dy, const float* gamma, const float* beta, const float* mean, const float* rstd, f16* dx, float* dgamma,
                         float* dbeta, int B, int HW, int C, int G, int silu, hipStream_t s) {
  LDIFF_CHECK(G > 0 && C % G == 0, LDIFF_ERR_INVALID, "gn_train_bwd: C=%d not divisible by groups=%d", C, G);
  if (B == 0 || HW == 0) return;
  hipLaunchKernelGGL(gn_train_bwd_kernel, dim3(G, B), dim3(256), 2 * (C / G) * sizeof(float), s, x, dy, gamma, beta, mean, rstd, dx, dgamma, dbeta, HW, C, G,
                     silu);
  HIP_CHECK(hipGetLastError());
}

// ---- LayerNorm backward (BasicTransformerBlock.norm1/2/3): one wave per row, statistics recomputed from x;
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;   dgamma[c] += dy*xhat, dbeta[c] += dy  (fp32 atomics) ----
__global__ __launch_bounds__(256) void ln_bwd_kernel(const f16* __restrict__ x, const f16* __restrict__ dy, const float* __restrict__ gamma, f16* __restrict__ dx,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C, float eps) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const f16* xr = x + (long long)row * C;
  const f16* dr = dy + (long long)row * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += (float)xr[c];
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
  for (int c = lane; c < C; c += 64) { const float d = (float)xr[c] - mu; q += d * d; }
  const float r = rsqrtf(wave_sum(q) / (float)C + eps);
  float g1 = 0.f, g2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float xh = ((float)xr[c] - mu) * r, gg = (float)dr[c] * gamma[c];
    g1 += gg; g2 += gg * xh;
  }
  g1 = wave_sum(g1) / (float)C; g2 = wave_sum(g2) / (float)C;
  for (int c = lane; c < C; c += 64) {
    const float xh = ((float)xr[c] - mu) * r, d = (float)dr[c];
    dx[(long long)row * C + c] = (f16)(r * (d * gamma[c] - g1 - xh * g2));
    atomicAdd(&dgamma[c], d * xh);
    atomicAdd(&dbeta[c], d);
  }
}
void launch_ln_bwd(const f16* x, const f16* dy, const float* gamma, f16* dx, float* dgamma, float* dbeta, int rows, int C, float eps, hipStream_t s) {
  if (rows <= 0) return;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, dy, gamma, dx, dgamma, dbeta, rows, C, eps);
  HIP_CHECK(hipGetLastError());
}

// ---- GEGLU backward: y = h * gelu(g), x = [h | g] (diffusers GEGLU, erf GELU) ----
__global__ void geglu_bwd_kernel(const f16* __restrict__ x, const f16* __restrict__ dy, f16* __restrict__ dx, long long M, int C4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M * C4) return;
  const long long m = i / C4;
  const int c = (int)(i - m * C4);
  const float h = (float)x[m * 2 * C4 + c], g = (float)x[m * 2 * C4 + C4 + c], d = (float)dy[i];
  const float cdf = 0.5f * (1.0f + erff(g * 0.70710678118654752f)), pdf = 0.3989422804014327f * __expf(-0.5f * g * g);
  dx[m * 2 * C4 + c] = (f16)(d * g * cdf);
  dx[m * 2 * C4 + C4 + c] = (f16)(d * h * (cdf + g * pdf));
}
void launch_geglu_bwd(const f16* x, const f16* dy, f16* dx, long long M, int C4, hipStream_t s) {
  if (M * C4 == 0) return;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(nblk(M * C4)), dim3(256), 0, s, x, dy, dx, M, C4);
  HIP_CHECK(hipGetLastError());
}

// ---- attention backward for the short sequences of the fine-tuning step (8 x 8 latents: L = 64, 16, 4, 1; context length <= 77):
// one workgroup per (batch, head); S = scale*Q K^T and P = softmax(S) are recomputed in LDS (fp32), then
//   dV = P^T dO,  dP = dO V^T,  dS = P * (dP - rowsum(dP * P)),  dQ = scale * dS K,  dK = scale * dS^T Q.
// Lq * Lk <= 8192 scores (32 KiB of LDS).  Layout as the forward kernel: [B, L, ld] with head h at column h*d. ----
__global__ __launch_bounds__(256) void attn_bwd_kernel(const f16* __restrict__ q, int ldq, const f16* __restrict__ k, int ldk, const f16* __restrict__ v, int ldv,
                                                       const f16* __restrict__ dO, int ldo, f16* __restrict__ dq, f16* __restrict__ dk, f16* __restrict__ dv, int Lq,
                                                       int Lk, int d, long long q_bs, long long kv_bs, long long o_bs, float scale, int staged) {
  extern __shared__ float sm[];   // P [Lq][Lk], dS [Lq][Lk], then (staged) the operand chunks
  float* P = sm;
  float* dS = sm + Lq * Lk;
  const int b = blockIdx.y, h = blockIdx.x, tid = threadIdx.x;
  const f16* qb = q + b * q_bs + h * d;
  const f16* kb = k + b * kv_bs + h * d;
  const f16* vb = v + b * kv_bs + h * d;
  const f16* ob = dO + b * o_bs + h * d;
  if (staged) {
    // scores and dP with the operands staged through LDS in chunks of 64 head-dim columns (coalesced global reads, rows padded to 33 words:
    // lanes differ in the key row).  The direct loop below walks q / k / dO / v rows with one lane per row: at d = 512 (VAE mid block,
    // 64 x 64 scores, ONE head) that was 2.7 ms of latency-bound loads per call
    constexpr int CH = 64, RW = CH / 2 + 1;
    unsigned* Qs = reinterpret_cast<unsigned*>(sm + 2 * Lq * Lk);
    unsigned* Os = Qs + Lq * RW;
    unsigned* Ks = Os + Lq * RW;
    unsigned* Vs = Ks + Lk * RW;
    float sacc[32], dacc[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) { sacc[r] = 0.f; dacc[r] = 0.f; }
    for (int c0 = 0; c0 < d; c0 += CH) {
      const int cw2 = min(CH, d - c0) >> 1;
      for (int e = tid; e < Lq * cw2; e += 256) {
        const int row = e / cw2, cc = e - row * cw2;
        Qs[row * RW + cc] = *reinterpret_cast<const unsigned*>(qb + (long long)row * ldq + c0 + 2 * cc);
        Os[row * RW + cc] = *reinterpret_cast<const unsigned*>(ob + (long long)row * ldo + c0 + 2 * cc);
      }
      for (int e = tid; e < Lk * cw2; e += 256) {
        const int row = e / cw2, cc = e - row * cw2;
        Ks[row * RW + cc] = *reinterpret_cast<const unsigned*>(kb + (long long)row * ldk + c0 + 2 * cc);
        Vs[row * RW + cc] = *reinterpret_cast<const unsigned*>(vb + (long long)row * ldv + c0 + 2 * cc);
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int i = tid + 256 * r;
        if (i < Lq * Lk) {
          const int iq = i / Lk, ik = i - iq * Lk;
          float a = 0.f, b2 = 0.f;
          for (int cc = 0; cc < cw2; ++cc) {
            const f16x2 q2 = __builtin_bit_cast(f16x2, Qs[iq * RW + cc]), k2 = __builtin_bit_cast(f16x2, Ks[ik * RW + cc]);
            const f16x2 o2 = __builtin_bit_cast(f16x2, Os[iq * RW + cc]), v2 = __builtin_bit_cast(f16x2, Vs[ik * RW + cc]);
            a += (float)q2[0] * (float)k2[0] + (float)q2[1] * (float)k2[1];
            b2 += (float)o2[0] * (float)v2[0] + (float)o2[1] * (float)v2[1];
          }
          sacc[r] += a; dacc[r] += b2;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const int i = tid + 256 * r;
      if (i < Lq * Lk) { P[i] = sacc[r] * scale; dS[i] = dacc[r]; }
    }
  } else
  for (int i = tid; i < Lq * Lk; i += 256) {   // scores and dP
    const int iq = i / Lk, ik = i - iq * Lk;
    float s = 0.f, dp = 0.f;
    for (int c = 0; c < d; ++c) {
      s += (float)qb[(long long)iq * ldq + c] * (float)kb[(long long)ik * ldk + c];
      dp += (float)ob[(long long)iq * ldo + c] * (float)vb[(long long)ik * ldv + c];
    }
    P[i] = s * scale;
    dS[i] = dp;
  }
  __syncthreads();
  for (int iq = tid; iq < Lq; iq += 256) {     // row softmax, then dS = P * (dP - sum(dP*P))
    float mx = -INFINITY;
    for (int j = 0; j < Lk; ++j) mx = fmaxf(mx, P[iq * Lk + j]);
    float den = 0.f;
    for (int j = 0; j < Lk; ++j) { const float e = __expf(P[iq * Lk + j] - mx); P[iq * Lk + j] = e; den += e; }
    float dot = 0.f;
    for (int j = 0; j < Lk; ++j) { P[iq * Lk + j] /= den; dot += dS[iq * Lk + j] * P[iq * Lk + j]; }
    for (int j = 0; j < Lk; ++j) dS[iq * Lk + j] = P[iq * Lk + j] * (dS[iq * Lk + j] - dot);
  }
  __syncthreads();
  for (int i = tid; i < Lq * d; i += 256) {    // dQ = scale * dS K
    const int iq = i / d, c = i - iq * d;
    float a = 0.f;
    for (int j = 0; j < Lk; ++j) a += dS[iq * Lk + j] * (float)kb[(long long)j * ldk + c];
    dq[b * q_bs + (long long)iq * ldq + h * d + c] = (f16)(a * scale);
  }
  for (int i = tid; i < Lk * d; i += 256) {    // dK = scale * dS^T Q,  dV = P^T dO
    const int ik = i / d, c = i - ik * d;
    float a = 0.f, w = 0.f;
    for (int j = 0; j < Lq; ++j) {
      a += dS[j * Lk + ik] * (float)qb[(long long)j * ldq + c];
      w += P[j * Lk + ik] * (float)ob[(long long)j * ldo + c];
    }
    dk[b * kv_bs + (long long)ik * ldk + h * d + c] = (f16)(a * scale);
    dv[b * kv_bs + (long long)ik * ldv + h * d + c] = (f16)w;
  }
}
void launch_attn_bwd(const AttnParams& p, const f16* dO, f16* dq, f16* dk, f16* dv, hipStream_t s) {
  LDIFF_CHECK((long long)p.Lq * p.Lk <= 8192, LDIFF_ERR_INVALID, "attn_bwd: Lq*Lk = %d*%d exceeds the 8192 scores of the short-sequence kernel", p.Lq, p.Lk);
  LDIFF_CHECK(p.kv_bstride != 0 || p.B == 1, LDIFF_ERR_INVALID, "attn_bwd: broadcast K/V need their gradients summed over the batch by the caller (pass expanded K/V)");
  if (p.B == 0) return;
  size_t smem = (size_t)2 * p.Lq * p.Lk * sizeof(float);
  const size_t stage = (size_t)2 * (p.Lq + p.Lk) * 33 * sizeof(unsigned);
  // the staged path reads q, k, v and dO as 32-bit words: head dim, row pitches AND batch strides even, base pointers 4-byte aligned
  const bool even = p.d % 2 == 0 && p.ldq % 2 == 0 && p.ldk % 2 == 0 && p.ldv % 2 == 0 && p.ldo % 2 == 0 && p.q_bstride % 2 == 0 && p.kv_bstride % 2 == 0 &&
                    p.o_bstride % 2 == 0 && (((size_t)p.q | (size_t)p.k | (size_t)p.v | (size_t)dO) & 3) == 0;
  const int staged = (smem + stage <= 150 * 1024 && even) ? 1 : 0;
  if (staged) smem += stage;
  ensure_dyn_smem(reinterpret_cast<const void*>(attn_bwd_kernel), (int)smem);
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(p.heads, p.B), dim3(256), smem, s, p.q, p.ldq, p.k, p.ldk, p.v, p.ldv, dO, p.ldo, dq, dk, dv, p.Lq, p.Lk, p.d,
                     p.q_bstride, p.kv_bstride, p.o_bstride, p.scale, staged);
  HIP_CHECK(hipGetLastError());
}

// ---- weight layouts of the training step in one pass each (the torch formulation was zeros + permute + cast + slice-assign per call) ----
// mode 0 (forward):  dst[n][ky][kx][c] = w[n][c][ky][kx]            dst rows R >= Cout, row length k*k*Cp, Cp >= Cin; zero elsewhere
// mode 1 (dgrad):    dst[c][ky][kx][n] = w[n][c][k-1-ky][k-1-kx]    dst rows R >= Cin,  row length k*k*Cp, Cp >= Cout; zero elsewhere
// Both are transposes of a matrix whose rows are far apart in memory; they go through LDS so that every global access is a contiguous
// run (the one-thread-per-destination formulation read the fp32 master with a stride of 36 B (forward) or Cin * 36 B (dgrad) between
// lanes: 10.5 ms per step for 860 M parameters whose traffic takes 2 ms).
constexpr int PW_CT = 256;   // forward: channels per workgroup tile; rows per tile: 1 (3x3) or 8 (1x1), i.e. <= 2304 elements
__device__ __forceinline__ void pack_fwd_tile(float* t, const float* __restrict__ w, f16* __restrict__ dst, int Cout, int Cin, int kk, int R, int Cp, int bx, int by) {
  const int rb = kk == 1 ? 8 : 1;
  const int r0 = by * rb, c0 = bx * PW_CT, tid = threadIdx.x;
  const int nc = min(PW_CT, Cp - c0);   // destination columns of this chunk
  for (int rr = 0; rr < rb; ++rr) {
    const int r = r0 + rr;
    if (r >= R) break;
    const int ncs = r < Cout ? max(0, min(PW_CT, Cin - c0)) : 0;   // of which exist in the source
    const float* src = w + ((long long)r * Cin + c0) * kk;
    float* tr = t + rr * PW_CT;   // (rb > 1 only with kk == 1)
    for (int e = tid; e < nc * kk; e += 256) tr[e] = e < ncs * kk ? src[e] : 0.f;   // [c][tap], contiguous in memory
  }
  __syncthreads();
  for (int rr = 0; rr < rb; ++rr) {
    const int r = r0 + rr;
    if (r >= R) break;
    const float* tr = t + rr * PW_CT;
    f16* d = dst + (long long)r * kk * Cp + c0;
    for (int tap = 0; tap < kk; ++tap)   // two channels per lane: 4-byte stores (Cp and c0 are even)
      for (int cc = 2 * tid; cc < nc; cc += 512) {
        const f16x2 v = {(f16)tr[cc * kk + tap], (f16)tr[(cc + 1) * kk + tap]};
        *reinterpret_cast<f16x2*>(d + (long long)tap * Cp + cc) = v;
      }
  }
}
template <int KK, int TC>   // dgrad: a tile of 64 output channels n x TC input channels c x KK taps
__device__ __forceinline__ void pack_dgrad_tile(float* t, const float* __restrict__ w, f16* __restrict__ dst, int Cout, int Cin, int R, int Cp, int bx, int by) {
  constexpr int ROW = TC * KK + 1;
  const int n0 = bx * 64, c0 = by * TC, tid = threadIdx.x;
  for (int e = tid; e < 64 * TC * KK; e += 256) {
    const int nl = e / (TC * KK), off = e - nl * (TC * KK);
    const int n = n0 + nl, c = c0 + off / KK;
    t[nl * ROW + off] = (n < Cout && c < Cin) ? w[((long long)n * Cin + c0) * KK + off] : 0.f;   // per n: TC * KK contiguous floats
  }
  __syncthreads();
  for (int e = tid; e < TC * KK * 32; e += 256) {   // two output channels per lane: 4-byte stores (Cp and n0 are even)
    const int nl = (e & 31) * 2, ct = e >> 5;      // ct = cc * KK + tap' (destination order)
    const int cc = ct / KK, tap = ct - cc * KK;
    const int c = c0 + cc, n = n0 + nl;
    if (c < R && n < Cp) {
      const int src = cc * KK + (KK - 1 - tap);    // taps flipped: (k-1-ky, k-1-kx) = KK-1-tap
      const f16x2 v = {(f16)t[nl * ROW + src], (f16)t[(nl + 1) * ROW + src]};
      *reinterpret_cast<f16x2*>(dst + ((long long)c * KK + tap) * Cp + n) = v;
    }
  }
}
constexpr int PW_LDS = 64 * (16 * 9 + 1);   // floats: the largest tile (dgrad 3x3); forward tiles use 2304, dgrad 1x1 64 * 65
__global__ __launch_bounds__(256) void pack_weight_fwd_kernel(const float* __restrict__ w, f16* __restrict__ dst, int Cout, int Cin, int kk, int R, int Cp) {
  __shared__ float t[PW_CT * 9];
  pack_fwd_tile(t, w, dst, Cout, Cin, kk, R, Cp, blockIdx.x, blockIdx.y);
}
template <int KK, int TC>
__global__ __launch_bounds__(256) void pack_weight_dgrad_kernel(const float* __restrict__ w, f16* __restrict__ dst, int Cout, int Cin, int R, int Cp) {
  __shared__ float t[64 * (TC * KK + 1)];
  pack_dgrad_tile<KK, TC>(t, w, dst, Cout, Cin, R, Cp, blockIdx.x, blockIdx.y);
}
void launch_pack_weight(const float* w, f16* dst, int Cout, int Cin, int k, int R, int Cp, int mode, hipStream_t s) {
  const long long total = (long long)R * k * k * Cp;
  if (total == 0) return;
  LDIFF_CHECK(k == 1 || k == 3, LDIFF_ERR_INVALID, "pack_weight: kernel size %d (1 or 3)", k);
  if (mode == 0) {
    hipLaunchKernelGGL(pack_weight_fwd_kernel, dim3((Cp + PW_CT - 1) / PW_CT, k == 1 ? (R + 7) / 8 : R), dim3(256), 0, s, w, dst, Cout, Cin, k * k, R, Cp);
  } else if (k == 3) {
    hipLaunchKernelGGL((pack_weight_dgrad_kernel<9, 16>), dim3((Cp + 63) / 64, (R + 15) / 16), dim3(256), 0, s, w, dst, Cout, Cin, R, Cp);
  } else {
    hipLaunchKernelGGL((pack_weight_dgrad_kernel<1, 64>), dim3((Cp + 63) / 64, (R + 63) / 64), dim3(256), 0, s, w, dst, Cout, Cin, R, Cp);
  }
  HIP_CHECK(hipGetLastError());
}
// Both layouts of MANY weight tensors in one launch (the fine-tuning step inside a captured graph: 642 launches of ~14 us each were
// 9 ms of a 38 ms replay).  entries[e] = one (tensor, layout); tile_prefix[e] = first workgroup of entry e (tile_prefix[n] = all).
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const PackEntry* __restrict__ entries, const int* __restrict__ tile_prefix, int n_entries) {
  __shared__ float t[PW_LDS];
  int lo = 0, hi = n_entries;   // the entry this workgroup belongs to: last e with tile_prefix[e] <= blockIdx.x
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile_prefix[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
  const PackEntry en = entries[lo];
  const int local = (int)blockIdx.x - tile_prefix[lo];
  const int bx = local % en.tiles_x, by = local / en.tiles_x;
  if (en.mode == 0) pack_fwd_tile(t, en.w, en.dst, en.Cout, en.Cin, en.kk, en.R, en.Cp, bx, by);
  else if (en.kk == 9) pack_dgrad_tile<9, 16>(t, en.w, en.dst, en.Cout, en.Cin, en.R, en.Cp, bx, by);
  else pack_dgrad_tile<1, 64>(t, en.w, en.dst, en.Cout, en.Cin, en.R, en.Cp, bx, by);
}
void launch_pack_weight_multi(const PackEntry* entries, const int* tile_prefix, int n_entries, int n_tiles, hipStream_t s) {
  if (n_entries == 0 || n_tiles == 0) return;
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(n_tiles), dim3(256), 0, s, entries, tile_prefix, n_entries);
  HIP_CHECK(hipGetLastError());
}
// wgrad GEMM output g[n][tap*Cx + c] (row pitch ldg) -> dw[n][c][ky][kx] in the parameter's own layout (the inverse transpose, per row n)
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ g, float* __restrict__ dw, int Cin, int kk, int Cx, int ldg) {
  __shared__ float t[9 * (PW_CT + 1)];
  const int n = blockIdx.y, c0 = blockIdx.x * PW_CT, tid = threadIdx.x;
  const int nc = min(PW_CT, Cin - c0);
  const float* src = g + (long long)n * ldg + c0;
  for (int e = tid; e < nc * kk; e += 256) {
    const int tap = e / nc, cc = e - tap * nc;
    t[tap * (PW_CT + 1) + cc] = src[(long long)tap * Cx + cc];
  }
  __syncthreads();
  float* d = dw + ((long long)n * Cin + c0) * kk;
  for (int e = tid; e < nc * kk; e += 256) {
    const int cc = e / kk, tap = e - cc * kk;
    d[e] = t[tap * (PW_CT + 1) + cc];
  }
}
void launch_unpack_wgrad(const float* g, float* dw, int Cout, int Cin, int k, int Cx, int ldg, hipStream_t s) {
  const long long total = (long long)Cout * Cin * k * k;
  if (total == 0) return;
  LDIFF_CHECK(k == 1 || k == 3, LDIFF_ERR_INVALID, "unpack_wgrad: kernel size %d (1 or 3)", k);
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((Cin + PW_CT - 1) / PW_CT, Cout), dim3(256), 0, s, g, dw, Cin, k * k, Cx, ldg);
  HIP_CHECK(hipGetLastError());
}

// ---- AdamW (torch.optim.AdamW semantics, the optimiser of the reference's DeepSpeed config ldiffusion.py:168-171: lr 1e-5): fp32 master ----
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n, float lr, float b1,
                             float b2, float eps, float wd, float bc1, float bc2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = b1 * m[i] + (1.0f - b1) * gi, vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  m[i] = mi; v[i] = vi;
  float pi = p[i] * (1.0f - lr * wd);
  pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  p[i] = pi;
}
// All parameters of a model in ONE launch (688 tensors at SD-v1.5 width: one launch each was 8.6 ms of a step whose AdamW traffic, 28 B
// per parameter, takes 4.8 ms at HBM speed).  `tensors[t]` = {p, m, v} base pointers (static), `grads[t]` the gradient of tensor t (new
// every step: autograd allocates them), `chunks[c]` = {tensor, first element}: workgroup c updates ADAMW_CHUNK elements of its tensor.
__global__ __launch_bounds__(256) void adamw_multi_kernel(const AdamTensor* __restrict__ tensors, const float* const* __restrict__ grads,
                                                          const AdamChunk* __restrict__ chunks, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                          float bc2) {
  const AdamChunk c = chunks[blockIdx.x];
  const AdamTensor t = tensors[c.tensor];
  const float* __restrict__ g = grads[c.tensor];
  const long long end = c.first + ADAMW_CHUNK < t.n ? c.first + ADAMW_CHUNK : t.n;
  for (long long i = c.first + threadIdx.x; i < end; i += 256) {
    const float gi = g[i];
    const float mi = b1 * t.m[i] + (1.0f - b1) * gi, vi = b2 * t.v[i] + (1.0f - b2) * gi * gi;
    t.m[i] = mi; t.v[i] = vi;
    float pi = t.p[i] * (1.0f - lr * wd);
    pi -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    t.p[i] = pi;
  }
}
void launch_adamw_multi(const AdamTensor* tensors, const float* const* grads, const AdamChunk* chunks, long long nchunks, float lr, float b1, float b2, float eps,
                        float wd, int step, hipStream_t s) {
  LDIFF_CHECK(step >= 1, LDIFF_ERR_INVALID, "adamw: step counts from 1");
  if (nchunks == 0) return;
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)nchunks), dim3(256), 0, s, tensors, grads, chunks, lr, b1, b2, eps, wd, bc1, bc2);
  HIP_CHECK(hipGetLastError());
}
void launch_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd, int step, hipStream_t s) {
  LDIFF_CHECK(step >= 1, LDIFF_ERR_INVALID, "adamw: step counts from 1");
  if (n == 0) return;
  const float bc1 = 1.0f - powf(b1, (float)step), bc2 = 1.0f - powf(b2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(nblk(n)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2);
  HIP_CHECK(hipGetLastError());
}


// ---- contrastive (InfoNCE) feature loss of the fine-tuning step, forward AND gradient in one launch ----------------------------------
// /root/reference/model/loss.py:89-109 for GIVEN sample triples (the draws stay on the host: they consume the torch random stream in the
// reference's order, ldiffusion_amd/loss.py sample_triples):  feat[b, pixel, :] = features[b, :, pixel] (n planes, one per V5 pass);
//   logits_t = [a.p | a.n_1 .. a.n_K] / temperature,   loss = mean_t (logsumexp(logits_t) - logits_t[0]).
// One workgroup per triple: the K + 1 dot products (n <= 32 strided loads each; K = 1024 in the reference) into LDS, a block max / sum,
// then dL/dlogits = (softmax - onehot) / (T temperature) scattered back into dfeatures with float atomics (pixels repeat across triples;
// ~T (K + 2) n adds per step: thousands, nowhere near the atomic rate).  loss and dfeatures are zeroed by the launcher.
__global__ __launch_bounds__(256) void infonce_kernel(const float* __restrict__ feat, int n, long long HW, const int* __restrict__ bi, const int* __restrict__ ai,
                                                      const int* __restrict__ pi, const int* __restrict__ ni, int T_arg, const int* __restrict__ t_dev, int K, float inv_temp,
                                                      float* __restrict__ loss, float* __restrict__ dfeat) {
  extern __shared__ float lg[];   // K + 1 logits, then 2 x 4 reduction slots, then the anchor (n)
  const int t = blockIdx.x, tid = threadIdx.x;
  const int T = t_dev ? min(max(*t_dev, 0), T_arg) : T_arg;   // count from device memory, clamped to the capacity of the index arrays: the launch (grid = capacity) is shape-stable, e.g. inside a captured graph
  if (t >= T) return;
  const long long base = (long long)bi[t] * n * HW;
  float* red = lg + K + 1;
  float* av = red + 8;
  const int a = ai[t];
  for (int j = tid; j < n; j += 256) av[j] = feat[base + j * HW + a];
  __syncthreads();
  float mx = -3.0e38f;
  for (int k = tid; k <= K; k += 256) {
    const int px = k == 0 ? pi[t] : ni[(long long)t * K + k - 1];
    float d = 0.f;
    for (int j = 0; j < n; ++j) d += av[j] * feat[base + j * HW + px];
    d *= inv_temp;
    lg[k] = d;
    mx = fmaxf(mx, d);
  }
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float se = 0.f;
  for (int k = tid; k <= K; k += 256) se += expf(lg[k] - mx);
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
  if ((tid & 63) == 0) red[4 + (tid >> 6)] = se;
  __syncthreads();
  se = red[4] + red[5] + red[6] + red[7];
  const float invT = 1.0f / (float)T;
  if (tid == 0) atomicAdd(loss, (logf(se) + mx - lg[0]) * invT);
  // gradient: dl_k = (softmax_k - [k == 0]) / (T temperature);  d anchor += sum_k dl_k x_k,  d x_k += dl_k anchor
  float da[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) da[j] = 0.f;
  const float gs = invT * inv_temp / se;
  for (int k = tid; k <= K; k += 256) {
    const int px = k == 0 ? pi[t] : ni[(long long)t * K + k - 1];
    const float dl = expf(lg[k] - mx) * gs - (k == 0 ? invT * inv_temp : 0.f);
#pragma unroll
    for (int j = 0; j < 32; ++j)
      if (j < n) {
        da[j] += dl * feat[base + j * HW + px];
        atomicAdd(dfeat + base + j * HW + px, dl * av[j]);
      }
  }
#pragma unroll
  for (int j = 0; j < 32; ++j)
    if (j < n) {
      float v = da[j];
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if ((tid & 63) == 0) atomicAdd(dfeat + base + j * HW + a, v);
    }
}
void launch_infonce(const float* feat, int B, int n, long long HW, const int* bi, const int* ai, const int* pi, const int* ni, int T, const int* t_dev, int K,
                    float temperature, float* loss, float* dfeat, hipStream_t s) {
  LDIFF_CHECK(n >= 1 && n <= 32 && K >= 1 && K <= 8192 && temperature > 0.f, LDIFF_ERR_INVALID, "infonce: 1..32 feature planes, 1..8192 negatives");
  // (zeroed by a kernel, not hipMemsetAsync: launch_zero_bytes, kernels_elem.hip)
  launch_zero_bytes(dfeat, (size_t)B * n * HW * sizeof(float), s);
  launch_zero_bytes(loss, sizeof(float), s);
  if (T == 0) return;
  const size_t smem = (size_t)(K + 1 + 8 + n) * sizeof(float);
  hipLaunchKernelGGL(infonce_kernel, dim3(T), dim3(256), smem, s, feat, n, HW, bi, ai, pi, ni, T, t_dev, K, 1.0f / temperature, loss, dfeat);
  HIP_CHECK(hipGetLastError());
}

// ---- SiLU on fp16 rows and its backward (the time-embedding MLP of the fine-tuning step: train.py TrainableUNet.forward) ----
__global__ void silu_f16_kernel(const f16* __restrict__ x, f16* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = (float)x[i];
  y[i] = (f16)(v / (1.0f + expf(-v)));
}
__global__ void silu_bwd_f16_kernel(const f16* __restrict__ x, const f16* __restrict__ dy, f16* __restrict__ dx, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = (float)x[i], sg = 1.0f / (1.0f + expf(-v));
  dx[i] = (f16)((float)dy[i] * sg * (1.0f + v * (1.0f - sg)));   // d/dx x sigma(x) = sigma (1 + x (1 - sigma))
}
void launch_silu_f16(const f16* x, f16* y, long long n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(silu_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
  HIP_CHECK(hipGetLastError());
}
void launch_silu_bwd_f16(const f16* x, const f16* dy, f16* dx, long long n, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(silu_bwd_f16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, dy, dx, n);
  HIP_CHECK(hipGetLastError());
}
