// GroupNorm statistics and LayerNorm for gfx950 (MI355X).  HBM-bound: 16-byte coalesced NHWC reads.
//
// GroupNorm (diffusers ResnetBlock2D.norm1/norm2, Transformer2DModel.norm, conv_norm_out, VAE norms):
//   the apply (+SiLU) is folded into the consumer conv's A-tile load (kernels_igemm.hip); here only the
//   statistics are computed: ONE read of the tensor -> per-(b,channel) fp32 partial sums per pixel chunk,
//   then a finalize that combines chunks/channels per group in fp64 and emits
//       scale[b,c] = rstd*gamma[c],  shift[b,c] = beta[c] - mean*rstd*gamma[c].
//   Two NHWC sources (the UNet skip concat) are handled as one channel space, so a group may straddle
//   the concat boundary (e.g. 1280+640 channels, 60 per group).
// LayerNorm (BasicTransformerBlock.norm1/2/3): one wave per row, row held in registers, two-pass variance.
#include "common.h"

#define GN_PIX_PER_CHUNK_MIN 32

// Non-finite detector (include/ldiff.h "Non-finite detection"): every activation of a graph passes through GroupNorm statistics sooner or later (an fp16
// inf written by a conv / GEMM epilogue, or the NaN it turns into downstream, makes the sums below non-finite), and the workgroup that finalizes a
// group sees its totals in fp64 anyway: one predicated store into the handle's sticky flag, no extra read, no extra launch.
__device__ __forceinline__ void flag_nonfinite(int* flag, double a, double q) {
  if (flag && !(a > -1e300 && a < 1e300 && q < 1e300)) *flag = 1;   // (written negated: NaN fails every comparison)
}

static int gn_chunks(int HW) {
  // ~1024 blocks at (B=8, HW=4096); cap the chunk count so the partial buffer stays small at 512x512.
  int pix = GN_PIX_PER_CHUNK_MIN;
  while (HW / pix > 1024) pix *= 2;
  return (HW + pix - 1) / pix;
}

size_t gn_partial_bytes(int B, int HW, int C) { return (size_t)B * gn_chunks(HW) * C * 2 * sizeof(float); }

// 8 channels of one pixel as fp32: plain fp16 or split (hi + lo)
__device__ __forceinline__ void load8(const f16* p, int lo, float (&f)[8]) {
  const f16x8 h = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p));
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (float)h[j];
  if (lo) {
    const f16x8 l = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(p + lo));
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] += (float)l[j];
  }
}

// grid (nchunk, B, nsrc); block 256.  Thread (r, cc): channel chunk cc (8 channels), pixel rows r, r+R, ...
__global__ __launch_bounds__(256) void gn_partial_kernel(SrcView s1, SrcView s2, int HW, int pix_per_chunk, int nchunk, float* __restrict__ partial) {
  const int src = blockIdx.z;
  const SrcView sv = src ? s2 : s1;
  const f16* x = sv.p;
  const int C = sv.C, ld = sv.ld, lo = sv.lo, coff = src ? s1.C : 0, Ct = s1.C + (s2.p ? s2.C : 0);
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int CC = C >> 3;              // 16-byte chunks per pixel (<= 256)
  const int R = 256 / CC;
  const int tid = threadIdx.x;
  const int r = tid / CC, cc = tid - r * CC;
  const int p0 = chunk * pix_per_chunk;
  const int p1 = min(HW, p0 + pix_per_chunk);
  float s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; ss[j] = 0.f; }
  if (r < R) {
    const f16* base = x + ((long long)b * HW) * ld + cc * 8;
    // 4 independent 16-byte loads in flight per thread (the tensor is read exactly once: pure HBM streaming)
    int pix = p0 + r;
    for (; pix + 3 * R < p1; pix += 4 * R) {
      float f0[8], f1[8], f2[8], f3[8];
      load8(base + (long long)pix * ld, lo, f0);
      load8(base + (long long)(pix + R) * ld, lo, f1);
      load8(base + (long long)(pix + 2 * R) * ld, lo, f2);
      load8(base + (long long)(pix + 3 * R) * ld, lo, f3);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s[j] += (f0[j] + f1[j]) + (f2[j] + f3[j]);
        ss[j] += (f0[j] * f0[j] + f1[j] * f1[j]) + (f2[j] * f2[j] + f3[j] * f3[j]);
      }
    }
    for (; pix < p1; pix += R) {
      float f[8];
      load8(base + (long long)pix * ld, lo, f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += f[j]; ss[j] += f[j] * f[j]; }
    }
  }
  // reduce over r through LDS: red[r][c] for sums and squares
  extern __shared__ float red[];  // [2][R][C]
  if (r < R) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[r * C + cc * 8 + j] = s[j]; red[(R + r) * C + cc * 8 + j] = ss[j]; }
  }
  __syncthreads();
  float* out = partial + (((long long)b * nchunk + chunk) * Ct + coff) * 2;
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, q = 0.f;
    for (int rr = 0; rr < R; ++rr) { a += red[rr * C + c]; q += red[(R + rr) * C + c]; }
    out[c * 2] = a; out[c * 2 + 1] = q;
  }
}

// grid (groups, B); block 64: one wave per (b, group).
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ partial, int nchunk, int C, int groups, int HW, float eps,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ scale, float* __restrict__ shift, int* __restrict__ nonfinite) {
  const int grp = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int Cg = C / groups, c0 = grp * Cg;
  double a = 0.0, q = 0.0;
  const int total = nchunk * Cg;
  for (int i = lane; i < total; i += 64) {
    const int ch = i / Cg, c = c0 + (i - ch * Cg);
    const float* pp = partial + (((long long)b * nchunk + ch) * C + c) * 2;
    a += (double)pp[0]; q += (double)pp[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
  if (lane == 0) flag_nonfinite(nonfinite, a, q);
  const double n = (double)HW * Cg;
  const double mean = a / n;
  double var = q / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int c = c0 + lane; c < c0 + Cg; c += 64) {
    const float gsc = gamma[c] * rstd;
    scale[(long long)b * C + c] = gsc;
    shift[(long long)b * C + c] = beta[c] - (float)mean * gsc;
  }
}

// Finalize from producer-fused partials (common.h "GroupNorm statistics fused into the producing kernel's epilogue"):
// channels [0,C1) come from part1 ([B][C1][R1][2]), channels [C1,C1+C2) from part2 ([B][C2][R2][2]); each channel's R
// partial sums are contiguous, so a (image, group) block streams Cg contiguous runs.
__global__ __launch_bounds__(256) void gn_finalize2_kernel(const float* __restrict__ p1, int R1, int C1, const float* __restrict__ p2, int R2, int C2,
                                                           int groups, int HW, float eps, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift,
                                                           int* __restrict__ nonfinite) {
  const int grp = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int C = C1 + C2, Cg = C / groups, c0 = grp * Cg, c1 = c0 + Cg;
  // gamma / beta of this thread's first output channel go out with the partials (one round trip instead of a second, dependent one behind the
  // reduction: the launch is a pure latency chain, 6.5 us of which the kernel boundary is 1.5)
  const bool own = c0 + tid < c1;
  const float g_own = own ? gamma[c0 + tid] : 0.f, b_own = own ? beta[c0 + tid] : 0.f;
  // the partials of the group's channels are (up to) two contiguous runs of float2: channels [c0, min(c1, C1)) of part 1 and
  // [max(c0, C1), c1) of part 2.  All loads of a run are independent (a per-channel loop would chain Cg memory round trips:
  // 10-60 of them on the UNet levels, where R is small and most threads idle).
  double a = 0.0, q = 0.0;
  auto sum_run = [&](const float2* base, long long n) {   // n float2 elements, 8-byte aligned
    long long i = tid;
    for (; i + 768 < n; i += 1024) {
      const float2 v0 = base[i], v1 = base[i + 256], v2 = base[i + 512], v3 = base[i + 768];
      a += (double)((v0.x + v1.x) + (v2.x + v3.x));
      q += (double)((v0.y + v1.y) + (v2.y + v3.y));
    }
    for (; i < n; i += 256) { const float2 v = base[i]; a += (double)v.x; q += (double)v.y; }
  };
  const int e1 = c1 < C1 ? c1 : C1;
  if (c0 < e1) sum_run(reinterpret_cast<const float2*>(p1 + ((long long)b * C1 + c0) * R1 * 2), (long long)(e1 - c0) * R1);
  const int s2 = c0 > C1 ? c0 : C1;
  if (s2 < c1) sum_run(reinterpret_cast<const float2*>(p2 + ((long long)b * C2 + (s2 - C1)) * R2 * 2), (long long)(c1 - s2) * R2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
  __shared__ double red[2][4];
  if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = q; }
  __syncthreads();
  a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  if (tid == 0) flag_nonfinite(nonfinite, a, q);
  const double n = (double)HW * Cg;
  const double mean = a / n;
  double var = q / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (own) {
    const float gsc = g_own * rstd;
    scale[(long long)b * C + c0 + tid] = gsc;
    shift[(long long)b * C + c0 + tid] = b_own - (float)mean * gsc;
  }
  for (int c = c0 + tid + 256; c < c0 + Cg; c += 256) {
    const float gsc = gamma[c] * rstd;
    scale[(long long)b * C + c] = gsc;
    shift[(long long)b * C + c] = beta[c] - (float)mean * gsc;
  }
}

void launch_gn_finalize(const float* part1, int R1, int C1, const float* part2, int R2, int C2, int B, int HW, int groups, float eps,
                        const float* gamma, const float* beta, float* scale, float* shift, hipStream_t s, int* nonfinite) {
  LDIFF_CHECK((C1 + C2) % groups == 0 && part1 && R1 > 0 && (C2 == 0 || (part2 && R2 > 0)), LDIFF_ERR_INVALID, "gn_finalize: bad arguments");
  ProfScope prof("gn_finalize", 0.0, 8.0 * B * ((double)R1 * C1 + (double)R2 * C2), s);
  hipLaunchKernelGGL(gn_finalize2_kernel, dim3(groups, B), dim3(256), 0, s, part1, R1, C1, part2, R2, C2, groups, HW, eps, gamma, beta, scale, shift, nonfinite);
  HIP_CHECK(hipGetLastError());
}

static inline SrcView norm_view(SrcView v) { if (v.p && v.ld == 0) v.ld = v.C; return v; }

// Small maps (UNet levels 1-3: 8x8 ... 32x32 pixels): statistics AND scale / shift in ONE launch.  grid (groups, B), block 256: the
// workgroup reads the group's channels of every pixel (the 16-byte chunks that overlap [c0, c0 + Cg), elements outside masked; a chunk
// never straddles the two sources because C1 % 8 == 0), fp32 per thread, fp64 across threads.  The two-launch form (partial + finalize)
// costs 18-30 us on these tensors of 1-10 MB: both launches are latency-, not bandwidth-bound.
__global__ __launch_bounds__(256) void gn_small_kernel(SrcView s1, SrcView s2, int HW, int groups, float eps, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift,
                                                       int* __restrict__ nonfinite) {
  const int grp = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int C1 = s1.C, C = C1 + (s2.p ? s2.C : 0), Cg = C / groups, c0 = grp * Cg, c1 = c0 + Cg;
  const int k0 = c0 >> 3, nk = ((c1 + 7) >> 3) - k0;   // chunks [k0, k0 + nk) overlap the group
  const int items = HW * nk;
  float a = 0.f, q = 0.f;
  for (int i = tid; i < items; i += 256) {
    const int pix = i / nk, k = k0 + (i - pix * nk), c = k * 8;
    float f[8];
    if (c < C1) load8(s1.p + ((long long)b * HW + pix) * s1.ld + c, s1.lo, f);
    else load8(s2.p + ((long long)b * HW + pix) * s2.ld + (c - C1), s2.lo, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = (c + j >= c0 && c + j < c1) ? f[j] : 0.f;
      a += v; q += v * v;
    }
  }
  double da = (double)a, dq = (double)q;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { da += __shfl_xor(da, o); dq += __shfl_xor(dq, o); }
  __shared__ double red[2][4];
  if ((tid & 63) == 0) { red[0][tid >> 6] = da; red[1][tid >> 6] = dq; }
  __syncthreads();
  da = red[0][0] + red[0][1] + red[0][2] + red[0][3];
  dq = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  if (tid == 0) flag_nonfinite(nonfinite, da, dq);
  const double n = (double)HW * Cg;
  const double mean = da / n;
  double var = dq / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int c = c0 + tid; c < c1; c += 256) {
    const float gsc = gamma[c] * rstd;
    scale[(long long)b * C + c] = gsc;
    shift[(long long)b * C + c] = beta[c] - (float)mean * gsc;
  }
}

void launch_gn_stats(SrcView x1, SrcView x2, int B, int HW, int groups, float eps, const float* gamma,
                     const float* beta, float* partial, size_t partial_bytes, float* scale, float* shift, hipStream_t s, int* nonfinite) {
  x1 = norm_view(x1); x2 = norm_view(x2);
  if (!x2.p) x2.C = 0;
  const int C1 = x1.C, C2 = x2.C, C = C1 + C2;
  LDIFF_CHECK(C1 % 8 == 0 && C2 % 8 == 0 && C1 > 0 && C1 <= 2048 && C2 <= 2048, LDIFF_ERR_INVALID, "gn_stats: bad channels C1=%d C2=%d", C1, C2);
  LDIFF_CHECK(x1.ld % 8 == 0 && x1.lo % 8 == 0 && (!x2.p || (x2.ld % 8 == 0 && x2.lo % 8 == 0)), LDIFF_ERR_INVALID, "gn_stats: pitches must be multiples of 8");
  LDIFF_CHECK(C % groups == 0, LDIFF_ERR_INVALID, "gn_stats: C=%d not divisible by groups=%d", C, groups);
  LDIFF_CHECK(gn_partial_bytes(B, HW, C) <= partial_bytes, LDIFF_ERR_INVALID, "gn_stats: workspace too small");
  // small map with enough (image, group) workgroups, or a batch so small that either form is a latency chain (B = 1, the reference's own batch:
  // the two-launch form costs 22 us per tensor there, 43 tensors per UNet pass -- profiles/r05_unet_launches_b1.txt): one launch.
  // LDIFF_GN_SMALL_BATCH=0 restores the two-launch form for small batches (A/B).
  static const bool small_batch = [] { const char* e = getenv("LDIFF_GN_SMALL_BATCH"); return !e || atoi(e) != 0; }();
  if ((HW <= 1024 && B * groups >= 64) || (small_batch && B * groups < 256 && HW <= 4096)) {
    ProfScope prof("gn_stats", 3.0 * B * HW * (double)C, 2.0 * B * HW * ((double)C1 * (x1.lo ? 2 : 1) + (double)C2 * (x2.lo ? 2 : 1)), s);
    hipLaunchKernelGGL(gn_small_kernel, dim3(groups, B), dim3(256), 0, s, x1, x2, HW, groups, eps, gamma, beta, scale, shift, nonfinite);
    HIP_CHECK(hipGetLastError());
    return;
  }
  const int nchunk = gn_chunks(HW);
  const int pix = (HW + nchunk - 1) / nchunk;
  const size_t smem = 2 * 2048 * sizeof(float);  // [2][R][C] with R*C <= 256*8
  ProfScope prof("gn_stats", 3.0 * B * HW * (double)C, 2.0 * B * HW * ((double)C1 * (x1.lo ? 2 : 1) + (double)C2 * (x2.lo ? 2 : 1)), s);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, B, C2 ? 2 : 1), dim3(256), smem, s, x1, x2, HW, pix, nchunk, partial);
  HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, B), dim3(64), 0, s, partial, nchunk, C, groups, HW, eps, gamma, beta, scale, shift, nonfinite);
  HIP_CHECK(hipGetLastError());
}

// ---- GroupNorm-apply (+SiLU) as its own pass (common.h launch_norm_apply): one thread per 8 channels of one pixel ----
__global__ __launch_bounds__(256) void norm_apply_kernel(SrcView s1, SrcView s2, int HW, long long rows, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int silu, f16* __restrict__ y, int ldy, int y_lo, int lo8) {
  const int Ct = s1.C + (s2.p ? s2.C : 0), cpr = Ct >> 3;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cpr) return;
  const long long m = i / cpr;
  const int c = (int)(i - m * cpr) * 8;
  const int b = (int)(m / HW);
  float f[8];
  if (c < s1.C) load8(s1.p + m * s1.ld + c, s1.lo, f);
  else load8(s2.p + m * s2.ld + (c - s1.C), s2.lo, f);
  const float* sc = scale + (long long)b * Ct + c;
  const float* sh = shift + (long long)b * Ct + c;
  const float4 a0 = *reinterpret_cast<const float4*>(sc), a1 = *reinterpret_cast<const float4*>(sc + 4);
  const float4 t0 = *reinterpret_cast<const float4*>(sh), t1 = *reinterpret_cast<const float4*>(sh + 4);
  const float sv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, tv[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
  f16x8 hi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = f[j] * sv[j] + tv[j];
    if (silu) v = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
    hi[j] = (f16)v;
    lo[j] = (f16)(v - (float)hi[j]);
  }
  *reinterpret_cast<uint4*>(y + m * ldy + c) = __builtin_bit_cast(uint4, hi);
  if (lo8) {   // lo half as e4m3 of lo * 2^LO8_SHIFT (saturating), one byte per channel behind the Ct fp16 hi values (ConvParams::lo8_slab0)
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = __builtin_amdgcn_fmed3f((float)lo[j] * (float)(1 << LO8_SHIFT), -448.f, 448.f);
    int q0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false), q1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], 0, false);
    q0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], q0, true); q1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], q1, true);
    *reinterpret_cast<int2*>(reinterpret_cast<unsigned char*>(y + m * ldy + y_lo) + c) = make_int2(q0, q1);
  } else if (y_lo) *reinterpret_cast<uint4*>(y + m * ldy + y_lo + c) = __builtin_bit_cast(uint4, lo);
}
void launch_norm_apply(SrcView x1, SrcView x2, int B, int HW, const float* scale, const float* shift, int silu, f16* y, int ldy, int y_lo, int lo8,
                       hipStream_t s) {
  x1 = norm_view(x1); x2 = norm_view(x2);
  if (!x2.p) x2.C = 0;
  const int Ct = x1.C + x2.C;
  LDIFF_CHECK(x1.p && y && scale && shift && x1.C % 8 == 0 && x2.C % 8 == 0 && x1.ld % 8 == 0 && x1.lo % 8 == 0 && (!x2.p || (x2.ld % 8 == 0 && x2.lo % 8 == 0)) &&
                  ldy % 8 == 0 && y_lo % 8 == 0 && (y_lo == 0 || y_lo >= Ct) && ldy >= (lo8 ? y_lo + Ct / 2 : Ct + (y_lo ? y_lo : 0)) && (!lo8 || (y_lo > 0 && Ct % 16 == 0)),
              LDIFF_ERR_INVALID, "norm_apply: bad layout (C=%d+%d ldy=%d y_lo=%d)", x1.C, x2.C, ldy, y_lo);
  const long long rows = (long long)B * HW, n = rows * (Ct >> 3);
  if (n == 0) return;
  ProfScope prof("norm_apply", 4.0 * rows * Ct, 2.0 * rows * ((double)x1.C * (x1.lo ? 2 : 1) + (double)x2.C * (x2.lo ? 2 : 1) + (double)Ct * (y_lo ? 2 : 1)), s);
  hipLaunchKernelGGL(norm_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x1, x2, HW, rows, scale, shift, silu, y, ldy, y_lo, lo8);
  HIP_CHECK(hipGetLastError());
}

// ---- LayerNorm: one wave per row, up to 2560 channels (5 chunks of 8 per lane) ----------------
#define LN_MAXCH 5
__global__ __launch_bounds__(256) void layernorm_kernel(SrcView xs, f16* __restrict__ y, int rows,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int C = xs.C, CC = C >> 3;
  const f16* xr = xs.p + (long long)row * xs.ld;
  float v[LN_MAXCH][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int cc = lane + i * 64;
    if (cc < CC) {
      load8(xr + cc * 8, xs.lo, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float var = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int cc = lane + i * 64;
    if (cc < CC) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { float dlt = v[i][j] - mean; var += dlt * dlt; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
  const float rstd = rsqrtf(var / (float)C + eps);
  f16* yr = y + (long long)row * C;
#pragma unroll
  for (int i = 0; i < LN_MAXCH; ++i) {
    const int cc = lane + i * 64;
    if (cc < CC) {
      f16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (f16)((v[i][j] - mean) * rstd * gamma[cc * 8 + j] + beta[cc * 8 + j]);
      *reinterpret_cast<uint4*>(yr + cc * 8) = __builtin_bit_cast(uint4, o);
    }
  }
}

// (Round 5: a form with 64 / LPR rows per wave for C = 320 / 640 / 1280 -- every lane busy, ten 16-byte loads per lane in flight -- was built,
// parity-green, and measured on the UNet pass: 12.42 / 12.41 ms with this kernel, 12.46 / 12.39 ms with it (profiles/r05_unet_ab_ln_rows.txt).
// No gain: removed again.)
void launch_layernorm(SrcView x, f16* y, int rows, const float* gamma, const float* beta, float eps, hipStream_t s) {
  x = norm_view(x);
  LDIFF_CHECK(x.C % 8 == 0 && x.C <= 8 * 64 * LN_MAXCH && x.ld % 8 == 0 && x.lo % 8 == 0, LDIFF_ERR_INVALID, "layernorm: C=%d unsupported", x.C);
  if (rows <= 0) return;
  ProfScope prof("layernorm", 8.0 * rows * x.C, 2.0 * rows * x.C * (x.lo ? 3.0 : 2.0), s);
  hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, y, rows, gamma, beta, eps);
  HIP_CHECK(hipGetLastError());
}


// ---- GroupNorm folded into a following 1x1 convolution / linear layer (no activation in between) -------------------------
// y = W (s_b * x + t_b) + bias = (W diag(s_b)) x + (bias + W t_b): per image b the normalisation becomes a scaled copy of the
// weights and a bias vector, and the contraction itself runs as a plain GEMM on the LDS-DMA kernel (transformer proj_in and the
// VAE attention q/k/v: GroupNorm -> conv1x1 / Linear without SiLU).  One workgroup per (image, 4 weight rows).
__global__ __launch_bounds__(256) void fold_gn_weights_kernel(const f16* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, f16* __restrict__ wb, float* __restrict__ biasb,
                                                               int Nrows, int C) {
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + wave;
  if (n >= Nrows) return;
  const f16* wr = w + (long long)n * C;
  f16* o = wb + ((long long)b * Nrows + n) * C;
  const float* sc = scale + (long long)b * C;
  const float* sh = shift + (long long)b * C;
  float acc = 0.f;
  for (int c = lane * 8; c < C; c += 512) {   // C % 8 == 0
    const f16x8 v = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(wr + c));
    f16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float wv = (float)v[j];
      r[j] = (f16)(wv * sc[c + j]);
      acc += wv * sh[c + j];
    }
    *reinterpret_cast<uint4*>(o + c) = __builtin_bit_cast(uint4, r);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) biasb[(long long)b * Nrows + n] = acc + (bias ? bias[n] : 0.f);
}
void launch_fold_gn_weights(const f16* w, const float* bias, const float* scale, const float* shift, f16* wb, float* biasb, int B, int Nrows, int C,
                            hipStream_t s) {
  LDIFF_CHECK(C % 8 == 0, LDIFF_ERR_INVALID, "fold_gn_weights: C=%d must be a multiple of 8", C);
  hipLaunchKernelGGL(fold_gn_weights_kernel, dim3((Nrows + 3) / 4, B), dim3(256), 0, s, w, bias, scale, shift, wb, biasb, Nrows, C);
  HIP_CHECK(hipGetLastError());
}
