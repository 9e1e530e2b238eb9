// Semantics check for the block-scaled fp8 MFMA as the split-operand convs use it (diagnostic, not part of the library):
//   * operands: 32 bytes per lane = the 16-byte chunks g and 4 + g of a 128-byte row (the two k-half reads of the fp16 path side by side);
//   * A and B quantised by v_cvt_pk_fp8_f32 (OCP e4m3 on gfx950) from fp32 values * 2^shift, products rescaled by the E8M0 scale operands
//     (127 - shift_a, 127 - shift_b);
//   * result against a host dot product over the SAME decoded values.
// build: hipcc --offload-arch=gfx950 -O2 scripts/micro/mfma_fp8_scaled.hip -o /tmp/mfma_fp8_scaled
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void quant(const float* x, unsigned char* q, int n, float s) {
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  float a = __builtin_amdgcn_fmed3f(x[i] * s, -448.f, 448.f), b = __builtin_amdgcn_fmed3f(x[i + 1] * s, -448.f, 448.f);
  float c = __builtin_amdgcn_fmed3f(x[i + 2] * s, -448.f, 448.f), d = __builtin_amdgcn_fmed3f(x[i + 3] * s, -448.f, 448.f);
  int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
  *reinterpret_cast<int*>(q + i) = v;
}
// A: [16 rows][128 B], B: [16 cols][128 B], out[16][16] (row-major: out[i][j] = sum_k A[i][k] B[j][k])
__global__ void mm(const unsigned char* A, const unsigned char* B, float* out, int sa, int sb) {
  const int lane = threadIdx.x, g = lane >> 4, l15 = lane & 15;
  const v4i a0 = *reinterpret_cast<const v4i*>(A + l15 * 128 + g * 16), a1 = *reinterpret_cast<const v4i*>(A + l15 * 128 + 64 + g * 16);
  const v4i b0 = *reinterpret_cast<const v4i*>(B + l15 * 128 + g * 16), b1 = *reinterpret_cast<const v4i*>(B + l15 * 128 + 64 + g * 16);
  const v8i a = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]}, b = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
  for (int r = 0; r < 4; ++r) out[(g * 4 + r) * 16 + l15] = acc[r];   // C layout: rows 4g + r (A's index), column l15 (B's index)
}
static float dec_e4m3(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
  if (e == 15 && m == 7) f = NAN;
  return s ? -f : f;
}
int main() {
  const int n = 16 * 128;
  std::vector<float> ha(n), hb(n);
  srand(1);
  for (int i = 0; i < n; ++i) { ha[i] = ((rand() % 2001) - 1000) * 1e-3f * 0.2f; hb[i] = ((rand() % 2001) - 1000) * 1e-6f * (1 + (i % 7)); }
  ha[5] = 10.0f;   // saturates at 448 / 2^shift_a
  const int shift_a = 11, shift_b = 15;
  float *da, *db, *dout; unsigned char *qa, *qb;
  hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&qa, n); hipMalloc(&qb, n); hipMalloc(&dout, 256 * 4);
  hipMemcpy(da, ha.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), n * 4, hipMemcpyHostToDevice);
  quant<<<(n / 4 + 63) / 64, 64>>>(da, qa, n, std::ldexp(1.0f, shift_a));
  quant<<<(n / 4 + 63) / 64, 64>>>(db, qb, n, std::ldexp(1.0f, shift_b));
  mm<<<1, 64>>>(qa, qb, dout, 127 - shift_a, 127 - shift_b);
  std::vector<unsigned char> ca(n), cb(n); std::vector<float> out(256);
  hipMemcpy(ca.data(), qa, n, hipMemcpyDeviceToHost); hipMemcpy(cb.data(), qb, n, hipMemcpyDeviceToHost); hipMemcpy(out.data(), dout, 1024, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0, qerr = 0;
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double ref = 0, exact = 0;
      for (int k = 0; k < 128; ++k) {
        ref += (double)dec_e4m3(ca[i * 128 + k]) * dec_e4m3(cb[j * 128 + k]);
        exact += (double)ha[i * 128 + k] * hb[j * 128 + k];
      }
      ref = std::ldexp(ref, -(shift_a + shift_b));
      maxerr = std::fmax(maxerr, std::fabs(out[i * 16 + j] - ref)); maxref = std::fmax(maxref, std::fabs(ref));
      qerr = std::fmax(qerr, std::fabs(ref - exact));
    }
  printf("e4m3 of 10.0 * 2^11 (saturated): 0x%02x = %g; of ha[0] = %g * 2^11: 0x%02x = %g\n", ca[5], dec_e4m3(ca[5]), ha[0], ca[0], dec_e4m3(ca[0]));
  printf("MFMA vs host dot product over the decoded operands: max |diff| %.3e of max |ref| %.3e (relative %.2e); quantisation itself: %.3e\n", maxerr, maxref, maxerr / maxref, qerr);
  return maxerr / maxref < 1e-5 ? 0 : 1;
}
