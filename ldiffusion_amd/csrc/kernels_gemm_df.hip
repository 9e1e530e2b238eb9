// Producer / consumer ("dataflow") GEMM for the UNet's 1x1 convs and linears: y[M,N] = A[M,K] W[N,K]^T + bias (+ residual), fp16 operands,
// fp32 MFMA accumulate.  Serves every nn.Linear of diffusers' BasicTransformerBlock that is not fused with its LayerNorm (attn1/attn2 to_out,
// ff.net.2, the q/k/v / to_q / GEGLU projections at C = 640 / 1280), proj_in / proj_out of Transformer2DModel and the 1x1 conv_shortcut of
// ResnetBlock2D (reached from /root/reference/segmentor.py:103,526 and pixel_latent_vector.py:78).
//
// Why another GEMM (round 5; the round-4 verdict names it): in gemm_dma_kernel (kernels_gemm.hip) every wave does everything -- issues the LDS-DMA
// pieces of both operands, reads fragments, runs the MFMAs, meets the other three waves at a barrier per 64-deep K step, then runs an epilogue
// (bias, fp16 hi|lo residual, split output: ~10 VALU per output element) whose loads start when the last MFMA has finished.  With K = 320 ... 1280
// a tile is 5-20 K steps long: prologue and epilogue are a third to a half of its life, and the kernel sits at 0.2-0.6 PFLOP/s.  Here the roles
// are split inside one persistent 512-thread workgroup per CU, the way conv3x3d_kernel does it, and nothing in the K loop is synchronised:
//   * waves 0-3, one per SIMD: CONSUMERS.  A unit = 16 MT rows x 64 NTW columns of y over the whole K; consumer wave w owns every row and the
//     column tiles {4 a + w}: MT x NTW accumulator tiles.  The WEIGHTS of a step come straight from global memory (L2) into registers -- the
//     matrix is stored fragment-packed per layer (launch_pack_gemm_frag: every 16 x 32 MFMA A fragment is one contiguous KiB), NTW
//     global_load_dwordx4 per k-half issued one k-half ahead, counted vmcnt.  The ACTIVATIONS come from a ring of F_S slots in LDS (one slot =
//     16 MT rows x 64 k, the swizzled image of kernels_gemm.hip), row tiles in groups of two one group ahead, counted lgkmcnt.  At the end of a unit
//     a consumer writes its fp32 sums to LDS in 64-column slices and goes on with the next unit.
//   * waves 4 .. 4+NLOAD-1: LOADERS.  Free-running LDS-DMA of the activation slots, up to F_S - 1 steps ahead across unit boundaries (counted
//     vmcnt; nothing asynchronous targets a VGPR).
//   * the remaining waves: EPILOGUE.  Plain compiled code (loads, waits and stores are the compiler's): take a slice from LDS, add the bias, then the
//     residual (fp16 hi, then lo, in fp32: gemm_dma_kernel's order, the two kernels agree bit for bit), round once, write y (plain, or split hi | lo) as full 128-byte lines.  Runs beside the next unit's MFMAs.
//   * progress words in LDS (one per wave, as in conv3x3d_kernel): loaders "steps landed", consumers "steps whose fragments are in registers"
//     and "slices staged", epilogue waves "slices taken".
// A workgroup walks a contiguous run of units (column unit fastest, XCD-aware: the runs of one XCD's workgroups are adjacent).
//
// Scope (gemm_df_selected): ks == 1, K % 64 == 0, one or two row-major sources (channel concat at a multiple of 64), fp16 output plain or split,
// optional plain / split residual, optional GEGLU epilogue or fused GroupNorm partial statistics; no per-image weights, no split-K (those stay on
// gemm_dma_kernel).
#include "common.h"
#include <map>
#include <mutex>

namespace {

typedef __attribute__((address_space(3))) void lptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int V> using ic_t = std::integral_constant<int, V>;

// (tuning builds override these: scripts/gemm_df_stamps.py --def=...; measured in profiles/r05_gemm_df.md: three slots + three slice buffers and
// one loader wave beat five slots + two buffers and two loader waves on every level-0 / level-1 shape -- the consumers wait for slice buffers,
// never for slots)
#ifndef GDF_RING
#define GDF_RING 3
#endif
#ifndef GDF_SLICES
#define GDF_SLICES 3
#endif
#ifndef GDF_NLOAD
#define GDF_NLOAD 1
#endif
constexpr int F_S = GDF_RING;                            // ring slots; a loader runs at most F_S - 1 steps ahead
constexpr int F_NH = GDF_SLICES;                         // hand-off slice buffers
constexpr unsigned F_SLOT = 128 * 128;                   // one slot: up to 128 rows x 128 B (64 k)
constexpr unsigned F_HP = 272;                           // hand-off slice: row pitch (64 fp32 + 16 B: conflict-free on both sides)
constexpr unsigned F_HS = 128 * F_HP;                    // one slice buffer
constexpr unsigned F_HAND = F_S * F_SLOT;                // the slice buffers
constexpr unsigned F_FLAGS = F_HAND + F_NH * F_HS;       // [loader steps x4][consumer steps x4][consumer slices x4][epilogue slices x4]
constexpr unsigned F_DUMP = F_FLAGS + 64;                // 8 waves x 256 B: where the lanes other than 0 put their copy of a progress word
constexpr unsigned F_LDS = F_DUMP + 8 * 256;
static_assert(F_LDS <= 160 * 1024, "LDS budget of one workgroup per CU");
constexpr unsigned F_OOR = 0x80000000u;                  // beyond num_records of every descriptor used here
enum { FG_RES = 1, FG_RES_SPLIT = 2, FG_OUT_SPLIT = 4, FG_GEGLU = 8, FG_STATS = 16 };

__device__ __forceinline__ int swz8(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <int OFF>
__device__ __forceinline__ void lds_read128(f16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int CNT>
__device__ __forceinline__ void lds_wait2(f16x8& a, f16x8& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(CNT)); }
// (every register exactly once: a register named twice is COPIED in front of the statement, i.e. read while its load is still in flight)
__device__ __forceinline__ void vm_wait_frags(f16x8& a) { asm volatile("s_waitcnt vmcnt(1)" : "+v"(a)); }
__device__ __forceinline__ void vm_wait_frags(f16x8& a, f16x8& b) { asm volatile("s_waitcnt vmcnt(2)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void vm_wait_frags(f16x8& a, f16x8& b, f16x8& c) { asm volatile("s_waitcnt vmcnt(3)" : "+v"(a), "+v"(b), "+v"(c)); }
__device__ __forceinline__ void vm_wait_frags(f16x8& a, f16x8& b, f16x8& c, f16x8& d) { asm volatile("s_waitcnt vmcnt(4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
__device__ __forceinline__ void vm_wait_frags(f16x8& a, f16x8& b, f16x8& c, f16x8& d, f16x8& e) {
  asm volatile("s_waitcnt vmcnt(5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
}
template <int OFF>
__device__ __forceinline__ void glb_read128(f16x8& d, unsigned voff, const char* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
__device__ __forceinline__ unsigned flags_min_now(unsigned addr) {   // read four progress words and wait for them
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  const unsigned m = min(min(v[0], v[1]), min(v[2], v[3]));
  return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}
__device__ __forceinline__ void lds_write32(unsigned addr, unsigned v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_write128f(unsigned addr, const f32x4& v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void vm_wait_n(int n) {   // s_waitcnt vmcnt(n), n uniform at run time (the immediate is an instruction field)
  switch (n) {
#define VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    VMW(0) VMW(1) VMW(2) VMW(3) VMW(4) VMW(5) VMW(6) VMW(7) VMW(8) VMW(9) VMW(10) VMW(11) VMW(12) VMW(13) VMW(14) VMW(15) VMW(16) VMW(17) VMW(18) VMW(19)
    VMW(20) VMW(21) VMW(22) VMW(23) VMW(24) VMW(25) VMW(26) VMW(27) VMW(28) VMW(29) VMW(30) VMW(31) VMW(32) VMW(33) VMW(34) VMW(35) VMW(36) VMW(37) VMW(38)
    VMW(39) VMW(40) VMW(41) VMW(42) VMW(43) VMW(44) VMW(45) VMW(46) VMW(47) VMW(48)
#undef VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}
__device__ __forceinline__ float u2f(unsigned v) { return __builtin_bit_cast(float, v); }

#ifdef GDF_STAMPS   // diagnostic build only (scripts/gemm_df_stamps.py): cycle sums / poll counts of one wave per role of workgroup 0
__device__ unsigned long long gdf_dbg[64];
__device__ __forceinline__ unsigned long long g_stamp() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define GSTAMP(v) const unsigned long long v = g_stamp()
#define GACC(i, expr) dbg[i] += (expr)
#else
#define GSTAMP(v)
#define GACC(i, expr)
#endif

// MT: 16-row tiles per unit (8 = 128 rows, 4 = 64 rows); NTW: 16-column tiles per consumer wave (a unit has 64 NTW columns);
// FL: epilogue flags; NLOAD: loader waves (the other 4 - NLOAD producer-side waves run the epilogue)
template <int MT, int NTW, int FL, int NLOAD>
__global__ __launch_bounds__(512, 2) void gemm_df_kernel(const ConvParams p, const int units, const int nunits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int BM = MT * 16, BN = NTW * 64, NP_ALL = MT * 2, NEPI = 4 - NLOAD;
  constexpr bool RES = (FL & FG_RES) != 0, RES_SPLIT = (FL & FG_RES_SPLIT) != 0, OUT_SPLIT = (FL & FG_OUT_SPLIT) != 0, GEGLU = (FL & FG_GEGLU) != 0;
  constexpr bool STATS = (FL & FG_STATS) != 0;   // fused GroupNorm partial statistics (common.h): 32-row blocks, so row groups go to the epilogue waves in fours
  static_assert(!STATS || (!GEGLU && (NLOAD == 2 || MT == 4)), "statistics: two epilogue waves (or one 64-row half each of three)");
  static_assert(NLOAD >= 1 && NLOAD <= 3 && (MT == 8 || MT == 4) && NTW >= 1 && NTW <= 5, "configuration");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(lptr_t*)smem_raw;
  const int nk = p.K >> 6;

  // this workgroup's run of units (column unit fastest, then row block); the runs of the workgroups that share an XCD are adjacent
  const int G = gridDim.x, id = blockIdx.x;
  const int q8 = G >> 3, r8 = G & 7, xcd = id & 7, idx = id >> 3;
  const int sw = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  const int u0 = (int)((long long)sw * units / G), u1 = (int)((long long)(sw + 1) * units / G);
  const int n_u = u1 - u0;
  const int total = n_u * nk;   // K steps of the run

  if (tid < 16) {   // unused words of a class read as "infinitely far": the minimum over four words is the minimum over the live waves
    unsigned v = 0u;
    if ((tid < 4 && tid >= NLOAD) || (tid >= 12 && tid - 12 >= NEPI)) v = 0xffffffffu;
    *reinterpret_cast<volatile unsigned*>(smem_raw + F_FLAGS + tid * 4) = v;
  }
  __syncthreads();
  if (n_u <= 0) return;
  const unsigned pflags = lds0 + F_FLAGS, cflags = lds0 + F_FLAGS + 16u, eflags = lds0 + F_FLAGS + 32u, dflags = lds0 + F_FLAGS + 48u;
  const unsigned dump = lds0 + F_DUMP + (unsigned)wave * 256u + (unsigned)lane * 4u;   // progress word by ONE unmasked ds_write_b32: lane 0 hits the word

  if (wave >= 4 && wave < 4 + NLOAD) {
    // =================================================== LOADERS ===================================================
    const int lw = wave - 4;
    // pieces (8 rows x 128 B = one wave-instruction) of a slot this wave moves: [first, first + np)
    const int first = NLOAD == 3 ? (lw == 0 ? 0 : (NP_ALL * lw + 1) / 3) : lw * (NP_ALL / NLOAD);
    const int last = NLOAD == 3 ? (lw == 2 ? NP_ALL : (NP_ALL * (lw + 1) + 1) / 3) : (lw + 1) * (NP_ALL / NLOAD);
    const int np = last - first;
    constexpr int NPMAX = (NP_ALL + NLOAD - 1) / NLOAD + (NLOAD == 3 ? 1 : 0);
    const int pitch1 = p.ld1 ? p.ld1 : p.C1, pitch2 = p.ld2 ? p.ld2 : p.C2;
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((long long)p.M * pitch1 * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x2 ? p.x2 : p.x), 0, (int)((long long)p.M * (p.x2 ? pitch2 : pitch1) * 2), 0x00020000);
    const int kt2 = p.C1 >> 6;   // first K step of the second concat source
    const unsigned pflag_addr = lane == 0 ? pflags + (unsigned)lw * 4u : dump;
    int a_sw[NPMAX], a_row[NPMAX];
#pragma unroll
    for (int i = 0; i < NPMAX; ++i) {
      const int r = (first + i) * 8 + (lane >> 3);
      a_row[i] = r;
      a_sw[i] = swz8(r, lane & 7) * 16 - (i & 3) * 1024;
    }
    // issue cursor: step `issued` = K step ikt of local unit iu
    int issued = 0, iu = 0, ikt = 0, im0 = 0, in0 = 0;
    int v1[NPMAX], v2[NPMAX];
    auto set_unit = [&]() __attribute__((always_inline)) {
      const int u = u0 + iu, mb = u / nunits;
      im0 = __builtin_amdgcn_readfirstlane(mb * BM);
      in0 = __builtin_amdgcn_readfirstlane((u - mb * nunits) * BN);
#pragma unroll
      for (int i = 0; i < NPMAX; ++i) {
        int m = im0 + a_row[i];
        m = m < p.M ? m : p.M - 1;
        v1[i] = m * (pitch1 * 2) + a_sw[i];
        v2[i] = m * (pitch2 * 2) + a_sw[i];
      }
    };
    auto issue = [&]() __attribute__((always_inline)) {
      if (ikt == 0) set_unit();
      const bool second = ikt >= kt2;
      const int soff = (second ? ikt - kt2 : ikt) * 128;
      unsigned char* dst = smem_raw + (unsigned)(issued % F_S) * F_SLOT + (unsigned)first * 1024u;
#if defined(__HIP_DEVICE_COMPILE__)
      static_for<0, NPMAX>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i < np) {
          if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lptr_t*)(dst + (i >> 2) * 4096), 16, v2[i], soff, (i & 3) * 1024, 0);
          else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lptr_t*)(dst + (i >> 2) * 4096), 16, v1[i], soff, (i & 3) * 1024, 0);
        }
      });
#endif
      ++issued;
      if (++ikt == nk) { ikt = 0; ++iu; }
    };
#ifdef GDF_STAMPS
    unsigned long long dbg[16] = {0};
    GSTAMP(l_t0);
#endif
    const int ahead = total < F_S - 1 ? total : F_S - 1;
    for (int k = 0; k < ahead; ++k) issue();
    for (int j = 0; j < total; ++j) {
      GSTAMP(q0);
      vm_wait_n((issued - j - 1) * np);   // step j's pieces are older than the steps issued after it (a bias DMA among those only makes the wait longer)
      lds_write32(pflag_addr, (unsigned)j + 1u);
      GSTAMP(q1);
      if (issued < total) {
        // slot issued % F_S held step issued - F_S: free once every consumer has that step's fragments in registers
        while ((int)flags_min_now(cflags) < issued - F_S + 1) { __builtin_amdgcn_s_sleep(2); GACC(4, 1); }
        GSTAMP(q2);
        issue();
        GSTAMP(q3);
        GACC(2, q2 - q1); GACC(3, q3 - q2);
      }
      GACC(1, q1 - q0); GACC(5, 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef GDF_STAMPS
    { GSTAMP(l_t1); dbg[0] = l_t1 - l_t0; if (blockIdx.x == 0 && lw == 0 && lane == 0) for (int i = 0; i < 16; ++i) gdf_dbg[16 + i] = dbg[i]; }
#endif
    return;
  }

  if (wave >= 4 + NLOAD) {
    // =================================================== EPILOGUE ===================================================
    const int ew = wave - 4 - NLOAD;
    const unsigned dflag_addr = lane == 0 ? dflags + (unsigned)ew * 4u : dump;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((long long)p.M * p.ldy * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : (const f16*)p.y), 0, (int)((long long)p.M * (p.res ? p.ld_res : p.ldy) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : (const float*)p.y), 0, p.bias ? p.Nrows * 4 : 0, 0x00020000);   // no bias: zero records, loads return 0
    // a wave-instruction takes ROWS_I rows of the slice: 8 rows x 8 octets (a 64-column slice row = one 128-byte line of y), or, GEGLU, 16 rows x 4
    // octets (a slice carries 32 x | 32 gate columns = 32 outputs).  Row groups are dealt round-robin over the epilogue waves.
    constexpr int ROWS_I = GEGLU ? 16 : 8, NG = BM / ROWS_I, NIT = STATS ? 4 * ((NG / 4 + NEPI - 1) / NEPI) : (NG + NEPI - 1) / NEPI;
    // row group of iteration `it`: dealt round-robin -- or, with statistics, in runs of four (one 32-row block) round-robin
    auto row_group = [&](int it) __attribute__((always_inline)) -> int { return STATS ? ((it >> 2) * NEPI + ew) * 4 + (it & 3) : ew + it * NEPI; };
    const int px = GEGLU ? lane >> 2 : lane >> 3, o = GEGLU ? lane & 3 : lane & 7;
#ifdef GDF_STAMPS
    unsigned long long dbg[16] = {0};
    GSTAMP(e_t0);
#endif
    int sidx = 0;
    for (int i = 0; i < n_u; ++i) {
      const int u = u0 + i, mb = u / nunits;
      const int m0 = mb * BM, n0 = (u - mb * nunits) * BN;
      for (int a = 0; a < NTW; ++a, ++sidx) {
        const unsigned hb = lds0 + F_HAND + (unsigned)(sidx % F_NH) * F_HS;
        const int ncol = GEGLU ? n0 / 2 + a * 32 + o * 8 : n0 + a * 64 + o * 8;        // first of this lane's 8 output columns
        const bool nok = ncol < (GEGLU ? p.N / 2 : p.N);
        // bias of this lane's columns (GEGLU: of its 8 x and its 8 gate weight rows), added to the sums FIRST, then the residual hi, then lo:
        // the order of gemm_dma_kernel's epilogue, so that the two kernels agree bit for bit (a launch may go to either, by its row count)
        // (GEGLU: the weight rows are x / gate interleaved by 16: output column 16 q + c <- x row 32 q + c, gate row 32 q + 16 + c)
        const int brow = GEGLU ? n0 + a * 64 + (o >> 1) * 32 + (o & 1) * 8 : ncol;
        f32x4 Bv[GEGLU ? 4 : 2];
        Bv[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, brow * 4, 0, 0));
        Bv[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, brow * 4 + 16, 0, 0));
        if constexpr (GEGLU) {
          Bv[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (brow + 16) * 4, 0, 0));
          Bv[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, (brow + 16) * 4 + 16, 0, 0));
        }
        int yoff[NIT];
        u32x4 Rh[NIT], Rl[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int rg = row_group(it), m = m0 + rg * ROWS_I + px;
          const bool ok = rg < NG && m < p.M && nok;
          yoff[it] = ok ? (m * p.ldy + ncol) * 2 : (int)F_OOR;
          if constexpr (RES) {
            const int roff = ok ? (m * p.ld_res + ncol) * 2 : (int)F_OOR;
            Rh[it] = __builtin_amdgcn_raw_buffer_load_b128(rrs, roff, 0, 0);
            if constexpr (RES_SPLIT) Rl[it] = __builtin_amdgcn_raw_buffer_load_b128(rrs, ok ? roff + p.res_lo * 2 : (int)F_OOR, 0, 0);
          }
        }
        GSTAMP(w0);
        while ((int)flags_min_now(eflags) < sidx + 1) { __builtin_amdgcn_s_sleep(2); GACC(3, 1); }
        GSTAMP(w1);
        GACC(1, w1 - w0);
        f32x4 U[NIT][GEGLU ? 4 : 2];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int rg = row_group(it);
          const unsigned ra = hb + (unsigned)((rg < NG ? rg : 0) * ROWS_I + px) * F_HP + (unsigned)o * 32u;
          U[it][0] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)ra);
          U[it][1] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)(ra + 16u));
          if constexpr (GEGLU) {
            U[it][2] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)(ra + 128u));
            U[it][3] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((size_t)(ra + 144u));
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        lds_write32(dflag_addr, (unsigned)sidx + 1u);   // the slice is in registers: its buffer is free
        // statistics: two accumulator sets per 32-row block -- rows {px, px + 16} and {px + 8, px + 24} of the block -- reduced separately over the
        // row lanes and added last: the association of gemm_dma_kernel's wave_stats_store (a lane's two rows, quad, octet, 16 rows), bit for bit
        float x16[2][16];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          if constexpr (STATS) {
            if ((it & 3) == 0) {
#pragma unroll
              for (int j = 0; j < 16; ++j) { x16[0][j] = 0.f; x16[1][j] = 0.f; }
            }
          }
          float v[8];
          if constexpr (GEGLU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (U[it][e >> 2][e & 3] + Bv[e >> 2][e & 3]) * gelu_erf(U[it][2 + (e >> 2)][e & 3] + Bv[2 + (e >> 2)][e & 3]);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = U[it][e >> 2][e & 3] + Bv[e >> 2][e & 3];
          }
          if constexpr (RES) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const unsigned hd = Rh[it][d];
              const f16x2 h = __builtin_bit_cast(f16x2, hd);
              v[2 * d] += (float)h[0]; v[2 * d + 1] += (float)h[1];
              if constexpr (RES_SPLIT) {
                const unsigned ld = Rl[it][d];
                const f16x2 l = __builtin_bit_cast(f16x2, ld);
                v[2 * d] += (float)l[0]; v[2 * d + 1] += (float)l[1];
              }
            }
          }
          u32x4 oh, ol;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const f16x2 h = (f16x2){(f16)v[2 * d], (f16)v[2 * d + 1]};
            oh[d] = __builtin_bit_cast(unsigned, h);
            if constexpr (OUT_SPLIT) {
              const f16x2 l = (f16x2){(f16)(v[2 * d] - (float)h[0]), (f16)(v[2 * d + 1] - (float)h[1])};
              ol[d] = __builtin_bit_cast(unsigned, l);
            }
          }
          __builtin_amdgcn_raw_buffer_store_b128(oh, yrs, yoff[it], 0, 0);
          if constexpr (OUT_SPLIT) __builtin_amdgcn_raw_buffer_store_b128(ol, yrs, yoff[it] == (int)F_OOR ? (int)F_OOR : yoff[it] + p.y_lo * 2, 0, 0);
          if constexpr (STATS) {
            // what gemm_dma_kernel sums: the fp32 value of a split output, the rounded value of a plain one; rows beyond M count nothing
            const bool rok = yoff[it] != (int)F_OOR;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
              const unsigned hd = oh[d];
              const f16x2 h = __builtin_bit_cast(f16x2, hd);
              const float f0 = rok ? (OUT_SPLIT ? v[2 * d] : (float)h[0]) : 0.f, f1 = rok ? (OUT_SPLIT ? v[2 * d + 1] : (float)h[1]) : 0.f;
              float (&xs)[16] = x16[it & 1];
              xs[2 * d] += f0; xs[2 * d + 1] += f1;
              xs[8 + 2 * d] = __builtin_fmaf(f0, f0, xs[8 + 2 * d]); xs[8 + 2 * d + 1] = __builtin_fmaf(f1, f1, xs[8 + 2 * d + 1]);
            }
            if ((it & 3) == 3) {
              // a 32-row block is complete in this wave: per set 8 sums + 8 sums of squares per lane, reduced over the eight row lanes in three halving
              // steps (row_ror:8, v_permlane16_swap, v_permlane32_swap: the scheme of conv3x3d_kernel's store_unit): every lane ends with
              // (sum, sum of squares) of ONE channel: one 8-byte store
              const bool b0 = (lane & 8) != 0;
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                float (&xs)[16] = x16[h];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                  const float keep = b0 ? xs[2 * q + 1] : xs[2 * q], send = b0 ? xs[2 * q] : xs[2 * q + 1];
                  xs[q] = keep + dpp_f<0x128>(send);   // row_ror:8 = lane ^ 8
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, xs[2 * q + 1]), __builtin_bit_cast(unsigned, xs[2 * q]), false, false);
                  xs[q] = u2f(sw[0]) + u2f(sw[1]);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                  auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, xs[2 * q + 1]), __builtin_bit_cast(unsigned, xs[2 * q]), false, false);
                  xs[q] = u2f(sw[0]) + u2f(sw[1]);
                }
              }
              const float ssum = x16[0][0] + x16[1][0], ssq = x16[0][1] + x16[1][1];
              const int chn = ((lane >> 3) & 1) | ((((lane >> 4) & 1) ^ 1) << 1) | ((((lane >> 5) & 1) ^ 1) << 2);
              const int rg0 = row_group(it - 3), mblk = m0 + rg0 * ROWS_I;   // first row of the block (a multiple of 32)
              const int hw = p.Hout * p.Wout, bimg = mblk / hw, rblk = (mblk - bimg * hw) >> 5, n = ncol + chn;
              if (rg0 < NG && mblk < p.M && n < p.N)
                *reinterpret_cast<float2*>(p.stats + (((long long)bimg * p.N + n) * p.stats_R + rblk) * 2) = make_float2(ssum, ssq);
            }
          }
        }
        GSTAMP(w2);
        GACC(2, w2 - w1); GACC(4, 1);
      }
    }
#ifdef GDF_STAMPS
    { GSTAMP(e_t1); dbg[0] = e_t1 - e_t0; if (blockIdx.x == 0 && ew == 0 && lane == 0) for (int i = 0; i < 16; ++i) gdf_dbg[32 + i] = dbg[i]; }
#endif
    return;
  }

  // =================================================== CONSUMERS ===================================================
  const int cw = wave;
  const int g = lane >> 4, l15 = lane & 15;
  // activation fragment of row tile m, k-half kh: row m * 16 + l15, 16-byte chunk (kh * 4 + g) ^ swizzle(row); (16 m) >> 1 is a multiple of 8, so the
  // swizzle depends on l15 only and m, kh, slot go into the instruction offset / an XOR of 64
  const unsigned xa0 = lds0 + (unsigned)(l15 * 128 + ((g ^ ((l15 >> 1) & 7)) << 4));
  const unsigned cflag_addr = lane == 0 ? cflags + (unsigned)cw * 4u : dump;
  const unsigned eflag_addr = lane == 0 ? eflags + (unsigned)cw * 4u : dump;
  const int ntiles = p.Nrows >> 4;                     // 16-column tiles the packed matrix holds
  const char* wbase = reinterpret_cast<const char*>(p.w_frag);

#ifdef GDF_STAMPS
  unsigned long long dbg[16] = {0};
#endif
  // activation fragments travel in groups of GR = 2 row tiles, XD - 1 groups ahead (XD register sets).  ONE group ahead is the kept form: three
  // groups ahead (GDF_XD=4: 6 NTW MFMAs of lead instead of 2 NTW) made the K loops 10 % SLOWER (L1 q/k/v 28.9 -> 34.9 us, profiles/r05_gemm_df.md):
  // the next slot is then asked for a third of a step into the current one, and with three ring slots the consumers end up polling for it.
  // The read stream runs through step AND unit boundaries (the next unit's first slot is simply the ring's next one); behind the run's last
  // step it reads a slot nobody filled, unused.
#ifndef GDF_XD
#define GDF_XD 2
#endif
  constexpr int GR = 2, NGH = MT / GR, NQ = 2 * NGH, XD = NTW >= 5 ? 2 : GDF_XD, XA = XD - 1;
  static_assert(NQ % XD == 0 && XA <= NQ, "a step's group count must be a multiple of the register sets");
  f32x4 acc[NTW][MT];
  f16x8 Wf[2][NTW], X[XD][GR];
  unsigned woff[NTW];   // per column tile: byte offset of its fragments of K step 0 (+ lane * 16)

  auto set_unit_w = [&](int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < NTW; ++a) {
      // plain: column tile 4 a + cw of the unit; GEGLU (weight rows x / gate interleaved by 16): slice a = [x(2a) x(2a+1) gate(2a) gate(2a+1)]
      int t = (n0 >> 4) + (GEGLU ? 2 * (2 * a + (cw & 1)) + (cw >> 1) : 4 * a + cw);
      t = t < ntiles ? t : ntiles - 1;   // beyond N: computed on the last tile, never stored
      woff[a] = (unsigned)t * (unsigned)nk * 2048u + (unsigned)lane * 16u;
    }
  };
  auto issue_w = [&](auto khc, const char* wq) __attribute__((always_inline)) {
    constexpr int kh = decltype(khc)::value;
#ifdef GDF_ABL_NOW   // ablation (timing only, results wrong): no weight loads
    return;
#endif
    static_for<0, NTW>([&](auto ac) { constexpr int a = decltype(ac)::value; glb_read128<kh * 1024>(Wf[kh][a], woff[a], wq); });
  };
  auto vm_wait_w = [&](auto khc) __attribute__((always_inline)) {   // all but the NTW youngest loads: k-half kh's fragments have landed
    constexpr int kh = decltype(khc)::value;
#ifdef GDF_ABL_NOW
    return;
#endif
    if constexpr (NTW == 1) vm_wait_frags(Wf[kh][0]);
    else if constexpr (NTW == 2) vm_wait_frags(Wf[kh][0], Wf[kh][1]);
    else if constexpr (NTW == 3) vm_wait_frags(Wf[kh][0], Wf[kh][1], Wf[kh][2]);
    else if constexpr (NTW == 4) vm_wait_frags(Wf[kh][0], Wf[kh][1], Wf[kh][2], Wf[kh][3]);
    else vm_wait_frags(Wf[kh][0], Wf[kh][1], Wf[kh][2], Wf[kh][3], Wf[kh][4]);
  };
  auto issue_x = [&](auto gc, auto khc, unsigned sb, f16x8 (&dst)[GR]) __attribute__((always_inline)) {   // row tiles GR gr, GR gr + 1 of k-half kh from the slot at sb
    constexpr int gr = decltype(gc)::value, kh = decltype(khc)::value;
    const unsigned b0 = (xa0 + sb) ^ (kh ? 64u : 0u);
#ifdef GDF_ABL_NOX   // ablation (timing only): no activation fragment reads
    return;
#endif
    static_for<0, GR>([&](auto rc) { constexpr int r = decltype(rc)::value; lds_read128<(GR * gr + r) * 2048>(dst[r], b0); });
  };
  auto lds_wait_g = [&](auto cc, f16x8 (&x)[GR]) __attribute__((always_inline)) {
    constexpr int CNT = decltype(cc)::value;
#ifdef GDF_ABL_NOX
    return;
#endif
    lds_wait2<CNT>(x[0], x[1]);
  };
  auto mfma_g = [&](auto khc, auto gc, f16x8 (&x)[GR]) __attribute__((always_inline)) {
    constexpr int kh = decltype(khc)::value, gr = decltype(gc)::value;
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
      for (int r = 0; r < GR; ++r) acc[a][GR * gr + r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wf[kh][a], x[r], acc[a][GR * gr + r], 0, 0, 0);
  };
  unsigned have = 0;   // steps known to have landed
  auto need_steps = [&](unsigned n) __attribute__((always_inline)) {
#ifdef GDF_ABL_NOP   // ablation (timing only): the consumers never wait for a slot to land
    return;
#endif
    while (have < n) { have = flags_min_now(pflags); GACC(1, 1); }
  };
  unsigned taken = 0;  // slices known to be out of their buffers
  auto zero_sums = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[a][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

  GSTAMP(c_t0);
  int n0 = 0;
  {
    const int mb = u0 / nunits;
    n0 = (u0 - mb * nunits) * BN;
  }
  set_unit_w(n0);
  const char* wq = wbase;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (nothing of the compiler's may be younger than the hand-counted loads below)
  issue_w(ic_t<0>{}, wq);
  need_steps(1u);
  zero_sums();
  static_for<0, XA>([&](auto tc) { constexpr int t = decltype(tc)::value; issue_x(ic_t<t % NGH>{}, ic_t<t / NGH>{}, 0u, X[t]); });
  GSTAMP(c_t1);
  GACC(0, c_t1 - c_t0);

  int j = 0, sidx = 0;   // global K step / slice of this workgroup
  for (int i = 0; i < n_u; ++i) {
    const bool has_next = i + 1 < n_u;
    int n0_next = n0;
    if (has_next) { const int u = u0 + i + 1, mb = u / nunits; n0_next = (u - mb * nunits) * BN; }
    for (int kt = 0; kt < nk; ++kt, ++j) {
      const bool last = kt == nk - 1;
      const unsigned sb = (unsigned)(j % F_S) * F_SLOT, sbn = (unsigned)((j + 1) % F_S) * F_SLOT;
      // entry: in flight are this step's first XA row groups (-> X[0 .. XA-1]) and the weights of k-half 0 (NTW global loads -> Wf[0]).
      // The step is NQ group-steps (k-half 0's row groups, then k-half 1's); group-step q computes with X[q % XD] while the reads of q + 1 .. q + XA travel.
      const bool more = j + 1 < total;
      static_for<0, NQ>([&](auto qc) {
        constexpr int q = decltype(qc)::value, kh = q / NGH, gr = q % NGH, t = q + XA;
        if constexpr (q == 0) issue_w(ic_t<1>{}, wq);
        if constexpr (q == NGH) {
          // the weights of the next step: the next K step of this unit, or step 0 of the next unit (after the last unit: loaded again, unused)
          if (last) { set_unit_w(n0_next); wq = wbase; } else wq += 2048;
          issue_w(ic_t<0>{}, wq);
        }
        if constexpr (t < NQ) issue_x(ic_t<t % NGH>{}, ic_t<t / NGH>{}, sb, X[t % XD]);
        else {   // the next step's groups (its slot must have landed before the first of them is read)
          if constexpr (t == NQ) { if (more) need_steps((unsigned)j + 2u); }
          issue_x(ic_t<(t - NQ) % NGH>{}, ic_t<(t - NQ) / NGH>{}, sbn, X[t % XD]);
        }
        lds_wait_g(ic_t<XA * GR>{}, X[q % XD]);
        // every fragment of step j is in registers once its last group has landed: its slot is free
        if constexpr (q == NQ - 1) lds_write32(cflag_addr, (unsigned)j + 1u);
        if constexpr (gr == 0) vm_wait_w(ic_t<kh>{});
        mfma_g(ic_t<kh>{}, ic_t<gr>{}, X[q % XD]);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    // ---- end of the unit: hand the sums to the epilogue waves, slice by slice (two buffers) ----
    GSTAMP(e0);
    static_for<0, NTW>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
#ifdef GDF_ABL_NOSTAGE   // ablation (timing only): no hand-over of the sums (the epilogue waves see the flags only)
      ++sidx; lds_write32(eflag_addr, (unsigned)sidx); return;
#endif
      while ((int)taken < sidx - (F_NH - 1)) { taken = flags_min_now(dflags); GACC(2, 1); }
      const unsigned hb = lds0 + F_HAND + (unsigned)(sidx % F_NH) * F_HS + (unsigned)l15 * F_HP + (unsigned)(cw * 16 + g * 4) * 4u;
#pragma unroll
      for (int m = 0; m < MT; ++m) lds_write128f(hb + (unsigned)m * 16u * F_HP, acc[a][m]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ++sidx;
      lds_write32(eflag_addr, (unsigned)sidx);
    });
    GSTAMP(e1);
    GACC(3, e1 - e0); GACC(4, 1);
    if (has_next) {
      zero_sums();   // (the next unit's first fragments are already on their way: the read stream does not stop at a unit's end)
      n0 = n0_next;
    }
    GSTAMP(e2);
    GACC(5, e2 - e1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the unused weights of the step after the last
#ifdef GDF_STAMPS
  { GSTAMP(c_t2); dbg[6] = c_t2 - c_t0; if (blockIdx.x == 0 && cw == 0 && lane == 0) for (int i = 0; i < 16; ++i) gdf_dbg[i] = dbg[i]; }
#endif
}

// fragment-packed copy of a weight matrix [Nrows][K] (K-major) for the consumers: 16-column tile t, K step kt, k-half kh is one contiguous KiB:
// wf[((t nk + kt) 2 + kh) * 512 + lane * 8 ..] = w[t 16 + (lane & 15)][kt 64 + kh 32 + (lane >> 4) 8 .. + 8]
__global__ void pack_gemm_frag_kernel(const f16* __restrict__ w, f16* __restrict__ wf, int Nrows, int K) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // one 16-byte chunk each
  if (t >= (long long)Nrows * K / 8) return;
  const int lane = (int)(t & 63);
  long long f = t >> 6;
  const int kh = (int)(f & 1); f >>= 1;
  const int nk = K >> 6;
  const int kt = (int)(f % nk), tile = (int)(f / nk);
  *reinterpret_cast<uint4*>(wf + t * 8) = *reinterpret_cast<const uint4*>(w + (long long)(tile * 16 + (lane & 15)) * K + kt * 64 + kh * 32 + (lane >> 4) * 8);
}

int f_num_cus() {
  static std::mutex mu;
  static std::map<int, int> cus;
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  int n = 0;
  HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
  return cus[dev] = n > 0 ? n : 256;
}

struct DfPlan { int mt, ntw; };
// Unit shape: the plan with the shortest estimated time = (units per workgroup, rounded up) x (K steps x the cost of a step + the hand-over of the
// sums).  A step costs its matrix cycles or the cycles its operand bytes take to arrive at ~28 B per cycle and CU, whichever is longer -- at these
// tile sizes it is nearly always the bytes (profiles/r05_gemm_df.md: the unit shapes rank as this model says on all 22 UNet shapes measured).
DfPlan df_plan(const ConvParams& p) {
  static const int force_mt = [] { const char* e = getenv("LDIFF_GEMM_DF_MT"); return e ? atoi(e) : 0; }();
  static const int force_ntw = [] { const char* e = getenv("LDIFF_GEMM_DF_NTW"); return e ? atoi(e) : 0; }();
  const int cus = f_num_cus();
  DfPlan best{8, 5};
  double best_t = 1e30;
  const int arg_mt = p.df_force >= 16 ? p.df_force >> 4 : 0, arg_ntw = p.df_force >= 16 ? p.df_force & 15 : 0;
  LDIFF_CHECK(p.df_force < 16 || ((arg_mt == 4 || arg_mt == 8) && (arg_ntw == 2 || arg_ntw == 4 || arg_ntw == 5)), LDIFF_ERR_INVALID,
              "gemm (dataflow): unit shape %d x %d is not built", arg_mt * 16, arg_ntw * 64);
  for (int mt : {8, 4})
    for (int ntw : {5, 4, 2}) {
      if (arg_mt ? mt != arg_mt : (force_mt && mt != force_mt)) continue;
      if (arg_ntw ? ntw != arg_ntw : (force_ntw && ntw != force_ntw)) continue;
      const int bm = mt * 16, bn = ntw * 64;
      const long long units = (long long)((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn);
      const double rounds = (double)((units + cus - 1) / cus);
      const double mfma = (double)mt * ntw * 4.0 * 2.0 * 16.0;                      // matrix cycles of a K step on one SIMD
      const double feed = ((double)bm * 128.0 + (double)bn * 128.0) / 28.0;          // operand bytes of a K step at ~28 B per cycle and CU
      const double t = rounds * ((p.K / 64.0) * (mfma > feed ? mfma : feed) + 500.0 * ntw * mt / 8.0);
      if (t < best_t) { best_t = t; best = DfPlan{mt, ntw}; }
    }
  return best;
}

template <int MT, int NTW, int FL, int NLOAD>
void launch_df(const ConvParams& p, hipStream_t s) {
  auto kern = gemm_df_kernel<MT, NTW, FL, NLOAD>;
  ensure_dyn_smem(reinterpret_cast<const void*>(kern), (int)F_LDS);
  const int bm = MT * 16, bn = NTW * 64;
  const int nunits = (p.N + bn - 1) / bn, units = ((p.M + bm - 1) / bm) * nunits;
  const int cus = f_num_cus();
  // run length (units per workgroup): LDIFF_GEMM_DF_RUN > 0 caps it (diagnostic: a persistent workgroup holds its CU for the whole launch, which
  // matters where another stream's kernels wait for CUs -- the sampler's decodes beside the UNet pass); 0 (default) = one workgroup per CU
  static const int run_cap = [] { const char* e = getenv("LDIFF_GEMM_DF_RUN"); return e ? atoi(e) : 0; }();
  int grid = units < cus ? units : cus;
  if (run_cap > 0 && units > grid * run_cap) grid = (units + run_cap - 1) / run_cap;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), F_LDS, s, p, units, nunits);
  HIP_CHECK(hipGetLastError());
}
template <int MT, int NTW, int FL>
void launch_df_l(const ConvParams& p, hipStream_t s) {
  // loader waves: ONE moves a 128-row slot per step (16 pieces at ~64 cycles each against >= 1024 matrix cycles per step at NTW >= 4); the other
  // three run the epilogue, which is what the consumers end up waiting for
#ifdef GDF_TUNE   // tuning build: LDIFF_GEMM_DF_NLOAD = 1 / 2 / 3 (three times the instantiations)
  static const int nl = [] { const char* e = getenv("LDIFF_GEMM_DF_NLOAD"); return e ? atoi(e) : GDF_NLOAD; }();
  if constexpr ((FL & FG_STATS) == 0) {
    if (nl == 1) return launch_df<MT, NTW, FL, 1>(p, s);
    if (nl == 2) return launch_df<MT, NTW, FL, 2>(p, s);
    if (nl == 3) return launch_df<MT, NTW, FL, 3>(p, s);
  }
#endif
  if constexpr ((FL & FG_STATS) != 0) launch_df<MT, NTW, FL, 2>(p, s);   // 32-row statistics blocks: two epilogue waves take two of a 128-row unit's four each
  else launch_df<MT, NTW, FL, GDF_NLOAD>(p, s);
}
template <int FL>
void launch_df_f(const ConvParams& p, hipStream_t s) {
  const DfPlan pl = df_plan(p);
  if (pl.mt == 8) {
    if (pl.ntw == 5) launch_df_l<8, 5, FL>(p, s);
    else if (pl.ntw == 4) launch_df_l<8, 4, FL>(p, s);
    else launch_df_l<8, 2, FL>(p, s);
  } else {
    if (pl.ntw == 5) launch_df_l<4, 5, FL>(p, s);
    else if (pl.ntw == 4) launch_df_l<4, 4, FL>(p, s);
    else launch_df_l<4, 2, FL>(p, s);
  }
}

}  // namespace

void launch_pack_gemm_frag(const f16* w, f16* wf, int Nrows, int K, hipStream_t s) {
  LDIFF_CHECK(Nrows % 16 == 0 && K % 64 == 0, LDIFF_ERR_INVALID, "gemm (dataflow): fragment packing needs Nrows %% 16 == 0 and K %% 64 == 0");
  const long long n = (long long)Nrows * K / 8;
  hipLaunchKernelGGL(pack_gemm_frag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, wf, Nrows, K);
  HIP_CHECK(hipGetLastError());
}
size_t gemm_df_frag_bytes(const ConvParams& p) { return (size_t)p.Nrows * p.K * sizeof(f16); }

// LDIFF_GEMM_DF: 0 = off, 1 (default) = where the unit list fills the chip, 2 = every eligible launch (tests, A/B timing)
bool gemm_df_selected(const ConvParams& p) {
  static const int mode = [] { const char* e = getenv("LDIFF_GEMM_DF"); return e ? atoi(e) : 1; }();
  if (p.df_force < 0 || (mode == 0 && p.df_force == 0) || !gemm_dma_eligible(p)) return false;
  if (p.w_bstride != 0 || p.splitk > 1 || p.out_f32 || p.M <= 0) return false;
  if (p.stats && (p.geglu || (p.Hout * p.Wout) % 32 != 0 || p.stats_R != (p.Hout * p.Wout) / 32)) return false;
  if (p.C1 % 64 != 0 || p.C2 % 64 != 0 || p.Nrows % 16 != 0 || p.Nrows < p.N) return false;
  if ((p.N & 7) || (p.ldy & 7) || (p.y_lo & 7) || (p.res && ((p.ld_res & 7) || (p.res_lo & 7)))) return false;
  if (p.geglu && (p.N % 64 != 0 || p.res || p.y_lo)) return false;
  if ((long long)p.M * p.ldy * 2 >= (1LL << 31) || (p.res && (long long)p.M * p.ld_res * 2 >= (1LL << 31)) || (long long)p.Nrows * p.K * 2 >= (1LL << 31)) return false;
  if (mode == 2 || p.df_force > 0) return true;
  // fused statistics: built and bit-identical to gemm_dma's, but with two epilogue waves summing beside their stores the launch takes twice
  // gemm_dma's time (proj_out at level 0: 36 -> 68 us): only on request
  if (p.stats) return false;
  // Measured against gemm_dma on the UNet's shapes at B = 8 (profiles/r05_gemm_df.md): ahead by 5-25 % wherever a launch has >= 4,096 rows, and on
  // the wide launches (N >= 3,840) of the 2,048-row level; behind on that level's narrow launches (16 row blocks: gemm_dma's 64 x 64 tiles and
  // split-K fill the chip better) and on GEGLU at K < 512, where the erf of the epilogue outweighs the matrix work whoever runs it.
  if (p.geglu && p.K < 512) return false;
  if (p.M < 4096) return p.M >= 1024 && p.N >= 3840;
  return true;
}

#ifdef GDF_STAMPS
extern "C" int ldiff_debug_gdf_stamps(unsigned long long* out) {   // diagnostic build only
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gdf_dbg), sizeof(unsigned long long) * 64);
}
#endif

void launch_gemm_df(const ConvParams& p, hipStream_t s) {
  LDIFF_CHECK(p.w_frag != nullptr, LDIFF_ERR_INVALID, "gemm (dataflow): the caller did not provide the fragment-packed weights (launch_pack_gemm_frag)");
  const double bytes = (double)p.M * p.K * 2.0 + (double)p.N * p.K * 2.0 + (double)p.M * (p.geglu ? p.N / 2 : p.N) * (p.y_lo ? 4.0 : 2.0) + (p.res ? (double)p.M * p.N * (p.res_lo ? 4.0 : 2.0) : 0.0);
  ProfScope prof(p.geglu ? "gemm_df<geglu>" : "gemm_df", 2.0 * p.M * (double)p.N * p.K, bytes, s);
  const int fl = (p.res ? FG_RES : 0) | (p.res && p.res_lo ? FG_RES_SPLIT : 0) | (p.y_lo ? FG_OUT_SPLIT : 0) | (p.geglu ? FG_GEGLU : 0) | (p.stats ? FG_STATS : 0);
  switch (fl) {
    case 0: launch_df_f<0>(p, s); break;
    case FG_RES: launch_df_f<FG_RES>(p, s); break;
    case FG_OUT_SPLIT: launch_df_f<FG_OUT_SPLIT>(p, s); break;
    case FG_RES | FG_OUT_SPLIT: launch_df_f<FG_RES | FG_OUT_SPLIT>(p, s); break;
    case FG_RES | FG_RES_SPLIT: launch_df_f<FG_RES | FG_RES_SPLIT>(p, s); break;
    case FG_RES | FG_RES_SPLIT | FG_OUT_SPLIT: launch_df_f<FG_RES | FG_RES_SPLIT | FG_OUT_SPLIT>(p, s); break;
    case FG_GEGLU: launch_df_f<FG_GEGLU>(p, s); break;
    // fused statistics: the layers that ask for them (proj_out of Transformer2DModel: split residual, split output; plain graphs: plain residual)
    case FG_STATS | FG_RES | FG_RES_SPLIT | FG_OUT_SPLIT: launch_df_f<FG_STATS | FG_RES | FG_RES_SPLIT | FG_OUT_SPLIT>(p, s); break;
    case FG_STATS | FG_RES: launch_df_f<FG_STATS | FG_RES>(p, s); break;
    case FG_STATS | FG_OUT_SPLIT: launch_df_f<FG_STATS | FG_OUT_SPLIT>(p, s); break;
    case FG_STATS: launch_df_f<FG_STATS>(p, s); break;
    default: LDIFF_CHECK(false, LDIFF_ERR_INVALID, "gemm (dataflow): unsupported epilogue %d", fl);
  }
}
