// Host-side executors: weight store, op helpers over the device arena, UNet / VAE graphs.
// Internal to libldiff_hip.so.
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ldiff.h"
#include "common.h"

// NHWC fp16 activation (tokens [M, C] are B=1,H=1,W=M or keep the image shape).  A SPLIT activation stores every row as
// [hi(C) | lo(C)] with value = hi + lo (~22 significant bits): the residual stream is kept this way so that
// the reference's fp32 residual adds survive ~40 chained blocks; plain consumers read the hi half with row pitch 2C, split
// consumers (the contractions that carry the whole stream: shortcut / proj_in / proj_out / resampling convs) read all 2C
// channels against weights duplicated along K.
struct Act {
  f16* p = nullptr;
  int B = 0, H = 0, W = 0, C = 0;
  bool split = false;
  bool lo8 = false;      // split with an fp8 lo half: a row is [C fp16 | C e4m3 of lo * 2^LO8_SHIFT] = 3C bytes (ConvParams::lo8_slab0); conv operands only
  float* st = nullptr;   // producer-fused GroupNorm partial statistics [B][st_R][C][2] (nullptr: none)
  int st_R = 0;
  long long rows() const { return (long long)B * H * W; }
  int ld() const { return lo8 ? C + C / 2 : (split ? 2 * C : C); }       // row pitch in elements
  int lo() const { return split ? C : 0; }           // offset of the lo half inside a row
  size_t bytes() const { return (size_t)rows() * ld() * sizeof(f16); }
  SrcView view() const { return SrcView{p, C, ld(), lo()}; }
};

struct Derived { f16* p = nullptr; int gen = -1; int key = 0; };   // lazily built weights derived from a checkpoint matrix

struct MatW {   // [Nrows][K] fp16 K-major + fp32 bias
  f16* w = nullptr;
  float* b = nullptr;  // nullptr => no bias
  int N = 0, Nrows = 0, K = 0, ks = 1, Cin = 0;  // Cin = padded input channels (K = ks*ks*Cin)
  bool geglu = false;   // rows stored x/gate-interleaved by 16 so that the GEMM epilogue can apply x * gelu(gate) (ConvParams::geglu)
  int Cin_logical = 0;            // unpadded input channels of a first-layer conv (Cin padded to 8): the split form keeps hi | lo inside the pad
  // derived weights, built on first use by the Exec that needs them and rebuilt when the checkpoint is reloaded:
  mutable Derived par;            //   upsampler convs: parity weights [4][Nrows][4*Cin]
  mutable Derived dup;            //   split operand: [Nrows][taps][2*Cin] (same weights against the hi and the lo half); key = C1 of a concat
  mutable Derived dup_par;        //   parity weights of the duplicated matrix
  mutable Derived frag;           //   MFMA-fragment-packed copy for the dataflow conv3x3 kernel (kernels_conv3x3d.hip)
  mutable Derived frag_par;       //   ... of the parity weights (upsampling convs on the dataflow kernel)
  mutable Derived frag_sc;        //   ... with a folded shortcut's weights behind the nine taps (key = the shortcut matrix's address bits) and the summed bias
  mutable float* bias_sc = nullptr;
  mutable Derived gfrag;          //   MFMA-fragment-packed copy for the dataflow GEMM (kernels_gemm_df.hip)
  mutable Derived gfrag_dup;      //   the same of the duplicated (split-operand) matrix; key = C1 of a concat
  mutable Derived tiled;          //   panel-tiled copy for the LayerNorm-fused GEMM (kernels_gemm_ast.hip)
  mutable Derived lo8;            //   split operand with an fp8 lo half: [Nrows][taps][Cin fp16 | Cin e4m3] + one int (the E8M0 scale operand) behind it
};
struct NormW { float* g = nullptr; float* b = nullptr; int C = 0; };
struct GNss { float* scale = nullptr; float* shift = nullptr; };

// One expected checkpoint tensor and where/how it lands on the device.
struct LoadSpec {
  enum Kind { MATRIX, VECTOR } kind;
  std::vector<int64_t> shape;   // expected torch shape
  f16* mat = nullptr; int row_off = 0, K = 0, ks = 1, Cin_pad = 0;   // MATRIX
  float* vec = nullptr; int vec_off = 0;                             // VECTOR
  int geglu_half = 0;   // > 0: GEGLU projection of width 2*geglu_half: row r lands at geglu_row(r) (x / gate interleaved by 16 rows)
  bool loaded = false;
};

class WeightStore {
 public:
  ~WeightStore();
  // allocation helpers (device memory owned by the store, zero-initialised)
  f16* alloc_mat(int Nrows, int K);
  float* alloc_vec(int n);
  // registration
  MatW add_conv(const std::string& prefix, int Cin, int Cout, int ks, bool bias = true, int Cin_pad = -1, int min_rows = 0, bool geglu = false);
  void add_rows(const std::string& wname, const std::string& bname, f16* mat, int K, int ks, int Cin, int Cin_pad, int row_off, int rows,
                float* bias_vec, bool has_bias);
  NormW add_norm(const std::string& prefix, int C);
  void alias(const std::string& alias_name, const std::string& name);
  // loading
  void load(const char* name, const void* host, int dtype, const int64_t* shape, int ndim);
  int missing() const;
  const char* missing_name(int i) const;
  int generation = 0;   // bumped by every load(): derived weights are rebuilt when it changes

 private:
  std::unordered_map<std::string, LoadSpec> specs_;
  std::unordered_map<std::string, std::string> alias_;
  std::vector<std::string> order_;
  std::vector<void*> allocs_;
  mutable std::vector<std::string> missing_cache_;
};

// Sticky non-finite detector of a handle (include/ldiff.h "Non-finite detection").  One int per graph in host-mapped pinned memory: the device sets it
// (a plain store of 1 from a GroupNorm finalize workgroup whose totals are not finite), the host reads it without a device round trip.
struct NonFiniteFlag {
  int* words = nullptr;   // [4], host pointer == device pointer (hipHostMallocMapped under unified addressing)
  void create();
  void destroy();
  bool test_and_clear();  // true if any word was set (only meaningful for work that has completed)
};

struct ConvOpts {
  int stride = 1, pad_t = -1 /* -1 => (ks-1)/2 */, pad_l = -1, ups = 0;
  int Hout = -1, Wout = -1;     // override output size (asymmetric-pad downsample)
  const GNss* gn = nullptr; int silu = 0;
  const float* temb = nullptr; int ld_temb = 0;
  const Act* res = nullptr;      // residual operand (plain or split)
  bool split_in = false;         // consume the (split) sources as a split operand: K doubled, duplicated weights
  bool split_out = false;        // write the output as a split activation
  void* out_f32 = nullptr; int ldy_f32 = 0;   // write fp32 [M, ldy] here instead of allocating an fp16 Act
  int N_override = 0;           // columns to store (multiple of 4), default = roundup4(w.N)
  int ldy = 0;                  // fp16 output channel stride (default N stored rounded up to 8)
  bool want_stats = false;      // also emit GroupNorm partial statistics of the output (consumed by Exec::gn)
  bool geglu = false;           // apply x * gelu(gate) in the epilogue (weights must be MatW::geglu); output has N/2 channels
  // decode_latents tail in the epilogue (VAE conv_out): applied iff the narrow-output kernel takes the launch; *post_done says whether it did
  float* post_img = nullptr; uint8_t* post_rgb = nullptr; uint8_t* post_luma = nullptr; int post_slots = 0, post_slot = 0; bool post_only = false;
  bool* post_done = nullptr;
  // the block's 1x1 conv_shortcut folded into this (its second) conv where the dataflow conv3x3 kernel takes the launch (ConvParams::xs): sc_x = the block's
  // input, sc_w = the shortcut's weights; *sc_done says whether the fold happened (else the caller runs the shortcut conv and passes its output as res)
  const Act* sc_x = nullptr; const struct MatW* sc_w = nullptr; bool* sc_done = nullptr;
};

class Exec {
 public:
  Arena arena;
  hipStream_t s = nullptr;
  float* gn_partial = nullptr;
  size_t gn_partial_cap = 0;
  std::vector<void*> owned;   // lazily built derived weights (parity weights of the upsampler convs)
  const int* weights_gen = nullptr;   // -> WeightStore::generation of the owning model
  bool short_runs = false;            // this graph runs beside another stream's (ConvParams::short_runs)
  int* nonfinite = nullptr;           // -> the owning handle's sticky non-finite flag (host-mapped; set by the GroupNorm finalize kernels, NonFiniteFlag below)
  const char* trace_tag = nullptr;    // LDIFF_TRACE_ABSMAX=1: name of the graph whose stages trace() reports (diagnostic, synchronises)
  void trace(const char* stage, const Act& a);   // max |value| of a stage's output to stderr when LDIFF_TRACE_ABSMAX is set; otherwise nothing
  ~Exec();
  void ensure_gn_partial(size_t bytes);
  Act new_act(int B, int H, int W, int C, bool split = false, bool lo8 = false);
  const f16* derived_dup(const MatW& w, int C1_logical, int C2_logical);
  const f16* derived_par(const MatW& w, const f16* src, int Cin, Derived& d);
  const f16* derived_frag(const MatW& w, const ConvParams& p);
  const f16* derived_frag_par(const MatW& w, const ConvParams& p);   // ... of the parity-folded weights p.w_par (ups = 1)
  const f16* derived_frag_sc(const MatW& w, const MatW& sc, const ConvParams& p, const float** bias_sum);   // ... + the folded shortcut
  const f16* derived_tiled(const MatW& w, int N);
  const f16* derived_gfrag(const MatW& w, const f16* src, int K, Derived& d, int key);   // fragment-packed copy of `src` [Nrows][K] for the dataflow GEMM
  const f16* derived_lo8(const MatW& w, const int** scale);   // fp8-lo weights of a split operand + the device int holding their E8M0 scale operand
  bool lo8_conv_ok(const MatW& w, const Act& x, bool res, bool split_out) const;   // would conv(w, norm_apply(x) with an fp8 lo half) run on the ping-pong kernel?
  Act norm_apply(const Act& x, const Act* x2, const GNss& g, bool silu, bool split_out, bool lo8 = false);
  void release(Act& a);
  template <typename T> T* tmp(size_t n) { return reinterpret_cast<T*>(arena.alloc(n * sizeof(T))); }
  GNss gn(const Act& x, const Act* x2, const NormW& w, int groups, float eps);
  void release(GNss& g);
  Act conv(const MatW& w, const Act& x, const Act* x2, const ConvOpts& o);
  Act layernorm(const Act& x, const NormW& w);
  // LayerNorm + linear (+ GEGLU) in one launch where the activation-stationary kernel takes the shape (kernels_gemm_ast.hip), else the two launches
  // qcols > 0: ask for columns [0, qcols) multiplied by qscale before the rounding; *scaled says whether the fused kernel took the launch and did it
  Act ln_linear(const MatW& w, const Act& x, const NormW& ln, bool geglu, int qcols = 0, float qscale = 1.0f, bool* scaled = nullptr);
  Act geglu(const Act& x);
  // ResnetBlock2D (UNet: with time embedding and optional skip concat; VAE: neither) under storage policy `prec`
  Act resnet(const struct ResnetW& r, const Act& x, const Act* skip, const float* temb, int ld_temb, int groups, float eps, int prec);
};

// ---- UNet -------------------------------------------------------------------------------------
struct ResnetW { NormW n1, n2; MatW c1, c2, sc; bool has_sc = false; int temb_off = 0; int Cin = 0, Cout = 0; };
struct TransformerW {
  NormW gn, ln1, ln2, ln3;
  MatW proj_in, qkv, out1, q2, kv2, out2, ff1, ff2, proj_out;
  int C = 0;
  f16* kv_ctx = nullptr;  // [Bctx*L, 2C] precomputed by set_context
};

// Storage policy of a graph (ldiff_*_set_precision):
//   0  everything fp16 in HBM (fastest; one UNet pass ~2e-3 of the output range from the fp32 reference)
//   1  split residual stream: stream tensors hi|lo, residual adds to fp32 round-off, stream-carrying contractions on split operands
//   2  every conv / linear operand split (K doubled everywhere): ~1e-4; used for the VAE encoder, whose error every later pass inherits
enum { PREC_FAST = 0, PREC_STREAM = 1, PREC_FULL = 2 };

struct ldiff_unet {
  ldiff_unet_cfg cfg;
  int device = 0;
  int precision = PREC_STREAM;
  WeightStore ws;
  Exec ex;
  MatW conv_in, conv_out, t_lin1, t_lin2, temb_proj_all;
  NormW norm_out;
  int temb_total = 0;
  std::vector<std::vector<ResnetW>> down_res, up_res;
  std::vector<std::vector<TransformerW>> down_attn, up_attn;
  std::vector<MatW> down_sample, up_sample;
  std::vector<bool> has_down, has_up;
  ResnetW mid_res[2];
  TransformerW mid_attn;
  std::vector<TransformerW*> all_tf;
  NonFiniteFlag nf;
  int ctx_B = 0, ctx_L = 0, ctx_gen = 0;
  f16* ctx_buf = nullptr; size_t ctx_cap = 0;     // all kv_ctx live in one allocation
  void build();
  void set_context(const float* ctx, int Bc, int L, hipStream_t s);
  // forward = the ~390-450 launches of one pass (384 at B = 8, 443 at B = 1 at SD-v1.5 size).  With graphs on (default) the launch sequence of a (B, h, w, precision, context)
  // configuration is captured into a hipGraph on its second use and replayed afterwards: input, timestep and output go through
  // handle-owned staging buffers, so the replay is valid for any caller pointers and any timestep.
  void forward(const float* x, int B, int h, int w, float t, float* out, hipStream_t s);
  void forward_impl(const float* x, int B, int h, int w, float t, const float* t_dev, float* out, hipStream_t s);
  // ControlNet inputs of the next forward (down_block_additional_residuals, mid_block_additional_residual: segmentor.py:366-372);
  // float32 NCHW device pointers in skip-stack order, consumed (cleared) by that forward, which then runs eagerly
  std::vector<const float*> extra_down;
  const float* extra_mid = nullptr;
  int n_skips() const;
  struct GraphCache {
    bool enabled = true;
    int uses = 0;                       // forwards seen with the current key (0: none, 1: ran eagerly once, >= 2: graph ready)
    long long key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap_stream = nullptr;
    float *in = nullptr, *out = nullptr, *t = nullptr;
    size_t in_cap = 0;
    long long replays = 0, captures = 0, nodes = 0;   // nodes: kernel launches of the captured forward
    void drop();
  } gc;
  ~ldiff_unet();
  Act transformer(const TransformerW& t, const Act& x);
};

// ---- VAE --------------------------------------------------------------------------------------
struct VaeAttnW { NormW gn; MatW qkv, out; int C = 0; };
struct ldiff_vae {
  ldiff_vae_cfg cfg;
  int device = 0;
  int prec_enc = PREC_FULL, prec_dec = PREC_FAST;   // the decoder feeds only uint8 images / luma (never the latents): see DESIGN.md section 3
  int prec() const { return cur == &ex_enc ? prec_enc : prec_dec; }
  WeightStore ws;
  // Two workspaces: the decoder's and the encoder's.  A pipelined sampler decodes batch k on the side stream while the encoder
  // of batch k+1 already runs on the caller's stream; ex() is the one the running graph builder uses.
  Exec ex_dec, ex_enc;
  NonFiniteFlag nf;   // word 0: encoder graph, word 1: decoder graph
  ~ldiff_vae() { nf.destroy(); }
  Exec* cur = &ex_dec;
  Exec& ex() { return *cur; }
  // side stream for decodes that only feed the feature tensor (ldiff_sample); ev_side = "everything queued on it so far is done"
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_side = nullptr;
  bool side_used = false;
  void wait_side(hipStream_t s);   // make s wait for the side stream's queued work (no-op if it was never used)
  // encoder
  MatW e_conv_in, e_conv_out, quant;
  std::vector<std::vector<ResnetW>> e_res;
  std::vector<MatW> e_down;
  ResnetW e_mid[2]; VaeAttnW e_attn; NormW e_norm_out;
  // decoder
  MatW post_quant, d_conv_in, d_conv_out;
  std::vector<std::vector<ResnetW>> d_res;
  std::vector<MatW> d_up;
  ResnetW d_mid[2]; VaeAttnW d_attn; NormW d_norm_out;
  void build();
  void encode(const float* x, int B, int H, int W, float* moments, hipStream_t s);
  // writes the fp32 NHWC decoder output [B*8h*8w, 4] into the arena and post-processes it
  void decode(const float* z, int B, int h, int w, float z_scale, float* sample_nchw, float* image_nhwc, uint8_t* rgb, uint8_t* luma,
              int n_slots, int slot, hipStream_t s);
  Act mid_attention(const VaeAttnW& a, const Act& x);
};

struct ldiff_pipeline {
  ldiff_unet* unet;
  ldiff_vae* vae;
  Arena arena;   // latents / eps history
  float abar[1000];
  // decode side stream: the VAE decode of pass k (needed only for the features) runs beside the UNet pass k+1
  // 0: everything on the caller's stream; 1: decodes on the VAE's side stream, the caller's stream joins before ldiff_sample
  // returns (default); 2: as 1 but the join is deferred to ldiff_pipeline_join (lets the next batch start under the decodes)
  int overlap = 1;
  bool join_pending = false;
  hipEvent_t ev_latents = nullptr, ev_decoded = nullptr;
};

void pndm_alphas_cumprod(float* out1000);
int plms_timesteps(int n_passes, int64_t* out, int cap);
