// Host-side executors for the UNet / VAE graphs of the sampling path (gfx950, MI355X).
// Graph structure follows diffusers 0.34.0 UNet2DConditionModel / AutoencoderKL as restated in
// SURVEY.md 8a R1-R5; every contraction runs in kernels_igemm.hip / kernels_attn.hip, every GroupNorm
// is a statistics pass (kernels_norm.hip) whose apply+SiLU is folded into the consuming conv's load,
// nearest-2x upsampling and the skip concat are gathers inside the conv (no copies).
#include "model.h"

#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <cmath>

// ---- error message (thread-local) --------------------------------------------------------------
static thread_local char g_err[1024] = "";
void ldiff_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* ldiff_error_message() { return g_err; }

static inline int roundup(int x, int m) { return (x + m - 1) / m * m; }

// ---- per-(device, kernel) launch attribute ---------------------------------------------------------
#include <mutex>
#include <set>
void ensure_dyn_smem(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  HIP_CHECK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({dev, kernel})) return;
  HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({dev, kernel});
}

// ---- Arena -------------------------------------------------------------------------------------
Arena::~Arena() {
  if (base_) (void)hipFree(base_);
}
void Arena::reserve(size_t bytes) {
  if (bytes <= cap_) return;
  if (base_) {
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipFree(base_));
    base_ = nullptr;
    cap_ = 0;
  }
  HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&base_), bytes));
  cap_ = bytes;
  reset();
}
void Arena::reset() {
  blocks_.clear();
  if (cap_) blocks_.push_back({0, cap_, false});
}
void* Arena::alloc(size_t bytes) {
  bytes = (bytes + 255) & ~size_t(255);
  if (bytes == 0) bytes = 256;
  for (size_t i = 0; i < blocks_.size(); ++i) {
    if (!blocks_[i].used && blocks_[i].size >= bytes) {
      if (blocks_[i].size > bytes) {
        Block rest{blocks_[i].off + bytes, blocks_[i].size - bytes, false};
        blocks_[i].size = bytes;
        blocks_.insert(blocks_.begin() + i + 1, rest);
      }
      blocks_[i].used = true;
      high_ = std::max(high_, blocks_[i].off + bytes);
      return base_ + blocks_[i].off;
    }
  }
  ldiff_set_error("device arena exhausted: need %zu bytes, capacity %zu", bytes, cap_);
  throw LdiffError{LDIFF_ERR_RUNTIME};
}
void Arena::free(void* p) {
  if (!p) return;
  size_t off = (size_t)((char*)p - base_);
  for (size_t i = 0; i < blocks_.size(); ++i) {
    if (blocks_[i].off == off && blocks_[i].used) {
      blocks_[i].used = false;
      if (i + 1 < blocks_.size() && !blocks_[i + 1].used) { blocks_[i].size += blocks_[i + 1].size; blocks_.erase(blocks_.begin() + i + 1); }
      if (i > 0 && !blocks_[i - 1].used) { blocks_[i - 1].size += blocks_[i].size; blocks_.erase(blocks_.begin() + i); }
      return;
    }
  }
  ldiff_set_error("arena: free of unknown pointer");
  throw LdiffError{LDIFF_ERR_RUNTIME};
}

// ---- WeightStore -------------------------------------------------------------------------------
WeightStore::~WeightStore() {
  for (void* p : allocs_) (void)hipFree(p);
}
f16* WeightStore::alloc_mat(int Nrows, int K) {
  void* p = nullptr;
  size_t bytes = (size_t)Nrows * K * sizeof(f16);
  HIP_CHECK(hipMalloc(&p, bytes));
  HIP_CHECK(hipMemset(p, 0, bytes));
  allocs_.push_back(p);
  return (f16*)p;
}
float* WeightStore::alloc_vec(int n) {
  void* p = nullptr;
  HIP_CHECK(hipMalloc(&p, (size_t)n * sizeof(float)));
  HIP_CHECK(hipMemset(p, 0, (size_t)n * sizeof(float)));
  allocs_.push_back(p);
  return (float*)p;
}
void WeightStore::add_rows(const std::string& wname, const std::string& bname, f16* mat, int K, int ks, int Cin, int Cin_pad, int row_off,
                           int rows, float* bias_vec, bool has_bias) {
  LoadSpec w;
  w.kind = LoadSpec::MATRIX;
  w.shape = {rows, Cin, ks, ks};
  w.mat = mat; w.row_off = row_off; w.K = K; w.ks = ks; w.Cin_pad = Cin_pad;
  specs_[wname] = w;
  order_.push_back(wname);
  if (has_bias) {
    LoadSpec b;
    b.kind = LoadSpec::VECTOR;
    b.shape = {rows};
    b.vec = bias_vec; b.vec_off = row_off;
    specs_[bname] = b;
    order_.push_back(bname);
  }
}
MatW WeightStore::add_conv(const std::string& prefix, int Cin, int Cout, int ks, bool bias, int Cin_pad, int min_rows, bool geglu) {
  if (Cin_pad < 0) Cin_pad = roundup(Cin, 8);
  MatW m;
  m.N = Cout; m.Nrows = roundup(std::max(Cout, min_rows), 16); m.ks = ks; m.Cin = Cin_pad; m.K = ks * ks * Cin_pad;
  m.Cin_logical = 2 * Cin <= Cin_pad ? Cin : 0;   // hi | lo of a <= 4-channel input fit into the 8 padded channels
  m.w = alloc_mat(m.Nrows, m.K);
  m.b = bias ? alloc_vec(m.Nrows) : nullptr;
  add_rows(prefix + ".weight", prefix + ".bias", m.w, m.K, ks, Cin, Cin_pad, 0, Cout, m.b, bias);
  if (geglu) {   // Linear(C, 2*half) whose output is [x | gate]: store x rows 16j..16j+15 at 32j.., gate rows 16j.. at 32j+16..
    LDIFF_CHECK(Cout % 32 == 0, LDIFF_ERR_INVALID, "geglu projection width %d must be a multiple of 32", Cout);
    m.geglu = true;
    specs_[prefix + ".weight"].geglu_half = Cout / 2;
    if (bias) specs_[prefix + ".bias"].geglu_half = Cout / 2;
  }
  return m;
}
static inline int geglu_row(int r, int half) { const int q = r < half ? r : r - half; return (q / 16) * 32 + (r < half ? 0 : 16) + q % 16; }
NormW WeightStore::add_norm(const std::string& prefix, int C) {
  NormW n;
  n.C = C; n.g = alloc_vec(C); n.b = alloc_vec(C);
  LoadSpec g; g.kind = LoadSpec::VECTOR; g.shape = {C}; g.vec = n.g; g.vec_off = 0;
  LoadSpec b = g; b.vec = n.b;
  specs_[prefix + ".weight"] = g; order_.push_back(prefix + ".weight");
  specs_[prefix + ".bias"] = b; order_.push_back(prefix + ".bias");
  return n;
}
void WeightStore::alias(const std::string& alias_name, const std::string& name) { alias_[alias_name] = name; }

static inline float host_to_float(const void* p, int dtype, size_t i) {
  if (dtype == LDIFF_F32) return ((const float*)p)[i];
  if (dtype == LDIFF_F16) return (float)((const f16*)p)[i];
  uint32_t u = (uint32_t)((const uint16_t*)p)[i] << 16;  // bf16
  float f;
  memcpy(&f, &u, 4);
  return f;
}

void WeightStore::load(const char* name_c, const void* host, int dtype, const int64_t* shape, int ndim) {
  LDIFF_CHECK(name_c && host && shape, LDIFF_ERR_INVALID, "load: null argument");
  LDIFF_CHECK(dtype == LDIFF_F32 || dtype == LDIFF_F16 || dtype == LDIFF_BF16, LDIFF_ERR_INVALID, "load(%s): unsupported dtype %d", name_c, dtype);
  std::string name(name_c);
  auto al = alias_.find(name);
  if (al != alias_.end()) name = al->second;
  auto it = specs_.find(name);
  LDIFF_CHECK(it != specs_.end(), LDIFF_ERR_INVALID, "load: unexpected tensor name '%s'", name_c);
  LoadSpec& sp = it->second;
  size_t numel = 1, expect = 1;
  for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
  for (auto d : sp.shape) expect *= (size_t)d;
  bool ok = numel == expect && ndim >= 1 && shape[0] == sp.shape[0];
  if (sp.kind == LoadSpec::MATRIX) ok = ok && ndim >= 2 && shape[1] == sp.shape[1] && (ndim == 4 || (ndim == 2 && sp.ks == 1));
  else ok = ok && ndim == 1;
  if (!ok) {
    std::string got;
    for (int i = 0; i < ndim; ++i) got += (i ? "," : "") + std::to_string((long long)shape[i]);
    std::string want;
    for (size_t i = 0; i < sp.shape.size(); ++i) want += (i ? "," : "") + std::to_string((long long)sp.shape[i]);
    ldiff_set_error("load(%s): shape [%s] does not match expected [%s]", name_c, got.c_str(), want.c_str());
    throw LdiffError{LDIFF_ERR_INVALID};
  }
  if (sp.kind == LoadSpec::VECTOR) {
    std::vector<float> tmp(numel);
    for (size_t i = 0; i < numel; ++i) tmp[sp.geglu_half ? (size_t)geglu_row((int)i, sp.geglu_half) : i] = host_to_float(host, dtype, i);
    HIP_CHECK(hipMemcpy(sp.vec + sp.vec_off, tmp.data(), numel * sizeof(float), hipMemcpyHostToDevice));
  } else {
    const int rows = (int)sp.shape[0], Cin = (int)sp.shape[1], ks = sp.ks, taps = ks * ks;
    std::vector<f16> tmp((size_t)rows * sp.K, (f16)0.f);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < Cin; ++c)
        for (int t = 0; t < taps; ++t)
          tmp[(size_t)(sp.geglu_half ? geglu_row(r, sp.geglu_half) : r) * sp.K + (size_t)t * sp.Cin_pad + c] =
              (f16)host_to_float(host, dtype, ((size_t)r * Cin + c) * taps + t);
    HIP_CHECK(hipMemcpy(sp.mat + (size_t)sp.row_off * sp.K, tmp.data(), tmp.size() * sizeof(f16), hipMemcpyHostToDevice));
  }
  sp.loaded = true;
  ++generation;
}
int WeightStore::missing() const {
  missing_cache_.clear();
  for (auto& n : order_)
    if (!specs_.at(n).loaded) missing_cache_.push_back(n);
  return (int)missing_cache_.size();
}
const char* WeightStore::missing_name(int i) const {
  if (i < 0 || i >= (int)missing_cache_.size()) return "";
  return missing_cache_[i].c_str();
}

// ---- non-finite flag ------------------------------------------------------------------------------
void NonFiniteFlag::create() {
  if (words) return;
  HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&words), 4 * sizeof(int), hipHostMallocMapped));
  for (int i = 0; i < 4; ++i) words[i] = 0;
}
void NonFiniteFlag::destroy() {
  if (words) (void)hipHostFree(words);
  words = nullptr;
}
bool NonFiniteFlag::test_and_clear() {
  if (!words) return false;
  bool any = false;
  for (int i = 0; i < 4; ++i) {
    volatile int* w = words + i;
    if (*w) { any = true; *w = 0; }
  }
  return any;
}

// ---- Exec: op helpers --------------------------------------------------------------------------
void launch_absmax(const f16* x, long long rows, int C, int ld, int lo, float* out, hipStream_t s);   // kernels_elem.hip
void Exec::trace(const char* stage, const Act& a) {
  static const bool on = [] { const char* e = getenv("LDIFF_TRACE_ABSMAX"); return e && atoi(e) != 0; }();
  if (!on || !a.p || a.lo8) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (s) (void)hipStreamIsCapturing(s, &cs);
  if (cs != hipStreamCaptureStatusNone) return;
  float* d = tmp<float>(1);
  launch_absmax(a.p, a.rows(), a.C, a.ld(), a.lo(), d, s);
  float h = 0.f;
  HIP_CHECK(hipMemcpyAsync(&h, d, sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  arena.free(d);
  fprintf(stderr, "[absmax] %-8s %-28s [%d,%d,%d,%d]%s max|x| = %.6g (fp16 limit 65504)\n", trace_tag ? trace_tag : "", stage, a.B, a.H, a.W, a.C, a.split ? " split" : "", (double)h);
}
Exec::~Exec() {
  if (gn_partial) (void)hipFree(gn_partial);
  for (void* q : owned) (void)hipFree(q);
}
void Exec::ensure_gn_partial(size_t bytes) {
  if (bytes <= gn_partial_cap) return;
  if (gn_partial) { HIP_CHECK(hipDeviceSynchronize()); HIP_CHECK(hipFree(gn_partial)); gn_partial = nullptr; }
  HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&gn_partial), bytes));
  gn_partial_cap = bytes;
}
Act Exec::new_act(int B, int H, int W, int C, bool split, bool lo8) {
  Act a;
  a.B = B; a.H = H; a.W = W; a.C = C; a.split = split || lo8; a.lo8 = lo8;
  a.p = (f16*)arena.alloc(a.bytes());
  return a;
}
void Exec::release(Act& a) {
  if (a.p) arena.free(a.p);
  if (a.st) arena.free(a.st);
  a.p = nullptr;
  a.st = nullptr;
}
GNss Exec::gn(const Act& x, const Act* x2, const NormW& w, int groups, float eps) {
  const int C = x.C + (x2 ? x2->C : 0);
  LDIFF_CHECK(C == w.C, LDIFF_ERR_INVALID, "group norm: %d channels, weight has %d", C, w.C);
  GNss g;
  g.scale = tmp<float>((size_t)x.B * C);
  g.shift = tmp<float>((size_t)x.B * C);
  if (x.st && (!x2 || x2->st)) {   // statistics were accumulated by the producing kernels: finalize only, no extra read
    launch_gn_finalize(x.st, x.st_R, x.C, x2 ? x2->st : nullptr, x2 ? x2->st_R : 0, x2 ? x2->C : 0, x.B, x.H * x.W, groups, eps, w.g, w.b,
                       g.scale, g.shift, s, nonfinite);
    return g;
  }
  LDIFF_CHECK(gn_partial_bytes(x.B, x.H * x.W, C) <= gn_partial_cap, LDIFF_ERR_RUNTIME, "group norm workspace too small");
  launch_gn_stats(x.view(), x2 ? x2->view() : SrcView{nullptr, 0, 0, 0}, x.B, x.H * x.W, groups, eps, w.g, w.b, gn_partial, gn_partial_cap, g.scale,
                  g.shift, s, nonfinite);
  return g;
}
void Exec::release(GNss& g) {
  arena.free(g.scale);
  arena.free(g.shift);
  g.scale = g.shift = nullptr;
}
// duplicated weights of a split-operand contraction: per tap [a(C1) a(C1) b(C2) b(C2)] from [a(C1) b(C2)]; first-layer convs whose
// input channels are padded to 8 keep hi | lo inside the pad ([a(c) a(c) 0..] with c = Cin_logical)
const f16* Exec::derived_dup(const MatW& w, int C1, int C2) {
  const int gen = weights_gen ? *weights_gen : 0;
  const bool small = w.Cin_logical > 0;
  const int dst_stride = small ? w.Cin : 2 * w.Cin;
  if (!w.dup.p) {
    void* q = nullptr;
    HIP_CHECK(hipMalloc(&q, (size_t)w.Nrows * w.ks * w.ks * dst_stride * sizeof(f16)));
    owned.push_back(q);
    w.dup.p = (f16*)q;
  }
  if (w.dup.gen != gen || w.dup.key != C1) {
    launch_dup_weights(w.w, w.dup.p, w.Nrows, w.ks * w.ks, w.Cin, small ? w.Cin_logical : C1, small ? 0 : C2, dst_stride, s);
    w.dup.gen = gen; w.dup.key = C1;
  }
  return w.dup.p;
}
// split operand with an fp8 lo half (ConvParams::lo8_slab0): per (row, tap) [Cin fp16 | Cin e4m3 of w * 2^sw]; the int behind the matrix is 127 - sw
const f16* Exec::derived_lo8(const MatW& w, const int** scale) {
  const int gen = weights_gen ? *weights_gen : 0;
  const size_t bytes = (size_t)w.Nrows * w.ks * w.ks * w.Cin * 3;
  if (!w.lo8.p) {
    void* q = nullptr;
    HIP_CHECK(hipMalloc(&q, bytes + 16));
    owned.push_back(q);
    w.lo8.p = (f16*)q;
  }
  int* sc = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(w.lo8.p) + bytes);
  if (w.lo8.gen != gen) {
    launch_lo8_weights(w.w, w.lo8.p, sc, w.Nrows, w.ks * w.ks, w.Cin, s);
    w.lo8.gen = gen;
  }
  *scale = sc;
  return w.lo8.p;
}
// LDIFF_LO8: 1 (default) = the lo half of a split conv operand travels as fp8 where the 16 x 16 ping-pong kernel takes it, 0 = fp16 lo halves everywhere
bool Exec::lo8_conv_ok(const MatW& w, const Act& x, bool res, bool split_out) const {
  static const int mode = [] { const char* e = getenv("LDIFF_LO8"); return e ? atoi(e) : 1; }();
  if (!mode || w.ks != 3 || x.C != w.Cin || x.C % 128 != 0 || w.Cin_logical > 0 || !split_out) return false;
  ConvParams p;
  memset(&p, 0, sizeof(p));
  static const int one = 127;
  p.x = x.p; p.C1 = x.C + x.C / 2; p.lo8_slab0 = x.C / 64; p.lo8_sb = 127 - LO8_SHIFT; p.lo8_sa = &one;
  p.B = x.B; p.Hin = p.Hout = x.H; p.Win = p.Wout = x.W; p.ks = 3; p.stride = 1; p.pad_t = p.pad_l = 1;
  p.w = w.w; p.N = roundup(w.N, 4); p.n_real = w.N; p.Nrows = w.Nrows; p.K = 9 * p.C1; p.M = x.B * x.H * x.W;
  if (p.N != w.N || p.N % 8 != 0) return false;
  p.y = x.p; p.ldy = 2 * p.N; p.y_lo = p.N; p.stats = reinterpret_cast<float*>(x.p);   // (placeholders: only null / non-null and the layout matter)
  if (res) { p.res = x.p; p.ld_res = 2 * p.N; p.res_lo = p.N; }
  return conv3x3_eligible(p) && conv3x3_splitk_plan(p) <= 1 && conv3x3p_selected(p);
}
const f16* Exec::derived_frag(const MatW& w, const ConvParams& p) {
  const int gen = weights_gen ? *weights_gen : 0;
  if (!w.frag.p) {
    void* q = nullptr;
    HIP_CHECK(hipMalloc(&q, conv3x3d_frag_bytes(p)));
    owned.push_back(q);
    w.frag.p = (f16*)q;
  }
  if (w.frag.gen != gen) {   // first use, or the checkpoint was reloaded since
    launch_pack_frag_weights(w.w, w.frag.p, p.N, p.C1, s);
    w.frag.gen = gen;
  }
  return w.frag.p;
}
const f16* Exec::derived_frag_par(const MatW& w, const ConvParams& p) {
  const int gen = weights_gen ? *weights_gen : 0;
  if (!w.frag_par.p) {
    void* q = nullptr;
    HIP_CHECK(hipMalloc(&q, conv3x3d_frag_bytes(p)));
    HIP_CHECK(hipMemset(q, 0, conv3x3d_frag_bytes(p)));   // (the padding step behind the last block is loaded, never used)
    owned.push_back(q);
    w.frag_par.p = (f16*)q;
  }
  if (w.frag_par.gen != gen) {   // first use, or the checkpoint was reloaded since (p.w_par has been rebuilt by derived_par just before)
    launch_pack_frag_weights_par(p.w_par, w.frag_par.p, p.N, p.Nrows, p.C1, s);
    w.frag_par.gen = gen;
  }
  return w.frag_par.p;
}
const f16* Exec::derived_frag_sc(const MatW& w, const MatW& sc, const ConvParams& p, const float** bias_sum) {
  const int gen = weights_gen ? *weights_gen : 0;
  if (!w.frag_sc.p) {
    void* q = nullptr;
    HIP_CHECK(hipMalloc(&q, conv3x3d_frag_bytes(p)));
    HIP_CHECK(hipMemset(q, 0, conv3x3d_frag_bytes(p)));   // (the padding step behind the shortcut's blocks is loaded, never used)
    owned.push_back(q);
    w.frag_sc.p = (f16*)q;
    HIP_CHECK(hipMalloc(&q, (size_t)w.Nrows * sizeof(float)));
    owned.push_back(q);
    w.bias_sc = (float*)q;
  }
  // key: the shortcut matrix this copy was packed with and its width (a conv2 reused with another shortcut would otherwise read stale / too few weights)
  const int key = (int)((reinterpret_cast<uintptr_t>(sc.w) >> 4) & 0x3fffffff) ^ (p.Cs << 20);
  LDIFF_CHECK(w.frag_sc.gen < 0 || w.frag_sc.key == key, LDIFF_ERR_RUNTIME, "conv: the folded-shortcut weights of this layer were packed for another shortcut matrix");
  if (w.frag_sc.gen != gen) {   // first use, or the checkpoint was reloaded since
    w.frag_sc.key = key;
    launch_pack_frag_weights(w.w, w.frag_sc.p, p.N, p.C1, s);
    launch_pack_frag_weights_sc(sc.w, w.frag_sc.p, p.N, p.C1, p.Cs, sc.K, s);
    launch_add_vectors(w.b, sc.b, w.bias_sc, w.Nrows, s);
    w.frag_sc.gen = gen;
  }
  *bias_sum = w.bias_sc;
  return w.frag_sc.p;
}
const f16* Exec::derived_tiled(const MatW& w, int N) {
  const int gen = weights_gen ? *weights_gen : 0;
  if (!w.tiled.p) {
    void* q = nullptr;
    HIP_CHECK(hipMalloc(&q, (size_t)N * w.K * sizeof(f16)));
    owned.push_back(q);
    w.tiled.p = (f16*)q;
  }
  if (w.tiled.gen != gen) {   // first use, or the checkpoint was reloaded since
    launch_lngemm_tile_weights(w.w, w.tiled.p, N, w.K, s);
    w.tiled.gen = gen;
  }
  return w.tiled.p;
}
const f16* Exec::derived_gfrag(const MatW& w, const f16* src, int K, Derived& d, int key) {
  const int gen = weights_gen ? *weights_gen : 0;
  if (!d.p) {
    void* q = nullptr;
    HIP_CHECK(hipMalloc(&q, (size_t)w.Nrows * K * sizeof(f16)));
    owned.push_back(q);
    d.p = (f16*)q;
  }
  if (d.gen != gen || d.key != key) {   // first use, the checkpoint was reloaded since, or the duplicated source was rebuilt for another concat split
    launch_pack_gemm_frag(src, d.p, w.Nrows, K, s);
    d.gen = gen; d.key = key;
  }
  return d.p;
}
const f16* Exec::derived_par(const MatW& w, const f16* src, int Cin, Derived& d) {
  const int gen = weights_gen ? *weights_gen : 0;
  if (!d.p) {
    void* q = nullptr;
    HIP_CHECK(hipMalloc(&q, (size_t)4 * w.Nrows * 4 * Cin * sizeof(f16)));
    owned.push_back(q);
    d.p = (f16*)q;
  }
  if (d.gen != gen) {   // first use, or the checkpoint was reloaded since
    launch_make_parity_weights(src, d.p, w.Nrows, Cin, s);
    d.gen = gen;
  }
  return d.p;
}
Act Exec::conv(const MatW& w, const Act& x, const Act* x2, const ConvOpts& o) {
  ConvParams p;
  memset(&p, 0, sizeof(p));
  const bool small = w.Cin_logical > 0;   // first-layer conv: <= 4 logical channels inside 8 padded ones (hi | lo fit in the pad)
  LDIFF_CHECK(!o.split_in || (x.split && (!x2 || x2->split) && !o.gn) || (small && !x.split), LDIFF_ERR_INVALID,
              "conv: a split operand needs split sources and no GroupNorm prologue");
  p.x = x.p; p.x2 = x2 ? x2->p : nullptr;
  if (o.split_in && small) {             // hi | lo live inside the 8 padded channels of ONE source
    LDIFF_CHECK(!x2, LDIFF_ERR_INVALID, "conv: first-layer split operand takes one source");
    p.C1 = x.split ? x.ld() : x.C;
  } else if (o.split_in && x.lo8) {      // hi halves fp16, lo halves fp8: 3C/2 "elements" per row (ConvParams::lo8_slab0)
    LDIFF_CHECK(!x2 && x.C % 128 == 0, LDIFF_ERR_INVALID, "conv: an fp8 lo half takes one source with C %% 128 == 0");
    p.C1 = x.C + x.C / 2; p.lo8_slab0 = x.C / 64; p.lo8_sb = 127 - LO8_SHIFT;
  } else if (o.split_in) {               // all 2C channels of each source, K doubled
    p.C1 = 2 * x.C; p.C2 = x2 ? 2 * x2->C : 0;
  } else {                               // hi halves only (row pitch 2C for a split source)
    p.C1 = x.C; p.C2 = x2 ? x2->C : 0;
    p.ld1 = x.split ? x.ld() : 0; p.ld2 = (x2 && x2->split) ? x2->ld() : 0;
  }
  const int Cin_eff = (o.split_in && x.lo8) ? w.Cin + w.Cin / 2 : (o.split_in && !small) ? 2 * w.Cin : w.Cin;
  LDIFF_CHECK(p.C1 + p.C2 == Cin_eff, LDIFF_ERR_INVALID, "conv: input has %d channels, weight expects %d", p.C1 + p.C2, Cin_eff);
  p.B = x.B; p.Hin = x.H; p.Win = x.W;
  p.ks = w.ks; p.stride = o.stride; p.ups = o.ups;
  p.short_runs = short_runs ? 1 : 0;   // (before anything that asks which kernel takes the launch: the choice may depend on it)
  p.pad_t = o.pad_t < 0 ? (w.ks - 1) / 2 : o.pad_t;
  p.pad_l = o.pad_l < 0 ? (w.ks - 1) / 2 : o.pad_l;
  const int He = x.H << o.ups, We = x.W << o.ups;
  p.Hout = o.Hout > 0 ? o.Hout : (He + 2 * p.pad_t - w.ks) / o.stride + 1;
  p.Wout = o.Wout > 0 ? o.Wout : (We + 2 * p.pad_l - w.ks) / o.stride + 1;
  const f16* wsrc = (o.split_in && x.lo8) ? derived_lo8(w, &p.lo8_sa) : o.split_in ? derived_dup(w, x.C, x2 ? x2->C : 0) : w.w;
  p.w = wsrc; p.Nrows = w.Nrows; p.K = w.ks * w.ks * Cin_eff;
  p.N = o.N_override ? o.N_override : roundup(w.N, 4);
  p.n_real = o.N_override ? 0 : w.N;
  p.bias = w.b;
  f16* wfold = nullptr;
  float* bfold = nullptr;
  if (o.gn) { p.gn_scale = o.gn->scale; p.gn_shift = o.gn->shift; p.silu_in = o.silu; }
  p.temb = o.temb; p.ld_temb = o.ld_temb;
  p.M = x.B * p.Hout * p.Wout;
  if (o.res) {
    LDIFF_CHECK(o.res->rows() == p.M && o.res->C >= p.N, LDIFF_ERR_INVALID, "conv: residual shape mismatch");
    p.res = o.res->p; p.ld_res = o.res->ld(); p.res_lo = o.res->lo();
  }
  // GroupNorm -> 1x1 conv / Linear with no activation in between (VAE attention q/k/v; transformer proj_in under PREC_FAST): fold the
  // normalisation into per-image weights and bias and run the plain LDS-DMA GEMM instead of the register-staged GN prologue
  if (o.gn && !o.silu && w.ks == 1 && !x2 && !o.out_f32 && !o.geglu && !o.want_stats && !o.split_in) {
    ConvParams q = p;
    q.gn_scale = nullptr; q.gn_shift = nullptr; q.silu_in = 0;
    q.w_bstride = (long long)w.Nrows * w.K; q.bias_bstride = w.Nrows;
    q.ldy = roundup(p.N, 8) * (o.split_out ? 2 : 1);
    if (gemm_dma_eligible(q)) {          // same predicate the dispatcher uses: no silent fall-through to a kernel without per-image weights
      wfold = tmp<f16>((size_t)x.B * w.Nrows * w.K);
      bfold = tmp<float>((size_t)x.B * w.Nrows);
      launch_fold_gn_weights(w.w, w.b, o.gn->scale, o.gn->shift, wfold, bfold, x.B, w.Nrows, w.K, s);
      p = q;
      p.w = wfold; p.bias = bfold;
    }
  }
  Act y;
  if (o.geglu) {
    LDIFF_CHECK(w.geglu && !o.out_f32 && !o.res && !o.want_stats && !o.split_out && p.N % 32 == 0, LDIFF_ERR_INVALID, "conv: GEGLU epilogue on a layer that was not built for it");
    p.geglu = 1;
    y = new_act(x.B, p.Hout, p.Wout, p.N / 2);
    p.y = y.p; p.ldy = p.N / 2;
    LDIFF_CHECK(gemm_dma_eligible(p), LDIFF_ERR_INVALID, "conv: GEGLU epilogue needs the DMA GEMM (K %% 64 == 0)");
  } else if (o.out_f32) {
    LDIFF_CHECK(!o.split_out, LDIFF_ERR_INVALID, "conv: fp32 output cannot be split");
    p.y = o.out_f32; p.ldy = o.ldy_f32; p.out_f32 = 1;
    if (o.post_done) {   // fused only where the narrow-output kernel runs (it is the one epilogue that knows the tail)
      ConvParams t = p;
      t.splitk = 1;
      const bool fused = conv3x3_eligible(t) && conv3x3n_selected(t);
      *o.post_done = fused;
      if (fused) { p.post_img = o.post_img; p.post_rgb = o.post_rgb; p.post_luma = o.post_luma; p.post_slots = o.post_slots; p.post_slot = o.post_slot; p.post_only = o.post_only ? 1 : 0; }
    }
  } else {
    const int C = o.ldy ? o.ldy : (o.split_out ? roundup(p.N, 4) : roundup(p.N, 8));
    y = new_act(x.B, p.Hout, p.Wout, C, o.split_out);
    if (C > p.N) launch_zero_bytes(y.p, y.bytes(), s);  // zero the pad columns (a kernel: the forward may be inside a captured graph)
    p.y = y.p; p.ldy = y.ld(); p.y_lo = y.lo();
    if (o.ups && conv3x3_eligible(p))   // nearest-2x upsample folded algebraically (4 parity convs with pre-summed taps)
      p.w_par = derived_par(w, wsrc, Cin_eff, o.split_in ? w.dup_par : w.par);
    // split-K launches write raw partials: their reduce kernel applies the epilogue and (round 6) emits the GroupNorm statistics too, in 32-row blocks
    p.splitk = conv3x3_eligible(p) ? conv3x3_splitk_plan(p) : gemm_dma_eligible(p) ? gemm_dma_splitk_plan(p) : igemm_splitk_plan(p);
    if (o.want_stats && C == p.N && p.N == w.N) {
      const int R = conv_stats_blocks_per_image(p);
      if (R > 0) {
        y.st = tmp<float>((size_t)x.B * R * p.N * 2);
        y.st_R = R;
        p.stats = y.st;
        p.stats_R = R;
      }
    }
  }
  if (!o.out_f32 && p.splitk > 1) p.splitk_ws = tmp<float>((size_t)p.splitk * p.M * p.N);
  else p.splitk = 0;
  if (o.sc_done) *o.sc_done = false;
  if (o.sc_x && o.sc_w && o.sc_done && !o.split_in && !o.res && !o.sc_x->split && o.sc_w->ks == 1 && o.sc_w->Nrows == w.Nrows) {
    // fold the block's 1x1 shortcut into this conv where the dataflow kernel takes the launch (LDIFF_C3D_FOLD_SC=0: never)
    static const bool fold = [] { const char* e = getenv("LDIFF_C3D_FOLD_SC"); return !e || atoi(e) != 0; }();
    ConvParams q = p;
    q.xs = o.sc_x->p; q.Cs = o.sc_x->C; q.lds = o.sc_x->ld();
    if (fold && o.sc_w->K == q.Cs && conv3x3_eligible(q) && conv3x3d_selected(q)) {   // (selected() is false under a split-K plan: q.splitk was decided above)
      p = q;
      p.w_frag = derived_frag_sc(w, *o.sc_w, p, &p.bias);
      *o.sc_done = true;
    }
  }
  if (o.sc_x && !p.xs) {   // asked for the folded form only: nothing is launched, the caller takes the two-launch form
    if (p.splitk_ws) arena.free(p.splitk_ws);
    release(y);
    return Act{};
  }
  if (p.xs) {}
  else if (!o.split_in && conv3x3_eligible(p) && conv3x3d_selected(p)) p.w_frag = p.ups ? derived_frag_par(w, p) : derived_frag(w, p);   // dataflow kernel: MFMA-fragment-packed weights
  else if (!wfold && w.ks == 1 && !(o.split_in && x.lo8) && !conv3x3_eligible(p) && gemm_df_selected(p))   // dataflow GEMM: the same, of the matrix this launch reads
    p.w_frag = derived_gfrag(w, wsrc, p.K, o.split_in ? w.gfrag_dup : w.gfrag, o.split_in ? x.C : 0);
  launch_igemm(p, s);
  if (p.splitk_ws) arena.free(p.splitk_ws);   // stream-ordered reuse: safe once the launches are enqueued
  if (wfold) { arena.free(bfold); arena.free(wfold); }
  return y;
}
Act Exec::norm_apply(const Act& x, const Act* x2, const GNss& g, bool silu, bool split_out, bool lo8) {
  const int C = x.C + (x2 ? x2->C : 0);
  Act y = new_act(x.B, x.H, x.W, C, split_out, lo8);
  launch_norm_apply(x.view(), x2 ? x2->view() : SrcView{nullptr, 0, 0, 0}, x.B, x.H * x.W, g.scale, g.shift, silu ? 1 : 0, y.p, y.ld(), y.lo(), lo8 ? 1 : 0, s);
  return y;
}
Act Exec::layernorm(const Act& x, const NormW& w) {
  Act y = new_act(x.B, x.H, x.W, x.C);
  launch_layernorm(x.view(), y.p, (int)x.rows(), w.g, w.b, 1e-5f, s);
  return y;
}
Act Exec::ln_linear(const MatW& w, const Act& x, const NormW& ln, bool geglu, int qcols, float qscale, bool* scaled) {
  if (scaled) *scaled = false;
  const int N = roundup(w.N, 4);
  const bool fused_geglu = geglu && w.geglu;
  const int Cout = fused_geglu ? N / 2 : N, ldy = roundup(Cout, 8);
  if (w.ks == 1 && w.K == x.C && w.Nrows >= N && lngemm_eligible(x.C, N, x.ld(), x.lo(), ldy, fused_geglu) && ldy == Cout && (!geglu || fused_geglu)) {
    Act y = new_act(x.B, x.H, x.W, Cout);
    const bool sc = qcols > 0 && qcols % 64 == 0 && !fused_geglu;
    launch_lngemm(x.p, x.ld(), x.lo(), (int)x.rows(), x.C, ln.g, ln.b, 1e-5f, derived_tiled(w, N), N, w.b, fused_geglu, y.p, y.ld(), s, sc ? qcols : 0, qscale);
    if (scaled) *scaled = sc;
    return y;
  }
  Act n = layernorm(x, ln);
  ConvOpts o;
  o.geglu = fused_geglu;
  Act y = conv(w, n, nullptr, o);
  release(n);
  if (geglu && !fused_geglu) {   // channel counts the fused epilogue does not take: projection, then the standalone activation
    Act gg = this->geglu(y);
    release(y);
    return gg;
  }
  return y;
}
Act Exec::geglu(const Act& x) {
  Act y = new_act(x.B, x.H, x.W, x.C / 2);
  launch_geglu(x.p, y.p, x.rows(), x.C / 2, s);
  return y;
}

// ResnetBlock2D: GN -> SiLU -> conv3x3 (+temb) -> GN -> SiLU -> conv3x3, + shortcut(x) (1x1 conv iff Cin != Cout).
//   PREC_FAST   : plain fp16 tensors, GroupNorm-apply+SiLU in the conv's load path.
//   PREC_STREAM : x / skip / output split; conv1 reads the hi halves (pitch 2C) through the GN prologue, the shortcut conv and the
//                 residual add see hi + lo.
//   PREC_FULL   : additionally the normalised operands are materialised split (norm_apply) and both convs run on split operands.
Act Exec::resnet(const ResnetW& r, const Act& x, const Act* skip, const float* temb, int ld_temb, int groups, float eps, int prec) {
  const bool st = prec >= PREC_STREAM, full = prec >= PREC_FULL;
  LDIFF_CHECK(r.has_sc || skip == nullptr, LDIFF_ERR_RUNTIME, "resnet: concat input without shortcut conv");
  GNss g1 = gn(x, skip, r.n1, groups, eps);
  ConvOpts o1;
  o1.temb = temb; o1.ld_temb = ld_temb; o1.want_stats = true;
  Act h;
  if (full) {
    Act a = norm_apply(x, skip, g1, true, true, !skip && lo8_conv_ok(r.c1, x, false, true));
    o1.split_in = true; o1.split_out = true;
    h = conv(r.c1, a, nullptr, o1);
    release(a);
  } else {
    o1.gn = &g1; o1.silu = 1;
    h = conv(r.c1, x, skip, o1);
  }
  release(g1);
  trace("  resnet.conv1", h);
  GNss g2 = gn(h, nullptr, r.n2, groups, eps);
  Act sc;
  const Act* resp = &x;
  ConvOpts o2;
  o2.want_stats = true; o2.split_out = st;
  Act out;
  if (r.has_sc && !st && !skip) {
    // plain graph (the VAE decoder's two width-changing blocks): try the shortcut as extra centre-tap slabs of conv2 on the dataflow kernel -- no
    // 1x1 launch, no round trip of its output through memory, the sum in fp32.  Where the kernel does not take the launch: the two-launch form below.
    bool folded = false;
    ConvOpts of = o2;
    of.gn = &g2; of.silu = 1; of.sc_x = &x; of.sc_w = &r.sc; of.sc_done = &folded;
    out = conv(r.c2, h, nullptr, of);   // (launches nothing and returns an empty tensor when it cannot fold)
    if (folded) {
      release(g2);
      release(h);
      return out;
    }
  }
  if (r.has_sc) {
    ConvOpts os;
    os.split_in = st; os.split_out = st;
    sc = conv(r.sc, x, skip, os);
    resp = &sc;
  }
  o2.res = resp;
  if (full) {
    Act a = norm_apply(h, nullptr, g2, true, true, lo8_conv_ok(r.c2, h, true, o2.split_out));
    o2.split_in = true;
    out = conv(r.c2, a, nullptr, o2);
    release(a);
  } else {
    o2.gn = &g2; o2.silu = 1;
    out = conv(r.c2, h, nullptr, o2);
  }
  release(g2);
  release(h);
  if (r.has_sc) release(sc);
  return out;
}

// ================================================================================================
// UNet
// ================================================================================================
static ResnetW make_resnet(WeightStore& ws, const std::string& p, int Cin, int Cout, bool temb) {
  ResnetW r;
  r.Cin = Cin; r.Cout = Cout;
  r.n1 = ws.add_norm(p + ".norm1", Cin);
  r.c1 = ws.add_conv(p + ".conv1", Cin, Cout, 3);
  r.n2 = ws.add_norm(p + ".norm2", Cout);
  r.c2 = ws.add_conv(p + ".conv2", Cout, Cout, 3);
  r.has_sc = Cin != Cout;
  if (r.has_sc) r.sc = ws.add_conv(p + ".conv_shortcut", Cin, Cout, 1);
  (void)temb;
  return r;
}

static TransformerW make_transformer(WeightStore& ws, const std::string& p, int C, int ctx) {
  TransformerW t;
  t.C = C;
  t.gn = ws.add_norm(p + ".norm", C);
  t.proj_in = ws.add_conv(p + ".proj_in", C, C, 1);
  const std::string b = p + ".transformer_blocks.0";
  t.ln1 = ws.add_norm(b + ".norm1", C);
  t.ln2 = ws.add_norm(b + ".norm2", C);
  t.ln3 = ws.add_norm(b + ".norm3", C);
  // fused self-attention QKV [3C][C], no bias
  t.qkv.N = 3 * C; t.qkv.Nrows = roundup(3 * C, 16); t.qkv.ks = 1; t.qkv.Cin = C; t.qkv.K = C;
  t.qkv.w = ws.alloc_mat(t.qkv.Nrows, C); t.qkv.b = nullptr;
  ws.add_rows(b + ".attn1.to_q.weight", "", t.qkv.w, C, 1, C, C, 0, C, nullptr, false);
  ws.add_rows(b + ".attn1.to_k.weight", "", t.qkv.w, C, 1, C, C, C, C, nullptr, false);
  ws.add_rows(b + ".attn1.to_v.weight", "", t.qkv.w, C, 1, C, C, 2 * C, C, nullptr, false);
  t.out1 = ws.add_conv(b + ".attn1.to_out.0", C, C, 1);
  t.q2 = ws.add_conv(b + ".attn2.to_q", C, C, 1, false);
  // fused cross-attention KV [2C][ctx], no bias
  t.kv2.N = 2 * C; t.kv2.Nrows = roundup(2 * C, 16); t.kv2.ks = 1; t.kv2.Cin = ctx; t.kv2.K = ctx;
  t.kv2.w = ws.alloc_mat(t.kv2.Nrows, ctx); t.kv2.b = nullptr;
  ws.add_rows(b + ".attn2.to_k.weight", "", t.kv2.w, ctx, 1, ctx, ctx, 0, C, nullptr, false);
  ws.add_rows(b + ".attn2.to_v.weight", "", t.kv2.w, ctx, 1, ctx, ctx, C, C, nullptr, false);
  t.out2 = ws.add_conv(b + ".attn2.to_out.0", C, C, 1);
  t.ff1 = ws.add_conv(b + ".ff.net.0.proj", C, 8 * C, 1, true, -1, 0, /*geglu=*/C % 64 == 0);   // fused x * gelu(gate) epilogue when the DMA GEMM applies
  t.ff2 = ws.add_conv(b + ".ff.net.2", 4 * C, C, 1);
  t.proj_out = ws.add_conv(p + ".proj_out", C, C, 1);
  return t;
}

void ldiff_unet::build() {
  ex.weights_gen = &ws.generation;
  nf.create();
  ex.nonfinite = nf.words;
  ex.trace_tag = "unet";
  const int nb = cfg.n_blocks;
  const int* boc = cfg.block_out_channels;
  const int temb_dim = boc[0] * 4, ctx = cfg.cross_attention_dim, lpb = cfg.layers_per_block;
  LDIFF_CHECK(nb >= 1 && nb <= LDIFF_MAX_BLOCKS, LDIFF_ERR_INVALID, "unet: n_blocks=%d out of range", nb);
  LDIFF_CHECK(cfg.in_channels <= 8 && cfg.out_channels <= 8 && cfg.in_channels > 0, LDIFF_ERR_INVALID, "unet: in/out channels must be <= 8");
  LDIFF_CHECK(ctx % 8 == 0 && cfg.heads > 0, LDIFF_ERR_INVALID, "unet: cross_attention_dim must be a multiple of 8");
  for (int i = 0; i < nb; ++i) {
    LDIFF_CHECK(boc[i] % cfg.norm_num_groups == 0 && boc[i] % 8 == 0, LDIFF_ERR_INVALID, "unet: channels %d not divisible by groups/8", boc[i]);
    LDIFF_CHECK(boc[i] % cfg.heads == 0 && (boc[i] / cfg.heads) % 8 == 0, LDIFF_ERR_INVALID, "unet: head dim %d/%d must be a multiple of 8", boc[i], cfg.heads);
  }
  conv_in = ws.add_conv("conv_in", cfg.in_channels, boc[0], 3, true, 8);
  t_lin1 = ws.add_conv("time_embedding.linear_1", boc[0], temb_dim, 1);
  t_lin2 = ws.add_conv("time_embedding.linear_2", temb_dim, temb_dim, 1);

  // first pass: count resnet output channels for the fused time-embedding projection
  std::vector<int> temb_couts;
  {
    int ch = boc[0];
    for (int i = 0; i < nb; ++i) { for (int j = 0; j < lpb; ++j) { temb_couts.push_back(boc[i]); ch = boc[i]; } }
    temb_couts.push_back(ch); temb_couts.push_back(ch);  // mid
    for (int i = 0; i < nb; ++i) for (int j = 0; j < lpb + 1; ++j) temb_couts.push_back(boc[nb - 1 - i]);
  }
  temb_total = 0;
  for (int c : temb_couts) temb_total += c;
  temb_proj_all.N = temb_total; temb_proj_all.Nrows = roundup(temb_total, 16); temb_proj_all.ks = 1; temb_proj_all.Cin = temb_dim;
  temb_proj_all.K = temb_dim;
  temb_proj_all.w = ws.alloc_mat(temb_proj_all.Nrows, temb_dim);
  temb_proj_all.b = ws.alloc_vec(temb_proj_all.Nrows);
  int temb_off = 0;
  auto add_res = [&](const std::string& p, int Cin, int Cout) {
    ResnetW r = make_resnet(ws, p, Cin, Cout, true);
    r.temb_off = temb_off;
    ws.add_rows(p + ".time_emb_proj.weight", p + ".time_emb_proj.bias", temb_proj_all.w, temb_dim, 1, temb_dim, temb_dim, temb_off, Cout,
                temb_proj_all.b, true);
    temb_off += Cout;
    return r;
  };

  std::vector<int> skip_ch{boc[0]};
  int ch = boc[0];
  down_res.resize(nb); down_attn.resize(nb); down_sample.resize(nb); has_down.assign(nb, false);
  for (int i = 0; i < nb; ++i) {
    for (int j = 0; j < lpb; ++j) {
      const std::string p = "down_blocks." + std::to_string(i);
      down_res[i].push_back(add_res(p + ".resnets." + std::to_string(j), ch, boc[i]));
      ch = boc[i];
      if (cfg.down_has_attn[i]) down_attn[i].push_back(make_transformer(ws, p + ".attentions." + std::to_string(j), ch, ctx));
      skip_ch.push_back(ch);
    }
    if (i != nb - 1) {
      down_sample[i] = ws.add_conv("down_blocks." + std::to_string(i) + ".downsamplers.0.conv", ch, ch, 3);
      has_down[i] = true;
      skip_ch.push_back(ch);
    }
  }
  mid_res[0] = add_res("mid_block.resnets.0", ch, ch);
  mid_attn = make_transformer(ws, "mid_block.attentions.0", ch, ctx);
  mid_res[1] = add_res("mid_block.resnets.1", ch, ch);
  up_res.resize(nb); up_attn.resize(nb); up_sample.resize(nb); has_up.assign(nb, false);
  for (int i = 0; i < nb; ++i) {
    const int oc = boc[nb - 1 - i];
    const std::string p = "up_blocks." + std::to_string(i);
    for (int j = 0; j < lpb + 1; ++j) {
      const int sc = skip_ch.back();
      skip_ch.pop_back();
      up_res[i].push_back(add_res(p + ".resnets." + std::to_string(j), ch + sc, oc));
      ch = oc;
      if (cfg.up_has_attn[i]) up_attn[i].push_back(make_transformer(ws, p + ".attentions." + std::to_string(j), ch, ctx));
    }
    if (i != nb - 1) { up_sample[i] = ws.add_conv(p + ".upsamplers.0.conv", ch, ch, 3); has_up[i] = true; }
  }
  norm_out = ws.add_norm("conv_norm_out", ch);
  conv_out = ws.add_conv("conv_out", ch, cfg.out_channels, 3);
  LDIFF_CHECK(temb_off == temb_total, LDIFF_ERR_RUNTIME, "unet: internal temb bookkeeping error");

  for (auto& v : down_attn) for (auto& t : v) all_tf.push_back(&t);
  all_tf.push_back(&mid_attn);
  for (auto& v : up_attn) for (auto& t : v) all_tf.push_back(&t);
}

void ldiff_unet::set_context(const float* ctx, int Bc, int L, hipStream_t s) {
  LDIFF_CHECK(ctx && Bc >= 1 && L >= 1, LDIFF_ERR_INVALID, "set_context: need B_ctx >= 1 and L >= 1 (got %d, %d)", Bc, L);
  LDIFF_CHECK(ws.missing() == 0, LDIFF_ERR_STATE, "unet: %d weight tensors not loaded (first: %s)", ws.missing(), ws.missing_name(0));
  HIP_CHECK(hipSetDevice(device));
  const int D = cfg.cross_attention_dim, rows = Bc * L;
  size_t need = 0;
  for (auto* t : all_tf) need += (size_t)rows * 2 * t->C;
  if (need * sizeof(f16) > ctx_cap) {
    if (ctx_buf) { HIP_CHECK(hipDeviceSynchronize()); HIP_CHECK(hipFree(ctx_buf)); ctx_buf = nullptr; }
    HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&ctx_buf), need * sizeof(f16)));
    ctx_cap = need * sizeof(f16);
  }
  ex.s = s;
  ex.arena.reserve((size_t)rows * D * 2 + (1 << 20));
  Act c16 = ex.new_act(1, 1, rows, D);
  launch_nchw_f32_to_nhwc_f16(ctx, c16.p, rows, D, 1, 1, D, s);
  size_t off = 0;
  for (auto* t : all_tf) {
    t->kv_ctx = ctx_buf + off;
    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.x = c16.p; p.C1 = D; p.B = 1; p.Hin = 1; p.Win = rows; p.Hout = 1; p.Wout = rows; p.ks = 1; p.stride = 1;
    p.w = t->kv2.w; p.N = 2 * t->C; p.Nrows = t->kv2.Nrows; p.K = D;
    p.y = t->kv_ctx; p.ldy = 2 * t->C; p.M = rows;
    launch_igemm(p, s);
    off += (size_t)rows * 2 * t->C;
  }
  ex.release(c16);
  ctx_B = Bc; ctx_L = L;
  ++ctx_gen;   // the K/V buffer may have moved: a captured forward graph holds its old address
}

Act ldiff_unet::transformer(const TransformerW& t, const Act& x) {
  const int C = t.C, heads = cfg.heads, d = C / heads, L = x.H * x.W;
  const bool st = precision >= PREC_STREAM;
  GNss g = ex.gn(x, nullptr, t.gn, cfg.norm_num_groups, 1e-6f);
  Act h;
  if (st) {   // proj_in carries the whole stream: GroupNorm-apply to a split tensor, then a split-operand GEMM
    Act xn = ex.norm_apply(x, nullptr, g, false, true);
    ConvOpts oi;
    oi.split_in = true; oi.split_out = true;
    h = ex.conv(t.proj_in, xn, nullptr, oi);
    ex.release(xn);
  } else {
    ConvOpts oi;
    oi.gn = &g; oi.silu = 0;
    h = ex.conv(t.proj_in, x, nullptr, oi);
  }
  ex.release(g);
  // self-attention
  // LDIFF_ATTN_PRESCALE=1 (default 0): level-0 self-attention (d = 40) in its prescaled form -- q leaves the fused q/k/v projection already
  // multiplied by scale * log2(e) (fp32, before its one rounding) and the attention kernel drops the per-score FMA (kernels_attn.hip PRE).
  // Measured: -7 % per attention launch, -0.5 % of a UNet pass (12.82 / 12.87 -> 12.74 / 12.80 ms same box; d = 80: no gain), and another
  // rounding pattern of q: at the bench configuration the latents' max error went 3.4e-4 -> 4.1e-4 of range and the probe masks 7 -> 8
  // differing pixels (profiles/r04_attention_prescaled.txt).  Not worth the default.
  static const bool pre_on = [] { const char* e = getenv("LDIFF_ATTN_PRESCALE"); return e && atoi(e) != 0; }();
  bool pre = false;
  const float att_scale = 1.0f / sqrtf((float)d);
  Act qkv = ex.ln_linear(t.qkv, h, t.ln1, false, (pre_on && d == 40 && attention_prescale_supported(d)) ? C : 0, att_scale * 1.4426950408889634f, &pre);
  Act a1 = ex.new_act(x.B, x.H, x.W, C);
  AttnParams ap;
  ap.q = qkv.p; ap.ldq = 3 * C; ap.k = qkv.p + C; ap.ldk = 3 * C; ap.v = qkv.p + 2 * C; ap.ldv = 3 * C;
  ap.o = a1.p; ap.ldo = C; ap.B = x.B; ap.heads = heads; ap.Lq = L; ap.Lk = L; ap.d = d;
  ap.q_bstride = (long long)L * 3 * C; ap.kv_bstride = (long long)L * 3 * C; ap.o_bstride = (long long)L * C;
  ap.scale = att_scale;
  ap.prescaled = pre ? 1 : 0;
  launch_attention(ap, ex.s);
  ap.prescaled = 0;
  ex.release(qkv);
  ConvOpts o1;
  o1.res = &h; o1.split_out = st;
  Act h2 = ex.conv(t.out1, a1, nullptr, o1);
  ex.release(a1);
  ex.release(h);
  // cross-attention (K/V precomputed per prompt)
  Act q2 = ex.ln_linear(t.q2, h2, t.ln2, false);
  Act a2 = ex.new_act(x.B, x.H, x.W, C);
  ap.q = q2.p; ap.ldq = C; ap.k = t.kv_ctx; ap.ldk = 2 * C; ap.v = t.kv_ctx + C; ap.ldv = 2 * C;
  ap.o = a2.p; ap.Lk = ctx_L; ap.q_bstride = (long long)L * C;
  ap.kv_bstride = ctx_B == 1 ? 0 : (long long)ctx_L * 2 * C;
  launch_attention(ap, ex.s);
  ex.release(q2);
  ConvOpts o2;
  o2.res = &h2; o2.split_out = st;
  Act h3 = ex.conv(t.out2, a2, nullptr, o2);
  ex.release(a2);
  ex.release(h2);
  // GEGLU feed-forward
  Act gg = ex.ln_linear(t.ff1, h3, t.ln3, true);
  ConvOpts o3;
  o3.res = &h3; o3.split_out = st;
  Act h4 = ex.conv(t.ff2, gg, nullptr, o3);
  ex.release(gg);
  ex.release(h3);
  ConvOpts oo;
  oo.res = &x; oo.want_stats = true; oo.split_in = st; oo.split_out = st;
  Act out = ex.conv(t.proj_out, h4, nullptr, oo);
  ex.release(h4);
  return out;
}

int ldiff_unet::n_skips() const {
  int n = 1;
  for (int i = 0; i < cfg.n_blocks; ++i) n += cfg.layers_per_block + (i != cfg.n_blocks - 1 ? 1 : 0);
  return n;
}
void ldiff_unet::GraphCache::drop() {
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  exec = nullptr; graph = nullptr; uses = 0;
}
ldiff_unet::~ldiff_unet() {
  nf.destroy();
  gc.drop();
  if (gc.in) (void)hipFree(gc.in);
  if (gc.out) (void)hipFree(gc.out);
  if (gc.t) (void)hipFree(gc.t);
  if (gc.cap_stream) (void)hipStreamDestroy(gc.cap_stream);
}

void ldiff_unet::forward(const float* x, int B, int h, int w, float tval, float* out, hipStream_t s) {
  static const bool env_off = getenv("LDIFF_NO_GRAPH") != nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (s) (void)hipStreamIsCapturing(s, &cs);   // the legacy default stream cannot be captured
  if (!gc.enabled || env_off || prof_enabled() || cs != hipStreamCaptureStatusNone || !x || !out || B < 1 || h < 1 || w < 1 || !extra_down.empty() ||
      extra_mid) {
    forward_impl(x, B, h, w, tval, nullptr, out, s);   // (argument errors are reported by forward_impl)
    return;
  }
  HIP_CHECK(hipSetDevice(device));
  const size_t n_in = (size_t)B * cfg.in_channels * h * w, n_out = (size_t)B * cfg.out_channels * h * w;
  const long long key[8] = {B, h, w, precision, (long long)ctx_B * 65536 + ctx_L, ws.generation, ctx_gen, (long long)ex.arena.capacity()};
  if (memcmp(key, gc.key, sizeof(key)) != 0) { gc.drop(); memcpy(gc.key, key, sizeof(key)); }
  if (gc.uses == 0) {               // first use of this configuration: eager (builds lazily derived weights, sizes the workspaces)
    forward_impl(x, B, h, w, tval, nullptr, out, s);
    gc.uses = 1;
    return;
  }
  if (gc.uses == 1) {               // second use: capture the same launch sequence on staging buffers
    const size_t need = std::max(n_in, n_out) * sizeof(float);
    if (need > gc.in_cap) {
      if (gc.in) { HIP_CHECK(hipDeviceSynchronize()); HIP_CHECK(hipFree(gc.in)); HIP_CHECK(hipFree(gc.out)); gc.in = gc.out = nullptr; }
      HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&gc.in), need));
      HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&gc.out), need));
      gc.in_cap = need;
    }
    if (!gc.t) HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&gc.t), sizeof(float)));
    // capture on a handle-owned stream (the caller's may be the legacy default stream, which cannot be captured); nothing
    // executes during capture, and the instantiated graph is launched on the caller's stream
    if (!gc.cap_stream) HIP_CHECK(hipStreamCreateWithFlags(&gc.cap_stream, hipStreamNonBlocking));
    HIP_CHECK(hipStreamBeginCapture(gc.cap_stream, hipStreamCaptureModeThreadLocal));
    hipGraph_t g = nullptr;
    try {
      forward_impl(gc.in, B, h, w, 0.f, gc.t, gc.out, gc.cap_stream);
    } catch (...) {
      (void)hipStreamEndCapture(gc.cap_stream, &g);
      if (g) (void)hipGraphDestroy(g);
      gc.enabled = false;           // this configuration cannot be captured: stay eager (same kernels, same results)
      forward_impl(x, B, h, w, tval, nullptr, out, s);
      return;
    }
    HIP_CHECK(hipStreamEndCapture(gc.cap_stream, &g));
    gc.graph = g;
    {
      size_t n_nodes = 0;
      if (hipGraphGetNodes(g, nullptr, &n_nodes) == hipSuccess) gc.nodes = (long long)n_nodes;
      static const bool debug = getenv("LDIFF_DEBUG") != nullptr;
      if (debug) fprintf(stderr, "[ldiff_unet] captured forward B=%d %dx%d precision %d: %lld graph nodes (= kernel launches per pass)\n", B, h, w, precision, gc.nodes);
    }
    HIP_CHECK(hipGraphInstantiate(&gc.exec, g, nullptr, nullptr, 0));
    gc.uses = 2;
    ++gc.captures;
  }
  HIP_CHECK(hipMemcpyAsync(gc.in, x, n_in * sizeof(float), hipMemcpyDeviceToDevice, s));
  launch_set_scalar(gc.t, tval, s);
  HIP_CHECK(hipGraphLaunch(gc.exec, s));
  HIP_CHECK(hipMemcpyAsync(out, gc.out, n_out * sizeof(float), hipMemcpyDeviceToDevice, s));
  ++gc.replays;
}

void ldiff_unet::forward_impl(const float* x, int B, int h, int w, float tval, const float* t_dev, float* out, hipStream_t s) {
  LDIFF_CHECK(x && out && B >= 1 && h >= 1 && w >= 1, LDIFF_ERR_INVALID, "unet_forward: bad arguments (B=%d h=%d w=%d)", B, h, w);
  const int nb = cfg.n_blocks;
  LDIFF_CHECK(h % (1 << (nb - 1)) == 0 && w % (1 << (nb - 1)) == 0, LDIFF_ERR_INVALID, "unet_forward: latent size %dx%d must be divisible by %d", h, w, 1 << (nb - 1));
  LDIFF_CHECK(ws.missing() == 0, LDIFF_ERR_STATE, "unet: %d weight tensors not loaded (first: %s)", ws.missing(), ws.missing_name(0));
  LDIFF_CHECK(ctx_L > 0, LDIFF_ERR_STATE, "unet: set_context has not been called");
  LDIFF_CHECK(ctx_B == 1 || ctx_B == B, LDIFF_ERR_INVALID, "unet: context batch %d does not match sample batch %d", ctx_B, B);
  HIP_CHECK(hipSetDevice(device));
  ex.s = s;
  ex.arena.reset();
  const int C0 = cfg.block_out_channels[0];
  int Cmax = 0;
  for (int i = 0; i < nb; ++i) Cmax = std::max(Cmax, cfg.block_out_channels[i]);
  ex.arena.reserve((size_t)B * h * w * C0 * 2 * (precision >= PREC_STREAM ? 160 : 96) + (size_t)B * temb_total * 16 + (64u << 20));
  ex.ensure_gn_partial(gn_partial_bytes(B, h * w, 2 * Cmax));

  // time embedding: sinusoid -> linear_1 -> SiLU -> linear_2, then SiLU once and all 22 per-resnet projections in one GEMM
  const int td = C0 * 4;
  Act e16 = ex.new_act(1, 1, B, C0);
  launch_timestep_embed(tval, t_dev, e16.p, B, C0, cfg.flip_sin_to_cos, cfg.freq_shift, s);
  float* l1 = ex.tmp<float>((size_t)B * td);
  { ConvOpts o; o.out_f32 = l1; o.ldy_f32 = td; ex.conv(t_lin1, e16, nullptr, o); }
  Act l1h = ex.new_act(1, 1, B, td);
  launch_silu_f32_to_f16(l1, l1h.p, (long long)B * td, s);
  float* l2 = ex.tmp<float>((size_t)B * td);
  { ConvOpts o; o.out_f32 = l2; o.ldy_f32 = td; ex.conv(t_lin2, l1h, nullptr, o); }
  Act l2h = ex.new_act(1, 1, B, td);
  launch_silu_f32_to_f16(l2, l2h.p, (long long)B * td, s);
  float* temb_all = ex.tmp<float>((size_t)B * temb_total);
  { ConvOpts o; o.out_f32 = temb_all; o.ldy_f32 = temb_total; ex.conv(temb_proj_all, l2h, nullptr, o); }
  ex.release(e16); ex.arena.free(l1); ex.release(l1h); ex.arena.free(l2); ex.release(l2h);

  const bool st = precision >= PREC_STREAM;
  auto rb = [&](const ResnetW& r, const Act& xin, const Act* skip) {
    return ex.resnet(r, xin, skip, temb_all + r.temb_off, temb_total, cfg.norm_num_groups, cfg.norm_eps, precision);
  };
  Act x16 = ex.new_act(B, h, w, 8);
  const bool split_first = st && conv_in.Cin_logical > 0;   // the fp32 latents enter as hi | lo inside the 8 padded channels
  launch_nchw_f32_to_nhwc_f16(x, x16.p, B, cfg.in_channels, h, w, 8, s, split_first ? cfg.in_channels : 0);
  ConvOpts oci;
  oci.want_stats = true; oci.split_in = split_first; oci.split_out = st;
  Act cur = ex.conv(conv_in, x16, nullptr, oci);
  ex.release(x16);

  std::vector<Act> skips{cur};
  bool cur_is_skip = true;
  int stage_no = 0;
  auto advance = [&](Act nxt) {
    if (!cur_is_skip) ex.release(cur);
    cur = nxt;
    cur_is_skip = false;
    char nm[32];
    snprintf(nm, sizeof(nm), "stage %d", stage_no++);
    ex.trace(nm, cur);
  };
  for (int i = 0; i < nb; ++i) {
    for (size_t j = 0; j < down_res[i].size(); ++j) {
      advance(rb(down_res[i][j], cur, nullptr));
      if (cfg.down_has_attn[i]) advance(transformer(down_attn[i][j], cur));
      skips.push_back(cur);
      cur_is_skip = true;
    }
    if (has_down[i]) {
      ConvOpts o;
      o.stride = 2; o.want_stats = true; o.split_in = st; o.split_out = st;
      advance(ex.conv(down_sample[i], cur, nullptr, o));
      skips.push_back(cur);
      cur_is_skip = true;
    }
  }
  auto add_extra = [&](Act& a, const float* r) {   // a += r (fp32 NCHW); the producer's fused GroupNorm statistics no longer describe a
    launch_add_nchw_residual(a.p, a.ld(), a.lo(), r, a.B, a.C, a.H * a.W, s);
    if (a.st) { ex.arena.free(a.st); a.st = nullptr; a.st_R = 0; }
  };
  if (!extra_down.empty()) {   // UNet2DConditionModel.forward: down_block_res_samples = [s + r for s, r in zip(res_samples, additional)]
    LDIFF_CHECK(extra_down.size() == skips.size(), LDIFF_ERR_INVALID, "unet: %zu additional down-block residuals for %zu skip tensors",
                extra_down.size(), skips.size());
    // the last skip tensor shares its buffer with the mid block's input, which diffusers leaves unmodified (the sums are new
    // tensors): give the stack its own copy of that one before adding
    Act own = ex.new_act(cur.B, cur.H, cur.W, cur.C, cur.split);
    HIP_CHECK(hipMemcpyAsync(own.p, cur.p, cur.bytes(), hipMemcpyDeviceToDevice, s));
    skips.back() = own;
    cur_is_skip = false;     // cur's buffer now belongs to the chain only
    for (size_t i = 0; i < skips.size(); ++i) add_extra(skips[i], extra_down[i]);
  }
  advance(rb(mid_res[0], cur, nullptr));
  advance(transformer(mid_attn, cur));
  advance(rb(mid_res[1], cur, nullptr));
  if (extra_mid) add_extra(cur, extra_mid);
  extra_down.clear();
  extra_mid = nullptr;
  for (int i = 0; i < nb; ++i) {
    for (size_t j = 0; j < up_res[i].size(); ++j) {
      Act sk = skips.back();
      skips.pop_back();
      advance(rb(up_res[i][j], cur, &sk));
      ex.release(sk);
      if (cfg.up_has_attn[i]) advance(transformer(up_attn[i][j], cur));
    }
    if (has_up[i]) {
      ConvOpts o;
      o.ups = 1; o.want_stats = true; o.split_in = st; o.split_out = st;
      advance(ex.conv(up_sample[i], cur, nullptr, o));
    }
  }
  GNss g = ex.gn(cur, nullptr, norm_out, cfg.norm_num_groups, cfg.norm_eps);
  const int Nst = roundup(cfg.out_channels, 4);
  float* o32 = ex.tmp<float>((size_t)B * h * w * Nst);
  if (precision >= PREC_FULL) {
    Act a = ex.norm_apply(cur, nullptr, g, true, true);
    ConvOpts o; o.split_in = true; o.out_f32 = o32; o.ldy_f32 = Nst;
    ex.conv(conv_out, a, nullptr, o);
    ex.release(a);
  } else {
    ConvOpts o; o.gn = &g; o.silu = 1; o.out_f32 = o32; o.ldy_f32 = Nst;
    ex.conv(conv_out, cur, nullptr, o);
  }
  launch_nhwc_f32_to_nchw_f32(o32, out, B, cfg.out_channels, h, w, Nst, s);
  ex.release(g);
  ex.arena.free(o32);
  ex.release(cur);
  ex.arena.free(temb_all);
  LDIFF_CHECK(skips.empty(), LDIFF_ERR_RUNTIME, "unet: skip stack not empty at exit");
}

// ================================================================================================
// VAE
// ================================================================================================
static VaeAttnW make_vae_attn(WeightStore& ws, const std::string& p, int C) {
  VaeAttnW a;
  a.C = C;
  a.gn = ws.add_norm(p + ".group_norm", C);
  a.qkv.N = 3 * C; a.qkv.Nrows = roundup(3 * C, 16); a.qkv.ks = 1; a.qkv.Cin = C; a.qkv.K = C;
  a.qkv.w = ws.alloc_mat(a.qkv.Nrows, C);
  a.qkv.b = ws.alloc_vec(a.qkv.Nrows);
  const char* nn[3] = {"to_q", "to_k", "to_v"};
  const char* old[3] = {"query", "key", "value"};
  for (int i = 0; i < 3; ++i) {
    ws.add_rows(p + "." + nn[i] + ".weight", p + "." + nn[i] + ".bias", a.qkv.w, C, 1, C, C, i * C, C, a.qkv.b, true);
    ws.alias(p + "." + old[i] + ".weight", p + "." + nn[i] + ".weight");
    ws.alias(p + "." + old[i] + ".bias", p + "." + nn[i] + ".bias");
  }
  a.out = ws.add_conv(p + ".to_out.0", C, C, 1);
  ws.alias(p + ".proj_attn.weight", p + ".to_out.0.weight");
  ws.alias(p + ".proj_attn.bias", p + ".to_out.0.bias");
  return a;
}

void ldiff_vae::wait_side(hipStream_t s) {
  if (side_used && ev_side) HIP_CHECK(hipStreamWaitEvent(s, ev_side, 0));
}

void ldiff_vae::build() {
  ex_dec.weights_gen = &ws.generation;
  ex_enc.weights_gen = &ws.generation;
  nf.create();
  ex_enc.nonfinite = nf.words; ex_dec.nonfinite = nf.words + 1;
  ex_enc.trace_tag = "vae.enc"; ex_dec.trace_tag = "vae.dec";
  const int nb = cfg.n_blocks, lpb = cfg.layers_per_block, lat = cfg.latent_channels;
  const int* boc = cfg.block_out_channels;
  LDIFF_CHECK(nb >= 1 && nb <= LDIFF_MAX_BLOCKS, LDIFF_ERR_INVALID, "vae: n_blocks=%d out of range", nb);
  LDIFF_CHECK(cfg.in_channels <= 8 && cfg.out_channels <= 4 && lat <= 4 && lat >= 1, LDIFF_ERR_INVALID, "vae: in<=8, out<=4, latent<=4 channels supported");
  for (int i = 0; i < nb; ++i)
    LDIFF_CHECK(boc[i] % cfg.norm_num_groups == 0 && boc[i] % 8 == 0, LDIFF_ERR_INVALID, "vae: channels %d not divisible by groups/8", boc[i]);
  // encoder
  e_conv_in = ws.add_conv("encoder.conv_in", cfg.in_channels, boc[0], 3, true, 8);
  int ch = boc[0];
  e_res.resize(nb); e_down.resize(nb);
  for (int i = 0; i < nb; ++i) {
    for (int j = 0; j < lpb; ++j) {
      e_res[i].push_back(make_resnet(ws, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), ch, boc[i], false));
      ch = boc[i];
    }
    if (i != nb - 1) e_down[i] = ws.add_conv("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv", ch, ch, 3);
  }
  e_mid[0] = make_resnet(ws, "encoder.mid_block.resnets.0", ch, ch, false);
  e_attn = make_vae_attn(ws, "encoder.mid_block.attentions.0", ch);
  e_mid[1] = make_resnet(ws, "encoder.mid_block.resnets.1", ch, ch, false);
  e_norm_out = ws.add_norm("encoder.conv_norm_out", ch);
  e_conv_out = ws.add_conv("encoder.conv_out", ch, 2 * lat, 3);
  quant = ws.add_conv("quant_conv", 2 * lat, 2 * lat, 1, true, 8);
  // decoder
  post_quant = ws.add_conv("post_quant_conv", lat, lat, 1, true, 8);
  ch = boc[nb - 1];
  d_conv_in = ws.add_conv("decoder.conv_in", lat, ch, 3, true, 8);
  d_mid[0] = make_resnet(ws, "decoder.mid_block.resnets.0", ch, ch, false);
  d_attn = make_vae_attn(ws, "decoder.mid_block.attentions.0", ch);
  d_mid[1] = make_resnet(ws, "decoder.mid_block.resnets.1", ch, ch, false);
  d_res.resize(nb); d_up.resize(nb);
  for (int i = 0; i < nb; ++i) {
    const int oc = boc[nb - 1 - i];
    for (int j = 0; j < lpb + 1; ++j) {
      d_res[i].push_back(make_resnet(ws, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), ch, oc, false));
      ch = oc;
    }
    if (i != nb - 1) d_up[i] = ws.add_conv("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv", ch, ch, 3);
  }
  d_norm_out = ws.add_norm("decoder.conv_norm_out", ch);
  d_conv_out = ws.add_conv("decoder.conv_out", ch, cfg.out_channels, 3);
}

Act ldiff_vae::mid_attention(const VaeAttnW& a, const Act& x) {
  const int C = a.C, L = x.H * x.W;
  const bool st = prec() >= PREC_STREAM, full = prec() >= PREC_FULL;
  GNss g = ex().gn(x, nullptr, a.gn, cfg.norm_num_groups, 1e-6f);
  Act qkv;
  if (full) {   // normalised operand materialised split, q/k/v on the split operand
    Act xn = ex().norm_apply(x, nullptr, g, false, true);
    ConvOpts oq;
    oq.split_in = true;
    qkv = ex().conv(a.qkv, xn, nullptr, oq);
    ex().release(xn);
  } else {      // GroupNorm folded into per-image q/k/v weights (reads the hi half of a split x)
    ConvOpts oq;
    oq.gn = &g; oq.silu = 0;
    qkv = ex().conv(a.qkv, x, nullptr, oq);
  }
  ex().release(g);
  Act o = ex().new_act(x.B, x.H, x.W, C);
  AttnParams ap;
  ap.q = qkv.p; ap.ldq = 3 * C; ap.k = qkv.p + C; ap.ldk = 3 * C; ap.v = qkv.p + 2 * C; ap.ldv = 3 * C;
  ap.o = o.p; ap.ldo = C; ap.B = x.B; ap.heads = 1; ap.Lq = L; ap.Lk = L; ap.d = C;
  ap.q_bstride = (long long)L * 3 * C; ap.kv_bstride = ap.q_bstride; ap.o_bstride = (long long)L * C;
  ap.scale = 1.0f / sqrtf((float)C);
  launch_attention(ap, ex().s);
  ex().release(qkv);
  ConvOpts oo;
  oo.res = &x; oo.want_stats = true; oo.split_out = st;
  Act out = ex().conv(a.out, o, nullptr, oo);
  ex().release(o);
  return out;
}

void ldiff_vae::encode(const float* x, int B, int H, int W, float* moments, hipStream_t s) {
  const int nb = cfg.n_blocks, f = 1 << (nb - 1);
  LDIFF_CHECK(x && moments && B >= 1, LDIFF_ERR_INVALID, "vae_encode: bad arguments");
  LDIFF_CHECK(H >= f && W >= f && H % f == 0 && W % f == 0, LDIFF_ERR_INVALID, "vae_encode: image size %dx%d must be a positive multiple of %d", H, W, f);
  LDIFF_CHECK(ws.missing() == 0, LDIFF_ERR_STATE, "vae: %d weight tensors not loaded (first: %s)", ws.missing(), ws.missing_name(0));
  HIP_CHECK(hipSetDevice(device));
  struct UseEnc { ldiff_vae* v; UseEnc(ldiff_vae* v_) : v(v_) { v->cur = &v->ex_enc; } ~UseEnc() { v->cur = &v->ex_dec; } } use_enc(this);
  const int pr = prec_enc;
  const bool st = pr >= PREC_STREAM, full = pr >= PREC_FULL;
  ex().arena.reset();
  ex().s = s;
  const int* boc = cfg.block_out_channels;
  int Cmax = 0;
  for (int i = 0; i < nb; ++i) Cmax = std::max(Cmax, boc[i]);
  const size_t mult = full ? 3 : (st ? 2 : 1);
  ex().arena.reserve(mult * ((size_t)B * H * W * boc[0] * 2 * 10 + (size_t)B * (H / f) * (W / f) * Cmax * 2 * 24) + (64u << 20));
  ex().ensure_gn_partial(std::max(gn_partial_bytes(B, H * W, boc[0]), gn_partial_bytes(B, (H / f) * (W / f), Cmax)));
  for (int i = 0; i < nb; ++i) ex().ensure_gn_partial(gn_partial_bytes(B, (H >> i) * (W >> i), boc[i]));
  auto rb = [&](const ResnetW& r, const Act& xin) { return ex().resnet(r, xin, nullptr, nullptr, 0, cfg.norm_num_groups, 1e-6f, pr); };

  Act x16 = ex().new_act(B, H, W, 8);
  const bool split_first = st && e_conv_in.Cin_logical > 0;   // the fp32 image enters as hi | lo inside the 8 padded channels
  launch_nchw_f32_to_nhwc_f16(x, x16.p, B, cfg.in_channels, H, W, 8, s, split_first ? cfg.in_channels : 0);
  ConvOpts oci;
  oci.want_stats = true; oci.split_in = split_first; oci.split_out = st;
  Act cur = ex().conv(e_conv_in, x16, nullptr, oci);
  ex().release(x16);
  ex().trace("conv_in", cur);
  auto advance = [&](Act nxt, const char* stage) { ex().release(cur); cur = nxt; ex().trace(stage, cur); };
  for (int i = 0; i < nb; ++i) {
    char nm[64];
    int j = 0;
    for (auto& r : e_res[i]) { snprintf(nm, sizeof(nm), "down_blocks.%d.resnets.%d", i, j++); advance(rb(r, cur), nm); }
    if (i != nb - 1) {
      ConvOpts o;  // Downsample2D(padding=0): F.pad(x,(0,1,0,1)) then stride-2 conv without padding
      o.stride = 2; o.pad_t = 0; o.pad_l = 0; o.Hout = cur.H / 2; o.Wout = cur.W / 2; o.want_stats = true; o.split_in = st; o.split_out = st;
      snprintf(nm, sizeof(nm), "down_blocks.%d.downsamplers.0", i);
      advance(ex().conv(e_down[i], cur, nullptr, o), nm);
    }
  }
  advance(rb(e_mid[0], cur), "mid.resnets.0");
  advance(mid_attention(e_attn, cur), "mid.attentions.0");
  advance(rb(e_mid[1], cur), "mid.resnets.1");
  GNss g = ex().gn(cur, nullptr, e_norm_out, cfg.norm_num_groups, 1e-6f);
  Act m;   // [B,h,w,8] (2*latent channels)
  if (full) {
    Act a = ex().norm_apply(cur, nullptr, g, true, true);
    ConvOpts oc;
    oc.split_in = true; oc.split_out = true; oc.ldy = roundup(2 * cfg.latent_channels, 8);
    m = ex().conv(e_conv_out, a, nullptr, oc);
    ex().release(a);
  } else {
    ConvOpts oc;
    oc.gn = &g; oc.silu = 1;
    m = ex().conv(e_conv_out, cur, nullptr, oc);
  }
  ex().release(g);
  ex().release(cur);
  const int Nst = roundup(2 * cfg.latent_channels, 4);
  float* q32 = ex().tmp<float>((size_t)m.rows() * Nst);
  { ConvOpts o; o.out_f32 = q32; o.ldy_f32 = Nst; o.split_in = m.split; ex().conv(quant, m, nullptr, o); }
  launch_nhwc_f32_to_nchw_f32(q32, moments, B, 2 * cfg.latent_channels, m.H, m.W, Nst, s);
  ex().arena.free(q32);
  ex().release(m);
}

void ldiff_vae::decode(const float* z, int B, int h, int w, float z_scale, float* sample_nchw, float* image_nhwc, uint8_t* rgb, uint8_t* luma,
                       int n_slots, int slot, hipStream_t s) {
  const int nb = cfg.n_blocks, f = 1 << (nb - 1);
  LDIFF_CHECK(z && B >= 1 && h >= 1 && w >= 1, LDIFF_ERR_INVALID, "vae_decode: bad arguments");
  LDIFF_CHECK(ws.missing() == 0, LDIFF_ERR_STATE, "vae: %d weight tensors not loaded (first: %s)", ws.missing(), ws.missing_name(0));
  LDIFF_CHECK(!luma || (slot >= 0 && slot < n_slots), LDIFF_ERR_INVALID, "vae_decode: luma slot %d out of range [0,%d)", slot, n_slots);
  HIP_CHECK(hipSetDevice(device));
  const int pr = prec_dec;
  const bool st = pr >= PREC_STREAM, full = pr >= PREC_FULL;
  ex().arena.reset();   // decodes of one VAE run on one stream at a time: the workspace is reused in stream order
  ex().s = s;
  const int* boc = cfg.block_out_channels;
  const int H = h * f, W = w * f;
  int Cmax = 0;
  for (int i = 0; i < nb; ++i) Cmax = std::max(Cmax, boc[i]);
  const size_t mult = full ? 3 : (st ? 2 : 1);
  ex().arena.reserve(mult * ((size_t)B * H * W * boc[0] * 2 * 10 + (size_t)B * h * w * Cmax * 2 * 24) + (64u << 20));
  for (int i = 0; i < nb; ++i) ex().ensure_gn_partial(gn_partial_bytes(B, (H >> i) * (W >> i), boc[std::min(i + 1, nb - 1)]));
  ex().ensure_gn_partial(gn_partial_bytes(B, h * w, Cmax));
  auto rb = [&](const ResnetW& r, const Act& xin) { return ex().resnet(r, xin, nullptr, nullptr, 0, cfg.norm_num_groups, 1e-6f, pr); };

  const long long nz = (long long)B * cfg.latent_channels * h * w;
  float* zs = ex().tmp<float>((size_t)nz);
  launch_scale_f32(z, zs, z_scale, nz, s);
  Act z16 = ex().new_act(B, h, w, 8);
  const bool split_first = st && post_quant.Cin_logical > 0 && d_conv_in.Cin_logical > 0;   // z and post_quant(z) as hi | lo in 8 channels
  launch_nchw_f32_to_nhwc_f16(zs, z16.p, B, cfg.latent_channels, h, w, 8, s, split_first ? cfg.latent_channels : 0);
  ex().arena.free(zs);
  Act pq;
  if (split_first) {
    ConvOpts opq;
    opq.split_in = true; opq.split_out = true; opq.N_override = roundup(cfg.latent_channels, 4);
    pq = ex().conv(post_quant, z16, nullptr, opq);   // [B,h,w, hi(4) | lo(4)]
  } else {
    ConvOpts opq;
    opq.N_override = 8; opq.ldy = 8;   // rows >= latent_channels are zero => pad channels come out zero
    pq = ex().conv(post_quant, z16, nullptr, opq);
  }
  ex().release(z16);
  ConvOpts odi;
  odi.want_stats = true; odi.split_in = split_first; odi.split_out = st;
  Act cur = ex().conv(d_conv_in, pq, nullptr, odi);
  ex().release(pq);
  ex().trace("conv_in", cur);
  auto advance = [&](Act nxt, const char* stage) { ex().release(cur); cur = nxt; ex().trace(stage, cur); };
  advance(rb(d_mid[0], cur), "mid.resnets.0");
  advance(mid_attention(d_attn, cur), "mid.attentions.0");
  advance(rb(d_mid[1], cur), "mid.resnets.1");
  for (int i = 0; i < nb; ++i) {
    char nm[64];
    int j = 0;
    for (auto& r : d_res[i]) { snprintf(nm, sizeof(nm), "up_blocks.%d.resnets.%d", i, j++); advance(rb(r, cur), nm); }
    if (i != nb - 1) {
      ConvOpts o;
      o.ups = 1; o.want_stats = true; o.split_in = st; o.split_out = st;
      snprintf(nm, sizeof(nm), "up_blocks.%d.upsamplers.0", i);
      advance(ex().conv(d_up[i], cur, nullptr, o), nm);
    }
  }
  GNss g = ex().gn(cur, nullptr, d_norm_out, cfg.norm_num_groups, 1e-6f);
  const int Nst = 4;
  float* o32 = ex().tmp<float>((size_t)B * H * W * Nst);
  const bool want_post = image_nhwc || rgb || luma;
  LDIFF_CHECK(!want_post || cfg.out_channels == 3, LDIFF_ERR_INVALID, "vae_decode: image outputs need 3 output channels");
  bool post_done = false;
  auto post = [&](ConvOpts& o) {   // the decode_latents tail inside conv_out's epilogue; the fp32 tensor only if `sample` is wanted too
    if (!want_post) return;
    o.post_img = image_nhwc; o.post_rgb = rgb; o.post_luma = luma; o.post_slots = n_slots; o.post_slot = slot; o.post_only = sample_nchw == nullptr;
    o.post_done = &post_done;
  };
  if (full) {
    Act a = ex().norm_apply(cur, nullptr, g, true, true);
    ConvOpts o; o.split_in = true; o.out_f32 = o32; o.ldy_f32 = Nst;
    post(o);
    ex().conv(d_conv_out, a, nullptr, o);
    ex().release(a);
  } else {
    ConvOpts o; o.gn = &g; o.silu = 1; o.out_f32 = o32; o.ldy_f32 = Nst;
    post(o);
    ex().conv(d_conv_out, cur, nullptr, o);
  }
  ex().release(g);
  ex().release(cur);
  if (sample_nchw) launch_nhwc_f32_to_nchw_f32(o32, sample_nchw, B, cfg.out_channels, H, W, Nst, s);
  if (want_post && !post_done) launch_decode_post(o32, Nst, B, H, W, image_nhwc, rgb, luma, n_slots, slot, s);
  ex().arena.free(o32);
}

// ================================================================================================
// PNDM host logic (SD-v1.5 scheduler_config.json)
// ================================================================================================
void pndm_alphas_cumprod(float* out) {
  // torch.linspace(sqrt(b0), sqrt(b1), 1000, dtype=float32) ** 2 -> cumprod(1 - beta), all float32.
  // (python evaluates beta ** 0.5 in double and torch narrows it to float32.)  ATen's vectorised fill rounds
  // start + step*i in two steps, so this table agrees with torch's to ~2 ulp, not bit for bit; callers that need
  // torch's exact table pass it through ldiff_pipeline_set_alphas_cumprod (the python shim does).
  const float start = (float)sqrt(0.00085), end = (float)sqrt(0.012);
  // ATen linspace: step = (end-start)/(steps-1); first half start + i*step, second half end - (steps-1-i)*step
  const int steps = 1000, halfway = steps / 2;
  const float step = (end - start) / (float)(steps - 1);
  float acc = 1.0f;
  for (int i = 0; i < steps; ++i) {
    float v = i < halfway ? start + step * (float)i : end - step * (float)(steps - i - 1);
    float beta = v * v;
    acc = acc * (1.0f - beta);
    out[i] = acc;
  }
}
int plms_timesteps(int n_passes, int64_t* out, int cap) {
  // PNDMScheduler.set_timesteps(n), skip_prk_steps, leading spacing, steps_offset 1; n = N-1 (N>=3) or 1 (N==1)
  LDIFF_CHECK(n_passes == 1 || n_passes >= 3, LDIFF_ERR_INVALID,
              "n_passes=%d: the reference's set_timesteps(N-1) yields N passes only for N>=3 (N=1 via set_timesteps(1))", n_passes);
  const int n = n_passes == 1 ? 1 : n_passes - 1;
  LDIFF_CHECK(n <= 1000, LDIFF_ERR_INVALID, "n_passes=%d exceeds the 1000 training timesteps", n_passes);
  const int ratio = 1000 / n;
  std::vector<int64_t> ts(n);
  for (int i = 0; i < n; ++i) ts[i] = (int64_t)i * ratio + 1;
  std::vector<int64_t> pl;
  for (int i = 0; i < n - 1; ++i) pl.push_back(ts[i]);
  if (n >= 2) pl.push_back(ts[n - 2]);
  pl.push_back(ts[n - 1]);
  std::reverse(pl.begin(), pl.end());
  LDIFF_CHECK((int)pl.size() <= cap, LDIFF_ERR_INVALID, "plms_timesteps: output capacity %d < %zu", cap, pl.size());
  for (size_t i = 0; i < pl.size(); ++i) out[i] = pl[i];
  return (int)pl.size();
}
