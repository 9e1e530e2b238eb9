// Per-launch HIP-event profiling used by bench.py for the roofline line (see include/ldiff.h ldiff_prof_*).
#include <map>
#include <string.h>

#include "common.h"

namespace {
struct Rec { std::string name; hipEvent_t e0, e1; double flops, bytes; };
bool g_on = false;
std::string g_filter;   // empty = every profiled launch
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}
}  // namespace

bool prof_enabled() { return g_on && g_filter.empty(); }   // a filtered profile leaves the UNet forward on its graph (those launches are not bracketed)
bool prof_on(const char* name) { return g_on && (g_filter.empty() || g_filter == name); }
void prof_begin(const char* name, double flops, double bytes, hipStream_t s) {
  Rec r;
  r.name = name; r.flops = flops; r.bytes = bytes;
  r.e0 = get_event(); r.e1 = get_event();
  HIP_CHECK(hipEventRecord(r.e0, s));
  g_recs.push_back(r);
}
void prof_end(hipStream_t s) { HIP_CHECK(hipEventRecord(g_recs.back().e1, s)); }

extern "C" int ldiff_prof_enable(int on) {
  g_on = on != 0;
  return LDIFF_OK;
}
extern "C" int ldiff_prof_set_filter(const char* kernel_name_or_null) {
  g_filter = kernel_name_or_null ? kernel_name_or_null : "";
  return LDIFF_OK;
}
extern "C" int ldiff_prof_collect(ldiff_prof_row* rows, int cap) {
  try {
    std::map<std::string, ldiff_prof_row> agg;
    const bool dump = getenv("LDIFF_PROF_DUMP") != nullptr;   // diagnostic: one stderr line per launch, in launch order
    for (auto& r : g_recs) {
      HIP_CHECK(hipEventSynchronize(r.e1));
      float ms = 0.f;
      HIP_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
      if (dump) fprintf(stderr, "[ldiff_prof] %-28s %9.1f us %9.2f GFLOP %8.1f TFLOP/s %9.2f MB %7.1f GB/s\n", r.name.c_str(), ms * 1e3, r.flops * 1e-9,
                        ms > 0 ? r.flops / (ms * 1e-3) * 1e-12 : 0.0, r.bytes * 1e-6, ms > 0 ? r.bytes / (ms * 1e-3) * 1e-9 : 0.0);
      auto it = agg.find(r.name);
      if (it == agg.end()) {
        ldiff_prof_row z;
        memset(&z, 0, sizeof(z));
        strncpy(z.name, r.name.c_str(), sizeof(z.name) - 1);
        it = agg.insert({r.name, z}).first;
      }
      it->second.launches += 1; it->second.ms += ms; it->second.flops += r.flops; it->second.bytes += r.bytes;
      g_pool.push_back(r.e0); g_pool.push_back(r.e1);
    }
    g_recs.clear();
    int n = 0;
    for (auto& kv : agg) { if (n < cap) rows[n] = kv.second; ++n; }
    return n;
  } catch (const LdiffError& e) { return e.code; }
}
