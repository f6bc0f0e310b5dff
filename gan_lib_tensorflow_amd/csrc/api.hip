// Error reporting, version, and the opt-in HIP-event profiler of libgank.
#include "gank_common.h"
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

static thread_local char g_err[512] = "";

int gank_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return 1;
}

extern "C" const char* gank_last_error(void) { return g_err; }
extern "C" int gank_version(void) { return GANK_VERSION; }
extern "C" int gank_act_dtype(void) { return GANK_ACT_DTYPE; }

// ---- profiler: one (start, stop) event pair per launch of a kernel family, recorded on the launch
// stream.  Off by default; never active during graph capture (bench.py enables it for one eager pass).
namespace {
constexpr int kFamilies = 4;
struct Rec { hipEvent_t a, b; double flops, bytes; const char* tag; };
bool g_on = false;
std::vector<Rec> g_recs[kFamilies];
hipEvent_t g_open[kFamilies];
}  // namespace

void gank_prof_begin(int family, double flops, hipStream_t s, double bytes) {
  if (!g_on || family < 0 || family >= kFamilies) return;
  Rec r;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  r.flops = flops;
  r.bytes = bytes;
  r.tag = "";
  hipEventRecord(r.a, s);
  g_recs[family].push_back(r);
}

void gank_prof_tag(int family, const char* kernel_name) {
  if (!g_on || family < 0 || family >= kFamilies || g_recs[family].empty()) return;
  g_recs[family].back().tag = kernel_name;
}

void gank_prof_end(int family, hipStream_t s) {
  if (!g_on || family < 0 || family >= kFamilies || g_recs[family].empty()) return;
  hipEventRecord(g_recs[family].back().b, s);
}

extern "C" int gank_prof_enable(int on) { g_on = on != 0; return 0; }

// What an event pair costs by itself: n launches of an empty kernel, each between its own pair, average ms per pair.
// bench.py reports it next to the family times (an event pair reads a few microseconds longer than the kernel's own
// begin-to-end time that rocprofv3 reports).
__global__ void prof_empty_kernel() {}
extern "C" double gank_prof_calibrate(int n, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) return 0.0;
  std::vector<Rec> recs(n);
  for (auto& r : recs) {
    r.flops = r.bytes = 0; r.tag = "";
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1.0;
    hipEventRecord(r.a, s);
    hipLaunchKernelGGL(prof_empty_kernel, dim3(1), dim3(64), 0, s);
    hipEventRecord(r.b, s);
  }
  double ms = 0;
  for (auto& r : recs) {
    float t = 0.f;
    hipEventSynchronize(r.b);
    hipEventElapsedTime(&t, r.a, r.b);
    ms += t;
    hipEventDestroy(r.a); hipEventDestroy(r.b);
  }
  return ms / n;
}

extern "C" int gank_prof_reset(void) {
  for (int f = 0; f < kFamilies; f++) {
    for (auto& r : g_recs[f]) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    g_recs[f].clear();
  }
  return 0;
}

// per-kernel breakdown of a family: entry `index` (sorted by total time, longest first) -> its symbol name as the
// launcher recorded it, launches, total ms, FLOPs and algorithmic bytes.  Returns 0 when index is past the end.
extern "C" int gank_prof_kernel_stats(int family, int index, char* name, int name_cap, int* launches, double* total_ms,
                                      double* total_flops, double* total_bytes) {
  if (family < 0 || family >= kFamilies || index < 0) return 0;
  struct Agg { int n = 0; double ms = 0, fl = 0, by = 0; };
  std::map<std::string, Agg> m;
  for (auto& r : g_recs[family]) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    Agg& a = m[r.tag ? r.tag : ""];
    a.n++; a.ms += t; a.fl += r.flops; a.by += r.bytes;
  }
  std::vector<std::pair<std::string, Agg>> v(m.begin(), m.end());
  std::sort(v.begin(), v.end(), [](const auto& x, const auto& y) { return x.second.ms > y.second.ms; });
  if (index >= (int)v.size()) return 0;
  if (name && name_cap > 0) { strncpy(name, v[index].first.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (launches) *launches = v[index].second.n;
  if (total_ms) *total_ms = v[index].second.ms;
  if (total_flops) *total_flops = v[index].second.fl;
  if (total_bytes) *total_bytes = v[index].second.by;
  return 1;
}

extern "C" double gank_prof_bytes(int family) {
  double b = 0;
  if (family >= 0 && family < kFamilies)
    for (auto& r : g_recs[family]) b += r.bytes;
  return b;
}

extern "C" int gank_prof_collect(int family, double* total_ms, double* total_flops) {
  if (family < 0 || family >= kFamilies) return 0;
  double ms = 0, fl = 0;
  int n = 0;
  for (auto& r : g_recs[family]) {
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    ms += t; fl += r.flops; n++;
  }
  if (total_ms) *total_ms = ms;
  if (total_flops) *total_flops = fl;
  return n;
}
