// Thread-local error string + version entry points of libpgdvs_hip.so.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <stdio.h>

#include <stdlib.h>

#include "common.h"

namespace pgdvs {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char *last_error_text() { return g_err; }

// ---- per-kernel timing (pgdvs_prof_*) --------------------------------------
struct ProfRec {
  const char *name;
  hipEvent_t a, b;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static thread_local ProfRec g_prof_cur;

bool prof_enabled() { return g_prof_on; }

static hipEvent_t prof_event() {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

void prof_begin(const char *name, hipStream_t s) {
  g_prof_cur.name = name;
  g_prof_cur.a = prof_event();
  g_prof_cur.b = prof_event();
  (void)hipEventRecord(g_prof_cur.a, s);
}

void prof_end(hipStream_t s) {
  (void)hipEventRecord(g_prof_cur.b, s);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_recs.push_back(g_prof_cur);
}
}  // namespace pgdvs

// ---- options ------------------------------------------------------------------
namespace pgdvs {
static Options options_from_env() {
  auto flag = [](const char *name) {
    const char *v = getenv(name);
    return v != nullptr && v[0] == '1' ? 1 : 0;
  };
  Options o;
  o.agg_ordered = flag("PGDVS_AGG_ORDERED");
  o.agg_stage = getenv("PGDVS_AGG_STAGE") != nullptr && getenv("PGDVS_AGG_STAGE")[0] == '0' ? 0 : 1;
  o.gnt_fp32 = flag("PGDVS_GNT_FP32");
  o.knn_no_tpq = flag("PGDVS_KNN_NO_TPQ");
  o.knn_stats = flag("PGDVS_KNN_STATS");
  const char *d = getenv("PGDVS_RASTER_BOUND_DENSITY");
  o.raster_bound_density = d != nullptr ? (float)atof(d) : 2.2f;
  o.side_thread = getenv("PGDVS_SIDE_THREAD") != nullptr && getenv("PGDVS_SIDE_THREAD")[0] == '0' ? 0 : 1;
  return o;
}
static Options g_options = options_from_env();  // (dynamic initialisation at load time: the only read of the environment)
const Options &options() { return g_options; }
int option_int(const int &field) { return __atomic_load_n(&field, __ATOMIC_RELAXED); }
float option_float(const float &field) {
  float v;
  __atomic_load(&field, &v, __ATOMIC_RELAXED);
  return v;
}
struct OptionName {
  const char *name;
  int *i;
  float *f;
};
static const OptionName kOptionNames[] = {
    {"agg_ordered", &g_options.agg_ordered, nullptr},   {"agg_stage", &g_options.agg_stage, nullptr},
    {"gnt_fp32", &g_options.gnt_fp32, nullptr},
    {"knn_no_tpq", &g_options.knn_no_tpq, nullptr},     {"knn_stats", &g_options.knn_stats, nullptr},
    {"raster_bound_density", nullptr, &g_options.raster_bound_density},
    {"side_thread", &g_options.side_thread, nullptr},
};
}  // namespace pgdvs

extern "C" __attribute__((visibility("default"))) int pgdvs_option_set(const char *name, double value) {
  for (const auto &o : pgdvs::kOptionNames) {
    if (name != nullptr && strcmp(name, o.name) == 0) {
      if (o.i != nullptr) {
        __atomic_store_n(o.i, value != 0.0 ? 1 : 0, __ATOMIC_RELAXED);
      } else {
        float v = (float)value;
        __atomic_store(o.f, &v, __ATOMIC_RELAXED);
      }
      return PGDVS_OK;
    }
  }
  pgdvs::set_error("pgdvs_option_set: unknown option '%s'", name ? name : "(null)");
  return PGDVS_ERR_INVALID;
}

extern "C" __attribute__((visibility("default"))) double pgdvs_option_get(const char *name) {
  for (const auto &o : pgdvs::kOptionNames) {
    if (name != nullptr && strcmp(name, o.name) == 0)
      return o.i != nullptr ? (double)pgdvs::option_int(*o.i) : (double)pgdvs::option_float(*o.f);
  }
  return __builtin_nan("");
}

namespace pgdvs {
__global__ void prof_null_kernel() {}

// What an (event, launch, event) bracket measures beyond the kernel itself: the dispatch of an
// empty kernel between two event markers on an idle stream (minimum of 32 tries).  Subtracted
// from every record so that the durations of short kernels agree with a kernel trace.
static double prof_bracket_overhead_ms() {
  static double cached = -1.0;
  if (cached >= 0.0) return cached;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 0.0;
  double best = 1e30;
  (void)hipDeviceSynchronize();
  for (int i = 0; i < 32; ++i) {
    (void)hipEventRecord(a, nullptr);
    prof_null_kernel<<<1, 64, 0, nullptr>>>();
    (void)hipEventRecord(b, nullptr);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess && ms > 0.f && ms < best) best = ms;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  cached = best < 1e29 ? best : 0.0;
  return cached;
}
}  // namespace pgdvs

extern "C" __attribute__((visibility("default"))) double pgdvs_prof_overhead_ms(void) {
  return pgdvs::prof_bracket_overhead_ms();
}

extern "C" __attribute__((visibility("default"))) void pgdvs_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(pgdvs::g_prof_mu);
  pgdvs::g_prof_on = on != 0;
}

// Synchronises the recorded events, writes "name calls total_ms\n" lines into buf
// (truncated to buf_len) and clears the records.  Returns the number of distinct names.
extern "C" __attribute__((visibility("default"))) int pgdvs_prof_report(char *buf, int buf_len) {
  using namespace pgdvs;
  std::vector<ProfRec> recs;
  {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    recs.swap(g_prof_recs);
  }
  std::map<std::string, std::pair<int, double>> agg;
  const double overhead = recs.empty() ? 0.0 : prof_bracket_overhead_ms();
  for (auto &r : recs) {
    (void)hipEventSynchronize(r.b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, r.a, r.b);
    auto &e = agg[r.name];
    e.first += 1;
    e.second += ms > overhead ? ms - overhead : 0.0;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_pool.push_back(r.a);
    g_prof_pool.push_back(r.b);
  }
  std::string out;
  for (auto &kv : agg) {
    char line[256];
    snprintf(line, sizeof(line), "%s %d %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if (buf && buf_len > 0) {
    strncpy(buf, out.c_str(), (size_t)buf_len - 1);
    buf[buf_len - 1] = 0;
  }
  return (int)agg.size();
}

extern "C" __attribute__((visibility("default"))) const char *pgdvs_last_error(void) {
  return pgdvs::g_err;
}
extern "C" __attribute__((visibility("default"))) int pgdvs_abi_version(void) { return 1; }
extern "C" __attribute__((visibility("default"))) const char *pgdvs_build_arch(void) {
  return "gfx950";
}
