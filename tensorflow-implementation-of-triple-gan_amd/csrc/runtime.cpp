// Runtime services of libtg_hip.so: version, error strings, hipGraph capture, event profiler.
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>
#include "tg_common.h"

namespace tg {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return TG_ERR_HIP;
}

static const char* kClassNames[PC_COUNT] = {"igemm_f32", "wgrad_f32", "prep", "norm", "elementwise", "loss", "optim"};

struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; char desc[160]; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;
static std::mutex g_mu;

static hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

ProfScope::ProfScope(int cls, double flops, double bytes, hipStream_t s, const char* desc) : idx(-1), stream(s) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRec r{get_event(), get_event(), cls, flops, bytes, {0}};
  if (desc) snprintf(r.desc, sizeof(r.desc), "%s", desc);
  (void)hipEventRecord(r.a, s);
  idx = (int)g_recs.size();
  g_recs.push_back(r);
}

ProfScope::~ProfScope() {
  if (idx < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipEventRecord(g_recs[idx].b, stream);
}

}  // namespace tg

using namespace tg;

extern "C" {

int tg_version(void) { return 100; }
const char* tg_last_error_string(void) { return g_err; }

int tg_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
  return n;
}

int tg_graph_begin_capture(void* stream) {
  hipError_t e = hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) return hip_fail(e, "hipStreamBeginCapture");
  return TG_OK;
}

int tg_graph_end_capture(void* stream, void** graph_exec_out) {
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(as_stream(stream), &g);
  if (e != hipSuccess) return hip_fail(e, "hipStreamEndCapture");
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return hip_fail(e, "hipGraphInstantiate");
  *graph_exec_out = ex;
  return TG_OK;
}

int tg_graph_launch(void* graph_exec, void* stream) {
  if (!graph_exec) { set_error("tg_graph_launch: null graph"); return TG_ERR_STATE; }
  hipError_t e = hipGraphLaunch(reinterpret_cast<hipGraphExec_t>(graph_exec), as_stream(stream));
  if (e != hipSuccess) return hip_fail(e, "hipGraphLaunch");
  return TG_OK;
}

int tg_graph_destroy(void* graph_exec) {
  if (!graph_exec) return TG_OK;
  hipError_t e = hipGraphExecDestroy(reinterpret_cast<hipGraphExec_t>(graph_exec));
  if (e != hipSuccess) return hip_fail(e, "hipGraphExecDestroy");
  return TG_OK;
}

int tg_prof_enable(int on) { g_prof_on = on != 0; return TG_OK; }

int tg_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) { g_pool.push_back(r.a); g_pool.push_back(r.b); }
  g_recs.clear();
  return TG_OK;
}

int tg_prof_dump(const char* path) {
  std::lock_guard<std::mutex> lk(g_mu);
  FILE* f = fopen(path, "w");
  if (!f) { set_error("tg_prof_dump: cannot open %s", path); return TG_ERR_INVALID; }
  fprintf(f, "class,ms,gflop,gbytes,desc\n");
  for (auto& r : g_recs) {
    float dt = 0;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&dt, r.a, r.b) != hipSuccess) dt = -1;
    fprintf(f, "%s,%.5f,%.4f,%.5f,%s\n", kClassNames[r.cls], dt, r.flops / 1e9, r.bytes / 1e9, r.desc);
  }
  fclose(f);
  return TG_OK;
}

int tg_prof_num_classes(void) { return PC_COUNT; }
const char* tg_prof_class_name(int cls) { return (cls >= 0 && cls < PC_COUNT) ? kClassNames[cls] : "?"; }

int tg_prof_collect(int cls, double* ms, int64_t* launches, double* flops, double* bytes) {
  std::lock_guard<std::mutex> lk(g_mu);
  double t = 0, f = 0, b = 0;
  int64_t n = 0;
  for (auto& r : g_recs) {
    if (r.cls != cls) continue;
    hipError_t e = hipEventSynchronize(r.b);
    if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize");
    float dt = 0;
    e = hipEventElapsedTime(&dt, r.a, r.b);
    if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime");
    t += dt; f += r.flops; b += r.bytes; ++n;
  }
  if (ms) *ms = t;
  if (launches) *launches = n;
  if (flops) *flops = f;
  if (bytes) *bytes = b;
  return TG_OK;
}

}  // extern "C"
