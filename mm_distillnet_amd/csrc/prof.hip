// Optional hipEvent bracketing of kernel families, used by bench.py's roofline leg: events are
// recorded on the SAME stream the kernel is launched on (torch.cuda.Event would only see torch's
// current stream).  Off by default: zero overhead on the product path.
#include "common.h"
#include <vector>
#include <mutex>
#include <cstdio>
#include <cstring>

struct ProfRec { hipEvent_t e0, e1; double flops, bytes; char tag[48]; };
static int g_on[MMD_FAM_COUNT] = {0};
static std::vector<ProfRec> g_recs[MMD_FAM_COUNT];
static hipEvent_t g_cur[MMD_FAM_COUNT];
static std::mutex g_mu;

extern "C" int mmd_prof_is_on(int family) { return (family >= 0 && family < MMD_FAM_COUNT) ? g_on[family] : 0; }

extern "C" int mmd_prof_enable(int family, int on) {
  if (family < 0 || family >= MMD_FAM_COUNT) return MMD_EINVAL;
  g_on[family] = on;
  return MMD_OK;
}

void mmd_prof_begin(int family, hipStream_t s) {
  if (!g_on[family]) return;
  hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, s);
  g_cur[family] = e;
}
static char g_tag[MMD_FAM_COUNT][48];
void mmd_prof_tag(int family, const char* fmt, long long a, long long b, long long c, long long d) {
  if (!g_on[family]) return;
  snprintf(g_tag[family], sizeof(g_tag[family]), fmt, a, b, c, d);
}
static FILE* g_dump = nullptr;
extern "C" int mmd_prof_dump_to(const char* path) {
  if (g_dump) { fclose(g_dump); g_dump = nullptr; }
  if (path) g_dump = fopen(path, "w");
  return MMD_OK;
}
void mmd_prof_end(int family, hipStream_t s, double flops, double bytes) {
  if (!g_on[family]) return;
  hipEvent_t e; hipEventCreate(&e); hipEventRecord(e, s);
  std::lock_guard<std::mutex> lk(g_mu);
  ProfRec r{g_cur[family], e, flops, bytes, {0}};
  memcpy(r.tag, g_tag[family], sizeof(r.tag));
  g_tag[family][0] = 0;
  g_recs[family].push_back(r);
}

// Synchronises the recorded events and returns totals; clears the records.
// out[0]=launches, out[1]=total ms, out[2]=total flops, out[3]=total algorithmic bytes
extern "C" int mmd_prof_collect(int family, double* out) {
  if (family < 0 || family >= MMD_FAM_COUNT || !out) return MMD_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  double ms = 0, fl = 0, by = 0; int n = 0;
  for (auto& r : g_recs[family]) {
    hipEventSynchronize(r.e1);
    float t = 0; hipEventElapsedTime(&t, r.e0, r.e1);
    ms += t; fl += r.flops; by += r.bytes; ++n;
    if (g_dump) fprintf(g_dump, "%d,%s,%.3f,%.0f,%.0f\n", family, r.tag, t * 1e3, r.flops, r.bytes);
    hipEventDestroy(r.e0); hipEventDestroy(r.e1);
  }
  g_recs[family].clear();
  out[0] = n; out[1] = ms; out[2] = fl; out[3] = by;
  return MMD_OK;
}
