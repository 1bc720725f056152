// Library identity + error strings.
#include "dm_common.h"

extern "C" const char *dm_version(void) { return "detmatch_hip 0.1 (gfx950)"; }

extern "C" const char *dm_error_string(int code) {
  switch (code) {
    case DM_OK: return "ok";
    case DM_ERR_INVALID_ARG: return "invalid argument";
    case DM_ERR_WORKSPACE: return "workspace too small";
    case DM_ERR_INT32_RANGE: return "batch * volume exceeds the int32 cell-id range";
    case DM_ERR_UNSUPPORTED: return "unsupported channel count / kernel volume";
    case DM_ERR_LAUNCH: return "HIP launch / runtime error";
    default: return "unknown error";
  }
}

// ---- per-launch event timing ------------------------------------------------
#include <vector>
namespace {
struct ProfRec {
  hipEvent_t e0, e1;
  int kind, a, b, c, rows, kvol;
  unsigned long long table;
  bool closed;
};
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
}  // namespace

int dm_prof_begin(hipStream_t st, int kind, int a, int b, int c, int rows, int kvol,
                  const void *table) {
  if (!g_prof_on) return -1;
  ProfRec r;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
  r.kind = kind; r.a = a; r.b = b; r.c = c; r.rows = rows; r.kvol = kvol;
  r.table = (unsigned long long)(uintptr_t)table;
  r.closed = false;
  (void)hipEventRecord(r.e0, st);
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

int dm_prof_open(int kind, int a, int b, int c, int rows, int kvol, const void *table, hipEvent_t *e0,
                 hipEvent_t *e1) {
  if (!g_prof_on) return -1;
  ProfRec r;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return -1;
  r.kind = kind; r.a = a; r.b = b; r.c = c; r.rows = rows; r.kvol = kvol;
  r.table = (unsigned long long)(uintptr_t)table;
  r.closed = true;               // the launch itself fills both events
  *e0 = r.e0;
  *e1 = r.e1;
  g_recs.push_back(r);
  return (int)g_recs.size() - 1;
}

void dm_prof_end(int idx, hipStream_t st) {
  if (idx < 0 || idx >= (int)g_recs.size()) return;
  (void)hipEventRecord(g_recs[idx].e1, st);
  g_recs[idx].closed = true;
}

extern "C" int dm_profile_enable(int on) {
  for (auto &r : g_recs) {
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_recs.clear();
  g_prof_on = on != 0;
  return DM_OK;
}

extern "C" int dm_profile_count(void) { return (int)g_recs.size(); }

extern "C" int dm_profile_get(int i, int *kind, int *a, int *b, int *c, int *rows, int *kvol,
                              unsigned long long *table, float *ms) {
  if (i < 0 || i >= (int)g_recs.size() || !g_recs[i].closed) return DM_ERR_INVALID_ARG;
  ProfRec &r = g_recs[i];
  if (hipEventSynchronize(r.e1) != hipSuccess) return DM_ERR_LAUNCH;
  float t = 0.f;
  if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return DM_ERR_LAUNCH;
  *kind = r.kind; *a = r.a; *b = r.b; *c = r.c; *rows = r.rows; *kvol = r.kvol;
  *table = r.table; *ms = t;
  return DM_OK;
}
