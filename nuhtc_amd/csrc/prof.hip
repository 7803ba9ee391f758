// Optional per-kernel timing with HIP events on the launch stream (bench.py's live roofline numbers).
// Disabled by default: a disabled scope costs one branch per launch.
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"

struct ProfRec { const char* tag; double flops, bytes; hipEvent_t a, b; };
static bool g_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
static size_t g_pool_next = 0;

bool prof_enabled() { return g_on; }

ProfScope::ProfScope(const char* tag, double flops, double bytes, hipStream_t s) : s_(s), idx_(-1) {
  if (!g_on) return;
  if (g_pool_next == g_pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    g_pool.push_back({a, b});
  }
  auto& ev = g_pool[g_pool_next++];
  g_recs.push_back(ProfRec{tag, flops, bytes, ev.first, ev.second});
  idx_ = (int)g_recs.size() - 1;
  hipEventRecord(ev.first, s_);
}

ProfScope::~ProfScope() {
  if (idx_ >= 0) hipEventRecord(g_recs[idx_].b, s_);
}

extern "C" int nuhtc_profile_enable(int on) {
  g_on = on != 0;
  g_recs.clear();
  g_pool_next = 0;
  return 0;
}

// Synchronises the device and writes one line per kernel tag: "tag launches total_ms flops bytes\n". Resets the records.
extern "C" int nuhtc_profile_read(char* buf, size_t cap) {
  if (!buf || cap == 0) return NUHTC_E_INVALID;
  if (hipDeviceSynchronize() != hipSuccess) return NUHTC_E_HIP;
  struct Acc { long n = 0; double ms = 0, flops = 0, bytes = 0; };
  std::map<std::string, Acc> acc;
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    Acc& a = acc[r.tag];
    a.n++; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
  }
  std::string out;
  char line[256];
  for (auto& kv : acc) {
    snprintf(line, sizeof(line), "%s %ld %.6f %.6e %.6e\n", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
    out += line;
  }
  g_recs.clear();
  g_pool_next = 0;
  if (out.size() + 1 > cap) return NUHTC_E_CAPACITY;
  memcpy(buf, out.c_str(), out.size() + 1);
  return 0;
}
