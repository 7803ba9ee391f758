// Optional per-kernel timing with HIP events on the launch stream (bench.py's live roofline numbers).
// Disabled by default: a disabled scope costs one branch per launch.
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.h"

struct ProfRec { const char* tag; double flops, bytes; hipEvent_t a, b; const int* m_dev; int m_mul, m_cap; };
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
  g_recs.push_back(ProfRec{tag, flops, bytes, ev.first, ev.second, nullptr, 1, 0});
  idx_ = (int)g_recs.size() - 1;
  hipEventRecord(ev.first, s_);
}

// the launch's row count lives on the device (RoI / detection lists): flops and bytes were given for the capacity `m_cap`
// and are scaled to min(m_cap, *m_dev * m_mul) rows when the records are read
void ProfScope::device_rows(const int* m_dev, int m_mul, int m_cap) {
  if (idx_ >= 0 && m_dev && m_cap > 0) { g_recs[idx_].m_dev = m_dev; g_recs[idx_].m_mul = m_mul; g_recs[idx_].m_cap = m_cap; }
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
  std::map<const int*, int> rows;          // device-side row counts (values of the last launch sequence: steady state)
  for (auto& r : g_recs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    double scale = 1.0;
    if (r.m_dev) {
      auto it = rows.find(r.m_dev);
      if (it == rows.end()) {
        int v = 0;
        if (hipMemcpy(&v, r.m_dev, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) v = r.m_cap;
        it = rows.emplace(r.m_dev, v).first;
      }
      const long long m = (long long)it->second * r.m_mul;
      scale = (double)(m < r.m_cap ? m : r.m_cap) / (double)r.m_cap;
    }
    Acc& a = acc[r.tag];
    a.n++; a.ms += ms; a.flops += r.flops * scale; a.bytes += r.bytes * scale;
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

// ---- development knobs: integer switches of the tile heuristics.  In a -DNUHTC_DEV build they are initialised from the environment
// (NUHTC_<NAME>) on first use and settable at run time so that two settings can be A/B-ed inside one process (tools/dev/knob_ab.py);
// the default build compiles them to their defaults: no environment reads, nuhtc_dev_knob refuses.
#include <cstdlib>
#include <mutex>
static std::map<std::string, int> g_knobs;      // nodes never move: launch code keeps references to the values
static std::mutex g_knob_mu;
int& dev_knob_ref(const char* name, int dflt) {
  std::lock_guard<std::mutex> lock(g_knob_mu);
  auto it = g_knobs.find(name);
  if (it != g_knobs.end()) return it->second;
#ifdef NUHTC_DEV
  const char* e = getenv((std::string("NUHTC_") + name).c_str());
  return g_knobs.emplace(name, e ? atoi(e) : dflt).first->second;
#else
  return g_knobs.emplace(name, dflt).first->second;
#endif
}
int dev_knob(const char* name, int dflt) { return dev_knob_ref(name, dflt); }
extern "C" int nuhtc_dev_knob(const char* name, int value) {
  if (!name) return NUHTC_E_INVALID;
#ifdef NUHTC_DEV
  dev_knob_ref(name, value) = value;
  return 0;
#else
  (void)value;
  return NUHTC_E_STATE;      // not a development build (-DNUHTC_DEV)
#endif
}

// ---- shader clock under load: one wave spins for `ticks` periods of the 100 MHz reference clock (s_memrealtime) and reports how
// many shader cycles (s_memtime) went by.  Launched on a stream of its own beside the kernels being measured it occupies one
// wave slot of one CU; cycles / ticks x 100 MHz is the clock the chip held while those kernels ran (bench.py: roofline.shader_clock).
__global__ void clock_probe_kernel(unsigned long long ticks, unsigned long long* out) {
  if (threadIdx.x != 0) return;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long r = r0;
  while (r - r0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    r = __builtin_amdgcn_s_memrealtime();
  }
  out[0] = __builtin_amdgcn_s_memtime() - c0;
  out[1] = r - r0;
}

extern "C" int nuhtc_clock_probe(int device, uint64_t ticks_100mhz, uint64_t* out_dev, void* stream) {
  if (!out_dev || ticks_100mhz == 0 || ticks_100mhz > 1000000000ull) return NUHTC_E_INVALID;
  if (hipSetDevice(device) != hipSuccess) return NUHTC_E_HIP;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long)ticks_100mhz, (unsigned long long*)out_dev);
  return hipGetLastError() == hipSuccess ? 0 : NUHTC_E_HIP;
}
