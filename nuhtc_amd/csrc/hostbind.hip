// Host-thread placement.  On the two-socket hosts of MI355X nodes the command processor fetches every AQL packet (kernel dispatches, event
// markers) from host memory whose cache lines the SUBMITTING thread wrote last: when that thread runs on the socket the GPU is not attached
// to, each packet costs 1.4-2.9 us more (cross-socket snoop in front of the PCIe read), which the back-to-back dense launches of a step
// see as 0.3-0.4 ms per step (the "slow state" of rounds 3-4: DESIGN.md section 5; tools/dev/r04_state_numa*.sh).  Moving the thread moves
// the state at once -- same engine, buffers and queues -- so the remedy is the usual one: run the submitter on the GPU's NUMA node.
#include <sched.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "common.h"

// "0-63,128-191" -> cpu_set_t; false if nothing parses
static bool parse_cpulist(const char* s, cpu_set_t* out) {
  CPU_ZERO(out);
  int n = 0;
  while (*s) {
    while (*s == ',' || isspace((unsigned char)*s)) ++s;
    if (!isdigit((unsigned char)*s)) break;
    char* end = nullptr;
    long a = strtol(s, &end, 10), b = a;
    s = end;
    if (*s == '-') { b = strtol(s + 1, &end, 10); s = end; }
    if (a < 0 || b < a) return false;
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, out); ++n; }
  }
  return n > 0;
}

// a PCI address as sysfs spells it -- "0000:75:00.0": hex digits, ':' and '.' only (the string becomes part of a path)
static bool valid_bdf(const std::string& b) {
  if (b.empty() || b.size() > 32) return false;
  for (char c : b)
    if (!isxdigit((unsigned char)c) && c != ':' && c != '.') return false;
  return b.find(':') != std::string::npos;
}

static bool read_local_cpulist(const char* root, const char* bdf, std::string* out) {
  std::string path = std::string(root) + "/bus/pci/devices/" + bdf + "/local_cpulist";
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[4096];
  const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[n] = 0;
  *out = buf;
  return true;
}

// The mask a thread had before its FIRST placement: every later placement intersects the device's node with THIS mask, not with the
// narrowed one (so a thread that served a GPU of one socket can be re-placed for a GPU of the other), and nuhtc_restore_host_thread
// gives it back.
static thread_local bool t_saved = false;
static thread_local cpu_set_t t_orig;

static int bind_at(const char* root, const char* pci_bdf) {
  if (!root || !pci_bdf) return NUHTC_E_INVALID;
  std::string bdf(pci_bdf);
  for (auto& c : bdf) c = (char)tolower((unsigned char)c);
  if (!valid_bdf(bdf)) return NUHTC_E_INVALID;
  std::string list;
  if (!read_local_cpulist(root, bdf.c_str(), &list)) return NUHTC_E_NOTFOUND;
  cpu_set_t local, cur, both;
  if (!parse_cpulist(list.c_str(), &local)) return NUHTC_E_NOTFOUND;      // no NUMA information for the device (single-node host, VM)
  if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return NUHTC_E_STATE;
  const cpu_set_t base = t_saved ? t_orig : cur;
  CPU_AND(&both, &local, &base);
  if (CPU_COUNT(&both) == 0) return NUHTC_E_STATE;                        // the caller's own mask excludes the local node: left as it is
  if (CPU_EQUAL(&both, &cur)) return 0;                                   // already there
  if (sched_setaffinity(0, sizeof(both), &both) != 0) return NUHTC_E_STATE;
  if (!t_saved) { t_orig = cur; t_saved = true; }
  return 0;
}

extern "C" int nuhtc_bind_host_thread_pci(const char* pci_bdf) { return bind_at("/sys", pci_bdf); }

// test entry point: the same parser and mask arithmetic against a sysfs tree somewhere else (tests/test_host.py builds one)
extern "C" int nuhtc_bind_host_thread_at(const char* sysfs_root, const char* pci_bdf) { return bind_at(sysfs_root, pci_bdf); }

extern "C" int nuhtc_bind_host_thread(int device) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return NUHTC_E_HIP;
  return nuhtc_bind_host_thread_pci(bdf);
}

extern "C" int nuhtc_restore_host_thread(void) {
  if (!t_saved) return 0;
  if (sched_setaffinity(0, sizeof(t_orig), &t_orig) != 0) return NUHTC_E_STATE;
  t_saved = false;
  return 0;
}
