// Host-thread placement.  On the two-socket hosts of MI355X nodes the command processor fetches every AQL packet (kernel dispatches, event
// markers) from host memory whose cache lines the SUBMITTING thread wrote last: when that thread runs on the socket the GPU is not attached
// to, each packet costs 1.4-2.9 us more (cross-socket snoop in front of the PCIe read), which the back-to-back dense launches of a step
// see as 0.3-0.4 ms per step (the "slow state" of rounds 3-4: DESIGN.md section 5; tools/dev/r04_state_numa*.sh).  Moving the thread moves
// the state at once -- same engine, buffers and queues -- so the remedy is the usual one: run the submitter on the GPU's NUMA node.
#include <sched.h>

#include <cctype>
#include <cstdio>
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

// the cpulist file of a PCI function, optionally under another sysfs root (tests)
static bool read_local_cpulist(const char* bdf, std::string* out) {
  const char* root = getenv("NUHTC_SYSFS_ROOT");
  std::string path = std::string(root ? root : "/sys") + "/bus/pci/devices/" + bdf + "/local_cpulist";
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[4096];
  const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[n] = 0;
  *out = buf;
  return true;
}

extern "C" int nuhtc_bind_host_thread_pci(const char* pci_bdf) {
  if (!pci_bdf) return NUHTC_E_INVALID;
  std::string bdf(pci_bdf);
  for (auto& c : bdf) c = (char)tolower((unsigned char)c);
  std::string list;
  if (!read_local_cpulist(bdf.c_str(), &list)) return NUHTC_E_NOTFOUND;
  cpu_set_t local, cur, both;
  if (!parse_cpulist(list.c_str(), &local)) return NUHTC_E_NOTFOUND;      // no NUMA information for the device (single-node host, VM)
  if (sched_getaffinity(0, sizeof(cur), &cur) != 0) return NUHTC_E_STATE;
  CPU_AND(&both, &local, &cur);
  if (CPU_COUNT(&both) == 0) return NUHTC_E_STATE;                        // the caller's mask excludes the local node: left as it is
  if (CPU_EQUAL(&both, &cur)) return 0;                                   // already there
  return sched_setaffinity(0, sizeof(both), &both) == 0 ? 0 : NUHTC_E_STATE;
}

extern "C" int nuhtc_bind_host_thread(int device) {
  char bdf[64] = {0};
  if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return NUHTC_E_HIP;
  return nuhtc_bind_host_thread_pci(bdf);
}
