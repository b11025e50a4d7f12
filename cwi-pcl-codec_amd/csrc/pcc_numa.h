// pcc_numa.h -- which cores of the host a GPU's pipeline should use (one pipeline per GPU, no exchange between them).
//
// A node with eight GPUs has two sockets; a GPU hangs off one of them.  The entropy threads of a GPU's pipeline read what its
// device->host copies landed (occupancy bytes, quantised JPEG coefficients) and the GPU-stage threads feed its queues: both
// belong on the cores of the GPU's own socket, or every landing crosses the socket link twice.  (The reference has no
// counterpart: its only parallel loop is the OpenMP one of the inter-frame path, impl.hpp:974.)
//
// Host-only, no HIP: the PCI address of a device comes from the caller (pcc_api.cpp asks the runtime), everything else is
// read from sysfs, whose root is a parameter so that the planning can be tested against a made-up tree
// (tests/test_numa_plan.py: two nodes, eight devices).
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unistd.h>
#include <vector>
#include <sys/syscall.h>

namespace pcc {
namespace numa {

inline bool read_text(const std::string& path, std::string* out) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) return false;
  char buf[4096];
  const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[n] = 0;
  *out = buf;
  return true;
}

// "0-3,8,10-11" -> 0 1 2 3 8 10 11
inline std::vector<int> parse_cpulist(const std::string& s) {
  std::vector<int> out;
  const char* p = s.c_str();
  while (*p) {
    while (*p && !isdigit((unsigned char)*p)) ++p;
    if (!*p) break;
    char* e = nullptr;
    const long a = strtol(p, &e, 10);
    long b = a;
    p = e;
    if (*p == '-') {
      b = strtol(p + 1, &e, 10);
      p = e;
    }
    for (long c = a; c <= b && c - a < 65536; ++c) out.push_back((int)c);
  }
  return out;
}

// <root>/bus/pci/devices/<domain:bus:device.function>/numa_node; -1 where the file is missing or says -1 (one-socket hosts,
// virtual machines).  The runtime prints the address in upper- or lower-case hex; sysfs names are lower-case.
inline int pci_numa_node(const std::string& root, const char* bdf) {
  if (!bdf || !*bdf) return -1;
  std::string name(bdf);
  for (char& ch : name) ch = (char)tolower((unsigned char)ch);
  std::string text;
  if (!read_text(root + "/bus/pci/devices/" + name + "/numa_node", &text)) return -1;
  char* e = nullptr;
  const long v = strtol(text.c_str(), &e, 10);
  return (e == text.c_str() || v < 0) ? -1 : (int)v;
}

// the node of every CPU of `cpus` (-1 where sysfs does not say): <root>/devices/system/node/node<k>/cpulist
inline std::vector<int> nodes_of_cpus(const std::string& root, const std::vector<int>& cpus) {
  std::vector<int> out(cpus.size(), -1);
  int missing = 0;
  for (int k = 0; k < 1024 && missing < 8; ++k) {  // node numbers may have holes (memory-less or offline nodes)
    std::string text;
    if (!read_text(root + "/devices/system/node/node" + std::to_string(k) + "/cpulist", &text)) { ++missing; continue; }
    missing = 0;
    const std::vector<int> of_node = parse_cpulist(text);
    for (size_t i = 0; i < cpus.size(); ++i)
      if (std::find(of_node.begin(), of_node.end(), cpus[i]) != of_node.end()) out[i] = k;
  }
  return out;
}

// What one pipeline gets: the physical cores (one logical CPU each) its entropy threads are spread over, and the node they
// are on (-1: the host did not say, the cores are a plain share of all allowed cores).
struct Share {
  int node = -1;
  std::vector<int> cores;
};

// `cores`: one logical CPU per physical core this process may use, ascending; `core_node`: their nodes; `device_node`: the node
// of every pipeline's GPU, in pipeline order (the same GPU may appear twice).  Pipelines whose GPUs share a node split that
// node's cores evenly, in pipeline order.  If any GPU's node is unknown, or has no core this process may use, every pipeline
// gets a plain share of all cores instead (the placement before there was one): a half-placed host would put two pipelines
// on the same cores.
inline std::vector<Share> plan(const std::vector<int>& cores, const std::vector<int>& core_node, const std::vector<int>& device_node) {
  const size_t nd = device_node.size();
  std::vector<Share> out(nd);
  if (!nd || cores.empty()) return out;
  bool placed = core_node.size() == cores.size();
  for (size_t d = 0; placed && d < nd; ++d)
    placed = device_node[d] >= 0 && std::find(core_node.begin(), core_node.end(), device_node[d]) != core_node.end();
  if (!placed) {
    const size_t span = std::max<size_t>(cores.size() / nd, 1);
    for (size_t d = 0; d < nd; ++d)
      for (size_t k = 0; k < span; ++k) out[d].cores.push_back(cores[(d * span + k) % cores.size()]);
    return out;
  }
  for (size_t d = 0; d < nd; ++d) {
    std::vector<int> of_node;
    for (size_t i = 0; i < cores.size(); ++i)
      if (core_node[i] == device_node[d]) of_node.push_back(cores[i]);
    size_t sharers = 0, mine = 0;  // pipelines on this node, and which of them this one is
    for (size_t e = 0; e < nd; ++e)
      if (device_node[e] == device_node[d]) {
        if (e == d) mine = sharers;
        ++sharers;
      }
    const size_t span = std::max<size_t>(of_node.size() / sharers, 1);
    out[d].node = device_node[d];
    for (size_t k = 0; k < span; ++k) out[d].cores.push_back(of_node[(mine * span + k) % of_node.size()]);
  }
  return out;
}

// the node the page behind `p` lives on right now (move_pages with no target nodes only reports), -1 if the kernel does not
// say (page not touched yet, no permission, not a NUMA kernel): how a test or the bench sees where a landing buffer went
inline int node_of_address(const void* p) {
#ifdef SYS_move_pages
  const long page = sysconf(_SC_PAGESIZE);
  void* addr = (void*)((uintptr_t)p & ~((uintptr_t)page - 1));
  int status = -1;
  if (syscall(SYS_move_pages, 0, 1ul, &addr, (const int*)nullptr, &status, 0) != 0) return -1;
  return status >= 0 ? status : -1;
#else
  (void)p;
  return -1;
#endif
}

}  // namespace numa
}  // namespace pcc
