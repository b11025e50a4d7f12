// Linked into the libraries tools/known_good/build.sh makes from OLDER commits (libpcc_hip_r02.so, libpcc_hip_r04x.so): the
// entry points include/pcc_codec*.h have gained since, in terms of what those builds had, so that HEAD's binding, the parity
// subset and the timing tools can load them.  Every definition is weak: where the old build has the symbol itself, its own wins.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

struct pcc_pipeline;
struct pcc_entropy_batch;
#define PCC_COMPAT extern "C" __attribute__((weak))
extern "C" {
size_t pcc_host_range_encode(const uint8_t* in, size_t n, uint8_t* out, size_t out_cap);
int pcc_pipeline_workers(pcc_pipeline*);    // both old builds have these two
int pcc_pipeline_contexts(pcc_pipeline*);
}
PCC_COMPAT int pcc_pipeline_last_entropy_mode(pcc_pipeline*) { return 0; }   // (round 2's pipeline codes on the host unless it is told otherwise)
PCC_COMPAT int pcc_host_range_encode_many(int count, const uint8_t* const* in, const size_t* n, uint8_t* const* out, const size_t* out_cap, size_t* out_len) {
  if (count < 1 || count > 4) return -1;   // PCC_ERR_ARG
  for (int i = 0; i < count; ++i) out_len[i] = pcc_host_range_encode(in[i], n[i], out[i], out_cap[i]);
  return 0;
}
PCC_COMPAT int pcc_pipeline_get(pcc_pipeline* p, const char* name) {
  if (!p || !name) return -1;
  if (!strcmp(name, "workers")) return pcc_pipeline_workers(p);
  if (!strcmp(name, "contexts")) return pcc_pipeline_contexts(p);
  if (!strcmp(name, "last_entropy_mode")) return pcc_pipeline_last_entropy_mode(p);
  if (!strcmp(name, "frames_per_coder_call")) return 0;   // (not known from outside those builds)
  if (!strcmp(name, "numa_node")) return -100;            // PCC_NO_NUMA_NODE: those builds split the allowed cores evenly
  if (!strcmp(name, "gpu_threads") || !strcmp(name, "rc_device_lanes") || !strcmp(name, "entropy_gpu_batch")) return 0;
  return -1;
}
PCC_COMPAT int pcc_entropy_batch_set_option(pcc_entropy_batch*, const char*, int) { return 0; }   // (the form was a process-wide option there)
PCC_COMPAT int pcc_debug_host_rc_wide(void) { return 0; }
PCC_COMPAT int pcc_debug_pipeline_cpus(pcc_pipeline*, int, int*, int) { return -1; }
// (round 6: placement by NUMA node; the older builds split the allowed cores evenly and say so)
PCC_COMPAT int pcc_debug_device_pci_bus_id(int, char* out, int cap) { if (out && cap > 0) out[0] = 0; return -2; }
PCC_COMPAT int pcc_debug_device_numa_node(int, const char*) { return -1; }
PCC_COMPAT int pcc_debug_numa_plan(const char*, const char* const*, int, const int*, int, int*, int*, int*, int) { return -1; }
PCC_COMPAT int pcc_debug_address_node(const void*) { return -1; }
