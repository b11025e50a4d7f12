// Linked into libpcc_hip_r02.so only (tools/known_good/build.sh): the two entry points include/pcc_codec.h has gained since
// the commit that library is built from, in terms of what it had, so that HEAD's binding, tests and timing tools can load it.
#include <stddef.h>
#include <stdint.h>

struct pcc_pipeline;
extern "C" {
size_t pcc_host_range_encode(const uint8_t* in, size_t n, uint8_t* out, size_t out_cap);
int pcc_pipeline_last_entropy_mode(pcc_pipeline*) { return 0; }   // that build's pipeline codes on the host unless it is told otherwise
int pcc_host_range_encode_many(int count, const uint8_t* const* in, const size_t* n, uint8_t* const* out, const size_t* out_cap, size_t* out_len) {
  if (count < 1 || count > 4) return -1;   // PCC_ERR_ARG
  for (int i = 0; i < count; ++i) out_len[i] = pcc_host_range_encode(in[i], n[i], out[i], out_cap[i]);
  return 0;
}
}
