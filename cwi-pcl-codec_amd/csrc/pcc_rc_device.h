// pcc_rc_device.h -- the static range coder for many independent streams on the GPU (pcc_rc_device.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcc {

struct RcJob {            // one stream (device pointers)
  const uint8_t* in;      // symbols (16-byte aligned for the lane-per-stream form's vector loads; anything else is read byte by byte)
  uint32_t n;
  const uint32_t* hist;   // 256 symbol counts if somebody has them already (k_occ_histogram), else null
  uint8_t* out;           // 1028-byte table + payload + 4 flush bytes; room for 1028 + n + n / 2 + 64 bytes, 4-byte aligned
  uint32_t* out_len;      // bytes written
};

// dev_hists: room for 256 counts per job (the lane-per-stream form makes the counts of streams that come without them
// there first); null: the wave-per-stream form whatever the mode
void launch_range_encode(const RcJob* dev_jobs, uint32_t n_jobs, uint32_t* dev_hists, hipStream_t stream);
// after the coder: the streams packed side by side (stream j at dev_packed + dev_offsets[j], offsets multiples of 16),
// so that one device-to-host copy of the coded bytes brings everything back
void launch_pack_streams(const RcJob* dev_jobs, const uint32_t* dev_offsets, uint8_t* dev_packed, uint32_t n_jobs, hipStream_t stream);

}  // namespace pcc
