// pcc_dev.h -- developer switches (traces, bisecting forms, forced shapes).
//
// The shipped library reads none of them: dev_env() is a constant nullptr there and the code behind every switch folds
// away.  `make dev` (-DPCC_DEV -> ../libpcc_hip_dev.so, "pcc_hip 0.1 (gfx950, dev build)") and the CPU executor of
// tests/emu read them from the environment; tools/ and the bisect ladder of the GPU session use that library through
// PCC_LIB.  What the shipped library does read from the environment is the pipeline's deployment configuration (threads,
// core pinning, entropy stage: pcc_pipeline.cpp) -- see include/pcc_codec.h, "Environment".
//
//   PCC_LEAF_PROBES=uniform   k_leaf_tile: evenly spaced first probes of the parent search (round 2's layout)
//   PCC_LEAF_ROWS=linear      k_leaf_tile: block row = blockIdx instead of per-XCD ranges (round 2's layout)
//   PCC_SORT_SHAPE=narrow|wide   force the 512 x 8 or the 1024 x 4 workgroup shape of the sort passes and the leaf scan
//   PCC_WAIT=event            wait_stream: hipEventSynchronize instead of polling between sleeps
//   PCC_WAIT_STATS, PCC_FINISH_TRACE, PCC_TRACE_DELTA, PCC_DECODE_TRACE, PCC_PIPELINE_TRACE     where the time goes (stderr)
//   PCC_DECODE_SERIAL         decoder: no second host thread
//   PCC_PIPELINE_SPREAD=<x>, PCC_PIPELINE_OWN_STREAMS=1   pipeline: launch spreading, one stream per context
//   PCC_RC_WIDE=0             host range coder: scalar loop although AVX-512 is there
//   PCC_RC_DEVICE=lanes       device range coder: process default of the option "rc_device_lanes"
//   PCC_PACK_UPLOAD=1         host input: default of the option "pack_upload"
//   PCC_SYSFS_ROOT=<dir>      pipelines: the tree the NUMA placement reads instead of /sys (tests: a made-up two-socket host)
#pragma once
#include <stdlib.h>

namespace pcc {
#if defined(PCC_DEV) || defined(PCC_EMU)
inline const char* dev_env(const char* name) { return getenv(name); }
#else
inline const char* dev_env(const char*) { return nullptr; }
#endif
}  // namespace pcc
