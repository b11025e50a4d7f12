// Known-answer kernels for the executor's happens-before checker (race.cpp): TEST INFRASTRUCTURE ONLY.
// Compiled like the product's kernels in the `race` build (outline access instrumentation, -DPCC_EMU_RACE) into
// _build/race_selftest; prints "<case> <distinct reports>" per case.  tests/test_emu_race.py holds the expected numbers:
// every racy hand-off must be reported, every correctly ordered one must not.
#include <hip/hip_runtime.h>

extern "C" size_t pcc_emu_race_report(char* out, size_t cap);

// ---- global memory, workgroup to workgroup.  Workgroup 0 produces data[0..63] and raises flag; workgroup 1 waits and reads.
enum Mode : int {
  kPlainFlag = 0,        // flag written and polled with plain accesses: flag AND data race
  kRelaxedFlag,          // agent-scope relaxed flag, plain data: the flag is fine, the data races (no release / acquire)
  kReleaseAcquire,       // store-release / load-acquire at agent scope: ordered
  kFences,               // relaxed flag between an agent-scope release fence and an acquire fence: ordered
  kWorkgroupScope,       // release / acquire at WORKGROUP scope: orders nothing between workgroups -> data (and flag) race
  kWorkgroupFences,      // fences at workgroup scope around an agent-scope relaxed flag: the data races
  kSelfDescribing,       // the payload travels inside the agent-scope atomic word itself: nothing to report
  kRmwChain,             // three workgroups: 0 and 1 write their halves and fetch_add(release) a counter, 2 waits for 2 (acquire) and reads both: ordered
  kRmwChainRelaxed,      // the same with relaxed fetch_add / load: both halves race
};

__global__ void k_handoff(uint32_t* data, uint32_t* flag, uint32_t* out, int mode) {
  const uint32_t t = threadIdx.x;
  if (mode == kRmwChain || mode == kRmwChainRelaxed) {
    const int order_rel = mode == kRmwChain ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, order_acq = mode == kRmwChain ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED;
    if (blockIdx.x < 2) {
      data[blockIdx.x * 64 + t] = t + 1;
      __syncthreads();
      if (t == 0) {
        if (order_rel == __ATOMIC_RELEASE) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if (t == 0) {
        if (order_acq == __ATOMIC_ACQUIRE) while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 2u) __builtin_amdgcn_s_sleep(1);
        else while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      out[t] = data[t] + data[64 + t];
    }
    return;
  }
  if (blockIdx.x == 0) {
    if (mode == kSelfDescribing) {
      __hip_atomic_store(data + t, 0x80000000u | t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    data[t] = t + 1;
    __syncthreads();
    if (t == 0) {
      switch (mode) {
        case kPlainFlag: *(volatile uint32_t*)flag = 1u; break;
        case kRelaxedFlag: __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
        case kReleaseAcquire: __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); break;
        case kFences:
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        case kWorkgroupScope: __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); break;
        case kWorkgroupFences:
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
      }
    }
  } else {
    if (mode == kSelfDescribing) {
      uint32_t v;
      while (((v = __hip_atomic_load(data + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0x80000000u) == 0u) __builtin_amdgcn_s_sleep(1);
      out[t] = v;
      return;
    }
    if (t == 0) {
      switch (mode) {
        case kPlainFlag: while (*(volatile uint32_t*)flag != 1u) __builtin_amdgcn_s_sleep(1); break;
        case kRelaxedFlag: while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(1); break;
        case kReleaseAcquire: while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(1); break;
        case kFences:
          while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          break;
        case kWorkgroupScope: while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 1u) __builtin_amdgcn_s_sleep(1); break;
        case kWorkgroupFences:
          while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          break;
      }
    }
    __syncthreads();
    out[t] = data[t];
  }
}

// ---- disjoint bytes of one dword from two workgroups (legitimate: the chip writes masked bytes), then the same byte (a race)
__global__ void k_bytes(uint8_t* bytes, int overlap) {
  if (threadIdx.x == 0) bytes[overlap ? 1 : blockIdx.x] = (uint8_t)blockIdx.x;
}

// ---- a buffer written in one launch and read in the next: ordered by the launch boundary
__global__ void k_fill(uint32_t* data) { data[blockIdx.x * 64 + threadIdx.x] = threadIdx.x; }
__global__ void k_read_neighbour(const uint32_t* data, uint32_t* out) { out[blockIdx.x * 64 + threadIdx.x] = data[((blockIdx.x + 1) % gridDim.x) * 64 + threadIdx.x]; }

// ---- LDS, wave to wave inside a workgroup of two waves
__global__ void k_lds(uint32_t* out, int with_barrier, int atomics) {
  __shared__ uint32_t s_v[64];
  __shared__ uint32_t s_cnt;
  const uint32_t t = threadIdx.x;
  if (atomics) {   // both waves add to one LDS word: atomic against atomic is no race; reading it needs the barrier
    if (t == 0) s_cnt = 0;
    __syncthreads();
    atomicAdd(&s_cnt, 1u);
    if (with_barrier) __syncthreads();
    if (t == 64) out[blockIdx.x * 64] = s_cnt;
    return;
  }
  if (t < 64) s_v[t] = t;                 // wave 0 writes
  if (with_barrier) __syncthreads();
  if (t >= 64) out[blockIdx.x * 64 + t - 64] = s_v[t - 64];  // wave 1 reads (every workgroup has an LDS of its own)
}

// ---- LDS read before anybody in the workgroup has written it
__global__ void k_lds_uninitialised(uint32_t* out, int init) {
  __shared__ uint32_t s_u[64];
  const uint32_t t = threadIdx.x;
  if (init) s_u[t] = t;
  __syncthreads();
  out[blockIdx.x * 64 + t] = s_u[63 - t];
}

// ---- global memory, wave to wave inside a workgroup
__global__ void k_global_in_workgroup(uint32_t* buf, uint32_t* out, int with_barrier) {
  const uint32_t t = threadIdx.x;
  if (t < 64) buf[t] = t;
  if (with_barrier) __syncthreads();
  if (t >= 64) out[t - 64] = buf[t - 64];
}

// ---- what the executor refuses to stand in for (it aborts): lanes of ONE wave at cross-lane operations, or at barriers, of two
//      different source lines -- the chip runs each line with the lanes that are there, the executor would serve them as one
__global__ void k_divergent_ballot(uint32_t* out) {
  uint64_t m;
  if (threadIdx.x & 1u) m = __ballot(1);
  else
    m = __ballot(1);
  out[threadIdx.x] = (uint32_t)m;
}
__global__ void k_divergent_barrier(uint32_t* out) {
  if (threadIdx.x < 32u) __syncthreads();
  else
    __syncthreads();
  out[threadIdx.x] = 1u;
}

static int reports() {
  static char text[1 << 16];
  const int n = (int)pcc_emu_race_report(text, sizeof(text));
  if (getenv("RACE_SELFTEST_VERBOSE")) fputs(text, stdout);
  return n;
}

int main(int argc, char** argv) {
  uint32_t *data, *flag, *out;
  hipMalloc(&data, 4096);
  hipMalloc(&flag, 256);
  hipMalloc(&out, 4096);
  if (argc > 1 && !strcmp(argv[1], "divergent_ballot")) { hipLaunchKernelGGL(k_divergent_ballot, dim3(1), dim3(64), 0, nullptr, out); printf("survived\n"); return 0; }
  if (argc > 1 && !strcmp(argv[1], "divergent_barrier")) { hipLaunchKernelGGL(k_divergent_barrier, dim3(1), dim3(64), 0, nullptr, out); printf("survived\n"); return 0; }
  const char* names[] = {"plain_flag", "relaxed_flag_plain_data", "release_acquire", "agent_fences", "workgroup_scope_atomics", "workgroup_scope_fences",
                         "self_describing_words", "rmw_release_chain", "rmw_relaxed_chain"};
  for (int mode = 0; mode <= kRmwChainRelaxed; ++mode) {
    hipMemset(data, 0, 4096);
    hipMemset(flag, 0, 256);
    const bool chain = mode == kRmwChain || mode == kRmwChainRelaxed;
    hipLaunchKernelGGL(k_handoff, dim3(chain ? 3 : 2), dim3(64), 0, nullptr, data, flag, out, mode);
    printf("%s %d\n", names[mode], reports());
  }
  hipMemset(data, 0, 4096);
  hipLaunchKernelGGL(k_bytes, dim3(2), dim3(64), 0, nullptr, (uint8_t*)data, 0);
  printf("disjoint_bytes_of_a_dword %d\n", reports());
  hipLaunchKernelGGL(k_bytes, dim3(2), dim3(64), 0, nullptr, (uint8_t*)data, 1);
  printf("same_byte_two_workgroups %d\n", reports());
  hipLaunchKernelGGL(k_fill, dim3(4), dim3(64), 0, nullptr, data);
  hipLaunchKernelGGL(k_read_neighbour, dim3(4), dim3(64), 0, nullptr, data, out);
  printf("across_launches %d\n", reports());
  hipLaunchKernelGGL(k_lds, dim3(2), dim3(128), 0, nullptr, out, 1, 0);
  printf("lds_with_barrier %d\n", reports());
  hipLaunchKernelGGL(k_lds, dim3(2), dim3(128), 0, nullptr, out, 0, 0);
  printf("lds_missing_barrier %d\n", reports());
  hipLaunchKernelGGL(k_lds, dim3(2), dim3(128), 0, nullptr, out, 1, 1);
  printf("lds_atomics_with_barrier %d\n", reports());
  hipLaunchKernelGGL(k_lds, dim3(2), dim3(128), 0, nullptr, out, 0, 1);
  printf("lds_atomics_missing_barrier %d\n", reports());
  hipLaunchKernelGGL(k_lds_uninitialised, dim3(2), dim3(64), 0, nullptr, out, 1);
  printf("lds_written_before_read %d\n", reports());
  hipLaunchKernelGGL(k_lds_uninitialised, dim3(2), dim3(64), 0, nullptr, out, 0);
  printf("lds_read_uninitialised %d\n", reports());
  hipLaunchKernelGGL(k_global_in_workgroup, dim3(1), dim3(128), 0, nullptr, data, out, 1);
  printf("global_in_workgroup_with_barrier %d\n", reports());
  hipLaunchKernelGGL(k_global_in_workgroup, dim3(1), dim3(128), 0, nullptr, data, out, 0);
  printf("global_in_workgroup_missing_barrier %d\n", reports());
  return 0;
}
