// hip/hip_runtime.h of the wave64 executor (tests/emu): TEST INFRASTRUCTURE ONLY.
//
// The product's kernel sources (cwi-pcl-codec_amd/csrc/*.hip, pcc_api.cpp) are compiled unmodified by g++ against
// this header into tests/emu/_build/libpcc_emu.so, and run on the CPU: a workgroup is an OS thread, a lane is a fibre,
// the 64 lanes of a wave meet at every cross-lane operation (ballot, DPP, readlane, shuffles) and at every barrier, LDS
// is thread-local storage of the workgroup's OS thread, workgroups of a launch run side by side on a pool of threads
// (so look-back polls and tickets really wait for each other).  It exists so that the kernels' LOGIC can be held
// against the oracle in the `-m "not gpu"` tests when no GPU is reachable; it says nothing about speed, and the
// product (libpcc_hip.so, bench.py, smoke()) never loads it.
//
// What is modelled from the CDNA3/4 ISA documents rather than observed: the DPP controls (row_shr, row_bcast:15/31,
// wave_shr:1, quad_perm, row masks, bank masks, bound_ctrl), readlane, ds_bpermute shuffles.  Where a kernel relies on
// the 64 lanes of a wave executing LDS operations in program order (the match ranking of k_sort_pass) the source says
// so with __builtin_amdgcn_wave_barrier(), which is a meeting point here and no instruction on the GPU.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <tuple>
#include <type_traits>
#include <utility>

#define PCC_EMU 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __constant__
#define HIP_SYMBOL(x) (&(x))

// ---------------------------------------------------------------- vector types
struct dim3 {
  uint32_t x, y, z;
  constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) double2 { double x, y; };
struct float3 { float x, y, z; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

namespace emu {
struct Idx3 { uint32_t x, y, z; };
// cross-lane operations: every active lane of the wave calls the same one; the wave's scheduler computes the results
// `site` = the source line of the call (__builtin_LINE() as a default argument is evaluated where the call is written): the lanes
// of a wave that meet at a cross-lane operation, and the lanes of a wave that wait at a barrier, must have come there through the
// same line -- on the chip lanes at two different instructions never execute them together (each runs with the lanes that are
// there), which the executor, serving all waiting lanes as one operation, could not reproduce.  (The line, not the return
// address: the host compiler may clone a loop body, e.g. for a loop-invariant `lane == 0`, which a GPU compiler must not do
// around convergent operations.)
uint64_t ballot(int pred, int site = __builtin_LINE());
int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int site = __builtin_LINE());
uint32_t readlane(uint32_t v, int lane, int site = __builtin_LINE());
uint32_t readfirstlane(uint32_t v, int site = __builtin_LINE());
uint64_t shfl64(uint64_t v, int src_lane, int width, int mode, int site);  // mode 0: idx, 1: xor, 2: up, 3: down
void wave_barrier(int site = __builtin_LINE());
void syncthreads(int site = __builtin_LINE());
int syncthreads_and(int pred, int site = __builtin_LINE());
int syncthreads_or(int pred, int site = __builtin_LINE());
int syncthreads_count(int pred, int site = __builtin_LINE());
void sleep(int n);
int getreg(int imm);  // HW_REG_XCC_ID (id 20): workgroup b "runs on XCD" b mod 8, as observed on the chip; others: 0
unsigned long long wall_clock();
struct Launcher {
  virtual void run_lane() = 0;
  virtual ~Launcher() {}
};
void launch(void* stream, const char* name, dim3 grid, dim3 block, Launcher& l);

template <typename... KArgs>
struct BoundKernel : Launcher {
  void (*fn)(KArgs...);
  std::tuple<std::decay_t<KArgs>...> args;
  template <typename... A>
  BoundKernel(void (*f)(KArgs...), A&&... a) : fn(f), args(std::forward<A>(a)...) {}
  void run_lane() override { std::apply(fn, args); }
};
template <typename... KArgs, typename... A>
inline void launch_kernel(void* stream, const char* name, void (*fn)(KArgs...), dim3 grid, dim3 block, A&&... a) {
  BoundKernel<KArgs...> b(fn, std::forward<A>(a)...);
  launch(stream, name, grid, block, b);
}
}  // namespace emu

extern thread_local emu::Idx3 threadIdx, blockIdx, blockDim, gridDim;
static constexpr int warpSize = 64;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch_kernel((void*)(stream), #kernel, kernel, grid, block, ##__VA_ARGS__)

// ---------------------------------------------------------------- device intrinsics
#define __syncthreads() emu::syncthreads()
#define __syncthreads_and(p) emu::syncthreads_and(p)
#define __syncthreads_or(p) emu::syncthreads_or(p)
#define __syncthreads_count(p) emu::syncthreads_count(p)
#define __ballot(p) emu::ballot((p) ? 1 : 0)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu::update_dpp(old, src, ctrl, rm, bm, bc)
#define __builtin_amdgcn_readlane(v, l) ((int)emu::readlane((uint32_t)(v), l))
#define __builtin_amdgcn_readfirstlane(v) ((int)emu::readfirstlane((uint32_t)(v)))
#define __builtin_amdgcn_s_sleep(n) emu::sleep(n)
#define __builtin_amdgcn_s_getreg(imm) emu::getreg(imm)
#define __builtin_amdgcn_wave_barrier() emu::wave_barrier()
#ifndef PCC_EMU_RACE
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(order)
#endif
#define __builtin_amdgcn_s_barrier() emu::syncthreads()
#define wall_clock64() emu::wall_clock()
#define clock64() ((long long)emu::wall_clock())

#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#ifndef PCC_EMU_RACE
#define __hip_atomic_load(p, order, scope) __atomic_load_n(p, order)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(p, v, order)
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add(p, v, order)
#define __hip_atomic_fetch_or(p, v, order, scope) __atomic_fetch_or(p, v, order)
#define EMU_ATOMIC(p, kind) do { } while (0)
#else
// The `race` build (tests/emu/race.cpp): every atomic tells the happens-before checker its address, memory order and
// scope before it executes (the checker books it as an atomic access and does the release half of the bookkeeping) and
// after it (the acquire half); the instrumented load / store inside is then not counted a second time.
namespace emu {
namespace race {
void atomic_begin(const void* p, size_t n, int order, int scope, int kind, void* pc);
void atomic_end();
void fence(int order, const char* scope);
}
struct RaceAtomic {
  __attribute__((noinline)) RaceAtomic(const void* p, size_t n, int order, int scope, int kind) { race::atomic_begin(p, n, order, scope, kind, __builtin_return_address(0)); }
  __attribute__((noinline)) ~RaceAtomic() { race::atomic_end(); }
};
template <typename T> static inline T race_load(const T* p, int order, int scope) { RaceAtomic g(p, sizeof(T), order, scope, 0); return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
template <typename T, typename V> static inline void race_store(T* p, V v, int order, int scope) { RaceAtomic g(p, sizeof(T), order, scope, 1); __atomic_store_n(p, (T)v, __ATOMIC_SEQ_CST); }
template <typename T, typename V> static inline T race_fetch_add(T* p, V v, int order, int scope) { RaceAtomic g(p, sizeof(T), order, scope, 2); return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }
template <typename T, typename V> static inline T race_fetch_or(T* p, V v, int order, int scope) { RaceAtomic g(p, sizeof(T), order, scope, 2); return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); }
}  // namespace emu
#define __hip_atomic_load(p, order, scope) emu::race_load(p, order, scope)
#define __hip_atomic_store(p, v, order, scope) emu::race_store(p, v, order, scope)
#define __hip_atomic_fetch_add(p, v, order, scope) emu::race_fetch_add(p, v, order, scope)
#define __hip_atomic_fetch_or(p, v, order, scope) emu::race_fetch_or(p, v, order, scope)
#define __builtin_amdgcn_fence(order, scope) emu::race::fence(order, scope)   /* (the executor's memory is sequentially consistent: nothing else to do) */
// atomicAdd and friends: relaxed, agent scope (what HIP's definitions are); kind 2 = read-modify-write
#define EMU_ATOMIC(p, kind) emu::RaceAtomic emu_guard_(p, sizeof(*(p)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT, kind)
#endif

template <typename T>
static inline T emu_shfl(T v, int a, int width, int mode, int site) {
  static_assert(sizeof(T) <= 8, "shuffle of at most 64 bits");
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  raw = emu::shfl64(raw, a, width, mode, site);
  T out;
  memcpy(&out, &raw, sizeof(T));
  return out;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64, int site = __builtin_LINE()) { return emu_shfl(v, src, width, 0, site); }
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64, int site = __builtin_LINE()) { return emu_shfl(v, mask, width, 1, site); }
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64, int site = __builtin_LINE()) { return emu_shfl(v, (int)d, width, 2, site); }
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64, int site = __builtin_LINE()) { return emu_shfl(v, (int)d, width, 3, site); }

// atomics (workgroups are OS threads: these have to be real ones)
template <typename T> static inline T atomicAdd(T* p, T v) { EMU_ATOMIC(p, 2); return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline float atomicAdd(float* p, float v) {
  EMU_ATOMIC(p, 2);
  uint32_t* q = reinterpret_cast<uint32_t*>(p);
  uint32_t o = __atomic_load_n(q, __ATOMIC_RELAXED);
  for (;;) {
    float old, want;
    memcpy(&old, &o, 4);
    want = old + v;
    uint32_t w;
    memcpy(&w, &want, 4);
    if (__atomic_compare_exchange_n(q, &o, w, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return old;
  }
}
static inline double atomicAdd(double* p, double v) {
  EMU_ATOMIC(p, 2);
  uint64_t* q = reinterpret_cast<uint64_t*>(p);
  uint64_t o = __atomic_load_n(q, __ATOMIC_RELAXED);
  for (;;) {
    double old, want;
    memcpy(&old, &o, 8);
    want = old + v;
    uint64_t w;
    memcpy(&w, &want, 8);
    if (__atomic_compare_exchange_n(q, &o, w, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return old;
  }
}
template <typename T> static inline T atomicSub(T* p, T v) { EMU_ATOMIC(p, 2); return __atomic_fetch_sub(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicOr(T* p, T v) { EMU_ATOMIC(p, 2); return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicAnd(T* p, T v) { EMU_ATOMIC(p, 2); return __atomic_fetch_and(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicXor(T* p, T v) { EMU_ATOMIC(p, 2); return __atomic_fetch_xor(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicExch(T* p, T v) { EMU_ATOMIC(p, 2); return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <typename T> static inline T atomicCAS(T* p, T expect, T v) {
  EMU_ATOMIC(p, 2);
  __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return expect;
}
template <typename T> static inline T atomicMin(T* p, T v) {
  EMU_ATOMIC(p, 2);
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
template <typename T> static inline T atomicMax(T* p, T v) {
  EMU_ATOMIC(p, 2);
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v > old && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
// the mixed-signedness calls HIP's overload set accepts
static inline unsigned atomicAdd(unsigned* p, int v) { EMU_ATOMIC(p, 2); return __atomic_fetch_add(p, (unsigned)v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicOr(unsigned long long* p, uint64_t v) { EMU_ATOMIC(p, 2); return __atomic_fetch_or(p, (unsigned long long)v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicMin(unsigned long long* p, uint64_t v) { return atomicMin<unsigned long long>(p, (unsigned long long)v); }
static inline unsigned atomicOr(unsigned* p, int v) { EMU_ATOMIC(p, 2); return __atomic_fetch_or(p, (unsigned)v, __ATOMIC_RELAXED); }

// bit operations with the device's results for zero
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline unsigned __brev(unsigned v) {
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
  return __builtin_bswap32(v);
}
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline uint64_t __umul64hi(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

// individually rounded arithmetic: the file is compiled with -ffp-contract=off -msse2 (no x87, no FMA)
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fsub_rn(float a, float b) { return a - b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fsqrt_rn(float a) { return sqrtf(a); }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline double __dadd_rn(double a, double b) { return a + b; }
static inline double __dsub_rn(double a, double b) { return a - b; }
static inline double __dmul_rn(double a, double b) { return a * b; }
static inline double __ddiv_rn(double a, double b) { return a / b; }
static inline double __dsqrt_rn(double a) { return sqrt(a); }
static inline float __double2float_rn(double d) { return (float)d; }
static inline float __double2float_ru(double d) {
  float f = (float)d;
  if ((double)f < d) f = nextafterf(f, INFINITY);
  return f;
}
static inline float __double2float_rd(double d) {
  float f = (float)d;
  if ((double)f > d) f = nextafterf(f, -INFINITY);
  return f;
}
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline long long __double_as_longlong(double d) { long long u; memcpy(&u, &d, 8); return u; }
static inline double __longlong_as_double(long long u) { double d; memcpy(&d, &u, 8); return d; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

// HIP's min / max overload set in the global namespace
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, int b) { return min(a, (unsigned)b); }
static inline unsigned min(int a, unsigned b) { return min((unsigned)a, b); }
static inline unsigned max(unsigned a, int b) { return max(a, (unsigned)b); }
static inline unsigned max(int a, unsigned b) { return max((unsigned)a, b); }
static inline long min(long a, long b) { return a < b ? a : b; }
static inline long max(long a, long b) { return a > b ? a : b; }
static inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
static inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
using std::isfinite;
using std::isnan;
using std::isinf;

// ---------------------------------------------------------------- host API (one device, synchronous streams)
typedef int hipError_t;
enum : int { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorNoDevice = 100, hipErrorInvalidDevice = 101 };
typedef struct emuStream* hipStream_t;
typedef struct emuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum : unsigned { hipStreamDefault = 0, hipStreamNonBlocking = 1 };
enum : unsigned { hipEventDefault = 0, hipEventBlockingSync = 1, hipEventDisableTiming = 2 };
enum : unsigned { hipHostMallocDefault = 0, hipHostRegisterDefault = 0 };
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1, hipDeviceAttributeMultiprocessorCount = 2, hipDeviceAttributeHostNumaId = 3 };
struct hipDeviceProp_t {
  char name[256];
  char gcnArchName[256];
  size_t totalGlobalMem;
  int multiProcessorCount;
  size_t maxSharedMemoryPerMultiProcessor, sharedMemPerBlock;
  int regsPerMultiprocessor, maxThreadsPerMultiProcessor, warpSize, clockRate;
};
struct hipFuncAttributes { int numRegs; size_t sharedSizeBytes, localSizeBytes; int maxThreadsPerBlock; };

const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipDeviceSynchronize();
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int d);
hipError_t hipDeviceGetPCIBusId(char* out, int len, int d);   // "0000:<d + 1, two hex digits>:00.0": a name for made-up sysfs trees
hipError_t hipMalloc(void** p, size_t bytes);
template <typename T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes); }
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags = 0);
template <typename T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags = 0) { return hipHostMalloc(reinterpret_cast<void**>(p), bytes, flags); }
hipError_t hipHostFree(void* p);
hipError_t hipHostRegister(void* p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind k, hipStream_t s = nullptr);
hipError_t hipMemsetAsync(void* dst, int v, size_t bytes, hipStream_t s = nullptr);
hipError_t hipMemset(void* dst, int v, size_t bytes);
hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t bytes, size_t off, hipMemcpyKind k);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
template <typename F> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* blocks, F, int, size_t) { *blocks = 1; return hipSuccess; }
template <typename F> static inline hipError_t hipFuncGetAttributes(hipFuncAttributes* a, F) { memset(a, 0, sizeof(*a)); return hipSuccess; }
