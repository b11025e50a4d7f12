// Happens-before checker of the wave64 executor: TEST INFRASTRUCTURE ONLY (see race.cpp).
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace emu {
namespace race {

// what the executor tells the checker about the lane that is running (one per OS thread = workgroup)
struct Where {
  uint32_t wg = 0;        // linear workgroup index of the launch
  int lane = -1;          // linear thread index inside the workgroup (-1: no lane is running -- host code)
  uint32_t bepoch = 0;    // workgroup barriers passed so far
  const char* kname = "";
  void* wgstate = nullptr;   // the checker's per-workgroup state
  char* lds_lo = nullptr;    // this OS thread's block of thread-local storage of the library = the LDS of the workgroup
  char* lds_hi = nullptr;
  char* own_lo = nullptr;    // the executor's own thread-local variables that kernels read (threadIdx ... gridDim): not LDS
  char* own_hi = nullptr;
  void* lds_shadow = nullptr;
  uint32_t lds_stamp = 0;
};
extern thread_local Where tl_where;

bool on();                                   // PCC_EMU_RACE=1
void* arena_alloc(size_t bytes);             // device (and pinned host) memory comes out of one arena: it has a shadow
bool arena_owns(const void* p);
void launch_begin(const char* kernel, uint32_t workgroups);
void launch_end();
void workgroup_begin();                      // on the workgroup's OS thread
void workgroup_end();
void access(uintptr_t addr, size_t n, bool store, void* pc);   // a plain load or store of instrumented code
void atomic_begin(const void* p, size_t n, int order, int scope, int kind, void* pc);  // kind 0 load, 1 store, 2 read-modify-write
void atomic_end();
void fence(int order, const char* scope);
// an atomic builtin / fence of instrumented code that did not come through a hook (no scope known: booked as agent scope);
// inside a hook's own atomic: nothing
void atomic_begin_bare(const void* p, size_t n, int order, int kind, void* pc);
void atomic_end_bare();
void fence_bare(int order);

}  // namespace race
}  // namespace emu
