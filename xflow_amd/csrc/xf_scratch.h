// xf_scratch.h — scratch memory of the device-side builders: a bump allocator over one
// persistent device arena per host thread.  A build synchronises its stream before it
// returns, so the arena can be reused by the next one; Scratch objects nest like a stack
// (xf_batch_compile_gpu -> xf_batch_compile_dev -> cells_build).
#ifndef XF_SCRATCH_H_
#define XF_SCRATCH_H_

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <vector>

#include "xf_common.h"

namespace xf {

struct Arena {
  char *base = nullptr;
  size_t cap = 0, used = 0, high = 0;  // high = bytes the deepest nesting wanted
};
inline Arena &arena() {
  static thread_local Arena a;
  return a;
}

struct Scratch {
  size_t mark;
  std::vector<void *> overflow;  // when the arena was too small: plain allocations
  size_t spilled = 0;
  Scratch() : mark(arena().used) {}
  Scratch(const Scratch &) = delete;
  Scratch &operator=(const Scratch &) = delete;
  ~Scratch() {
    Arena &a = arena();
    if (device_poisoned()) {  // (xf_common.h: leak, do not wait for a stuck stream)
      a.used = mark;
      return;
    }
    for (void *p : overflow) (void)hipFree(p);
    a.high = std::max(a.high, a.used + spilled);
    a.used = mark;
    if (mark == 0 && a.high > a.cap) {  // outermost scope: grow for the next build
      if (a.base) (void)hipFree(a.base);
      a.base = nullptr;
      a.cap = 0;
      const size_t want = a.high + a.high / 4;
      if (hipMalloc((void **)&a.base, want) == hipSuccess) a.cap = want;
    }
  }
  template <typename T>
  int get(T **p, size_t n) {
    const size_t bytes = ((std::max<size_t>(n, 1) * sizeof(T)) + 255) & ~(size_t)255;
    Arena &a = arena();
    if (a.used == 0 && bytes > a.cap) {
      // nothing is handed out: size the arena for a whole build now (the first request of a
      // build is one of its NNZ-sized arrays, a build takes about eight of them) instead of
      // spilling every request of the first build into its own hipMalloc (50 ms, once)
      if (a.base) (void)hipFree(a.base);
      a.base = nullptr;
      a.cap = 0;
      const size_t want = std::max(bytes * 10, a.high + a.high / 4);
      if (hipMalloc((void **)&a.base, want) == hipSuccess) a.cap = want;
    }
    if (a.used + bytes <= a.cap) {
      *p = (T *)(a.base + a.used);
      a.used += bytes;
      return XF_OK;
    }
    spilled += bytes;
    XF_HIP(hipMalloc((void **)p, bytes));
    overflow.push_back(*p);
    return XF_OK;
  }
};

}  // namespace xf
#endif  // XF_SCRATCH_H_
