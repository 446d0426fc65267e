// xf_common.h — error plumbing and small host/device helpers shared by the library.
#ifndef XF_COMMON_H_
#define XF_COMMON_H_

#include <hip/hip_runtime_api.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "xflow_amd.h"

namespace xf {

int set_error(int code, const char *fmt, ...);
int parse_threads();
void set_parse_threads(int n);
// A stream of this process holds work that will never finish (a collective whose peer died):
// device memory must not be freed any more — hipFree would wait for that work, or release what
// queued kernels still write.  Set once by the sharded trainer's stream wait on a time-out;
// Scratch and the trainer's Dev<> buffers leak their memory from then on.
bool device_poisoned();
void scratch_poison();
int exp_knob();
void set_exp_knob(int v);

#define XF_HIP(expr)                                                                  \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess)                                                            \
      return xf::set_error(XF_EHIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr,      \
                           hipGetErrorString(e__));                                   \
  } while (0)

#define XF_REQUIRE(cond, ...)                                   \
  do {                                                          \
    if (!(cond)) return xf::set_error(XF_EINVAL, __VA_ARGS__);  \
  } while (0)

#define XF_TRY(expr)           \
  do {                         \
    int rc__ = (expr);         \
    if (rc__ != XF_OK) return rc__; \
  } while (0)

// Key-range sharding exactly as ps-lite's default slicer would place a key
// (SURVEY §5/§8e): server i owns [span*i, span*(i+1)), the last one takes the remainder.
struct ShardRange {
  uint64_t lo;    // first key owned
  uint64_t span;  // UINT64_MAX / nshards
  uint32_t shard, nshards;
};

static inline ShardRange shard_range(uint32_t shard, uint32_t nshards) {
  ShardRange r;
  r.shard = shard;
  r.nshards = nshards ? nshards : 1;
  r.span = UINT64_MAX / r.nshards;
  r.lo = r.span * shard;
  return r;
}

}  // namespace xf
#endif  // XF_COMMON_H_
