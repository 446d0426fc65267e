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
// Named switches between code paths that are ALL product code — which one runs is normally
// decided by the shape; tests pin one to run each against the oracle (xf_tune, by name):
//   key_build    0 by shape; 1 the sort-based build (a probe of the table per nonzero + a radix
//                pass on the cell number: the limits' fallback and the tests' independent second
//                implementation); 2 the two-level partition also where one level would do; 3 the keys
//                of a table's first minibatch through the arrival index (not settled at once)
//   old_weight   0 by shape (xf_cells_grad.hip); 1 the gradient + Push kernels READ a step's old w;
//                2 they derive it from (n, z) wherever the table vouches for it
//   lr_gradient  0 by shape; 1 the general kernel also where the dense one applies; 2 / 3 the
//                dense kernel with byte-masked / whole-line stores whatever the touch density
//   owner_pass   an owner's gradient + Pushes for several workers: 0 by shape; 1 the general
//                loop (a sweep per worker); 2 the workers' phases merged (k_lr_grad_ranked);
//                3 the same with 32-bit worker masks; 4 a phase per worker (k_lr_grad_multi)
enum PathSwitch { kPathKeyBuild = 0, kPathOldWeight, kPathLrGradient, kPathOwnerPass, kPathCount };
int path_switch(int which);
void set_path_switch(int which, int v);
inline int key_build_mode() { return path_switch(kPathKeyBuild); }
#ifdef XF_EXPERIMENTS  // timing experiments (tools/): never in a product build
int exp_knob();
void set_exp_knob(int v);
#endif

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
