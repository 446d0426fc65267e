// xf_device.h — device-side structs and the scalar arithmetic of the path (gfx950).
// Compiled with -ffp-contract=off: the reference is built without FMA contraction
// (CMakeLists.txt:6-8), and hipcc's default correctly-rounded fp32 sqrt/divide is kept.
#ifndef XF_DEVICE_H_
#define XF_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xflow_amd.h"

namespace xf {

constexpr uint64_t kEmptyKey = 0xFFFFFFFFFFFFFFFFull;
constexpr uint32_t kNoRow = 0xFFFFFFFFu;
constexpr unsigned kErrFull = 1u;
constexpr unsigned kErrForeignKey = 2u;
constexpr unsigned kErrDupKey = 4u;
constexpr int kBaseWin = 4;  // settled-tier keys compared per probe round
constexpr int kCoarse = 16;  // keys per bucket of the coarse directory

struct TableStat {
  unsigned long long count;  // state rows handed out == keys stored
  unsigned int err;          // kErr* bits, sticky
  unsigned int spare_used;   // the reserved key value has been stored
};

struct TableDev {
  uint64_t cap;       // positions of the key index; position `cap` is the spare one
  uint64_t max_rows;  // state rows allocated; row `max_rows` is a write-off row for errors
  uint64_t lo;        // first key of this shard's range
  uint64_t span;      // UINT64_MAX / nshards
  uint64_t mult;      // floor(cap * 2^64 / span)
  uint64_t seed;
  uint64_t *keys;     // [cap+1]   key index (open addressing, order-preserving home)
  uint32_t *rows;     // [cap+1]   state row of the key stored at that position
  float *w;           // [(max_rows+1)*dim] dense weights, row-major
  float2 *nz;         // [(max_rows+1)*dim] FTRL accumulators {n, z} side by side (FTRL only):
                      // the Push reads and writes them together, one 8-byte access each
  // settled tier (built by xf_table_defrag): the keys known at the last defrag, sorted and
  // dense; the key at rank r owns state row r.  Found through a bucket directory over the
  // same order-preserving map as `home`, so a sorted key list sweeps bkeys[] once, at 8 bytes
  // per stored key instead of the 24 of the half-empty open-addressing index.  Keys that
  // arrive later live in keys[]/rows[] until the next defrag.
  uint64_t nbase;         // keys in the settled tier == rows 0..nbase-1
  uint64_t ndir;          // directory buckets
  uint64_t dmult;         // bucket = mulhi64(key - lo, dmult), clamped to ndir-1
  const uint64_t *bkeys;  // [nbase + kBaseWin] ascending, padded
  const uint32_t *bdir;   // [ndir+1] bdir[b] = keys in buckets < b
  // a second, COARSE directory over the same keys (kCoarse keys per bucket on average) for key
  // lists in no order (the raw keys of a minibatch): it is small enough to live in an XCD's L2
  // (2.5 MB for 1e7 keys), so a lookup costs one HBM access — the keys around the interpolated
  // position — instead of two
  uint64_t ncdir, cmult;
  const uint32_t *cdir;   // [ncdir+1]
  TableStat *stat;
  int dim, init_kind;
  float init_const;
  float alpha, beta, lambda1, lambda2, lr;
  double inv_alpha;  // 1.0 / (double)alpha (ftrl_step)
  bool last_shard, single;
  // FTRL, dim 1: every row's w is ftrl_w_of(n, z) under the table's hyper-parameters — true of a
  // table whose rows were all written by ftrl_step (or never: w = n = z = 0); the table keeps
  // track (xf_table.hip: w_tainted).  The gradient + Push kernels then do not READ w: they
  // derive the old weight from the (n, z) they load anyway — 4 of a step's 28 bytes.
  bool w_of_nz;
};

__device__ __forceinline__ void load_nz(const TableDev &T, size_t o, float &n, float &z) {
  const float2 s = T.nz[o];
  n = s.x;
  z = s.y;
}
__device__ __forceinline__ void store_nz(const TableDev &T, size_t o, float n, float z) {
  T.nz[o] = make_float2(n, z);
}

// ps-lite uniform key-range ownership
__device__ __forceinline__ bool owns(const TableDev &T, uint64_t key) {
  if (T.single) return true;
  if (key < T.lo) return false;
  return T.last_shard || (key - T.lo) < T.span;
}

// order-preserving home slot: floor((key - lo) * cap / span), clamped
__device__ __forceinline__ uint64_t home_of(const TableDev &T, uint64_t key) {
  const uint64_t h = __umul64hi(key - T.lo, T.mult);
  return h < T.cap ? h : T.cap - 1;
}

__device__ __forceinline__ uint64_t bucket_of(const TableDev &T, uint64_t key) {
  const uint64_t h = __umul64hi(key - T.lo, T.dmult);
  return h < T.ndir ? h : T.ndir - 1;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

// Deterministic first-touch value for FTRL v rows.  The reference draws N(0,1)*1e-2 from a
// time-seeded engine (ftrl.h:114-120, base.h:33-44), which no implementation can
// reproduce; this is an integer Irwin-Hall(12) normal, identical bits on any machine.
__device__ __forceinline__ float hashnorm(uint64_t seed, uint64_t key, uint32_t j) {
  const uint64_t base = mix64(seed ^ mix64(key)) + (uint64_t)j * 0xd1342543de82ef95ull;
  uint64_t sum = 0;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const uint64_t r = mix64(base + (uint64_t)t);
    sum += (r & 0xffff) + ((r >> 16) & 0xffff) + ((r >> 32) & 0xffff) + (r >> 48);
  }
  const double x = ((double)(int64_t)sum - 393210.0) / 65536.0;
  return (float)(x * 1e-2);
}

// One FTRL-proximal coordinate step, ftrl.h:59-74 statement for statement: fp32, the OLD w
// in the z update, left-to-right evaluation, no fused multiply-add.
// x / d for a divisor that does not change from call to call (inv_d = 1.0 / (double)d, from the
// host or hoisted by the compiler): the double product with the reciprocal, rounded to float.
// An fp32 division is ~11 instructions of a VALU-bound step, this is three, and it is the SAME
// float as x / d: the quotient of two floats with a normal result is never a rounding midpoint
// and lies at least 2^-49 (relative) away from one, the double product is within 2^-52 of it.
// Where that argument does not reach — a result in or next to the subnormal range, where exact
// midpoints exist (x = 25000 * 2^-149, d = 50000; the bare product differs at 1308 values of x for
// that d, all with subnormal quotients) — the division itself is done (a branch no wavefront
// takes in practice; x == 0 stays on the fast path: +-0 either way).  Checked on the host for
// all 2^32 values of x at five divisors and for random (x, d) pairs
// (tests/test_div_by_alpha_cpu.py), on the GPU by the bit-exact parity tests.
//
// `exact` (round 5): the table has checked its alpha on the GPU — every one of the 2^32 values of
// x, the bare product against the division, when the hyper-parameters were set (xf_table.hip:
// div_exact_for) — and found no difference: the guard's two compares and its branch go too.
// TableDev::inv_alpha carries the verdict in its sign (negative: exact).
__device__ __forceinline__ float div_by_const(float x, float d, double inv_d, bool exact = false) {
#pragma clang fp contract(off)
  const float q = (float)((double)x * inv_d);
  if (exact || fabsf(q) >= 0x1p-125f || x == 0.0f) return q;
  return x / d;
}

// `inv_alpha` = +-1.0 / (double)alpha (TableDev::inv_alpha, set on the host with the
// hyper-parameters; its sign: see div_by_const): the reference divides by alpha twice per step
// (ftrl.h:63,70)
// the weight a step leaves next to (n, z): ftrl.h:66-73 (ftrl_step's last statement).  It is a
// function of the stored n and z alone, so a row that a step wrote holds w == ftrl_w_of(n, z),
// bit for bit — what TableDev::w_of_nz is about.
__device__ __forceinline__ float ftrl_w_of(float alpha, double inv_alpha, float beta, float lambda1,
                                           float lambda2, float n, float z) {
#pragma clang fp contract(off)
  const bool exact = __double2hiint(inv_alpha) < 0;  // (scalar: a kernel argument's sign bit)
  const double ia = __hiloint2double(__double2hiint(inv_alpha) & 0x7FFFFFFF,
                                     __double2loint(inv_alpha));
  if (fabsf(z) <= lambda1) return 0.0f;
  float tmpr = 0.0f;
  if (z > 0.0f) tmpr = z - lambda1;
  if (z < 0.0f) tmpr = z + lambda1;
  const float tmpl = -1.0f * (div_by_const(beta + sqrtf(n), alpha, ia, exact) + lambda2);
  return tmpr / tmpl;
}

__device__ __forceinline__ void ftrl_step(float alpha, double inv_alpha, float beta, float lambda1,
                                          float lambda2, float g, float &w, float &n,
                                          float &z) {
#pragma clang fp contract(off)
  const bool exact = __double2hiint(inv_alpha) < 0;  // (scalar: a kernel argument's sign bit)
  const double ia = __hiloint2double(__double2hiint(inv_alpha) & 0x7FFFFFFF,
                                     __double2loint(inv_alpha));
  const float old_n = n;
  const float nn = old_n + g * g;
  z = z + (g - div_by_const(sqrtf(nn) - sqrtf(old_n), alpha, ia, exact) * w);
  n = nn;
  w = ftrl_w_of(alpha, inv_alpha, beta, lambda1, lambda2, nn, z);
}

// sgd.h:52,96
__device__ __forceinline__ float sgd_step(float lr, float g, float w) {
#pragma clang fp contract(off)
  return w - lr * g;
}

// x / R exactly as the reference computes it — `float /= 1.0 * line_num`, i.e. the fp32 value
// divided in double and rounded back to fp32 (lr_worker.cc:117, fm_worker.cc:150-156) — at the
// price of an fp32 division: for x an fp32 number and R an integer below 2^24 the exact
// quotient is either a float midpoint or at least 2^-49 (relative) away from one, far more
// than the 2^-53 the intermediate double rounding can move it, so rounding the double
// quotient to float gives the correctly rounded fp32 quotient, which is what x / (float)R
// is (hipcc keeps fp32 division correctly rounded).  One fp64 division per (key, factor) was a
// fifth of the FM gradient kernel.
// (div_by_const with 1.0 / R was measured against this: with its guard it saves two
// instructions per call, and R = 50000 is a divisor whose subnormal quotients need the guard)
__device__ __forceinline__ float div_by_rows(float x, uint32_t R) {
#pragma clang fp contract(off)
  if (R < (1u << 24)) return x / (float)R;
  return (float)((double)x / (1.0 * R));
}

// Base::sigmoid, base.h:54-63: double pow with base 2.718281828 (not e), clamp to 1e-6
// below -30 and to exactly 1 above +30.
__device__ __forceinline__ float sigmoid_ref(float x) {
  if (x < -30.0f) return 1e-6f;
  if (x > 30.0f) return 1.0f;
  const double ex = pow(2.718281828, (double)x);
  return (float)(ex / (1.0 + ex));
}

}  // namespace xf
#endif  // XF_DEVICE_H_
