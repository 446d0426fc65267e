// xf_cells_grad.h — device helpers of the gradient (+ Push) kernels (internal; included by
// xf_cells_grad.hip and xf_cells_grad_dense.hip).
#ifndef XF_CELLS_GRAD_H_
#define XF_CELLS_GRAD_H_

#include "xf_cells_impl.h"

namespace {
// ----------------------------------------------------------------------------- gradient
// One workgroup per work item = (chunk, slice).  The chunk's kChunk (2048) key sums live in LDS as
// fp64; the item walks its share of the chunk's nwin cells: coalesced entry loads, loss
// gathers that ascend through the window (a cell is sorted by row), one LDS atomic each.
// Unsplit chunks (all of them unless a chunk holds > kSliceMax entries) finish in place:
//   MODE 0  g = sum / R (lr_worker.cc:117), then the optimizer step on the key's state row
//           (ftrl.h:59-74 / sgd.h:52): the Push fused into the gradient, state read and
//           written once, coalesced, and only for the keys this minibatch touched
//   MODE 1  g_out[idx] = g (the multi-GPU worker side: gradients travel to the owners)
// The slices of a split chunk (power-law heads) add their sums into the chunk's accumulators
// in HBM (fp64 atomics) and k_lr_grad_split_finish does the rest.
__device__ __forceinline__ void apply_key(const xf::TableDev &T, int opt, size_t row, float g) {
  if (opt == XF_OPT_FTRL) {
    float w, nn, z;
    xf::load_nz(T, row, nn, z);
    w = T.w_of_nz ? xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, nn, z)
                  : T.w[row];
    xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g, w, nn, z);
    T.w[row] = w;
    xf::store_nz(T, row, nn, z);
  } else {
    T.w[row] = xf::sgd_step(T.lr, g, T.w[row]);
  }
}

// add `val` to acc[k] for every active lane.  Same-address LDS atomics serialise (a wavefront
// whose lanes all hold one key takes ~1 us for one ds_add_f64), and the head keys of a
// power-law minibatch fill whole wavefronts.  So: a key that sits in two neighbouring lanes
// (the cheap test: one cross-lane compare) is summed over all its lanes in registers and lands
// as ONE atomic; up to three such keys per call, everything else one atomic per lane.
// `touched` is a byte per key written with plain stores: every writer stores the same 1.
__device__ __forceinline__ void add_keys(double *acc, uint8_t *touched, bool on, uint32_t k,
                                         double val) {
  const unsigned lane = threadIdx.x & 63u;
  uint32_t kk = on ? k : (0x80000000u | lane);  // inactive lanes: a key nobody else has
  for (int round = 0; round < 3; ++round) {
    const unsigned long long cand = __ballot(kk == (uint32_t)__shfl_xor((int)kk, 1));
    if (!cand) break;  // wave-uniform
    const int leader = __ffsll((long long)cand) - 1;
    const uint32_t k0 = (uint32_t)__shfl((int)kk, leader);
    const bool mine = kk == k0;
    double sum = mine ? val : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((int)lane == leader) {
      atomicAdd(&acc[k0], sum);
      touched[k0] = 1;
    }
    if (mine) kk = 0x80000000u | lane;
  }
  if (!(kk & 0x80000000u)) {
    atomicAdd(&acc[kk], val);
    touched[kk] = 1;
  }
}
// ... of a lane's E entries at once.  A power-law head key fills most of the E slots of every
// lane of its chunk's workgroups, and taken slot by slot it went through E cross-lane
// reductions per round (~150 instructions per entry slot: they were the power-law gradient
// kernel).  Here a key found in two neighbouring lanes of slot `round` is summed over ALL the
// slots of all lanes in registers (fp64 sums of fp32 terms), then over the lanes, and lands as
// ONE atomic; up to three such keys per call, every other entry one atomic.
template <int E>
__device__ __forceinline__ void add_keys_folded(double *acc, uint8_t *touched,
                                                const uint32_t (&ent)[E], const float (&l)[E]) {
  const unsigned lane = threadIdx.x & 63u;
  uint32_t kk[E];  // the key's place in the chunk; bit 31: nothing (left) to add
#pragma unroll
  for (int q = 0; q < E; ++q)
    kk[q] = ent[q] != 0xFFFFFFFFu ? (ent[q] & (kChunk - 1)) : 0x80000000u;
#pragma unroll
  for (int round = 0; round < 3 && round < E; ++round) {
    const uint32_t probe = kk[round];
    const unsigned long long cand =
        __ballot(probe == (uint32_t)__shfl_xor((int)probe, 1) && !(probe & 0x80000000u));
    if (!cand) break;  // wave-uniform
    const int leader = __ffsll((long long)cand) - 1;
    const uint32_t k0 = (uint32_t)__shfl((int)probe, leader);
    double sum = 0.0;
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const bool m = kk[q] == k0;
      sum += m ? (double)l[q] : 0.0;
      kk[q] = m ? 0x80000000u : kk[q];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((int)lane == leader) {
      atomicAdd(&acc[k0], sum);
      touched[k0] = 1;
    }
  }
#pragma unroll
  for (int q = 0; q < E; ++q)
    if (!(kk[q] & 0x80000000u)) {
      atomicAdd(&acc[kk[q]], (double)l[q]);
      touched[kk[q]] = 1;
    }
}

#ifndef XF_GRAD_E
#define XF_GRAD_E 8
#endif
// waves / SIMD the multi-source variant's registers must allow: unbounded the compiler takes
// 176 (2 waves: 451 us for the pass of an owner of 8 workers), 5 -> 85 registers, 6 -> 80 with
// 24 bytes of scratch per lane (245 us)
#ifndef XF_GRAD_MULTI_WAVES
#define XF_GRAD_MULTI_WAVES 6
#endif
constexpr int kGradE = XF_GRAD_E;  // entries per lane and round
constexpr uint32_t kGradWin = 32;  // windows whose slice bounds fit the LDS table
constexpr uint32_t kSparseKeys = 2 * 256;  // touched keys of a chunk that are stepped from a list
}  // namespace

#endif  // XF_CELLS_GRAD_H_
