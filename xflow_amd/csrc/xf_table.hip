// xf_table.hip — the feature-key-range-sharded parameter table of one GPU and its
// resolve / gather / update kernels (gfx950).
//
// Replaces: ps::KVWorker<float>::Pull/Push/Wait (src/model/lr/lr_worker.cc:170,175;
// src/model/fm/fm_worker.cc:228-242), the server handlers FTRL::KVServerFTRLHandle_{w,v}
// (src/optimizer/ftrl.h:38-152) and SGD::KVServerSGDHandle_{w,v} (src/optimizer/sgd.h:30-109)
// and their std::unordered_map stores (ftrl.h:84, sgd.h:62).
//
// Layout in HBM (per shard):
//   key index   keys[cap+1] u64, rows[cap+1] u32     open addressing, EMPTY = 2^64-1
//               (position `cap` is a spare for the reserved key value)
//   state       w[(max_rows+1)*dim], and for FTRL n[], z[] likewise; DENSE: a key gets the
//               next free row on first touch and keeps it.  No slack in the state arrays
//               (they are what Push streams through), the slack of open addressing is paid
//               only on 12 bytes per key.
// A key's position is found by linear probing from an ORDER-PRESERVING home
//   home = floor((key - range_lo) * cap / range_span)
// Keys are already uniform 64-bit hashes (io.h:53), so a linear map spreads them as well as
// any hash — and the index stays (almost) sorted by key.  Pull/Push key lists are sorted (the
// ps-lite contract), so lane i and lane i+1 of a wave probe neighbouring positions: the
// pointer-chasing unordered_map walk becomes a monotone sweep over keys[] / rows[].
// Rows are handed out per wavefront in key order (one atomicAdd per wave): the keys a sorted
// list inserts together occupy runs of consecutive rows, so later sorted lists touch the
// state in contiguous runs as well.  Same idea as ps-lite's key-range sharding across
// servers, continued inside the GPU.
#include <hip/hip_runtime.h>
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <map>
#include <mutex>
#include <numeric>
#include <vector>

#include "xf_common.h"
#include "xf_device.h"
#include "xf_scratch.h"

namespace {

__global__ void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {  // four loads in flight per lane
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a;
    dst[i + stride] = b;
    dst[i + 2 * stride] = c;
    dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void k_copy4(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const uint32_t a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a;
    dst[i + stride] = b;
    dst[i + 2 * stride] = c;
    dst[i + 3 * stride] = d;
  }
  for (; i < n; i += stride) dst[i] = src[i];
}
__global__ void k_copy1(const unsigned char *__restrict__ src, unsigned char *__restrict__ dst,
                        size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

constexpr int kBlock = 256;
constexpr double kMaxLoad = 0.75;  // state rows allocated per index position

inline int grid_for(size_t n) {
  size_t g = (n + kBlock - 1) / kBlock;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

inline hipStream_t S(void *s) { return (hipStream_t)s; }

// a device allocation that is released on every return path unless handed over with take()
template <typename T>
struct DevBuf {
  T *p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)); }
  T *take() {
    T *q = p;
    p = nullptr;
    return q;
  }
  operator T *() const { return p; }
};

// ---------------------------------------------------------------------------- kernels

// append slot for this lane if `want`: one atomicAdd per wavefront (10^7 lanes bumping one
// counter one by one took milliseconds).  Every lane of the wave must call it.
__device__ __forceinline__ unsigned long long wave_append(unsigned long long *counter,
                                                          bool want) {
  const unsigned long long m = __ballot(want);
  if (!m) return 0;
  const unsigned lane = threadIdx.x & 63u;
  const int leader = __ffsll((long long)m) - 1;
  unsigned long long base = 0;
  if ((int)lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
  base = __shfl(base, leader);
  return base + __popcll(m & ((1ull << lane) - 1ull));
}


// key -> state row with insert-on-miss: `store[key]` of ftrl.h:56 / sgd.h:46, first-touch
// init of ftrl.h:112-121 / sgd.h:67-72.  One key per lane.  Each probe round reads a window
// of kWin consecutive index positions with independent loads (one memory round trip instead
// of up to kWin dependent ones: a wavefront waits for its slowest lane, and the longest of
// 64 linear-probe chains is several positions even at load 0.5).
// The same key may appear several times in one launch (key lists of several workers): the
// lane that wins the insert publishes the row, the others wait for it.
// GATHER (dim == 1, zero-initialised tables or unique keys): also emit the Pull payload
// w[row] (ftrl.h:75-77).
constexpr int kWin = 4;

template <bool GATHER>
__global__ void __launch_bounds__(kBlock)
k_resolve(xf::TableDev T, const uint64_t *__restrict__ keys, size_t n,
          uint32_t *__restrict__ rows_out, float *__restrict__ wu,
          const uint32_t *__restrict__ list, const unsigned long long *__restrict__ list_n,
          const uint32_t *__restrict__ out_idx /* null, or where entry i's results go */) {
  // with a work list (the keys k_pull_settled did not find): entries list[0 .. *list_n)
  if (list) n = (size_t)*list_n;
  __shared__ unsigned int wcount[kBlock / 64];
  __shared__ unsigned long long wbase;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // workgroup-uniform trip count: every thread reaches the barriers of the row hand-out
  for (size_t b0 = (size_t)blockIdx.x * blockDim.x; b0 < n; b0 += stride) {
    const size_t i0 = b0 + (threadIdx.x & ~63u);
    const bool active = i0 + lane < n;
    const size_t i = !active ? 0 : list ? (size_t)list[i0 + lane] : i0 + lane;
    const uint64_t key = active ? keys[i] : 0;
    bool inserted = false, bad = false, in_base = false;
    uint32_t base_row = 0;
    uint64_t pos = T.cap;
    if (active) {
      if (key == xf::kEmptyKey) {  // reserved value lives at the spare position
        inserted = atomicExch(&T.stat->spare_used, 1u) == 0u;
        if (inserted) T.keys[T.cap] = key;
      } else if (!xf::owns(T, key)) {
        atomicOr(&T.stat->err, xf::kErrForeignKey);
        bad = true;
      } else {
        bool done = false;
        if (T.nbase) {  // settled tier: one directory word pair, then a short dense run
          const uint64_t bk = xf::bucket_of(T, key);
          uint32_t s = T.bdir[bk];
          const uint32_t e = T.bdir[bk + 1];
          for (; s < e && !done; s += xf::kBaseWin) {
            uint64_t c[xf::kBaseWin];
#pragma unroll
            for (int t = 0; t < xf::kBaseWin; ++t) c[t] = T.bkeys[s + t];
#pragma unroll
            for (int t = 0; t < xf::kBaseWin; ++t)
              if (s + t < e && c[t] == key) {
                base_row = s + t;
                done = true;
              }
          }
          in_base = done;
        }
        uint64_t p = xf::home_of(T, key);
        for (uint64_t probes = 0; probes < T.cap && !done; probes += kWin) {
          uint64_t idx[kWin], cur[kWin];
#pragma unroll
          for (int t = 0; t < kWin; ++t) {
            idx[t] = p + t;
            if (idx[t] >= T.cap) idx[t] -= T.cap;
          }
#pragma unroll
          for (int t = 0; t < kWin; ++t) cur[t] = T.keys[idx[t]];
#pragma unroll
          for (int t = 0; t < kWin; ++t) {
            if (done) break;
            uint64_t c = cur[t];
            if (c == xf::kEmptyKey) {
              // claim; the atomic is served at the coherent point, so a stale EMPTY read
              // (another XCD inserted meanwhile) is corrected by the returned value
              c = atomicCAS((unsigned long long *)&T.keys[idx[t]], xf::kEmptyKey, key);
              if (c == xf::kEmptyKey) {
                inserted = true;
                c = key;
              }
            }
            if (c == key) {
              pos = idx[t];
              done = true;
            }
          }
          p = idx[kWin - 1] + 1;
          if (p >= T.cap) p -= T.cap;
        }
        if (!done) {
          atomicOr(&T.stat->err, xf::kErrFull);
          bad = true;
        }
      }
    }
    // hand out state rows: consecutive rows for the workgroup's new keys, in key order.  One
    // atomicAdd per workgroup: same-address atomics serialise (~10 ns each), and one per
    // wavefront made a batch of 6e6 first-touch keys a 1.25 ms launch.
    uint32_t row = (uint32_t)T.max_rows;  // write-off row
    const unsigned long long m = __ballot(inserted);
    if (lane == 0) wcount[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned total = 0;
      for (int w = 0; w < kBlock / 64; ++w) total += wcount[w];
      wbase = total ? atomicAdd(&T.stat->count, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    if (m) {
      unsigned long long base = wbase;
      for (unsigned w = 0; w < wave; ++w) base += wcount[w];
      if (inserted) {
        const unsigned long long r = base + __popcll(m & ((1ull << lane) - 1ull));
        if (r < T.max_rows) {
          row = (uint32_t)r;
          if (T.init_kind != XF_INIT_ZERO) {  // memory is pre-zeroed for XF_INIT_ZERO
            float *dst = T.w + (size_t)row * T.dim;
            for (int j = 0; j < T.dim; ++j)
              dst[j] = T.init_kind == XF_INIT_CONST ? T.init_const
                                                    : xf::hashnorm(T.seed, key, (uint32_t)j);
          }
        } else {
          atomicOr(&T.stat->err, xf::kErrFull);
        }
        // publish the row (agent scope: lanes of other CUs that meet this key in the same
        // launch wait for it below); an overflowed key publishes the write-off row
        __hip_atomic_store(&T.rows[pos], row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (active) {
      if (in_base) {
        row = base_row;  // rank in the settled tier == state row
      } else if (!inserted && !bad) {
        row = T.rows[pos];
        // The key may have been inserted by another lane of THIS launch (the same key sent by
        // several workers): its row is published right after the insert; the inserter never
        // waits for anybody, so this bounded wait cannot deadlock.
        for (int spin = 0; row == xf::kNoRow && spin < (1 << 22); ++spin) {
          __builtin_amdgcn_s_sleep(1);
          row = __hip_atomic_load(&T.rows[pos], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (row == xf::kNoRow) {
          atomicOr(&T.stat->err, xf::kErrDupKey);
          row = (uint32_t)T.max_rows;
        }
      }
      const size_t o = out_idx ? (size_t)out_idx[i] : i;
      rows_out[o] = row;
      if (GATHER) wu[o] = T.w[row];
    }
    __syncthreads();  // wcount / wbase are reused by the next pass
  }
}

// The steady-state Pull: every key of the list is looked up in the settled tier, kIlp keys per
// lane so that kIlp independent chains (key -> directory pair -> dense key run -> weight) are
// in flight per lane.  Read-only on the table.  Keys the
// tier does not hold (new since the last defrag, the reserved key value, foreign keys) go to
// a work list that k_resolve finishes with the general insert-on-miss path.
template <bool GATHER, int ILP, int WIN>
__global__ void __launch_bounds__(kBlock)
k_pull_settled(xf::TableDev T, const uint64_t *__restrict__ keys, size_t n,
               uint32_t *__restrict__ rows_out, float *__restrict__ wu,
               uint32_t *__restrict__ miss, unsigned long long *__restrict__ miss_n,
               const uint32_t *__restrict__ out_idx) {
  const size_t chunk = (size_t)kBlock * ILP;
  for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
    uint64_t key[ILP];
    uint32_t s[ILP], e[ILP], row[ILP];
    bool act[ILP], hit[ILP];
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      const size_t i = base + (size_t)q * kBlock + threadIdx.x;
      act[q] = i < n;
      key[q] = act[q] ? keys[i] : 0;
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      hit[q] = false;
      const bool look = act[q] && key[q] != xf::kEmptyKey && xf::owns(T, key[q]);
      const uint64_t bk = look ? xf::bucket_of(T, key[q]) : 0;
      s[q] = look ? T.bdir[bk] : 0u;
      e[q] = look ? T.bdir[bk + 1] : 0u;
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      uint64_t c[WIN];
#pragma unroll
      for (int t = 0; t < WIN; ++t) c[t] = T.bkeys[s[q] + t];  // padded: always readable
#pragma unroll
      for (int t = 0; t < WIN; ++t)
        if (s[q] + t < e[q] && c[t] == key[q]) {
          row[q] = s[q] + t;
          hit[q] = true;
        }
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q) {  // buckets longer than one window (rare)
      for (uint32_t p = s[q] + WIN; p < e[q] && !hit[q]; ++p)
        if (T.bkeys[p] == key[q]) {
          row[q] = p;
          hit[q] = true;
        }
    }
    float wv[ILP];
#pragma unroll
    for (int q = 0; q < ILP; ++q)
      if (GATHER && hit[q]) wv[q] = T.w[row[q]];
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      const size_t i = base + (size_t)q * kBlock + threadIdx.x;
      if (hit[q]) {
        const size_t o = out_idx ? (size_t)out_idx[i] : i;
        rows_out[o] = row[q];
        if (GATHER) wu[o] = wv[q];
      }
      // (all lanes reach this: the loop bounds are workgroup-uniform)
      const bool lost = act[q] && !hit[q];
      const unsigned long long p = wave_append(miss_n, lost);
      if (lost) miss[p] = (uint32_t)i;
    }
  }
}

// The settled-tier lookup for key lists in NO order (every raw key of a minibatch, duplicates
// included): coarse directory (L2-resident) -> the bucket's key run [s, e) -> the position the
// key would have if the bucket's keys were evenly spaced -> kBaseWin keys around it, then the
// rest of the bucket if need be.  Misses go to the work list of the insert kernel.
// two neighbouring words as ONE load: a divergent load costs ~2 TA cycles per lane whatever its
// width, and the lookup is TA-bound (six narrow loads per key: 338 us for 1e7 keys)
struct __attribute__((packed, aligned(4))) U32x2 {
  uint32_t a, b;
};
struct __attribute__((packed, aligned(8))) U64x2 {
  uint64_t a, b;
};

template <int ILP>
__global__ void __launch_bounds__(kBlock)
k_lookup_any(xf::TableDev T, const uint64_t *__restrict__ keys, size_t n,
             uint32_t *__restrict__ rows_out, uint32_t *__restrict__ miss,
             unsigned long long *__restrict__ miss_n) {
  const size_t chunk = (size_t)kBlock * ILP;
  for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
    uint64_t key[ILP];
    uint32_t s[ILP], e[ILP], est[ILP], row[ILP];
    bool act[ILP], hit[ILP];
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      const size_t i = base + (size_t)q * kBlock + threadIdx.x;
      act[q] = i < n;
      key[q] = act[q] ? keys[i] : 0;
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      hit[q] = false;
      const bool look = act[q] && key[q] != xf::kEmptyKey && xf::owns(T, key[q]);
      uint64_t bk = look ? __umul64hi(key[q] - T.lo, T.cmult) : 0;
      if (bk >= T.ncdir) bk = T.ncdir - 1;
      const U32x2 se = look ? *reinterpret_cast<const U32x2 *>(T.cdir + bk) : U32x2{0u, 0u};
      s[q] = se.a;
      e[q] = se.b;
      // fraction of the bucket's key range below the key: the low 64 bits of (key-lo)*cmult
      const uint64_t frac = (key[q] - T.lo) * T.cmult;
      const uint32_t len = e[q] - s[q];
      uint32_t p = s[q] + (uint32_t)__umul64hi(frac, (uint64_t)len);
      p = p > s[q] + 1 ? p - 2 : s[q];  // the window [p, p + kBaseWin) around the estimate
      est[q] = p;
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      static_assert(xf::kBaseWin == 4, "the window is read as two 16-byte loads");
      const U64x2 c01 = *reinterpret_cast<const U64x2 *>(T.bkeys + est[q]);  // padded: readable
      const U64x2 c23 = *reinterpret_cast<const U64x2 *>(T.bkeys + est[q] + 2);
      const uint64_t c[xf::kBaseWin] = {c01.a, c01.b, c23.a, c23.b};
#pragma unroll
      for (int t = 0; t < xf::kBaseWin; ++t)
        if (est[q] + t < e[q] && c[t] == key[q]) {
          row[q] = est[q] + t;
          hit[q] = true;
        }
      if (!hit[q] && e[q] > s[q]) {  // walk from the window towards the key (sorted run)
        if (c[0] > key[q]) {
          for (uint32_t p = est[q]; p > s[q] && !hit[q];) {
            --p;
            const uint64_t k2 = T.bkeys[p];
            if (k2 == key[q]) {
              row[q] = p;
              hit[q] = true;
            }
            if (k2 < key[q]) break;
          }
        } else {
          for (uint32_t p = est[q] + xf::kBaseWin; p < e[q] && !hit[q]; ++p) {
            const uint64_t k2 = T.bkeys[p];
            if (k2 == key[q]) {
              row[q] = p;
              hit[q] = true;
            }
            if (k2 > key[q]) break;
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < ILP; ++q) {
      const size_t i = base + (size_t)q * kBlock + threadIdx.x;
      if (hit[q]) rows_out[i] = row[q];
      const bool lost = act[q] && !hit[q];
      const unsigned long long p = wave_append(miss_n, lost);  // every lane reaches this
      if (lost) miss[p] = (uint32_t)i;
    }
  }
}

// Pull payload: vals[i][j] = w[row[i]][j] (ftrl.h:75-77).  One element per lane; rows are
// contiguous so a wave reads 64/dim rows as whole segments.
__global__ void __launch_bounds__(kBlock)
k_gather(const float *__restrict__ w, int dim, const uint32_t *__restrict__ rows, size_t n,
         float *__restrict__ vals) {
  const size_t total = n * (size_t)dim;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const size_t i = dim == 1 ? e : e / (size_t)dim;
    const size_t j = e - i * (size_t)dim;
    vals[e] = w[(size_t)rows[i] * dim + j];
  }
}

// 16-byte variant for dim % 4 == 0 (FM factor rows): four coordinates per lane
__global__ void __launch_bounds__(kBlock)
k_gather4(const float4 *__restrict__ w, int dim4, const uint32_t *__restrict__ rows, size_t n,
          float4 *__restrict__ vals) {
  const size_t total = n * (size_t)dim4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride * 2) {
    const size_t e1 = e + stride;
    const size_t i0 = e / (size_t)dim4, j0 = e - i0 * (size_t)dim4;
    const bool two = e1 < total;
    const size_t i1 = two ? e1 / (size_t)dim4 : 0, j1 = two ? e1 - i1 * (size_t)dim4 : 0;
    const float4 a = w[(size_t)rows[i0] * dim4 + j0];
    const float4 b = two ? w[(size_t)rows[i1] * dim4 + j1] : float4{};
    vals[e] = a;
    if (two) vals[e1] = b;
  }
}

// Push: one optimizer step per (row, j).  FTRL: ftrl.h:59-74 / :126-141.  SGD: sgd.h:52,96.
template <int OPT>
__global__ void __launch_bounds__(kBlock)
k_update(xf::TableDev T, const uint32_t *__restrict__ rows, size_t n,
         const float *__restrict__ grads) {
  const size_t total = n * (size_t)T.dim;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const size_t i = T.dim == 1 ? e : e / (size_t)T.dim;
    const size_t j = e - i * (size_t)T.dim;
    const size_t o = (size_t)rows[i] * T.dim + j;
    const float g = grads[e];
    if (OPT == XF_OPT_FTRL) {
      float w, nn, z;
      xf::load_nz(T, o, nn, z);
      // (the old weight: derived from the row's n and z where the table vouches for it — one
      // sector less per key, TableDev::w_of_nz)
      w = T.w_of_nz ? xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, nn, z)
                    : T.w[o];
      xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g, w, nn, z);
      T.w[o] = w;
      xf::store_nz(T, o, nn, z);
    } else {
      T.w[o] = xf::sgd_step(T.lr, g, T.w[o]);
    }
  }
}

// Push of several workers' gradients in ONE pass over the owner's state.  The entries of all
// source ranks are visited in key order (keys_sorted = the concatenated per-source key lists,
// stably sorted; order[i] = position of sorted entry i in the per-source layout that rows[] and
// grads[] use), so a key's entries are adjacent, in source-rank order.  The first of them
// applies all of that key's optimizer steps one after the other — exactly the rank-ordered
// sequence of per-source passes — with the state row read and written once.  A pass per source
// instead sweeps (nearly) every cache line of the state once per source: each source's list
// touches a tenth of the keys, but sixteen 8-byte accumulators share a line.
template <int OPT>
__global__ void __launch_bounds__(kBlock)
k_update_merged(xf::TableDev T, const uint64_t *__restrict__ keys_sorted,
                const uint32_t *__restrict__ order, size_t n,
                const uint32_t *__restrict__ rows, const float *__restrict__ grads) {
  const size_t total = n * (size_t)T.dim;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const size_t i = T.dim == 1 ? e : e / (size_t)T.dim;
    const size_t j = e - i * (size_t)T.dim;
    const uint64_t key = keys_sorted[i];
    if (i > 0 && keys_sorted[i - 1] == key) continue;  // not the first entry of its key
    const size_t o = (size_t)rows[order[i]] * T.dim + j;
    if (OPT == XF_OPT_FTRL) {
      float w, nn, z;
      xf::load_nz(T, o, nn, z);
      // (the old weight: derived from the row's n and z where the table vouches for it — one
      // sector less per key, TableDev::w_of_nz)
      w = T.w_of_nz ? xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, nn, z)
                    : T.w[o];
      for (size_t s = i; s < n && keys_sorted[s] == key; ++s)
        xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2,
                      grads[(size_t)order[s] * T.dim + j], w, nn, z);
      T.w[o] = w;
      xf::store_nz(T, o, nn, z);
    } else {
      float w = T.w[o];
      for (size_t s = i; s < n && keys_sorted[s] == key; ++s)
        w = xf::sgd_step(T.lr, grads[(size_t)order[s] * T.dim + j], w);
      T.w[o] = w;
    }
  }
}

// The merged walk with its static part precomputed (k_head_rows, once per row numbering): entry
// i of the key-sorted order carries the state row of its key when it is the FIRST entry of the
// key and kNotHead otherwise.  Per entry the update then reads 4 B of that, 4 B of the order
// and the gradient instead of two 8-byte keys, the order, the row through the order and the
// gradient: 32 instead of 44 bytes per entry with the FTRL state.
constexpr uint32_t kNotHead = 0xFFFFFFFFu;

__global__ void __launch_bounds__(kBlock)
k_head_rows(const uint64_t *__restrict__ keys_sorted, const uint32_t *__restrict__ order,
            const uint32_t *__restrict__ rows, size_t n, uint32_t *__restrict__ hrow) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    hrow[i] = (i > 0 && keys_sorted[i - 1] == keys_sorted[i]) ? kNotHead : rows[order[i]];
}

template <int OPT>
__global__ void __launch_bounds__(kBlock)
k_update_heads(xf::TableDev T, const uint32_t *__restrict__ hrow,
               const uint32_t *__restrict__ order, size_t n, const float *__restrict__ grads) {
  const size_t total = n * (size_t)T.dim;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const size_t i = T.dim == 1 ? e : e / (size_t)T.dim;
    const size_t j = e - i * (size_t)T.dim;
    const uint32_t row = hrow[i];
    if (row == kNotHead) continue;  // a later push of a key whose first entry does the walk
    const size_t o = (size_t)row * T.dim + j;
    if (OPT == XF_OPT_FTRL) {
      float w, nn, z;
      xf::load_nz(T, o, nn, z);
      // (the old weight: derived from the row's n and z where the table vouches for it — one
      // sector less per key, TableDev::w_of_nz)
      w = T.w_of_nz ? xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, nn, z)
                    : T.w[o];
      size_t s = i;
      do {
        xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2,
                      grads[(size_t)order[s] * T.dim + j], w, nn, z);
        ++s;
      } while (s < n && hrow[s] == kNotHead);
      T.w[o] = w;
      xf::store_nz(T, o, nn, z);
    } else {
      float w = T.w[o];
      size_t s = i;
      do {
        w = xf::sgd_step(T.lr, grads[(size_t)order[s] * T.dim + j], w);
        ++s;
      } while (s < n && hrow[s] == kNotHead);
      T.w[o] = w;
    }
  }
}

// export / import of one FTRL accumulator (comp 0 = n, 1 = z)
__global__ void __launch_bounds__(kBlock)
k_gather_nz(const float2 *__restrict__ nz, int comp, int dim, const uint32_t *__restrict__ rows,
            size_t n, float *__restrict__ vals) {
  const size_t total = n * (size_t)dim;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const size_t i = e / (size_t)dim, j = e - i * (size_t)dim;
    const float2 s = nz[(size_t)rows[i] * dim + j];
    vals[e] = comp ? s.y : s.x;
  }
}

__global__ void __launch_bounds__(kBlock)
k_scatter_nz(float2 *__restrict__ nz, int comp, int dim, const uint32_t *__restrict__ rows,
             size_t n, const float *__restrict__ src) {
  const size_t total = n * (size_t)dim;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const size_t i = e / (size_t)dim, j = e - i * (size_t)dim;
    float *cell = reinterpret_cast<float *>(&nz[(size_t)rows[i] * dim + j]);
    cell[comp] = src[e];
  }
}

// import: scatter host rows into their state rows
__global__ void __launch_bounds__(kBlock)
k_scatter_rows(float *__restrict__ dst, int dim, const uint32_t *__restrict__ rows,
               size_t n, const float *__restrict__ src) {
  const size_t total = n * (size_t)dim;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const size_t i = e / (size_t)dim, j = e - i * (size_t)dim;
    dst[(size_t)rows[i] * dim + j] = src[e];
  }
}

// export: append every key of the index (and the spare) with its row, after the settled
// tier's entries (arbitrary order; the host sorts by key)
__global__ void __launch_bounds__(kBlock)
k_list_occupied(xf::TableDev T, uint64_t *__restrict__ out_keys,
                uint32_t *__restrict__ out_rows, unsigned long long *__restrict__ counter,
                size_t out_cap) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t s0 = (size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); s0 <= T.cap;
       s0 += stride) {  // wave-uniform trip count
    const size_t s = s0 + (threadIdx.x & 63u);
    const uint64_t key = s <= T.cap ? T.keys[s] : xf::kEmptyKey;
    const bool occ = s < T.cap ? key != xf::kEmptyKey : s == T.cap && T.stat->spare_used != 0u;
    const unsigned long long p = T.nbase + wave_append(counter, occ);
    if (occ && p < out_cap) {
      out_keys[p] = key;
      out_rows[p] = T.rows[s];
    }
  }
}

__global__ void k_fill_u64(uint64_t *p, size_t n, uint64_t v) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// grow the key index: re-insert every stored key of `O` into the (empty, larger) index `T`;
// state rows stay where they are
__global__ void __launch_bounds__(kBlock)
k_rehash(xf::TableDev O, xf::TableDev T) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s <= O.cap; s += stride) {
    const uint64_t key = O.keys[s];
    uint64_t dst;
    if (s == O.cap) {
      if (O.stat->spare_used == 0u) continue;
      dst = T.cap;
      T.keys[T.cap] = key;
    } else {
      if (key == xf::kEmptyKey) continue;
      uint64_t pos = xf::home_of(T, key);
      while (atomicCAS((unsigned long long *)&T.keys[pos], xf::kEmptyKey, key) !=
             xf::kEmptyKey)
        if (++pos == T.cap) pos = 0;
      dst = pos;
    }
    T.rows[dst] = O.rows[s];
  }
}

// ---- defrag: every stored key into the settled tier ----------------------------------------
// (key, current row) of the keys held by the open-addressing index, appended in any order
// (the sort that follows fixes the order); the spare position is handled by the host code
__global__ void __launch_bounds__(kBlock)
k_list_index(xf::TableDev T, uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_rows,
             unsigned long long *__restrict__ counter, size_t out_cap) {
  // one atomicAdd per workgroup and pass: same-address atomics serialise at ~10 ns each, and
  // one per wavefront still made this a 3.7 ms kernel for 2e7 index positions
  __shared__ unsigned int wcount[kBlock / 64];
  __shared__ unsigned long long wbase;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  for (size_t s0 = (size_t)blockIdx.x * blockDim.x; s0 < T.cap; s0 += stride) {  // block-uniform
    const size_t s = s0 + threadIdx.x;
    const uint64_t key = s < T.cap ? T.keys[s] : xf::kEmptyKey;
    const uint32_t row = s < T.cap ? T.rows[s] : xf::kNoRow;
    const bool occ = key != xf::kEmptyKey && row != xf::kNoRow;
    const unsigned long long m = __ballot(occ);
    if (lane == 0) wcount[wave] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned total = 0;
      for (int w = 0; w < kBlock / 64; ++w) total += wcount[w];
      wbase = total ? atomicAdd(counter, (unsigned long long)total) : 0ull;
    }
    __syncthreads();
    if (occ) {
      unsigned long long p = wbase + __popcll(m & ((1ull << lane) - 1ull));
      for (unsigned w = 0; w < wave; ++w) p += wcount[w];
      if (p < out_cap) {
        out_keys[p] = key;
        out_rows[p] = row;
      }
    }
    __syncthreads();
  }
}

// the settled tier's own (key, row == rank) pairs
__global__ void __launch_bounds__(kBlock)
k_list_base(xf::TableDev T, uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_rows) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < T.nbase; r += stride) {
    out_keys[r] = T.bkeys[r];
    out_rows[r] = (uint32_t)r;
  }
}

// new row r <- old row oldrow[r]
__global__ void __launch_bounds__(kBlock)
k_move_rows(const float *__restrict__ w, const float2 *__restrict__ nz, int dim,
            const uint32_t *__restrict__ oldrow, size_t n, size_t dst_first,
            float *__restrict__ w2, float2 *__restrict__ nz2) {
  const size_t total = n * (size_t)dim;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const size_t r = dim == 1 ? e : e / (size_t)dim;
    const size_t j = e - r * (size_t)dim;
    const size_t src = (size_t)oldrow[r] * dim + j, dst = (dst_first + r) * dim + j;
    w2[dst] = w[src];
    if (nz2) nz2[dst] = nz[src];
  }
}

// The two directories over the sorted keys, in one pass (round 6: they were a kernel each, 40 +
// 37 us per 6e6 keys): dir[b] = number of keys whose bucket is < b — work item r closes the
// buckets between key r-1's and key r's (r == n closes the tail) — and the coarse directory
// likewise, cdir[b] = first rank whose coarse bucket is >= b.
__global__ void __launch_bounds__(kBlock)
k_build_dirs(xf::TableDev T, const uint64_t *__restrict__ skeys, size_t n,
             uint32_t *__restrict__ dir, uint32_t *__restrict__ cdir) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  auto coarse = [&](uint64_t key) -> uint64_t {
    const uint64_t h = __umul64hi(key - T.lo, T.cmult);
    return h < T.ncdir ? h : T.ncdir - 1;
  };
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r <= n; r += stride) {
    const uint64_t kp = r ? skeys[r - 1] : 0ull, kr = r < n ? skeys[r] : 0ull;
    const uint64_t first = r == 0 ? 0 : xf::bucket_of(T, kp) + 1;
    const uint64_t last = r == n ? T.ndir : xf::bucket_of(T, kr);
    for (uint64_t b = first; b <= last; ++b) dir[b] = (uint32_t)r;
    const uint64_t cfirst = r == 0 ? 0 : coarse(kp) + 1;
    const uint64_t clast = r == n ? T.ncdir : coarse(kr);
    for (uint64_t b = cfirst; b <= clast; ++b) cdir[b] = (uint32_t)r;  // first rank with bucket >= b
  }
}

// ---- defrag without a library sort (round 6) ---------------------------------------------
// The arrival index is ORDER-PRESERVING: a key sits at or after its home = floor((key - lo) *
// cap / span), inside the same CLUSTER (a maximal run of occupied positions) as its home.  So
// every key of a cluster is smaller than every key of the next one, and the index read front to
// back is sorted up to the order inside the clusters (a handful of keys each at load 0.6).
//   k_df_count  per block of kDfBlock positions: the keys of the clusters that START in it (a
//               cluster that began in the block before is that block's, to its end)
//   k_df_scan   where every block's keys go
//   k_df_list   a block's keys written out in position order with their rows and the cluster
//               starts; k_df_fix: every cluster sorted in place by one thread
//   k_df_merge  the sorted new keys merged with the settled tier's (merge path: one diagonal
//               search per workgroup, the rest in LDS)
// instead of a 64-bit radix sort of every key the table holds (8 passes over 12 bytes per key).
// The one cluster that may wrap around the end of the index holds keys of both ends: an entry
// whose home lies AFTER its position has wrapped (it belongs to the end), the others to block 0.
// A cluster of more than kDfCluster keys (keys that are not hashes: one home for all) sets a
// flag and the radix sort runs instead.
constexpr uint32_t kDfBlock = 16384;  // index positions per block (a workgroup: 16 per thread)
constexpr uint32_t kDfMax = 8192;    // (4 x this: how far a block follows its last cluster)
constexpr int kDfThreads = 1024;

struct DfBlock {
  uint32_t lead;  // positions at the block's start that belong to the cluster before it
  uint32_t ext;   // positions past the block's end that its last cluster runs on for
};

// lead / ext of block b (every thread of the workgroup gets them); occ[]: LDS scratch of
// kDfThreads / 64 words
__device__ __forceinline__ DfBlock df_extent(const xf::TableDev &T, uint32_t b, uint32_t *sh) {
  const uint32_t tid = threadIdx.x;
  const uint64_t p0 = (uint64_t)b * kDfBlock, p1 = min(p0 + kDfBlock, T.cap);
  const uint32_t len = (uint32_t)(p1 - p0);
  if (tid == 0) {
    sh[0] = len;  // first empty position of the block (relative)
    sh[1] = 0;
  }
  __syncthreads();
  uint32_t first_empty = len;
  for (uint32_t i = tid; i < len; i += kDfThreads)
    if (T.keys[p0 + i] == xf::kEmptyKey) first_empty = min(first_empty, i);
  if (first_empty < len) atomicMin(&sh[0], first_empty);
  __syncthreads();
  first_empty = sh[0];
  const bool pred_occ = T.keys[p0 ? p0 - 1 : T.cap - 1] != xf::kEmptyKey;
  DfBlock r;
  r.lead = pred_occ ? first_empty : 0u;
  r.ext = 0;
  // the block's last position is occupied and belongs to a cluster the block owns (or, block 0:
  // to the wrapped cluster, whose far end then is the last block's): how far does it run on?
  if (r.lead < len && T.keys[p1 - 1] != xf::kEmptyKey) {
    if (tid < 64) {  // one wavefront, 64 positions at a time
      uint32_t ext = 0;
      for (;;) {  // wave-uniform
        uint64_t q = p1 + ext + tid;
        if (q >= T.cap) q -= T.cap;
        const unsigned long long m = __ballot(T.keys[q] == xf::kEmptyKey);
        if (m) {
          ext += (uint32_t)__ffsll((long long)m) - 1;
          break;
        }
        ext += 64;
        if (ext > 4 * kDfMax) break;  // (too long: the caller's count trips the flag)
      }
      if (tid == 0) sh[1] = ext;
    }
  }
  __syncthreads();
  r.ext = sh[1];
  return r;
}

// is the entry at position p (key k) one of the block's?  In the block's own range: everything
// past `lead` — and, block 0 only, the entries of the leading run that have NOT wrapped; in the
// extension: everything, except that past the end of the index only the wrapped entries count
__device__ __forceinline__ bool df_owned(const xf::TableDev &T, uint32_t b, const DfBlock &e,
                                         uint32_t rel /* position - p0 */, uint64_t key,
                                         uint64_t *pos_out) {
  const uint64_t p0 = (uint64_t)b * kDfBlock;
  uint64_t p = p0 + rel;
  const bool past_end = p >= T.cap;
  if (past_end) p -= T.cap;
  *pos_out = p;
  if (key == xf::kEmptyKey) return false;
  if (past_end) return xf::home_of(T, key) > p;                      // wrapped: the end's
  if (b == 0 && rel < e.lead) return xf::home_of(T, key) <= p;       // not wrapped: block 0's
  return rel >= e.lead;
}

__global__ void __launch_bounds__(kDfThreads)
k_df_count(xf::TableDev T, uint32_t nblk, uint32_t *__restrict__ cnt, unsigned int *__restrict__ flag) {
  __shared__ uint32_t sh[2];
  __shared__ uint32_t total;
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  const uint64_t p0 = (uint64_t)b * kDfBlock, p1 = min(p0 + kDfBlock, T.cap);
  const DfBlock e = df_extent(T, b, sh);
  if (tid == 0) total = 0;
  __syncthreads();
  const uint32_t span = (uint32_t)(p1 - p0) + min(e.ext, 4 * kDfMax);
  uint32_t mine = 0;
  for (uint32_t i = tid; i < span; i += kDfThreads) {
    uint64_t p = p0 + i;
    if (p >= T.cap) p -= T.cap;
    uint64_t pp;
    mine += df_owned(T, b, e, i, T.keys[p], &pp) ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if ((tid & 63u) == 0 && mine) atomicAdd(&total, mine);
  __syncthreads();
  if (tid == 0) {
    cnt[b] = total;
    if (e.ext > 4 * kDfMax) atomicOr(flag, 1u);  // (a cluster of tens of thousands of keys)
  }
}

// cnt[0 .. n) -> its exclusive scan in place, the total in cnt[n]
__global__ void __launch_bounds__(kDfThreads)
k_df_scan(uint32_t *__restrict__ cnt, uint32_t n) {
  __shared__ uint32_t wsum[kDfThreads / 64];
  __shared__ uint32_t carry_s;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t base = 0; base < n; base += kDfThreads) {
    const uint32_t i = base + tid;
    const uint32_t x = i < n ? cnt[i] : 0u;
    uint32_t inc = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t y = __shfl_up(inc, o);
      if ((int)lane >= o) inc += y;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t pre = carry_s;
    for (uint32_t w = 0; w < wave; ++w) pre += wsum[w];
    if (i < n) cnt[i] = pre + inc - x;
    __syncthreads();
    if (tid == kDfThreads - 1) carry_s = pre + inc;
    __syncthreads();
  }
  if (tid == 0) cnt[n] = carry_s;
}

// a block's keys in POSITION order (ordered compaction: ballots + a scan over the wavefronts),
// with their rows, and for every key whether it begins a cluster — the block's first key does
__global__ void __launch_bounds__(kDfThreads)
k_df_list(xf::TableDev T, const uint32_t *__restrict__ base, uint64_t *__restrict__ out_keys,
          uint32_t *__restrict__ out_rows, uint8_t *__restrict__ start) {
  __shared__ uint32_t sh[2];
  __shared__ uint32_t wsum[kDfThreads / 64];
  const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t at = base[b], n = base[b + 1] - at;
  if (n == 0) return;  // (workgroup-uniform)
  const uint64_t p0 = (uint64_t)b * kDfBlock, p1 = min(p0 + kDfBlock, T.cap);
  const DfBlock e = df_extent(T, b, sh);
  const uint32_t span = (uint32_t)(p1 - p0) + e.ext;
  uint32_t run = 0;  // keys written so far (the same number in every thread)
  for (uint32_t i0 = 0; i0 < span; i0 += kDfThreads) {  // workgroup-uniform
    const uint32_t i = i0 + tid;
    uint64_t key = xf::kEmptyKey, p = 0;
    bool own = false, pred_empty = false;
    if (i < span) {
      uint64_t q = p0 + i;
      if (q >= T.cap) q -= T.cap;
      key = T.keys[q];
      own = df_owned(T, b, e, i, key, &p);
      if (own) pred_empty = T.keys[q ? q - 1 : T.cap - 1] == xf::kEmptyKey;
    }
    const unsigned long long m = __ballot(own);
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < kDfThreads / 64; ++w) {
      if (w < wave) before += wsum[w];
      total += wsum[w];
    }
    if (own) {
      const uint32_t slot = run + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      out_keys[at + slot] = key;
      out_rows[at + slot] = T.rows[p];
      start[at + slot] = (slot == 0 || pred_empty) ? 1 : 0;
    }
    run += total;
    __syncthreads();  // (wsum is rewritten by the next pass)
  }
}

// every cluster sorted by the thread of its first key (an insertion sort in memory: a cluster is
// a handful of keys at load 0.6, and already in order where no key was displaced); a cluster of
// more than kDfCluster keys sets the flag (keys that are not hashes) and the radix sort runs
constexpr uint32_t kDfCluster = 1024;
__global__ void __launch_bounds__(256)
k_df_fix(uint64_t *__restrict__ keys, uint32_t *__restrict__ rows,
         const uint8_t *__restrict__ start, size_t n, unsigned int *__restrict__ flag) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !start[i]) return;
  size_t j = i + 1;
  while (j < n && !start[j] && j - i <= kDfCluster) ++j;
  if (j - i > kDfCluster) {
    atomicOr(flag, 1u);
    return;
  }
  for (size_t a = i + 1; a < j; ++a) {
    const uint64_t k = keys[a];
    const uint32_t r = rows[a];
    size_t c = a;
    while (c > i && keys[c - 1] > k) {
      keys[c] = keys[c - 1];
      rows[c] = rows[c - 1];
      --c;
    }
    if (c != a) {
      keys[c] = k;
      rows[c] = r;
    }
  }
}

// merge A = the settled tier (keys ascending, row = rank) with B = the sorted new keys
constexpr uint32_t kMgTile = 4096;
constexpr int kMgThreads = 512;
__device__ __forceinline__ uint64_t merge_split(const uint64_t *__restrict__ A, uint64_t nA,
                                                const uint64_t *__restrict__ B, uint64_t nB,
                                                uint64_t d) {
  uint64_t lo = d > nB ? d - nB : 0, hi = d < nA ? d : nA;  // elements of A among the first d
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (A[mid] < B[d - 1 - mid]) lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}
__global__ void __launch_bounds__(kMgThreads)
k_df_merge(const uint64_t *__restrict__ A, uint64_t nA, const uint64_t *__restrict__ B,
           const uint32_t *__restrict__ Brow, uint64_t nB, uint64_t *__restrict__ out_keys,
           uint32_t *__restrict__ out_rows) {
  __shared__ uint64_t la[kMgTile], lb[kMgTile];
  __shared__ uint64_t cut[2];
  const uint32_t tid = threadIdx.x;
  const uint64_t n = nA + nB, d0 = (uint64_t)blockIdx.x * kMgTile, d1 = min(d0 + kMgTile, n);
  if (tid < 2) cut[tid] = merge_split(A, nA, B, nB, tid ? d1 : d0);
  __syncthreads();
  const uint64_t i0 = cut[0], i1 = cut[1], j0 = d0 - i0, j1 = d1 - i1;
  const uint32_t na = (uint32_t)(i1 - i0), nb = (uint32_t)(j1 - j0);
  for (uint32_t k = tid; k < na; k += kMgThreads) la[k] = A[i0 + k];
  for (uint32_t k = tid; k < nb; k += kMgThreads) lb[k] = B[j0 + k];
  __syncthreads();
  auto lower = [](const uint64_t *x, uint32_t m, uint64_t key) -> uint32_t {  // x[i] < key
    uint32_t lo = 0, hi = m;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (x[mid] < key) lo = mid + 1;
      else
        hi = mid;
    }
    return lo;
  };
  for (uint32_t k = tid; k < na; k += kMgThreads) {
    const uint64_t o = d0 + k + lower(lb, nb, la[k]);
    out_keys[o] = la[k];
    out_rows[o] = (uint32_t)(i0 + k);
  }
  for (uint32_t k = tid; k < nb; k += kMgThreads) {
    const uint64_t o = d0 + k + lower(la, na, lb[k]);
    out_keys[o] = lb[k];
    out_rows[o] = Brow[j0 + k];
  }
}

// table_take_early / table_put_early: the few keys the host API put into the arrival index,
// taken out (state rows into a buffer, the rows zero again, the positions empty) and put back
// where the resolve finds or puts them
__global__ void __launch_bounds__(kBlock)
k_early_find(xf::TableDev T, const uint64_t *__restrict__ keys, size_t n,
             unsigned long long *__restrict__ pos, float *__restrict__ tw,
             float2 *__restrict__ tnz) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const uint64_t key = keys[j];
  uint64_t p = xf::home_of(T, key);
  bool found = false;
  for (uint64_t probes = 0; probes < T.cap; ++probes) {
    const uint64_t c = T.keys[p];
    if (c == key) {
      found = true;
      break;
    }
    if (c == xf::kEmptyKey) break;
    if (++p >= T.cap) p -= T.cap;
  }
  if (!found) {
    atomicOr(&T.stat->err, xf::kErrDupKey);
    pos[j] = ~0ull;
    return;
  }
  pos[j] = p;
  const size_t row = T.rows[p];
  for (int d = 0; d < T.dim; ++d) {
    tw[j * T.dim + d] = T.w[row * T.dim + d];
    if (T.nz) tnz[j * T.dim + d] = T.nz[row * T.dim + d];
  }
}
__global__ void __launch_bounds__(kBlock)
k_early_clear(xf::TableDev T, const unsigned long long *__restrict__ pos, size_t n) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n || pos[j] == ~0ull) return;
  const size_t row = T.rows[pos[j]];
  for (int d = 0; d < T.dim; ++d) {
    T.w[row * T.dim + d] = 0.0f;
    if (T.nz) T.nz[row * T.dim + d] = make_float2(0.0f, 0.0f);
  }
  T.keys[pos[j]] = xf::kEmptyKey;
  T.rows[pos[j]] = xf::kNoRow;
}
__global__ void __launch_bounds__(kBlock)
k_early_put(xf::TableDev T, const uint32_t *__restrict__ rows, size_t n,
            const float *__restrict__ tw, const float2 *__restrict__ tnz) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const size_t row = rows[j];
  for (int d = 0; d < T.dim; ++d) {
    T.w[row * T.dim + d] = tw[j * T.dim + d];
    if (T.nz) T.nz[row * T.dim + d] = tnz[j * T.dim + d];
  }
}

// table_settle_first: the initial weights of the first d rows (row r belongs to keys[r]) for the
// init kinds that are not "zero" (zeroed memory), and the table's key count
__global__ void __launch_bounds__(kBlock)
k_first_rows(xf::TableDev T, const uint64_t *__restrict__ keys, size_t d) {
  const size_t stride = (size_t)gridDim.x * blockDim.x, n = d * (size_t)T.dim;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const size_t row = i / (size_t)T.dim;
    const uint32_t j = (uint32_t)(i - row * (size_t)T.dim);
    T.w[i] = T.init_kind == XF_INIT_CONST ? T.init_const : xf::hashnorm(T.seed, keys[row], j);
  }
}
__global__ void k_set_count(xf::TableStat *st, unsigned long long d) { st->count = d; }

__global__ void k_fill_u32(uint32_t *p, size_t n, uint32_t v) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

}  // namespace

// ------------------------------------------------------------------------------ table
struct xf_table {
  xf_table_config cfg;
  int dev = 0;
  xf::TableDev T{};
  // identity of the row numbering: `uid` is unique per table object, `epoch` counts the
  // renumberings (xf_table_defrag).  Row numbers cached outside the table (the cells of a
  // compiled minibatch, the owner-side rows of a sharded step) are valid for one (uid, epoch).
  uint64_t uid = 0, epoch = 0;
  // scratch for the host-pointer API
  uint64_t *s_keys = nullptr;
  uint32_t *s_rows = nullptr;
  float *s_vals = nullptr;
  size_t s_n = 0, s_vals_n = 0;
  // work list of the two-stage resolve (keys the settled tier did not hold)
  uint32_t *miss = nullptr;
  unsigned long long *miss_n = nullptr;
  size_t miss_cap = 0;
  // one device allocation derived from the settled tier by another module (xf_keybuild.hip:
  // chunk boundaries and their directories), valid for `aux_epoch`; freed with the table
  void *aux = nullptr;
  uint64_t aux_epoch = ~0ull;
  // a second derived allocation: a fixed-size record per state row, owned by xf_model.hip (the
  // FM forward's per-key (sum_k v, sum_k v^2, w) next to the factors they come from), and what
  // its validity hangs on: `rec_gen` counts (re)allocations, `writes` the weight writes by
  // anything that does not keep the records up to date
  void *rec = nullptr;
  size_t rec_rows = 0, rec_row_bytes = 0;
  uint64_t rec_gen = 0, rec_tag = 0, writes = 0;
  // "the record of EVERY settled row is current", as of the key the records' owner set it with
  // (table_records_all_*): a minibatch compiled against the settled tier then needs no pass
  // over its keys' rows before its first step
  bool rec_all_ok = false;
  uint64_t rec_all[5] = {0, 0, 0, 0, 0};
  // a row's w may differ from ftrl_w_of(n, z): rows were imported with a w that is not (checked
  // on the GPU, row by row), or the hyper-parameters changed with rows in the table.  Sticky.
  bool w_tainted = false;
  // xf_table_defrag's second state buffer (as large as the first; the two swap roles)
  float *w_alt = nullptr;
  float2 *nz_alt = nullptr;
  size_t alt_elems = 0;
  // The keys the HOST API (xf_table_pull / xf_table_push) has put into a table without a settled
  // tier, while they are few (lr_worker.cc:180-182 pushes key 0 before the first minibatch).
  // When they are ALL the table holds — the key count says so — and no row number has left the
  // table (rows_out), the build of the first minibatch may still settle the table at once:
  // table_early_keys, xf_keybuild.hip "an empty table".
  std::vector<uint64_t> early;
  bool early_over = false, rows_out = false;
  // the settled tier's keys and its two directories are ONE allocation (T.bkeys / bdir / cdir
  // point into it); tier_next: the one a build has asked for and not yet handed back
  void *tier = nullptr, *tier_next = nullptr;
  // ... and a retired one kept for the next build (a table that settles again and again — a
  // first epoch — would otherwise pay a hipMalloc and a hipFree of hundreds of MB per defrag:
  // 3 ms as a rule, 30-170 ms now and then, tools/r6/call54.sh); their sizes; tier_full: every
  // tier allocation is sized for max_rows keys (xf_table_prepare_defrag)
  void *tier_spare = nullptr;
  size_t tier_bytes = 0, tier_next_bytes = 0, tier_spare_bytes = 0;
  bool tier_full = false;
};
// the tier of n keys: where its parts lie in one allocation
struct TierLayout {
  uint64_t ndir, ncdir;
  size_t o_dir, o_cdir, bytes;
};
static TierLayout tier_layout(size_t n) {
  TierLayout L;
  L.ndir = n + 1;
  L.ncdir = std::max<uint64_t>(1, n / xf::kCoarse);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  L.o_dir = al((n + xf::kBaseWin) * sizeof(uint64_t));
  L.o_cdir = L.o_dir + al((L.ndir + 1) * sizeof(uint32_t));
  L.bytes = L.o_cdir + al((L.ncdir + 1) * sizeof(uint32_t));
  return L;
}
// the second state buffer a defrag moves the rows into (the two swap roles)
static int alt_state(xf_table *t, size_t elems) {
  if (t->w_alt && t->alt_elems == elems) return XF_OK;
  if (t->w_alt) XF_HIP(hipFree(t->w_alt));
  if (t->nz_alt) XF_HIP(hipFree(t->nz_alt));
  t->w_alt = nullptr;
  t->nz_alt = nullptr;
  t->alt_elems = 0;
  XF_HIP(hipMalloc((void **)&t->w_alt, elems * sizeof(float)));
  XF_HIP(hipMemset(t->w_alt, 0, elems * sizeof(float)));
  if (t->T.nz) {
    XF_HIP(hipMalloc((void **)&t->nz_alt, elems * sizeof(float2)));
    XF_HIP(hipMemset(t->nz_alt, 0, elems * sizeof(float2)));
  }
  t->alt_elems = elems;
  return XF_OK;
}
static int tier_alloc(xf_table *t, size_t n, uint64_t **keys) {
  const size_t need = tier_layout(n).bytes;
  if (t->tier_next && t->tier_next_bytes < need) {
    (void)hipFree(t->tier_next);
    t->tier_next = nullptr;
  }
  if (!t->tier_next && t->tier_spare && t->tier_spare_bytes >= need) {  // the retired one again
    t->tier_next = t->tier_spare;
    t->tier_next_bytes = t->tier_spare_bytes;
    t->tier_spare = nullptr;
    t->tier_spare_bytes = 0;
  }
  if (!t->tier_next) {
    const size_t bytes = t->tier_full ? std::max(need, tier_layout(t->T.max_rows).bytes) : need;
    XF_HIP(hipMalloc(&t->tier_next, bytes));
    t->tier_next_bytes = bytes;
  }
  *keys = (uint64_t *)t->tier_next;
  return XF_OK;
}
// the tier in use makes way for tier_next (the caller has waited for the device): kept as the
// spare when it is the larger of the two
static void tier_swap(xf_table *t) {
  if (t->tier) {
    if (t->tier_bytes > t->tier_spare_bytes) {
      if (t->tier_spare) (void)hipFree(t->tier_spare);
      t->tier_spare = t->tier;
      t->tier_spare_bytes = t->tier_bytes;
    } else {
      (void)hipFree(t->tier);
    }
  }
  t->tier = t->tier_next;
  t->tier_bytes = t->tier_next_bytes;
  t->tier_next = nullptr;
  t->tier_next_bytes = 0;
}
// tier_next (n keys in place) becomes the table's tier: directories built on `s`, N's tier fields
// set; the old tier is the caller's to free (after the device has finished with it)
static void tier_install(xf_table *t, xf::TableDev &N, size_t n, hipStream_t s) {
  const TierLayout L = tier_layout(n);
  char *base = (char *)t->tier_next;
  uint64_t *keys = (uint64_t *)base;
  N.nbase = n;
  N.ndir = L.ndir;
  N.dmult = (uint64_t)((((unsigned __int128)N.ndir) << 64) / N.span);
  N.ncdir = L.ncdir;
  N.cmult = (uint64_t)((((unsigned __int128)N.ncdir) << 64) / N.span);
  N.bkeys = keys;
  N.bdir = (uint32_t *)(base + L.o_dir);
  N.cdir = (uint32_t *)(base + L.o_cdir);
  hipLaunchKernelGGL(k_fill_u64, dim3(1), dim3(kBlock), 0, s, keys + n, (size_t)xf::kBaseWin,
                     xf::kEmptyKey);
  hipLaunchKernelGGL(k_build_dirs, dim3(grid_for(n + 1)), dim3(kBlock), 0, s, N, keys, n,
                     (uint32_t *)(base + L.o_dir), (uint32_t *)(base + L.o_cdir));
}
constexpr size_t kEarlyMax = 4096;
static void note_early(xf_table *t, const uint64_t *keys, size_t n) {
  if (t->T.nbase != 0 || t->early_over) return;
  if (t->early.size() + n > kEarlyMax) {
    t->early_over = true;
    t->early.clear();
    return;
  }
  t->early.insert(t->early.end(), keys, keys + n);
}

// Is (float)((double)x * (1.0 / (double)d)) the float x / d for EVERY finite x?  All 2^32 bit
// patterns on the GPU (~10 ms), once per divisor and process.  The step's two divisions by alpha
// then go without their subnormal-range guard (xf_device.h: div_by_const); a divisor that fails
// — 50000 does, at 1308 subnormal quotients — keeps it.  Any failure to run the check keeps it too.
namespace {
__global__ void __launch_bounds__(256)
k_div_exact(float d, double inv_d, unsigned int *__restrict__ bad) {
#pragma clang fp contract(off)
  unsigned int mine = 0;
  for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < (1ull << 32);
       u += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)u);
    if ((((uint32_t)u >> 23) & 0xFFu) == 0xFFu) continue;  // inf / nan
    const float q1 = x / d, q2 = (float)((double)x * inv_d);
    mine |= __float_as_uint(q1) != __float_as_uint(q2) ? 1u : 0u;
  }
  if (__any((int)mine) && (threadIdx.x & 63u) == 0) atomicOr(bad, 1u);
}
}  // namespace

static bool div_exact_for(float d) {
  static std::mutex mu;
  static std::map<uint32_t, bool> known;
  if (!(d > 0.0f) || !std::isfinite(d)) return false;
  uint32_t bits;
  memcpy(&bits, &d, 4);
  std::lock_guard<std::mutex> lk(mu);
  auto it = known.find(bits);
  if (it != known.end()) return it->second;
  bool ok = false;
  unsigned int *dbad = nullptr, hbad = 1;
  if (hipMalloc((void **)&dbad, 4) == hipSuccess) {
    if (hipMemset(dbad, 0, 4) == hipSuccess) {
      hipLaunchKernelGGL(k_div_exact, dim3(256 * 16), dim3(256), 0, nullptr, d, 1.0 / (double)d,
                         dbad);
      if (hipGetLastError() == hipSuccess &&
          hipMemcpy(&hbad, dbad, 4, hipMemcpyDeviceToHost) == hipSuccess)
        ok = hbad == 0;
    }
    (void)hipFree(dbad);
  }
  known[bits] = ok;
  return ok;
}

static void refresh_hyper(xf_table *t) {
  t->T.alpha = t->cfg.alpha;
  t->T.inv_alpha = 1.0 / (double)t->cfg.alpha;
  // (FTRL tables only: the check runs on the GPU; its verdict rides in the sign, xf_device.h)
  if (t->cfg.opt_kind == XF_OPT_FTRL && div_exact_for(t->cfg.alpha))
    t->T.inv_alpha = -t->T.inv_alpha;
  t->T.beta = t->cfg.beta;
  t->T.lambda1 = t->cfg.lambda1;
  t->T.lambda2 = t->cfg.lambda2;
  t->T.lr = t->cfg.lr;
  // (TableDev::w_of_nz; a fresh row is w = n = z = 0 = ftrl_w_of(0, 0) when lambda1 >= 0)
  t->T.w_of_nz = t->cfg.opt_kind == XF_OPT_FTRL && t->cfg.dim == 1 &&
                 t->cfg.init_kind == XF_INIT_ZERO && t->cfg.lambda1 >= 0.0f && !t->w_tainted;
}

static int ensure_scratch(xf_table *t, size_t n) {
  if (n > t->s_n) {
    if (t->s_keys) XF_HIP(hipFree(t->s_keys));
    if (t->s_rows) XF_HIP(hipFree(t->s_rows));
    size_t m = std::max<size_t>(n, 1024);
    XF_HIP(hipMalloc((void **)&t->s_keys, m * sizeof(uint64_t)));
    XF_HIP(hipMalloc((void **)&t->s_rows, m * sizeof(uint32_t)));
    t->s_n = m;
  }
  const size_t need = n * (size_t)t->cfg.dim;
  if (need > t->s_vals_n) {
    if (t->s_vals) XF_HIP(hipFree(t->s_vals));
    size_t m = std::max<size_t>(need, 1024);
    XF_HIP(hipMalloc((void **)&t->s_vals, m * sizeof(float)));
    t->s_vals_n = m;
  }
  return XF_OK;
}

static int alloc_index(xf::TableDev &T) {
  const size_t slots = (size_t)T.cap + 1;
  XF_HIP(hipMalloc((void **)&T.keys, slots * sizeof(uint64_t)));
  XF_HIP(hipMalloc((void **)&T.rows, slots * sizeof(uint32_t)));
  XF_HIP(hipMemset(T.rows, 0xFF, slots * sizeof(uint32_t)));  // kNoRow
  hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(slots)), dim3(kBlock), 0, 0, T.keys, slots,
                     xf::kEmptyKey);
  XF_HIP(hipGetLastError());
  XF_HIP(hipDeviceSynchronize());
  return XF_OK;
}

// (re)allocate the dense state for `max_rows` rows, keeping the first `keep` rows
static int alloc_state(xf::TableDev &T, uint64_t max_rows, bool ftrl, uint64_t keep) {
  const size_t elems = ((size_t)max_rows + 1) * (size_t)T.dim;
  {
    float *fresh = nullptr;
    XF_HIP(hipMalloc((void **)&fresh, elems * sizeof(float)));
    XF_HIP(hipMemset(fresh, 0, elems * sizeof(float)));
    if (T.w) {
      if (keep)
        XF_HIP(hipMemcpy(fresh, T.w, (size_t)keep * T.dim * sizeof(float),
                         hipMemcpyDeviceToDevice));
      XF_HIP(hipFree(T.w));
    }
    T.w = fresh;
  }
  if (ftrl) {
    float2 *fresh = nullptr;
    XF_HIP(hipMalloc((void **)&fresh, elems * sizeof(float2)));
    XF_HIP(hipMemset(fresh, 0, elems * sizeof(float2)));
    if (T.nz) {
      if (keep)
        XF_HIP(hipMemcpy(fresh, T.nz, (size_t)keep * T.dim * sizeof(float2),
                         hipMemcpyDeviceToDevice));
      XF_HIP(hipFree(T.nz));
    }
    T.nz = fresh;
  }
  T.max_rows = max_rows;
  return XF_OK;
}

static void set_geometry(xf::TableDev &T, uint64_t cap, uint32_t shard, uint32_t nshards) {
  const xf::ShardRange r = xf::shard_range(shard, nshards);
  T.cap = cap;
  T.lo = r.lo;
  T.span = r.span;
  T.last_shard = shard == nshards - 1;
  T.single = nshards == 1;
  // home = mulhi64(key - lo, mult), mult = floor(cap * 2^64 / span)
  T.mult = (uint64_t)((((unsigned __int128)cap) << 64) / r.span);
}

static uint64_t rows_for(uint64_t cap) { return (uint64_t)((double)cap * kMaxLoad) + 1; }

extern "C" void xf_table_config_default(xf_table_config *c) {
  c->opt_kind = XF_OPT_FTRL;
  c->dim = 1;
  c->init_kind = XF_INIT_ZERO;
  c->init_const = 0.0f;
  c->seed = 0;
  c->alpha = 5e-2f;  // ftrl.h:17-20
  c->beta = 1.0f;
  c->lambda1 = 5e-5f;
  c->lambda2 = 10.0f;
  c->lr = 0.001f;  // sgd.h:16
  c->capacity = 1u << 20;
  c->shard = 0;
  c->nshards = 1;
}

extern "C" int xf_table_create(xf_table **out, const xf_table_config *cfg) {
  XF_REQUIRE(out && cfg, "xf_table_create: null argument");
  XF_REQUIRE(cfg->dim >= 1 && cfg->dim <= 4096, "xf_table_create: dim %d", cfg->dim);
  XF_REQUIRE(cfg->opt_kind == XF_OPT_FTRL || cfg->opt_kind == XF_OPT_SGD,
             "xf_table_create: opt_kind %d", cfg->opt_kind);
  XF_REQUIRE(cfg->capacity >= 16 && cfg->capacity < 0xFFFFFFF0ull,
             "xf_table_create: capacity %llu out of range",
             (unsigned long long)cfg->capacity);
  XF_REQUIRE(cfg->nshards >= 1 && cfg->shard < cfg->nshards, "xf_table_create: shard %u/%u",
             cfg->shard, cfg->nshards);
  // (the step divides by alpha twice, ftrl.h:63,70 — and TableDev::inv_alpha carries a verdict in
  // its sign: a negative or zero alpha is refused rather than silently re-signed)
  XF_REQUIRE(cfg->opt_kind != XF_OPT_FTRL || (cfg->alpha > 0.0f && std::isfinite(cfg->alpha)),
             "xf_table_create: FTRL needs a finite alpha > 0 (got %g)", (double)cfg->alpha);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return xf::set_error(XF_ENOGPU, "xf_table_create: no HIP device (the table lives in HBM)");
  xf_table *t = new xf_table;
  t->cfg = *cfg;
  {
    static std::atomic<uint64_t> next_uid{1};
    t->uid = next_uid.fetch_add(1);
  }
  XF_HIP(hipGetDevice(&t->dev));
  xf::TableDev &T = t->T;
  T.dim = cfg->dim;
  T.init_kind = cfg->init_kind;
  T.init_const = cfg->init_const;
  T.seed = cfg->seed;
  set_geometry(T, cfg->capacity, cfg->shard, cfg->nshards);
  refresh_hyper(t);
  XF_TRY(alloc_index(T));
  XF_TRY(alloc_state(T, rows_for(cfg->capacity), cfg->opt_kind == XF_OPT_FTRL, 0));
  XF_HIP(hipMalloc((void **)&T.stat, sizeof(xf::TableStat)));
  XF_HIP(hipMemset(T.stat, 0, sizeof(xf::TableStat)));
  *out = t;
  return XF_OK;
}

extern "C" int xf_table_destroy(xf_table *t) {
  if (!t) return XF_OK;
  void *ps[] = {t->T.keys, t->T.rows, t->T.w, t->T.nz, t->T.stat, t->tier, t->tier_next, t->tier_spare, t->s_keys, t->s_rows, t->s_vals, t->miss,
                t->miss_n, t->aux, t->rec, t->w_alt, t->nz_alt};
  for (void *p : ps)
    if (p) hipFree(p);
  delete t;
  return XF_OK;
}

extern "C" int xf_table_set_hyper(xf_table *t, float alpha, float beta, float l1, float l2,
                                  float lr) {
  XF_REQUIRE(t, "xf_table_set_hyper: null table");
  XF_REQUIRE(t->cfg.opt_kind != XF_OPT_FTRL || (alpha > 0.0f && std::isfinite(alpha)),
             "xf_table_set_hyper: FTRL needs a finite alpha > 0 (got %g)", (double)alpha);
  // the rows in the table hold the w of the OLD hyper-parameters, and the next step of a key uses
  // that w (ftrl.h:63): from here on the kernels read it
  if (t->cfg.opt_kind == XF_OPT_FTRL &&
      (memcmp(&alpha, &t->cfg.alpha, 4) || memcmp(&beta, &t->cfg.beta, 4) ||
       memcmp(&l1, &t->cfg.lambda1, 4) || memcmp(&l2, &t->cfg.lambda2, 4))) {
    uint64_t nk = 0;
    XF_TRY(xf_table_size(t, &nk));
    if (nk) t->w_tainted = true;
  }
  t->cfg.alpha = alpha;
  t->cfg.beta = beta;
  t->cfg.lambda1 = l1;
  t->cfg.lambda2 = l2;
  t->cfg.lr = lr;
  refresh_hyper(t);
  return XF_OK;
}

static int read_stat(xf_table *t, xf::TableStat *st) {
  XF_HIP(hipMemcpy(st, t->T.stat, sizeof(*st), hipMemcpyDeviceToHost));
  return XF_OK;
}

extern "C" int xf_table_settled(xf_table *t, uint64_t *nkeys) {
  XF_REQUIRE(t && nkeys, "xf_table_settled: null argument");
  *nkeys = t->T.nbase;
  return XF_OK;
}

extern "C" int xf_table_size(xf_table *t, uint64_t *nkeys) {
  XF_REQUIRE(t && nkeys, "xf_table_size: null argument");
  XF_HIP(hipDeviceSynchronize());
  xf::TableStat st;
  XF_TRY(read_stat(t, &st));
  *nkeys = std::min<uint64_t>(st.count, t->T.max_rows);
  return XF_OK;
}

// What the table's maintenance step allocates, allocated now: the second state buffer and both
// tier allocations at the size the table's rows allow.  For a caller whose clock is about to
// start (the worker before its first block): a defrag then calls the driver's allocator no more
// (hundreds of MB per call: 3 ms as a rule, 30-170 ms now and then — tools/r6/call54.sh).
extern "C" int xf_table_prepare_defrag(xf_table *t) {
  XF_REQUIRE(t, "xf_table_prepare_defrag: null table");
  XF_TRY(alt_state(t, ((size_t)t->T.max_rows + 1) * (size_t)t->T.dim));
  t->tier_full = true;
  const size_t full = tier_layout(t->T.max_rows).bytes;
  if (!t->tier_next || t->tier_next_bytes < full) {
    if (t->tier_next) (void)hipFree(t->tier_next);
    t->tier_next = nullptr;
    XF_HIP(hipMalloc(&t->tier_next, full));
    t->tier_next_bytes = full;
  }
  if (!t->tier_spare || t->tier_spare_bytes < full) {
    if (t->tier_spare) (void)hipFree(t->tier_spare);
    t->tier_spare = nullptr;
    XF_HIP(hipMalloc(&t->tier_spare, full));
    t->tier_spare_bytes = full;
  }
  return XF_OK;
}

extern "C" int xf_table_capacity(xf_table *t, uint64_t *slots) {
  XF_REQUIRE(t && slots, "xf_table_capacity: null argument");
  *slots = t->T.cap;
  return XF_OK;
}

extern "C" int xf_table_check(xf_table *t, void *stream) {
  XF_REQUIRE(t, "xf_table_check: null table");
  XF_HIP(hipStreamSynchronize(S(stream)));
  xf::TableStat st;
  XF_TRY(read_stat(t, &st));
  if (st.err & xf::kErrFull)
    return xf::set_error(XF_EFULL,
                         "table full: %llu keys for %llu state rows / %llu index positions "
                         "(shard %u/%u); raise capacity or call xf_table_reserve",
                         (unsigned long long)st.count, (unsigned long long)t->T.max_rows,
                         (unsigned long long)t->T.cap, t->cfg.shard, t->cfg.nshards);
  if (st.err & xf::kErrForeignKey)
    return xf::set_error(XF_EINVAL, "a key outside shard %u/%u's range was sent to it",
                         t->cfg.shard, t->cfg.nshards);
  if (st.err & xf::kErrDupKey)
    return xf::set_error(XF_EINVAL, "a key's row was never published (internal error)");
  return XF_OK;
}

// Re-house the key index in `new_capacity` positions and extend the state (rows keep their
// numbers, so row arrays from earlier resolve calls stay valid).
extern "C" int xf_table_reserve(xf_table *t, uint64_t new_capacity) {
  XF_REQUIRE(t, "xf_table_reserve: null table");
  if (new_capacity <= t->T.cap) return XF_OK;
  XF_REQUIRE(new_capacity < 0xFFFFFFF0ull, "xf_table_reserve: capacity out of range");
  XF_HIP(hipDeviceSynchronize());
  xf::TableStat st;
  XF_TRY(read_stat(t, &st));
  if (st.err & xf::kErrFull)
    return xf::set_error(XF_EFULL, "xf_table_reserve: the table already overflowed");
  xf::TableDev N = t->T;
  set_geometry(N, new_capacity, t->cfg.shard, t->cfg.nshards);
  XF_TRY(alloc_index(N));
  hipLaunchKernelGGL(k_rehash, dim3(grid_for((size_t)t->T.cap + 1)), dim3(kBlock), 0, 0, t->T, N);
  XF_HIP(hipGetLastError());
  XF_HIP(hipDeviceSynchronize());
  XF_HIP(hipFree(t->T.keys));
  XF_HIP(hipFree(t->T.rows));
  XF_TRY(alloc_state(N, rows_for(new_capacity), t->cfg.opt_kind == XF_OPT_FTRL, st.count));
  if (t->w_alt) (void)hipFree(t->w_alt);  // (the second state buffer is of the old size)
  if (t->nz_alt) (void)hipFree(t->nz_alt);
  t->w_alt = nullptr;
  t->nz_alt = nullptr;
  t->alt_elems = 0;
  t->T = N;
  t->cfg.capacity = new_capacity;
  return XF_OK;
}

// Settle the table: every stored key moves into the settled tier — sorted, dense, the key of
// rank r owning state row r — and the open-addressing index is emptied for the keys still
// to come.  Rows are handed out in arrival order, so before this a sorted key list hops
// between runs of rows and pays 24 bytes of half-empty index per stored key to find them;
// afterwards the lookup sweeps 8 bytes per stored key plus a small directory, and the Pull's
// weight gather and the Push's state pass walk the state arrays front to back.
// Row numbers change: call it between steps, never between a resolve and its update.
extern "C" int xf_table_defrag(xf_table *t) {
  XF_REQUIRE(t, "xf_table_defrag: null table");
  XF_HIP(hipDeviceSynchronize());
  xf::TableStat st;
  XF_TRY(read_stat(t, &st));
  if (st.err) return xf_table_check(t, nullptr);
  if (st.count == 0) return XF_OK;
  xf::TableDev &T = t->T;
  const size_t spare = st.spare_used ? 1 : 0;
  const size_t n = (size_t)st.count - spare;  // ordinary keys: settled tier + index
  const size_t n_idx = n - (size_t)T.nbase;
  if (n_idx == 0) return XF_OK;                // nothing arrived since the last defrag
  const size_t elems = ((size_t)T.max_rows + 1) * (size_t)T.dim;
  // Temporaries from the builders' arena (no hipMalloc / hipFree per call: a free waits for the
  // device and a GB-sized one takes milliseconds); what outlives the call — the new tier's keys
  // and directories — is allocated, and the state moves into the table's SECOND state buffer
  // (kept between calls: the two swap roles).  Rows at or beyond the table's key count are zero
  // in both: a buffer only ever held rows below the count of its time, and the count only grows.
  xf::Scratch sc;
  uint64_t *k_all = nullptr;
  uint32_t *r_all = nullptr, *r_sorted = nullptr, *bcnt = nullptr;
  uint8_t *cstart = nullptr;
  unsigned int *dflag = nullptr;
  uint64_t *k_sorted = nullptr;  // (the new tier's allocation: keys, then the directories)
  XF_TRY(sc.get(&k_all, n));
  XF_TRY(sc.get(&r_all, n));
  XF_TRY(sc.get(&r_sorted, n));
  XF_TRY(tier_alloc(t, n, &k_sorted));
  // the index's keys in key order without a sort (kernels: "defrag without a library sort"):
  // into k_all / r_all, then merged with the settled tier's into k_sorted / r_sorted
  bool sorted_ok = false;
  const uint64_t nblk64 = (T.cap + kDfBlock - 1) / kDfBlock;
  if (nblk64 < (1u << 24) && n_idx < 0xFFFFFFFFull && xf::key_build_mode() != 1) {
    const uint32_t nblk = (uint32_t)nblk64;
    XF_TRY(sc.get(&bcnt, (size_t)nblk + 1));
    XF_TRY(sc.get(&cstart, n_idx));
    XF_TRY(sc.get(&dflag, 1));
    XF_HIP(hipMemsetAsync(dflag, 0, 4, 0));
    hipLaunchKernelGGL(k_df_count, dim3(nblk), dim3(kDfThreads), 0, 0, T, nblk, bcnt, dflag);
    hipLaunchKernelGGL(k_df_scan, dim3(1), dim3(kDfThreads), 0, 0, bcnt, nblk);
    XF_HIP(hipGetLastError());
    unsigned int hflag = 1;
    uint32_t listed32 = 0;
    XF_HIP(hipMemcpy(&hflag, dflag, 4, hipMemcpyDeviceToHost));
    XF_HIP(hipMemcpy(&listed32, bcnt + nblk, 4, hipMemcpyDeviceToHost));
    if (!hflag) {
      if (listed32 != n_idx)
        return xf::set_error(XF_EINVAL, "xf_table_defrag: index holds %u keys, %zu expected",
                             listed32, n_idx);
      uint64_t *bk = T.nbase ? k_all : k_sorted;  // (no settled tier: nothing to merge with)
      uint32_t *br = T.nbase ? r_all : r_sorted;
      hipLaunchKernelGGL(k_df_list, dim3(nblk), dim3(kDfThreads), 0, 0, T, bcnt, bk, br, cstart);
      hipLaunchKernelGGL(k_df_fix, dim3((unsigned)((n_idx + 255) / 256)), dim3(256), 0, 0, bk, br,
                         cstart, n_idx, dflag);
      XF_HIP(hipGetLastError());
      XF_HIP(hipMemcpy(&hflag, dflag, 4, hipMemcpyDeviceToHost));
      if (!hflag) {
        if (T.nbase)
          hipLaunchKernelGGL(k_df_merge, dim3((unsigned)((n + kMgTile - 1) / kMgTile)),
                             dim3(kMgThreads), 0, 0, T.bkeys, (uint64_t)T.nbase, k_all, r_all,
                             (uint64_t)n_idx, k_sorted, r_sorted);
        XF_HIP(hipGetLastError());
        sorted_ok = true;
      }
    }
  }
  if (!sorted_ok) {  // keys that are not hashes (one home for thousands of them): a radix sort
    unsigned long long *d_cnt = nullptr;
    XF_TRY(sc.get(&d_cnt, 1));
    XF_HIP(hipMemsetAsync(d_cnt, 0, 8, 0));
    if (T.nbase)
      hipLaunchKernelGGL(k_list_base, dim3(grid_for(T.nbase)), dim3(kBlock), 0, 0, T, k_all, r_all);
    hipLaunchKernelGGL(k_list_index, dim3(grid_for(T.cap)), dim3(kBlock), 0, 0, T,
                       k_all + T.nbase, r_all + T.nbase, d_cnt, n_idx);
    XF_HIP(hipGetLastError());
    unsigned long long listed = 0;
    XF_HIP(hipMemcpy(&listed, d_cnt, 8, hipMemcpyDeviceToHost));
    if (listed != n_idx)
      return xf::set_error(XF_EINVAL, "xf_table_defrag: index holds %llu keys, %zu expected",
                           listed, n_idx);
    size_t tb = 0;
    XF_HIP(rocprim::radix_sort_pairs(nullptr, tb, k_all, k_sorted, r_all, r_sorted, n, 0, 64,
                                     (hipStream_t)0));
    char *tmp = nullptr;
    XF_TRY(sc.get(&tmp, tb));
    XF_HIP(rocprim::radix_sort_pairs((void *)tmp, tb, k_all, k_sorted, r_all, r_sorted, n, 0, 64,
                                     (hipStream_t)0));
  }
  // state in rank order, into the second buffer; the spare key's row, if any, follows the tier
  XF_TRY(alt_state(t, elems));
  float *w2 = t->w_alt;
  float2 *nz2 = T.nz ? t->nz_alt : nullptr;
  hipLaunchKernelGGL(k_move_rows, dim3(grid_for(n * T.dim)), dim3(kBlock), 0, 0, T.w, T.nz, T.dim,
                     r_sorted, n, (size_t)0, w2, nz2);
  if (spare)
    hipLaunchKernelGGL(k_move_rows, dim3(1), dim3(kBlock), 0, 0, T.w, T.nz, T.dim,
                       T.rows + T.cap, (size_t)1, n, w2, nz2);
  // the directories: one key per bucket on average, and the coarse one
  xf::TableDev N = T;
  tier_install(t, N, n, 0);
  XF_HIP(hipGetLastError());
  XF_HIP(hipDeviceSynchronize());
  // everything is built: from here on nothing fails half-way.  Empty the index (the spare
  // position keeps its key), point the spare at its new row, swap the arrays in.
  hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(T.cap)), dim3(kBlock), 0, 0, T.keys, (size_t)T.cap,
                     xf::kEmptyKey);
  hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(T.cap)), dim3(kBlock), 0, 0, T.rows, (size_t)T.cap,
                     xf::kNoRow);
  if (spare)
    hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(kBlock), 0, 0, T.rows + T.cap, (size_t)1,
                       (uint32_t)n);
  XF_HIP(hipGetLastError());
  XF_HIP(hipDeviceSynchronize());
  tier_swap(t);
  t->w_alt = T.w;  // (the old state: rows below the old count hold data, all below n)
  t->nz_alt = T.nz;
  N.w = w2;
  N.nz = nz2;
  T = N;
  ++t->epoch;  // every row number handed out before this call is stale
  return XF_OK;
}

// key -> row for a device key list (+ the weight payload when GATHER).  With a settled tier:
// the read-only tier lookup for all keys, then the insert-on-miss kernel over what is left.
// The work list belongs to the table: one resolve per table at a time.
// Measured on the config-2 shape (6.3e6 sorted keys against 1e7 settled ones): one key per
// lane 63 us, two 67, four 75; one key per directory bucket on average 63 us, two 67, four
// 80 (the 4-byte directory word is cheaper than a longer run of 8-byte keys); the old
// single-tier index took 95.  At 63 us the kernel moves ~260 MB: HBM-bound.
constexpr int kIlp = 1;

template <bool GATHER>
static int launch_resolve(xf_table *t, const uint64_t *d_keys, size_t n, uint32_t *d_rows,
                          float *d_vals, hipStream_t s, const uint32_t *d_out_idx = nullptr) {
  XF_REQUIRE(n < 0xFFFFFFFFull, "resolve: %zu keys in one call", n);
  if (t->T.nbase == 0) {
    hipLaunchKernelGGL(k_resolve<GATHER>, dim3(grid_for(n)), dim3(kBlock), 0, s, t->T, d_keys, n,
                       d_rows, d_vals, (const uint32_t *)nullptr,
                       (const unsigned long long *)nullptr, d_out_idx);
    XF_HIP(hipGetLastError());
    return XF_OK;
  }
  if (n > t->miss_cap) {
    XF_HIP(hipStreamSynchronize(s));
    if (t->miss) XF_HIP(hipFree(t->miss));
    t->miss = nullptr;
    t->miss_cap = 0;
    const size_t want = n + n / 4 + 1024;
    XF_HIP(hipMalloc((void **)&t->miss, want * 4));
    t->miss_cap = want;
  }
  if (!t->miss_n) XF_HIP(hipMalloc((void **)&t->miss_n, 8));
  XF_HIP(hipMemsetAsync(t->miss_n, 0, 8, s));
  const size_t chunk = (size_t)kBlock * kIlp;
  const size_t blocks = std::min<size_t>((n + chunk - 1) / chunk, 1u << 16);
  hipLaunchKernelGGL((k_pull_settled<GATHER, kIlp, xf::kBaseWin>), dim3((unsigned)blocks),
                     dim3(kBlock), 0, s, t->T, d_keys, n, d_rows, d_vals, t->miss, t->miss_n,
                     d_out_idx);
  // few keys miss in the steady state: a small grid, whose waves find the count in memory
  hipLaunchKernelGGL(k_resolve<GATHER>, dim3(std::min(grid_for(n), 2048)), dim3(kBlock), 0, s,
                     t->T, d_keys, n, d_rows, d_vals, (const uint32_t *)t->miss,
                     (const unsigned long long *)t->miss_n, d_out_idx);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

extern "C" int xf_table_resolve_dev(xf_table *t, const uint64_t *d_keys, size_t n,
                                    uint32_t *d_rows, void *stream) {
  XF_REQUIRE(t && (n == 0 || (d_keys && d_rows)), "xf_table_resolve_dev: null argument");
  if (n == 0) return XF_OK;
  t->rows_out = true;
  return launch_resolve<false>(t, d_keys, n, d_rows, nullptr, S(stream));
}

// Pull in one pass: resolve + the weight payload, for dim-1 tables.
extern "C" int xf_table_pull_dev(xf_table *t, const uint64_t *d_keys, size_t n,
                                 uint32_t *d_rows, float *d_vals, void *stream) {
  XF_REQUIRE(t && (n == 0 || (d_keys && d_rows && d_vals)), "xf_table_pull_dev: null argument");
  XF_REQUIRE(t->T.dim == 1, "xf_table_pull_dev: dim must be 1 (use resolve + gather)");
  // a lane that finds its key inserted by another lane of the same launch reads w[row] without
  // waiting for the inserter's first-touch stores: only pre-zeroed rows make that safe
  XF_REQUIRE(t->T.init_kind == XF_INIT_ZERO,
             "xf_table_pull_dev: only zero-initialised tables (use resolve + gather)");
  if (n == 0) return XF_OK;
  t->rows_out = true;
  return launch_resolve<true>(t, d_keys, n, d_rows, d_vals, S(stream));
}

extern "C" int xf_table_gather_dev(xf_table *t, const uint32_t *d_rows, size_t n,
                                   float *d_vals, void *stream) {
  XF_REQUIRE(t && (n == 0 || (d_rows && d_vals)), "xf_table_gather_dev: null argument");
  if (n == 0) return XF_OK;
  if (t->T.dim % 4 == 0 && ((uintptr_t)d_vals & 15) == 0) {
    const int dim4 = t->T.dim / 4;
    hipLaunchKernelGGL(k_gather4, dim3(grid_for(n * dim4)), dim3(kBlock), 0, S(stream),
                       (const float4 *)t->T.w, dim4, d_rows, n, (float4 *)d_vals);
  } else {
    hipLaunchKernelGGL(k_gather, dim3(grid_for(n * t->T.dim)), dim3(kBlock), 0, S(stream),
                       t->T.w, t->T.dim, d_rows, n, d_vals);
  }
  XF_HIP(hipGetLastError());
  return XF_OK;
}

extern "C" int xf_table_update_dev(xf_table *t, const uint32_t *d_rows, size_t n,
                                   const float *d_grads, void *stream) {
  XF_REQUIRE(t && (n == 0 || (d_rows && d_grads)), "xf_table_update_dev: null argument");
  if (n == 0) return XF_OK;
  ++t->writes;
  const dim3 g(grid_for(n * t->T.dim)), b(kBlock);
  if (t->cfg.opt_kind == XF_OPT_FTRL)
    hipLaunchKernelGGL(k_update<XF_OPT_FTRL>, g, b, 0, S(stream), t->T, d_rows, n, d_grads);
  else
    hipLaunchKernelGGL(k_update<XF_OPT_SGD>, g, b, 0, S(stream), t->T, d_rows, n, d_grads);
  XF_HIP(hipGetLastError());
  return XF_OK;
}


// Owner-side Pull over the key lists of ALL source ranks at once, visited in key order (see
// k_update_merged): the settled tier and the weights are swept once instead of once per
// source.  d_rows / d_vals are written in the per-source layout (entry d_order[i]).
extern "C" int xf_table_pull_ordered_dev(xf_table *t, const uint64_t *d_keys_sorted,
                                         const uint32_t *d_order, size_t n, uint32_t *d_rows,
                                         float *d_vals, void *stream) {
  XF_REQUIRE(t && (n == 0 || (d_keys_sorted && d_order && d_rows)),
             "xf_table_pull_ordered_dev: null argument");
  XF_REQUIRE(!d_vals || (t->T.dim == 1 && t->T.init_kind == XF_INIT_ZERO),
             "xf_table_pull_ordered_dev: values only for dim-1 zero-initialised tables");
  if (n == 0) return XF_OK;
  t->rows_out = true;
  if (d_vals) return launch_resolve<true>(t, d_keys_sorted, n, d_rows, d_vals, S(stream), d_order);
  return launch_resolve<false>(t, d_keys_sorted, n, d_rows, nullptr, S(stream), d_order);
}

extern "C" int xf_table_update_merged_dev(xf_table *t, const uint64_t *d_keys_sorted,
                                          const uint32_t *d_order, size_t n,
                                          const uint32_t *d_rows, const float *d_grads,
                                          void *stream) {
  XF_REQUIRE(t && (n == 0 || (d_keys_sorted && d_order && d_rows && d_grads)),
             "xf_table_update_merged_dev: null argument");
  if (n == 0) return XF_OK;
  ++t->writes;
  const dim3 g(grid_for(n * t->T.dim)), b(kBlock);
  if (t->cfg.opt_kind == XF_OPT_FTRL)
    hipLaunchKernelGGL(k_update_merged<XF_OPT_FTRL>, g, b, 0, S(stream), t->T, d_keys_sorted,
                       d_order, n, d_rows, d_grads);
  else
    hipLaunchKernelGGL(k_update_merged<XF_OPT_SGD>, g, b, 0, S(stream), t->T, d_keys_sorted,
                       d_order, n, d_rows, d_grads);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

// ---- ps-lite-shaped host API: Pull / Push followed by Wait ---------------------------
extern "C" int xf_table_pull(xf_table *t, const uint64_t *keys, size_t n, float *vals) {
  XF_REQUIRE(t && (n == 0 || (keys && vals)), "xf_table_pull: null argument");
  if (n == 0) return XF_OK;
  XF_TRY(ensure_scratch(t, n));
  XF_HIP(hipMemcpy(t->s_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice));
  XF_TRY(launch_resolve<false>(t, t->s_keys, n, t->s_rows, nullptr, nullptr));  // (rows stay inside)
  XF_TRY(xf_table_gather_dev(t, t->s_rows, n, t->s_vals, nullptr));
  XF_HIP(hipMemcpy(vals, t->s_vals, n * t->T.dim * sizeof(float), hipMemcpyDeviceToHost));
  XF_TRY(xf_table_check(t, nullptr));
  note_early(t, keys, n);
  return XF_OK;
}

extern "C" int xf_table_push(xf_table *t, const uint64_t *keys, size_t n, const float *grads) {
  XF_REQUIRE(t && (n == 0 || (keys && grads)), "xf_table_push: null argument");
  if (n == 0) return XF_OK;
  XF_TRY(ensure_scratch(t, n));
  XF_HIP(hipMemcpy(t->s_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice));
  XF_HIP(hipMemcpy(t->s_vals, grads, n * t->T.dim * sizeof(float), hipMemcpyHostToDevice));
  XF_TRY(launch_resolve<false>(t, t->s_keys, n, t->s_rows, nullptr, nullptr));  // (rows stay inside)
  XF_TRY(xf_table_update_dev(t, t->s_rows, n, t->s_vals, nullptr));
  XF_TRY(xf_table_check(t, nullptr));
  note_early(t, keys, n);
  return XF_OK;
}

// ---- state dump / load ----------------------------------------------------------------
extern "C" int xf_table_export(xf_table *t, uint64_t *keys, float *w, float *n_, float *z_,
                               size_t cap_entries, size_t *n_out) {
  XF_REQUIRE(t && n_out, "xf_table_export: null argument");
  uint64_t nk = 0;
  XF_TRY(xf_table_size(t, &nk));
  *n_out = (size_t)nk;
  if (!keys) return XF_OK;  // size query
  XF_REQUIRE(cap_entries >= nk, "xf_table_export: %zu entries offered, %llu needed",
             cap_entries, (unsigned long long)nk);
  if (nk == 0) return XF_OK;
  const int dim = t->T.dim;
  uint64_t *d_keys = nullptr;
  uint32_t *d_rows = nullptr;
  unsigned long long *d_cnt = nullptr;
  float *d_vals = nullptr;
  XF_HIP(hipMalloc((void **)&d_keys, nk * sizeof(uint64_t)));
  XF_HIP(hipMalloc((void **)&d_rows, nk * sizeof(uint32_t)));
  XF_HIP(hipMalloc((void **)&d_cnt, sizeof(unsigned long long)));
  XF_HIP(hipMalloc((void **)&d_vals, nk * dim * sizeof(float)));
  XF_HIP(hipMemset(d_cnt, 0, sizeof(unsigned long long)));
  if (t->T.nbase)
    hipLaunchKernelGGL(k_list_base, dim3(grid_for(t->T.nbase)), dim3(kBlock), 0, 0, t->T, d_keys,
                       d_rows);
  hipLaunchKernelGGL(k_list_occupied, dim3(grid_for((size_t)t->T.cap + 1)), dim3(kBlock), 0, 0,
                     t->T, d_keys, d_rows, d_cnt, (size_t)nk);
  XF_HIP(hipGetLastError());
  std::vector<uint64_t> hk(nk);
  XF_HIP(hipMemcpy(hk.data(), d_keys, nk * sizeof(uint64_t), hipMemcpyDeviceToHost));
  std::vector<size_t> order(nk);
  std::iota(order.begin(), order.end(), (size_t)0);
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return hk[a] < hk[b]; });
  for (size_t i = 0; i < nk; ++i) keys[i] = hk[order[i]];
  std::vector<float> vals(nk * dim);
  float *dsts[3] = {w, n_, z_};
  for (int a = 0; a < 3; ++a) {
    if (!dsts[a]) continue;
    if (a > 0 && !t->T.nz) {  // SGD tables have no n/z: report zeros
      std::fill(dsts[a], dsts[a] + nk * dim, 0.0f);
      continue;
    }
    if (a == 0)
      hipLaunchKernelGGL(k_gather, dim3(grid_for(nk * dim)), dim3(kBlock), 0, 0, t->T.w, dim,
                         d_rows, (size_t)nk, d_vals);
    else
      hipLaunchKernelGGL(k_gather_nz, dim3(grid_for(nk * dim)), dim3(kBlock), 0, 0, t->T.nz, a - 1,
                         dim, d_rows, (size_t)nk, d_vals);
    XF_HIP(hipGetLastError());
    XF_HIP(hipMemcpy(vals.data(), d_vals, nk * dim * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nk; ++i)
      std::copy(vals.begin() + order[i] * dim, vals.begin() + (order[i] + 1) * dim,
                dsts[a] + i * dim);
  }
  XF_HIP(hipFree(d_keys));
  XF_HIP(hipFree(d_rows));
  XF_HIP(hipFree(d_cnt));
  XF_HIP(hipFree(d_vals));
  return XF_OK;
}

// rows whose w is not ftrl_w_of(n, z), bit for bit
__global__ void __launch_bounds__(kBlock)
k_check_w_of_nz(xf::TableDev T, const uint32_t *__restrict__ rows, size_t n,
                unsigned long long *__restrict__ bad) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = rows[i];
    const float2 s = T.nz[r];
    const float w = xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, s.x, s.y);
    if (__float_as_uint(w) != __float_as_uint(T.w[r])) atomicAdd(bad, 1ull);
  }
}

extern "C" int xf_table_import(xf_table *t, const uint64_t *keys, size_t n, const float *w,
                               const float *n_, const float *z_) {
  XF_REQUIRE(t && (n == 0 || keys), "xf_table_import: null argument");
  if (n == 0) return XF_OK;
  ++t->writes;
  XF_TRY(ensure_scratch(t, n));
  XF_HIP(hipMemcpy(t->s_keys, keys, n * sizeof(uint64_t), hipMemcpyHostToDevice));
  XF_TRY(xf_table_resolve_dev(t, t->s_keys, n, t->s_rows, nullptr));
  const float *srcs[3] = {w, n_, z_};
  for (int a = 0; a < 3; ++a) {
    if (!srcs[a] || (a > 0 && !t->T.nz)) continue;
    XF_HIP(hipMemcpy(t->s_vals, srcs[a], n * t->T.dim * sizeof(float), hipMemcpyHostToDevice));
    if (a == 0)
      hipLaunchKernelGGL(k_scatter_rows, dim3(grid_for(n * t->T.dim)), dim3(kBlock), 0, 0, t->T.w,
                         t->T.dim, t->s_rows, n, t->s_vals);
    else
      hipLaunchKernelGGL(k_scatter_nz, dim3(grid_for(n * t->T.dim)), dim3(kBlock), 0, 0, t->T.nz,
                         a - 1, t->T.dim, t->s_rows, n, t->s_vals);
    XF_HIP(hipGetLastError());
  }
  if (t->T.w_of_nz) {  // do the imported rows hold the w their (n, z) give?  (a model file this
                       // library wrote under the same hyper-parameters does)
    unsigned long long *d_bad = nullptr, bad = 0;
    XF_HIP(hipMalloc((void **)&d_bad, 8));
    XF_HIP(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(k_check_w_of_nz, dim3(grid_for(n)), dim3(kBlock), 0, 0, t->T, t->s_rows, n,
                       d_bad);
    const hipError_t e1 = hipGetLastError();
    const hipError_t e2 = hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    XF_HIP(e1);
    XF_HIP(e2);
    if (bad) {
      t->w_tainted = true;
      refresh_hyper(t);
    }
  }
  return xf_table_check(t, nullptr);
}

// does the gradient + Push of this table derive a key's old weight from its (n, z)
// (TableDev::w_of_nz) instead of reading it?  (tests, tools)
extern "C" int xf_table_w_derived(xf_table *t, int *yes) {
  XF_REQUIRE(t && yes, "xf_table_w_derived: null argument");
  *yes = t->T.w_of_nz ? 1 : 0;
  return XF_OK;
}

// keys[i] (or keys[list[i]]) for i < n into out
__global__ void __launch_bounds__(kBlock)
k_take_keys(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ list, size_t n,
            uint64_t *__restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = keys[list ? list[i] : i];
}
// number of positions of a sorted list that start a run
__global__ void __launch_bounds__(kBlock)
k_count_heads(const uint64_t *__restrict__ sorted, size_t n, unsigned long long *__restrict__ out) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    c += (i == 0 || sorted[i] != sorted[i - 1]) ? 1u : 0u;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// distinct keys among keys[list[0..n)) (list == null: keys[0..n)): one radix sort.  Only run
// when a table is about to be grown on the strength of a count that includes duplicates.
static int count_distinct(const uint64_t *d_keys, const uint32_t *d_list, size_t n,
                          hipStream_t s, size_t *out) {
  xf::Scratch sc;
  uint64_t *a = nullptr, *b = nullptr;
  unsigned long long *d_c = nullptr;
  XF_TRY(sc.get(&a, n));
  XF_TRY(sc.get(&b, n));
  XF_TRY(sc.get(&d_c, 1));
  hipLaunchKernelGGL(k_take_keys, dim3(grid_for(n)), dim3(kBlock), 0, s, d_keys, d_list, n, a);
  size_t tb = 0;
  XF_HIP(rocprim::radix_sort_keys(nullptr, tb, a, b, n, 0, 64, s));
  void *tmp = nullptr;
  XF_TRY(sc.get((char **)&tmp, tb));
  XF_HIP(rocprim::radix_sort_keys(tmp, tb, a, b, n, 0, 64, s));
  XF_HIP(hipMemsetAsync(d_c, 0, 8, s));
  hipLaunchKernelGGL(k_count_heads, dim3(std::min(grid_for(n), 1024)), dim3(kBlock), 0, s, b, n,
                     d_c);
  XF_HIP(hipGetLastError());
  unsigned long long c = 0;
  XF_HIP(hipMemcpyAsync(&c, d_c, 8, hipMemcpyDeviceToHost, s));
  XF_HIP(hipStreamSynchronize(s));
  *out = (size_t)c;
  return XF_OK;
}

// key -> row for ANY device key list (duplicates allowed, any order), inserting the missing
// keys; with allow_grow the table is first grown (xf_table_reserve) when the keys that may be
// new would push the load past 0.6.  The raw keys of a minibatch are resolved with this
// (xf_batch_compile_local_dev).  Synchronises the stream.
namespace xf {
int table_resolve_any(xf_table *t, const uint64_t *d_keys, size_t n, uint32_t *d_rows,
                      hipStream_t s, bool allow_grow) {
  XF_REQUIRE(t && (n == 0 || (d_keys && d_rows)), "table_resolve_any: null argument");
  XF_REQUIRE(n < 0xFFFFFFFFull, "table_resolve_any: %zu keys in one call", n);
  if (n == 0) return XF_OK;
  t->rows_out = true;
  size_t maybe_new = n;
  const bool tiered = t->T.nbase != 0;
  if (tiered) {
    if (n > t->miss_cap) {
      XF_HIP(hipStreamSynchronize(s));
      if (t->miss) XF_HIP(hipFree(t->miss));
      t->miss = nullptr;
      t->miss_cap = 0;
      const size_t want = n + n / 4 + 1024;
      XF_HIP(hipMalloc((void **)&t->miss, want * 4));
      t->miss_cap = want;
    }
    if (!t->miss_n) XF_HIP(hipMalloc((void **)&t->miss_n, 8));
    XF_HIP(hipMemsetAsync(t->miss_n, 0, 8, s));
    // an unsorted list: every lookup is two dependent random reads (directory, key run), so
    // eight keys per lane are in flight (4: 0.806, 8: 0.78, 16: 0.80 ms per key build + step; a
    // sorted list sweeps the tier and gains nothing from it)
    constexpr int kAnyIlp = 8;
    const size_t chunk = (size_t)kBlock * kAnyIlp;
    const size_t blocks = std::min<size_t>((n + chunk - 1) / chunk, 1u << 16);
    hipLaunchKernelGGL(k_lookup_any<kAnyIlp>, dim3((unsigned)blocks), dim3(kBlock), 0, s, t->T,
                       d_keys, n, d_rows, t->miss, t->miss_n);
    XF_HIP(hipGetLastError());
    unsigned long long misses = 0;
    XF_HIP(hipMemcpyAsync(&misses, t->miss_n, 8, hipMemcpyDeviceToHost, s));
    XF_HIP(hipStreamSynchronize(s));
    maybe_new = (size_t)misses;
    if (maybe_new == 0) return XF_OK;
  }
  if (allow_grow) {
    XF_HIP(hipStreamSynchronize(s));
    xf::TableStat st;
    XF_TRY(read_stat(t, &st));
    const uint64_t cap = t->T.cap;
    if ((st.count + maybe_new) * 10 > cap * 6) {
      // `maybe_new` counts nonzeros, duplicates included (a 1e7-nonzero minibatch over 1e6
      // keys would grow a 4M table to 32M positions): before growing, count the distinct ones
      size_t distinct = maybe_new;
      if (tiered) {
        XF_TRY(count_distinct(d_keys, t->miss, maybe_new, s, &distinct));
      } else {
        XF_TRY(count_distinct(d_keys, nullptr, n, s, &distinct));
      }
      if ((st.count + distinct) * 10 > cap * 6) {
        uint64_t want = cap * 2;
        while ((st.count + distinct) * 10 > want * 6) want *= 2;
        XF_TRY(xf_table_reserve(t, want));
      }
    }
  }
  hipLaunchKernelGGL(k_resolve<false>, dim3(std::min(grid_for(maybe_new), 4096)), dim3(kBlock), 0,
                     s, t->T, d_keys, n, d_rows, (float *)nullptr,
                     tiered ? (const uint32_t *)t->miss : (const uint32_t *)nullptr,
                     tiered ? (const unsigned long long *)t->miss_n
                            : (const unsigned long long *)nullptr,
                     (const uint32_t *)nullptr);
  XF_HIP(hipGetLastError());
  XF_HIP(hipStreamSynchronize(s));
  return XF_OK;
}

// Room for the keys of d_keys[0 .. n) that may be new (duplicates allowed, any order): the table is
// grown (xf_table_reserve) when they would push the load past 0.6 — on the strength of a count
// of the DISTINCT keys when the number of nonzeros alone would say so.  The arrival build calls
// this before it inserts (xf_keybuild.hip).  Synchronises the stream; may move the table's arrays.
int table_grow_for(xf_table *t, const uint64_t *d_keys, size_t n, hipStream_t s) {
  XF_HIP(hipStreamSynchronize(s));
  xf::TableStat st;
  XF_TRY(read_stat(t, &st));
  const uint64_t cap = t->T.cap;
  if ((st.count + n) * 10 <= cap * 6) return XF_OK;
  size_t distinct = n;
  XF_TRY(count_distinct(d_keys, nullptr, n, s, &distinct));
  if ((st.count + distinct) * 10 > cap * 6) {
    uint64_t want = cap * 2;
    while ((st.count + distinct) * 10 > want * 6) want *= 2;
    XF_TRY(xf_table_reserve(t, want));
  }
  return XF_OK;
}

// the keys the table holds (waits for the stream: inserts on it have finished)
int table_count(xf_table *t, hipStream_t s, uint64_t *count) {
  XF_HIP(hipStreamSynchronize(s));
  xf::TableStat st;
  XF_TRY(read_stat(t, &st));
  *count = st.count;
  return XF_OK;
}

// A table that holds NOTHING takes a list of distinct keys, ascending, as its settled tier: key r
// owns state row r, fresh (what the first Pull of every key would have made of it, ftrl.h:56).
// What xf_table_defrag builds from the arrival index, without the index and without a state move
// — for the first minibatch of a run (xf_keybuild.hip: "an empty table").  The keys go where
// table_first_tier says (the tier's allocation: keys and directories in one); no row number
// existed before, so none goes stale: the epoch stays.  In stream order, nothing is waited for.
int table_first_tier(xf_table *t, size_t d, uint64_t **keys) { return tier_alloc(t, d, keys); }
int table_settle_first(xf_table *t, size_t d, hipStream_t s) {
  xf::TableDev &T = t->T;
  XF_REQUIRE(T.nbase == 0 && t->tier_next && d > 0 && d <= T.max_rows && d < 0xFFFFFFF0ull,
             "table_settle_first: %zu keys into a table of %llu rows, %llu settled", d,
             (unsigned long long)T.max_rows, (unsigned long long)T.nbase);
  xf::TableDev N = T;
  tier_install(t, N, d, s);
  if (T.init_kind != XF_INIT_ZERO)
    hipLaunchKernelGGL(k_first_rows, dim3(grid_for(d * (size_t)T.dim)), dim3(kBlock), 0, s, N,
                       N.bkeys, d);
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, s, T.stat, (unsigned long long)d);
  XF_HIP(hipGetLastError());
  tier_swap(t);  // (a table without a settled tier has none)
  T = N;
  ++t->writes;  // (rows that records derived from the state know nothing of)
  t->rec_all_ok = false;
  return XF_OK;
}

// The keys of a table without a settled tier when the host API put every one of them there and
// nobody outside holds a row number of it: sorted, distinct.  `count`: the table's key count
// (the caller has just read it); false: not known, or not so.
bool table_early_keys(xf_table *t, uint64_t count, std::vector<uint64_t> *out) {
  out->clear();
  if (t->T.nbase != 0 || t->early_over || t->rows_out || t->early.empty()) return false;
  std::vector<uint64_t> k = t->early;
  std::sort(k.begin(), k.end());
  k.erase(std::unique(k.begin(), k.end()), k.end());
  if (k.size() != count || k.back() == xf::kEmptyKey) return false;
  *out = std::move(k);
  return true;
}
void table_note_rows_out(xf_table *t) { t->rows_out = true; }
// could this table's keys be none, or only what the host API put there?  (the host's knowledge:
// the count is on the device)
bool table_maybe_first(const xf_table *t) {
  return t->T.nbase == 0 && !t->early_over && !t->rows_out;
}
// ... taken out of the arrival index: their state into tw / tnz ([n * dim], key order of d_keys),
// their rows zeroed, their positions emptied, the key count 0.  d_pos: n words of scratch.
int table_take_early(xf_table *t, const uint64_t *d_keys, size_t n, float *tw, float2 *tnz,
                     unsigned long long *d_pos, hipStream_t s) {
  const xf::TableDev &T = t->T;
  hipLaunchKernelGGL(k_early_find, dim3(grid_for(n)), dim3(kBlock), 0, s, T, d_keys, n, d_pos, tw,
                     tnz);
  hipLaunchKernelGGL(k_early_clear, dim3(grid_for(n)), dim3(kBlock), 0, s, T, d_pos, n);
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, s, T.stat, 0ull);
  XF_HIP(hipGetLastError());
  return XF_OK;
}
// ... and put back after table_settle_first: found in the tier or inserted behind it, the state
// written to the row the key has now.  Their rows have changed: the epoch moves.
int table_put_early(xf_table *t, const uint64_t *d_keys, size_t n, const float *tw,
                    const float2 *tnz, uint32_t *d_rows, hipStream_t s) {
  XF_TRY(launch_resolve<false>(t, d_keys, n, d_rows, nullptr, s));
  hipLaunchKernelGGL(k_early_put, dim3(grid_for(n)), dim3(kBlock), 0, s, t->T, d_rows, n, tw, tnz);
  XF_HIP(hipGetLastError());
  ++t->epoch;
  t->early.clear();
  return XF_OK;
}

// grow the table (xf_table_reserve) when `incoming` more keys would push the load past 0.6.
// Synchronises the device.
int table_ensure_room(xf_table *t, size_t incoming) {
  XF_HIP(hipDeviceSynchronize());
  xf::TableStat st;
  XF_TRY(read_stat(t, &st));
  const uint64_t cap = t->T.cap;
  if ((st.count + incoming) * 10 > cap * 6) {
    uint64_t want = cap * 2;
    while ((st.count + incoming) * 10 > want * 6) want *= 2;
    XF_TRY(xf_table_reserve(t, want));
  }
  return XF_OK;
}

// dst[i] = src[rows[i]] for any float array indexed by state row (parity hook of the cells path)
int gather_f32(const float *src, const uint32_t *rows, size_t n, float *dst, hipStream_t s) {
  if (n == 0) return XF_OK;
  hipLaunchKernelGGL(k_gather, dim3(grid_for(n)), dim3(kBlock), 0, s, src, 1, rows, n, dst);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

// used by xf_model.hip / xf_cells.hip
const TableDev &table_dev(const xf_table *t) { return t->T; }
void **table_aux(xf_table *t, uint64_t **epoch) {
  *epoch = &t->aux_epoch;
  return &t->aux;
}
// device-to-device copy by a kernel on the caller's stream.  (hipMemcpyAsync picks its engine
// itself: a 25 MB copy in the middle of a step was seen at 2.3 TB/s — blit kernel — in one run
// and at 0.27 TB/s with ~80 us of cross-queue waits around it — SDMA — in the next.)
int device_copy(void *dst, const void *src, size_t bytes, hipStream_t s) {
  if (bytes == 0) return XF_OK;
  if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 15u) == 0) {
    const size_t n = bytes / 16;
    const size_t blocks = std::min<size_t>(std::max<size_t>(n / (kBlock * 4), 1), 4096);
    hipLaunchKernelGGL(k_copy16, dim3((unsigned)blocks), dim3(kBlock), 0, s, (const uint4 *)src,
                       (uint4 *)dst, n);
  } else if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 3u) == 0) {
    // (float / index lists of odd length: one byte per lane moved 25 MB in 53 us)
    const size_t n = bytes / 4;
    const size_t blocks = std::min<size_t>(std::max<size_t>(n / (kBlock * 4), 1), 4096);
    hipLaunchKernelGGL(k_copy4, dim3((unsigned)blocks), dim3(kBlock), 0, s,
                       (const uint32_t *)src, (uint32_t *)dst, n);
  } else {
    hipLaunchKernelGGL(k_copy1, dim3(grid_for(bytes)), dim3(kBlock), 0, s,
                       (const unsigned char *)src, (unsigned char *)dst, bytes);
  }
  XF_HIP(hipGetLastError());
  return XF_OK;
}
uint64_t table_row_bound(const xf_table *t) { return t->T.max_rows + 1; }  // > every state row
const float *table_weights(const xf_table *t) { return t->T.w; }
int table_dim(const xf_table *t) { return t->T.dim; }
// static part of the owner's merged update (valid for one row numbering of `t`)
int table_head_rows(const uint64_t *d_keys_sorted, const uint32_t *d_order,
                    const uint32_t *d_rows, size_t n, uint32_t *d_hrow, hipStream_t s) {
  if (n == 0) return XF_OK;
  hipLaunchKernelGGL(k_head_rows, dim3(grid_for(n)), dim3(kBlock), 0, s, d_keys_sorted, d_order,
                     d_rows, n, d_hrow);
  XF_HIP(hipGetLastError());
  return XF_OK;
}
// the Pushes of all sources, applied key by key in the merged order (xf_table_update_merged_dev
// with the keys' rows and run heads precomputed)
int table_update_heads(xf_table *t, const uint32_t *d_hrow, const uint32_t *d_order, size_t n,
                       const float *d_grads, hipStream_t s) {
  if (n == 0) return XF_OK;
  ++t->writes;
  const dim3 g(grid_for(n * t->T.dim)), b(kBlock);
  if (t->cfg.opt_kind == XF_OPT_FTRL)
    hipLaunchKernelGGL(k_update_heads<XF_OPT_FTRL>, g, b, 0, s, t->T, d_hrow, d_order, n, d_grads);
  else
    hipLaunchKernelGGL(k_update_heads<XF_OPT_SGD>, g, b, 0, s, t->T, d_hrow, d_order, n, d_grads);
  XF_HIP(hipGetLastError());
  return XF_OK;
}
uint64_t table_uid(const xf_table *t) { return t->uid; }
uint64_t table_epoch(const xf_table *t) { return t->epoch; }
// the per-row records of another module: `row_bytes` per state row, (re)allocated — contents
// undefined, *gen changed — whenever the state has been reallocated since the last call or the
// caller's `tag` (what else the records were derived from) is another one
int table_records(xf_table *t, size_t row_bytes, uint64_t tag, void **rec, uint64_t *gen) {
  const size_t rows = (size_t)t->T.max_rows + 1;
  if (t->rec && t->rec_tag != tag) ++t->rec_gen;
  t->rec_tag = tag;
  if (!t->rec || t->rec_rows != rows || t->rec_row_bytes != row_bytes) {
    if (t->rec) {
      XF_HIP(hipDeviceSynchronize());  // (a step that still reads the old records)
      XF_HIP(hipFree(t->rec));
      t->rec = nullptr;
    }
    XF_HIP(hipMalloc(&t->rec, rows * row_bytes));
    t->rec_rows = rows;
    t->rec_row_bytes = row_bytes;
    ++t->rec_gen;
  }
  *rec = t->rec;
  *gen = t->rec_gen;
  return XF_OK;
}
void table_records_all_set(xf_table *t, const uint64_t key[5]) {
  t->rec_all_ok = true;
  for (int i = 0; i < 5; ++i) t->rec_all[i] = key[i];
}
bool table_records_all_is(const xf_table *t, const uint64_t key[5]) {
  if (!t->rec_all_ok) return false;
  for (int i = 0; i < 5; ++i)
    if (t->rec_all[i] != key[i]) return false;
  return true;
}
// weight writes so far by code that does not maintain the records; table_note_write: one more
uint64_t table_writes(const xf_table *t) { return t->writes; }
void table_note_write(xf_table *t) { ++t->writes; }
}  // namespace xf
