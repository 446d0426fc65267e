// xf_cells.hip — the cell-sorted minibatch and the LR kernels that stream it (gfx950).
//
// Replaces (paths relative to /root/reference):
//   key build of LRWorker::update      src/model/lr/lr_worker.cc:146-166  (cells_build)
//   LRWorker::calculate_loss           src/model/lr/lr_worker.cc:121-143  (k_lr_fwd_cells,
//                                                                          k_lr_finalize_cells)
//   LRWorker::calculate_gradient       src/model/lr/lr_worker.cc:100-119  (k_lr_grad_cells)
//   KVWorker::Push -> FTRL / SGD       src/optimizer/ftrl.h:54-74, sgd.h:52 (fused in the same)
// Layout and rationale: xf_cells.h.  HBM-bound integer/byte work, no MFMA.
//
// Numerics: a row's / a key's sum is accumulated in fp64 (LDS atomics) and rounded to fp32
// once, where the reference holds an fp32 value.  fp64 addition of fp32 addends is exact —
// hence independent of the order the atomics land in — as long as the addends of one sum span
// fewer than 2^(29 - log2 n) in magnitude (n addends); beyond that the LAST BIT of the fp64
// sum may depend on the order, which changes the fp32 result with probability ~n * 2^-29.
// The reference's own order inside a key is std::sort's (unspecified, lr_worker.cc:162).
// Chunks with more than kSliceMax entries (power-law heads) are cut into slices whose partial
// sums are added in slice order by a second kernel.
#include <hip/hip_runtime.h>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <vector>

#include "xf_batch.h"
#include "xf_cells.h"
#include "xf_device.h"
#include "xf_scratch.h"

namespace {

using xf::kBlk;
using xf::kChunk;
using xf::kChunkBits;
using xf::kNoDump;
using xf::kSliceMax;
using xf::kWinMax;
using xf::kRowMask;
using xf::kTagShift;
using xf::kTagMask;

constexpr int kBlock = 256;
#ifndef XF_FWD_BLOCK
#define XF_FWD_BLOCK 1024
#endif
#ifndef XF_FWD_GROUPS
#define XF_FWD_GROUPS 256
#endif
constexpr int kFwdBlock = XF_FWD_BLOCK;  // one forward workgroup per CU (its LDS holds a row window)
constexpr uint32_t kFwdGroups = XF_FWD_GROUPS;  // workgroup slots of the chip at that size

inline int grid_for(size_t n, int block = kBlock) {
  size_t g = (n + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}
#define XF_GRID_STRIDE(i, n)                                                     \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(n); \
       i += (size_t)gridDim.x * blockDim.x)

// ------------------------------------------------------------------------------- build
// (cell number, entry) of every nonzero, row-major.  One wavefront per row.
__global__ void __launch_bounds__(kBlock)
k_cell_keys(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ src,
            const uint32_t *__restrict__ map, uint32_t R, uint32_t W, uint32_t nchunk,
            uint32_t chunk0, uint32_t *__restrict__ cid, uint32_t *__restrict__ ent) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nw = gridDim.x * (kBlock / 64);
  for (uint32_t r = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; r < R; r += nw) {
    const uint32_t v = r / W, rin = r - v * W;
    for (uint32_t j = rowptr[r] + lane; j < rowptr[r + 1]; j += 64) {
      const uint32_t s = src[j];
      const uint32_t idx = map ? map[s] : s;
      const uint32_t chunk = (idx >> kChunkBits) - chunk0;
      cid[j] = v * nchunk + chunk;
      ent[j] = ((chunk & kTagMask) << kTagShift) | (rin << kChunkBits) | (idx & (kChunk - 1));
    }
  }
}

// cellptr[c] = number of entries in cells < c: a lower bound in the sorted cell numbers per
// cell (a pass over the entries that closes cells at every change took 80 us for 1e7 entries;
// 22k searches of 24 steps take 5)
__global__ void __launch_bounds__(kBlock)
k_cellptr(const uint32_t *__restrict__ cid_s, uint32_t n, uint32_t ncell,
          uint32_t *__restrict__ cellptr) {
  XF_GRID_STRIDE(c, (size_t)ncell + 1) {
    uint32_t lo = 0, hi = n;  // first j with cid_s[j] >= c
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (cid_s[mid] < (uint32_t)c) lo = mid + 1;
      else
        hi = mid;
    }
    cellptr[c] = lo;
  }
}

__global__ void __launch_bounds__(kBlock)
k_blk_cell(const uint32_t *__restrict__ cid_s, uint32_t nblk, uint32_t ncell,
           uint32_t *__restrict__ blk_cell) {
  XF_GRID_STRIDE(b, (size_t)nblk + 1)
  blk_cell[b] = b < nblk ? cid_s[(size_t)b * kBlk] : ncell - 1;
}

// gradient work items: chunk c is cut into ceil(n_c / kSliceMax) slices (none when empty).
// One workgroup: slices per chunk, their exclusive scan (off: first item of the chunk) and the
// scan of the "is split" flags (soff: index among the split chunks); totals in [nchunk].
constexpr int kPlanBlock = 1024;
__global__ void __launch_bounds__(kPlanBlock)
k_plan_items(const uint32_t *__restrict__ cellptr, uint32_t nchunk, uint32_t nwin,
             uint32_t *__restrict__ nsl, uint32_t *__restrict__ off,
             uint32_t *__restrict__ soff, uint32_t *__restrict__ poff) {
  __shared__ uint32_t wsum[3][kPlanBlock / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t per = (nchunk + kPlanBlock - 1) / kPlanBlock;
  const uint32_t c0 = min(tid * per, nchunk), c1 = min(c0 + per, nchunk);
  auto slices = [&](uint32_t c) -> uint32_t {
    uint32_t n = 0, v = 0;
    for (; v + 4 <= nwin; v += 4) {  // (four windows' bounds requested at a time)
      uint32_t b[4], e[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        b[k] = cellptr[(size_t)(v + k) * nchunk + c];
        e[k] = cellptr[(size_t)(v + k) * nchunk + c + 1];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) n += e[k] - b[k];
    }
    for (; v < nwin; ++v)
      n += cellptr[(size_t)v * nchunk + c + 1] - cellptr[(size_t)v * nchunk + c];
    return (n + kSliceMax - 1) / kSliceMax;
  };
  uint32_t a = 0, b = 0, p = 0;
  for (uint32_t c = c0; c < c1; ++c) {
    const uint32_t S = slices(c);
    a += S;
    b += S > 1 ? 1u : 0u;
    p += S > 1 ? S : 0u;
  }
  uint32_t ia = a, ib = b, ip = p;  // inclusive scan over the threads
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t ta = __shfl_up(ia, o), tb = __shfl_up(ib, o), tp = __shfl_up(ip, o);
    if ((int)lane >= o) {
      ia += ta;
      ib += tb;
      ip += tp;
    }
  }
  if (lane == 63) {
    wsum[0][wave] = ia;
    wsum[1][wave] = ib;
    wsum[2][wave] = ip;
  }
  __syncthreads();
  uint32_t ba = 0, bb = 0, bp = 0;
  for (uint32_t w = 0; w < wave; ++w) {
    ba += wsum[0][w];
    bb += wsum[1][w];
    bp += wsum[2][w];
  }
  // exclusive prefix of this thread's range
  uint32_t ea = ba + ia - a, eb = bb + ib - b, ep = bp + ip - p;
  for (uint32_t c = c0; c < c1; ++c) {
    const uint32_t S = slices(c);  // (recomputed: a reload of nsl[] would wait for the stores)
    nsl[c] = S;
    off[c] = ea;
    soff[c] = eb;
    poff[c] = ep;
    ea += S;
    eb += S > 1 ? 1u : 0u;
    ep += S > 1 ? S : 0u;
  }
  if (tid == kPlanBlock - 1) {
    nsl[nchunk] = 0;
    off[nchunk] = ba + ia;
    soff[nchunk] = bb + ib;
    poff[nchunk] = bp + ip;
  }
}

__global__ void __launch_bounds__(kBlock)
k_items_fill(const uint32_t *__restrict__ nsl, const uint32_t *__restrict__ off,
             const uint32_t *__restrict__ soff, const uint32_t *__restrict__ poff,
             uint32_t nchunk, uint32_t *__restrict__ item_chunk,
             uint32_t *__restrict__ item_slice, uint32_t *__restrict__ item_dump,
             uint32_t *__restrict__ split_chunk) {
  // Longest items first: the slices of the split chunks (kSliceMax entries each) take the first
  // poff[nchunk] places of the list, the unsplit chunks follow in chunk order — a slice that
  // starts in the last round of workgroups was the tail of the power-law gradient kernel.
  // (a wavefront per 64 chunks; the slices of a split chunk are written by all its lanes: a
  // power-law head chunk has a hundred and more, one lane writing them all took 96 us)
  const uint32_t P = poff[nchunk];
  const uint32_t lane = threadIdx.x & 63u;
  const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const size_t nwave = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t cb = wave0 * 64; cb < nchunk; cb += nwave * 64) {
    const size_t c = cb + lane;
    const uint32_t S = c < nchunk ? nsl[c] : 0u;
    if (S == 1) {
      const uint32_t i = P + (off[c] - poff[c]);
      item_chunk[i] = (uint32_t)c;
      item_slice[i] = 1u << 16;
      item_dump[i] = kNoDump;
    }
    unsigned long long m = __ballot(S > 1);
    while (m) {  // wave-uniform
      const int l = __ffsll((long long)m) - 1;
      m &= m - 1;
      const uint32_t cs = (uint32_t)cb + (uint32_t)l, Ss = (uint32_t)__shfl((int)S, l);
      const uint32_t p0 = poff[cs], so = soff[cs];
      for (uint32_t s = lane; s < Ss; s += 64) {
        item_chunk[p0 + s] = cs;
        item_slice[p0 + s] = s | (Ss << 16);
        item_dump[p0 + s] = so;
      }
      if (lane == 0) split_chunk[so] = cs;
    }
  }
}

// The forward's copy of the cells: inside every cell the entries ordered by their position in
// the chunk (the entry's low bits), so that neighbouring lanes gather neighbouring weights.  A
// counting sort in LDS over the chunk's kChunk positions — a wavefront per cell (config 2: ~680
// entries per cell, an owner's 32-window minibatch: ~50), the whole workgroup on a cell of more
// than kSortWave entries, and a cell beyond kSortMax (a power-law head key's: one position holds
// most of it) is copied as it is.  Not stable — the order of a position's entries is whatever
// the LDS cursors make of it: the forward's fp64 row sums do not depend on it (exact).
// Replaces rocprim::segmented_radix_sort_keys (2 x 112 us per 1e7 entries, round 2).
constexpr uint32_t kSortWave = 4096, kSortMax = 1u << 17;
constexpr int kSortCells = 4;  // cells (wavefronts) per workgroup
__device__ __forceinline__ uint32_t sortp(uint32_t i) { return i + (i >> 5); }  // (bank padding)
__device__ __forceinline__ void lds_wave_sync() {
  // LDS accesses of ONE wavefront are served in order; this keeps the compiler from moving them
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <int NT>  // threads that share `bins`: 64 (a wavefront, no barriers) or kBlock
__device__ __forceinline__ void cell_sort_pos(uint32_t *bins, const uint32_t *__restrict__ in,
                                              uint32_t *__restrict__ out, uint32_t n, uint32_t t) {
  auto sync = [&]() {
    if (NT == 64) lds_wave_sync();
    else
      __syncthreads();
  };
  constexpr uint32_t kPer = kChunk / NT;  // consecutive bins a thread scans
  for (uint32_t i = t; i < kChunk + kChunk / 32; i += NT) bins[i] = 0;
  sync();
  for (uint32_t i = t; i < n; i += NT) atomicAdd(&bins[sortp(in[i] & (kChunk - 1))], 1u);
  sync();
  uint32_t sum = 0;
#pragma unroll 8
  for (uint32_t k = 0; k < kPer; ++k) sum += bins[sortp(t * kPer + k)];
  // exclusive scan of the threads' sums: over the wavefront, then (NT > 64) over the wavefronts
  const uint32_t lane = t & 63u;
  uint32_t inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(inc, o);
    if ((int)lane >= o) inc += y;
  }
  uint32_t run = inc - sum;
  if (NT > 64) {
    uint32_t *wsum = bins + kChunk + kChunk / 32;  // [NT / 64] behind the bins
    if (lane == 63) wsum[t >> 6] = inc;
    __syncthreads();
    for (uint32_t w = 0; w < (t >> 6); ++w) run += wsum[w];
  }
#pragma unroll 8
  for (uint32_t k = 0; k < kPer; ++k) {
    const uint32_t x = bins[sortp(t * kPer + k)];
    bins[sortp(t * kPer + k)] = run;
    run += x;
  }
  sync();
  for (uint32_t i = t; i < n; i += NT) {
    const uint32_t e = in[i];
    out[atomicAdd(&bins[sortp(e & (kChunk - 1))], 1u)] = e;
  }
  sync();
}

__global__ void __launch_bounds__(kBlock)
k_cells_sort_pos(const uint32_t *__restrict__ entries, uint32_t *__restrict__ out,
                 const uint32_t *__restrict__ cellptr, uint32_t ncell) {
  static_assert(kBlock == 64 * kSortCells, "a wavefront per cell");
  __shared__ uint32_t bins[kSortCells][kChunk + kChunk / 32 + 8];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  const uint32_t c0 = blockIdx.x * kSortCells;
  {
    const uint32_t c = c0 + wave;
    if (c < ncell) {  // wave-uniform
      const uint32_t b = cellptr[c], n = cellptr[c + 1] - b;
      if (n && n <= kSortWave) cell_sort_pos<64>(bins[wave], entries + b, out + b, n, lane);
    }
  }
  __syncthreads();
  for (uint32_t k = 0; k < (uint32_t)kSortCells && c0 + k < ncell; ++k) {  // workgroup-uniform
    const uint32_t b = cellptr[c0 + k], n = cellptr[c0 + k + 1] - b;
    if (n <= kSortWave) continue;
    if (n > kSortMax)
      for (uint32_t i = tid; i < n; i += kBlock) out[b + i] = entries[b + i];
    else
      cell_sort_pos<kBlock>(bins[0], entries + b, out + b, n, tid);
  }
}

// the same for nonzeros that come with their row number instead of in CSR order (the owner side
// of the owner-compute step: nonzeros of several workers' minibatches, rows numbered window by
// window across the workers)
__global__ void __launch_bounds__(kBlock)
k_cell_keys_rowid(const uint32_t *__restrict__ rowid, const uint32_t *__restrict__ src, size_t n,
                  uint32_t W, uint32_t nchunk, uint32_t chunk0, uint32_t *__restrict__ cid,
                  uint32_t *__restrict__ ent) {
  XF_GRID_STRIDE(j, n) {
    const uint32_t r = rowid[j], v = r / W, rin = r - v * W;
    const uint32_t idx = src[j], chunk = (idx >> kChunkBits) - chunk0;
    cid[j] = v * nchunk + chunk;
    ent[j] = ((chunk & kTagMask) << kTagShift) | (rin << kChunkBits) | (idx & (kChunk - 1));
  }
}

// ------------------------------------------------------------------------------ forward
// the cell of entry j, known to lie in [lo, hi]: the largest c with cellptr[c] <= j
__device__ __forceinline__ uint32_t cell_of(const uint32_t *__restrict__ cellptr, uint32_t lo,
                                            uint32_t hi, uint32_t j) {
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo + 1) / 2;
    if (cellptr[mid] <= j) lo = mid;
    else
      hi = mid - 1;
  }
  return lo;
}

// Workgroup (v, g) takes the g-th of G slices of window v's entries, cut at multiples of kBlk
// in the entry stream.  Its 16 wavefronts pull blocks of kBlk entries (kFwdE per lane) off a
// counter in LDS and keep the NEXT block's index loads in flight while they work on the
// current one: the kernel is a chain of dependent loads — block bounds -> entries -> weights —
// and with the row window in LDS only one workgroup fits a CU, so the memory parallelism has to
// come from inside the wavefront (measured on the config-2 shape: the LDS atomics cost nothing,
// the weight gathers ~26 us, the bare index stream ~20 us at 4 entries per lane).  The
// window's row sums live in LDS as fp64; every entry costs one coalesced index load, one 4-byte
// gather inside the cell's 8 KiB chunk (kChunk = 2048 weights) of the weight array and one LDS
// atomic.  An entry's cell follows from the block's first cell and the 5 chunk-number bits the entry carries; only
// a block that spans 32 or more chunks (a sparse minibatch on a big table) searches cellptr.
// The partial row sums of the G workgroups of a window are added by k_lr_finalize_cells.
constexpr int kFwdE = (int)(kBlk / 64);

struct FwdBlock {
  uint32_t ent[kFwdE];
  uint32_t lo, hi;
};

__device__ __forceinline__ void fwd_load(FwdBlock &B, const uint32_t *__restrict__ entries,
                                         const uint32_t *__restrict__ blk_cell, uint32_t b,
                                         uint32_t pb, uint32_t pe, uint32_t lane) {
  B.lo = blk_cell[b];
  B.hi = blk_cell[b + 1];
#pragma unroll
  for (int q = 0; q < kFwdE; ++q) {
    const uint32_t j = b * kBlk + q * 64 + lane;
    B.ent[q] = (j >= pb && j < pe) ? entries[j] : 0xFFFFFFFFu;
  }
}

__device__ __forceinline__ void fwd_process(const FwdBlock &B, uint32_t b, uint32_t c0,
                                            uint32_t nchunk, uint32_t lane,
                                            const uint32_t *__restrict__ cellptr,
                                            const float *__restrict__ w, double *wx) {
  // the block may begin in the window before and end in the one after
  const uint32_t lo = B.lo < c0 ? c0 : B.lo;
  const uint32_t hi = B.hi >= c0 + nchunk ? c0 + nchunk - 1 : B.hi;
  const bool near = hi - lo <= kTagMask;  // wave-uniform
  float wv[kFwdE];
#pragma unroll
  for (int q = 0; q < kFwdE; ++q) {
    const uint32_t e = B.ent[q];
    if (e == 0xFFFFFFFFu) continue;
    const uint32_t cell = near ? lo + (((e >> kTagShift) - (lo - c0)) & kTagMask)
                               : cell_of(cellptr, lo, hi, b * kBlk + q * 64 + lane);
    wv[q] = w[(size_t)(cell - c0) * kChunk + (e & (kChunk - 1))];
  }
  // (Power-law minibatches: a block that lies in ONE cell — a head key's — holds runs of
  // neighbouring lanes with the same row, whose same-address LDS atomics serialise.  Adding a
  // run up in registers first, a segmented suffix sum over the wavefront, was tried in round 5:
  // the Zipf(1.1) forward went from 48 to 60 us — twelve ds_bpermute per entry slot load the
  // LDS pipe more than the serialised atomics they replace.)
#pragma unroll
  for (int q = 0; q < kFwdE; ++q) {
    const uint32_t e = B.ent[q];
    if (e != 0xFFFFFFFFu) atomicAdd(&wx[(e >> kChunkBits) & kRowMask], (double)wv[q]);
  }
}

__global__ void __launch_bounds__(kFwdBlock)
k_lr_fwd_cells(const uint32_t *__restrict__ entries, const uint32_t *__restrict__ cellptr,
               const uint32_t *__restrict__ blk_cell, uint32_t nchunk, uint32_t W, uint32_t G,
               const float *__restrict__ w, double *__restrict__ partial, int accumulate) {
  __shared__ double wx[kWinMax];
  __shared__ uint32_t next_blk;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  // block b runs on XCD b % 8 (observed placement; a different one only costs speed): group g
  // of window v is block ((g / 8) * nwin + v) * 8 + g % 8
  const uint32_t nwin = gridDim.x / G;
  const uint32_t slot = blockIdx.x >> 3, v = slot % nwin, g = (slot / nwin) * 8 + (blockIdx.x & 7u);
  // (a later segment of the batch's cells starts from the sums of the segments before it)
  double *out = partial + ((size_t)v * G + g) * W;
  for (uint32_t r = tid; r < W; r += kFwdBlock) wx[r] = accumulate ? out[r] : 0.0;
  const uint32_t c0 = v * nchunk;
  const uint32_t wb = cellptr[c0], we = cellptr[c0 + nchunk];
  const uint64_t n = we - wb;
  uint32_t pb = g == 0 ? wb : (uint32_t)((wb + n * g / G) & ~(uint64_t)(kBlk - 1));
  uint32_t pe = g == G - 1 ? we : (uint32_t)((wb + n * (g + 1) / G) & ~(uint64_t)(kBlk - 1));
  if (pb < wb) pb = wb;
  if (pe < pb) pe = pb;
  const uint32_t b_first = pb / kBlk, b_end = pb < pe ? (pe - 1) / kBlk + 1 : b_first;
  if (tid == 0) next_blk = b_first;
  __syncthreads();
  // (one block counter in HBM shared by the window's workgroups instead of a fixed split was
  // tried against the 1.4x spread between the median and the slowest workgroup: 54 -> 67 us,
  // the ticket's round trip to L2 sits in the dependent chain)
  auto grab = [&]() -> uint32_t {  // the wavefront's next block
    uint32_t b = 0;
    if (lane == 0) b = atomicAdd(&next_blk, 1u);
    return (uint32_t)__shfl((int)b, 0);
  };
  // Two block buffers that swap roles (no register copies: a copy of the prefetched block
  // would wait for its loads and undo the prefetch)
  FwdBlock A, B;
  uint32_t ba = grab(), bb = 0;
  if (ba < b_end) fwd_load(A, entries, blk_cell, ba, pb, pe, lane);
  while (ba < b_end) {
    bb = grab();
    if (bb < b_end) fwd_load(B, entries, blk_cell, bb, pb, pe, lane);
    fwd_process(A, ba, c0, nchunk, lane, cellptr, w, wx);
    if (bb >= b_end) break;
    ba = grab();
    if (ba < b_end) fwd_load(A, entries, blk_cell, ba, pb, pe, lane);
    fwd_process(B, bb, c0, nchunk, lane, cellptr, w, wx);
  }
  __syncthreads();
  for (uint32_t r = tid; r < W; r += kFwdBlock) out[r] = wx[r];
}

// loss[r] = sigmoid(sum of the window workgroups' partial sums of row r) - label.  Four lanes
// per row, lane q adding the workgroups g = q mod 4; combined as (s0 + s1) + (s2 + s3): a
// fixed association, the same bits every run.  (Sixteen lanes per row, every lane's loads in
// flight at once, were tried: forward + finalize 38.3 -> 41.4 us — a wavefront then reads four
// rows of sixteen partial arrays, 32 bytes of every line it touches.)
__global__ void __launch_bounds__(kBlock)
k_lr_finalize_cells(const double *__restrict__ partial, const int32_t *__restrict__ labels,
                    uint32_t R, uint32_t W, uint32_t G, float *__restrict__ loss,
                    float *__restrict__ pctr) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = t >> 2, q = t & 3u;
  double a = 0.0;
  if (r < R) {
    const uint32_t v = r / W, rin = r - v * W;
    const double *p = partial + (size_t)v * G * W + rin;
    // four loads in flight per lane (fp64 sums of fp32 addends: exact in any association)
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    uint32_t g = q;
    for (; g + 12 < G; g += 16) {
      a += p[(size_t)g * W];
      a1 += p[(size_t)(g + 4) * W];
      a2 += p[(size_t)(g + 8) * W];
      a3 += p[(size_t)(g + 12) * W];
    }
    for (; g < G; g += 4) a += p[(size_t)g * W];
    a += a1 + (a2 + a3);
  }
  a += __shfl_xor(a, 1);
  a += __shfl_xor(a, 2);
  if (r >= R || q != 0) return;
  const float pr = xf::sigmoid_ref((float)a);  // lr_worker.cc:141, base.h:54-63
  if (pctr) pctr[r] = pr;
  if (loss) loss[r] = pr - (float)labels[r];
}

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

// Timeline of the work items (tools/grad_timeline.py; build with -DXF_GRAD_TIMELINE, never by
// default): wall_clock64 ticks (10 ns) of every workgroup's phases.
#ifdef XF_GRAD_TIMELINE
constexpr int kTlSlots = 8;
__device__ unsigned long long xf_grad_tl[16384 * kTlSlots];
#define GRAD_T(slot)                                                                  \
  do {                                                                                \
    if (threadIdx.x == 0 && blockIdx.x < 16384)                                       \
      xf_grad_tl[blockIdx.x * kTlSlots + (slot)] = wall_clock64();                    \
  } while (0)
#else
#define GRAD_T(slot) do { } while (0)
#endif

template <int OPT, int MODE, bool SRC, bool MULTI = false /* SRC with more than one source */>
__global__ void __launch_bounds__(kBlock, MULTI ? XF_GRAD_MULTI_WAVES : SRC ? 1 : 6)
k_lr_grad_cells(xf::TableDev T, const uint32_t *__restrict__ entries,
                const uint32_t *__restrict__ cellptr, uint32_t nchunk, uint32_t nwin, uint32_t W,
                const uint32_t *__restrict__ item_chunk, const uint32_t *__restrict__ item_slice,
                const uint32_t *__restrict__ item_dump, const float *__restrict__ loss,
                uint32_t R, uint32_t M, float *__restrict__ g_out, double *__restrict__ gsum,
                uint8_t *__restrict__ gtouched, uint32_t nsrc, const uint32_t *__restrict__ src_win,
                const uint32_t *__restrict__ src_rows, uint32_t nsplit,
                const uint32_t *__restrict__ loss_base, uint32_t chunk0,
                const uint8_t *__restrict__ item_done) {
  __shared__ double acc[kChunk];
  __shared__ uint8_t touched[kChunk];
  __shared__ uint32_t cum[kGradWin + 1], sbase[kGradWin];
  __shared__ uint32_t nlist;
  const uint32_t tid = threadIdx.x;
  // the old weight of a step derived from the row's (n, z) instead of read (TableDev::w_of_nz)
  const bool wnz = OPT == XF_OPT_FTRL && MODE == 0 && T.w_of_nz;
  // (several sources: k_lr_grad_multi has run first and says which items it has taken)
  if (SRC && item_done && item_done[blockIdx.x]) return;
  GRAD_T(0);
  if (tid == 0) nlist = 0;
  const uint32_t c = item_chunk[blockIdx.x];
  const uint32_t sl = item_slice[blockIdx.x], s = sl & 0xFFFFu, S = sl >> 16;
  for (uint32_t k = tid; k < kChunk; k += kBlock) {
    acc[k] = 0.0;
    touched[k] = 0;
  }
  // SOURCES (src_win != null, the owner-compute step): the windows are grouped by the worker
  // whose rows they hold; a key's gradient from worker q is sum/R_q and is its own optimizer
  // step, the workers' steps applied in rank order — one accumulate + update phase
  // per worker inside the same pass over the chunk, the state row read and written by the same
  // thread every phase (it stays in L2 in between).  One source: the whole minibatch.
  if constexpr (SRC && MULTI && MODE == 0) {
    // Several sources, an unsplit chunk whose entries fit one round of registers.  Taken source
    // after source (the general loop below) the pass of an owner of 8 workers took 403 us for
    // what one source does in 115: per source and chunk a chain of cell bounds -> entries ->
    // losses, four barriers, and a state row that is loaded right after the previous source's
    // store to it.  Here the loads of ALL sources' entries and losses go out together, the
    // state of every key the chunk touches is loaded ONCE into its thread's registers (a
    // thread owns keys tid, tid + 256, ...), the sources' phases only add in LDS and step
    // registers, and the state is stored once.  Same sums, same order of the steps.
    __shared__ uint8_t wsrc[kGradWin];
    __shared__ uint8_t any[kChunk];
    if (S == 1 && nwin <= kGradWin && nsrc > 1 && !g_out) {  // workgroup-uniform
      for (uint32_t k = tid; k < kChunk; k += kBlock) any[k] = 0;
      __syncthreads();
      if (tid < nwin) {
        const size_t cell = (size_t)tid * nchunk + c;
        const uint32_t b = cellptr[cell], e = cellptr[cell + 1];
        sbase[tid] = b;
        cum[tid + 1] = e - b;
        uint32_t q = 0;
        while (q + 1 < nsrc && tid >= src_win[q + 1]) ++q;
        wsrc[tid] = (uint8_t)q;
      }
      __syncthreads();
      if (tid == 0) {
        cum[0] = 0;
        for (uint32_t v = 0; v < nwin; ++v) cum[v + 1] += cum[v];
      }
      __syncthreads();
      const uint32_t total = cum[nwin];
      if (total <= kBlock * kGradE) {  // workgroup-uniform
        constexpr int kOwn = (int)(kChunk / kBlock);  // keys per thread
        // per entry: the key's place in the chunk and its source in ONE register, the loss in
        // another (the pass lives on its occupancy: every register counts)
        uint32_t ek[kGradE];
        float l[kGradE];
        uint32_t v = 0;
#pragma unroll
        for (int q = 0; q < kGradE; ++q) {
          const uint32_t p = q * kBlock + tid;
          ek[q] = 0xFFFFFFFFu;
          l[q] = 0.0f;
          if (p < total) {
            while (p >= cum[v + 1]) ++v;  // p ascends with q: v never goes back
            const uint32_t e = entries[sbase[v] + (p - cum[v])];
            if (e != 0xFFFFFFFFu) {  // (a hole: the key went to the arrival segment)
              l[q] = loss[(size_t)loss_base[v] + ((e >> kChunkBits) & kRowMask)];
              ek[q] = (e & (kChunk - 1)) | ((uint32_t)wsrc[v] << 16);
              any[e & (kChunk - 1)] = 1;
            }
          }
        }
        __syncthreads();
        const size_t idx0 = (size_t)(chunk0 + c) * kChunk + tid;
        float sw[kOwn], sn[kOwn], sz[kOwn];
#pragma unroll
        for (int i = 0; i < kOwn; ++i) {
          sw[i] = sn[i] = sz[i] = 0.0f;
          if (any[tid + i * kBlock] && idx0 + i * kBlock < M) {
            sw[i] = T.w[idx0 + i * kBlock];
            if (OPT == XF_OPT_FTRL) xf::load_nz(T, idx0 + i * kBlock, sn[i], sz[i]);
          }
        }
        for (uint32_t q = 0; q < nsrc; ++q) {
          // source q's entries are the positions [pb, pe) of the index space: register slots
          // pb / kBlock .. (pe - 1) / kBlock of every thread (one or two of the eight)
          const uint32_t pb = cum[src_win[q]], pe = cum[src_win[q + 1]];
          if (pb == pe) continue;  // workgroup-uniform: nothing of this worker in the chunk
          const int ib = (int)(pb / kBlock), ie = (int)((pe - 1) / kBlock);
#pragma unroll
          for (int i = 0; i < kGradE; ++i)
            if (i >= ib && i <= ie)  // workgroup-uniform
              add_keys(acc, touched, (ek[i] >> 16) == q, ek[i] & (kChunk - 1), (double)l[i]);
          __syncthreads();
          const uint32_t rq = src_rows[q];
#pragma unroll
          for (int i = 0; i < kOwn; ++i) {
            const uint32_t k = tid + i * kBlock;
            if (!touched[k]) continue;
            const double sum = acc[k];
            acc[k] = 0.0;  // the next worker's phase starts from zero
            touched[k] = 0;
            const float g = xf::div_by_rows((float)sum, rq);  // lr_worker.cc:117
            if (OPT == XF_OPT_FTRL)
              xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g, sw[i], sn[i], sz[i]);
            else
              sw[i] = xf::sgd_step(T.lr, g, sw[i]);
          }
          __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < kOwn; ++i)
          if (any[tid + i * kBlock] && idx0 + i * kBlock < M) {
            T.w[idx0 + i * kBlock] = sw[i];
            if (OPT == XF_OPT_FTRL) xf::store_nz(T, idx0 + i * kBlock, sn[i], sz[i]);
          }
        return;
      }
    }
  }
  const uint32_t ns = SRC ? nsrc : 1u;
  for (uint32_t q = 0; q < ns; ++q) {
  const uint32_t wbeg = SRC ? src_win[q] : 0u, wend = SRC ? src_win[q + 1] : nwin;
  const uint32_t Rq = SRC ? src_rows[q] : R;
  if (SRC && wbeg == wend) continue;  // workgroup-uniform
  // The item's share of the chunk's cells as ONE index space: windows v0, v0+1, ... side
  // by side (cum = running entry counts), so that a thread's loads of a round — entries, then
  // the losses they point at — are all in flight together instead of window after window.
  for (uint32_t v0 = wbeg; v0 < wend; v0 += kGradWin) {
    const uint32_t nv = min(wend - v0, kGradWin);
    __syncthreads();
    if (tid < nv) {
      const size_t cell = (size_t)(v0 + tid) * nchunk + c;
      const uint32_t b = cellptr[cell], e = cellptr[cell + 1];
      const uint64_t n = e - b;
      sbase[tid] = b + (uint32_t)(n * s / S);
      cum[tid + 1] = (uint32_t)(n * (s + 1) / S) - (uint32_t)(n * s / S);
    }
    __syncthreads();
    if (tid == 0) {
      cum[0] = 0;
      for (uint32_t v = 0; v < nv; ++v) cum[v + 1] += cum[v];
    }
    __syncthreads();
    const uint32_t total = cum[nv];
    GRAD_T(1);
#ifdef XF_GRAD_TIMELINE
    if (tid == 0 && blockIdx.x < 16384) xf_grad_tl[blockIdx.x * kTlSlots + 7] = total;
#endif
    for (uint32_t p0 = 0; p0 < total; p0 += kBlock * kGradE) {  // workgroup-uniform trip count
      uint32_t ent[kGradE], vq[kGradE];
      float l[kGradE];
      // (the window of position p by a walk: p ascends with q, v never goes back.  A search of
      // selects — log2(windows) LDS reads per entry, the eight entries' searches side by side —
      // was measured in round 5: no different at 32 windows (107 vs 108 us), and the power-law
      // gradient, three windows, 7 us slower with it.)
      uint32_t v = 0;
#pragma unroll
      for (int q = 0; q < kGradE; ++q) {
        const uint32_t p = p0 + q * kBlock + tid;
        ent[q] = 0xFFFFFFFFu;
        vq[q] = 0;
        if (p < total) {
          while (p >= cum[v + 1]) ++v;
          vq[q] = v;
          ent[q] = entries[sbase[v] + (p - cum[v])];
        }
      }
#pragma unroll
      for (int q = 0; q < kGradE; ++q)
        l[q] = ent[q] != 0xFFFFFFFFu
                   ? loss[(loss_base ? (size_t)loss_base[v0 + vq[q]] : (size_t)(v0 + vq[q]) * W) +
                          ((ent[q] >> kChunkBits) & kRowMask)]
                   : 0.0f;
      if constexpr (SRC)  // (the owner's passes: their registers stay where they were)
#pragma unroll
        for (int q = 0; q < kGradE; ++q)
          add_keys(acc, touched, ent[q] != 0xFFFFFFFFu, ent[q] & (kChunk - 1), (double)l[q]);
      else
        add_keys_folded<kGradE>(acc, touched, ent, l);
    }
  }
  GRAD_T(2);
  __syncthreads();
  GRAD_T(3);
  if (S > 1) {  // a slice of a split chunk: into the chunk's accumulators in HBM
    const size_t slot = (size_t)q * nsplit + item_dump[blockIdx.x];
    for (uint32_t k = tid; k < kChunk; k += kBlock)
      if (touched[k]) {
        unsafeAtomicAdd(&gsum[slot * kChunk + k], acc[k]);
        gtouched[slot * kChunk + k] = 1;
        if (SRC) {
          acc[k] = 0.0;
          touched[k] = 0;
        }
      }
    GRAD_T(6);
    continue;
  }
  // eight keys per thread at a time in three sweeps — every state row requested, stepped,
  // stored (k_lr_grad_dense below says why: row after row the stores keep the next row's loads
  // from being issued early, one dependent trip to memory per key)
  const size_t row0 = (size_t)(chunk0 + c) * kChunk;
  if constexpr (SRC) {  // (several phases per chunk: the row-after-row loop keeps the registers
                        // of the multi-source pass where they were)
    for (uint32_t k = tid; k < kChunk; k += kBlock) {
      if (!touched[k]) continue;
      const double sum = acc[k];
      acc[k] = 0.0;  // the next worker's phase starts from zero
      touched[k] = 0;
      if (row0 + k >= M) continue;
      const float g = xf::div_by_rows((float)sum, Rq);  // lr_worker.cc:117
      if (g_out) g_out[row0 + k] = g;
      if (MODE == 0) apply_key(T, OPT, row0 + k, g);
    }
  } else {
  // A chunk of which the minibatch touched few keys (a power-law minibatch touches a tenth of
  // a chunk's keys: the sweeps below would run every optimizer step with a tenth of the lanes):
  // the touched keys are listed (in the bytes of `touched`, once every thread has read its own)
  // and stepped with all lanes busy.  Same sums, same steps.
  if constexpr (kChunk == kBlock * 8) {
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (touched[i * kBlock + tid] && row0 + i * kBlock + tid < M) mine |= 1u << i;
    const uint32_t cnt = (uint32_t)__popc(mine), lane = tid & 63u;
    uint32_t inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o);
      if ((int)lane >= o) inc += u;
    }
    uint32_t base = 0;
    if (lane == 63) base = atomicAdd(&nlist, inc);
    base = (uint32_t)__shfl((int)base, 63);
    __syncthreads();
    const uint32_t n = nlist;
    if (n <= kSparseKeys) {  // workgroup-uniform
      uint16_t *list = (uint16_t *)touched;
      uint32_t pos = base + inc - cnt;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (mine >> i & 1u) list[pos++] = (uint16_t)(i * kBlock + tid);
      __syncthreads();
      GRAD_T(4);
      constexpr int kS = (int)(kSparseKeys / kBlock);
      uint32_t k[kS];
      float g[kS], sw[kS], sn[kS], sz[kS];
#pragma unroll
      for (int i = 0; i < kS; ++i) {
        const uint32_t j = i * kBlock + tid;
        k[i] = j < n ? list[j] : 0xFFFFFFFFu;
        const size_t r = row0 + (j < n ? k[i] : 0u);
        g[i] = j < n ? xf::div_by_rows((float)acc[k[i]], Rq) : 0.0f;  // lr_worker.cc:117
        if (MODE == 0) {
          sw[i] = 0.0f;
          if (!wnz) sw[i] = T.w[r];
          sn[i] = sz[i] = 0.0f;
          if (OPT == XF_OPT_FTRL) xf::load_nz(T, r, sn[i], sz[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < kS; ++i) {
        if (k[i] == 0xFFFFFFFFu) continue;
        if (g_out) g_out[row0 + k[i]] = g[i];
        if (MODE == 0) {
          if (wnz) sw[i] = xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, sn[i], sz[i]);
          if (OPT == XF_OPT_FTRL)
            xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g[i], sw[i], sn[i], sz[i]);
          else
            sw[i] = xf::sgd_step(T.lr, g[i], sw[i]);
        }
      }
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < kS; ++i) {
          if (k[i] == 0xFFFFFFFFu) continue;
          T.w[row0 + k[i]] = sw[i];
          if (OPT == XF_OPT_FTRL) xf::store_nz(T, row0 + k[i], sn[i], sz[i]);
        }
      }
      GRAD_T(5);
      return;
    }
  }
  for (uint32_t kb = 0; kb < kChunk; kb += kBlock * 8) {
    bool t[8];
    float g[8], sw[8], sn[8], sz[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t k = kb + i * kBlock + tid;
      t[i] = k < kChunk && touched[k] != 0;
      g[i] = 0.0f;
      if (t[i]) {
        g[i] = xf::div_by_rows((float)acc[k], Rq);  // lr_worker.cc:117
        t[i] = row0 + k < M;
      }
      if (MODE == 0) {  // (an idle slot loads the chunk's first row: no load under a branch)
        const size_t r = t[i] ? row0 + k : row0;
        sw[i] = 0.0f;
        if (!wnz) sw[i] = T.w[r];
        sn[i] = sz[i] = 0.0f;
        if (OPT == XF_OPT_FTRL) xf::load_nz(T, r, sn[i], sz[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (!t[i]) continue;
      if (g_out) g_out[row0 + kb + i * kBlock + tid] = g[i];
      if (MODE == 0) {
        if (wnz) sw[i] = xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, sn[i], sz[i]);
        if (OPT == XF_OPT_FTRL)
          xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g[i], sw[i], sn[i], sz[i]);
        else
          sw[i] = xf::sgd_step(T.lr, g[i], sw[i]);
      }
    }
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (!t[i]) continue;
        T.w[row0 + kb + i * kBlock + tid] = sw[i];
        if (OPT == XF_OPT_FTRL) xf::store_nz(T, row0 + kb + i * kBlock + tid, sn[i], sz[i]);
      }
    }
  }
  GRAD_T(6);
  }
  }  // sources
}

#ifdef XF_GRAD_TIMELINE
}  // namespace
extern "C" int xf_debug_grad_timeline(unsigned long long *out, size_t n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(xf_grad_tl), n * 8) == hipSuccess ? 0 : -1;
}
namespace {
#endif

// ---- the gradient + Pushes of an owner of SEVERAL workers (XF_UPDATE_RANK_ORDERED on the
// owner-compute dataflow): every worker's gradient sum / R_q is its own optimizer step, the steps
// of a key applied in rank order (lr_worker.cc:116-118 + ftrl.h:54-74 once per Push).
//
// Round 4's pass (k_lr_grad_cells<.., SRC, MULTI>) kept a thread's eight state rows in
// registers and ran, per worker, a sweep over them: 8 workers x 8 rows = 64 optimizer steps per
// thread, each issued for the whole wavefront although a worker touches a tenth of a chunk's
// keys — ~5 400 VALU instructions per wavefront and chunk, 108 us of issue time at the N = 8
// shard shape: the pass was VALU-bound on steps that 90 % of the lanes sat out.
//
// Here the chunk's state lives in LDS for the time of the pass (w 8 KiB, {n, z} 16 KiB: loaded
// and stored once, coalesced, whole lines), so ANY lane can step ANY key, and the lane that
// steps key k for worker q is one of the lanes that hold an entry (k, q): after a worker's
// entries have been added to the key sums, every such lane reads back the mark its key carries
// (the last writer's position) and the one whose position it is takes the step.  A worker's
// entries are neighbours in the index space, so its ~200 steps per chunk run in three or four
// wavefronts with nearly every lane busy; the other wavefronts skip the phase.  Same sums (fp64
// LDS atomics: exact, any order), same steps in the same order per key: the bits of the general
// loop (tests/test_gpu_sharded.py, world 2 / 3 / 8).
//
// Takes the unsplit chunks whose entries fit one round of registers (NT x E = 2048) and that
// span at most kMultiWin row windows; item_done[item] says which, the general kernel
// (k_lr_grad_cells<.., SRC>) is launched behind it for the others.
constexpr uint32_t kMultiWin = 64;
constexpr uint32_t kMultiSrc = 64;
constexpr uint32_t kMultiAny = 0x8000u;  // mark: the key has been touched by some worker
constexpr uint32_t kMultiCap = 512;      // SLOTS: entries of one worker in a chunk

// SLOTS: the key sums of a phase live in kMultiCap accumulators indexed by the STEPPING lane's
// position among the worker's entries instead of 2048 indexed by key (4 KiB of LDS instead of
// 16: four workgroups per CU instead of three).  A phase is then mark -> barrier -> add to the
// stepping lane's slot -> barrier -> step -> barrier: one barrier more.  (Tried: two mark arrays
// that swap roles, the next worker's marks written in the interval of this worker's steps — two
// barriers per phase instead of three: 142 -> 175 us at the N = 8 shard shape, 72 registers
// instead of 64 and the marks' stores in the way of the steps' LDS traffic.)
template <int OPT, int NT, bool SLOTS = false>
__global__ void __launch_bounds__(NT)
k_lr_grad_multi(xf::TableDev T, const uint32_t *__restrict__ entries,
                const uint32_t *__restrict__ cellptr, uint32_t nchunk, uint32_t nwin,
                const uint32_t *__restrict__ item_chunk, const uint32_t *__restrict__ item_slice,
                const float *__restrict__ loss, uint32_t M, uint32_t nsrc,
                const uint32_t *__restrict__ src_win, const uint32_t *__restrict__ src_rows,
                const uint32_t *__restrict__ loss_base, uint32_t chunk0, int full_store,
                uint8_t *__restrict__ item_done) {
  constexpr int E = (int)(kChunk / NT);  // entries per lane = state rows per thread
  constexpr bool FTRL = OPT == XF_OPT_FTRL;
  __shared__ double acc[SLOTS ? kMultiCap : kChunk];
  __shared__ uint32_t over;
  __shared__ float sw[kChunk];
  __shared__ float2 snz[FTRL ? kChunk : 1];
  __shared__ uint16_t mark[kChunk];
  __shared__ uint32_t cum[kMultiWin + 1], sbase[kMultiWin];
  __shared__ uint32_t spos[kMultiSrc + 1], srow[kMultiSrc];
  __shared__ uint8_t wsrc[kMultiWin];
  const uint32_t tid = threadIdx.x;
  const bool wnz = FTRL && T.w_of_nz;  // the rows' w derived from their (n, z), not read
  const uint32_t c = item_chunk[blockIdx.x];
  const uint32_t S = item_slice[blockIdx.x] >> 16;
  if (S != 1 || nwin > kMultiWin || nsrc > kMultiSrc) {  // workgroup-uniform
    if (tid == 0) item_done[blockIdx.x] = 0;
    return;
  }
  // the chunk's state rows: requested now, written to LDS below (rows past M: a table whose
  // last chunk is not full)
  const size_t row0 = (size_t)(chunk0 + c) * kChunk;
  float rw[E];
  float2 rnz[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const size_t r = row0 + tid + i * NT < M ? row0 + tid + i * NT : row0;
    rw[i] = 0.0f;
    if (!wnz) rw[i] = T.w[r];
    rnz[i] = make_float2(0.0f, 0.0f);
    if (FTRL) rnz[i] = T.nz[r];
  }
  if (tid < nwin) {
    const size_t cell = (size_t)tid * nchunk + c;
    const uint32_t b = cellptr[cell], e = cellptr[cell + 1];
    sbase[tid] = b;
    cum[tid + 1] = e - b;
    uint32_t q = 0;
    while (q + 1 < nsrc && tid >= src_win[q + 1]) ++q;
    wsrc[tid] = (uint8_t)q;
  }
  if (tid < nsrc) srow[tid] = src_rows[tid];
  if (tid == 0) over = 0;
  __syncthreads();
  if (tid < 64) {  // cum = inclusive scan of the windows' entry counts (one wavefront)
    uint32_t inc = tid < nwin ? cum[tid + 1] : 0u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o);
      if ((int)tid >= o) inc += u;
    }
    if (tid < nwin) cum[tid + 1] = inc;
    if (tid == 0) cum[0] = 0;
  }
  __syncthreads();
  const uint32_t total = cum[nwin];
  if (total > (uint32_t)(NT * E)) {  // workgroup-uniform: the general kernel's
    if (tid == 0) item_done[blockIdx.x] = 0;
    return;
  }
  if (tid <= nsrc) spos[tid] = cum[src_win[tid]];  // worker q's entries: positions [spos[q], spos[q+1])
  if (SLOTS) {  // every worker's entries must fit the slots (else: the general kernel's)
    if (tid < nsrc && cum[src_win[tid + 1]] - cum[src_win[tid]] > kMultiCap) over = 1;
    __syncthreads();
    if (over) {  // workgroup-uniform
      if (tid == 0) item_done[blockIdx.x] = 0;
      return;
    }
  }
  if (tid == 0) item_done[blockIdx.x] = 1;
  // per entry: the key's place in the chunk and its worker in ONE register, the loss in another
  uint32_t ek[E];
  float l[E];
  if (total) {
    uint32_t vq[E], en[E];
    uint32_t st0 = 1;
    while (st0 * 2 < nwin) st0 *= 2;
#pragma unroll
    for (int q = 0; q < E; ++q) {  // (the window of a position: see k_lr_grad_cells)
      const uint32_t p = q * NT + tid;
      const uint32_t pc = min(p, total - 1);
      uint32_t v = 0;
      for (uint32_t st = st0; st > 0; st >>= 1) {  // wave-uniform trip count
        const uint32_t t = v + st;
        v = (t < nwin && cum[t] <= pc) ? t : v;
      }
      vq[q] = v;
      uint32_t e = 0xFFFFFFFFu;
      if (p < total) e = entries[sbase[v] + (pc - cum[v])];
      en[q] = e;
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const bool on = en[q] != 0xFFFFFFFFu;  // (a hole: the key went to the arrival segment)
      float x = 0.0f;
      if (on) x = loss[(size_t)loss_base[vq[q]] + ((en[q] >> kChunkBits) & kRowMask)];
      l[q] = x;
      ek[q] = on ? (en[q] & (kChunk - 1)) | ((uint32_t)wsrc[vq[q]] << 16) : 0xFFFFFFFFu;
    }
  } else {
#pragma unroll
    for (int q = 0; q < E; ++q) {
      ek[q] = 0xFFFFFFFFu;
      l[q] = 0.0f;
    }
  }
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const uint32_t k = tid + i * NT;
    if (!SLOTS) acc[k] = 0.0;
    mark[k] = 0;
    sw[k] = wnz ? xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, rnz[i].x, rnz[i].y)
                : rw[i];
    if (FTRL) snz[k] = rnz[i];
  }
  if (SLOTS)
    for (uint32_t k = tid; k < kMultiCap; k += NT) acc[k] = 0.0;
  __syncthreads();
  for (uint32_t q = 0; q < nsrc; ++q) {
    const uint32_t pb = spos[q], pe = spos[q + 1];
    if (pb == pe) continue;  // workgroup-uniform: nothing of this worker in the chunk
    const int ib = (int)(pb / NT), ie = (int)((pe - 1) / NT);
#pragma unroll
    for (int i = 0; i < E; ++i)
      if (i >= ib && i <= ie && (ek[i] >> 16) == q) {  // (a hole's worker is 0xFFFF)
        const uint32_t k = ek[i] & (kChunk - 1);
        if (!SLOTS) atomicAdd(&acc[k], (double)l[i]);
        mark[k] = (uint16_t)(kMultiAny | (uint32_t)(i * NT + tid));
      }
    __syncthreads();
    if (SLOTS) {  // the sums where the stepping lanes will look for them
#pragma unroll
      for (int i = 0; i < E; ++i)
        if (i >= ib && i <= ie && (ek[i] >> 16) == q) {
          const uint32_t k = ek[i] & (kChunk - 1);
          atomicAdd(&acc[(mark[k] & (kMultiAny - 1u)) - pb], (double)l[i]);
        }
      __syncthreads();
    }
    const uint32_t rq = srow[q];
#pragma unroll
    for (int i = 0; i < E; ++i)
      if (i >= ib && i <= ie && (ek[i] >> 16) == q) {
        const uint32_t k = ek[i] & (kChunk - 1);
        if ((mark[k] & (kMultiAny - 1u)) != (uint32_t)(i * NT + tid)) continue;  // not its stepper
        const uint32_t slot = SLOTS ? (uint32_t)(i * NT + tid) - pb : k;
        const double sum = acc[slot];
        acc[slot] = 0.0;  // the next worker's phase starts from zero
        const float g = xf::div_by_rows((float)sum, rq);  // lr_worker.cc:117
        float w = sw[k];
        if (FTRL) {
          float2 s2 = snz[k];
          xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g, w, s2.x, s2.y);
          snz[k] = s2;
        } else {
          w = xf::sgd_step(T.lr, g, w);
        }
        sw[k] = w;
      }
    __syncthreads();
  }
  // back to the table: every row of the chunk (whole lines; an untouched row gets the bits it
  // had), or the touched ones only when the minibatch touches the table thinly
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const uint32_t k = tid + i * NT;
    if (row0 + k >= M) continue;
    if (!full_store && !(mark[k] & kMultiAny)) continue;
    T.w[row0 + k] = sw[k];
    if (FTRL) T.nz[row0 + k] = snz[k];
  }
}

// ---- the same pass with the workers' phases merged (round 5, second version).  The kernel above
// takes a worker at a time — mark, add, step, three barriers each — and a phase costs ~5 us of a
// workgroup's life whatever it holds (measured: 101 us + 5 us per worker at the N = 8 shard
// shape): a chain of LDS round trips and one optimizer step's ~140 dependent instructions, with
// half the wavefronts waiting at the barrier.  But only the steps of ONE key have an order; here:
//   * every entry ORs its worker into the key's mask (LDS atomic);
//   * one scan over the chunk's keys numbers the (key, worker) pairs — a key's pairs are
//     neighbours, in rank order — and lists the touched keys and the keys of several workers;
//   * every entry adds its loss to its pair's sum (fp64 LDS atomic: exact, any order);
//   * all touched keys take their FIRST worker's step at once, a lane per key off the list
//     (full wavefronts: a step is ~140 instructions, idle lanes were what made round 4's pass
//     VALU-bound); then the keys of several workers — a quarter of the touched ones at N = 8 —
//     take their remaining steps, a lane per key, in rank order, no barrier in between.
// Nine barriers per chunk instead of 3 x workers + 3; the same sums, the same steps in the same
// order per key: the bits of the general loop (tests/test_gpu_sharded.py).  MB = width of a
// key's mask: 8 (packed four to a word, 52 KB of LDS: three workgroups per CU) or 32.
// Measured (tools/r5/call18.sh, N = 8 shard shape, 2 / 4 / 8 / 16 pretended workers): 134 / 141 /
// 143 / 186 us against the kernel above's 185 (over its slots: the general kernel) / 123 / 141 /
// 184 — the scan, the second pass over the entries and a third workgroup less per CU cost what
// eight phases cost; it is the pass for two or three workers, whose entries per chunk do not fit
// the other kernel's slots (launch_grad decides).
template <int OPT, int MB>
__global__ void __launch_bounds__(512)
k_lr_grad_ranked(xf::TableDev T, const uint32_t *__restrict__ entries,
                 const uint32_t *__restrict__ cellptr, uint32_t nchunk, uint32_t nwin,
                 const uint32_t *__restrict__ item_chunk, const uint32_t *__restrict__ item_slice,
                 const float *__restrict__ loss, uint32_t M, uint32_t nsrc,
                 const uint32_t *__restrict__ src_win, const uint32_t *__restrict__ src_rows,
                 const uint32_t *__restrict__ loss_base, uint32_t chunk0, int full_store,
                 uint8_t *__restrict__ item_done) {
  constexpr int NT = 512;
  constexpr int E = (int)(kChunk / NT);  // entries per lane = state rows per thread = keys it scans
  constexpr bool FTRL = OPT == XF_OPT_FTRL;
  static_assert(MB == 8 || MB == 32, "mask width");
  static_assert(E == 4, "the scan below gives a thread four neighbouring keys");
  __shared__ double acc[kChunk];                       // the pairs' sums
  __shared__ float sw[kChunk];
  __shared__ float2 snz[FTRL ? kChunk : 1];
  __shared__ uint32_t wm[MB == 8 ? kChunk / 4 : kChunk];  // per key: the workers that touch it
  __shared__ uint16_t base[kChunk];                    // per key: its first pair
  __shared__ uint16_t lists[kChunk];  // touched keys from the bottom, keys of several workers from the top
  __shared__ uint32_t cum[kMultiWin + 1], sbase[kMultiWin];
  __shared__ uint32_t srow[kMultiSrc];
  __shared__ uint8_t wsrc[kMultiWin];
  __shared__ unsigned long long wtot[NT / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const bool wnz = FTRL && T.w_of_nz;  // the rows' w derived from their (n, z), not read
  const uint32_t c = item_chunk[blockIdx.x];
  const uint32_t S = item_slice[blockIdx.x] >> 16;
  if (S != 1 || nwin > kMultiWin || nsrc > (uint32_t)MB) {  // workgroup-uniform
    if (tid == 0) item_done[blockIdx.x] = 0;
    return;
  }
  auto mask_of = [&](uint32_t k) -> uint32_t {
    return MB == 8 ? (wm[k >> 2] >> ((k & 3u) * 8u)) & 0xFFu : wm[k];
  };
  const size_t row0 = (size_t)(chunk0 + c) * kChunk;
  float rw[E];
  float2 rnz[E];
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const size_t r = row0 + tid + i * NT < M ? row0 + tid + i * NT : row0;
    rw[i] = 0.0f;
    if (!wnz) rw[i] = T.w[r];
    rnz[i] = make_float2(0.0f, 0.0f);
    if (FTRL) rnz[i] = T.nz[r];
  }
  if (tid < nwin) {
    const size_t cell = (size_t)tid * nchunk + c;
    const uint32_t b = cellptr[cell], e = cellptr[cell + 1];
    sbase[tid] = b;
    cum[tid + 1] = e - b;
    uint32_t q = 0;
    while (q + 1 < nsrc && tid >= src_win[q + 1]) ++q;
    wsrc[tid] = (uint8_t)q;
  }
  if (tid < nsrc) srow[tid] = src_rows[tid];
  __syncthreads();
  if (tid < 64) {  // cum = inclusive scan of the windows' entry counts (one wavefront)
    uint32_t inc = tid < nwin ? cum[tid + 1] : 0u;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t u = __shfl_up(inc, o);
      if ((int)tid >= o) inc += u;
    }
    if (tid < nwin) cum[tid + 1] = inc;
    if (tid == 0) cum[0] = 0;
  }
  __syncthreads();
  const uint32_t total = cum[nwin];
  if (total > (uint32_t)(NT * E)) {  // workgroup-uniform: the general kernel's
    if (tid == 0) item_done[blockIdx.x] = 0;
    return;
  }
  if (tid == 0) item_done[blockIdx.x] = 1;
  // per entry: the key's place in the chunk and its worker in ONE register, the loss in another
  uint32_t ek[E];
  float l[E];
  if (total) {
    uint32_t vq[E], en[E];
    uint32_t st0 = 1;
    while (st0 * 2 < nwin) st0 *= 2;
#pragma unroll
    for (int q = 0; q < E; ++q) {  // (the window of a position: see k_lr_grad_multi)
      const uint32_t p = q * NT + tid;
      const uint32_t pc = min(p, total - 1);
      uint32_t v = 0;
      for (uint32_t st = st0; st > 0; st >>= 1) {  // wave-uniform trip count
        const uint32_t t = v + st;
        v = (t < nwin && cum[t] <= pc) ? t : v;
      }
      vq[q] = v;
      uint32_t e = 0xFFFFFFFFu;
      if (p < total) e = entries[sbase[v] + (pc - cum[v])];
      en[q] = e;
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const bool on = en[q] != 0xFFFFFFFFu;  // (a hole: the key went to the arrival segment)
      float x = 0.0f;
      if (on) x = loss[(size_t)loss_base[vq[q]] + ((en[q] >> kChunkBits) & kRowMask)];
      l[q] = x;
      ek[q] = on ? (en[q] & (kChunk - 1)) | ((uint32_t)wsrc[vq[q]] << 16) : 0xFFFFFFFFu;
    }
  } else {
#pragma unroll
    for (int q = 0; q < E; ++q) {
      ek[q] = 0xFFFFFFFFu;
      l[q] = 0.0f;
    }
  }
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const uint32_t k = tid + i * NT;
    acc[k] = 0.0;
    if (MB == 32) wm[k] = 0;
    sw[k] = wnz ? xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, rnz[i].x, rnz[i].y)
                : rw[i];
    if (FTRL) snz[k] = rnz[i];
  }
  if (MB == 8) wm[tid] = 0;  // (kChunk / 4 == NT words)
  __syncthreads();
#pragma unroll
  for (int i = 0; i < E; ++i)
    if (ek[i] != 0xFFFFFFFFu) {
      const uint32_t k = ek[i] & (kChunk - 1), q = ek[i] >> 16;
      if (MB == 8) atomicOr(&wm[k >> 2], (1u << q) << ((k & 3u) * 8u));
      else
        atomicOr(&wm[k], 1u << q);
    }
  __syncthreads();
  // the scan: thread t takes keys 4t .. 4t + 3; pairs | touched keys << 16 | keys of several
  // workers << 32 in one 64-bit number (each count is at most 2048)
  uint32_t km[E];
  unsigned long long mine = 0;
#pragma unroll
  for (int j = 0; j < E; ++j) {
    km[j] = mask_of(tid * E + j);
    const uint32_t pc = (uint32_t)__popc(km[j]);
    mine += (unsigned long long)pc | ((unsigned long long)(pc > 0) << 16) |
            ((unsigned long long)(pc > 1) << 32);
  }
  unsigned long long inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned long long u = __shfl_up(inc, o);
    if ((int)lane >= o) inc += u;
  }
  if (lane == 63) wtot[wave] = inc;
  __syncthreads();
  unsigned long long before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const unsigned long long x = wtot[w];
    if (w < (int)wave) before += x;
    all += x;
  }
  {
    unsigned long long run = before + inc - mine;
#pragma unroll
    for (int j = 0; j < E; ++j) {
      const uint32_t k = tid * E + j, pc = (uint32_t)__popc(km[j]);
      base[k] = (uint16_t)(run & 0xFFFFu);
      if (pc > 0) lists[(run >> 16) & 0xFFFFu] = (uint16_t)k;
      if (pc > 1) lists[kChunk - 1u - (uint32_t)((run >> 32) & 0xFFFFu)] = (uint16_t)k;
      run += (unsigned long long)pc | ((unsigned long long)(pc > 0) << 16) |
             ((unsigned long long)(pc > 1) << 32);
    }
  }
  const uint32_t ntouched = (uint32_t)((all >> 16) & 0xFFFFu), nmulti = (uint32_t)((all >> 32) & 0xFFFFu);
  __syncthreads();
  // the pairs' sums
#pragma unroll
  for (int i = 0; i < E; ++i)
    if (ek[i] != 0xFFFFFFFFu) {
      const uint32_t k = ek[i] & (kChunk - 1), q = ek[i] >> 16;
      const uint32_t m = mask_of(k);
      atomicAdd(&acc[(uint32_t)base[k] + (uint32_t)__popc(m & ((1u << q) - 1u))], (double)l[i]);
    }
  __syncthreads();
  auto step_key = [&](uint32_t k, uint32_t q, uint32_t pair, float &w, float2 &s2) {
    const float g = xf::div_by_rows((float)acc[pair], srow[q]);  // lr_worker.cc:117
    if (FTRL) xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g, w, s2.x, s2.y);
    else
      w = xf::sgd_step(T.lr, g, w);
  };
  // every touched key: its first worker's step
  for (uint32_t idx = tid; idx < ntouched; idx += NT) {
    const uint32_t k = lists[idx];
    const uint32_t m = mask_of(k);
    float w = sw[k];
    float2 s2 = make_float2(0.0f, 0.0f);
    if (FTRL) s2 = snz[k];
    step_key(k, (uint32_t)__ffs((int)m) - 1u, base[k], w, s2);
    sw[k] = w;
    if (FTRL) snz[k] = s2;
  }
  if (nmulti) {  // workgroup-uniform
    __syncthreads();
    // the keys of several workers: the other workers' steps, in rank order, a lane per key
    for (uint32_t idx = tid; idx < nmulti; idx += NT) {
      const uint32_t k = lists[kChunk - 1u - idx];
      uint32_t m = mask_of(k);
      m &= m - 1u;  // (the first worker has stepped)
      float w = sw[k];
      float2 s2 = make_float2(0.0f, 0.0f);
      if (FTRL) s2 = snz[k];
      uint32_t pair = (uint32_t)base[k] + 1u;
      while (m) {
        step_key(k, (uint32_t)__ffs((int)m) - 1u, pair, w, s2);
        m &= m - 1u;
        ++pair;
      }
      sw[k] = w;
      if (FTRL) snz[k] = s2;
    }
  }
  __syncthreads();
  // back to the table: every row of the chunk (whole lines; an untouched row gets the bits it
  // had), or the touched ones only when the minibatch touches the table thinly
#pragma unroll
  for (int i = 0; i < E; ++i) {
    const uint32_t k = tid + i * NT;
    if (row0 + k >= M) continue;
    if (!full_store && !mask_of(k)) continue;
    T.w[row0 + k] = sw[k];
    if (FTRL) T.nz[row0 + k] = snz[k];
  }
}

// ---- the gradient + Push of the steady state: ONE source, no split chunk, at most kDenseWin
// row windows (config 2: three) — the launch that takes most of the LR step.  What the general
// kernel above spends there (ISA count, 58 registers): ~140 VALU instructions per optimizer step
// (two correctly rounded square roots, three correctly rounded divisions) + an fp64 division for
// sum / R, executed for EVERY 64 rows of the chunk with the ~63 % of the lanes whose key the
// minibatch touched — ~760 cycles x 156 000 wavefront iterations / 1024 SIMDs ~ 48 us of pure
// issue time in a 78 us kernel; the accumulate phase another ~13 us (window search in LDS per
// entry, 64-bit address arithmetic).  The kernel is VALU-bound, not HBM-bound.  Here:
//   * g = sum / R as ONE fp32 division (div_by_rows: the same number as the reference's double
//     division for R < 2^24);
//   * kDenseCompact: after the sums are complete every wavefront compacts the touched keys of
//     its share of the chunk into a list in LDS and steps them with all lanes busy;
//   * the cell bounds of the <= 4 windows in registers (scalar loads): no LDS table, no serial
//     scan, two barriers fewer, window of an entry = two compares;
//   * kDensePrefetch (instead of the compaction): the chunk's state rows requested before the
//     entries, so that the optimizer steps find them in registers;
//   * kDenseWide: 512 threads per chunk.
// Same sums (fp64 LDS atomics: exact, any order), same step (ftrl_step / sgd_step): the table
// bits of the general kernel (tests/test_gpu_cells.py runs every variant against it).
constexpr uint32_t kDenseWin = 4;
enum { kDenseCompact = 1, kDensePrefetch = 2, kDenseWide = 4,
       // timing experiments only (WRONG results; reachable through exp_knob, never by default):
       kDiagNoStore = 8,    // the optimizer steps run, nothing is stored
       kDiagNoUpdate = 16,  // accumulate phase alone
       kDiagNoAccum = 32,   // update phase alone (every row of the chunk takes a step, g = 0)
       kDiagCopy = 64,      // with kDiagNoAccum: the state is loaded and stored, no arithmetic
       kDenseFullStore = 128,    // every row of the chunk is stored back, touched or not (whole
                                 // lines instead of byte-masked ones; same table afterwards)
       kDenseQuad = 256 };       // four chunks per workgroup (see the kernel)
#ifndef XF_GRAD_DENSE_VAR
#define XF_GRAD_DENSE_VAR 128
#endif

// a chunk's team: 8 keys per thread (kDenseWide: 4), at most 1024 threads
constexpr int dense_team(int var) {
  return (int)kChunk / ((var & kDenseWide) ? 4 : 8) > 1024 ? 1024
         : (int)kChunk / ((var & kDenseWide) ? 4 : 8) < 64 ? 64
                                                            : (int)kChunk / ((var & kDenseWide) ? 4 : 8);
}
constexpr int dense_threads(int var) {
  return dense_team(var) * ((var & kDenseQuad) ? 4 : 1) > 1024 ? 1024
                                                               : dense_team(var) * ((var & kDenseQuad) ? 4 : 1);
}

template <int OPT, int VAR>
__global__ void __launch_bounds__(dense_threads(VAR),
                                  (VAR & kDenseCompact) ? 1 : 6 /* <= 80 registers */)
k_lr_grad_dense(xf::TableDev T, const uint32_t *__restrict__ entries,
                const uint32_t *__restrict__ cellptr, uint32_t nchunk, uint32_t nwin, uint32_t W,
                const uint32_t *__restrict__ item_chunk, const float *__restrict__ loss,
                uint32_t R, uint32_t M, uint32_t chunk0, uint32_t nitems) {
  constexpr int NT = dense_team(VAR);                 // threads of a chunk's team
  constexpr int SUB = (VAR & kDenseQuad) && NT * 4 <= 1024 ? 4 : 1;  // chunks per workgroup
  constexpr int kOwn = (int)(kChunk / NT);  // keys per thread = entries per lane and round
  constexpr int NW = NT / 64;
  constexpr uint32_t KW = kChunk / NW;      // keys per wavefront in the compaction
  constexpr bool COMPACT = (VAR & kDenseCompact) != 0;
  constexpr bool PREFETCH = !COMPACT && (VAR & kDensePrefetch) != 0;
  __shared__ double acc_all[SUB * kChunk];
  __shared__ uint8_t touched_all[SUB * kChunk];
  __shared__ uint16_t list_all[COMPACT ? SUB * kChunk : 1];
  // kDenseQuad: four teams, four consecutive chunks, one workgroup.  The teams walk the row
  // windows' losses side by side, so a line of losses that one team's gather brings into the
  // CU's L1 serves the other three (the L1's misses in flight x their latency is what bounds
  // this kernel: 6.0 M read requests to L2 per launch, 4.4 M of them loss gathers).
  const uint32_t team = SUB > 1 ? threadIdx.x / NT : 0u, tid = SUB > 1 ? threadIdx.x % NT : threadIdx.x;
  double *acc = acc_all + team * kChunk;
  uint8_t *touched = touched_all + team * kChunk;
  uint16_t *list = list_all + (COMPACT ? team * kChunk : 0);
  const uint32_t item = blockIdx.x * SUB + team;
  const bool live = item < nitems;  // (the last workgroup's spare teams only keep the barriers)
  const uint32_t c = item_chunk[live ? item : blockIdx.x * SUB];  // (no chunk is split)
  const size_t row0 = (size_t)(chunk0 + c) * kChunk;
  if (!live) M = 0;  // no row of a spare team is in range: nothing accumulated, nothing stored
  // the old weight of a step derived from the row's (n, z) instead of read (TableDev::w_of_nz):
  // 40 of the launch's 284 MB at the config-2 shape
  const bool wnz = OPT == XF_OPT_FTRL && T.w_of_nz;
  float sw[kOwn], sn[kOwn], sz[kOwn];
  if constexpr (PREFETCH) {
#pragma unroll
    for (int i = 0; i < kOwn; ++i) {
      const size_t r = row0 + tid + i * NT < M ? row0 + tid + i * NT : row0;
      sw[i] = 0.0f;
      if (!wnz) sw[i] = T.w[r];
      sn[i] = sz[i] = 0.0f;
      if (OPT == XF_OPT_FTRL) xf::load_nz(T, r, sn[i], sz[i]);
    }
  }
  // the chunk's cells: window v holds entries [cb[v], cb[v] + (cum[v + 1] - cum[v]))
  uint32_t cb0 = 0, cb1 = 0, cb2 = 0, cb3 = 0, c1 = 0, c2 = 0, c3 = 0, total = 0;  // NOLINT
  {
    uint32_t b, e;
    b = cellptr[c], e = cellptr[c + 1], cb0 = b, c1 = e - b, c2 = c3 = total = c1;
    if (nwin > 1) b = cellptr[(size_t)nchunk + c], e = cellptr[(size_t)nchunk + c + 1], cb1 = b,
                  c2 = c1 + (e - b), c3 = total = c2;
    if (nwin > 2) b = cellptr[2 * (size_t)nchunk + c], e = cellptr[2 * (size_t)nchunk + c + 1],
                  cb2 = b, c3 = c2 + (e - b), total = c3;
    if (nwin > 3) b = cellptr[3 * (size_t)nchunk + c], e = cellptr[3 * (size_t)nchunk + c + 1],
                  cb3 = b, total = c3 + (e - b);
  }
#pragma unroll
  for (int i = 0; i < kOwn; ++i) acc[tid + i * NT] = 0.0;
  if (tid < kChunk / 4) ((uint32_t *)touched)[tid] = 0u;
  if (NT < (int)(kChunk / 4) && tid + NT < kChunk / 4) ((uint32_t *)touched)[tid + NT] = 0u;
  __syncthreads();
  if (!live) total = 0;
  if constexpr (VAR & kDiagNoAccum) {
    total = 0;
    for (uint32_t k = tid; k < kChunk; k += NT) touched[k] = 1;
  }
  // Entry p of the chunk's index space sits at entries[p + d], d = the offset of p's window
  // (wave-uniform numbers).  Selects, not branches: written with `if (p < total)` around nested
  // ?: the compiler built a tree of divergent branches — ~75 instructions, a dozen s_cbranch
  // among them, per entry loaded (round 4's "~45 instructions per entry").  Only the loads stay
  // under the lane's mask: a chunk's last round is mostly idle lanes (2087 entries: 39 in the
  // second round), and with every lane loading — the same address, dropped afterwards — the
  // kernel took 1.5 us more, the power-law gradient 7 (a lane's load costs its TA cycle whatever
  // it hits).
  const uint32_t d0 = cb0, d1 = cb1 - c1, d2 = cb2 - c2, d3 = cb3 - c3;
  for (uint32_t p0 = 0; p0 < total; p0 += NT * kOwn) {  // workgroup-uniform trip count
    uint32_t ent[kOwn], lidx[kOwn];
    float l[kOwn];
#pragma unroll
    for (int q = 0; q < kOwn; ++q) {
      const uint32_t p = p0 + q * NT + tid;
      const uint32_t pc = min(p, total - 1);
      uint32_t d = d0, li = 0;
      d = pc >= c1 ? d1 : d;
      li = pc >= c1 ? W : li;
      d = pc >= c2 ? d2 : d;
      li = pc >= c2 ? 2u * W : li;
      d = pc >= c3 ? d3 : d;
      li = pc >= c3 ? 3u * W : li;
      uint32_t e = 0xFFFFFFFFu;
      if (p < total) e = entries[pc + d];  // (the load alone under the lane's mask)
      ent[q] = e;
      lidx[q] = li;
    }
#pragma unroll
    for (int q = 0; q < kOwn; ++q) {
      float x = 0.0f;
      if (ent[q] != 0xFFFFFFFFu) x = loss[lidx[q] + ((ent[q] >> kChunkBits) & kRowMask)];
      l[q] = x;
    }
#pragma unroll
    for (int q = 0; q < kOwn; ++q)
      add_keys(acc, touched, ent[q] != 0xFFFFFFFFu, ent[q] & (kChunk - 1), l[q]);
  }
  __syncthreads();
  if constexpr (VAR & kDiagNoUpdate) {
    if (acc[tid] == 12345.0) T.w[row0] = 0.0f;  // (keeps the sums alive)
    return;
  }
  // The optimizer steps in three sweeps — request every state row, step, store — so that a
  // thread's rows are all in flight together.  (Row after row — load, step, store, next — the
  // stores to T.w keep the compiler from moving the next row's loads up: eight dependent trips
  // to memory per thread, 10 of the ~35 us a workgroup lived.)
  if constexpr (COMPACT) {
    // wavefront w lists the touched keys among [w * KW, (w + 1) * KW), ascending, in its own
    // part of `list` (LDS operations of one wavefront execute in order: no barrier), then takes
    // them 64 at a time with every lane busy
    constexpr int kSlots = (int)(KW / 64);
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    uint16_t *wl = list + wave * KW;
    uint32_t cnt = 0;
#pragma unroll
    for (uint32_t j = 0; j < KW; j += 64) {
      const uint32_t k = wave * KW + j + lane;
      const bool t = touched[k] != 0 && row0 + k < M;
      const unsigned long long m = __ballot(t);
      if (t) wl[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)k;
      cnt += (uint32_t)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t kk[kSlots];
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
      const uint32_t p = lane + 64u * i;
      kk[i] = p < cnt ? (uint32_t)wl[p] : 0xFFFFFFFFu;
      // (an idle slot loads the chunk's first row: a load under a branch would have to be
      // waited for where the branch ends, one slot after the other)
      const size_t r = row0 + (kk[i] != 0xFFFFFFFFu ? kk[i] : 0u);
      sw[i] = 0.0f;
      if (!wnz) sw[i] = T.w[r];
      sn[i] = sz[i] = 0.0f;
      if (OPT == XF_OPT_FTRL) xf::load_nz(T, r, sn[i], sz[i]);
    }
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
      if (kk[i] == 0xFFFFFFFFu) continue;
      const float g = xf::div_by_rows((float)acc[kk[i]], R);  // lr_worker.cc:117
      if (wnz) sw[i] = xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, sn[i], sz[i]);
      if (OPT == XF_OPT_FTRL)
        xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g, sw[i], sn[i], sz[i]);
      else
        sw[i] = xf::sgd_step(T.lr, g, sw[i]);
    }
#pragma unroll
    for (int i = 0; i < kSlots; ++i) {
      if (kk[i] == 0xFFFFFFFFu) continue;
      T.w[row0 + kk[i]] = sw[i];
      if (OPT == XF_OPT_FTRL) xf::store_nz(T, row0 + kk[i], sn[i], sz[i]);
    }
  } else {
    bool t[kOwn];
#pragma unroll
    for (int i = 0; i < kOwn; ++i) {
      const uint32_t k = tid + i * NT;
      t[i] = touched[k] != 0 && row0 + k < M;
      if constexpr (!PREFETCH) {  // (unconditional, see above; row0 itself is below M)
        const size_t r = (t[i] || ((VAR & kDenseFullStore) && row0 + k < M)) ? row0 + k : row0;
        sw[i] = 0.0f;
        if (!wnz) sw[i] = T.w[r];
        sn[i] = sz[i] = 0.0f;
        if (OPT == XF_OPT_FTRL) xf::load_nz(T, r, sn[i], sz[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < kOwn; ++i) {
      // (every row: kDenseFullStore stores the untouched ones too — the bits they had)
      if (wnz) sw[i] = xf::ftrl_w_of(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, sn[i], sz[i]);
      if (!t[i] || (VAR & kDiagCopy)) continue;
      const float g = xf::div_by_rows((float)acc[tid + i * NT], R);  // lr_worker.cc:117
      if (OPT == XF_OPT_FTRL)
        xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g, sw[i], sn[i], sz[i]);
      else
        sw[i] = xf::sgd_step(T.lr, g, sw[i]);
    }
    if constexpr (VAR & kDiagNoStore) {
      float x = 0.0f;
#pragma unroll
      for (int i = 0; i < kOwn; ++i) x += sw[i] + sn[i] + sz[i];
      if (x == 12345.0f) T.w[row0] = x;  // (keeps the steps alive)
      return;
    }
#pragma unroll
    for (int i = 0; i < kOwn; ++i) {
      if constexpr (VAR & kDenseFullStore) {
        if (row0 + tid + i * NT >= M) continue;
      } else {
        if (!t[i]) continue;
      }
      T.w[row0 + tid + i * NT] = sw[i];
      if (OPT == XF_OPT_FTRL) xf::store_nz(T, row0 + tid + i * NT, sn[i], sz[i]);
    }
  }
}

// the keys of the split chunks: one lane per key
template <int OPT, int MODE>
__global__ void __launch_bounds__(kBlock)
k_lr_grad_split_finish(xf::TableDev T, const uint32_t *__restrict__ split_chunk,
                       double *__restrict__ gsum, uint8_t *__restrict__ gtouched,
                       uint32_t R, uint32_t M, float *__restrict__ g_out, uint32_t nsrc,
                       const uint32_t *__restrict__ src_rows, uint32_t nsplit, uint32_t chunk0,
                       int clean) {
  const uint32_t slot = blockIdx.x / (kChunk / kBlock);
  const uint32_t k = (blockIdx.x % (kChunk / kBlock)) * kBlock + threadIdx.x;
  const size_t idx = (size_t)(chunk0 + split_chunk[slot]) * kChunk + k;
  if (idx >= M) return;  // (no entry names a row the table does not hold: never touched)
  const uint32_t ns = src_rows ? nsrc : 1u;
  for (uint32_t q = 0; q < ns; ++q) {  // the workers' steps in rank order
    const size_t o = ((size_t)q * nsplit + slot) * kChunk + k;
    if (!gtouched[o]) continue;
    const double sum = gsum[o];
    if (clean) {  // the accumulators go back to zero here: no memset before the next pass
      gsum[o] = 0.0;
      gtouched[o] = 0;
    }
    const float g = xf::div_by_rows((float)sum, src_rows ? src_rows[q] : R);
    if (g_out) g_out[idx] = g;
    if (MODE == 0) apply_key(T, OPT, idx, g);
  }
}

}  // namespace

namespace xf {

const TableDev &table_dev(const xf_table *t);
void table_note_write(xf_table *t);
uint64_t table_uid(const xf_table *t);
uint64_t table_epoch(const xf_table *t);
int table_resolve_any(xf_table *t, const uint64_t *d_keys, size_t n, uint32_t *d_rows,
                      hipStream_t s, bool allow_grow);

void cells_free(xf_cells *c) {
  while (c) {
    xf_cells *n = c->next;
    if (c->blob) blob_free(c->blob, c->blob_bytes);
    if (c->blob2) blob_free(c->blob2, c->blob2_bytes);
    delete c;
    c = n;
  }
}

size_t cells_partial_doubles(const xf_cells *c) { return (size_t)c->G * c->nwin * c->W; }

uint32_t cells_split_chunks(const xf_cells *c) {
  uint32_t n = 0;
  for (; c; c = c->next) n += c->nsplit_chunks;
  return n;
}

// geometry + the allocation whose size the shape decides: entries (two copies when the
// key-sorted one is wanted), cellptr, blk_cell, the item plan
int cells_alloc(xf_cells **out, uint32_t R, uint32_t NNZ, uint32_t M, int mode,
                bool key_sorted_copy, uint32_t w_fixed, uint32_t chunk0) {
  XF_REQUIRE(out, "cells_alloc: null argument");
  XF_REQUIRE(w_fixed <= kWinMax, "cells_alloc: %u rows per window", w_fixed);
  xf_cells *c = new xf_cells;
  c->R = R;
  c->NNZ = NNZ;
  c->M = M;
  c->mode = mode;
  c->chunk0 = chunk0;
  if (w_fixed) {  // the caller numbered the rows window by window
    c->W = w_fixed;
    c->nwin = std::max<uint32_t>(1, (R + w_fixed - 1) / w_fixed);
  } else {
    c->nwin = std::max<uint32_t>(1, (R + kWinMax - 1) / kWinMax);
    c->W = std::max<uint32_t>(1, (R + c->nwin - 1) / c->nwin);
  }
  const uint64_t nchunk_all = std::max<uint64_t>(1, ((uint64_t)M + kChunk - 1) / kChunk);
  c->nchunk = (uint32_t)std::max<uint64_t>(1, nchunk_all > chunk0 ? nchunk_all - chunk0 : 1);
  const uint64_t ncell64 = (uint64_t)c->nwin * c->nchunk;
  if (ncell64 >= 0x7FFFFFFFull) {
    delete c;
    return xf::set_error(XF_EINVAL, "cells: %llu cells", (unsigned long long)ncell64);
  }
  c->ncell = (uint32_t)ncell64;
  c->nblk = (NNZ + kBlk - 1) / kBlk;
  // Groups per window: a multiple of 8 with at most 32 workgroups per XCD (one per CU: the row
  // window fills the LDS).  Group g of EVERY window runs on XCD g % 8 (k_lr_fwd_cells), so the
  // windows read a weight range through one L2: once from HBM instead of once per window.
  c->G = c->nwin <= 32 ? (kFwdGroups / 8 / c->nwin) * 8 : 8;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_ent = 0;
  const size_t o_entk = o_ent + al((size_t)NNZ * 4);
  const size_t o_cellptr = o_entk + al(key_sorted_copy ? (size_t)NNZ * 4 : 0);
  const size_t o_blk = o_cellptr + al(((size_t)c->ncell + 1) * 4);
  const size_t o_plan = o_blk + al(((size_t)c->nblk + 1) * 4);
  const size_t total = o_plan + al(4 * ((size_t)c->nchunk + 1) * 4) + 256;
  int rc = blob_alloc((void **)&c->blob, total, &c->blob_bytes);
  if (rc != XF_OK) {
    delete c;
    return rc;
  }
  c->entries = (uint32_t *)(c->blob + o_ent);
  c->entries_k = key_sorted_copy ? (uint32_t *)(c->blob + o_entk) : c->entries;
  c->cellptr = (uint32_t *)(c->blob + o_cellptr);
  c->blk_cell = (uint32_t *)(c->blob + o_blk);
  c->plan = (uint32_t *)(c->blob + o_plan);
  *out = c;
  return XF_OK;
}

// the forward's copy: every cell ordered by its entries' low bits (the position within the
// chunk), so that neighbouring lanes gather neighbouring weights (forward kernel 58 -> 42 us on
// the config-2 shape): k_cells_sort_pos.  A minibatch that is stepped once goes without and the
// forward reads the row-sorted cells.  In stream order, nothing is waited for.
int cells_key_sorted_copy(xf_cells *c, hipStream_t s) {
  if (!c->NNZ || c->entries_k == c->entries) return XF_OK;
  hipLaunchKernelGGL(k_cells_sort_pos, dim3((c->ncell + kSortCells - 1) / kSortCells), dim3(kBlock),
                     0, s, c->entries, c->entries_k, c->cellptr, c->ncell);
  XF_HIP(hipGetLastError());
  c->entries_k_ready = true;
  return XF_OK;
}

// gradient work items, part 1 (on the device, into the cells' own allocation): slices per
// chunk and their scans; the totals are plan[2*(nchunk+1) - 1] (items), plan[3*(nchunk+1) - 1]
// (split chunks) and plan[4*(nchunk+1) - 1] (their slices)
int cells_plan_items(xf_cells *c, hipStream_t s) {
  const size_t nc1 = (size_t)c->nchunk + 1;
  hipLaunchKernelGGL(k_plan_items, dim3(1), dim3(kPlanBlock), 0, s, c->cellptr, c->nchunk,
                     c->nwin, c->plan, c->plan + nc1, c->plan + 2 * nc1, c->plan + 3 * nc1);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

// part 2, once the host knows the totals: the item lists and the split chunks' accumulators
int cells_fill_items(xf_cells *c, uint32_t nitems, uint32_t nsplit, hipStream_t s) {
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  c->nitems = nitems;
  c->nsplit_chunks = nsplit;
  const size_t o_ic = 0;
  const size_t o_is = o_ic + al((size_t)c->nitems * 4);
  const size_t o_id = o_is + al((size_t)c->nitems * 4);
  const size_t o_sc = o_id + al((size_t)c->nitems * 4);
  const size_t o_gd = o_sc + al((size_t)c->nsplit_chunks * 4);
  const size_t o_td = o_gd + al((size_t)c->nsplit_chunks * kChunk * 8);
  const size_t o_dn = o_td + al((size_t)c->nsplit_chunks * kChunk);
  const size_t total2 = o_dn + al((size_t)c->nitems) + 256;
  c->split_bytes = o_dn - o_gd;
  XF_TRY(blob_alloc((void **)&c->blob2, total2, &c->blob2_bytes));
  char *d = c->blob2;
  c->item_chunk = (uint32_t *)(d + o_ic);
  c->item_slice = (uint32_t *)(d + o_is);
  c->item_dump = (uint32_t *)(d + o_id);
  c->split_chunk = (uint32_t *)(d + o_sc);
  c->gsum = (double *)(d + o_gd);
  c->gtouched = (uint8_t *)(d + o_td);
  c->item_done = (uint8_t *)(d + o_dn);
  const size_t nc1 = (size_t)c->nchunk + 1;
  if (c->nitems)
    hipLaunchKernelGGL(k_items_fill, dim3(grid_for(c->nchunk)), dim3(kBlock), 0, s, c->plan,
                       c->plan + nc1, c->plan + 2 * nc1, c->plan + 3 * nc1, c->nchunk, c->item_chunk,
                       c->item_slice, c->item_dump, c->split_chunk);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

int cells_build(xf_cells **out, const uint32_t *d_src, const uint32_t *d_map,
                const uint32_t *d_rowptr, uint32_t R, uint32_t NNZ, uint32_t M, int mode,
                bool key_sorted_copy, hipStream_t s, const uint32_t *d_rowid, uint32_t w_fixed,
                uint32_t chunk0) {
  XF_REQUIRE(out && (d_rowptr || d_rowid || NNZ == 0) && (NNZ == 0 || d_src),
             "cells_build: null argument");
  XF_REQUIRE(!d_rowid || (w_fixed >= 1 && w_fixed <= kWinMax && !d_map),
             "cells_build: row ids need a window size");
  xf_cells *c = nullptr;
  XF_TRY(cells_alloc(&c, R, NNZ, M, mode, key_sorted_copy, w_fixed, chunk0));
  struct Guard {
    xf_cells *c;
    ~Guard() {
      if (c) cells_free(c);
    }
  } guard{c};
  {
    Scratch sc;
    uint32_t *cid = nullptr, *ent = nullptr, *cid_s = nullptr;
    XF_TRY(sc.get(&cid, NNZ));
    XF_TRY(sc.get(&ent, NNZ));
    XF_TRY(sc.get(&cid_s, NNZ));
    if (NNZ) {
      if (d_rowid)
        hipLaunchKernelGGL(k_cell_keys_rowid, dim3(grid_for((size_t)NNZ)), dim3(kBlock), 0, s,
                           d_rowid, d_src, (size_t)NNZ, c->W, c->nchunk, chunk0, cid, ent);
      else
        hipLaunchKernelGGL(k_cell_keys, dim3(grid_for((size_t)R * 64)), dim3(kBlock), 0, s,
                           d_rowptr, d_src, d_map, R, c->W, c->nchunk, chunk0, cid, ent);
      int bits = 1;
      while (bits < 32 && (1ull << bits) < (uint64_t)c->ncell) ++bits;
      size_t tb = 0;
      XF_HIP(rocprim::radix_sort_pairs(nullptr, tb, cid, cid_s, ent, c->entries, (size_t)NNZ, 0,
                                       bits, s));
      void *tmp = nullptr;
      XF_TRY(sc.get((char **)&tmp, tb));
      XF_HIP(rocprim::radix_sort_pairs(tmp, tb, cid, cid_s, ent, c->entries, (size_t)NNZ, 0, bits,
                                       s));
    }
    hipLaunchKernelGGL(k_cellptr, dim3(grid_for((size_t)c->ncell + 1)), dim3(kBlock), 0, s, cid_s,
                       NNZ, c->ncell, c->cellptr);
    hipLaunchKernelGGL(k_blk_cell, dim3(grid_for((size_t)c->nblk + 1)), dim3(kBlock), 0, s, cid_s,
                       c->nblk, c->ncell, c->blk_cell);
    XF_TRY(cells_plan_items(c, s));
    uint32_t totals[2] = {0, 0};
    const size_t nc1 = (size_t)c->nchunk + 1;
    XF_HIP(hipMemcpyAsync(&totals[0], c->plan + 2 * nc1 - 1, 4, hipMemcpyDeviceToHost, s));
    XF_HIP(hipMemcpyAsync(&totals[1], c->plan + 3 * nc1 - 1, 4, hipMemcpyDeviceToHost, s));
    XF_HIP(hipGetLastError());
    XF_HIP(hipStreamSynchronize(s));
    XF_TRY(cells_fill_items(c, totals[0], totals[1], s));
  }
  XF_TRY(cells_key_sorted_copy(c, s));
  XF_HIP(hipStreamSynchronize(s));
  guard.c = nullptr;
  *out = c;
  return XF_OK;
}

int cells_lr_forward(const xf_cells *c, const float *d_w, const int32_t *d_labels,
                     double *d_partial, float *d_loss, float *d_pctr, hipStream_t s) {
  XF_REQUIRE(c && d_w && d_partial && (d_loss || d_pctr), "cells_lr_forward: null argument");
  if (c->R == 0) return XF_OK;
  int acc = 0;
  for (const xf_cells *q = c; q; q = q->next, acc = 1)
    hipLaunchKernelGGL(k_lr_fwd_cells, dim3(q->nwin * q->G), dim3(kFwdBlock), 0, s, q->fwd_entries(),
                       q->cellptr, q->blk_cell, q->nchunk, q->W, q->G,
                       d_w + (size_t)q->chunk0 * kChunk, d_partial, acc);
  hipLaunchKernelGGL(k_lr_finalize_cells,
                     dim3((unsigned)(((size_t)c->R * 4 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                     d_partial, d_labels, c->R, c->W, c->G, d_loss, d_pctr);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

// sources: the workers whose rows the windows hold (CellSources, owner-compute step); null: one
struct CellSources {
  uint32_t n = 0;
  const uint32_t *d_win = nullptr;   // [n + 1] first window of every worker
  const uint32_t *d_rows = nullptr;  // [n] rows of every worker's minibatch
  const uint32_t *d_loss_base = nullptr;  // [nwin] where a window's losses start in d_loss
  uint32_t rows_host = 0;            // n == 1: the source's rows (all workers'), known on the host
  double *gsum = nullptr;            // [n * nsplit_chunks * kChunk] split chunks' sums per worker
  uint8_t *gtouched = nullptr;       // [n * nsplit_chunks * kChunk]
};

// whole-line stores of a chunk's state pay when most 128-byte lines hold a touched row (63 % of
// the rows at the config-2 shape: 74 -> 70 us); a minibatch that touches the table thinly (a
// 1e8-key shard: a tenth of the rows) would write ten times what it changes.  Entries per chunk
// of the items the pass runs over stand in for the touch density (uniform keys: 0.3 entries
// per row = a quarter of the rows touched, three lines in four hold one).
static bool dense_touch(const xf_cells *c) {
  return (double)c->NNZ >= 0.3 * (double)c->nitems * (double)kChunk;
}

template <int OPT, int MODE>
static int launch_grad(const xf_cells *c, const TableDev &T_, const float *d_loss, float *d_g,
                       hipStream_t s, const CellSources *src = nullptr) {
  if (c->nitems == 0) return XF_OK;
  TableDev T = T_;
  if (path_switch(kPathOldWeight) == 1) T.w_of_nz = false;  // (old_weight = read, xf_common.h)
  double *gsum = src ? src->gsum : c->gsum;
  uint8_t *gtouched = src ? src->gtouched : c->gtouched;
  const uint8_t *no_skip = nullptr;
  if (c->nsplit_chunks) {
    if (src) {
      const size_t cells = (size_t)src->n * c->nsplit_chunks * kChunk;
      XF_HIP(hipMemsetAsync(gsum, 0, cells * 8, s));
      XF_HIP(hipMemsetAsync(gtouched, 0, cells, s));
    } else if (c->split_dirty) {
      XF_HIP(hipMemsetAsync(c->gsum, 0, c->split_bytes, s));
    }
    c->split_dirty = true;  // (until the finish kernel that cleans them is in the stream)
  }
  if (src && src->n == 1 && src->rows_host) {
    // ONE source (sum_then_step, or a group of one): the plain instantiation — its optimizer
    // steps in three sweeps — with the windows' losses where the exchange left them and the
    // row count of all workers together
    hipLaunchKernelGGL((k_lr_grad_cells<OPT, MODE, false>), dim3(c->nitems), dim3(kBlock), 0, s,
                       T, c->entries, c->cellptr, c->nchunk, c->nwin, c->W, c->item_chunk,
                       c->item_slice, c->item_dump, d_loss, src->rows_host, c->M, d_g, gsum,
                       gtouched, 1u, (const uint32_t *)nullptr, (const uint32_t *)nullptr,
                       c->nsplit_chunks, src->d_loss_base, c->chunk0, no_skip);
    if (c->nsplit_chunks)
      hipLaunchKernelGGL((k_lr_grad_split_finish<OPT, MODE>),
                         dim3(c->nsplit_chunks * (kChunk / kBlock)), dim3(kBlock), 0, s, T,
                         c->split_chunk, gsum, gtouched, src->rows_host, c->M, d_g, 1u,
                         (const uint32_t *)nullptr, c->nsplit_chunks, c->chunk0, 0);
    XF_HIP(hipGetLastError());
    return XF_OK;
  }
  if (src && src->n > 1) {
    // several workers: k_lr_grad_multi takes the unsplit chunks that fit one round of
    // registers, the general loop the others (the slices of split chunks, chunks with more
    // entries, more row windows or workers than its LDS tables hold)
    bool multi = false;
    const int pass = path_switch(kPathOwnerPass);  // (xf_common.h: 0 by shape)
    if constexpr (MODE == 0) multi = !d_g && c->item_done && pass != 1;
    if (multi) {
      if constexpr (MODE == 0) {
        const int full = dense_touch(c) ? 1 : 0;
        // 512 threads per chunk (four entries per lane, 64 registers), the key sums in the
        // stepping lanes' slots (SLOTS: 34 KB of LDS, four workgroups = 32 wavefronts per CU):
        // 140 us at the N = 8 shard shape; the sums indexed by key (46 KB, three workgroups): 155;
        // 256 threads: 164-193; 1024: 183 (the variants: -DXF_EXPERIMENTS)
#ifdef XF_EXPERIMENTS
        if (exp_knob() == 295)  // (the key sums indexed by key: three workgroups per CU, 155 us)
          hipLaunchKernelGGL((k_lr_grad_multi<OPT, 512, false>), dim3(c->nitems), dim3(512), 0, s,
                             T, c->entries, c->cellptr, c->nchunk, c->nwin, c->item_chunk,
                             c->item_slice, d_loss, c->M, src->n, src->d_win, src->d_rows,
                             src->d_loss_base, c->chunk0, full, c->item_done);
        else if (exp_knob() == 296)  // (experiment: 1024 threads, two workgroups = 32 wavefronts per CU)
          hipLaunchKernelGGL((k_lr_grad_multi<OPT, 1024>), dim3(c->nitems), dim3(1024), 0, s, T,
                             c->entries, c->cellptr, c->nchunk, c->nwin, c->item_chunk,
                             c->item_slice, d_loss, c->M, src->n, src->d_win, src->d_rows,
                             src->d_loss_base, c->chunk0, full, c->item_done);
        else if (exp_knob() == 297)
          hipLaunchKernelGGL((k_lr_grad_multi<OPT, 256>), dim3(c->nitems), dim3(256), 0, s, T,
                             c->entries, c->cellptr, c->nchunk, c->nwin, c->item_chunk,
                             c->item_slice, d_loss, c->M, src->n, src->d_win, src->d_rows,
                             src->d_loss_base, c->chunk0, full, c->item_done);
        // a phase per worker (k_lr_grad_multi, SLOTS: the key sums in the stepping lanes' slots,
        // 140 us at the N = 8 shard shape) where a worker's entries in a chunk fit its slots; the
        // merged phases (k_lr_grad_ranked: ~140 us whatever the number of workers — 134 / 141 /
        // 143 us for 2 / 4 / 8 of them against 185 (the general kernel: over the slots) / 123 /
        // 141) where they do not: two or three workers.  (owner_pass = 4 / 2: one or the other.)
        else
#endif
        if (pass == 4 || src->n > 32 ||
            (pass != 2 && pass != 3 && (double)c->NNZ / c->nitems / src->n <= 0.9 * kMultiCap))
          hipLaunchKernelGGL((k_lr_grad_multi<OPT, 512, true>), dim3(c->nitems), dim3(512), 0, s,
                             T, c->entries, c->cellptr, c->nchunk, c->nwin, c->item_chunk,
                             c->item_slice, d_loss, c->M, src->n, src->d_win, src->d_rows,
                             src->d_loss_base, c->chunk0, full, c->item_done);
        else if (src->n <= 8 && pass != 3)  // the workers' phases merged (k_lr_grad_ranked;
                                            // owner_pass = 3: its 32-bit masks whatever the number)
          hipLaunchKernelGGL((k_lr_grad_ranked<OPT, 8>), dim3(c->nitems), dim3(512), 0, s, T,
                             c->entries, c->cellptr, c->nchunk, c->nwin, c->item_chunk,
                             c->item_slice, d_loss, c->M, src->n, src->d_win, src->d_rows,
                             src->d_loss_base, c->chunk0, full, c->item_done);
        else
          hipLaunchKernelGGL((k_lr_grad_ranked<OPT, 32>), dim3(c->nitems), dim3(512), 0, s, T,
                             c->entries, c->cellptr, c->nchunk, c->nwin, c->item_chunk,
                             c->item_slice, d_loss, c->M, src->n, src->d_win, src->d_rows,
                             src->d_loss_base, c->chunk0, full, c->item_done);
      }
      hipLaunchKernelGGL((k_lr_grad_cells<OPT, MODE, true>), dim3(c->nitems), dim3(kBlock), 0, s,
                         T, c->entries, c->cellptr, c->nchunk, c->nwin, c->W, c->item_chunk,
                         c->item_slice, c->item_dump, d_loss, c->R, c->M, d_g, gsum, gtouched,
                         src->n, src->d_win, src->d_rows, c->nsplit_chunks, src->d_loss_base,
                         c->chunk0, (const uint8_t *)c->item_done);
    } else {  // (the general loop, a sweep per worker: gradients wanted, or owner_pass = 1)
      hipLaunchKernelGGL((k_lr_grad_cells<OPT, MODE, true, true>), dim3(c->nitems), dim3(kBlock),
                         0, s, T, c->entries, c->cellptr, c->nchunk, c->nwin, c->W, c->item_chunk,
                         c->item_slice, c->item_dump, d_loss, c->R, c->M, d_g, gsum, gtouched,
                         src->n, src->d_win, src->d_rows, c->nsplit_chunks, src->d_loss_base,
                         c->chunk0, no_skip);
    }
  } else if (src) {
    hipLaunchKernelGGL((k_lr_grad_cells<OPT, MODE, true>), dim3(c->nitems), dim3(kBlock), 0, s, T,
                       c->entries, c->cellptr, c->nchunk, c->nwin, c->W, c->item_chunk,
                       c->item_slice, c->item_dump, d_loss, c->R, c->M, d_g, gsum, gtouched,
                       src->n, src->d_win, src->d_rows, c->nsplit_chunks, src->d_loss_base,
                       c->chunk0, no_skip);
  } else {
    // one source: the unsplit chunks go to k_lr_grad_dense (gradient + Push, no dense copy of
    // the gradients wanted, few windows), the general kernel keeps the split ones
    bool dense = false;
    if constexpr (MODE == 0) {
      // (a minibatch with split chunks stays with the general kernel: its long unsplit chunks
      // and the slices of the split ones share one launch there; two launches one after the other
      // add their tails — Zipf 1.1: 118 us instead of 88)
      if (!d_g && c->nwin <= kDenseWin && c->nsplit_chunks == 0) {
        // whole-line stores (variant kDenseFullStore) where the minibatch touches most lines of
        // the chunks it runs over, byte-masked stores of the touched rows where it does not
        int var = dense_touch(c) ? XF_GRAD_DENSE_VAR : (XF_GRAD_DENSE_VAR & ~kDenseFullStore);
#ifdef XF_EXPERIMENTS
        const int knob = exp_knob();
        if (knob >= 300 && knob < 812) var = knob - 300;  // (tools/cells_knobs.py)
#endif
        const int lg = path_switch(kPathLrGradient);  // (xf_common.h)
        if (lg == 2) var = 0;
        if (lg == 3) var = kDenseFullStore;
        dense = lg != 1;
        // (Measured and dropped: fewer workgroups per CU — 4 .. 7 instead of 8, by a pad of dynamic
        // LDS — so that the rounds of workgroups come out even (4883 chunks are 2.38 rounds of
        // 2048): 74.5-76.6 us at every occupancy against 74.6-75.4, tools/r5/call13.sh.)
        // The old weights derived from (n, z) instead of read (TableDev::w_of_nz) where the
        // kernel waits for lines of state: a table whose state does not fit the 256 MiB Infinity
        // Cache (2 / 3 / 10 x 10^7 keys: 125.6 -> 120.4, 159.8 -> 148.4, 377.7 -> 310.6 us), or a
        // minibatch that touches the chunks thinly.  A small table touched densely (config 2:
        // 120 MB of state, every line of w needed anyway) has the lines on hand and the ~45
        // instructions of the derivation per row are not hidden (this kernel's phases add up,
        // DESIGN 3): 74.6 -> 77.7 us with it, so there the kernel reads w.  (old_weight = derive:
        // derived whatever the table.)
        TableDev Td = T;
        if (dense_touch(c) && (size_t)c->M * 12 <= ((size_t)180 << 20) &&
            path_switch(kPathOldWeight) != 2)
          Td.w_of_nz = false;
        if (dense) {
#define XF_DENSE(V)                                                                              \
  case V:                                                                                        \
    hipLaunchKernelGGL((k_lr_grad_dense<OPT, V>),                                                \
                       dim3(((V) & kDenseQuad) && dense_team(V) * 4 <= 1024 ? (c->nitems + 3) / 4 \
                                                                            : c->nitems),        \
                       dim3(dense_threads(V)), 0,                                                \
                       s, Td, c->entries, c->cellptr, c->nchunk, c->nwin, c->W, c->item_chunk,    \
                       d_loss, c->R, c->M, c->chunk0, c->nitems);                                \
    break
          switch (var) {
            XF_DENSE(0);
            XF_DENSE(128);
#ifdef XF_EXPERIMENTS  // the measured-and-not-adopted variants and the timing
                       // experiments (some with WRONG results): never in a product build
            XF_DENSE(1);
            XF_DENSE(2);
            XF_DENSE(4);
            XF_DENSE(128 + 4);
            XF_DENSE(5);
            XF_DENSE(6);
            XF_DENSE(8);
            XF_DENSE(16);
            XF_DENSE(32);
            XF_DENSE(32 + 64);
            XF_DENSE(32 + 8);
            XF_DENSE(256);
            XF_DENSE(256 + 128);
            XF_DENSE(256 + 1);
            XF_DENSE(256 + 16);
            XF_DENSE(4 + 16);
            XF_DENSE(4 + 32);
            XF_DENSE(4 + 32 + 64);
#endif
            default:
              return xf::set_error(XF_EINVAL, "gradient kernel variant %d (the timing-only "
                                   "variants need a library built with -DXF_EXPERIMENTS)", var);
          }
#undef XF_DENSE
        }
      }
    }
    if (!dense)
      hipLaunchKernelGGL((k_lr_grad_cells<OPT, MODE, false>), dim3(c->nitems), dim3(kBlock), 0, s,
                         T, c->entries, c->cellptr, c->nchunk, c->nwin, c->W, c->item_chunk,
                         c->item_slice, c->item_dump, d_loss, c->R, c->M, d_g, gsum, gtouched, 1u,
                         (const uint32_t *)nullptr, (const uint32_t *)nullptr, c->nsplit_chunks,
                         (const uint32_t *)nullptr, c->chunk0, no_skip);
  }
  if (c->nsplit_chunks)
    hipLaunchKernelGGL((k_lr_grad_split_finish<OPT, MODE>),
                       dim3(c->nsplit_chunks * (kChunk / kBlock)), dim3(kBlock), 0, s, T,
                       c->split_chunk, gsum, gtouched, c->R, c->M, d_g, src ? src->n : 1u,
                       src ? src->d_rows : (const uint32_t *)nullptr, c->nsplit_chunks,
                       c->chunk0, src ? 0 : 1);
  XF_HIP(hipGetLastError());
  if (!src) c->split_dirty = false;
  return XF_OK;
}

// gradient only: g_out[idx] for every index position the minibatch touches
int cells_lr_grad(const xf_cells *c, const float *d_loss, float *d_g, hipStream_t s) {
  XF_REQUIRE(c && d_loss && d_g, "cells_lr_grad: null argument");
  for (; c; c = c->next) XF_TRY((launch_grad<XF_OPT_SGD, 1>(c, TableDev{}, d_loss, d_g, s)));
  return XF_OK;
}

// gradient + Push on the table the cells were compiled against (d_g: optional dense copy of
// the gradients, indexed by state row — the parity hook)
int cells_lr_grad_update(const xf_cells *c, const xf_table *t, const float *d_loss, float *d_g,
                         hipStream_t s) {
  XF_REQUIRE(c && t && d_loss, "cells_lr_grad_update: null argument");
  XF_REQUIRE(c->mode == kCellsTableRows, "cells_lr_grad_update: cells are not table rows");
  const TableDev &T = table_dev(t);
  XF_REQUIRE(T.dim == 1, "cells_lr_grad_update: dim must be 1");
  table_note_write(const_cast<xf_table *>(t));
  for (; c; c = c->next) {
    if (T.nz != nullptr) XF_TRY((launch_grad<XF_OPT_FTRL, 0>(c, T, d_loss, d_g, s)));
    else
      XF_TRY((launch_grad<XF_OPT_SGD, 0>(c, T, d_loss, d_g, s)));
  }
  return XF_OK;
}

// The owner-compute step's gradient + Pushes: the cells hold the rows of `n` workers (windows
// [d_win[q], d_win[q+1]) are worker q's), the losses of window v start at d_loss[d_loss_base[v]]
// (the workers' losses back to back, as they arrive); every
// worker's gradient (its sum / d_rows[q]) is its own optimizer step, applied in rank order.
// d_gsum / d_gtouched: n * cells_split_chunks(c) * kChunk elements of scratch (null when no
// chunk of the cells is split).
int cells_lr_grad_update_sources(const xf_cells *c, const xf_table *t, const float *d_loss,
                                 uint32_t n, const uint32_t *d_win, const uint32_t *d_rows,
                                 const uint32_t *d_loss_base, double *d_gsum,
                                 uint8_t *d_gtouched, hipStream_t s, uint32_t rows_if_one) {
  XF_REQUIRE(c && t && d_loss && n && d_win && d_rows && d_loss_base,
             "cells_lr_grad_update_sources: null");
  XF_REQUIRE(c->mode == kCellsTableRows, "cells_lr_grad_update_sources: cells are not table rows");
  XF_REQUIRE(cells_split_chunks(c) == 0 || (d_gsum && d_gtouched),
             "cells_lr_grad_update_sources: no scratch for the split chunks");
  const TableDev &T = table_dev(t);
  XF_REQUIRE(T.dim == 1, "cells_lr_grad_update_sources: dim must be 1");
  table_note_write(const_cast<xf_table *>(t));
  size_t used = 0;  // split chunks of the segments before this one
  for (; c; c = c->next) {
    CellSources src;
    src.n = n;
    src.d_win = d_win;
    src.d_rows = d_rows;
    src.d_loss_base = d_loss_base;
    src.rows_host = n == 1 ? rows_if_one : 0u;
    src.gsum = d_gsum ? d_gsum + (size_t)n * used * kChunk : nullptr;
    src.gtouched = d_gtouched ? d_gtouched + (size_t)n * used * kChunk : nullptr;
    used += c->nsplit_chunks;
    if (T.nz != nullptr) XF_TRY((launch_grad<XF_OPT_FTRL, 0>(c, T, d_loss, nullptr, s, &src)));
    else
      XF_TRY((launch_grad<XF_OPT_SGD, 0>(c, T, d_loss, nullptr, s, &src)));
  }
  return XF_OK;
}

// forward up to the row sums: d_rowsum[window * W + row-in-window] = sum of the row's weights
// (fp64); the owner-compute step sends them to the rows' workers, who add the owners' sums
namespace {
__global__ void __launch_bounds__(kBlock)
k_sum_partials(const double *__restrict__ partial, uint32_t n, uint32_t W, uint32_t G,
               const uint32_t *__restrict__ out_base, const uint32_t *__restrict__ out_rows,
               double *__restrict__ rowsum) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t r = t >> 2, q = t & 3u;
  const uint32_t v = r < n ? r / W : 0u, rin = r - v * W;
  const bool live = r < n && rin < out_rows[v];  // (the last window of a worker is not full)
  double a = 0.0;
  if (live) {
    const double *p = partial + (size_t)v * G * W + rin;
    for (uint32_t g = q; g < G; g += 4) a += p[(size_t)g * W];
  }
  a += __shfl_xor(a, 1);
  a += __shfl_xor(a, 2);
  if (live && q == 0) rowsum[out_base[v] + rin] = a;
}
}  // namespace

// d_rowsum[d_out_base[window] + row-in-window] for the first d_out_rows[window] rows of every
// window: the workers' rows back to back, ready to be sent
int cells_lr_forward_sums(const xf_cells *c, const float *d_w, double *d_partial,
                          const uint32_t *d_out_base, const uint32_t *d_out_rows,
                          double *d_rowsum, hipStream_t s) {
  XF_REQUIRE(c && d_w && d_partial && d_rowsum && d_out_base && d_out_rows,
             "cells_lr_forward_sums: null argument");
  if (c->R == 0) return XF_OK;
  int acc = 0;
  for (const xf_cells *q = c; q; q = q->next, acc = 1)
    hipLaunchKernelGGL(k_lr_fwd_cells, dim3(q->nwin * q->G), dim3(kFwdBlock), 0, s, q->fwd_entries(),
                       q->cellptr, q->blk_cell, q->nchunk, q->W, q->G,
                       d_w + (size_t)q->chunk0 * kChunk, d_partial, acc);
  const uint32_t n = c->nwin * c->W;
  hipLaunchKernelGGL(k_sum_partials, dim3((unsigned)(((size_t)n * 4 + kBlock - 1) / kBlock)),
                     dim3(kBlock), 0, s, d_partial, n, c->W, c->G, d_out_base, d_out_rows,
                     d_rowsum);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

}  // namespace xf

// ---------------------------------------------------------------- cells of a compiled batch
namespace xf {

// Make b->cells the cells of `b` against table `t`'s current row numbering.  Batches with a
// key list (xf_batch_compile*): Pull's key -> row resolve (insert on first touch, ftrl.h:56)
// over the sorted unique keys, then uidx -> row.  Local batches: the retained raw keys are
// resolved again.  Either way this is where the minibatch's keys enter the table — what the
// Pull of LRWorker::update does (lr_worker.cc:170).
int ensure_cells(xf_batch *b, xf_table *t, hipStream_t s) {
  const uint64_t uid = table_uid(t), ep = table_epoch(t);
  if (b->cells && b->cells->mode == kCellsTableRows && b->cells->table_uid == uid &&
      b->cells->epoch == ep)
    return XF_OK;
  if (b->cells) {
    XF_HIP(hipDeviceSynchronize());
    cells_free(b->cells);
    b->cells = nullptr;
  }
  xf_cells *c = nullptr;
  if (b->local) {
    XF_REQUIRE(b->raw_keys || b->NNZ == 0,
               "this minibatch was compiled against another table (or the table has renumbered "
               "its rows since) and did not keep its keys: compile it again, or with "
               "retain_keys = 1");
    XF_TRY(cells_build_keyed(&c, t, b->raw_keys, b->raw_rowptr, nullptr, b->R, b->NNZ, true, 0,
                             s));
  } else {
    XF_TRY(xf_batch_upload(b, s));
    if (!b->d_rows_u) XF_HIP(hipMalloc((void **)&b->d_rows_u, std::max<size_t>(b->U, 1) * 4));
    XF_TRY(xf_table_resolve_dev(t, b->view.ukeys, b->U, b->d_rows_u, s));
    const uint64_t M = table_dev(t).max_rows + 1;
    XF_TRY(cells_build(&c, b->view.uidx, b->d_rows_u, b->view.rowptr, b->R, b->NNZ, (uint32_t)M,
                       kCellsTableRows, true, s));
  }
  c->table_uid = uid;
  c->epoch = ep;
  b->cells = c;
  return XF_OK;
}

}  // namespace xf

// The key build of LRWorker::update (lr_worker.cc:146-166) for a table on THIS GPU, without
// the sort: every raw key is resolved straight to its state row (insert on first touch,
// growing the table when needed) and the nonzeros are grouped into cells by a stable radix
// pass on (row window, row chunk).  No unique-key list is formed: the key's state row IS its
// identity, the forward reads the table's weights in place and the gradient pass applies the
// optimizer step where it sums.  retain_keys: keep a device copy of the raw arrays so that the
// cells can be rebuilt after xf_table_defrag renumbers the rows.
namespace xf {
// defer != null: the build's host wait is left to cells_build_keyed_finish(*defer) (xf_lr_update_dev:
// the forward runs under it)
int batch_compile_local_dev(xf_batch **out, xf_table *t, const uint64_t *d_keys,
                            const uint32_t *d_rowptr, const int32_t *d_labels, uint32_t R,
                            uint32_t NNZ, int retain_keys, void *stream, KbDeferred **defer) {
  XF_REQUIRE(out && t && d_rowptr && (R == 0 || d_labels) && (NNZ == 0 || d_keys),
             "xf_batch_compile_local_dev: null argument");
  hipStream_t s = (hipStream_t)stream;
  xf_batch *b = new xf_batch;
  struct Guard {
    xf_batch *b;
    ~Guard() {
      if (b) xf_batch_free(b);
    }
  } guard{b};
  b->R = R;
  b->NNZ = NNZ;
  b->local = true;
  b->on_device_only = true;
  // labels (and, when asked, the raw keys / row offsets) in the batch's own allocation
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_lab = 0;
  const size_t o_rp = o_lab + al((size_t)R * 4);
  const size_t o_keys = o_rp + al(retain_keys ? ((size_t)R + 1) * 4 : 0);
  const size_t total = o_keys + al(retain_keys ? (size_t)NNZ * 8 : 0) + 256;
  XF_TRY(xf::blob_alloc(&b->d_raw, total, &b->d_raw_bytes));
  char *d = (char *)b->d_raw;
  if (R) XF_HIP(hipMemcpyAsync(d + o_lab, d_labels, (size_t)R * 4, hipMemcpyDeviceToDevice, s));
  b->raw_labels = (const int32_t *)(d + o_lab);
  if (retain_keys) {
    XF_HIP(hipMemcpyAsync(d + o_rp, d_rowptr, ((size_t)R + 1) * 4, hipMemcpyDeviceToDevice, s));
    if (NNZ) XF_HIP(hipMemcpyAsync(d + o_keys, d_keys, (size_t)NNZ * 8, hipMemcpyDeviceToDevice, s));
  }
  XF_TRY(xf::cells_build_keyed(&b->cells, t, d_keys, d_rowptr, nullptr, R, NNZ, retain_keys != 0,
                               0, s, defer));
  b->cells->table_uid = xf::table_uid(t);
  b->cells->epoch = xf::table_epoch(t);
  if (retain_keys) {
    b->raw_rowptr = (const uint32_t *)(d + o_rp);
    b->raw_keys = (const uint64_t *)(d + o_keys);
  }
  guard.b = nullptr;
  *out = b;
  return XF_OK;
}
}  // namespace xf

extern "C" int xf_batch_compile_local_dev(xf_batch **out, xf_table *t, const uint64_t *d_keys,
                                          const uint32_t *d_rowptr, const int32_t *d_labels,
                                          uint32_t R, uint32_t NNZ, int retain_keys,
                                          void *stream) {
  return xf::batch_compile_local_dev(out, t, d_keys, d_rowptr, d_labels, R, NNZ, retain_keys,
                                     stream, nullptr);
}

// host-array front end (the reader's block arrays and a row slice, like xf_batch_compile)
extern "C" int xf_batch_compile_local(xf_batch **out, xf_table *t, const uint64_t *rowptr,
                                      const uint64_t *keys, const int32_t *labels,
                                      size_t row_begin, size_t row_end, int retain_keys,
                                      void *stream) {
  XF_REQUIRE(out && t && rowptr && labels && row_end >= row_begin,
             "xf_batch_compile_local: bad argument");
  const size_t R = row_end - row_begin;
  const uint64_t base = rowptr[row_begin];
  const size_t NNZ = (size_t)(rowptr[row_end] - base);
  XF_REQUIRE(NNZ == 0 || keys, "xf_batch_compile_local: null keys");
  XF_REQUIRE(R < 0xFFFFFFFFull && NNZ < 0xFFFFFFFFull, "xf_batch_compile_local: batch too large");
  hipStream_t s = (hipStream_t)stream;
  std::vector<uint32_t> rp(R + 1);
  for (size_t r = 0; r <= R; ++r) rp[r] = (uint32_t)(rowptr[row_begin + r] - base);
  xf::Scratch sc;
  uint64_t *d_keys = nullptr;
  uint32_t *d_rp = nullptr;
  int32_t *d_lab = nullptr;
  XF_TRY(sc.get(&d_keys, NNZ));
  XF_TRY(sc.get(&d_rp, R + 1));
  XF_TRY(sc.get(&d_lab, R));
  if (NNZ) XF_HIP(hipMemcpyAsync(d_keys, keys + base, NNZ * 8, hipMemcpyHostToDevice, s));
  XF_HIP(hipMemcpyAsync(d_rp, rp.data(), (R + 1) * 4, hipMemcpyHostToDevice, s));
  if (R) XF_HIP(hipMemcpyAsync(d_lab, labels + row_begin, R * 4, hipMemcpyHostToDevice, s));
  XF_HIP(hipStreamSynchronize(s));
  return xf_batch_compile_local_dev(out, t, d_keys, d_rp, d_lab, (uint32_t)R, (uint32_t)NNZ,
                                    retain_keys, stream);
}

// shape of a batch's cells (tests / bench): out[0..8) = W, nwin, nchunk, G, nitems,
// segments, nsplit_chunks, M
extern "C" int xf_batch_cells_info(const xf_batch *b, uint32_t *out) {
  XF_REQUIRE(b && out, "xf_batch_cells_info: null argument");
  XF_REQUIRE(b->cells, "xf_batch_cells_info: the batch has no cells yet");
  const xf_cells *c = b->cells;
  uint32_t nchunk = 0, nitems = 0, nseg = 0;
  for (const xf_cells *q = c; q; q = q->next) {  // the segments of the batch's cells
    nchunk = std::max(nchunk, q->chunk0 + q->nchunk);
    nitems += q->nitems;
    ++nseg;
  }
  const uint32_t v[8] = {c->W, c->nwin, nchunk, c->G, nitems, nseg, xf::cells_split_chunks(c),
                         c->M};
  for (int i = 0; i < 8; ++i) out[i] = v[i];
  return XF_OK;
}
