// xf_cells_build.hip — the cells of a minibatch (xf_cells.h): the general build from index-space
// positions, the gradient's work items, the forward's position-ordered copy, and the local
// minibatch compile (gfx950).
//
// Replaces (paths relative to /root/reference):
//   key build of LRWorker::update      src/model/lr/lr_worker.cc:146-166  (cells_build; the
//                                      range-partitioned builds: xf_keybuild.hip)
// Layout and rationale: xf_cells.h.  HBM-bound integer/byte work, no MFMA.
#include <rocprim/device/device_radix_sort.hpp>

#include "xf_cells_impl.h"

namespace {
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
// most of it) is copied, by as many workgroups as it has kBlk blocks.  Not stable — the order of a position's entries is whatever
// the LDS cursors make of it: the forward's fp64 row sums do not depend on it (exact).
// Replaces rocprim::segmented_radix_sort_keys (2 x 112 us per 1e7 entries, round 2).
constexpr uint32_t kSortWave = 4096, kSortMax = 1u << 15;
constexpr uint32_t kSortLine = 128;  // entries of a cell ordered by line only (64 bins)
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
  // Neighbouring lanes take inputs `stride` apart (NT > 64: a large cell): the cell arrives in
  // row order, and a power-law head key occurs many times in ONE row — placed in that order the
  // forward's lanes would add a run of weights to the same row sum, and same-address LDS atomics
  // serialise (Zipf 1.1: the head key 20 times per row).  Taken apart here, at no cost.
  const uint32_t stride = NT > 64 ? (n + NT - 1) / NT : 1u;
  if (NT > 64 && stride > 1) {
    for (uint32_t k = 0; k < stride; ++k) {
      const uint32_t i = t * stride + k;
      if (i < n) {
        const uint32_t e = in[i];
        out[atomicAdd(&bins[sortp(e & (kChunk - 1))], 1u)] = e;
      }
    }
  } else {
    for (uint32_t i = t; i < n; i += NT) {
      const uint32_t e = in[i];
      out[atomicAdd(&bins[sortp(e & (kChunk - 1))], 1u)] = e;
    }
  }
  sync();
}

__global__ void __launch_bounds__(kBlock)
k_cells_sort_pos(const uint32_t *__restrict__ entries, uint32_t *__restrict__ out,
                 const uint32_t *__restrict__ cellptr, const uint32_t *__restrict__ blk_cell,
                 uint32_t ncell, uint32_t nsort) {
  static_assert(kBlock == 64 * kSortCells, "a wavefront per cell");
  __shared__ uint32_t bins[kSortCells][kChunk + kChunk / 32 + 8];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
  if (blockIdx.x >= nsort) {
    // The cells beyond kSortMax, kBlk entries per workgroup (a cell of a million entries — Zipf
    // 1.1 has four — was one workgroup's copy loop: 740 us).  Such a cell cannot lie inside one
    // kBlk block: the block meets it as the cell of its first entry or as the cell of the next
    // block's first entry.  Copied with its row order taken apart the way the sort does it:
    // output j = input (j % 64) * m + j / 64 over the first 64 * m entries (neighbouring lanes
    // of the forward: inputs m apart).
    const uint32_t blk = blockIdx.x - nsort;
    const uint32_t j0 = blk * kBlk, c1 = blk_cell[blk], c2 = blk_cell[blk + 1];
    for (int k = 0; k < 2; ++k) {
      const uint32_t c = k ? c2 : c1;
      if (k && c2 == c1) break;
      const uint32_t b = cellptr[c], e = cellptr[c + 1], n = e - b;
      if (n <= kSortMax) continue;
      const uint32_t m = n / 64, jb = max(b, j0), je = min(e, j0 + kBlk);
      for (uint32_t j = jb + tid; j < je; j += kBlock) {
        const uint32_t rel = j - b;
        out[j] = entries[b + (rel < 64 * m ? (rel & 63u) * m + (rel >> 6) : rel)];
      }
    }
    return;
  }
  const uint32_t c0 = blockIdx.x * kSortCells;
  {
    const uint32_t c = c0 + wave;
    if (c < ncell) {  // wave-uniform
      const uint32_t b = cellptr[c], n = cellptr[c + 1] - b;
      if (n && n <= kSortLine) {
        // A small cell (an owner's 32-window minibatch: ~50 entries): ordered by the 128-byte
        // LINE of its position — what the forward's gathers coalesce by — with 64 bins, one per
        // lane: the 2112 bins of the sort by position, cleared and scanned per cell, were most of
        // this kernel at that shape (137 us per 10^7 entries).
        uint32_t *lb = bins[wave];
        lb[lane] = 0;
        lds_wave_sync();
        for (uint32_t i = lane; i < n; i += 64) atomicAdd(&lb[(entries[b + i] >> 5) & 63u], 1u);
        lds_wave_sync();
        const uint32_t x = lb[lane];
        uint32_t inc = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const uint32_t y = __shfl_up(inc, o);
          if ((int)lane >= o) inc += y;
        }
        lb[lane] = inc - x;
        lds_wave_sync();
        for (uint32_t i = lane; i < n; i += 64) {
          const uint32_t e = entries[b + i];
          out[b + atomicAdd(&lb[(e >> 5) & 63u], 1u)] = e;
        }
        lds_wave_sync();
      } else if (n && n <= kSortWave) {
        cell_sort_pos<64>(bins[wave], entries + b, out + b, n, lane);
      }
    }
  }
  __syncthreads();
  for (uint32_t k = 0; k < (uint32_t)kSortCells && c0 + k < ncell; ++k) {  // workgroup-uniform
    const uint32_t b = cellptr[c0 + k], n = cellptr[c0 + k + 1] - b;
    if (n > kSortWave && n <= kSortMax) cell_sort_pos<kBlock>(bins[0], entries + b, out + b, n, tid);
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
}  // namespace

namespace xf {

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
  const uint32_t nsort = (c->ncell + kSortCells - 1) / kSortCells;
  hipLaunchKernelGGL(k_cells_sort_pos, dim3(nsort + c->nblk), dim3(kBlock), 0, s, c->entries,
                     c->entries_k, c->cellptr, c->blk_cell, c->ncell, nsort);
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
  c->epoch = table_epoch(t);  // (after the build: a table's first build may renumber early rows)
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
