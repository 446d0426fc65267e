// xf_keybuild.hip — the key build of LRWorker::update (src/model/lr/lr_worker.cc:146-166 of
// /root/reference) plus the key -> state row step of its Pull (lr_worker.cc:170, ftrl.h:56),
// raw keys in, cells (xf_cells.h) out, for a table that has a settled tier (gfx950).
//
// The reference sorts (fid, row) by fid and walks the sorted list against the sorted unique
// keys.  A sort of 1e7 64-bit keys is ~1 ms on this GPU and a random probe of the table per
// nonzero moves a 128-byte line for an 8-byte key (round 2: 2.2 GB and 300 us per minibatch).
// Here the nonzeros are PARTITIONED by key range instead, and the lookup happens where the
// table's keys of that range sit in LDS:
//
//   the settled tier holds the table's keys sorted, key of rank r = state row r; chunk c of the
//   index space (kChunk rows) is therefore a key RANGE [bkeys[c * kChunk], bkeys[(c+1) * kChunk)),
//   a super-chunk S is kSC chunks (8192 keys = 64 KiB of LDS).
//
//   k_kb_hist     nonzeros per cell (row window, chunk) and per (workgroup, super-chunk): the
//                 chunk of a key from the chunk boundaries and their directory (in LDS),
//                 counted in LDS histograms
//   k_kb_scan     cellptr (window-major scan of the cell counts), the gradient's work items,
//                 where every scatter workgroup's records of every super-chunk begin
//   k_kb_scatter  (key, row) records grouped by super-chunk: a tile of 8192 nonzeros is
//                 grouped in LDS and leaves in runs of neighbouring records (one partition pass
//                 into ~1200 buckets, no atomics on memory)
//   k_kb_resolve  one workgroup per super-chunk: its 8192 keys and a directory over them in
//                 LDS, every record finds its key there (position = state row), takes the next
//                 slot of its cell and becomes a 4-byte entry
//
// Table size.  The scatter's per-super-chunk arrays must fit the LDS next to its stage of
// records: 8192 nonzeros per tile up to ~1900 super-chunks = 1.5e7 settled keys per GPU
// (configs[1]: 1e7, configs[2]: 1.25e7 per GPU), 4096 per tile up to ~4200 = 3.4e7 keys.  A
// larger table (round 5; ftrl.h:84 is an unbounded map) takes the TWO-LEVEL build: the nonzeros
// are partitioned by GROUP of 2^g super-chunks first (as few groups as the full-tile scatter
// holds: the same histogram / scan / scatter kernels over coarser ranges), then every group's
// records — a few thousand, L2-resident — are split by super-chunk and counted by cell by one
// workgroup (k_kb_regroup), and the resolve runs as before:
//   k_kb_hist_groups  nonzeros per (workgroup, group)
//   k_kb_scan         (its per-range part) where every scatter workgroup's records of a group begin
//   k_kb_scatter      records grouped by group
//   k_kb_regroup      per group: records per super-chunk and per cell (row window, chunk), the
//                     records again grouped by super-chunk, the resolve's work items
//   k_kb_scan         (its cell part) cellptr, the gradient's work items
//   k_kb_resolve      unchanged
// up to 65 000 super-chunks (5e8 keys per GPU).  Other limits (beyond them the general build
// runs: a probe of the table per nonzero + a radix sort): 3.3e7 nonzeros per minibatch; 4e6
// cells.
//
// Keys the settled tier does not hold (new since the last xf_table_defrag, or the reserved key
// value) leave a hole in their cell (an entry the kernels skip) and go to a miss list; they are
// inserted by the first-touch build (round 6, "first touch" below: the same partition over
// UNIFORM key ranges, the arrival index probed where a range's positions sit in L2) and form a
// second SEGMENT of the batch's cells (xf_cells::next) over the arrival rows.  A table that
// holds nothing yet — a run's first minibatch — is settled by its build at once ("an empty
// table" below: the key ranges sorted in LDS).  In the steady state (every key settled) the
// list is empty and the build is four streaming passes and ONE host synchronisation.
// Power-law streams: whatever works on a range's or a super-chunk's records takes them in work
// items of kPart records (k_kb_resolve, k_ar_insert, k_eb_rank / k_eb_cells, k_fm_count /
// k_fm_regroup), and takes the places of a head key's records with one atomic per workgroup and
// round, not one per wavefront (same-address atomics serialise: ~10 ns each on memory).
// HBM-bound integer work, no MFMA.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "xf_batch.h"
#include "xf_cells.h"
#include "xf_device.h"
#include "xf_scratch.h"

namespace xf {
const TableDev &table_dev(const xf_table *t);
void **table_aux(xf_table *t, uint64_t **epoch);
uint64_t table_epoch(const xf_table *t);
int table_resolve_any(xf_table *t, const uint64_t *d_keys, size_t n, uint32_t *d_rows,
                      hipStream_t s, bool allow_grow);
int table_grow_for(xf_table *t, const uint64_t *d_keys, size_t n, hipStream_t s);
int table_count(xf_table *t, hipStream_t s, uint64_t *count);
int table_first_tier(xf_table *t, size_t d, uint64_t **keys);
int table_settle_first(xf_table *t, size_t d, hipStream_t s);
bool table_early_keys(xf_table *t, uint64_t count, std::vector<uint64_t> *out);
bool table_maybe_first(const xf_table *t);
void table_note_rows_out(xf_table *t);
int table_take_early(xf_table *t, const uint64_t *d_keys, size_t n, float *tw, float2 *tnz,
                     unsigned long long *d_pos, hipStream_t s);
int table_put_early(xf_table *t, const uint64_t *d_keys, size_t n, const float *tw,
                    const float2 *tnz, uint32_t *d_rows, hipStream_t s);
}  // namespace xf

namespace {

using xf::kBlk;
using xf::kChunk;
using xf::kChunkBits;
using xf::kTagShift;

constexpr int kKb = 1024;                        // threads of the workgroups here
#ifndef XF_KB_SC_SHIFT
#define XF_KB_SC_SHIFT 2
#endif
constexpr int kSCShift = XF_KB_SC_SHIFT;
constexpr uint32_t kSC = 1u << kSCShift;         // chunks per super-chunk
constexpr uint32_t kSCKeys = kSC * kChunk;       // keys per super-chunk (in LDS: 8 B each)
constexpr uint32_t kDirMax = kSCKeys / 2;        // buckets of a super-chunk's directory
constexpr uint32_t kDirStride = kDirMax + 8;     // u16 per super-chunk in the aux allocation
#ifndef XF_KB_TILE
#define XF_KB_TILE 8192
#endif
constexpr uint32_t kTile = XF_KB_TILE;           // nonzeros per scatter tile
constexpr uint32_t kPart = 32768;                // records per resolve work item
constexpr uint32_t kBatch = 8192;                // ... taken into registers at a time
constexpr uint32_t kRowSeg = 1024;               // row offsets a scatter tile keeps in LDS
constexpr uint32_t kHole = 0xFFFFFFFFu;          // an entry the step kernels skip
constexpr size_t kLdsMax = 160 * 1024;
constexpr size_t kDynMax = kLdsMax - 1024;       // dynamic LDS a kernel here may ask for
constexpr uint32_t kRinBits = 15;                // record row = window << 15 | row in window
constexpr uint32_t kLocalCells = 1024;           // cell cursors a resolve item keeps in LDS

// Ranges of the key space and the directory that finds a key's range with two LDS reads: range
// r begins at key bnd[r]; bucket(key) = mulhi32((key - lo) >> 32, mult) cuts the shard's key span
// into as many equal buckets as there are ranges (32-bit arithmetic: a 64-bit mulhi is six
// quarter-rate multiplies, and the histogram pass is VALU-bound); dir[b] = number of ranges
// that begin in buckets < b.
// A key of bucket b lies in range dir[b] - 1 + (ranges that begin in bucket b at or below it):
// the keys are uniform hashes, a bucket holds one boundary, sometimes none or two.
struct KbRanges {
  const uint64_t *bnd;   // [n]
  const uint16_t *dir;   // [n + 1]
  uint32_t n;
  uint32_t mult;
};

// a record of the scatter: key and window << 15 | row in window, 12 bytes, one store / one load
struct __attribute__((packed, aligned(4))) Rec3 {
  uint32_t klo, khi, rp;
};

// FM: the record carries two payloads — the nonzero's row (packed like Rec3::rp) for the
// key-grouped occurrence lists of the gradient, and its position in the CSR for the forward's
// per-nonzero record index
struct __attribute__((aligned(16))) Rec4 {
  uint32_t klo, khi, rp, pos;
};
constexpr uint32_t kFmRpBits = 19;  // rp of a minibatch of <= 16 row windows; above it, in the
                                    // staged word, the nonzero's place in its tile (13 bits)

// what the host reads back after the build (one copy)
struct KbSummary {
  unsigned long long miss;
  uint32_t nitems, nsplit;
};

struct KbArgs {
  const uint64_t *keys;
  const uint32_t *rowptr, *rowid;  // one of the two
  uint32_t R, NNZ, W, nwin;
  uint32_t cA, nS, ntile;          // chunks / super-chunks of the settled tier; scatter tiles
  uint32_t span, nW;               // nonzeros per histogram / scatter workgroup; workgroups
  uint32_t tile;                   // nonzeros per scatter tile: kTile, or less for big tables
  uint32_t nbase;
  const uint64_t *bkeys;           // the tier's keys, ascending; chunk c begins at bkeys[c << 11]
  uint64_t lo;
  KbRanges ch, sc;                 // chunks, super-chunks
  const uint16_t *sdirs;           // [nS * kDirStride] directory of every super-chunk's keys
  const uint64_t *smult;           // [nS] its bucket multiplier
  uint32_t *hist;                  // [nwin * cA]
  uint32_t *cellptr;               // [nwin * cA + 1]
  uint32_t *cellcur;               // [nwin * cA] next free slot of every cell
  uint32_t *scount;                // [nS] records of every super-chunk
  uint32_t *sstart;                // [nS + 1] first record of every super-chunk
  uint32_t *wgcnt;                 // [nW * nS] records of workgroup w for super-chunk S; scanned
                                   // over w: where the workgroup's share of S begins
  uint32_t *tile_r0;               // [ntile] row of every tile's first nonzero (CSR input)
  uint32_t *plan;                  // the cells' item plan (xf_cells::plan)
  uint32_t *blk_cell;              // the cells' blk_cell
  uint32_t *items;                 // resolve work items: super-chunk | part << 16
  uint32_t *nitems;                // their number
  uint32_t flags;                  // experiments (exp_knob)
  uint32_t scan_part;              // k_kb_scan: 0 all of it, 1 the per-range part, 2 the cell part
  uint32_t npc;                    // k_kb_scan: workgroups that scan the cell counts (pieces)
  uint32_t *psum;                  // [npc] cell counts per piece (k_kb_psum; npc > 1)
  // two-level build: groups of 1 << gshift super-chunks
  uint32_t gshift, nG;
  const uint32_t *gstart;          // [nG + 1] first record of every group (records by group)
  const Rec3 *grec;                // [NNZ] records grouped by group (k_kb_regroup reads them)
  Rec3 *rec;                       // [NNZ] records, grouped by super-chunk
  Rec4 *rec4;                      // FM build: the records with both payloads
  uint32_t *fm_vrow;               // FM build: [NNZ] state row of every record (kHole: a miss)
  uint32_t *fm_ridx;               // FM build: [NNZ] state row of every nonzero's key, CSR order
  uint32_t *fm_rp;                 // FM build: [NNZ] the records' rows (packed), record order
  uint32_t *entries;               // [NNZ] the cells
  uint64_t *missK;                 // [NNZ] miss list: key,
  uint32_t *missR;                 // [NNZ]            row
  KbSummary *sum;                  // device copy
  unsigned long long *dbg;         // phase timestamps (tools/kb_timeline.py), or null
};
// timing-only switches of the kernels (parts skipped, WRONG results): constants in a product build
#ifdef XF_EXPERIMENTS
#define KB_FLAG(bit) (a.flags & (bit))
#else
#define KB_FLAG(bit) false
#endif
#define KB_T(slot)                                                                   \
  do {                                                                               \
    if (a.dbg && threadIdx.x == 0) dbg_base[(slot)] = wall_clock64();                \
  } while (0)
constexpr int kDbgSlots = 24;

__device__ __forceinline__ uint32_t kb_bucket(uint64_t key, uint64_t lo, uint32_t mult,
                                              uint32_t n) {
  return key < lo ? 0u : min(__umulhi((uint32_t)((key - lo) >> 32), mult), n - 1);
}
// the bucket of a key inside one super-chunk (k_kb_sdir): sm = multiplier | shift << 32
__device__ __forceinline__ uint32_t kb_sbucket(uint64_t key, uint64_t kfirst, uint64_t sm,
                                               uint32_t nb) {
  if (key < kfirst) return 0u;
  const uint64_t d = (key - kfirst) >> (uint32_t)(sm >> 32);
  return d > 0xFFFFFFFFull ? nb - 1 : min(__umulhi((uint32_t)d, (uint32_t)sm), nb - 1);
}
// the number of ranges that begin at or below `key`, from its bucket's directory pair [s, e) and
// the bucket's first two boundaries (x0 = bnd[s], x1 = bnd[s + 1], read by the caller for all
// its keys at once)
template <typename B>
__device__ __forceinline__ uint32_t kb_count(const B &bnd, uint32_t s, uint32_t e, uint64_t x0,
                                             uint64_t x1, uint64_t key) {
  uint32_t r = s + ((s < e && x0 <= key) ? 1u : 0u) + ((s + 1 < e && x1 <= key) ? 1u : 0u);
  if (e - s > 2 && r == s + 2)
    while (r < e && bnd(r) <= key) ++r;
  return r;
}
// ... and the range the key lies in (keys below every boundary count as range 0)
template <typename B>
__device__ __forceinline__ uint32_t kb_range(const B &bnd, uint32_t s, uint32_t e, uint64_t x0,
                                             uint64_t x1, uint64_t key) {
  const uint32_t r = kb_count(bnd, s, e, x0, x1, key);
  return r ? r - 1 : 0u;
}

// bnd / dir of the chunks or the super-chunks of a settled tier (one launch per table epoch)
__global__ void __launch_bounds__(256)
k_kb_index(const uint64_t *__restrict__ bkeys, uint64_t lo, uint32_t n, int shift, uint32_t mult,
           uint64_t *__restrict__ bnd, uint16_t *__restrict__ dir) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  if (r < n) bnd[r] = bkeys[(size_t)r << shift];
  const uint32_t first = r == 0 ? 0u : kb_bucket(bkeys[(size_t)(r - 1) << shift], lo, mult, n) + 1;
  const uint32_t last = r == n ? n : kb_bucket(bkeys[(size_t)r << shift], lo, mult, n);
  for (uint32_t b = first; b <= last; ++b) dir[b] = (uint16_t)r;
}

// The directory of one super-chunk's keys (one workgroup per super-chunk, once per table
// epoch): a linear map of the super-chunk's key range onto nk / 2 buckets, dir[b] = first
// position whose key lies in a bucket >= b (kb_sbucket: any monotone map does as long as the
// look-up uses the same).
__global__ void __launch_bounds__(256)
k_kb_sdir(const uint64_t *__restrict__ bkeys, uint32_t nbase, uint16_t *__restrict__ sdirs,
          uint64_t *__restrict__ smult) {
  const uint32_t S = blockIdx.x, k0 = S * kSCKeys, nk = min(kSCKeys, nbase - k0);
  const uint64_t *__restrict__ lk = bkeys + k0;
  uint16_t *__restrict__ dir = sdirs + (size_t)S * kDirStride;
  const uint64_t kfirst = lk[0], range = lk[nk - 1] - kfirst;
  const uint32_t nb = max(nk / 2, 1u);
  // the key range shifted into 32 bits; multiplier = nb * 2^32 / (shifted range + 1)
  const uint32_t sh = range >> 32 ? 32u - (uint32_t)__clzll((long long)range) : 0u;
  const uint64_t m = ((uint64_t)nb << 32) / ((range >> sh) + 1);
  const uint64_t sm = (m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m) | ((uint64_t)sh << 32);
  if (threadIdx.x == 0) smult[S] = sm;
  auto bucket = [&](uint64_t key) -> uint32_t { return kb_sbucket(key, kfirst, sm, nb); };
  for (uint32_t i = threadIdx.x; i < nk; i += blockDim.x) {
    const uint32_t bi = bucket(lk[i]);
    const uint32_t bp = i ? bucket(lk[i - 1]) + 1 : 0u;
    for (uint32_t b = bp; b <= bi; ++b) dir[b] = (uint16_t)i;
    if (i == nk - 1)
      for (uint32_t b = bi + 1; b <= nb; ++b) dir[b] = (uint16_t)nk;
  }
}

// the largest r in [0, n) with p[r] <= j (p ascending, p[0] <= j), found by ONE wavefront: 64
// probes per round instead of a chain of log2(n) dependent loads
__device__ __forceinline__ uint32_t wave_last_le(const uint32_t *__restrict__ p, uint32_t n,
                                                 uint32_t j) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t lo = 0, hi = n;
  while (hi - lo > 1) {  // wave-uniform
    const uint32_t step = (hi - lo + 63) / 64;
    const uint32_t i = lo + lane * step;
    const bool le = i < hi && p[i] <= j;  // monotone over the lanes, lane 0 true
    const uint32_t k = (uint32_t)__popcll(__ballot(le)) - 1u;
    lo += k * step;
    hi = min(hi, lo + step);
  }
  return lo;
}

// the window of entry j (CSR order): the largest v with rowptr[min(v * W, R)] <= j
__device__ __forceinline__ uint32_t window_of_entry(const uint32_t *__restrict__ rowptr,
                                                    uint32_t R, uint32_t W, uint32_t nwin,
                                                    uint32_t j) {
  uint32_t a = 0, b = nwin;
  while (b - a > 1) {
    const uint32_t m = a + (b - a) / 2;
    if (rowptr[min((uint64_t)m * W, (uint64_t)R)] <= j) a = m;
    else
      b = m;
  }
  return a;
}

// a workgroup barrier that orders LDS accesses only: __syncthreads() also waits for the
// wavefront's outstanding global loads and stores (its fence), which is exactly what a
// prefetch in flight across the barrier must not be made to do
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <bool LDS_ONLY = false>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t x, uint32_t *wsum, uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o);
    if ((int)lane >= o) inc += t;
  }
  if (LDS_ONLY) lds_barrier();  // wsum may still be read from the previous call
  else
    __syncthreads();
  if (lane == 63) wsum[wave] = inc;
  if (LDS_ONLY) lds_barrier();
  else
    __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t w = 0; w < kKb / 64; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  *total = tot;
  return base + inc - x;
}

// ------------------------------------------------------------------------------ histogram
// Workgroup w takes the nonzeros [w * span, (w + 1) * span), span a multiple of kTile (the same
// workgroup of the scatter takes the same nonzeros).  hist[v * cA + c] = nonzeros of window v
// whose key falls into chunk c; wgcnt[w * nS + S] = nonzeros of the workgroup whose key falls
// into super-chunk S; tile_r0[t] = the row of the first nonzero of tile t (CSR input).  The
// counts of a window gather in LDS and are flushed when the workgroup's nonzeros move on to
// the next window (CSR input: windows are ranges of nonzeros; with row ids the first nonzero's
// window is the one in LDS and the others' counts go straight to memory).
__host__ __device__ inline size_t hist_lds_bytes(uint32_t cA, uint32_t nS, bool ldsb) {
  return (ldsb ? (size_t)cA * 8 + ((size_t)cA * 4 + 7) / 8 * 8 : 0) + (size_t)cA * 4 +
         (size_t)nS * 4;
}

template <bool ROWID, bool LDSB>
__global__ void __launch_bounds__(kKb)
k_kb_hist(KbArgs a) {
  extern __shared__ uint64_t smem[];
  uint64_t *lb = smem;
  uint32_t *ld2 = (uint32_t *)(smem + a.cA);  // dir[b] | dir[b + 1] << 16: one read per key
  uint32_t *lh = (uint32_t *)(smem + (LDSB ? a.cA + ((size_t)a.cA + 1) / 2 : 0));
  uint32_t *ls = lh + a.cA;
  __shared__ uint32_t s_v, s_end;
  const uint32_t tid = threadIdx.x;
  unsigned long long *dbg_base = a.dbg ? a.dbg + (size_t)blockIdx.x * kDbgSlots : nullptr;
  KB_T(0);
  const uint32_t e0 = blockIdx.x * a.span, e1 = min(e0 + a.span, a.NNZ);
  const uint64_t *__restrict__ keys = a.keys;
  const uint64_t *__restrict__ gb = a.ch.bnd;
  const uint16_t *__restrict__ gd = a.ch.dir;
  constexpr int E = 4;
  constexpr uint32_t kRound = kKb * E;
  // two register sets of keys that swap roles: the next round's keys are on their way while
  // this round's are counted (no register copies: a copy would wait for the loads)
  auto load_keys = [&](uint64_t *key, uint32_t j0) {
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t j = j0 + q * kKb + tid;
      key[q] = j < e1 ? keys[j] : 0;
    }
  };
  uint64_t keyA[E], keyB[E];
  load_keys(keyA, e0);
  if (LDSB) {
    for (uint32_t c = tid; c < a.cA; c += kKb) {
      lb[c] = gb[c];
      ld2[c] = (uint32_t)gd[c] | ((uint32_t)gd[c + 1] << 16);
    }
  }
  for (uint32_t c = tid; c < a.cA; c += kKb) lh[c] = 0;
  for (uint32_t S = tid; S < a.nS; S += kKb) ls[S] = 0;
  if (tid == 0) {
    if (ROWID) {
      s_v = a.rowid[e0] / a.W;
      s_end = e1;
    } else {
      const uint32_t v = window_of_entry(a.rowptr, a.R, a.W, a.nwin, e0);
      s_v = v;
      s_end = v + 1 < a.nwin ? a.rowptr[min((uint64_t)(v + 1) * a.W, (uint64_t)a.R)] : a.NNZ;
    }
  }
  if (!ROWID) {  // a wavefront per tile of the workgroup: the row of the tile's first nonzero
    const uint32_t wave = tid >> 6, ntl = (e1 - e0 + a.tile - 1) / a.tile;
    for (uint32_t k = wave; k < ntl; k += kKb / 64) {
      const uint32_t r = wave_last_le(a.rowptr, a.R, e0 + k * a.tile);
      if ((tid & 63u) == 0) a.tile_r0[e0 / a.tile + k] = r;
    }
  }
  __syncthreads();
  KB_T(1);
  uint32_t v0 = s_v, wend = min(s_end, e1);  // the window in LDS; its nonzeros end at wend
  auto bnd = [&](uint32_t c) -> uint64_t { return LDSB ? lb[c] : gb[c]; };
  auto dir2 = [&](uint32_t b) -> uint32_t {
    return LDSB ? ld2[b] : (uint32_t)gd[b] | ((uint32_t)gd[b + 1] << 16);
  };
  auto process = [&](const uint64_t *key, uint32_t j0) {
    if (ROWID) {
      // nonzeros with row numbers arrive worker after worker, rows ascending: when a round begins
      // in another window than the one in LDS, that window's counts are flushed and the new one
      // moves in (every lane reads the round's first row: a uniform load, no barrier unless the
      // window changes).  Before, whatever followed the workgroup's first window went to memory
      // one atomic per nonzero: the workgroups that straddle a window's end — one in eight at 32
      // windows — were the kernel's tail (85 us where the CSR form takes 36).
      const uint32_t r = a.rowid[j0], off = r - v0 * a.W;
      const uint32_t nv = off < a.W ? v0 : r / a.W;
      if (nv != v0) {  // workgroup-uniform
        __syncthreads();
        for (uint32_t c = tid; c < a.cA; c += kKb) {
          const uint32_t n = lh[c];
          if (n) {
            atomicAdd(&a.hist[(size_t)v0 * a.cA + c], n);
            atomicAdd(&ls[c >> kSCShift], n);
          }
          lh[c] = 0;
        }
        __syncthreads();
        v0 = nv;
      }
    }
    if (!ROWID && j0 >= wend) {  // (workgroup-uniform) the nonzeros have left the window
      __syncthreads();
      for (uint32_t c = tid; c < a.cA; c += kKb) {
        const uint32_t n = lh[c];
        if (n) {
          atomicAdd(&a.hist[(size_t)v0 * a.cA + c], n);
          atomicAdd(&ls[c >> kSCShift], n);
        }
        lh[c] = 0;
      }
      if (tid == 0) {
        const uint32_t v = window_of_entry(a.rowptr, a.R, a.W, a.nwin, j0);
        s_v = v;
        s_end = v + 1 < a.nwin ? a.rowptr[min((uint64_t)(v + 1) * a.W, (uint64_t)a.R)] : a.NNZ;
      }
      __syncthreads();
      v0 = s_v;
      wend = min(s_end, e1);
    }
    uint64_t x0[E], x1[E];
    uint32_t dp[E];
#pragma unroll
    for (int q = 0; q < E; ++q) dp[q] = dir2(kb_bucket(key[q], a.lo, a.ch.mult, a.cA));
#pragma unroll
    for (int q = 0; q < E; ++q) {
      x0[q] = bnd(min(dp[q] & 0xFFFFu, a.cA - 1));
      x1[q] = bnd(min((dp[q] & 0xFFFFu) + 1, a.cA - 1));
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t j = j0 + q * kKb + tid;
      if (j >= e1) continue;
      const uint32_t c = kb_range(bnd, dp[q] & 0xFFFFu, dp[q] >> 16, x0[q], x1[q], key[q]);
      bool in_lds;
      uint32_t v = v0;
      if (ROWID) {  // (rows ascend within a worker's share: the window rarely changes — no
                    // division while it does not)
        const uint32_t r = a.rowid[j], off = r - v0 * a.W;
        in_lds = off < a.W;
        if (!in_lds) v = r / a.W;
      } else {
        in_lds = j < wend;  // (a round that straddles the window's end: its tail, rare)
        if (!in_lds) v = window_of_entry(a.rowptr, a.R, a.W, a.nwin, j);
      }
      if (KB_FLAG(64)) continue;
      if (in_lds) {
        atomicAdd(&lh[c], 1u);
      } else {
        atomicAdd(&a.hist[(size_t)v * a.cA + c], 1u);
        atomicAdd(&ls[c >> kSCShift], 1u);
      }
    }
  };
  for (uint32_t j0 = e0; j0 < e1; j0 += 2 * kRound) {
    if (j0 + kRound < e1) load_keys(keyB, j0 + kRound);
    process(keyA, j0);
    if (j0 + kRound >= e1) break;
    if (j0 + 2 * kRound < e1) load_keys(keyA, j0 + 2 * kRound);
    process(keyB, j0 + kRound);
  }
  KB_T(2);
  __syncthreads();
  KB_T(3);
  if (!KB_FLAG(1))
    for (uint32_t c = tid; c < a.cA; c += kKb) {
      const uint32_t n = lh[c];
      if (n) {
        atomicAdd(&a.hist[(size_t)v0 * a.cA + c], n);
        atomicAdd(&ls[c >> kSCShift], n);
      }
    }
  __syncthreads();
  uint32_t *__restrict__ out = a.wgcnt + (size_t)blockIdx.x * a.nS;
  for (uint32_t S = tid; S < a.nS; S += kKb) out[S] = ls[S];
  KB_T(4);
}

// ----------------------------------------------------------------------------------- scan
// Workgroup 0: cellptr = exclusive scan of hist (window-major), cellcur = its copy (the resolve
// hands out the slots of a cell of a split super-chunk from it).  Workgroup 1: the gradient's
// work items (what k_plan_items of xf_cells.hip computes, from the histogram).  The other
// workgroups: one wavefront per super-chunk S, exclusive scan of wgcnt[. * nS + S] over the
// scatter workgroups (where workgroup w's records of S begin, relative to the super-chunk's
// first record) and scount[S] = the super-chunk's records.
// The scans of workgroups 0 and 1 go through LDS in pieces of kScanPiece elements: coalesced
// loads in, a thread scans 16 neighbouring elements (17-word stride: no bank conflicts),
// coalesced stores out.
constexpr uint32_t kScanPiece = 16 * kKb;
constexpr uint32_t kPlanWgs = 3;  // k_kb_scan: workgroups of the item plan (a scan each)
__device__ __forceinline__ uint32_t sp(uint32_t i) { return i + (i >> 4); }

template <typename F>
__device__ __forceinline__ uint32_t staged_excl_scan(F in, uint32_t n, uint32_t *__restrict__ o1,
                                                     uint32_t *__restrict__ o2, uint32_t *sbuf,
                                                     uint32_t *wsum, uint32_t carry0 = 0) {
  const uint32_t tid = threadIdx.x;
  uint32_t carry = carry0;
  for (uint32_t base = 0; base < n; base += kScanPiece) {
    const uint32_t m = min(kScanPiece, n - base);
    for (uint32_t i = tid; i < kScanPiece; i += kKb) sbuf[sp(i)] = i < m ? in(base + i) : 0u;
    __syncthreads();
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += sbuf[sp(tid * 16 + k)];
    uint32_t total;
    uint32_t run = carry + block_excl_scan(sum, wsum, &total);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t x = sbuf[sp(tid * 16 + k)];
      sbuf[sp(tid * 16 + k)] = run;
      run += x;
    }
    __syncthreads();
    for (uint32_t i = tid; i < m; i += kKb) {
      const uint32_t x = sbuf[sp(i)];
      o1[base + i] = x;
      if (o2) o2[base + i] = x;
    }
    carry += total;
    __syncthreads();
  }
  return carry;
}

// the cell counts of piece p (kScanPiece cells), for the scan's workgroups to start from
// ... and, by the workgroups beyond the pieces, the gradient's slices per chunk (the item plan's
// input): kKb chunks per workgroup.  (In the scan's plan workgroup — a load per window, its wait,
// an add, 48 times over for the 48 829 chunks of a 10^8-key table — they were 100 of that
// kernel's 157 us.)
__global__ void __launch_bounds__(kKb)
k_kb_psum(KbArgs a) {
  __shared__ uint32_t wsum[kKb / 64];
  if (blockIdx.x >= a.npc) {
    const uint32_t c = (blockIdx.x - a.npc) * kKb + threadIdx.x;
    if (c > a.cA) return;
    uint32_t n = 0;
    if (c < a.cA)
      for (uint32_t v = 0; v < a.nwin; ++v) n += a.hist[(size_t)v * a.cA + c];
    a.plan[c] = (n + xf::kSliceMax - 1) / xf::kSliceMax;  // (plan[cA] = 0)
    return;
  }
  const uint32_t ncell = a.nwin * a.cA, p = blockIdx.x;
  const uint32_t b = p * kScanPiece, e = min(b + kScanPiece, ncell);
  uint32_t sum = 0;
  for (uint32_t i = b + threadIdx.x; i < e; i += kKb) sum += a.hist[i];
  uint32_t total;
  (void)block_excl_scan(sum, wsum, &total);
  if (threadIdx.x == 0) a.psum[p] = total;
}

// Workgroups [0, npc): the cell counts, a piece of kScanPiece cells each (one workgroup walking
// 195 000 cells — an owner's 32 row windows — took 77 us).  Workgroup npc: the work items.  The
// others: the per-range columns.
__global__ void __launch_bounds__(kKb)
k_kb_scan(KbArgs a) {
  __shared__ uint32_t sbuf[kScanPiece + kScanPiece / 16];
  __shared__ uint32_t wsum[kKb / 64];
  const uint32_t tid = threadIdx.x;
  const uint32_t *__restrict__ hist = a.hist;
  if (blockIdx.x >= a.npc + kPlanWgs ? a.scan_part == 2 : a.scan_part == 1) return;  // (two-level build)
  if (blockIdx.x >= a.npc + kPlanWgs) {
    const uint32_t S = (blockIdx.x - a.npc - kPlanWgs) * (kKb / 64) + (tid >> 6), lane = tid & 63u;
    if (S >= a.nS) return;
    uint32_t *__restrict__ col = a.wgcnt + S;
    uint32_t carry = 0;
    constexpr int kCol = 4;  // 256 scatter workgroups at most: every load of the column first
    uint32_t x[kCol];
#pragma unroll
    for (int q = 0; q < kCol; ++q) {
      const uint32_t w = q * 64 + lane;
      x[q] = w < a.nW ? col[(size_t)w * a.nS] : 0u;
    }
#pragma unroll
    for (int q = 0; q < kCol; ++q) {
      const uint32_t w = q * 64 + lane;
      uint32_t inc = x[q];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(inc, o);
        if ((int)lane >= o) inc += y;
      }
      if (w < a.nW) col[(size_t)w * a.nS] = carry + inc - x[q];
      carry += (uint32_t)__shfl((int)inc, 63);
    }
    for (uint32_t w0 = kCol * 64; w0 < a.nW; w0 += 64) {  // (more workgroups than that: never)
      const uint32_t w = w0 + lane;
      const uint32_t y0 = w < a.nW ? col[(size_t)w * a.nS] : 0u;
      uint32_t inc = y0;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(inc, o);
        if ((int)lane >= o) inc += y;
      }
      if (w < a.nW) col[(size_t)w * a.nS] = carry + inc - y0;
      carry += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) a.scount[S] = carry;
    return;
  }
  if (blockIdx.x >= a.npc) {  // slices per chunk, first item of every chunk, index among the split
    // (three workgroups, a scan each — unless one packed scan does for two of them: a 10^8-key
    // table's 48 829 chunks, three scans one behind the other, were 84 us)
    const uint32_t which = blockIdx.x - a.npc;
    const size_t nc1 = (size_t)a.cA + 1;
    uint32_t *nsl = a.plan, *off = a.plan + nc1, *soff = a.plan + 2 * nc1;
    auto slices = [&](uint32_t c) -> uint32_t {
      // (eight windows' counts requested at a time: one after the other — a load, its wait, an
      // add — the 32 windows of an owner's minibatch made this workgroup the kernel: 51 us)
      uint32_t n = 0, v = 0;
      for (; v + 8 <= a.nwin; v += 8) {
        uint32_t x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = hist[(size_t)(v + k) * a.cA + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) n += x[k];
      }
      for (; v < a.nwin; ++v) n += hist[(size_t)v * a.cA + c];
      return (n + xf::kSliceMax - 1) / xf::kSliceMax;
    };
    if (a.npc <= 1) {  // (else k_kb_psum has them)
      for (uint32_t c = tid; c < a.cA; c += kKb) nsl[c] = slices(c);
      if (tid == 0) nsl[a.cA] = 0;
    }
    __syncthreads();  // (its fence: the slices are in L2 before this workgroup reads them back)
    const volatile uint32_t *vn = nsl;
    // both scans in one pass: the slices' prefix in a word's low 20 bits, the split chunks'
    // in its high 12, whenever both fit (else one pass each)
    uint32_t ta, tb;
    const bool packed = a.cA < (1u << 12) && a.NNZ / xf::kSliceMax + a.cA < (1u << 20);
    if (packed && which == 1) return;  // (workgroup 0 scans both)
    uint32_t *poff = a.plan + 3 * nc1;
    if (which == 2) {
      // the split chunks' slices, scanned: they go first in the item list (k_items_fill)
      const uint32_t tp = staged_excl_scan(
          [&](uint32_t c) { const uint32_t S = vn[c]; return S > 1 ? S : 0u; }, a.cA, poff, nullptr,
          sbuf, wsum);
      if (tid == 0) poff[a.cA] = tp;
      return;
    }
    if (packed) {
      const uint32_t t = staged_excl_scan(
          [&](uint32_t c) { const uint32_t S = vn[c]; return S | ((S > 1 ? 1u : 0u) << 20); },
          a.cA, off, nullptr, sbuf, wsum);
      for (uint32_t c = tid; c < a.cA; c += kKb) {  // (this thread's own elements: unpack)
        const uint32_t v = ((const volatile uint32_t *)off)[c];
        off[c] = v & 0xFFFFFu;
        soff[c] = v >> 20;
      }
      ta = t & 0xFFFFFu;
      tb = t >> 20;
    } else if (which == 0) {
      ta = staged_excl_scan([&](uint32_t c) { return (uint32_t)vn[c]; }, a.cA, off, nullptr, sbuf,
                            wsum);
      if (tid == 0) {
        off[a.cA] = ta;
        a.sum->nitems = ta;
      }
      return;
    } else {
      tb = staged_excl_scan([&](uint32_t c) { return vn[c] > 1 ? 1u : 0u; }, a.cA, soff, nullptr,
                            sbuf, wsum);
      if (tid == 0) {
        soff[a.cA] = tb;
        a.sum->nsplit = tb;
      }
      return;
    }
    if (tid == 0) {  // (the packed scan's two totals)
      off[a.cA] = ta;
      soff[a.cA] = tb;
      a.sum->nitems = ta;
      a.sum->nsplit = tb;
    }
    return;
  }
  const uint32_t ncell = a.nwin * a.cA, p = blockIdx.x;
  const uint32_t b = p * kScanPiece, m = min(kScanPiece, ncell - b);
  uint32_t base = 0;
  if (a.npc > 1) {  // the cells before this piece
    uint32_t x = 0;
    for (uint32_t q = tid; q < p; q += kKb) x += a.psum[q];
    (void)block_excl_scan(x, wsum, &base);
    __syncthreads();  // (wsum is used again below)
  }
  const uint32_t total = staged_excl_scan([&](uint32_t c) { return hist[b + c]; }, m,
                                          a.cellptr + b, a.cellcur + b, sbuf, wsum, base);
  if (p == a.npc - 1 && tid == 0) a.cellptr[ncell] = total;
}

// ------------------------------------------------------------- two-level build, level one
// Nonzeros per (workgroup, group): a.sc holds the GROUP ranges here, a.nS their number.  Same
// split of the nonzeros over the workgroups as the scatter's; also tile_r0 (CSR input).
__host__ __device__ inline size_t hist_groups_lds_bytes(uint32_t nG) {
  return (size_t)nG * 8 + (size_t)nG * 4 + (size_t)nG * 4 + 64;
}

template <bool ROWID>
__global__ void __launch_bounds__(kKb)
k_kb_hist_groups(KbArgs a) {
  extern __shared__ uint64_t smem[];
  uint64_t *lb = smem;                        // group boundaries
  uint32_t *ld2 = (uint32_t *)(smem + a.nS);  // dir[b] | dir[b + 1] << 16
  uint32_t *ls = ld2 + a.nS;                  // counts
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t e0 = blockIdx.x * a.span, e1 = min(e0 + a.span, a.NNZ);
  const uint64_t *__restrict__ keys = a.keys;
  for (uint32_t g = tid; g < a.nS; g += kKb) {
    lb[g] = a.sc.bnd[g];
    ld2[g] = (uint32_t)a.sc.dir[g] | ((uint32_t)a.sc.dir[g + 1] << 16);
    ls[g] = 0;
  }
  if (!ROWID) {  // a wavefront per tile of the workgroup: the row of the tile's first nonzero
    const uint32_t wave = tid >> 6, ntl = (e1 - e0 + a.tile - 1) / a.tile;
    for (uint32_t k = wave; k < ntl; k += kKb / 64) {
      const uint32_t r = wave_last_le(a.rowptr, a.R, e0 + k * a.tile);
      if (lane == 0) a.tile_r0[e0 / a.tile + k] = r;
    }
  }
  __syncthreads();
  auto bnd = [&](uint32_t g) -> uint64_t { return lb[g]; };
  constexpr int E = 4;
  for (uint32_t j0 = e0; j0 < e1; j0 += kKb * E) {
    uint64_t key[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t j = j0 + q * kKb + tid;
      key[q] = j < e1 ? keys[j] : 0;
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t j = j0 + q * kKb + tid;
      const uint32_t dp = ld2[kb_bucket(key[q], a.lo, a.sc.mult, a.nS)];
      const uint32_t s0 = dp & 0xFFFFu, s1 = dp >> 16;
      const uint32_t g = kb_range(bnd, s0, s1, lb[min(s0, a.nS - 1)], lb[min(s0 + 1, a.nS - 1)],
                                  key[q]);
      // (one LDS atomic per key: a wavefront's 64 keys fall into ~60 of the ~1500 groups — a
      // round of ballots per distinct group, the owner partition's trick for EIGHT owners, made
      // this kernel 383 us)
      if (j < e1) atomicAdd(&ls[g], 1u);
    }
  }
  __syncthreads();
  uint32_t *__restrict__ out = a.wgcnt + (size_t)blockIdx.x * a.nS;
  for (uint32_t g = tid; g < a.nS; g += kKb) out[g] = ls[g];
}

// ------------------------------------------------------------- two-level build, level two
// One workgroup per group g = super-chunks [g << gshift, ...): its records [gstart[g],
// gstart[g + 1]) of a.grec (a few thousand: read twice, the second time from L2).  Pass 1 counts
// them per super-chunk and per cell (row window, chunk) — a record's chunk from the group's
// chunk boundaries in LDS — and leaves scount / sstart of its super-chunks, their resolve work
// items and its chunks' columns of the cell histogram in memory (a group owns them: no
// atomics on memory).  Pass 2 writes the records grouped by super-chunk into a.rec.
constexpr uint32_t kGrpShiftMax = 6;
__host__ __device__ inline size_t regroup_lds_bytes(uint32_t gshift, uint32_t nwin) {
  const size_t ncg = (size_t)kSC << gshift;
  return ncg * 8 + ncg * nwin * 4 + ((size_t)1 << gshift) * 12 + 64;
}

__global__ void __launch_bounds__(kKb)
k_kb_regroup(KbArgs a) {
  extern __shared__ uint64_t smem[];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t g = blockIdx.x, nSg_max = 1u << a.gshift;
  const uint32_t S0 = g << a.gshift, nSg = min(nSg_max, a.nS - S0);
  const uint32_t c0 = S0 << kSCShift, ncg = min(nSg << kSCShift, a.cA - c0);
  uint64_t *cb = smem;                                         // [ncg] chunk boundaries
  uint32_t *ccnt = (uint32_t *)(smem + ((size_t)kSC << a.gshift));  // [nwin * ncg]
  uint32_t *scnt = ccnt + (size_t)a.nwin * ((size_t)kSC << a.gshift);  // [nSg] counts
  uint32_t *soff = scnt + nSg_max;                             // [nSg] first record (relative)
  uint32_t *scur = soff + nSg_max;                             // [nSg] cursors
  const uint32_t rb = a.gstart[g], re = a.gstart[g + 1];
  for (uint32_t c = tid; c < ncg; c += kKb) cb[c] = a.ch.bnd[c0 + c];
  for (uint32_t i = tid; i < a.nwin * ncg; i += kKb) ccnt[i] = 0;
  if (tid < nSg_max) scnt[tid] = scur[tid] = 0;
  __syncthreads();
  const Rec3 *__restrict__ src = a.grec;
  // the chunk of a key among the group's: the last c with cb[c] <= key (0 below all of them)
  auto chunk_of = [&](uint64_t key) -> uint32_t {
    uint32_t lo = 0, hi = ncg;  // cb[lo] <= key < cb[hi] (cb[0] taken as -inf)
    while (hi - lo > 1) {
      const uint32_t m = lo + (hi - lo) / 2;
      if (cb[m] <= key) lo = m;
      else
        hi = m;
    }
    return lo;
  };
  for (uint32_t i0 = rb; i0 < re; i0 += kKb) {
    const uint32_t i = i0 + tid;
    if (i < re) {
      const Rec3 r = src[i];
      const uint64_t key = (uint64_t)r.klo | ((uint64_t)r.khi << 32);
      const uint32_t c = chunk_of(key), v = r.rp >> kRinBits;
      atomicAdd(&ccnt[(size_t)v * ncg + c], 1u);
      atomicAdd(&scnt[c >> kSCShift], 1u);
    }
  }
  __syncthreads();
  if (tid == 0) {  // (at most 64 super-chunks)
    uint32_t run = 0;
    for (uint32_t j = 0; j < nSg; ++j) {
      soff[j] = run;
      run += scnt[j];
    }
  }
  __syncthreads();
  if (tid < nSg) {
    const uint32_t n = scnt[tid], S = S0 + tid;
    a.scount[S] = n;
    a.sstart[S] = rb + soff[tid];
    const uint32_t parts = (n + kPart - 1) / kPart;
    if (parts) {
      const uint32_t at = atomicAdd(a.nitems, parts);
      for (uint32_t q = 0; q < parts; ++q) a.items[at + q] = S | (q << 16);
    }
  }
  if (g == gridDim.x - 1 && tid == 0) a.sstart[a.nS] = re;
  for (uint32_t i = tid; i < a.nwin * ncg; i += kKb) {
    const uint32_t v = i / ncg, c = i - v * ncg;
    a.hist[(size_t)v * a.cA + c0 + c] = ccnt[i];
  }
  Rec3 *__restrict__ dst = a.rec;
  for (uint32_t i0 = rb; i0 < re; i0 += kKb) {
    const uint32_t i = i0 + tid;
    const bool ok = i < re;
    Rec3 r{0u, 0u, 0u};
    uint32_t sl = 0xFFFFFFFFu;
    if (ok) {
      r = src[i];
      sl = chunk_of((uint64_t)r.klo | ((uint64_t)r.khi << 32)) >> kSCShift;
    }
    unsigned long long todo = __ballot(ok);
    uint32_t slot = 0;
    while (todo) {  // wave-uniform: one cursor atomic per super-chunk the wavefront holds
      const int l = __ffsll((long long)todo) - 1;
      const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)sl, l);
      const unsigned long long m = __ballot(ok && sl == cur);
      uint32_t base = 0;
      if ((int)lane == l) base = atomicAdd(&scur[cur], (uint32_t)__popcll(m));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, l);
      if (ok && sl == cur) slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      todo &= ~m;
    }
    if (ok) dst[rb + soff[sl] + slot] = r;
  }
}

// -------------------------------------------------------------------------------- scatter
// Workgroup w walks the tiles of its nonzeros (the histogram's split).  Per tile of kTile
// nonzeros: every nonzero's super-chunk (boundaries and their directory in LDS) and its rank
// among the tile's nonzeros of that super-chunk (an LDS counter); the records laid out by
// super-chunk in LDS and written out in that order — neighbouring lanes write neighbouring
// records — to the workgroup's share of every super-chunk's record space (from the scan, moved
// on tile by tile: no atomics on memory, same-address ones serialise at ~0.1 us each).  A
// thread takes kTile / 1024 consecutive nonzeros (one row search, then a walk); the next tile's
// keys are on their way while this one's records are written.  Workgroup 0 also leaves
// sstart[] and the resolve's work items in memory.
struct ScatterLds {
  uint64_t *sb, *stK;
  uint32_t *cnt, *lofs, *base, *stR, *rowseg, *sd2, *tr;
  uint16_t *stB;
};
constexpr uint32_t kMaxSub = 32;  // tiles per scatter workgroup
__host__ __device__ inline size_t scatter_lds_bytes(uint32_t nS, uint32_t tile) {
  return (size_t)nS * 8 + (size_t)tile * 8 +
         ((size_t)nS * 4 + 1 + tile + 2 * (kRowSeg + 2) + kMaxSub + 2) * 4 + (size_t)tile * 2;
}

// TILE: kTile, or a half / a quarter of it when the table has so many super-chunks that their
// per-tile arrays would not fit the LDS next to a full stage (shorter runs of records, more
// barriers per nonzero: a big table's price)
template <bool ROWID, uint32_t TILE, bool FM = false>
__global__ void __launch_bounds__(kKb)
k_kb_scatter(KbArgs a) {
  extern __shared__ uint64_t smem[];
  ScatterLds L;
  L.sb = smem;
  L.stK = L.sb + a.nS;
  L.cnt = (uint32_t *)(L.stK + TILE);
  L.lofs = L.cnt + a.nS;
  L.base = L.lofs + a.nS + 1;
  L.sd2 = L.base + a.nS;  // dir[b] | dir[b + 1] << 16
  L.stR = L.sd2 + a.nS;
  L.rowseg = L.stR + TILE;  // two buffers: this tile's and the next one's
  L.tr = L.rowseg + 2 * (kRowSeg + 2);
  L.stB = (uint16_t *)(L.tr + kMaxSub + 2);
  __shared__ uint32_t wsum[kKb / 64];
  const uint32_t tid = threadIdx.x;
  unsigned long long *dbg_base =
      a.dbg ? a.dbg + ((size_t)a.nW + blockIdx.x) * kDbgSlots : nullptr;
  KB_T(0);
  int dslot = 1;
  const uint32_t w0 = blockIdx.x * a.span, w1 = min(w0 + a.span, a.NNZ);
  const uint32_t ntl = (w1 - w0 + TILE - 1) / TILE;  // <= kMaxSub
  const uint64_t *__restrict__ keys = a.keys;
  const uint32_t *__restrict__ rowptr = a.rowptr;
  Rec3 *__restrict__ rec = a.rec;
  constexpr int E = (int)(TILE / kKb);
  static_assert(E % 4 == 0, "a thread's keys are read in pairs and looked up four at a time");
  static_assert(TILE <= 8192 && TILE >= 4 * kKb, "rank and super-chunk share a word: 13 bits of rank");
  static_assert(kRowSeg + 2 <= 2 * kKb, "a thread stages two row offsets");
  auto load_keys = [&](uint32_t e0, uint64_t *key) {  // keys[e0 + tid * E .. + E), 0 past w1
    const uint32_t j0 = e0 + tid * E;
    if (j0 + E <= w1 && ((uintptr_t)(keys + j0) & 15u) == 0) {
#pragma unroll
      for (int q = 0; q < E; q += 2) {
        const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(keys + j0 + q);
        key[q] = t.x;
        key[q + 1] = t.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < E; ++q) key[q] = j0 + q < w1 ? keys[j0 + q] : 0;
    }
  };
  uint64_t key[E], nkey[E];
  load_keys(w0, key);
  {
    const uint64_t *__restrict__ gb = a.sc.bnd;
    const uint16_t *__restrict__ gd = a.sc.dir;
    const uint32_t *__restrict__ wc = a.wgcnt + (size_t)blockIdx.x * a.nS;
    const uint32_t *__restrict__ cn = a.scount;
    for (uint32_t S = tid; S < a.nS; S += kKb) {
      L.sb[S] = gb[S];
      L.sd2[S] = (uint32_t)gd[S] | ((uint32_t)gd[S + 1] << 16);
    }
    if (!ROWID)  // the first row of every tile of the workgroup, and of the tile after them
      for (uint32_t k = tid; k <= ntl; k += kKb) {
        const uint32_t t = w0 / TILE + k;
        L.tr[k] = t < a.ntile ? a.tile_r0[t] : a.R - 1;
      }
    // sstart = exclusive scan of the super-chunks' record counts
    const uint32_t per = (a.nS + kKb - 1) / kKb;
    const uint32_t s0 = min(tid * per, a.nS), s1 = min(s0 + per, a.nS);
    uint32_t sum = 0, nit = 0;
    for (uint32_t S = s0; S < s1; ++S) {
      const uint32_t n = cn[S];
      sum += n;
      nit += (n + kPart - 1) / kPart;
    }
    uint32_t total, itotal = 0;
    uint32_t run = block_excl_scan(sum, wsum, &total);
    uint32_t irun = blockIdx.x == 0 ? block_excl_scan(nit, wsum, &itotal) : 0u;
    for (uint32_t S = s0; S < s1; ++S) {
      const uint32_t n = cn[S];
      L.base[S] = run + wc[S];
      if (blockIdx.x == 0) {
        a.sstart[S] = run;
        const uint32_t parts = (n + kPart - 1) / kPart;
        for (uint32_t q = 0; q < parts; ++q) a.items[irun + q] = S | (q << 16);
        irun += parts;
      }
      run += n;
    }
    if (blockIdx.x == 0 && tid == 0) {
      a.sstart[a.nS] = total;
      *a.nitems = itotal;
    }
  }
  __syncthreads();
  // rowptr[r0 .. r1 + 1] of the rows tile k touches go into row-offset buffer k & 1 (when they
  // fit: nseg <= kRowSeg + 2; else the tile searches rowptr in memory)
  auto seg_of = [&](uint32_t k, uint32_t *r0, uint32_t *nseg) {
    *r0 = L.tr[k];
    *nseg = L.tr[k + 1] - L.tr[k] + 2;
  };
  if (!ROWID) {
    uint32_t r0, nseg;
    seg_of(0, &r0, &nseg);
    if (nseg <= kRowSeg + 2)
      for (uint32_t i = tid; i < nseg; i += kKb) L.rowseg[i] = rowptr[r0 + i];
  }
  auto bnd = [&](uint32_t S) -> uint64_t { return L.sb[S]; };
  KB_T(dslot++);
  for (uint32_t k = 0; k < ntl; ++k) {
    const uint32_t e0 = w0 + k * TILE, n = min(TILE, w1 - e0);
    const bool more = k + 1 < ntl;
    uint32_t r0 = 0, nseg = 0, nr0 = 0, nnseg = 0, nrs[2] = {0, 0};
    if (!ROWID) seg_of(k, &r0, &nseg);
    const bool seg_lds = !ROWID && nseg <= kRowSeg + 2;
    const uint32_t *seg = L.rowseg + (k & 1) * (kRowSeg + 2);
    // the next tile's keys and row offsets: issued now, taken over (registers / LDS) before this
    // tile's records are stored — loads and stores retire through one in-order counter, a wait
    // for a load issued after the stores would drain the stores
    if (more) {
      load_keys(e0 + TILE, nkey);
      if (!ROWID) {
        seg_of(k + 1, &nr0, &nnseg);
        if (nnseg <= kRowSeg + 2) {
          if (tid < nnseg) nrs[0] = rowptr[nr0 + tid];
          if (tid + kKb < nnseg) nrs[1] = rowptr[nr0 + tid + kKb];
        }
      }
    }
    for (uint32_t S = tid; S < a.nS; S += kKb) L.cnt[S] = 0;
    lds_barrier();
    if (dslot < kDbgSlots - 8) KB_T(dslot++);
    uint32_t rp[E], sr[E];  // sr = super-chunk << 13 | rank among the tile's nonzeros of it
#pragma unroll
    for (int h = 0; h < E; h += 4) {  // (four at a time: the registers)
      uint64_t x0[4], x1[4];
      uint32_t dp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) dp[q] = L.sd2[kb_bucket(key[h + q], a.lo, a.sc.mult, a.nS)];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        x0[q] = L.sb[min(dp[q] & 0xFFFFu, a.nS - 1)];
        x1[q] = L.sb[min((dp[q] & 0xFFFFu) + 1, a.nS - 1)];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        sr[h + q] = kb_range(bnd, dp[q] & 0xFFFFu, dp[q] >> 16, x0[q], x1[q], key[h + q]);
    }
    const uint32_t i0 = tid * E;  // the thread's first nonzero of the tile
    if (i0 < n) {
      if (ROWID) {
        // one division per thread and tile: its first nonzero's window; the others are in that
        // window or (where a window ends inside the thread's run) divide for themselves
        // (rowid == null: a nonzero's "row" is its position — the (key, position) sort's payload)
        const uint32_t *__restrict__ rid = a.rowid;
        const uint32_t r0w = (rid ? rid[e0 + i0] : e0 + i0) / a.W;
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const uint32_t r = i0 + q < n ? (rid ? rid[e0 + i0 + q] : e0 + i0 + q) : r0w * a.W;
          uint32_t v = r0w, rin = r - r0w * a.W;
          if (rin >= a.W) {
            v = r / a.W;
            rin = r - v * a.W;
          }
          rp[q] = (v << kRinBits) | rin;
        }
      } else {
        const uint32_t j0 = e0 + i0;
        uint32_t lo = 0, hi = nseg - 1;  // the last r in the segment with rowptr[r] <= j0
        uint32_t next;                   // rowptr[row + 1]
        if (seg_lds) {
          while (hi - lo > 1) {
            const uint32_t m = lo + (hi - lo) / 2;
            if (seg[m] <= j0) lo = m;
            else
              hi = m;
          }
          next = seg[lo + 1];
        } else {
          while (hi - lo > 1) {
            const uint32_t m = lo + (hi - lo) / 2;
            if (rowptr[r0 + m] <= j0) lo = m;
            else
              hi = m;
          }
          next = rowptr[r0 + lo + 1];
        }
        uint32_t row = r0 + lo;
        uint32_t v = row / a.W, rin = row - v * a.W;
#pragma unroll
        for (int q = 0; q < E; ++q) {
          const uint32_t j = j0 + q;
          while (j >= next && row + 1 < a.R) {  // (rows without nonzeros are stepped over)
            ++row;
            next = seg_lds ? seg[row + 1 - r0] : rowptr[row + 1];
            if (++rin == a.W) {
              rin = 0;
              ++v;
            }
          }
          rp[q] = (v << kRinBits) | rin;
        }
      }
#pragma unroll
      for (int q = 0; q < E; ++q)
        sr[q] = (sr[q] << 13) | (i0 + q < n ? atomicAdd(&L.cnt[sr[q]], 1u) : 0u);
    }
    if (dslot < kDbgSlots - 8) KB_T(dslot++);
    lds_barrier();
    if (dslot < kDbgSlots - 8) KB_T(dslot++);
    {  // lofs = exclusive scan of cnt
      const uint32_t per = (a.nS + kKb - 1) / kKb;
      const uint32_t s0 = min(tid * per, a.nS), s1 = min(s0 + per, a.nS);
      uint32_t sum = 0;
      for (uint32_t S = s0; S < s1; ++S) sum += L.cnt[S];
      uint32_t total;
      uint32_t run = block_excl_scan<true>(sum, wsum, &total);
      for (uint32_t S = s0; S < s1; ++S) {
        L.lofs[S] = run;
        run += L.cnt[S];
      }
    }
    lds_barrier();
    if (dslot < kDbgSlots - 8) KB_T(dslot++);
#pragma unroll
    for (int q = 0; q < E; ++q) {
      if (i0 + q >= n) continue;
      const uint32_t S = sr[q] >> 13, p = L.lofs[S] + (sr[q] & 8191u);
      L.stK[p] = key[q];
      L.stR[p] = FM ? rp[q] | ((i0 + q) << kFmRpBits) : rp[q];
      L.stB[p] = (uint16_t)S;
    }
    if (more) {  // take the prefetch over: the loads were issued a tile's work ago
#pragma unroll
      for (int q = 0; q < E; ++q) key[q] = nkey[q];
      if (!ROWID && nnseg <= kRowSeg + 2) {
        uint32_t *nseg_buf = L.rowseg + ((k + 1) & 1) * (kRowSeg + 2);
        if (tid < nnseg) nseg_buf[tid] = nrs[0];
        if (tid + kKb < nnseg) nseg_buf[tid + kKb] = nrs[1];
      }
    }
    if (dslot < kDbgSlots - 8) KB_T(dslot++);
    lds_barrier();
    if (dslot < kDbgSlots - 8) KB_T(dslot++);
    if (!KB_FLAG(2))
      for (uint32_t i = tid; i < n; i += kKb) {
        const uint32_t S = L.stB[i];
        const uint64_t kk = L.stK[i];
        if (FM)
          a.rec4[L.base[S] + (i - L.lofs[S])] =
              Rec4{(uint32_t)kk, (uint32_t)(kk >> 32), L.stR[i] & ((1u << kFmRpBits) - 1u),
                   e0 + (L.stR[i] >> kFmRpBits)};
        else
          rec[L.base[S] + (i - L.lofs[S])] = Rec3{(uint32_t)kk, (uint32_t)(kk >> 32), L.stR[i]};
      }
    if (dslot < kDbgSlots - 8) KB_T(dslot++);
    lds_barrier();
    if (dslot < kDbgSlots - 8) KB_T(dslot++);
    for (uint32_t S = tid; S < a.nS; S += kKb) L.base[S] += L.cnt[S];
  }
  KB_T(kDbgSlots - 1);
}

// -------------------------------------------------------------------------------- resolve
// Every lane with `on` adds one to cnt[l] and learns the count before it (WANT) — the lanes that
// share the first lane's key with ONE atomic: a wavefront of a head key's records is one LDS
// operation instead of 64 on one address.
template <bool WANT>
__device__ __forceinline__ uint32_t lds_add_shared(uint32_t *cnt, uint32_t l, bool on) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long m_on = __ballot(on);
  if (!m_on) return 0;  // wave-uniform
  const int leader = __ffsll((long long)m_on) - 1;
  const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)l, leader);
  const bool same = on && l == l0;
  const unsigned long long ms = __ballot(same);
  uint32_t base = 0;
  if ((int)lane == leader) {
    if (WANT) base = atomicAdd(&cnt[l0], (uint32_t)__popcll(ms));
    else
      atomicAdd(&cnt[l0], (uint32_t)__popcll(ms));
  }
  if (WANT) base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
  if (same) return base + (uint32_t)__popcll(ms & ((1ull << lane) - 1ull));
  if (on) {
    if (WANT) return atomicAdd(&cnt[l], 1u);
    atomicAdd(&cnt[l], 1u);
  }
  return 0;
}

// Work item = up to kPart records of one super-chunk, whose keys sit in LDS with the
// directory over them (k_kb_sdir).  A record finds its key with the directory pair and the
// bucket's first two keys (a plain binary search would put the 64 lanes' probes of its first
// steps on ONE bank: 130 us of bank conflicts per minibatch); the position IS the state row.
// It takes the next free slot of its cell (lanes of a wavefront that fill the same cell share
// one atomic — on a cursor in LDS when the item has the super-chunk to itself) and becomes
// entry = tag | row in window | position in chunk.  A key the tier does not hold leaves a hole
// and joins the miss list.  The workgroups beyond the last item compute blk_cell.
constexpr int kRes = 512;  // threads per resolve workgroup (two of them per CU)
__global__ void __launch_bounds__(kRes, 4)
k_kb_resolve(KbArgs a) {
  __shared__ uint64_t lk[kSCKeys];
  __shared__ uint16_t dir[kDirStride];
  __shared__ uint32_t lcur[kLocalCells];
  __shared__ uint32_t s_mcnt;
  __shared__ unsigned long long s_mbase;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t nitems = *a.nitems;
  if (blockIdx.x >= nitems) {  // blk_cell[b] = the largest cell with cellptr[cell] <= kBlk * b
    const uint32_t ncell = a.nwin * a.cA, nblk = (a.NNZ + kBlk - 1) / kBlk;
    const uint32_t spare = gridDim.x - nitems;
    for (uint32_t b = (blockIdx.x - nitems) * kRes + tid; b <= nblk; b += spare * kRes) {
      uint32_t lo = 0, hi = ncell;  // cellptr[0] = 0
      if (b == nblk) lo = ncell - 1;
      else
        while (hi - lo > 1) {
          const uint32_t m = lo + (hi - lo) / 2;
          if (a.cellptr[m] <= b * kBlk) lo = m;
          else
            hi = m;
        }
      a.blk_cell[b] = lo;
    }
    return;
  }
  unsigned long long *dbg_base =
      a.dbg ? a.dbg + ((size_t)2 * a.nW + blockIdx.x) * kDbgSlots : nullptr;
  KB_T(0);
  int dslot = 2;
  const uint32_t item = a.items[blockIdx.x];
  const uint32_t S = item & 0xFFFFu, part = item >> 16;
  const uint32_t sb = a.sstart[S], se = a.sstart[S + 1];
  const uint32_t rb = sb + part * kPart, re = min(se, rb + kPart);
  const uint32_t k0 = S * kSCKeys, nk = min(kSCKeys, a.nbase - k0);
  const uint32_t nb = max(nk / 2, 1u);
  const bool local = se - sb <= kPart && a.nwin * kSC <= kLocalCells;  // workgroup-uniform
  // a super-chunk of several items (a power-law head's): the items share its cells' cursors in
  // memory — a round's records are counted per cell in LDS and the workgroup takes their places
  // with ONE atomic per cell (one per wavefront and cell, all on the few cursors of the head
  // key's cells, made this kernel 323 us on the Zipf(1.1) stream: 115 on the uniform one)
  const uint32_t nlc = a.nwin * kSC;
  const bool shared = !local && nlc <= kLocalCells / 2;
  uint32_t *const lbase = lcur + kLocalCells / 2;
  uint32_t *__restrict__ entries = a.entries;
  constexpr int E = 4, kRounds = (int)(kBatch / (kRes * E));
  // A batch of kBatch records goes into registers at once (16 per thread): the rounds below end
  // in scattered stores, and on this GPU loads and stores retire through one in-order counter —
  // a wait for a load issued after a round's stores would drain the stores
  Rec3 rec[kRounds * E];
  auto load_batch = [&](uint32_t b0) {
    const Rec3 *__restrict__ src = a.rec + b0;
#pragma unroll
    for (int q = 0; q < kRounds * E; ++q) {
      const uint32_t i = q * kRes + tid;
      rec[q] = i < re - b0 ? src[i] : Rec3{0u, 0u, 0u};
    }
  };
  {
    const ulonglong2 *__restrict__ src = (const ulonglong2 *)(a.bkeys + k0);  // 16-byte aligned
    for (uint32_t i = tid; i < (nk + 1) / 2; i += kRes) {  // (bkeys is padded past nbase)
      const ulonglong2 t = src[i];
      lk[2 * i] = t.x;
      lk[2 * i + 1] = t.y;
    }
    const uint32_t *__restrict__ gd = (const uint32_t *)(a.sdirs + (size_t)S * kDirStride);
    uint32_t *ld = (uint32_t *)dir;
    for (uint32_t i = tid; i < (nb + 2 + 1) / 2; i += kRes) ld[i] = gd[i];
    if (local)
      for (uint32_t i = tid; i < a.nwin * kSC; i += kRes) {
        const uint32_t c = (S << kSCShift) + (i & (kSC - 1));
        lcur[i] = c < a.cA ? a.cellptr[(size_t)(i >> kSCShift) * a.cA + c] : 0u;
      }
    if (shared)
      for (uint32_t i = tid; i < nlc; i += kRes) lcur[i] = 0;
    if (tid == 0) s_mcnt = 0;
  }
  const uint64_t sm = a.smult[S];
  // the records after the keys: loads come back in order, the look-ups of the first round can
  // start when the keys and the first records are there, the other records still on their way
  load_batch(rb);
  lds_barrier();
  KB_T(1);
  const uint64_t kfirst = lk[0];
  auto lkf = [&](uint32_t i) -> uint64_t { return lk[i]; };
  for (uint32_t b0 = rb; b0 < re; b0 += kBatch) {
  if (b0 != rb) load_batch(b0);  // (an item of a heavy super-chunk: several batches)
#pragma unroll
  for (int g = 0; g < kRounds; ++g) {
    const uint32_t i0 = b0 + g * kRes * E;
    if (i0 >= re) break;  // workgroup-uniform
    if (dslot < kDbgSlots - 4) KB_T(dslot++);
    uint64_t key[E], x0[E], x1[E];
    uint32_t rp[E], cell[E], ent[E], ds[E], de[E], at[E] = {0u, 0u, 0u, 0u};
    bool ok[E], miss[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      ok[q] = i0 + q * kRes + tid < re;
      key[q] = (uint64_t)rec[g * E + q].klo | ((uint64_t)rec[g * E + q].khi << 32);
      rp[q] = rec[g * E + q].rp;
      const uint32_t b = kb_sbucket(key[q], kfirst, sm, nb);
      ds[q] = dir[b];
      de[q] = dir[b + 1];
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
      x0[q] = lk[min(ds[q], nk - 1)];
      x1[q] = lk[min(ds[q] + 1, nk - 1)];
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
      // ub = number of the super-chunk's keys <= key
      uint32_t ub = key[q] < kfirst ? 0u : kb_count(lkf, ds[q], de[q], x0[q], x1[q], key[q]);
      if (KB_FLAG(8)) ub = 1;
      const uint32_t p = ub ? ub - 1 : 0u;  // the largest position with lk <= key
      const bool found = ok[q] && ub > 0 && lk[p] == key[q];
      const uint32_t cl = p >> kChunkBits, c = (S << kSCShift) + cl;
      const uint32_t v = rp[q] >> kRinBits, rin = rp[q] & ((1u << kRinBits) - 1u);
      cell[q] = local || shared ? (v << kSCShift) + cl : v * a.cA + c;
      ent[q] = found ? ((c & xf::kTagMask) << kTagShift) | (rin << kChunkBits) | (p & (kChunk - 1))
                     : kHole;
      miss[q] = ok[q] && !found && !KB_FLAG(8);
    }
    if (dslot < kDbgSlots - 4) KB_T(dslot++);
    if (KB_FLAG(4)) {
#pragma unroll
      for (int q = 0; q < E; ++q)
        if (ok[q]) entries[i0 + q * kRes + tid] = ent[q];
      continue;
    }
    // slots: the lanes of the wavefront that fill the same cell share one atomic.  A
    // wavefront's records mostly come from ONE row window, i.e. from the kSC cells of that
    // window: kSC ballots (wave-uniform masks and counts), lane cl adds cell cl's count to the
    // cell's cursor, every lane takes its base from that lane.  Records of other windows
    // (where the tiles of two windows meet) go through a loop over their cells.
    if (local && !KB_FLAG(16)) {
      // every lane takes its slot with its own LDS atomic: 64 lanes on the ~4 cursors of a
      // window's cells serialise in the LDS, but the ~40 VALU instructions per record of the
      // ballot rounds below go — the kernel is VALU-bound after its load burst: 122 -> 113 us
      // (exp_knob 116 runs the ballot rounds)
#pragma unroll
      for (int q = 0; q < E; ++q)
        if (ok[q]) entries[atomicAdd(&lcur[cell[q]], 1u)] = ent[q];
    } else if (shared) {
#pragma unroll
      for (int q = 0; q < E; ++q)  // (rank in the round; the slots after the barriers below)
        at[q] = lds_add_shared<true>(lcur, cell[q], ok[q]);  // (a head key's records: one cell)
    } else
#pragma unroll
    for (int q = 0; q < E; ++q) {
      unsigned long long todo = __ballot(ok[q]);
      if (!todo) continue;  // wave-uniform
      uint32_t slot = 0;
      bool placed = false;
      if (local) {  // cell = window * kSC + chunk: the first lane's window
        const int l0 = __ffsll((long long)todo) - 1;
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)cell[q], l0) & ~(kSC - 1);
        const uint32_t cl = cell[q] - w0;  // < kSC for the lanes of that window
        const bool in_w = ok[q] && cl < kSC;
        unsigned long long mine = 0;
        uint32_t cnt = 0;  // on lane i < kSC: records of the wavefront for cell w0 + i
#pragma unroll
        for (uint32_t i = 0; i < kSC; ++i) {
          const unsigned long long m = __ballot(in_w && cl == i);
          if (cl == i) mine = m;
          if (lane == i) cnt = (uint32_t)__popcll(m);
          todo &= ~m;
        }
        uint32_t base = 0;
        if (lane < kSC && cnt) base = atomicAdd(&lcur[w0 + lane], cnt);
        const uint32_t b0 = (uint32_t)__shfl((int)base, (int)(in_w ? cl : 0u));
        if (in_w) {
          slot = b0 + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
          placed = true;
        }
      }
      while (todo) {  // wave-uniform: the cells not settled above, one at a time
        const int l = __ffsll((long long)todo) - 1;
        const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)cell[q], l);
        const bool me = ok[q] && !placed && cell[q] == lc;
        const unsigned long long m = __ballot(me);
        uint32_t base = 0;
        if ((int)lane == l)
          base = local ? atomicAdd(&lcur[lc], (uint32_t)__popcll(m))
                       : atomicAdd(&a.cellcur[lc], (uint32_t)__popcll(m));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, l);
        if (me) {
          slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
          placed = true;
        }
        todo &= ~m;
      }
      if (ok[q]) entries[slot] = ent[q];
    }
    // the miss list: the round's misses are counted in LDS and the workgroup takes their places
    // with ONE atomic on the list's end (same-address atomics on memory serialise: a minibatch
    // with 1.4e6 misses, one atomic per wavefront and slot, made this kernel 1.9 ms instead of
    // 0.11 — round 6's first trace of a table's first epoch)
    uint32_t moff[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const unsigned long long m = __ballot(miss[q]);
      moff[q] = 0;
      if (!m) continue;  // wave-uniform
      const int leader = __ffsll((long long)m) - 1;
      uint32_t at = 0;
      if ((int)lane == leader) at = atomicAdd(&s_mcnt, (uint32_t)__popcll(m));
      moff[q] = (uint32_t)__builtin_amdgcn_readlane((int)at, leader) +
                (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    lds_barrier();
    if (tid == 0) {
      const uint32_t n = s_mcnt;
      s_mbase = n ? atomicAdd(&a.sum->miss, (unsigned long long)n) : 0ull;
      s_mcnt = 0;  // (the next round's: nobody adds before the barrier below)
    }
    if (shared)
      for (uint32_t c = tid; c < nlc; c += kRes) {
        const uint32_t n = lcur[c];
        if (n) {
          lbase[c] = atomicAdd(
              &a.cellcur[(size_t)(c >> kSCShift) * a.cA + (S << kSCShift) + (c & (kSC - 1))], n);
          lcur[c] = 0;
        }
      }
    lds_barrier();
    if (shared)
#pragma unroll
      for (int q = 0; q < E; ++q)
        if (ok[q]) entries[lbase[cell[q]] + at[q]] = ent[q];
    const unsigned long long mbase = s_mbase;
#pragma unroll
    for (int q = 0; q < E; ++q)
      if (miss[q]) {
        a.missK[mbase + moff[q]] = key[q];
        a.missR[mbase + moff[q]] = (rp[q] >> kRinBits) * a.W + (rp[q] & ((1u << kRinBits) - 1u));
      }
  }
  }
  KB_T(kDbgSlots - 1);
}


// --------------------------------------------------------------------- first touch (round 6)
// Keys the settled tier does not hold — every key of a run's first minibatches (ftrl.h:56 inserts
// on the first Pull, lr_worker.cc:183-188 starts from an empty store).  Until round 5 they went
// through a probe of the 24-bytes-per-key arrival index per nonzero (a 128-byte line from HBM
// each: 457 us per 1e7) and two rocPRIM radix passes on the cell number.  Now the same partition
// idea as above, with UNIFORM key ranges (the arrival index is order-preserving: a key range is a
// range of index positions):
//   k_ar_ranges                       range r = bucket r of kb_bucket (boundaries + directory,
//                                     so that the partition kernels above serve unchanged)
//   k_kb_hist_groups / _scan / _scatter   (key, row) records grouped by key range
//   k_ar_insert   one workgroup per range (parts of a heavy one): its records probe the arrival
//                 index — the range's positions, L2-resident while the workgroup runs — with an
//                 insert on miss; ONE atomic on the table's row counter per batch of 4096 records,
//                 the new keys' rows consecutive; every record's state row is written out and
//                 counted into its cell (row window, chunk of arrival rows)
//   k_kb_scan     (cell part) cellptr, the gradient's work items
//   k_ar_place    every record takes the next slot of its cell and becomes an entry
// No sort, no probe that leaves the L2, no library call.
struct ArArgs {
  xf::TableDev T;
  const Rec3 *rec;         // [n] records grouped by key range
  const uint32_t *sstart;  // [nR + 1]
  const uint32_t *items;   // range | part << 16
  const uint32_t *nitems;
  uint32_t n, nwin, nchunk, chunk0;
  const uint64_t *bnd;     // [nR] first key of every range
  uint32_t nR;
  uint32_t *rec_row;       // [n] state row of every record
  uint32_t *hist;          // [nwin * nchunk]
  uint32_t *cellcur;       // [nwin * nchunk] next free slot of every cell (after the scan)
  const uint32_t *cellptr; // [nwin * nchunk + 1]
  uint32_t *entries;       // [n]
  uint32_t *blk_cell;      // [n / kBlk + 1]
};

__global__ void __launch_bounds__(256)
k_ar_ranges(uint64_t lo, uint32_t n, uint32_t mult, uint64_t *__restrict__ bnd,
            uint16_t *__restrict__ dir) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  dir[r] = (uint16_t)r;  // ranges that begin in buckets < r
  // the smallest h = (key - lo) >> 32 with mulhi32(h, mult) >= r
  if (r < n) bnd[r] = lo + ((r == 0 ? 0ull : (((uint64_t)r << 32) + mult - 1) / mult) << 32);
}

// A wavefront's records fall into a few cells (a range's new keys get consecutive rows): the
// lanes that hold the same cell share one atomic on the cell's counter; after kArRounds distinct
// cells the lanes left over go one by one.
constexpr int kArRounds = 6;
// ... counting: nobody reads what the atomics return, so they are issued one behind the other
// (with the returns read — one trip to the L2 per round, eight slots of six rounds per thread —
// this was most of k_ar_insert's 0.7-0.9 ms per 1e7 records)
__device__ __forceinline__ void ar_cell_count(uint32_t *__restrict__ counter, bool on,
                                              uint32_t cell) {
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long todo = __ballot(on);
  bool placed = !on;
  for (int round = 0; todo && round < kArRounds; ++round) {  // wave-uniform
    const int l = __ffsll((long long)todo) - 1;
    const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)cell, l);
    const bool me = !placed && cell == lc;
    const unsigned long long m = __ballot(me);
    if ((int)lane == l) atomicAdd(&counter[lc], (uint32_t)__popcll(m));
    placed = placed || me;
    todo &= ~m;
  }
  if (!placed) atomicAdd(&counter[cell], 1u);
}
// ... taking slots: the rounds' atomics do not depend on one another (only the ballots do, and
// those are register work), so all of them are in flight before the first return is read
__device__ __forceinline__ uint32_t ar_cell_slot(uint32_t *__restrict__ counter, bool on,
                                                 uint32_t cell) {
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long todo = __ballot(on);
  uint32_t ret[kArRounds], rank = 0;
  int lead[kArRounds], mine = -1;
#pragma unroll
  for (int round = 0; round < kArRounds; ++round) {
    ret[round] = 0;
    lead[round] = 0;
    if (!todo) continue;  // wave-uniform
    const int l = __ffsll((long long)todo) - 1;
    const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)cell, l);
    const bool me = on && mine < 0 && cell == lc;
    const unsigned long long m = __ballot(me);
    if ((int)lane == l) ret[round] = atomicAdd(&counter[lc], (uint32_t)__popcll(m));
    lead[round] = l;
    if (me) {
      mine = round;
      rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    todo &= ~m;
  }
  uint32_t slot = 0;
  if (on && mine < 0) slot = atomicAdd(&counter[cell], 1u);
#pragma unroll
  for (int round = 0; round < kArRounds; ++round) {
    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)ret[round], lead[round]);
    if (mine == round) slot = base + rank;
  }
  return slot;
}

constexpr int kAr = 512;                       // threads of an insert workgroup (a thread's eight
                                               // windows in flight: ~150 registers)
constexpr int kArE = 8;                        // records per thread and batch
constexpr uint32_t kArBatch = kAr * kArE;      // records whose rows are handed out together
constexpr int kArWin = 4;                      // index positions read per probe round

// LOCAL (the first launch): a range whose records are ONE work item has its slice of the index —
// positions [home(bnd[S]), home(bnd[S + 1])) — to itself: nobody else claims a position there, so
// its claims and its row publications are WORKGROUP-scope operations (no waiting protocol: a
// barrier orders the batch's publications and reads).  Where the kernel's 0.95 ms per 1e7 first
// touches go, measured with its parts switched off (tools/r6/call12.sh): the records in and the
// rows out 54 us, the probe windows 246 (4e7 divergent 8-byte loads), the 6.3e6 claims ~480
// (a compare-and-swap per new key, whatever its scope), the cell counts 170.  A record
// whose probe leaves the slice (a cluster across the slice's end, a home on the boundary) and
// every record of a range cut into several items are DEFERRED: marked in rec_row and taken by the
// second launch (LOCAL = false: agent scope, the waiting protocol), after the first has finished.
constexpr uint32_t kArDeferred = 0xFFFFFFFEu;

template <bool LOCAL>
__global__ void __launch_bounds__(kAr)
k_ar_insert(ArArgs a) {
  __shared__ uint32_t s_new, s_cur;
  __shared__ unsigned long long s_base;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  if (blockIdx.x >= *a.nitems) return;
  const xf::TableDev &T = a.T;
  const uint32_t item = a.items[blockIdx.x];
  const uint32_t S = item & 0xFFFFu, part = item >> 16;
  const uint32_t sb = a.sstart[S], se = a.sstart[S + 1];
  const uint32_t rb = sb + part * kPart, re = min(se, rb + kPart);
  uint64_t h0 = 0, h1 = T.cap;  // the positions this workgroup may claim
  if (LOCAL) {
    if (se - sb > kPart) {  // a range of several items: all of it to the second launch
      for (uint32_t i = rb + tid; i < re; i += kAr) a.rec_row[i] = kArDeferred;
      return;
    }
    h0 = xf::home_of(T, a.bnd[S]);
    h1 = S + 1 < a.nR ? xf::home_of(T, a.bnd[S + 1]) : T.cap;
  }
  constexpr int kScope = LOCAL ? __HIP_MEMORY_SCOPE_WORKGROUP : __HIP_MEMORY_SCOPE_AGENT;
  auto claim = [&](uint64_t i, uint64_t key) -> uint64_t {  // the key the position holds now
    unsigned long long expect = xf::kEmptyKey;
    __hip_atomic_compare_exchange_strong((unsigned long long *)&T.keys[i], &expect,
                                         (unsigned long long)key, __ATOMIC_RELAXED,
                                         __ATOMIC_RELAXED, kScope);
    return expect;  // (kEmptyKey: the claim succeeded)
  };
  for (uint32_t b0 = rb; b0 < re; b0 += kArBatch) {  // workgroup-uniform
    uint64_t key[kArE];
    uint32_t rp[kArE], pos[kArE], row[kArE];
    bool ok[kArE], ins[kArE], bad[kArE];
    int any = 0;
#pragma unroll
    for (int q = 0; q < kArE; ++q) {
      const uint32_t i = b0 + q * kAr + tid;
      ok[q] = i < re && (LOCAL || a.rec_row[i] == kArDeferred);
      any |= ok[q] ? 1 : 0;
    }
    if (tid == 0) {
      s_new = 0;
      s_cur = 0;
    }
    if (!__syncthreads_or(any)) continue;  // (the second launch: nothing deferred in this batch)
#pragma unroll
    for (int q = 0; q < kArE; ++q) {
      const uint32_t i = b0 + q * kAr + tid;
      const Rec3 r = ok[q] ? a.rec[i] : Rec3{0u, 0u, 0u};
      key[q] = (uint64_t)r.klo | ((uint64_t)r.khi << 32);
      rp[q] = r.rp;
    }
    // (a) every record finds its key's position in the arrival index, or claims an empty one.
    // The thread's eight first probe windows are requested together, then its eight claims
    // (independent atomics).  A record whose first window holds neither its key nor an empty
    // position (or whose claim another key won) walks on alone.
    uint32_t mine = 0;
    uint64_t home[kArE], cur[kArE][kArWin];
    uint32_t slot[kArE];  // position to claim (kArWin: none), or the window's hit
    uint32_t lim[kArE];   // LOCAL: positions from the home on that lie inside the slice
    bool more[kArE];      // the walk goes on from home + start[q]
    bool defer[kArE];
    uint32_t start[kArE];
#pragma unroll
    for (int q = 0; q < kArE; ++q) {
      ins[q] = bad[q] = more[q] = defer[q] = false;
      pos[q] = (uint32_t)T.cap;
      row[q] = (uint32_t)T.max_rows;  // the write-off row
      slot[q] = kArWin;
      start[q] = 0;
      home[q] = 0;
      lim[q] = 0xFFFFFFFFu;
      const bool plain = ok[q] && key[q] != xf::kEmptyKey && xf::owns(T, key[q]);
      if (ok[q] && !plain) {
        if (key[q] == xf::kEmptyKey) {  // the reserved value lives at the spare position
          ins[q] = atomicExch(&T.stat->spare_used, 1u) == 0u;
          if (ins[q]) T.keys[T.cap] = key[q];
        } else {
          atomicOr(&T.stat->err, xf::kErrForeignKey);
          bad[q] = true;
        }
      }
      if (plain) home[q] = xf::home_of(T, key[q]);
      more[q] = plain;
      if (LOCAL && plain)
        lim[q] = home[q] >= h0 && home[q] < h1 ? (uint32_t)std::min<uint64_t>(h1 - home[q], 0xFFFFFFFFull)
                                               : 0u;
#pragma unroll
      for (int t = 0; t < kArWin; ++t) {
        uint64_t i = home[q] + t;
        if (i >= T.cap) i -= T.cap;
        cur[q][t] = plain && (uint32_t)t < lim[q] ? T.keys[i] : 0ull;
      }
    }
    bool hit[kArE];  // the window holds the key at slot[q] (no run-time index into cur[][]: an
                     // array indexed so lives in scratch memory)
#pragma unroll
    for (int q = 0; q < kArE; ++q) {
      hit[q] = false;
      if (!more[q]) continue;
#pragma unroll
      for (int t = kArWin - 1; t >= 0; --t)  // (the first hit or empty position of the window)
        if ((uint32_t)t < lim[q] && (cur[q][t] == key[q] || cur[q][t] == xf::kEmptyKey)) {
          slot[q] = (uint32_t)t;
          hit[q] = cur[q][t] == key[q];
        }
      if (slot[q] == kArWin && lim[q] <= (uint32_t)kArWin) {  // the slice ends inside the window
        defer[q] = true;
        more[q] = false;
      }
    }
    uint64_t won[kArE];
#pragma unroll
    for (int q = 0; q < kArE; ++q) {
      won[q] = 0;
      if (!more[q] || slot[q] == kArWin) continue;
      uint64_t i = home[q] + slot[q];
      if (i >= T.cap) i -= T.cap;
      if (hit[q]) {
        pos[q] = (uint32_t)i;
        more[q] = false;
      } else {
        // (a stale EMPTY read — somebody inserted meanwhile — is corrected by the returned value)
        won[q] = claim(i, key[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < kArE; ++q) {
      if (!more[q]) continue;
      if (slot[q] == kArWin) {
        start[q] = kArWin;  // a window of other keys: walk on behind it
        continue;
      }
      uint64_t i = home[q] + slot[q];
      if (i >= T.cap) i -= T.cap;
      if (won[q] == xf::kEmptyKey || won[q] == key[q]) {
        ins[q] = won[q] == xf::kEmptyKey;
        pos[q] = (uint32_t)i;
        more[q] = false;
      } else {
        start[q] = slot[q] + 1;  // another key took the position: walk on behind it
      }
    }
#pragma unroll
    for (int q = 0; q < kArE; ++q) {
      if (more[q]) {  // (rare at load <= 0.6: one position after the other)
        bool done = false;
        uint64_t p = home[q] + start[q];
        if (p >= T.cap) p -= T.cap;
        for (uint64_t probes = start[q]; probes < T.cap && !done; ++probes) {
          if (LOCAL && probes >= lim[q]) {  // the walk leaves the slice: the second launch's
            defer[q] = true;
            done = true;
            break;
          }
          uint64_t c = T.keys[p];
          if (c == xf::kEmptyKey) {
            c = claim(p, key[q]);
            if (c == xf::kEmptyKey) {
              ins[q] = true;
              c = key[q];
            }
          }
          if (c == key[q]) {
            pos[q] = (uint32_t)p;
            done = true;
          }
          if (++p >= T.cap) p -= T.cap;
        }
        if (!done) {
          atomicOr(&T.stat->err, xf::kErrFull);
          bad[q] = true;
        }
      }
      if (defer[q]) {
        a.rec_row[b0 + q * kAr + tid] = kArDeferred;
        ok[q] = false;
      }
      mine += ins[q] ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0 && mine) atomicAdd(&s_new, mine);
    __syncthreads();
    if (tid == 0) s_base = s_new ? atomicAdd(&T.stat->count, (unsigned long long)s_new) : 0ull;
    __syncthreads();
    // (b) consecutive rows for the batch's new keys, published at once.  Second launch: a record
    // of another workgroup that met the key in this launch waits for the row below, and nobody
    // waits before publishing.  First launch: only this workgroup reads them, after the barrier.
    const unsigned long long base = s_base;
#pragma unroll
    for (int q = 0; q < kArE; ++q) {
      const unsigned long long m = __ballot(ins[q]);
      if (!m) continue;  // wave-uniform
      const int l = __ffsll((long long)m) - 1;
      uint32_t at = 0;
      if ((int)lane == l) at = atomicAdd(&s_cur, (uint32_t)__popcll(m));
      at = (uint32_t)__builtin_amdgcn_readlane((int)at, l);
      if (ins[q]) {
        const unsigned long long r = base + at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (r < T.max_rows) {
          row[q] = (uint32_t)r;
          if (T.init_kind != XF_INIT_ZERO) {  // (the memory is pre-zeroed for XF_INIT_ZERO)
            float *dst = T.w + (size_t)row[q] * T.dim;
            for (int j = 0; j < T.dim; ++j)
              dst[j] = T.init_kind == XF_INIT_CONST ? T.init_const
                                                    : xf::hashnorm(T.seed, key[q], (uint32_t)j);
          }
        } else {
          atomicOr(&T.stat->err, xf::kErrFull);
        }
        __hip_atomic_store(&T.rows[pos[q]], row[q], __ATOMIC_RELAXED, kScope);
      }
    }
    if (LOCAL) __syncthreads();  // (the rows of this batch's new keys: stored, visible to the CU)
    // (c) the rows of the keys that were there (or were inserted by somebody else just now)
#pragma unroll
    for (int q = 0; q < kArE; ++q)  // (all of the thread's rows requested before the first wait)
      if (ok[q] && !ins[q] && !bad[q])
        row[q] = __hip_atomic_load(&T.rows[pos[q]], __ATOMIC_RELAXED, kScope);
#pragma unroll
    for (int q = 0; q < kArE; ++q) {
      if (!ok[q]) continue;
      if (!ins[q] && !bad[q]) {
        uint32_t r = row[q];
        for (int spin = 0; !LOCAL && r == xf::kNoRow && spin < (1 << 22); ++spin) {
          __builtin_amdgcn_s_sleep(1);
          r = __hip_atomic_load(&T.rows[pos[q]], __ATOMIC_RELAXED, kScope);
        }
        if (r == xf::kNoRow) {
          atomicOr(&T.stat->err, xf::kErrDupKey);
          r = (uint32_t)T.max_rows;
        }
        row[q] = r;
      }
      a.rec_row[b0 + q * kAr + tid] = row[q];
    }
#pragma unroll
    for (int q = 0; q < kArE; ++q) {
      const uint32_t cell =
          (rp[q] >> kRinBits) * a.nchunk + ((row[q] >> kChunkBits) - a.chunk0);
      ar_cell_count(a.hist, ok[q], cell);
    }
    __syncthreads();  // (s_new / s_cur are reset by the next batch)
  }
}

// How many NEW keys would these records bring?  (The table grows before the inserts when they
// would push its load past 0.6: ftrl.h:84 is an unbounded map.)  Per work item: every record's
// key is looked for in the arrival index, read-only; the absent ones go into a set in LDS, whose
// size is the item's count — duplicates of a new key inside an item count once, a key of a range
// cut into several items once per item, a set that has filled up counts what it cannot hold:
// an upper bound that is the number itself for hashed keys.  Replaces a 64-bit radix sort of
// every key of the minibatch (count_distinct, xf_table.hip: still there for the sort-based build).
constexpr uint32_t kAbsSlots = 8192;
__global__ void __launch_bounds__(kAr)
k_ar_absent(ArArgs a, unsigned long long *__restrict__ out) {
  __shared__ unsigned long long lset[kAbsSlots];
  __shared__ uint32_t s_cnt;
  const uint32_t tid = threadIdx.x;
  if (blockIdx.x >= *a.nitems) return;
  const xf::TableDev &T = a.T;
  const uint32_t item = a.items[blockIdx.x];
  const uint32_t S = item & 0xFFFFu, part = item >> 16;
  const uint32_t sb = a.sstart[S], se = a.sstart[S + 1];
  const uint32_t rb = sb + part * kPart, re = min(se, rb + kPart);
  for (uint32_t i = tid; i < kAbsSlots; i += kAr) lset[i] = xf::kEmptyKey;
  if (tid == 0) s_cnt = 0;
  __syncthreads();
  uint32_t mine = 0;
  for (uint32_t i = rb + tid; i < re; i += kAr) {
    const Rec3 r = a.rec[i];
    const uint64_t key = (uint64_t)r.klo | ((uint64_t)r.khi << 32);
    if (key == xf::kEmptyKey || !xf::owns(T, key)) continue;  // (the spare position / an error)
    bool absent = false;
    uint64_t p = xf::home_of(T, key);
    for (uint64_t probes = 0; probes < T.cap; ++probes) {
      const uint64_t c = T.keys[p];
      if (c == key) break;
      if (c == xf::kEmptyKey) {
        absent = true;
        break;
      }
      if (++p >= T.cap) p -= T.cap;
    }
    if (!absent) continue;
    uint32_t h = (uint32_t)xf::mix64(key) & (kAbsSlots - 1);
    bool counted = false;
    for (int t = 0; t < 32 && !counted; ++t) {
      const unsigned long long old = atomicCAS(&lset[h], (unsigned long long)xf::kEmptyKey,
                                               (unsigned long long)key);
      if (old == xf::kEmptyKey) {
        ++mine;
        counted = true;
      } else if (old == key) {
        counted = true;
      }
      h = (h + 1) & (kAbsSlots - 1);
    }
    if (!counted) ++mine;  // (the set is full around here: counted without being remembered)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
  if ((tid & 63u) == 0 && mine) atomicAdd(&s_cnt, mine);
  __syncthreads();
  if (tid == 0 && s_cnt) atomicAdd(out, (unsigned long long)s_cnt);
}

// blk_cell[b] = the largest cell with cellptr[cell] <= kBlk * b, by `spare` workgroups
template <typename A>
__device__ __forceinline__ void ar_blk_cell(const A &a, uint32_t wg, uint32_t spare) {
  const uint32_t ncell = a.nwin * a.nchunk, nblk = (a.n + kBlk - 1) / kBlk;
  for (uint32_t b = wg * blockDim.x + threadIdx.x; b <= nblk; b += spare * blockDim.x) {
    uint32_t lo = 0, hi = ncell;  // cellptr[0] = 0
    if (b == nblk) lo = ncell - 1;
    else
      while (hi - lo > 1) {
        const uint32_t m = lo + (hi - lo) / 2;
        if (a.cellptr[m] <= b * kBlk) lo = m;
        else
          hi = m;
      }
    a.blk_cell[b] = lo;
  }
}

// every record becomes the entry of its cell; the workgroups beyond the records compute blk_cell
__global__ void __launch_bounds__(kRes)
k_ar_place(ArArgs a) {
  constexpr int E = 4;
  const uint32_t tid = threadIdx.x;
  const uint32_t nwg = (a.n + kRes * E - 1) / (kRes * E);
  if (blockIdx.x >= nwg) {
    ar_blk_cell(a, blockIdx.x - nwg, gridDim.x - nwg);
    return;
  }
  Rec3 rec[E];
  uint32_t row[E];
  bool ok[E];
#pragma unroll
  for (int q = 0; q < E; ++q) {
    const uint32_t i = blockIdx.x * kRes * E + q * kRes + tid;
    ok[q] = i < a.n;
    rec[q] = ok[q] ? a.rec[i] : Rec3{0u, 0u, 0u};
    row[q] = ok[q] ? a.rec_row[i] : a.chunk0 << kChunkBits;
  }
#pragma unroll
  for (int q = 0; q < E; ++q) {
    const uint32_t v = rec[q].rp >> kRinBits, rin = rec[q].rp & ((1u << kRinBits) - 1u);
    const uint32_t c = (row[q] >> kChunkBits) - a.chunk0;
    const uint32_t slot = ar_cell_slot(a.cellcur, ok[q], v * a.nchunk + c);
    if (ok[q])
      a.entries[slot] =
          ((c & xf::kTagMask) << kTagShift) | (rin << kChunkBits) | (row[q] & (kChunk - 1));
  }
}

// ------------------------------------------------------------- an empty table (round 6)
// The first minibatch of a run meets a table that holds nothing (lr_worker.cc:183-188 starts
// from an empty store): every key is new, and what the arrival index would be asked to do — a
// claim per key, then xf_table_defrag to read the index back in key order — is a sort of the
// minibatch's keys.  The uniform key ranges above ARE the first level of that sort; the second
// runs in LDS, a workgroup per range:
//   k_eb_rank    the range's distinct keys, ascending, and every record's rank among them: an
//                order-preserving key set in LDS (below).  More distinct keys than the set holds,
//                clusters of hundreds (keys that are no hashes), a reserved or a foreign key: the
//                flag goes up and the build takes the arrival index after all.
//   k_eb_scan    ranks of the ranges' first keys (a scan over <= 4000 counts)
//   k_eb_count   the keys into the table's settled tier (rank = state row, fresh: zeroed memory,
//                or k_first_rows), every record's state row, its cell counted — in LDS: a range's
//                rows are consecutive, its records fall into nwin x (<= 6) cells
//   table_settle_first (xf_table.hip)   the tier's two directories
//   k_kb_scan    (cell part), then
//   k_eb_place   the entries: a range's records take their slots from LDS cursors, one atomic on
//                the cell's cursor per (range, cell) instead of one per wavefront and cell
// No claim, no probe, no defrag after the minibatch: the table is settled when the build returns.
// 10^7 nonzeros, 6.3e6 keys: DESIGN.md §3 "First touch".
constexpr int kEb = 512;                   // threads of k_eb_count / k_eb_place
constexpr int kEbR = 1024;                 // threads of k_eb_rank
constexpr uint32_t kEbSlots = 8192;        // positions of a range's key set in LDS ...
constexpr uint32_t kEbPad = 1024;          // ... and behind them, for the last homes' clusters
constexpr uint32_t kEbAll = kEbSlots + kEbPad;
constexpr uint32_t kEbPerT = kEbAll / kEbR;      // positions per thread: 9, groups of 3
constexpr uint32_t kEbCluster = 192;       // keys of one cluster one lane still sorts (half-full sets:
                                           // the longest of 2e7 positions' clusters ~90; a set
                                           // fuller than that is keys that are no hashes)
constexpr uint32_t kEbCells = 2048;        // cells a range's records may fall into (LDS counters)
static_assert(kEbPerT * kEbR == kEbAll && kEbPerT % 3 == 0, "k_eb_rank: a thread's groups");
struct EbArgs {
  xf::TableDev T;
  const Rec3 *rec;         // [n] records grouped by key range
  const uint32_t *sstart;  // [nR + 1]
  const uint64_t *bnd;     // [nR]
  uint32_t nR, n, max_items;
  uint32_t *rec_row;       // [n] k_eb_rank: rank of the record's key in its range; k_eb_count: row
  uint64_t *ukeys;         // [n] the ranges' distinct keys, range S's from sstart[S] on
  uint32_t *dbase;         // [nR + 1] distinct keys per range, then (k_eb_scan) their scan
  unsigned int *out;       // [0] flag: not this way; [1] distinct keys in all
  uint64_t *bkeys;         // the settled tier to be
  uint32_t nwin, nchunk;
  uint32_t *hist, *cellcur, *entries;
  const uint32_t *items, *nitems;  // the partition's work items (a range | its part << 16)
  uint64_t *pkeys;         // [n] a heavy range's parts' distinct keys, part q's from its first record on
  uint32_t *pmap;          // [n] ... their ranks among the range's keys (k_eb_rank<true>)
  uint32_t *pd;            // [max_items] ... how many a part has
};
// two workgroups per CU: 2 x (72 KB of keys + 6 KB of counts) of the 160 KB
constexpr size_t kEbLds = (size_t)kEbAll * 8 + (size_t)(kEbAll / 3) * 2;

// The range's DISTINCT keys as an order-preserving set in LDS (the arrival index's own scheme: a
// key sits at or behind its home = its share of the range's width, inside the home's cluster; no
// wrap-around: kEbPad positions behind the last home), so the records of a heavy key cost a read
// each and a range may hold any number of them.  The clusters (one or two keys at this load,
// dozens now and then) are then sorted where they lie — every key still sits at or behind its home (the positions from the cluster's start
// to a key's home hold keys with smaller homes: smaller keys) — and a key's rank is the number of
// occupied positions before it: a count per group of three positions.  Every record finds its
// key again and takes the rank.
// MERGE = false: a work item of the partition — a range, or kPart records of a heavy one (a
// power-law head key's 10^6 records streamed by ONE workgroup were 1.8 ms): a part ranks its
// records among ITS distinct keys, which go to pkeys.  MERGE = true (a second launch; the first
// part's workgroup of every heavy range): the set again, over the parts' distinct keys — the
// range's keys in order, and for every part's key its rank among them (pmap), through which
// k_eb_count reads the parts' records' rows.
template <bool MERGE>
__global__ void __launch_bounds__(kEbR)
k_eb_rank(EbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char eb_lds[];
  unsigned long long *tab = (unsigned long long *)eb_lds;  // [kEbAll]
  uint16_t *gp = (uint16_t *)(tab + kEbAll);               // [kEbAll / 3] keys before the group
  __shared__ uint32_t wsum[kEbR / 64], s_nlong;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (blockIdx.x >= *a.nitems) return;
  const uint32_t item = a.items[blockIdx.x];
  const uint32_t S = item & 0xFFFFu, part = item >> 16;
  const uint32_t s0 = a.sstart[S], s1 = a.sstart[S + 1];
  const uint32_t parts = (s1 - s0 + kPart - 1) / kPart;
  const bool heavy = parts > 1;  // workgroup-uniform
  if (MERGE && (!heavy || part != 0 || a.out[0] != 0)) return;  // (out[0]: the build goes the other way)
  const uint32_t sb = s0 + part * kPart, m = min(s1, sb + kPart) - sb;  // (this item's records)
  // home of a key: (key - first key of the range) >> sh, the range's width cut into at most
  // kEbSlots equal pieces (the last range ends where the shard's key span does; the keys of the
  // last shard beyond it — the division's remainder — share the last home)
  const uint64_t k0 = a.bnd[S];
  const uint64_t width1 = (S + 1 < a.nR ? a.bnd[S + 1] : a.T.lo + a.T.span) - k0 - 1ull;  // width - 1
  const int bits = 64 - __clzll((long long)(width1 | 1ull));
  const int sh = bits > 13 ? bits - 13 : 0;
  static_assert(kEbSlots == 1u << 13, "the shift above");
  for (uint32_t i = tid; i < kEbAll; i += kEbR) tab[i] = xf::kEmptyKey;
  __syncthreads();
  bool bad = false;
  auto insert = [&](uint64_t key) {
    if (key == xf::kEmptyKey || !xf::owns(a.T, key) || key < k0) {
      bad = true;  // the reserved value, a foreign key: the arrival index knows what to do
      return;
    }
    uint32_t h = (uint32_t)min((key - k0) >> sh, (uint64_t)(kEbSlots - 1));
    for (;;) {
      unsigned long long c = tab[h];
      if (c == xf::kEmptyKey)
        c = atomicCAS(&tab[h], (unsigned long long)xf::kEmptyKey, (unsigned long long)key);
      if (c == xf::kEmptyKey || c == key) break;
      if (++h >= kEbAll) {  // more keys than the set holds
        bad = true;
        break;
      }
    }
  };
  auto rank_of = [&](uint64_t key) -> uint32_t {
    uint32_t h = (uint32_t)min((key - k0) >> sh, (uint64_t)(kEbSlots - 1));
    while (tab[h] != key) ++h;  // (the key is there, at or behind its home)
    const uint32_t g = h / 3, g0 = g * 3;
    uint32_t rank = gp[g];
    if (h > g0) rank += tab[g0] != xf::kEmptyKey ? 1u : 0u;
    if (h > g0 + 1) rank += tab[g0 + 1] != xf::kEmptyKey ? 1u : 0u;
    return rank;
  };
  // records in flight per thread; an item of up to kEbR * E records (the usual one: ranges are
  // cut for 4096) keeps its keys in registers for the second pass
  constexpr int E = 5;
  const bool once = m <= (uint32_t)kEbR * E;  // workgroup-uniform
  uint64_t key[E];
  if (MERGE) {
    for (uint32_t q = 0; q < parts; ++q) {
      const uint32_t pb = s0 + q * kPart, pn = a.pd[blockIdx.x + q];
      for (uint32_t j = tid; j < pn; j += kEbR) insert(a.pkeys[pb + j]);
    }
  } else {
    for (uint32_t i0 = 0; i0 < m; i0 += kEbR * E) {
#pragma unroll
      for (int q = 0; q < E; ++q) {
        const uint32_t i = i0 + q * kEbR + tid;
        const Rec3 r = i < m ? a.rec[sb + i] : Rec3{0xFFFFFFFFu, 0xFFFFFFFFu, 0u};
        key[q] = (uint64_t)r.klo | ((uint64_t)r.khi << 32);
      }
#pragma unroll
      for (int q = 0; q < E; ++q)
        if (i0 + q * kEbR + tid < m) insert(key[q]);
    }
  }
  __syncthreads();
  // The clusters in key order.  A thread looks at its positions for clusters' first positions:
  // a cluster of up to four keys it sorts in registers (five reads in flight, a few
  // compare-and-swaps, the writes); a longer one goes on a list (in gp[], not yet in use) that
  // the wavefronts take in turns: a lane per key, the key's place = the smaller keys of the
  // cluster, counted from registers.  (One thread per cluster with an insertion sort in LDS was
  // the kernel's time: 20 of a workgroup's 31 us were its longest cluster — tools/r6/call22.sh.)
  if (tid == 0) s_nlong = 0;
  __syncthreads();
  const uint32_t p0 = tid * kEbPerT;
  {
    unsigned long long c[kEbPerT + 1];
    c[0] = p0 ? tab[p0 - 1] : xf::kEmptyKey;
#pragma unroll
    for (uint32_t k = 0; k < kEbPerT; ++k) c[k + 1] = tab[p0 + k];
#pragma unroll
    for (uint32_t k = 0; k < kEbPerT; ++k) {
      if (c[k + 1] == xf::kEmptyKey || c[k] != xf::kEmptyKey) continue;
      const uint32_t p = p0 + k;  // a cluster begins here
      auto at = [&](uint32_t i) { return i < kEbAll ? tab[i] : (unsigned long long)xf::kEmptyKey; };
      unsigned long long x0 = c[k + 1], x1 = at(p + 1), x2 = at(p + 2), x3 = at(p + 3);
      const unsigned long long x4 = at(p + 4);
      if (x1 == xf::kEmptyKey) continue;
      const uint32_t len = x2 == xf::kEmptyKey ? 2u : x3 == xf::kEmptyKey ? 3u : x4 == xf::kEmptyKey ? 4u : 5u;
      if (len == 5) {
        gp[atomicAdd(&s_nlong, 1u)] = (uint16_t)p;
        continue;
      }
      auto cs2 = [](unsigned long long &u, unsigned long long &v) {
        if (u > v) {
          const unsigned long long t = u;
          u = v;
          v = t;
        }
      };
      if (len < 4) x3 = xf::kEmptyKey;  // (the largest value: stays last)
      if (len < 3) x2 = xf::kEmptyKey;
      cs2(x0, x1);
      cs2(x2, x3);
      cs2(x0, x2);
      cs2(x1, x3);
      cs2(x1, x2);
      tab[p] = x0;
      tab[p + 1] = x1;
      if (len > 2) tab[p + 2] = x2;
      if (len > 3) tab[p + 3] = x3;
    }
  }
  __syncthreads();
  for (uint32_t idx = wave; idx < s_nlong; idx += kEbR / 64) {  // wave-uniform
    const uint32_t cs = gp[idx];
    const uint32_t pl = cs + lane;
    const unsigned long long k = pl < kEbAll ? tab[pl] : xf::kEmptyKey;
    const unsigned long long occm = __ballot(k != xf::kEmptyKey);
    if (~occm == 0ull) {  // 64 keys and more in one cluster: one lane, in place
      if (lane == 0) {
        uint32_t e = cs + 64;
        while (e < kEbAll && tab[e] != xf::kEmptyKey && e - cs <= kEbCluster) ++e;
        if (e - cs > kEbCluster) bad = true;  // keys that are no hashes
        else
          for (uint32_t i = cs + 1; i < e; ++i) {
            const unsigned long long x = tab[i];
            uint32_t j = i;
            while (j > cs && tab[j - 1] > x) {
              tab[j] = tab[j - 1];
              --j;
            }
            tab[j] = x;
          }
      }
      continue;
    }
    const uint32_t len = (uint32_t)__ffsll((long long)~occm) - 1;  // the first empty position
    uint32_t rank = 0;
    for (uint32_t j = 0; j < len; ++j) {
      const unsigned long long kj =
          (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, (int)j) |
          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), (int)j) << 32);
      rank += kj < k ? 1u : 0u;
    }
    if (lane < len) tab[cs + rank] = k;
  }
  if (__syncthreads_or(bad ? 1 : 0)) {  // not this way: the caller takes the arrival index
    if (tid == 0) atomicOr(&a.out[0], 1u);
    return;
  }
  // ranks: occupied positions before every group of three; the distinct keys go out in order
  uint64_t *__restrict__ dst = MERGE || !heavy ? a.ukeys + s0 : a.pkeys + sb;
  unsigned long long mine[kEbPerT];
  uint32_t occ = 0;
#pragma unroll
  for (uint32_t k = 0; k < kEbPerT; ++k) {
    mine[k] = tab[p0 + k];
    occ += mine[k] != xf::kEmptyKey ? 1u : 0u;
  }
  uint32_t inc = occ;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(inc, o);
    if ((int)lane >= o) inc += y;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = inc - occ, d = 0;
#pragma unroll
  for (uint32_t w = 0; w < (uint32_t)kEbR / 64; ++w) {
    const uint32_t x = wsum[w];
    if (w < wave) before += x;
    d += x;
  }
#pragma unroll
  for (uint32_t k = 0; k < kEbPerT; ++k) {
    if (k % 3 == 0) gp[(p0 + k) / 3] = (uint16_t)before;
    if (mine[k] != xf::kEmptyKey) {
      dst[before] = mine[k];
      ++before;
    }
  }
  __syncthreads();
  if (MERGE) {
    for (uint32_t q = 0; q < parts; ++q) {
      const uint32_t pb = s0 + q * kPart, pn = a.pd[blockIdx.x + q];
      for (uint32_t j = tid; j < pn; j += kEbR) a.pmap[pb + j] = rank_of(a.pkeys[pb + j]);
    }
    if (tid == 0) a.dbase[S] = d;
    return;
  }
  for (uint32_t i0 = 0; i0 < m; i0 += kEbR * E) {
    if (!once) {
#pragma unroll
      for (int q = 0; q < E; ++q) {
        const uint32_t i = i0 + q * kEbR + tid;
        const Rec3 r = i < m ? a.rec[sb + i] : Rec3{0u, 0u, 0u};
        key[q] = (uint64_t)r.klo | ((uint64_t)r.khi << 32);
      }
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t i = i0 + q * kEbR + tid;
      if (i < m) a.rec_row[sb + i] = rank_of(key[q]);
    }
  }
  if (tid == 0) {
    if (heavy) a.pd[blockIdx.x] = d;
    else
      a.dbase[S] = d;
  }
}

__global__ void __launch_bounds__(1024)
k_eb_scan(EbArgs a) {
  __shared__ uint32_t wsum[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  constexpr uint32_t kPerT = 4;  // 4096 >= kArMaxRanges
  uint32_t v[kPerT], sum = 0;
#pragma unroll
  for (uint32_t k = 0; k < kPerT; ++k) {
    const uint32_t r = tid * kPerT + k;
    v[k] = r < a.nR ? a.dbase[r] : 0u;
    sum += v[k];
  }
  uint32_t inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t y = __shfl_up(inc, o);
    if ((int)lane >= o) inc += y;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t run = inc - sum;
  for (uint32_t w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
  for (uint32_t k = 0; k < kPerT; ++k) {
    const uint32_t r = tid * kPerT + k;
    if (r < a.nR) a.dbase[r] = run;
    run += v[k];
    if (r + 1 == a.nR) {
      a.dbase[a.nR] = run;
      a.out[1] = run;
    }
  }
}

// the cells a range's records fall into: (row window) x (the chunks its consecutive rows span)
struct EbLocal {
  uint32_t base, d, c_lo, nloc;
};
__device__ __forceinline__ EbLocal eb_local(const EbArgs &a, uint32_t S) {
  EbLocal L;
  L.base = a.dbase[S];
  L.d = a.dbase[S + 1] - L.base;
  L.c_lo = L.base >> kChunkBits;
  L.nloc = L.d ? ((L.base + L.d - 1) >> kChunkBits) - L.c_lo + 1 : 1u;
  return L;
}

// PLACE = false: k_eb_count; true: k_eb_place (the workgroups beyond the ranges: blk_cell)
template <bool PLACE>
__global__ void __launch_bounds__(kEb)
k_eb_cells(EbArgs a, ArArgs r) {
  __shared__ uint32_t lcnt[kEbCells], lcur[kEbCells];
  const uint32_t tid = threadIdx.x;
  if (PLACE && blockIdx.x >= a.max_items) {
    ar_blk_cell(r, blockIdx.x - a.max_items, gridDim.x - a.max_items);
    return;
  }
  if (blockIdx.x >= *r.nitems) return;
  // a work item of the partition: a range, or kPart records of a heavy one (a power-law head's
  // 10^6 records are thirty workgroups' work, not one's)
  const uint32_t item = r.items[blockIdx.x];
  const uint32_t S = item & 0xFFFFu, part = item >> 16;
  const uint32_t s0 = a.sstart[S], s1 = a.sstart[S + 1];
  const uint32_t sb = s0 + part * kPart, m = min(s1, sb + kPart) - sb;
  const EbLocal L = eb_local(a, S);
  const uint32_t nl = a.nwin * L.nloc;  // <= kEbCells (the host saw to it)
  for (uint32_t i = tid; i < nl; i += kEb) lcnt[i] = 0;
  if (!PLACE && part == 0)
    for (uint32_t j = tid; j < L.d; j += kEb) a.bkeys[L.base + j] = a.ukeys[s0 + j];
  __syncthreads();
  for (uint32_t i0 = 0; i0 < m; i0 += kEb * 4) {
    uint32_t rp[4], row[4], lc[4], at[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t i = i0 + q * kEb + tid;
      ok[q] = i < m;
      rp[q] = ok[q] ? a.rec[sb + i].rp : 0u;
      row[q] = L.base;
      if (ok[q]) {
        const uint32_t rr = a.rec_row[sb + i];  // PLACE: the row; else the key's rank in its part / range
        row[q] = PLACE ? rr : (s1 - s0 > kPart ? a.pmap[sb + rr] : rr) + L.base;
      }
      lc[q] = (rp[q] >> kRinBits) * L.nloc + ((row[q] >> kChunkBits) - L.c_lo);
      if (ok[q]) at[q] = atomicAdd(&lcnt[lc[q]], 1u);
      if (!PLACE && ok[q]) a.rec_row[sb + i] = row[q];
    }
    if (PLACE) {
      // this round's records take their slots: the cells' cursors move once per round and cell
      __syncthreads();
      for (uint32_t c = tid; c < nl; c += kEb) {
        const uint32_t k = lcnt[c];
        if (k) {
          const uint32_t v = c / L.nloc, ch = L.c_lo + (c - v * L.nloc);
          lcur[c] = atomicAdd(&a.cellcur[v * a.nchunk + ch], k);
          lcnt[c] = 0;
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ok[q]) {
          const uint32_t ch = row[q] >> kChunkBits;
          a.entries[lcur[lc[q]] + at[q]] = ((ch & xf::kTagMask) << kTagShift) |
                                           ((rp[q] & ((1u << kRinBits) - 1u)) << kChunkBits) |
                                           (row[q] & (kChunk - 1));
        }
      __syncthreads();
    }
  }
  if (!PLACE) {
    __syncthreads();
    for (uint32_t c = tid; c < nl; c += kEb) {
      const uint32_t k = lcnt[c];
      if (k) {
        const uint32_t v = c / L.nloc, ch = L.c_lo + (c - v * L.nloc);
        atomicAdd(&a.hist[v * a.nchunk + ch], k);
      }
    }
  }
}

// ----------------------------------------------------------------------------------- FM
// The key build of FMWorker::update (fm_worker.cc:205-225) against the v table's settled tier.
// Same partition as above (histogram, scan, scatter by super-chunk) with 16-byte records that
// carry the nonzero's row AND its position in the CSR; then:
//   k_kb_resolve_fm  every record's state row (= the key's rank in the tier), in record order,
//                    and — scattered by position — the forward's per-nonzero record index
//   k_fm_count       touched keys per super-chunk (an LDS histogram of its records' rows)
//   k_fm_regroup     per super-chunk: its touched keys in row order (key list, rows, segment
//                    offsets), its records' rows grouped by key (an LDS counting sort): the
//                    gradient's key-grouped occurrence lists
// No sort of (key, position) pairs, no unique-key pass over sorted keys, no probe of the table
// per nonzero.  The order of a key's occurrences is the records' (not reproducible run to run:
// LDS cursors): the sums the gradient forms from them are exact in fp64.
__global__ void __launch_bounds__(kRes)
k_kb_resolve_fm(KbArgs a) {
  __shared__ uint64_t lk[kSCKeys];
  __shared__ uint16_t dir[kDirStride];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  if (blockIdx.x >= *a.nitems) return;
  const uint32_t item = a.items[blockIdx.x];
  const uint32_t S = item & 0xFFFFu, part = item >> 16;
  const uint32_t sb = a.sstart[S], se = a.sstart[S + 1];
  const uint32_t rb = sb + part * kPart, re = min(se, rb + kPart);
  const uint32_t k0 = S * kSCKeys, nk = min(kSCKeys, a.nbase - k0);
  const uint32_t nb = max(nk / 2, 1u);
  {
    const ulonglong2 *__restrict__ src = (const ulonglong2 *)(a.bkeys + k0);  // 16-byte aligned
    for (uint32_t i = tid; i < (nk + 1) / 2; i += kRes) {  // (bkeys is padded past nbase)
      const ulonglong2 t = src[i];
      lk[2 * i] = t.x;
      lk[2 * i + 1] = t.y;
    }
    const uint32_t *__restrict__ gd = (const uint32_t *)(a.sdirs + (size_t)S * kDirStride);
    uint32_t *ld = (uint32_t *)dir;
    for (uint32_t i = tid; i < (nb + 2 + 1) / 2; i += kRes) ld[i] = gd[i];
  }
  __syncthreads();
  const uint64_t sm = a.smult[S], kfirst = lk[0];
  auto lkf = [&](uint32_t i) -> uint64_t { return lk[i]; };
  constexpr int E = 4;
  uint32_t nmiss = 0;
  // (all of an item's records in registers up front, as k_kb_resolve has them, was tried: 208 ->
  // 221 us — the scattered record-index stores are this kernel, not the waits for its loads)
  for (uint32_t i0 = rb; i0 < re; i0 += kRes * E) {
    uint64_t key[E], x0[E], x1[E];
    uint32_t ds[E], de[E], pos[E], rpk[E];
    bool ok[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t i = i0 + q * kRes + tid;
      ok[q] = i < re;
      const Rec4 r = ok[q] ? a.rec4[i] : Rec4{0u, 0u, 0u, 0u};
      pos[q] = r.pos;
      rpk[q] = r.rp;
      key[q] = (uint64_t)r.klo | ((uint64_t)r.khi << 32);
      const uint32_t b = kb_sbucket(key[q], kfirst, sm, nb);
      ds[q] = dir[b];
      de[q] = dir[b + 1];
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
      x0[q] = lk[min(ds[q], nk - 1)];
      x1[q] = lk[min(ds[q] + 1, nk - 1)];
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t ub = key[q] < kfirst ? 0u : kb_count(lkf, ds[q], de[q], x0[q], x1[q], key[q]);
      const uint32_t p = ub ? ub - 1 : 0u;
      const bool found = ub > 0 && lk[p] == key[q];
      if (ok[q]) {
        a.fm_vrow[i0 + q * kRes + tid] = found ? k0 + p : kHole;
        a.fm_rp[i0 + q * kRes + tid] = rpk[q];  // (4 of the record's 16 bytes for the regroup)
        // the forward's record index of the nonzero (a scattered 4-byte store: here, where the
        // look-ups keep the workgroup busy, rather than in the regroup pass)
        if (found) a.fm_ridx[pos[q]] = k0 + p;
        nmiss += found ? 0u : 1u;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nmiss += __shfl_xor(nmiss, o);
  if (lane == 0 && nmiss) atomicAdd(&a.sum->miss, (unsigned long long)nmiss);
}

struct FmRegroup {
  const uint32_t *vrow;    // [NNZ] state row of every record
  const uint32_t *rp;      // [NNZ] row of every record (window << 15 | row in window)
  const uint32_t *sstart;  // [nS + 1] first record of every super-chunk
  const uint64_t *bkeys;   // the tier's keys
  uint32_t nS, W;
  uint64_t nbase;          // keys of the tier
  uint32_t *ucount;        // [nS] touched keys per super-chunk; [nS + 1] after the scan: ubase
  uint64_t *ukeys;         // [U] the minibatch's keys, ascending (= in row order)
  uint32_t *urow;          // [U] their state rows
  uint32_t *segptr;        // [U + 1]
  uint32_t *coo;           // [NNZ] rows of the occurrences, grouped by key
  // work items (the partition's: a super-chunk, or kPart records of a heavy one — a power-law
  // head key's 10^6 records in ONE workgroup made k_fm_count 1.5 ms and k_fm_regroup 3.1 ms on
  // the Zipf(1.1) stream: 0.013 and 0.13 on the uniform one)
  const uint32_t *items, *nitems;
  uint32_t *pcnt;          // [items][kSCKeys] a heavy super-chunk's records per key and part
  uint32_t *gbits;         // [nS][kSCKeys / 32] ... its touched keys (zeroed)
  uint32_t *done;          // [nS] ... its parts that have counted (zeroed)
};

struct FmItem {
  uint32_t S, part, parts, sb, se, rb, re;
};
__device__ __forceinline__ FmItem fm_item(const FmRegroup &g, uint32_t item) {
  FmItem it;
  it.S = item & 0xFFFFu;
  it.part = item >> 16;
  it.sb = g.sstart[it.S];
  it.se = g.sstart[it.S + 1];
  it.parts = (it.se - it.sb + kPart - 1) / kPart;
  it.rb = it.sb + it.part * kPart;
  it.re = min(it.se, it.rb + kPart);
  return it;
}

__global__ void __launch_bounds__(kKb)
k_fm_count(FmRegroup g) {
  __shared__ uint32_t cnt[kSCKeys];
  __shared__ uint32_t wsum[kKb / 64];
  __shared__ uint32_t s_last;
  const uint32_t tid = threadIdx.x;
  if (blockIdx.x >= *g.nitems) return;
  const FmItem it = fm_item(g, g.items[blockIdx.x]);
  const uint32_t k0 = it.S * kSCKeys;
  for (uint32_t k = tid; k < kSCKeys; k += kKb) cnt[k] = 0;
  __syncthreads();
  for (uint32_t i0 = it.rb; i0 < it.re; i0 += kKb * 8) {
    uint32_t r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint32_t i = i0 + q * kKb + tid;
      r[q] = i < it.re ? g.vrow[i] : kHole;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) (void)lds_add_shared<false>(cnt, r[q] - k0, r[q] != kHole);
  }
  __syncthreads();
  if (it.parts > 1) {  // workgroup-uniform: the counts out, the touched keys into the super-chunk's bits
    uint32_t *__restrict__ pc = g.pcnt + (size_t)blockIdx.x * kSCKeys;
    for (uint32_t k = tid; k < kSCKeys; k += kKb) {
      const uint32_t c = cnt[k];
      pc[k] = c;
      const unsigned long long mb = __ballot(c != 0);  // (k = 64 * wave + lane: two words)
      if ((tid & 63u) == 0 && mb) {
        const uint32_t w = it.S * (kSCKeys / 32) + (k >> 5);
        if ((uint32_t)mb) atomicOr(&g.gbits[w], (uint32_t)mb);
        if ((uint32_t)(mb >> 32)) atomicOr(&g.gbits[w + 1], (uint32_t)(mb >> 32));
      }
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&g.done[it.S], 1u) == it.parts - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    uint32_t n = 0;  // the last part to count: every part's bits are there
    for (uint32_t w = tid; w < kSCKeys / 32; w += kKb)
      n += (uint32_t)__popc(__hip_atomic_load(&g.gbits[it.S * (kSCKeys / 32) + w], __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT));
    uint32_t total;
    (void)block_excl_scan(n, wsum, &total);
    if (tid == 0) g.ucount[it.S] = total;
    return;
  }
  uint32_t n = 0;
  for (uint32_t k = tid; k < kSCKeys; k += kKb) n += cnt[k] ? 1u : 0u;
  uint32_t total;
  (void)block_excl_scan(n, wsum, &total);
  if (tid == 0) g.ucount[it.S] = total;
}

__global__ void __launch_bounds__(kKb)
k_fm_scan(uint32_t *__restrict__ ucount, uint32_t nS) {
  __shared__ uint32_t sbuf[kScanPiece + kScanPiece / 16];
  __shared__ uint32_t wsum[kKb / 64];
  const volatile uint32_t *vin = ucount;
  const uint32_t total =
      staged_excl_scan([&](uint32_t c) { return (uint32_t)vin[c]; }, nS, ucount, nullptr, sbuf, wsum);
  if (threadIdx.x == 0) ucount[nS] = total;
}

__global__ void __launch_bounds__(kKb)
k_fm_regroup(FmRegroup g) {
  __shared__ uint32_t cnt[kSCKeys], off[kSCKeys];
  __shared__ uint32_t wsum[kKb / 64];
  const uint32_t tid = threadIdx.x;
  if (blockIdx.x >= *g.nitems) return;
  const FmItem it = fm_item(g, g.items[blockIdx.x]);
  const uint32_t S = it.S, sb = it.sb, k0 = S * kSCKeys;
  const bool heavy = it.parts > 1;  // workgroup-uniform
  // a thread owns kSCKeys / kKb consecutive rows; their keys are asked for now — a load per
  // touched row inside the loop below was a dependent trip to memory each
  constexpr uint32_t kPer = kSCKeys / kKb;
  uint64_t mykey[kPer];
  uint32_t before[kPer];  // a heavy super-chunk: the key's records in the parts before this one
#pragma unroll
  for (uint32_t q = 0; q < kPer; ++q) {
    mykey[q] = g.bkeys[min((uint64_t)k0 + tid * kPer + q, g.nbase - 1)];
    before[q] = 0;
  }
  if (heavy) {
    // every part's counts per key (k_fm_count): the key's total, and what lies before this part
    const uint32_t *__restrict__ pc = g.pcnt + (size_t)(blockIdx.x - it.part) * kSCKeys;
#pragma unroll
    for (uint32_t q = 0; q < kPer; ++q) {
      const uint32_t k = tid * kPer + q;
      uint32_t tot = 0;
      for (uint32_t p = 0; p < it.parts; ++p) {
        const uint32_t c = pc[(size_t)p * kSCKeys + k];
        tot += c;
        if (p < it.part) before[q] += c;
      }
      cnt[k] = tot;
    }
  } else {
    for (uint32_t k = tid; k < kSCKeys; k += kKb) cnt[k] = 0;
    __syncthreads();
    for (uint32_t i0 = it.rb; i0 < it.re; i0 += kKb * 8) {
      uint32_t r[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t i = i0 + q * kKb + tid;
        r[q] = i < it.re ? g.vrow[i] : kHole;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) (void)lds_add_shared<false>(cnt, r[q] - k0, r[q] != kHole);
    }
  }
  __syncthreads();
  // where the rows' occurrences begin, and their numbers among the touched keys
  uint32_t occ = 0, used = 0;
#pragma unroll
  for (uint32_t q = 0; q < kPer; ++q) {
    const uint32_t c = cnt[tid * kPer + q];
    occ += c;
    used += c ? 1u : 0u;
  }
  uint32_t t0, t1;
  uint32_t run = block_excl_scan(occ, wsum, &t0);
  uint32_t urun = g.ucount[S] + block_excl_scan(used, wsum, &t1);
#pragma unroll
  for (uint32_t q = 0; q < kPer; ++q) {
    const uint32_t k = tid * kPer + q, c = cnt[k];
    off[k] = run;
    if (c && it.part == 0) {
      g.urow[urun] = k0 + k;
      g.ukeys[urun] = mykey[q];
      g.segptr[urun] = sb + run;
    }
    urun += c ? 1u : 0u;
    run += c;
  }
  if (S == g.nS - 1 && it.part == 0 && tid == kKb - 1) g.segptr[urun] = g.sstart[g.nS];  // segptr[U] = NNZ
  __syncthreads();
  // now the rows' cursors (a heavy super-chunk: behind the parts before this one)
#pragma unroll
  for (uint32_t q = 0; q < kPer; ++q) cnt[tid * kPer + q] = before[q];
  __syncthreads();
  constexpr int E = 8;  // a thread's loads of a round first, then its cursors and stores
  for (uint32_t i0 = it.rb; i0 < it.re; i0 += kKb * E) {
    uint32_t r[E], rp[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const uint32_t i = i0 + q * kKb + tid;
      r[q] = i < it.re ? g.vrow[i] : kHole;
      rp[q] = i < it.re ? g.rp[i] : 0u;
    }
#pragma unroll
    for (int q = 0; q < E; ++q) {
      const bool on = r[q] != kHole;
      const uint32_t l = on ? r[q] - k0 : 0u;
      const uint32_t at = lds_add_shared<true>(cnt, l, on);
      if (on)
        g.coo[sb + off[l] + at] = (rp[q] >> kRinBits) * g.W + (rp[q] & ((1u << kRinBits) - 1u));
    }
  }
}


// ------------------------------------------------- (key, position) in key order (round 6)
// The sort at the top of the reference's key build (lr_worker.cc:146-166: all_keys[(fid, sid)],
// std::sort by fid) where no table stands behind the keys — the worker side of the weight /
// gradient exchange, the FM fallback, an owner's merged walking order: until round 6 a 64-bit
// rocPRIM radix sort of (key, position).  The uniform key ranges above are its first level here
// as well (k_kb_hist_groups / _scan / _scatter with the position as the record's payload); the
// second runs in LDS, a workgroup per range:
//   k_sp_sort   the range's records (a few thousand) counted into 8192 equal pieces of the
//               range's width, staged piece by piece (a counting sort in LDS); a piece holds one
//               record, sometimes two or three — ranked by comparison, a lane per record — and
//               now and then the thousand records of one hot key: a piece of up to 128 records
//               is sorted where it lies by a wavefront (a bitonic network over (key, position)),
//               a longer one by the workgroup (sp_long_piece: one key's positions, a second
//               counting sort); every record leaves for its place in the range's window of the
//               output.
// The partition does not keep the positions' order (a tile's records of a range are ranked by an
// LDS atomic), the comparisons are on (key, position): the result is THE sorted order, as a
// stable sort's.  A range of more than kSpCap records (a power-law head — real click logs have
// them —, keys that are no hashes) is a merge sort: it goes on a list the host reads with the
// build's one wait; its parts of kSpCap records are sorted in LDS the same way (k_sp_parts) and
// merged pair by pair, a pass per doubling (k_sp_merge: an output tile's two inputs found by a
// search along its diagonal, merged in LDS, eight outputs per thread).  A stream that has shown
// itself skewed gets its hot keys ranges of their own first ("hot keys get ranges of their own"
// below): no merge for them.
constexpr int kSp = 1024;
constexpr uint32_t kSpCap = 8192;      // records of a range
constexpr uint32_t kSpPieces = 8192;   // equal pieces of a range's width
constexpr uint32_t kSpBrute = 32;      // records of a piece ranked by comparison
constexpr uint32_t kSpWave = 128;      // ... sorted by one wavefront (the bitonic network)
constexpr uint32_t kSpList = 256;      // longer pieces per range and kind (8192 / 33 < 256)
constexpr size_t kSpLds = (size_t)kSpCap * 14 + ((size_t)kSpPieces + 1 + 2 * kSpList) * 4;
constexpr uint32_t kSpTileSpan = 4096; // k_sp_tiles: a workgroup's tile groups begin inside so many records
static_assert(kSpLds <= kDynMax && kSp == kKb, "k_sp_sort: LDS, block_excl_scan's workgroup");
struct SpArgs {
  const Rec3 *rec;         // [n] records grouped by key range, rp = the nonzero's position
  const uint32_t *sstart;  // [nR + 1]
  const uint64_t *bnd;     // [nR]
  uint32_t nR;
  uint64_t last;           // the largest key of the span
  uint64_t *sk;            // [n] out: keys, ascending
  uint32_t *spos;          // [n] out: their positions (ascending inside a key)
  uint32_t tbits;          // payload = position: log2 of the partition's tile (a range's records
                           // lie tile after tile); 0: no use is made of that
  uint32_t rows;           // != 0: the payload is the nonzero's row and its order inside a key
                           // is nobody's concern (a range that is ONE key's is copied)
  unsigned int *heavy;     // [0] ranges of more than kSpCap records, [1] their parts of kSpCap
                           // records in all, [2] the longest one's records
  uint2 *hv;               // [nR] those ranges: (range, its first part's number among all parts)
  // heavy[4], [5] and hv2: the one-key ranges among them whose payload is the position — their
  // records lie in tile order, a tile's among themselves in none: sorted tile group by tile
  // group (k_sp_tiles), parts of kSpTileSpan records
  uint2 *hv2;
  // the merge passes (k_sp_merge): runs of kSpCap << pass records, from (mk, mp) to (ok, op)
  const uint64_t *mk;
  const uint32_t *mp;
  uint64_t *ok;
  uint32_t *op;
  uint32_t pass;
};

// K[0..c), P[0..c) in (key, position) order: the bitonic network whose every comparator puts the
// smaller element at the lower index (a stage's first step mirrors the upper half), so c need
// not be a power of two — the elements beyond c are +infinity and never move.  nt threads, t
// this one's number; sync() between the steps.
template <typename Sync>
__device__ __forceinline__ void sp_bitonic(unsigned long long *K, uint32_t *P, uint32_t c,
                                           uint32_t t, uint32_t nt, Sync sync) {
  uint32_t n2 = 2;
  while (n2 < c) n2 <<= 1;
  auto cx = [&](uint32_t l, uint32_t r) {
    if (r >= c) return;
    const unsigned long long kl = K[l], kr = K[r];
    const uint32_t pl = P[l], pr = P[r];
    if (kl > kr || (kl == kr && pl > pr)) {
      K[l] = kr;
      K[r] = kl;
      P[l] = pr;
      P[r] = pl;
    }
  };
  for (uint32_t k = 2; k <= n2; k <<= 1) {
    const uint32_t h = k >> 1;
    for (uint32_t i = t; i < n2 / 2; i += nt) {
      const uint32_t b0 = (i / h) * k, off = i & (h - 1);
      cx(b0 + off, b0 + k - 1 - off);
    }
    sync();
    for (uint32_t j = h >> 1; j >= 1; j >>= 1) {
      for (uint32_t i = t; i < n2 / 2; i += nt) {
        const uint32_t l = (i / j) * 2 * j + (i & (j - 1));
        cx(l, l + j);
      }
      sync();
    }
  }
}

// A piece of more than kSpWave records, the workgroup's: nearly always ONE key's (a hot key's
// records, and now and then a cold neighbour's) — then the order is the positions' order, and
// positions are spread evenly enough for a second counting sort: the other keys' few records
// aside, the hot key's positions counted into as many equal pieces of their span as there are
// records, staged (counters and stage lie where the piece's keys were: they are all the same),
// ranked by comparison inside a piece.  A piece of several keys with many records each: the
// bitonic network (91 steps over 8192 records: ~120 us, which made a power-law stream's sort
// 1.5 ms).  c <= kSpCap; all kSp threads.
constexpr uint32_t kSpOdd = 32;  // records of other keys a one-key piece may hold
template <int E = (int)(kSpCap / kSp)>  // (records per thread: c <= E * kSp)
__device__ __forceinline__ void sp_long_piece(unsigned long long *K, uint32_t *P, uint32_t c,
                                              uint32_t *wsum) {
  __shared__ unsigned long long exK[kSpOdd];
  __shared__ uint32_t exP[kSpOdd], s_nex, s_pmin, s_pmax;
  const uint32_t tid = threadIdx.x;
  const unsigned long long kh = K[c / 2];
  if (tid == 0) {
    s_nex = 0;
    s_pmin = 0xFFFFFFFFu;
    s_pmax = 0;
  }
  __syncthreads();
  uint32_t pp[E], pc[E], at[E], lo = 0xFFFFFFFFu, hi = 0;
  bool eq[E];
#pragma unroll
  for (int q = 0; q < E; ++q) {
    const uint32_t i = q * kSp + tid;
    eq[q] = false;
    pp[q] = 0;
    if (i < c) {
      const unsigned long long kk = K[i];
      pp[q] = P[i];
      eq[q] = kk == kh;
      if (eq[q]) {
        lo = min(lo, pp[q]);
        hi = max(hi, pp[q]);
      } else if (s_nex <= kSpOdd) {  // (a piece of several keys: not thousands of atomics on one word)
        const uint32_t e = atomicAdd(&s_nex, 1u);
        if (e < kSpOdd) {
          exK[e] = kk;
          exP[e] = pp[q];
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
    hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
  }
  if ((tid & 63u) == 0 && lo <= hi) {
    atomicMin(&s_pmin, lo);
    atomicMax(&s_pmax, hi);
  }
  __syncthreads();
  const uint32_t nex = s_nex;
  if (nex > kSpOdd) {  // (workgroup-uniform; nothing was moved)
    sp_bitonic(K, P, c, tid, (uint32_t)kSp, [] { lds_barrier(); });
    __syncthreads();
    return;
  }
  const uint32_t ceq = c - nex, pmin = s_pmin;
  uint32_t np = 1;  // pieces of the positions' span: the largest power of two <= ceq
  while (np * 2 <= ceq) np *= 2;
  const int bw = 32 - __clz((int)((s_pmax - pmin) | 1u)), lg = 31 - __clz((int)np);
  const int sh = bw > lg ? bw - lg : 0;
  uint32_t *cnt = (uint32_t *)K, *tmp = cnt + np;  // np + ceq <= 2 c words
  for (uint32_t i = tid; i < np; i += kSp) cnt[i] = 0;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < E; ++q)
    if (eq[q]) {
      pc[q] = (pp[q] - pmin) >> sh;
      at[q] = atomicAdd(&cnt[pc[q]], 1u);
    }
  __syncthreads();
  {
    const uint32_t per = (np + kSp - 1) / kSp;  // <= E
    uint32_t v[E], sum = 0;
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const uint32_t i = tid * per + k;
      v[k] = (uint32_t)k < per && i < np ? cnt[i] : 0u;
      sum += v[k];
    }
    uint32_t total;
    uint32_t run = block_excl_scan(sum, wsum, &total);
#pragma unroll
    for (int k = 0; k < E; ++k) {
      const uint32_t i = tid * per + k;
      if ((uint32_t)k < per && i < np) cnt[i] = run;
      run += v[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < E; ++q)
    if (eq[q]) tmp[cnt[pc[q]] + at[q]] = pp[q];
  uint32_t nlt = 0;
  for (uint32_t e = 0; e < nex; ++e) nlt += exK[e] < kh ? 1u : 0u;
  __syncthreads();
  for (uint32_t j = tid; j < ceq; j += kSp) {
    const uint32_t pos = tmp[j], x = (pos - pmin) >> sh;
    const uint32_t b = cnt[x], e = x + 1 < np ? cnt[x + 1] : ceq;
    uint32_t rank = 0;
    for (uint32_t q = b; q < e; ++q) rank += (tmp[q] < pos || (tmp[q] == pos && q < j)) ? 1u : 0u;
    P[nlt + b + rank] = pos;
  }
  __syncthreads();
  for (uint32_t j = tid; j < ceq; j += kSp) K[nlt + j] = kh;
  if (tid < nex) {
    const unsigned long long kk = exK[tid];
    const uint32_t pos = exP[tid];
    uint32_t r = 0;
    for (uint32_t e = 0; e < nex; ++e)
      r += (exK[e] < kk || (exK[e] == kk && (exP[e] < pos || (exP[e] == pos && e < tid)))) ? 1u : 0u;
    const uint32_t at2 = kk < kh ? r : ceq + r;
    K[at2] = kk;
    P[at2] = pos;
  }
  __syncthreads();
}

// records rec[0..m) of range S (all of them, or a part of a heavy range's) to out_k / out_p[0..m)
template <uint32_t CAP = kSpCap, uint32_t PIECES = kSpPieces>
__device__ __forceinline__ void sp_sort_block(const SpArgs &a, uint32_t S, const Rec3 *__restrict__ rec,
                                              uint32_t m, uint64_t *__restrict__ out_k,
                                              uint32_t *__restrict__ out_p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
  unsigned long long *stK = (unsigned long long *)sp_lds;  // [CAP] the staged records' keys
  uint32_t *stP = (uint32_t *)(stK + CAP);                 // [CAP] ... positions
  uint32_t *st = stP + CAP;                                // [PIECES + 1] counts, then starts
  uint32_t *lw = st + PIECES + 1;                       // pieces a wavefront sorts
  uint32_t *lg = lw + kSpList;                             // pieces the workgroup sorts
  __shared__ uint32_t wsum[kSp / 64], s_nw, s_ng;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint64_t k0 = a.bnd[S];
  const uint64_t width1 = (S + 1 < a.nR ? a.bnd[S + 1] - 1ull : a.last) - k0;  // width - 1
  const int bits = 64 - __clzll((long long)(width1 | 1ull));
  constexpr int lgp = PIECES == 8192 ? 13 : 12;
  static_assert(PIECES == 1u << lgp, "the pieces: 4096 or 8192");
  const int sh = bits > lgp ? bits - lgp : 0;
  // (a key below the range's first — one that lies below the span, in range 0 — goes with the
  // first piece, one beyond the span with the last: the comparisons below are on whole keys)
  auto piece = [&](uint64_t key) -> uint32_t {
    return key < k0 ? 0u : (uint32_t)min((key - k0) >> sh, (uint64_t)(PIECES - 1));
  };
  for (uint32_t i = tid; i <= PIECES; i += kSp) st[i] = 0;
  if (tid == 0) s_nw = s_ng = 0;
  __syncthreads();
  constexpr int E = (int)(CAP / kSp);
  uint64_t key[E];
  uint32_t pos[E], pc[E], at[E];
#pragma unroll
  for (int q = 0; q < E; ++q) {
    const uint32_t i = q * kSp + tid;
    const Rec3 r = i < m ? rec[i] : Rec3{0u, 0u, 0u};
    key[q] = (uint64_t)r.klo | ((uint64_t)r.khi << 32);
    pos[q] = r.rp;
  }
#pragma unroll
  for (int q = 0; q < E; ++q) {
    pc[q] = piece(key[q]);
    at[q] = q * kSp + tid < m ? atomicAdd(&st[pc[q]], 1u) : 0u;
  }
  __syncthreads();
  {  // st = exclusive scan of the counts; the long pieces on their lists
    constexpr uint32_t per = PIECES / kSp;
    uint32_t c[per], sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < per; ++k) {
      c[k] = st[tid * per + k];
      sum += c[k];
    }
    uint32_t total;
    uint32_t run = block_excl_scan(sum, wsum, &total);
#pragma unroll
    for (uint32_t k = 0; k < per; ++k) {
      st[tid * per + k] = run;
      run += c[k];
      if (c[k] > kSpBrute) {
        if (c[k] <= kSpWave) lw[atomicAdd(&s_nw, 1u)] = tid * per + k;
        else
          lg[atomicAdd(&s_ng, 1u)] = tid * per + k;
      }
    }
    if (tid == 0) st[PIECES] = m;
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < E; ++q)
    if (q * kSp + tid < m) {
      const uint32_t p = st[pc[q]] + at[q];
      stK[p] = key[q];
      stP[p] = pos[q];
    }
  __syncthreads();
  for (uint32_t idx = wave; idx < s_nw; idx += kSp / 64) {  // wave-uniform
    const uint32_t b = st[lw[idx]], c = st[lw[idx] + 1] - b;
    sp_bitonic(stK + b, stP + b, c, lane, 64u,
               [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); });
  }
  __syncthreads();
  for (uint32_t idx = 0; idx < s_ng; ++idx) {  // workgroup-uniform
    const uint32_t b = st[lg[idx]], c = st[lg[idx] + 1] - b;
    sp_long_piece<(int)(CAP / kSp)>(stK + b, stP + b, c, wsum);
  }
  // every record to its place — where it is staged, or, a short piece's, at its rank in the piece:
  // the places first (who goes to place i: inv[i], 16 bits in LDS), then out in order — the
  // records' 12 bytes stored straight to their places were 216 MB of write requests for 120 MB
  // per 1e7 keys (profiles/r06/pmc_traffic_sort_key_pos.json; held in registers until all places
  // are known they spill: 64 registers a thread)
  uint16_t *inv = (uint16_t *)(lg + kSpList);  // [CAP]
  for (uint32_t p = tid; p < m; p += kSp) {
    const unsigned long long kk = stK[p];
    const uint32_t pp = stP[p], pcs = piece(kk);
    const uint32_t b = st[pcs], c = st[pcs + 1] - b;
    uint32_t out = p;
    if (c > 1 && c <= kSpBrute) {
      uint32_t rank = 0;
      for (uint32_t q = 0; q < c; ++q) {  // (equal records — a key twice in a row, the payload the
        const unsigned long long kq = stK[b + q];  // row — in the order they are staged in)
        const uint32_t pq = stP[b + q];
        rank += (kq < kk || (kq == kk && (pq < pp || (pq == pp && b + q < p)))) ? 1u : 0u;
      }
      out = b + rank;
    }
    inv[out] = (uint16_t)p;
  }
  __syncthreads();
  for (uint32_t i = tid; i < m; i += kSp) {
    const uint32_t p = inv[i];
    out_k[i] = stK[p];
    out_p[i] = stP[p];
  }
}

// Hot keys get ranges of their own.  A key with thousands of records (a power-law head: real
// click logs have them) makes its uniform range a merge sort's, cold keys and all; found ahead
// of the partition — a strided sample of the keys, sorted by k_sp_sort as one range, runs of
// equal keys in it — it becomes the one-key range [key, key + 1) between the uniform boundaries:
// its records then need their payloads ordered (positions) or nothing at all (rows), and its
// neighbours are ordinary ranges again.
//   k_hot_gather   the sample as records of one range
//   k_sp_sort      (one workgroup) the sample in key order
//   k_hot_ranges   (one workgroup) the keys with runs of >= T samples (T the smallest that leaves
//                  at most kHotMax of them), the boundaries — nR0 uniform ones and two per hot key,
//                  merged; the rest of the nS = nR0 + 2 kHotMax ranges empty behind them — and the
//                  directory over them (kb_bucket's nS buckets)
constexpr uint32_t kHotSample = 8192, kHotMax = 128;
__global__ void k_hot_gather(const uint64_t *__restrict__ keys, uint32_t n, uint32_t ns,
                             Rec3 *__restrict__ samp, uint32_t *__restrict__ sstart,
                             uint64_t *__restrict__ bnd0) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    sstart[0] = 0;
    sstart[1] = ns;
    bnd0[0] = 0;
  }
  if (i >= ns) return;
  const uint32_t j = (uint32_t)(((uint64_t)i * n) / ns);
  const uint64_t k = keys[j];
  samp[i] = Rec3{(uint32_t)k, (uint32_t)(k >> 32), j};
}

__global__ void __launch_bounds__(kKb)
k_hot_ranges(const uint64_t *ssk, uint32_t ns, uint64_t lo, uint64_t span,
             uint32_t nR0, uint32_t nS, uint32_t mult, uint64_t *__restrict__ bnd,
             uint16_t *__restrict__ dir, unsigned int *__restrict__ nhot) {
  extern __shared__ uint64_t hr_lds[];
  uint64_t *lb = hr_lds;        // [nS] the boundaries
  uint64_t *hot = lb + nS;      // [kHotMax]
  uint64_t *sm = hot + kHotMax; // [kHotSample] the sorted sample
  __shared__ uint32_t wsum[kKb / 64];
  const uint32_t tid = threadIdx.x;
  constexpr uint32_t per = kHotSample / kKb;
  for (uint32_t i = tid; i < ns; i += kKb) sm[i] = ssk[i];
  __syncthreads();
  ssk = sm;
  // runs of equal keys in the sorted sample: a thread's `per` neighbouring elements
  auto long_run = [&](uint32_t i, uint32_t T) {
    return i < ns && (i == 0 || ssk[i] != ssk[i - 1]) && i + T - 1 < ns && ssk[i + T - 1] == ssk[i];
  };
  // (from 5 samples: at 1e7 keys a key of ~6000 records — one of 1500, the bench stream's hot
  // field, shows 3 samples one time in seven and would keep the search on for good)
  uint32_t T = 5, H = 0;
  for (;; T += (T + 1) / 2) {  // 5, 8, 12, 18, ... (workgroup-uniform)
    uint32_t c = 0;
    for (uint32_t k = 0; k < per; ++k) c += long_run(tid * per + k, T) ? 1u : 0u;
    uint32_t total;
    const uint32_t e = block_excl_scan(c, wsum, &total);
    if (total <= kHotMax) {
      H = total;
      uint32_t at = e;
      for (uint32_t k = 0; k < per; ++k)
        if (long_run(tid * per + k, T)) hot[at++] = ssk[tid * per + k];
      break;
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) *nhot = H;
  // the boundaries: uniform ones u_r = lo + r * step and, per hot key h, h and h + 1, in order
  // (a uniform one before an equal hot one); the ranges behind them begin at the largest key
  const uint64_t step = span / nR0;
  auto uniform_le = [&](uint64_t x) -> uint32_t {  // uniform boundaries <= x
    if (x < lo) return 0u;
    const uint64_t q = (x - lo) / (step ? step : 1ull);
    return (uint32_t)min(q + 1ull, (uint64_t)nR0);
  };
  auto hot_below = [&](uint64_t x) -> uint32_t {  // hot keys < x (hot[] ascends; a linear pass
    uint32_t l = 0, h = H;                        // per boundary was this kernel's time: 0.2 us a key)
    while (l < h) {
      const uint32_t mid = l + (h - l) / 2;
      if (hot[mid] < x) l = mid + 1;
      else
        h = mid;
    }
    return l;
  };
  for (uint32_t r = tid; r < nR0; r += kKb) {
    const uint64_t u = lo + (uint64_t)r * step;
    // hot boundaries < u: the keys h < u and the successors h + 1 < u, i.e. h < u - 1
    const uint32_t before = hot_below(u) + (u ? hot_below(u - 1ull) : 0u);
    lb[r + before] = u;
  }
  for (uint32_t j = tid; j < 2 * H; j += kKb) {
    const uint64_t h = hot[j >> 1];
    // (the key 2^64 - 1 has no successor: its second boundary is the key itself, an empty range)
    const uint64_t x = (j & 1u) && h != ~0ull ? h + 1ull : h;
    // (h0, h0 + 1, h1, h1 + 1, ... never descends: among the hot boundaries, equal ones in their
    // order, boundary j has j before it)
    lb[uniform_le(x) + j] = x;
  }
  for (uint32_t i = nR0 + 2 * H + tid; i < nS; i += kKb) lb[i] = ~0ull;
  __syncthreads();
  for (uint32_t i = tid; i < nS; i += kKb) bnd[i] = lb[i];
  // dir[b] = ranges that begin in buckets < b
  for (uint32_t b = tid; b <= nS; b += kKb) {
    uint32_t l = 0, h = nS;  // the first i with bucket(lb[i]) >= b
    while (l < h) {
      const uint32_t mid = l + (h - l) / 2;
      if (kb_bucket(lb[mid], lo, mult, nS) < b) l = mid + 1;
      else
        h = mid;
    }
    dir[b] = (uint16_t)l;
  }
}

// SMALL: the ranges of up to kSpSmall records — nearly all of them — with half the LDS (two
// workgroups per CU: a range's load, its few barriers and its store are latency); the launch
// without it takes the others (and lists the heavy ones)
constexpr uint32_t kSpSmall = 4096, kSpSmallPieces = 4096;
constexpr size_t kSpLdsSmall = (size_t)kSpSmall * 14 + ((size_t)kSpSmallPieces + 1 + 2 * kSpList) * 4;
template <bool SMALL>
__global__ void __launch_bounds__(kSp, SMALL ? 8 : 4)  // (waves per SIMD: two workgroups a CU)
k_sp_sort(SpArgs a) {
  const uint32_t S = blockIdx.x;
  const uint32_t s0 = a.sstart[S], m = a.sstart[S + 1] - s0;
  if (m == 0 || (m <= kSpSmall) != SMALL) return;
  if (SMALL) {
    if (a.rows && S + 1 < a.nR && a.bnd[S + 1] == a.bnd[S] + 1ull) return;  // (k_sp_copy's)
    sp_sort_block<kSpSmall, kSpSmallPieces>(a, S, a.rec + s0, m, a.sk + s0, a.spos + s0);
    return;
  }
  // (a hot key's own range, rows as payload: k_sp_copy's — the partition's work items)
  if (a.rows && S + 1 < a.nR && a.bnd[S + 1] == a.bnd[S] + 1ull) return;
  if (m > kSpCap && a.tbits && (1u << a.tbits) <= kSpCap - kSpTileSpan && S + 1 < a.nR &&
      a.bnd[S + 1] == a.bnd[S] + 1ull) {  // a hot key's own range, positions as payload
    if (threadIdx.x == 0) {
      const uint32_t parts = (m + kSpTileSpan - 1) / kSpTileSpan;
      a.hv2[atomicAdd(&a.heavy[4], 1u)] = make_uint2(S, atomicAdd(&a.heavy[5], parts));
    }
    return;
  }
  if (m > kSpCap) {  // (workgroup-uniform) a merge sort's: on the list
    if (threadIdx.x == 0) {
      const uint32_t parts = (m + kSpCap - 1) / kSpCap;
      a.hv[atomicAdd(&a.heavy[0], 1u)] = make_uint2(S, atomicAdd(&a.heavy[1], parts));
      atomicMax(&a.heavy[2], m);
    }
    return;
  }
  sp_sort_block(a, S, a.rec + s0, m, a.sk + s0, a.spos + s0);
}

// the records of the one-key ranges as they are (rows as payload: no order inside a key), a
// workgroup per work item of the partition (kPart records: a head key's 10^6 records copied by
// the range's ONE workgroup were 0.5 ms)
__global__ void __launch_bounds__(kSp)
k_sp_copy(SpArgs a, const uint32_t *__restrict__ items, const uint32_t *__restrict__ nitems) {
  if (blockIdx.x >= *nitems) return;
  const uint32_t item = items[blockIdx.x];
  const uint32_t S = item & 0xFFFFu, part = item >> 16;
  if (!(S + 1 < a.nR && a.bnd[S + 1] == a.bnd[S] + 1ull)) return;
  const uint32_t s0 = a.sstart[S], s1 = a.sstart[S + 1];
  const uint32_t sb = s0 + part * kPart, m = min(s1, sb + kPart) - sb;
  for (uint32_t i = threadIdx.x; i < m; i += kSp) {
    const Rec3 r = a.rec[sb + i];
    a.sk[sb + i] = (uint64_t)r.klo | ((uint64_t)r.khi << 32);
    a.spos[sb + i] = r.rp;
  }
}

// part b of the heavy ranges' parts: its range, the range's first record and length, the part's
// number in the range
struct SpPart {
  uint32_t S, s0, m, t;
};
__device__ __forceinline__ SpPart sp_part_of(const SpArgs &a, uint32_t b, const uint2 *list = nullptr,
                                             uint32_t nlist = 0, uint32_t granule = kSpCap) {
  __shared__ SpPart sp;
  if (!list) {
    list = a.hv;
    nlist = a.heavy[0];
  }
  for (uint32_t e = threadIdx.x; e < nlist; e += blockDim.x) {
    const uint2 h = list[e];
    const uint32_t s0 = a.sstart[h.x], m = a.sstart[h.x + 1] - s0;
    if (b >= h.y && b < h.y + (m + granule - 1) / granule) sp = SpPart{h.x, s0, m, b - h.y};
  }
  __syncthreads();
  return sp;
}

// A hot key's own range, positions as payload: the partition wrote its records tile after tile
// (a tile = 1 << tbits consecutive nonzeros), a tile's records among themselves in no order — so
// the range is in order once every tile group is.  Workgroup t of the range takes the groups
// that begin in its records [t, t + 1) * kSpTileSpan (at most kSpTileSpan + a tile of them).
__global__ void __launch_bounds__(kSp)
k_sp_tiles(SpArgs a) {
  __shared__ uint32_t s_cut[2];
  const SpPart p = sp_part_of(a, blockIdx.x, a.hv2, a.heavy[4], kSpTileSpan);
  const Rec3 *__restrict__ rec = a.rec + p.s0;
  if (threadIdx.x < 128) {  // the first group that begins at or behind x: a wavefront per cut
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = min((p.t + wv) * kSpTileSpan, p.m);
    uint32_t cut = x;
    if (x > 0 && x < p.m) {  // wave-uniform
      const uint32_t tau = rec[x].rp >> a.tbits;
      if ((rec[x - 1].rp >> a.tbits) == tau) {  // inside a group: where the next one begins
        // the first j in [l, h) whose tile is behind tau, or h: 64 probes a round
        uint32_t l = x + 1, h = min(x + (1u << a.tbits), p.m);
        while (l < h) {
          const uint32_t step = (h - l + 63u) / 64u, probe = l + lane * step;
          const bool valid = probe < h;
          const bool pr = valid && (rec[probe].rp >> a.tbits) > tau;
          const unsigned long long bal = __ballot(pr);
          if (!bal) l += (uint32_t)__popcll(__ballot(valid)) * step - step + 1;
          else {
            const uint32_t f = (uint32_t)__ffsll((long long)bal) - 1;
            h = l + f * step;
            l = f ? h - step + 1 : h;
          }
        }
        cut = l;
      }
    }
    if (lane == 0) s_cut[wv] = cut;
  }
  __syncthreads();
  const uint32_t b = s_cut[0], e = s_cut[1];
  if (e <= b) return;  // (workgroup-uniform)
  // one key's records: nothing to count by key (8192 LDS atomics on one counter: 14 us) — staged
  // as they come, ordered by position (sp_long_piece)
  extern __shared__ __attribute__((aligned(16))) unsigned char sp_lds[];
  unsigned long long *stK = (unsigned long long *)sp_lds;
  uint32_t *stP = (uint32_t *)(stK + kSpCap);
  __shared__ uint32_t wsum[kSp / 64];
  const uint32_t m = e - b;
  for (uint32_t i = threadIdx.x; i < m; i += kSp) {
    const Rec3 r = rec[b + i];
    stK[i] = (uint64_t)r.klo | ((uint64_t)r.khi << 32);
    stP[i] = r.rp;
  }
  __syncthreads();
  sp_long_piece(stK, stP, m, wsum);
  for (uint32_t i = threadIdx.x; i < m; i += kSp) {
    a.sk[p.s0 + b + i] = stK[i];
    a.spos[p.s0 + b + i] = stP[i];
  }
}

// a heavy range's parts of kSpCap records, each in (key, position) order where it lies
__global__ void __launch_bounds__(kSp)
k_sp_parts(SpArgs a) {
  const SpPart p = sp_part_of(a, blockIdx.x);
  const uint32_t o = p.t * kSpCap, m = min(kSpCap, p.m - o);
  sp_sort_block(a, p.S, a.rec + p.s0 + o, m, a.sk + p.s0 + o, a.spos + p.s0 + o);
}

// One pass of the heavy ranges' merge sort: runs of L = kSpCap << pass records of (mk, mp) merged
// pair by pair into (ok, op); a range whose runs are one by now takes no part (the host's last
// launch, pass = ~0, copies the ranges that ended in the other buffer).  A workgroup per output
// tile of kSpTile records: the tile's share of the two runs from a search along its first and
// its last diagonal (a wavefront each, 64 probes a round; elements that compare equal — a key
// twice in a row, rows as payload — are the same record twice: either order), both shares into
// LDS, a thread's eight outputs from a search of its own there.
constexpr int kSpM = 512;
constexpr uint32_t kSpTile = 4096;
static_assert(kSpCap % kSpTile == 0 && kSpTile % kSpM == 0, "k_sp_merge: tiles of a part");
__global__ void __launch_bounds__(kSpM)
k_sp_merge(SpArgs a) {
  __shared__ unsigned long long LK[kSpTile];
  __shared__ uint32_t LP[kSpTile];
  __shared__ uint32_t s_ai[2];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  constexpr uint32_t kPer = kSpCap / kSpTile;
  const SpPart p = sp_part_of(a, blockIdx.x / kPer);
  const uint32_t t0 = (p.t * kPer + blockIdx.x % kPer) * kSpTile;
  if (t0 >= p.m) return;
  const uint32_t tn = min(kSpTile, p.m - t0);
  const uint32_t parts = (p.m + kSpCap - 1) / kSpCap;  // >= 2
  const uint32_t passes = 32u - (uint32_t)__clz((int)(parts - 1));
  const uint64_t *__restrict__ mk = a.mk + p.s0;
  const uint32_t *__restrict__ mp = a.mp + p.s0;
  uint64_t *__restrict__ ok = a.ok + p.s0;
  uint32_t *__restrict__ op = a.op + p.s0;
  if (a.pass == 0xFFFFFFFFu) {
    if (passes & 1u)
      for (uint32_t i = tid; i < tn; i += kSpM) {
        ok[t0 + i] = mk[t0 + i];
        op[t0 + i] = mp[t0 + i];
      }
    return;
  }
  if (a.pass >= passes) return;
  const uint32_t L = kSpCap << a.pass;
  const uint32_t gb = (t0 / (2 * L)) * 2 * L;  // the pair's first record
  const uint32_t la = min(L, p.m - gb), lb = min(L, p.m - gb - la);
  const uint64_t *__restrict__ Ak = mk + gb, *__restrict__ Bk = mk + gb + la;
  const uint32_t *__restrict__ Ap = mp + gb, *__restrict__ Bp = mp + gb + la;
  auto less = [](uint64_t k1, uint32_t p1, uint64_t k2, uint32_t p2) {
    return k1 < k2 || (k1 == k2 && p1 < p2);
  };
  const uint32_t o0 = t0 - gb;
  if (tid < 128) {  // elements of A among the pair's first o outputs: wavefront 0 for o0, 1 for o0 + tn
    const uint32_t o = tid < 64 ? o0 : o0 + tn;
    uint32_t lo = o > lb ? o - lb : 0u, hi = min(o, la);
    while (lo < hi) {  // wave-uniform
      const uint32_t step = (hi - lo + 63u) / 64u, mid = lo + lane * step;
      const bool pr = mid < hi && less(Ak[mid], Ap[mid], Bk[o - 1 - mid], Bp[o - 1 - mid]);
      const uint32_t cnt = (uint32_t)__popcll(__ballot(pr));  // (true for the first cnt lanes)
      if (cnt == 0) hi = lo;
      else {
        hi = min(hi, lo + cnt * step);
        lo = lo + (cnt - 1) * step + 1;
      }
    }
    if (lane == 0) s_ai[tid >> 6] = lo;
  }
  __syncthreads();
  const uint32_t a0 = s_ai[0], na = s_ai[1] - a0, b0 = o0 - a0, nb = tn - na;
  for (uint32_t i = tid; i < tn; i += kSpM) {
    LK[i] = i < na ? Ak[a0 + i] : Bk[b0 + i - na];
    LP[i] = i < na ? Ap[a0 + i] : Bp[b0 + i - na];
  }
  __syncthreads();
  constexpr uint32_t E = kSpTile / kSpM;
  const uint32_t d = min(tid * E, tn), de = min(d + E, tn);
  uint32_t x;  // elements of the tile's A share among its first d outputs
  {
    uint32_t lo = d > nb ? d - nb : 0u, hi = min(d, na);
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (less(LK[mid], LP[mid], LK[na + d - 1 - mid], LP[na + d - 1 - mid])) lo = mid + 1;
      else
        hi = mid;
    }
    x = lo;
  }
  uint32_t y = d - x;
  for (uint32_t i = d; i < de; ++i) {
    bool from_a = y >= nb;
    if (!from_a && x < na) from_a = less(LK[x], LP[x], LK[na + y], LP[na + y]);
    const uint32_t j = from_a ? x++ : na + y++;
    ok[t0 + i] = LK[j];
    op[t0 + i] = LP[j];
  }
}


// ------------------------------------------- the worker side of the exchange, LR (round 6)
// LRWorker::update's key build (lr_worker.cc:146-166) where the table is NOT on this GPU — the
// worker side of the weight / gradient exchange (schedules sequential / stale1): the minibatch's
// sorted unique keys (what travels to the owners, once) and its cells over the unique-key index
// (the weights come back as a dense array in that order).  Until round 6: the library's radix sort
// of (key, position), a scatter of the unique index by position (k_uidx_coo: 1e7 4-byte stores),
// the panel-major and key-grouped views nobody reads on this path, and a second library sort on
// the cell number (cells_build).  Now:
//   the sort above with the nonzero's ROW as the payload (the partition walks the CSR anyway)
//   k_wl_count / _scan / _unique   the sorted list's heads: every record's unique index, the keys
//   k_wc_cells<false>              a range's records counted by cell — its keys' indices are
//                                  consecutive: (row windows) x (a chunk or two), LDS counters
//   k_kb_psum / k_kb_scan          (cell part) cellptr, the gradient's work items
//   k_wc_cells<true>               the entries: slots taken from LDS cursors, one atomic on a
//                                  cell's cursor per (work item, round, cell)
// Inside a cell the entries come in no particular order (as the keyed build's).
constexpr uint32_t kWlBlock = 4 * kKb;  // records per workgroup of k_wl_count / k_wl_unique
struct WlArgs {
  const uint64_t *sk;   // [n] sorted keys
  uint32_t n, nb;
  uint32_t *bsum;       // [nb + 1] heads per block, then (k_wl_scan) before every block; [nb] = U
  uint32_t *su;         // [n] out: unique index of every record
  uint64_t *ukeys;      // [U] out
};
__device__ __forceinline__ uint32_t wl_heads(const WlArgs &a, uint32_t j0, bool *h) {
  uint32_t cnt = 0;
  uint64_t prev = j0 > 0 && j0 <= a.n ? a.sk[j0 - 1] : 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t j = j0 + q;
    const uint64_t k = j < a.n ? a.sk[j] : 0;
    h[q] = j < a.n && (j == 0 || k != prev);
    cnt += h[q] ? 1u : 0u;
    prev = k;
  }
  return cnt;
}
__global__ void __launch_bounds__(kKb)
k_wl_count(WlArgs a) {
  __shared__ uint32_t wsum[kKb / 64];
  bool h[4];
  const uint32_t cnt = wl_heads(a, blockIdx.x * kWlBlock + threadIdx.x * 4, h);
  uint32_t total;
  (void)block_excl_scan(cnt, wsum, &total);
  if (threadIdx.x == 0) a.bsum[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kKb)
k_wl_scan(WlArgs a) {  // one workgroup
  __shared__ uint32_t wsum[kKb / 64];
  uint32_t carry = 0;
  for (uint32_t i0 = 0; i0 < a.nb; i0 += kKb) {  // workgroup-uniform
    const uint32_t i = i0 + threadIdx.x, x = i < a.nb ? a.bsum[i] : 0u;
    uint32_t total;
    const uint32_t e = block_excl_scan(x, wsum, &total);
    if (i < a.nb) a.bsum[i] = carry + e;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) a.bsum[a.nb] = carry;
}
__global__ void __launch_bounds__(kKb)
k_wl_unique(WlArgs a) {
  __shared__ uint32_t wsum[kKb / 64];
  bool h[4];
  const uint32_t j0 = blockIdx.x * kWlBlock + threadIdx.x * 4;
  const uint32_t cnt = wl_heads(a, j0, h);
  uint32_t total;
  uint32_t u = a.bsum[blockIdx.x] + block_excl_scan(cnt, wsum, &total);  // heads before j0
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const uint32_t j = j0 + q;
    if (j >= a.n) break;
    if (h[q]) a.ukeys[u++] = a.sk[j];
    a.su[j] = u - 1;
  }
}

struct WcArgs {
  const uint32_t *su;      // [n] the sorted records' unique-key indices
  const uint32_t *srp;     // [n] ... rows (window << kRinBits | row in window)
  const uint32_t *sstart;  // [nR + 1] the key ranges' records
  const uint32_t *items, *nitems;  // the partition's work items (a range | its part << 16)
  uint32_t n, max_items, nwin, nchunk;
  uint32_t *hist, *cellcur, *entries;
  const uint32_t *cellptr;
  uint32_t *blk_cell;
};
// counter l of every lane that is `on` up by one, the value before it returned: a range's records
// fall into a handful of cells (row windows x a chunk or two; ONE chunk for a hot key's range),
// so the lanes of a wavefront that hold the same counter share one LDS atomic — up to eight
// distinct counters a wavefront, the lanes left over one by one
__device__ __forceinline__ uint32_t wc_slot(uint32_t *cnt, uint32_t l, bool on) {
  const uint32_t lane = threadIdx.x & 63u;
  unsigned long long todo = __ballot(on);
  uint32_t slot = 0;
  bool mine = false;
  for (int round = 0; todo && round < 8; ++round) {  // wave-uniform
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t l0 = (uint32_t)__builtin_amdgcn_readlane((int)l, leader);
    const bool same = on && !mine && l == l0;
    const unsigned long long ms = __ballot(same);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&cnt[l0], (uint32_t)__popcll(ms));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
    if (same) {
      slot = base + (uint32_t)__popcll(ms & ((1ull << lane) - 1ull));
      mine = true;
    }
    todo &= ~ms;
  }
  if (on && !mine) slot = atomicAdd(&cnt[l], 1u);
  return slot;
}

// PLACE = false: the cell counts; true: the entries (the workgroups beyond the items: blk_cell)
template <bool PLACE>
__global__ void __launch_bounds__(kEb)
k_wc_cells(WcArgs a) {
  __shared__ uint32_t lcnt[kEbCells], lcur[kEbCells];
  const uint32_t tid = threadIdx.x;
  if (PLACE && blockIdx.x >= a.max_items) {
    ar_blk_cell(a, blockIdx.x - a.max_items, gridDim.x - a.max_items);
    return;
  }
  if (blockIdx.x >= *a.nitems) return;
  const uint32_t item = a.items[blockIdx.x];
  const uint32_t S = item & 0xFFFFu, part = item >> 16;
  const uint32_t s0 = a.sstart[S], s1 = a.sstart[S + 1];
  const uint32_t sb = s0 + part * kPart, m = min(s1, sb + kPart) - sb;
  // the range's keys have consecutive indices: a chunk or two (keys that are no hashes — a
  // range of thousands of chunks — count and take their slots in memory)
  const uint32_t c_lo = a.su[s0] >> kChunkBits, nloc = (a.su[s1 - 1] >> kChunkBits) - c_lo + 1;
  const uint32_t nl = a.nwin * nloc;
  const bool lds = nloc <= kEbCells && nl <= kEbCells;  // workgroup-uniform
  const bool crowded = s1 - s0 > kSpCap;
  if (lds)
    for (uint32_t i = tid; i < nl; i += kEb) lcnt[i] = 0;
  __syncthreads();
  for (uint32_t i0 = 0; i0 < m; i0 += kEb * 4) {
    uint32_t rp[4], u[4], lc[4], at[4];
    bool ok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t i = i0 + q * kEb + tid;
      ok[q] = i < m;
      rp[q] = ok[q] ? a.srp[sb + i] : 0u;
      u[q] = ok[q] ? a.su[sb + i] : c_lo << kChunkBits;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t v = rp[q] >> kRinBits, ch = u[q] >> kChunkBits;
      lc[q] = v * nloc + (ch - c_lo);
      at[q] = 0;
      if (lds) {  // (workgroup-uniform: every lane takes part in the ballots)
        // a range of thousands of records is a hot key's: its records fall into its chunk's few
        // cells (plain atomics: 91 + 112 us on a Zipf(1.1) minibatch, shared ones 66 + 65); an
        // ordinary range's cells are few as well, but its lanes' atomics do not queue for long
        // (shared ones there: 32 + 34 -> 47 + 48 us)
        if (crowded) at[q] = wc_slot(lcnt, lc[q], ok[q]);
        else if (ok[q])
          at[q] = atomicAdd(&lcnt[lc[q]], 1u);
        continue;
      }
      if (!ok[q]) continue;
      if (!PLACE)
        atomicAdd(&a.hist[v * a.nchunk + ch], 1u);
      else
        a.entries[atomicAdd(&a.cellcur[v * a.nchunk + ch], 1u)] =
            ((ch & xf::kTagMask) << kTagShift) | ((rp[q] & ((1u << kRinBits) - 1u)) << kChunkBits) |
            (u[q] & (kChunk - 1));
    }
    if (PLACE && lds) {  // this round's records take their slots: a cursor moves once per round and cell
      __syncthreads();
      for (uint32_t c = tid; c < nl; c += kEb) {
        const uint32_t k = lcnt[c];
        if (k) {
          const uint32_t v = c / nloc, ch = c_lo + (c - v * nloc);
          lcur[c] = atomicAdd(&a.cellcur[v * a.nchunk + ch], k);
          lcnt[c] = 0;
        }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ok[q]) {
          const uint32_t ch = u[q] >> kChunkBits;
          a.entries[lcur[lc[q]] + at[q]] = ((ch & xf::kTagMask) << kTagShift) |
                                           ((rp[q] & ((1u << kRinBits) - 1u)) << kChunkBits) |
                                           (u[q] & (kChunk - 1));
        }
      __syncthreads();
    }
  }
  if (!PLACE && lds) {
    __syncthreads();
    for (uint32_t c = tid; c < nl; c += kEb) {
      const uint32_t k = lcnt[c];
      if (k) {
        const uint32_t v = c / nloc;
        atomicAdd(&a.hist[v * a.nchunk + c_lo + (c - v * nloc)], k);
      }
    }
  }
}

}  // namespace

namespace xf {

// where the FM build's U-sized arrays go: ukeys [U], urow [U], segptr [U + 1], coo [NNZ]
typedef int (*FmKeyedOut)(void *ctx, uint32_t U, uint64_t **ukeys, uint32_t **urow,
                          uint32_t **segptr, uint32_t **coo);

static KbSummary *summary_buf() {
  static thread_local KbSummary *p = nullptr;
  if (!p && hipHostMalloc((void **)&p, sizeof(KbSummary)) != hipSuccess) p = nullptr;
  return p;
}

// phase timestamps of the last keyed build (exp_knob 200 turns them on): [hist workgroups |
// scatter workgroups | resolve workgroups] x kDbgSlots, wall_clock64 ticks (10 ns)
static unsigned long long *g_dbg = nullptr;
[[maybe_unused]] static size_t g_dbg_n = 0;
static size_t g_dbg_used = 0;
static uint32_t g_dbg_shape[3] = {0, 0, 0};

static int arrival_build(xf_cells **out, xf_table *t, const uint64_t *d_keys,
                         const uint32_t *d_rowptr, const uint32_t *d_rowid, uint32_t R, uint32_t n,
                         bool ksc, uint32_t w_fixed, uint32_t chunk0, hipStream_t s, bool *done);

// no settled tier (a table's first minibatches), or beyond the keyed build's limits
static int general_build(xf_cells **out, xf_table *t, const uint64_t *d_keys,
                         const uint32_t *d_rowptr, const uint32_t *d_rowid, uint32_t R,
                         uint32_t NNZ, bool ksc, uint32_t w_fixed, hipStream_t s) {
  if (table_dev(t).nbase == 0) {  // every key goes through the arrival index: the first-touch build
    bool done = false;
    XF_TRY(arrival_build(out, t, d_keys, d_rowptr, d_rowid, R, NNZ, ksc, w_fixed, 0, s, &done));
    if (done) return XF_OK;
  }
  // the sort-based build: a probe of the table per nonzero + a radix pass on the cell number
  Scratch sc;
  uint32_t *idx = nullptr;
  XF_TRY(sc.get(&idx, NNZ));
  if (NNZ) XF_TRY(table_resolve_any(t, d_keys, NNZ, idx, s, true));
  const uint64_t M = table_dev(t).max_rows + 1;
  return cells_build(out, idx, nullptr, d_rowptr, R, NNZ, (uint32_t)M, kCellsTableRows, ksc, s,
                     NNZ ? d_rowid : nullptr, w_fixed);
}

#define XF_KB_LAUNCH_N(kern, grid, threads, lds, args)                                         \
  do {                                                                                         \
    static bool attr_done = false;                                                             \
    if (!attr_done) {                                                                          \
      XF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)kDynMax));                                               \
      attr_done = true;                                                                        \
    }                                                                                          \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, args);                         \
  } while (0)

// The first-touch build (kernels: "first touch" above): the n nonzeros d_keys[] — CSR order with
// d_rowptr, or any order with their row numbers d_rowid (rows numbered window by window, w_fixed
// per window) — NONE of whose keys the settled tier holds are looked up in the arrival index
// (insert on miss, ftrl.h:56), and *out are their cells over the chunks from chunk0 on.  The
// table may grow first.  Synchronises the stream.  *done = false: beyond this build's limits,
// nothing was done (the caller takes the sort-based build).
constexpr uint32_t kArMaxRanges = 4000;  // (beyond ~1900 the scatter takes half tiles: its LDS)
static int arrival_build(xf_cells **out, xf_table *t, const uint64_t *d_keys,
                         const uint32_t *d_rowptr, const uint32_t *d_rowid, uint32_t R, uint32_t n,
                         bool ksc, uint32_t w_fixed, uint32_t chunk0, hipStream_t s, bool *done) {
  *done = false;
  KbSummary *sum = summary_buf();
  if (n == 0 || n >= (1u << 30) || !sum ||
      ((uint64_t)n + kTile / 2 - 1) / (kTile / 2) > (uint64_t)kMaxSub * 256 || key_build_mode() == 1)
    return XF_OK;
  {
    const TableDev &T0 = table_dev(t);
    const uint32_t nwin0 = w_fixed ? std::max<uint32_t>(1, (R + w_fixed - 1) / w_fixed)
                                   : std::max<uint32_t>(1, (R + kWinMax - 1) / kWinMax);
    // (the cell histogram: a table that grows below has more chunks — bounded by twice the rows
    // this minibatch can add)
    const uint64_t rows_bound = std::max<uint64_t>(T0.max_rows + 1, 2 * ((uint64_t)n + T0.max_rows));
    if ((rows_bound / kChunk + 1) * nwin0 >= (1ull << 22)) return XF_OK;
  }
  // the partition first: it needs nothing of the table but its key range — and the growth check
  // below counts over the partitioned records
  const TableDev T0 = table_dev(t);
  const uint32_t nwin = w_fixed ? std::max<uint32_t>(1, (R + w_fixed - 1) / w_fixed)
                                : std::max<uint32_t>(1, (R + kWinMax - 1) / kWinMax);
  const uint32_t W = w_fixed ? w_fixed : std::max<uint32_t>(1, (R + nwin - 1) / nwin);
  Scratch sc;
  const uint32_t nR = std::min<uint32_t>(kArMaxRanges, std::max<uint32_t>(1, (n + kArBatch - 1) / kArBatch));
  KbArgs a{};
  a.keys = d_keys;
  a.rowptr = d_rowid ? nullptr : d_rowptr;
  a.rowid = d_rowid;
  a.R = R;
  a.NNZ = n;
  a.W = W;
  a.nwin = nwin;
  a.nS = nR;
  a.tile = scatter_lds_bytes(nR, kTile) <= kDynMax ? kTile : kTile / 2;
  a.ntile = (n + a.tile - 1) / a.tile;
  a.lo = T0.lo;
  const uint32_t sub = std::max<uint32_t>(1, (a.ntile + 255) / 256);
  a.span = sub * a.tile;
  a.nW = (a.ntile + sub - 1) / sub;
  a.npc = 0;  // (the scan's per-range part alone)
  const unsigned max_items = nR + n / kPart + 1;
  uint32_t *part1 = nullptr;
  const size_t n_part1 = (size_t)nR * 2 + 1 + max_items + 1 + (size_t)a.nW * nR + a.ntile + 1 + 2;
  XF_TRY(sc.get(&part1, n_part1));
  a.scount = part1;
  a.sstart = a.scount + nR;
  a.items = a.sstart + nR + 1;
  a.nitems = a.items + max_items;
  a.wgcnt = a.nitems + 1;
  a.tile_r0 = a.wgcnt + (size_t)a.nW * nR;
  unsigned long long *d_absent = (unsigned long long *)(((uintptr_t)(a.tile_r0 + a.ntile + 1) + 7) & ~(uintptr_t)7);
  uint64_t *bnd = nullptr;
  uint16_t *dir = nullptr;
  uint32_t *rec_row = nullptr;
  XF_TRY(sc.get(&bnd, nR));
  XF_TRY(sc.get(&dir, nR + 1));
  XF_TRY(sc.get(&a.rec, n));
  XF_TRY(sc.get(&rec_row, n));
  a.sc.bnd = bnd;
  a.sc.dir = dir;
  a.sc.n = nR;
  a.sc.mult = (uint32_t)std::min<uint64_t>(((uint64_t)nR << 32) / ((T0.span >> 32) + 1), 0xFFFFFFFFull);
  hipLaunchKernelGGL(k_ar_ranges, dim3((nR + 256) / 256), dim3(256), 0, s, T0.lo, nR, a.sc.mult, bnd,
                     dir);
  a.scan_part = 1;
  if (d_rowid) XF_KB_LAUNCH_N((k_kb_hist_groups<true>), a.nW, kKb, hist_groups_lds_bytes(nR), a);
  else
    XF_KB_LAUNCH_N((k_kb_hist_groups<false>), a.nW, kKb, hist_groups_lds_bytes(nR), a);
  hipLaunchKernelGGL(k_kb_scan, dim3(kPlanWgs + (nR + kKb / 64 - 1) / (kKb / 64)), dim3(kKb), 0, s, a);
  const size_t sl = scatter_lds_bytes(nR, a.tile);
  if (a.tile == kTile) {
    if (d_rowid) XF_KB_LAUNCH_N((k_kb_scatter<true, kTile>), a.nW, kKb, sl, a);
    else
      XF_KB_LAUNCH_N((k_kb_scatter<false, kTile>), a.nW, kKb, sl, a);
  } else {
    if (d_rowid) XF_KB_LAUNCH_N((k_kb_scatter<true, kTile / 2>), a.nW, kKb, sl, a);
    else
      XF_KB_LAUNCH_N((k_kb_scatter<false, kTile / 2>), a.nW, kKb, sl, a);
  }
  XF_HIP(hipGetLastError());
  ArArgs r{};
  r.rec = a.rec;
  r.sstart = a.sstart;
  r.items = a.items;
  r.nitems = a.nitems;
  r.n = n;
  r.nwin = nwin;
  r.chunk0 = chunk0;
  r.rec_row = rec_row;
  r.bnd = bnd;
  r.nR = nR;
  // room for the keys that may be new (the table's count: a wait for the stream, which the
  // partition's kernels share) — and, of a table that holds nothing, the whole tier at once
  // (kernels: "an empty table")
  EbArgs e{};
  bool eb = false;
  uint64_t eb_keys = 0;
  // (the keys the host API put there before the first minibatch, lr_worker.cc:180-182: taken out
  // of the arrival index, put back when the tier stands — found in it, or inserted behind it)
  std::vector<uint64_t> early;
  uint64_t *d_early = nullptr;
  float *early_w = nullptr;
  float2 *early_nz = nullptr;
  uint32_t *early_rows = nullptr;
  unsigned long long *early_pos = nullptr;
  // (launched before the table's count is known — one wait for both; a table that cannot be a
  // first-minibatch table, the host knows without asking)
  const bool maybe = T0.nbase == 0 && chunk0 == 0 && key_build_mode() != 3 &&
                     (uint64_t)nwin * 6 <= kEbCells && table_maybe_first(t);
  unsigned int *hout = (unsigned int *)&sum->miss;  // (pinned: flag, distinct keys)
  if (maybe) {
    e.T = T0;
    e.rec = a.rec;
    e.sstart = a.sstart;
    e.bnd = bnd;
    e.nR = nR;
    e.n = n;
    e.max_items = max_items;
    e.rec_row = rec_row;
    e.items = a.items;
    e.nitems = a.nitems;
    XF_TRY(sc.get(&e.ukeys, n));
    XF_TRY(sc.get(&e.pkeys, n));
    XF_TRY(sc.get(&e.pmap, n));
    XF_TRY(sc.get(&e.dbase, (size_t)nR + 2 + 2 + max_items));
    e.out = e.dbase + nR + 2;
    e.pd = e.out + 2;
    // (an empty range has no item to write its count; a part that gives up leaves its count 0)
    XF_HIP(hipMemsetAsync(e.dbase, 0, ((size_t)nR + 4 + max_items) * 4, s));
    XF_KB_LAUNCH_N(k_eb_rank<false>, max_items, kEbR, kEbLds, e);
    XF_KB_LAUNCH_N(k_eb_rank<true>, max_items, kEbR, kEbLds, e);  // (heavy ranges' first parts)
    hipLaunchKernelGGL(k_eb_scan, dim3(1), dim3(1024), 0, s, e);
    XF_HIP(hipMemcpyAsync(hout, e.out, 8, hipMemcpyDeviceToHost, s));
    XF_HIP(hipGetLastError());
  }
  {
    uint64_t count = 0;
    XF_TRY(table_count(t, s, &count));
    eb = maybe && (count == 0 || table_early_keys(t, count, &early));
    if (eb) {
      const uint64_t distinct = hout[1];
      eb = hout[0] == 0 && distinct > 0;
      eb_keys = distinct;
      if (eb) {
        const uint64_t all = distinct + early.size();
        if (all * 10 > T0.cap * 6) {  // (the arrival index's load rule, for the rows' sake)
          uint64_t want = T0.cap * 2;
          while (all * 10 > want * 6) want *= 2;
          XF_TRY(xf_table_reserve(t, want));
        }
        XF_REQUIRE(all <= table_dev(t).max_rows, "first-touch build: %llu keys, %llu rows",
                   (unsigned long long)all, (unsigned long long)table_dev(t).max_rows);
        XF_TRY(table_first_tier(t, (size_t)distinct, &e.bkeys));
        if (!early.empty()) {
          const size_t ne = early.size(), dim = (size_t)table_dev(t).dim;
          unsigned long long *d_pos = nullptr;
          XF_TRY(sc.get(&d_early, ne));
          XF_TRY(sc.get(&d_pos, ne));
          XF_TRY(sc.get(&early_w, ne * dim));
          XF_TRY(sc.get(&early_nz, ne * dim));
          XF_TRY(sc.get(&early_rows, ne));
          XF_HIP(hipMemcpyAsync(d_early, early.data(), ne * 8, hipMemcpyHostToDevice, s));
          early_pos = d_pos;
        }
      }
    }
    if (!eb && (count + n) * 10 > T0.cap * 6) {  // the records alone would say "grow": count the new keys
      XF_HIP(hipMemsetAsync(d_absent, 0, 8, s));
      r.T = T0;
      hipLaunchKernelGGL(k_ar_absent, dim3(max_items), dim3(kAr), 0, s, r, d_absent);
      unsigned long long absent = 0;
      XF_HIP(hipMemcpyAsync(&absent, d_absent, 8, hipMemcpyDeviceToHost, s));
      XF_HIP(hipGetLastError());
      XF_HIP(hipStreamSynchronize(s));
      if ((count + absent) * 10 > T0.cap * 6) {
        uint64_t want = T0.cap * 2;
        while ((count + absent) * 10 > want * 6) want *= 2;
        XF_TRY(xf_table_reserve(t, want));
      }
    }
  }
  const TableDev T = table_dev(t);
  xf_cells *c = nullptr;
  XF_TRY(cells_alloc(&c, R, n, (uint32_t)(T.max_rows + 1), kCellsTableRows, ksc, w_fixed, chunk0));
  struct Guard {
    xf_cells *c;
    ~Guard() {
      if (c) cells_free(c);
    }
  } guard{c};
  XF_REQUIRE(c->W == W && c->nwin == nwin, "arrival_build: geometry");
  const size_t ncell = (size_t)c->nwin * c->nchunk;
  a.cA = c->nchunk;
  a.npc = (uint32_t)((ncell + kScanPiece - 1) / kScanPiece);
  uint32_t *small = nullptr;
  const size_t n_zero = 4 + ncell;  // the summary and the cell histogram: cleared together
  XF_TRY(sc.get(&small, n_zero + ncell + a.npc));
  a.sum = (KbSummary *)small;
  a.hist = small + 4;
  a.cellcur = a.hist + ncell;
  a.psum = a.cellcur + ncell;
  a.cellptr = c->cellptr;
  a.entries = c->entries;
  a.plan = c->plan;
  a.blk_cell = c->blk_cell;
  XF_HIP(hipMemsetAsync(small, 0, n_zero * 4, s));
  r.T = T;
  r.nchunk = c->nchunk;
  r.hist = a.hist;
  r.cellcur = a.cellcur;
  r.cellptr = c->cellptr;
  r.entries = c->entries;
  r.blk_cell = c->blk_cell;
  if (eb) {
    e.T = T;
    e.nwin = nwin;
    e.nchunk = c->nchunk;
    e.hist = a.hist;
    e.cellcur = a.cellcur;
    e.entries = c->entries;
    if (!early.empty())  // (from here on nothing fails before they are back)
      XF_TRY(table_take_early(t, d_early, early.size(), early_w, early_nz, early_pos, s));
    hipLaunchKernelGGL(k_eb_cells<false>, dim3(max_items), dim3(kEb), 0, s, e, r);
    XF_HIP(hipGetLastError());
    XF_TRY(table_settle_first(t, (size_t)eb_keys, s));
    if (!early.empty())
      XF_TRY(table_put_early(t, d_early, early.size(), early_w, early_nz, early_rows, s));
  } else {
    hipLaunchKernelGGL(k_ar_insert<true>, dim3(max_items), dim3(kAr), 0, s, r);
    hipLaunchKernelGGL(k_ar_insert<false>, dim3(max_items), dim3(kAr), 0, s, r);
  }
  a.scan_part = 2;
  if (a.npc > 1)
    hipLaunchKernelGGL(k_kb_psum, dim3(a.npc + a.cA / kKb + 1), dim3(kKb), 0, s, a);
  hipLaunchKernelGGL(k_kb_scan, dim3(a.npc + kPlanWgs), dim3(kKb), 0, s, a);
  if (eb) {
    hipLaunchKernelGGL(k_eb_cells<true>, dim3(max_items + std::min<uint32_t>(64, nR / 8 + 1)),
                       dim3(kEb), 0, s, e, r);
  } else {
    const uint32_t nwg = (n + kRes * 4 - 1) / (kRes * 4);
    hipLaunchKernelGGL(k_ar_place, dim3(nwg + std::min<uint32_t>(64, nwg / 8 + 1)), dim3(kRes), 0, s,
                       r);
  }
  XF_HIP(hipMemcpyAsync(sum, a.sum, sizeof(KbSummary), hipMemcpyDeviceToHost, s));
  XF_HIP(hipGetLastError());
  XF_HIP(hipStreamSynchronize(s));
  XF_TRY(cells_fill_items(c, sum->nitems, sum->nsplit, s));
  XF_TRY(cells_key_sorted_copy(c, s));
  XF_HIP(hipStreamSynchronize(s));  // (the scratch goes back)
  guard.c = nullptr;
  table_note_rows_out(t);  // (the cells hold row numbers)
  *out = c;
  *done = true;
  return XF_OK;
}

// the chunk / super-chunk boundaries of the table's settled tier, their directories and the
// directories over every super-chunk's keys: one device allocation kept with the table,
// rebuilt when the tier changes (xf_table_defrag)
// gshift != 0 (two-level build): also the ranges of the groups of 1 << gshift super-chunks (*gr)
static int kb_index(xf_table *t, const TableDev &T, uint32_t cA, uint32_t nS, KbArgs *a,
                    hipStream_t s, uint32_t gshift = 0, KbRanges *gr = nullptr) {
  uint64_t *ep = nullptr;
  void **slot = table_aux(t, &ep);
  auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
  const uint32_t nG = gshift ? (nS + (1u << gshift) - 1) >> gshift : 0u;
  const size_t o_cb = 0, o_sb = o_cb + al((size_t)cA * 8), o_sm = o_sb + al((size_t)nS * 8);
  const size_t o_cd = o_sm + al((size_t)nS * 8), o_sd = o_cd + al(((size_t)cA + 1) * 2);
  const size_t o_sdirs = o_sd + al(((size_t)nS + 1) * 2);
  const size_t o_gb = o_sdirs + al((size_t)nS * kDirStride * 2);
  const size_t o_gd = o_gb + al((size_t)nG * 8);
  const size_t total = o_gd + al(((size_t)nG + 1) * 2);
  // (the allocation is laid out for one group size: another one — a test forcing the two-level
  // build on a small table — rebuilds it)
  static thread_local void *aux_ptr = nullptr;
  static thread_local uint32_t aux_gshift = 0;
  KbRanges *ch = &a->ch, *sc = &a->sc;
  ch->n = cA;
  sc->n = nS;
  auto mult32 = [&](uint32_t n) -> uint32_t {
    const uint64_t m = ((uint64_t)n << 32) / ((T.span >> 32) + 1);
    return (uint32_t)std::min<uint64_t>(m, 0xFFFFFFFFull);
  };
  ch->mult = mult32(cA);
  sc->mult = mult32(nS);
  const bool fresh = !*slot || *ep != table_epoch(t) ||
                     (gshift && (aux_ptr != *slot || aux_gshift != gshift));
  if (fresh) {
    if (*slot) {
      XF_HIP(hipDeviceSynchronize());
      XF_HIP(hipFree(*slot));
      *slot = nullptr;
    }
    XF_HIP(hipMalloc(slot, total));
    *ep = table_epoch(t);
    aux_ptr = *slot;
    aux_gshift = gshift;
  }
  char *d = (char *)*slot;
  if (gr) {
    gr->n = nG;
    gr->bnd = (const uint64_t *)(d + o_gb);
    gr->dir = (const uint16_t *)(d + o_gd);
  }
  ch->bnd = (const uint64_t *)(d + o_cb);
  ch->dir = (const uint16_t *)(d + o_cd);
  sc->bnd = (const uint64_t *)(d + o_sb);
  sc->dir = (const uint16_t *)(d + o_sd);
  a->smult = (const uint64_t *)(d + o_sm);
  a->sdirs = (const uint16_t *)(d + o_sdirs);
  if (fresh) {
    hipLaunchKernelGGL(k_kb_index, dim3((cA + 256) / 256), dim3(256), 0, s, T.bkeys, T.lo, cA,
                       kChunkBits, ch->mult, (uint64_t *)ch->bnd, (uint16_t *)ch->dir);
    hipLaunchKernelGGL(k_kb_index, dim3((nS + 256) / 256), dim3(256), 0, s, T.bkeys, T.lo, nS,
                       kChunkBits + kSCShift, sc->mult, (uint64_t *)sc->bnd, (uint16_t *)sc->dir);
    hipLaunchKernelGGL(k_kb_sdir, dim3(nS), dim3(256), 0, s, T.bkeys, (uint32_t)T.nbase,
                       (uint16_t *)a->sdirs, (uint64_t *)a->smult);
    if (gr) {
      gr->mult = mult32(nG);
      hipLaunchKernelGGL(k_kb_index, dim3((nG + 256) / 256), dim3(256), 0, s, T.bkeys, T.lo, nG,
                         kChunkBits + kSCShift + (int)gshift, gr->mult, (uint64_t *)gr->bnd,
                         (uint16_t *)gr->dir);
    }
    XF_HIP(hipGetLastError());
  }
  if (gr) gr->mult = mult32(nG);
  return XF_OK;
}

// what a keyed build holds on to between its kernels and its one host wait
struct KbDeferred {
  Scratch sc;  // the records, the miss list, the small arrays
  KbArgs a{};
  xf_cells *c = nullptr;
  xf_table *t = nullptr;
  KbSummary *sum = nullptr;
  bool ksc = false, waited = false;
  uint32_t R = 0, NNZ = 0;
  uint64_t nbase = 0;
  // the build's kernels and the copy of the summary are done: its OWN event (two host threads
  // with a deferred build each must not wait on one another's record), on the device the build
  // runs on
  hipEvent_t ev = nullptr;
  ~KbDeferred() {
    if (ev && !device_poisoned()) (void)hipEventDestroy(ev);
  }
};

// after the build's kernels and the copy of the summary have finished: the work items, the
// key-sorted copy, and the second segment for the keys the tier does not hold
static int keyed_tail(KbDeferred &d, hipStream_t s, bool *more) {
  xf_cells *c = d.c;
  const unsigned long long misses = d.sum->miss;
  if (more) *more = misses != 0;
  XF_TRY(cells_fill_items(c, d.sum->nitems, d.sum->nsplit, s));
  XF_TRY(cells_key_sorted_copy(c, s));
  if (misses) {
    // segment B: the keys the tier did not hold, through the general path (insert on first
    // touch), over the rows from the tier's last chunk on
    XF_REQUIRE(misses <= d.NNZ, "cells_build_keyed: miss list");
    // the insertions may move the table's arrays: a step of ANOTHER minibatch that a caller
    // keeps running on a second stream while this one is built (the build itself reads only
    // the tier's keys, which no step writes) has to be over first
    XF_HIP(hipDeviceSynchronize());
    bool done = false;
    XF_TRY(arrival_build(&c->next, d.t, d.a.missK, nullptr, d.a.missR, d.R, (uint32_t)misses, d.ksc,
                         c->W, (uint32_t)(d.nbase >> kChunkBits), s, &done));
    if (!done) {  // (beyond the first-touch build's limits: the sort-based build)
      Scratch sc2;
      uint32_t *idx = nullptr;
      XF_TRY(sc2.get(&idx, (size_t)misses));
      XF_TRY(table_resolve_any(d.t, d.a.missK, (size_t)misses, idx, s, true));
      const uint64_t M = table_dev(d.t).max_rows + 1;
      XF_TRY(cells_build(&c->next, idx, nullptr, nullptr, d.R, (uint32_t)misses, (uint32_t)M,
                         kCellsTableRows, d.ksc, s, d.a.missR, c->W,
                         (uint32_t)(d.nbase >> kChunkBits)));
    }
    XF_REQUIRE(c->next->nwin == c->nwin && c->next->G == c->G, "cells_build_keyed: segments");
    // (the general build synchronises: every kernel that reads the scratch has finished)
  }
  return XF_OK;
}

int cells_build_keyed_finish(KbDeferred *d, hipStream_t s, bool *more) {
  XF_REQUIRE(d, "cells_build_keyed_finish: null argument");
  struct Free {
    KbDeferred *d;
    ~Free() { delete d; }
  } guard{d};
  XF_HIP(hipEventSynchronize(d->ev));
  return keyed_tail(*d, s, more);
}

int cells_build_keyed(xf_cells **out, xf_table *t, const uint64_t *d_keys,
                      const uint32_t *d_rowptr, const uint32_t *d_rowid, uint32_t R,
                      uint32_t NNZ, bool ksc, uint32_t w_fixed, hipStream_t s,
                      KbDeferred **defer) {
  if (defer) *defer = nullptr;
  XF_REQUIRE(out && t && (d_rowptr || d_rowid || NNZ == 0) && (NNZ == 0 || d_keys),
             "cells_build_keyed: null argument");
  XF_REQUIRE(!d_rowid || (w_fixed >= 1 && w_fixed <= kWinMax),
             "cells_build_keyed: row ids need a window size");
  const TableDev T = table_dev(t);
  const uint64_t cA64 = (T.nbase + kChunk - 1) / kChunk;
  const uint64_t nS64 = (cA64 + kSC - 1) / kSC;
  const uint32_t nwin = w_fixed ? std::max<uint32_t>(1, (R + w_fixed - 1) / w_fixed)
                                : std::max<uint32_t>(1, (R + kWinMax - 1) / kWinMax);
  KbSummary *sum = summary_buf();
  const bool common = T.nbase > 0 && T.nbase < 0xFFFF0000ull && NNZ > 0 && NNZ < (1u << 30) &&
                      sum != nullptr && cA64 * nwin < (1ull << 22) &&
                      ((uint64_t)NNZ + kTile / 2 - 1) / (kTile / 2) <= (uint64_t)kMaxSub * 256 &&
                      key_build_mode() != 1;
  bool fits = common && cA64 < 0xFFFFu &&
              scatter_lds_bytes((uint32_t)nS64, kTile / 2) <= kDynMax &&
              hist_lds_bytes((uint32_t)cA64, (uint32_t)nS64, false) <= kDynMax;
  // the two-level build (this file's header): groups of 1 << gshift super-chunks, as few of
  // them as the full-tile scatter holds.  key_build = 2 (xf_tune) takes it on a table that does
  // not need it (tests: two super-chunks per group).
  uint32_t gshift = 0;
  if (common && (!fits || key_build_mode() == 2) && nS64 >= 2 && nS64 < 0xFFFFu)
    for (uint32_t g = 1; g <= kGrpShiftMax && !gshift; ++g) {
      const uint32_t nG = (uint32_t)((nS64 + (1u << g) - 1) >> g);
      if (scatter_lds_bytes(nG, kTile) <= kDynMax && hist_groups_lds_bytes(nG) <= kDynMax &&
          regroup_lds_bytes(g, nwin) <= kDynMax)
        gshift = g;
    }
  const bool big = gshift != 0;
  if (big) fits = true;
  if (!fits) {
    // a table or a minibatch beyond the limits in this file's header: said once, loudly — the
    // general build is 3x slower and its user should know which path the numbers come from
    if (T.nbase > 0 && NNZ > 0 && key_build_mode() != 1) {
      static bool told = false;
      if (!told) {
        told = true;
        fprintf(stderr,
                "xflow_amd: the range-partitioned key build does not apply (%llu settled keys, "
                "%u nonzeros, %u row windows: beyond 5e8 keys / 3.3e7 nonzeros / 4e6 cells per "
                "GPU, xf_keybuild.hip): minibatches are built by the general path (a probe of the "
                "table per nonzero + a radix pass)\n",
                (unsigned long long)T.nbase, NNZ, nwin);
      }
    }
    return general_build(out, t, d_keys, d_rowptr, d_rowid, R, NNZ, ksc, w_fixed, s);
  }
  const uint32_t cA = (uint32_t)cA64, nS = (uint32_t)nS64;
  const uint32_t nG = big ? (nS + (1u << gshift) - 1) >> gshift : 0u;
  const uint32_t nR = big ? nG : nS;  // the ranges the histogram / scatter partition by
  // segment A: the settled tier's rows [0, nbase)
  xf_cells *c = nullptr;
  XF_TRY(cells_alloc(&c, R, NNZ, (uint32_t)T.nbase, kCellsTableRows, ksc, w_fixed, 0));
  struct Guard {
    xf_cells *c;
    ~Guard() {
      if (c) cells_free(c);
    }
  } guard{c};
  XF_REQUIRE(c->nchunk == cA && c->nwin == nwin, "cells_build_keyed: geometry");
  std::unique_ptr<KbDeferred> D(new KbDeferred);
  Scratch &sc = D->sc;  // (outlives the miss path: the list lives in it)
  KbArgs &a = D->a;
  D->c = c;
  D->t = t;
  D->sum = sum;
  D->ksc = ksc;
  D->R = R;
  D->NNZ = NNZ;
  D->nbase = T.nbase;
  {
    a.keys = d_keys;
    a.rowptr = d_rowid ? nullptr : d_rowptr;
    a.rowid = d_rowid;
    a.R = R;
    a.NNZ = NNZ;
    a.W = c->W;
    a.nwin = nwin;
    a.cA = cA;
    a.nS = nS;
    a.tile = scatter_lds_bytes(nR, kTile) <= kDynMax ? kTile : kTile / 2;
    a.ntile = (NNZ + a.tile - 1) / a.tile;
    a.nbase = (uint32_t)T.nbase;
    a.bkeys = T.bkeys;
    a.lo = T.lo;
    KbRanges gr{};
    XF_TRY(kb_index(t, T, cA, nS, &a, s, gshift, big ? &gr : nullptr));
    a.cellptr = c->cellptr;
    a.entries = c->entries;
    a.plan = c->plan;
    a.blk_cell = c->blk_cell;
    const size_t ncell = (size_t)nwin * cA;
    const unsigned max_items = nS + NNZ / kPart + 1;  // >= sum over S of ceil(n_S / kPart)
    const unsigned max_gitems = nG + NNZ / kPart + 1;
    // as many histogram / scatter workgroups as the GPU has CUs (one round), whole tiles each
    const uint32_t sub = std::max<uint32_t>(1, (a.ntile + 255) / 256);
    XF_REQUIRE(sub <= kMaxSub, "cells_build_keyed: %u nonzeros in one minibatch", NNZ);
    a.span = sub * a.tile;
    a.nW = (a.ntile + sub - 1) / sub;
    uint32_t *small = nullptr;
    const size_t n_zero = 4 + ncell;  // the summary and the histogram: cleared together
    a.npc = (uint32_t)((ncell + kScanPiece - 1) / kScanPiece);
    const size_t n_small = n_zero + ncell + (size_t)nS * 2 + 1 + max_items + 1 +
                           (size_t)a.nW * nR + a.ntile + 1 +
                           (big ? (size_t)nG * 2 + 1 + max_gitems + 1 : 0) + a.npc;
    XF_TRY(sc.get(&small, n_small));
    a.psum = small + (n_small - a.npc);
    a.sum = (KbSummary *)small;
    a.hist = small + 4;
    a.cellcur = a.hist + ncell;
    a.scount = a.cellcur + ncell;
    a.sstart = a.scount + nS;
    a.items = a.sstart + nS + 1;
    a.nitems = a.items + max_items;
    a.wgcnt = a.nitems + 1;
    a.tile_r0 = a.wgcnt + (size_t)a.nW * nR;
    uint32_t *gcount = a.tile_r0 + a.ntile + 1, *gstart = gcount + nG,
             *gitems = gstart + nG + 1, *gnitems = gitems + max_gitems;
#ifdef XF_EXPERIMENTS  // (tools/kb_knobs.py, tools/kb_timeline.py)
    a.flags = exp_knob() >= 100 && exp_knob() < 200 ? (uint32_t)(exp_knob() - 100) : 0u;
    if (exp_knob() == 200 && !big) {
      const size_t need = ((size_t)2 * a.nW + max_items) * kDbgSlots;
      if (need > g_dbg_n) {
        if (g_dbg) (void)hipFree(g_dbg);
        g_dbg = nullptr;
        XF_HIP(hipMalloc((void **)&g_dbg, need * 8));
        g_dbg_n = need;
      }
      XF_HIP(hipMemsetAsync(g_dbg, 0, need * 8, s));
      a.dbg = g_dbg;
      g_dbg_used = need;
      g_dbg_shape[0] = a.nW;
      g_dbg_shape[1] = a.nW;
      g_dbg_shape[2] = max_items;
    }
#endif
    XF_TRY(sc.get(&a.rec, NNZ));
    XF_TRY(sc.get(&a.missK, NNZ));
    XF_TRY(sc.get(&a.missR, NNZ));
    XF_HIP(hipMemsetAsync(small, 0, n_zero * 4, s));
#define XF_KB_LAUNCH(kern, grid, lds, args)                                                    \
  do {                                                                                         \
    static bool attr_done = false;                                                             \
    if (!attr_done) {                                                                          \
      XF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)kDynMax));                                               \
      attr_done = true;                                                                        \
    }                                                                                          \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kKb), lds, s, args);                             \
  } while (0)
    if (!big) {
      const bool ldsb = hist_lds_bytes(cA, nS, true) <= kDynMax;
      const size_t hl = hist_lds_bytes(cA, nS, ldsb);
      if (d_rowid) {
        if (ldsb) XF_KB_LAUNCH((k_kb_hist<true, true>), a.nW, hl, a);
        else
          XF_KB_LAUNCH((k_kb_hist<true, false>), a.nW, hl, a);
      } else {
        if (ldsb) XF_KB_LAUNCH((k_kb_hist<false, true>), a.nW, hl, a);
        else
          XF_KB_LAUNCH((k_kb_hist<false, false>), a.nW, hl, a);
      }
      if (a.npc > 1)
    hipLaunchKernelGGL(k_kb_psum, dim3(a.npc + a.cA / kKb + 1), dim3(kKb), 0, s, a);
      hipLaunchKernelGGL(k_kb_scan, dim3(a.npc + kPlanWgs + (nS + kKb / 64 - 1) / (kKb / 64)), dim3(kKb),
                         0, s, a);
      const size_t sl = scatter_lds_bytes(nS, a.tile);
      if (a.tile == kTile) {
        if (d_rowid) XF_KB_LAUNCH((k_kb_scatter<true, kTile>), a.nW, sl, a);
        else
          XF_KB_LAUNCH((k_kb_scatter<false, kTile>), a.nW, sl, a);
      } else {
        if (d_rowid) XF_KB_LAUNCH((k_kb_scatter<true, kTile / 2>), a.nW, sl, a);
        else
          XF_KB_LAUNCH((k_kb_scatter<false, kTile / 2>), a.nW, sl, a);
      }
    } else {
      // level one over the groups: the same scan and scatter kernels see the groups as their
      // ranges (ag); level two splits every group's records by super-chunk into a.rec
      Rec3 *grec = nullptr;
      XF_TRY(sc.get(&grec, NNZ));
      XF_HIP(hipMemsetAsync(a.nitems, 0, 4, s));
      KbArgs ag = a;
      ag.nS = nG;
      ag.sc = gr;
      ag.scount = gcount;
      ag.sstart = gstart;
      ag.items = gitems;
      ag.nitems = gnitems;
      ag.rec = grec;
      ag.scan_part = 1;
      const size_t hl = hist_groups_lds_bytes(nG);
      if (d_rowid) XF_KB_LAUNCH((k_kb_hist_groups<true>), a.nW, hl, ag);
      else
        XF_KB_LAUNCH((k_kb_hist_groups<false>), a.nW, hl, ag);
      hipLaunchKernelGGL(k_kb_scan, dim3(a.npc + kPlanWgs + (nG + kKb / 64 - 1) / (kKb / 64)), dim3(kKb),
                         0, s, ag);
      const size_t sl = scatter_lds_bytes(nG, a.tile);
      if (a.tile == kTile) {
        if (d_rowid) XF_KB_LAUNCH((k_kb_scatter<true, kTile>), a.nW, sl, ag);
        else
          XF_KB_LAUNCH((k_kb_scatter<false, kTile>), a.nW, sl, ag);
      } else {
        if (d_rowid) XF_KB_LAUNCH((k_kb_scatter<true, kTile / 2>), a.nW, sl, ag);
        else
          XF_KB_LAUNCH((k_kb_scatter<false, kTile / 2>), a.nW, sl, ag);
      }
      a.gshift = gshift;
      a.nG = nG;
      a.gstart = gstart;
      a.grec = grec;
      XF_KB_LAUNCH(k_kb_regroup, nG, regroup_lds_bytes(gshift, nwin), a);
      a.scan_part = 2;
      if (a.npc > 1)
    hipLaunchKernelGGL(k_kb_psum, dim3(a.npc + a.cA / kKb + 1), dim3(kKb), 0, s, a);
      hipLaunchKernelGGL(k_kb_scan, dim3(a.npc + kPlanWgs), dim3(kKb), 0, s, a);
    }
    hipLaunchKernelGGL(k_kb_resolve, dim3(max_items), dim3(kRes), 0, s, a);
#undef XF_KB_LAUNCH
    XF_HIP(hipMemcpyAsync(sum, a.sum, sizeof(KbSummary), hipMemcpyDeviceToHost, s));
    XF_HIP(hipGetLastError());
  }
  if (defer && hipEventCreateWithFlags(&D->ev, hipEventDisableTiming) == hipSuccess) {
    // the wait is the caller's (cells_build_keyed_finish)
    XF_HIP(hipEventRecord(D->ev, s));
    guard.c = nullptr;
    *out = c;
    *defer = D.release();
    return XF_OK;
  }
  XF_HIP(hipStreamSynchronize(s));  // the one synchronisation of the steady state
  XF_TRY(keyed_tail(*D, s, nullptr));
  guard.c = nullptr;
  *out = c;
  return XF_OK;
}


// The FM key build (kernels above).  The arrays whose size is the number of distinct keys are
// written where `place` says (the caller's allocation, sized for the bound `place` is called
// with: no copies afterwards, no wait for the number).  Nothing is waited for: *d_U and *d_miss
// (device, in `sc`) are the number of distinct keys and of keys the tier does not hold — with
// misses the arrays are to be discarded.  *ok = false: the fast path does not apply (no settled
// tier, a minibatch of more than 16 row windows, a table too large for the LDS tables, or a key
// the tier does not hold) — the caller takes the sort-based build.
int fm_build_keyed(xf_table *t, const uint64_t *d_keys, const uint32_t *d_rowptr, uint32_t R,
                   uint32_t NNZ, Scratch &sc, hipStream_t s, bool *ok, const uint32_t **d_U,
                   const unsigned long long **d_miss, uint32_t *ridx /* [NNZ], the caller's */,
                   FmKeyedOut place, void *ctx) {
  *ok = false;
  const TableDev T = table_dev(t);
  const uint64_t cA64 = (T.nbase + kChunk - 1) / kChunk;
  const uint64_t nS64 = (cA64 + kSC - 1) / kSC;
  const uint32_t nwin = std::max<uint32_t>(1, (R + kWinMax - 1) / kWinMax);
  KbSummary *sum = summary_buf();
  const bool fits = T.nbase > 0 && T.nbase < 0xFFFF0000ull && NNZ > 0 && NNZ < (1u << 30) &&
                    R > 0 && sum != nullptr && cA64 < 0xFFFFu && nwin <= 16 &&
                    scatter_lds_bytes((uint32_t)nS64, kTile / 2) <= kDynMax &&
                    hist_lds_bytes((uint32_t)cA64, (uint32_t)nS64, false) <= kDynMax &&
                    ((uint64_t)NNZ + kTile / 2 - 1) / (kTile / 2) <= (uint64_t)kMaxSub * 256;
  if (!fits) return XF_OK;
  const uint32_t cA = (uint32_t)cA64, nS = (uint32_t)nS64;
  KbArgs a{};
  a.keys = d_keys;
  a.rowptr = d_rowptr;
  a.R = R;
  a.NNZ = NNZ;
  a.W = std::max<uint32_t>(1, (R + nwin - 1) / nwin);  // (cells_alloc's window)
  a.nwin = nwin;
  a.cA = cA;
  a.nS = nS;
  a.tile = scatter_lds_bytes(nS, kTile) <= kDynMax ? kTile : kTile / 2;
  a.ntile = (NNZ + a.tile - 1) / a.tile;
  a.nbase = (uint32_t)T.nbase;
  a.bkeys = T.bkeys;
  a.lo = T.lo;
  XF_TRY(kb_index(t, T, cA, nS, &a, s));
  const size_t ncell = (size_t)nwin * cA;
  const unsigned max_items = nS + NNZ / kPart + 1;
  const uint32_t sub = std::max<uint32_t>(1, (a.ntile + 255) / 256);
  if (sub > kMaxSub) return XF_OK;
  a.span = sub * a.tile;
  a.nW = (a.ntile + sub - 1) / sub;
  uint32_t *small = nullptr;
  const size_t n_zero = 4 + ncell;
  a.npc = (uint32_t)((ncell + kScanPiece - 1) / kScanPiece);
  const size_t n_small = n_zero + ncell + (ncell + 1) + 4 * ((size_t)cA + 1) + (size_t)nS * 2 + 1 +
                         max_items + 1 + (size_t)a.nW * nS + a.ntile + 1 + ((size_t)nS + 2) + a.npc;
  XF_TRY(sc.get(&small, n_small));
  a.psum = small + (n_small - a.npc);
  a.sum = (KbSummary *)small;
  a.hist = small + 4;
  a.cellcur = a.hist + ncell;
  a.cellptr = a.cellcur + ncell;
  a.plan = a.cellptr + ncell + 1;
  a.scount = a.plan + 4 * ((size_t)cA + 1);
  a.sstart = a.scount + nS;
  a.items = a.sstart + nS + 1;
  a.nitems = a.items + max_items;
  a.wgcnt = a.nitems + 1;
  a.tile_r0 = a.wgcnt + (size_t)a.nW * nS;
  uint32_t *ucount = a.tile_r0 + a.ntile + 1;
  XF_TRY(sc.get(&a.rec4, NNZ));
  XF_TRY(sc.get(&a.fm_vrow, NNZ));
  XF_TRY(sc.get(&a.fm_rp, NNZ));
  a.fm_ridx = ridx;
  XF_HIP(hipMemsetAsync(small, 0, n_zero * 4, s));
  const bool ldsb = hist_lds_bytes(cA, nS, true) <= kDynMax;
  const size_t hl = hist_lds_bytes(cA, nS, ldsb);
#define XF_KB_LAUNCH(kern, grid, lds)                                                          \
  do {                                                                                         \
    static bool attr_done = false;                                                             \
    if (!attr_done) {                                                                          \
      XF_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)kDynMax));                                               \
      attr_done = true;                                                                        \
    }                                                                                          \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kKb), lds, s, a);                                \
  } while (0)
  if (ldsb) XF_KB_LAUNCH((k_kb_hist<false, true>), a.nW, hl);
  else
    XF_KB_LAUNCH((k_kb_hist<false, false>), a.nW, hl);
  if (a.npc > 1)
    hipLaunchKernelGGL(k_kb_psum, dim3(a.npc + a.cA / kKb + 1), dim3(kKb), 0, s, a);
  hipLaunchKernelGGL(k_kb_scan, dim3(a.npc + kPlanWgs + (nS + kKb / 64 - 1) / (kKb / 64)), dim3(kKb), 0, s,
                     a);
  const size_t sl = scatter_lds_bytes(nS, a.tile);
  if (a.tile == kTile) XF_KB_LAUNCH((k_kb_scatter<false, kTile, true>), a.nW, sl);
  else
    XF_KB_LAUNCH((k_kb_scatter<false, kTile / 2, true>), a.nW, sl);
#undef XF_KB_LAUNCH
  hipLaunchKernelGGL(k_kb_resolve_fm, dim3(max_items), dim3(kRes), 0, s, a);
  FmRegroup g{};
  g.vrow = a.fm_vrow;
  g.rp = a.fm_rp;
  g.sstart = a.sstart;
  g.bkeys = T.bkeys;
  g.nS = nS;
  g.W = a.W;
  g.nbase = T.nbase;
  g.ucount = ucount;
  g.items = a.items;
  g.nitems = a.nitems;
  XF_TRY(sc.get(&g.pcnt, (size_t)max_items * kSCKeys));
  XF_TRY(sc.get(&g.gbits, (size_t)nS * (kSCKeys / 32) + nS));
  g.done = g.gbits + (size_t)nS * (kSCKeys / 32);
  XF_HIP(hipMemsetAsync(g.gbits, 0, ((size_t)nS * (kSCKeys / 32) + nS) * 4, s));
  hipLaunchKernelGGL(k_fm_count, dim3(max_items), dim3(kKb), 0, s, g);
  hipLaunchKernelGGL(k_fm_scan, dim3(1), dim3(kKb), 0, s, ucount, nS);
  // no wait for the number of distinct keys: the arrays are placed for the most there can be
  // (every nonzero its own key, every settled key touched), the regroup takes the number from
  // the device, and the caller reads it — and the misses — when it next synchronises
  XF_TRY(place(ctx, (uint32_t)std::min<uint64_t>(NNZ, T.nbase), &g.ukeys, &g.urow, &g.segptr,
               &g.coo));
  hipLaunchKernelGGL(k_fm_regroup, dim3(max_items), dim3(kKb), 0, s, g);
  XF_HIP(hipGetLastError());
  *d_U = ucount + nS;
  *d_miss = &a.sum->miss;
  *ok = true;
  return XF_OK;
}


// what a caller that goes on with the sorted list takes along (valid while its Scratch lives)
struct SortExtra {
  // in: the nonzeros come in CSR order and a record's payload is its ROW (window << kRinBits |
  // row in window, W rows per window) instead of its position
  const uint32_t *rowptr = nullptr;
  uint32_t R = 0, W = 0;
  bool count_keys = false;  // in: the distinct keys counted as well (the same wait)
  // out (device unless said otherwise)
  uint32_t U = 0;           // (host) distinct keys
  uint32_t *bsum = nullptr; // distinct keys before every block of kWlBlock sorted records
  const uint32_t *sstart = nullptr, *items = nullptr, *nitems = nullptr;
  uint32_t nR = 0, max_items = 0;
};

// (key, payload) of d_keys[0..n) in (key, payload) order — the payload the nonzero's position:
// what a stable sort of the keys with their indices gives (kernels: "(key, position) in key
// order" above).  [lo, lo + span]: where the keys lie (a shard's key range; 0 and UINT64_MAX for
// any key) — keys outside it are sorted as well, only slower.  *done = false: not sorted (more
// nonzeros than the partition takes, xf_tune key_build = 1): the caller sorts some other way.
// Waits for the stream; the scratch is the caller's.
static int sort_key_pos_sc(Scratch &sc, const uint64_t *d_keys, uint32_t n, uint64_t lo,
                           uint64_t span, uint64_t *sk, uint32_t *spos, hipStream_t s, bool *done,
                           SortExtra *ex, uint32_t site) {
  *done = false;
  KbSummary *sum = summary_buf();
  if (n == 0 || n >= (1u << 30) || !sum || key_build_mode() == 1 ||
      ((uint64_t)n + kTile / 2 - 1) / (kTile / 2) > (uint64_t)kMaxSub * 256)
    return XF_OK;
  const bool csr = ex && ex->rowptr;
  // Records per range: as many as leave a range's Poisson spread (8 sigma) below the 4096 of the
  // 68 KB variant — fewer, longer runs out of the partition (1e7 keys: 0.255 ms at 3000 a range,
  // 0.249 at 3400, 0.243 at 3700; 2400 leaves no room for the hot keys' ranges); a hot key's
  // thousand or two on top stay below kSpCap.
  constexpr uint32_t per_range = 3600;
  // Room for the hot keys' own ranges (kernels: "hot keys get ranges of their own"), looked for
  // when the stream has shown itself skewed: the call site's last sort on this thread met a range
  // beyond a range's LDS, or found a hot key (the sample costs ~35 us a sort).  A state per call
  // site: a worker's minibatches and an owner's merged key lists — unique keys — are different
  // streams.
  static thread_local bool skewed_at[kSortSites] = {};
  bool &skewed = skewed_at[site < kSortSites ? site : 0];
  const uint32_t hot2 =
      skewed && n >= 65536 && (n + per_range - 1) / per_range + 2 * kHotMax <= kArMaxRanges ? 2 * kHotMax : 0;
  const uint32_t nR0 = std::min<uint32_t>(kArMaxRanges - hot2, (n + per_range - 1) / per_range);
  const uint32_t nR = nR0 + hot2;
  KbArgs a{};
  a.keys = d_keys;
  if (csr) {
    a.rowptr = ex->rowptr;
    a.R = ex->R;
    a.W = ex->W;
    a.nwin = std::max<uint32_t>(1, (ex->R + ex->W - 1) / ex->W);
  } else {
    // (rowid = null with ROWID kernels: the record's "row" is its position; window << kRinBits | row
    // in window with 2^kRinBits rows per window is the position itself)
    a.rowid = nullptr;
    a.W = 1u << kRinBits;
    a.R = n;
    a.nwin = (n + a.W - 1) / a.W;
  }
  a.NNZ = n;
  a.nS = nR;
  a.tile = scatter_lds_bytes(nR, kTile) <= kDynMax ? kTile : kTile / 2;
  a.ntile = (n + a.tile - 1) / a.tile;
  a.lo = lo;
  const uint32_t sub = std::max<uint32_t>(1, (a.ntile + 255) / 256);
  a.span = sub * a.tile;
  a.nW = (a.ntile + sub - 1) / sub;
  a.npc = 0;  // (the scan's per-range part alone)
  const unsigned max_items = nR + n / kPart + 1;
  uint32_t *part1 = nullptr;
  XF_TRY(sc.get(&part1, (size_t)nR * 2 + 1 + max_items + 1 + (size_t)a.nW * nR + a.ntile + 1 + 8));
  a.scount = part1;
  a.sstart = a.scount + nR;
  a.items = a.sstart + nR + 1;
  a.nitems = a.items + max_items;
  a.wgcnt = a.nitems + 1;
  a.tile_r0 = a.wgcnt + (size_t)a.nW * nR;
  unsigned int *d_heavy = (unsigned int *)(a.tile_r0 + a.ntile + 1);
  uint64_t *bnd = nullptr;
  uint16_t *dir = nullptr;
  SpArgs p{};
  XF_TRY(sc.get(&bnd, nR));
  XF_TRY(sc.get(&dir, nR + 1));
  XF_TRY(sc.get(&p.hv, nR));
  XF_TRY(sc.get(&p.hv2, nR));
  XF_TRY(sc.get(&a.rec, n));
  WlArgs wl{};
  if (ex && ex->count_keys) {
    wl.sk = sk;
    wl.n = n;
    wl.nb = (n + kWlBlock - 1) / kWlBlock;
    XF_TRY(sc.get(&wl.bsum, (size_t)wl.nb + 1));
  }
  a.sc.bnd = bnd;
  a.sc.dir = dir;
  a.sc.n = nR;
  a.sc.mult = (uint32_t)std::min<uint64_t>(((uint64_t)nR << 32) / ((span >> 32) + 1), 0xFFFFFFFFull);
  XF_HIP(hipMemsetAsync(d_heavy, 0, 24, s));
  if (hot2) {
    Rec3 *samp = nullptr;
    uint64_t *ssk = nullptr, *bnd0 = nullptr;
    uint32_t *ssp = nullptr, *sst = nullptr;
    XF_TRY(sc.get(&samp, kHotSample));
    XF_TRY(sc.get(&ssk, kHotSample));
    XF_TRY(sc.get(&ssp, kHotSample));
    XF_TRY(sc.get(&sst, 2));
    XF_TRY(sc.get(&bnd0, 1));
    hipLaunchKernelGGL(k_hot_gather, dim3(kHotSample / 256), dim3(256), 0, s, d_keys, n, kHotSample,
                       samp, sst, bnd0);
    SpArgs q{};
    q.rec = samp;
    q.sstart = sst;
    q.bnd = bnd0;
    q.nR = 1;
    q.last = ~0ull;
    q.sk = ssk;
    q.spos = ssp;
    q.heavy = d_heavy;
    q.hv = p.hv;
    XF_KB_LAUNCH_N(k_sp_sort<false>, 1, kSp, kSpLds, q);
    {
      static bool attr_done = false;
      if (!attr_done) {
        XF_HIP(hipFuncSetAttribute((const void *)k_hot_ranges,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDynMax));
        attr_done = true;
      }
    }
    hipLaunchKernelGGL(k_hot_ranges, dim3(1), dim3(kKb), ((size_t)nR + kHotMax + kHotSample) * 8, s,
                       ssk, kHotSample, lo, span, nR0, nR, a.sc.mult, bnd, dir, d_heavy + 3);
  } else {
    hipLaunchKernelGGL(k_ar_ranges, dim3((nR + 256) / 256), dim3(256), 0, s, lo, nR, a.sc.mult, bnd, dir);
  }
  a.scan_part = 1;
  if (csr) XF_KB_LAUNCH_N((k_kb_hist_groups<false>), a.nW, kKb, hist_groups_lds_bytes(nR), a);
  else
    XF_KB_LAUNCH_N((k_kb_hist_groups<true>), a.nW, kKb, hist_groups_lds_bytes(nR), a);
  hipLaunchKernelGGL(k_kb_scan, dim3(kPlanWgs + (nR + kKb / 64 - 1) / (kKb / 64)), dim3(kKb), 0, s, a);
  const size_t sl = scatter_lds_bytes(nR, a.tile);
  if (a.tile == kTile) {
    if (csr) XF_KB_LAUNCH_N((k_kb_scatter<false, kTile>), a.nW, kKb, sl, a);
    else
      XF_KB_LAUNCH_N((k_kb_scatter<true, kTile>), a.nW, kKb, sl, a);
  } else {
    if (csr) XF_KB_LAUNCH_N((k_kb_scatter<false, kTile / 2>), a.nW, kKb, sl, a);
    else
      XF_KB_LAUNCH_N((k_kb_scatter<true, kTile / 2>), a.nW, kKb, sl, a);
  }
  p.rec = a.rec;
  p.sstart = a.sstart;
  p.bnd = bnd;
  p.nR = nR;
  p.last = lo + span < lo ? ~0ull : lo + span;
  p.sk = sk;
  p.spos = spos;
  p.heavy = d_heavy;
  p.rows = csr ? 1u : 0u;
  p.tbits = csr ? 0u : (a.tile == kTile ? 13u : 12u);
  static_assert(kTile == 8192, "tbits above");
  XF_KB_LAUNCH_N(k_sp_sort<true>, nR, kSp, kSpLdsSmall, p);
  XF_KB_LAUNCH_N(k_sp_sort<false>, nR, kSp, kSpLds, p);
  if (csr && hot2) hipLaunchKernelGGL(k_sp_copy, dim3(max_items), dim3(kSp), 0, s, p, a.items, a.nitems);
  // (the distinct keys of a list without heavy ranges — the usual one — counted before the wait)
  auto count_keys = [&]() {
    hipLaunchKernelGGL(k_wl_count, dim3(wl.nb), dim3(kKb), 0, s, wl);
    hipLaunchKernelGGL(k_wl_scan, dim3(1), dim3(kKb), 0, s, wl);
  };
  // (pinned: the heavy ranges' three counts, the hot keys found, the distinct keys)
  static thread_local unsigned int *h_heavy = nullptr;
  if (!h_heavy) XF_HIP(hipHostMalloc((void **)&h_heavy, 32));
  constexpr int kU = 6;
  if (wl.bsum) {
    count_keys();
    XF_HIP(hipMemcpyAsync(h_heavy + kU, wl.bsum + wl.nb, 4, hipMemcpyDeviceToHost, s));
  }
  XF_HIP(hipMemcpyAsync(h_heavy, d_heavy, 24, hipMemcpyDeviceToHost, s));
  XF_HIP(hipGetLastError());
  XF_HIP(hipStreamSynchronize(s));
  skewed = h_heavy[0] != 0 || (hot2 && h_heavy[3] != 0);
  if (h_heavy[4]) {  // the hot keys' own ranges, tile group by tile group
    XF_KB_LAUNCH_N(k_sp_tiles, h_heavy[5], kSp, kSpLds, p);
    if (!h_heavy[0]) {
      if (wl.bsum) {
        count_keys();
        XF_HIP(hipMemcpyAsync(h_heavy + kU, wl.bsum + wl.nb, 4, hipMemcpyDeviceToHost, s));
      }
      XF_HIP(hipGetLastError());
      XF_HIP(hipStreamSynchronize(s));
    }
  }
  if (h_heavy[0]) {  // the merge sort of the ranges beyond a range's LDS
    const uint32_t nparts = h_heavy[1], longest = (h_heavy[2] + kSpCap - 1) / kSpCap;
    uint64_t *tk = nullptr;
    uint32_t *tp = nullptr;
    XF_TRY(sc.get(&tk, n));
    XF_TRY(sc.get(&tp, n));
    XF_KB_LAUNCH_N(k_sp_parts, nparts, kSp, kSpLds, p);
    bool in_tmp = false;
    const uint32_t nmerge = nparts * (kSpCap / kSpTile);
    for (uint32_t pass = 0; (1u << pass) < longest; ++pass) {
      p.pass = pass;
      p.mk = in_tmp ? tk : sk;
      p.mp = in_tmp ? tp : spos;
      p.ok = in_tmp ? sk : tk;
      p.op = in_tmp ? spos : tp;
      hipLaunchKernelGGL(k_sp_merge, dim3(nmerge), dim3(kSpM), 0, s, p);
      in_tmp = !in_tmp;
    }
    // (a range of 2^odd parts, or fewer, ended in the other buffer)
    p.pass = 0xFFFFFFFFu;
    p.mk = tk;
    p.mp = tp;
    p.ok = sk;
    p.op = spos;
    hipLaunchKernelGGL(k_sp_merge, dim3(nmerge), dim3(kSpM), 0, s, p);
    if (wl.bsum) {
      count_keys();
      XF_HIP(hipMemcpyAsync(h_heavy + kU, wl.bsum + wl.nb, 4, hipMemcpyDeviceToHost, s));
    }
    XF_HIP(hipGetLastError());
    XF_HIP(hipStreamSynchronize(s));
  }
  if (ex) {
    ex->U = wl.bsum ? h_heavy[kU] : 0;
    ex->bsum = wl.bsum;
    ex->sstart = a.sstart;
    ex->items = a.items;
    ex->nitems = a.nitems;
    ex->nR = nR;
    ex->max_items = max_items;
  }
  *done = true;
  return XF_OK;
}

int sort_key_pos(const uint64_t *d_keys, uint32_t n, uint64_t lo, uint64_t span, uint64_t *sk,
                 uint32_t *spos, hipStream_t s, bool *done, uint32_t site) {
  Scratch sc;
  return sort_key_pos_sc(sc, d_keys, n, lo, span, sk, spos, s, done, nullptr, site);
}

// The worker side of the weight / gradient exchange, LR (kernels: "the worker side of the
// exchange" above): *out = a minibatch with its sorted unique keys, row offsets and labels on the
// device (no CSR index of the unique keys, no key-grouped or panel-major view: the LR kernels of
// that path stream the cells), *cells = its cells over the unique-key index.  *done = false:
// beyond the sort's limits, nothing was built.  Waits for the stream.
int batch_compile_lr_dev(xf_batch **out, xf_cells **cells, const uint64_t *d_keys,
                         const uint32_t *d_rowptr, const int32_t *d_labels, uint32_t R,
                         uint32_t NNZ, bool key_sorted_copy, hipStream_t s, bool *done) {
  *done = false;
  if (NNZ == 0 || R == 0) return XF_OK;
  Scratch sc;
  uint64_t *sk = nullptr;
  uint32_t *srp = nullptr, *su = nullptr;
  XF_TRY(sc.get(&sk, NNZ));
  XF_TRY(sc.get(&srp, NNZ));
  XF_TRY(sc.get(&su, NNZ));
  SortExtra ex;
  ex.rowptr = d_rowptr;
  ex.R = R;
  {  // (cells_alloc's windows)
    const uint32_t nwin = std::max<uint32_t>(1, (R + kWinMax - 1) / kWinMax);
    ex.W = std::max<uint32_t>(1, (R + nwin - 1) / nwin);
  }
  ex.count_keys = true;
  XF_TRY(sort_key_pos_sc(sc, d_keys, NNZ, 0, ~0ull, sk, srp, s, done, &ex, kSortSiteLr));
  if (!*done) return XF_OK;
  *done = false;
  const uint32_t U = ex.U;
  xf_cells *c = nullptr;
  XF_TRY(cells_alloc(&c, R, NNZ, U, kCellsUidx, key_sorted_copy, 0, 0));
  struct Guard {
    xf_cells *c;
    xf_batch *b;
    ~Guard() {
      if (c) cells_free(c);
      if (b) xf_batch_free(b);
    }
  } guard{c, nullptr};
  if (c->W != ex.W) return XF_OK;  // (never: cells_alloc's windows are the ones above)
  xf_batch *b = new xf_batch;
  guard.b = b;
  b->R = R;
  b->NNZ = NNZ;
  b->U = U;
  b->on_device_only = true;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_ukeys = 0, o_rowptr = o_ukeys + al((size_t)U * 8);
  const size_t o_labels = o_rowptr + al(((size_t)R + 1) * 4);
  char *d = nullptr;
  XF_TRY(blob_alloc((void **)&d, o_labels + al((size_t)R * 4) + 256, &b->d_blob_bytes));
  b->d_blob = d;
  xf_dev_batch &v = b->view;
  v.R = R;
  v.NNZ = NNZ;
  v.U = U;
  v.ukeys = (const uint64_t *)(d + o_ukeys);
  v.rowptr = (const uint32_t *)(d + o_rowptr);
  v.labels = (const int32_t *)(d + o_labels);
  XF_HIP(hipMemcpyAsync(d + o_rowptr, d_rowptr, ((size_t)R + 1) * 4, hipMemcpyDeviceToDevice, s));
  XF_HIP(hipMemcpyAsync(d + o_labels, d_labels, (size_t)R * 4, hipMemcpyDeviceToDevice, s));
  WlArgs wl{};
  wl.sk = sk;
  wl.n = NNZ;
  wl.nb = (NNZ + kWlBlock - 1) / kWlBlock;
  wl.bsum = ex.bsum;
  wl.su = su;
  wl.ukeys = (uint64_t *)(d + o_ukeys);
  hipLaunchKernelGGL(k_wl_unique, dim3(wl.nb), dim3(kKb), 0, s, wl);
  // the cells (as the first-touch build's tail)
  const size_t ncell = (size_t)c->nwin * c->nchunk;
  KbArgs a{};
  a.NNZ = NNZ;
  a.nwin = c->nwin;
  a.cA = c->nchunk;
  a.npc = (uint32_t)((ncell + kScanPiece - 1) / kScanPiece);
  uint32_t *small = nullptr;
  const size_t n_zero = 4 + ncell;  // the summary and the cell histogram: cleared together
  XF_TRY(sc.get(&small, n_zero + ncell + a.npc));
  a.sum = (KbSummary *)small;
  a.hist = small + 4;
  a.cellcur = a.hist + ncell;
  a.psum = a.cellcur + ncell;
  a.cellptr = c->cellptr;
  a.entries = c->entries;
  a.plan = c->plan;
  a.blk_cell = c->blk_cell;
  XF_HIP(hipMemsetAsync(small, 0, n_zero * 4, s));
  WcArgs w{};
  w.su = su;
  w.srp = srp;
  w.sstart = ex.sstart;
  w.items = ex.items;
  w.nitems = ex.nitems;
  w.n = NNZ;
  w.max_items = ex.max_items;
  w.nwin = c->nwin;
  w.nchunk = c->nchunk;
  w.hist = a.hist;
  w.cellcur = a.cellcur;
  w.entries = c->entries;
  w.cellptr = c->cellptr;
  w.blk_cell = c->blk_cell;
  hipLaunchKernelGGL(k_wc_cells<false>, dim3(ex.max_items), dim3(kEb), 0, s, w);
  a.scan_part = 2;
  if (a.npc > 1) hipLaunchKernelGGL(k_kb_psum, dim3(a.npc + a.cA / kKb + 1), dim3(kKb), 0, s, a);
  hipLaunchKernelGGL(k_kb_scan, dim3(a.npc + kPlanWgs), dim3(kKb), 0, s, a);
  hipLaunchKernelGGL(k_wc_cells<true>, dim3(ex.max_items + std::min<uint32_t>(64, ex.nR / 8 + 1)),
                     dim3(kEb), 0, s, w);
  KbSummary *sum = summary_buf();
  XF_HIP(hipMemcpyAsync(sum, a.sum, sizeof(KbSummary), hipMemcpyDeviceToHost, s));
  XF_HIP(hipGetLastError());
  XF_HIP(hipStreamSynchronize(s));
  XF_TRY(cells_fill_items(c, sum->nitems, sum->nsplit, s));
  XF_TRY(cells_key_sorted_copy(c, s));
  XF_HIP(hipStreamSynchronize(s));  // (the scratch goes back)
  guard.c = nullptr;
  guard.b = nullptr;
  *out = b;
  *cells = c;
  *done = true;
  return XF_OK;
}

}  // namespace xf

// diagnostics (tools/kb_timeline.py): the phase timestamps of the last keyed build made with
// exp_knob = 200; shape[3] = workgroups of the three kernels, slots per workgroup returned
extern "C" int xf_kb_debug_read(unsigned long long *out, size_t cap, uint32_t *shape) {
  XF_REQUIRE(out && shape, "xf_kb_debug_read: null argument");
  XF_HIP(hipDeviceSynchronize());
  const size_t n = std::min(cap, xf::g_dbg_used);
  if (n) XF_HIP(hipMemcpy(out, xf::g_dbg, n * 8, hipMemcpyDeviceToHost));
  for (int i = 0; i < 3; ++i) shape[i] = xf::g_dbg_shape[i];
  return kDbgSlots;
}
