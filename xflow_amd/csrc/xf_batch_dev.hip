// xf_batch_dev.hip — the minibatch key build ON THE GPU (gfx950).
//
// Replaces the same reference code as xf_batch.cc (the key build at the top of
// LRWorker::update / FMWorker::update, src/model/lr/lr_worker.cc:146-166,
// src/model/fm/fm_worker.cc:205-225: flatten the slice to all_keys[(fid,sid)], std::sort by
// fid, sorted-unique key list) and produces the identical compiled batch — every array equals
// the host builder's (tests/test_gpu_parity.py::test_device_key_build_equals_host).
//
// Pipeline, all on one stream:
//   1. (key, position) pairs sorted by key           xf::sort_key_pos (round 6: uniform key ranges,
//                                                    a range sorted in LDS); rocPRIM's radix sort
//                                                    beyond 3.3e7 keys
//   2. segment heads -> unique index (scan: k_scan_*, by hand), ukeys / segptr / uidx / coo_row
//   3. heavy-key list and gradient tiles             flag + scan + scatter (xf_tiling.h rules)
//   4. panel-major forward view: cell counts (atomics), scan, stable in-row placement with
//      wave ballots; forward tiles by the same flag + scan + scatter
// The order is (key, position) — what a stable sort of the keys gives: inside a key the
// occurrences stay in row-major order, like the host builder's.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <algorithm>
#include <vector>

#include "xf_batch.h"
#include "xf_common.h"
#include "xf_device.h"
#include "xf_scratch.h"
#include "xf_tiling.h"

namespace xf {
double panel_slice_bytes();  // xf_batch.cc tuning knobs
double min_panel_nnz();
}  // namespace xf

namespace {

constexpr int kBlock = 256;
inline int grid_for(size_t n) {
  size_t g = (n + kBlock - 1) / kBlock;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}
#define XF_GRID_STRIDE(i, n)                                                     \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(n); \
       i += (size_t)gridDim.x * blockDim.x)

__global__ void k_iota(uint32_t *p, size_t n) { XF_GRID_STRIDE(i, n) p[i] = (uint32_t)i; }

__global__ void k_heads(const uint64_t *__restrict__ sk, size_t n, uint32_t *__restrict__ head) {
  XF_GRID_STRIDE(j, n) head[j] = (j == 0 || sk[j] != sk[j - 1]) ? 1u : 0u;
}

// uid1 = inclusive scan of head (1-based unique index)
__global__ void k_unique(const uint64_t *__restrict__ sk, const uint32_t *__restrict__ head,
                         const uint32_t *__restrict__ uid1, size_t n, uint32_t U,
                         uint64_t *__restrict__ ukeys, uint32_t *__restrict__ segptr) {
  XF_GRID_STRIDE(j, n) {
    if (head[j]) {
      ukeys[uid1[j] - 1] = sk[j];
      segptr[uid1[j] - 1] = (uint32_t)j;
    }
    if (j == 0) segptr[U] = (uint32_t)n;
  }
}

// row_of[pos] = row of the pos-th nonzero: one wave per row, coalesced writes
__global__ void k_row_of(const uint32_t *__restrict__ rowptr, uint32_t R,
                         uint32_t *__restrict__ row_of) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nw = gridDim.x * (kBlock / 64);
  for (uint32_t r = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; r < R; r += nw)
    for (uint32_t j = rowptr[r] + lane; j < rowptr[r + 1]; j += 64) row_of[j] = r;
}

__global__ void k_uidx_coo(const uint32_t *__restrict__ spos, const uint32_t *__restrict__ uid1,
                           const uint32_t *__restrict__ row_of, size_t n,
                           uint32_t *__restrict__ uidx, uint32_t *__restrict__ coo_row) {
  XF_GRID_STRIDE(j, n) {
    const uint32_t pos = spos[j];
    uidx[pos] = uid1[j] - 1;
    coo_row[j] = row_of[pos];
  }
}

// The heavy keys and the keys at which a gradient tile starts (xf_tiling.h), two ascending
// lists, without arrays of U flags and their U-element scans: a tile starts every ~192 occurrences and
// heavy keys are rare, so the lists are counted per block of kFlagBlk keys, the ~U / 4096 block
// counts scanned by one workgroup, and the lists written block by block (a thread owns 16
// neighbouring keys).  25 us instead of 115 at 6.3*10^6 keys.
constexpr uint32_t kFlagPer = 16, kFlagBlk = kBlock * kFlagPer;

// (U: from the device when the host does not know it yet; the launch is sized for a bound)
__device__ __forceinline__ uint32_t keys_of(uint32_t U, const uint32_t *__restrict__ d_U) {
  return d_U ? *d_U : U;
}
__device__ __forceinline__ void key_flag_bits(const uint32_t *__restrict__ segptr, uint32_t U,
                                              uint32_t u0, uint32_t &hbits, uint32_t &tbits) {
  hbits = tbits = 0;
#pragma unroll
  for (uint32_t i = 0; i < kFlagPer; ++i) {
    const uint32_t u = u0 + i;
    if (u >= U) break;
    if (segptr[u + 1] - segptr[u] > XF_HEAVY_SEG) hbits |= 1u << i;
    if (xf::grad_tile_starts_at(segptr, u)) tbits |= 1u << i;
  }
}

// sums over the workgroup; every thread gets the exclusive prefix of its own value
__device__ __forceinline__ uint32_t block_prefix(uint32_t x, uint32_t *wsum, uint32_t *total) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t inc = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o);
    if ((int)lane >= o) inc += t;
  }
  __syncthreads();  // (wsum may still be read from the previous call)
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t w = 0; w < kBlock / 64; ++w) {
    if (w < wave) base += wsum[w];
    tot += wsum[w];
  }
  *total = tot;
  return base + inc - x;
}

__global__ void __launch_bounds__(kBlock)
k_flag_counts(const uint32_t *__restrict__ segptr, uint32_t Ub, const uint32_t *__restrict__ d_U,
              uint32_t nb,
              uint32_t *__restrict__ bc /* [2 * nb]: heavy keys, tile starts per block */) {
  __shared__ uint32_t wsum[kBlock / 64];
  const uint32_t U = keys_of(Ub, d_U);
  if (blockIdx.x * kFlagBlk >= U) {  // (workgroup-uniform: the launch is sized for the bound)
    if (threadIdx.x == 0) bc[blockIdx.x] = bc[nb + blockIdx.x] = 0;
    return;
  }
  uint32_t hb, tb, th, tt;
  key_flag_bits(segptr, U, blockIdx.x * kFlagBlk + threadIdx.x * kFlagPer, hb, tb);
  block_prefix((uint32_t)__popc(hb), wsum, &th);
  block_prefix((uint32_t)__popc(tb), wsum, &tt);
  if (threadIdx.x == 0) {
    bc[blockIdx.x] = th;
    bc[nb + blockIdx.x] = tt;
  }
}

// one workgroup: both rows of bc scanned in place (exclusive); totals[0] = H, totals[1] = ntiles
__global__ void __launch_bounds__(kBlock)
k_flag_bases(uint32_t *__restrict__ bc, uint32_t nb, uint32_t *__restrict__ totals) {
  __shared__ uint32_t wsum[kBlock / 64];
  for (uint32_t row = 0; row < 2; ++row) {
    uint32_t *p = bc + (size_t)row * nb;
    uint32_t carry = 0;
    for (uint32_t i0 = 0; i0 < nb; i0 += kBlock) {  // workgroup-uniform trip count
      const uint32_t i = i0 + threadIdx.x;
      const uint32_t x = i < nb ? p[i] : 0u;
      uint32_t tot;
      const uint32_t e = block_prefix(x, wsum, &tot);
      if (i < nb) p[i] = carry + e;
      carry += tot;
    }
    if (threadIdx.x == 0) totals[row] = carry;
  }
}

__global__ void __launch_bounds__(kBlock)
k_flag_compact(const uint32_t *__restrict__ segptr, uint32_t Ub,
               const uint32_t *__restrict__ d_U, uint32_t nb, const uint32_t *__restrict__ bc,
               uint32_t *__restrict__ heavy, uint32_t *__restrict__ tile_ptr) {
  __shared__ uint32_t wsum[kBlock / 64];
  const uint32_t U = keys_of(Ub, d_U);
  if (blockIdx.x * kFlagBlk >= U) return;  // (workgroup-uniform)
  const uint32_t u0 = blockIdx.x * kFlagBlk + threadIdx.x * kFlagPer;
  uint32_t hb, tb, tot;
  key_flag_bits(segptr, U, u0, hb, tb);
  uint32_t ph = bc[blockIdx.x] + block_prefix((uint32_t)__popc(hb), wsum, &tot);
  uint32_t pt = bc[nb + blockIdx.x] + block_prefix((uint32_t)__popc(tb), wsum, &tot);
  for (; hb; hb &= hb - 1) heavy[ph++] = u0 + (uint32_t)__ffs((int)hb) - 1u;
  for (; tb; tb &= tb - 1) tile_ptr[pt++] = u0 + (uint32_t)__ffs((int)tb) - 1u;
}

// heavy [<= U] and tile_ptr [<= U + 1] filled; d_totals[0] = H, d_totals[1] = ntiles (device).
// d_U != null: the number of keys is on the device, U is a bound on it.
static int key_lists(xf::Scratch &sc, const uint32_t *segptr, uint32_t U, const uint32_t *d_U,
                     uint32_t *heavy, uint32_t *tile_ptr, uint32_t **d_totals, hipStream_t s) {
  const uint32_t nb = (U + kFlagBlk - 1) / kFlagBlk;
  uint32_t *bc = nullptr;
  XF_TRY(sc.get(&bc, (size_t)2 * nb + 2));
  *d_totals = bc + (size_t)2 * nb;
  if (!U) {
    XF_HIP(hipMemsetAsync(*d_totals, 0, 8, s));
    return XF_OK;
  }
  hipLaunchKernelGGL(k_flag_counts, dim3(nb), dim3(kBlock), 0, s, segptr, U, d_U, nb, bc);
  hipLaunchKernelGGL(k_flag_bases, dim3(1), dim3(kBlock), 0, s, bc, nb, *d_totals);
  hipLaunchKernelGGL(k_flag_compact, dim3(nb), dim3(kBlock), 0, s, segptr, U, d_U, nb, bc, heavy,
                     tile_ptr);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

// out[scan[i]] = i for flagged i (scan = exclusive scan of flag); out[total] = sentinel
__global__ void k_compact_index(const uint32_t *__restrict__ flag,
                                const uint32_t *__restrict__ scan, size_t n,
                                uint32_t *__restrict__ out) {
  XF_GRID_STRIDE(i, n) if (flag[i]) out[scan[i]] = (uint32_t)i;
}

// cnt[p*(R+1) + r + 1] = nonzeros of row r in panel p.  One wave per row; the per-panel
// counters live in LDS and are bumped once per distinct panel of a 64-nonzero chunk (ballot),
// so there are no global atomics.
__global__ void __launch_bounds__(kBlock)
k_cell_counts(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ uidx, uint32_t R,
              uint32_t P, uint32_t U, uint32_t *__restrict__ cnt /* P*(R+1) */) {
  __shared__ uint32_t off_all[kBlock / 64][256];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t *off = off_all[wv];
  const uint32_t nw = gridDim.x * (kBlock / 64);
  for (uint32_t r = blockIdx.x * (kBlock / 64) + wv; r < R; r += nw) {
    for (uint32_t p = lane; p < P; p += 64) off[p] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint32_t b = rowptr[r], e = rowptr[r + 1];
    for (uint32_t j0 = b; j0 < e; j0 += 64) {
      const uint32_t j = j0 + lane;
      const bool act = j < e;
      const uint32_t p = act ? xf::panel_of(uidx[j], P, U) : 0xFFFFFFFFu;
      unsigned long long todo = __ballot(act);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t p0 = __shfl(p, leader);
        const unsigned long long m = __ballot(p == p0);
        if ((int)lane == leader) off[p0] += __popcll(m);
        todo &= ~m;
      }
    }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t p = lane; p < P; p += 64) cnt[(size_t)p * (R + 1) + r + 1] = off[p];
    if (lane == 0 && r == 0)
      for (uint32_t p = 0; p < P; ++p) cnt[(size_t)p * (R + 1)] = 0;  // the panels' first cells
    __builtin_amdgcn_wave_barrier();
  }
}

// pptr = exclusive scan of cnt laid out [p][r+1] -> here cnt already holds counts shifted by
// one cell, so an INCLUSIVE scan over the flat array gives pptr directly.

// stable placement: within a (panel,row) cell the CSR order is kept
__global__ void __launch_bounds__(kBlock)
k_fill_pidx(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ uidx,
            const uint32_t *__restrict__ pptr, uint32_t R, uint32_t P, uint32_t U,
            uint32_t *__restrict__ pidx) {
  __shared__ uint32_t off_all[kBlock / 64][256];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t *off = off_all[wv];
  const uint32_t nw = gridDim.x * (kBlock / 64);
  for (uint32_t r = blockIdx.x * (kBlock / 64) + wv; r < R; r += nw) {
    for (uint32_t p = lane; p < P; p += 64) off[p] = 0;
    __builtin_amdgcn_wave_barrier();
    const uint32_t b = rowptr[r], e = rowptr[r + 1];
    for (uint32_t j0 = b; j0 < e; j0 += 64) {
      const uint32_t j = j0 + lane;
      const bool act = j < e;
      const uint32_t ui = act ? uidx[j] : 0;
      const uint32_t p = act ? xf::panel_of(ui, P, U) : 0xFFFFFFFFu;
      unsigned long long todo = __ballot(act);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t p0 = __shfl(p, leader);
        const unsigned long long m = __ballot(p == p0);
        if (p == p0) {
          const uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
          pidx[pptr[(size_t)p0 * (R + 1) + r] + off[p0] + rank] = ui;
        }
        __builtin_amdgcn_wave_barrier();
        if ((int)lane == leader) off[p0] += __popcll(m);
        __builtin_amdgcn_wave_barrier();
        todo &= ~m;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// chunks of XF_TILE_NNZ occurrences per heavy key
__global__ void k_heavy_chunk_counts(const uint32_t *__restrict__ heavy, uint32_t H,
                                     const uint32_t *__restrict__ segptr,
                                     uint32_t *__restrict__ cnt /* H+1, cnt[H] = 0 */) {
  XF_GRID_STRIDE(h, (size_t)H + 1) {
    const uint32_t u = h < H ? heavy[h] : 0;
    cnt[h] = h < H ? (segptr[u + 1] - segptr[u] + XF_TILE_NNZ - 1) / XF_TILE_NNZ : 0u;
  }
}

__global__ void k_cell_flags(const uint32_t *__restrict__ pptr, uint32_t R, uint32_t P,
                             uint32_t *__restrict__ flag /* P*(R+1) */) {
  const size_t n = (size_t)P * (R + 1);
  XF_GRID_STRIDE(s, n) {
    const uint32_t p = (uint32_t)(s / (R + 1)), r = (uint32_t)(s - (size_t)p * (R + 1));
    flag[s] = (r < R && xf::fwd_tile_starts_at(pptr + (size_t)p * (R + 1), r)) ? 1u : 0u;
  }
}

__global__ void k_panel_first(const uint32_t *__restrict__ scan, uint32_t R, uint32_t P,
                              uint32_t total, uint32_t *__restrict__ panel_first) {
  XF_GRID_STRIDE(p, (size_t)P + 1)
  panel_first[p] = p < P ? scan[(size_t)p * (R + 1)] : total;
}

using xf::Scratch;

// u32 prefix sums (round 6: by hand, the library's scan until then): blocks of kScanBlk elements
// — their sums, one workgroup over the sums, the blocks again from their bases.  in == out is
// allowed (a thread reads its 16 neighbouring elements before it writes them).
constexpr uint32_t kScanPer = 16, kScanBlk = kBlock * kScanPer;
__global__ void __launch_bounds__(kBlock)
k_scan_sums(const uint32_t *__restrict__ in, size_t n, uint32_t *__restrict__ bs) {
  __shared__ uint32_t wsum[kBlock / 64];
  const size_t i0 = (size_t)blockIdx.x * kScanBlk + (size_t)threadIdx.x * kScanPer;
  uint32_t x = 0;
#pragma unroll
  for (uint32_t k = 0; k < kScanPer; ++k) x += i0 + k < n ? in[i0 + k] : 0u;
  uint32_t total;
  (void)block_prefix(x, wsum, &total);
  if (threadIdx.x == 0) bs[blockIdx.x] = total;
}
__global__ void __launch_bounds__(kBlock)
k_scan_bases(uint32_t *__restrict__ bs, uint32_t nb) {  // one workgroup: exclusive, in place
  __shared__ uint32_t wsum[kBlock / 64];
  uint32_t carry = 0;
  for (uint32_t i0 = 0; i0 < nb; i0 += kBlock) {  // workgroup-uniform trip count
    const uint32_t i = i0 + threadIdx.x, x = i < nb ? bs[i] : 0u;
    uint32_t total;
    const uint32_t e = block_prefix(x, wsum, &total);
    if (i < nb) bs[i] = carry + e;
    carry += total;
  }
}
template <bool INCLUSIVE>
__global__ void __launch_bounds__(kBlock)
k_scan_apply(const uint32_t *in, uint32_t *out, size_t n, const uint32_t *__restrict__ bs) {
  __shared__ uint32_t wsum[kBlock / 64];
  const size_t i0 = (size_t)blockIdx.x * kScanBlk + (size_t)threadIdx.x * kScanPer;
  uint32_t v[kScanPer], x = 0;
#pragma unroll
  for (uint32_t k = 0; k < kScanPer; ++k) {
    v[k] = i0 + k < n ? in[i0 + k] : 0u;
    x += v[k];
  }
  uint32_t total;
  uint32_t run = bs[blockIdx.x] + block_prefix(x, wsum, &total);
#pragma unroll
  for (uint32_t k = 0; k < kScanPer; ++k) {
    if (INCLUSIVE) run += v[k];
    if (i0 + k < n) out[i0 + k] = run;
    if (!INCLUSIVE) run += v[k];
  }
}
template <bool INCLUSIVE>
int scan_u32(Scratch &sc, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
  if (!n) return XF_OK;
  const uint32_t nb = (uint32_t)((n + kScanBlk - 1) / kScanBlk);
  uint32_t *bs = nullptr;
  XF_TRY(sc.get(&bs, nb));
  hipLaunchKernelGGL(k_scan_sums, dim3(nb), dim3(kBlock), 0, s, in, n, bs);
  hipLaunchKernelGGL(k_scan_bases, dim3(1), dim3(kBlock), 0, s, bs, nb);
  hipLaunchKernelGGL(k_scan_apply<INCLUSIVE>, dim3(nb), dim3(kBlock), 0, s, in, out, n, bs);
  XF_HIP(hipGetLastError());
  return XF_OK;
}
int exclusive_scan_u32(Scratch &sc, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
  return scan_u32<false>(sc, in, out, n, s);
}
int inclusive_scan_u32(Scratch &sc, const uint32_t *in, uint32_t *out, size_t n, hipStream_t s) {
  return scan_u32<true>(sc, in, out, n, s);
}

}  // namespace

// (key, position) of d_keys[0..n) in key order, positions ascending inside a key.  Round 6: the
// partition by uniform key range + a range sorted in LDS (xf::sort_key_pos, xf_keybuild.hip);
// the library's radix sort beyond that sort's limits — more than 3.3e7 keys — and under
// xf_tune key_build = 1 (the tests' second implementation).
namespace xf {
int sort_key_pos_any(const uint64_t *d_keys, uint32_t n, uint64_t lo, uint64_t span, uint64_t *sk,
                     uint32_t *spos, hipStream_t s, bool *by_hand, uint32_t site) {
  bool sorted = false;
  if (n) XF_TRY(sort_key_pos(d_keys, n, lo, span, sk, spos, s, &sorted, site));
  if (by_hand) *by_hand = sorted;
  if (sorted || !n) return XF_OK;
  Scratch sc;
  uint32_t *pos = nullptr;
  XF_TRY(sc.get(&pos, n));
  hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(kBlock), 0, s, pos, (size_t)n);
  size_t tb = 0;
  XF_HIP(rocprim::radix_sort_pairs(nullptr, tb, d_keys, sk, pos, spos, (size_t)n, 0, 64, s));
  void *tmp = nullptr;
  XF_TRY(sc.get((char **)&tmp, tb));
  XF_HIP(rocprim::radix_sort_pairs(tmp, tb, d_keys, sk, pos, spos, (size_t)n, 0, 64, s));
  XF_HIP(hipStreamSynchronize(s));  // (the scratch goes back)
  return XF_OK;
}
}  // namespace xf

extern "C" int xf_sort_key_pos(const uint64_t *d_keys, uint32_t n, uint64_t lo, uint64_t span,
                               uint64_t *d_sorted_keys, uint32_t *d_sorted_pos, void *stream,
                               int *by_hand) {
  XF_REQUIRE(n == 0 || (d_keys && d_sorted_keys && d_sorted_pos), "xf_sort_key_pos: null argument");
  bool h = false;
  XF_TRY(xf::sort_key_pos_any(d_keys, n, lo, span, d_sorted_keys, d_sorted_pos,
                              (hipStream_t)stream, &h));
  if (by_hand) *by_hand = h ? 1 : 0;
  return XF_OK;
}

// Device key build.  d_keys[NNZ], d_rowptr[R+1] (row-relative, d_rowptr[0] == 0),
// d_labels[R] are device pointers; the compiled batch stays on the device
// (xf_batch_download brings the arrays to the host for inspection).
extern "C" int xf_batch_compile_dev(xf_batch **out, const uint64_t *d_keys,
                                    const uint32_t *d_rowptr, const int32_t *d_labels,
                                    uint32_t R, uint32_t NNZ, void *stream) {
  return xf::batch_compile_dev_ex(out, d_keys, d_rowptr, d_labels, R, NNZ, (hipStream_t)stream, true);
}

// panels = false: without the panel-major forward view (pptr / pidx and its tiles: what
// xf_lr_forward_dev streams — FM's kernels and the sharded trainer's never do; 0.35 ms of a 1e7-
// nonzero build)
int xf::batch_compile_dev_ex(xf_batch **out, const uint64_t *d_keys, const uint32_t *d_rowptr,
                             const int32_t *d_labels, uint32_t R, uint32_t NNZ, hipStream_t stream,
                             bool panels) {
  XF_REQUIRE(out && d_rowptr && (R == 0 || d_labels) && (NNZ == 0 || d_keys),
             "xf_batch_compile_dev: null argument");
  hipStream_t s = (hipStream_t)stream;
  Scratch sc;
  xf_batch *b = new xf_batch;
  b->R = R;
  b->NNZ = NNZ;
  b->on_device_only = true;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };

  // ---- 1. sort (key, position)
  uint64_t *sk = nullptr;
  uint32_t *pos = nullptr, *spos = nullptr, *head = nullptr, *uid1 = nullptr;
  XF_TRY(sc.get(&sk, NNZ));
  XF_TRY(sc.get(&pos, NNZ));
  XF_TRY(sc.get(&spos, NNZ));
  XF_TRY(sc.get(&head, NNZ));
  XF_TRY(sc.get(&uid1, NNZ));
  uint32_t U = 0;
  if (NNZ) {
    XF_TRY(xf::sort_key_pos_any(d_keys, NNZ, 0, ~0ull, sk, spos, s, nullptr, xf::kSortSiteBatch));
    // ---- 2. unique index
    hipLaunchKernelGGL(k_heads, dim3(grid_for(NNZ)), dim3(kBlock), 0, s, sk, (size_t)NNZ, head);
    XF_TRY(inclusive_scan_u32(sc, head, uid1, NNZ, s));
    XF_HIP(hipMemcpyAsync(&U, uid1 + (NNZ - 1), 4, hipMemcpyDeviceToHost, s));
    XF_HIP(hipStreamSynchronize(s));
  }
  b->U = U;
  uint64_t *ukeys = nullptr;
  uint32_t *segptr = nullptr, *uidx = nullptr, *coo = nullptr;
  XF_TRY(sc.get(&ukeys, U));
  XF_TRY(sc.get(&segptr, (size_t)U + 1));
  XF_TRY(sc.get(&uidx, NNZ));
  XF_TRY(sc.get(&coo, NNZ));
  if (NNZ) {
    hipLaunchKernelGGL(k_unique, dim3(grid_for(NNZ)), dim3(kBlock), 0, s, sk, head, uid1,
                       (size_t)NNZ, U, ukeys, segptr);
    hipLaunchKernelGGL(k_row_of, dim3(grid_for((size_t)R * 64)), dim3(kBlock), 0, s, d_rowptr, R,
                       pos /* reused: iota is no longer needed after the sort */);
    hipLaunchKernelGGL(k_uidx_coo, dim3(grid_for(NNZ)), dim3(kBlock), 0, s, spos, uid1, pos,
                       (size_t)NNZ, uidx, coo);
  } else {
    XF_HIP(hipMemsetAsync(segptr, 0, 4, s));
  }
  // ---- 3. heavy keys and gradient tiles
  uint32_t *heavy = nullptr, *tile_ptr = nullptr, *d_tot = nullptr;
  XF_TRY(sc.get(&heavy, (size_t)U + 1));
  XF_TRY(sc.get(&tile_ptr, (size_t)U + 2));
  uint32_t tot[2] = {0, 0};
  uint32_t &H = tot[0], &ntiles = tot[1];  // (on the host after the synchronisation below)
  XF_TRY(key_lists(sc, segptr, U, nullptr, heavy, tile_ptr, &d_tot, s));
  XF_HIP(hipMemcpyAsync(tot, d_tot, 8, hipMemcpyDeviceToHost, s));
  // ---- 4. panel-major forward view
  const uint32_t P =
      panels ? xf::panel_count(U, NNZ, xf::panel_slice_bytes(), xf::min_panel_nnz()) : 0u;
  const size_t ncell = (size_t)P * ((size_t)R + 1);
  uint32_t *pptr = nullptr, *pidx = nullptr, *cflag = nullptr, *cscan = nullptr;
  uint32_t *ftile = nullptr, *fpf = nullptr;
  uint32_t nft = 0;
  std::vector<uint32_t> h_fpf;
  if (P) {
    XF_TRY(sc.get(&pptr, ncell));
    XF_TRY(sc.get(&pidx, NNZ));
    XF_TRY(sc.get(&cflag, ncell + 1));
    XF_TRY(sc.get(&cscan, ncell + 1));
    XF_TRY(sc.get(&ftile, ncell + 2));
    XF_TRY(sc.get(&fpf, (size_t)P + 1));
    hipLaunchKernelGGL(k_cell_counts, dim3(grid_for((size_t)R * 64)), dim3(kBlock), 0, s,
                       d_rowptr, uidx, R, P, U, pptr);
    XF_TRY(inclusive_scan_u32(sc, pptr, pptr, ncell, s));  // shifted counts -> offsets
    hipLaunchKernelGGL(k_fill_pidx, dim3(grid_for((size_t)R * 64)), dim3(kBlock), 0, s, d_rowptr,
                       uidx, pptr, R, P, U, pidx);
    XF_HIP(hipMemsetAsync(cflag + ncell, 0, 4, s));
    hipLaunchKernelGGL(k_cell_flags, dim3(grid_for(ncell)), dim3(kBlock), 0, s, pptr, R, P, cflag);
    XF_TRY(exclusive_scan_u32(sc, cflag, cscan, ncell + 1, s));  // cscan[ncell] = nft
    hipLaunchKernelGGL(k_compact_index, dim3(grid_for(ncell)), dim3(kBlock), 0, s, cflag, cscan,
                       ncell, ftile);
    XF_HIP(hipMemcpyAsync(&nft, cscan + ncell, 4, hipMemcpyDeviceToHost, s));
  }
  XF_HIP(hipGetLastError());
  XF_HIP(hipStreamSynchronize(s));
  b->H = H;
  uint32_t *hch = nullptr;
  uint32_t n_hch = 0;
  if (H) {
    uint32_t *hcnt = nullptr;
    XF_TRY(sc.get(&hcnt, (size_t)H + 1));
    XF_TRY(sc.get(&hch, (size_t)H + 1));
    hipLaunchKernelGGL(k_heavy_chunk_counts, dim3(grid_for((size_t)H + 1)), dim3(kBlock), 0, s,
                       heavy, H, segptr, hcnt);
    XF_TRY(exclusive_scan_u32(sc, hcnt, hch, (size_t)H + 1, s));  // hch[H] = total chunks
    XF_HIP(hipMemcpyAsync(&n_hch, hch + H, 4, hipMemcpyDeviceToHost, s));
    XF_HIP(hipStreamSynchronize(s));
  }
  if (U) {  // close the tile lists
    XF_HIP(hipMemcpyAsync(tile_ptr + ntiles, &U, 4, hipMemcpyHostToDevice, s));
  } else {
    const uint32_t zero = 0;
    XF_HIP(hipMemcpyAsync(tile_ptr, &zero, 4, hipMemcpyHostToDevice, s));
  }
  if (P) {
    const uint32_t endcell = (uint32_t)((size_t)(P - 1) * ((size_t)R + 1) + R);
    XF_HIP(hipMemcpyAsync(ftile + nft, &endcell, 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_panel_first, dim3(1), dim3(kBlock), 0, s, cscan, R, P, nft, fpf);
    h_fpf.resize((size_t)P + 1);
    XF_HIP(hipMemcpyAsync(h_fpf.data(), fpf, ((size_t)P + 1) * 4, hipMemcpyDeviceToHost, s));
    XF_HIP(hipStreamSynchronize(s));
    uint32_t longest = 0;
    for (uint32_t x = 0; x < 8; ++x) {
      uint32_t len = 0;
      for (uint32_t p = x; p < P; p += 8) len += h_fpf[p + 1] - h_fpf[p];
      longest = std::max(longest, len);
    }
    b->fwd_grid = 8 * longest;
  }
  b->P = P;

  // ---- assemble the batch's own device allocation (same layout as xf_batch_upload)
  const size_t n_tile = (size_t)ntiles + 1, n_ftile = P ? (size_t)nft + 1 : 0;
  const size_t o_ukeys = 0;
  const size_t o_rowptr = o_ukeys + al((size_t)U * 8);
  const size_t o_uidx = o_rowptr + al(((size_t)R + 1) * 4);
  const size_t o_segptr = o_uidx + al((size_t)NNZ * 4);
  const size_t o_coo = o_segptr + al(((size_t)U + 1) * 4);
  const size_t o_labels = o_coo + al((size_t)NNZ * 4);
  const size_t o_heavy = o_labels + al((size_t)R * 4);
  const size_t o_pptr = o_heavy + al((size_t)H * 4);
  const size_t o_pidx = o_pptr + al(ncell * 4);
  const size_t o_scr = o_pidx + al(P ? (size_t)NNZ * 4 : 0);
  const size_t o_tile = o_scr + al((size_t)P * R * 8);
  const size_t o_ftile = o_tile + al(n_tile * 4);
  const size_t o_fpf = o_ftile + al(n_ftile * 4);
  const size_t o_hch = o_fpf + al(P ? ((size_t)P + 1) * 4 : 0);
  const size_t o_hscr = o_hch + al(H ? ((size_t)H + 1) * 4 : 0);
  const size_t total = o_hscr + al((size_t)n_hch * (1 + XF_HEAVY_KMAX) * 8) + 256;
  char *d = nullptr;
  XF_TRY(xf::blob_alloc((void **)&d, total, &b->d_blob_bytes));
  auto cp = [&](size_t off, const void *src, size_t bytes) -> hipError_t {
    if (!bytes) return hipSuccess;
    return hipMemcpyAsync(d + off, src, bytes, hipMemcpyDeviceToDevice, s);
  };
  XF_HIP(cp(o_ukeys, ukeys, (size_t)U * 8));
  XF_HIP(cp(o_rowptr, d_rowptr, ((size_t)R + 1) * 4));
  XF_HIP(cp(o_uidx, uidx, (size_t)NNZ * 4));
  XF_HIP(cp(o_segptr, segptr, ((size_t)U + 1) * 4));
  XF_HIP(cp(o_coo, coo, (size_t)NNZ * 4));
  XF_HIP(cp(o_labels, d_labels, (size_t)R * 4));
  XF_HIP(cp(o_heavy, heavy, (size_t)H * 4));
  XF_HIP(cp(o_tile, tile_ptr, n_tile * 4));
  if (H) XF_HIP(cp(o_hch, hch, ((size_t)H + 1) * 4));
  if (P) {
    XF_HIP(cp(o_pptr, pptr, ncell * 4));
    XF_HIP(cp(o_pidx, pidx, (size_t)NNZ * 4));
    XF_HIP(cp(o_ftile, ftile, n_ftile * 4));
    XF_HIP(cp(o_fpf, fpf, ((size_t)P + 1) * 4));
  }
  XF_HIP(hipStreamSynchronize(s));
  b->d_blob = d;
  xf_dev_batch &v = b->view;
  v.R = R;
  v.NNZ = NNZ;
  v.U = U;
  v.H = H;
  v.ukeys = (const uint64_t *)(d + o_ukeys);
  v.rowptr = (const uint32_t *)(d + o_rowptr);
  v.uidx = (const uint32_t *)(d + o_uidx);
  v.segptr = (const uint32_t *)(d + o_segptr);
  v.coo_row = (const uint32_t *)(d + o_coo);
  v.labels = (const int32_t *)(d + o_labels);
  v.heavy = H ? (const uint32_t *)(d + o_heavy) : nullptr;
  v.P = P;
  v.fwd_ntiles = P ? nft : 0;
  v.pptr = P ? (const uint32_t *)(d + o_pptr) : nullptr;
  v.pidx = P ? (const uint32_t *)(d + o_pidx) : nullptr;
  v.fwd_scratch = P ? (double *)(d + o_scr) : nullptr;
  v.fwd_tile_ptr = P ? (const uint32_t *)(d + o_ftile) : nullptr;
  v.fwd_panel_first = P ? (const uint32_t *)(d + o_fpf) : nullptr;
  v.fwd_grid = b->fwd_grid;
  v.pad3_ = 0;
  v.ntiles = ntiles;
  v.n_heavy_chunks = n_hch;
  v.heavy_chunk_ptr = H ? (const uint32_t *)(d + o_hch) : nullptr;
  v.heavy_scratch = H ? (double *)(d + o_hscr) : nullptr;
  v.tile_ptr = (const uint32_t *)(d + o_tile);
  *out = b;
  return XF_OK;
}

// Bring a device-built batch's arrays to the host vectors (inspection / tests).
extern "C" int xf_batch_download(xf_batch *b) {
  XF_REQUIRE(b && b->d_blob, "xf_batch_download: batch is not on the device");
  const xf_dev_batch &v = b->view;
  auto get = [&](auto &vec, const void *src, size_t n) -> hipError_t {
    vec.resize(n);
    if (!n) return hipSuccess;
    return hipMemcpy(vec.data(), src, n * sizeof(vec[0]), hipMemcpyDeviceToHost);
  };
  XF_HIP(get(b->ukeys, v.ukeys, v.U));
  XF_HIP(get(b->rowptr, v.rowptr, (size_t)v.R + 1));
  XF_HIP(get(b->uidx, v.uidx, v.NNZ));
  XF_HIP(get(b->segptr, v.segptr, (size_t)v.U + 1));
  XF_HIP(get(b->coo_row, v.coo_row, v.NNZ));
  XF_HIP(get(b->labels, v.labels, v.R));
  XF_HIP(get(b->heavy, v.heavy, v.H));
  if (v.H) XF_HIP(get(b->hchunk_ptr, v.heavy_chunk_ptr, (size_t)v.H + 1));
  else
    b->hchunk_ptr.assign(1, 0);
  XF_HIP(get(b->tile_ptr, v.tile_ptr, (size_t)v.ntiles + 1));
  if (v.P) {
    XF_HIP(get(b->pptr, v.pptr, (size_t)v.P * ((size_t)v.R + 1)));
    XF_HIP(get(b->pidx, v.pidx, v.NNZ));
    XF_HIP(get(b->ftile_ptr, v.fwd_tile_ptr, (size_t)v.fwd_ntiles + 1));
    XF_HIP(get(b->fpanel_first, v.fwd_panel_first, (size_t)v.P + 1));
  }
  b->on_device_only = false;
  return XF_OK;
}

// ---------------------------------------------------------------------------------------
// The FM key build against the tables themselves (fm_worker.cc:205-225 + the key -> row step
// of its two Pulls, :228,:231): when every key of the minibatch sits in the v table's settled
// tier and the w table numbers its rows the same way, the keyed build of xf_keybuild.hip gives
// the key list with its state rows, the key-grouped occurrence lists and the forward's
// per-nonzero record index without sorting (key, position) pairs (the 64-bit radix sort and
// the scattered index passes of xf_batch_compile_dev: 1.7 ms per 10^7 nonzeros).  Otherwise
// (first minibatches of a run, keys new since the last xf_table_defrag, a minibatch of more
// than 16 row windows) it IS xf_batch_compile_dev.  *keyed_out (optional) says which.
namespace xf {
const TableDev &table_dev(const xf_table *t);
uint64_t table_uid(const xf_table *t);
uint64_t table_epoch(const xf_table *t);
int table_dim(const xf_table *t);
bool fm_records_fit(int k);
typedef int (*FmKeyedOut)(void *ctx, uint32_t U, uint64_t **ukeys, uint32_t **urow,
                          uint32_t **segptr, uint32_t **coo);
int fm_build_keyed(xf_table *t, const uint64_t *d_keys, const uint32_t *d_rowptr, uint32_t R,
                   uint32_t NNZ, Scratch &sc, hipStream_t s, bool *ok, const uint32_t **d_U,
                   const unsigned long long **d_miss, uint32_t *ridx, FmKeyedOut place,
                   void *ctx);
}  // namespace xf

namespace {
__global__ void k_keys_differ(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b,
                              size_t n, unsigned int *__restrict__ flag) {
  XF_GRID_STRIDE(i, n) if (a[i] != b[i]) *flag = 1u;
}

// do the two tables' settled tiers hold the same keys (so that the key of rank r has state row
// r in both)?  Compared on the device once per pair of table epochs.
int same_numbering(xf_table *w, xf_table *v, hipStream_t s, bool *same) {
  struct Memo {
    uint64_t uw, ew, uv, ev;
    bool same;
  };
  static thread_local Memo memo{0, 0, 0, 0, false};
  const uint64_t uw = xf::table_uid(w), ew = xf::table_epoch(w);
  const uint64_t uv = xf::table_uid(v), ev = xf::table_epoch(v);
  if (memo.uw == uw && memo.ew == ew && memo.uv == uv && memo.ev == ev) {
    *same = memo.same;
    return XF_OK;
  }
  const xf::TableDev &TW = xf::table_dev(w), &TV = xf::table_dev(v);
  bool eq = TW.nbase == TV.nbase && TW.nbase > 0 && TW.lo == TV.lo && TW.span == TV.span;
  if (eq) {
    xf::Scratch sc;
    unsigned int *flag = nullptr, h = 0;
    XF_TRY(sc.get(&flag, 1));
    XF_HIP(hipMemsetAsync(flag, 0, 4, s));
    hipLaunchKernelGGL(k_keys_differ, dim3(grid_for(TW.nbase)), dim3(kBlock), 0, s, TW.bkeys,
                       TV.bkeys, (size_t)TW.nbase, flag);
    XF_HIP(hipMemcpyAsync(&h, flag, 4, hipMemcpyDeviceToHost, s));
    XF_HIP(hipStreamSynchronize(s));
    eq = h == 0;
  }
  memo = Memo{uw, ew, uv, ev, eq};
  *same = eq;
  return XF_OK;
}
}  // namespace

extern "C" int xf_batch_compile_fm_dev(xf_batch **out, xf_table *w, xf_table *v,
                                       const uint64_t *d_keys, const uint32_t *d_rowptr,
                                       const int32_t *d_labels, uint32_t R, uint32_t NNZ,
                                       void *stream, int *keyed_out) {
  XF_REQUIRE(out && w && v && d_rowptr && (R == 0 || d_labels) && (NNZ == 0 || d_keys),
             "xf_batch_compile_fm_dev: null argument");
  hipStream_t s = (hipStream_t)stream;
  if (keyed_out) *keyed_out = 0;
  bool same = false;
  // (a keyed minibatch steps on the table-resident records only: factor widths they exist for)
  const char *rec_off = getenv("XF_FM_TABLE_RECORDS");  // ("0": the records are switched off)
  if (NNZ && R && xf::fm_records_fit(xf::table_dim(v)) && !(rec_off && *rec_off == '0'))
    XF_TRY(same_numbering(w, v, s, &same));
  if (!same) return xf::batch_compile_dev_ex(out, d_keys, d_rowptr, d_labels, R, NNZ, (hipStream_t)stream, false);
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  uint32_t *ridx = nullptr;
  size_t ridx_bytes = 0;
  XF_TRY(xf::blob_alloc((void **)&ridx, (size_t)NNZ * 4, &ridx_bytes));
  struct RidxGuard {
    uint32_t *p;
    size_t n;
    ~RidxGuard() {
      if (p) xf::blob_free(p, n);
    }
  } rguard{ridx, ridx_bytes};
  // the batch and, once the number of distinct keys is known, its allocation: the build writes
  // the key list, its rows, the segment offsets and the occurrence lists straight into it
  xf_batch *b = new xf_batch;
  struct Guard {
    xf_batch *b;
    ~Guard() {
      if (b) xf_batch_free(b);
    }
  } guard{b};
  struct Place {
    xf_batch *b;
    uint32_t R, NNZ, Ub;
    size_t o_ukeys, o_rowptr, o_segptr, o_coo, o_labels;
    static int at(void *ctx, uint32_t U, uint64_t **ukeys, uint32_t **urow, uint32_t **segptr,
                  uint32_t **coo) {
      Place *p = (Place *)ctx;
      auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
      p->Ub = U;  // (a bound: the number itself is still on the device)
      p->o_ukeys = 0;
      p->o_rowptr = p->o_ukeys + al((size_t)U * 8);
      p->o_segptr = p->o_rowptr + al(((size_t)p->R + 1) * 4);
      p->o_coo = p->o_segptr + al(((size_t)U + 1) * 4);
      p->o_labels = p->o_coo + al((size_t)p->NNZ * 4);
      const size_t total = p->o_labels + al((size_t)p->R * 4) + 256;
      char *d = nullptr;
      XF_TRY(xf::blob_alloc((void **)&d, total, &p->b->d_blob_bytes));
      p->b->d_blob = d;
      for (int i = 0; i < 2; ++i)  // the key list's rows, in both tables (the same numbers)
        XF_TRY(xf::blob_alloc((void **)&p->b->d_fm_rows[i], std::max<size_t>(U, 1) * 4,
                              &p->b->fm_rows_bytes[i]));
      *ukeys = (uint64_t *)(d + p->o_ukeys);
      *urow = p->b->d_fm_rows[1];
      *segptr = (uint32_t *)(d + p->o_segptr);
      *coo = (uint32_t *)(d + p->o_coo);
      return XF_OK;
    }
  } place{b, R, NNZ, 0, 0, 0, 0, 0, 0};
  Scratch sc;
  bool ok = false;
  const uint32_t *d_U = nullptr;
  const unsigned long long *d_miss = nullptr;
  XF_TRY(xf::fm_build_keyed(v, d_keys, d_rowptr, R, NNZ, sc, s, &ok, &d_U, &d_miss, ridx,
                            &Place::at, &place));
  if (!ok) {
    XF_HIP(hipStreamSynchronize(s));
    return xf::batch_compile_dev_ex(out, d_keys, d_rowptr, d_labels, R, NNZ, (hipStream_t)stream, false);
  }
  char *d = (char *)b->d_blob;
  const uint32_t *segptr = (const uint32_t *)(d + place.o_segptr);
  XF_HIP(hipMemcpyAsync(b->d_fm_rows[0], b->d_fm_rows[1], (size_t)place.Ub * 4,
                        hipMemcpyDeviceToDevice, s));
  XF_HIP(hipMemcpyAsync(d + place.o_rowptr, d_rowptr, ((size_t)R + 1) * 4,
                        hipMemcpyDeviceToDevice, s));
  XF_HIP(hipMemcpyAsync(d + place.o_labels, d_labels, (size_t)R * 4, hipMemcpyDeviceToDevice, s));
  // ---- heavy keys and gradient tiles (as in xf_batch_compile_dev); the number of keys still
  // on the device: the lists are sized by what NNZ allows (xf_tiling.h)
  const size_t h_max = (size_t)NNZ / (XF_HEAVY_SEG + 1) + 1;
  const size_t t_max = (size_t)NNZ / xf::kGradQuantum + 2 * h_max + 2;
  uint32_t *heavy = nullptr, *tile_ptr = nullptr, *d_tot = nullptr;
  XF_TRY(sc.get(&heavy, h_max));
  XF_TRY(sc.get(&tile_ptr, t_max + 1));
  uint32_t tot[2] = {0, 0}, U = 0;
  unsigned long long misses = 0;
  XF_TRY(key_lists(sc, segptr, place.Ub, d_U, heavy, tile_ptr, &d_tot, s));
  XF_HIP(hipMemcpyAsync(tot, d_tot, 8, hipMemcpyDeviceToHost, s));
  XF_HIP(hipMemcpyAsync(&U, d_U, 4, hipMemcpyDeviceToHost, s));
  XF_HIP(hipMemcpyAsync(&misses, d_miss, 8, hipMemcpyDeviceToHost, s));
  XF_HIP(hipGetLastError());
  XF_HIP(hipStreamSynchronize(s));  // the only wait of the build
  if (misses)  // a key the tiers do not hold: the sort-based build (what was built is dropped)
    return xf::batch_compile_dev_ex(out, d_keys, d_rowptr, d_labels, R, NNZ, (hipStream_t)stream, false);
  const uint32_t H = tot[0], ntiles = tot[1];
  uint32_t *hch = nullptr;
  uint32_t n_hch = 0;
  if (H) {
    uint32_t *hcnt = nullptr;
    XF_TRY(sc.get(&hcnt, (size_t)H + 1));
    XF_TRY(sc.get(&hch, (size_t)H + 1));
    hipLaunchKernelGGL(k_heavy_chunk_counts, dim3(grid_for((size_t)H + 1)), dim3(kBlock), 0, s,
                       heavy, H, segptr, hcnt);
    XF_TRY(exclusive_scan_u32(sc, hcnt, hch, (size_t)H + 1, s));
    XF_HIP(hipMemcpyAsync(&n_hch, hch + H, 4, hipMemcpyDeviceToHost, s));
    XF_HIP(hipStreamSynchronize(s));
  }
  XF_HIP(hipMemcpyAsync(tile_ptr + ntiles, &U, 4, hipMemcpyHostToDevice, s));
  // ---- the lists whose sizes came last, in a second allocation
  b->R = R;
  b->NNZ = NNZ;
  b->U = U;
  b->H = H;
  b->on_device_only = true;
  b->fm_keyed = true;
  b->fm_same_rows = true;
  b->fm_nbase = xf::table_dev(v).nbase;
  const size_t n_tile = (size_t)ntiles + 1;
  const size_t o_heavy = 0;
  const size_t o_tile = o_heavy + al((size_t)H * 4);
  const size_t o_hch = o_tile + al(n_tile * 4);
  const size_t o_hscr = o_hch + al(H ? ((size_t)H + 1) * 4 : 0);
  const size_t total2 = o_hscr + al((size_t)n_hch * (1 + XF_HEAVY_KMAX) * 8) + 256;
  char *d2 = nullptr;
  XF_TRY(xf::blob_alloc((void **)&d2, total2, &b->d_blob2_bytes));
  b->d_blob2 = d2;
  auto cp = [&](size_t off, const void *src, size_t bytes) -> hipError_t {
    if (!bytes) return hipSuccess;
    return hipMemcpyAsync(d2 + off, src, bytes, hipMemcpyDeviceToDevice, s);
  };
  XF_HIP(cp(o_heavy, heavy, (size_t)H * 4));
  XF_HIP(cp(o_tile, tile_ptr, n_tile * 4));
  if (H) XF_HIP(cp(o_hch, hch, ((size_t)H + 1) * 4));
  b->d_fm_ridx = ridx;
  b->fm_ridx_bytes = ridx_bytes;
  rguard.p = nullptr;
  b->fm_uid[0] = xf::table_uid(w);
  b->fm_epoch[0] = xf::table_epoch(w);
  b->fm_uid[1] = b->fm_ridx_uid = xf::table_uid(v);
  b->fm_epoch[1] = b->fm_ridx_epoch = xf::table_epoch(v);
  XF_HIP(hipStreamSynchronize(s));  // (the scratch goes back)
  xf_dev_batch &vw = b->view;
  vw = xf_dev_batch{};
  vw.R = R;
  vw.NNZ = NNZ;
  vw.U = U;
  vw.H = H;
  vw.ukeys = (const uint64_t *)(d + place.o_ukeys);
  vw.rowptr = (const uint32_t *)(d + place.o_rowptr);
  vw.uidx = nullptr;
  vw.segptr = segptr;
  vw.coo_row = (const uint32_t *)(d + place.o_coo);
  vw.labels = (const int32_t *)(d + place.o_labels);
  vw.heavy = H ? (const uint32_t *)(d2 + o_heavy) : nullptr;
  vw.ntiles = ntiles;
  vw.n_heavy_chunks = n_hch;
  vw.heavy_chunk_ptr = H ? (const uint32_t *)(d2 + o_hch) : nullptr;
  vw.heavy_scratch = H ? (double *)(d2 + o_hscr) : nullptr;
  vw.tile_ptr = (const uint32_t *)(d2 + o_tile);
  if (keyed_out) *keyed_out = 1;
  guard.b = nullptr;
  *out = b;
  return XF_OK;
}

// Host-array front end of the device key build: same arguments as xf_batch_compile (the
// reader's block arrays and a row slice); uploads the raw slice and builds on the GPU.
extern "C" int xf_batch_compile_gpu(xf_batch **out, const uint64_t *rowptr, const uint64_t *keys,
                                    const int32_t *labels, size_t row_begin, size_t row_end,
                                    void *stream) {
  XF_REQUIRE(out && rowptr && labels && row_end >= row_begin, "xf_batch_compile_gpu: bad argument");
  const size_t R = row_end - row_begin;
  const uint64_t base = rowptr[row_begin];
  const size_t NNZ = (size_t)(rowptr[row_end] - base);
  XF_REQUIRE(NNZ == 0 || keys, "xf_batch_compile_gpu: null keys");
  XF_REQUIRE(R < 0xFFFFFFFFull && NNZ < 0xFFFFFFFFull, "xf_batch_compile_gpu: batch too large");
  hipStream_t s = (hipStream_t)stream;
  std::vector<uint32_t> rp(R + 1);
  for (size_t r = 0; r <= R; ++r) rp[r] = (uint32_t)(rowptr[row_begin + r] - base);
  Scratch sc;
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
  return xf_batch_compile_dev(out, d_keys, d_rp, d_lab, (uint32_t)R, (uint32_t)NNZ, stream);
}

// host-array front end of xf_batch_compile_fm_dev (the reader's block arrays and a row slice)
extern "C" int xf_batch_compile_fm(xf_batch **out, xf_table *w, xf_table *v,
                                   const uint64_t *rowptr, const uint64_t *keys,
                                   const int32_t *labels, size_t row_begin, size_t row_end,
                                   void *stream, int *keyed) {
  XF_REQUIRE(out && w && v && rowptr && labels && row_end >= row_begin,
             "xf_batch_compile_fm: bad argument");
  const size_t R = row_end - row_begin;
  const uint64_t base = rowptr[row_begin];
  const size_t NNZ = (size_t)(rowptr[row_end] - base);
  XF_REQUIRE(NNZ == 0 || keys, "xf_batch_compile_fm: null keys");
  XF_REQUIRE(R < 0xFFFFFFFFull && NNZ < 0xFFFFFFFFull, "xf_batch_compile_fm: batch too large");
  hipStream_t s = (hipStream_t)stream;
  std::vector<uint32_t> rp(R + 1);
  for (size_t r = 0; r <= R; ++r) rp[r] = (uint32_t)(rowptr[row_begin + r] - base);
  Scratch sc;
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
  return xf_batch_compile_fm_dev(out, w, v, d_keys, d_rp, d_lab, (uint32_t)R, (uint32_t)NNZ,
                                 stream, keyed);
}

// every row's unique-key indices in ascending order (parity mode "reference order": ascending
// index = ascending fid, the order the reference's merge-join adds a row's weights in).  Built
// once per minibatch, on the device, from the uploaded CSR view.
namespace xf {
int batch_sorted_uidx(xf_batch *b, hipStream_t s) {
  XF_REQUIRE(b && !b->local, "reference-order parity mode needs a minibatch with a key list "
             "(xf_batch_compile*)");
  if (b->d_uidx_sorted || b->NNZ == 0) return XF_OK;
  XF_TRY(xf_batch_upload(b, s));
  const xf_dev_batch &v = b->view;
  XF_HIP(hipMalloc((void **)&b->d_uidx_sorted, (size_t)b->NNZ * 4));
  int bits = 1;
  while (bits < 32 && (1ull << bits) < (uint64_t)std::max<uint32_t>(b->U, 1)) ++bits;
  Scratch sc;
  size_t tb = 0;
  XF_HIP(rocprim::segmented_radix_sort_keys(nullptr, tb, v.uidx, b->d_uidx_sorted, (size_t)b->NNZ,
                                            (unsigned)b->R, v.rowptr, v.rowptr + 1, 0, bits, s));
  void *tmp = nullptr;
  XF_TRY(sc.get((char **)&tmp, tb));
  XF_HIP(rocprim::segmented_radix_sort_keys(tmp, tb, v.uidx, b->d_uidx_sorted, (size_t)b->NNZ,
                                            (unsigned)b->R, v.rowptr, v.rowptr + 1, 0, bits, s));
  XF_HIP(hipStreamSynchronize(s));
  return XF_OK;
}

// The same mode, gradient side.  calculate_gradient (lr_worker.cc:104-118; FM: fm_worker.cc:
// 134-148) adds a key's losses into an fp32 sum in the order the key's occurrences have in
// all_keys AFTER `std::sort(all_keys.begin(), all_keys.end(), sort_finder)` (lr_worker.cc:162),
// and sort_finder compares fids only (base.h:71-73): among equal fids the order is whatever
// libstdc++'s introsort makes of this very input — all_keys filled row by row, sid = the row's
// number in the slice (:150-161).  That order is reproduced by running that sort, on the host,
// once per minibatch: over (unique-key index, sid) pairs — the key list ascends, so every
// comparison has the outcome it has on the fids, and std::sort's moves depend on nothing else —
// in the element layout of Base::sample_key.  Slow on purpose (a checking mode).
int batch_reference_coo(xf_batch *b, hipStream_t s) {
  XF_REQUIRE(b && !b->local, "reference-order parity mode needs a minibatch with a key list "
             "(xf_batch_compile*)");
  if (b->d_ref_coo || b->NNZ == 0) return XF_OK;
  XF_TRY(xf_batch_upload(b, s));
  if (b->on_device_only) XF_TRY(xf_batch_download(b));
  struct SampleKey {  // base.h:65-69
    size_t fgid;
    size_t fid;
    int sid;
  };
  std::vector<SampleKey> all_keys;
  all_keys.reserve(b->NNZ);
  for (uint32_t row = 0; row < b->R; ++row) {  // lr_worker.cc:150-161
    SampleKey sk;
    sk.fgid = 0;
    sk.sid = (int)row;
    for (uint32_t j = b->rowptr[row]; j < b->rowptr[row + 1]; ++j) {
      sk.fid = b->uidx[j];
      all_keys.push_back(sk);
    }
  }
  std::sort(all_keys.begin(), all_keys.end(),
            [](const SampleKey &a, const SampleKey &c) { return a.fid < c.fid; });  // :162
  std::vector<uint32_t> coo(b->NNZ);
  for (size_t j = 0; j < all_keys.size(); ++j) coo[j] = (uint32_t)all_keys[j].sid;
  XF_HIP(hipMalloc((void **)&b->d_ref_coo, (size_t)b->NNZ * 4));
  XF_HIP(hipMemcpyAsync(b->d_ref_coo, coo.data(), (size_t)b->NNZ * 4, hipMemcpyHostToDevice, s));
  XF_HIP(hipStreamSynchronize(s));
  return XF_OK;
}
}  // namespace xf
