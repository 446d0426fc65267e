// xf_cells_fwd.hip — the LR forward over cells (gfx950).
//
// Replaces (paths relative to /root/reference):
//   LRWorker::calculate_loss           src/model/lr/lr_worker.cc:121-143  (k_lr_fwd_cells,
//                                                                          k_lr_finalize_cells)
// Layout and rationale: xf_cells.h.  Numerics: a row's sum is accumulated in fp64 (LDS atomics)
// and rounded to fp32 once, where the reference holds an fp32 value — exact, hence independent
// of the order the atomics land in (xf_cells_grad.hip has the bound).  No MFMA.
#include "xf_cells_impl.h"

namespace {
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
}  // namespace

namespace xf {

int cells_lr_forward(const xf_cells *c, const float *d_w, const int32_t *d_labels,
                     double *d_partial, float *d_loss, float *d_pctr, hipStream_t s,
                     const xf_cells *resume_from) {
  XF_REQUIRE(c && d_w && d_partial && (d_loss || d_pctr), "cells_lr_forward: null argument");
  if (c->R == 0) return XF_OK;
  int acc = 0;
  const xf_cells *first = c;
  if (resume_from) {  // d_partial holds the sums of the segments before it (an earlier call)
    first = resume_from;
    acc = 1;
  }
  for (const xf_cells *q = first; q; q = q->next, acc = 1)
    hipLaunchKernelGGL(k_lr_fwd_cells, dim3(q->nwin * q->G), dim3(kFwdBlock), 0, s, q->fwd_entries(),
                       q->cellptr, q->blk_cell, q->nchunk, q->W, q->G,
                       d_w + (size_t)q->chunk0 * kChunk, d_partial, acc);
  hipLaunchKernelGGL(k_lr_finalize_cells,
                     dim3((unsigned)(((size_t)c->R * 4 + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                     d_partial, d_labels, c->R, c->W, c->G, d_loss, d_pctr);
  XF_HIP(hipGetLastError());
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
