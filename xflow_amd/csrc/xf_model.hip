// xf_model.hip — LR / FM forward and gradient kernels (gfx950) and the fused per-minibatch
// step built from them and the table kernels.
//
// Replaces (paths relative to /root/reference):
//   LRWorker::calculate_loss      src/model/lr/lr_worker.cc:121-143
//   LRWorker::calculate_gradient  src/model/lr/lr_worker.cc:100-119
//   LRWorker::update              src/model/lr/lr_worker.cc:145-177
//   LRWorker::calculate_pctr      src/model/lr/lr_worker.cc:25-71
//   FMWorker::calculate_loss      src/model/fm/fm_worker.cc:159-202
//   FMWorker::calculate_gradient  src/model/fm/fm_worker.cc:126-157
//   FMWorker::update              src/model/fm/fm_worker.cc:204-245
//   Base::sigmoid                 src/base/base.h:54-63
//
// This is a gather / segmented-reduce path: HBM- and L2-bound integer indexing plus a few
// flops per byte.  No MFMA.  Forward = one 64-lane wavefront (or a 16-lane group for short
// rows) per example: coalesced loads of the row's uidx[], gathers of w_u[uidx] from the
// compact pulled-weight array (U floats, L2/Infinity-Cache resident), butterfly reduction
// with __shfl_xor.  Gradient = one lane per unique key walking its occurrence list
// (coalesced across neighbouring keys), gathers of loss[row] from an R-float array that
// lives in L2; keys with long occurrence lists (power-law heads) take a wave-per-key path.
// Sums are accumulated in fp64 so the result does not depend on lane assignment or
// occurrence order (the reference's own order is unspecified: std::sort, lr_worker.cc:162),
// then rounded to fp32 where the reference holds an fp32 value.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "xf_batch.h"
#include "xf_cells.h"
#include "xf_common.h"
#include "xf_device.h"
#include "xf_scratch.h"

namespace xf {
const TableDev &table_dev(const xf_table *t);
int table_dim(const xf_table *t);
uint64_t table_uid(const xf_table *t);
uint64_t table_epoch(const xf_table *t);
int table_records(xf_table *t, size_t row_bytes, uint64_t tag, void **rec, uint64_t *gen);
uint64_t table_writes(const xf_table *t);
void table_note_write(xf_table *t);
void table_records_all_set(xf_table *t, const uint64_t key[5]);
bool table_records_all_is(const xf_table *t, const uint64_t key[5]);
int ensure_cells(xf_batch *b, xf_table *t, hipStream_t s);
int batch_compile_local_dev(xf_batch **out, xf_table *t, const uint64_t *d_keys,
                            const uint32_t *d_rowptr, const int32_t *d_labels, uint32_t R,
                            uint32_t NNZ, int retain_keys, void *stream, KbDeferred **defer);
int cells_lr_grad_update(const xf_cells *c, const xf_table *t, const float *d_loss, float *d_g,
                         hipStream_t s);
int gather_f32(const float *src, const uint32_t *rows, size_t n, float *dst, hipStream_t s);
int batch_sorted_uidx(xf_batch *b, hipStream_t s);
int batch_reference_coo(xf_batch *b, hipStream_t s);
}  // namespace xf

namespace {

constexpr int kBlock = 256;
inline hipStream_t S(void *s) { return (hipStream_t)s; }

using xf::div_by_rows;  // xf_device.h

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, G);
  return v;
}

// ------------------------------------------------------------------ LR forward (a5, a6)
// loss[r] = sigmoid(sum_{j in row r} w_u[uidx[j]]) - label[r]
template <int G>
__global__ void __launch_bounds__(kBlock)
k_lr_forward(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ uidx,
             const float *__restrict__ wu, const int32_t *__restrict__ labels, uint32_t R,
             float *__restrict__ loss, float *__restrict__ pctr) {
  const uint32_t lane = threadIdx.x % G;
  const uint32_t ngroups = gridDim.x * (kBlock / G);
  for (uint32_t r = blockIdx.x * (kBlock / G) + threadIdx.x / G; r < R; r += ngroups) {
    const uint32_t b = rowptr[r], e = rowptr[r + 1];
    double acc = 0.0;
    uint32_t j = b + lane;
    // 4 independent gathers in flight per lane
    for (; j + 3 * G < e; j += 4 * G) {
      const uint32_t i0 = uidx[j], i1 = uidx[j + G], i2 = uidx[j + 2 * G], i3 = uidx[j + 3 * G];
      const float w0 = wu[i0], w1 = wu[i1], w2 = wu[i2], w3 = wu[i3];
      acc += ((double)w0 + (double)w1) + ((double)w2 + (double)w3);
    }
    for (; j < e; j += G) acc += (double)wu[uidx[j]];
    acc = group_sum<G>(acc);
    if (lane == 0) {
      const float p = xf::sigmoid_ref((float)acc);
      if (pctr) pctr[r] = p;
      loss[r] = p - (float)labels[r];  // lr_worker.cc:141
    }
  }
}

// LR forward for large minibatches.  The compact weight array w_u (U floats, 25 MB at the
// config-2 shape) does not fit an XCD's 4 MiB L2, so the plain kernel's gathers are served by
// the Infinity Cache at fabric rate (1e7 gathers: 155 us).  The batch therefore carries a
// panel-major copy of the CSR: the uidx space is cut into P key-range panels (P a multiple of
// 8) so that one panel's slice of w_u fits an L2, and the partial sums of a row over the
// panels are added in panel order by k_lr_finalize (deterministic).
// A workgroup takes a run of (panel,row) cells holding
// <= XF_TILE_NNZ nonzeros, gathers w_u[pidx[j]] for all of them at once into LDS (one
// coalesced index read, every lane with independent gathers in flight), then one lane per
// cell adds the cell's short run in fp64.  Workgroup b takes the (b/8)-th tile of the panels p
// with p % 8 == b % 8 (observed placement: XCD b % 8), so an XCD's L2 holds its panels'
// w_u slices.
__global__ void __launch_bounds__(kBlock)
k_lr_forward_tiled(const uint32_t *__restrict__ tile_ptr,
                   const uint32_t *__restrict__ panel_first, uint32_t P,
                   const uint32_t *__restrict__ pptr, const uint32_t *__restrict__ pidx,
                   const float *__restrict__ wu, uint32_t R, double *__restrict__ partial,
                   uint32_t flat_tiles) {
  __shared__ float vals[XF_TILE_NNZ];
  __shared__ uint32_t sp[XF_TILE_KEYS + 2];
  __shared__ double red[kBlock / 64];
  const uint32_t tid = threadIdx.x;
  {
    // flat_tiles != 0: w_u as a whole fits an XCD's L2, tiles are taken in order by any XCD
    // (a power-law batch puts a fifth of its nonzeros into one panel: pinning panels to XCDs
    // would leave the other seven waiting for that one)
    uint32_t q = blockIdx.x >> 3, tile = flat_tiles ? blockIdx.x : 0xFFFFFFFFu;
    for (uint32_t p = blockIdx.x & 7u; !flat_tiles && p < P; p += 8) {  // <= P/8 iterations
      const uint32_t first = panel_first[p], cnt = panel_first[p + 1] - first;
      if (q < cnt) {
        tile = first + q;
        break;
      }
      q -= cnt;
    }
    if (tile == 0xFFFFFFFFu) return;
    const uint32_t sa = tile_ptr[tile], sb = tile_ptr[tile + 1], ns = sb - sa;
    const uint32_t p = sa / (R + 1), r0 = sa - p * (R + 1);
    const uint32_t j0 = pptr[sa], j1 = pptr[sb];
    if (j1 - j0 > XF_TILE_NNZ) {  // one oversized cell (a row with > XF_TILE_NNZ nonzeros)
      double acc = 0.0;
      for (uint32_t j = j0 + tid; j < j1; j += kBlock) acc += (double)wu[pidx[j]];
      acc = group_sum<64>(acc);
      if ((tid & 63) == 0) red[tid >> 6] = acc;
      __syncthreads();
      if (tid == 0) {
        double s = 0.0;
        for (int k = 0; k < kBlock / 64; ++k) s += red[k];
        partial[(size_t)p * R + r0] = s;
      }
      return;
    }
    for (uint32_t k = tid; k <= ns; k += kBlock) sp[k] = pptr[sa + k] - j0;
    for (uint32_t j = j0 + tid; j < j1; j += kBlock) vals[j - j0] = wu[pidx[j]];
    __syncthreads();
    for (uint32_t k = tid; k < ns; k += kBlock) {
      const uint32_t r = r0 + k;
      if (r < R) {  // r == R is the empty cell between two panels
        double acc = 0.0;
        for (uint32_t j = sp[k]; j < sp[k + 1]; ++j) acc += (double)vals[j];
        partial[(size_t)p * R + r] = acc;
      }
    }
  }
}

__global__ void __launch_bounds__(kBlock)
k_lr_finalize(const double *__restrict__ partial, const int32_t *__restrict__ labels,
              uint32_t R, uint32_t P, float *__restrict__ loss, float *__restrict__ pctr) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  double acc = 0.0;
  for (uint32_t p = 0; p < P; ++p) acc += partial[(size_t)p * R + r];
  const float pr = xf::sigmoid_ref((float)acc);
  if (pctr) pctr[r] = pr;
  loss[r] = pr - (float)labels[r];
}

// ------------------------------------------------------------------ LR gradient (a7)
// g[u] = (sum_{j in seg u} loss[coo_row[j]]) / R   (divide in double: lr_worker.cc:117)
__global__ void __launch_bounds__(kBlock)
k_lr_grad(const uint32_t *__restrict__ segptr, const uint32_t *__restrict__ coo_row,
          const float *__restrict__ loss, uint32_t U, uint32_t R, float *__restrict__ g) {
  const uint32_t stride = gridDim.x * blockDim.x;
  for (uint32_t u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += stride) {
    const uint32_t b = segptr[u], e = segptr[u + 1];
    if (e - b > XF_HEAVY_SEG) continue;  // wave-per-key path below
    double acc = 0.0;
    for (uint32_t j = b; j < e; ++j) acc += (double)loss[coo_row[j]];
    g[u] = (float)((double)(float)acc / (1.0 * R));
  }
}

__global__ void __launch_bounds__(kBlock)
k_lr_grad_heavy(const uint32_t *__restrict__ heavy, uint32_t H,
                const uint32_t *__restrict__ segptr, const uint32_t *__restrict__ coo_row,
                const float *__restrict__ loss, uint32_t R, float *__restrict__ g) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nwaves = gridDim.x * (kBlock / 64);
  for (uint32_t h = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; h < H; h += nwaves) {
    const uint32_t u = heavy[h];
    const uint32_t b = segptr[u], e = segptr[u + 1];
    double acc = 0.0;
    for (uint32_t j = b + lane; j < e; j += 64) acc += (double)loss[coo_row[j]];
    acc = group_sum<64>(acc);
    if (lane == 0) g[u] = (float)((double)(float)acc / (1.0 * R));
  }
}

// LDS-tiled gradient, optionally fused with the Push for a table that lives on the same GPU
// (single shard): g then never round-trips through HBM (it is still stored, once, for the
// parity hook).  The per-key kernel above walks a key's occurrences with a chain of dependent
// global loads (coo_row[j] -> loss[row]), and a wavefront waits for its longest chain.  Here a workgroup takes a tile of consecutive keys
// whose occurrence lists total <= XF_TILE_NNZ entries: phase 1 stages loss[coo_row[j]] for
// the whole tile into LDS with one coalesced read of coo_row[] and one round of gathers, all
// lanes busy; phase 2 sums each key's (short) run out of LDS in fp64 and, when UPDATE, applies
// the optimizer step to the key's state row, whose words were requested before the sums.
// kKeysPerThread keys' state rides in registers across the barrier.  The kernel waits on
// memory, so what matters is how many tiles a CU keeps in flight: __launch_bounds__(256, 5)
// holds the allocation at 96 VGPRs = 5 workgroups per CU (124 -> 111 us on the config-2
// shape); 6+ need spills and 512/1024-thread tiles were slower (131 / 152 us).
constexpr int kKeysPerThread = (XF_GRAD_TILE_KEYS + kBlock - 1) / kBlock;

template <int OPT, bool UPDATE>
__global__ void __launch_bounds__(kBlock, 5)
k_lr_grad_tiled(xf::TableDev T, const uint32_t *__restrict__ tile_ptr, uint32_t ntiles,
                const uint32_t *__restrict__ segptr, const uint32_t *__restrict__ coo_row,
                const float *__restrict__ loss, const uint32_t *__restrict__ slots,
                const float *__restrict__ wu /* pulled weights == current w, or null */,
                uint32_t R, float *__restrict__ g_out) {
  __shared__ float vals[XF_GRAD_TILE_NNZ];
  __shared__ uint32_t sp[XF_GRAD_TILE_KEYS + 1];
  const uint32_t tid = threadIdx.x;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t ua = tile_ptr[tile], ub = tile_ptr[tile + 1];
    const uint32_t nk = ub - ua;
    const uint32_t j0 = segptr[ua], j1 = segptr[ub];
    if (nk == 1 && j1 - j0 > XF_HEAVY_SEG) continue;  // a heavy key: wave-per-key path
    // Three rounds of loads, each issued as one batch so that a tile costs three memory round
    // trips, not one per dependent pointer: (1) the tile's bounds above; (2) everything
    // addressed by them — occurrence rows, key offsets, state-row numbers, pulled weights;
    // (3) everything addressed by round 2 — the loss gathers and the {n,z} words.
    constexpr int kOccPerThread = (XF_GRAD_TILE_NNZ + kBlock - 1) / kBlock;
    uint32_t orow[kOccPerThread];
#pragma unroll
    for (int q = 0; q < kOccPerThread; ++q) {
      const uint32_t j = j0 + tid + q * kBlock;
      orow[q] = j < j1 ? coo_row[j] : 0u;
    }
    uint32_t slot[kKeysPerThread];
    float w[kKeysPerThread], nn[kKeysPerThread], z[kKeysPerThread];
    if (UPDATE) {
#pragma unroll
      for (int q = 0; q < kKeysPerThread; ++q) {
        const uint32_t k = tid + q * kBlock;
        slot[q] = k < nk ? slots[ua + k] : 0u;
        if (wu) w[q] = k < nk ? wu[ua + k] : 0.0f;  // the Pull's copy is still current
      }
    }
    for (uint32_t k = tid; k <= nk; k += kBlock) sp[k] = segptr[ua + k] - j0;
#pragma unroll
    for (int q = 0; q < kOccPerThread; ++q) {
      const uint32_t j = j0 + tid + q * kBlock;
      if (j < j1) vals[j - j0] = loss[orow[q]];
    }
    if (UPDATE) {
#pragma unroll
      for (int q = 0; q < kKeysPerThread; ++q) {
        const uint32_t k = tid + q * kBlock;
        if (k < nk) {
          if (!wu) w[q] = T.w[slot[q]];
          if (OPT == XF_OPT_FTRL) {
            xf::load_nz(T, slot[q], nn[q], z[q]);
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kKeysPerThread; ++q) {
      const uint32_t k = tid + q * kBlock;
      if (k < nk) {
        double acc = 0.0;
        for (uint32_t j = sp[k]; j < sp[k + 1]; ++j) acc += (double)vals[j];
        const float g = (float)((double)(float)acc / (1.0 * R));
        g_out[ua + k] = g;
        if (UPDATE) {
          if (OPT == XF_OPT_FTRL) {
            xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g, w[q], nn[q], z[q]);
            T.w[slot[q]] = w[q];
            xf::store_nz(T, slot[q], nn[q], z[q]);
          } else {
            T.w[slot[q]] = xf::sgd_step(T.lr, g, w[q]);
          }
        }
      }
    }
    __syncthreads();
  }
}

// ---- heavy keys in chunks ------------------------------------------------------------------
// A power-law head key can own millions of a minibatch's occurrences; one wavefront walking
// them takes milliseconds.  Every heavy key is cut into chunks of XF_TILE_NNZ occurrences
// (heavy_chunk_ptr, built with the batch), one workgroup reduces one chunk, and a second
// small kernel adds a key's chunk sums in chunk order (deterministic) and applies the step.
__device__ __forceinline__ uint32_t heavy_of_chunk(const uint32_t *__restrict__ hch, uint32_t H,
                                                   uint32_t c) {
  uint32_t lo = 0, hi = H;  // largest h with hch[h] <= c
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (hch[mid] <= c) lo = mid;
    else
      hi = mid;
  }
  return lo;
}

// the same by a whole wavefront: 64 probes per round (three dependent loads for any H up to
// 2^18 instead of log2 H; every lane returns the answer)
__device__ __forceinline__ uint32_t heavy_of_chunk_wave(const uint32_t *__restrict__ hch,
                                                        uint32_t H, uint32_t c) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t lo = 0, hi = H;
  while (hi - lo > 1) {
    const uint32_t step = (hi - lo + 63u) / 64u;
    const uint32_t idx = lo + lane * step;
    const bool ok = idx < hi && hch[idx] <= c;  // a prefix of the lanes (hch ascends; lane 0)
    const unsigned long long m = __ballot(ok);
    const uint32_t top = 63u - (uint32_t)__builtin_clzll(m | 1ull);
    lo += top * step;
    hi = min(hi, lo + step);
  }
  return lo;
}

__device__ __forceinline__ double block_sum(double v, double *red) {
  v = group_sum<64>(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
  for (int k = 0; k < kBlock / 64; ++k) s += red[k];
  __syncthreads();
  return s;
}

__global__ void __launch_bounds__(kBlock)
k_lr_heavy_partial(const uint32_t *__restrict__ heavy, const uint32_t *__restrict__ hch,
                   uint32_t H, const uint32_t *__restrict__ segptr,
                   const uint32_t *__restrict__ coo_row, const float *__restrict__ loss,
                   double *__restrict__ partial) {
  __shared__ double red[kBlock / 64];
  const uint32_t c = blockIdx.x;
  const uint32_t h = heavy_of_chunk(hch, H, c);
  const uint32_t u = heavy[h];
  const uint32_t b = segptr[u] + (c - hch[h]) * XF_TILE_NNZ;
  const uint32_t e = min(segptr[u + 1], b + XF_TILE_NNZ);
  double acc = 0.0;
  for (uint32_t j = b + threadIdx.x; j < e; j += kBlock) acc += (double)loss[coo_row[j]];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partial[c] = acc;
}

template <int OPT, bool UPDATE>
__global__ void __launch_bounds__(kBlock)
k_lr_heavy_finish(xf::TableDev T, const uint32_t *__restrict__ heavy,
                  const uint32_t *__restrict__ hch, uint32_t H,
                  const double *__restrict__ partial, const uint32_t *__restrict__ slots,
                  uint32_t R, float *__restrict__ g_out) {
  // one wavefront per heavy key: the head of a power-law batch owns hundreds of chunks, and
  // one lane adding them one dependent load after the other took 70 us on its own.  Lanes
  // take the chunks round-robin, then a fixed butterfly: the same association every run.
  const uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  if (h >= H) return;  // wave-uniform
  double acc = 0.0;
  for (uint32_t c = hch[h] + lane; c < hch[h + 1]; c += 64) acc += partial[c];
  acc = group_sum<64>(acc);
  if (lane != 0) return;
  const uint32_t u = heavy[h];
  const float g = (float)((double)(float)acc / (1.0 * R));
  g_out[u] = g;
  if (UPDATE) {
    const uint32_t slot = slots[u];
    if (OPT == XF_OPT_FTRL) {
      float w = T.w[slot], nn, z;
      xf::load_nz(T, slot, nn, z);
      xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g, w, nn, z);
      T.w[slot] = w;
      xf::store_nz(T, slot, nn, z);
    } else {
      T.w[slot] = xf::sgd_step(T.lr, g, T.w[slot]);
    }
  }
}

// FM: per chunk and factor kk the sum of loss*(v_sum - v[u,kk]) (fp32 products, fm_worker.cc:141),
// plus the plain loss sum in column k.  partial is [chunk][k+1].  A thread is (slice of the
// chunk's occurrences, column): its column's factor in a register, the occurrences read from
// LDS as broadcasts, one LDS round to add the slices.  (One wavefront per column with a
// butterfly per column cost ~50 instructions per (chunk, column) whatever the chunk's length —
// and most heavy keys of a power-law minibatch have 65..300 occurrences: 217 us at k = 64.)
// A key of ONE chunk is finished here (gw, gv: what k_fm_heavy_finish would write).
__global__ void __launch_bounds__(kBlock)
k_fm_heavy_partial(const uint32_t *__restrict__ heavy, const uint32_t *__restrict__ hch,
                   uint32_t H, const uint32_t *__restrict__ segptr,
                   const uint32_t *__restrict__ coo_row, const float *__restrict__ loss,
                   const float *__restrict__ vsum, const float *__restrict__ vu, int k,
                   double *__restrict__ partial, uint32_t R, float *__restrict__ gw,
                   float *__restrict__ gv) {
#pragma clang fp contract(off)
  __shared__ float lv[XF_TILE_NNZ], sv[XF_TILE_NNZ];
  __shared__ double red[kBlock];
  const uint32_t c = blockIdx.x;
  const uint32_t h = heavy_of_chunk_wave(hch, H, c);
  const uint32_t u = heavy[h];
  const uint32_t b = segptr[u] + (c - hch[h]) * XF_TILE_NNZ;
  const uint32_t e = min(segptr[u + 1], b + XF_TILE_NNZ);
  const uint32_t n = e - b;
  const uint32_t ncol = (uint32_t)k + 1u, nsl = kBlock / ncol;  // k <= XF_HEAVY_KMAX: nsl >= 3
  const uint32_t col = threadIdx.x % ncol, sl = threadIdx.x / ncol;
  const bool fac = col < (uint32_t)k;
  const float v = (sl < nsl && fac) ? vu[(size_t)u * k + col] : 0.0f;
  for (uint32_t j = threadIdx.x; j < n; j += kBlock) {
    const uint32_t sid = coo_row[b + j];
    lv[j] = loss[sid];
    sv[j] = vsum[sid];
  }
  __syncthreads();
  double acc = 0.0;
  if (sl < nsl) {
    // four occurrences per round, four independent sums (one dependent chain of LDS reads and
    // fp64 adds per occurrence was the kernel's time), joined in a fixed order
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    uint32_t j = sl;
    for (; j + 3 * nsl < n; j += 4 * nsl) {
      const float l0 = lv[j], l1 = lv[j + nsl], l2 = lv[j + 2 * nsl], l3 = lv[j + 3 * nsl];
      if (fac) {
        const float s0 = sv[j], s1 = sv[j + nsl], s2 = sv[j + 2 * nsl], s3 = sv[j + 3 * nsl];
        a0 += (double)(l0 * (s0 - v));
        a1 += (double)(l1 * (s1 - v));
        a2 += (double)(l2 * (s2 - v));
        a3 += (double)(l3 * (s3 - v));
      } else {
        a0 += (double)l0;
        a1 += (double)l1;
        a2 += (double)l2;
        a3 += (double)l3;
      }
    }
    for (; j < n; j += nsl) a0 += fac ? (double)(lv[j] * (sv[j] - v)) : (double)lv[j];
    acc = (a0 + a1) + (a2 + a3);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (sl != 0) return;
  for (uint32_t q = 1; q < nsl; ++q) acc += red[q * ncol + col];
  if (hch[h + 1] - hch[h] == 1) {  // the key's only chunk
    if (fac) gv[(size_t)u * k + col] = (float)((double)(float)acc / (1.0 * R));
    else
      gw[u] = (float)((double)(float)(acc * (double)k) / (1.0 * R));
  } else {
    partial[(size_t)c * ncol + col] = acc;
  }
}

__global__ void __launch_bounds__(kBlock)
k_fm_heavy_finish(const uint32_t *__restrict__ heavy, const uint32_t *__restrict__ hch,
                  uint32_t H, const double *__restrict__ partial, uint32_t R, int k,
                  float *__restrict__ gw, float *__restrict__ gv) {
  // one workgroup per heavy key; a thread is (slice of the key's chunks, column), the columns
  // being the k factors and the loss sum: partial[chunk][column] is read coalesced, nothing is
  // shuffled (one wavefront per key with a butterfly per factor took 281 us at k = 64 for the
  // 2e4 heavy keys of a Zipf minibatch, most of them one chunk long)
  __shared__ double red[kBlock];
  const uint32_t h = blockIdx.x;
  const uint32_t ncol = (uint32_t)k + 1u, nsl = kBlock / ncol;  // k <= XF_HEAVY_KMAX: nsl >= 3
  const uint32_t col = threadIdx.x % ncol, sl = threadIdx.x / ncol;
  const uint32_t c0 = hch[h], c1 = hch[h + 1], u = heavy[h];
  if (c1 - c0 == 1) return;  // finished by k_fm_heavy_partial (workgroup-uniform)
  double acc = 0.0;
  if (sl < nsl) {
    // (the head key of a power-law minibatch owns hundreds of chunks: eight loads in flight —
    // one dependent load after the other was 78 us for that one key — added in chunk order)
    uint32_t c = c0 + sl;
    for (; c + 7 * nsl < c1; c += 8 * nsl) {
      double p[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = partial[(size_t)(c + i * nsl) * ncol + col];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += p[i];
    }
    for (; c < c1; c += nsl) acc += partial[(size_t)c * ncol + col];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (sl != 0) return;
  for (uint32_t q = 1; q < nsl; ++q) acc += red[q * ncol + col];
  if (col < (uint32_t)k) gv[(size_t)u * k + col] = (float)((double)(float)acc / (1.0 * R));
  else
    gw[u] = (float)((double)(float)(acc * (double)k) / (1.0 * R));
}

// the optimizer step for a listed subset of keys (the heavy ones)
template <int OPT>
__global__ void __launch_bounds__(kBlock)
k_update_listed(xf::TableDev T, const uint32_t *__restrict__ list, uint32_t H,
                const uint32_t *__restrict__ slots, const float *__restrict__ g) {
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  const uint32_t u = list[h], slot = slots[u];
  if (OPT == XF_OPT_FTRL) {
    float w = T.w[slot], nn, z;
      xf::load_nz(T, slot, nn, z);
    xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, g[u], w, nn, z);
    T.w[slot] = w;
    xf::store_nz(T, slot, nn, z);
  } else {
    T.w[slot] = xf::sgd_step(T.lr, g[u], T.w[slot]);
  }
}

// ------------------------------------------------------------------ FM forward (a11)
// Reference form, fm_worker.cc:159-202: v_sum and v_pow_sum are SCALARS per row pooled over
// all k factors; v_y = v_sum^2 - v_pow_sum (no 1/2); loss = sigmoid(wx + v_y) - label.
// One wavefront per example; lane e walks the row's (nnz, factor) elements so every v_u
// row is read as one contiguous k*4-byte segment.
template <int K>
__global__ void __launch_bounds__(kBlock)
k_fm_forward(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ uidx,
             const float *__restrict__ wu, const float *__restrict__ vu, int k_rt,
             const int32_t *__restrict__ labels, uint32_t R, float *__restrict__ loss,
             float *__restrict__ pctr, float *__restrict__ vsum_out) {
#pragma clang fp contract(off)
  const uint32_t k = K > 0 ? (uint32_t)K : (uint32_t)k_rt;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nwaves = gridDim.x * (kBlock / 64);
  for (uint32_t r = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; r < R; r += nwaves) {
    const uint32_t b = rowptr[r], e = rowptr[r + 1];
    const uint32_t nel = (e - b) * k;
    double wx = 0.0, vs = 0.0, vp = 0.0;
    if (K > 0 && 64 % (K > 0 ? K : 1) == 0) {
      // K divides the wavefront: lane = (nnz slot, factor); 4 nnz slots in flight per lane
      constexpr uint32_t kG = K > 0 ? 64 / (K > 0 ? K : 1) : 1;  // nnz per wave pass
      const uint32_t kk = lane % (K > 0 ? K : 1), sub = lane / (K > 0 ? K : 1);
      const uint32_t n = e - b;
      for (uint32_t j0 = sub; j0 < n; j0 += kG * 4) {
        uint32_t ui[4];
        float vv[4], ww[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ui[i] = j0 + i * kG < n ? uidx[b + j0 + i * kG] : 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          vv[i] = ui[i] != 0xFFFFFFFFu ? vu[(size_t)ui[i] * K + kk] : 0.0f;
          ww[i] = (ui[i] != 0xFFFFFFFFu && kk == 0) ? wu[ui[i]] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          vs += (double)vv[i];
          vp += (double)(vv[i] * vv[i]);  // fp32 product, as fm_worker.cc:187
          wx += (double)ww[i];
        }
      }
    } else
    for (uint32_t el = lane; el < nel; el += 64) {
      const uint32_t jj = el / k, kk = el - jj * k;
      const uint32_t ui = uidx[b + jj];
      const float vv = vu[(size_t)ui * k + kk];
      vs += (double)vv;
      vp += (double)(vv * vv);  // fp32 product, as fm_worker.cc:187
      if (kk == 0) wx += (double)wu[ui];
    }
    wx = group_sum<64>(wx);
    vs = group_sum<64>(vs);
    vp = group_sum<64>(vp);
    if (lane == 0) {
      const float vsf = (float)vs, vpf = (float)vp;
      const float vy = vsf * vsf - vpf;                       // :194-195
      const float p = xf::sigmoid_ref((float)wx + vy);        // :199
      if (pctr) pctr[r] = p;
      loss[r] = p - (float)labels[r];
      vsum_out[r] = vsf;
    }
  }
}

// ------------------------------------------------ FM forward over per-key scalars (a11)
// The reference pools its second-order sums over ALL k factors (fm_worker.cc:178-192):
//   v_sum[row] = sum_k sum_nnz v[u,k],   v_pow_sum[row] = sum_k sum_nnz v[u,k]^2
// so both are sums over the row's keys of per-key scalars a[u] = sum_k v[u,k] and
// b[u] = sum_k v[u,k]^2 (fp32 products, as :187).  The v-row gather of the Pull reads every
// row anyway: it forms (a, b) on the way — exactly, in fp64 — and stores them with the pulled
// w as one 32-byte record per key.  The per-nonzero pass then gathers 32 bytes per nonzero
// from a U x 32 B table instead of a k x 4-byte row from U x k x 4 B plus w_u from a third
// array: at k = 16, 1e7 nonzeros, 383 -> see DESIGN.md 8 (the gather count, not the bytes, is
// what costs: tools/exp/gather_widths.py).
struct __attribute__((aligned(32))) FmKey {
  double a, b;
  float w;
  float pad[3];
};

template <int DIM4>  // floats per factor row / 4; a power of two <= 16
__global__ void __launch_bounds__(kBlock)
k_fm_gather_scalars(const float4 *__restrict__ tv, const uint32_t *__restrict__ rows,
                    const float *__restrict__ wu, const uint32_t *__restrict__ wrows, size_t n,
                    float4 *__restrict__ vu, FmKey *__restrict__ ks,
                    const uint32_t *__restrict__ orows) {
#pragma clang fp contract(off)
  const size_t total = n * DIM4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const unsigned lane = threadIdx.x & 63u;
  // wave-uniform trip count (the shuffles below need every lane); a key's DIM4 lanes are
  // consecutive and never straddle a wavefront
  for (size_t e0 = (size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u); e0 < total;
       e0 += stride) {
    const size_t e = e0 + lane;
    const bool on = e < total;
    const size_t i = on ? e / DIM4 : 0, j = on ? e % DIM4 : 0;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
      v = tv[(size_t)(rows ? rows[i] : (uint32_t)i) * DIM4 + j];  // rows == null: tv is dense
      if (vu) vu[e] = v;
    }
    double a = (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    double b = (double)(v.x * v.x) + (double)(v.y * v.y) + (double)(v.z * v.z) +
               (double)(v.w * v.w);
#pragma unroll
    for (int off = DIM4 / 2; off > 0; off >>= 1) {
      a += __shfl_xor(a, off, DIM4);
      b += __shfl_xor(b, off, DIM4);
    }
    if (on && j == 0) {
      FmKey q;
      q.a = a;
      q.b = b;
      q.w = wu[wrows ? wrows[i] : (uint32_t)i];  // wrows: wu is the w table's weight column
      q.pad[0] = q.pad[1] = q.pad[2] = 0.f;
      ks[orows ? orows[i] : (uint32_t)i] = q;    // orows: the records sit at the keys' v rows
    }
  }
}

// one wavefront per example, one 32-byte record per nonzero, four in flight per lane
__global__ void __launch_bounds__(kBlock)
k_fm_forward_scalars(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ uidx,
                     const FmKey *__restrict__ ks, const int32_t *__restrict__ labels,
                     uint32_t R, float *__restrict__ loss, float *__restrict__ pctr,
                     float *__restrict__ vsum_out) {
#pragma clang fp contract(off)
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nwaves = gridDim.x * (kBlock / 64);
  for (uint32_t r = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; r < R; r += nwaves) {
    const uint32_t b = rowptr[r], e = rowptr[r + 1];
    double wx = 0.0, vs = 0.0, vp = 0.0;
    for (uint32_t j0 = b + lane; j0 < e; j0 += 64 * 4) {
      uint32_t ui[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ui[i] = j0 + 64 * i < e ? uidx[j0 + 64 * i] : 0xFFFFFFFFu;
      FmKey q[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ui[i] != 0xFFFFFFFFu) q[i] = ks[ui[i]];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ui[i] != 0xFFFFFFFFu) {
          wx += (double)q[i].w;
          vs += q[i].a;
          vp += q[i].b;
        }
    }
    wx = group_sum<64>(wx);
    vs = group_sum<64>(vs);
    vp = group_sum<64>(vp);
    if (lane == 0) {
      const float vsf = (float)vs, vpf = (float)vp;
      const float vy = vsf * vsf - vpf;                       // :194-195
      const float p = xf::sigmoid_ref((float)wx + vy);        // :199
      if (pctr) pctr[r] = p;
      loss[r] = p - (float)labels[r];
      vsum_out[r] = vsf;
    }
  }
}

// ------------------------------------------------------------------ FM gradient (a12)
// fm_worker.cc:126-157: for every factor kk (outer loop) and occurrence:
//   gw[u]    += loss[sid]                        -> gw is k x the LR gradient (:140)
//   gv[u,kk] += loss[sid] * (v_sum[sid] - v[u,kk])
// then both / R.  One lane per (u, kk); neighbouring lanes share the occurrence list.
__global__ void __launch_bounds__(kBlock)
k_fm_grad(const uint32_t *__restrict__ segptr, const uint32_t *__restrict__ coo_row,
          const float *__restrict__ loss, const float *__restrict__ vsum,
          const float *__restrict__ vu, uint32_t U, uint32_t R, int k,
          float *__restrict__ gw, float *__restrict__ gv) {
#pragma clang fp contract(off)
  const size_t total = (size_t)U * k;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t el = (size_t)blockIdx.x * blockDim.x + threadIdx.x; el < total; el += stride) {
    const uint32_t u = (uint32_t)(el / k);
    const uint32_t kk = (uint32_t)(el - (size_t)u * k);
    const uint32_t b = segptr[u], e = segptr[u + 1];
    if (e - b > XF_HEAVY_SEG) continue;
    const float v = vu[el];
    double accw = 0.0, accv = 0.0;
    for (uint32_t j = b; j < e; ++j) {
      const uint32_t sid = coo_row[j];
      const float l = loss[sid];
      accw += (double)l;
      accv += (double)(l * (vsum[sid] - v));
    }
    gv[el] = (float)((double)(float)accv / (1.0 * R));
    if (kk == 0) gw[u] = (float)((double)(float)(accw * (double)k) / (1.0 * R));
  }
}

// LDS-tiled FM gradient, optionally fused with the two Pushes (fm_worker.cc:241-242) when
// both tables live on this GPU.  The per-(key,factor) kernel above gathers loss[sid] and
// v_sum[sid] once per factor — k times too often.  Here a workgroup takes a gradient tile
// (the same key tiles as the LR kernel), stages (loss, v_sum) of every occurrence in LDS once,
// and its lanes walk the tile's (key, factor) pairs: v_u is read coalesced, the occurrence
// runs come out of LDS.  UPDATE: the lane then applies the optimizer step to its coordinate
// of the key's v row (the pulled value IS the current weight: nothing touched the row since
// the Pull) and, for factor 0, to the key's w row.
// REC (with UPDATE, K % 4 == 0): nothing was pulled into scratch — a lane reads its factors
// from the v table's rows (and the key's w from the w table), and after the optimizer step the
// key's lanes leave the forward's record (sum_k v, sum_k v^2, w) of the NEW weights at the
// key's v row: the factors are in registers here, the next forward of any minibatch finds the
// records up to date and no pass over the factor rows precedes it.
// waves per SIMD the register allocation must allow.  SGD: 86 registers unbounded = 5 tiles per
// CU; 80 = 6 tiles is 8 % faster, 64 = 8 tiles spills and is slower.  FTRL (166 registers: the
// (n, z) state of four quads in flight) loses more to spills than it gains (k = 64: 0.84 ->
// 0.93 ms at 4 waves).
#ifndef XF_FMG_SGD_WAVES
#define XF_FMG_SGD_WAVES 6
#define XF_FMG_FTRL_WAVES 1
#endif
template <int OPT, bool UPDATE, int K /* compile-time factor count, 0 = use k_rt */,
          bool REC = false>
__global__ void __launch_bounds__(kBlock, OPT == XF_OPT_SGD ? XF_FMG_SGD_WAVES : XF_FMG_FTRL_WAVES)
k_fm_grad_tiled(xf::TableDev TW, xf::TableDev TV, const uint32_t *__restrict__ tile_ptr,
                uint32_t ntiles, const uint32_t *__restrict__ segptr,
                const uint32_t *__restrict__ coo_row, const float *__restrict__ loss,
                const float *__restrict__ vsum, const float *__restrict__ wu,
                const float *__restrict__ vu, const uint32_t *__restrict__ rows_w,
                const uint32_t *__restrict__ rows_v, uint32_t R, int k_rt,
                float *__restrict__ gw, float *__restrict__ gv, FmKey *__restrict__ rec) {
#pragma clang fp contract(off)
  static_assert(!REC || (UPDATE && K > 0 && K % 4 == 0), "REC needs the fused quad path");
  __shared__ float lv[XF_GRAD_TILE_NNZ], sv[XF_GRAD_TILE_NNZ];
  __shared__ uint32_t sp[XF_GRAD_TILE_KEYS + 1];
  const int k = K > 0 ? K : k_rt;  // a constant for the common factor counts: no divisions
  const uint32_t tid = threadIdx.x;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t ua = tile_ptr[tile], ub = tile_ptr[tile + 1], nk = ub - ua;
    const uint32_t j0 = segptr[ua], j1 = segptr[ub];
    if (nk == 1 && j1 - j0 > XF_HEAVY_SEG) continue;  // heavy key: k_fm_grad_heavy
    if constexpr (K > 0 && K % 4 == 0) {
      // Factor counts that are a multiple of four: a lane's item is FOUR consecutive factors of
      // a key — one 16-byte load of the factors, two of the (n,z) state, one 16-byte store of
      // the weights — so that four times the bytes are in flight per lane and the key's
      // occurrences are read from LDS once per four factors.  (With one factor per lane the
      // kernel moved its 1.03 GB at 2.7 TB/s: 24 wavefronts x 4 loads x 256 B in flight per CU
      // against ~2 us of latency.)
      constexpr uint32_t Q = (uint32_t)(K > 0 ? K : 4) / 4u;
      const uint32_t nq = nk * Q;
      // quads in flight per lane.  With 2048-occurrence tiles FTRL wanted four (its (n, z)
      // state makes a quad 48 bytes of loads); with 256-occurrence tiles the registers are
      // worth more as occupancy: k = 64 + FTRL power-law 0.73 / 0.68 / 0.60 ms at 4 / 2 / 1
      // (166 / 112 / fewer registers)
#ifndef XF_FMG_FTRL_UQ
#define XF_FMG_FTRL_UQ 1
#endif
#ifndef XF_FMG_SGD_UQ
#define XF_FMG_SGD_UQ 2
#endif
      constexpr int kUq = OPT == XF_OPT_FTRL ? XF_FMG_FTRL_UQ : XF_FMG_SGD_UQ;
      // REC: the key's w coordinate is the job of ONE of its quad lanes (the second, when there
      // is one), so that the record leaves as one full 32-byte sector from neighbouring lanes:
      // (a, b) from the first lane, (w, 0, 0, 0) from the w lane.  (Written as 16 + 4 bytes at
      // different times the records cost 150 us of the kernel's 480: partial sectors.)
      constexpr uint32_t kWLane = Q > 1 ? 1u : 0u;
      uint32_t kq[kUq], qq[kUq];
      float4 v[kUq], nzA[kUq], nzB[kUq];
      size_t to[kUq];
      bool on[kUq];
      [[maybe_unused]] bool wl[kUq];
      [[maybe_unused]] uint32_t rw[kUq];
      [[maybe_unused]] float ww[kUq], wn[kUq], wz[kUq];
      // a batch of kUq quads per lane, in two halves: what depends on the tile's bounds only
      // (which quads, their rows) ...
      auto batch_rows = [&](uint32_t e0) {
#pragma unroll
        for (int i = 0; i < kUq; ++i) {
          const uint32_t el = e0 + i * kBlock;
          on[i] = el < nq;
          kq[i] = on[i] ? el / Q : 0;
          qq[i] = on[i] ? el - kq[i] * Q : 0;
        }
#pragma unroll
        for (int i = 0; i < kUq; ++i)
          to[i] = (UPDATE && on[i]) ? (size_t)rows_v[ua + kq[i]] * K + 4 * qq[i] : 0;
        if constexpr (REC) {
#pragma unroll
          for (int i = 0; i < kUq; ++i) {
            wl[i] = on[i] && qq[i] == kWLane;
            rw[i] = wl[i] ? rows_w[ua + kq[i]] : 0;
          }
        }
      };
      // ... and what is addressed by that: the factors, their state, the key's w
      auto batch_state = [&]() {
#pragma unroll
        for (int i = 0; i < kUq; ++i) {
          const float *src = REC ? TV.w + to[i] : vu + (size_t)(ua + kq[i]) * K + 4 * qq[i];
          v[i] = on[i] ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (UPDATE && OPT == XF_OPT_FTRL) {
#pragma unroll
          for (int i = 0; i < kUq; ++i) {
            nzA[i] = nzB[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on[i]) {  // (n0,z0,n1,z1), (n2,z2,n3,z3)
              nzA[i] = *reinterpret_cast<const float4 *>(TV.nz + to[i]);
              nzB[i] = *reinterpret_cast<const float4 *>(TV.nz + to[i] + 2);
            }
          }
        }
        if constexpr (REC) {
#pragma unroll
          for (int i = 0; i < kUq; ++i) {
            ww[i] = wn[i] = wz[i] = 0.f;
            if (wl[i]) {
              ww[i] = TW.w[rw[i]];
              if (OPT == XF_OPT_FTRL) xf::load_nz(TW, rw[i], wn[i], wz[i]);
            }
          }
        }
      };
      // the sums out of LDS, the optimizer steps, the stores
      auto batch_finish = [&]() {
#pragma unroll
        for (int i = 0; i < kUq; ++i) {
          if (!on[i]) continue;
          double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
          [[maybe_unused]] double aw = 0.0;
          for (uint32_t j = sp[kq[i]]; j < sp[kq[i] + 1]; ++j) {
            const float l = lv[j], sj = sv[j];
            if constexpr (REC) aw += (double)l;
            a0 += (double)(l * (sj - v[i].x));
            a1 += (double)(l * (sj - v[i].y));
            a2 += (double)(l * (sj - v[i].z));
            a3 += (double)(l * (sj - v[i].w));
          }
          const float4 g = make_float4(div_by_rows((float)a0, R), div_by_rows((float)a1, R),
                                       div_by_rows((float)a2, R), div_by_rows((float)a3, R));
          if (gv)
            *reinterpret_cast<float4 *>(gv + (size_t)(ua + kq[i]) * K + 4 * qq[i]) = g;
          if (UPDATE) {
            float4 w = v[i];
            if (OPT == XF_OPT_FTRL) {
              xf::ftrl_step(TV.alpha, TV.inv_alpha, TV.beta, TV.lambda1, TV.lambda2, g.x, w.x, nzA[i].x, nzA[i].y);
              xf::ftrl_step(TV.alpha, TV.inv_alpha, TV.beta, TV.lambda1, TV.lambda2, g.y, w.y, nzA[i].z, nzA[i].w);
              xf::ftrl_step(TV.alpha, TV.inv_alpha, TV.beta, TV.lambda1, TV.lambda2, g.z, w.z, nzB[i].x, nzB[i].y);
              xf::ftrl_step(TV.alpha, TV.inv_alpha, TV.beta, TV.lambda1, TV.lambda2, g.w, w.w, nzB[i].z, nzB[i].w);
              *reinterpret_cast<float4 *>(TV.nz + to[i]) = nzA[i];
              *reinterpret_cast<float4 *>(TV.nz + to[i] + 2) = nzB[i];
            } else {
              w.x = xf::sgd_step(TV.lr, g.x, w.x);
              w.y = xf::sgd_step(TV.lr, g.y, w.y);
              w.z = xf::sgd_step(TV.lr, g.z, w.z);
              w.w = xf::sgd_step(TV.lr, g.w, w.w);
            }
            *reinterpret_cast<float4 *>(TV.w + to[i]) = w;
            if constexpr (REC) {  // the same sums, in the same order, as k_fm_gather_scalars
              double ra = (double)w.x + (double)w.y + (double)w.z + (double)w.w;
              double rb = (double)(w.x * w.x) + (double)(w.y * w.y) + (double)(w.z * w.z) +
                          (double)(w.w * w.w);
#pragma unroll
              for (int off = (int)Q / 2; off > 0; off >>= 1) {
                ra += __shfl_xor(ra, off, (int)Q);
                rb += __shfl_xor(rb, off, (int)Q);
              }
              FmKey *rk = &rec[to[i] / K];
              if (qq[i] == 0) *reinterpret_cast<double2 *>(rk) = make_double2(ra, rb);
              if (wl[i]) {  // the key's w coordinate (gw = k x the LR gradient, fm_worker.cc:140)
                const float g1 = div_by_rows((float)(aw * (double)K), R);
                float w1 = ww[i];
                if (OPT == XF_OPT_FTRL) {
                  xf::ftrl_step(TW.alpha, TW.inv_alpha, TW.beta, TW.lambda1, TW.lambda2, g1, w1, wn[i], wz[i]);
                  xf::store_nz(TW, rw[i], wn[i], wz[i]);
                } else {
                  w1 = xf::sgd_step(TW.lr, g1, w1);
                }
                TW.w[rw[i]] = w1;
                *reinterpret_cast<float4 *>(&rk->w) = make_float4(w1, 0.f, 0.f, 0.f);
              }
            }
          }
        }
      };
      // The tile is a chain of dependent memory round trips (bounds -> rows / occurrence rows
      // -> factors / (loss, v_sum) -> stores) and a CU holds only eight tiles: the first
      // batch's rows and state are requested next to the occurrences, ahead of the barrier, so
      // that the two chains run side by side instead of one after the other.
      batch_rows(tid);
      for (uint32_t q = tid; q <= nk; q += kBlock) sp[q] = segptr[ua + q] - j0;
      uint32_t sid0 = 0;
      const bool occ0 = j0 + tid < j1;
      if (occ0) sid0 = coo_row[j0 + tid];
      batch_state();
      if (occ0) {
        lv[tid] = loss[sid0];
        sv[tid] = vsum[sid0];
      }
      for (uint32_t j = j0 + tid + kBlock; j < j1; j += kBlock) {  // (tiles beyond one pass)
        const uint32_t sid = coo_row[j];
        lv[j - j0] = loss[sid];
        sv[j - j0] = vsum[sid];
      }
      __syncthreads();
      batch_finish();
      for (uint32_t e0 = tid + kBlock * kUq; e0 < nq; e0 += kBlock * kUq) {
        batch_rows(e0);
        batch_state();
        batch_finish();
      }
    } else {
    for (uint32_t q = tid; q <= nk; q += kBlock) sp[q] = segptr[ua + q] - j0;
    for (uint32_t j = j0 + tid; j < j1; j += kBlock) {
      const uint32_t sid = coo_row[j];
      lv[j - j0] = loss[sid];
      sv[j - j0] = vsum[sid];
    }
    __syncthreads();
    const uint32_t nel = nk * (uint32_t)k;
    // (key,factor) items in flight per lane.  Everything an item reads from HBM — its factor,
    // its state row, the w row of the factor-0 lanes — is requested before the first
    // occurrence loop: with the state loads behind each item's loop a lane sat through three
    // HBM latencies per two items (k = 16, SGD: 508 us for 1.7 TB/s of state traffic).
    constexpr int kUn = 4;
    for (uint32_t el0 = tid; el0 < nel; el0 += kBlock * kUn) {
      uint32_t kq[kUn], kk[kUn];
      float v[kUn], vn[kUn], vz[kUn];
      size_t to[kUn];
      bool on[kUn];
#pragma unroll
      for (int i = 0; i < kUn; ++i) {
        const uint32_t el = el0 + i * kBlock;
        on[i] = el < nel;
        kq[i] = on[i] ? el / (uint32_t)k : 0;
        kk[i] = on[i] ? el - kq[i] * (uint32_t)k : 0;
      }
#pragma unroll
      for (int i = 0; i < kUn; ++i) {
        v[i] = on[i] ? vu[(size_t)(ua + kq[i]) * k + kk[i]] : 0.0f;
        to[i] = (UPDATE && on[i]) ? (size_t)rows_v[ua + kq[i]] * k + kk[i] : 0;
      }
      if (UPDATE && OPT == XF_OPT_FTRL) {
#pragma unroll
        for (int i = 0; i < kUn; ++i) {
          vn[i] = vz[i] = 0.0f;
          if (on[i]) xf::load_nz(TV, to[i], vn[i], vz[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < kUn; ++i) {
        if (!on[i]) continue;
        const size_t o = (size_t)(ua + kq[i]) * k + kk[i];
        double accv = 0.0;
        for (uint32_t j = sp[kq[i]]; j < sp[kq[i] + 1]; ++j)
          accv += (double)(lv[j] * (sv[j] - v[i]));
        const float g = div_by_rows((float)accv, R);
        if (gv) gv[o] = g;
        if (UPDATE) {
          if (OPT == XF_OPT_FTRL) {
            float w = v[i], nn = vn[i], z = vz[i];
            xf::ftrl_step(TV.alpha, TV.inv_alpha, TV.beta, TV.lambda1, TV.lambda2, g, w, nn, z);
            TV.w[to[i]] = w;
            xf::store_nz(TV, to[i], nn, z);
          } else {
            TV.w[to[i]] = xf::sgd_step(TV.lr, g, v[i]);
          }
        }
      }
    }
    }  // K % 4 != 0
    // the keys' w coordinate (gw = k x the LR gradient, fm_worker.cc:140), one lane per key: as
    // the factor-0 lanes' job inside the loop above it was four memory instructions per item
    // with one lane in k active (94 of 590 us at k = 16)
    if constexpr (!REC)
    for (uint32_t q = tid; q < nk; q += kBlock) {
      double accw = 0.0;
      for (uint32_t j = sp[q]; j < sp[q + 1]; ++j) accw += (double)lv[j];
      const float g1 = div_by_rows((float)(accw * (double)k), R);
      gw[ua + q] = g1;
      if (UPDATE) {
        const uint32_t rw = rows_w[ua + q];
        if (OPT == XF_OPT_FTRL) {
          float w = wu[ua + q], nn, z;
          xf::load_nz(TW, rw, nn, z);
          xf::ftrl_step(TW.alpha, TW.inv_alpha, TW.beta, TW.lambda1, TW.lambda2, g1, w, nn, z);
          TW.w[rw] = w;
          xf::store_nz(TW, rw, nn, z);
        } else {
          TW.w[rw] = xf::sgd_step(TW.lr, g1, wu[ua + q]);
        }
      }
    }
    __syncthreads();
  }
}

// optimizer step for the listed keys of a dim-k table (the heavy ones)
template <int OPT>
__global__ void __launch_bounds__(kBlock)
k_update_listed_rows(xf::TableDev T, const uint32_t *__restrict__ list, uint32_t H,
                     const uint32_t *__restrict__ rows, const float *__restrict__ g) {
  const size_t total = (size_t)H * T.dim;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
    const size_t h = e / T.dim, j = e - h * T.dim;
    const uint32_t u = list[h];
    const size_t o = (size_t)rows[u] * T.dim + j;
    const float gg = g[(size_t)u * T.dim + j];
    if (OPT == XF_OPT_FTRL) {
      float w = T.w[o], nn, z;
      xf::load_nz(T, o, nn, z);
      xf::ftrl_step(T.alpha, T.inv_alpha, T.beta, T.lambda1, T.lambda2, gg, w, nn, z);
      T.w[o] = w;
      xf::store_nz(T, o, nn, z);
    } else {
      T.w[o] = xf::sgd_step(T.lr, gg, T.w[o]);
    }
  }
}

// heavy keys: one block per key, wave w takes factors w, w+4, ...
__global__ void __launch_bounds__(kBlock)
k_fm_grad_heavy(const uint32_t *__restrict__ heavy, uint32_t H,
                const uint32_t *__restrict__ segptr, const uint32_t *__restrict__ coo_row,
                const float *__restrict__ loss, const float *__restrict__ vsum,
                const float *__restrict__ vu, uint32_t R, int k, float *__restrict__ gw,
                float *__restrict__ gv) {
#pragma clang fp contract(off)
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t h = blockIdx.x; h < H; h += gridDim.x) {
    const uint32_t u = heavy[h];
    const uint32_t b = segptr[u], e = segptr[u + 1];
    for (uint32_t kk = wave; kk < (uint32_t)k; kk += kBlock / 64) {
      const float v = vu[(size_t)u * k + kk];
      double accw = 0.0, accv = 0.0;
      for (uint32_t j = b + lane; j < e; j += 64) {
        const uint32_t sid = coo_row[j];
        const float l = loss[sid];
        accw += (double)l;
        accv += (double)(l * (vsum[sid] - v));
      }
      accw = group_sum<64>(accw);
      accv = group_sum<64>(accv);
      if (lane == 0) {
        gv[(size_t)u * k + kk] = (float)((double)(float)accv / (1.0 * R));
        if (kk == 0) gw[u] = (float)((double)(float)(accw * (double)k) / (1.0 * R));
      }
    }
  }
}

// one tile per workgroup while the grid stays reasonable (measured: a workgroup that walks
// several tiles serialises their load->barrier->sum chains; 1e7 occurrences take 50 us with
// one tile per workgroup, 58 us with five)
inline int tile_grid(uint32_t ntiles) { return (int)std::min<uint32_t>(ntiles, 1u << 16); }

// the forward takes its tiles in plain order when the pulled weights fit one XCD's L2 (4 MiB)
// next to the index stream
constexpr size_t kFlatForwardBytes = (size_t)4 << 20;

// blocks for one wavefront per item
inline unsigned waves_grid(uint32_t n) { return (unsigned)(((size_t)n * 64 + kBlock - 1) / kBlock); }

inline int blocks_for_groups(uint32_t n_items, int items_per_block) {
  size_t g = ((size_t)n_items + items_per_block - 1) / items_per_block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// ------------------------------------------------ reference-order forward (parity mode 1)
// The reference accumulates a row's sum as an fp32 running sum while it walks the nonzeros in
// ascending fid (the merge-join of lr_worker.cc:127-138 over the sorted all_keys; FM:
// fm_worker.cc:166-192, the second-order sums k-outer and pooled over k).  These kernels do
// exactly that, one thread per example over the row's unique-key indices in ascending order:
// the loss is then bit for bit the reference arithmetic's (the oracle's mode 0).  Slow on
// purpose (one lane walks a row); a checking mode, not the product path.
__global__ void __launch_bounds__(kBlock)
k_lr_forward_reforder(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ us,
                      const uint32_t *__restrict__ rows_u, const float *__restrict__ w,
                      const int32_t *__restrict__ labels, uint32_t R, float *__restrict__ loss,
                      float *__restrict__ pctr) {
#pragma clang fp contract(off)
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  float wx = 0.0f;
  for (uint32_t j = rowptr[r]; j < rowptr[r + 1]; ++j) {
    const uint32_t u = us[j];
    wx += w[rows_u ? rows_u[u] : u];  // lr_worker.cc:132
  }
  const float p = xf::sigmoid_ref(wx);  // :141
  if (pctr) pctr[r] = p;
  if (loss) loss[r] = p - (float)labels[r];
}

__global__ void __launch_bounds__(kBlock)
k_fm_forward_reforder(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ us,
                      const float *__restrict__ wu, const float *__restrict__ vu, int k,
                      const int32_t *__restrict__ labels, uint32_t R, float *__restrict__ loss,
                      float *__restrict__ pctr, float *__restrict__ vsum_out) {
#pragma clang fp contract(off)
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const uint32_t b = rowptr[r], e = rowptr[r + 1];
  float wx = 0.0f, vs = 0.0f, vp = 0.0f;
  for (uint32_t j = b; j < e; ++j) wx += wu[us[j]];  // fm_worker.cc:166-176
  for (int kk = 0; kk < k; ++kk)                      // :177-192, k outer
    for (uint32_t j = b; j < e; ++j) {
      const float v = vu[(size_t)us[j] * k + kk];
      vs += v;
      vp += v * v;
    }
  const float vy = vs * vs - vp;              // :193-196
  const float p = xf::sigmoid_ref(wx + vy);   // :198-201
  if (pctr) pctr[r] = p;
  if (loss) loss[r] = p - (float)labels[r];
  vsum_out[r] = vs;
}

// ------------------------------------------------ reference-order gradient (parity mode 1)
// calculate_gradient as written: a key's gradient is an fp32 running sum over its occurrences
// in the order of the sorted all_keys (xf_batch_dev.hip: batch_reference_coo has that order),
// then `/= 1.0 * rows` (lr_worker.cc:104-118).  One wavefront per key: it fetches 64
// occurrences at a time and adds them one after the other (the adds are the dependent chain,
// the loads are not).  The optimizer step on the result is the ordinary Push
// (xf_table_update_dev), so g, w, n, z equal the oracle's reference arithmetic bit for bit.
__global__ void __launch_bounds__(kBlock)
k_lr_grad_reforder(const uint32_t *__restrict__ segptr, const uint32_t *__restrict__ coo,
                   const float *__restrict__ loss, uint32_t U, uint32_t R,
                   float *__restrict__ g_out) {
#pragma clang fp contract(off)
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t nw = gridDim.x * (kBlock / 64);
  for (uint32_t u = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; u < U; u += nw) {
    const uint32_t b = segptr[u], e = segptr[u + 1];
    float acc = 0.0f;  // push_gradient starts at 0 (lr_worker.cc:168)
    for (uint32_t j0 = b; j0 < e; j0 += 64) {
      const uint32_t n = min(64u, e - j0);
      const float v = lane < n ? loss[coo[j0 + lane]] : 0.0f;
      for (uint32_t i = 0; i < n; ++i) acc = acc + __shfl(v, (int)i);  // :111
    }
    if (lane == 0) g_out[u] = div_by_rows(acc, R);  // :117
  }
}

// fm_worker.cc:134-156: k outer; inside it every occurrence adds loss[sid] to gw (so gw runs
// through the key's occurrences k times) and loss[sid] * (v_sum[sid] - v[i,k]) to gv[i,k].
// One wavefront per key, lane = factor (k > 64: in rounds of 64); the gw chain is carried by
// every lane alike.
__global__ void __launch_bounds__(kBlock)
k_fm_grad_reforder(const uint32_t *__restrict__ segptr, const uint32_t *__restrict__ coo,
                   const float *__restrict__ loss, const float *__restrict__ vsum,
                   const float *__restrict__ vu, int k, uint32_t U, uint32_t R,
                   float *__restrict__ gw_out, float *__restrict__ gv_out) {
#pragma clang fp contract(off)
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t nw = gridDim.x * (kBlock / 64);
  for (uint32_t u = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; u < U; u += nw) {
    const uint32_t b = segptr[u], e = segptr[u + 1];
    float gw = 0.0f;
    for (int k0 = 0; k0 < k; k0 += 64) {
      const int kk = k0 + (int)lane;
      const float vk = kk < k ? vu[(size_t)u * k + kk] : 0.0f;
      float gv = 0.0f;
      for (uint32_t j0 = b; j0 < e; j0 += 64) {
        const uint32_t n = min(64u, e - j0);
        const uint32_t sid = lane < n ? coo[j0 + lane] : 0u;
        const float l = lane < n ? loss[sid] : 0.0f, vs = lane < n ? vsum[sid] : 0.0f;
        for (uint32_t i = 0; i < n; ++i) {
          const float li = __shfl(l, (int)i), vi = __shfl(vs, (int)i);
          gv = gv + li * (vi - vk);  // :141-142
        }
      }
      if (kk < k) gv_out[(size_t)u * k + kk] = div_by_rows(gv, R);  // :153-155
    }
    for (int kk = 0; kk < k; ++kk)  // :140, once per factor
      for (uint32_t j0 = b; j0 < e; j0 += 64) {
        const uint32_t n = min(64u, e - j0);
        const float l = lane < n ? loss[coo[j0 + lane]] : 0.0f;
        for (uint32_t i = 0; i < n; ++i) gw = gw + __shfl(l, (int)i);
      }
    if (lane == 0) gw_out[u] = div_by_rows(gw, R);  // :150-152
  }
}

// ------------------------------------------------------------------------ C entry points
extern "C" int xf_lr_forward_dev(const xf_dev_batch *b, const float *d_wu, float *d_loss,
                                 float *d_pctr, void *stream) {
  XF_REQUIRE(b && d_wu && d_loss, "xf_lr_forward_dev: null argument");
  if (b->R == 0) return XF_OK;
  const double avg = (double)b->NNZ / b->R;
  if (b->P >= 8 && b->fwd_grid && b->fwd_tile_ptr && b->fwd_panel_first && b->fwd_scratch) {
    // ... or when pinning would leave XCDs idle: fwd_grid is 8 x the longest per-XCD tile list
    // (Zipf 1.05: 125 us pinned, 54 us flat; uniform batches are balanced and stay pinned)
    const bool flat = (size_t)b->U * 4 <= kFlatForwardBytes ||
                      (double)b->fwd_grid > 1.25 * (double)b->fwd_ntiles;
    hipLaunchKernelGGL(k_lr_forward_tiled, dim3(flat ? b->fwd_ntiles : b->fwd_grid), dim3(kBlock),
                       0, S(stream), b->fwd_tile_ptr, b->fwd_panel_first, b->P, b->pptr, b->pidx,
                       d_wu, b->R, b->fwd_scratch, flat ? b->fwd_ntiles : 0u);
    XF_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_lr_finalize, dim3((b->R + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       S(stream), b->fwd_scratch, b->labels, b->R, b->P, d_loss, d_pctr);
    XF_HIP(hipGetLastError());
    return XF_OK;
  }
  if (avg <= 48.0) {  // short rows: four examples per wavefront
    hipLaunchKernelGGL(k_lr_forward<16>, dim3(blocks_for_groups(b->R, kBlock / 16)),
                       dim3(kBlock), 0, S(stream), b->rowptr, b->uidx, d_wu, b->labels, b->R,
                       d_loss, d_pctr);
  } else {
    hipLaunchKernelGGL(k_lr_forward<64>, dim3(blocks_for_groups(b->R, kBlock / 64)),
                       dim3(kBlock), 0, S(stream), b->rowptr, b->uidx, d_wu, b->labels, b->R,
                       d_loss, d_pctr);
  }
  XF_HIP(hipGetLastError());
  return XF_OK;
}

extern "C" int xf_lr_grad_dev(const xf_dev_batch *b, const float *d_loss, float *d_g,
                              void *stream) {
  XF_REQUIRE(b && d_loss && d_g, "xf_lr_grad_dev: null argument");
  if (b->U == 0) return XF_OK;
  if (b->ntiles && b->tile_ptr) {
    hipLaunchKernelGGL((k_lr_grad_tiled<XF_OPT_SGD, false>), dim3(tile_grid(b->ntiles)),
                       dim3(kBlock), 0, S(stream), xf::TableDev{}, b->tile_ptr, b->ntiles,
                       b->segptr, b->coo_row, d_loss, (const uint32_t *)nullptr,
                       (const float *)nullptr, b->R, d_g);
  } else {
    hipLaunchKernelGGL(k_lr_grad, dim3(blocks_for_groups(b->U, kBlock)), dim3(kBlock), 0,
                       S(stream), b->segptr, b->coo_row, d_loss, b->U, b->R, d_g);
  }
  XF_HIP(hipGetLastError());
  if (b->H && b->heavy_chunk_ptr && b->heavy_scratch) {
    hipLaunchKernelGGL(k_lr_heavy_partial, dim3(b->n_heavy_chunks), dim3(kBlock), 0, S(stream),
                       b->heavy, b->heavy_chunk_ptr, b->H, b->segptr, b->coo_row, d_loss,
                       b->heavy_scratch);
    hipLaunchKernelGGL((k_lr_heavy_finish<XF_OPT_SGD, false>), dim3(waves_grid(b->H)),
                       dim3(kBlock), 0, S(stream), xf::TableDev{}, b->heavy, b->heavy_chunk_ptr,
                       b->H, b->heavy_scratch, (const uint32_t *)nullptr, b->R, d_g);
    XF_HIP(hipGetLastError());
  } else if (b->H) {
    hipLaunchKernelGGL(k_lr_grad_heavy, dim3(blocks_for_groups(b->H, kBlock / 64)),
                       dim3(kBlock), 0, S(stream), b->heavy, b->H, b->segptr, b->coo_row,
                       d_loss, b->R, d_g);
    XF_HIP(hipGetLastError());
  }
  return XF_OK;
}

extern "C" int xf_lr_grad_update_dev(xf_table *t, const xf_dev_batch *b, const uint32_t *d_slots,
                                     const float *d_wu, const float *d_loss, float *d_g,
                                     void *stream) {
  XF_REQUIRE(t && b && d_slots && d_loss && d_g, "xf_lr_grad_update_dev: null argument");
  XF_REQUIRE(xf::table_dim(t) == 1, "xf_lr_grad_update_dev: dim must be 1");
  if (b->U == 0) return XF_OK;
  if (!(b->ntiles && b->tile_ptr)) {  // a view without gradient tiles: unfused
    XF_TRY(xf_lr_grad_dev(b, d_loss, d_g, stream));
    return xf_table_update_dev(t, d_slots, b->U, d_g, stream);
  }
  const xf::TableDev &T = xf::table_dev(t);
  const bool ftrl = T.nz != nullptr;
  const dim3 blk(kBlock);
  {
    const dim3 gt(tile_grid(b->ntiles));
    if (ftrl)
      hipLaunchKernelGGL((k_lr_grad_tiled<XF_OPT_FTRL, true>), gt, blk, 0, S(stream), T,
                         b->tile_ptr, b->ntiles, b->segptr, b->coo_row, d_loss, d_slots, d_wu,
                         b->R, d_g);
    else
      hipLaunchKernelGGL((k_lr_grad_tiled<XF_OPT_SGD, true>), gt, blk, 0, S(stream), T,
                         b->tile_ptr, b->ntiles, b->segptr, b->coo_row, d_loss, d_slots, d_wu,
                         b->R, d_g);
  }
  XF_HIP(hipGetLastError());
  if (b->H && b->heavy_chunk_ptr && b->heavy_scratch) {
    const dim3 gh(waves_grid(b->H));  // a wavefront per heavy key
    hipLaunchKernelGGL(k_lr_heavy_partial, dim3(b->n_heavy_chunks), blk, 0, S(stream), b->heavy,
                       b->heavy_chunk_ptr, b->H, b->segptr, b->coo_row, d_loss, b->heavy_scratch);
    if (ftrl)
      hipLaunchKernelGGL((k_lr_heavy_finish<XF_OPT_FTRL, true>), gh, blk, 0, S(stream), T,
                         b->heavy, b->heavy_chunk_ptr, b->H, b->heavy_scratch, d_slots, b->R, d_g);
    else
      hipLaunchKernelGGL((k_lr_heavy_finish<XF_OPT_SGD, true>), gh, blk, 0, S(stream), T,
                         b->heavy, b->heavy_chunk_ptr, b->H, b->heavy_scratch, d_slots, b->R, d_g);
    XF_HIP(hipGetLastError());
  } else if (b->H) {
    hipLaunchKernelGGL(k_lr_grad_heavy, dim3(blocks_for_groups(b->H, kBlock / 64)),
                       dim3(kBlock), 0, S(stream), b->heavy, b->H, b->segptr, b->coo_row,
                       d_loss, b->R, d_g);
    XF_HIP(hipGetLastError());
    const dim3 gh((b->H + kBlock - 1) / kBlock);
    if (ftrl)
      hipLaunchKernelGGL(k_update_listed<XF_OPT_FTRL>, gh, blk, 0, S(stream), T, b->heavy, b->H,
                         d_slots, d_g);
    else
      hipLaunchKernelGGL(k_update_listed<XF_OPT_SGD>, gh, blk, 0, S(stream), T, b->heavy, b->H,
                         d_slots, d_g);
    XF_HIP(hipGetLastError());
  }
  return XF_OK;
}

extern "C" int xf_fm_forward_dev(const xf_dev_batch *b, int k, const float *d_wu,
                                 const float *d_vu, float *d_loss, float *d_pctr,
                                 float *d_vsum, void *stream) {
  XF_REQUIRE(b && d_wu && d_vu && d_loss && d_vsum && k >= 1, "xf_fm_forward_dev: bad argument");
  if (b->R == 0) return XF_OK;
  const dim3 g(blocks_for_groups(b->R, kBlock / 64)), blk(kBlock);
#define XF_FM_FWD(KK)                                                                       \
  hipLaunchKernelGGL(k_fm_forward<KK>, g, blk, 0, S(stream), b->rowptr, b->uidx, d_wu, d_vu, \
                     k, b->labels, b->R, d_loss, d_pctr, d_vsum)
  switch (k) {
    case 8: XF_FM_FWD(8); break;
    case 10: XF_FM_FWD(10); break;
    case 16: XF_FM_FWD(16); break;
    case 32: XF_FM_FWD(32); break;
    case 64: XF_FM_FWD(64); break;
    default: XF_FM_FWD(0); break;
  }
#undef XF_FM_FWD
  XF_HIP(hipGetLastError());
  return XF_OK;
}

// FM forward from dense pulled rows (w_u[U], v_u[U x k]) through the per-key records: one pass
// over v_u forms (sum_k v, sum_k v^2, w) per key into `d_ks` (U x 32 bytes of caller-owned
// scratch), the per-nonzero pass gathers one record.  Same sums as xf_fm_forward_dev (exact
// fp64), a fraction of its time at larger k (k = 64: one 256-byte row per nonzero there).
// Returns XF_EINVAL-free false when k does not fit the record kernel (the caller falls back).
namespace xf {
bool fm_records_fit(int k) {
  const int dim4 = k / 4;
  return k % 4 == 0 && dim4 >= 1 && dim4 <= 16 && (dim4 & (dim4 - 1)) == 0;
}
size_t fm_record_bytes(size_t U) { return std::max<size_t>(U, 1) * sizeof(FmKey); }
int fm_forward_records(const xf_dev_batch *b, int k, const float *d_wu, const float *d_vu,
                       void *d_ks, float *d_loss, float *d_pctr, float *d_vsum,
                       hipStream_t s) {
  XF_REQUIRE(b && d_wu && d_vu && d_ks && d_loss && d_vsum && fm_records_fit(k),
             "fm_forward_records: bad argument");
  if (b->R == 0) return XF_OK;
  const int dim4 = k / 4;
  const size_t tot = (size_t)b->U * dim4;
  const dim3 g((unsigned)std::min<size_t>((tot + kBlock - 1) / kBlock, 8192)), blk(kBlock);
#define XF_FM_GS(D)                                                                          \
  hipLaunchKernelGGL(k_fm_gather_scalars<D>, g, blk, 0, s, (const float4 *)d_vu,             \
                     (const uint32_t *)nullptr, d_wu, (const uint32_t *)nullptr,             \
                     (size_t)b->U, (float4 *)nullptr, (FmKey *)d_ks, (const uint32_t *)nullptr)
  if (b->U) {
    switch (dim4) {
      case 1: XF_FM_GS(1); break;
      case 2: XF_FM_GS(2); break;
      case 4: XF_FM_GS(4); break;
      case 8: XF_FM_GS(8); break;
      default: XF_FM_GS(16); break;
    }
  }
#undef XF_FM_GS
  hipLaunchKernelGGL(k_fm_forward_scalars, dim3(blocks_for_groups(b->R, kBlock / 64)),
                     dim3(kBlock), 0, s, b->rowptr, b->uidx, (const FmKey *)d_ks, b->labels, b->R,
                     d_loss, d_pctr, d_vsum);
  XF_HIP(hipGetLastError());
  return XF_OK;
}
}  // namespace xf

extern "C" int xf_fm_grad_dev(const xf_dev_batch *b, int k, const float *d_vu,
                              const float *d_vsum, const float *d_loss, float *d_gw,
                              float *d_gv, void *stream) {
  XF_REQUIRE(b && d_vu && d_vsum && d_loss && d_gw && d_gv && k >= 1,
             "xf_fm_grad_dev: bad argument");
  if (b->U == 0) return XF_OK;
  if (b->ntiles && b->tile_ptr) {
#define XF_FM_GRAD(KK)                                                                       \
  hipLaunchKernelGGL((k_fm_grad_tiled<XF_OPT_SGD, false, KK>), dim3(tile_grid(b->ntiles)),    \
                     dim3(kBlock), 0, S(stream), xf::TableDev{}, xf::TableDev{}, b->tile_ptr, \
                     b->ntiles, b->segptr, b->coo_row, d_loss, d_vsum, (const float *)nullptr, \
                     d_vu, (const uint32_t *)nullptr, (const uint32_t *)nullptr, b->R, k, d_gw, \
                     d_gv, (FmKey *)nullptr)
    switch (k) {
      case 8: XF_FM_GRAD(8); break;
      case 10: XF_FM_GRAD(10); break;
      case 16: XF_FM_GRAD(16); break;
      case 32: XF_FM_GRAD(32); break;
      case 64: XF_FM_GRAD(64); break;
      default: XF_FM_GRAD(0); break;
    }
#undef XF_FM_GRAD
  } else {
    const size_t total = (size_t)b->U * k;
    size_t g = (total + kBlock - 1) / kBlock;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_fm_grad, dim3((int)g), dim3(kBlock), 0, S(stream), b->segptr,
                       b->coo_row, d_loss, d_vsum, d_vu, b->U, b->R, k, d_gw, d_gv);
  }
  XF_HIP(hipGetLastError());
  if (b->H) {
    if (b->heavy_chunk_ptr && b->heavy_scratch && k <= XF_HEAVY_KMAX) {
      hipLaunchKernelGGL(k_fm_heavy_partial, dim3(b->n_heavy_chunks), dim3(kBlock), 0, S(stream),
                         b->heavy, b->heavy_chunk_ptr, b->H, b->segptr, b->coo_row, d_loss, d_vsum,
                         d_vu, k, b->heavy_scratch, b->R, d_gw, d_gv);
      hipLaunchKernelGGL(k_fm_heavy_finish, dim3(b->H), dim3(kBlock), 0,
                         S(stream), b->heavy, b->heavy_chunk_ptr, b->H, b->heavy_scratch, b->R, k,
                         d_gw, d_gv);
    } else {
      hipLaunchKernelGGL(k_fm_grad_heavy, dim3(std::min<uint32_t>(b->H, 4096)), dim3(kBlock), 0,
                         S(stream), b->heavy, b->H, b->segptr, b->coo_row, d_loss, d_vsum, d_vu,
                         b->R, k, d_gw, d_gv);
    }
    XF_HIP(hipGetLastError());
  }
  return XF_OK;
}

// FM gradient fused with the two Pushes, for tables on this GPU (single shard).  rows_w /
// rows_v as returned by the Pulls of b->ukeys; gw, gv are still written (parity hook).
// REC mode, heavy keys (their gradient is reduced by kernels of their own that read the pulled
// rows from scratch): pull just those keys' rows into their places of d_wu / d_vu ...
__global__ void __launch_bounds__(kBlock)
k_fm_pull_listed(xf::TableDev TW, xf::TableDev TV, const uint32_t *__restrict__ list, uint32_t H,
                 const uint32_t *__restrict__ rows_w, const uint32_t *__restrict__ rows_v,
                 float *__restrict__ wu, float *__restrict__ vu) {
  const int k = TV.dim;
  const size_t total = (size_t)H * k;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (size_t)gridDim.x * blockDim.x) {
    const uint32_t u = list[e / k], kk = (uint32_t)(e % k);
    vu[(size_t)u * k + kk] = TV.w[(size_t)rows_v[u] * k + kk];
    if (kk == 0) wu[u] = TW.w[rows_w[u]];
  }
}
// ... and, after their optimizer steps, bring their records up to date: the sums of
// k_fm_gather_scalars in its order (four factors in sequence, then the pairwise tree)
__global__ void __launch_bounds__(kBlock)
k_fm_records_listed(xf::TableDev TW, xf::TableDev TV, const uint32_t *__restrict__ list,
                    uint32_t H, const uint32_t *__restrict__ rows_w,
                    const uint32_t *__restrict__ rows_v, FmKey *__restrict__ rec) {
#pragma clang fp contract(off)
  const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  const uint32_t u = list[h], rv = rows_v[u];
  const int dim4 = TV.dim / 4;  // a power of two <= 16
  double a[16], b[16];
  for (int j = 0; j < dim4; ++j) {
    const float4 v = *reinterpret_cast<const float4 *>(TV.w + (size_t)rv * TV.dim + 4 * j);
    a[j] = (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    b[j] = (double)(v.x * v.x) + (double)(v.y * v.y) + (double)(v.z * v.z) +
           (double)(v.w * v.w);
  }
  for (int off = dim4 / 2; off > 0; off >>= 1)
    for (int j = 0; j < off; ++j) {
      a[j] += a[j + off];
      b[j] += b[j + off];
    }
  FmKey q;
  q.a = a[0];
  q.b = b[0];
  q.w = TW.w[rows_w[u]];
  q.pad[0] = q.pad[1] = q.pad[2] = 0.f;
  rec[rv] = q;
}

// store_gv = false: the per-coordinate gradients of the tiled keys stay in registers (they
// are consumed by the fused Push); d_gv then only carries the heavy keys' rows between the
// chunk reduction and their optimizer step.  U x k x 4 bytes less to write per step.
// rec != null: REC mode of k_fm_grad_tiled (d_wu / d_vu are scratch for the heavy keys only).
static int fm_grad_update(xf_table *tw, xf_table *tv, const xf_dev_batch *b,
                          const uint32_t *d_rows_w, const uint32_t *d_rows_v, float *d_wu,
                          float *d_vu, const float *d_vsum, const float *d_loss,
                          float *d_gw, float *d_gv, bool store_gv, void *stream,
                          FmKey *rec = nullptr) {
  XF_REQUIRE(tw && tv && b && d_rows_w && d_rows_v && d_wu && d_vu && d_vsum && d_loss && d_gw &&
                 d_gv, "xf_fm_grad_update_dev: null argument");
  if (b->U == 0) return XF_OK;
  const xf::TableDev &TW = xf::table_dev(tw), &TV = xf::table_dev(tv);
  const int k = TV.dim;
  const bool ftrl = TV.nz != nullptr;
  XF_REQUIRE((TW.nz != nullptr) == ftrl, "xf_fm_grad_update_dev: w and v use different optimizers");
  XF_REQUIRE(b->ntiles && b->tile_ptr, "xf_fm_grad_update_dev: batch has no gradient tiles");
  const dim3 gt(tile_grid(b->ntiles)), blk(kBlock);
  if (rec) {
    XF_REQUIRE(!store_gv && (k == 4 || k == 8 || k == 16 || k == 32 || k == 64),
               "fm_grad_update: per-row records with k = %d", k);
    if (b->H)
      hipLaunchKernelGGL(k_fm_pull_listed, dim3(blocks_for_groups(b->H * k, kBlock)), blk, 0,
                         S(stream), TW, TV, b->heavy, b->H, d_rows_w, d_rows_v, d_wu, d_vu);
#define XF_FM_GR(OPTV, KK)                                                                     \
  hipLaunchKernelGGL((k_fm_grad_tiled<OPTV, true, KK, true>), gt, blk, 0, S(stream), TW, TV,    \
                     b->tile_ptr, b->ntiles, b->segptr, b->coo_row, d_loss, d_vsum, d_wu, d_vu, \
                     d_rows_w, d_rows_v, b->R, k, d_gw, (float *)nullptr, rec)
#define XF_FM_GR_K(OPTV)                       \
  switch (k) {                                 \
    case 4: XF_FM_GR(OPTV, 4); break;          \
    case 8: XF_FM_GR(OPTV, 8); break;          \
    case 16: XF_FM_GR(OPTV, 16); break;        \
    case 32: XF_FM_GR(OPTV, 32); break;        \
    default: XF_FM_GR(OPTV, 64); break;        \
  }
    if (ftrl) {
      XF_FM_GR_K(XF_OPT_FTRL)
    } else {
      XF_FM_GR_K(XF_OPT_SGD)
    }
#undef XF_FM_GR_K
#undef XF_FM_GR
  } else {
  // (weights written without the per-row records following them: whoever keeps records of
  // these tables' rows must rebuild them)
  xf::table_note_write(tw);
  xf::table_note_write(tv);
#define XF_FM_GU(OPTV, KK)                                                                     \
  hipLaunchKernelGGL((k_fm_grad_tiled<OPTV, true, KK>), gt, blk, 0, S(stream), TW, TV,          \
                     b->tile_ptr, b->ntiles, b->segptr, b->coo_row, d_loss, d_vsum, d_wu, d_vu, \
                     d_rows_w, d_rows_v, b->R, k, d_gw, store_gv ? d_gv : (float *)nullptr,     \
                     (FmKey *)nullptr)
#define XF_FM_GU_K(OPTV)                       \
  switch (k) {                                 \
    case 8: XF_FM_GU(OPTV, 8); break;          \
    case 10: XF_FM_GU(OPTV, 10); break;        \
    case 16: XF_FM_GU(OPTV, 16); break;        \
    case 32: XF_FM_GU(OPTV, 32); break;        \
    case 64: XF_FM_GU(OPTV, 64); break;        \
    default: XF_FM_GU(OPTV, 0); break;         \
  }
  if (ftrl) {
    XF_FM_GU_K(XF_OPT_FTRL)
  } else {
    XF_FM_GU_K(XF_OPT_SGD)
  }
#undef XF_FM_GU_K
#undef XF_FM_GU
  }
  XF_HIP(hipGetLastError());
  if (b->H) {
    if (b->heavy_chunk_ptr && b->heavy_scratch && k <= XF_HEAVY_KMAX) {
      hipLaunchKernelGGL(k_fm_heavy_partial, dim3(b->n_heavy_chunks), dim3(kBlock), 0, S(stream),
                         b->heavy, b->heavy_chunk_ptr, b->H, b->segptr, b->coo_row, d_loss, d_vsum,
                         d_vu, k, b->heavy_scratch, b->R, d_gw, d_gv);
      hipLaunchKernelGGL(k_fm_heavy_finish, dim3(b->H), dim3(kBlock), 0,
                         S(stream), b->heavy, b->heavy_chunk_ptr, b->H, b->heavy_scratch, b->R, k,
                         d_gw, d_gv);
    } else {
      hipLaunchKernelGGL(k_fm_grad_heavy, dim3(std::min<uint32_t>(b->H, 4096)), dim3(kBlock), 0,
                         S(stream), b->heavy, b->H, b->segptr, b->coo_row, d_loss, d_vsum, d_vu,
                         b->R, k, d_gw, d_gv);
    }
    XF_HIP(hipGetLastError());
    const dim3 gh((b->H + kBlock - 1) / kBlock), gk(blocks_for_groups(b->H * k, kBlock));
    if (ftrl) {
      hipLaunchKernelGGL(k_update_listed<XF_OPT_FTRL>, gh, blk, 0, S(stream), TW, b->heavy, b->H,
                         d_rows_w, d_gw);
      hipLaunchKernelGGL(k_update_listed_rows<XF_OPT_FTRL>, gk, blk, 0, S(stream), TV, b->heavy,
                         b->H, d_rows_v, d_gv);
    } else {
      hipLaunchKernelGGL(k_update_listed<XF_OPT_SGD>, gh, blk, 0, S(stream), TW, b->heavy, b->H,
                         d_rows_w, d_gw);
      hipLaunchKernelGGL(k_update_listed_rows<XF_OPT_SGD>, gk, blk, 0, S(stream), TV, b->heavy,
                         b->H, d_rows_v, d_gv);
    }
    if (rec)
      hipLaunchKernelGGL(k_fm_records_listed, gh, blk, 0, S(stream), TW, TV, b->heavy, b->H,
                         d_rows_w, d_rows_v, rec);
    XF_HIP(hipGetLastError());
  }
  return XF_OK;
}

extern "C" int xf_fm_grad_update_dev(xf_table *tw, xf_table *tv, const xf_dev_batch *b,
                                     const uint32_t *d_rows_w, const uint32_t *d_rows_v,
                                     const float *d_wu, const float *d_vu, const float *d_vsum,
                                     const float *d_loss, float *d_gw, float *d_gv, void *stream) {
  return fm_grad_update(tw, tv, b, d_rows_w, d_rows_v, const_cast<float *>(d_wu),
                        const_cast<float *>(d_vu), d_vsum, d_loss, d_gw, d_gv, true, stream);
}

// ---------------------------------------------------------------------------- workspace
enum { kEvResolve = 0, kEvGather, kEvForward, kEvGrad, kEvUpdate, kEvCount };

struct xf_workspace {
  uint32_t *slots = nullptr, *slots2 = nullptr;
  float *wu = nullptr, *g = nullptr, *vu = nullptr, *gv = nullptr;
  float *loss = nullptr, *pctr = nullptr, *vsum = nullptr;
  void *ks = nullptr;  // FM: per-key (a, b, w) records, 32 B each
  void *owner_rec = nullptr;  // fm_owner_partials -> fm_owner_grad_update: the v table's records
  size_t capU = 0, capUK = 0, capR = 0;
  double *partial = nullptr;  // LR forward: partial row sums of the window workgroups
  size_t capPartial = 0;
  float *gdense = nullptr;    // capture mode: gradients indexed by state row
  size_t capGdense = 0;
  // capture: keep the step's intermediates per unique key (pulled weights, gradients) for
  // xf_workspace_fetch — the parity hook.  Off by default: the production step never forms them.
  bool capture = false;
  // parity mode: 0 = row / key sums exact in fp64 (the production path); 1 = "reference order":
  // the forward's row sums as fp32 running sums in the reference's own order (slow kernels)
  int parity = 0;
  uint32_t lastU = 0, lastR = 0;
  // optional per-kernel HIP-event timing (same stream, inside the caller's timed region)
  bool profiling = false;
  // A ring of event sets: a step records into the oldest set, whose own step finished long
  // ago — reading it back never makes the host wait for the step just launched (one set meant a
  // hipEventSynchronize on the previous step before every launch: the host could not run
  // ahead and every step paid its launch latencies, ~10 us of a 145 us step).
  static constexpr int kEvSets = 8;
  struct EvSet {
    hipEvent_t ev[kEvCount + 1] = {};
    bool pending = false;
    int nseg = 0;                 // segments recorded by the step
    int seg_slot[kEvCount] = {};  // segment k is accounted to ms_sum[seg_slot[k]]
  } sets[kEvSets];
  int cur = 0;  // the set the running step records into
  // every kProfileEvery-th step is the one that records (an event is a barrier packet of its
  // own on the stream: three per step cost ~2 us of a 136 us step)
  static constexpr long kProfileEvery = 4;
  long step_no = 0;
  bool rec = false;  // this step records
  double ms_sum[kEvCount] = {};
  long steps_timed = 0;
  int nmark = 0;
};

static int ws_reserve(xf_workspace *ws, size_t U, size_t UK, size_t R) {
  auto grow = [](void **p, size_t bytes) -> hipError_t {
    if (*p) {
      hipError_t e = hipFree(*p);
      if (e != hipSuccess) return e;
    }
    return hipMalloc(p, bytes);
  };
  if (!ws->slots || U > ws->capU) {
    const size_t m = std::max<size_t>(U + U / 8, 1024);
    XF_HIP(grow((void **)&ws->slots, m * 4));
    XF_HIP(grow((void **)&ws->slots2, m * 4));
    XF_HIP(grow((void **)&ws->wu, m * 4));
    XF_HIP(grow((void **)&ws->g, m * 4));
    ws->capU = m;
  }
  if (!ws->vu || UK > ws->capUK) {
    const size_t m = std::max<size_t>(UK + UK / 8, 1024);
    XF_HIP(grow((void **)&ws->vu, m * 4));
    XF_HIP(grow((void **)&ws->gv, m * 4));
    XF_HIP(grow(&ws->ks, std::max<size_t>(U + U / 8, 1024) * sizeof(FmKey)));
    ws->capUK = m;
  }
  if (!ws->loss || R > ws->capR) {
    const size_t m = std::max<size_t>(R + R / 8, 1024);
    XF_HIP(grow((void **)&ws->loss, m * 4));
    XF_HIP(grow((void **)&ws->pctr, m * 4));
    XF_HIP(grow((void **)&ws->vsum, m * 4));
    ws->capR = m;
  }
  return XF_OK;
}

extern "C" int xf_workspace_create(xf_workspace **out) {
  XF_REQUIRE(out, "xf_workspace_create: null argument");
  *out = new xf_workspace;
  return XF_OK;
}

extern "C" int xf_workspace_destroy(xf_workspace *ws) {
  if (!ws) return XF_OK;
  void *ps[] = {ws->slots, ws->slots2, ws->wu,   ws->g,  ws->vu,      ws->gv,
                ws->loss,  ws->pctr,   ws->vsum, ws->ks, ws->partial, ws->gdense};
  for (void *p : ps)
    if (p) hipFree(p);
  for (auto &e : ws->sets)
    for (auto &ev : e.ev)
      if (ev) hipEventDestroy(ev);
  delete ws;
  return XF_OK;
}

static int ws_collect_set(xf_workspace *ws, xf_workspace::EvSet &e) {  // fold into the sums
  if (!e.pending) return XF_OK;
  XF_HIP(hipEventSynchronize(e.ev[e.nseg]));
  for (int i = 0; i < e.nseg; ++i) {
    float ms = 0.f;
    XF_HIP(hipEventElapsedTime(&ms, e.ev[i], e.ev[i + 1]));
    ws->ms_sum[e.seg_slot[i]] += ms;
  }
  ++ws->steps_timed;
  e.pending = false;
  return XF_OK;
}
static int ws_collect(xf_workspace *ws) {  // every set still pending (end of a timed run)
  for (auto &e : ws->sets) XF_TRY(ws_collect_set(ws, e));
  return XF_OK;
}
// a step is about to record: take the oldest set
static int ws_next_set(xf_workspace *ws) {
  ws->cur = (ws->cur + 1) % xf_workspace::kEvSets;
  return ws_collect_set(ws, ws->sets[ws->cur]);
}

extern "C" int xf_workspace_profile(xf_workspace *ws, int enable) {
  XF_REQUIRE(ws, "xf_workspace_profile: null workspace");
  if (enable && !ws->sets[0].ev[0])
    for (auto &e : ws->sets)
      for (auto &ev : e.ev) XF_HIP(hipEventCreate(&ev));
  if (!enable) XF_TRY(ws_collect(ws));
  ws->profiling = enable != 0;
  if (enable) {
    for (auto &m : ws->ms_sum) m = 0.0;
    ws->steps_timed = 0;
    for (auto &e : ws->sets) e.pending = false;
    ws->step_no = 0;
  }
  return XF_OK;
}

// ms_sum[5] = resolve, gather, forward, gradient, update (summed over *steps steps)
extern "C" int xf_workspace_profile_read(xf_workspace *ws, double *ms_sum, long *steps) {
  XF_REQUIRE(ws && ms_sum && steps, "xf_workspace_profile_read: null argument");
  XF_TRY(ws_collect(ws));
  for (int i = 0; i < kEvCount; ++i) ms_sum[i] = ws->ms_sum[i];
  *steps = ws->steps_timed;
  return XF_OK;
}

// XF_BEGIN opens the step's first segment; XF_END(slot) closes the running segment,
// accounting it to ms_sum[slot], and opens the next one.  One event per boundary.
#define XF_BEGIN()                                                      \
  do {                                                                  \
    if (ws->rec) {                                                      \
      ws->nmark = 0;                                                    \
      XF_HIP(hipEventRecord(ws->sets[ws->cur].ev[0], S(stream)));       \
    }                                                                   \
  } while (0)
#define XF_END(slot)                                                    \
  do {                                                                  \
    if (ws->rec) {                                                      \
      ws->sets[ws->cur].seg_slot[ws->nmark] = (slot);                   \
      ++ws->nmark;                                                      \
      XF_HIP(hipEventRecord(ws->sets[ws->cur].ev[ws->nmark], S(stream))); \
      ws->sets[ws->cur].nseg = ws->nmark;                               \
    }                                                                   \
  } while (0)

// ---------------------------------------------------------------------------- fused steps
static int ws_reserve_cells(xf_workspace *ws, const xf_cells *c, bool dense_g) {
  const size_t need = xf::cells_partial_doubles(c);
  if (need > ws->capPartial) {
    if (ws->partial) XF_HIP(hipFree(ws->partial));
    ws->partial = nullptr;
    ws->capPartial = 0;
    XF_HIP(hipMalloc((void **)&ws->partial, (need + need / 8 + 1024) * 8));
    ws->capPartial = need + need / 8 + 1024;
  }
  // (the gradient pass writes g at (chunk0 + chunk) * kChunk + k for every segment of the chain)
  size_t gd = 0;
  if (dense_g)
    for (const xf_cells *q = c; q; q = q->next)
      gd = std::max(gd, ((size_t)q->chunk0 + q->nchunk) * xf::kChunk);
  if (gd > ws->capGdense) {
    if (ws->gdense) XF_HIP(hipFree(ws->gdense));
    ws->gdense = nullptr;
    ws->capGdense = 0;
    XF_HIP(hipMalloc((void **)&ws->gdense, gd * 4));
    ws->capGdense = gd;
  }
  return XF_OK;
}

extern "C" int xf_workspace_parity(xf_workspace *ws, int mode) {
  XF_REQUIRE(ws && (mode == XF_PARITY_EXACT_SUMS || mode == XF_PARITY_REFERENCE_ORDER),
             "xf_workspace_parity: bad argument");
  ws->parity = mode;
  return XF_OK;
}

// the LR forward of parity mode 1 for a minibatch with a key list whose rows are resolved
static int lr_forward_reforder(xf_table *w, xf_batch *b, float *loss, float *pctr,
                               hipStream_t s) {
  XF_TRY(xf::batch_sorted_uidx(b, s));
  if (b->R == 0) return XF_OK;
  XF_REQUIRE(b->d_rows_u || b->U == 0, "reference-order forward: the keys' rows are not resolved");
  hipLaunchKernelGGL(k_lr_forward_reforder, dim3((b->R + kBlock - 1) / kBlock), dim3(kBlock), 0,
                     s, b->view.rowptr, b->d_uidx_sorted, b->d_rows_u, xf::table_dev(w).w,
                     b->view.labels, b->R, loss, pctr);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

static int fm_forward_reforder(xf_batch *b, int k, const float *wu, const float *vu, float *loss,
                               float *pctr, float *vsum, hipStream_t s) {
  XF_TRY(xf::batch_sorted_uidx(b, s));
  if (b->R == 0) return XF_OK;
  hipLaunchKernelGGL(k_fm_forward_reforder, dim3((b->R + kBlock - 1) / kBlock), dim3(kBlock), 0,
                     s, b->view.rowptr, b->d_uidx_sorted, wu, vu, k, b->view.labels, b->R, loss,
                     pctr, vsum);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

extern "C" int xf_workspace_capture(xf_workspace *ws, int enable) {
  XF_REQUIRE(ws, "xf_workspace_capture: null workspace");
  ws->capture = enable != 0;
  return XF_OK;
}

// One LRWorker::update (lr_worker.cc:167-176) against a table on this GPU.  The Pull's key ->
// row resolve (and its insert-on-first-touch) happens once per (minibatch, table row
// numbering), when the batch's cells are built; a step is then two passes over the cells:
// forward (reads the table's weights in place) and gradient + Push.
// `deferred`: the minibatch's key build has not been waited for yet (xf_lr_update_dev): its
// cells serve the forward as they are, the wait — for the work items of the gradient pass and
// for the keys the table did not hold — happens on the host while the forward runs on the GPU.
static int lr_step(xf_table *w, xf_batch *b, xf_workspace *ws, void *stream,
                   xf::KbDeferred *deferred) {
  struct Settle {  // (an early error return must not leave the build unfinished)
    xf::KbDeferred *d;
    void *stream;
    ~Settle() {
      if (d) (void)xf::cells_build_keyed_finish(d, S(stream), nullptr);
    }
  } settle{deferred, stream};
  XF_TRY(xf::ensure_cells(b, w, S(stream)));  // Pull (:170): keys -> state rows, inserts
  const xf_cells *c = b->cells;
  const bool cap = ws->capture && !b->local && b->d_rows_u;
  if (settle.d && (cap || ws->parity == XF_PARITY_REFERENCE_ORDER)) {
    xf::KbDeferred *d = settle.d;
    settle.d = nullptr;
    XF_TRY(xf::cells_build_keyed_finish(d, S(stream), nullptr));
  }
  XF_TRY(ws_reserve(ws, b->U, 0, b->R));
  XF_TRY(ws_reserve_cells(ws, c, cap));
  ws->rec = ws->profiling && ws->step_no++ % xf_workspace::kProfileEvery == 0;
  if (ws->rec) XF_TRY(ws_next_set(ws));
  ws->lastU = cap ? b->U : 0;
  ws->lastR = b->R;
  const xf::TableDev &T = xf::table_dev(w);
  const int32_t *labels = b->local ? b->raw_labels : b->view.labels;
  XF_BEGIN();
  if (ws->parity == XF_PARITY_REFERENCE_ORDER)
    XF_TRY(lr_forward_reforder(w, b, ws->loss, nullptr, S(stream)));
  else
    XF_TRY(xf::cells_lr_forward(c, T.w, labels, ws->partial, ws->loss, nullptr, S(stream)));  // :172
  if (settle.d) {  // the build's wait, under the forward
    xf::KbDeferred *d = settle.d;
    settle.d = nullptr;
    bool more = false;
    XF_TRY(xf::cells_build_keyed_finish(d, S(stream), &more));
    if (more) {  // keys the table did not hold: they are in now, as a second segment of cells —
                 // its row sums on top of the first segment's (still in ws->partial)
      const size_t had = ws->capPartial;
      XF_TRY(ws_reserve_cells(ws, c, cap));
      XF_TRY(xf::cells_lr_forward(c, xf::table_dev(w).w, labels, ws->partial, ws->loss, nullptr,
                                  S(stream), ws->capPartial == had ? c->next : nullptr));
    }
  }
  XF_END(kEvForward);
  if (cap) XF_TRY(xf::gather_f32(T.w, b->d_rows_u, b->U, ws->wu, S(stream)));
  if (ws->parity == XF_PARITY_REFERENCE_ORDER) {
    // the reference's own fp32 running sums per key, then the Push as the server does it
    XF_TRY(xf::batch_reference_coo(b, S(stream)));
    if (b->U) {
      hipLaunchKernelGGL(k_lr_grad_reforder, dim3(blocks_for_groups(b->U, kBlock / 64)),
                         dim3(kBlock), 0, S(stream), b->view.segptr, b->d_ref_coo, ws->loss, b->U,
                         b->R, ws->g);  // :173
      XF_HIP(hipGetLastError());
      XF_TRY(xf_table_update_dev(w, b->d_rows_u, b->U, ws->g, stream));  // :175
    }
    XF_END(kEvGrad);
    if (ws->rec) ws->sets[ws->cur].pending = true;
    return XF_OK;
  }
  // gradient (:173) + Push (:175) in one pass over the cells
  XF_TRY(xf::cells_lr_grad_update(c, w, ws->loss, cap ? ws->gdense : nullptr, S(stream)));
  XF_END(kEvGrad);
  if (cap) XF_TRY(xf::gather_f32(ws->gdense, b->d_rows_u, b->U, ws->g, S(stream)));
  if (ws->rec) ws->sets[ws->cur].pending = true;
  return XF_OK;
}

extern "C" int xf_lr_step(xf_table *w, xf_batch *b, xf_workspace *ws, void *stream) {
  XF_REQUIRE(w && b && ws, "xf_lr_step: null argument");
  XF_REQUIRE(xf::table_dim(w) == 1, "xf_lr_step: the w table must have dim 1");
  return lr_step(w, b, ws, stream, nullptr);
}

// The whole LRWorker::update (lr_worker.cc:145-177) on a fresh minibatch in one call: the key
// build against the table (xf_batch_compile_local_dev) and the step (xf_lr_step) — same
// kernels, same results, but the build's one host wait is taken while the forward, which needs
// none of what the host waits for, already runs (the wait was ~70 us of idle GPU per
// minibatch).  *out (optional): the minibatch, for replays; freed here when null.
extern "C" int xf_lr_update_dev(xf_batch **out, xf_table *w, const uint64_t *d_keys,
                                const uint32_t *d_rowptr, const int32_t *d_labels, uint32_t R,
                                uint32_t NNZ, int retain_keys, xf_workspace *ws, void *stream) {
  XF_REQUIRE(w && ws, "xf_lr_update_dev: null argument");
  XF_REQUIRE(xf::table_dim(w) == 1, "xf_lr_update_dev: the w table must have dim 1");
  if (out) *out = nullptr;
  xf_batch *b = nullptr;
  xf::KbDeferred *def = nullptr;
  XF_TRY(xf::batch_compile_local_dev(&b, w, d_keys, d_rowptr, d_labels, R, NNZ, retain_keys,
                                     stream, &def));
  const int rc = lr_step(w, b, ws, stream, def);  // (finishes the build on every path)
  if (rc != XF_OK || !out) {
    // (the step's kernels read the minibatch: wait for them before it goes)
    if (hipStreamSynchronize(S(stream)) != hipSuccess) (void)hipGetLastError();
    xf_batch_free(b);
    return rc;
  }
  *out = b;
  return XF_OK;
}

// two Pulls (fm_worker.cc:228,231): each table resolves the key list itself — once per
// (minibatch, row numbering of the table): rows only move in a defrag, so a replayed
// minibatch finds its keys' rows where it left them and only gathers w
// want_w = false: the caller reads w where it lives (only a first resolve pulls it anyway);
// *any_fresh: a table resolved the list anew (first use, or its rows were renumbered)
__global__ void __launch_bounds__(kBlock)
k_fm_ridx(const uint32_t *uidx, const uint32_t *__restrict__ rows_v, size_t n, uint32_t *ridx) {
  // (ridx may be uidx: the translation of a keyed minibatch's index after a renumbering)
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
       j += (size_t)gridDim.x * blockDim.x)
    ridx[j] = rows_v[uidx[j]];
}

__global__ void __launch_bounds__(kBlock)
k_fm_row_map(const uint32_t *__restrict__ old_rows, const uint32_t *__restrict__ new_rows,
             size_t n, uint32_t *__restrict__ map) {
  for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < n;
       u += (size_t)gridDim.x * blockDim.x)
    map[old_rows[u]] = new_rows[u];
}

static int fm_resolve_rows(xf_table *w, xf_table *vt, xf_batch *b, xf_workspace *ws,
                           void *stream, bool want_w = true, bool *any_fresh = nullptr) {
  const xf_dev_batch &v = b->view;
  xf_table *tabs[2] = {w, vt};
  if (b->fm_keyed) {
    // compiled against these tables' settled tiers (xf_batch_compile_fm*): the rows came with
    // the key list.  After a renumbering (defrag) the key list's rows are looked up again and
    // the per-nonzero record index — there is no CSR index of the key list to rebuild it from —
    // is translated through a map old row -> new row.
    bool moved = false;
    for (int i = 0; i < 2; ++i) {
      XF_REQUIRE(b->fm_uid[i] == xf::table_uid(tabs[i]),
                 "this FM minibatch was compiled against the row numbering of other tables "
                 "(xf_batch_compile_fm*): %s is not the one it was compiled for",
                 i ? "the v table" : "the w table");
      moved = moved || b->fm_epoch[i] != xf::table_epoch(tabs[i]);
    }
    if (moved && v.U) {
      xf::Scratch sc;
      uint32_t *old_v = nullptr, *map = nullptr;
      XF_TRY(sc.get(&old_v, (size_t)v.U));
      XF_TRY(sc.get(&map, (size_t)b->fm_nbase + 1));
      XF_HIP(hipMemcpyAsync(old_v, b->d_fm_rows[1], (size_t)v.U * 4, hipMemcpyDeviceToDevice,
                            S(stream)));
      XF_TRY(xf_table_pull_dev(w, v.ukeys, v.U, b->d_fm_rows[0], ws->wu, stream));
      XF_TRY(xf_table_resolve_dev(vt, v.ukeys, v.U, b->d_fm_rows[1], stream));
      hipLaunchKernelGGL(k_fm_row_map, dim3(blocks_for_groups(v.U, kBlock)), dim3(kBlock), 0,
                         S(stream), old_v, b->d_fm_rows[1], (size_t)v.U, map);
      hipLaunchKernelGGL(k_fm_ridx, dim3(blocks_for_groups(v.NNZ, kBlock)), dim3(kBlock), 0,
                         S(stream), b->d_fm_ridx, map, (size_t)v.NNZ, b->d_fm_ridx);
      XF_HIP(hipGetLastError());
      XF_HIP(hipStreamSynchronize(S(stream)));  // (the scratch goes back)
      for (int i = 0; i < 2; ++i) b->fm_epoch[i] = xf::table_epoch(tabs[i]);
      b->fm_ridx_uid = xf::table_uid(vt);
      b->fm_ridx_epoch = xf::table_epoch(vt);
      b->fm_same_rows = false;  // (the two tables number their rows independently)
      if (any_fresh) *any_fresh = true;
      return XF_OK;
    }
    if (want_w && v.U)
      XF_TRY(xf::gather_f32(xf::table_dev(w).w, b->d_fm_rows[0], v.U, ws->wu, S(stream)));
    return XF_OK;
  }
  for (int i = 0; i < 2 && v.U; ++i) {
    const uint64_t uid = xf::table_uid(tabs[i]), ep = xf::table_epoch(tabs[i]);
    const bool fresh = !b->d_fm_rows[i] || b->fm_uid[i] != uid || b->fm_epoch[i] != ep;
    if (!b->d_fm_rows[i])
      XF_TRY(xf::blob_alloc((void **)&b->d_fm_rows[i], (size_t)v.U * 4, &b->fm_rows_bytes[i]));
    if (fresh) {
      if (i == 0) XF_TRY(xf_table_pull_dev(w, v.ukeys, v.U, b->d_fm_rows[0], ws->wu, stream));
      else
        XF_TRY(xf_table_resolve_dev(vt, v.ukeys, v.U, b->d_fm_rows[1], stream));
      b->fm_uid[i] = uid;
      b->fm_epoch[i] = ep;
      if (any_fresh) *any_fresh = true;
    } else if (i == 0 && want_w) {
      XF_TRY(xf::gather_f32(xf::table_dev(w).w, b->d_fm_rows[0], v.U, ws->wu, S(stream)));
    }
  }
  return XF_OK;
}

// ---- the forward's per-key records kept at the v table's rows ("REC mode")
// The step below used to begin with a pass over the minibatch's factor rows (gather U x k x 4 B,
// write them to scratch with the 32-byte records: 1.2 GB, ~205 us at k = 16) although the only
// thing that changes a row between two steps is the gradient + Push kernel — which has the new
// factors in registers.  So the records live in an array of the v table (one per state row),
// that kernel rewrites the records of the keys it steps, the forward gathers records by the
// nonzeros' v rows, and the gradient kernel reads its factors from the table: no scratch copy
// of the rows, no pass before the forward.  The records of a minibatch's keys are rebuilt from
// the tables (k_fm_gather_scalars, to the rows) when they cannot be trusted: first use of the
// minibatch, a renumbering, a reallocation of the table's state, or any weight write by code
// that does not maintain them (xf_table_update*, import, the unfused paths) since this
// minibatch last looked.  XF_FM_TABLE_RECORDS=0 turns the mode off (a measuring aid).
static bool fm_table_records_enabled() {
  static const bool on = [] {
    const char *e = getenv("XF_FM_TABLE_RECORDS");
    return !(e && *e == '0');
  }();
  return on;
}

// brings the records of b's keys up to date if needed; *rec_out = the v table's records
static int fm_prepare_records(xf_table *w, xf_table *vt, xf_batch *b, bool fresh, int k,
                              FmKey **rec_out, void *stream) {
  const xf_dev_batch &v = b->view;
  void *recp = nullptr;
  uint64_t gen = 0;
  XF_TRY(xf::table_records(vt, sizeof(FmKey), xf::table_uid(w), &recp, &gen));
  const uint64_t uidv = xf::table_uid(vt), epv = xf::table_epoch(vt);
  if (!b->d_fm_ridx || b->fm_ridx_uid != uidv || b->fm_ridx_epoch != epv) {
    XF_REQUIRE(!b->fm_keyed, "fm_prepare_records: a keyed minibatch without its record index");
    if (!b->d_fm_ridx)
      XF_TRY(xf::blob_alloc((void **)&b->d_fm_ridx, (size_t)v.NNZ * 4, &b->fm_ridx_bytes));
    hipLaunchKernelGGL(k_fm_ridx, dim3(blocks_for_groups(v.NNZ, kBlock)), dim3(kBlock), 0,
                       S(stream), v.uidx, b->d_fm_rows[1], (size_t)v.NNZ, b->d_fm_ridx);
    XF_HIP(hipGetLastError());
    b->fm_ridx_uid = uidv;
    b->fm_ridx_epoch = epv;
    fresh = true;
  }
  const uint64_t w0 = xf::table_writes(w), w1 = xf::table_writes(vt);
  // A minibatch compiled against the settled tiers (xf_batch_compile_fm_dev: all its keys are
  // settled, rank r = state row r in both tables) rides on the TABLE's records: one pass over
  // all settled rows (10^7 rows: ~0.3 ms) the first time, nothing afterwards for as long as
  // only this step (which rewrites the records of the keys it steps) writes the two tables —
  // instead of a pass over its own keys' rows (0.2 ms) before every fresh minibatch's step.
  const uint64_t allkey[5] = {gen, w0, w1, xf::table_epoch(w), epv};
  const bool all_rows = b->fm_keyed && b->fm_same_rows && xf::table_dev(vt).nbase > 0;
  if (all_rows && xf::table_records_all_is(vt, allkey)) {
    b->fm_rec_ok = true;
    b->fm_rec_gen = gen;
    b->fm_rec_writes[0] = w0;
    b->fm_rec_writes[1] = w1;
  } else if (fresh || !b->fm_rec_ok || b->fm_rec_gen != gen || b->fm_rec_writes[0] != w0 ||
             b->fm_rec_writes[1] != w1) {
    const int dim4 = k / 4;
    const float4 *tv = (const float4 *)xf::table_dev(vt).w;
    const size_t nrows = all_rows ? (size_t)xf::table_dev(vt).nbase : (size_t)v.U;
    const uint32_t *rv = all_rows ? nullptr : b->d_fm_rows[1];
    const uint32_t *rw = all_rows ? nullptr : b->d_fm_rows[0];
    const size_t tot = nrows * dim4;
    const dim3 g((unsigned)std::min<size_t>((tot + kBlock - 1) / kBlock, 8192)), blk(kBlock);
#define XF_FM_GS(D)                                                                           \
  hipLaunchKernelGGL(k_fm_gather_scalars<D>, g, blk, 0, S(stream), tv, rv, xf::table_dev(w).w, \
                     rw, nrows, (float4 *)nullptr, (FmKey *)recp, rv)
    switch (dim4) {
      case 1: XF_FM_GS(1); break;
      case 2: XF_FM_GS(2); break;
      case 4: XF_FM_GS(4); break;
      case 8: XF_FM_GS(8); break;
      default: XF_FM_GS(16); break;
    }
#undef XF_FM_GS
    XF_HIP(hipGetLastError());
    b->fm_rec_ok = true;
    b->fm_rec_gen = gen;
    b->fm_rec_writes[0] = w0;
    b->fm_rec_writes[1] = w1;
    if (all_rows) xf::table_records_all_set(vt, allkey);
  }
  *rec_out = (FmKey *)recp;
  return XF_OK;
}

extern "C" int xf_fm_step(xf_table *w, xf_table *vt, xf_batch *b, xf_workspace *ws,
                          void *stream) {
  XF_REQUIRE(w && vt && b && ws, "xf_fm_step: null argument");
  XF_REQUIRE(xf::table_dim(w) == 1, "xf_fm_step: the w table must have dim 1");
  XF_REQUIRE(!b->local, "xf_fm_step: needs a minibatch with a key list (xf_batch_compile*)");
  const int k = xf::table_dim(vt);
  XF_TRY(xf_batch_upload(b, stream));
  XF_TRY(ws_reserve(ws, b->U, (size_t)b->U * k, b->R));
  ws->rec = ws->profiling && ws->step_no++ % xf_workspace::kProfileEvery == 0;
  if (ws->rec) XF_TRY(ws_next_set(ws));
  const xf_dev_batch &v = b->view;
  ws->lastU = b->U;
  ws->lastR = b->R;
  XF_BEGIN();
  const bool recmode = fm_table_records_enabled() && xf::fm_records_fit(k) && !ws->capture &&
                       ws->parity == XF_PARITY_EXACT_SUMS && v.U && v.R;
  XF_REQUIRE(!b->fm_keyed || recmode || !v.U,
             "a minibatch of xf_batch_compile_fm_dev steps on the table-resident records only "
             "(k in {4, 8, 16, 32, 64}, no capture, no parity mode): it has no index of its key "
             "list.  Compile it again after the parity / capture mode is set (xf_sharded_compile "
             "then picks the sort-based build), or with xf_batch_compile_dev");
  bool fresh = false;
  XF_TRY(fm_resolve_rows(w, vt, b, ws, stream, !recmode, &fresh));
  const uint32_t *rows_w = v.U ? b->d_fm_rows[0] : ws->slots;
  const uint32_t *rows_v = v.U ? b->d_fm_rows[1] : ws->slots2;
  if (recmode) {
    ws->lastU = 0;  // no per-key intermediates in scratch (xf_workspace_capture keeps them)
    FmKey *rec = nullptr;
    XF_TRY(fm_prepare_records(w, vt, b, fresh, k, &rec, stream));
    XF_END(kEvResolve);
    hipLaunchKernelGGL(k_fm_forward_scalars, dim3(blocks_for_groups(v.R, kBlock / 64)),
                       dim3(kBlock), 0, S(stream), v.rowptr, b->d_fm_ridx, (const FmKey *)rec,
                       v.labels, v.R, ws->loss, (float *)nullptr, ws->vsum);  // :237
    XF_HIP(hipGetLastError());
    XF_END(kEvForward);
    XF_TRY(fm_grad_update(w, vt, &v, rows_w, rows_v, ws->wu, ws->vu, ws->vsum, ws->loss, ws->g,
                          ws->gv, false, stream, rec));
    XF_END(kEvGrad);
    if (ws->rec) ws->sets[ws->cur].pending = true;
    return XF_OK;
  }
  XF_END(kEvResolve);
  const int dim4 = k / 4;
  const bool scalars = k % 4 == 0 && dim4 <= 16 && (dim4 & (dim4 - 1)) == 0 && v.U && v.R;
  if (scalars) {  // v-row gather that also forms the per-key scalars of the forward
    const float4 *tv = (const float4 *)xf::table_dev(vt).w;
    const size_t tot = (size_t)v.U * dim4;
    const dim3 g((unsigned)std::min<size_t>((tot + kBlock - 1) / kBlock, 8192)), blk(kBlock);
#define XF_FM_GS(D)                                                                        \
  hipLaunchKernelGGL(k_fm_gather_scalars<D>, g, blk, 0, S(stream), tv, rows_v, ws->wu, \
                     (const uint32_t *)nullptr, (size_t)v.U, (float4 *)ws->vu,          \
                     (FmKey *)ws->ks, (const uint32_t *)nullptr)
    switch (dim4) {
      case 1: XF_FM_GS(1); break;
      case 2: XF_FM_GS(2); break;
      case 4: XF_FM_GS(4); break;
      case 8: XF_FM_GS(8); break;
      default: XF_FM_GS(16); break;
    }
#undef XF_FM_GS
    XF_HIP(hipGetLastError());
    XF_END(kEvGather);
    if (ws->parity == XF_PARITY_REFERENCE_ORDER)
      XF_TRY(fm_forward_reforder(b, k, ws->wu, ws->vu, ws->loss, nullptr, ws->vsum, S(stream)));
    else
      hipLaunchKernelGGL(k_fm_forward_scalars, dim3(blocks_for_groups(v.R, kBlock / 64)),
                         dim3(kBlock), 0, S(stream), v.rowptr, v.uidx, (const FmKey *)ws->ks,
                         v.labels, v.R, ws->loss, (float *)nullptr, ws->vsum);  // :237
    XF_HIP(hipGetLastError());
  } else {
    XF_TRY(xf_table_gather_dev(vt, rows_v, v.U, ws->vu, stream));
    XF_END(kEvGather);
    if (ws->parity == XF_PARITY_REFERENCE_ORDER)
      XF_TRY(fm_forward_reforder(b, k, ws->wu, ws->vu, ws->loss, nullptr, ws->vsum, S(stream)));
    else
      XF_TRY(xf_fm_forward_dev(&v, k, ws->wu, ws->vu, ws->loss, nullptr, ws->vsum, stream));
  }
  XF_END(kEvForward);
  if (ws->parity == XF_PARITY_REFERENCE_ORDER && v.U) {
    // the reference's own fp32 running sums per key and factor, then the two Pushes
    XF_TRY(xf::batch_reference_coo(b, S(stream)));
    hipLaunchKernelGGL(k_fm_grad_reforder, dim3(blocks_for_groups(v.U, kBlock / 64)), dim3(kBlock),
                       0, S(stream), v.segptr, b->d_ref_coo, ws->loss, ws->vsum, ws->vu, k, v.U,
                       v.R, ws->g, ws->gv);  // :238
    XF_HIP(hipGetLastError());
    XF_TRY(xf_table_update_dev(w, rows_w, v.U, ws->g, stream));    // :241
    XF_TRY(xf_table_update_dev(vt, rows_v, v.U, ws->gv, stream));  // :242
    XF_END(kEvGrad);
    if (ws->rec) ws->sets[ws->cur].pending = true;
    return XF_OK;
  }
  // gradient (:238) and the two Pushes (:241-242) in one pass: both tables are on this GPU
  XF_TRY(fm_grad_update(w, vt, &v, rows_w, rows_v, ws->wu, ws->vu, ws->vsum, ws->loss,
                        ws->g, ws->gv, false, stream));
  XF_END(kEvGrad);
  if (ws->rec) ws->sets[ws->cur].pending = true;
  return XF_OK;
}

// ------------------------------------------------ FM at the owner of a share of the keys
// Sharded owner-compute dataflow (xf_sharded.hip): `b` holds the nonzeros of EVERY worker's
// rows whose keys this GPU owns, rows numbered worker after worker.  The forward's three row
// sums (fm_worker.cc:166-192: wx, v_sum, v_pow_sum) are sums over the row's keys, so every
// owner forms its share — exactly, in fp64 — and the rows' worker adds the shares before the
// rounding steps of :194-199.
__global__ void __launch_bounds__(kBlock)
k_fm_key_scalars_any(const float *__restrict__ vu, const float *__restrict__ wu, uint32_t U,
                     int k, FmKey *__restrict__ ks) {
#pragma clang fp contract(off)
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  double a = 0.0, b = 0.0;
  for (int j = 0; j < k; ++j) {
    const float v = vu[(size_t)u * k + j];
    a += (double)v;
    b += (double)(v * v);  // fp32 product, as fm_worker.cc:187
  }
  FmKey q;
  q.a = a;
  q.b = b;
  q.w = wu[u];
  q.pad[0] = q.pad[1] = q.pad[2] = 0.f;
  ks[u] = q;
}

// one wavefront per row: out[3 r .. 3 r + 2] = this owner's share of (wx, v_sum, v_pow_sum)
__global__ void __launch_bounds__(kBlock)
k_fm_row_partials(const uint32_t *__restrict__ rowptr, const uint32_t *__restrict__ uidx,
                  const FmKey *__restrict__ ks, uint32_t R, double *__restrict__ out) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nwaves = gridDim.x * (kBlock / 64);
  for (uint32_t r = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; r < R; r += nwaves) {
    const uint32_t b = rowptr[r], e = rowptr[r + 1];
    double wx = 0.0, vs = 0.0, vp = 0.0;
    for (uint32_t j0 = b + lane; j0 < e; j0 += 64 * 4) {
      uint32_t ui[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) ui[i] = j0 + 64 * i < e ? uidx[j0 + 64 * i] : 0xFFFFFFFFu;
      FmKey q[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ui[i] != 0xFFFFFFFFu) q[i] = ks[ui[i]];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ui[i] != 0xFFFFFFFFu) {
          wx += (double)q[i].w;
          vs += q[i].a;
          vp += q[i].b;
        }
    }
    wx = group_sum<64>(wx);
    vs = group_sum<64>(vs);
    vp = group_sum<64>(vp);
    if (lane == 0) {
      out[(size_t)r * 3 + 0] = wx;
      out[(size_t)r * 3 + 1] = vs;
      out[(size_t)r * 3 + 2] = vp;
    }
  }
}

namespace xf {
// the owner's Pull (key -> row once per row numbering, rows gathered) and its share of the
// row sums: d_part[3 * b->R]
// pulled_copies: the rows are gathered into the workspace (what a worker's Pull returns) whatever
// the record mode — the rank-ordered update rule forms every worker's gradient from them
int fm_owner_partials(xf_table *w, xf_table *vt, xf_batch *b, xf_workspace *ws, double *d_part,
                      hipStream_t s, bool pulled_copies) {
  XF_REQUIRE(w && vt && b && ws && d_part && !b->local, "fm_owner_partials: bad argument");
  const int k = xf::table_dim(vt);
  XF_TRY(xf_batch_upload(b, s));
  XF_TRY(ws_reserve(ws, b->U, (size_t)b->U * k, b->R));
  const xf_dev_batch &v = b->view;
  if (!v.R) return XF_OK;
  if (!v.U) {
    XF_HIP(hipMemsetAsync(d_part, 0, (size_t)v.R * 24, s));
    return XF_OK;
  }
  const bool recmode = fm_table_records_enabled() && fm_records_fit(k) && !ws->capture &&
                       ws->parity == XF_PARITY_EXACT_SUMS && !pulled_copies;
  bool fresh = false;
  XF_TRY(fm_resolve_rows(w, vt, b, ws, s, !recmode, &fresh));
  ws->owner_rec = nullptr;
  if (recmode) {  // records at the v table's rows, kept up to date by the gradient + Push kernel
    FmKey *rec = nullptr;
    XF_TRY(fm_prepare_records(w, vt, b, fresh, k, &rec, s));
    hipLaunchKernelGGL(k_fm_row_partials, dim3(blocks_for_groups(v.R, kBlock / 64)),
                       dim3(kBlock), 0, s, v.rowptr, b->d_fm_ridx, (const FmKey *)rec, v.R,
                       d_part);
    XF_HIP(hipGetLastError());
    ws->owner_rec = rec;
    return XF_OK;
  }
  const int dim4 = k / 4;
  if (fm_records_fit(k)) {
    const float4 *tv = (const float4 *)xf::table_dev(vt).w;
    const size_t tot = (size_t)v.U * dim4;
    const dim3 g((unsigned)std::min<size_t>((tot + kBlock - 1) / kBlock, 8192)), blk(kBlock);
#define XF_FM_GS(D)                                                                         \
  hipLaunchKernelGGL(k_fm_gather_scalars<D>, g, blk, 0, s, tv, b->d_fm_rows[1], ws->wu,     \
                     (const uint32_t *)nullptr, (size_t)v.U, (float4 *)ws->vu,              \
                     (FmKey *)ws->ks, (const uint32_t *)nullptr)
    switch (dim4) {
      case 1: XF_FM_GS(1); break;
      case 2: XF_FM_GS(2); break;
      case 4: XF_FM_GS(4); break;
      case 8: XF_FM_GS(8); break;
      default: XF_FM_GS(16); break;
    }
#undef XF_FM_GS
  } else {
    XF_TRY(xf_table_gather_dev(vt, b->d_fm_rows[1], v.U, ws->vu, s));
    hipLaunchKernelGGL(k_fm_key_scalars_any, dim3((v.U + kBlock - 1) / kBlock), dim3(kBlock), 0,
                       s, ws->vu, ws->wu, v.U, k, (FmKey *)ws->ks);
  }
  hipLaunchKernelGGL(k_fm_row_partials, dim3(blocks_for_groups(v.R, kBlock / 64)), dim3(kBlock),
                     0, s, v.rowptr, v.uidx, (const FmKey *)ws->ks, v.R, d_part);
  XF_HIP(hipGetLastError());
  return XF_OK;
}

// gradient (fm_worker.cc:126-157) over all the rows at once and the two Pushes (:241-242), with
// the rows' (loss, v_sum) as their workers formed them.  Uses the rows pulled by
// fm_owner_partials of the same step (ws->wu, ws->vu).
int fm_owner_grad_update(xf_table *w, xf_table *vt, xf_batch *b, xf_workspace *ws,
                         const float *d_loss, const float *d_vsum, hipStream_t s) {
  XF_REQUIRE(w && vt && b && ws && d_loss && d_vsum, "fm_owner_grad_update: bad argument");
  const xf_dev_batch &v = b->view;
  if (!v.U || !v.R) return XF_OK;
  return fm_grad_update(w, vt, &v, b->d_fm_rows[0], b->d_fm_rows[1], ws->wu, ws->vu, d_vsum,
                        d_loss, ws->g, ws->gv, false, s, (FmKey *)ws->owner_rec);
}

// XF_UPDATE_RANK_ORDERED at the owner: `b` holds ONE worker's nonzeros of this owner's keys.
// The worker's gradient (fm_worker.cc:126-157, its own 1 / R) from the rows its Pull returned
// (fm_owner_partials(.., pulled_copies) of the same step: ws->vu), kept in the workspace ...
int fm_owner_grad_pulled(xf_table *vt, xf_batch *b, xf_workspace *ws, const float *d_loss,
                         const float *d_vsum, hipStream_t s) {
  XF_REQUIRE(vt && b && ws && d_loss && d_vsum, "fm_owner_grad_pulled: bad argument");
  const xf_dev_batch &v = b->view;
  if (!v.U || !v.R) return XF_OK;
  return xf_fm_grad_dev(&v, xf::table_dim(vt), ws->vu, d_vsum, d_loss, ws->g, ws->gv, s);
}
// ... and its two Pushes (fm_worker.cc:241-242 -> ftrl.h:54-74 / sgd.h:52 per key and factor):
// called worker after worker once ALL workers' gradients exist
int fm_owner_push_pulled(xf_table *w, xf_table *vt, xf_batch *b, xf_workspace *ws,
                         hipStream_t s) {
  XF_REQUIRE(w && vt && b && ws, "fm_owner_push_pulled: bad argument");
  const xf_dev_batch &v = b->view;
  if (!v.U || !v.R) return XF_OK;
  XF_TRY(xf_table_update_dev(w, b->d_fm_rows[0], v.U, ws->g, s));
  return xf_table_update_dev(vt, b->d_fm_rows[1], v.U, ws->gv, s);
}
}  // namespace xf

// forward only: calculate_pctr (lr_worker.cc:25-71).  The pull inserts unseen keys, as the
// reference's test-time Pull does (ftrl.h:56).
extern "C" int xf_lr_predict(xf_table *w, xf_batch *b, xf_workspace *ws, float *pctr_out) {
  XF_REQUIRE(w && b && ws && pctr_out, "xf_lr_predict: null argument");
  XF_REQUIRE(xf::table_dim(w) == 1, "xf_lr_predict: the w table must have dim 1");
  XF_TRY(xf::ensure_cells(b, w, nullptr));
  XF_TRY(ws_reserve(ws, b->U, 0, b->R));
  XF_TRY(ws_reserve_cells(ws, b->cells, false));
  const int32_t *labels = b->local ? b->raw_labels : b->view.labels;
  if (ws->parity == XF_PARITY_REFERENCE_ORDER)
    XF_TRY(lr_forward_reforder(w, b, ws->loss, ws->pctr, nullptr));
  else
    XF_TRY(xf::cells_lr_forward(b->cells, xf::table_dev(w).w, labels, ws->partial, ws->loss,
                                ws->pctr, nullptr));
  if (b->R) XF_HIP(hipMemcpy(pctr_out, ws->pctr, (size_t)b->R * 4, hipMemcpyDeviceToHost));
  return xf_table_check(w, nullptr);
}

extern "C" int xf_fm_predict(xf_table *w, xf_table *vt, xf_batch *b, xf_workspace *ws,
                             float *pctr_out) {
  XF_REQUIRE(w && vt && b && ws && pctr_out, "xf_fm_predict: null argument");
  XF_REQUIRE(!b->local, "xf_fm_predict: needs a minibatch with a key list (xf_batch_compile*)");
  const int k = xf::table_dim(vt);
  XF_TRY(xf_batch_upload(b, nullptr));
  XF_TRY(ws_reserve(ws, b->U, (size_t)b->U * k, b->R));
  const xf_dev_batch &v = b->view;
  if (b->fm_keyed) {  // the step's forward (records at the v table's rows), loss left aside
    XF_REQUIRE(fm_table_records_enabled() && xf::fm_records_fit(k) && !ws->capture &&
                   ws->parity == XF_PARITY_EXACT_SUMS,
               "xf_fm_predict: a minibatch of xf_batch_compile_fm* is scored from the "
               "table-resident records only (k in {4, 8, 16, 32, 64}, no capture, no parity "
               "mode): use xf_batch_compile_dev");
    if (v.U && v.R) {
      bool fresh = false;
      XF_TRY(fm_resolve_rows(w, vt, b, ws, nullptr, false, &fresh));
      FmKey *rec = nullptr;
      XF_TRY(fm_prepare_records(w, vt, b, fresh, k, &rec, nullptr));
      hipLaunchKernelGGL(k_fm_forward_scalars, dim3(blocks_for_groups(v.R, kBlock / 64)),
                         dim3(kBlock), 0, S(nullptr), v.rowptr, b->d_fm_ridx, (const FmKey *)rec,
                         v.labels, v.R, ws->loss, ws->pctr, ws->vsum);
      XF_HIP(hipGetLastError());
      XF_HIP(hipMemcpy(pctr_out, ws->pctr, (size_t)b->R * 4, hipMemcpyDeviceToHost));
    }
    XF_TRY(xf_table_check(w, nullptr));
    return xf_table_check(vt, nullptr);
  }
  XF_TRY(xf_table_pull_dev(w, v.ukeys, v.U, ws->slots, ws->wu, nullptr));
  XF_TRY(xf_table_resolve_dev(vt, v.ukeys, v.U, ws->slots2, nullptr));
  XF_TRY(xf_table_gather_dev(vt, ws->slots2, v.U, ws->vu, nullptr));
  if (ws->parity == XF_PARITY_REFERENCE_ORDER)
    XF_TRY(fm_forward_reforder(b, k, ws->wu, ws->vu, ws->loss, ws->pctr, ws->vsum, nullptr));
  else if (xf::fm_records_fit(k) && v.U)
    XF_TRY(xf::fm_forward_records(&v, k, ws->wu, ws->vu, ws->ks, ws->loss, ws->pctr, ws->vsum,
                                  nullptr));
  else
    XF_TRY(xf_fm_forward_dev(&v, k, ws->wu, ws->vu, ws->loss, ws->pctr, ws->vsum, nullptr));
  if (b->R) XF_HIP(hipMemcpy(pctr_out, ws->pctr, (size_t)b->R * 4, hipMemcpyDeviceToHost));
  XF_TRY(xf_table_check(w, nullptr));
  return xf_table_check(vt, nullptr);
}

// parity hook: intermediates of the last step
extern "C" int xf_workspace_fetch(xf_workspace *ws, float *wu, float *loss, float *g, size_t U,
                                  size_t R) {
  XF_REQUIRE(ws, "xf_workspace_fetch: null workspace");
  XF_REQUIRE(R <= ws->lastR && (!(wu || g) || U <= ws->lastU),
             "xf_workspace_fetch: sizes exceed the last step's (per-key intermediates of a "
             "step exist only after xf_workspace_capture(ws, 1))");
  XF_HIP(hipDeviceSynchronize());
  if (wu && U) XF_HIP(hipMemcpy(wu, ws->wu, U * 4, hipMemcpyDeviceToHost));
  if (loss && R) XF_HIP(hipMemcpy(loss, ws->loss, R * 4, hipMemcpyDeviceToHost));
  if (g && U) XF_HIP(hipMemcpy(g, ws->g, U * 4, hipMemcpyDeviceToHost));
  return XF_OK;
}

extern "C" int xf_stream_sync(void *stream) {
  XF_HIP(hipStreamSynchronize(S(stream)));
  return XF_OK;
}
