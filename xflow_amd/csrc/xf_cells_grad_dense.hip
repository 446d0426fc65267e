// xf_cells_grad_dense.hip — the gradient + Push of the steady state (one source, no split chunk,
// at most four row windows): k_lr_grad_dense, the launch that takes most of the LR step (gfx950).
//
// Replaces (paths relative to /root/reference):
//   LRWorker::calculate_gradient       src/model/lr/lr_worker.cc:100-119
//   KVWorker::Push -> FTRL / SGD       src/optimizer/ftrl.h:54-74, sgd.h:52 (fused in the same)
// HBM-bound integer/byte work, no MFMA.
#include "xf_cells_grad.h"

namespace {
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
using xf::kDenseCompact;
using xf::kDensePrefetch;
using xf::kDenseWide;
using xf::kDiagNoStore;
using xf::kDiagNoUpdate;
using xf::kDiagNoAccum;
using xf::kDiagCopy;
using xf::kDenseFullStore;
using xf::kDenseQuad;
using xf::kDenseWin;

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
      sw[i] = 0.0f;
      sn[i] = sz[i] = 0.0f;
      // (a slot NO lane of the wavefront fills — cnt is the wavefront's, the test a scalar one —
      // issues nothing: a load costs its lanes' TA cycles whatever it hits, and at a tenth of
      // the rows touched seven of the eight slots are idle.  Inside a slot in use an idle lane
      // loads the chunk's first row: a load under a divergent branch would have to be waited for
      // where the branch ends, one slot after the other)
      if (64u * (uint32_t)i < cnt) {
        const size_t r = row0 + (kk[i] != 0xFFFFFFFFu ? kk[i] : 0u);
        if (!wnz) sw[i] = T.w[r];
        if (OPT == XF_OPT_FTRL) xf::load_nz(T, r, sn[i], sz[i]);
      }
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
}  // namespace

namespace xf {

int cells_launch_grad_dense(int opt, int var, const xf_cells *c, const TableDev &Td,
                            const float *d_loss, hipStream_t s) {
#define XF_DENSE_O(O, V)                                                                         \
  hipLaunchKernelGGL((k_lr_grad_dense<O, V>),                                                    \
                     dim3(((V) & kDenseQuad) && dense_team(V) * 4 <= 1024 ? (c->nitems + 3) / 4  \
                                                                          : c->nitems),          \
                     dim3(dense_threads(V)), 0, s, Td, c->entries, c->cellptr, c->nchunk,        \
                     c->nwin, c->W, c->item_chunk, d_loss, c->R, c->M, c->chunk0, c->nitems)
#define XF_DENSE(V)                              \
  case V:                                        \
    if (opt == XF_OPT_FTRL) XF_DENSE_O(XF_OPT_FTRL, V); \
    else                                         \
      XF_DENSE_O(XF_OPT_SGD, V);                 \
    break
  switch (var) {
    XF_DENSE(0);
    XF_DENSE(128);
#ifdef XF_EXPERIMENTS  // the measured-and-not-adopted variants (round 6: prefetch, 512
                       // threads and both on a 1e8-key table, tools/r6/sweep_variants.py) and the
                       // timing experiments (some with WRONG results): never in a product build
    XF_DENSE(1);
    XF_DENSE(2);
    XF_DENSE(4);
    XF_DENSE(6);
    XF_DENSE(128 + 4);
    XF_DENSE(5);
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
      return xf::set_error(XF_EINVAL, "gradient kernel variant %d (the timing-only variants need "
                           "a library built with -DXF_EXPERIMENTS)", var);
  }
#undef XF_DENSE
#undef XF_DENSE_O
  XF_HIP(hipGetLastError());
  return XF_OK;
}

}  // namespace xf
