// xf_cells_grad.hip — the LR gradient (+ Push) over cells: the general kernel, the passes of an
// owner of several workers, and the choice of the kernel for a minibatch (gfx950).
//
// Replaces (paths relative to /root/reference):
//   LRWorker::calculate_gradient       src/model/lr/lr_worker.cc:100-119  (k_lr_grad_cells, ...)
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
#include "xf_cells_grad.h"

namespace {
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
        int var = dense_touch(c) ? kDenseFullStore : 0;
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
        if (dense) XF_TRY(cells_launch_grad_dense(OPT, var, c, Td, d_loss, s));
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

}  // namespace xf
