// xf_sharded.hip — LRWorker / FMWorker::update across `world` GPUs, one process per GPU, with
// the parameter table sharded by key range (gfx950 + RCCL).
//
// Replaces ps-lite's worker <-> server routing (src/model/lr/lr_worker.cc:170,175;
// src/model/fm/fm_worker.cc:228-242): Pull(keys) = keys to their owners, weights back;
// Push(keys, grads) = gradients to the owners, owner-side FTRL / SGD (ftrl.h:54-74, sgd.h:52).
// Like ps-lite's default slicer, a SORTED key list splits into one contiguous range per owner
// (owner = min(key / (UINT64_MAX / N), N-1)), so every exchange is a plain all-to-all-v without
// permutation (xf_group_alltoallv: grouped ncclSend/ncclRecv over xGMI).
//
// What depends only on the minibatch is exchanged ONCE, when it is compiled: the split
// counts, the keys (so a step moves weights one way and gradients the other, never keys), the
// order in which the owner walks all sources' keys merged by key, and — cached per table row
// numbering — the owner-side state rows of those keys (the Pull's resolve).
//
// Update semantics with N workers: every worker's gradient is its own optimizer
// step, scaled by its own 1/R (lr_worker.cc:116-118); the owner applies the N pushes of a
// step in RANK ORDER after all N pulls — one legal, deterministic serialisation of what
// ps-lite does asynchronously.  Two schedules:
//   sequential  Pull, compute, Push of step t finish before Pull(t+1)
//   stale1      Push(t) runs on a second stream: its gradient exchange overlaps Pull(t+1) on
//               the owner, its optimizer pass — ordered by events after Pull(t+1) has read
//               the table and before Pull(t+2) — overlaps the next weights exchange, forward
//               and gradient.  Weights are exactly one step stale (inside ps-lite's
//               asynchronous semantics), still deterministic.
// With world == 1 there is nothing to exchange: the step IS the fused single-shard step
// (xf_lr_step / xf_fm_step) on a locally compiled minibatch.
#include <hip/hip_runtime.h>


#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <thread>
#include <vector>

#include "xf_batch.h"
#include "xf_cells.h"
#include "xf_common.h"
#include "xf_device.h"
#include "xf_scratch.h"

namespace xf {
int cells_lr_grad(const xf_cells *c, const float *d_loss, float *d_g, hipStream_t s);
uint64_t table_uid(const xf_table *t);
uint64_t table_epoch(const xf_table *t);
int table_ensure_room(xf_table *t, size_t incoming);
int table_resolve_any(xf_table *t, const uint64_t *d_keys, size_t n, uint32_t *d_rows,
                      hipStream_t s, bool allow_grow);
uint64_t table_row_bound(const xf_table *t);
const float *table_weights(const xf_table *t);
bool fm_records_fit(int k);
size_t fm_record_bytes(size_t U);
int fm_owner_partials(xf_table *w, xf_table *vt, xf_batch *b, xf_workspace *ws, double *d_part,
                      hipStream_t s, bool pulled_copies = false);
int fm_owner_grad_pulled(xf_table *vt, xf_batch *b, xf_workspace *ws, const float *d_loss,
                         const float *d_vsum, hipStream_t s);
int fm_owner_push_pulled(xf_table *w, xf_table *vt, xf_batch *b, xf_workspace *ws,
                         hipStream_t s);
int fm_owner_grad_update(xf_table *w, xf_table *vt, xf_batch *b, xf_workspace *ws,
                         const float *d_loss, const float *d_vsum, hipStream_t s);
int fm_forward_records(const xf_dev_batch *b, int k, const float *d_wu, const float *d_vu,
                       void *d_ks, float *d_loss, float *d_pctr, float *d_vsum, hipStream_t s);
const TableDev &table_dev(const xf_table *t);
int table_head_rows(const uint64_t *d_keys_sorted, const uint32_t *d_order,
                    const uint32_t *d_rows, size_t n, uint32_t *d_hrow, hipStream_t s);
int table_update_heads(xf_table *t, const uint32_t *d_hrow, const uint32_t *d_order, size_t n,
                       const float *d_grads, hipStream_t s);
}  // namespace xf

namespace {

constexpr int kBlock = 256;
inline int grid_for(size_t n) {
  size_t g = (n + kBlock - 1) / kBlock;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

// split[p] = first index of the sorted unique keys that belongs to shard >= p (p = 0..world)
__global__ void k_owner_split(const uint64_t *__restrict__ ukeys, uint32_t U, uint32_t world,
                              uint32_t *__restrict__ split) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > world) return;
  if (p == 0 || p == world) {
    split[p] = p == 0 ? 0u : U;
    return;
  }
  const uint64_t bound = (UINT64_MAX / world) * p;  // first key of shard p (xf_common.h)
  uint32_t lo = 0, hi = U;
  while (lo < hi) {
    const uint32_t mid = lo + (hi - lo) / 2;
    if (ukeys[mid] < bound) lo = mid + 1;
    else
      hi = mid;
  }
  split[p] = lo;
}

template <typename T>
struct Dev {
  // (from the pool of device allocations the minibatches' arrays come from, xf::blob_alloc: a
  // fresh minibatch per step — the first epoch — would otherwise pay a dozen hipMalloc and as
  // many hipFree, each of which waits for the device)
  T *p = nullptr;
  size_t n = 0, bytes = 0;
  Dev() = default;
  Dev(const Dev &) = delete;
  Dev &operator=(const Dev &) = delete;
  ~Dev() {
    if (p && !xf::device_poisoned()) xf::blob_free(p, bytes);  // (poisoned: leaked, see wait_stream)
  }
  int reserve(size_t want) {
    if (want <= n && p) return XF_OK;
    if (p) {  // growing: work in flight may still read the old array (hipFree used to wait)
      XF_HIP(hipDeviceSynchronize());
      xf::blob_free(p, bytes);
    }
    p = nullptr;
    n = 0;
    bytes = 0;
    XF_TRY(xf::blob_alloc((void **)&p, std::max<size_t>(want, 1) * sizeof(T), &bytes));
    n = want;
    return XF_OK;
  }
};

enum { kEvPull = 0, kEvA2aW, kEvForward, kEvGrad, kEvA2aG, kEvUpdate, kEvN };

}  // namespace

// per-step buffers of one minibatch; two sets, used alternately, so that a Push still in
// flight on the side stream (stale1) never shares a buffer with the next step on this batch
struct StepBuf {
  Dev<float> w_send, wu, g, g_recv, loss, vsum;
  Dev<char> ks;  // FM: the per-key records of the forward
  Dev<float> v_send, vu, gv, gv_recv;
};

struct xf_sbatch {
  xf_batch *b = nullptr;      // the compiled minibatch (world 1: compiled against the table)
  xf_cells *cells = nullptr;  // world > 1, LR: the cells over the batch's unique-key index
  uint32_t R = 0, NNZ = 0, U = 0;
  std::vector<uint64_t> send_counts, recv_counts;  // per peer: keys I send / keys I own
  size_t n_recv = 0;
  Dev<uint64_t> rkeys, rkeys_sorted;  // the keys this rank owns, per source; merged by key
  Dev<uint32_t> rorder;               // sorted entry i sits at rorder[i] of the per-source layout
  Dev<uint32_t> rows_w, rows_v;       // their state rows (valid for one table epoch)
  Dev<uint32_t> hrow_w, hrow_v;       // merged order: row of a key's first entry, else kNotHead
  uint64_t rows_uid = 0, rows_epoch = ~0ull;
  StepBuf buf[2];
  int flip = 0;
  xf_sharded *owner = nullptr;
  // ---- XF_SCHEDULE_OWNER: this rank's share of EVERY worker's nonzeros (the ones whose keys it
  // owns), rows numbered window by window across the workers
  bool oc = false, oc_keep = true;           // oc_keep: will be replayed (key-sorted forward copy)
  uint32_t oW = 1, o_rpad = 0;               // rows per window; windows * oW
  std::vector<uint32_t> o_rows, o_win;       // per worker: rows of its minibatch; first window
  std::vector<uint64_t> o_rows64;            // o_rows as exchange counts
  size_t o_n = 0;                            // nonzeros received
  Dev<uint64_t> o_keys;                      // their keys (kept: re-resolved after a defrag)
  Dev<uint32_t> o_rowid;                     // their rows (window-major numbering)
  // the small per-minibatch tables: one device allocation, one host copy kept beside it (the
  // upload reads it asynchronously); the names below point into o_small
  struct U32View {
    uint32_t *p = nullptr;
  };
  Dev<uint32_t> o_small;
  std::vector<uint32_t> o_small_host;
  U32View d_win, d_rows;                     // o_win, o_rows on the device
  U32View d_win1, d_rows1;                   // {0, windows}, {all rows}: sum_then_step
  U32View d_winT, d_rowsT;                   // measuring aid (XF_OWNER_TIMING_SOURCES, world 1)
  uint32_t nT = 0;
  U32View d_wbase, d_wrows;                  // per window: first row in the back-to-back
                                             // layout of the workers' rows; rows it holds
  Dev<int32_t> d_labels;                     // this worker's labels
  xf_cells *ocells = nullptr;                // cells over the shard's state rows
  uint64_t oc_uid = 0, oc_epoch = ~0ull;
  Dev<double> rs_send, rs_recv, gsum;
  Dev<float> oloss, loss_rep, loss_recv, opctr;
  Dev<float> loss_recv2;                     // owner_stale1: the gradient pass of step t reads one
  int oflip = 0;                             // buffer while the losses of step t+1 arrive in the other
  Dev<uint8_t> gtouched;
  // FM, XF_UPDATE_RANK_ORDERED: one minibatch (with a key list) and one workspace PER WORKER —
  // its Pull's copies of the rows and its gradients live there from the forward to the Pushes;
  // q_rowoff[q] = first of worker q's rows in the back-to-back layout
  std::vector<xf_batch *> bq;
  std::vector<xf_workspace *> wsq;
  // owner_stale1, FM: the workspaces of the step whose Pushes are still outstanding — its pulled
  // rows and gradients — while the next step of the SAME minibatch pulls into the other set
  std::vector<xf_workspace *> wsq2;
  std::vector<uint32_t> q_rowoff;
  // FM (sum_then_step): b = the received nonzeros as one minibatch with a key list, rows
  // numbered worker after worker; (loss, v_sum) pairs of the rows
  uint32_t o_total = 0;                      // rows of all workers
  uint32_t o_rows_all = 0;                   // the same, LR (rows1 on the host)
  Dev<float> lv, lv_rep, lv_recv, vsum_recv;
};

struct xf_sharded {
  xf_group *g = nullptr;
  int rank = 0, world = 1;
  bool poisoned = false;  // a stream wait timed out: every later call fails fast
  bool fused = true;  // world 1: the fused single-shard step (XF_SHARDED_GENERAL=1 runs the
                      // exchange path with its self-copies instead: a measuring aid)
  xf_sharded_config cfg{};
  xf_table *tw = nullptr, *tv = nullptr;
  xf_workspace *ws = nullptr;  // world 1: the fused step's scratch
  int parity_mode = XF_PARITY_EXACT_SUMS;  // (what xf_sharded_set_parity last set)
  uint64_t seen_upper = 0;     // world 1: host-side upper bound on the keys in the table
  Dev<double> partial;         // LR forward scratch
  // the owner-compute compile's staging: the worker's nonzeros grouped by owner (sent from
  // here), their rows, the partition's counts; pinned host memory for what the host reads back
  Dev<uint64_t> stage_keys, stage_pairs;
  Dev<uint32_t> stage_rows, stage_rowof, stage_cnt;
  uint64_t *h_counts = nullptr;
  hipStream_t main = nullptr, side = nullptr;
  hipEvent_t ev_pulled = nullptr, ev_graded = nullptr, ev_applied = nullptr;
  bool have_applied = false, have_graded = false, have_pulled = false;
  // stale1: the step whose Push is outstanding
  xf_sbatch *pending = nullptr;
  int pending_flip = 0;
  // per-stage timing (sequential schedule only)
  bool profiling = false;
  // a ring of event sets, every kProfileEvery-th step records: the host never waits for the
  // step it has just launched (see xf_workspace)
  static constexpr int kEvSets = 8;
  static constexpr long kProfileEvery = 4;
  struct EvSet {
    hipEvent_t ev[kEvN + 1] = {};
    bool pending = false;
  } sets[kEvSets];
  int cur = 0;
  long step_no = 0;
  bool rec = false;
  double ms_sum[kEvN] = {};
  long steps_timed = 0;
};

namespace {

// exchanges on the side stream (stale1: Push(t) under Pull(t+1)) go over the group's second
// communicator: one communicator driven from two streams would serialise them, and the ranks'
// enqueue orders on it could differ
int a2a(xf_sharded *st, const void *send, const std::vector<uint64_t> &sc, void *recv,
        const std::vector<uint64_t> &rc, size_t elem_bytes, hipStream_t s) {
  return xf_group_alltoallv_ch(st->g, s == st->side ? 1 : 0, send, sc.data(), recv, rc.data(),
                               elem_bytes, 0, (void *)s);
}

// Wait for a stream that may hold a collective.  A peer that has failed (or an exchange the
// ranks entered in different orders) would leave hipStreamSynchronize waiting forever: with
// more than one rank the stream is polled instead, and after XF_COLLECTIVE_TIMEOUT_S seconds
// (default 300) the call returns XF_EIO on this rank — every rank of a stuck exchange gets there.
int wait_stream(xf_sharded *st, hipStream_t s) {
  if (st->world <= 1) {
    XF_HIP(hipStreamSynchronize(s));
    return XF_OK;
  }
  static const double limit = [] {
    const char *v = getenv("XF_COLLECTIVE_TIMEOUT_S");
    const double x = v ? atof(v) : 0.0;
    return x > 0 ? x : 300.0;
  }();
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return XF_OK;
    if (e != hipErrorNotReady) XF_HIP(e);
    if (spins > 4096) {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      const double dt =
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (dt > limit) {
        // The stream still holds the stuck collective and whatever was queued behind it: the
        // callers' device scratch must not be freed under it (hipFree would wait for it — the
        // hang again — or release memory that queued kernels write).  The trainer is poisoned:
        // the communicators are aborted so that the stuck kernels end, scratch is leaked from
        // here on (xf::scratch_poison) and every later call fails fast.
        st->poisoned = true;
        xf::scratch_poison();
        xf_group_abort(st->g);
        return xf::set_error(XF_EIO, "rank %d: the work on a stream with a collective did not "
                             "finish within %.0f s (a peer has failed, or the ranks entered an "
                             "exchange in different orders); this trainer is unusable now",
                             st->rank, limit);
      }
    }
  }
}

int collect_set(xf_sharded *st, xf_sharded::EvSet &e) {
  if (!e.pending) return XF_OK;
  XF_HIP(hipEventSynchronize(e.ev[kEvN]));
  for (int i = 0; i < kEvN; ++i) {
    float ms = 0.f;
    XF_HIP(hipEventElapsedTime(&ms, e.ev[i], e.ev[i + 1]));
    st->ms_sum[i] += ms;
  }
  ++st->steps_timed;
  e.pending = false;
  return XF_OK;
}
int collect_profile(xf_sharded *st) {  // every set still pending
  for (auto &e : st->sets) XF_TRY(collect_set(st, e));
  return XF_OK;
}
// a step begins: does it record, and into which set
inline bool owner_dataflow(int schedule) {
  return schedule == XF_SCHEDULE_OWNER || schedule == XF_SCHEDULE_OWNER_STALE1;
}
inline bool overlapped(int schedule) {  // two streams: no per-stage events
  return schedule == XF_SCHEDULE_STALE1 || schedule == XF_SCHEDULE_OWNER_STALE1;
}

int begin_profiled_step(xf_sharded *st) {
  st->rec = st->profiling && !overlapped(st->cfg.schedule) &&
            st->step_no++ % xf_sharded::kProfileEvery == 0;
  if (!st->rec) return XF_OK;
  st->cur = (st->cur + 1) % xf_sharded::kEvSets;
  return collect_set(st, st->sets[st->cur]);
}

#define XF_MARK(i)                                                                \
  do {                                                                            \
    if (st->rec) XF_HIP(hipEventRecord(st->sets[st->cur].ev[i], st->main));       \
  } while (0)

// owner side of the Pull: state rows of the keys this rank owns (resolved once per row
// numbering, inserting on first touch — ftrl.h:56) and the weight payload
int front_pull(xf_sharded *st, xf_sbatch *b, StepBuf &B, hipStream_t s) {
  const size_t n = b->n_recv;
  const bool fm = st->cfg.model == 1;
  const uint64_t uid = xf::table_uid(st->tw), ep = xf::table_epoch(st->tw);
  if (b->rows_uid != uid || b->rows_epoch != ep) {
    XF_TRY(b->rows_w.reserve(n));
    if (fm) XF_TRY(b->rows_v.reserve(n));
    if (n) {
      // every key may be new to this shard: make room first (the table grows by itself)
      XF_TRY(xf::table_ensure_room(st->tw, n));
      if (fm) XF_TRY(xf::table_ensure_room(st->tv, n));
      // all sources' lists in one pass over the shard, visited in key order (the same key may
      // come from several workers: the table's resolve handles that)
      XF_TRY(xf_table_pull_ordered_dev(st->tw, b->rkeys_sorted.p, b->rorder.p, n, b->rows_w.p,
                                       nullptr, s));
      if (fm)
        XF_TRY(xf_table_pull_ordered_dev(st->tv, b->rkeys_sorted.p, b->rorder.p, n, b->rows_v.p,
                                         nullptr, s));
      // the static part of the Push's merged walk
      XF_TRY(b->hrow_w.reserve(n));
      XF_TRY(xf::table_head_rows(b->rkeys_sorted.p, b->rorder.p, b->rows_w.p, n, b->hrow_w.p, s));
      if (fm) {
        XF_TRY(b->hrow_v.reserve(n));
        XF_TRY(xf::table_head_rows(b->rkeys_sorted.p, b->rorder.p, b->rows_v.p, n, b->hrow_v.p, s));
      }
    }
    b->rows_uid = uid;
    b->rows_epoch = ep;
  }
  XF_TRY(B.w_send.reserve(n));
  if (n) XF_TRY(xf_table_gather_dev(st->tw, b->rows_w.p, n, B.w_send.p, s));
  if (fm) {
    XF_TRY(B.v_send.reserve(n * st->cfg.k));
    if (n) XF_TRY(xf_table_gather_dev(st->tv, b->rows_v.p, n, B.v_send.p, s));
  }
  return XF_OK;
}

// weights back to the workers, forward, gradient
int front_compute(xf_sharded *st, xf_sbatch *b, StepBuf &B, float *d_pctr, bool want_grad,
                  hipStream_t s) {
  const bool fm = st->cfg.model == 1;
  const int k = st->cfg.k;
  XF_TRY(B.wu.reserve(b->U));
  XF_TRY(B.loss.reserve(b->R));
  XF_TRY(a2a(st, B.w_send.p, b->recv_counts, B.wu.p, b->send_counts, 4, s));
  if (fm) {
    XF_TRY(B.vu.reserve((size_t)b->U * k));
    XF_TRY(B.vsum.reserve(b->R));
    XF_TRY(a2a(st, B.v_send.p, b->recv_counts, B.vu.p, b->send_counts, 4 * (size_t)k, s));
  }
  XF_MARK(kEvA2aW + 1);
  if (!fm) {
    XF_TRY(st->partial.reserve(xf::cells_partial_doubles(b->cells)));
    XF_TRY(xf::cells_lr_forward(b->cells, B.wu.p, b->b->view.labels, st->partial.p, B.loss.p,
                                d_pctr, s));
    XF_MARK(kEvForward + 1);
    if (want_grad) {
      XF_TRY(B.g.reserve(b->U));
      XF_TRY(xf::cells_lr_grad(b->cells, B.loss.p, B.g.p, s));
    }
  } else {
    if (xf::fm_records_fit(k) && b->U) {
      XF_TRY(B.ks.reserve(xf::fm_record_bytes(b->U)));
      XF_TRY(xf::fm_forward_records(&b->b->view, k, B.wu.p, B.vu.p, B.ks.p, B.loss.p, d_pctr,
                                    B.vsum.p, s));
    } else {
      XF_TRY(xf_fm_forward_dev(&b->b->view, k, B.wu.p, B.vu.p, B.loss.p, d_pctr, B.vsum.p, s));
    }
    XF_MARK(kEvForward + 1);
    if (want_grad) {
      XF_TRY(B.g.reserve(b->U));
      XF_TRY(B.gv.reserve((size_t)b->U * k));
      XF_TRY(xf_fm_grad_dev(&b->b->view, k, B.vu.p, B.vsum.p, B.loss.p, B.g.p, B.gv.p, s));
    }
  }
  XF_MARK(kEvGrad + 1);
  return XF_OK;
}

// Push: gradients to the owners (the keys are already there) ...
int back_exchange(xf_sharded *st, xf_sbatch *b, StepBuf &B, hipStream_t s) {
  XF_TRY(B.g_recv.reserve(b->n_recv));
  XF_TRY(a2a(st, B.g.p, b->send_counts, B.g_recv.p, b->recv_counts, 4, s));
  if (st->cfg.model == 1) {
    XF_TRY(B.gv_recv.reserve(b->n_recv * st->cfg.k));
    XF_TRY(a2a(st, B.gv.p, b->send_counts, B.gv_recv.p, b->recv_counts, 4 * (size_t)st->cfg.k, s));
  }
  return XF_OK;
}

// ... and the owner-side optimizer steps: ONE pass over the shard, a key's pushes applied one
// after the other in source-rank order
int back_apply(xf_sharded *st, xf_sbatch *b, StepBuf &B, hipStream_t s) {
  if (!b->n_recv) return XF_OK;
  XF_TRY(xf::table_update_heads(st->tw, b->hrow_w.p, b->rorder.p, b->n_recv, B.g_recv.p, s));
  if (st->cfg.model == 1)
    XF_TRY(xf::table_update_heads(st->tv, b->hrow_v.p, b->rorder.p, b->n_recv, B.gv_recv.p, s));
  return XF_OK;
}

int owner_flush_pending(xf_sharded *st);

int flush_pending(xf_sharded *st) {
  if (st->cfg.schedule == XF_SCHEDULE_OWNER_STALE1) return owner_flush_pending(st);
  if (st->pending) {
    xf_sbatch *pb = st->pending;
    StepBuf &PB = pb->buf[st->pending_flip];
    if (st->have_graded) XF_HIP(hipStreamWaitEvent(st->side, st->ev_graded, 0));
    XF_TRY(back_exchange(st, pb, PB, st->side));
    if (st->have_pulled) XF_HIP(hipStreamWaitEvent(st->side, st->ev_pulled, 0));
    XF_TRY(back_apply(st, pb, PB, st->side));
    XF_HIP(hipEventRecord(st->ev_applied, st->side));
    st->have_applied = true;
    st->pending = nullptr;
  }
  if (st->have_applied) XF_HIP(hipStreamWaitEvent(st->main, st->ev_applied, 0));
  return XF_OK;
}

}  // namespace


// ------------------------------------------------------------------ XF_SCHEDULE_OWNER
// The worker's side of the compile: its nonzeros grouped by the OWNER of their key (xf_shard_of:
// min(key / (UINT64_MAX / world), world - 1), ps-lite's default slicer), row-major order kept
// within an owner.  Round 4 did it with a rocPRIM radix pass over (owner, position) pairs and a
// gather of the keys through the permutation; the owner number has at most 8 bits and the list
// is already in the order wanted inside every owner, so this is a partition: a histogram per
// workgroup (k_own_hist), the workgroups' write positions from one scan (k_own_scan), and a
// stable scatter (k_own_scatter) — 8 bytes of key + 4 of row read twice and written once per
// nonzero, no sort, no permutation array, and the per-owner counts stay on the device until the
// ONE host wait of the compile (they travel to the peers through the group's all-to-all).
__global__ void __launch_bounds__(kBlock)
k_rows_of_nnz(const uint32_t *__restrict__ rowptr, uint32_t R, uint32_t *__restrict__ row_of) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t nw = gridDim.x * (kBlock / 64);
  for (uint32_t r = blockIdx.x * (kBlock / 64) + threadIdx.x / 64; r < R; r += nw)
    for (uint32_t j = rowptr[r] + lane; j < rowptr[r + 1]; j += 64) row_of[j] = r;
}

constexpr int kOwn = 1024;           // threads of the partition workgroups
constexpr uint32_t kOwnMax = 256;    // owners (compile_owner requires world <= 255)

__device__ __forceinline__ uint32_t owner_of(uint64_t key, uint64_t span, uint32_t world) {
  const uint64_t q = key / span;
  return q < world ? (uint32_t)q : world - 1;
}

// wgcnt[w * world + o] = nonzeros of workgroup w's share [w * span_nnz, (w + 1) * span_nnz) whose
// key belongs to owner o.  The lanes of a wavefront that hold the same owner count as one LDS
// atomic (eight owners: 64 lanes on 8 addresses would serialise).
__global__ void __launch_bounds__(kOwn)
k_own_hist(const uint64_t *__restrict__ keys, uint32_t n, uint32_t world, uint32_t span_nnz,
           uint32_t *__restrict__ wgcnt) {
  __shared__ uint32_t h[kOwnMax];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  if (tid < kOwnMax) h[tid] = 0;
  __syncthreads();
  const uint64_t span = UINT64_MAX / world;
  const uint32_t e0 = blockIdx.x * span_nnz, e1 = min(e0 + span_nnz, n);
  for (uint32_t j0 = e0; j0 < e1; j0 += kOwn) {
    const uint32_t j = j0 + tid;
    const bool ok = j < e1;
    const uint32_t o = ok ? owner_of(keys[j], span, world) : 0xFFFFFFFFu;
    unsigned long long todo = __ballot(ok);
    while (todo) {  // wave-uniform: one round per owner the wavefront holds
      const int l = __ffsll((long long)todo) - 1;
      const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)o, l);
      const unsigned long long m = __ballot(o == cur);
      if ((int)lane == l) atomicAdd(&h[cur], (uint32_t)__popcll(m));
      todo &= ~m;
    }
  }
  __syncthreads();
  if (tid < world) wgcnt[(size_t)blockIdx.x * world + tid] = h[tid];
}

// One workgroup.  wgcnt[w * world + o] -> where workgroup w's nonzeros of owner o begin in the
// owner-grouped list; first[o] = where owner o's begin (first[world] = n); pairs[o] = {nonzeros
// for owner o, this worker's rows}: what owner o is told (one 16-byte element per peer).
// A wavefront per owner: the column of the (at most 256) workgroups' counts in four loads per
// lane and a shuffle scan (one lane walking the column was 2 x 256 dependent trips: 41 us).
__global__ void __launch_bounds__(kOwnMax)
k_own_scan(uint32_t *__restrict__ wgcnt, uint32_t nwg, uint32_t world, uint32_t R,
           uint32_t *__restrict__ first, uint64_t *__restrict__ pairs) {
  __shared__ uint32_t tot[kOwnMax + 1];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  constexpr uint32_t NWV = kOwnMax / 64;
  constexpr int kCol = 4;  // 256 workgroups at most (compile_owner_dev's split)
  for (uint32_t o = wave; o < world; o += NWV) {
    uint32_t carry = 0;
#pragma unroll
    for (int q = 0; q < kCol; ++q) {
      const uint32_t w = q * 64 + lane;
      const uint32_t x = w < nwg ? wgcnt[(size_t)w * world + o] : 0u;
      uint32_t inc = x;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(inc, d);
        if ((int)lane >= d) inc += y;
      }
      if (w < nwg) wgcnt[(size_t)w * world + o] = carry + inc - x;  // (relative to the owner's first)
      carry += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) tot[o] = carry;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t a = 0;
    for (uint32_t q = 0; q < world; ++q) {
      const uint32_t x = tot[q];
      tot[q] = a;
      a += x;
      pairs[2 * q] = x;
      pairs[2 * q + 1] = R;
    }
    tot[world] = a;
  }
  __syncthreads();
  for (uint32_t o = wave; o < world; o += NWV) {
    const uint32_t base = tot[o];
#pragma unroll
    for (int q = 0; q < kCol; ++q) {
      const uint32_t w = q * 64 + lane;
      if (w < nwg) wgcnt[(size_t)w * world + o] += base;
    }
  }
  if (tid <= world) first[tid] = tot[tid];
}

// The stable scatter: workgroup w walks its share in rounds of kOwn consecutive nonzeros.  A
// nonzero's place = the workgroup's running position for its owner + the nonzeros of that owner
// in the round's earlier wavefronts + those in the lower lanes of its own (ballots).
__global__ void __launch_bounds__(kOwn)
k_own_scatter(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ row_of, uint32_t n,
              uint32_t world, uint32_t span_nnz, const uint32_t *__restrict__ wgbase,
              uint64_t *__restrict__ out_keys, uint32_t *__restrict__ out_rows) {
  constexpr int NW = kOwn / 64;
  __shared__ uint32_t run[kOwnMax];
  __shared__ uint32_t wcnt[NW][kOwnMax];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid < world) run[tid] = wgbase[(size_t)blockIdx.x * world + tid];
  const uint64_t span = UINT64_MAX / world;
  const uint32_t e0 = blockIdx.x * span_nnz, e1 = min(e0 + span_nnz, n);
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint64_t nkey = 0;
  uint32_t nrow = 0;
  if (e0 + tid < e1) {
    nkey = keys[e0 + tid];
    nrow = row_of[e0 + tid];
  }
  for (uint32_t j0 = e0; j0 < e1; j0 += kOwn) {
    const uint32_t j = j0 + tid;
    const bool ok = j < e1;
    const uint64_t key = nkey;
    const uint32_t row = nrow;
    if (j + kOwn < e1) {  // the next round's, on their way while this one is placed
      nkey = keys[j + kOwn];
      nrow = row_of[j + kOwn];
    }
    const uint32_t o = ok ? owner_of(key, span, world) : 0xFFFFFFFFu;
    for (uint32_t q = lane; q < world; q += 64) wcnt[wave][q] = 0;  // (this wavefront's row)
    unsigned long long todo = __ballot(ok);
    uint32_t rank = 0;
    while (todo) {  // wave-uniform
      const int l = __ffsll((long long)todo) - 1;
      const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)o, l);
      const unsigned long long m = __ballot(o == cur);
      if (o == cur) rank = (uint32_t)__popcll(m & lt);
      if ((int)lane == l) wcnt[wave][cur] = (uint32_t)__popcll(m);
      todo &= ~m;
    }
    __syncthreads();
    if (ok) {
      uint32_t pos = run[o] + rank;
      for (uint32_t w = 0; w < wave; ++w) pos += wcnt[w][o];
      out_keys[pos] = key;
      out_rows[pos] = row;
    }
    __syncthreads();
    if (tid < world) {
      uint32_t a = 0;
#pragma unroll
      for (int w = 0; w < NW; ++w) a += wcnt[w][tid];
      run[tid] += a;
    }
    __syncthreads();
  }
}

// received row numbers: worker q's nonzeros sit at [segoff[q], segoff[q+1]); + win[q] * W
__global__ void __launch_bounds__(kBlock)
k_rows_to_padded(uint32_t *__restrict__ rowid, size_t n, uint32_t nsrc,
                 const uint32_t *__restrict__ segoff, const uint32_t *__restrict__ win,
                 uint32_t W) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
       j += (size_t)gridDim.x * blockDim.x) {
    uint32_t q = 0;
    while (q + 1 < nsrc && j >= segoff[q + 1]) ++q;
    rowid[j] += win[q] * W;
  }
}

// the worker's side of the forward: add the owners' partial row sums (recv[o * R + r], fp64:
// exact), then lr_worker.cc:141
__global__ void __launch_bounds__(kBlock)
k_owner_finalize(const double *__restrict__ recv, uint32_t nown, uint32_t R,
                 const int32_t *__restrict__ labels, float *__restrict__ loss,
                 float *__restrict__ pctr) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  double a = 0.0;
  for (uint32_t o = 0; o < nown; ++o) a += recv[(size_t)o * R + r];
  const float p = xf::sigmoid_ref((float)a);
  if (pctr) pctr[r] = p;
  if (loss) loss[r] = p - (float)labels[r];
}

__global__ void __launch_bounds__(kBlock)
k_replicate_f32(const float *__restrict__ in, uint32_t n, uint32_t copies,
                float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)n * copies) out[i] = in[i % n];
}

// rowptr of rows whose nonzeros arrive in ascending row order (empty rows allowed)
__global__ void __launch_bounds__(kBlock)
k_rowptr_of_sorted_rows(const uint32_t *__restrict__ rowid, uint32_t n, uint32_t R,
                        uint32_t *__restrict__ rowptr) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j <= n;
       j += (size_t)gridDim.x * blockDim.x) {
    const uint32_t lo = j == 0 ? 0u : rowid[j - 1] + 1;  // rows (prev, cur] begin at j
    const uint32_t hi = j == n ? R : rowid[j];
    for (uint32_t r = lo; r <= hi; ++r) rowptr[r] = (uint32_t)j;
  }
}

// FM: the worker's side of the forward.  recv[(o * R + r) * 3 ..] = owner o's share of the
// row's (wx, v_sum, v_pow_sum), fp64: their sums are the row's exact sums; then
// fm_worker.cc:194-199.  lv[r] = (loss, v_sum) for the owners' gradient.
__global__ void __launch_bounds__(kBlock)
k_owner_finalize_fm(const double *__restrict__ recv, uint32_t nown, uint32_t R,
                    const int32_t *__restrict__ labels, float *__restrict__ loss,
                    float *__restrict__ pctr, float2 *__restrict__ lv) {
#pragma clang fp contract(off)
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  double wx = 0.0, vs = 0.0, vp = 0.0;
  for (uint32_t o = 0; o < nown; ++o) {
    const double *q = recv + ((size_t)o * R + r) * 3;
    wx += q[0];
    vs += q[1];
    vp += q[2];
  }
  const float vsf = (float)vs, vpf = (float)vp;
  const float vy = vsf * vsf - vpf;
  const float p = xf::sigmoid_ref((float)wx + vy);
  const float l = p - (float)labels[r];
  if (pctr) pctr[r] = p;
  if (loss) loss[r] = l;
  if (lv) lv[r] = make_float2(l, vsf);
}

__global__ void __launch_bounds__(kBlock)
k_split_pairs(const float2 *__restrict__ in, uint32_t n, float *__restrict__ a,
              float *__restrict__ b) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float2 q = in[i];
  a[i] = q.x;
  b[i] = q.y;
}

// Compile for the owner-compute dataflow: the nonzeros of this worker's rows go to the owners
// of their keys (once), with the row each belongs to.  Device-resident input (the raw keys in CSR
// order, 32-bit row offsets, labels); ONE host wait — for the per-owner counts, which the ranks
// tell each other through the group's all-to-all (16 bytes per peer: nonzeros for that owner,
// rows of this worker) — and one copy of the small per-minibatch tables.  The input arrays may
// be released when this returns.
static int compile_owner_dev(xf_sharded *st, xf_sbatch *b, const uint64_t *d_keys,
                             const uint32_t *d_rowptr, const int32_t *d_labels, uint32_t R,
                             size_t NNZ) {
  const int W = st->world;
  XF_REQUIRE(W <= 255, "xf_sharded_compile: owner-compute dataflow with %d workers", W);
  XF_REQUIRE(NNZ < 0xFFFFFFFFull, "xf_sharded_compile: %zu nonzeros in one minibatch", NNZ);
  hipStream_t s = st->main;
  b->oc = true;
  b->R = R;
  b->NNZ = (uint32_t)NNZ;
  b->U = 0;
  // the nonzeros grouped by the owner of their key, row-major order kept within an owner: a
  // stable partition into the trainer's staging buffers (stream order protects them: the
  // exchange of the previous compile has read them before this one's scatter writes)
  const uint32_t span_nnz = (uint32_t)(((NNZ + 255) / 256 + kOwn - 1) / kOwn * kOwn);
  const uint32_t nwg = NNZ ? (uint32_t)((NNZ + span_nnz - 1) / span_nnz) : 0u;
  XF_TRY(st->stage_keys.reserve(NNZ));
  XF_TRY(st->stage_rows.reserve(NNZ));
  XF_TRY(st->stage_rowof.reserve(NNZ));
  XF_TRY(st->stage_cnt.reserve((size_t)nwg * W + (size_t)W + 1));
  XF_TRY(st->stage_pairs.reserve((size_t)4 * W));
  if (!st->h_counts)
    XF_HIP(hipHostMalloc((void **)&st->h_counts, ((size_t)W + 1) * 4 + (size_t)2 * W * 8 + 64));
  uint32_t *h_first = (uint32_t *)((char *)st->h_counts + (size_t)2 * W * 8);
  uint64_t *h_pairs = st->h_counts;
  uint32_t *d_wgcnt = st->stage_cnt.p, *d_first = st->stage_cnt.p + (size_t)nwg * W;
  uint64_t *d_psend = st->stage_pairs.p, *d_precv = st->stage_pairs.p + (size_t)2 * W;
  if (NNZ) {
    hipLaunchKernelGGL(k_rows_of_nnz, dim3(grid_for((size_t)R * 64)), dim3(kBlock), 0, s, d_rowptr,
                       R, st->stage_rowof.p);
    hipLaunchKernelGGL(k_own_hist, dim3(nwg), dim3(kOwn), 0, s, d_keys, (uint32_t)NNZ, (uint32_t)W,
                       span_nnz, d_wgcnt);
  }
  hipLaunchKernelGGL(k_own_scan, dim3(1), dim3(kOwnMax), 0, s, d_wgcnt, nwg, (uint32_t)W, R,
                     d_first, d_psend);
  if (NNZ)
    hipLaunchKernelGGL(k_own_scatter, dim3(nwg), dim3(kOwn), 0, s, d_keys,
                       (const uint32_t *)st->stage_rowof.p, (uint32_t)NNZ, (uint32_t)W, span_nnz,
                       (const uint32_t *)d_wgcnt, st->stage_keys.p, st->stage_rows.p);
  XF_HIP(hipGetLastError());
  XF_TRY(b->d_labels.reserve(R));
  if (R)
    XF_HIP(hipMemcpyAsync(b->d_labels.p, d_labels, (size_t)R * 4, hipMemcpyDeviceToDevice, s));
  // who sends how much to whom, and how many rows every worker has
  const std::vector<uint64_t> ones(W, 1);
  XF_TRY(a2a(st, d_psend, ones, d_precv, ones, 16, s));
  XF_HIP(hipMemcpyAsync(h_first, d_first, ((size_t)W + 1) * 4, hipMemcpyDeviceToHost, s));
  XF_HIP(hipMemcpyAsync(h_pairs, d_precv, (size_t)2 * W * 8, hipMemcpyDeviceToHost, s));
  XF_TRY(wait_stream(st, s));  // the compile's one wait (the input arrays are not read after it)
  std::vector<uint64_t> cnt(W), recv(W), rows_all(W);
  std::vector<uint32_t> segoff(W + 1, 0);
  uint64_t o_n = 0;
  for (int p = 0; p < W; ++p) {
    cnt[p] = h_first[p + 1] - h_first[p];
    recv[p] = h_pairs[2 * p];
    rows_all[p] = h_pairs[2 * p + 1];
    o_n += recv[p];
    XF_REQUIRE(o_n < 0xFFFFFFFFull, "xf_sharded_compile: %llu nonzeros for one owner",
               (unsigned long long)o_n);
    segoff[p + 1] = (uint32_t)o_n;
  }
  b->o_n = (size_t)o_n;
  b->n_recv = b->o_n;
  XF_TRY(b->o_keys.reserve(b->o_n));
  XF_TRY(b->o_rowid.reserve(b->o_n));
  XF_TRY(a2a(st, st->stage_keys.p, cnt, b->o_keys.p, recv, 8, s));
  XF_TRY(a2a(st, st->stage_rows.p, cnt, b->o_rowid.p, recv, 4, s));
  // windows: every worker's rows fill whole windows of one common size
  uint64_t maxR = 1;
  for (int p = 0; p < W; ++p) maxR = std::max<uint64_t>(maxR, rows_all[p]);
  // windows per worker: as few as the LDS allows, or up to two more when that fills the chip
  // better (the forward runs windows x G workgroups, G = 8 * (32 / windows): 8 workers x 3
  // windows would leave a quarter of the CUs idle, 8 x 4 uses all of them)
  uint64_t nw = (maxR + xf::kWinMax - 1) / xf::kWinMax;
  {
    auto groups = [&](uint64_t per_worker) {
      const uint64_t total = per_worker * (uint64_t)W;
      return total <= 32 ? total * ((32 / total) * 8) : (uint64_t)256;
    };
    uint64_t best = nw;
    for (uint64_t c = nw + 1; (c <= nw + 2 || c * (uint64_t)W <= 32) && c <= maxR; ++c)
      if (groups(c) > groups(best)) best = c;
    nw = best;
  }
  // (Measured at the N = 8 owner shape staged on one rank, 4 x 10^5 rows — tools/r5/call27.sh:
  // 23 / 26 / 32 / 40 / 48 windows give a forward of 66.9 / 62.1 / 57.4 / 83.0 / 106.5 us, the
  // gradient + Pushes 138-141 us whatever the number; the rule picks 32.)
  b->oW = (uint32_t)((maxR + nw - 1) / nw);
  b->o_rows.assign(W, 0);
  b->o_rows64.assign(W, 0);
  b->o_win.assign(W + 1, 0);
  std::vector<uint32_t> rowoff(W + 1, 0);
  for (int p = 0; p < W; ++p) {
    b->o_rows[p] = (uint32_t)rows_all[p];
    b->o_rows64[p] = rows_all[p];
    b->o_win[p + 1] = b->o_win[p] + (uint32_t)((rows_all[p] + b->oW - 1) / b->oW);
    rowoff[p + 1] = rowoff[p] + (uint32_t)rows_all[p];
  }
  XF_REQUIRE((uint64_t)b->o_win[W] * b->oW < 0x7FFFFFFFull, "xf_sharded_compile: too many rows");
  b->o_rpad = std::max<uint32_t>(1, b->o_win[W]) * b->oW;
  b->o_rows_all = rowoff[W];
  b->o_total = rowoff[W];
  // the small per-minibatch tables: one host array (kept with the minibatch), one copy
  std::vector<uint32_t> &H = b->o_small_host;
  H.clear();
  auto put = [&](const std::vector<uint32_t> &v) {
    const size_t at = H.size();
    H.insert(H.end(), v.begin(), v.end());
    return at;
  };
  const size_t at_win = put(b->o_win), at_rows = put(b->o_rows);
  // all workers as ONE source (XF_UPDATE_SUM_THEN_STEP)
  const size_t at_win1 = put({0u, b->o_win[W]}), at_rows1 = put({rowoff[W]});
  size_t at_winT = 0, at_rowsT = 0;
  b->nT = 0;
  if (const char *e = getenv("XF_OWNER_TIMING_SOURCES")) {
    // MEASURING AID, one rank only: the gradient + Push pass as an owner of n workers runs it —
    // this rank's windows dealt out to n pretended workers, each its own optimizer step.  The
    // arithmetic is not that of one LRWorker::update any more: for timing the pass on one GPU.
    const uint32_t n = (uint32_t)atoi(e), nwt = b->o_win[W];
    if (W == 1 && n > 1 && nwt >= n) {
      static bool said = false;
      if (!said) {
        said = true;
        fprintf(stderr, "xflow_amd: XF_OWNER_TIMING_SOURCES=%u — a measuring aid: this rank's rows "
                "are dealt out to %u pretended workers and every key takes %u optimizer steps; "
                "the results are NOT those of LRWorker::update\n", n, n, n);
      }
      std::vector<uint32_t> winT(n + 1), rowsT(n);
      for (uint32_t q = 0; q <= n; ++q) winT[q] = (uint32_t)((uint64_t)nwt * q / n);
      for (uint32_t q = 0; q < n; ++q)
        rowsT[q] = std::min<uint32_t>(rowoff[W], winT[q + 1] * b->oW) -
                   std::min<uint32_t>(rowoff[W], winT[q] * b->oW);
      at_winT = put(winT);
      at_rowsT = put(rowsT);
      b->nT = n;
    }
  }
  size_t at_wbase, at_wrows;
  {
    std::vector<uint32_t> wbase(std::max<uint32_t>(1, b->o_win[W]), 0), wrows(wbase.size(), 0);
    for (int p = 0; p < W; ++p)
      for (uint32_t v = b->o_win[p]; v < b->o_win[p + 1]; ++v) {
        const uint32_t first = (v - b->o_win[p]) * b->oW;
        wbase[v] = rowoff[p] + first;
        wrows[v] = std::min<uint32_t>(b->oW, b->o_rows[p] - first);
      }
    at_wbase = put(wbase);
    at_wrows = put(wrows);
  }
  const size_t at_seg = put(segoff), at_off = put(rowoff);
  XF_TRY(b->o_small.reserve(H.size()));
  XF_HIP(hipMemcpyAsync(b->o_small.p, H.data(), H.size() * 4, hipMemcpyHostToDevice, s));
  uint32_t *D = b->o_small.p;
  b->d_win.p = D + at_win;
  b->d_rows.p = D + at_rows;
  b->d_win1.p = D + at_win1;
  b->d_rows1.p = D + at_rows1;
  b->d_winT.p = b->nT ? D + at_winT : nullptr;
  b->d_rowsT.p = b->nT ? D + at_rowsT : nullptr;
  b->d_wbase.p = D + at_wbase;
  b->d_wrows.p = D + at_wrows;
  const uint32_t *d_seg = D + at_seg, *d_off = D + at_off;
  b->q_rowoff = rowoff;
  if (st->cfg.model == 1 && st->cfg.update_rule == XF_UPDATE_RANK_ORDERED) {
    // FM, every worker's Push its own optimizer step: the nonzeros of worker q (they arrive
    // worker after worker, row-major within a worker) become minibatch q, rows numbered as the
    // worker numbers them
    b->bq.assign(W, nullptr);
    b->wsq.assign(W, nullptr);
    size_t Usum = 0;
    for (int q = 0; q < W; ++q) {
      const uint32_t nq = segoff[q + 1] - segoff[q], Rq = b->o_rows[q];
      if (!nq) continue;
      Dev<uint32_t> d_rp;
      Dev<int32_t> d_lab;
      XF_TRY(d_rp.reserve((size_t)Rq + 1));
      XF_TRY(d_lab.reserve(Rq));
      hipLaunchKernelGGL(k_rowptr_of_sorted_rows, dim3(grid_for((size_t)nq + 1)), dim3(kBlock), 0,
                         s, b->o_rowid.p + segoff[q], nq, Rq, d_rp.p);
      XF_HIP(hipGetLastError());
      XF_HIP(hipMemsetAsync(d_lab.p, 0, (size_t)Rq * 4, s));  // (labels stay with the workers)
      XF_TRY(xf::batch_compile_dev_ex(&b->bq[q], b->o_keys.p + segoff[q], d_rp.p, d_lab.p, Rq, nq, s, false));
      XF_TRY(wait_stream(st, s));
      uint32_t Uq = 0;
      XF_TRY(xf_batch_dims(b->bq[q], nullptr, nullptr, &Uq, nullptr));
      Usum += Uq;
      XF_TRY(xf_workspace_create(&b->wsq[q]));
    }
    b->U = (uint32_t)std::min<size_t>(Usum, 0xFFFFFFFFu);
    // first-touch keys are inserted by the step's Pulls: room for all of them now
    XF_TRY(xf::table_ensure_room(st->tw, Usum));
    XF_TRY(xf::table_ensure_room(st->tv, Usum));
  } else if (st->cfg.model == 1) {
    // FM: the received nonzeros become one minibatch with a key list (the owner's two tables
    // resolve it), rows numbered worker after worker — they arrive in that order
    if (b->o_n) {
      Dev<uint32_t> d_rp;
      Dev<int32_t> d_lab;
      XF_TRY(d_rp.reserve((size_t)b->o_total + 1));
      XF_TRY(d_lab.reserve(b->o_total));
      const uint32_t one = 1u;
      hipLaunchKernelGGL(k_rows_to_padded, dim3(grid_for(b->o_n)), dim3(kBlock), 0, s,
                         b->o_rowid.p, b->o_n, (uint32_t)W, d_seg, d_off, one);
      hipLaunchKernelGGL(k_rowptr_of_sorted_rows, dim3(grid_for(b->o_n + 1)), dim3(kBlock), 0, s,
                         b->o_rowid.p, (uint32_t)b->o_n, b->o_total, d_rp.p);
      XF_HIP(hipGetLastError());
      XF_HIP(hipMemsetAsync(d_lab.p, 0, (size_t)b->o_total * 4, s));  // (labels stay with the
                                                                      // rows' workers)
      XF_TRY(xf::batch_compile_dev_ex(&b->b, b->o_keys.p, d_rp.p, d_lab.p, b->o_total,
                                      (uint32_t)b->o_n, s, false));
      XF_TRY(wait_stream(st, s));
      XF_TRY(xf_batch_dims(b->b, nullptr, nullptr, &b->U, nullptr));
      // first-touch keys are inserted by the step's Pull: room for all of them now
      XF_TRY(xf::table_ensure_room(st->tw, b->U));
      XF_TRY(xf::table_ensure_room(st->tv, b->U));
    }
  } else if (b->o_n) {
    hipLaunchKernelGGL(k_rows_to_padded, dim3(grid_for(b->o_n)), dim3(kBlock), 0, s, b->o_rowid.p,
                       b->o_n, (uint32_t)W, d_seg, (const uint32_t *)b->d_win.p, b->oW);
    XF_HIP(hipGetLastError());
  }
  return XF_OK;
}

// host-array front end (the reader's block arrays and a row slice): upload, then the above
static int compile_owner(xf_sharded *st, xf_sbatch *b, const uint64_t *rowptr,
                         const uint64_t *keys, const int32_t *labels, size_t row_begin,
                         size_t row_end) {
  hipStream_t s = st->main;
  const size_t R = row_end - row_begin;
  const uint64_t base = rowptr[row_begin];
  const size_t NNZ = (size_t)(rowptr[row_end] - base);
  XF_REQUIRE(R < 0xFFFFFFFFull && NNZ < 0xFFFFFFFFull, "xf_sharded_compile: minibatch too large");
  std::vector<uint32_t> rp(R + 1);
  for (size_t r = 0; r <= R; ++r) rp[r] = (uint32_t)(rowptr[row_begin + r] - base);
  xf::Scratch sc;
  uint64_t *d_k = nullptr;
  uint32_t *d_rp = nullptr;
  int32_t *d_lab = nullptr;
  XF_TRY(sc.get(&d_k, NNZ));
  XF_TRY(sc.get(&d_rp, R + 1));
  XF_TRY(sc.get(&d_lab, R));
  if (NNZ) XF_HIP(hipMemcpyAsync(d_k, keys + base, NNZ * 8, hipMemcpyHostToDevice, s));
  XF_HIP(hipMemcpyAsync(d_rp, rp.data(), (R + 1) * 4, hipMemcpyHostToDevice, s));
  if (R) XF_HIP(hipMemcpyAsync(d_lab, labels + row_begin, R * 4, hipMemcpyHostToDevice, s));
  // (its wait is also the wait for these uploads: `rp` and the scratch may go when it returns)
  return compile_owner_dev(st, b, d_k, d_rp, d_lab, (uint32_t)R, NNZ);
}

// the cells of the received nonzeros against the shard's current row numbering (keys resolved —
// inserted on first touch, ftrl.h:56 — here: this is the Pull's key -> row step)
static int ensure_ocells(xf_sharded *st, xf_sbatch *b) {
  const uint64_t uid = xf::table_uid(st->tw), ep = xf::table_epoch(st->tw);
  if (b->ocells && b->oc_uid == uid && b->oc_epoch == ep) return XF_OK;
  hipStream_t s = st->main;
  if (b->ocells) {
    XF_TRY(wait_stream(st, s));
    xf::cells_free(b->ocells);
    b->ocells = nullptr;
  }
  XF_TRY(xf::cells_build_keyed(&b->ocells, st->tw, b->o_keys.p, nullptr,
                               b->o_n ? b->o_rowid.p : nullptr, b->o_rpad, (uint32_t)b->o_n,
                               b->oc_keep, b->oW, s));
  b->ocells->table_uid = uid;
  b->ocells->epoch = b->oc_epoch = xf::table_epoch(st->tw);  // (after the build, which may move it)
  b->oc_uid = uid;
  const size_t sp = (size_t)std::max<uint32_t>((uint32_t)st->world, b->nT) *
                    xf::cells_split_chunks(b->ocells) * xf::kChunk;
  XF_TRY(b->gsum.reserve(sp));
  XF_TRY(b->gtouched.reserve(sp));
  return XF_OK;
}

// forward at the owners, row sums to the rows' workers, sigmoid there
static int owner_forward(xf_sharded *st, xf_sbatch *b, float *d_loss, float *d_pctr,
                         hipStream_t s, hipEvent_t table_read = nullptr) {
  const int W = st->world;
  XF_TRY(ensure_ocells(st, b));
  const xf_cells *c = b->ocells;
  XF_TRY(st->partial.reserve(xf::cells_partial_doubles(c)));
  uint32_t total = 0;
  for (uint32_t r : b->o_rows) total += r;
  XF_TRY(b->rs_send.reserve(total));
  XF_TRY(b->rs_recv.reserve((size_t)W * b->R));
  // the row sums land worker after worker, ready to be sent
  XF_TRY(xf::cells_lr_forward_sums(c, xf::table_weights(st->tw), st->partial.p, b->d_wbase.p,
                                   b->d_wrows.p, b->rs_send.p, s));
  if (table_read) XF_HIP(hipEventRecord(table_read, s));  // (the weights are not read after this)
  XF_MARK(1);
  const std::vector<uint64_t> mine(W, b->R);
  XF_TRY(a2a(st, b->rs_send.p, b->o_rows64, b->rs_recv.p, mine, 8, s));
  if (b->R)
    hipLaunchKernelGGL(k_owner_finalize, dim3(grid_for(b->R)), dim3(kBlock), 0, s, b->rs_recv.p,
                       (uint32_t)W, b->R, b->d_labels.p, d_loss, d_pctr);
  XF_HIP(hipGetLastError());
  XF_MARK(2);
  return XF_OK;
}

static int step_owner_fm(xf_sharded *st, xf_sbatch *b);

// gradient + the workers' Pushes in rank order, one pass over the shard (the losses are read
// where they arrived: worker after worker)
static int owner_grad(xf_sharded *st, xf_sbatch *b, const float *d_loss_recv, hipStream_t s) {
  if (b->nT)  // (XF_OWNER_TIMING_SOURCES)
    return xf::cells_lr_grad_update_sources(b->ocells, st->tw, d_loss_recv, b->nT, b->d_winT.p,
                                            b->d_rowsT.p, b->d_wbase.p, b->gsum.p,
                                            b->gtouched.p, s);
  if (st->cfg.update_rule == XF_UPDATE_SUM_THEN_STEP)  // one source: every row of the step
    return xf::cells_lr_grad_update_sources(b->ocells, st->tw, d_loss_recv, 1u, b->d_win1.p,
                                            b->d_rows1.p, b->d_wbase.p, b->gsum.p,
                                            b->gtouched.p, s, b->o_rows_all);
  return xf::cells_lr_grad_update_sources(b->ocells, st->tw, d_loss_recv, (uint32_t)st->world,
                                          b->d_win.p, b->d_rows.p, b->d_wbase.p, b->gsum.p,
                                          b->gtouched.p, s, st->world == 1 ? b->o_rows_all : 0u);
}

// owner_stale1: the outstanding gradient + Pushes (step t-1) on the side stream, once the
// losses of that step have arrived (ev_graded) and the forward of step t has read the weights
// (ev_pulled)
static int owner_apply_pending(xf_sharded *st) {
  xf_sbatch *pb = st->pending;
  if (!pb) return XF_OK;
  if (st->have_graded) XF_HIP(hipStreamWaitEvent(st->side, st->ev_graded, 0));
  if (st->have_pulled) XF_HIP(hipStreamWaitEvent(st->side, st->ev_pulled, 0));
  if (st->cfg.model == 1) {
    // FM (XF_UPDATE_RANK_ORDERED): the workers' two Pushes, worker after worker, of the gradients
    // the step left in its workspaces (fm_worker.cc:241-242)
    std::vector<xf_workspace *> &W = st->pending_flip ? pb->wsq2 : pb->wsq;
    for (size_t q = 0; q < pb->bq.size(); ++q)
      if (pb->bq[q]) XF_TRY(xf::fm_owner_push_pulled(st->tw, st->tv, pb->bq[q], W[q], st->side));
  } else {
    XF_TRY(owner_grad(st, pb, (st->pending_flip ? pb->loss_recv2 : pb->loss_recv).p, st->side));
  }
  XF_HIP(hipEventRecord(st->ev_applied, st->side));
  st->have_applied = true;
  st->pending = nullptr;
  return XF_OK;
}

namespace {
int owner_flush_pending(xf_sharded *st) {
  XF_TRY(owner_apply_pending(st));
  if (st->have_applied) XF_HIP(hipStreamWaitEvent(st->main, st->ev_applied, 0));
  return XF_OK;
}
}  // namespace

// one LRWorker::update of every rank, owner-compute dataflow
static int step_owner(xf_sharded *st, xf_sbatch *b) {
  const int W = st->world;
  hipStream_t s = st->main;
  XF_REQUIRE(b->oc, "xf_sharded_step: the minibatch was not compiled for the owner-compute "
             "dataflow");
  if (st->cfg.model == 1) return step_owner_fm(st, b);
  const bool stale = st->cfg.schedule == XF_SCHEDULE_OWNER_STALE1;
  XF_TRY(begin_profiled_step(st));
  XF_MARK(0);
  XF_TRY(b->oloss.reserve(b->R));
  uint32_t total = 0;
  for (uint32_t r : b->o_rows) total += r;
  const int flip = stale ? b->oflip : 0;
  if (stale) b->oflip ^= 1;
  Dev<float> &lrecv = flip ? b->loss_recv2 : b->loss_recv;
  XF_TRY(b->loss_rep.reserve((size_t)W * b->R));
  XF_TRY(lrecv.reserve(total));
  if (stale) {
    // forward(t) sees the table with the Pushes of step t-2 (ev_applied) and without those of
    // step t-1, which start on the side stream as soon as forward(t) has read the weights and
    // run under this step's two exchanges
    if (st->have_applied) XF_HIP(hipStreamWaitEvent(s, st->ev_applied, 0));
    XF_TRY(ensure_ocells(st, b));  // (may rebuild cells: before the event that frees the table)
    XF_TRY(owner_forward(st, b, b->oloss.p, nullptr, s, st->ev_pulled));
    st->have_pulled = true;
    XF_TRY(owner_apply_pending(st));
  } else {
    XF_TRY(owner_forward(st, b, b->oloss.p, nullptr, s));
  }
  // the losses to every owner
  if (b->R)
    hipLaunchKernelGGL(k_replicate_f32, dim3(grid_for((size_t)W * b->R)), dim3(kBlock), 0, s,
                       b->oloss.p, b->R, (uint32_t)W, b->loss_rep.p);
  const std::vector<uint64_t> mine(W, b->R);
  XF_TRY(a2a(st, b->loss_rep.p, mine, lrecv.p, b->o_rows64, 4, s));
  XF_HIP(hipGetLastError());
  XF_MARK(3);
  if (stale) {
    XF_HIP(hipEventRecord(st->ev_graded, s));  // the losses of step t are here
    st->have_graded = true;
    st->pending = b;
    st->pending_flip = flip;
    return XF_OK;
  }
  XF_TRY(owner_grad(st, b, lrecv.p, s));
  XF_MARK(4);
  XF_MARK(5);
  XF_MARK(6);
  if (st->rec) st->sets[st->cur].pending = true;
  return XF_OK;
}

// ---- FM on the owner-compute dataflow (sum_then_step): the owners' shares of the three row
// sums to the rows' workers, (loss, v_sum) back, one gradient + optimizer step per key over all
// the rows of the step
static int owner_forward_fm(xf_sharded *st, xf_sbatch *b, float *d_loss, float *d_pctr,
                            float2 *d_lv, hipStream_t s,
                            std::vector<xf_workspace *> *wsq_use = nullptr,
                            hipEvent_t table_read = nullptr) {
  const int W = st->world;
  std::vector<xf_workspace *> &WS = wsq_use ? *wsq_use : b->wsq;
  XF_TRY(b->rs_send.reserve((size_t)b->o_total * 3));
  XF_TRY(b->rs_recv.reserve((size_t)W * b->R * 3));
  if (!b->bq.empty()) {  // XF_UPDATE_RANK_ORDERED: every worker's Pull and its share of its rows
    if (b->o_total) XF_HIP(hipMemsetAsync(b->rs_send.p, 0, (size_t)b->o_total * 24, s));
    for (size_t q = 0; q < b->bq.size(); ++q)
      if (b->bq[q])
        XF_TRY(xf::fm_owner_partials(st->tw, st->tv, b->bq[q], WS[q],
                                     b->rs_send.p + (size_t)3 * b->q_rowoff[q], s, true));
  } else if (b->b)
    XF_TRY(xf::fm_owner_partials(st->tw, st->tv, b->b, st->ws, b->rs_send.p, s));
  else if (b->o_total)
    XF_HIP(hipMemsetAsync(b->rs_send.p, 0, (size_t)b->o_total * 24, s));
  if (table_read) XF_HIP(hipEventRecord(table_read, s));  // (the tables are not read after this)
  XF_MARK(1);
  const std::vector<uint64_t> mine(W, b->R);
  XF_TRY(a2a(st, b->rs_send.p, b->o_rows64, b->rs_recv.p, mine, 24, s));
  if (b->R)
    hipLaunchKernelGGL(k_owner_finalize_fm, dim3(grid_for(b->R)), dim3(kBlock), 0, s,
                       b->rs_recv.p, (uint32_t)W, b->R, b->d_labels.p, d_loss, d_pctr, d_lv);
  XF_HIP(hipGetLastError());
  XF_MARK(2);
  return XF_OK;
}

static int step_owner_fm(xf_sharded *st, xf_sbatch *b) {
  const int W = st->world;
  hipStream_t s = st->main;
  XF_TRY(begin_profiled_step(st));
  XF_MARK(0);
  XF_TRY(b->oloss.reserve(b->R));
  XF_TRY(b->lv.reserve((size_t)b->R * 2));
  // owner_stale1 (XF_UPDATE_RANK_ORDERED): the two Pushes of every worker of step t-1 run on the
  // side stream under this step's exchanges — after this step's Pulls have read the tables
  // (ev_pulled) and step t-1's gradients exist (ev_graded); this step's Pulls wait for the
  // Pushes of step t-2 (ev_applied).  Rows exactly one step stale, reader and writer of the
  // tables never concurrent: the stale1 schedule of the weight / gradient exchange.
  const bool stale = st->cfg.schedule == XF_SCHEDULE_OWNER_STALE1;
  const int flip = stale ? b->oflip : 0;
  std::vector<xf_workspace *> *WS = &b->wsq;
  if (stale) {
    XF_REQUIRE(!b->bq.empty() || (!b->b && b->o_total == 0),
               "xf_sharded_step: owner_stale1 for FM takes minibatches compiled under "
               "update_rule rank_ordered");
    b->oflip ^= 1;
    if (flip) {
      if (b->wsq2.size() != b->wsq.size()) b->wsq2.assign(b->wsq.size(), nullptr);
      for (size_t q = 0; q < b->wsq.size(); ++q)
        if (b->wsq[q] && !b->wsq2[q]) XF_TRY(xf_workspace_create(&b->wsq2[q]));
      WS = &b->wsq2;
    }
    if (st->have_applied) XF_HIP(hipStreamWaitEvent(s, st->ev_applied, 0));
    XF_TRY(owner_forward_fm(st, b, b->oloss.p, nullptr, (float2 *)b->lv.p, s, WS, st->ev_pulled));
    st->have_pulled = true;
    XF_TRY(owner_apply_pending(st));
  } else {
    XF_TRY(owner_forward_fm(st, b, b->oloss.p, nullptr, (float2 *)b->lv.p, s));
  }
  XF_TRY(b->lv_rep.reserve((size_t)W * b->R * 2));
  XF_TRY(b->lv_recv.reserve((size_t)b->o_total * 2));
  XF_TRY(b->loss_recv.reserve(b->o_total));
  XF_TRY(b->vsum_recv.reserve(b->o_total));
  if (b->R)
    hipLaunchKernelGGL(k_replicate_f32, dim3(grid_for((size_t)W * b->R * 2)), dim3(kBlock), 0, s,
                       b->lv.p, b->R * 2, (uint32_t)W, b->lv_rep.p);
  const std::vector<uint64_t> mine(W, b->R);
  XF_TRY(a2a(st, b->lv_rep.p, mine, b->lv_recv.p, b->o_rows64, 8, s));
  if (b->o_total)
    hipLaunchKernelGGL(k_split_pairs, dim3(grid_for(b->o_total)), dim3(kBlock), 0, s,
                       (const float2 *)b->lv_recv.p, b->o_total, b->loss_recv.p,
                       b->vsum_recv.p);
  XF_HIP(hipGetLastError());
  XF_MARK(3);
  if (!b->bq.empty()) {
    // every worker's gradient from what its Pull returned (all Pulls precede all Pushes), then
    // the Pushes worker after worker: the state a key's second step starts from is the first's
    for (size_t q = 0; q < b->bq.size(); ++q)
      if (b->bq[q])
        XF_TRY(xf::fm_owner_grad_pulled(st->tv, b->bq[q], (*WS)[q],
                                        b->loss_recv.p + b->q_rowoff[q],
                                        b->vsum_recv.p + b->q_rowoff[q], s));
    if (stale) {  // the Pushes wait for the next step's Pulls (or a flush)
      XF_HIP(hipEventRecord(st->ev_graded, s));
      st->have_graded = true;
      st->pending = b;
      st->pending_flip = flip;
      XF_MARK(4);
      XF_MARK(5);
      XF_MARK(6);
      if (st->rec) st->sets[st->cur].pending = true;
      return XF_OK;
    }
    for (size_t q = 0; q < b->bq.size(); ++q)
      if (b->bq[q]) XF_TRY(xf::fm_owner_push_pulled(st->tw, st->tv, b->bq[q], b->wsq[q], s));
  } else if (b->b)
    XF_TRY(xf::fm_owner_grad_update(st->tw, st->tv, b->b, st->ws, b->loss_recv.p,
                                    b->vsum_recv.p, s));
  XF_MARK(4);
  XF_MARK(5);
  XF_MARK(6);
  if (st->rec) st->sets[st->cur].pending = true;
  return XF_OK;
}

// a trainer whose stream wait timed out (wait_stream) holds stuck work: nothing runs on it again
#define XF_ALIVE(st)                                                                         \
  do {                                                                                       \
    if ((st)->poisoned)                                                                      \
      return xf::set_error(XF_EIO, "this trainer is unusable: an earlier wait for a stream " \
                           "with a collective timed out (destroy it and the group)");        \
  } while (0)

extern "C" void xf_sharded_config_default(xf_sharded_config *c) {
  memset(c, 0, sizeof(*c));
  c->model = 0;
  c->optimizer = XF_OPT_FTRL;
  c->k = 10;                 // fm_worker.h:92
  c->capacity = 1u << 22;
  c->alpha = 5e-2f;          // ftrl.h:17-20
  c->beta = 1.0f;
  c->lambda1 = 5e-5f;
  c->lambda2 = 10.0f;
  c->lr = 0.001f;            // sgd.h:16
  c->seed = 0;
  c->schedule = XF_SCHEDULE_SEQUENTIAL;
}

extern "C" int xf_sharded_create(xf_sharded **out, xf_group *g, const xf_sharded_config *cfg) {
  XF_REQUIRE(out && cfg, "xf_sharded_create: null argument");
  XF_REQUIRE(cfg->model == 0 || cfg->model == 1, "xf_sharded_create: model %d", cfg->model);
  XF_REQUIRE(cfg->schedule == XF_SCHEDULE_SEQUENTIAL || cfg->schedule == XF_SCHEDULE_STALE1 ||
                 owner_dataflow(cfg->schedule),
             "xf_sharded_create: schedule %d", cfg->schedule);
  XF_REQUIRE(cfg->update_rule == XF_UPDATE_RANK_ORDERED ||
                 (cfg->update_rule == XF_UPDATE_SUM_THEN_STEP && owner_dataflow(cfg->schedule)),
             "xf_sharded_create: update_rule %d (sum_then_step needs the owner-compute dataflow: "
             "there the workers' sums meet exactly)", cfg->update_rule);
  // (FM's fused gradient + Push reads the factors where they live, at Push time: only the rule
  // that forms every worker's gradient from what its Pull returned can run one step late)
  XF_REQUIRE(cfg->schedule != XF_SCHEDULE_OWNER_STALE1 || cfg->model == 0 ||
                 cfg->update_rule == XF_UPDATE_RANK_ORDERED,
             "xf_sharded_create: owner_stale1 for FM needs update_rule rank_ordered (sum_then_step: "
             "schedule owner)");
  xf_sharded *st = new xf_sharded;
  st->g = g;
  st->cfg = *cfg;
  if (g) XF_TRY(xf_group_info(g, &st->rank, &st->world, nullptr));
  {
    const char *e = getenv("XF_SHARDED_GENERAL");
    st->fused = st->world == 1 && !(g && e && *e == '1');
  }
  struct Guard {
    xf_sharded *s;
    ~Guard() {
      if (s) xf_sharded_destroy(s);
    }
  } guard{st};
  xf_table_config c;
  xf_table_config_default(&c);
  c.opt_kind = cfg->optimizer;
  c.alpha = cfg->alpha;
  c.beta = cfg->beta;
  c.lambda1 = cfg->lambda1;
  c.lambda2 = cfg->lambda2;
  c.lr = cfg->lr;
  c.capacity = cfg->capacity;
  c.shard = (uint32_t)st->rank;
  c.nshards = (uint32_t)st->world;
  c.dim = 1;
  c.init_kind = XF_INIT_ZERO;
  XF_TRY(xf_table_create(&st->tw, &c));
  if (cfg->model == 1) {  // server.h:22-31: app 1 serves v
    c.dim = cfg->k;
    if (cfg->optimizer == XF_OPT_FTRL) {
      c.init_kind = XF_INIT_HASHNORM;  // ftrl.h:114-120 (deterministic stand-in)
      c.seed = cfg->seed;
    } else {
      c.init_kind = XF_INIT_CONST;  // sgd.h:67-72
      c.init_const = 0.001f;
    }
    XF_TRY(xf_table_create(&st->tv, &c));
  }
  XF_TRY(xf_workspace_create(&st->ws));
  XF_HIP(hipStreamCreateWithFlags(&st->main, hipStreamNonBlocking));
  XF_HIP(hipStreamCreateWithFlags(&st->side, hipStreamNonBlocking));
  XF_HIP(hipEventCreateWithFlags(&st->ev_pulled, hipEventDisableTiming));
  XF_HIP(hipEventCreateWithFlags(&st->ev_graded, hipEventDisableTiming));
  XF_HIP(hipEventCreateWithFlags(&st->ev_applied, hipEventDisableTiming));
  guard.s = nullptr;
  *out = st;
  return XF_OK;
}

extern "C" int xf_sharded_destroy(xf_sharded *st) {
  if (!st) return XF_OK;
  if (st->poisoned) return XF_OK;  // stuck work on its streams: everything it owns is leaked
  (void)hipDeviceSynchronize();
  if (st->h_counts) (void)hipHostFree(st->h_counts);
  if (st->ws) xf_workspace_destroy(st->ws);
  if (st->tw) xf_table_destroy(st->tw);
  if (st->tv) xf_table_destroy(st->tv);
  if (st->main) (void)hipStreamDestroy(st->main);
  if (st->side) (void)hipStreamDestroy(st->side);
  for (hipEvent_t e : {st->ev_pulled, st->ev_graded, st->ev_applied})
    if (e) (void)hipEventDestroy(e);
  for (auto &e : st->sets)
    for (auto &ev : e.ev)
      if (ev) (void)hipEventDestroy(ev);
  delete st;
  return XF_OK;
}

extern "C" int xf_sharded_tables(xf_sharded *st, xf_table **w, xf_table **v) {
  XF_REQUIRE(st, "xf_sharded_tables: null trainer");
  if (w) *w = st->tw;
  if (v) *v = st->tv;
  return XF_OK;
}

extern "C" int xf_sbatch_free(xf_sbatch *b) {
  if (!b) return XF_OK;
  // its Push may still be outstanding (stale1): apply it first.  That is an exchange — every
  // rank frees its minibatches at the same point of the program, as it steps them.
  if (b->owner && b->owner->pending == b) (void)flush_pending(b->owner);
  (void)hipDeviceSynchronize();
  if (b->cells) xf::cells_free(b->cells);
  if (b->ocells) xf::cells_free(b->ocells);
  if (b->b) xf_batch_free(b->b);
  for (xf_batch *q : b->bq)
    if (q) xf_batch_free(q);
  for (xf_workspace *q : b->wsq)
    if (q) xf_workspace_destroy(q);
  for (xf_workspace *q : b->wsq2)
    if (q) xf_workspace_destroy(q);
  delete b;
  return XF_OK;
}

extern "C" int xf_sbatch_fm_keyed(const xf_sbatch *b) {
  if (!b) return -1;
  return b->b && b->b->fm_keyed ? 1 : 0;
}

extern "C" int xf_sbatch_dims(const xf_sbatch *b, uint32_t *R, uint32_t *NNZ, uint32_t *U,
                              uint64_t *n_owned) {
  XF_REQUIRE(b, "xf_sbatch_dims: null batch");
  if (R) *R = b->R;
  if (NNZ) *NNZ = b->NNZ;
  if (U) *U = b->U;
  if (n_owned) *n_owned = b->n_recv;
  return XF_OK;
}

// the static part of the weight / gradient exchange of a minibatch with a key list (b->b):
// contiguous owner ranges of the sorted unique keys (ps-lite's slicer), the keys to their
// owners — once — and the owner's merged walking order
static int compile_exchange_tail(xf_sharded *st, xf_sbatch *b, int keep) {
  hipStream_t s = st->main;
  XF_TRY(xf_batch_dims(b->b, &b->R, &b->NNZ, &b->U, nullptr));
  const xf_dev_batch &v = b->b->view;
  if (st->cfg.model == 0 && !b->cells)  // the LR kernels stream cells over the batch's unique-key index
    XF_TRY(xf::cells_build(&b->cells, v.uidx, nullptr, v.rowptr, b->R, b->NNZ, b->U,
                           xf::kCellsUidx, keep != 0, s));
  const int W = st->world;
  std::vector<uint32_t> split(W + 1);
  {
    xf::Scratch sc;
    uint32_t *d_split = nullptr;
    XF_TRY(sc.get(&d_split, (size_t)W + 1));
    hipLaunchKernelGGL(k_owner_split, dim3((W + 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                       v.ukeys, b->U, (uint32_t)W, d_split);
    XF_HIP(hipMemcpyAsync(split.data(), d_split, ((size_t)W + 1) * 4, hipMemcpyDeviceToHost, s));
    XF_TRY(wait_stream(st, s));
  }
  b->send_counts.resize(W);
  for (int p = 0; p < W; ++p) b->send_counts[p] = split[p + 1] - split[p];
  std::vector<uint64_t> all((size_t)W * W);
  XF_TRY(xf_group_allgather_host(st->g, b->send_counts.data(), (size_t)W * 8, all.data()));
  b->recv_counts.resize(W);
  b->n_recv = 0;
  for (int p = 0; p < W; ++p) {
    b->recv_counts[p] = all[(size_t)p * W + st->rank];
    b->n_recv += (size_t)b->recv_counts[p];
  }
  XF_REQUIRE(b->n_recv < 0xFFFFFFFFull, "xf_sharded_compile: %zu keys for one owner", b->n_recv);
  // the keys travel once, here
  XF_TRY(b->rkeys.reserve(b->n_recv));
  XF_TRY(a2a(st, v.ukeys, b->send_counts, b->rkeys.p, b->recv_counts, 8, s));
  // ... and the owner's walking order: all sources' lists merged by key (stable: a key's
  // entries stay in source-rank order).  Walking the lists one source after the other would
  // sweep the shard's index and state once per source.
  XF_TRY(b->rkeys_sorted.reserve(b->n_recv));
  XF_TRY(b->rorder.reserve(b->n_recv));
  if (b->n_recv) {
    // (round 6: this shard's key range cut into uniform ranges, a range sorted in LDS —
    // xf::sort_key_pos; the library's radix sort beyond its limits)
    const xf::TableDev &T = xf::table_dev(st->tw);
    XF_TRY(xf::sort_key_pos_any(b->rkeys.p, (uint32_t)b->n_recv, T.lo, T.span, b->rkeys_sorted.p,
                                b->rorder.p, s, nullptr, xf::kSortSiteMerged));
    XF_TRY(wait_stream(st, s));
  }
  return XF_OK;
}

// batches with a key list insert at step time (the Pull): make room now.  A host-side upper
// bound on the key count keeps the exact (synchronising) check rare.
static int fused_room(xf_sharded *st, xf_sbatch *b) {
  XF_TRY(xf_batch_dims(b->b, &b->R, &b->NNZ, &b->U, nullptr));
  if (b->U) {
    uint64_t cap = 0;
    XF_TRY(xf_table_capacity(st->tw, &cap));
    if ((st->seen_upper + b->U) * 10 > cap * 6) {
      XF_TRY(xf::table_ensure_room(st->tw, b->U));
      if (st->tv) XF_TRY(xf::table_ensure_room(st->tv, b->U));
      uint64_t n = 0;
      XF_TRY(xf_table_size(st->tw, &n));
      st->seen_upper = n;
    }
    st->seen_upper += b->U;
  }
  return XF_OK;
}

// host-array front end of xf::batch_compile_lr_dev (as xf_batch_compile_gpu's: the raw slice
// uploaded, the build on the GPU)
static int compile_lr_from_host(xf_sbatch *b, const uint64_t *rowptr, const uint64_t *keys,
                                const int32_t *labels, size_t row_begin, size_t row_end, int keep,
                                hipStream_t s, bool *lean) {
  *lean = false;
  const size_t R = row_end - row_begin;
  const uint64_t base = rowptr[row_begin];
  const size_t NNZ = (size_t)(rowptr[row_end] - base);
  if (!R || !NNZ || !keys || R >= 0xFFFFFFFFull || NNZ >= 0xFFFFFFFFull) return XF_OK;
  std::vector<uint32_t> rp(R + 1);
  for (size_t r = 0; r <= R; ++r) rp[r] = (uint32_t)(rowptr[row_begin + r] - base);
  xf::Scratch sc;
  uint64_t *d_keys = nullptr;
  uint32_t *d_rp = nullptr;
  int32_t *d_lab = nullptr;
  XF_TRY(sc.get(&d_keys, NNZ));
  XF_TRY(sc.get(&d_rp, R + 1));
  XF_TRY(sc.get(&d_lab, R));
  XF_HIP(hipMemcpyAsync(d_keys, keys + base, NNZ * 8, hipMemcpyHostToDevice, s));
  XF_HIP(hipMemcpyAsync(d_rp, rp.data(), (R + 1) * 4, hipMemcpyHostToDevice, s));
  XF_HIP(hipMemcpyAsync(d_lab, labels + row_begin, R * 4, hipMemcpyHostToDevice, s));
  XF_HIP(hipStreamSynchronize(s));
  return xf::batch_compile_lr_dev(&b->b, &b->cells, d_keys, d_rp, d_lab, (uint32_t)R, (uint32_t)NNZ,
                                  keep != 0, s, lean);
}

// Compile a minibatch: the key build (lr_worker.cc:146-166) and the static part of the
// exchange.  COLLECTIVE: every rank of the group calls it, in the same order.  Host arrays
// (the reader's block arrays and a row slice).  keep != 0: the batch will be replayed.
extern "C" int xf_sharded_compile(xf_sharded *st, xf_sbatch **out, const uint64_t *rowptr,
                                  const uint64_t *keys, const int32_t *labels, size_t row_begin,
                                  size_t row_end, int keep) {
  XF_REQUIRE(st && out && rowptr && labels && row_end >= row_begin,
             "xf_sharded_compile: bad argument");
  XF_ALIVE(st);
  xf_sbatch *b = new xf_sbatch;
  struct Guard {
    xf_sbatch *b;
    ~Guard() {
      if (b) xf_sbatch_free(b);
    }
  } guard{b};
  hipStream_t s = st->main;
  b->owner = st;
  if (st->fused) {  // one shard: the table is local
    if (st->cfg.host_key_build)
      XF_TRY(xf_batch_compile(&b->b, rowptr, keys, labels, row_begin, row_end));
    else if (st->cfg.model == 0)
      XF_TRY(xf_batch_compile_local(&b->b, st->tw, rowptr, keys, labels, row_begin, row_end,
                                    keep, s));
    else if (st->tv && st->parity_mode == XF_PARITY_EXACT_SUMS)
      // FM: against the tables' settled tiers when every key sits there (no sort; otherwise
      // this is xf_batch_compile_gpu)
      XF_TRY(xf_batch_compile_fm(&b->b, st->tw, st->tv, rowptr, keys, labels, row_begin, row_end,
                                 s, nullptr));
    else
      XF_TRY(xf_batch_compile_gpu(&b->b, rowptr, keys, labels, row_begin, row_end, s));
    XF_TRY(fused_room(st, b));
    guard.b = nullptr;
    *out = b;
    return XF_OK;
  }
  if (owner_dataflow(st->cfg.schedule)) {
    b->oc_keep = keep != 0;
    XF_TRY(compile_owner(st, b, rowptr, keys, labels, row_begin, row_end));
    guard.b = nullptr;
    *out = b;
    return XF_OK;
  }
  if (st->cfg.host_key_build) {
    XF_TRY(xf_batch_compile(&b->b, rowptr, keys, labels, row_begin, row_end));
    XF_TRY(xf_batch_upload(b->b, s));
  } else {
    bool lean = false;
    if (st->cfg.model == 0) XF_TRY(compile_lr_from_host(b, rowptr, keys, labels, row_begin, row_end, keep, s, &lean));
    if (!lean) XF_TRY(xf_batch_compile_gpu(&b->b, rowptr, keys, labels, row_begin, row_end, s));
  }
  XF_TRY(compile_exchange_tail(st, b, keep));
  guard.b = nullptr;
  *out = b;
  return XF_OK;
}

// The same from device-resident arrays (raw keys in CSR order, 32-bit row offsets, labels): what
// a reader that parses straight into HBM, or a caller that keeps its blocks there, hands over.
// The arrays may be released when this returns.  COLLECTIVE.
extern "C" int xf_sharded_compile_dev(xf_sharded *st, xf_sbatch **out, const uint64_t *d_keys,
                                      const uint32_t *d_rowptr, const int32_t *d_labels,
                                      uint32_t R, uint32_t NNZ, int keep) {
  XF_REQUIRE(st && out && d_rowptr && (R == 0 || d_labels) && (NNZ == 0 || d_keys),
             "xf_sharded_compile_dev: null argument");
  XF_ALIVE(st);
  XF_REQUIRE(!st->cfg.host_key_build, "xf_sharded_compile_dev: the trainer builds its keys on "
             "the host (host_key_build): hand it host arrays");
  xf_sbatch *b = new xf_sbatch;
  struct Guard {
    xf_sbatch *b;
    ~Guard() {
      if (b) xf_sbatch_free(b);
    }
  } guard{b};
  hipStream_t s = st->main;
  b->owner = st;
  if (st->fused) {
    if (st->cfg.model == 0)
      XF_TRY(xf_batch_compile_local_dev(&b->b, st->tw, d_keys, d_rowptr, d_labels, R, NNZ, keep,
                                        s));
    else if (st->tv && st->parity_mode == XF_PARITY_EXACT_SUMS)
      XF_TRY(xf_batch_compile_fm_dev(&b->b, st->tw, st->tv, d_keys, d_rowptr, d_labels, R, NNZ, s,
                                     nullptr));
    else
      XF_TRY(xf_batch_compile_dev(&b->b, d_keys, d_rowptr, d_labels, R, NNZ, s));
    XF_TRY(fused_room(st, b));
  } else if (owner_dataflow(st->cfg.schedule)) {
    b->oc_keep = keep != 0;
    XF_TRY(compile_owner_dev(st, b, d_keys, d_rowptr, d_labels, R, NNZ));
  } else {
    // LR: the keys and the cells, nothing else (xf::batch_compile_lr_dev); FM, or beyond that
    // build's limits: the minibatch with all its views
    bool lean = false;
    if (st->cfg.model == 0)
      XF_TRY(xf::batch_compile_lr_dev(&b->b, &b->cells, d_keys, d_rowptr, d_labels, R, NNZ, keep != 0,
                                      s, &lean));
    if (!lean) XF_TRY(xf::batch_compile_dev_ex(&b->b, d_keys, d_rowptr, d_labels, R, NNZ, s, false));
    XF_TRY(compile_exchange_tail(st, b, keep));
  }
  guard.b = nullptr;
  *out = b;
  return XF_OK;
}

// One LRWorker::update / FMWorker::update of every rank (COLLECTIVE), asynchronous.
extern "C" int xf_sharded_step(xf_sharded *st, xf_sbatch *b) {
  XF_REQUIRE(st && b, "xf_sharded_step: null argument");
  XF_ALIVE(st);
  if (st->fused) {
    if (st->cfg.model == 0) return xf_lr_step(st->tw, b->b, st->ws, st->main);
    return xf_fm_step(st->tw, st->tv, b->b, st->ws, st->main);
  }
  if (owner_dataflow(st->cfg.schedule)) return step_owner(st, b);
  XF_REQUIRE(!b->oc, "xf_sharded_step: the minibatch was compiled for the owner-compute dataflow");
  XF_TRY(begin_profiled_step(st));
  const int flip = b->flip;
  b->flip ^= 1;
  StepBuf &B = b->buf[flip];
  if (st->cfg.schedule == XF_SCHEDULE_SEQUENTIAL) {
    XF_MARK(0);
    XF_TRY(front_pull(st, b, B, st->main));
    XF_MARK(kEvPull + 1);
    XF_TRY(front_compute(st, b, B, nullptr, true, st->main));
    XF_TRY(back_exchange(st, b, B, st->main));
    XF_MARK(kEvA2aG + 1);
    XF_TRY(back_apply(st, b, B, st->main));
    XF_MARK(kEvUpdate + 1);
    if (st->rec) st->sets[st->cur].pending = true;
    return XF_OK;
  }
  // ---- stale1
  xf_sbatch *pb = st->pending;
  const int pflip = st->pending_flip;
  if (st->have_applied)  // the Push of step t-2 is in the table
    XF_HIP(hipStreamWaitEvent(st->main, st->ev_applied, 0));
  XF_TRY(front_pull(st, b, B, st->main));  // reads the table before Push(t-1) lands
  XF_HIP(hipEventRecord(st->ev_pulled, st->main));
  st->have_pulled = true;
  if (pb) {
    StepBuf &PB = pb->buf[pflip];
    XF_HIP(hipStreamWaitEvent(st->side, st->ev_graded, 0));  // gradient of step t-1 is complete
    XF_TRY(back_exchange(st, pb, PB, st->side));
    XF_HIP(hipStreamWaitEvent(st->side, st->ev_pulled, 0));  // do not write while Pull(t) reads
    XF_TRY(back_apply(st, pb, PB, st->side));
    XF_HIP(hipEventRecord(st->ev_applied, st->side));
    st->have_applied = true;
  }
  XF_TRY(front_compute(st, b, B, nullptr, true, st->main));
  XF_HIP(hipEventRecord(st->ev_graded, st->main));
  st->have_graded = true;
  st->pending = b;
  st->pending_flip = flip;
  return XF_OK;
}

// apply the outstanding Push of the stale1 schedule (end of training, before export / predict)
extern "C" int xf_sharded_flush(xf_sharded *st) {
  XF_REQUIRE(st, "xf_sharded_flush: null trainer");
  XF_ALIVE(st);
  if (!st->fused) XF_TRY(flush_pending(st));
  XF_TRY(wait_stream(st, st->main));
  XF_TRY(wait_stream(st, st->side));
  return XF_OK;
}

// forward only over this rank's rows (calculate_pctr, lr_worker.cc:25-71); COLLECTIVE: ranks
// without rows of their own call it with an empty minibatch and serve the Pulls.  The pulls
// insert unseen keys, as in the reference (ftrl.h:56).
extern "C" int xf_sharded_predict(xf_sharded *st, xf_sbatch *b, float *pctr_out) {
  XF_REQUIRE(st && b && (b->R == 0 || pctr_out), "xf_sharded_predict: null argument");
  XF_ALIVE(st);
  if (st->fused) {
    XF_TRY(wait_stream(st, st->main));
    if (st->cfg.model == 0) return xf_lr_predict(st->tw, b->b, st->ws, pctr_out);
    return xf_fm_predict(st->tw, st->tv, b->b, st->ws, pctr_out);
  }
  XF_TRY(xf_sharded_flush(st));
  st->rec = false;  // a forward-only pass records no step profile
  if (owner_dataflow(st->cfg.schedule)) {
    XF_REQUIRE(b->oc, "xf_sharded_predict: the minibatch was not compiled for the owner-compute "
               "dataflow");
    XF_TRY(b->opctr.reserve(b->R));
    if (st->cfg.model == 1)
      XF_TRY(owner_forward_fm(st, b, nullptr, b->opctr.p, nullptr, st->main));
    else
      XF_TRY(owner_forward(st, b, nullptr, b->opctr.p, st->main));
    XF_TRY(wait_stream(st, st->main));
    if (b->R) XF_HIP(hipMemcpy(pctr_out, b->opctr.p, (size_t)b->R * 4, hipMemcpyDeviceToHost));
    return XF_OK;
  }
  XF_REQUIRE(!b->oc, "xf_sharded_predict: the minibatch was compiled for the owner-compute "
             "dataflow");
  StepBuf &B = b->buf[b->flip];
  Dev<float> pctr;
  XF_TRY(pctr.reserve(b->R));
  XF_TRY(front_pull(st, b, B, st->main));
  XF_TRY(front_compute(st, b, B, pctr.p, false, st->main));
  XF_TRY(wait_stream(st, st->main));
  if (b->R) XF_HIP(hipMemcpy(pctr_out, pctr.p, (size_t)b->R * 4, hipMemcpyDeviceToHost));
  return XF_OK;
}

// renumber this shard's state rows in key order (table maintenance between steps; local to the
// rank).  Applies an outstanding stale1 Push first: row numbers held by a step in flight would
// go stale.
extern "C" int xf_sharded_defrag(xf_sharded *st) {
  XF_REQUIRE(st, "xf_sharded_defrag: null trainer");
  XF_TRY(xf_sharded_flush(st));
  XF_TRY(xf_table_defrag(st->tw));
  if (st->tv) XF_TRY(xf_table_defrag(st->tv));
  return XF_OK;
}

extern "C" int xf_sharded_check(xf_sharded *st) {
  XF_REQUIRE(st, "xf_sharded_check: null trainer");
  XF_TRY(xf_sharded_flush(st));
  XF_TRY(xf_table_check(st->tw, nullptr));
  if (st->tv) XF_TRY(xf_table_check(st->tv, nullptr));
  return XF_OK;
}

// per-stage timing with HIP events on the step's stream (sequential schedule; at world 1 the
// fused step's own profile): ms_sum[6] = owner pull, weights exchange, forward, gradient,
// gradients exchange, owner update
extern "C" int xf_sharded_profile(xf_sharded *st, int enable) {
  XF_REQUIRE(st, "xf_sharded_profile: null trainer");
  if (st->fused) return xf_workspace_profile(st->ws, enable);
  if (enable && !st->sets[0].ev[0])
    for (auto &e : st->sets)
      for (auto &ev : e.ev) XF_HIP(hipEventCreate(&ev));
  if (!enable) XF_TRY(collect_profile(st));
  st->profiling = enable != 0;
  if (enable) {
    for (auto &m : st->ms_sum) m = 0.0;
    st->steps_timed = 0;
    st->step_no = 0;
    for (auto &e : st->sets) e.pending = false;
  }
  return XF_OK;
}

extern "C" int xf_sharded_profile_read(xf_sharded *st, double *ms_sum, long *steps) {
  XF_REQUIRE(st && ms_sum && steps, "xf_sharded_profile_read: null argument");
  if (st->fused) {
    double m5[5];
    XF_TRY(xf_workspace_profile_read(st->ws, m5, steps));
    ms_sum[0] = m5[0] + m5[1];
    ms_sum[1] = 0;
    ms_sum[2] = m5[2];
    ms_sum[3] = m5[3];
    ms_sum[4] = 0;
    ms_sum[5] = m5[4];
    return XF_OK;
  }
  XF_TRY(collect_profile(st));
  for (int i = 0; i < kEvN; ++i) ms_sum[i] = st->ms_sum[i];
  if (owner_dataflow(st->cfg.schedule)) {
    // recorded in the order of the owner-compute step: forward at the owners | row sums to the
    // workers + sigmoid | losses to the owners | gradient + Pushes
    ms_sum[0] = 0;
    ms_sum[1] = st->ms_sum[1];
    ms_sum[2] = st->ms_sum[0];
    ms_sum[3] = st->ms_sum[3];
    ms_sum[4] = st->ms_sum[2];
    ms_sum[5] = 0;
  }
  *steps = st->steps_timed;
  return XF_OK;
}

extern "C" int xf_sharded_set_schedule(xf_sharded *st, int schedule) {
  XF_REQUIRE(st && (schedule == XF_SCHEDULE_SEQUENTIAL || schedule == XF_SCHEDULE_STALE1 ||
                    owner_dataflow(schedule)),
             "xf_sharded_set_schedule: bad argument");
  XF_REQUIRE(owner_dataflow(st->cfg.schedule) == owner_dataflow(schedule),
             "xf_sharded_set_schedule: minibatches compiled for the owner-compute dataflow "
             "cannot be stepped on the weight / gradient exchange, nor the other way round");
  XF_REQUIRE(st->cfg.model == 0 || schedule != XF_SCHEDULE_OWNER_STALE1 ||
                 st->cfg.update_rule == XF_UPDATE_RANK_ORDERED,
             "xf_sharded_set_schedule: owner_stale1 for FM needs update_rule rank_ordered");
  XF_TRY(xf_sharded_flush(st));
  st->cfg.schedule = schedule;
  return XF_OK;
}

extern "C" int xf_sharded_set_parity(xf_sharded *st, int mode) {
  XF_REQUIRE(st, "xf_sharded_set_parity: null trainer");
  XF_REQUIRE(st->fused && st->ws, "xf_sharded_set_parity: one rank only (the fused step)");
  XF_REQUIRE(mode == XF_PARITY_EXACT_SUMS || st->cfg.model == 1 || st->cfg.host_key_build,
             "xf_sharded_set_parity: the reference-order forward needs minibatches with a key "
             "list (host_key_build)");
  XF_TRY(xf_workspace_parity(st->ws, mode));
  // (FM minibatches compiled so far in the default mode may be keyed — no index of their key
  // list: the reference-order step refuses those; minibatches compiled from here on are built
  // for the mode set)
  st->parity_mode = mode;
  return XF_OK;
}

extern "C" int xf_sharded_stream(xf_sharded *st, void **stream) {
  XF_REQUIRE(st && stream, "xf_sharded_stream: null argument");
  *stream = (void *)st->main;
  return XF_OK;
}

// ---- checkpoint of the sharded table --------------------------------------------------------
namespace xf {
int model_write(const char *path, xf_table *tw, xf_table *tv, int k);
int model_read(const char *path, xf_table *tw, xf_table *tv, int k, uint32_t shard,
               uint32_t nshards);
}  // namespace xf

static std::string shard_path(const char *prefix, int r, int n) {
  char buf[64];
  snprintf(buf, sizeof(buf), ".shard-%05d-of-%05d", r, n);
  return std::string(prefix) + buf;
}

extern "C" int xf_sharded_save(xf_sharded *st, const char *prefix) {
  XF_REQUIRE(st && prefix, "xf_sharded_save: null argument");
  XF_TRY(xf_sharded_flush(st));
  int rc = xf::model_write(shard_path(prefix, st->rank, st->world).c_str(), st->tw, st->tv,
                           st->cfg.k);
  if (rc == XF_OK && st->rank == 0) {
    const std::string m = std::string(prefix) + ".manifest";
    FILE *f = fopen(m.c_str(), "w");
    if (!f) rc = xf::set_error(XF_EIO, "xf_sharded_save: cannot write %s", m.c_str());
    else {
      fprintf(f, "xflow_amd sharded model\nshards %d\nmodel %d\nk %d\n", st->world, st->cfg.model,
              st->cfg.model == 1 ? st->cfg.k : 0);
      fclose(f);
    }
  }
  // everybody learns whether everybody succeeded
  int32_t ok = rc == XF_OK ? 1 : 0;
  if (st->g && st->world > 1) {
    std::vector<int32_t> all(st->world);
    XF_TRY(xf_group_allgather_host(st->g, &ok, 4, all.data()));
    for (int r = 0; r < st->world; ++r)
      if (!all[r] && rc == XF_OK)
        rc = xf::set_error(XF_EIO, "xf_sharded_save: rank %d failed to write its shard", r);
  }
  return rc;
}

extern "C" int xf_sharded_load(xf_sharded *st, const char *prefix) {
  XF_REQUIRE(st && prefix, "xf_sharded_load: null argument");
  XF_TRY(xf_sharded_flush(st));
  int saved = 0;
  {
    const std::string m = std::string(prefix) + ".manifest";
    // a single-table model file of the same name that is NEWER than the manifest wins (a
    // one-worker save after a several-worker one; Worker::save_model also removes the manifest)
    struct stat sm, sf;
    const bool single_newer = stat(m.c_str(), &sm) == 0 && stat(prefix, &sf) == 0 &&
                              S_ISREG(sf.st_mode) &&
                              (sf.st_mtim.tv_sec > sm.st_mtim.tv_sec ||
                               (sf.st_mtim.tv_sec == sm.st_mtim.tv_sec &&
                                sf.st_mtim.tv_nsec > sm.st_mtim.tv_nsec));
    FILE *f = single_newer ? nullptr : fopen(m.c_str(), "r");
    if (f) {
      char line[256];
      while (fgets(line, sizeof(line), f))
        if (sscanf(line, "shards %d", &saved) == 1) break;
      fclose(f);
      XF_REQUIRE(saved >= 1, "xf_sharded_load: %s names no shard count", m.c_str());
    }
  }
  int rc = XF_OK;
  if (saved == 0) {  // a single-table model file (XFSaveModel)
    rc = xf::model_read(prefix, st->tw, st->tv, st->cfg.k, (uint32_t)st->rank,
                        (uint32_t)st->world);
  } else {
    // saved shard s holds keys [span_s * s, span_s * (s+1)): read the ones that overlap mine
    const uint64_t my_span = UINT64_MAX / st->world, sv_span = UINT64_MAX / saved;
    const uint64_t my_lo = my_span * st->rank;
    const uint64_t my_hi = st->rank == st->world - 1 ? UINT64_MAX : my_lo + my_span - 1;
    for (int s = 0; s < saved && rc == XF_OK; ++s) {
      const uint64_t lo = sv_span * s, hi = s == saved - 1 ? UINT64_MAX : lo + sv_span - 1;
      if (hi < my_lo || lo > my_hi) continue;
      rc = xf::model_read(shard_path(prefix, s, saved).c_str(), st->tw, st->tv, st->cfg.k,
                          (uint32_t)st->rank, (uint32_t)st->world);
    }
  }
  if (rc == XF_OK) {  // the host-side bound on the key count starts from what was loaded
    uint64_t n = 0;
    rc = xf_table_size(st->tw, &n);
    st->seen_upper = n;
  }
  int32_t ok = rc == XF_OK ? 1 : 0;
  if (st->g && st->world > 1) {
    std::vector<int32_t> all(st->world);
    XF_TRY(xf_group_allgather_host(st->g, &ok, 4, all.data()));
    for (int r = 0; r < st->world; ++r)
      if (!all[r] && rc == XF_OK)
        rc = xf::set_error(XF_EIO, "xf_sharded_load: rank %d failed to read its shard", r);
  }
  return rc;
}
