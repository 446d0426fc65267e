// xf_batch.h — the compiled minibatch (host arrays + device mirror).
#ifndef XF_BATCH_H_
#define XF_BATCH_H_

#include <hip/hip_runtime_api.h>

#include <vector>

#include "xf_common.h"

struct xf_cells;

struct xf_batch {
  uint32_t R = 0, NNZ = 0, U = 0, H = 0;
  std::vector<uint64_t> ukeys;
  std::vector<uint32_t> rowptr, uidx, segptr, coo_row, heavy;
  std::vector<uint32_t> hchunk_ptr;  // chunks of XF_TILE_NNZ occurrences per heavy key
  uint32_t P = 0;  // forward panels (0 = none)
  std::vector<uint32_t> pptr, pidx;
  std::vector<uint32_t> tile_ptr;  // gradient tiles (key ranges)
  std::vector<uint32_t> ftile_ptr, fpanel_first;  // forward tiles over (panel,row) cells
  uint32_t fwd_grid = 0;
  std::vector<int32_t> labels;
  bool on_device_only = false;  // built by xf_batch_compile_dev and not downloaded yet
  void *d_blob = nullptr;  // one device allocation holding all arrays
  size_t d_blob_bytes = 0;
  uint64_t fm_nbase = 0;     // fm_keyed: rows of the settled tier the batch was compiled against
  bool fm_same_rows = false; // fm_keyed: key of rank r has row r in BOTH tables (until a renumbering)
  void *d_blob2 = nullptr;  // fm_keyed: the lists whose sizes are known last (heavy keys, tiles)
  size_t d_blob2_bytes = 0;
  xf_dev_batch view{};
  // the cell-sorted form the LR kernels stream (xf_cells.h), compiled against ONE table's row
  // numbering (rebuilt when another table or another epoch of it comes along)
  xf_cells *cells = nullptr;
  uint32_t *d_rows_u = nullptr;  // state row of each unique key [U] (batches with a key list)
  // parity mode "reference order": every row's unique-key indices in ascending order (= in
  // ascending fid: the order of the reference's merge-join, lr_worker.cc:127-138), built once
  uint32_t *d_uidx_sorted = nullptr;
  // ... and, for the gradient, the rows of every key's occurrences in the order the reference's
  // own key build leaves them (std::sort by fid of the row-major all_keys, lr_worker.cc:150-162):
  // [NNZ], grouped by key like view.coo_row (view.segptr holds the offsets)
  uint32_t *d_ref_coo = nullptr;
  // FM: the unique keys' rows in the w and the v table, valid for one (uid, epoch) of each —
  // the Pulls of a replayed minibatch resolve nothing
  uint32_t *d_fm_rows[2] = {nullptr, nullptr};
  size_t fm_rows_bytes[2] = {0, 0};  // (pooled allocations, xf::blob_alloc: a fresh minibatch
  size_t fm_ridx_bytes = 0;          //  per step would otherwise pay three hipMalloc / hipFree)
  uint64_t fm_uid[2] = {0, 0}, fm_epoch[2] = {0, 0};
  // FM with the per-key records kept next to the v table's rows (xf_model.hip): the v row of
  // every nonzero's key [NNZ] (valid with d_fm_rows[1]); what the records of this minibatch's
  // keys were last brought up to date against (records allocation, foreign writes to w and v)
  uint32_t *d_fm_ridx = nullptr;
  uint64_t fm_ridx_uid = 0, fm_ridx_epoch = ~0ull;
  uint64_t fm_rec_gen = 0, fm_rec_writes[2] = {~0ull, ~0ull};
  bool fm_rec_ok = false;
  // built by xf_batch_compile_fm_dev against the settled tiers of ONE (w, v) pair of tables with
  // the same row numbering: the key list comes with its state rows and the per-nonzero record
  // index, there is no CSR index of unique keys (view.uidx == null).  Valid for that numbering
  // only: after a defrag the minibatch must be compiled again.
  bool fm_keyed = false;
  // "local" batches (xf_batch_compile_local_*): no key list at all — the raw keys were resolved
  // straight to state rows.  The raw arrays are kept (device) when the cells must be
  // rebuildable after the table renumbers its rows.
  bool local = false;
  void *d_raw = nullptr;  // keys u64[NNZ] | rowptr u32[R+1] | labels i32[R]
  size_t d_raw_bytes = 0;
  const uint64_t *raw_keys = nullptr;
  const uint32_t *raw_rowptr = nullptr;
  const int32_t *raw_labels = nullptr;
};

namespace xf {
// Device blobs of compiled minibatches are recycled through a small pool: a worker that
// streams blocks (compile -> step -> free per block) would otherwise pay a hipMalloc and a
// hipFree of ~200 MB per minibatch, which costs more than the key build itself.
int blob_alloc(void **p, size_t bytes, size_t *got);
void blob_free(void *p, size_t bytes);
void blob_pool_limit(int blobs);  // 0 = no pooling (every free goes back to the driver)
// (key, position) of d_keys[0..n) in key order, positions ascending inside a key (xf_keybuild.hip:
// uniform key ranges over [lo, lo + span], a range sorted in LDS).  *done = false: not sorted —
// beyond its limits — and the caller sorts some other way.  Waits for the stream.
// site: who sorts — the sort remembers per call site (and host thread) whether the stream it saw
// last was skewed, and looks for hot keys ahead of the partition when it was
enum { kSortSiteAny = 0, kSortSiteBatch = 1, kSortSiteMerged = 2, kSortSiteLr = 3, kSortSites = 4 };
int sort_key_pos(const uint64_t *d_keys, uint32_t n, uint64_t lo, uint64_t span, uint64_t *sk,
                 uint32_t *spos, hipStream_t s, bool *done, uint32_t site = kSortSiteAny);
// xf_batch_compile_dev, with or without the panel-major forward view (xf_batch_dev.hip)
int batch_compile_dev_ex(xf_batch **out, const uint64_t *d_keys, const uint32_t *d_rowptr,
                         const int32_t *d_labels, uint32_t R, uint32_t NNZ, hipStream_t stream,
                         bool panels);
// LR, the worker side of the weight / gradient exchange (xf_keybuild.hip): a minibatch with its
// sorted unique keys, row offsets and labels on the device — nothing else — and its cells over
// the unique-key index.  *done = false: beyond that build's limits, nothing was built.
int batch_compile_lr_dev(xf_batch **out, xf_cells **cells, const uint64_t *d_keys,
                         const uint32_t *d_rowptr, const int32_t *d_labels, uint32_t R,
                         uint32_t NNZ, bool key_sorted_copy, hipStream_t s, bool *done);
// ... by that sort, or beyond its limits by the library's radix sort (xf_batch_dev.hip)
int sort_key_pos_any(const uint64_t *d_keys, uint32_t n, uint64_t lo, uint64_t span, uint64_t *sk,
                     uint32_t *spos, hipStream_t s, bool *by_hand, uint32_t site = kSortSiteAny);
}  // namespace xf

#endif  // XF_BATCH_H_
