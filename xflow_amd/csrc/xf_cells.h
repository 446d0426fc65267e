// xf_cells.h — the cell-sorted minibatch ("cells") the LR kernels stream.
//
// One array of 4-byte entries, one per nonzero, grouped by CELL = (row window, key chunk):
//   row window  W <= kWinMax consecutive examples of the minibatch (their fp64 row accumulators
//               fit a CU's LDS: 17408 x 8 B = 136 KiB),
//   key chunk   kChunk = 2048 consecutive positions of the INDEX SPACE the batch was compiled
//               against: state rows of a table on this GPU (mode kCellsTableRows), or the
//               batch's own sorted-unique-key index (mode kCellsUidx, the multi-GPU worker side,
//               where the weights arrive as a dense U-array from the owning shards).
//   entry       (chunk number & kTagMask) << kTagShift | (row within the window) << kChunkBits |
//               (position within the chunk)          [2048-position chunks: & 31, << 27, << 11]
// and cellptr[nwin * nchunk + 1], the offsets of the cells in window-major order.  Inside a
// cell the entries keep the row-major order of the input (the grouping is a stable sort on the
// cell number), i.e. they are sorted by row.
//
// Both sparse products of the step read this one stream:
//   forward   wx[row] += w[idx]     a workgroup owns a slice of ONE window's entries; the
//                                   window's row sums live in LDS (fp64 atomics), the gathers
//                                   of w stay inside one 8 KiB chunk at a time (L1-resident)
//   gradient  g[idx]  += loss[row]  a workgroup owns ONE chunk (all windows); the chunk's 2048
//                                   sums live in LDS, the loss gathers of a cell walk the
//                                   window's rows in ascending order
// which replaces the CSR (rowptr/uidx), its panel-major copy (pptr/pidx), the key-grouped COO
// (segptr/coo_row) and the tile lists of the round-1 layout: 4 bytes per nonzero instead of
// ~20, and no Pull pass (the forward reads the table's weight array in place).
#ifndef XF_CELLS_H_
#define XF_CELLS_H_

#include <stdint.h>

#include "xf_common.h"

namespace xf {

#ifndef XF_CHUNK_BITS
#define XF_CHUNK_BITS 11
#endif
#ifndef XF_WIN_MAX
#define XF_WIN_MAX 17408
#endif
constexpr int kChunkBits = XF_CHUNK_BITS;
constexpr uint32_t kChunk = 1u << kChunkBits;  // index positions per chunk
constexpr uint32_t kWinMax = XF_WIN_MAX;       // rows per window (136 KiB of fp64 in LDS)
constexpr uint32_t kRowMask = 0x7FFFu;         // 15 bits of row-in-window
// the low bits of the chunk number ride in the bits the row and the position leave free: 5 of
// them up to 4096-position chunks, 4 with 8192
constexpr int kTagShift = kChunkBits + 15 > 27 ? kChunkBits + 15 : 27;
constexpr uint32_t kTagMask = (1u << (32 - kTagShift)) - 1u;
static_assert(kChunkBits >= 8 && kChunkBits <= 13, "chunk of 256 .. 8192 index positions");
constexpr uint32_t kBlk = 1024;                // entries per forward block (blk_cell granule)
// entries one gradient workgroup takes of a chunk (a chunk with more is cut into slices)
#ifndef XF_SLICE_MAX
#define XF_SLICE_MAX (4 * kChunk > 8192 ? 4 * kChunk : 8192)
#endif
constexpr uint32_t kSliceMax = XF_SLICE_MAX;
constexpr uint32_t kNoDump = 0xFFFFFFFFu;

enum { kCellsTableRows = 0, kCellsUidx = 1 };

}  // namespace xf

struct xf_cells {
  uint32_t R = 0, NNZ = 0, M = 0;  // rows, nonzeros, size of the index space
  uint32_t W = 1, nwin = 1, nchunk = 1, ncell = 1, nblk = 0;
  uint32_t G = 1;                  // forward workgroups per window
  // A batch's cells may come in SEGMENTS (xf_keybuild.hip): a chain of xf_cells over the same
  // rows (same W, nwin, G) and disjoint sets of state rows.  Segment chunk numbers are local:
  // chunk c of a segment is chunk chunk0 + c of the index space (M stays the bound of the
  // whole index space).  The forward adds the segments' row sums, the gradient pass runs once
  // per segment.
  uint32_t chunk0 = 0;
  xf_cells *next = nullptr;
  uint32_t nitems = 0, nsplit_chunks = 0;
  int mode = xf::kCellsUidx;
  uint64_t table_uid = 0, epoch = 0;  // kCellsTableRows: valid for this table at this epoch
  char *blob = nullptr, *blob2 = nullptr;  // device allocations: entries + cell offsets; items
  size_t blob_bytes = 0, blob2_bytes = 0;
  uint32_t *entries = nullptr;      // [NNZ] cells sorted by row (the gradient's stream)
  uint32_t *entries_k = nullptr;    // [NNZ] the same cells sorted by key (the forward's stream;
                                    //       == entries when the copy was not built)
  bool entries_k_ready = false;     // (a forward before cells_key_sorted_copy reads `entries`)
  const uint32_t *fwd_entries() const { return entries_k_ready ? entries_k : entries; }
  uint32_t *cellptr = nullptr;      // [ncell + 1]
  uint32_t *blk_cell = nullptr;     // [nblk + 1] cell of entry kBlk*b; [nblk] = ncell - 1
  uint32_t *plan = nullptr;         // [4 * (nchunk + 1)] slices per chunk and three scans of them
  uint32_t *item_chunk = nullptr;   // [nitems]   gradient work items: chunk,
  uint32_t *item_slice = nullptr;   // [nitems]   slice | nslices << 16,
  uint32_t *item_dump = nullptr;    // [nitems]   index of the chunk among the split ones
  uint32_t *split_chunk = nullptr;  // [nsplit_chunks] chunks cut into several items
  uint8_t *item_done = nullptr;     // [nitems] scratch of the several-worker gradient pass: 1 = the
                                    //          item was taken by k_lr_grad_multi (written per pass)
  double *gsum = nullptr;           // [nsplit_chunks * kChunk] their key sums (fp64 atomics)
  uint8_t *gtouched = nullptr;      // [nsplit_chunks * kChunk] 1 = the minibatch holds the key
  size_t split_bytes = 0;           // gsum + gtouched: zero between gradient passes (the finish
                                    // kernel leaves them so; a memset only while split_dirty)
  mutable bool split_dirty = true;
};

namespace xf {

// Build the cells of a minibatch on the device.  idx[NNZ]: index-space position of every
// nonzero in row-major order — map == null: idx[j] = src[j]; else idx[j] = map[src[j]] (the
// unique-key index of a compiled batch mapped to table rows).  Synchronises `stream`.
// d_rowid != null: the nonzeros come in any order with their row number (d_rowptr unused, no
// map); the rows are then numbered window by window by the caller: w_fixed rows per window.
// chunk0 != 0: a segment over the chunks from chunk0 on (every position is >= chunk0 * kChunk).
int cells_build(xf_cells **out, const uint32_t *d_src, const uint32_t *d_map,
                const uint32_t *d_rowptr, uint32_t R, uint32_t NNZ, uint32_t M, int mode,
                bool key_sorted_copy, hipStream_t stream, const uint32_t *d_rowid = nullptr,
                uint32_t w_fixed = 0, uint32_t chunk0 = 0);
void cells_free(xf_cells *c);  // the whole chain

// The key build of LRWorker::update (lr_worker.cc:146-166) against table `t` on this GPU,
// raw keys in, cells out (xf_keybuild.hip): the keys' state rows are found — and first-touch
// keys inserted (ftrl.h:56) — on the way.  Nonzeros in CSR order (d_rowptr) or with their row
// numbers (d_rowid, rows numbered window by window, w_fixed rows per window).  Synchronises
// `stream`.
//
// `defer` (optional): the build's one host wait — for the number of gradient work items and
// for the keys the settled tier does not hold — is left to cells_build_keyed_finish.  Until
// then *out serves the FORWARD only (entries, cell offsets: in stream order); keys the tier
// does not hold are holes in it.  *defer stays null when there is nothing left to wait for
// (the general build ran).  One deferred build at a time.
struct KbDeferred;
int cells_build_keyed(xf_cells **out, xf_table *t, const uint64_t *d_keys,
                      const uint32_t *d_rowptr, const uint32_t *d_rowid, uint32_t R,
                      uint32_t NNZ, bool key_sorted_copy, uint32_t w_fixed, hipStream_t stream,
                      KbDeferred **defer = nullptr);
// waits for the build's kernels (an event, not the stream: what was launched after them keeps
// running), then the work items and — *more = true — a second segment of cells over the keys
// the tier did not hold (inserted now): the forward has to be run again over both.  Frees d.
int cells_build_keyed_finish(KbDeferred *d, hipStream_t stream, bool *more);

// pieces of cells_build shared with the keyed build
int cells_alloc(xf_cells **out, uint32_t R, uint32_t NNZ, uint32_t M, int mode,
                bool key_sorted_copy, uint32_t w_fixed, uint32_t chunk0);
int cells_plan_items(xf_cells *c, hipStream_t stream);            // after cellptr is final
int cells_fill_items(xf_cells *c, uint32_t nitems, uint32_t nsplit, hipStream_t stream);
int cells_key_sorted_copy(xf_cells *c, hipStream_t stream);
uint32_t cells_split_chunks(const xf_cells *c);  // over the chain

// scratch of the forward: G * nwin * W partial row sums (fp64)
size_t cells_partial_doubles(const xf_cells *c);

// forward: loss[r] = sigmoid(sum_j w[idx_j]) - label[r]   (lr_worker.cc:121-143)
// resume_from: a later segment of c's chain — d_partial still holds the partial row sums a call
// over the segments before it left there: only the rest is run (and the rows finalized again)
int cells_lr_forward(const xf_cells *c, const float *d_w, const int32_t *d_labels,
                     double *d_partial, float *d_loss, float *d_pctr, hipStream_t stream,
                     const xf_cells *resume_from = nullptr);

// owner-compute step (xf_sharded.hip): forward up to the fp64 row sums, and the gradient with
// the Pushes of several workers applied in rank order (see xf_cells.hip)
int cells_lr_forward_sums(const xf_cells *c, const float *d_w, double *d_partial,
                          const uint32_t *d_out_base, const uint32_t *d_out_rows,
                          double *d_rowsum, hipStream_t stream);
int cells_lr_grad_update_sources(const xf_cells *c, const xf_table *t, const float *d_loss,
                                 uint32_t n, const uint32_t *d_win, const uint32_t *d_rows,
                                 const uint32_t *d_loss_base, double *d_gsum,
                                 uint8_t *d_gtouched, hipStream_t stream,
                                 uint32_t rows_if_one = 0 /* n == 1: d_rows[0], for the host */);

}  // namespace xf

#endif  // XF_CELLS_H_
