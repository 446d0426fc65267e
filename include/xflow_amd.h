/*
 * xflow_amd.h — C ABI of libxflow_amd.so: the MI355X-native replacement for the
 * ps-lite push/pull parameter server + worker math of xswang/xflow.
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.  Every
 * function returns 0 on success and a non-zero XF_E* code on failure; xf_last_error()
 * returns a thread-local description.  Handles are opaque and NOT thread-safe (one
 * driver thread per device/stream, like one ps-lite customer thread per server).
 *
 * What each group replaces in the reference (paths relative to /root/reference):
 *   xf_hash_bytes / xf_reader_*   src/io/io.h:53, src/io/load_data_from_disk.cc:103-210
 *   xf_table_*                    ps::KVWorker<float>::{Pull,Push,Wait} as used at
 *                                 src/model/lr/lr_worker.cc:170,175 and
 *                                 src/model/fm/fm_worker.cc:228-242, plus the server
 *                                 handlers src/optimizer/ftrl.h:38-152, sgd.h:30-109 and
 *                                 their wiring src/model/server.h:22-31
 *   xf_batch_*                    key build in LRWorker::update, lr_worker.cc:146-166
 *   xf_lr_* / xf_fm_*             calculate_loss / calculate_gradient / update,
 *                                 lr_worker.cc:100-177, fm_worker.cc:126-245
 *   xf_auc_logloss                Base::calculate_auc, src/base/base.h:84-110
 *   XFCreate / XFStartTrain       src/c_api/c_api.h:26-29 (signatures kept verbatim)
 */
#ifndef XFLOW_AMD_H_
#define XFLOW_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XF_OK 0
#define XF_EINVAL 1   /* bad argument */
#define XF_EHIP 2     /* HIP runtime error */
#define XF_EFULL 3    /* table capacity exhausted */
#define XF_EIO 4      /* file open / read error */
#define XF_EPARSE 5   /* malformed libsvm-style input */
#define XF_ENOGPU 6   /* no usable HIP device */

const char *xf_last_error(void);
int xf_version(void);
int xf_device_count(int *count);

/* ---------------------------------------------------------------- key hash / sharding */
/* libstdc++ std::hash<std::string> (io.h:53), bit-exact */
uint64_t xf_hash_bytes(const void *ptr, size_t len);
/* out[i] = hash of the decimal string of (start + i): synthetic fids "0","1",... */
int xf_hash_decimal_range(uint64_t start, size_t n, uint64_t *out);
/* out[i] = hash of the decimal string of ids[i] */
int xf_hash_decimal_ids(const uint64_t *ids, size_t n, uint64_t *out);
/* ps-lite default key-range owner: min(key / (UINT64_MAX / nshards), nshards-1) */
uint32_t xf_shard_of(uint64_t key, uint32_t nshards);

/* ---------------------------------------------------------------- text block reader   */
/* load_minibatch_hash_data_fread (load_data_from_disk.cc:103-210): block = what fits in
 * cap_bytes-1 bytes, cut at the last newline when the buffer fills; label = atof > 1e-7;
 * key = hash of the middle field of fgid:fid:val; val is never read. */
typedef struct xf_reader xf_reader;
int xf_reader_open(xf_reader **out, const char *path, size_t cap_bytes);
/* The same reader over a binarized block cache: when `cache_path` holds the
 * blocks of this exact file (size, mtime) at this block size, they are served from it and the
 * text is not touched (*from_cache = 1); otherwise the text is parsed as above and every block
 * is also written to the cache, which becomes valid once a pass has reached end of file. */
int xf_reader_open_cached(xf_reader **out, const char *path, size_t cap_bytes,
                          const char *cache_path, int *from_cache);
int xf_reader_close(xf_reader *r);
/* *yes = 1: the text is a memory mapping (a non-empty regular file) — what xf_reader_peek_text /
 * _copy_text / _skip_text (the GPU tokeniser's feed) work on; 0: an empty file, a pipe: only
 * xf_reader_next* serve it. */
int xf_reader_mapped(xf_reader *r, int *yes);
/* rows_out = 0 at end of file.  Arrays are owned by the reader, valid until next call. */
int xf_reader_next(xf_reader *r, size_t *rows_out, size_t *nnz_out,
                   const uint64_t **rowptr, const uint64_t **keys, const int32_t **fgid,
                   const int32_t **labels);

/* The same, into a block object the caller owns: the arrays stay valid until the block is
 * reused or destroyed, also while the reader already parses the next block on another thread. */
typedef struct xf_block xf_block;
int xf_block_create(xf_block **out);
int xf_block_destroy(xf_block *b);
int xf_reader_next_into(xf_reader *r, xf_block *blk, size_t *rows_out, size_t *nnz_out,
                        const uint64_t **rowptr, const uint64_t **keys, const int32_t **fgid,
                        const int32_t **labels);

/* ---- the block as text, for a caller that tokenises it itself (xf_ingest_*) ----
 * xf_reader_peek_text: the next block's text by the block rule above (*len = 0 at the end of the
 * file; mapped files without a block cache); nothing moves until xf_reader_skip_text — any
 * xf_reader_next* call instead parses that block on the host and moves on.
 * xf_reader_copy_text: the peeked text copied to `dst` by `threads` host threads.
 * xf_reader_parse_text: a block's text in memory parsed on the host into `blk` (what
 * xf_reader_next_into yields for that block). */
int xf_reader_peek_text(xf_reader *r, const char **text, size_t *len);
int xf_reader_skip_text(xf_reader *r);
int xf_reader_copy_text(xf_reader *r, char *dst, size_t cap, size_t *len, int threads);
int xf_reader_parse_text(xf_reader *r, const char *text, size_t len, xf_block *blk,
                         size_t *rows_out, size_t *nnz_out, const uint64_t **rowptr,
                         const uint64_t **keys, const int32_t **fgid, const int32_t **labels);

/* ---------------------------------------------------------------- GPU tokeniser       */
/* load_minibatch_hash_data_fread's token loop (load_data_from_disk.cc:126-208) and
 * std::hash<std::string> of every fid (io.h:53) as byte-parallel kernels, for blocks of the
 * common shape — lines "('0'|'1') '\t' field0:fid:rest (' ' field0:fid:rest)* '\n'" with single
 * blanks and no control bytes: label = the digit, one key per token = _Hash_bytes(fid), bit for
 * bit the host parser's arrays.  Any other block (other labels, empty tokens — which the
 * reference duplicates —, CR LF, NUL, a token without two colons, ...) comes back with *ok = 0 and
 * is the host parser's (xf_reader_parse_text).  Not in the reference: its io path is host code;
 * this feeds xf_sharded_compile_dev / xf_lr_update_dev without the keys ever being host arrays. */
typedef struct xf_ingest xf_ingest;
int xf_ingest_create(xf_ingest **out, size_t max_text_bytes);
int xf_ingest_destroy(xf_ingest *g);
/* the pinned staging buffer the next block's text goes to (xf_reader_copy_text) */
int xf_ingest_staging(xf_ingest *g, char **buf, size_t *cap);
/* the staged text on its way to the device, asynchronously on `stream` (a staging thread's own:
 * the copy of the next block runs while the GPU works on this one); the next
 * xf_ingest_block(g, NULL, len, ...) waits for that copy instead of making one */
int xf_ingest_upload(xf_ingest *g, size_t len, void *stream);
/* text == NULL: `len` bytes are in the staging buffer already.  Uploads, tokenises, waits for
 * the counts.  d_keys [nnz] u64, d_rowptr [rows + 1] u32, d_labels [rows] i32: device arrays
 * owned by `g`, valid until its next call. */
int xf_ingest_block(xf_ingest *g, const char *text, size_t len, void *stream,
                    const uint64_t **d_keys, const uint32_t **d_rowptr, const int32_t **d_labels,
                    uint32_t *rows, uint32_t *nnz, int *ok);

/* device memory -> host, blocking (the tokeniser's arrays in tests and small tools) */
int xf_copy_to_host(void *dst, const void *d_src, size_t bytes);

typedef struct xf_table xf_table;

/* ---------------------------------------------------------------- compiled minibatch  */
/* Host-side key build (lr_worker.cc:146-166): sorted unique keys (== unique_keys, the
 * Pull/Push key list) + the two views of all_keys the kernels walk:
 *   CSR:  rowptr[R+1], uidx[NNZ]   (index of each nnz's key in ukeys, row-major order)
 *   COO grouped by key: segptr[U+1], coo_row[NNZ]  (row ids of each key's occurrences)
 *   heavy[H]: indices u whose segment is longer than XF_HEAVY_SEG (wave-per-key path) */
#define XF_HEAVY_SEG 64
#define XF_TILE_NNZ 2048
#define XF_TILE_KEYS 2048
/* gradient tiles are much smaller than forward tiles: a gradient workgroup goes through
 * load -> barrier -> sum -> store once per tile and waits on memory in between, so what counts
 * is how many independent tiles a CU has in flight (FM k = 16, 1e7 occurrences: 482 us with
 * 2048-occurrence tiles, 402 / 340 / 324 us with 1024 / 512 / 256). */
#ifndef XF_GRAD_TILE_NNZ /* (overridable at build time: experiments) */
#define XF_GRAD_TILE_NNZ 256
#endif
#define XF_GRAD_TILE_KEYS XF_GRAD_TILE_NNZ
#define XF_HEAVY_KMAX 64
typedef struct xf_batch xf_batch; /* host arrays + (after upload) device mirror */
int xf_batch_compile(xf_batch **out, const uint64_t *rowptr, const uint64_t *keys,
                     const int32_t *labels, size_t row_begin, size_t row_end);
/* The same key build on the GPU (rocPRIM radix sort + flag/scan/scatter kernels): the
 * compiled batch is born on the device, identical array for array to xf_batch_compile's.
 * _dev takes raw device arrays (d_rowptr row-relative, d_rowptr[0] == 0); _gpu takes the
 * reader's host arrays and a row slice like xf_batch_compile and uploads the raw slice. */
int xf_batch_compile_dev(xf_batch **out, const uint64_t *d_keys, const uint32_t *d_rowptr,
                         const int32_t *d_labels, uint32_t R, uint32_t NNZ, void *stream);
int xf_batch_compile_gpu(xf_batch **out, const uint64_t *rowptr, const uint64_t *keys,
                         const int32_t *labels, size_t row_begin, size_t row_end, void *stream);
/* The FM key build (fm_worker.cc:205-225) against the tables themselves: when every key of the
 * minibatch sits in the v table's settled tier (xf_table_defrag) and the w table numbers its rows
 * the same way, the key list comes with its state rows, the key-grouped occurrence lists and
 * the forward's per-nonzero record index from the range-partitioned build (no sort of (key,
 * position) pairs); otherwise this IS xf_batch_compile_dev.  *keyed (optional): 1 when the
 * fast build ran.  Such a minibatch is for xf_fm_step / xf_fm_predict on these two tables
 * (k in {4, 8, 16, 32, 64}, no capture, no parity mode); after a defrag its rows are looked up
 * again and its record index translated at its next use.  xf_batch_compile_fm: the host-array
 * front end (the reader's block arrays and a row slice, like xf_batch_compile_gpu). */
int xf_batch_compile_fm_dev(xf_batch **out, xf_table *w, xf_table *v, const uint64_t *d_keys,
                            const uint32_t *d_rowptr, const int32_t *d_labels, uint32_t R,
                            uint32_t NNZ, void *stream, int *keyed);
int xf_batch_compile_fm(xf_batch **out, xf_table *w, xf_table *v, const uint64_t *rowptr,
                        const uint64_t *keys, const int32_t *labels, size_t row_begin,
                        size_t row_end, void *stream, int *keyed);
/* The key build for a table on THIS GPU, without the sort (LR): every raw key is resolved
 * straight to its state row in `t` (insert on first touch, ftrl.h:56; the table grows when
 * needed) and the nonzeros are grouped into cells (row window x 4096-row chunk of the state)
 * by a stable radix pass — no unique-key list, the state row is the key's identity.  The
 * result feeds xf_lr_step / xf_lr_predict on `t` only.  retain_keys != 0 is for minibatches that
 * are kept and replayed (epochs >= 2): the raw arrays stay on the device so that the batch
 * survives a renumbering of the rows (xf_table_defrag), and the cells get the key-sorted copy
 * the forward reads fastest; 0 is for a minibatch that is stepped once and freed. */
int xf_batch_compile_local_dev(xf_batch **out, xf_table *t, const uint64_t *d_keys,
                               const uint32_t *d_rowptr, const int32_t *d_labels, uint32_t R,
                               uint32_t NNZ, int retain_keys, void *stream);
int xf_batch_compile_local(xf_batch **out, xf_table *t, const uint64_t *rowptr,
                           const uint64_t *keys, const int32_t *labels, size_t row_begin,
                           size_t row_end, int retain_keys, void *stream);
/* shape of a batch's cells once a step / predict has built them: out[0..8) = rows per window,
 * windows, chunks, forward workgroups per window, gradient work items, 0 (reserved), chunks
 * cut into several work items, size of the index space */
int xf_batch_cells_info(const xf_batch *b, uint32_t *out);
/* copy a device-built batch's arrays into its host views (xf_batch_host etc. do it on demand) */
int xf_batch_download(xf_batch *b);
int xf_batch_free(xf_batch *b);
int xf_batch_dims(const xf_batch *b, uint32_t *R, uint32_t *NNZ, uint32_t *U, uint32_t *H);
/* host views (owned by the batch) */
int xf_batch_host(const xf_batch *b, const uint64_t **ukeys, const uint32_t **rowptr,
                  const uint32_t **uidx, const uint32_t **segptr,
                  const uint32_t **coo_row, const int32_t **labels,
                  const uint32_t **heavy);
/* panel view (host arrays); *P == 0 when the batch is too small to need panels */
int xf_batch_panels(const xf_batch *b, uint32_t *P, const uint32_t **pptr,
                    const uint32_t **pidx);
int xf_batch_fwd_tiles(const xf_batch *b, uint32_t *ntiles, const uint32_t **tile_ptr,
                       const uint32_t **panel_first, uint32_t *grid);
/* chunk offsets of the heavy keys (host array, H+1 entries) */
int xf_batch_heavy_chunks(const xf_batch *b, uint32_t *H, const uint32_t **chunk_ptr);
/* gradient tiles (host array, ntiles+1 key indices) */
int xf_batch_tiles(const xf_batch *b, uint32_t *ntiles, const uint32_t **tile_ptr);
/* tuning knobs (process-wide): "panel_slice_bytes" (default 1.5 MiB: w_u bytes per panel),
 * "min_panel_nnz" (default 4e6: smaller batches keep the plain CSR forward),
 * "parse_threads" (default 64: threads of the block text parser),
 * "batch_pool_blobs" (default 8: device allocations of freed minibatches kept for reuse by
 * the next compile/upload; 0 returns every one to the driver).
 * Named switches between code paths that are all product code and normally chosen by the shape
 * (tests pin one to run each against the oracle; 0 = by shape): "key_build" (1 the sort-based
 * build, 2 the two-level partition, 3 an empty table's keys through the arrival index too), "old_weight" (1 read, 2 derive from (n, z)), "lr_gradient"
 * (1 the general kernel, 2 / 3 the dense kernel with byte-masked / whole-line stores),
 * "owner_pass" (1 the general loop, 2 / 3 the merged phases, 4 a phase per worker) —
 * xflow_amd/csrc/xf_common.h.  An unknown name, or a value a switch does not take, is XF_EINVAL
 * (a library built with -DXF_EXPERIMENTS also has the experiments' numeric "exp_knob"). */
int xf_tune(const char *name, double value);
/* Size the calling thread's scratch arena of the device-side builders ahead of its first build
 * (a minibatch of NNZ nonzeros needs about 40 B x NNZ + 64 MiB when it meets new keys): a run's
 * first minibatches otherwise grow it build by build.  No-op when it is that large already. */
int xf_scratch_reserve(size_t bytes);
/* Set a device allocation of `bytes` aside in the pool of freed minibatches ("batch_pool_blobs")
 * for the first compile that asks for that much or less (and at least an eighth of it): a run's
 * first minibatch otherwise waits for the driver to map its cells (4-8 B x NNZ; 0.25 ms for
 * 40 MB, measured in its first xf_lr_update_dev).  No-op when such an allocation is waiting
 * already, or without pooling. */
int xf_batch_pool_reserve(size_t bytes);
/* copy to the current device (async on stream); idempotent */
int xf_batch_upload(xf_batch *b, void *stream);

/* device view of a compiled batch: raw device pointers (e.g. torch tensors) */
typedef struct {
  uint32_t R, NNZ, U, H;
  const uint32_t *rowptr;  /* R+1  */
  const uint32_t *uidx;    /* NNZ  */
  const uint64_t *ukeys;   /* U    */
  const uint32_t *segptr;  /* U+1  */
  const uint32_t *coo_row; /* NNZ  */
  const int32_t *labels;   /* R    */
  const uint32_t *heavy;   /* H (may be NULL when H == 0) */
  /* LR forward, panel-major view of the CSR (P == 0: absent, the plain CSR is used).
   * Panel p holds the nonzeros whose uidx lies in [U*p/P, U*(p+1)/P): row r's part is
   * pidx[pptr[p*(R+1)+r] .. pptr[p*(R+1)+r+1]).  Blocks working on panel p gather from
   * one L2-sized slice of w_u; p % 8 selects the XCD. */
  uint32_t P, fwd_ntiles;
  const uint32_t *pptr;    /* P*(R+1) */
  const uint32_t *pidx;    /* NNZ     */
  double *fwd_scratch;     /* P*R partial sums (device scratch owned by the batch) */
  /* forward tiles over the (panel,row) cells s = p*(R+1)+r: tile t = cells
   * [fwd_tile_ptr[t], fwd_tile_ptr[t+1]) of ONE panel, <= XF_TILE_NNZ nonzeros;
   * fwd_panel_first[p] = first tile of panel p.  Workgroup b takes the (b/8)-th tile of the
   * panels p with p % 8 == b % 8 (observed placement: XCD b % 8); fwd_grid = 8 x the longest
   * such list. */
  const uint32_t *fwd_tile_ptr;    /* fwd_ntiles+1 */
  const uint32_t *fwd_panel_first; /* P+1 */
  uint32_t fwd_grid, pad3_;
  /* gradient tiles: tile t covers keys [tile_ptr[t], tile_ptr[t+1]) whose occurrences
   * (<= XF_GRAD_TILE_NNZ of them, <= XF_GRAD_TILE_KEYS keys) are staged through LDS by one
   * workgroup; a heavy key (> XF_HEAVY_SEG occurrences) is a tile of its own, skipped by
   * the tile kernel and handled by the wave-per-key path. */
  uint32_t ntiles, n_heavy_chunks;
  const uint32_t *tile_ptr; /* ntiles+1 */
  /* heavy keys are reduced in chunks of XF_TILE_NNZ occurrences spread over the whole chip
   * (a power-law head key can own millions of occurrences): heavy key h owns chunks
   * [heavy_chunk_ptr[h], heavy_chunk_ptr[h+1]); heavy_scratch holds
   * n_heavy_chunks * (1 + XF_HEAVY_KMAX) partial sums. */
  const uint32_t *heavy_chunk_ptr; /* H+1 (NULL when H == 0) */
  double *heavy_scratch;
} xf_dev_batch;
int xf_batch_dev_view(const xf_batch *b, xf_dev_batch *view);

/* ---------------------------------------------------------------- sharded table       */
enum { XF_OPT_FTRL = 0, XF_OPT_SGD = 1 };
enum {
  XF_INIT_ZERO = 0,    /* ftrl.h:27-36, sgd.h:22-27 */
  XF_INIT_CONST = 1,   /* sgd.h:67-72 (0.001) */
  XF_INIT_HASHNORM = 2 /* deterministic N(0,1)*1e-2 stand-in for ftrl.h:114-120 */
};
typedef struct {
  int32_t opt_kind;     /* XF_OPT_* */
  int32_t dim;          /* 1 for w, k for v */
  int32_t init_kind;    /* XF_INIT_* */
  float init_const;
  uint64_t seed;        /* HASHNORM seed */
  float alpha, beta, lambda1, lambda2; /* ftrl.h:17-20 : 0.05, 1, 5e-5, 10 */
  float lr;                            /* sgd.h:16     : 0.001 */
  uint64_t capacity;    /* slots on THIS shard (keys stored <= capacity) */
  uint32_t shard, nshards; /* key range owned: ps-lite uniform range rule */
} xf_table_config;
void xf_table_config_default(xf_table_config *cfg); /* reference defaults, FTRL, dim 1 */

int xf_table_create(xf_table **out, const xf_table_config *cfg); /* on current device */
int xf_table_destroy(xf_table *t);
int xf_table_size(xf_table *t, uint64_t *nkeys); /* synchronises */
int xf_table_capacity(xf_table *t, uint64_t *slots);
/* keys of the settled tier (sorted, key r owns state row r): what the last xf_table_defrag —
 * or the first-touch build of an empty table's first minibatch — left there; 0 before */
int xf_table_settled(xf_table *t, uint64_t *nkeys);
/* allocate now what xf_table_defrag would allocate (its second state buffer, both allocations of
 * the settled tier at the size the table's rows allow): for a caller whose clock is about to
 * start — a defrag then calls the driver's allocator no more.  Optional. */
int xf_table_prepare_defrag(xf_table *t);
/* grow to new_capacity slots (rehash on device); earlier slot arrays become invalid */
int xf_table_reserve(xf_table *t, uint64_t new_capacity);
/* renumber the state rows in key order (locality of the Pull gather and the Push pass once
 * the key set has settled); row numbers change — call it between steps.  The table keeps a
 * second state buffer of the same size from its first call on (the two swap roles: no
 * allocation, no clearing per call). */
int xf_table_defrag(xf_table *t);
int xf_table_set_hyper(xf_table *t, float alpha, float beta, float l1, float l2, float lr);

/* ps-lite-shaped host API: keys sorted & unique (the KVWorker contract), host pointers,
 * blocking (== Push/Pull followed by Wait).  Pull inserts missing keys (ftrl.h:56). */
int xf_table_pull(xf_table *t, const uint64_t *keys, size_t n, float *vals /* n*dim */);
int xf_table_push(xf_table *t, const uint64_t *keys, size_t n, const float *grads);

/* device-pointer API, asynchronous on `stream` (a hipStream_t; NULL = default stream).
 * resolve: key -> slot with insert-on-miss (+ first-touch init); keys of other shards
 * are an error.  gather: vals[i*dim+j] = w[slot[i]*dim+j].  update: one optimizer step
 * per (slot, j) — slots must be unique within one call. */
int xf_table_resolve_dev(xf_table *t, const uint64_t *d_keys, size_t n, uint32_t *d_slots,
                         void *stream);
int xf_table_gather_dev(xf_table *t, const uint32_t *d_slots, size_t n, float *d_vals,
                        void *stream);
/* resolve + gather in one pass (dim-1 tables, keys unique within the call) */
int xf_table_pull_dev(xf_table *t, const uint64_t *d_keys, size_t n, uint32_t *d_slots,
                      float *d_vals, void *stream);
/* The owner side of a step with N source ranks, in ONE pass over the shard instead of one per
 * source: d_keys_sorted = the concatenated per-source key lists in (stable) ascending key
 * order, d_order[i] = position of sorted entry i in the per-source layout.  Rows / weights
 * are written, gradients read, in the per-source layout.  A key that several sources sent gets
 * their optimizer steps one after the other in source order, as N separate
 * xf_table_update_dev passes in rank order would apply them. */
int xf_table_pull_ordered_dev(xf_table *t, const uint64_t *d_keys_sorted,
                              const uint32_t *d_order, size_t n, uint32_t *d_slots,
                              float *d_vals /* may be null: resolve only */, void *stream);
int xf_table_update_merged_dev(xf_table *t, const uint64_t *d_keys_sorted,
                               const uint32_t *d_order, size_t n, const uint32_t *d_slots,
                               const float *d_grads, void *stream);
int xf_table_update_dev(xf_table *t, const uint32_t *d_slots, size_t n,
                        const float *d_grads, void *stream);
/* raises XF_EFULL / XF_EINVAL recorded by earlier async calls; synchronises the stream */
int xf_table_check(xf_table *t, void *stream);

/* state dump sorted by key (parity hook / checkpoint).  n_/z_ may be NULL. */
int xf_table_export(xf_table *t, uint64_t *keys, float *w, float *n_, float *z_,
                    size_t cap_entries, size_t *n_out);
int xf_table_import(xf_table *t, const uint64_t *keys, size_t n, const float *w,
                    const float *n_, const float *z_);
/* *yes = 1: the LR gradient + Push kernels derive a key's old weight from the (n, z) they load
 * anyway instead of reading it — the w an FTRL step leaves is a function of the n and z it
 * leaves (ftrl.h:66-73), so the stored w is that function's value, bit for bit, as long as every
 * row was written by steps under the table's present hyper-parameters (or never: 0, 0, 0).
 * An FTRL table of dim 1 starts that way; it stops (for good: the kernels read w again) when
 * xf_table_import brings a row whose w is not the w of its (n, z) — checked row by row on the
 * GPU, a model file saved under the same hyper-parameters passes — or xf_table_set_hyper
 * changes alpha / beta / lambda1 / lambda2 with keys in the table.  Same table bits either way. */
int xf_table_w_derived(xf_table *t, int *yes);

/* ---------------------------------------------------------------- model kernels       */
/* All pointers are device pointers; asynchronous on `stream`. */
/* LR forward (lr_worker.cc:121-143): loss[r] = sigmoid(sum w_u[uidx]) - label */
int xf_lr_forward_dev(const xf_dev_batch *b, const float *d_wu, float *d_loss,
                      float *d_pctr /* may be NULL */, void *stream);
/* LR gradient (lr_worker.cc:100-119): g[u] = (sum_{occurrences} loss[row]) / R */
int xf_lr_grad_dev(const xf_dev_batch *b, const float *d_loss, float *d_g, void *stream);
/* LR gradient fused with the Push for a table on the same GPU (single shard): g is still
 * written (parity hook); slots as returned by resolve/pull for b->ukeys; d_wu = the weights
 * that Pull returned, if nothing has touched those rows since (saves re-reading w), or NULL */
int xf_lr_grad_update_dev(xf_table *t, const xf_dev_batch *b, const uint32_t *d_slots,
                          const float *d_wu, const float *d_loss, float *d_g, void *stream);
/* FM forward, reference form (fm_worker.cc:159-202); v_u is U x k row-major */
int xf_fm_forward_dev(const xf_dev_batch *b, int k, const float *d_wu, const float *d_vu,
                      float *d_loss, float *d_pctr, float *d_vsum, void *stream);
/* FM gradient (fm_worker.cc:126-157): gw U, gv U x k */
int xf_fm_grad_dev(const xf_dev_batch *b, int k, const float *d_vu, const float *d_vsum,
                   const float *d_loss, float *d_gw, float *d_gv, void *stream);

/* FM gradient fused with the two Pushes (fm_worker.cc:241-242), both tables on this GPU */
int xf_fm_grad_update_dev(xf_table *w, xf_table *v, const xf_dev_batch *b,
                          const uint32_t *d_rows_w, const uint32_t *d_rows_v, const float *d_wu,
                          const float *d_vu, const float *d_vsum, const float *d_loss,
                          float *d_gw, float *d_gv, void *stream);

/* ---------------------------------------------------------------- fused steps         */
typedef struct xf_workspace xf_workspace; /* per-stream scratch: slots, w_u, g, loss... */
int xf_workspace_create(xf_workspace **out);
int xf_workspace_destroy(xf_workspace *ws);
/* One LRWorker::update (lr_worker.cc:167-176) on a single-shard table, device-resident:
 * pull(resolve+gather) -> loss -> gradient -> push(update).  Asynchronous. */
int xf_lr_step(xf_table *w, xf_batch *b, xf_workspace *ws, void *stream);
/* The whole LRWorker::update (lr_worker.cc:145-177) on a fresh minibatch: xf_batch_compile_local_dev
 * + xf_lr_step in one call — same kernels, same results; the key build's one host wait (work
 * items of the gradient pass, keys the table does not hold yet) is taken while the forward
 * already runs.  *out (optional): the minibatch, for replays (xf_batch_free); with out == NULL
 * the call waits for the step and frees it. */
int xf_lr_update_dev(xf_batch **out, xf_table *w, const uint64_t *d_keys, const uint32_t *d_rowptr,
                     const int32_t *d_labels, uint32_t R, uint32_t NNZ, int retain_keys,
                     xf_workspace *ws, void *stream);
/* One FMWorker::update (fm_worker.cc:226-242).  For k in {4, 8, 16, 32, 64} the forward's
 * per-key records (sum_k v, sum_k v^2, w) are kept in an array next to v's state rows and are
 * rewritten by the step's gradient + Push kernel; a minibatch that is stepped again starts with
 * the forward.  Other writers of w or v (xf_table_push / _update* / _import, a defrag, a step
 * with capture or a parity mode on) are noticed through the tables' write counters: the
 * minibatch's records are rebuilt on its next step.  Per-key intermediates (xf_workspace_fetch
 * of w_u / g) exist only with xf_workspace_capture.  Environment: XF_FM_TABLE_RECORDS=0 turns
 * the records off (the step then gathers the factor rows before every forward). */
int xf_fm_step(xf_table *w, xf_table *v, xf_batch *b, xf_workspace *ws, void *stream);
/* forward only (calculate_pctr, lr_worker.cc:25-71 / fm_worker.cc:25-95): pulls (and so
 * inserts) the batch's keys, writes R probabilities to host `pctr_out`.  Blocking. */
int xf_lr_predict(xf_table *w, xf_batch *b, xf_workspace *ws, float *pctr_out);
int xf_fm_predict(xf_table *w, xf_table *v, xf_batch *b, xf_workspace *ws,
                  float *pctr_out);
/* parity hook of the LR / FM step: when enabled, the step also stores the pulled weights and the
 * gradients per unique key of the minibatch for xf_workspace_fetch (the production step never
 * forms them: the forward reads the table in place, the gradient is consumed where it is
 * summed).  Needs a minibatch with a key list (xf_batch_compile*). */
int xf_workspace_capture(xf_workspace *ws, int enable);
/* Parity mode of the forward (xf_lr_step / xf_fm_step / xf_*_predict with this workspace):
 *   XF_PARITY_EXACT_SUMS       (default) row sums accumulated in fp64 and rounded once
 *   XF_PARITY_REFERENCE_ORDER  fp32 running sums in the reference's own order (ascending fid,
 *                              lr_worker.cc:127-138; FM: k-outer pooled sums, fm_worker.cc:
 *                              166-192): the loss equals the reference arithmetic's bit for
 *                              bit.  One thread per example: a checking mode, ~100x slower.
 *                              Needs a minibatch with a key list (xf_batch_compile*). */
#define XF_PARITY_EXACT_SUMS 0
#define XF_PARITY_REFERENCE_ORDER 1
int xf_workspace_parity(xf_workspace *ws, int mode);
/* copies of the last step's intermediates to host (parity hook): any pointer may be NULL */
int xf_workspace_fetch(xf_workspace *ws, float *wu, float *loss, float *g, size_t U,
                       size_t R);
/* optional per-kernel timing with HIP events recorded on the step's own stream:
 * ms_sum[5] = resolve, gather, forward, gradient, update, summed over *steps steps.
 * xf_lr_step fuses gather into resolve and update into gradient: slots 1 and 4 read 0. */
int xf_workspace_profile(xf_workspace *ws, int enable);
int xf_workspace_profile_read(xf_workspace *ws, double *ms_sum, long *steps);
int xf_stream_sync(void *stream);
/* measurement support: stream `bytes` of scratch with a known access width, `repeat` times
 * (kind 0/1/2 = read 4/8/16 B per lane, 3/4/5 = write 4/8/16 B per lane): calibrates the
 * rocprofv3 FETCH_SIZE / WRITE_SIZE counters (tools/pmc_traffic.py) */
int xf_calib_stream(int kind, size_t bytes, int repeat); /* == ps KVWorker::Wait */

/* ---------------------------------------------------------------- process group       */
/* One process per GPU.  Replaces ps-lite's postoffice / van as xflow uses them (main.cc:22-47,
 * scripts/local.sh:3-14) and the transport under KVWorker::Push/Pull: the exchange steps of
 * the sharded table are all-to-all-v's over RCCL (grouped ncclSend/ncclRecv per peer over
 * xGMI, asynchronous on the caller's stream).  The bootstrap is a TCP star around rank 0
 * (ps-lite's scheduler): it carries the ncclUniqueId and the small host-side collectives.
 * XF_TRANSPORT_HOST stages the exchange through the bootstrap sockets (tests: several ranks on
 * one GPU, or none). */
#define XF_TRANSPORT_RCCL 0
#define XF_TRANSPORT_HOST 1
/* RCCL when it comes up on every rank (library, communicator and one all-to-all-v on the device
 * are checked collectively), otherwise the host transport with a warning on rank 0's stderr;
 * xf_group_info reports which one the group runs on */
#define XF_TRANSPORT_AUTO 2
typedef struct xf_group xf_group;
/* rank < 0 / world <= 0 / addr NULL / port <= 0: from the environment — WORLD_SIZE | XF_WORLD |
 * DMLC_NUM_WORKER; RANK | XF_RANK | DMLC_RANK (absent: ranks are handed out in arrival order,
 * the process that binds the port first is rank 0, as ps-lite's scheduler numbers its nodes);
 * MASTER_ADDR | DMLC_PS_ROOT_URI (127.0.0.1); MASTER_PORT | DMLC_PS_ROOT_PORT (29512).
 * Collective: returns when all `world` processes have joined.  device: this process's GPU —
 * >= 0 a device index, -1 the current device, -2 by rank (LOCAL_RANK, else rank modulo the
 * visible devices; ranks handed out on arrival are only known inside this call). */
int xf_group_create(xf_group **out, int rank, int world, const char *addr, int port,
                    int transport, int device);
int xf_group_destroy(xf_group *g);
/* Give up on the group's communicators (ncclCommAbort): work of theirs stuck on a stream — a
 * peer has died — ends; device exchanges fail with XF_EIO afterwards.  The sharded trainer calls
 * it when a stream wait exceeds XF_COLLECTIVE_TIMEOUT_S. */
int xf_group_abort(xf_group *g);
int xf_group_info(const xf_group *g, int *rank, int *world, int *transport);
int xf_group_barrier(xf_group *g);
/* host-side collectives over the bootstrap (small payloads: counts, flags, metrics) */
int xf_group_allgather_host(xf_group *g, const void *in, size_t bytes, void *out /* world x */);
int xf_group_gatherv_host(xf_group *g, const void *in, size_t bytes, void *out_rank0,
                          const uint64_t *sizes_rank0);
/* all-to-all-v: send[...] holds one slice per destination rank (send_counts[p] elements of
 * elem_bytes, in rank order), recv gets one slice per source rank.  RCCL: device pointers,
 * asynchronous on `stream`.  Host transport: blocking; host_buffers != 0 says the pointers are
 * host memory (no GPU involved at all). */
int xf_group_alltoallv(xf_group *g, const void *send, const uint64_t *send_counts, void *recv,
                       const uint64_t *recv_counts, size_t elem_bytes, int host_buffers,
                       void *stream);
/* The same on one of the group's XF_GROUP_CHANNELS communicators (xf_group_alltoallv = channel
 * 0).  RCCL serialises the work of one communicator and needs every rank to enqueue on it in
 * the same order: exchanges that a rank issues from two streams independently of each other
 * (the sharded trainer's stale1 schedule) take a channel per stream.  Every rank must use the
 * same channel for the same exchange. */
#define XF_GROUP_CHANNELS 2
int xf_group_alltoallv_ch(xf_group *g, int channel, const void *send,
                          const uint64_t *send_counts, void *recv, const uint64_t *recv_counts,
                          size_t elem_bytes, int host_buffers, void *stream);

/* COLLECTIVE diagnostic: both channels driven at the same time from two streams of every rank
 * (grouped send / recv of `bytes` bytes with every rank, this one included), results checked. */
int xf_group_selftest(xf_group *g, size_t bytes);

/* The hash of the sources this library was built from (xflow_amd/build.py: source_hash); the
 * Python binding refuses a library whose hash differs from the sources next to it. */
const char *xf_source_hash(void);

/* The sort at the top of the reference's key build (lr_worker.cc:146-166: all_keys[(fid, sid)],
 * std::sort by fid) as a device routine: d_sorted_keys = d_keys[0..n) ascending, d_sorted_pos =
 * their positions, ascending inside a key (what a stable sort gives).  [lo, lo + span] = where
 * the keys lie (0, UINT64_MAX: anywhere); n < 2^30.  Hand-written (*by_hand = 1): a partition by
 * uniform key range + a range sorted in LDS; a range beyond the LDS (a power-law head, keys that
 * are no hashes) is merged from sorted parts, a skewed stream's hot keys get ranges of their own.
 * The library's radix sort (*by_hand = 0) beyond 3.3e7 keys and under xf_tune("key_build", 1).
 * Waits for the stream. */
int xf_sort_key_pos(const uint64_t *d_keys, uint32_t n, uint64_t lo, uint64_t span,
                    uint64_t *d_sorted_keys, uint32_t *d_sorted_pos, void *stream, int *by_hand);

/* Diagnostic (tools/kb_timeline.py): wall_clock64 stamps of the phases of the last keyed build
 * made with xf_tune("exp_knob", 200) (a library built with -DXF_EXPERIMENTS; empty otherwise):
 * [histogram | scatter | resolve] workgroups x slots;
 * returns the slots per workgroup, shape[3] = workgroups per kernel. */
int xf_kb_debug_read(unsigned long long *out, size_t cap, uint32_t *shape);

/* ---------------------------------------------------------------- sharded trainer     */
/* LRWorker / FMWorker::update across the ranks of a group: the table sharded by key range
 * (ps-lite's default slicer), examples by worker, one all-to-all-v each way per step (weights
 * back to the workers, gradients to the owners; the keys travel once, when the minibatch is
 * compiled).  The owner applies the ranks' pushes of a step in rank order after all pulls.
 * With a group of one rank (or g == NULL) it is the fused single-shard step. */
#define XF_SCHEDULE_SEQUENTIAL 0 /* Pull, compute, Push of step t before Pull(t+1) */
#define XF_SCHEDULE_STALE1 1     /* Push(t) overlaps Pull/forward/gradient of t+1 (one step stale) */
/* LR and FM (FM: three fp64 row sums per row instead of one; (loss, v_sum) back; with
 * XF_UPDATE_RANK_ORDERED the owner keeps one minibatch per worker and forms every worker's
 * gradient from the rows that worker's Pull returned, with XF_UPDATE_SUM_THEN_STEP it runs one
 * fused gradient + Push pass over all workers' rows).
 * Owner-compute dataflow: a minibatch's NONZEROS go to the key owners once, when it is
 * compiled; a step then runs the table-resident forward and gradient + Push at the owners and
 * exchanges only row-level scalars (fp64 partial row sums to the rows' workers, their losses
 * back) instead of a weight and a gradient per key.  Same results as XF_SCHEDULE_SEQUENTIAL
 * (every worker's gradient its own optimizer step, applied in rank order). */
#define XF_SCHEDULE_OWNER 2
/* The owner-compute dataflow with the gradient + Pushes of step t on a second HIP stream under
 * the exchanges of step t+1 (north_star: "overlapped with the next batch's forward on a second
 * HIP stream"): forward(t+1) reads the table BEFORE the Pushes of step t land, they are applied
 * while the row sums and losses of step t+1 travel, and forward(t+2) waits for them — weights
 * exactly one step stale, deterministic (events order the table's reader and writer), inside
 * ps-lite's asynchronous semantics; the results of XF_SCHEDULE_STALE1.  Same compiled
 * minibatches as XF_SCHEDULE_OWNER (xf_sharded_set_schedule switches between the two).
 * FM: under XF_UPDATE_RANK_ORDERED (every worker's gradient from the rows ITS Pull returned,
 * fm_worker.cc:226-242) the workers' two Pushes are what runs one step late; the fused pass of
 * XF_UPDATE_SUM_THEN_STEP reads the factors where they live at Push time and is refused here. */
#define XF_SCHEDULE_OWNER_STALE1 3
typedef struct {
  int32_t model;     /* 0 LR, 1 FM */
  int32_t optimizer; /* XF_OPT_* */
  int32_t k;         /* FM factors */
  int32_t schedule;  /* XF_SCHEDULE_* */
  uint64_t capacity; /* index positions of THIS rank's shard */
  uint64_t seed;
  float alpha, beta, lambda1, lambda2, lr;
  int32_t host_key_build; /* != 0: the sorted-unique-key build on the host (xf_batch_compile) */
  /* How the owner applies the ranks' pushes of one step:
   *   XF_UPDATE_RANK_ORDERED  every worker's gradient (its sum / its R) is its own optimizer
   *                           step, applied in rank order — one legal ps-lite interleaving
   *   XF_UPDATE_SUM_THEN_STEP the workers' sums added, ONE step with 1 / (all rows): the result
   *                           of one update() on the concatenation of the ranks' minibatches,
   *                           whatever the number of GPUs (XF_SCHEDULE_OWNER / _OWNER_STALE1
   *                           only: there the sums meet exactly, in fp64) */
  int32_t update_rule;
} xf_sharded_config;
#define XF_UPDATE_RANK_ORDERED 0
#define XF_UPDATE_SUM_THEN_STEP 1
void xf_sharded_config_default(xf_sharded_config *cfg);
typedef struct xf_sharded xf_sharded;
typedef struct xf_sbatch xf_sbatch;
int xf_sharded_create(xf_sharded **out, xf_group *g, const xf_sharded_config *cfg);
int xf_sharded_destroy(xf_sharded *st);
/* COLLECTIVE (every rank, same order): key build + the static part of the exchange */
int xf_sharded_compile(xf_sharded *st, xf_sbatch **out, const uint64_t *rowptr,
                       const uint64_t *keys, const int32_t *labels, size_t row_begin,
                       size_t row_end, int keep);
/* The same from DEVICE-resident arrays: the raw keys in CSR order, 32-bit row offsets (R + 1),
 * labels.  On the owner-compute dataflow the worker's side is a stable partition by key owner
 * on the device and ONE host wait (for the per-owner counts, which travel through the group's
 * all-to-all).  The arrays may be released when the call returns.  COLLECTIVE.  Replaces the
 * key build + the Pull's routing of lr_worker.cc:146-170 for a minibatch that is already in
 * HBM (nothing in the reference: its blocks live in host vectors). */
int xf_sharded_compile_dev(xf_sharded *st, xf_sbatch **out, const uint64_t *d_keys,
                           const uint32_t *d_rowptr, const int32_t *d_labels, uint32_t R,
                           uint32_t NNZ, int keep);
int xf_sbatch_free(xf_sbatch *b);
int xf_sbatch_dims(const xf_sbatch *b, uint32_t *R, uint32_t *NNZ, uint32_t *U,
                   uint64_t *n_owned /* keys of this minibatch (all ranks) this rank owns */);
/* 1 when the minibatch came from the FM build against the tables' settled tiers
 * (xf_batch_compile_fm: one shard, every key settled), else 0; < 0: error */
int xf_sbatch_fm_keyed(const xf_sbatch *b);
/* COLLECTIVE: one update() of every rank; asynchronous on the trainer's stream */
int xf_sharded_step(xf_sharded *st, xf_sbatch *b);
/* COLLECTIVE: forward only over this rank's rows (ranks without rows pass an empty minibatch) */
int xf_sharded_predict(xf_sharded *st, xf_sbatch *b, float *pctr_out);
int xf_sharded_flush(xf_sharded *st);  /* apply an outstanding stale1 Push; synchronises */
int xf_sharded_defrag(xf_sharded *st); /* local table maintenance (flushes first) */
int xf_sharded_check(xf_sharded *st);  /* flush + the tables' sticky errors */
int xf_sharded_set_schedule(xf_sharded *st, int schedule);
/* the parity mode of the forward (xf_workspace_parity) for a one-rank trainer whose minibatches
 * carry a key list (host_key_build, or FM).  Set it BEFORE compiling the minibatches it is to
 * apply to: an FM minibatch compiled in the default mode against the tables' settled tiers
 * (xf_sbatch_fm_keyed) has no index of its key list, and the reference-order kernels refuse it
 * (the call says so and names this remedy) */
int xf_sharded_set_parity(xf_sharded *st, int mode);
int xf_sharded_tables(xf_sharded *st, xf_table **w, xf_table **v);
int xf_sharded_stream(xf_sharded *st, void **stream);
/* ms_sum[6] = owner pull, weights exchange, forward, gradient, gradients exchange, owner
 * update (HIP events on the step's stream; sequential schedule) */
int xf_sharded_profile(xf_sharded *st, int enable);
int xf_sharded_profile_read(xf_sharded *st, double *ms_sum, long *steps);
/* Checkpoint of the sharded table (the reference never saves its model): every rank writes
 * <prefix>.shard-RRRRR-of-NNNNN (key-sorted (key, w, n, z) of its shard), rank 0 also
 * <prefix>.manifest.  Load reads the shards of ANY saved world size (or a single-GPU model
 * file saved by XFSaveModel) and keeps the keys this rank owns.  Both COLLECTIVE. */
int xf_sharded_save(xf_sharded *st, const char *prefix);
int xf_sharded_load(xf_sharded *st, const char *prefix);

/* ---------------------------------------------------------------- metrics             */
/* Base::calculate_auc (base.h:84-110): reference-format logloss (mean of y*log2 p +
 * (1-y)*log2(1-p), negative), AUC by descending-pctr rank sum, plus the conventional
 * natural-log logloss (positive).  acc_logloss_inout is the never-reset member. */
int xf_auc_logloss(const int32_t *labels, const float *pctr, size_t n,
                   float *acc_logloss_inout, float *auc, int *tp, int *fp,
                   double *nat_logloss);

/* ---------------------------------------------------------------- worker-level C API  */
/* Signatures of src/c_api/c_api.h:26-29 kept verbatim.  The handle owns an LR or FM
 * worker (XFSetParam "model") whose table lives on the current HIP device. */
int XFCreate(void **h, const char *train_path, const char *test_path);
int XFStartTrain(void **h);
/* additive (the reference has no equivalents) */
int XFDestroy(void **h);
/* names: model(0 LR,1 FM) epochs block_size_mb core_num k optimizer(ftrl|sgd) capacity
 *        rank pred_path alpha beta lambda1 lambda2 lr seed cache_batches key_build(gpu|host)
 *        update(rank_ordered|sum_then_step: with schedule=owner)
 *        parity(exact|reference_order: the forward's row sums in the reference's own fp32
 *        order — one worker, checking mode)
 *        model_in model_out (model file to load before / save after training)
 *        block_cache(0|1) block_cache_dir (binarized block cache of the text files)
 *        ingest(host|gpu: the text of a block tokenised and hashed on the GPU, xf_ingest_*;
 *        blocks that are not of the common shape go to the host parser; core_num 1, no block
 *        cache).  XFGetMetric: ... blocks_gpu blocks_host (how the blocks were parsed) */
int XFSetParam(void *h, const char *name, const char *value);
/* after XFStartTrain: logloss_ref, logloss_nat, auc, tp, fp, rows_trained, train_seconds,
 * examples_per_sec, keys */
int XFGetMetric(void *h, const char *name, double *value);
/* model file (the reference never saves its model): key-sorted (key, w, n, z) dumps of the
 * worker's tables; XFPredict scores the test file with the current tables, no training */
int XFSaveModel(void *h, const char *path);
int XFLoadModel(void *h, const char *path);
int XFPredict(void *h);
/* the worker's tables, for export/checkpoint (NULL v for LR) */
int XFGetTables(void *h, xf_table **w, xf_table **v);

#ifdef __cplusplus
}
#endif
#endif /* XFLOW_AMD_H_ */
