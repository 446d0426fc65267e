// xf_batch.cc — host-side minibatch key build and its device mirror.
//
// Replaces the key build at the top of LRWorker::update / FMWorker::update
// (src/model/lr/lr_worker.cc:146-166, src/model/fm/fm_worker.cc:205-225): flatten the
// slice to all_keys[(fid,sid)], sort by fid, sorted-unique key list.  The reference keeps
// `all_keys` as an array of 24-byte structs and walks it with merge-joins; here the same
// information is stored as the two index views the kernels stream:
//   CSR  rowptr/uidx   forward  (one wave per example gathers w_u[uidx])
//   COO  segptr/coo_row gradient (one lane per key walks its occurrences)
// The result depends only on the input rows, so a compiled batch can be cached across
// epochs (the reference re-parses and re-sorts every epoch, lr_worker.cc:184).
#include "xf_batch.h"
#include "xf_scratch.h"
#include "xf_tiling.h"

#include <atomic>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct KeyPos {
  uint64_t key;
  uint32_t pos;  // nnz position in row-major order
};

inline bool keypos_less(const KeyPos &a, const KeyPos &b) {
  return a.key < b.key || (a.key == b.key && a.pos < b.pos);
}

// Sort (key,pos) by key, ties by position.  Keys are uniform 64-bit hashes, so one
// counting pass on the top 16 bits leaves ~NNZ/65536 elements per bucket, which are
// finished with std::sort in parallel.
void sort_keypos(std::vector<KeyPos> &a) {
  const size_t n = a.size();
  if (n < (1u << 16)) {
    std::sort(a.begin(), a.end(), keypos_less);
    return;
  }
  constexpr int kBits = 16;
  constexpr size_t kBuckets = (size_t)1 << kBits;
  std::vector<size_t> start(kBuckets + 1, 0);
  for (size_t i = 0; i < n; ++i) ++start[(a[i].key >> (64 - kBits)) + 1];
  for (size_t b = 0; b < kBuckets; ++b) start[b + 1] += start[b];
  std::vector<KeyPos> tmp(n);
  {
    std::vector<size_t> cur(start.begin(), start.end() - 1);
    for (size_t i = 0; i < n; ++i) tmp[cur[a[i].key >> (64 - kBits)]++] = a[i];
  }
  a.swap(tmp);
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > 32) nt = 32;
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) {
    th.emplace_back([&, t]() {
      for (size_t b = t; b < kBuckets; b += nt)
        std::sort(a.begin() + start[b], a.begin() + start[b + 1], keypos_less);
    });
  }
  for (auto &x : th) x.join();
}

double g_panel_slice_bytes = 1.5 * 1024 * 1024;
double g_min_panel_nnz = 4e6;

// Panel-major copy of the CSR for the LR forward: the uidx space [0,U) is cut into P equal
// ranges (P a multiple of 8 = XCDs) so that one range of w_u (U/P floats) fits an XCD's
// 4 MiB L2 beside the index stream; within a (panel,row) cell the CSR order is kept.
void build_panels(xf_batch *b) {
  b->P = 0;
  b->pptr.clear();
  b->pidx.clear();
  b->ftile_ptr.clear();
  b->fpanel_first.clear();
  b->fwd_grid = 0;
  const uint32_t P = xf::panel_count(b->U, b->NNZ, g_panel_slice_bytes, g_min_panel_nnz);
  if (P == 0) return;
  const uint64_t R = b->R;
  const uint32_t U = b->U;
  b->pptr.assign((size_t)P * (R + 1), 0);
  // counts per (panel,row) -> exclusive prefix in panel-major order
  for (uint64_t r = 0; r < R; ++r)
    for (uint32_t j = b->rowptr[r]; j < b->rowptr[r + 1]; ++j)
      ++b->pptr[(size_t)xf::panel_of(b->uidx[j], P, U) * (R + 1) + r + 1];
  uint32_t run = 0;
  for (uint32_t p = 0; p < P; ++p) {
    uint32_t *pp = &b->pptr[(size_t)p * (R + 1)];
    pp[0] = run;
    for (uint64_t r = 0; r < R; ++r) {
      const uint32_t c = pp[r + 1];
      pp[r + 1] = pp[r] + c;
    }
    run = pp[R];
  }
  b->pidx.resize(b->NNZ);
  std::vector<uint32_t> cur((size_t)P * R);
  for (uint32_t p = 0; p < P; ++p)
    for (uint64_t r = 0; r < R; ++r) cur[(size_t)p * R + r] = b->pptr[(size_t)p * (R + 1) + r];
  for (uint64_t r = 0; r < R; ++r)
    for (uint32_t j = b->rowptr[r]; j < b->rowptr[r + 1]; ++j) {
      const uint32_t ui = b->uidx[j];
      b->pidx[cur[(size_t)xf::panel_of(ui, P, U) * R + r]++] = ui;
    }
  b->P = P;
  // forward tiles (xf_tiling.h): tile t = cells [ftile_ptr[t], ftile_ptr[t+1]) of one panel;
  // the last tile of a panel also covers the empty cell p*(R+1)+R between two panels
  b->fpanel_first.assign(P + 1, 0);
  for (uint32_t p = 0; p < P; ++p) {
    b->fpanel_first[p] = (uint32_t)b->ftile_ptr.size();
    const uint32_t s0 = (uint32_t)((size_t)p * (R + 1));
    const uint32_t *pp = &b->pptr[s0];
    for (uint32_t r = 0; r < R; ++r)
      if (xf::fwd_tile_starts_at(pp, r)) b->ftile_ptr.push_back(s0 + r);
  }
  b->fpanel_first[P] = (uint32_t)b->ftile_ptr.size();
  b->ftile_ptr.push_back((uint32_t)((size_t)(P - 1) * (R + 1) + R));  // end of the last panel
  // workgroup b serves XCD b % 8: it takes the (b / 8)-th tile of the panels p % 8 == b % 8
  uint32_t longest = 0;
  for (uint32_t x = 0; x < 8; ++x) {
    uint32_t len = 0;
    for (uint32_t p = x; p < P; p += 8) len += b->fpanel_first[p + 1] - b->fpanel_first[p];
    longest = std::max(longest, len);
  }
  b->fwd_grid = 8 * longest;
}

// Gradient tiles (xf_tiling.h): consecutive key ranges whose occurrences fit one workgroup's LDS
void build_tiles(xf_batch *b) {
  b->tile_ptr.clear();
  for (uint32_t u = 0; u < b->U; ++u)
    if (xf::grad_tile_starts_at(b->segptr.data(), u)) b->tile_ptr.push_back(u);
  b->tile_ptr.push_back(b->U);
}

}  // namespace

namespace xf {
double panel_slice_bytes() { return g_panel_slice_bytes; }
double min_panel_nnz() { return g_min_panel_nnz; }
}  // namespace xf

extern "C" int xf_batch_download(xf_batch *b);

static int need_host(const xf_batch *b) {
  if (b && b->on_device_only) return xf_batch_download(const_cast<xf_batch *>(b));
  return XF_OK;
}

namespace xf {
namespace {
struct Blob {
  void *p;
  size_t bytes;
  int dev;
  bool reserved;  // xf_batch_pool_reserve: serves ONE request of an eighth of its size up to its size
};
std::mutex g_pool_mu;
std::vector<Blob> g_pool;
int g_pool_limit = 64;
}  // namespace

int blob_alloc(void **p, size_t bytes, size_t *got) {
  int dev = 0;
  XF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    size_t best = g_pool.size();
    for (size_t i = 0; i < g_pool.size(); ++i) {
      const Blob &c = g_pool[i];
      if (c.dev != dev || c.bytes < bytes || c.bytes > bytes + bytes / 4 + (1u << 20)) continue;
      if (best == g_pool.size() || c.bytes < g_pool[best].bytes) best = i;
    }
    if (best == g_pool.size())  // a blob set aside for a run's first minibatch
      for (size_t i = 0; i < g_pool.size(); ++i)
        if (g_pool[i].reserved && g_pool[i].dev == dev && g_pool[i].bytes >= bytes &&
            bytes >= g_pool[i].bytes / 8 &&  // (not for a minibatch's small arrays)
            (best == g_pool.size() || g_pool[i].bytes < g_pool[best].bytes))
          best = i;
    if (best != g_pool.size()) {
      *p = g_pool[best].p;
      *got = g_pool[best].bytes;
      g_pool.erase(g_pool.begin() + best);
      return XF_OK;
    }
  }
  XF_HIP(hipMalloc(p, bytes));
  *got = bytes;
  return XF_OK;
}

void blob_free(void *p, size_t bytes) {
  if (!p) return;
  int dev = 0;
  void *evict = nullptr;
  if (hipGetDevice(&dev) == hipSuccess) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    if (g_pool_limit > 0) {
      if ((int)g_pool.size() >= g_pool_limit) {  // drop the oldest
        evict = g_pool.front().p;
        g_pool.erase(g_pool.begin());
      }
      g_pool.push_back({p, bytes, dev, false});
      p = nullptr;
    }
  }
  if (evict) (void)hipFree(evict);
  if (p) (void)hipFree(p);
}

void blob_pool_limit(int blobs) {
  std::vector<Blob> drop;
  {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_limit = blobs < 0 ? 0 : blobs;
    while ((int)g_pool.size() > g_pool_limit) {
      drop.push_back(g_pool.back());
      g_pool.pop_back();
    }
  }
  for (const Blob &b : drop) (void)hipFree(b.p);
}
}  // namespace xf

namespace xf {
static int g_path[kPathCount] = {0, 0, 0, 0};
int path_switch(int which) { return g_path[which]; }
void set_path_switch(int which, int v) { g_path[which] = v; }
#ifdef XF_EXPERIMENTS
static int g_exp_knob = 0;
int exp_knob() { return g_exp_knob; }
void set_exp_knob(int v) { g_exp_knob = v; }
#endif
static std::atomic<bool> g_device_poisoned{false};
bool device_poisoned() { return g_device_poisoned.load(std::memory_order_relaxed); }
void scratch_poison() { g_device_poisoned.store(true); }
}  // namespace xf

// a device allocation set aside in the pool for a run's first cells (include/xflow_amd.h): the
// driver maps tens of MB in ~0.25 ms when asked under the first build (profiles/r06/)
extern "C" int xf_batch_pool_reserve(size_t bytes) {
  XF_REQUIRE(bytes > 0, "xf_batch_pool_reserve: no bytes");
  int dev = 0;
  XF_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lk(xf::g_pool_mu);
    if (xf::g_pool_limit <= 0) return XF_OK;  // (no pooling: every blob goes back to the driver)
    for (const xf::Blob &b : xf::g_pool)
      if (b.reserved && b.dev == dev && b.bytes >= bytes) return XF_OK;
  }
  void *p = nullptr;
  XF_HIP(hipMalloc(&p, bytes));
  std::lock_guard<std::mutex> lk(xf::g_pool_mu);
  xf::g_pool.push_back({p, bytes, dev, true});
  return XF_OK;
}

// the calling thread's builder arena (xf_scratch.h) sized ahead of its first build: a run's first
// minibatches otherwise grow it build by build — a hipFree and a hipMalloc of ~0.5 GB, ~0.8 ms,
// each time a nested builder asked for more than the last one had (the gaps in the timeline of a
// table's first minibatches, profiles/r06/)
extern "C" int xf_scratch_reserve(size_t bytes) {
  xf::Arena &a = xf::arena();
  XF_REQUIRE(a.used == 0, "xf_scratch_reserve: a build is in progress on this thread");
  if (bytes <= a.cap) return XF_OK;
  if (a.base) XF_HIP(hipFree(a.base));
  a.base = nullptr;
  a.cap = 0;
  XF_HIP(hipMalloc((void **)&a.base, bytes));
  a.cap = bytes;
  return XF_OK;
}

extern "C" int xf_tune(const char *name, double value) {
  XF_REQUIRE(name, "xf_tune: null name");
  if (!strcmp(name, "panel_slice_bytes")) g_panel_slice_bytes = value;
  else if (!strcmp(name, "min_panel_nnz")) g_min_panel_nnz = value;
  else if (!strcmp(name, "parse_threads")) xf::set_parse_threads((int)value);
  else if (!strcmp(name, "key_build") && value >= 0 && value <= 3)
    xf::set_path_switch(xf::kPathKeyBuild, (int)value);
  else if (!strcmp(name, "old_weight") && value >= 0 && value <= 2)
    xf::set_path_switch(xf::kPathOldWeight, (int)value);
  else if (!strcmp(name, "lr_gradient") && value >= 0 && value <= 3)
    xf::set_path_switch(xf::kPathLrGradient, (int)value);
  else if (!strcmp(name, "owner_pass") && value >= 0 && value <= 4)
    xf::set_path_switch(xf::kPathOwnerPass, (int)value);
#ifdef XF_EXPERIMENTS
  else if (!strcmp(name, "exp_knob")) xf::set_exp_knob((int)value);
#endif
  else if (!strcmp(name, "batch_pool_blobs")) xf::blob_pool_limit((int)value);
  else
    return xf::set_error(XF_EINVAL, "xf_tune: unknown knob '%s' (or a value it does not take)", name);
  return XF_OK;
}

extern "C" int xf_batch_heavy_chunks(const xf_batch *b, uint32_t *H, const uint32_t **chunk_ptr) {
  XF_REQUIRE(b && H, "xf_batch_heavy_chunks: null argument");
  XF_TRY(need_host(b));
  *H = b->H;
  if (chunk_ptr) *chunk_ptr = b->hchunk_ptr.data();
  return XF_OK;
}

extern "C" int xf_batch_tiles(const xf_batch *b, uint32_t *ntiles, const uint32_t **tile_ptr) {
  XF_REQUIRE(b && ntiles, "xf_batch_tiles: null argument");
  XF_TRY(need_host(b));
  *ntiles = (uint32_t)(b->tile_ptr.size() - 1);
  if (tile_ptr) *tile_ptr = b->tile_ptr.data();
  return XF_OK;
}

extern "C" int xf_batch_fwd_tiles(const xf_batch *b, uint32_t *ntiles, const uint32_t **tile_ptr,
                                  const uint32_t **panel_first, uint32_t *grid) {
  XF_REQUIRE(b && ntiles, "xf_batch_fwd_tiles: null argument");
  XF_TRY(need_host(b));
  *ntiles = b->P ? (uint32_t)b->ftile_ptr.size() - 1 : 0;
  if (tile_ptr) *tile_ptr = b->ftile_ptr.data();
  if (panel_first) *panel_first = b->fpanel_first.data();
  if (grid) *grid = b->fwd_grid;
  return XF_OK;
}

extern "C" int xf_batch_panels(const xf_batch *b, uint32_t *P, const uint32_t **pptr,
                               const uint32_t **pidx) {
  XF_REQUIRE(b && P, "xf_batch_panels: null argument");
  XF_TRY(need_host(b));
  *P = b->P;
  if (pptr) *pptr = b->pptr.data();
  if (pidx) *pidx = b->pidx.data();
  return XF_OK;
}

extern "C" int xf_batch_compile(xf_batch **out, const uint64_t *rowptr, const uint64_t *keys,
                                const int32_t *labels, size_t row_begin, size_t row_end) {
  XF_REQUIRE(out && rowptr && labels && row_end >= row_begin, "xf_batch_compile: bad argument");
  const size_t R = row_end - row_begin;
  const uint64_t base = rowptr[row_begin];
  const size_t NNZ = (size_t)(rowptr[row_end] - base);
  XF_REQUIRE(NNZ == 0 || keys, "xf_batch_compile: null keys");
  XF_REQUIRE(R < 0xFFFFFFFFull && NNZ < 0xFFFFFFFFull, "xf_batch_compile: batch too large");
  xf_batch *b = new xf_batch;
  b->R = (uint32_t)R;
  b->NNZ = (uint32_t)NNZ;
  b->rowptr.resize(R + 1);
  b->labels.assign(labels + row_begin, labels + row_end);
  std::vector<uint32_t> row_of(NNZ);
  for (size_t r = 0; r < R; ++r) {
    b->rowptr[r] = (uint32_t)(rowptr[row_begin + r] - base);
    for (uint64_t j = rowptr[row_begin + r]; j < rowptr[row_begin + r + 1]; ++j)
      row_of[j - base] = (uint32_t)r;
  }
  b->rowptr[R] = (uint32_t)NNZ;
  std::vector<KeyPos> kp(NNZ);
  for (size_t j = 0; j < NNZ; ++j) {
    kp[j].key = keys[base + j];
    kp[j].pos = (uint32_t)j;
  }
  sort_keypos(kp);
  b->uidx.resize(NNZ);
  b->coo_row.resize(NNZ);
  b->segptr.clear();
  b->ukeys.clear();
  for (size_t j = 0; j < NNZ; ++j) {
    if (j == 0 || kp[j].key != kp[j - 1].key) {
      b->segptr.push_back((uint32_t)j);
      b->ukeys.push_back(kp[j].key);
    }
    b->uidx[kp[j].pos] = (uint32_t)(b->ukeys.size() - 1);
    b->coo_row[j] = row_of[kp[j].pos];
  }
  b->segptr.push_back((uint32_t)NNZ);
  b->U = (uint32_t)b->ukeys.size();
  for (uint32_t u = 0; u < b->U; ++u)
    if (b->segptr[u + 1] - b->segptr[u] > XF_HEAVY_SEG) b->heavy.push_back(u);
  b->H = (uint32_t)b->heavy.size();
  b->hchunk_ptr.assign(1, 0);
  for (uint32_t u : b->heavy)
    b->hchunk_ptr.push_back(b->hchunk_ptr.back() +
                            (b->segptr[u + 1] - b->segptr[u] + XF_TILE_NNZ - 1) / XF_TILE_NNZ);
  build_panels(b);
  build_tiles(b);
  *out = b;
  return XF_OK;
}

namespace xf {
void cells_free(xf_cells *c);
}

extern "C" int xf_batch_free(xf_batch *b) {
  if (!b) return XF_OK;
  if (b->d_blob || b->cells || b->d_raw || b->d_rows_u || b->d_fm_rows[0] || b->d_uidx_sorted ||
      b->d_ref_coo) {
    // kernels still running on the batch must finish first (hipFree used to imply that)
    (void)hipDeviceSynchronize();
  }
  if (b->d_blob) xf::blob_free(b->d_blob, b->d_blob_bytes);
  if (b->d_blob2) xf::blob_free(b->d_blob2, b->d_blob2_bytes);
  if (b->cells) xf::cells_free(b->cells);
  if (b->d_raw) xf::blob_free(b->d_raw, b->d_raw_bytes);
  if (b->d_rows_u) (void)hipFree(b->d_rows_u);
  if (b->d_uidx_sorted) (void)hipFree(b->d_uidx_sorted);
  if (b->d_ref_coo) (void)hipFree(b->d_ref_coo);
  for (int i = 0; i < 2; ++i)
    if (b->d_fm_rows[i]) xf::blob_free(b->d_fm_rows[i], b->fm_rows_bytes[i]);
  if (b->d_fm_ridx) xf::blob_free(b->d_fm_ridx, b->fm_ridx_bytes);
  delete b;
  return XF_OK;
}

extern "C" int xf_batch_dims(const xf_batch *b, uint32_t *R, uint32_t *NNZ, uint32_t *U,
                             uint32_t *H) {
  XF_REQUIRE(b, "xf_batch_dims: null batch");
  if (R) *R = b->R;
  if (NNZ) *NNZ = b->NNZ;
  if (U) *U = b->U;
  if (H) *H = b->H;
  return XF_OK;
}

extern "C" int xf_batch_host(const xf_batch *b, const uint64_t **ukeys, const uint32_t **rowptr,
                             const uint32_t **uidx, const uint32_t **segptr,
                             const uint32_t **coo_row, const int32_t **labels,
                             const uint32_t **heavy) {
  XF_REQUIRE(b, "xf_batch_host: null batch");
  XF_TRY(need_host(b));
  if (ukeys) *ukeys = b->ukeys.data();
  if (rowptr) *rowptr = b->rowptr.data();
  if (uidx) *uidx = b->uidx.data();
  if (segptr) *segptr = b->segptr.data();
  if (coo_row) *coo_row = b->coo_row.data();
  if (labels) *labels = b->labels.data();
  if (heavy) *heavy = b->heavy.data();
  return XF_OK;
}

extern "C" int xf_batch_upload(xf_batch *b, void *stream) {
  XF_REQUIRE(b, "xf_batch_upload: null batch");
  if (b->d_blob) return XF_OK;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_ukeys = 0;
  const size_t o_rowptr = o_ukeys + al((size_t)b->U * 8);
  const size_t o_uidx = o_rowptr + al(((size_t)b->R + 1) * 4);
  const size_t o_segptr = o_uidx + al((size_t)b->NNZ * 4);
  const size_t o_coo = o_segptr + al(((size_t)b->U + 1) * 4);
  const size_t o_labels = o_coo + al((size_t)b->NNZ * 4);
  const size_t o_heavy = o_labels + al((size_t)b->R * 4);
  const size_t o_pptr = o_heavy + al((size_t)b->H * 4);
  const size_t o_pidx = o_pptr + al((size_t)b->P * ((size_t)b->R + 1) * 4);
  const size_t o_scr = o_pidx + al(b->P ? (size_t)b->NNZ * 4 : 0);
  const size_t o_tile = o_scr + al((size_t)b->P * b->R * 8);
  const size_t o_ftile = o_tile + al(b->tile_ptr.size() * 4);
  const size_t o_forder = o_ftile + al(b->ftile_ptr.size() * 4);
  const size_t o_hch = o_forder + al(b->fpanel_first.size() * 4);
  const size_t n_hch = b->H ? b->hchunk_ptr.back() : 0;
  const size_t o_hscr = o_hch + al(b->H ? ((size_t)b->H + 1) * 4 : 0);
  const size_t total = o_hscr + al(n_hch * (1 + XF_HEAVY_KMAX) * 8) + 256;
  char *d = nullptr;
  XF_TRY(xf::blob_alloc((void **)&d, total, &b->d_blob_bytes));
  hipStream_t s = (hipStream_t)stream;
  auto put = [&](size_t off, const void *src, size_t bytes) -> hipError_t {
    if (bytes == 0) return hipSuccess;
    return hipMemcpyAsync(d + off, src, bytes, hipMemcpyHostToDevice, s);
  };
  XF_HIP(put(o_ukeys, b->ukeys.data(), (size_t)b->U * 8));
  XF_HIP(put(o_rowptr, b->rowptr.data(), ((size_t)b->R + 1) * 4));
  XF_HIP(put(o_uidx, b->uidx.data(), (size_t)b->NNZ * 4));
  XF_HIP(put(o_segptr, b->segptr.data(), ((size_t)b->U + 1) * 4));
  XF_HIP(put(o_coo, b->coo_row.data(), (size_t)b->NNZ * 4));
  XF_HIP(put(o_labels, b->labels.data(), (size_t)b->R * 4));
  XF_HIP(put(o_heavy, b->heavy.data(), (size_t)b->H * 4));
  if (b->P) {
    XF_HIP(put(o_pptr, b->pptr.data(), b->pptr.size() * 4));
    XF_HIP(put(o_pidx, b->pidx.data(), b->pidx.size() * 4));
  }
  XF_HIP(put(o_tile, b->tile_ptr.data(), b->tile_ptr.size() * 4));
  if (b->H) XF_HIP(put(o_hch, b->hchunk_ptr.data(), b->hchunk_ptr.size() * 4));
  if (b->P) {
    XF_HIP(put(o_ftile, b->ftile_ptr.data(), b->ftile_ptr.size() * 4));
    XF_HIP(put(o_forder, b->fpanel_first.data(), b->fpanel_first.size() * 4));
  }
  XF_HIP(hipStreamSynchronize(s));  // host vectors are pageable: finish before returning
  b->d_blob = d;
  b->view.R = b->R;
  b->view.NNZ = b->NNZ;
  b->view.U = b->U;
  b->view.H = b->H;
  b->view.ukeys = (const uint64_t *)(d + o_ukeys);
  b->view.rowptr = (const uint32_t *)(d + o_rowptr);
  b->view.uidx = (const uint32_t *)(d + o_uidx);
  b->view.segptr = (const uint32_t *)(d + o_segptr);
  b->view.coo_row = (const uint32_t *)(d + o_coo);
  b->view.labels = (const int32_t *)(d + o_labels);
  b->view.heavy = b->H ? (const uint32_t *)(d + o_heavy) : nullptr;
  b->view.P = b->P;
  b->view.fwd_ntiles = b->P ? (uint32_t)b->ftile_ptr.size() - 1 : 0;
  b->view.fwd_tile_ptr = b->P ? (const uint32_t *)(d + o_ftile) : nullptr;
  b->view.fwd_panel_first = b->P ? (const uint32_t *)(d + o_forder) : nullptr;
  b->view.fwd_grid = b->fwd_grid;
  b->view.pptr = b->P ? (const uint32_t *)(d + o_pptr) : nullptr;
  b->view.pidx = b->P ? (const uint32_t *)(d + o_pidx) : nullptr;
  b->view.fwd_scratch = b->P ? (double *)(d + o_scr) : nullptr;
  b->view.ntiles = (uint32_t)(b->tile_ptr.size() - 1);
  b->view.n_heavy_chunks = (uint32_t)n_hch;
  b->view.heavy_chunk_ptr = b->H ? (const uint32_t *)(d + o_hch) : nullptr;
  b->view.heavy_scratch = b->H ? (double *)(d + o_hscr) : nullptr;
  b->view.tile_ptr = (const uint32_t *)(d + o_tile);
  return XF_OK;
}

extern "C" int xf_batch_dev_view(const xf_batch *b, xf_dev_batch *view) {
  XF_REQUIRE(b && view, "xf_batch_dev_view: null argument");
  XF_REQUIRE(b->d_blob, "xf_batch_dev_view: batch not uploaded");
  *view = b->view;
  return XF_OK;
}
