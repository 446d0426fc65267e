// xf_worker.cc — LRWorker / FMWorker and the XFCreate / XFStartTrain C API.
//
// Mirrors the reference's worker surface (same public members and call order):
//   xflow::LRWorker  src/model/lr/lr_worker.{h,cc}  (ctor(train,test), epochs, train(),
//                    batch_training(), update(), predict(), calculate_pctr())
//   xflow::FMWorker  src/model/fm/fm_worker.{h,cc}
//   xflow::Server    src/model/server.h:22-31        (which optimizer serves w and v)
//   XFCreate/XFStartTrain  src/c_api/c_api.{h,cc}
// but every Pull / Push / loss / gradient runs on the GPU through the extern "C" shim
// (xf_table_*, xf_lr_step, xf_fm_step).  What stays on the host is what north_star keeps
// there: the libsvm-format block reader and the per-block key build.
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include "xf_worker.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <fstream>
#include <initializer_list>
#include <iostream>

namespace xf {
int model_write(const char *path, xf_table *tw, xf_table *tv, int k);
int model_read(const char *path, xf_table *tw, xf_table *tv, int k, uint32_t shard,
               uint32_t nshards);
}  // namespace xf

namespace xflow_amd {

static double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

Worker::Worker(int model, const char *train_file, const char *test_file)
    : model_(model), train_file_path(train_file ? train_file : ""),
      test_file_path(test_file ? test_file : "") {}

Worker::~Worker() {
  for (std::thread &t : closers_) t.join();
  for (xf_block *b : blocks_)
    if (b) xf_block_destroy(b);
  for (xf_ingest *g : ingest_)
    if (g) xf_ingest_destroy(g);
  for (xf_sbatch *b : cache_) xf_sbatch_free(b);
  if (sharded_) xf_sharded_destroy(sharded_);  // owns the tables
  if (group_) xf_group_destroy(group_);
}

static const char *env_first(std::initializer_list<const char *> names) {
  for (const char *n : names) {
    const char *v = getenv(n);
    if (v && *v) return v;
  }
  return nullptr;
}

// xflow::Server (server.h:22-31): app 0 serves w, app 1 serves v; FTRL by default, the SGD
// handlers are the commented-out alternative (server.h:25,29).  ps::Start / MyRank
// (main.cc:22-47): with more than one worker (world=, or WORLD_SIZE / DMLC_NUM_WORKER in the
// environment, as scripts/local.sh sets it) this process joins the group — one process per
// GPU — and its tables are one key-range shard of the parameter table.
int Worker::create_tables() {
  if (sharded_) return XF_OK;
  int w = world;
  if (w <= 0) {
    const char *v = env_first({"WORLD_SIZE", "XF_WORLD", "DMLC_NUM_WORKER"});
    w = v ? atoi(v) : 1;
  }
  if (w > 1) {
    XF_TRY(xf_group_create(&group_, rank_given_ ? rank : -1, w, nullptr, 0,
                           transport, -2));
    int r = 0;
    XF_TRY(xf_group_info(group_, &r, &w, nullptr));
    rank = r;  // ps::MyRank(): selects <prefix>-%05d (lr_worker.cc:208-210)
  }
  world = w;
  xf_sharded_config c;
  xf_sharded_config_default(&c);
  c.model = model_;
  c.optimizer = optimizer;
  c.k = v_dim_;
  c.schedule = schedule;
  c.capacity = capacity;
  c.seed = seed;
  c.alpha = alpha;
  c.beta = beta;
  c.lambda1 = lambda1;
  c.lambda2 = lambda2;
  c.lr = learning_rate;
  c.host_key_build = key_build_gpu && parity == XF_PARITY_EXACT_SUMS ? 0 : 1;
  c.update_rule = update_rule;
  XF_TRY(xf_sharded_create(&sharded_, group_, &c));
  if (parity != XF_PARITY_EXACT_SUMS) XF_TRY(xf_sharded_set_parity(sharded_, parity));
  XF_TRY(xf_sharded_tables(sharded_, &table_w_, &table_v_));
  // One update() and one predict of a two-row minibatch on a private single-shard trainer: the
  // first launch of a kernel loads its code object and the builders size their scratch on first
  // use (~45 ms together) — start-up work, done here instead of inside the first block of the
  // training loop.  The real tables are not touched.
  {
    xf_sharded_config wc = c;
    wc.capacity = 1024;
    xf_sharded *warm = nullptr;
    XF_TRY(xf_sharded_create(&warm, nullptr, &wc));
    const uint64_t rp[3] = {0, 2, 4}, keys[4] = {11, 12, 12, 13};
    const int32_t lab[2] = {0, 1};
    float p[2];
    xf_sbatch *b = nullptr;
    int rc = xf_sharded_compile(warm, &b, rp, keys, lab, 0, 2, 1);
    if (rc == XF_OK) rc = xf_sharded_step(warm, b);
    if (rc == XF_OK) rc = xf_sharded_predict(warm, b, p);
    if (b) xf_sbatch_free(b);
    xf_sharded_destroy(warm);
    XF_TRY(rc);
  }
  return XF_OK;
}

// ingest = gpu: the two staging buffers (pinning 2 x block_size of host memory costs tens of ms)
// and the tokeniser's first launches: start-up work like the two-row update of create_tables
int Worker::start_ingest() {
  if (!gpu_ingest_applies()) return XF_OK;  // (nothing is pinned for a path that will not run)
  for (int i = 0; i < 2; ++i)
    if (!ingest_[i] && xf_ingest_create(&ingest_[i], (size_t)block_size << 20) != XF_OK) {
      // (a block size the tokeniser does not take, or no memory to pin: the host parser's run)
      fprintf(stderr, "xflow_amd: ingest=gpu is not available (%s): the host parser reads the text\n",
              xf_last_error());
      ingest_gpu = false;
      return XF_OK;
    }
  static const char two_rows[] = "0\t1:2:0.5 3:4:1\n1\t5:6:0.25\n";
  void *stream = nullptr;
  XF_TRY(xf_sharded_stream(sharded_, &stream));
  uint32_t R = 0, NNZ = 0;
  int ok = 0;
  return xf_ingest_block(ingest_[0], two_rows, sizeof(two_rows) - 1, stream, nullptr, nullptr,
                         nullptr, &R, &NNZ, &ok);
}

// percent: settle the table when more than that share of its keys has arrived since the last
// time (5 at the epoch boundaries; inside the first epoch, one worker: 30 — the key build of
// the blocks still to come finds settled keys where they sit in LDS, arrival keys by a probe
// per nonzero, and a defrag is a sort of the table's keys)
int Worker::defrag_if_grown(int percent) {
  uint64_t n = 0, settled = 0;
  XF_TRY(xf_table_size(table_w_, &n));
  // (the settled tier as the table reports it: the first minibatch's build may have settled the
  // table already)
  XF_TRY(xf_table_settled(table_w_, &settled));
  settled = std::min(n, settled);
  const uint64_t arrived = n - settled, fresh = n - std::min(n, keys_seen_);
  keys_seen_ = n;
  // grown by `percent` since the table was last settled — or (inside an epoch) the inflow has
  // stopped: the last minibatch brought less than 0.1 % new keys while more than 0.5 % of the
  // table's keys still sit in the arrival index, where every minibatch that meets them pays the
  // first-touch path for them (measured, DESIGN 3: a defrag of 1e7 keys 1.1 ms, the path ~0.15 ms
  // per minibatch at 14 % unsettled keys)
  const bool grown = n > settled + settled / 100 * percent + (percent > 5 ? 4096 : 0);
  const bool settled_down = percent > 5 && arrived * 200 > n && fresh * 1000 < n && arrived > 4096;
  if (grown || settled_down) XF_TRY(xf_sharded_defrag(sharded_));
  return XF_OK;
}

// "does any rank still have a block?" — the ranks' files differ in length, the steps are
// collective: a rank that has run out keeps stepping empty minibatches until all are done
int Worker::any_rank(bool mine, bool *any) {
  *any = mine;
  if (world <= 1) return XF_OK;
  const int32_t flag = mine ? 1 : 0;
  std::vector<int32_t> all(world);
  XF_TRY(xf_group_allgather_host(group_, &flag, 4, all.data()));
  for (int32_t f : all)
    if (f) *any = true;
  return XF_OK;
}

int Worker::open_reader(xf_reader **rd, const char *path, size_t cap, bool *writes_cache) {
  if (writes_cache) *writes_cache = false;
  if (!block_cache) return xf_reader_open(rd, path, cap);
  std::string base = path;
  if (!block_cache_dir.empty()) {
    const size_t slash = base.find_last_of('/');
    base = block_cache_dir + "/" + (slash == std::string::npos ? base : base.substr(slash + 1));
  }
  const std::string cpath = base + ".xfcsr" + std::to_string(cap);
  int from_cache = 0;
  XF_TRY(xf_reader_open_cached(rd, path, cap, cpath.c_str(), &from_cache));
  if (writes_cache) *writes_cache = !from_cache;
  return XF_OK;
}

// batch_training (lr_worker.cc:179-205, fm_worker.cc:247-275)
int Worker::batch_training() {
  // init push of key 0 with a zero gradient (lr_worker.cc:180-182, fm_worker.cc:248-252): on
  // the shard that owns key 0
  if (xf_shard_of(0, (uint32_t)world) == (uint32_t)rank) {
    const uint64_t key0 = 0;
    const float zero = 0.0f;
    XF_TRY(xf_table_push(table_w_, &key0, 1, &zero));
    if (model_ == 1) {
      std::vector<float> zv(v_dim_, 0.0f);
      XF_TRY(xf_table_push(table_v_, &key0, 1, zv.data()));
    }
  }
  // the builders' scratch arena for a block's worth of nonzeros (a token is at least four bytes of
  // text), sized before the clock starts instead of growing over the first blocks
  XF_TRY(xf_scratch_reserve(((size_t)block_size << 20) / 4 * 40 + ((size_t)64 << 20)));
  // ... and the first block's cells (4 bytes per nonzero, 8 when the minibatches are kept for
  // replay, + the cell offsets): set aside now, not mapped by the driver under the first build
  XF_TRY(xf_batch_pool_reserve(((size_t)block_size << 20) / 4 * 8 + ((size_t)16 << 20)));
  // ... and what the table's maintenance step inside the epoch (defrag_if_grown) would allocate
  if (world <= 1 && model_ == 0 && table_w_) XF_TRY(xf_table_prepare_defrag(table_w_));
  const double t0 = now_s();
  rows_trained_ = 0;
  blocks_gpu = blocks_host = 0;
  for (xf_sbatch *b : cache_) xf_sbatch_free(b);  // a second XFStartTrain starts from the files
  cache_.clear();
  bool cached = false;
  // compiled minibatches are kept in HBM for the epochs that replay them — with one epoch
  // there is none: the batches are one-shot (no key-sorted copy, their blobs go back to the pool)
  const int keep = cache_batches != 0 && epochs > 1;
  static const uint64_t kNoRows[1] = {0};
  for (int epoch = 0; epoch < epochs; ++epoch) {
    if (cached) {
      // The compiled batches depend only on the file, so later epochs replay them from
      // HBM instead of re-reading and re-sorting the text (the reference re-opens and
      // re-parses per epoch, lr_worker.cc:184).  Every rank holds the same number of them.
      for (xf_sbatch *b : cache_) {
        XF_TRY(xf_sharded_step(sharded_, b));
        uint32_t R = 0;
        xf_sbatch_dims(b, &R, nullptr, nullptr, nullptr);
        rows_trained_ += R;
      }
      XF_TRY(xf_sharded_check(sharded_));
    } else {
      // ingest = gpu: the text tokenised and hashed on the GPU (xf_ingest.hip), compiled from
      // device arrays — unless this rank's shard file cannot be mapped (empty, not a regular
      // file): the host reader below yields zero rows for such a file and the rank still joins
      // every collective with empty minibatches
      bool took = false;
      if (gpu_ingest_applies()) XF_TRY(text_epoch(epoch, keep, &took));
      if (took) {
        cached = keep != 0;
        goto epoch_done;
      }
      xf_reader *rd = nullptr;
      const bool trace = getenv("XF_TRACE_WORKER") != nullptr;  // per-block timeline on stderr
      const double te0 = now_s();
      bool writes_cache = false;
      XF_TRY(open_reader(&rd, train_data_path, (size_t)block_size << 20, &writes_cache));
      if (trace) fprintf(stderr, "epoch %d: reader open %.2f ms\n", epoch, (now_s() - te0) * 1e3);
      // The text of block i+1 is parsed on a second host thread while the GPU builds the keys
      // of block i and trains on it: two caller-owned blocks go round between the two threads.
      struct Parsed {
        xf_block *blk = nullptr;
        size_t rows = 0, nnz = 0;
        const uint64_t *rowptr = nullptr, *keys = nullptr;
        const int32_t *labels = nullptr;
        int rc = XF_OK;
        std::string err;
      };
      Parsed slot[2];
      for (int i = 0; i < 2; ++i) {
        int rc = blocks_[i] ? XF_OK : xf_block_create(&blocks_[i]);
        if (rc != XF_OK) {
          xf_reader_close(rd);
          return rc;
        }
        slot[i].blk = blocks_[i];
      }
      std::mutex mu;
      std::condition_variable cv;
      int filled[2] = {0, 0};  // 0 = free for the parser, 1 = parsed, waiting for the trainer
      bool stop = false;
      int trainer_dev = 0;
      (void)hipGetDevice(&trainer_dev);
      std::thread parser([&, trainer_dev] {
        // the block arrays are page-locked by this thread: on the trainer's GPU, not on
        // device 0 (the current device is per thread; every rank of a node would otherwise
        // create a context on GPU 0)
        (void)hipSetDevice(trainer_dev);
        for (int k = 0;; k ^= 1) {
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop || !filled[k]; });
            if (stop) return;
          }
          Parsed &p = slot[k];
          const int32_t *fgid = nullptr;
          p.rc = xf_reader_next_into(rd, p.blk, &p.rows, &p.nnz, &p.rowptr, &p.keys, &fgid,
                                     &p.labels);
          if (p.rc != XF_OK) p.err = xf_last_error();  // the message is thread-local
          const bool last = p.rc != XF_OK || p.rows == 0;
          {
            std::lock_guard<std::mutex> lk(mu);
            filled[k] = 1;
          }
          cv.notify_all();
          if (last) return;
        }
      });
      int rc = XF_OK;
      bool mine_done = false;
      for (int k = 0; rc == XF_OK;) {
        Parsed *p = nullptr;
        const double tw0 = now_s();
        if (!mine_done) {
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return filled[k] != 0; });
          }
          p = &slot[k];
          if (p->rc != XF_OK) {
            rc = xf::set_error(p->rc, "%s", p->err.c_str());
            break;  // (a parse error ends this rank; its peers time out on the next exchange)
          }
          if (p->rows == 0) mine_done = true;
        }
        bool any = false;
        rc = any_rank(!mine_done, &any);
        if (rc != XF_OK || !any) break;
        const size_t rows = mine_done ? 0 : p->rows;
        const double tw1 = now_s();
        double t_compile = 0, t_step = 0;
        const size_t thread_size = rows / core_num;  // remainder dropped, lr_worker.cc:190
        for (int i = 0; i < core_num && rc == XF_OK; ++i) {
          const size_t start = i * thread_size, end = (i + 1) * thread_size;
          if (end == start && world <= 1) continue;
          xf_sbatch *b = nullptr;
          const double tc0 = now_s();
          // a rank without rows of its own still takes part in the (collective) step
          rc = end > start ? xf_sharded_compile(sharded_, &b, p->rowptr, p->keys, p->labels,
                                                start, end, keep)
                           : xf_sharded_compile(sharded_, &b, kNoRows, nullptr,
                                                (const int32_t *)kNoRows, 0, 0, keep);
          if (rc != XF_OK) break;
          const double tc1 = now_s();
          rc = xf_sharded_step(sharded_, b);
          if (rc == XF_OK) rc = xf_sharded_check(sharded_);
          t_compile += tc1 - tc0;
          t_step += now_s() - tc1;
          if (rc != XF_OK) {
            xf_sbatch_free(b);
            break;
          }
          rows_trained_ += (long)(end - start);
          if (keep) cache_.push_back(b);
          else
            xf_sbatch_free(b);
        }
        // (the table's maintenance step inside an epoch: local to the shard, but its flush is
        // collective — one worker only)
        if (rc == XF_OK && world <= 1 && model_ == 0) rc = defrag_if_grown(30);
        if (trace)
          fprintf(stderr, "block: %zu rows  waited for the parser %.2f ms  key build %.2f ms  "
                  "step+check %.2f ms\n", rows, (tw1 - tw0) * 1e3, t_compile * 1e3, t_step * 1e3);
        if (!mine_done) {
          {
            std::lock_guard<std::mutex> lk(mu);
            filled[k] = 0;
          }
          cv.notify_all();
          k ^= 1;
        }
      }
      const double te1 = now_s();
      {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
      }
      cv.notify_all();
      parser.join();
      // unmapping a GB of text takes tens of ms: not on the training loop's clock (a reader
      // that is writing the block cache finishes the file when it closes: that one now)
      if (writes_cache) xf_reader_close(rd);
      else
      {
        for (std::thread &t : closers_) t.join();  // (the previous epoch's: long done)
        closers_.clear();
        closers_.emplace_back([rd] { xf_reader_close(rd); });
      }
      if (trace)
        fprintf(stderr, "epoch %d: blocks done after %.2f ms, reader and block buffers released "
                "in %.2f ms\n", epoch, (te1 - te0) * 1e3, (now_s() - te1) * 1e3);
      if (rc != XF_OK) return rc;
      cached = keep != 0;
    }
  epoch_done:
    // table maintenance at the epoch boundary: when this epoch inserted a noticeable share of
    // the keys, renumber the state rows in key order for the epochs that replay them.  The
    // flush is collective (every rank, every epoch); the defrag itself is local to the shard.
    const double tf0 = now_s();
    XF_TRY(xf_sharded_flush(sharded_));
    if (epoch + 1 < epochs) XF_TRY(defrag_if_grown(5));
    if (getenv("XF_TRACE_WORKER"))
      fprintf(stderr, "epoch %d: flush + table maintenance %.2f ms\n", epoch, (now_s() - tf0) * 1e3);
    if ((epoch + 1) % 30 == 0) std::cout << "epoch : " << epoch << std::endl;  // :202
  }
  XF_TRY(xf_sharded_flush(sharded_));
  train_seconds_ = now_s() - t0;
  return XF_OK;
}

// One epoch from the text with the GPU tokeniser (ingest = gpu).  A staging thread copies the
// next block's text from the mapped file into pinned memory (several host threads: a 64 MiB
// block is ~2 ms) while the GPU tokenises, compiles and steps the current one; two staging /
// tokeniser buffers go round between the two threads.  A block the tokeniser hands back (not of
// the common shape, xf_ingest.hip) is parsed from the staged text by the host parser: the same
// arrays, the reference's quirks in one place.  Block boundaries are the reader's (a1).
// the conditions of the GPU tokeniser's epoch that do not depend on the file
bool Worker::gpu_ingest_applies() const { return ingest_gpu && core_num == 1 && !block_cache; }

// *took = false: this rank's shard file is not a mapped regular file — nothing was done
int Worker::text_epoch(int epoch, int keep, bool *took) {
  *took = false;
  const bool trace = getenv("XF_TRACE_WORKER") != nullptr;
  const double te0 = now_s();
  xf_reader *rd = nullptr;
  XF_TRY(xf_reader_open(&rd, train_data_path, (size_t)block_size << 20));
  struct ReaderGuard {
    xf_reader *r;
    ~ReaderGuard() {
      if (r) xf_reader_close(r);
    }
  } rguard{rd};
  {
    int mapped = 0;
    XF_TRY(xf_reader_mapped(rd, &mapped));
    if (!mapped) return XF_OK;  // (xf_reader_peek_text / copy_text work on a mapping)
  }
  *took = true;
  for (int i = 0; i < 2; ++i)
    if (!ingest_[i]) XF_TRY(xf_ingest_create(&ingest_[i], (size_t)block_size << 20));
  if (!blocks_[0]) XF_TRY(xf_block_create(&blocks_[0]));
  void *stream = nullptr;
  XF_TRY(xf_sharded_stream(sharded_, &stream));
  struct Staged {
    size_t len = 0;
    int rc = XF_OK;
    std::string err;
  } slot[2];
  std::mutex mu;
  std::condition_variable cv;
  int filled[2] = {0, 0};  // 0 = free for the staging thread, 1 = staged
  bool stop = false;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const int copy_threads = std::max(2, std::min(16, xf::parse_threads()));
  std::thread stager([&, dev] {
    (void)hipSetDevice(dev);
    // the staged text goes up on this thread's own stream: the copy of block i + 1 (64 MiB: ~2.5 ms
    // of PCIe) runs while the GPU tokenises, compiles and steps block i
    hipStream_t up = nullptr;
    if (hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess) up = nullptr;
    struct StreamGuard {
      hipStream_t s;
      ~StreamGuard() {
        if (s) {
          (void)hipStreamSynchronize(s);
          (void)hipStreamDestroy(s);
        }
      }
    } sguard{up};
    for (int k = 0;; k ^= 1) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !filled[k]; });
        if (stop) return;
      }
      Staged &p = slot[k];
      char *buf = nullptr;
      size_t cap = 0;
      p.len = 0;
      p.rc = xf_ingest_staging(ingest_[k], &buf, &cap);
      if (p.rc == XF_OK) p.rc = xf_reader_copy_text(rd, buf, cap, &p.len, copy_threads);
      if (p.rc == XF_OK && p.len) p.rc = xf_reader_skip_text(rd);
      if (p.rc == XF_OK && p.len && up) p.rc = xf_ingest_upload(ingest_[k], p.len, up);
      if (p.rc != XF_OK) p.err = xf_last_error();
      const bool last = p.rc != XF_OK || p.len == 0;
      {
        std::lock_guard<std::mutex> lk(mu);
        filled[k] = 1;
      }
      cv.notify_all();
      if (last) return;
    }
  });
  static const uint64_t kNoRows[1] = {0};
  int rc = XF_OK;
  bool mine_done = false;
  for (int k = 0; rc == XF_OK;) {
    Staged *p = nullptr;
    const double tw0 = now_s();
    if (!mine_done) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return filled[k] != 0; });
      }
      p = &slot[k];
      if (p->rc != XF_OK) {
        rc = xf::set_error(p->rc, "%s", p->err.c_str());
        break;
      }
      if (p->len == 0) mine_done = true;
    }
    bool any = false;
    rc = any_rank(!mine_done, &any);
    if (rc != XF_OK || !any) break;
    const double tw1 = now_s();
    xf_sbatch *b = nullptr;
    size_t rows = 0;
    bool on_gpu = false;
    if (!mine_done) {
      const uint64_t *dk = nullptr;
      const uint32_t *drp = nullptr;
      const int32_t *dl = nullptr;
      uint32_t R = 0, NNZ = 0;
      int ok = 0;
      rc = xf_ingest_block(ingest_[k], nullptr, p->len, stream, &dk, &drp, &dl, &R, &NNZ, &ok);
      if (rc == XF_OK && ok) {
        rc = xf_sharded_compile_dev(sharded_, &b, dk, drp, dl, R, NNZ, keep);
        rows = R;
        on_gpu = true;
        ++blocks_gpu;
      } else if (rc == XF_OK) {  // not of the common shape: the host parser's
        char *buf = nullptr;
        size_t nnz = 0;
        const uint64_t *rowptr = nullptr, *keys = nullptr;
        const int32_t *fgid = nullptr, *labels = nullptr;
        rc = xf_ingest_staging(ingest_[k], &buf, nullptr);
        if (rc == XF_OK)
          rc = xf_reader_parse_text(rd, buf, p->len, blocks_[0], &rows, &nnz, &rowptr, &keys,
                                    &fgid, &labels);
        if (rc == XF_OK)
          rc = rows ? xf_sharded_compile(sharded_, &b, rowptr, keys, labels, 0, rows, keep)
                    : xf_sharded_compile(sharded_, &b, kNoRows, nullptr,
                                         (const int32_t *)kNoRows, 0, 0, keep);
        ++blocks_host;
      }
    } else {  // a rank without rows of its own still takes part in the (collective) step
      rc = xf_sharded_compile(sharded_, &b, kNoRows, nullptr, (const int32_t *)kNoRows, 0, 0, keep);
    }
    const double tc1 = now_s();
    if (rc == XF_OK) rc = xf_sharded_step(sharded_, b);
    if (rc == XF_OK) rc = xf_sharded_check(sharded_);
    if (rc != XF_OK) {
      if (b) xf_sbatch_free(b);
      break;
    }
    rows_trained_ += (long)rows;
    if (keep) cache_.push_back(b);
    else
      xf_sbatch_free(b);
    if (world <= 1 && model_ == 0) rc = defrag_if_grown(30);
    if (trace)
      fprintf(stderr, "block: %zu rows  waited for the parser %.2f ms  key build %.2f ms  "
              "step+check %.2f ms  (%s)\n", rows, (tw1 - tw0) * 1e3, (tc1 - tw1) * 1e3,
              (now_s() - tc1) * 1e3, on_gpu ? "tokenised on the GPU" : "parsed on the host");
    if (!mine_done) {
      {
        std::lock_guard<std::mutex> lk(mu);
        filled[k] = 0;
      }
      cv.notify_all();
      k ^= 1;
    }
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    stop = true;
  }
  cv.notify_all();
  stager.join();
  if (trace)
    fprintf(stderr, "epoch %d: blocks done after %.2f ms (%ld tokenised on the GPU, %ld parsed on "
            "the host)\n", epoch, (now_s() - te0) * 1e3, blocks_gpu, blocks_host);
  return rc;
}

// predict + calculate_pctr (lr_worker.cc:25-98, fm_worker.cc:25-124).  As in the reference
// only rank 0 scores its test file (lr_worker.cc:212-215) — against the WHOLE table: the other
// ranks serve its Pulls (`reader` == false: they step empty minibatches until rank 0 is done).
int Worker::predict(int rank, int block, bool reader) {
  std::ofstream md;
  xf_reader *rd = nullptr;
  if (reader) {
    char name[1200];
    if (pred_path.empty()) snprintf(name, sizeof(name), "pred_%d_%d.txt", rank, block);
    else
      snprintf(name, sizeof(name), "%s", pred_path.c_str());
    md.open(name);
    if (!md.is_open()) std::cout << "open pred file failure!" << std::endl;
    snprintf(test_data_path, sizeof(test_data_path), "%s-%05d", test_file_path.c_str(), rank);
    // 4 MiB blocks for LR (lr_worker.cc:80), 2 MiB for FM (fm_worker.cc:106)
    const size_t cap = model_ == 0 ? ((size_t)4 << 20) : ((size_t)2 << 20);
    XF_TRY(open_reader(&rd, test_data_path, cap));
  }
  struct ReaderGuard {  // every return path closes the reader
    xf_reader *r;
    ~ReaderGuard() {
      if (r) xf_reader_close(r);
    }
  } rd_guard{rd};
  static const uint64_t kNoRows[1] = {0};
  std::vector<int32_t> all_labels;
  std::vector<float> all_pctr, pctr;
  bool mine_done = !reader;
  while (true) {
    size_t rows = 0, nnz = 0;
    const uint64_t *rowptr = nullptr, *keys = nullptr;
    const int32_t *fgid = nullptr, *labels = nullptr;
    if (!mine_done) {
      XF_TRY(xf_reader_next(rd, &rows, &nnz, &rowptr, &keys, &fgid, &labels));
      if (rows == 0) mine_done = true;
    }
    bool any = false;
    XF_TRY(any_rank(!mine_done, &any));
    if (!any) break;
    const size_t thread_size = rows / core_num;
    for (int i = 0; i < core_num; ++i) {
      const size_t start = i * thread_size, end = (i + 1) * thread_size;
      if (end == start && world <= 1) continue;
      xf_sbatch *b = nullptr;
      if (end > start) XF_TRY(xf_sharded_compile(sharded_, &b, rowptr, keys, labels, start, end, 0));
      else
        XF_TRY(xf_sharded_compile(sharded_, &b, kNoRows, nullptr, (const int32_t *)kNoRows, 0, 0,
                                  0));
      struct BatchGuard {  // ... and frees the minibatch
        xf_sbatch *b;
        ~BatchGuard() { xf_sbatch_free(b); }
      } b_guard{b};
      pctr.resize(end - start);
      XF_TRY(xf_sharded_predict(sharded_, b, pctr.data()));
      for (size_t r = 0; r < end - start; ++r) {
        const int label = labels[start + r];
        all_labels.push_back(label);
        all_pctr.push_back(pctr[r]);
        md << pctr[r] << "\t" << 1 - label << "\t" << label << std::endl;  // :67
      }
    }
  }
  XF_TRY(xf_sharded_check(sharded_));
  if (!reader) return XF_OK;
  md.close();
  // Base::calculate_auc (base.h:84-110) and its stdout line
  XF_TRY(xf_auc_logloss(all_labels.data(), all_pctr.data(), all_labels.size(), &logloss_acc_,
                        &auc_, &tp_, &fp_, &logloss_nat_));
  std::cout << "logloss: " << logloss_acc_ << "\t";
  if (isnan(auc_)) std::cout << "tp_n = " << tp_ << std::endl;
  else
    std::cout << "auc = " << auc_ << "\ttp = " << tp_ << " fp = " << fp_ << std::endl;
  return XF_OK;
}

// train (lr_worker.cc:207-217, fm_worker.cc:277-287)
int Worker::train() {
  XF_TRY(create_tables());
  XF_TRY(start_ingest());
  std::cout << "my rank is = " << rank << std::endl;
  snprintf(train_data_path, sizeof(train_data_path), "%s-%05d", train_file_path.c_str(), rank);
  if (!model_in.empty()) {  // resume from a model file (XFLoadModel)
    void *self = this;
    XF_TRY(XFLoadModel(self, model_in.c_str()));
  }
  XF_TRY(batch_training());
  if (!model_out.empty()) {
    void *self = this;
    XF_TRY(XFSaveModel(self, model_out.c_str()));
  }
  if (rank == 0) std::cout << (model_ == 0 ? "LR AUC: " : "FM AUC: ") << std::endl;
  if (rank == 0 || world > 1) XF_TRY(predict(rank, 0, rank == 0));
  std::cout << "train end......" << std::endl;
  return XF_OK;
}

// one worker: one file at `path`; several: one file per shard + a manifest (xf_sharded_save)
int Worker::save_model(const char *path) {
  if (world > 1) return xf_sharded_save(sharded_, path);
  XF_TRY(xf_sharded_flush(sharded_));
  XF_TRY(xf::model_write(path, table_w_, table_v_, v_dim_));
  // a sharded checkpoint saved under the same name earlier is stale now: without this its
  // manifest would send a later load to the old shard files
  (void)unlink((std::string(path) + ".manifest").c_str());
  return XF_OK;
}

// a single file or a sharded checkpoint of ANY world size: every rank keeps the keys it owns
int Worker::load_model(const char *path) { return xf_sharded_load(sharded_, path); }

int Worker::set_param(const char *name, const char *value) {
  XF_REQUIRE(name && value, "XFSetParam: null argument");
  const std::string n = name;
  if (n == "model") {
    XF_REQUIRE(!sharded_, "XFSetParam: model cannot change after training started");
    model_ = atoi(value);
    XF_REQUIRE(model_ == 0 || model_ == 1, "XFSetParam: model must be 0 (LR) or 1 (FM)");
  } else if (n == "epochs") epochs = atoi(value);
  else if (n == "block_size_mb") block_size = atoi(value);
  else if (n == "core_num") core_num = atoi(value) > 0 ? atoi(value) : 1;
  else if (n == "k") v_dim_ = atoi(value);
  else if (n == "optimizer") {
    if (!strcmp(value, "ftrl")) optimizer = XF_OPT_FTRL;
    else if (!strcmp(value, "sgd")) optimizer = XF_OPT_SGD;
    else
      return xf::set_error(XF_EINVAL, "XFSetParam: optimizer must be ftrl or sgd");
  } else if (n == "capacity") capacity = strtoull(value, nullptr, 10);
  else if (n == "rank") {
    rank = atoi(value);
    rank_given_ = true;
  } else if (n == "world") world = atoi(value);
  else if (n == "transport") {
    if (!strcmp(value, "rccl")) transport = XF_TRANSPORT_RCCL;
    else if (!strcmp(value, "host")) transport = XF_TRANSPORT_HOST;
    else if (!strcmp(value, "auto")) transport = XF_TRANSPORT_AUTO;
    else
      return xf::set_error(XF_EINVAL, "XFSetParam: transport must be rccl, host or auto");
  } else if (n == "schedule") {
    if (!strcmp(value, "sequential")) schedule = XF_SCHEDULE_SEQUENTIAL;
    else if (!strcmp(value, "stale1")) schedule = XF_SCHEDULE_STALE1;
    else if (!strcmp(value, "owner")) schedule = XF_SCHEDULE_OWNER;
    else if (!strcmp(value, "owner_stale1")) schedule = XF_SCHEDULE_OWNER_STALE1;
    else
      return xf::set_error(XF_EINVAL, "XFSetParam: schedule must be sequential, stale1 or owner "
                           "(owner_stale1: the owner-compute dataflow, overlapped)");
  }
  else if (n == "pred_path") pred_path = value;
  else if (n == "alpha") alpha = (float)atof(value);
  else if (n == "beta") beta = (float)atof(value);
  else if (n == "lambda1") lambda1 = (float)atof(value);
  else if (n == "lambda2") lambda2 = (float)atof(value);
  else if (n == "lr") learning_rate = (float)atof(value);
  else if (n == "seed") seed = strtoull(value, nullptr, 10);
  else if (n == "cache_batches") cache_batches = atoi(value);
  else if (n == "parse_threads") return xf_tune("parse_threads", atof(value));
  else if (n == "ingest") {
    if (!strcmp(value, "gpu")) ingest_gpu = true;
    else if (!strcmp(value, "host")) ingest_gpu = false;
    else
      return xf::set_error(XF_EINVAL, "XFSetParam: ingest must be gpu or host");
  }
  else if (n == "block_cache") block_cache = atoi(value);
  else if (n == "block_cache_dir") block_cache_dir = value;
  else if (n == "model_in") model_in = value;
  else if (n == "model_out") model_out = value;
  else if (n == "update") {
    if (!strcmp(value, "rank_ordered")) update_rule = XF_UPDATE_RANK_ORDERED;
    else if (!strcmp(value, "sum_then_step")) update_rule = XF_UPDATE_SUM_THEN_STEP;
    else
      return xf::set_error(XF_EINVAL, "XFSetParam: update must be rank_ordered or sum_then_step");
  } else if (n == "parity") {
    if (!strcmp(value, "exact")) parity = XF_PARITY_EXACT_SUMS;
    else if (!strcmp(value, "reference_order")) parity = XF_PARITY_REFERENCE_ORDER;
    else
      return xf::set_error(XF_EINVAL, "XFSetParam: parity must be exact or reference_order");
  } else if (n == "key_build") {
    if (!strcmp(value, "gpu")) key_build_gpu = true;
    else if (!strcmp(value, "host")) key_build_gpu = false;
    else
      return xf::set_error(XF_EINVAL, "XFSetParam: key_build must be gpu or host");
  }
  else
    return xf::set_error(XF_EINVAL, "XFSetParam: unknown parameter '%s'", name);
  return XF_OK;
}

int Worker::get_metric(const char *name, double *value) {
  XF_REQUIRE(name && value, "XFGetMetric: null argument");
  const std::string n = name;
  if (n == "logloss_ref") *value = logloss_acc_;
  else if (n == "logloss_nat") *value = logloss_nat_;
  else if (n == "auc") *value = auc_;
  else if (n == "tp") *value = tp_;
  else if (n == "fp") *value = fp_;
  else if (n == "rows_trained") *value = (double)rows_trained_;
  else if (n == "blocks_gpu") *value = (double)blocks_gpu;
  else if (n == "blocks_host") *value = (double)blocks_host;
  else if (n == "train_seconds") *value = train_seconds_;
  else if (n == "examples_per_sec") *value = train_seconds_ > 0 ? rows_trained_ / train_seconds_ : 0;
  else if (n == "keys") {
    uint64_t k = 0;
    if (table_w_) XF_TRY(xf_table_size(table_w_, &k));
    *value = (double)k;
  } else
    return xf::set_error(XF_EINVAL, "XFGetMetric: unknown metric '%s'", name);
  return XF_OK;
}

}  // namespace xflow_amd

// ------------------------------------------------------------------------------- C API
// c_api.h:26-41: the handle is an XFlow object owning the worker.
extern "C" int XFCreate(void **h, const char *train_path, const char *test_path) {
  XF_REQUIRE(h && train_path && test_path, "XFCreate: null argument");
  *h = new xflow_amd::Worker(0, train_path, test_path);
  return XF_OK;
}

extern "C" int XFStartTrain(void **h) {
  XF_REQUIRE(h && *h, "XFStartTrain: null handle");
  return reinterpret_cast<xflow_amd::Worker *>(*h)->train();
}

extern "C" int XFDestroy(void **h) {
  if (h && *h) {
    delete reinterpret_cast<xflow_amd::Worker *>(*h);
    *h = nullptr;
  }
  return XF_OK;
}

extern "C" int XFSetParam(void *h, const char *name, const char *value) {
  XF_REQUIRE(h, "XFSetParam: null handle");
  return reinterpret_cast<xflow_amd::Worker *>(h)->set_param(name, value);
}

extern "C" int XFGetMetric(void *h, const char *name, double *value) {
  XF_REQUIRE(h, "XFGetMetric: null handle");
  return reinterpret_cast<xflow_amd::Worker *>(h)->get_metric(name, value);
}

// ---- model file (xf_modelfile.cc; per shard and resharding on load when the worker is sharded)
extern "C" int XFSaveModel(void *h, const char *path) {
  XF_REQUIRE(h && path, "XFSaveModel: null argument");
  xflow_amd::Worker *wk = reinterpret_cast<xflow_amd::Worker *>(h);
  XF_REQUIRE(wk->table_w(), "XFSaveModel: nothing trained or loaded yet");
  return wk->save_model(path);
}

extern "C" int XFLoadModel(void *h, const char *path) {
  XF_REQUIRE(h && path, "XFLoadModel: null argument");
  xflow_amd::Worker *wk = reinterpret_cast<xflow_amd::Worker *>(h);
  XF_TRY(wk->ensure_tables());
  wk->note_external_keys();
  return wk->load_model(path);
}

// score the test file with the current tables (no training): predict + AUC/logloss
extern "C" int XFPredict(void *h) {
  XF_REQUIRE(h, "XFPredict: null handle");
  xflow_amd::Worker *wk = reinterpret_cast<xflow_amd::Worker *>(h);
  XF_TRY(wk->ensure_tables());
  return wk->predict(wk->rank, 0, wk->rank == 0);
}

extern "C" int XFGetTables(void *h, xf_table **w, xf_table **v) {
  XF_REQUIRE(h, "XFGetTables: null handle");
  xflow_amd::Worker *wk = reinterpret_cast<xflow_amd::Worker *>(h);
  if (w) *w = wk->table_w();
  if (v) *v = wk->table_v();
  return XF_OK;
}
