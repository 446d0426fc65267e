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
#include <iostream>

namespace xflow_amd {

static double now_s() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

Worker::Worker(int model, const char *train_file, const char *test_file)
    : model_(model), train_file_path(train_file ? train_file : ""),
      test_file_path(test_file ? test_file : "") {}

Worker::~Worker() {
  for (xf_batch *b : cache_) xf_batch_free(b);
  if (ws_) xf_workspace_destroy(ws_);
  if (table_w_) xf_table_destroy(table_w_);
  if (table_v_) xf_table_destroy(table_v_);
}

// xflow::Server (server.h:22-31): app 0 serves w, app 1 serves v; FTRL by default, the
// SGD handlers are the commented-out alternative (server.h:25,29).
int Worker::create_tables() {
  if (table_w_) return XF_OK;
  xf_table_config c;
  xf_table_config_default(&c);
  c.opt_kind = optimizer;
  c.alpha = alpha;
  c.beta = beta;
  c.lambda1 = lambda1;
  c.lambda2 = lambda2;
  c.lr = learning_rate;
  c.capacity = capacity;
  c.dim = 1;
  c.init_kind = XF_INIT_ZERO;
  XF_TRY(xf_table_create(&table_w_, &c));
  if (model_ == 1) {
    c.dim = v_dim_;
    if (optimizer == XF_OPT_FTRL) {  // ftrl.h:114-120
      c.init_kind = XF_INIT_HASHNORM;
      c.seed = seed;
    } else {  // sgd.h:67-72
      c.init_kind = XF_INIT_CONST;
      c.init_const = 0.001f;
    }
    XF_TRY(xf_table_create(&table_v_, &c));
  }
  XF_TRY(xf_workspace_create(&ws_));
  return XF_OK;
}

// Make room for up to `incoming` new keys before a pull can insert them: keep the load
// factor <= 0.6.  `seen_upper_` is a host-side upper bound on the key count so that the
// exact (synchronising) size query only runs when the bound gets close.
int Worker::grow_if_needed(size_t incoming) {
  xf_table *ts[2] = {table_w_, table_v_};
  uint64_t cap = 0;
  XF_TRY(xf_table_capacity(table_w_, &cap));
  if (size_known_stale_) {  // keys came in behind our back (model file): count them once
    uint64_t n = 0;
    XF_TRY(xf_table_size(table_w_, &n));
    seen_upper_ = std::max<uint64_t>(seen_upper_, n);
    size_known_stale_ = false;
  }
  if ((seen_upper_ + incoming) * 10 <= cap * 6) {
    seen_upper_ += incoming;
    return XF_OK;
  }
  uint64_t n = 0;
  XF_TRY(xf_table_size(table_w_, &n));
  if ((n + incoming) * 10 > cap * 6) {
    uint64_t want = cap * 2;
    while ((n + incoming) * 10 > want * 6) want *= 2;
    for (xf_table *t : ts)
      if (t) XF_TRY(xf_table_reserve(t, want));
  }
  seen_upper_ = n + incoming;
  return XF_OK;
}

int Worker::defrag_if_grown() {
  uint64_t n = 0;
  XF_TRY(xf_table_size(table_w_, &n));
  if (n > keys_at_defrag_ + keys_at_defrag_ / 20) {  // > 5 % new keys since the last one
    XF_TRY(xf_table_defrag(table_w_));
    if (table_v_) XF_TRY(xf_table_defrag(table_v_));
    keys_at_defrag_ = n;
  }
  return XF_OK;
}

// the key build of update() (lr_worker.cc:146-166).  LR: straight against the table (raw keys
// -> state rows -> cells, no sort; the table grows by itself); `keep` = the batch will be
// replayed in later epochs.  FM (and key_build=host): the sorted-unique-key build.
int Worker::compile(xf_batch **b, const uint64_t *rowptr, const uint64_t *keys,
                    const int32_t *labels, size_t start, size_t end, bool keep) {
  if (model_ == 0 && key_build_gpu)
    return xf_batch_compile_local(b, table_w_, rowptr, keys, labels, start, end, keep ? 1 : 0,
                                  nullptr);
  if (key_build_gpu) return xf_batch_compile_gpu(b, rowptr, keys, labels, start, end, nullptr);
  return xf_batch_compile(b, rowptr, keys, labels, start, end);
}

// LRWorker::update / FMWorker::update (lr_worker.cc:145-177, fm_worker.cc:204-245)
int Worker::update(xf_batch *b) {
  uint32_t U = 0;
  xf_batch_dims(b, nullptr, nullptr, &U, nullptr);
  XF_TRY(grow_if_needed(U));
  if (model_ == 0) XF_TRY(xf_lr_step(table_w_, b, ws_, nullptr));
  else
    XF_TRY(xf_fm_step(table_w_, table_v_, b, ws_, nullptr));
  return XF_OK;
}

int Worker::open_reader(xf_reader **rd, const char *path, size_t cap) {
  if (!block_cache) return xf_reader_open(rd, path, cap);
  std::string base = path;
  if (!block_cache_dir.empty()) {
    const size_t slash = base.find_last_of('/');
    base = block_cache_dir + "/" + (slash == std::string::npos ? base : base.substr(slash + 1));
  }
  const std::string cpath = base + ".xfcsr" + std::to_string(cap);
  return xf_reader_open_cached(rd, path, cap, cpath.c_str(), nullptr);
}

// batch_training (lr_worker.cc:179-205, fm_worker.cc:247-275)
int Worker::batch_training() {
  {  // init push of key 0 with a zero gradient (lr_worker.cc:180-182, fm_worker.cc:248-252)
    const uint64_t key0 = 0;
    const float zero = 0.0f;
    XF_TRY(xf_table_push(table_w_, &key0, 1, &zero));
    if (model_ == 1) {
      std::vector<float> zv(v_dim_, 0.0f);
      XF_TRY(xf_table_push(table_v_, &key0, 1, zv.data()));
    }
  }
  const double t0 = now_s();
  rows_trained_ = 0;
  for (xf_batch *b : cache_) xf_batch_free(b);  // a second XFStartTrain starts from the files
  cache_.clear();
  bool cached = false;
  for (int epoch = 0; epoch < epochs; ++epoch) {
    if (cached) {
      // The compiled batches depend only on the file, so later epochs replay them from
      // HBM instead of re-reading and re-sorting the text (the reference re-opens and
      // re-parses per epoch, lr_worker.cc:184).
      for (xf_batch *b : cache_) {
        XF_TRY(update(b));
        uint32_t R = 0;
        xf_batch_dims(b, &R, nullptr, nullptr, nullptr);
        rows_trained_ += R;
      }
      XF_TRY(xf_table_check(table_w_, nullptr));
    } else {
      xf_reader *rd = nullptr;
      XF_TRY(open_reader(&rd, train_data_path, (size_t)block_size << 20));
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
      for (auto &p : slot) {
        int rc = xf_block_create(&p.blk);
        if (rc != XF_OK) {
          xf_reader_close(rd);
          return rc;
        }
      }
      std::mutex mu;
      std::condition_variable cv;
      int filled[2] = {0, 0};  // 0 = free for the parser, 1 = parsed, waiting for the trainer
      bool stop = false;
      std::thread parser([&] {
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
      for (int k = 0; rc == XF_OK; k ^= 1) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return filled[k] != 0; });
        }
        Parsed &p = slot[k];
        if (p.rc != XF_OK) {
          rc = xf::set_error(p.rc, "%s", p.err.c_str());
          break;
        }
        if (p.rows == 0) break;
        const size_t thread_size = p.rows / core_num;  // remainder dropped, lr_worker.cc:190
        for (int i = 0; i < core_num && rc == XF_OK; ++i) {
          const size_t start = i * thread_size, end = (i + 1) * thread_size;
          if (end == start) continue;
          xf_batch *b = nullptr;
          rc = compile(&b, p.rowptr, p.keys, p.labels, start, end, cache_batches != 0);
          if (rc != XF_OK) break;
          rc = update(b);
          if (rc == XF_OK) rc = xf_table_check(table_w_, nullptr);
          if (rc == XF_OK && table_v_) rc = xf_table_check(table_v_, nullptr);
          if (rc != XF_OK) {
            xf_batch_free(b);
            break;
          }
          rows_trained_ += (long)(end - start);
          if (cache_batches) cache_.push_back(b);
          else
            xf_batch_free(b);
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          filled[k] = 0;
        }
        cv.notify_all();
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        stop = true;
      }
      cv.notify_all();
      parser.join();
      for (auto &p : slot) xf_block_destroy(p.blk);
      if (rc != XF_OK) {
        xf_reader_close(rd);
        return rc;
      }
      xf_reader_close(rd);
      cached = cache_batches != 0;
    }
    // table maintenance at the epoch boundary: when this epoch inserted a noticeable share of
    // the keys, renumber the state rows in key order for the epochs that replay them
    if (epoch + 1 < epochs) XF_TRY(defrag_if_grown());
    if ((epoch + 1) % 30 == 0) std::cout << "epoch : " << epoch << std::endl;  // :202
  }
  XF_TRY(xf_stream_sync(nullptr));
  train_seconds_ = now_s() - t0;
  return XF_OK;
}

// predict + calculate_pctr (lr_worker.cc:25-98, fm_worker.cc:25-124)
int Worker::predict(int rank, int block) {
  char name[1200];
  if (pred_path.empty()) snprintf(name, sizeof(name), "pred_%d_%d.txt", rank, block);
  else
    snprintf(name, sizeof(name), "%s", pred_path.c_str());
  std::ofstream md(name);
  if (!md.is_open()) std::cout << "open pred file failure!" << std::endl;
  snprintf(test_data_path, sizeof(test_data_path), "%s-%05d", test_file_path.c_str(), rank);
  // 4 MiB blocks for LR (lr_worker.cc:80), 2 MiB for FM (fm_worker.cc:106)
  const size_t cap = model_ == 0 ? ((size_t)4 << 20) : ((size_t)2 << 20);
  xf_reader *rd = nullptr;
  XF_TRY(open_reader(&rd, test_data_path, cap));
  struct ReaderGuard {  // every return path closes the reader
    xf_reader *r;
    ~ReaderGuard() { xf_reader_close(r); }
  } rd_guard{rd};
  std::vector<int32_t> all_labels;
  std::vector<float> all_pctr, pctr;
  while (true) {
    size_t rows = 0, nnz = 0;
    const uint64_t *rowptr, *keys;
    const int32_t *fgid, *labels;
    XF_TRY(xf_reader_next(rd, &rows, &nnz, &rowptr, &keys, &fgid, &labels));
    if (rows == 0) break;
    const size_t thread_size = rows / core_num;
    for (int i = 0; i < core_num; ++i) {
      const size_t start = i * thread_size, end = (i + 1) * thread_size;
      if (end == start) continue;
      xf_batch *b = nullptr;
      XF_TRY(compile(&b, rowptr, keys, labels, start, end, false));
      struct BatchGuard {  // ... and frees the minibatch
        xf_batch *b;
        ~BatchGuard() { xf_batch_free(b); }
      } b_guard{b};
      pctr.resize(end - start);
      uint32_t U = 0;
      xf_batch_dims(b, nullptr, nullptr, &U, nullptr);
      XF_TRY(grow_if_needed(U));
      XF_TRY(model_ == 0 ? xf_lr_predict(table_w_, b, ws_, pctr.data())
                         : xf_fm_predict(table_w_, table_v_, b, ws_, pctr.data()));
      for (size_t r = 0; r < end - start; ++r) {
        const int label = labels[start + r];
        all_labels.push_back(label);
        all_pctr.push_back(pctr[r]);
        md << pctr[r] << "\t" << 1 - label << "\t" << label << std::endl;  // :67
      }
    }
  }
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
  if (rank == 0) {
    std::cout << (model_ == 0 ? "LR AUC: " : "FM AUC: ") << std::endl;
    XF_TRY(predict(rank, 0));
  }
  std::cout << "train end......" << std::endl;
  return XF_OK;
}

int Worker::set_param(const char *name, const char *value) {
  XF_REQUIRE(name && value, "XFSetParam: null argument");
  const std::string n = name;
  if (n == "model") {
    XF_REQUIRE(!table_w_, "XFSetParam: model cannot change after training started");
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
  else if (n == "rank") rank = atoi(value);
  else if (n == "pred_path") pred_path = value;
  else if (n == "alpha") alpha = (float)atof(value);
  else if (n == "beta") beta = (float)atof(value);
  else if (n == "lambda1") lambda1 = (float)atof(value);
  else if (n == "lambda2") lambda2 = (float)atof(value);
  else if (n == "lr") learning_rate = (float)atof(value);
  else if (n == "seed") seed = strtoull(value, nullptr, 10);
  else if (n == "cache_batches") cache_batches = atoi(value);
  else if (n == "parse_threads") return xf_tune("parse_threads", atof(value));
  else if (n == "block_cache") block_cache = atoi(value);
  else if (n == "block_cache_dir") block_cache_dir = value;
  else if (n == "model_in") model_in = value;
  else if (n == "model_out") model_out = value;
  else if (n == "key_build") {
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

// ---- model file: the tables' (key, w, n, z) dumps, sorted by key -------------------------
// (the reference never saves its model; this is the export/import hook with a
// file format around it)
namespace {
const char kMagic[8] = {'X', 'F', 'A', 'M', 'D', '0', '0', '1'};

int save_table(FILE *f, xf_table *t, int dim) {
  size_t n = 0;
  XF_TRY(xf_table_export(t, nullptr, nullptr, nullptr, nullptr, 0, &n));
  std::vector<uint64_t> keys(n);
  std::vector<float> w(n * dim), nn(n * dim), z(n * dim);
  if (n) XF_TRY(xf_table_export(t, keys.data(), w.data(), nn.data(), z.data(), n, &n));
  const uint64_t hdr[2] = {(uint64_t)n, (uint64_t)dim};
  if (fwrite(hdr, 8, 2, f) != 2 || fwrite(keys.data(), 8, n, f) != n ||
      fwrite(w.data(), 4, n * dim, f) != n * dim || fwrite(nn.data(), 4, n * dim, f) != n * dim ||
      fwrite(z.data(), 4, n * dim, f) != n * dim)
    return xf::set_error(XF_EIO, "XFSaveModel: short write");
  return XF_OK;
}

int load_table(FILE *f, xf_table *t, int dim) {
  uint64_t hdr[2];
  if (fread(hdr, 8, 2, f) != 2) return xf::set_error(XF_EIO, "XFLoadModel: truncated file");
  if ((int)hdr[1] != dim)
    return xf::set_error(XF_EINVAL, "XFLoadModel: file has dim %llu, table has %d",
                         (unsigned long long)hdr[1], dim);
  const size_t n = (size_t)hdr[0];
  {  // the header's count must fit what is left of the file (a corrupt or foreign file must
     // not turn into a giant allocation)
    const long here = ftell(f);
    fseek(f, 0, SEEK_END);
    const long end = ftell(f);
    fseek(f, here, SEEK_SET);
    if (here < 0 || end < here || (unsigned long long)n * (8ull + 12ull * dim) >
                                      (unsigned long long)(end - here))
      return xf::set_error(XF_EIO, "XFLoadModel: the file claims %zu keys but holds %ld bytes",
                           n, end - here);
  }
  std::vector<uint64_t> keys(n);
  std::vector<float> w(n * dim), nn(n * dim), z(n * dim);
  if (fread(keys.data(), 8, n, f) != n || fread(w.data(), 4, n * dim, f) != n * dim ||
      fread(nn.data(), 4, n * dim, f) != n * dim || fread(z.data(), 4, n * dim, f) != n * dim)
    return xf::set_error(XF_EIO, "XFLoadModel: truncated file");
  uint64_t cap = 0;
  XF_TRY(xf_table_capacity(t, &cap));
  if ((uint64_t)n * 10 > cap * 6) XF_TRY(xf_table_reserve(t, (uint64_t)n * 2 + 1024));
  return xf_table_import(t, keys.data(), n, w.data(), nn.data(), z.data());
}
}  // namespace

extern "C" int XFSaveModel(void *h, const char *path) {
  XF_REQUIRE(h && path, "XFSaveModel: null argument");
  xflow_amd::Worker *wk = reinterpret_cast<xflow_amd::Worker *>(h);
  XF_REQUIRE(wk->table_w(), "XFSaveModel: nothing trained or loaded yet");
  FILE *f = fopen(path, "wb");
  if (!f) return xf::set_error(XF_EIO, "XFSaveModel: cannot open %s", path);
  const uint64_t nt = wk->table_v() ? 2 : 1;
  int rc = fwrite(kMagic, 1, 8, f) == 8 && fwrite(&nt, 8, 1, f) == 1 ? XF_OK
                                                                       : xf::set_error(XF_EIO, "XFSaveModel: short write");
  if (rc == XF_OK) rc = save_table(f, wk->table_w(), 1);
  if (rc == XF_OK && wk->table_v()) rc = save_table(f, wk->table_v(), wk->v_dim_);
  fclose(f);
  return rc;
}

extern "C" int XFLoadModel(void *h, const char *path) {
  XF_REQUIRE(h && path, "XFLoadModel: null argument");
  xflow_amd::Worker *wk = reinterpret_cast<xflow_amd::Worker *>(h);
  XF_TRY(wk->ensure_tables());
  wk->note_external_keys();
  FILE *f = fopen(path, "rb");
  if (!f) return xf::set_error(XF_EIO, "XFLoadModel: cannot open %s", path);
  char magic[8];
  uint64_t nt = 0;
  int rc = XF_OK;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kMagic, 8) != 0 || fread(&nt, 8, 1, f) != 1)
    rc = xf::set_error(XF_EINVAL, "XFLoadModel: %s is not an xflow_amd model file", path);
  if (rc == XF_OK && nt != (wk->table_v() ? 2u : 1u))
    rc = xf::set_error(XF_EINVAL, "XFLoadModel: file holds %llu table(s), worker has %d",
                       (unsigned long long)nt, wk->table_v() ? 2 : 1);
  if (rc == XF_OK) rc = load_table(f, wk->table_w(), 1);
  if (rc == XF_OK && wk->table_v()) rc = load_table(f, wk->table_v(), wk->v_dim_);
  fclose(f);
  return rc;
}

// score the test file with the current tables (no training): predict + AUC/logloss
extern "C" int XFPredict(void *h) {
  XF_REQUIRE(h, "XFPredict: null handle");
  xflow_amd::Worker *wk = reinterpret_cast<xflow_amd::Worker *>(h);
  XF_TRY(wk->ensure_tables());
  return wk->predict(wk->rank, 0);
}

extern "C" int XFGetTables(void *h, xf_table **w, xf_table **v) {
  XF_REQUIRE(h, "XFGetTables: null handle");
  xflow_amd::Worker *wk = reinterpret_cast<xflow_amd::Worker *>(h);
  if (w) *w = wk->table_w();
  if (v) *v = wk->table_v();
  return XF_OK;
}
