// xf_worker.h — the worker classes (see xf_worker.cc).
#ifndef XF_WORKER_H_
#define XF_WORKER_H_

#include <string>
#include <vector>

#include "xf_common.h"

namespace xflow_amd {

// One class serves both models (model 0 = LRWorker, model 1 = FMWorker of the reference);
// LRWorker / FMWorker below keep the reference's constructor shape.
class Worker {
 public:
  Worker(int model, const char *train_file, const char *test_file);
  virtual ~Worker();

  int train();                       // lr_worker.cc:207-217
  int batch_training();              // lr_worker.cc:179-205
  int update(xf_batch *b);           // lr_worker.cc:145-177
  int predict(int rank, int block);  // lr_worker.cc:73-98

  int set_param(const char *name, const char *value);
  int get_metric(const char *name, double *value);
  int ensure_tables() { return create_tables(); }
  // keys entered the tables without passing through grow_if_needed (XFLoadModel)
  void note_external_keys() { size_known_stale_ = true; }
  xf_table *table_w() { return table_w_; }
  xf_table *table_v() { return table_v_; }

 public:
  int epochs = 60;  // lr_worker.h:63

  // reference constants, now parameters
  int rank = 0;
  int core_num = 1;      // slices per block; 1 = the deterministic reference schedule
  int block_size = 2;    // MiB, lr_worker.h:68
  int v_dim_ = 10;       // fm_worker.h:92
  int optimizer = XF_OPT_FTRL;  // server.h:24,28
  uint64_t capacity = 1ull << 22;
  float alpha = 5e-2f, beta = 1.0f, lambda1 = 5e-5f, lambda2 = 10.0f;  // ftrl.h:17-20
  float learning_rate = 0.001f;                                       // sgd.h:16
  uint64_t seed = 0;
  int cache_batches = 1;
  bool key_build_gpu = true;  // key build of update() on the GPU (xf_batch_compile_gpu)
  std::string pred_path;
  std::string model_in, model_out;  // load before / save after training (model file)
  // binarized block cache of the text files (xf_reader_open_cached): 0 = off; the cache
  // files go next to the data (<file>.xfcsr<cap>) or into block_cache_dir
  int block_cache = 0;
  std::string block_cache_dir;
  int open_reader(xf_reader **rd, const char *path, size_t cap);

 private:
  int create_tables();
  int compile(xf_batch **b, const uint64_t *rowptr, const uint64_t *keys, const int32_t *labels,
              size_t start, size_t end, bool keep);
  int grow_if_needed(size_t incoming);
  int defrag_if_grown();
  uint64_t keys_at_defrag_ = 0;
  uint64_t seen_upper_ = 0;
  bool size_known_stale_ = false;

  int model_;
  std::string train_file_path, test_file_path;
  char train_data_path[1200];
  char test_data_path[1200];
  xf_table *table_w_ = nullptr, *table_v_ = nullptr;  // kv_w_ / kv_v of the reference
  xf_workspace *ws_ = nullptr;
  std::vector<xf_batch *> cache_;
  long rows_trained_ = 0;
  double train_seconds_ = 0.0;
  float logloss_acc_ = 0.0f, auc_ = 0.0f;
  double logloss_nat_ = 0.0;
  int tp_ = 0, fp_ = 0;
};

class LRWorker : public Worker {
 public:
  LRWorker(const char *train_file, const char *test_file) : Worker(0, train_file, test_file) {}
};

class FMWorker : public Worker {
 public:
  FMWorker(const char *train_file, const char *test_file) : Worker(1, train_file, test_file) {}
};

}  // namespace xflow_amd
#endif  // XF_WORKER_H_
