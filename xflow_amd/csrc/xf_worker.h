// xf_worker.h — the worker classes (see xf_worker.cc).
#ifndef XF_WORKER_H_
#define XF_WORKER_H_

#include <string>
#include <thread>
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
  int predict(int rank, int block, bool reader = true);  // lr_worker.cc:73-98

  int set_param(const char *name, const char *value);
  int get_metric(const char *name, double *value);
  int ensure_tables() { return create_tables(); }
  void note_external_keys() {}  // (the tables count their keys themselves)
  xf_table *table_w() { return table_w_; }
  xf_table *table_v() { return table_v_; }
  int save_model(const char *path);
  int load_model(const char *path);

 public:
  int epochs = 60;  // lr_worker.h:63

  // reference constants, now parameters
  int rank = 0;          // ps::MyRank(): given (rank=), from the environment, or handed out
  int world = 0;         // workers == GPUs; 0: WORLD_SIZE / XF_WORLD / DMLC_NUM_WORKER, else 1
  int transport = 0;                     // XF_TRANSPORT_*: RCCL; host = through the bootstrap sockets (tests)
  int schedule = XF_SCHEDULE_SEQUENTIAL; // order of Push(t) and Pull(t+1) when world > 1
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
  int parity = 0;             // XF_PARITY_*: the forward's row sums (one worker)
  int update_rule = 0;        // XF_UPDATE_*: how an owner applies the workers' pushes of a step
  std::string pred_path;
  std::string model_in, model_out;  // load before / save after training (model file)
  // binarized block cache of the text files (xf_reader_open_cached): 0 = off; the cache
  // files go next to the data (<file>.xfcsr<cap>) or into block_cache_dir
  int block_cache = 0;
  std::string block_cache_dir;
  int open_reader(xf_reader **rd, const char *path, size_t cap, bool *writes_cache = nullptr);
  // ingest = gpu: the text of a block is tokenised and hashed on the GPU (xf_ingest.hip) and
  // compiled from device arrays; a block that is not of the common shape is parsed on the host
  // as before.  Needs core_num = 1 and no block cache (both checked where the epoch starts).
  bool ingest_gpu = false;
  long blocks_gpu = 0, blocks_host = 0;  // how the blocks of the last training run were parsed

 private:
  int create_tables();
  int defrag_if_grown(int percent);
  int any_rank(bool mine, bool *any);
  uint64_t keys_seen_ = 0;  // the table's keys at the last look (defrag_if_grown: the inflow per minibatch)
  bool rank_given_ = false;

  int model_;
  std::string train_file_path, test_file_path;
  char train_data_path[1200];
  char test_data_path[1200];
  xf_group *group_ = nullptr;      // ps-lite's postoffice: the other workers
  xf_sharded *sharded_ = nullptr;  // owns the tables; with one worker: the fused single shard
  xf_table *table_w_ = nullptr, *table_v_ = nullptr;  // kv_w_ / kv_v of the reference
  std::vector<xf_sbatch *> cache_;
  // the two block buffers that go round between the parser thread and the trainer: pinned host
  // memory, allocated once per worker (pinning and unpinning ~200 MB per epoch cost 30-80 ms)
  xf_block *blocks_[2] = {nullptr, nullptr};
  xf_ingest *ingest_[2] = {nullptr, nullptr};  // ingest = gpu: two staging / tokeniser buffers
  int start_ingest();                          // the buffers + first launches, before the clock
  bool gpu_ingest_applies() const;
  int text_epoch(int epoch, int keep, bool *took);  // one epoch from the text, tokenised on the GPU
  std::vector<std::thread> closers_;  // readers being closed (munmap of the text) off the clock
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
