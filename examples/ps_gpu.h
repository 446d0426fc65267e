// ps_gpu.h — the reference-side binding of INTEGRATION.md §A, complete and compilable:
// xflow's worker code (`ps::KVWorker<float>::Pull/Push/Wait`, lr_worker.cc:170,175;
// fm_worker.cc:228-242) bound to libxflow_amd.so's host-pointer table API.
//
// A maintainer of the reference drops this in place of `ps/ps.h` for a single-process,
// GPU-backed run: the server side (src/model/server.h, src/optimizer/*.h) is no longer
// compiled — the optimizer a server installed per app id (server.h:22-31) becomes the
// table's configuration.  Nothing here is used to build reference sources inside this
// repository; examples/kv_demo.cc drives it with its own calls.
#ifndef EXAMPLES_PS_GPU_H_
#define EXAMPLES_PS_GPU_H_

#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "xflow_amd.h"

namespace ps {

typedef uint64_t Key;

// what `xflow::Server` (server.h:22-31) decided on the server side
struct GpuStoreOptions {
  int optimizer = XF_OPT_FTRL;   // main.cc picks FTRL handles; SGD ones exist in sgd.h
  int v_dim = 10;                // ftrl.h:16
  uint64_t capacity = 1u << 22;  // index positions per table
};
inline GpuStoreOptions &gpu_store_options() {
  static GpuStoreOptions o;
  return o;
}

template <typename V>
class KVWorker {
 public:
  // app 0 = w (dim 1), app 1 = v (dim v_dim): lr_worker.h:38, fm_worker.h:37-38
  explicit KVWorker(int app_id) {
    static_assert(sizeof(V) == sizeof(float), "the table stores fp32");
    const GpuStoreOptions &o = gpu_store_options();
    xf_table_config c;
    xf_table_config_default(&c);  // FTRL hyper-parameters of ftrl.h:17-20, lr of sgd.h:16
    c.opt_kind = o.optimizer;
    c.capacity = o.capacity;
    if (app_id == 1) {
      c.dim = o.v_dim;
      if (o.optimizer == XF_OPT_FTRL) {
        c.init_kind = XF_INIT_HASHNORM;  // ftrl.h:114-120 (deterministic replacement)
      } else {
        c.init_kind = XF_INIT_CONST;     // sgd.h:69
        c.init_const = 0.001f;
      }
    }
    dim_ = c.dim;
    check(xf_table_create(&t_, &c));
  }
  ~KVWorker() { xf_table_destroy(t_); }
  KVWorker(const KVWorker &) = delete;
  KVWorker &operator=(const KVWorker &) = delete;

  // keys sorted ascending and unique (the ps-lite contract); blocking, so Wait is a no-op
  int Pull(const std::vector<Key> &keys, std::vector<V> *vals) {
    vals->resize(keys.size() * dim_);
    check(xf_table_pull(t_, keys.data(), keys.size(), reinterpret_cast<float *>(vals->data())));
    return 0;
  }
  int Push(const std::vector<Key> &keys, const std::vector<V> &vals) {
    if (vals.size() != keys.size() * dim_)  // CHECK_EQ(keys, vals/dim), ftrl.h:48
      throw std::runtime_error("Push: vals.size() != keys.size() * dim");
    check(xf_table_push(t_, keys.data(), keys.size(), reinterpret_cast<const float *>(vals.data())));
    return 0;
  }
  void Wait(int) {}
  xf_table *table() { return t_; }

 private:
  static void check(int rc) {
    if (rc != XF_OK) throw std::runtime_error(std::string("libxflow_amd: ") + xf_last_error());
  }
  xf_table *t_ = nullptr;
  size_t dim_ = 1;
};

// process-level calls of main.cc:22-47 for the one-process run
inline bool IsServer() { return false; }
inline bool IsWorker() { return true; }
inline bool IsScheduler() { return false; }
inline int MyRank() { return 0; }
inline void Start() {}
inline void Finalize() {}

}  // namespace ps
#endif  // EXAMPLES_PS_GPU_H_
