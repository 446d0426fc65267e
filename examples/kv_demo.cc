// kv_demo.cc — drives examples/ps_gpu.h the way LRWorker::update drives ps-lite
// (lr_worker.cc:167-176): Pull the unique keys, compute a gradient, Push it, twice, and print
// the pulled weights.  tests/test_gpu_parity.py runs it on the GPU box and checks the output
// against the oracle's store.
//   kv_demo <nkeys> <steps>     keys = std::hash of "0".."nkeys-1", gradient g_i = 0.01*(i%7-3)
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "ps_gpu.h"

int main(int argc, char **argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 16;
  const int steps = argc > 2 ? atoi(argv[2]) : 2;
  try {
    ps::Start();
    ps::KVWorker<float> kv_w(0);
    std::vector<ps::Key> keys(n);
    for (size_t i = 0; i < n; ++i) {
      char buf[32];
      const int len = snprintf(buf, sizeof(buf), "%zu", i);
      keys[i] = xf_hash_bytes(buf, (size_t)len);  // io.h:53
    }
    std::sort(keys.begin(), keys.end());
    std::vector<float> w, g(n);
    for (int s = 0; s < steps; ++s) {
      kv_w.Wait(kv_w.Pull(keys, &w));
      for (size_t i = 0; i < n; ++i) g[i] = 0.01f * (float)((int)((i + s) % 7) - 3);
      kv_w.Wait(kv_w.Push(keys, g));
    }
    kv_w.Wait(kv_w.Pull(keys, &w));
    for (size_t i = 0; i < n; ++i) printf("%llu %a\n", (unsigned long long)keys[i], w[i]);
    ps::Finalize();
  } catch (const std::exception &e) {
    fprintf(stderr, "kv_demo: %s\n", e.what());
    return 1;
  }
  return 0;
}
