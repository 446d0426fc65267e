// xf_cli.cc — `xflow_lr <train_prefix> <test_prefix> <model 0|1> <epochs> [name=value ...]`
// Same positional arguments as the reference's entry point (src/model/main.cc:27-44).  Started
// by the reference's scripts/local.sh it behaves like this: DMLC_ROLE=scheduler and =server
// have nothing to do (the "servers" are the key-range shards of the table in the workers' HBM,
// rank 0 is the rendezvous) and return at once; every DMLC_ROLE=worker is one GPU of a run of
// DMLC_NUM_WORKER, meeting the others at DMLC_PS_ROOT_URI:DMLC_PS_ROOT_PORT.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <string>

#include "xf_worker.h"

int main(int argc, char *argv[]) {
  if (argc < 5) {
    std::cout << "usage: xflow_lr <train_prefix> <test_prefix> <model: 0 LR | 1 FM> <epochs>"
                 " [name=value ...]\n"
              << "LR model example: xflow_lr data/small_train data/small_test 0 100\n"
              << "FM model example: xflow_lr data/small_train data/small_test 1 100\n";
    return 2;
  }
  if (const char *role = getenv("DMLC_ROLE")) {  // main.cc:22-26: ps::IsServer / scheduler
    if (!strcmp(role, "scheduler") || !strcmp(role, "server")) {
      std::cout << "xflow_lr: no " << role << " process in this build (the table lives in the "
                << "workers' HBM)" << std::endl;
      return 0;
    }
  }
  const int model = argv[3][0] - '0';
  if (model == 2) {
    std::cerr << "MVM (model 2) is out of scope of this build (SURVEY.md §2 row 11)\n";
    return 2;
  }
  xflow_amd::Worker *worker;
  if (model == 0) {
    std::cout << "start LR " << std::endl;
    worker = new xflow_amd::LRWorker(argv[1], argv[2]);
  } else {
    std::cout << "start FM " << std::endl;
    worker = new xflow_amd::FMWorker(argv[1], argv[2]);
  }
  worker->epochs = std::atoi(argv[4]);
  for (int i = 5; i < argc; ++i) {
    std::string kv = argv[i];
    const size_t eq = kv.find('=');
    if (eq == std::string::npos ||
        worker->set_param(kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str()) != XF_OK) {
      std::cerr << "bad option '" << kv << "': " << xf_last_error() << "\n";
      return 2;
    }
  }
  const int rc = worker->train();
  if (rc != XF_OK) {
    std::cerr << "xflow_lr: error " << rc << ": " << xf_last_error() << "\n";
    return 1;
  }
  double eps = 0;
  worker->get_metric("examples_per_sec", &eps);
  std::cout << "examples/sec (train loop): " << eps << std::endl;
  delete worker;
  return 0;
}
