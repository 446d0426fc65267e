// xf_metrics.cc — AUC / logloss of a scored test set.
//
// Replaces Base::calculate_auc (src/base/base.h:84-110): sort by descending pctr
// (std::sort, so the tie order is libstdc++'s, as in the reference), AUC by rank sum,
// and the reference's "logloss" = mean of y*log2(p) + (1-y)*log2(1-p) — base 2 and
// negative — accumulated into a member that is never reset (base.h:113).  The
// conventional natural-log logloss is returned beside it.
#include <algorithm>
#include <cmath>
#include <vector>

#include "xf_common.h"

namespace {
struct AucKey {
  int label;
  float pctr;
};
}  // namespace

extern "C" int xf_auc_logloss(const int32_t *labels, const float *pctr, size_t n,
                              float *acc_logloss_inout, float *auc, int *tp, int *fp,
                              double *nat_logloss) {
  XF_REQUIRE(n == 0 || (labels && pctr), "xf_auc_logloss: null argument");
  XF_REQUIRE(acc_logloss_inout && auc && tp && fp, "xf_auc_logloss: null output");
  std::vector<AucKey> v(n);
  for (size_t i = 0; i < n; ++i) {
    v[i].label = labels[i];
    v[i].pctr = pctr[i];
  }
  std::sort(v.begin(), v.end(), [](const AucKey &a, const AucKey &b) { return a.pctr > b.pctr; });
  float logloss = *acc_logloss_inout;
  float area = 0.0f;
  int tp_n = 0;
  double nat = 0.0;
  for (size_t i = 0; i < n; ++i) {
    if (v[i].label == 1) tp_n += 1;
    else
      area += tp_n;
    // float log2 for the positive term, double for the negative one: base.h:97-98
    logloss += v[i].label * std::log2(v[i].pctr) + (1.0 - v[i].label) * std::log2(1.0 - v[i].pctr);
    const double p = std::min(std::max((double)v[i].pctr, 1e-15), 1.0 - 1e-15);
    nat -= v[i].label ? std::log(p) : std::log(1.0 - p);
  }
  logloss /= n;
  *acc_logloss_inout = logloss;
  if (tp_n == 0 || (size_t)tp_n == n) {
    *auc = NAN;  // the reference prints only tp_n in this case (base.h:102-103)
  } else {
    area /= 1.0 * (tp_n * (n - tp_n));
    *auc = area;
  }
  *tp = tp_n;
  *fp = (int)(n - tp_n);
  if (nat_logloss) *nat_logloss = n ? nat / (double)n : 0.0;
  return XF_OK;
}
