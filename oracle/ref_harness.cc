/*
 * ref_harness.cc — thin extern "C" window onto the parts of the REAL reference that
 * compile from their own sources in this image (no ps-lite needed):
 *     src/io/load_data_from_disk.{h,cc} + src/io/io.h   (a1 parser, a2 key hash)
 *     src/base/base.h                                   (a6 sigmoid, a15 AUC/logloss)
 *
 * TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libxflow_ref.so
 * straight from /root/reference (sources are compiled where they lie, never copied).
 * The worker/optimizer sources (src/model, src/optimizer) include "ps/ps.h" from the
 * empty ps-lite submodule and are therefore NOT buildable here; no stand-in is written.
 */
#include <sstream>
#include <string>
#include <vector>

#define private public /* reach Base::logloss (src/base/base.h:113) */
#include "src/base/base.h"
#undef private
#include "src/io/load_data_from_disk.h"

extern "C" {

unsigned long long ref_hash(const char *p, size_t len) {
  return std::hash<std::string>()(std::string(p, len)); /* src/io/io.h:53 */
}

float ref_sigmoid(float x) {
  xflow::Base b;
  return b.sigmoid(x);
}

void *ref_loader_open(const char *path, size_t cap_bytes) {
  return new xflow::LoadData(path, cap_bytes);
}
void ref_loader_close(void *h) { delete (xflow::LoadData *)h; }
/* runs load_minibatch_hash_data_fread(); returns rows, total nnz via *nnz */
long ref_loader_next(void *h, size_t *nnz) {
  xflow::LoadData *ld = (xflow::LoadData *)h;
  ld->load_minibatch_hash_data_fread();
  size_t t = 0;
  for (size_t i = 0; i < ld->m_data.fea_matrix.size(); ++i)
    t += ld->m_data.fea_matrix[i].size();
  *nnz = t;
  return (long)ld->m_data.fea_matrix.size();
}
/* flatten the current block: rowptr[rows+1], keys[nnz], fgid[nnz], labels[rows] */
void ref_loader_get(void *h, unsigned long long *rowptr, unsigned long long *keys,
                    int *fgid, int *labels) {
  xflow::LoadData *ld = (xflow::LoadData *)h;
  size_t o = 0;
  rowptr[0] = 0;
  for (size_t i = 0; i < ld->m_data.fea_matrix.size(); ++i) {
    for (size_t j = 0; j < ld->m_data.fea_matrix[i].size(); ++j) {
      keys[o] = ld->m_data.fea_matrix[i][j].fid;
      fgid[o] = ld->m_data.fea_matrix[i][j].fgid;
      ++o;
    }
    rowptr[i + 1] = o;
    labels[i] = ld->m_data.label[i];
  }
}

/* Base::calculate_auc prints "logloss: L\tauc = A\ttp = T fp = F" to std::cout
 * (base.h:101-108).  Returns that line in `line` and the raw float member. */
void ref_auc(const int *labels, const float *pctr, size_t n, float *logloss_out,
             char *line, size_t line_cap) {
  std::vector<xflow::Base::auc_key> v(n);
  for (size_t i = 0; i < n; ++i) {
    v[i].label = labels[i];
    v[i].pctr = pctr[i];
  }
  xflow::Base b;
  std::ostringstream cap;
  std::streambuf *old = std::cout.rdbuf(cap.rdbuf());
  b.calculate_auc(v);
  std::cout.rdbuf(old);
  *logloss_out = b.logloss;
  std::string s = cap.str();
  size_t m = s.size() < line_cap - 1 ? s.size() : line_cap - 1;
  for (size_t i = 0; i < m; ++i) line[i] = s[i];
  line[m] = '\0';
}

} /* extern "C" */
