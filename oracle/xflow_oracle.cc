/*
 * xflow_oracle.cc — CPU restatement of the xswang/xflow LR/FM + FTRL/SGD hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see xflow_oracle.h).  C-style C++: g++ is used rather than
 * gcc so that the three places where the reference's result depends on libstdc++
 * behaviour are reproduced by the same library code the reference would run here:
 *   - std::sort tie order of all_keys        (src/model/lr/lr_worker.cc:162)
 *   - std::sort tie order inside the AUC     (src/base/base.h:85-88)
 *   - std::log2(float) vs std::log2(double)  (src/base/base.h:97-98)
 * Build: see oracle/Makefile (-O2 -ffp-contract=off: the reference is built without
 * FMA contraction, CMakeLists.txt:6-8).
 *
 * PINNING (details in oracle/README.md):
 *   a1/a2/a6/a15 (parser, key hash, sigmoid, AUC/logloss) are checked against the real
 *   reference sources compiled into oracle/_ref (tests/test_oracle.py, and the three-way
 *   fuzzes of tests/test_capi_cpu.py) and the committed golden vectors made from them
 *   (tests/golden/).  a3-a10, a13, a14 (LR worker math, FTRL/SGD handlers, train loop)
 *   cannot be compiled here without a stand-in for the absent ps-lite headers, so they
 *   are pinned only by the end-to-end values SURVEY.md §4/§8c records from the survey's
 *   run of the reference (logloss -0.886206, auc 0.547149, tp 46, fp 154, 525/877 keys).
 *   a11/a12 (FM loss / gradient) have no reference-side number at all: PARITY UNPINNED
 *   beyond the line-by-line citation.
 */
#include "xflow_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <thread>
#include <vector>

/* ======================================================================= a2 */
/* libstdc++ hash_bytes.cc (64-bit size_t variant), reached through
 * std::hash<std::string> at src/io/io.h:53 / load_data_from_disk.cc:151. */
static inline uint64_t xo_shift_mix(uint64_t v) { return v ^ (v >> 47); }

extern "C" uint64_t xo_hash_bytes(const void *ptr, size_t len) {
  const uint64_t mul = (((uint64_t)0xc6a4a793UL) << 32) + (uint64_t)0x5bd1e995UL;
  const unsigned char *buf = (const unsigned char *)ptr;
  const size_t len_aligned = len & ~(size_t)7;
  uint64_t hash = 0xc70f6907UL ^ (len * mul);
  for (size_t off = 0; off < len_aligned; off += 8) {
    uint64_t word;
    memcpy(&word, buf + off, 8); /* little-endian unaligned load */
    const uint64_t data = xo_shift_mix(word * mul) * mul;
    hash ^= data;
    hash *= mul;
  }
  const size_t tail = len & 7;
  if (tail != 0) {
    uint64_t data = 0;
    for (size_t i = tail; i-- > 0;) data = (data << 8) + buf[len_aligned + i];
    hash ^= data;
    hash *= mul;
  }
  hash = xo_shift_mix(hash) * mul;
  hash = xo_shift_mix(hash);
  return hash;
}

extern "C" uint32_t xo_shard_of(uint64_t key, uint32_t nshards) {
  if (nshards <= 1) return 0;
  const uint64_t span = UINT64_MAX / nshards;
  const uint64_t s = key / span;
  return (uint32_t)(s < nshards - 1 ? s : nshards - 1);
}

/* ======================================================================= a6 */
extern "C" float xo_sigmoid(float x) { /* base.h:54-63 */
  if (x < -30) {
    return 1e-6;
  } else if (x > 30) {
    return 1.0;
  } else {
    double ex = pow(2.718281828, x); /* base is NOT e */
    return ex / (1.0 + ex);
  }
}

/* ======================================================================= a1 */
struct xo_reader {
  FILE *fp;
  size_t cap;
  std::vector<char> buf;
  size_t have; /* bytes carried / valid at the front of buf */
  std::vector<uint64_t> rowptr, keys;
  std::vector<int32_t> fgid, labels;
};

extern "C" xo_reader *xo_reader_open(const char *path, size_t cap_bytes) {
  FILE *fp = fopen(path, "r");
  if (!fp || cap_bytes < 2) {
    if (fp) fclose(fp);
    return NULL;
  }
  xo_reader *r = new xo_reader;
  r->fp = fp;
  r->cap = cap_bytes;
  r->buf.resize(cap_bytes);
  r->have = 0;
  return r;
}
extern "C" void xo_reader_close(xo_reader *r) {
  if (!r) return;
  fclose(r->fp);
  delete r;
}

/* One token "fgid:fid:val" (load_data_from_disk.cc:141-156): fgid = (int)atof(field0),
 * fid = hash(field1); field2 is never read.  A token without two ':' is malformed
 * (the reference would run off the buffer). */
static bool xo_parse_token(const char *t, const char *te, int32_t *fg, uint64_t *fid) {
  const char *c1 = (const char *)memchr(t, ':', (size_t)(te - t));
  if (!c1) return false;
  const char *c2 = (const char *)memchr(c1 + 1, ':', (size_t)(te - (c1 + 1)));
  if (!c2) return false;
  char tmp[64];
  size_t l0 = (size_t)(c1 - t);
  if (l0 >= sizeof(tmp)) return false;
  memcpy(tmp, t, l0);
  tmp[l0] = '\0';
  *fg = (int32_t)atof(tmp);
  *fid = xo_hash_bytes(c1 + 1, (size_t)(c2 - (c1 + 1)));
  return true;
}

extern "C" long xo_reader_next(xo_reader *r) {
  r->rowptr.clear();
  r->keys.clear();
  r->fgid.clear();
  r->labels.clear();
  r->rowptr.push_back(0);
  /* fill: the reference keeps one byte for the terminator (:108-110) */
  const size_t room = r->cap - 1 - r->have;
  r->have += fread(&r->buf[r->have], 1, room, r->fp);
  size_t text_len, consumed;
  if (r->have + 1 == r->cap) { /* full buffer: cut at the last newline (:112-121) */
    size_t cut = r->have;
    while (cut > 0 && r->buf[cut - 1] != (char)EOF && r->buf[cut - 1] != '\n') --cut;
    if (cut == 0) return -1; /* a line longer than the block */
    text_len = cut - 1;      /* the newline itself becomes the terminator */
    consumed = cut;
  } else {
    text_len = r->have;
    consumed = r->have;
  }
  const char *p = &r->buf[0];
  const char *end = p + text_len;
  while (p < end && *p != '\0') {
    const char *le = (const char *)memchr(p, '\n', (size_t)(end - p));
    if (!le) le = end;
    const char *tab = (const char *)memchr(p, '\t', (size_t)(le - p));
    if (!tab) return -1;
    char tmp[64];
    size_t ll = (size_t)(tab - p);
    if (ll >= sizeof(tmp)) return -1;
    memcpy(tmp, p, ll);
    tmp[ll] = '\0';
    float y_tmp = std::atof(tmp); /* :129-134 */
    float y;
    if (y_tmp > 0.0000001) y = 1;
    else
      y = 0;
    r->labels.push_back((int32_t)y);
    const char *t = tab + 1;
    const size_t row_first = r->keys.size();
    /* An EMPTY token (two blanks in a row, or a blank right before the block terminator)
     * makes the reference push its `keyval` again without having parsed anything into it
     * (:178-196 with pp == qq, :160-177 for the terminator case): the row gets a duplicate of
     * its previous token.  A blank right before '\n' is not a token at all (:138 ends the
     * row).  With no previous token in the row the stale value comes from an earlier row or
     * is uninitialised: rejected. */
    while (t < le) {
      const char *te = (const char *)memchr(t, ' ', (size_t)(le - t));
      if (!te) te = le;
      if (te == t) {
        if (r->keys.size() == row_first) return -1;
        r->keys.push_back(r->keys.back());
        r->fgid.push_back(r->fgid.back());
        t = te + 1;
        continue;
      }
      int32_t fg;
      uint64_t fid;
      if (!xo_parse_token(t, te, &fg, &fid)) return -1;
      r->keys.push_back(fid);
      r->fgid.push_back(fg);
      t = te + 1;
    }
    if (le == end && le > tab + 1 && le[-1] == ' ') { /* blank, then the terminator */
      if (r->keys.size() == row_first) return -1;
      r->keys.push_back(r->keys.back());
      r->fgid.push_back(r->fgid.back());
    } else if (le == end && le == tab + 1) {
      return -1; /* a row without tokens at the terminator: the reference pushes a stale token */
    }
    r->rowptr.push_back(r->keys.size());
    p = le + 1;
  }
  /* carry the unconsumed tail to the front (:104-107) */
  if (consumed < r->have) memmove(&r->buf[0], &r->buf[consumed], r->have - consumed);
  r->have -= consumed;
  return (long)r->labels.size();
}
extern "C" size_t xo_reader_rows(const xo_reader *r) { return r->labels.size(); }
extern "C" size_t xo_reader_nnz(const xo_reader *r) { return r->keys.size(); }
extern "C" const uint64_t *xo_reader_rowptr(const xo_reader *r) { return r->rowptr.data(); }
extern "C" const uint64_t *xo_reader_keys(const xo_reader *r) { return r->keys.data(); }
extern "C" const int32_t *xo_reader_fgid(const xo_reader *r) { return r->fgid.data(); }
extern "C" const int32_t *xo_reader_labels(const xo_reader *r) { return r->labels.data(); }

/* ============================================================ a4/a8/a9/a10 */
static inline uint64_t xo_mix64(uint64_t x) { /* splitmix64 finaliser */
  x += 0x9e3779b97f4a7c15ULL;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}

/* Deterministic replacement for the time-seeded N(0,1)*1e-2 of ftrl.h:114-120:
 * sum of twelve 16-bit uniforms (Irwin-Hall, variance 1) computed in integers so the
 * CPU and the GPU produce the same bits.  Statistical, not bitwise, match with the
 * reference. */
extern "C" float xo_hashnorm(uint64_t seed, uint64_t key, uint32_t j) {
  uint64_t base = xo_mix64(seed ^ xo_mix64(key)) + (uint64_t)j * 0xd1342543de82ef95ULL;
  uint64_t sum = 0;
  for (int t = 0; t < 3; ++t) {
    uint64_t r = xo_mix64(base + (uint64_t)t);
    sum += (r & 0xffff) + ((r >> 16) & 0xffff) + ((r >> 32) & 0xffff) + (r >> 48);
  }
  double x = ((double)(int64_t)sum - 393210.0) / 65536.0;
  return (float)(x * 1e-2);
}

struct xo_store {
  int opt, dim, init_kind;
  float init_const;
  uint64_t seed;
  float alpha, beta, l1, l2, lr;
  /* exact map: open addressing over (key -> entry index); entries in insertion order */
  std::vector<uint64_t> hkeys;
  std::vector<uint32_t> hidx; /* 0 = empty, else entry+1 */
  std::vector<uint64_t> ekeys;
  std::vector<float> w, n, z;
};

static void xo_store_rehash(xo_store *s, size_t ncap) {
  s->hkeys.assign(ncap, 0);
  s->hidx.assign(ncap, 0);
  const size_t mask = ncap - 1;
  for (size_t e = 0; e < s->ekeys.size(); ++e) {
    size_t h = (size_t)xo_mix64(s->ekeys[e]) & mask;
    while (s->hidx[h]) h = (h + 1) & mask;
    s->hkeys[h] = s->ekeys[e];
    s->hidx[h] = (uint32_t)(e + 1);
  }
}

/* `store[key]` with insert-on-miss: ftrl.h:56 (zero entry), ftrl.h:112-121 (random v),
 * sgd.h:46/90 (zero / 0.001 entry). */
static size_t xo_store_entry(xo_store *s, uint64_t key) {
  size_t mask = s->hkeys.size() - 1;
  size_t h = (size_t)xo_mix64(key) & mask;
  while (s->hidx[h]) {
    if (s->hkeys[h] == key) return s->hidx[h] - 1;
    h = (h + 1) & mask;
  }
  if ((s->ekeys.size() + 1) * 2 > s->hkeys.size()) {
    xo_store_rehash(s, s->hkeys.size() * 2);
    mask = s->hkeys.size() - 1;
    h = (size_t)xo_mix64(key) & mask;
    while (s->hidx[h]) h = (h + 1) & mask;
  }
  const size_t e = s->ekeys.size();
  s->ekeys.push_back(key);
  s->hkeys[h] = key;
  s->hidx[h] = (uint32_t)(e + 1);
  for (int j = 0; j < s->dim; ++j) {
    float w0 = 0.0f;
    if (s->init_kind == XO_INIT_CONST) w0 = s->init_const;
    if (s->init_kind == XO_INIT_HASHNORM) w0 = xo_hashnorm(s->seed, key, (uint32_t)j);
    s->w.push_back(w0);
    s->n.push_back(0.0f);
    s->z.push_back(0.0f);
  }
  return e;
}

extern "C" xo_store *xo_store_create(int opt_kind, int dim, int init_kind,
                                     float init_const, uint64_t seed) {
  xo_store *s = new xo_store;
  s->opt = opt_kind;
  s->dim = dim;
  s->init_kind = init_kind;
  s->init_const = init_const;
  s->seed = seed;
  s->alpha = 5e-2; /* ftrl.h:17-20 */
  s->beta = 1.0;
  s->l1 = 5e-5;
  s->l2 = 10.0;
  s->lr = 0.001; /* sgd.h:16 */
  xo_store_rehash(s, 1024);
  return s;
}
extern "C" void xo_store_destroy(xo_store *s) { delete s; }
extern "C" void xo_store_set_ftrl(xo_store *s, float a, float b, float l1, float l2) {
  s->alpha = a;
  s->beta = b;
  s->l1 = l1;
  s->l2 = l2;
}
extern "C" void xo_store_set_sgd(xo_store *s, float lr) { s->lr = lr; }
extern "C" size_t xo_store_size(const xo_store *s) { return s->ekeys.size(); }

extern "C" void xo_ftrl_step(float alpha, float beta, float lambda1, float lambda2,
                             float g, float *pw, float *pn, float *pz) {
  /* ftrl.h:59-74, statement for statement, fp32 throughout */
  float old_n = *pn;
  float n = old_n + g * g;
  *pz += g - (std::sqrt(n) - std::sqrt(old_n)) / alpha * *pw;
  *pn = n;
  if (std::abs(*pz) <= lambda1) {
    *pw = 0.0;
  } else {
    float tmpr = 0.0;
    if (*pz > 0.0) tmpr = *pz - lambda1;
    if (*pz < 0.0) tmpr = *pz + lambda1;
    float tmpl = -1 * ((beta + std::sqrt(*pn)) / alpha + lambda2);
    *pw = tmpr / tmpl;
  }
}

extern "C" void xo_store_pull(xo_store *s, const uint64_t *keys, size_t nk, float *out) {
  for (size_t i = 0; i < nk; ++i) {
    const size_t e = xo_store_entry(s, keys[i]);
    for (int j = 0; j < s->dim; ++j) out[i * s->dim + j] = s->w[e * s->dim + j];
  }
}

extern "C" void xo_store_push(xo_store *s, const uint64_t *keys, size_t nk,
                              const float *grads) {
  for (size_t i = 0; i < nk; ++i) {
    const size_t e = xo_store_entry(s, keys[i]);
    for (int j = 0; j < s->dim; ++j) {
      const float g = grads[i * s->dim + j];
      const size_t o = e * s->dim + j;
      if (s->opt == XO_OPT_FTRL) {
        xo_ftrl_step(s->alpha, s->beta, s->l1, s->l2, g, &s->w[o], &s->n[o], &s->z[o]);
      } else {
        s->w[o] -= s->lr * g; /* sgd.h:52,96 */
      }
    }
  }
}

extern "C" void xo_store_export(const xo_store *s, uint64_t *keys, float *w, float *nn,
                                float *z) {
  const size_t ne = s->ekeys.size();
  /* (key, entry) pairs sorted by key: keys are unique, so the order is the key order whatever
   * the sort does with ties; sorting the pairs instead of entry numbers through an indirect
   * comparison matters at 10^8 entries only */
  std::vector<std::pair<uint64_t, uint32_t> > kv(ne);
  for (size_t i = 0; i < ne; ++i) kv[i] = std::make_pair(s->ekeys[i], (uint32_t)i);
  std::sort(kv.begin(), kv.end());
  std::vector<uint32_t> order(ne);
  for (size_t i = 0; i < ne; ++i) order[i] = kv[i].second;
  std::vector<std::pair<uint64_t, uint32_t> >().swap(kv);
  for (size_t i = 0; i < ne; ++i) {
    const size_t e = order[i];
    keys[i] = s->ekeys[e];
    for (int j = 0; j < s->dim; ++j) {
      if (w) w[i * s->dim + j] = s->w[e * s->dim + j];
      if (nn) nn[i * s->dim + j] = s->n[e * s->dim + j];
      if (z) z[i * s->dim + j] = s->z[e * s->dim + j];
    }
  }
}

/* room for `n` keys without a rehash on the way (test-harness convenience: a 10^8-key run) */
extern "C" void xo_store_reserve(xo_store *s, size_t n) {
  size_t cap = s->hkeys.size();
  while (cap < 2 * (n + 1)) cap *= 2;
  if (cap != s->hkeys.size()) xo_store_rehash(s, cap);
  s->ekeys.reserve(n);
  s->w.reserve(n * s->dim);
  s->n.reserve(n * s->dim);
  s->z.reserve(n * s->dim);
}

extern "C" void xo_store_import(xo_store *s, const uint64_t *keys, size_t nk,
                                const float *w, const float *nn, const float *z) {
  for (size_t i = 0; i < nk; ++i) {
    const size_t e = xo_store_entry(s, keys[i]);
    for (int j = 0; j < s->dim; ++j) {
      const size_t o = e * s->dim + j;
      if (w) s->w[o] = w[i * s->dim + j];
      if (nn) s->n[o] = nn[i * s->dim + j];
      if (z) s->z[o] = z[i * s->dim + j];
    }
  }
}

/* ======================================================================= a3 */
struct xo_batch {
  size_t rows, nnz, nu;
  std::vector<uint64_t> ukeys;
  std::vector<uint32_t> rowptr, uidx, segptr, coo_row;
  std::vector<int32_t> labels;
};

struct xo_sample_key { /* base.h:65-69 (fgid is never set on the LR/FM path) */
  uint64_t fid;
  int sid;
};
static bool xo_sort_finder(const xo_sample_key &a, const xo_sample_key &b) {
  return a.fid < b.fid; /* base.h:71-73 */
}

extern "C" xo_batch *xo_batch_build(const uint64_t *rowptr, const uint64_t *keys,
                                    const int32_t *labels, size_t row_begin,
                                    size_t row_end) {
  xo_batch *b = new xo_batch;
  b->rows = row_end - row_begin;
  std::vector<xo_sample_key> all_keys; /* lr_worker.cc:146-161 */
  std::vector<uint64_t> unique_keys;
  b->rowptr.push_back(0);
  int line_num = 0;
  for (size_t row = row_begin; row < row_end; ++row) {
    xo_sample_key sk;
    sk.sid = line_num;
    for (uint64_t j = rowptr[row]; j < rowptr[row + 1]; ++j) {
      sk.fid = keys[j];
      all_keys.push_back(sk);
      unique_keys.push_back(keys[j]);
    }
    b->rowptr.push_back((uint32_t)all_keys.size());
    b->labels.push_back(labels[row]);
    ++line_num;
  }
  std::sort(all_keys.begin(), all_keys.end(), xo_sort_finder); /* :162 */
  std::sort(unique_keys.begin(), unique_keys.end());           /* :163 */
  unique_keys.erase(std::unique(unique_keys.begin(), unique_keys.end()),
                    unique_keys.end()); /* :164-165 */
  b->nnz = all_keys.size();
  b->nu = unique_keys.size();
  b->ukeys.swap(unique_keys);
  /* group the sorted all_keys by key: segment u = [segptr[u], segptr[u+1]) */
  b->segptr.assign(b->nu + 1, 0);
  b->coo_row.resize(b->nnz);
  size_t u = 0;
  for (size_t j = 0; j < b->nnz; ++j) {
    while (b->ukeys[u] != all_keys[j].fid) {
      ++u;
      b->segptr[u] = (uint32_t)j;
    }
    b->coo_row[j] = (uint32_t)all_keys[j].sid;
  }
  for (size_t t = u + 1; t <= b->nu; ++t) b->segptr[t] = (uint32_t)b->nnz;
  /* CSR-order index of every nnz into ukeys */
  b->uidx.resize(b->nnz);
  size_t o = 0;
  for (size_t row = row_begin; row < row_end; ++row)
    for (uint64_t j = rowptr[row]; j < rowptr[row + 1]; ++j)
      b->uidx[o++] = (uint32_t)(std::lower_bound(b->ukeys.begin(), b->ukeys.end(),
                                                 keys[j]) -
                                b->ukeys.begin());
  return b;
}
extern "C" void xo_batch_free(xo_batch *b) { delete b; }
extern "C" size_t xo_batch_rows(const xo_batch *b) { return b->rows; }
extern "C" size_t xo_batch_nnz(const xo_batch *b) { return b->nnz; }
extern "C" size_t xo_batch_nuniq(const xo_batch *b) { return b->nu; }
extern "C" const uint64_t *xo_batch_ukeys(const xo_batch *b) { return b->ukeys.data(); }
extern "C" const uint32_t *xo_batch_rowptr(const xo_batch *b) { return b->rowptr.data(); }
extern "C" const uint32_t *xo_batch_uidx(const xo_batch *b) { return b->uidx.data(); }
extern "C" const uint32_t *xo_batch_segptr(const xo_batch *b) { return b->segptr.data(); }
extern "C" const uint32_t *xo_batch_coo_row(const xo_batch *b) { return b->coo_row.data(); }
extern "C" const int32_t *xo_batch_labels(const xo_batch *b) { return b->labels.data(); }

/* Summation mode.  0 (default) = the reference's own arithmetic: every per-row / per-key
 * sum is an fp32 running sum in the reference's visiting order.  1 = "exact-sum" variant
 * of the SAME algorithm: the sums (and only the sums) are accumulated in fp64 and rounded
 * to fp32 where the reference stores an fp32 value.  The reference's order within a key is
 * std::sort's (unspecified, lr_worker.cc:162), so mode 0 is one of several legal roundings;
 * mode 1 is order-independent and is what the GPU kernels compute (tests compare the GPU
 * with mode 1 bit-for-bit, and mode 1 with mode 0 within the fp32 accumulation noise). */
static int g_sum_mode = 0;
extern "C" void xo_set_sum_mode(int mode) { g_sum_mode = mode; }
extern "C" int xo_get_sum_mode(void) { return g_sum_mode; }

/* ==================================================================== a5/a7 */
/* The reference walks all_keys (sorted) against unique_keys with a merge-join
 * (lr_worker.cc:127-138); iterating segment by segment visits the same (j, i) pairs in
 * the same order, so every fp32 accumulation below happens in the reference's order. */
static void xo_wx(const xo_batch *b, const float *w, std::vector<float> &wx) {
  wx.assign(b->rows, 0.0f);
  if (g_sum_mode == 1) {
    std::vector<double> acc(b->rows, 0.0);
    for (size_t u = 0; u < b->nu; ++u)
      for (uint32_t j = b->segptr[u]; j < b->segptr[u + 1]; ++j)
        acc[b->coo_row[j]] += (double)w[u];
    for (size_t i = 0; i < b->rows; ++i) wx[i] = (float)acc[i];
    return;
  }
  for (size_t u = 0; u < b->nu; ++u)
    for (uint32_t j = b->segptr[u]; j < b->segptr[u + 1]; ++j) wx[b->coo_row[j]] += w[u];
}

extern "C" void xo_lr_loss(const xo_batch *b, const float *w, float *loss, float *pctr) {
  std::vector<float> wx;
  xo_wx(b, w, wx);
  for (size_t i = 0; i < b->rows; ++i) { /* lr_worker.cc:139-142 */
    float p = xo_sigmoid(wx[i]);
    if (pctr) pctr[i] = p;
    loss[i] = p - b->labels[i];
  }
}

extern "C" void xo_lr_grad(const xo_batch *b, const float *loss, float *g) {
  for (size_t u = 0; u < b->nu; ++u) { /* lr_worker.cc:104-115 */
    if (g_sum_mode == 1) {
      double acc = 0.0;
      for (uint32_t j = b->segptr[u]; j < b->segptr[u + 1]; ++j)
        acc += (double)loss[b->coo_row[j]];
      g[u] = (float)acc;
      continue;
    }
    float acc = 0.0f;
    for (uint32_t j = b->segptr[u]; j < b->segptr[u + 1]; ++j) acc += loss[b->coo_row[j]];
    g[u] = acc;
  }
  for (size_t u = 0; u < b->nu; ++u) g[u] /= 1.0 * b->rows; /* :116-118 (double divide) */
}

extern "C" void xo_lr_update(xo_store *ws, const xo_batch *b) { /* lr_worker.cc:167-176 */
  std::vector<float> w(b->nu), g(b->nu), loss(b->rows);
  xo_store_pull(ws, b->ukeys.data(), b->nu, w.data());
  xo_lr_loss(b, w.data(), loss.data(), NULL);
  xo_lr_grad(b, loss.data(), g.data());
  xo_store_push(ws, b->ukeys.data(), b->nu, g.data());
}

/* ============================================================== a11/a12/a13 */
extern "C" void xo_fm_loss(const xo_batch *b, int k, const float *w, const float *v,
                           float *loss, float *pctr, float *v_sum) {
  std::vector<float> wx;
  xo_wx(b, w, wx); /* fm_worker.cc:166-176 */
  std::vector<float> v_pow_sum(b->rows, 0.0f);
  for (size_t i = 0; i < b->rows; ++i) v_sum[i] = 0.0f;
  if (g_sum_mode == 1) {
    std::vector<double> vs(b->rows, 0.0), vp(b->rows, 0.0);
    for (int kk = 0; kk < k; ++kk)
      for (size_t u = 0; u < b->nu; ++u)
        for (uint32_t j = b->segptr[u]; j < b->segptr[u + 1]; ++j) {
          const uint32_t sid = b->coo_row[j];
          float v_weight = v[u * k + kk];
          vs[sid] += (double)v_weight;
          vp[sid] += (double)(v_weight * v_weight);
        }
    for (size_t i = 0; i < b->rows; ++i) {
      v_sum[i] = (float)vs[i];
      v_pow_sum[i] = (float)vp[i];
    }
  } else
  for (int kk = 0; kk < k; ++kk) { /* k-outer, pooled over k: fm_worker.cc:177-192 */
    for (size_t u = 0; u < b->nu; ++u) {
      for (uint32_t j = b->segptr[u]; j < b->segptr[u + 1]; ++j) {
        const uint32_t sid = b->coo_row[j];
        float v_weight = v[u * k + kk];
        v_sum[sid] += v_weight;
        v_pow_sum[sid] += v_weight * v_weight;
      }
    }
  }
  for (size_t i = 0; i < b->rows; ++i) {
    float v_y = v_sum[i] * v_sum[i] - v_pow_sum[i]; /* :193-196, no 1/2 */
    float p = xo_sigmoid(wx[i] + v_y);              /* :198-201 */
    if (pctr) pctr[i] = p;
    loss[i] = p - b->labels[i];
  }
}

extern "C" void xo_fm_grad(const xo_batch *b, int k, const float *v, const float *v_sum,
                           const float *loss, float *gw, float *gv) {
  for (size_t u = 0; u < b->nu; ++u) gw[u] = 0.0f;
  for (size_t u = 0; u < b->nu * (size_t)k; ++u) gv[u] = 0.0f;
  if (g_sum_mode == 1) {
    for (size_t u = 0; u < b->nu; ++u) {
      double accw = 0.0;
      for (uint32_t j = b->segptr[u]; j < b->segptr[u + 1]; ++j)
        accw += (double)loss[b->coo_row[j]];
      gw[u] = (float)(accw * (double)k); /* :140 adds loss once per factor */
      for (int kk = 0; kk < k; ++kk) {
        double accv = 0.0;
        for (uint32_t j = b->segptr[u]; j < b->segptr[u + 1]; ++j) {
          const uint32_t sid = b->coo_row[j];
          accv += (double)(loss[sid] * (v_sum[sid] - v[u * k + kk]));
        }
        gv[u * k + kk] = (float)accv;
      }
    }
  } else
  for (int kk = 0; kk < k; ++kk) { /* fm_worker.cc:134-148: gw accumulates k times */
    for (size_t u = 0; u < b->nu; ++u) {
      for (uint32_t j = b->segptr[u]; j < b->segptr[u + 1]; ++j) {
        const uint32_t sid = b->coo_row[j];
        gw[u] += loss[sid];
        gv[u * k + kk] += loss[sid] * (v_sum[sid] - v[u * k + kk]);
      }
    }
  }
  const size_t line_num = b->rows; /* :150-156 */
  for (size_t u = 0; u < b->nu; ++u) gw[u] /= 1.0 * line_num;
  for (size_t u = 0; u < b->nu * (size_t)k; ++u) gv[u] /= 1.0 * line_num;
}

extern "C" void xo_fm_update(xo_store *ws, xo_store *vs, const xo_batch *b) {
  const int k = vs->dim; /* fm_worker.cc:226-242 */
  std::vector<float> w(b->nu), v(b->nu * k), gw(b->nu), gv(b->nu * k);
  std::vector<float> loss(b->rows), v_sum(b->rows);
  xo_store_pull(ws, b->ukeys.data(), b->nu, w.data());
  xo_store_pull(vs, b->ukeys.data(), b->nu, v.data());
  xo_fm_loss(b, k, w.data(), v.data(), loss.data(), NULL, v_sum.data());
  xo_fm_grad(b, k, v.data(), v_sum.data(), loss.data(), gw.data(), gv.data());
  xo_store_push(ws, b->ukeys.data(), b->nu, gw.data());
  xo_store_push(vs, b->ukeys.data(), b->nu, gv.data());
}

/* ====================================================================== a15 */
struct xo_auc_key { /* base.h:79-82 */
  int label;
  float pctr;
};

extern "C" void xo_auc_logloss(const int32_t *labels, const float *pctr, size_t n,
                               float *acc_logloss_inout, float *auc, int *tp, int *fp) {
  std::vector<xo_auc_key> auc_vec(n);
  for (size_t i = 0; i < n; ++i) {
    auc_vec[i].label = labels[i];
    auc_vec[i].pctr = pctr[i];
  }
  std::sort(auc_vec.begin(), auc_vec.end(),
            [](const xo_auc_key &a, const xo_auc_key &b) { return a.pctr > b.pctr; });
  float logloss = *acc_logloss_inout;
  float area = 0.0;
  int tp_n = 0;
  for (size_t i = 0; i < auc_vec.size(); ++i) { /* base.h:91-100 */
    if (auc_vec[i].label == 1) {
      tp_n += 1;
    } else {
      area += tp_n;
    }
    logloss += auc_vec[i].label * std::log2(auc_vec[i].pctr) +
               +(1.0 - auc_vec[i].label) * std::log2(1.0 - auc_vec[i].pctr);
  }
  logloss /= auc_vec.size();
  *acc_logloss_inout = logloss;
  if (tp_n == 0 || (size_t)tp_n == auc_vec.size()) {
    *auc = NAN; /* reference prints only tp_n (base.h:102-103) */
  } else {
    area /= 1.0 * (tp_n * (auc_vec.size() - tp_n));
    *auc = area;
  }
  *tp = tp_n;
  *fp = (int)(auc_vec.size() - tp_n);
}

/* ====================================================================== a14 */
extern "C" long xo_train(int model, xo_store *ws, xo_store *vs, const char *train_path,
                         int epochs, size_t block_bytes, int core_num) {
  { /* init push of key 0: lr_worker.cc:180-182, fm_worker.cc:248-252 */
    uint64_t key0 = 0;
    float zero = 0.0f;
    xo_store_push(ws, &key0, 1, &zero);
    if (model == 1) {
      std::vector<float> zv(vs->dim, 0.0f);
      xo_store_push(vs, &key0, 1, zv.data());
    }
  }
  long consumed = 0;
  for (int epoch = 0; epoch < epochs; ++epoch) {
    xo_reader *rd = xo_reader_open(train_path, block_bytes); /* re-opened per epoch :184 */
    if (!rd) return -1;
    while (1) {
      long rows = xo_reader_next(rd);
      if (rows < 0) {
        xo_reader_close(rd);
        return -1;
      }
      if (rows == 0) break;
      const size_t thread_size = (size_t)rows / core_num; /* :190, remainder dropped */
      for (int i = 0; i < core_num; ++i) {
        const size_t start = i * thread_size, end = (i + 1) * thread_size;
        xo_batch *b = xo_batch_build(xo_reader_rowptr(rd), xo_reader_keys(rd),
                                     xo_reader_labels(rd), start, end);
        if (model == 0) xo_lr_update(ws, b);
        else
          xo_fm_update(ws, vs, b);
        xo_batch_free(b);
        consumed += (long)(end - start);
      }
    }
    xo_reader_close(rd);
  }
  return consumed;
}

/* The reference's slice fan-out as a timed baseline (lr_worker.cc:186-200): the block's rows
 * are cut into core_num slices, each slice's update() — its own key build (the two std::sorts),
 * Pull, loss, gradient, Push — runs on its own thread (the reference enqueues them on a
 * ThreadPool of hardware_concurrency() threads and spins until all are done); the server side
 * is ps-lite's single customer thread, i.e. Pull and Push are served one at a time (a mutex
 * here).  Not a parity path: the order in which the slices' pushes land is a race in the
 * reference too. */
extern "C" long xo_lr_update_slices_mt(xo_store *ws, const uint64_t *rowptr, const uint64_t *keys,
                                       const int32_t *labels, size_t rows, int core_num) {
  if (core_num < 1) core_num = 1;
  const size_t thread_size = rows / (size_t)core_num; /* :190, remainder dropped */
  std::mutex server;
  std::vector<std::thread> pool;
  for (int i = 0; i < core_num; ++i) {
    const size_t start = i * thread_size, end = (i + 1) * thread_size;
    if (end == start) continue;
    pool.emplace_back([=, &server]() {
      xo_batch *b = xo_batch_build(rowptr, keys, labels, start, end);
      std::vector<float> w(b->nu), g(b->nu), loss(b->rows);
      {
        std::lock_guard<std::mutex> lk(server);
        xo_store_pull(ws, b->ukeys.data(), b->nu, w.data());
      }
      xo_lr_loss(b, w.data(), loss.data(), NULL);
      xo_lr_grad(b, loss.data(), g.data());
      {
        std::lock_guard<std::mutex> lk(server);
        xo_store_push(ws, b->ukeys.data(), b->nu, g.data());
      }
      xo_batch_free(b);
    });
  }
  for (auto &t : pool) t.join();
  return (long)(thread_size * (size_t)core_num);
}

extern "C" long xo_predict(int model, xo_store *ws, xo_store *vs, const char *test_path,
                           size_t block_bytes, int core_num, int32_t *labels_out,
                           float *pctr_out, size_t cap) {
  xo_reader *rd = xo_reader_open(test_path, block_bytes);
  if (!rd) return -1;
  size_t n = 0;
  while (1) {
    long rows = xo_reader_next(rd);
    if (rows < 0) {
      xo_reader_close(rd);
      return -1;
    }
    if (rows == 0) break;
    const size_t thread_size = (size_t)rows / core_num;
    for (int i = 0; i < core_num; ++i) {
      const size_t start = i * thread_size, end = (i + 1) * thread_size;
      xo_batch *b = xo_batch_build(xo_reader_rowptr(rd), xo_reader_keys(rd),
                                   xo_reader_labels(rd), start, end);
      std::vector<float> w(b->nu), loss(b->rows), p(b->rows);
      xo_store_pull(ws, b->ukeys.data(), b->nu, w.data()); /* inserts zeros: ftrl.h:56 */
      if (model == 0) {
        xo_lr_loss(b, w.data(), loss.data(), p.data());
      } else {
        const int k = vs->dim;
        std::vector<float> v(b->nu * k), v_sum(b->rows);
        xo_store_pull(vs, b->ukeys.data(), b->nu, v.data());
        xo_fm_loss(b, k, w.data(), v.data(), loss.data(), p.data(), v_sum.data());
      }
      for (size_t r = 0; r < b->rows; ++r) {
        if (n >= cap) {
          xo_batch_free(b);
          xo_reader_close(rd);
          return -1;
        }
        labels_out[n] = b->labels[r];
        pctr_out[n] = p[r];
        ++n;
      }
      xo_batch_free(b);
    }
  }
  xo_reader_close(rd);
  return (long)n;
}
