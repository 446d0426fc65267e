// xf_modelfile.cc — the model file: key-sorted (key, w, n, z) dumps of the worker's tables.
// The reference never saves its model; this is the xf_table_export / import parity
// hook with a file format around it, and what the sharded checkpoint is made of: one such file
// per shard, and a reader that keeps only the keys a given shard owns (so a checkpoint of N
// shards loads into any other number).
//   "XFAMD001" | u64 ntables | per table: u64 nkeys, u64 dim, keys u64[n], w, n, z f32[n*dim]
#include <stdio.h>
#include <string.h>

#include <vector>

#include "xf_common.h"

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
    return xf::set_error(XF_EIO, "model file: short write");
  return XF_OK;
}

int load_table(FILE *f, xf_table *t, int dim, uint32_t shard, uint32_t nshards) {
  uint64_t hdr[2];
  if (fread(hdr, 8, 2, f) != 2) return xf::set_error(XF_EIO, "model file: truncated");
  if ((int)hdr[1] != dim)
    return xf::set_error(XF_EINVAL, "model file has dim %llu, the table has %d",
                         (unsigned long long)hdr[1], dim);
  const size_t n = (size_t)hdr[0];
  {  // the header's count must fit what is left of the file (a corrupt or foreign file must
     // not turn into a giant allocation)
    const long here = ftell(f);
    fseek(f, 0, SEEK_END);
    const long end = ftell(f);
    fseek(f, here, SEEK_SET);
    if (here < 0 || end < here ||
        (unsigned long long)n * (8ull + 12ull * dim) > (unsigned long long)(end - here))
      return xf::set_error(XF_EIO, "model file claims %zu keys but holds %ld bytes", n,
                           end - here);
  }
  std::vector<uint64_t> keys(n);
  std::vector<float> w(n * dim), nn(n * dim), z(n * dim);
  if (fread(keys.data(), 8, n, f) != n || fread(w.data(), 4, n * dim, f) != n * dim ||
      fread(nn.data(), 4, n * dim, f) != n * dim || fread(z.data(), 4, n * dim, f) != n * dim)
    return xf::set_error(XF_EIO, "model file: truncated");
  size_t m = n;
  if (nshards > 1) {  // keep the keys this shard owns (keys are sorted: one contiguous range)
    m = 0;
    for (size_t i = 0; i < n; ++i) {
      if (xf_shard_of(keys[i], nshards) != shard) continue;
      keys[m] = keys[i];
      for (int j = 0; j < dim; ++j) {
        w[m * dim + j] = w[i * dim + j];
        nn[m * dim + j] = nn[i * dim + j];
        z[m * dim + j] = z[i * dim + j];
      }
      ++m;
    }
  }
  if (m == 0) return XF_OK;
  uint64_t cap = 0, have = 0;
  XF_TRY(xf_table_capacity(t, &cap));
  XF_TRY(xf_table_size(t, &have));
  if ((have + m) * 10 > cap * 6) XF_TRY(xf_table_reserve(t, (have + m) * 2 + 1024));
  return xf_table_import(t, keys.data(), m, w.data(), nn.data(), z.data());
}
}  // namespace

namespace xf {

int model_write(const char *path, xf_table *tw, xf_table *tv, int k) {
  FILE *f = fopen(path, "wb");
  if (!f) return set_error(XF_EIO, "cannot open %s for writing", path);
  const uint64_t nt = tv ? 2 : 1;
  int rc = fwrite(kMagic, 1, 8, f) == 8 && fwrite(&nt, 8, 1, f) == 1
               ? XF_OK
               : set_error(XF_EIO, "%s: short write", path);
  if (rc == XF_OK) rc = save_table(f, tw, 1);
  if (rc == XF_OK && tv) rc = save_table(f, tv, k);
  if (fclose(f) != 0 && rc == XF_OK) rc = set_error(XF_EIO, "%s: close failed", path);
  return rc;
}

// import into (tw, tv) the keys of the file that shard `shard` of `nshards` owns
int model_read(const char *path, xf_table *tw, xf_table *tv, int k, uint32_t shard,
               uint32_t nshards) {
  FILE *f = fopen(path, "rb");
  if (!f) return set_error(XF_EIO, "cannot open %s", path);
  char magic[8];
  uint64_t nt = 0;
  int rc = XF_OK;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, kMagic, 8) != 0 || fread(&nt, 8, 1, f) != 1)
    rc = set_error(XF_EINVAL, "%s is not an xflow_amd model file", path);
  if (rc == XF_OK && nt != (tv ? 2u : 1u))
    rc = set_error(XF_EINVAL, "%s holds %llu table(s), the worker has %d", path,
                   (unsigned long long)nt, tv ? 2 : 1);
  if (rc == XF_OK) rc = load_table(f, tw, 1, shard, nshards);
  if (rc == XF_OK && tv) rc = load_table(f, tv, k, shard, nshards);
  fclose(f);
  return rc;
}

}  // namespace xf
