// xf_io.cc — key hash, key-range owner rule, the block text reader, error plumbing.
//
// Replaces src/io/io.h + src/io/load_data_from_disk.{h,cc} of the reference for the one
// loader the workers call, LoadData::load_minibatch_hash_data_fread
// (load_data_from_disk.cc:103-210; callers lr_worker.cc:85,188, fm_worker.cc:110,258).
// Host-side by design (north_star keeps the libsvm-format io path on the host); must be
// bit-exact on keys, labels and on which rows fall in which block.
#include <errno.h>
#include <stdio.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <exception>
#include <mutex>
#include <functional>
#include <condition_variable>
#include <new>
#include <utility>
#include <string>
#include <thread>
#include <vector>

#include "xf_common.h"

namespace xf {

// Parser threads.  On a box whose cgroup grants fewer CPUs than it shows (16 of 256 on the GPU
// boxes) 64 threads burn the period's quota in bursts and get parked for the rest of it — but
// the parse is then limited by the quota itself (a 64 MiB block costs ~90 thread-ms), and
// bursting reaches that limit (4.8e6 examples/s end to end) where quota-many threads leave
// pipeline bubbles (3.2e6): measured both ways, tools/e2e_text.py.
static int g_parse_threads = 64;
int parse_threads() { return g_parse_threads; }
void set_parse_threads(int n) { g_parse_threads = n < 1 ? 1 : n; }

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace xf

extern "C" const char *xf_last_error(void) { return xf::g_err; }
extern "C" int xf_version(void) { return 100; }

extern "C" int xf_device_count(int *count) {
  XF_REQUIRE(count, "xf_device_count: null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return XF_OK;
}

// ------------------------------------------------------------------------------- hash
// std::hash<std::string> of libstdc++ (io.h:53): _Hash_bytes, a 64-bit Murmur-style mix
// with multiplier 0xc6a4a7935bd1e995 and seed 0xc70f6907; 8-byte little-endian words, the
// 1..7 tail bytes packed little-endian, two closing shift-mix rounds (v ^= v >> 47).
extern "C" uint64_t xf_hash_bytes(const void *ptr, size_t len) {
  const uint64_t m = 0xc6a4a7935bd1e995ull;
  const unsigned char *p = static_cast<const unsigned char *>(ptr);
  uint64_t h = 0xc70f6907ull ^ (len * m);
  size_t left = len;
  while (left >= 8) {
    uint64_t w;
    memcpy(&w, p, 8);
    w *= m;
    w ^= w >> 47;
    w *= m;
    h = (h ^ w) * m;
    p += 8;
    left -= 8;
  }
  if (left) {
    uint64_t w = 0;
    memcpy(&w, p, left);  // little-endian host: same packing as libstdc++'s load_bytes
    h = (h ^ w) * m;
  }
  h ^= h >> 47;
  h *= m;
  h ^= h >> 47;
  return h;
}

// keys of the decimal strings "start", "start+1", ... (synthetic-data generator helper:
// the reference's fids are decimal strings, e.g. data/small_train "2:1163:0.3651")
extern "C" int xf_hash_decimal_range(uint64_t start, size_t n, uint64_t *out) {
  XF_REQUIRE(out || n == 0, "xf_hash_decimal_range: null output");
  auto run = [=](size_t lo, size_t hi) {
    char buf[24], tmp[24];
    for (size_t i = lo; i < hi; ++i) {
      uint64_t v = start + i;
      int len = 0;
      do {
        tmp[len++] = (char)('0' + v % 10);
        v /= 10;
      } while (v);
      for (int j = 0; j < len; ++j) buf[j] = tmp[len - 1 - j];
      out[i] = xf_hash_bytes(buf, (size_t)len);
    }
  };
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 32) nt = 32;  // several ranks of one node generate their key tables at once
  if (nt < 2 || n < (1u << 16)) {
    run(0, n);
    return XF_OK;
  }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) th.emplace_back(run, n * t / nt, n * (t + 1) / nt);
  for (auto &x : th) x.join();
  return XF_OK;
}

// out[i] = hash of the decimal string of ids[i] (key spaces too large for a table of every
// hash: 10^9 fids drawn from a power law)
extern "C" int xf_hash_decimal_ids(const uint64_t *ids, size_t n, uint64_t *out) {
  XF_REQUIRE((ids && out) || n == 0, "xf_hash_decimal_ids: null argument");
  auto run = [=](size_t lo, size_t hi) {
    char buf[24], tmp[24];
    for (size_t i = lo; i < hi; ++i) {
      uint64_t v = ids[i];
      int len = 0;
      do {
        tmp[len++] = (char)('0' + v % 10);
        v /= 10;
      } while (v);
      for (int j = 0; j < len; ++j) buf[j] = tmp[len - 1 - j];
      out[i] = xf_hash_bytes(buf, (size_t)len);
    }
  };
  unsigned nt = std::thread::hardware_concurrency();
  if (nt > 32) nt = 32;
  if (nt < 2 || n < (1u << 16)) {
    run(0, n);
    return XF_OK;
  }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t) th.emplace_back(run, n * t / nt, n * (t + 1) / nt);
  for (auto &x : th) x.join();
  return XF_OK;
}

extern "C" uint32_t xf_shard_of(uint64_t key, uint32_t nshards) {
  if (nshards <= 1) return 0;
  const uint64_t s = key / (UINT64_MAX / nshards);
  return s >= nshards ? nshards - 1 : (uint32_t)s;
}

// ----------------------------------------------------------------------------- reader
// The arrays of a block are what the GPU key build uploads (80 MB of keys per 1e7-nonzero
// block): they live in page-locked host memory when there is a GPU, so that the upload is one
// DMA at PCIe rate instead of the driver's staged copy out of pageable memory (8-16 ms per
// block), and growing them does not zero-fill what is about to be overwritten.
namespace {
bool pinned_host_memory() {
  static const bool yes = [] {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
  }();
  return yes;
}

template <typename T>
struct BlockAlloc {
  typedef T value_type;
  BlockAlloc() = default;
  template <typename U>
  BlockAlloc(const BlockAlloc<U> &) {}
  T *allocate(size_t n) {
    void *p = nullptr;
    if (pinned_host_memory()) {
      if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) p = nullptr;
    } else {
      p = malloc(n * sizeof(T));
    }
    if (!p) throw std::bad_alloc();
    return (T *)p;
  }
  void deallocate(T *p, size_t) {
    if (pinned_host_memory()) (void)hipHostFree(p);
    else
      free(p);
  }
  template <typename U>
  void construct(U *p) {  // default-initialise: no zero fill on resize
    ::new ((void *)p) U;
  }
  template <typename U, typename... A>
  void construct(U *p, A &&... a) {
    ::new ((void *)p) U(std::forward<A>(a)...);
  }
  template <typename U>
  bool operator==(const BlockAlloc<U> &) const { return true; }
  template <typename U>
  bool operator!=(const BlockAlloc<U> &) const { return false; }
};
template <typename T>
using BlockVec = std::vector<T, BlockAlloc<T>>;
// resize with headroom: blocks of one file differ by a few rows, and a block that is one row
// longer than every block before it would otherwise re-pin a whole array (10-20 ms per 40 MB)
template <typename T>
static void block_resize(BlockVec<T> &v, size_t n) {
  if (n > v.capacity()) v.reserve(n + n / 8 + 64);
  v.resize(n);
}
}  // namespace

// what one parser thread produces for its run of lines
struct Piece {
  std::vector<uint64_t> keys, rowend;
  std::vector<int32_t> fgid, labels;
  const char *err = nullptr;
  bool hit_nul = false;
};

// A team of host threads that lives as long as the reader: run(n, f) calls f(0..n-1) on them
// and returns when all are done.  (Two rounds of 64 std::thread creations per block — parse,
// then place — were ~4 ms of a 6.5 ms block.)
class Team {
 public:
  ~Team() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    for (auto &t : th_) t.join();
  }
  void run(unsigned n, const std::function<void(unsigned)> &f) {
    if (n <= 1) {
      if (n) f(0);
      return;
    }
    while (th_.size() < n - 1) {
      const unsigned id = (unsigned)th_.size() + 1;
      th_.emplace_back([this, id] { loop(id); });
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      job_ = &f;
      n_ = n;
      left_ = n - 1;
      ++gen_;
    }
    cv_.notify_all();
    f(0);  // the caller is member 0
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [&] { return left_ == 0; });
    job_ = nullptr;
  }

 private:
  void loop(unsigned id) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(unsigned)> *f = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return quit_ || gen_ != seen; });
        if (quit_) return;
        seen = gen_;
        if (id < n_) f = job_;
      }
      if (!f) continue;
      (*f)(id);
      {
        std::lock_guard<std::mutex> lk(mu_);
        --left_;
      }
      done_.notify_one();
    }
  }
  std::vector<std::thread> th_;
  std::mutex mu_;
  std::condition_variable cv_, done_;
  const std::function<void(unsigned)> *job_ = nullptr;
  unsigned n_ = 0, left_ = 0;
  uint64_t gen_ = 0;
  bool quit_ = false;
};

struct xf_reader {
  FILE *fp = nullptr;
  // A regular file is mapped and parsed in place (no copy of the text through a stdio
  // buffer, which cost as much as the 64-thread parse); anything else is read with fread.
  const char *map = nullptr;
  size_t map_size = 0, pos = 0;
  std::string path;
  size_t cap = 0;
  std::vector<char> buf;
  size_t held = 0;  // bytes at the front of buf not yet parsed (carry + fresh read)
  BlockVec<uint64_t> rowptr, keys;
  BlockVec<int32_t> fgid, labels;
  // per-thread pieces, kept between blocks: fresh 100 MB vectors per block meant page faults
  // and munmap under 64 threads every time
  std::vector<Piece> pieces;
  Team team;
  Team copy_team;  // xf_reader_copy_text: on the producer's thread while `team` parses elsewhere
  // block cache (xf_reader_open_cached): either replaying `cfp`, or teeing into `tfp`
  FILE *cfp = nullptr, *tfp = nullptr;
  const char *cmap = nullptr;  // the cache file, mapped
  size_t cmap_size = 0, cpos = 0;
  std::string cache_path, tmp_path;
  uint64_t cache_blocks = 0, served = 0, teed = 0;
  uint64_t src_size = 0, src_mtime_ns = 0;
};

namespace {
// The binarized block cache: what xf_reader_next returned for every block of one pass over
// a text file at one block size, so that later passes skip the text (the
// reference re-parses every epoch, lr_worker.cc:184).  Blocks are byte-determined, so a
// cache is only valid for the block size and the exact source file it was made from.
//   header  : magic[8] "XFCSR001", cap_bytes, src_size, src_mtime_ns, nblocks   (u64 each)
//   block   : rows, nnz (u64), rowptr[rows+1] u64, keys[nnz] u64, fgid[nnz] i32, labels[rows] i32
// Written to <cache>.tmp.<pid> and renamed when the pass reached end of file.
const char kCacheMagic[8] = {'X', 'F', 'C', 'S', 'R', '0', '0', '1'};
struct CacheHeader {
  char magic[8];
  uint64_t cap, src_size, src_mtime_ns, nblocks;
};

bool source_stamp(const char *path, uint64_t *size, uint64_t *mtime_ns) {
  struct stat st;
  if (stat(path, &st) != 0) return false;
  *size = (uint64_t)st.st_size;
  *mtime_ns = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
  return true;
}

template <typename T>
bool put(FILE *f, const T *p, size_t n) { return n == 0 || fwrite(p, sizeof(T), n, f) == n; }
template <typename T>
bool get(FILE *f, T *p, size_t n) { return n == 0 || fread(p, sizeof(T), n, f) == n; }
}  // namespace

extern "C" int xf_reader_open(xf_reader **out, const char *path, size_t cap_bytes) {
  XF_REQUIRE(out && path, "xf_reader_open: null argument");
  XF_REQUIRE(cap_bytes >= 2, "xf_reader_open: block of %zu bytes", cap_bytes);
  FILE *fp = fopen(path, "r");
  if (!fp) return xf::set_error(XF_EIO, "open file %s error: %s", path, strerror(errno));
  xf_reader *r = new xf_reader;
  r->path = path;
  r->cap = cap_bytes;
  struct stat st;
  if (fstat(fileno(fp), &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
    void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(fp), 0);
    if (m != MAP_FAILED) {
      (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
      r->map = (const char *)m;
      r->map_size = (size_t)st.st_size;
      fclose(fp);
      fp = nullptr;
    }
  }
  r->fp = fp;
  if (fp) r->buf.resize(cap_bytes);
  *out = r;
  return XF_OK;
}

extern "C" int xf_reader_open_cached(xf_reader **out, const char *path, size_t cap_bytes,
                                     const char *cache_path, int *from_cache) {
  XF_REQUIRE(out && path && cache_path, "xf_reader_open_cached: null argument");
  if (from_cache) *from_cache = 0;
  uint64_t size = 0, mtime = 0;
  if (!source_stamp(path, &size, &mtime))
    return xf::set_error(XF_EIO, "open file %s error: %s", path, strerror(errno));
  if (FILE *c = fopen(cache_path, "rb")) {
    CacheHeader h;
    if (fread(&h, sizeof(h), 1, c) == 1 && !memcmp(h.magic, kCacheMagic, 8) &&
        h.cap == (uint64_t)cap_bytes && h.src_size == size && h.src_mtime_ns == mtime) {
      xf_reader *r = new xf_reader;
      r->path = path;
      r->cap = cap_bytes;
      r->cfp = c;
      r->cache_path = cache_path;
      r->cache_blocks = h.nblocks;
      {  // the blocks are copied out of a mapping of the file by several threads (one fread
         // through a stdio buffer moved 125 MB per block at ~3 GB/s: 40 ms)
        struct stat st;
        if (fstat(fileno(c), &st) == 0 && st.st_size > 0) {
          void *m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fileno(c), 0);
          if (m != MAP_FAILED) {
            r->cmap = (const char *)m;
            r->cmap_size = (size_t)st.st_size;
            r->cpos = sizeof(CacheHeader);
            (void)madvise(m, r->cmap_size, MADV_SEQUENTIAL);
          }
        }
      }
      if (from_cache) *from_cache = 1;
      *out = r;
      return XF_OK;
    }
    fclose(c);  // stale or foreign: rebuilt below
  }
  XF_TRY(xf_reader_open(out, path, cap_bytes));
  xf_reader *r = *out;
  r->cache_path = cache_path;
  r->tmp_path = r->cache_path + ".tmp." + std::to_string((long)getpid());
  r->src_size = size;
  r->src_mtime_ns = mtime;
  r->tfp = fopen(r->tmp_path.c_str(), "wb");
  if (r->tfp) {  // a cache that cannot be written is not an error: the text is still parsed
    CacheHeader h{};
    if (fwrite(&h, sizeof(h), 1, r->tfp) != 1) {
      fclose(r->tfp);
      r->tfp = nullptr;
      unlink(r->tmp_path.c_str());
    }
  }
  return XF_OK;
}

static void abandon_cache(xf_reader *r) {
  if (r->tfp) {
    fclose(r->tfp);
    r->tfp = nullptr;
    unlink(r->tmp_path.c_str());
  }
}

extern "C" int xf_reader_close(xf_reader *r) {
  if (!r) return XF_OK;
  abandon_cache(r);  // a pass that did not reach end of file leaves no cache
  if (r->map) munmap((void *)r->map, r->map_size);
  if (r->cmap) munmap((void *)r->cmap, r->cmap_size);
  if (r->fp) fclose(r->fp);
  if (r->cfp) fclose(r->cfp);
  delete r;
  return XF_OK;
}

// one block out of the cache file
static int next_from_cache(xf_reader *r, size_t *rows_out, size_t *nnz_out) {
  *rows_out = 0;
  if (nnz_out) *nnz_out = 0;
  r->rowptr.assign(1, 0);
  r->keys.clear();
  r->fgid.clear();
  r->labels.clear();
  if (r->served == r->cache_blocks) return XF_OK;
  uint64_t dims[2];
  if (r->cmap) {  // mapped cache: bounds-checked parallel copy into the (pinned) block arrays
    bool ok = r->cpos + 16 <= r->cmap_size;
    if (ok) memcpy(dims, r->cmap + r->cpos, 16);
    if (ok && (dims[0] > r->cap / 2 + 1 || dims[1] > r->cap))
      return xf::set_error(XF_EIO, "%s: block %llu claims %llu rows / %llu nonzeros for %zu-byte "
                           "blocks (corrupt block cache)", r->cache_path.c_str(),
                           (unsigned long long)r->served, (unsigned long long)dims[0],
                           (unsigned long long)dims[1], r->cap);
    const size_t R = ok ? (size_t)dims[0] : 0, N = ok ? (size_t)dims[1] : 0;
    const size_t need = 16 + (R + 1) * 8 + N * 8 + N * 4 + R * 4;
    ok = ok && r->cpos + need <= r->cmap_size;
    if (!ok)
      return xf::set_error(XF_EIO, "%s: truncated block cache (block %llu of %llu)",
                           r->cache_path.c_str(), (unsigned long long)r->served,
                           (unsigned long long)r->cache_blocks);
    block_resize(r->rowptr, R + 1);
    block_resize(r->keys, N);
    block_resize(r->fgid, N);
    block_resize(r->labels, R);
    const char *src = r->cmap + r->cpos + 16;
    struct Part {
      void *dst;
      const char *src;
      size_t bytes;
    } parts[4] = {{r->rowptr.data(), src, (R + 1) * 8},
                  {r->keys.data(), src + (R + 1) * 8, N * 8},
                  {r->fgid.data(), src + (R + 1) * 8 + N * 8, N * 4},
                  {r->labels.data(), src + (R + 1) * 8 + N * 12, R * 4}};
    const unsigned nt = (unsigned)std::max(1, std::min(xf::parse_threads(), 16));
    const size_t total = need - 16, slice = (total + nt - 1) / nt;
    auto copy = [&](unsigned t) {  // thread t copies bytes [t*slice, (t+1)*slice) of the block
      size_t lo = (size_t)t * slice, hi = std::min(total, lo + slice), at = 0;
      for (const Part &p : parts) {
        const size_t a = std::max(lo, at), b = std::min(hi, at + p.bytes);
        if (a < b) memcpy((char *)p.dst + (a - at), p.src + (a - at), b - a);
        at += p.bytes;
      }
    };
    if (nt == 1 || total < (1u << 20)) {
      for (unsigned t = 0; t < nt; ++t) copy(t);
    } else {
      std::vector<std::thread> th;
      for (unsigned t = 0; t < nt; ++t) th.emplace_back(copy, t);
      for (auto &x : th) x.join();
    }
    r->cpos += need;
    if (r->rowptr[0] != 0 || r->rowptr[R] != N ||
        !std::is_sorted(r->rowptr.begin(), r->rowptr.end()))
      return xf::set_error(XF_EIO, "%s: block %llu has inconsistent row offsets (corrupt block "
                           "cache)", r->cache_path.c_str(), (unsigned long long)r->served);
    ++r->served;
    *rows_out = R;
    if (nnz_out) *nnz_out = N;
    return XF_OK;
  }
  bool ok = get(r->cfp, dims, 2);
  // a block of `cap` text bytes holds at most cap/2 rows ("0\t") and cap/4 tokens ("a:b:c ", and
  // an empty token repeats the previous one): anything larger is a corrupt or foreign file
  if (ok && (dims[0] > r->cap / 2 + 1 || dims[1] > r->cap))
    return xf::set_error(XF_EIO, "%s: block %llu claims %llu rows / %llu nonzeros for %zu-byte "
                         "blocks (corrupt block cache)", r->cache_path.c_str(),
                         (unsigned long long)r->served, (unsigned long long)dims[0],
                         (unsigned long long)dims[1], r->cap);
  if (ok) {
    block_resize(r->rowptr, dims[0] + 1);
    block_resize(r->keys, dims[1]);
    block_resize(r->fgid, dims[1]);
    block_resize(r->labels, dims[0]);
    ok = get(r->cfp, r->rowptr.data(), r->rowptr.size()) && get(r->cfp, r->keys.data(), dims[1]) &&
         get(r->cfp, r->fgid.data(), dims[1]) && get(r->cfp, r->labels.data(), dims[0]);
  }
  if (!ok)
    return xf::set_error(XF_EIO, "%s: truncated block cache (block %llu of %llu)",
                         r->cache_path.c_str(), (unsigned long long)r->served,
                         (unsigned long long)r->cache_blocks);
  if (r->rowptr[0] != 0 || r->rowptr[dims[0]] != dims[1] ||
      !std::is_sorted(r->rowptr.begin(), r->rowptr.end()))
    return xf::set_error(XF_EIO, "%s: block %llu has inconsistent row offsets (corrupt block "
                         "cache)", r->cache_path.c_str(), (unsigned long long)r->served);
  ++r->served;
  *rows_out = dims[0];
  if (nnz_out) *nnz_out = dims[1];
  return XF_OK;
}

// tee the block just parsed; at end of file seal the cache
static void tee_block(xf_reader *r, size_t rows) {
  if (!r->tfp) return;
  bool ok = true;
  if (rows) {
    const uint64_t dims[2] = {(uint64_t)rows, (uint64_t)r->keys.size()};
    ok = put(r->tfp, dims, 2) && put(r->tfp, r->rowptr.data(), r->rowptr.size()) &&
         put(r->tfp, r->keys.data(), r->keys.size()) && put(r->tfp, r->fgid.data(), r->fgid.size()) &&
         put(r->tfp, r->labels.data(), r->labels.size());
    ++r->teed;
  } else {
    CacheHeader h;
    memcpy(h.magic, kCacheMagic, 8);
    h.cap = r->cap;
    h.src_size = r->src_size;
    h.src_mtime_ns = r->src_mtime_ns;
    h.nblocks = r->teed;
    ok = fseek(r->tfp, 0, SEEK_SET) == 0 && fwrite(&h, sizeof(h), 1, r->tfp) == 1;
    ok = (fclose(r->tfp) == 0) && ok;
    r->tfp = nullptr;
    if (ok) ok = rename(r->tmp_path.c_str(), r->cache_path.c_str()) == 0;
    if (!ok) unlink(r->tmp_path.c_str());
    return;
  }
  if (!ok) abandon_cache(r);
}

namespace {

// atof on the byte range [b, e).  Plain decimal integers (what fgid and most labels are) are
// converted directly — the same value atof gives — everything else goes through atof.
inline double field_atof(const char *b, const char *e, bool *ok) {
  if (e - b >= 1 && e - b <= 9) {
    uint32_t v = 0;
    const char *c = b;
    for (; c < e && *c >= '0' && *c <= '9'; ++c) v = v * 10 + (uint32_t)(*c - '0');
    if (c == e) return (double)v;
  }
  char tmp[48];
  const size_t n = (size_t)(e - b);
  if (n >= sizeof(tmp)) {
    *ok = false;
    return 0.0;
  }
  memcpy(tmp, b, n);
  tmp[n] = '\0';
  return atof(tmp);
}


// xf_hash_bytes for the parser's short fids: the 1..7 tail bytes come from one 8-byte load and a
// mask when those 8 bytes are inside the buffer (`safe_end`), instead of a variable-size memcpy
inline uint64_t hash_fid(const char *ptr, size_t len, const char *safe_end) {
  if (len >= 8 || ptr + 8 > safe_end) return xf_hash_bytes(ptr, len);
  const uint64_t m = 0xc6a4a7935bd1e995ull;
  uint64_t h = 0xc70f6907ull ^ (len * m);
  if (len) {
    uint64_t w;
    memcpy(&w, ptr, 8);
    w &= ~0ull >> (64 - 8 * len);
    h = (h ^ w) * m;
  }
  h ^= h >> 47;
  h *= m;
  h ^= h >> 47;
  return h;
}

// One contiguous run of whole lines (load_data_from_disk.cc:126-208).  `safe_end`: the end of
// the buffer the text sits in (bytes up to there may be read, not interpreted).
void parse_piece(const char *p, const char *end, bool last_piece, const char *safe_end,
                 Piece *out) {
  // one reservation per piece instead of reallocating under 64 threads' malloc contention:
  // a token is at least "0:0:0 " (6 bytes), a row at least "0\t0:0:0\n" (8 bytes)
  const size_t bytes = (size_t)(end - p);
  out->keys.reserve(bytes / 6 + 16);
  out->fgid.reserve(bytes / 6 + 16);
  out->labels.reserve(bytes / 8 + 16);
  out->rowend.reserve(bytes / 8 + 16);
  while (p < end) {
    if (*p == '\0') {
      out->hit_nul = true;
      return;
    }
    const char *eol = (const char *)memchr(p, '\n', (size_t)(end - p));
    if (!eol) eol = end;
    const char *tab = (const char *)memchr(p, '\t', (size_t)(eol - p));
    if (!tab) {
      out->err = "no '\\t' after the label";
      return;
    }
    bool ok = true;
    const float y_tmp = (float)field_atof(p, tab, &ok);  // :129
    if (!ok) {
      out->err = "label field too long";
      return;
    }
    out->labels.push_back(y_tmp > 0.0000001 ? 1 : 0);  // :131-134
    const char *t = tab + 1;
    const size_t row_first = out->keys.size();
    // An EMPTY token (two blanks in a row, or a blank right before the block terminator) makes
    // the reference push its `keyval` again without parsing anything into it (:178-196 with an
    // empty field, :160-177 at the terminator): the row gets a duplicate of its previous
    // token.  A single blank before '\n' is not a token (:138 ends the row).  Without a
    // previous token in the row the value is stale or uninitialised there: rejected here.
    while (t < eol) {
      {
        // the common token, digits ':' fid ':' val: one pass, no second look at any byte.
        // Anything else (empty token, no colons, a non-digit or long fgid) takes the general
        // path below, which yields the same values for the tokens this path accepts.
        const char *q = t;
        uint32_t fgv = 0;
        while (q < eol && (unsigned)(*q - '0') <= 9u) fgv = fgv * 10 + (uint32_t)(*q++ - '0');
        if (q < eol && *q == ':' && q > t && q - t <= 9) {
          const char *f = q + 1;
          q = f;
          while (q < eol && *q != ':' && *q != ' ') ++q;
          if (q < eol && *q == ':') {
            const char *c2f = q;
            while (q < eol && *q != ' ') ++q;
            out->fgid.push_back((int32_t)fgv);
            out->keys.push_back(hash_fid(f, (size_t)(c2f - f), safe_end));
            t = q + 1;
            continue;
          }
        }
      }
      const char *te = t;
      const char *c1 = nullptr, *c2 = nullptr;
      while (te < eol && *te != ' ') {
        if (*te == ':') {
          if (!c1) c1 = te;
          else if (!c2)
            c2 = te;
        }
        ++te;
      }
      if (te == t) {  // empty token
        if (out->keys.size() == row_first) {
          out->labels.pop_back();
          out->err = "empty first token";
          return;
        }
        out->keys.push_back(out->keys.back());
        out->fgid.push_back(out->fgid.back());
        t = te + 1;
        continue;
      }
      if (!c1 || !c2) {  // the reference would scan past the terminator here
        out->labels.pop_back();
        out->err = "token without fgid:fid:val";
        return;
      }
      const double fg = field_atof(t, c1, &ok);  // :149
      if (!ok) {
        out->labels.pop_back();
        out->err = "fgid field too long";
        return;
      }
      out->fgid.push_back((int32_t)fg);
      out->keys.push_back(xf_hash_bytes(c1 + 1, (size_t)(c2 - (c1 + 1))));  // :151
      t = te + 1;
    }
    if (eol == end && last_piece) {  // this line is closed by the block terminator, not by '\n'
      if (eol > tab + 1 && eol[-1] == ' ' && out->keys.size() > row_first) {
        out->keys.push_back(out->keys.back());
        out->fgid.push_back(out->fgid.back());
      } else if (eol == tab + 1 || (eol[-1] == ' ' && out->keys.size() == row_first)) {
        out->labels.pop_back();
        out->err = "row without tokens at the end of the block";
        return;
      }
    }
    out->rowend.push_back(out->keys.size());
    p = eol + 1;
  }
}

}  // namespace

// a parsed block owned by the caller: xf_reader_next_into moves the reader's arrays into it, so
// the block stays valid while the reader parses the next one (the worker's prefetch thread)
struct xf_block {
  BlockVec<uint64_t> rowptr, keys;
  BlockVec<int32_t> fgid, labels;
};

extern "C" int xf_block_create(xf_block **out) {
  XF_REQUIRE(out, "xf_block_create: null argument");
  *out = new xf_block;
  return XF_OK;
}

extern "C" int xf_block_destroy(xf_block *b) {
  delete b;
  return XF_OK;
}

extern "C" int xf_reader_next_into(xf_reader *r, xf_block *blk, size_t *rows_out, size_t *nnz_out,
                                   const uint64_t **rowptr, const uint64_t **keys,
                                   const int32_t **fgid, const int32_t **labels) {
  XF_REQUIRE(r && blk && rows_out, "xf_reader_next_into: null argument");
  XF_TRY(xf_reader_next(r, rows_out, nnz_out, nullptr, nullptr, nullptr, nullptr));
  blk->rowptr.swap(r->rowptr);
  blk->keys.swap(r->keys);
  blk->fgid.swap(r->fgid);
  blk->labels.swap(r->labels);
  if (rowptr) *rowptr = blk->rowptr.data();
  if (keys) *keys = blk->keys.data();
  if (fgid) *fgid = blk->fgid.data();
  if (labels) *labels = blk->labels.data();
  return XF_OK;
}

static int reader_next(xf_reader *r, size_t *rows_out, size_t *nnz_out, const uint64_t **rowptr,
                       const uint64_t **keys, const int32_t **fgid, const int32_t **labels);

// no exception crosses the C boundary: the block arrays are (pinned) host allocations
extern "C" int xf_reader_next(xf_reader *r, size_t *rows_out, size_t *nnz_out,
                              const uint64_t **rowptr, const uint64_t **keys,
                              const int32_t **fgid, const int32_t **labels) {
  XF_REQUIRE(r && rows_out, "xf_reader_next: null argument");
  try {
    return reader_next(r, rows_out, nnz_out, rowptr, keys, fgid, labels);
  } catch (const std::exception &e) {
    return xf::set_error(XF_EIO, "xf_reader_next: %s (out of host memory for a block?)", e.what());
  }
}

// The block rule (load_data_from_disk.cc:104-121): the reference fills a cap-byte buffer up to
// cap-1 bytes; a buffer that filled completely is cut after its last newline and the rest
// carried; a short read means end of file and everything is parsed.  *base: the block's first
// byte (the mapping, or the read buffer after this call's fread), *text bytes of it are parsed,
// *take bytes consumed.
static int block_extent(xf_reader *r, const char **base_out, size_t *text_out, size_t *take_out) {
  const char *base;
  if (r->map) {  // the next cap-1 bytes of the mapping == carried tail + fresh read
    base = r->map + r->pos;
    r->held = std::min(r->map_size - r->pos, r->cap - 1);
  } else {
    char *buf = r->buf.data();
    r->held += fread(buf + r->held, 1, r->cap - 1 - r->held, r->fp);
    base = buf;
  }
  size_t take = r->held;   // bytes consumed by this block
  size_t text = r->held;   // bytes of text to parse
  if (r->held == r->cap - 1) {
    size_t cut = r->held;
    while (cut > 0 && base[cut - 1] != '\n' && base[cut - 1] != (char)EOF) --cut;
    if (cut == 0)
      return xf::set_error(XF_EPARSE, "%s: a line does not fit in a %zu-byte block",
                           r->path.c_str(), r->cap);
    take = cut;
    text = cut - 1;  // the newline is replaced by the terminator (:116)
  }
  *base_out = base;
  *text_out = text;
  *take_out = take;
  return XF_OK;
}

// [base, base + text) parsed into the reader's arrays.  Lines are independent, so a large
// block is cut at newlines into one piece per thread; the pieces are concatenated in order
// (bit-identical to a serial pass).  safe_end: bytes up to there may be READ beyond a token.
static int parse_text(xf_reader *r, const char *base, size_t text, const char *safe_end) {
  r->rowptr.assign(1, 0);
  r->keys.clear();
  r->fgid.clear();
  r->labels.clear();
  const char *p0 = base, *end = base + text;
  unsigned nt = std::thread::hardware_concurrency();
  if (nt == 0) nt = 1;
  if (nt > (unsigned)xf::parse_threads()) nt = (unsigned)xf::parse_threads();
  const size_t min_piece = 256 << 10;
  if ((size_t)(end - p0) / min_piece + 1 < nt) nt = (unsigned)((size_t)(end - p0) / min_piece + 1);
  std::vector<const char *> cut(nt + 1, end);
  cut[0] = p0;
  for (unsigned t = 1; t < nt; ++t) {
    const char *c = p0 + (size_t)(end - p0) * t / nt;
    if (c < cut[t - 1]) c = cut[t - 1];
    while (c < end && *c != '\n') ++c;  // a piece ends after a newline
    cut[t] = c < end ? c + 1 : end;
  }
  if (r->pieces.size() < nt) r->pieces.resize(nt);
  std::vector<Piece> &pieces = r->pieces;
  for (unsigned t = 0; t < nt; ++t) {
    Piece &pc = pieces[t];
    pc.keys.clear();
    pc.rowend.clear();
    pc.fgid.clear();
    pc.labels.clear();
    pc.err = nullptr;
    pc.hit_nul = false;
  }
  r->team.run(nt, [&](unsigned t) {
    parse_piece(cut[t], cut[t + 1], cut[t + 1] == end, safe_end, &pieces[t]);
  });
  // offsets of every piece in the block's arrays, then a parallel copy
  std::vector<size_t> key_off(nt + 1, 0), row_off(nt + 1, 0);
  unsigned used = nt;
  for (unsigned t = 0; t < nt; ++t) {
    Piece &pc = pieces[t];
    if (pc.err) {
      size_t row = row_off[t] + pc.labels.size();
      return xf::set_error(XF_EPARSE, "%s: %s in row %zu of the block", r->path.c_str(), pc.err,
                           row);
    }
    key_off[t + 1] = key_off[t] + pc.keys.size();
    row_off[t + 1] = row_off[t] + pc.labels.size();
    if (pc.hit_nul) {  // the reference stops at a NUL byte (:126)
      used = t + 1;
      break;
    }
  }
  const size_t nkeys = key_off[used], nrows = row_off[used];
  block_resize(r->keys, nkeys);
  block_resize(r->fgid, nkeys);
  block_resize(r->labels, nrows);
  block_resize(r->rowptr, nrows + 1);
  r->rowptr[0] = 0;
  auto place = [&](unsigned t) {
    Piece &pc = pieces[t];
    if (!pc.keys.empty()) {
      memcpy(&r->keys[key_off[t]], pc.keys.data(), pc.keys.size() * sizeof(uint64_t));
      memcpy(&r->fgid[key_off[t]], pc.fgid.data(), pc.fgid.size() * sizeof(int32_t));
    }
    if (!pc.labels.empty())
      memcpy(&r->labels[row_off[t]], pc.labels.data(), pc.labels.size() * sizeof(int32_t));
    for (size_t i = 0; i < pc.rowend.size(); ++i)
      r->rowptr[row_off[t] + i + 1] = key_off[t] + pc.rowend[i];
  };
  r->team.run(used, place);
  return XF_OK;
}

static int reader_next(xf_reader *r, size_t *rows_out, size_t *nnz_out, const uint64_t **rowptr,
                       const uint64_t **keys, const int32_t **fgid, const int32_t **labels) {
  if (r->cfp) {
    XF_TRY(next_from_cache(r, rows_out, nnz_out));
    if (rowptr) *rowptr = r->rowptr.data();
    if (keys) *keys = r->keys.data();
    if (fgid) *fgid = r->fgid.data();
    if (labels) *labels = r->labels.data();
    return XF_OK;
  }
  const char *base = nullptr;
  size_t text = 0, take = 0;
  XF_TRY(block_extent(r, &base, &text, &take));
  // bytes that may be READ beyond a token (the short-fid hash loads 8 at once): the mapping or
  // the read buffer
  const char *safe_end = r->map ? r->map + r->map_size : r->buf.data() + r->buf.size();
  XF_TRY(parse_text(r, base, text, safe_end));
  if (r->map) {
    r->pos += take;
    r->held = 0;
  } else {
    if (take < r->held) memmove(r->buf.data(), r->buf.data() + take, r->held - take);
    r->held -= take;
  }
  tee_block(r, r->labels.size());
  *rows_out = r->labels.size();
  if (nnz_out) *nnz_out = r->keys.size();
  if (rowptr) *rowptr = r->rowptr.data();
  if (keys) *keys = r->keys.data();
  if (fgid) *fgid = r->fgid.data();
  if (labels) *labels = r->labels.data();
  return XF_OK;
}

// ---- the block as TEXT (xf_ingest.hip tokenises it on the GPU) --------------------------------
extern "C" int xf_reader_mapped(xf_reader *r, int *yes) {
  XF_REQUIRE(r && yes, "xf_reader_mapped: null argument");
  *yes = r->map && !r->cfp && !r->tfp ? 1 : 0;
  return XF_OK;
}

// The next block's text by the block rule above, without parsing it: *text points into the
// mapping, *len bytes (0 at the end of the file).  Mapped files without a block cache only.
// Nothing moves until xf_reader_skip_text; a xf_reader_next* call instead parses this very block
// on the host and moves on.
extern "C" int xf_reader_peek_text(xf_reader *r, const char **text, size_t *len) {
  XF_REQUIRE(r && text && len, "xf_reader_peek_text: null argument");
  XF_REQUIRE(r->map && !r->cfp && !r->tfp,
             "xf_reader_peek_text: a mapped text file without a block cache only");
  const char *base = nullptr;
  size_t t = 0, take = 0;
  XF_TRY(block_extent(r, &base, &t, &take));
  *text = base;
  *len = t;
  return XF_OK;
}

extern "C" int xf_reader_skip_text(xf_reader *r) {
  XF_REQUIRE(r && r->map && !r->cfp && !r->tfp, "xf_reader_skip_text: bad reader");
  const char *base = nullptr;
  size_t t = 0, take = 0;
  XF_TRY(block_extent(r, &base, &t, &take));
  r->pos += take;
  r->held = 0;
  return XF_OK;
}

// the peeked block's text copied to dst (pinned staging memory) by `threads` host threads
extern "C" int xf_reader_copy_text(xf_reader *r, char *dst, size_t cap, size_t *len,
                                   int threads) {
  XF_REQUIRE(r && dst && len, "xf_reader_copy_text: null argument");
  const char *text = nullptr;
  XF_TRY(xf_reader_peek_text(r, &text, len));
  XF_REQUIRE(*len <= cap, "xf_reader_copy_text: a block of %zu bytes, room for %zu", *len, cap);
  const size_t n = *len;
  unsigned nt = (unsigned)std::max(1, std::min(threads, 64));
  if (n / (1 << 20) + 1 < nt) nt = (unsigned)(n / (1 << 20) + 1);
  r->copy_team.run(nt, [&](unsigned t) {
    const size_t lo = n * t / nt, hi = n * (t + 1) / nt;
    if (hi > lo) memcpy(dst + lo, text + lo, hi - lo);
  });
  return XF_OK;
}

// A block's text in memory (whole lines, as xf_reader_peek_text delimits them) parsed on the
// host into `blk`: what xf_reader_next_into yields for that block.  For the blocks the GPU
// tokeniser hands back.
extern "C" int xf_reader_parse_text(xf_reader *r, const char *text, size_t len, xf_block *blk,
                                    size_t *rows_out, size_t *nnz_out, const uint64_t **rowptr,
                                    const uint64_t **keys, const int32_t **fgid,
                                    const int32_t **labels) {
  XF_REQUIRE(r && blk && rows_out && (text || len == 0), "xf_reader_parse_text: null argument");
  try {
    XF_TRY(parse_text(r, text, len, text + len));
  } catch (const std::exception &e) {
    return xf::set_error(XF_EIO, "xf_reader_parse_text: %s", e.what());
  }
  *rows_out = r->labels.size();
  if (nnz_out) *nnz_out = r->keys.size();
  blk->rowptr.swap(r->rowptr);
  blk->keys.swap(r->keys);
  blk->fgid.swap(r->fgid);
  blk->labels.swap(r->labels);
  if (rowptr) *rowptr = blk->rowptr.data();
  if (keys) *keys = blk->keys.data();
  if (fgid) *fgid = blk->fgid.data();
  if (labels) *labels = blk->labels.data();
  return XF_OK;
}
