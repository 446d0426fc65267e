// xf_ingest.hip — the libsvm-style text of one block tokenised and hashed ON THE GPU (gfx950):
// LoadData::load_minibatch_hash_data_fread's token loop (src/io/load_data_from_disk.cc:126-208
// of /root/reference) and std::hash<std::string> of every fid (src/io/io.h:53) as byte-parallel
// kernels.  north_star keeps the libsvm io path on the host and the multi-threaded parser of
// xf_io.cc IS that path; this is the SURVEY 8(f) rank-1 "next" row taken one step further: at
// 16 host threads the text of a 10^7-nonzero minibatch (140 MB) parses at 4.7e6 examples/s, 1 %
// of the rate the step consumes them at, while the same bytes cross PCIe in ~3 ms.
//
// What the GPU takes is the COMMON SHAPE of a block and nothing else — every line
//     ('0' | '1') '\t' token (' ' token)* '\n'        token = field0 ':' fid ':' rest
// with single blanks, no empty token, no blank before the line's end, no NUL byte, one tab per
// line, field0 of at most 16 bytes, fid of at most 32 — for which the reference's parser
// yields: label = the digit (atof("1") > 1e-7), one key per token = _Hash_bytes(fid), rows in
// file order.  (Every other byte is an ordinary byte to the reference and to this: the CR of a
// CR LF file — the reference's own sample data is one — ends up in the last token's third
// field, which nobody reads.)  Anything else (a label like "0.5", an empty token — which the
// reference turns into a DUPLICATE of the previous token —, a NUL byte — where the reference
// stops reading the block —, a token without two colons ...) raises a flag and the caller parses that block with xf_reader_next on the host: the
// quirks live in one place, and a block is either wholly the GPU's or wholly the host's.
// Results for accepted blocks are bit for bit the host parser's (tests/test_gpu_ingest.py: the
// reference's sample files, the golden parses, a fuzz over well- and ill-formed blocks).
//
//   k_tok_count   per workgroup span of the text: newlines, separators ('\t' and ' ': one
//                 precedes every token)
//   k_tok_scan    the spans' first row / first token
//   k_tok_emit    per 4 KiB tile (in LDS with a halo): every separator's token parsed from LDS
//                 — first and second ':' — and its fid hashed; rowptr at the newlines, labels
//                 at the line starts; the shape checks
// Byte work, HBM-bound (text in once per pass, 8 bytes of key out per token), no MFMA.
#include <hip/hip_runtime.h>

#include <string.h>

#include <algorithm>

#include "xf_common.h"

namespace {

constexpr int kTok = 256;                   // threads per workgroup
constexpr uint32_t kTokB = 16;              // bytes per thread and tile
constexpr uint32_t kTile = kTok * kTokB;    // 4 KiB of text per tile
constexpr uint32_t kHalo = 64;              // bytes read past a tile (a token's head)
constexpr uint32_t kMaxField0 = 16, kMaxFid = 32;
constexpr uint32_t kMaxWG = 1024;

struct TokCounts {
  uint32_t rows, nnz, bad, pad;
};

__device__ __forceinline__ bool is_sep(uint32_t c) { return c == ' ' || c == '\t'; }

// _Hash_bytes(p, len, 0xc70f6907) over bytes in LDS (xf_io.cc: xf_hash_bytes)
__device__ __forceinline__ uint64_t hash_lds(const uint8_t *p, uint32_t len) {
  const uint64_t m = 0xc6a4a7935bd1e995ull;
  uint64_t h = 0xc70f6907ull ^ ((uint64_t)len * m);
  uint32_t i = 0;
  for (; i + 8 <= len; i += 8) {
    uint64_t w = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) w |= (uint64_t)p[i + k] << (8 * k);
    w *= m;
    w ^= w >> 47;
    w *= m;
    h = (h ^ w) * m;
  }
  if (i < len) {
    uint64_t w = 0;
    for (uint32_t k = 0; i + k < len; ++k) w |= (uint64_t)p[i + k] << (8 * k);
    h = (h ^ w) * m;
  }
  h ^= h >> 47;
  h *= m;
  h ^= h >> 47;
  return h;
}

// newlines | separators << 16 in a thread's 16 bytes
__device__ __forceinline__ uint32_t count16(uint4 v) {
  uint32_t nl = 0, sp = 0;
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const uint32_t c = (w[k] >> (8 * b)) & 0xFFu;
      nl += c == '\n' ? 1u : 0u;
      sp += is_sep(c) ? 1u : 0u;
    }
  return nl | (sp << 16);
}

// text: n bytes, readable (and '\n'-padded) up to the next multiple of kTile + kHalo
__global__ void __launch_bounds__(kTok)
k_tok_count(const uint8_t *__restrict__ text, uint32_t n, uint32_t span,
            uint2 *__restrict__ wgcnt) {
  __shared__ uint32_t wsum[2][kTok / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t b0 = blockIdx.x * span, b1 = min(b0 + span, n);
  uint32_t nl = 0, sp = 0;
  for (uint32_t t0 = b0; t0 < b1; t0 += kTile) {
    const uint32_t i0 = t0 + tid * kTokB;
    if (i0 >= b1) continue;
    const uint4 v = *reinterpret_cast<const uint4 *>(text + i0);
    uint32_t c = count16(v);
    if (i0 + kTokB > n) {  // the last, partial 16 bytes: the padding is not text
      c = 0;
      for (uint32_t k = 0; i0 + k < n; ++k) {
        const uint32_t ch = text[i0 + k];
        c += (ch == '\n' ? 1u : 0u) | ((is_sep(ch) ? 1u : 0u) << 16);
      }
    }
    nl += c & 0xFFFFu;
    sp += c >> 16;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    nl += __shfl_xor(nl, o);
    sp += __shfl_xor(sp, o);
  }
  if (lane == 0) {
    wsum[0][wave] = nl;
    wsum[1][wave] = sp;
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t a = 0, b = 0;
    for (int w = 0; w < kTok / 64; ++w) {
      a += wsum[0][w];
      b += wsum[1][w];
    }
    wgcnt[blockIdx.x] = make_uint2(a, b);
  }
}

// wgcnt -> exclusive prefix; the totals and the capacity check
__global__ void __launch_bounds__(kMaxWG)
k_tok_scan(uint2 *__restrict__ wgcnt, uint32_t nwg, uint32_t cap_rows, uint32_t cap_nnz,
           TokCounts *__restrict__ out) {
  __shared__ uint32_t wsum[2][kMaxWG / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint2 x = tid < nwg ? wgcnt[tid] : make_uint2(0u, 0u);
  uint32_t a = x.x, b = x.y;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t ta = __shfl_up(a, o), tb = __shfl_up(b, o);
    if ((int)lane >= o) {
      a += ta;
      b += tb;
    }
  }
  if (lane == 63) {
    wsum[0][wave] = a;
    wsum[1][wave] = b;
  }
  __syncthreads();
  uint32_t ba = 0, bb = 0, ta = 0, tb = 0;
  for (uint32_t w = 0; w < kMaxWG / 64; ++w) {
    if (w < wave) {
      ba += wsum[0][w];
      bb += wsum[1][w];
    }
    ta += wsum[0][w];
    tb += wsum[1][w];
  }
  if (tid < nwg) wgcnt[tid] = make_uint2(ba + a - x.x, bb + b - x.y);
  if (tid == 0) {
    out->rows = ta;
    out->nnz = tb;
    out->bad = (ta > cap_rows || tb > cap_nnz) ? 1u : 0u;
  }
}

__global__ void __launch_bounds__(kTok)
k_tok_emit(const uint8_t *__restrict__ text, uint32_t n, uint32_t span,
           const uint2 *__restrict__ wgbase, uint32_t cap_rows, uint32_t cap_nnz,
           uint64_t *__restrict__ keys, uint32_t *__restrict__ rowptr,
           int32_t *__restrict__ labels, TokCounts *__restrict__ out) {
  // [16 bytes: the two bytes before the tile at their end | the tile | the halo]
  __shared__ __attribute__((aligned(16))) uint8_t lt[16 + kTile + kHalo];
  __shared__ uint32_t wsum[kTok / 64];
  __shared__ uint32_t run_nl, run_sp, s_full;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // more rows or tokens than the output arrays hold (k_tok_scan's verdict): nothing to emit.  One
  // thread reads the word — other workgroups of this launch OR their shape verdicts into it, and
  // wavefronts that read it at different times must not part ways ahead of the barriers below
  if (tid == 0) s_full = out->bad;
  __syncthreads();
  if (s_full) return;
  const uint32_t b0 = blockIdx.x * span, b1 = min(b0 + span, n);
  if (tid == 0) {
    const uint2 base = wgbase[blockIdx.x];
    run_nl = base.x;
    run_sp = base.y;
  }
  if (blockIdx.x == 0 && tid == 0) rowptr[0] = 0;
  uint8_t *T = lt + 16;  // T[-1], T[-2]: the bytes before the tile
  uint32_t bad = 0;
  for (uint32_t t0 = b0; t0 < b1; t0 += kTile) {
    __syncthreads();  // (the previous tile's LDS reads, run_nl / run_sp)
    // the tile, the byte before it ('\n' before the first byte of the text) and the halo
    *reinterpret_cast<uint4 *>(T + tid * kTokB) =
        *reinterpret_cast<const uint4 *>(text + t0 + tid * kTokB);
    if (tid < kHalo / 16)
      *reinterpret_cast<uint4 *>(T + kTile + tid * 16) =
          *reinterpret_cast<const uint4 *>(text + t0 + kTile + tid * 16);
    if (tid == kTok - 1) {  // (before the text: as if a line had just ended)
      T[-1] = t0 >= 1 ? text[t0 - 1] : (uint8_t)'\n';
      T[-2] = t0 >= 2 ? text[t0 - 2] : (uint8_t)'\n';
    }
    __syncthreads();
    const uint32_t i0 = tid * kTokB;       // this thread's bytes: T[i0 .. i0 + 16)
    const uint32_t lim = b1 - t0;          // bytes of the tile that are text of this span
    uint32_t c = 0;
#pragma unroll
    for (uint32_t k = 0; k < kTokB; ++k) {
      const uint32_t ch = T[i0 + k];
      if (i0 + k < lim) c += (ch == '\n' ? 1u : 0u) | ((is_sep(ch) ? 1u : 0u) << 16);
    }
    // exclusive prefix of (newlines | separators << 16) over the workgroup
    uint32_t inc = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o);
      if ((int)lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t pre = inc - c, tot = 0;
    for (uint32_t w = 0; w < kTok / 64; ++w) {
      if (w < wave) pre += wsum[w];
      tot += wsum[w];
    }
    uint32_t nl = run_nl + (pre & 0xFFFFu), sp = run_sp + (pre >> 16);
#pragma unroll 1
    for (uint32_t k = 0; k < kTokB; ++k) {
      const int i = (int)(i0 + k);
      if ((uint32_t)i >= lim) break;
      const uint32_t ch = T[i], prev = T[i - 1], prev2 = T[i - 2], next = T[i + 1];
      bad |= ch == 0u ? 1u : 0u;  // (the reference stops at a NUL: the host parser's)
      if (prev == '\n') {  // a line starts: '0' | '1', then the tab (an empty line fails here)
        bad |= (ch != '0' && ch != '1') ? 1u : 0u;
        bad |= next != '\t' ? 1u : 0u;
        if (nl < cap_rows) labels[nl] = ch == '1' ? 1 : 0;
      }
      if (ch == '\t') bad |= prev2 != '\n' ? 1u : 0u;  // the tab is a line's second byte only
      if (ch == ' ') bad |= (is_sep(prev) || prev == '\n') ? 1u : 0u;
      if (is_sep(ch)) {
        // single blanks, no blank before a line's end, no row without tokens
        bad |= (next == ' ' || next == '\n' || next == '\t') ? 1u : 0u;
        // the token after this separator: field0 ':' fid ':' ... (first and second colon
        // before the token ends: load_data_from_disk.cc:147-153)
        uint32_t j = (uint32_t)i + 1, c1 = 0, c2 = 0;
        const uint32_t stop = j + kMaxField0 + 1 + kMaxFid + 1;  // < kTile + kHalo
        for (; j < stop; ++j) {
          const uint32_t x = T[j];
          if (x == ' ' || x == '\n') break;
          if (x == ':') {
            if (!c1) c1 = j;
            else {
              c2 = j;
              break;
            }
          }
        }
        const bool ok = c1 && c2 && c1 - ((uint32_t)i + 1) <= kMaxField0 && c2 - c1 - 1 <= kMaxFid;
        bad |= ok ? 0u : 1u;
        if (ok && sp < cap_nnz) keys[sp] = hash_lds(T + c1 + 1, c2 - c1 - 1);
        ++sp;
      }
      if (ch == '\n') {
        ++nl;
        if (nl <= cap_rows) rowptr[nl] = sp;
      }
    }
    __syncthreads();
    if (tid == 0) {
      run_nl += tot & 0xFFFFu;
      run_sp += tot >> 16;
    }
  }
  if (__any((int)bad) && lane == 0) atomicOr(&out->bad, 1u);
}

}  // namespace

struct xf_ingest {
  size_t uploaded = 0;          // xf_ingest_upload: bytes of text on their way to d_text (0: none)
  size_t uploaded_n = 0;        // ... with the closing newline
  hipEvent_t ev_up = nullptr;   // ... recorded behind that copy
  size_t cap_text = 0;          // bytes of text a block may hold
  uint32_t cap_rows = 0, cap_nnz = 0;
  uint8_t *d_text = nullptr;    // [cap_text rounded up + kTile + kHalo]
  uint8_t *h_text = nullptr;    // pinned staging of the same size
  uint64_t *d_keys = nullptr;
  uint32_t *d_rowptr = nullptr;
  int32_t *d_labels = nullptr;
  uint2 *d_wgcnt = nullptr;
  TokCounts *d_counts = nullptr, *h_counts = nullptr;
};

extern "C" int xf_ingest_destroy(xf_ingest *g) {
  if (!g) return XF_OK;
  (void)hipDeviceSynchronize();
  if (g->d_text) (void)hipFree(g->d_text);
  if (g->h_text) (void)hipHostFree(g->h_text);
  if (g->d_keys) (void)hipFree(g->d_keys);
  if (g->d_rowptr) (void)hipFree(g->d_rowptr);
  if (g->d_labels) (void)hipFree(g->d_labels);
  if (g->d_wgcnt) (void)hipFree(g->d_wgcnt);
  if (g->d_counts) (void)hipFree(g->d_counts);
  if (g->h_counts) (void)hipHostFree(g->h_counts);
  if (g->ev_up) (void)hipEventDestroy(g->ev_up);
  delete g;
  return XF_OK;
}

static size_t padded(size_t n) { return (n + kTile - 1) / kTile * kTile + kTile + kHalo; }

extern "C" int xf_ingest_create(xf_ingest **out, size_t max_text_bytes) {
  XF_REQUIRE(out && max_text_bytes > 0 && max_text_bytes < 0xF0000000ull,
             "xf_ingest_create: bad argument");
  xf_ingest *g = new xf_ingest;
  struct Guard {
    xf_ingest *g;
    ~Guard() {
      if (g) xf_ingest_destroy(g);
    }
  } guard{g};
  g->cap_text = max_text_bytes;
  // a token of the sample data is "12:3456:0.37 " (13 bytes), the shortest well-formed one
  // "0:a: " (5), the shortest row "0\t0:a:\n" (7).  The arrays are sized for 4 bytes per token
  // and 6 per row; a block with more raises the flag (k_tok_scan).
  g->cap_nnz = (uint32_t)(max_text_bytes / 4 + 16);
  g->cap_rows = (uint32_t)(max_text_bytes / 6 + 16);
  XF_HIP(hipMalloc((void **)&g->d_text, padded(max_text_bytes)));
  XF_HIP(hipHostMalloc((void **)&g->h_text, padded(max_text_bytes)));
  XF_HIP(hipMalloc((void **)&g->d_keys, (size_t)g->cap_nnz * 8));
  XF_HIP(hipMalloc((void **)&g->d_rowptr, ((size_t)g->cap_rows + 2) * 4));
  XF_HIP(hipMalloc((void **)&g->d_labels, ((size_t)g->cap_rows + 1) * 4));
  XF_HIP(hipMalloc((void **)&g->d_wgcnt, kMaxWG * sizeof(uint2)));
  XF_HIP(hipMalloc((void **)&g->d_counts, sizeof(TokCounts)));
  XF_HIP(hipHostMalloc((void **)&g->h_counts, sizeof(TokCounts)));
  XF_HIP(hipEventCreateWithFlags(&g->ev_up, hipEventDisableTiming));
  guard.g = nullptr;
  *out = g;
  return XF_OK;
}

// The staging buffer of the next block (pinned host memory, xf_ingest's own): the caller copies
// the block's text there — from the mapped file, with as many threads as it likes — and hands
// the length to xf_ingest_block.
extern "C" int xf_ingest_staging(xf_ingest *g, char **buf, size_t *cap) {
  XF_REQUIRE(g && buf, "xf_ingest_staging: null argument");
  *buf = (char *)g->h_text;
  if (cap) *cap = g->cap_text;
  return XF_OK;
}

// every line of the text the kernels see ends in '\n' (a block that was cut at a newline comes
// without it, a file may end without one), and the bytes past the text up to the tile + halo
// the kernels may read are newlines: returns the length with the closing newline
static size_t close_and_pad(xf_ingest *g, size_t len) {
  size_t n = len;
  if (g->h_text[n - 1] != '\n') g->h_text[n++] = '\n';
  memset(g->h_text + n, '\n', padded(n) - n);
  return n;
}

// The staged text (`len` bytes in the staging buffer) on its way to the device, asynchronously on
// `stream` — the staging thread's own, so that the copy of block i + 1 runs while the GPU works
// on block i.  The next xf_ingest_block(g, NULL, len, ...) waits for it on its stream instead of
// copying; the staging buffer is free again when the copy has finished (xf_ingest_block returns).
extern "C" int xf_ingest_upload(xf_ingest *g, size_t len, void *stream) {
  XF_REQUIRE(g && len > 0 && len <= g->cap_text, "xf_ingest_upload: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const size_t n = close_and_pad(g, len);
  XF_HIP(hipMemcpyAsync(g->d_text, g->h_text, padded(n), hipMemcpyHostToDevice, s));
  XF_HIP(hipEventRecord(g->ev_up, s));
  g->uploaded = len;
  g->uploaded_n = n;
  return XF_OK;
}

// Tokenise `len` bytes of text: `text` == null: they are in the staging buffer (and, after
// xf_ingest_upload, on their way to the device); else they are copied there first.  Uploads, runs
// the kernels, waits for the counts.  *ok = 0: the block is
// not of the common shape (or holds more rows / tokens than the arrays): the device arrays are
// not to be used — parse the block on the host.  The device arrays stay valid until the next
// call.
extern "C" int xf_ingest_block(xf_ingest *g, const char *text, size_t len, void *stream,
                               const uint64_t **d_keys, const uint32_t **d_rowptr,
                               const int32_t **d_labels, uint32_t *rows, uint32_t *nnz,
                               int *ok) {
  XF_REQUIRE(g && rows && nnz && ok, "xf_ingest_block: null argument");
  XF_REQUIRE(len <= g->cap_text, "xf_ingest_block: %zu bytes of text, room for %zu", len,
             g->cap_text);
  hipStream_t s = (hipStream_t)stream;
  *rows = *nnz = 0;
  *ok = 1;
  if (d_keys) *d_keys = g->d_keys;
  if (d_rowptr) *d_rowptr = g->d_rowptr;
  if (d_labels) *d_labels = g->d_labels;
  const size_t up = g->uploaded, up_n = g->uploaded_n;
  g->uploaded = 0;
  if (len == 0) {
    const uint32_t zero = 0;
    XF_HIP(hipMemcpyAsync(g->d_rowptr, &zero, 4, hipMemcpyHostToDevice, s));
    XF_HIP(hipStreamSynchronize(s));
    return XF_OK;
  }
  size_t n;
  if (!text && up == len) {  // xf_ingest_upload has sent it: its copy first, then the kernels
    XF_HIP(hipStreamWaitEvent(s, g->ev_up, 0));
    n = up_n;
  } else {
    if (text) memcpy(g->h_text, text, len);
    n = close_and_pad(g, len);
    XF_HIP(hipMemcpyAsync(g->d_text, g->h_text, padded(n), hipMemcpyHostToDevice, s));
  }
  const uint32_t ntile = (uint32_t)((n + kTile - 1) / kTile);
  const uint32_t per = (ntile + kMaxWG - 1) / kMaxWG;
  const uint32_t span = per * kTile, nwg = (ntile + per - 1) / per;
  hipLaunchKernelGGL(k_tok_count, dim3(nwg), dim3(kTok), 0, s, g->d_text, (uint32_t)n, span,
                     g->d_wgcnt);
  hipLaunchKernelGGL(k_tok_scan, dim3(1), dim3(kMaxWG), 0, s, g->d_wgcnt, nwg, g->cap_rows,
                     g->cap_nnz, g->d_counts);
  hipLaunchKernelGGL(k_tok_emit, dim3(nwg), dim3(kTok), 0, s, g->d_text, (uint32_t)n, span,
                     (const uint2 *)g->d_wgcnt, g->cap_rows, g->cap_nnz, g->d_keys, g->d_rowptr,
                     g->d_labels, g->d_counts);
  XF_HIP(hipGetLastError());
  XF_HIP(hipMemcpyAsync(g->h_counts, g->d_counts, sizeof(TokCounts), hipMemcpyDeviceToHost, s));
  XF_HIP(hipStreamSynchronize(s));
  *ok = g->h_counts->bad ? 0 : 1;
  *rows = g->h_counts->rows;
  *nnz = g->h_counts->nnz;
  return XF_OK;
}

// device memory -> host (tests and small tools read the tokeniser's arrays back with it)
extern "C" int xf_copy_to_host(void *dst, const void *d_src, size_t bytes) {
  XF_REQUIRE((dst && d_src) || bytes == 0, "xf_copy_to_host: null argument");
  if (bytes) XF_HIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
  return XF_OK;
}
