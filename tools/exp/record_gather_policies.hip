// Microbenchmark (GPU box): does a cache-policy bit make the L2 fetch LESS than a 128-byte line
// for a random 32-byte record?  The FM forward gathers 10^7 32-byte records out of 320 MB and
// is bound by the 128-byte requests that costs (1.30 GB at 6.8 TB/s).  Variants:
// plain loads, __builtin_nontemporal_load, and global_load_dwordx4 with the nt / sc0 / sc1 bits
// set by hand.  Two lanes per record (16 bytes each), 4 records in flight per lane pair.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/record_gather_policies.hip -o /tmp/rg && /tmp/rg
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int POL>
__device__ __forceinline__ u4 load16(const u4 *p) {
  if constexpr (POL == 0) return *p;
  if constexpr (POL == 1) return __builtin_nontemporal_load(p);
  u4 v;
  if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if constexpr (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if constexpr (POL == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int POL>
__global__ void __launch_bounds__(256) k(const unsigned char *__restrict__ p, size_t nrec,
                                         unsigned long long *sink) {
  const size_t pairs = (size_t)gridDim.x * blockDim.x / 2;
  unsigned long long acc = 0;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 2; i < nrec; i += pairs) {
    const size_t u = (i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & (nrec - 1);
    const u4 v = load16<POL>((const u4 *)(p + u * 32 + (threadIdx.x & 1) * 16));
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 0x123456789abcdefull) *sink = acc;
}

template <int POL>
static void run(const char *what, const unsigned char *d, size_t nrec, unsigned long long *sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9f, ms = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<POL>, dim3(8192), dim3(256), 0, 0, d, nrec, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
  }
  printf("%-44s %8.1f us  %6.2f TB/s of records  (%5.2f TB/s if every record cost a 128-byte line)\n",
         what, best * 1e3, nrec * 32.0 / (best * 1e-3) / 1e12, nrec * 128.0 / (best * 1e-3) / 1e12);
}

int main() {
  const size_t nrec = (size_t)1 << 23;  // 8.4 M records of 32 bytes = 268 MB
  unsigned char *d;
  unsigned long long *sink;
  hipMalloc(&d, nrec * 32);
  hipMemset(d, 1, nrec * 32);
  hipMalloc(&sink, 8);
  run<0>("plain global_load_dwordx4", d, nrec, sink);
  run<1>("__builtin_nontemporal_load", d, nrec, sink);
  run<2>("nt", d, nrec, sink);
  run<3>("sc0", d, nrec, sink);
  run<4>("sc1", d, nrec, sink);
  run<5>("sc0 sc1", d, nrec, sink);
  run<6>("sc0 sc1 nt", d, nrec, sink);
  run<0>("plain again", d, nrec, sink);
  return 0;
}
