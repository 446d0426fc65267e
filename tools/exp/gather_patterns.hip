// Microbenchmark (GPU box): rate of 4-byte gathers w[idx[j]] for index streams of different
// locality — the forward's inner loop without the LDS atomics.  One 1024-thread workgroup per
// CU (the forward's geometry), 16 gathers in flight per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/gather_patterns.hip -o /tmp/gp && /tmp/gp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

constexpr int kE = 16;

__global__ void __launch_bounds__(1024) k(const uint32_t *__restrict__ idx, const float *__restrict__ w,
                                           uint32_t per_wg, float *out) {
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t *p = idx + (size_t)blockIdx.x * per_wg;
  float acc = 0.f;
  for (uint32_t b = wave * 64 * kE; b < per_wg; b += 16 * 64 * kE) {
    uint32_t e[kE];
#pragma unroll
    for (int q = 0; q < kE; ++q) e[q] = p[b + q * 64 + lane];
    float v[kE];
#pragma unroll
    for (int q = 0; q < kE; ++q) v[q] = w[e[q]];
#pragma unroll
    for (int q = 0; q < kE; ++q) acc += v[q];
  }
  if (acc == 1.2345f) out[0] = acc;
}

int main() {
  const uint32_t per_wg = 16 * 64 * kE * 3;  // three rounds per wave
  const uint32_t nwg = 256;
  const size_t n = (size_t)per_wg * nwg;
  const uint32_t M = 10000000;
  std::vector<uint32_t> h(n);
  uint32_t *d_idx;
  float *d_w, *d_out;
  hipMalloc(&d_idx, n * 4);
  hipMalloc(&d_w, (size_t)M * 4);
  hipMalloc(&d_out, 4);
  hipMemset(d_w, 0, (size_t)M * 4);
  const char *names[] = {"random over 40 MB", "random inside a 16 KiB chunk per 1024 entries",
                         "sorted inside a 16 KiB chunk, 1365 of 4096 (cell sorted by key)",
                         "sorted, 1024 of 4096 per block", "contiguous"};
  for (int pat = 0; pat < 5; ++pat) {
    srand(1);
    for (size_t blk = 0; blk * 1024 < n; ++blk) {
      const uint32_t base = (uint32_t)((blk * 7919) % (M / 4096 - 1)) * 4096;
      uint32_t tmp[1024];
      for (int i = 0; i < 1024; ++i) {
        if (pat == 0) tmp[i] = (uint32_t)(((uint64_t)rand() * 32768 + rand()) % M);
        else if (pat == 4) tmp[i] = base + i;
        else tmp[i] = base + (rand() & 4095);
      }
      if (pat == 2) {  // 1365 draws sorted, take the first 1024 (same density as a cell)
        uint32_t t2[1365];
        for (int i = 0; i < 1365; ++i) t2[i] = base + (rand() & 4095);
        std::sort(t2, t2 + 1365);
        for (int i = 0; i < 1024; ++i) tmp[i] = t2[i];
      }
      if (pat == 3) std::sort(tmp, tmp + 1024);
      for (int i = 0; i < 1024; ++i) h[blk * 1024 + i] = tmp[i];
    }
    hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(nwg), dim3(1024), 0, 0, d_idx, d_w, per_wg, d_out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-66s %7.1f us  %5.2f lanes/clk/CU\n", names[pat], ms * 1e3,
           (double)per_wg / (ms * 1e-3 * 2.4e9));
  }
  return 0;
}
