// Microbenchmark (GPU box): what a sorted 4-byte gather through a WINDOW costs as a function of
// how many entries share a 128-byte line — the gradient kernel's loss gathers (a cell's entries
// ascend through the row window's losses: 680 entries over 521 lines with 2048-key chunks, 2720
// with 8192-key chunks) and the forward's weight gathers (a cell's entries ascend through its
// chunk's weights).  Geometry of the gradient kernel: 256-thread workgroups, 8 per CU, every
// workgroup sweeping its own cell (its own subset of the window's words), 8 gathers in flight
// per lane.  Prints lanes / clock / CU and the time 10^7 gathers would take on 256 CUs.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/gather_lines.hip -o /tmp/gl && /tmp/gl
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

constexpr int kE = 8;

__global__ void __launch_bounds__(256) k(const uint32_t *__restrict__ idx,
                                         const float *__restrict__ w, uint32_t per_wg,
                                         float *out) {
  const uint32_t *p = idx + (size_t)blockIdx.x * per_wg;
  float acc = 0.f;
  for (uint32_t b = 0; b < per_wg; b += 256 * kE) {
    uint32_t e[kE];
#pragma unroll
    for (int q = 0; q < kE; ++q) e[q] = p[b + q * 256 + threadIdx.x];
    float v[kE];
#pragma unroll
    for (int q = 0; q < kE; ++q) v[q] = w[e[q]];
#pragma unroll
    for (int q = 0; q < kE; ++q) acc += v[q];
  }
  if (acc == 1.2345f) out[0] = acc;
}

int main() {
  const uint32_t nwg = 2048 * 3;  // three rounds of 8 workgroups per CU
  const uint32_t win_words[] = {2048, 4096, 16667, 16667, 16667, 16667, 16667, 50000};
  const uint32_t per_cell[] = {680, 680, 680, 1360, 2720, 5440, 10880, 2048};
  const char *what[] = {"forward today: 680 sorted entries over an 8 KiB chunk of weights",
                        "680 sorted entries over 16 KiB",
                        "gradient today: 680 sorted entries over a 65 KiB window (2048-key chunk)",
                        "4096-key chunk: 1360 entries over the window",
                        "8192-key chunk: 2720 entries over the window",
                        "16384-key chunk: 5440 entries over the window",
                        "32768-key chunk: 10880 entries over the window",
                        "no windows: 2048 entries over all 50 000 losses"};
  float *d_w, *d_out;
  hipMalloc(&d_w, (size_t)64 << 20);
  hipMemset(d_w, 0, (size_t)64 << 20);
  hipMalloc(&d_out, 4);
  for (int pat = 0; pat < 8; ++pat) {
    // a workgroup walks cells of `per_cell` entries until it has done ~8192 entries
    const uint32_t cells = std::max(1u, 8192u / per_cell[pat]);
    uint32_t per_wg = cells * per_cell[pat];
    per_wg = (per_wg + 256 * kE - 1) / (256 * kE) * (256 * kE);
    std::vector<uint32_t> h((size_t)per_wg * nwg);
    srand(7);
    std::vector<uint32_t> t;
    for (uint32_t g = 0; g < nwg; ++g) {
      size_t o = (size_t)g * per_wg, filled = 0;
      while (filled < per_wg) {
        // every cell gathers from one of 3 windows (the three row windows of a minibatch)
        const uint32_t base = (uint32_t)(rand() % 3) * win_words[pat];
        const uint32_t n = (uint32_t)std::min<size_t>(per_cell[pat], per_wg - filled);
        t.resize(n);
        for (uint32_t i = 0; i < n; ++i) t[i] = base + (uint32_t)(rand() % win_words[pat]);
        std::sort(t.begin(), t.end());
        for (uint32_t i = 0; i < n; ++i) h[o + filled + i] = t[i];
        filled += n;
      }
    }
    uint32_t *d_idx;
    hipMalloc(&d_idx, h.size() * 4);
    hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, d_idx, d_w, per_wg, d_out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      best = std::min(best, ms);
    }
    const double lanes = (double)per_wg * nwg;
    printf("%-78s %7.1f us  %5.2f lanes/clk/CU  1e7 gathers: %5.1f us\n", what[pat], best * 1e3,
           lanes / 256 / (best * 1e-3 * 2.4e9), best * 1e3 * 1e7 / lanes);
    hipFree(d_idx);
  }
  return 0;
}
