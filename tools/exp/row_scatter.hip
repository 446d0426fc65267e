// Microbenchmark (GPU box): what it costs to regroup 10^7 (row, value) pairs that arrive in
// key order into row-grouped order (the FM forward's per-row record indices from the keyed
// build's cells): slot = rowptr[row] + atomicAdd(&cursor[row], 1); out[slot] = value.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/row_scatter.hip -o /tmp/rs && /tmp/rs
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

__global__ void __launch_bounds__(256) k_atomic(const uint32_t *__restrict__ row, uint32_t n,
                                                uint32_t nnz_per_row, uint32_t *cur,
                                                uint32_t *__restrict__ out) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t r = row[i];
    const uint32_t s = r * nnz_per_row + atomicAdd(&cur[r], 1u);
    out[s] = i;
  }
}
__global__ void __launch_bounds__(256) k_noatomic(const uint32_t *__restrict__ row,
                                                  const uint32_t *__restrict__ slot, uint32_t n,
                                                  uint32_t *__restrict__ out) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[slot[i]] = i;
}
__global__ void __launch_bounds__(256) k_atomic_only(const uint32_t *__restrict__ row, uint32_t n,
                                                     uint32_t *cur, uint32_t *__restrict__ out) {
  uint32_t acc = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    acc += atomicAdd(&cur[row[i]], 1u);
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const uint32_t R = 50000, per = 200, n = R * per;
  // arrival order: grouped by (window of 16667 rows, chunk of 2048 keys), rows ascending in a cell
  std::vector<uint32_t> row(n), slot(n);
  {
    std::vector<std::vector<uint32_t>> cell(3 * 4883);
    srand(3);
    for (uint32_t r = 0; r < R; ++r)
      for (uint32_t j = 0; j < per; ++j) cell[(r / 16667) * 4883 + rand() % 4883].push_back(r);
    size_t o = 0;
    // chunk-major like the regroup kernel would walk: chunk c, windows 0..2
    for (uint32_t c = 0; c < 4883; ++c)
      for (uint32_t v = 0; v < 3; ++v)
        for (uint32_t r : cell[v * 4883 + c]) row[o++] = r;
    std::vector<uint32_t> cur(R, 0);
    for (uint32_t i = 0; i < n; ++i) slot[i] = row[i] * per + cur[row[i]]++;
  }
  uint32_t *d_row, *d_slot, *d_cur, *d_out;
  hipMalloc(&d_row, n * 4);
  hipMalloc(&d_slot, n * 4);
  hipMalloc(&d_cur, R * 4);
  hipMalloc(&d_out, n * 4);
  hipMemcpy(d_row, row.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_slot, slot.data(), n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int which = 0; which < 3; ++which) {
    float best = 1e9f, ms;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemset(d_cur, 0, R * 4);
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_atomic, dim3(4096), dim3(256), 0, 0, d_row, n, per, d_cur, d_out);
      if (which == 1) hipLaunchKernelGGL(k_noatomic, dim3(4096), dim3(256), 0, 0, d_row, d_slot, n, d_out);
      if (which == 2) hipLaunchKernelGGL(k_atomic_only, dim3(4096), dim3(256), 0, 0, d_row, n, d_cur, d_out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      best = std::min(best, ms);
    }
    const char *nm[] = {"atomic cursor per row + scattered 4-byte store", "scattered 4-byte store alone (slots precomputed)", "atomic cursors alone"};
    printf("%-52s %8.1f us\n", nm[which], best * 1e3);
  }
  return 0;
}
