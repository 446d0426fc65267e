// Microbenchmark (GPU box): throughput of LDS atomics with random addresses, per CU.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/exp/lds_atomics.hip -o /tmp/lds_atomics && /tmp/lds_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int kRows = 16384;
constexpr int kIter = 256;

template <int KIND>
__global__ void __launch_bounds__(1024) k(const uint32_t *__restrict__ idx, double *out, int same) {
  __shared__ double acc[kRows];
  for (int r = threadIdx.x; r < kRows; r += 1024) acc[r] = 0.0;
  __syncthreads();
  uint32_t x = idx[blockIdx.x * 1024 + threadIdx.x];
  for (int i = 0; i < kIter; ++i) {
    x = x * 1664525u + 1013904223u;
    const uint32_t r = same ? (x >> 18) & ~63u : (x >> 18);  // 14 bits
    if (KIND == 0) atomicAdd(&acc[r], 1.5);
    if (KIND == 1) atomicAdd((unsigned long long *)&acc[r], 3ull);
    if (KIND == 2) atomicAdd((float *)&acc[r], 1.5f);
    if (KIND == 3) atomicAdd((unsigned int *)&acc[r], 3u);
    if (KIND == 4) acc[r] = 1.5;
    if (KIND == 5) ((float *)acc)[r] = 1.5f;
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc[x & (kRows - 1)];
}

template <int KIND>
void run(const char *name, const uint32_t *d_idx, double *d_out, int same) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, d_idx, d_out, same);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_cu = 1024.0 * kIter;  // lane-ops per CU (one workgroup per CU)
  printf("%-28s %s  %8.1f us   %6.2f lane-ops / clk / CU (2.4 GHz)\n", name,
         same ? "64 lanes -> 1 address" : "random addresses   ", ms * 1e3,
         per_cu / (ms * 1e-3 * 2.4e9));
}

int main() {
  uint32_t *d_idx;
  double *d_out;
  hipMalloc(&d_idx, 256 * 1024 * 4);
  hipMalloc(&d_out, 256 * 8);
  uint32_t *h = (uint32_t *)malloc(256 * 1024 * 4);
  for (int i = 0; i < 256 * 1024; ++i) h[i] = 2654435761u * (uint32_t)(i + 1);
  hipMemcpy(d_idx, h, 256 * 1024 * 4, hipMemcpyHostToDevice);
  for (int same = 0; same < 2; ++same) {
    run<0>("ds_add_f64", d_idx, d_out, same);
    run<1>("ds_add_u64", d_idx, d_out, same);
    run<2>("ds_add_f32", d_idx, d_out, same);
    run<3>("ds_add_u32", d_idx, d_out, same);
    run<4>("ds_write_b64", d_idx, d_out, same);
    run<5>("ds_write_b32", d_idx, d_out, same);
  }
  return 0;
}
