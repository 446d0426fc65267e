// Experiment: how fast do 240 workgroups add 17408 fp64 row sums each into a per-XCD array with
// global atomics that stay in the XCD's L2 (workgroup scope, array picked by the hardware XCC
// id) or go device-wide (agent scope, one array), against writing 240 x 17408 doubles of
// partials and reading them back?    hipcc --offload-arch=gfx950 -O3 l2_atomics.hip -o l2a
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int W = 17408, NW = 3, G = 80;

__device__ inline unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}

template <int MODE>  // 0 partial stores, 1 wg-scope atomics into acc[xcc], 2 agent-scope into acc[0]
__global__ void __launch_bounds__(1024) k_add(double *out, unsigned *xccs) {
  __shared__ double wx[W];
  const unsigned slot = blockIdx.x >> 3, v = slot % NW, g = (slot / NW) * 8 + (blockIdx.x & 7u);
  for (int r = threadIdx.x; r < W; r += 1024) wx[r] = (double)(r % 7) + 1.0;
  __syncthreads();
  if (MODE == 0) {
    double *o = out + ((size_t)v * G + g) * W;
    for (int r = threadIdx.x; r < W; r += 1024) o[r] = wx[r];
  } else if (MODE == 1) {
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) xccs[blockIdx.x] = x;
    double *o = out + ((size_t)x * NW + v) * W;
    for (int r = threadIdx.x; r < W; r += 1024)
      __hip_atomic_fetch_add(o + r, wx[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    double *o = out + (size_t)v * W;
    for (int r = threadIdx.x; r < W; r += 1024)
      __hip_atomic_fetch_add(o + r, wx[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int NP>  // sum NP partial arrays per (window,row), zero them
__global__ void k_fin(double *p, double *res, size_t stride) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)NW * W) return;
  const size_t v = i / W, r = i % W;
  double a = 0;
  for (int q = 0; q < NP; ++q) {
    double *x = NP == G ? p + ((v * G + q) * W + r) : p + (((size_t)q * NW + v) * W + r);
    a += *x;
    if (NP != G) *x = 0.0;
  }
  res[i] = a;
}

int main() {
  double *part, *acc, *res;
  unsigned *xccs;
  CK(hipMalloc(&part, (size_t)NW * G * W * 8));
  CK(hipMalloc(&acc, (size_t)8 * NW * W * 8));
  CK(hipMalloc(&res, (size_t)NW * W * 8));
  CK(hipMalloc(&xccs, 1024 * 4));
  CK(hipMemset(acc, 0, (size_t)8 * NW * W * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 12; ++rep) {
      if (rep == 2) CK(hipEventRecord(e0));
      if (mode == 0) {
        hipLaunchKernelGGL(k_add<0>, dim3(NW * G), dim3(1024), 0, 0, part, xccs);
        hipLaunchKernelGGL(k_fin<G>, dim3((NW * W + 255) / 256), dim3(256), 0, 0, part, res, 0);
      } else if (mode == 1) {
        hipLaunchKernelGGL(k_add<1>, dim3(NW * G), dim3(1024), 0, 0, acc, xccs);
        hipLaunchKernelGGL(k_fin<8>, dim3((NW * W + 255) / 256), dim3(256), 0, 0, acc, res, 0);
      } else {
        hipLaunchKernelGGL(k_add<2>, dim3(NW * G), dim3(1024), 0, 0, acc, xccs);
        hipLaunchKernelGGL(k_fin<1>, dim3((NW * W + 255) / 256), dim3(256), 0, 0, acc, res, 0);
      }
    }
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> h((size_t)NW * W);
    CK(hipMemcpy(h.data(), res, h.size() * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < h.size(); ++i)
      if (h[i] != G * ((double)((i % W) % 7) + 1.0)) ++bad;
    printf("mode %d: %.2f us per add+finalize pair, wrong sums %zu of %zu\n", mode, ms * 1e3 / 10, bad, h.size());
    if (mode == 1) {
      std::vector<unsigned> x(NW * G);
      CK(hipMemcpy(x.data(), xccs, x.size() * 4, hipMemcpyDeviceToHost));
      int agree = 0;
      for (int b = 0; b < NW * G; ++b) agree += x[b] == (unsigned)(b % 8);
      printf("  blocks on XCD b %% 8: %d of %d\n", agree, NW * G);
    }
  }
  return 0;
}
