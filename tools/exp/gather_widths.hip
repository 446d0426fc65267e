// gather_widths.hip — how long do N random gathers of W bytes from a table of S bytes take?
// (design input for the FM forward: 64-byte factor rows vs per-key scalars; shows the L2 /
// Infinity-Cache / HBM regimes).  Built and driven by tools/exp/gather_widths.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int W4>  // row width in float4 units (W = 16*W4 bytes); W4 = 0 -> 4-byte rows
__global__ void __launch_bounds__(256) e_gather(const uint32_t *__restrict__ idx,
                                                const float *__restrict__ tab, size_t n,
                                                float *__restrict__ out) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t r = idx[i];
    if (W4 == 0) {
      acc += tab[r];
    } else {
      const float4 *row = reinterpret_cast<const float4 *>(tab) + (size_t)r * W4;
#pragma unroll
      for (int q = 0; q < W4; ++q) {
        const float4 v = row[q];
        acc += v.x + v.y + v.z + v.w;
      }
    }
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// 8-byte rows
__global__ void __launch_bounds__(256) e_gather8(const uint32_t *__restrict__ idx,
                                                 const float2 *__restrict__ tab, size_t n,
                                                 float *__restrict__ out) {
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float2 v = tab[idx[i]];
    acc += v.x + v.y;
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

extern "C" int x_gather(int width_bytes, const void *idx, const void *tab, size_t n, void *out,
                        int grid, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint32_t *i = (const uint32_t *)idx;
  const float *t = (const float *)tab;
  float *o = (float *)out;
  switch (width_bytes) {
    case 4: hipLaunchKernelGGL(e_gather<0>, dim3(grid), dim3(256), 0, s, i, t, n, o); break;
    case 8: hipLaunchKernelGGL(e_gather8, dim3(grid), dim3(256), 0, s, i, (const float2 *)tab, n, o); break;
    case 16: hipLaunchKernelGGL(e_gather<1>, dim3(grid), dim3(256), 0, s, i, t, n, o); break;
    case 32: hipLaunchKernelGGL(e_gather<2>, dim3(grid), dim3(256), 0, s, i, t, n, o); break;
    case 64: hipLaunchKernelGGL(e_gather<4>, dim3(grid), dim3(256), 0, s, i, t, n, o); break;
    default: return 1;
  }
  return (int)hipGetLastError();
}
