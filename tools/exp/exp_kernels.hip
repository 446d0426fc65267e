// Scratch ablation kernels (tuning aid, not shipped): where does the time of the
// gather + segmented-reduce kernels go?
#include <hip/hip_runtime.h>
#include <stdint.h>
#define XF_TILE_NNZ 2048
#define XF_TILE_KEYS 2048

__global__ void __launch_bounds__(256) e_flat(const uint32_t* __restrict__ idx, const float* __restrict__ tab, size_t n, float* __restrict__ out) {
  size_t s = (size_t)gridDim.x * blockDim.x;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += s) out[j] = tab[idx[j]];
}
// flat gather, sum kept in registers, one store per thread (no 40 MB output)
__global__ void __launch_bounds__(256) e_flat_nostore(const uint32_t* __restrict__ idx, const float* __restrict__ tab, size_t n, float* __restrict__ out) {
  size_t s = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += s) acc += tab[idx[j]];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// tile staging only: gather into LDS, barrier, one store per thread
template <int BLK>
__global__ void __launch_bounds__(BLK) e_stage(const uint32_t* __restrict__ tile_ptr, uint32_t ntiles, const uint32_t* __restrict__ segptr,
    const uint32_t* __restrict__ idx, const float* __restrict__ tab, float* __restrict__ out) {
  __shared__ float vals[XF_TILE_NNZ];
  float acc = 0.f;
  for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    uint32_t ua = tile_ptr[t], ub = tile_ptr[t + 1];
    uint32_t j0 = segptr[ua], j1 = segptr[ub];
    if (j1 - j0 > XF_TILE_NNZ) continue;
    for (uint32_t j = j0 + threadIdx.x; j < j1; j += BLK) vals[j - j0] = tab[idx[j]];
    __syncthreads();
    acc += vals[(threadIdx.x * 7) % (j1 - j0 ? j1 - j0 : 1)];
    __syncthreads();
  }
  out[(size_t)blockIdx.x * BLK + threadIdx.x] = acc;
}
// full tiled segmented sum (float out per key)
template <int BLK>
__global__ void __launch_bounds__(BLK) e_tiled(const uint32_t* __restrict__ tile_ptr, uint32_t ntiles, const uint32_t* __restrict__ segptr,
    const uint32_t* __restrict__ idx, const float* __restrict__ tab, float* __restrict__ g) {
  __shared__ float vals[XF_TILE_NNZ];
  __shared__ uint32_t sp[XF_TILE_KEYS + 1];
  for (uint32_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    uint32_t ua = tile_ptr[t], ub = tile_ptr[t + 1], nk = ub - ua;
    uint32_t j0 = segptr[ua], j1 = segptr[ub];
    if (j1 - j0 > XF_TILE_NNZ) continue;
    for (uint32_t k = threadIdx.x; k <= nk; k += BLK) sp[k] = segptr[ua + k] - j0;
    for (uint32_t j = j0 + threadIdx.x; j < j1; j += BLK) vals[j - j0] = tab[idx[j]];
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < nk; k += BLK) {
      double acc = 0.0;
      for (uint32_t j = sp[k]; j < sp[k + 1]; ++j) acc += (double)vals[j];
      g[ua + k] = (float)acc;
    }
    __syncthreads();
  }
}
// wave-synchronous tiles: each wave owns a sub-tile of <= 512 nnz, no workgroup barrier
__global__ void __launch_bounds__(256) e_wavetile(const uint32_t* __restrict__ wt_ptr, uint32_t nwt, const uint32_t* __restrict__ segptr,
    const uint32_t* __restrict__ idx, const float* __restrict__ tab, float* __restrict__ g) {
  __shared__ float vals_all[4][512];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float* vals = vals_all[wv];
  const uint32_t nw = gridDim.x * 4;
  for (uint32_t t = blockIdx.x * 4 + wv; t < nwt; t += nw) {
    uint32_t ua = wt_ptr[t], ub = wt_ptr[t + 1], nk = ub - ua;
    uint32_t j0 = segptr[ua], j1 = segptr[ub];
    if (j1 - j0 > 512) continue;
    for (uint32_t j = j0 + lane; j < j1; j += 64) vals[j - j0] = tab[idx[j]];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0);
    for (uint32_t k = lane; k < nk; k += 64) {
      uint32_t b = segptr[ua + k] - j0, e = segptr[ua + k + 1] - j0;
      double acc = 0.0;
      for (uint32_t j = b; j < e; ++j) acc += (double)vals[j];
      g[ua + k] = (float)acc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// LDS-resident table (first NL floats) + wave-synchronous tiles, 1024-thread workgroups
#define NL 32768
__global__ void __launch_bounds__(1024) e_wavetile_lds(const uint32_t* __restrict__ wt_ptr, uint32_t nwt, const uint32_t* __restrict__ segptr,
    const uint32_t* __restrict__ idx, const float* __restrict__ tab, uint32_t ntab, float* __restrict__ g) {
  __shared__ float ltab[NL];
  __shared__ float vals_all[16][512];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t nl = ntab < NL ? ntab : NL;
  for (uint32_t i = threadIdx.x; i < nl; i += 1024) ltab[i] = tab[i];
  __syncthreads();
  float* vals = vals_all[wv];
  const uint32_t nw = gridDim.x * 16;
  for (uint32_t t = blockIdx.x * 16 + wv; t < nwt; t += nw) {
    uint32_t ua = wt_ptr[t], ub = wt_ptr[t + 1], nk = ub - ua;
    uint32_t j0 = segptr[ua], j1 = segptr[ub];
    if (j1 - j0 > 512) continue;
    for (uint32_t j = j0 + lane; j < j1; j += 64) { uint32_t r = idx[j]; vals[j - j0] = r < nl ? ltab[r] : tab[r]; }
    __builtin_amdgcn_wave_barrier();
    for (uint32_t k = lane; k < nk; k += 64) {
      uint32_t b = segptr[ua + k] - j0, e = segptr[ua + k + 1] - j0;
      double acc = 0.0;
      for (uint32_t j = b; j < e; ++j) acc += (double)vals[j];
      g[ua + k] = (float)acc;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

extern "C" {
void x_wavetile_lds(const void* tp, uint32_t nt, const void* sp, const void* idx, const void* tab, uint32_t ntab, void* out, int grid, void* s) { hipLaunchKernelGGL(e_wavetile_lds, dim3(grid), dim3(1024), 0, (hipStream_t)s, (const uint32_t*)tp, nt, (const uint32_t*)sp, (const uint32_t*)idx, (const float*)tab, ntab, (float*)out); }
void x_flat(const void* idx, const void* tab, size_t n, void* out, int grid, void* s) { hipLaunchKernelGGL(e_flat, dim3(grid), dim3(256), 0, (hipStream_t)s, (const uint32_t*)idx, (const float*)tab, n, (float*)out); }
void x_flat_nostore(const void* idx, const void* tab, size_t n, void* out, int grid, void* s) { hipLaunchKernelGGL(e_flat_nostore, dim3(grid), dim3(256), 0, (hipStream_t)s, (const uint32_t*)idx, (const float*)tab, n, (float*)out); }
void x_stage256(const void* tp, uint32_t nt, const void* sp, const void* idx, const void* tab, void* out, int grid, void* s) { hipLaunchKernelGGL(e_stage<256>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const uint32_t*)tp, nt, (const uint32_t*)sp, (const uint32_t*)idx, (const float*)tab, (float*)out); }
void x_tiled256(const void* tp, uint32_t nt, const void* sp, const void* idx, const void* tab, void* out, int grid, void* s) { hipLaunchKernelGGL(e_tiled<256>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const uint32_t*)tp, nt, (const uint32_t*)sp, (const uint32_t*)idx, (const float*)tab, (float*)out); }
void x_tiled1024(const void* tp, uint32_t nt, const void* sp, const void* idx, const void* tab, void* out, int grid, void* s) { hipLaunchKernelGGL(e_tiled<1024>, dim3(grid), dim3(1024), 0, (hipStream_t)s, (const uint32_t*)tp, nt, (const uint32_t*)sp, (const uint32_t*)idx, (const float*)tab, (float*)out); }
void x_wavetile(const void* tp, uint32_t nt, const void* sp, const void* idx, const void* tab, void* out, int grid, void* s) { hipLaunchKernelGGL(e_wavetile, dim3(grid), dim3(256), 0, (hipStream_t)s, (const uint32_t*)tp, nt, (const uint32_t*)sp, (const uint32_t*)idx, (const float*)tab, (float*)out); }
}
