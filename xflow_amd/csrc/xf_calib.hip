// xf_calib.hip — known-traffic streaming kernels for calibrating the rocprofv3 HBM counters.
//
// MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x and
// "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in
// your own access pattern".  The hot-path kernels read with 4- and 8-byte lanes and write with
// 4-byte lanes, so tools/pmc_traffic.py runs these kernels over a buffer larger than the
// 256 MiB Infinity Cache in the same rocprofv3 --pmc pass and derives bytes-per-count factors
// per access width.  Measurement support only; nothing on the training path calls it.
#include <hip/hip_runtime.h>

#include "xf_common.h"

namespace {

template <typename T>
__global__ void __launch_bounds__(256) k_calib_read(const T *__restrict__ p, size_t n,
                                                    unsigned long long *sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T v = p[i];
    const unsigned int *w = reinterpret_cast<const unsigned int *>(&v);
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += w[k];
  }
  if (acc == 0x123456789abcdefull) *sink = acc;  // keep the loads alive
}

template <typename T>
__global__ void __launch_bounds__(256) k_calib_write(T *__restrict__ p, size_t n, T v) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

}  // namespace

// kind: 0 read 4 B/lane, 1 read 8 B/lane, 2 read 16 B/lane, 3 write 4 B/lane, 4 write 8 B/lane,
// 5 write 16 B/lane.  Streams `bytes` bytes of an internal scratch buffer `repeat` times.
extern "C" int xf_calib_stream(int kind, size_t bytes, int repeat) {
  XF_REQUIRE(kind >= 0 && kind <= 5 && bytes >= 4096 && repeat >= 1, "xf_calib_stream: bad argument");
  static void *buf = nullptr;
  static size_t cap = 0;
  static unsigned long long *sink = nullptr;
  bytes &= ~(size_t)4095;
  if (bytes > cap) {
    if (buf) XF_HIP(hipFree(buf));
    XF_HIP(hipMalloc(&buf, bytes));
    XF_HIP(hipMemset(buf, 1, bytes));
    cap = bytes;
  }
  if (!sink) XF_HIP(hipMalloc((void **)&sink, 8));
  const dim3 g(4096), b(256);
  for (int r = 0; r < repeat; ++r) {
    switch (kind) {
      case 0: hipLaunchKernelGGL(k_calib_read<uint32_t>, g, b, 0, 0, (const uint32_t *)buf, bytes / 4, sink); break;
      case 1: hipLaunchKernelGGL(k_calib_read<uint64_t>, g, b, 0, 0, (const uint64_t *)buf, bytes / 8, sink); break;
      case 2: hipLaunchKernelGGL(k_calib_read<uint4>, g, b, 0, 0, (const uint4 *)buf, bytes / 16, sink); break;
      case 3: hipLaunchKernelGGL(k_calib_write<uint32_t>, g, b, 0, 0, (uint32_t *)buf, bytes / 4, 7u); break;
      case 4: hipLaunchKernelGGL(k_calib_write<uint64_t>, g, b, 0, 0, (uint64_t *)buf, bytes / 8, (uint64_t)7); break;
      case 5: hipLaunchKernelGGL(k_calib_write<uint4>, g, b, 0, 0, (uint4 *)buf, bytes / 16, make_uint4(7, 7, 7, 7)); break;
    }
    XF_HIP(hipGetLastError());
  }
  XF_HIP(hipDeviceSynchronize());
  return XF_OK;
}
