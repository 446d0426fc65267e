// xf_tiling.h — how a compiled minibatch is cut into workgroup tiles.  The rules are local
// predicates ("does a tile start at this key / cell?") so that the host builder (xf_batch.cc)
// and the device builder (xf_batch_dev.hip) produce identical tilings with a flag + scan.
#ifndef XF_TILING_H_
#define XF_TILING_H_

#include <stdint.h>

#include "xflow_amd.h"

#if defined(__HIPCC__)
#define XF_HD __host__ __device__
#else
#define XF_HD
#endif

namespace xf {

// ---- gradient tiles: runs of consecutive keys -------------------------------------------
// All keys of a tile start inside one quantum of the occurrence stream and none is heavy, so
// a tile holds < kGradQuantum + XF_HEAVY_SEG = XF_GRAD_TILE_NNZ occurrences and <= kGradQuantum
// keys.  A heavy key (> XF_HEAVY_SEG occurrences) is a tile of its own.
static_assert(XF_GRAD_TILE_NNZ > XF_HEAVY_SEG, "a gradient tile must hold any light key");
constexpr uint32_t kGradQuantum = XF_GRAD_TILE_NNZ - XF_HEAVY_SEG;

XF_HD inline bool grad_tile_starts_at(const uint32_t *segptr, uint32_t u) {
  if (u == 0) return true;
  const uint32_t len = segptr[u + 1] - segptr[u], plen = segptr[u] - segptr[u - 1];
  if (len > XF_HEAVY_SEG || plen > XF_HEAVY_SEG) return true;
  return segptr[u] / kGradQuantum != segptr[u - 1] / kGradQuantum;
}

// ---- forward tiles: runs of consecutive (panel,row) cells of ONE panel -------------------
// `pp` points at the panel's R+1 cell offsets.  Cells longer than kCellBig are tiles of their
// own; the others start inside one quantum, so a tile holds < XF_TILE_NNZ nonzeros; empty
// cells do not advance the offsets, so the cell count is bounded separately.
constexpr uint32_t kCellBig = 256;
constexpr uint32_t kFwdQuantum = XF_TILE_NNZ - kCellBig;

XF_HD inline bool fwd_tile_starts_at(const uint32_t *pp, uint32_t r) {
  if (r == 0) return true;
  const uint32_t len = pp[r + 1] - pp[r], plen = pp[r] - pp[r - 1];
  if (len > kCellBig || plen > kCellBig) return true;
  if (r % XF_TILE_KEYS == 0) return true;
  return pp[r] / kFwdQuantum != pp[r - 1] / kFwdQuantum;
}

// number of forward panels for U unique keys (0 = no panel view)
inline uint32_t panel_count(uint32_t U, uint32_t NNZ, double slice_bytes, double min_nnz) {
  if ((double)NNZ < min_nnz || U == 0) return 0;
  const double slices = (double)U * 4.0 / slice_bytes;
  uint32_t P = 8 * (uint32_t)((slices + 7.999999) / 8.0);
  if (P < 8) P = 8;
  if (P > 256) P = 256;
  return P;
}

XF_HD inline uint32_t panel_of(uint32_t uidx, uint32_t P, uint32_t U) {
  return (uint32_t)(((uint64_t)uidx * P) / U);
}

}  // namespace xf
#endif  // XF_TILING_H_
