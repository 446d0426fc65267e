// xf_cells_impl.h — what the four files of the cells path share (internal): launch constants,
// the table module's hooks, and the gradient pass's hand-over between its two files.
//   xf_cells_build.hip       cells of a minibatch (general build, item plan, the forward's copy)
//   xf_cells_fwd.hip         forward: k_lr_fwd_cells, k_lr_finalize_cells
//   xf_cells_grad.hip        gradient (+ Push): the general kernel, the several-workers passes,
//                            the choice between them
//   xf_cells_grad_dense.hip  the steady state's gradient + Push: k_lr_grad_dense
#ifndef XF_CELLS_IMPL_H_
#define XF_CELLS_IMPL_H_

#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "xf_batch.h"
#include "xf_cells.h"
#include "xf_device.h"
#include "xf_scratch.h"

namespace {

using xf::kBlk;
using xf::kChunk;
using xf::kChunkBits;
using xf::kNoDump;
using xf::kSliceMax;
using xf::kWinMax;
using xf::kRowMask;
using xf::kTagShift;
using xf::kTagMask;

constexpr int kBlock = 256;
#ifndef XF_FWD_BLOCK
#define XF_FWD_BLOCK 1024
#endif
#ifndef XF_FWD_GROUPS
#define XF_FWD_GROUPS 256
#endif
constexpr int kFwdBlock = XF_FWD_BLOCK;  // one forward workgroup per CU (its LDS holds a row window)
constexpr uint32_t kFwdGroups = XF_FWD_GROUPS;  // workgroup slots of the chip at that size

inline int grid_for(size_t n, int block = kBlock) {
  size_t g = (n + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}
#define XF_GRID_STRIDE(i, n)                                                     \
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)(n); \
       i += (size_t)gridDim.x * blockDim.x)
}  // namespace

namespace xf {

const TableDev &table_dev(const xf_table *t);
void table_note_write(xf_table *t);
uint64_t table_uid(const xf_table *t);
uint64_t table_epoch(const xf_table *t);
int table_resolve_any(xf_table *t, const uint64_t *d_keys, size_t n, uint32_t *d_rows,
                      hipStream_t s, bool allow_grow);

// the gradient + Push of the steady state (xf_cells_grad_dense.hip): k_lr_grad_dense<opt, var>
// over the cells' work items; var = 0 (byte-masked stores) or kDenseFullStore
constexpr uint32_t kDenseWin = 4;  // row windows whose cell bounds the kernel keeps in registers
enum { kDenseCompact = 1, kDensePrefetch = 2, kDenseWide = 4,
       // timing experiments only (WRONG results; -DXF_EXPERIMENTS, never by default):
       kDiagNoStore = 8,    // the optimizer steps run, nothing is stored
       kDiagNoUpdate = 16,  // accumulate phase alone
       kDiagNoAccum = 32,   // update phase alone (every row of the chunk takes a step, g = 0)
       kDiagCopy = 64,      // with kDiagNoAccum: the state is loaded and stored, no arithmetic
       kDenseFullStore = 128,    // every row of the chunk is stored back, touched or not (whole
                                 // lines instead of byte-masked ones; same table afterwards)
       kDenseQuad = 256 };       // four chunks per workgroup (see the kernel)
int cells_launch_grad_dense(int opt, int var, const xf_cells *c, const TableDev &T,
                            const float *d_loss, hipStream_t s);
}  // namespace xf

#endif  // XF_CELLS_IMPL_H_
