// The tail of a backward pass: every short reduction that is left when the last data gradient is done, in one launch.
// Part of the fp32 / bf16 pair-step path of libssp_hip.so (include/ssp_hip.h: ssp_backward, ssp_pair_step).
#pragma once
#include "conv1x1_group.hip.h"
#include "conv_wino.hip.h"
#include "sem_kernels.hip.h"

namespace sspk {

// The short reductions at the end of a backward pass in ONE launch (each of them alone is a dependent launch of 20-140 us whose
// blocks wait on a few loads in flight): the Winograd weight-gradient slabs (WredJobs), the slabs of the grouped pointwise weight
// gradient (wgrad1x1_reduce_block) and the column sums of the segmentation gradient = convSout's bias gradient (colsum_block).
// The latency-bound ones come first in the grid.
struct TailJobs {
  G1WArgs g1;
  int g1_blocks;               // 0: none pending
  const float* cs_m[2]; float* cs_out;
  int cs_rows, cs_C, cs_cs;
  int cs_blocks, cs_views;     // blocks per view; 0 views: none pending
};
__global__ __launch_bounds__(256) void wgrad_wino_reduce_multi_kernel(const WredJobs J, const TailJobs T) {
  __shared__ __attribute__((aligned(16))) float red[4][WC][64];
  int b = (int)blockIdx.x;
  if (b < T.g1_blocks) { wgrad1x1_reduce_block(T.g1, b); return; }
  b -= T.g1_blocks;
  if (b < T.cs_blocks * T.cs_views) {
    const int v = b / T.cs_blocks;
    colsum_block(T.cs_m[v], T.cs_out, T.cs_rows, T.cs_C, T.cs_cs, b - v * T.cs_blocks, reinterpret_cast<float4(*)[64]>(&red[0][0][0]));
    return;
  }
  b -= T.cs_blocks * T.cs_views;
  // last job first: the encoder's first layers come last in a backward pass and have the most splits per block
  b = J.j[J.n - 1].block0 + J.j[J.n - 1].nblocks - 1 - b;
  int k = 0;
  while (k + 1 < J.n && b >= J.j[k + 1].block0) ++k;
  const WredJob& q = J.j[k];
  if (q.f4 == 2) wgrad_fused12_reduce_block(q.partial, q.dw, q.cin, q.cout, q.ncob, q.nsplit, b - q.block0, &red[0][0][0]);
  else if (q.f4) wgrad_wino4_reduce_block(q.partial, q.dw, q.cin, q.cout, q.ncob, q.nsplit, b - q.block0, &red[0][0][0]);
  else wgrad_wino_reduce_block(q.partial, q.dw, q.cin, q.cout, q.ncob, q.nsplit, b - q.block0, red);
}

}  // namespace sspk
