// Instantiations of the persistent, LDS-DMA pipelined form of the conv kernel (conv_kernel.h, PIPE = 1): stride-1
// f16x3 convolutions whose input is a split tensor -- the 3x3x3 layers of the 3-D aggregation networks at every
// resolution and (D = 1) the 3x3 layers of the 2-D backbones.  Own translation unit: compiles in parallel with conv3d.hip.
#include "conv_kernel.h"

namespace osa {

#define OSA_PIPE(TU, MT, NT, WM, WN, TH, TW, OUTS) conv_mfma_kernel<PREC_F16X3, 1, TU, MT, NT, WM, WN, TH, TW, 0, OUTS, 1>

// [split output ? 1 : 0].  256 voxels x 32 channels, brick 4x8x8, ping-pong operand pipeline: 140 / 161 registers, so a SIMD
// holds 3 of the 10 waves two workgroups (4 compute + 1 loader wave each) put on a CU.  (The B-ring form and the 64- / 128-
// channel tiles need more than the 168 registers that residency allows and spill.)
static void (*const g_pipe[2])(const ConvArgs) = { OSA_PIPE(1, 2, 1, 4, 1, 8, 8, 0), OSA_PIPE(1, 2, 1, 4, 1, 8, 8, 1) };

void (*pipe_kernel(int tile, int ring, int outs))(const ConvArgs) { return (tile == 0) ? g_pipe[outs ? 1 : 0] : nullptr; }

}  // namespace osa
