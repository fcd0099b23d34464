// Kernel tables of conv_mfma_kernel, one per arithmetic mode, each instantiated in a translation unit of its own
// (conv_inst_f32.hip, conv_inst_f16x3.hip, conv_inst_f16.hip: they compile in parallel; conv3d.hip holds the host side only).
#pragma once
#include "conv_kernel.h"

namespace osa {

typedef void (*ConvFn)(const ConvArgs);
// fn: ping-pong operand pipeline; fn3: B-ring pipeline (tap counts % 3 == 0) or null; fns / fns3: the same with OUTS = 1 (the output
// is a split tensor in the f16x3 mode, an fp16 tensor in the f16 mode) or null; fnb / fnbs: B operands through the LDS ring (BL = 1; f16x3 and
// f16 modes, tiles whose waves can split a step's fragments evenly) or null
struct KernelFns { ConvFn fn, fn3, fns, fns3, fnb, fnbs; };
struct ConvFnTables {
    const KernelFns* cfgs; int n_cfgs;     // conv_cfgs.def order
    const KernelFns* ks;                   // the 4 split-K tiles
    const KernelFns* deconv;               // [0] fused 8-class transposed conv + redir (<= 32 ch), [1] + redir (<= 64 ch), [2] plain, [3] 2-D (4 classes)
};
const ConvFnTables& conv_tables_f32();
const ConvFnTables& conv_tables_f16x3();
const ConvFnTables& conv_tables_f16();

}  // namespace osa
