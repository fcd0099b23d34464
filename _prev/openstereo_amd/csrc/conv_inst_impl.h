// Included by conv_inst_<mode>.hip after defining
//   OSA_INST_PREC   PREC_F32 | PREC_F16X3 | PREC_F16
//   OSA_INST_FUNC   conv_tables_f32 | conv_tables_f16x3 | conv_tables_f16
//   OSA_INST_OUTS   1: instantiate the OUTS = 1 variants (split / fp16 outputs)
//   OSA_INST_REDIR  1: instantiate the fused-redir transposed convs
#include "conv_inst.h"

namespace osa {

#define OSA_K(NCLS, TU, MT, NT, WM, WN, TH, TW, REDIR, OUTS, KS) conv_mfma_kernel<OSA_INST_PREC, NCLS, TU, MT, NT, WM, WN, TH, TW, REDIR, OUTS, 0, KS>
#if OSA_INST_OUTS
#define OSA_KO(NCLS, TU, MT, NT, WM, WN, TH, TW, REDIR, KS) OSA_K(NCLS, TU, MT, NT, WM, WN, TH, TW, REDIR, 1, KS)
#else
#define OSA_KO(NCLS, TU, MT, NT, WM, WN, TH, TW, REDIR, KS) nullptr
#endif
#define OSA_RING_0(x) nullptr
#define OSA_RING_1(x) x
#define OSA_RING_2(x) nullptr
// RING = 2: a tile that exists in the ring form (BL = 1) only
#define OSA_BASE_0(x) x
#define OSA_BASE_1(x) x
#define OSA_BASE_2(x) nullptr
// BL = 1 variants (B operands through the LDS ring): every mode but exact f32, tiles with (2 * JO * WN * NT) % (WM * WN) == 0
template <int NCLS, int MT, int NT, int WM, int WN, int TH, int TW, int REDIR, int OUTS>
static constexpr ConvFn bl_fn() {
    if constexpr (OSA_INST_PREC != PREC_F32 && (OUTS == 0 || OSA_INST_OUTS) && (2 * JO * WN * NT) % (WM * WN) == 0)
        return conv_mfma_kernel<OSA_INST_PREC, NCLS, 1, MT, NT, WM, WN, TH, TW, REDIR, OUTS, 0, 1, 1>;
    else return nullptr;
}
#define OSA_KB(NCLS, MT, NT, WM, WN, TH, TW, REDIR, OUTS) bl_fn<NCLS, MT, NT, WM, WN, TH, TW, REDIR, OUTS>()

static const KernelFns g_cfg_fns[] = {
#define OSA_CFG_X(RING, MT, NT, WM, WN, TH, TW)                                                                          \
    { OSA_BASE_##RING(OSA_K(1, 1, MT, NT, WM, WN, TH, TW, 0, 0, 1)), OSA_RING_##RING(OSA_K(1, 3, MT, NT, WM, WN, TH, TW, 0, 0, 1)),       \
      OSA_BASE_##RING(OSA_KO(1, 1, MT, NT, WM, WN, TH, TW, 0, 1)), OSA_RING_##RING(OSA_KO(1, 3, MT, NT, WM, WN, TH, TW, 0, 1)),           \
      OSA_KB(1, MT, NT, WM, WN, TH, TW, 0, 0), OSA_KB(1, MT, NT, WM, WN, TH, TW, 0, 1) },
#define OSA_KS_X(MT, NT, WM, WN, TH, TW, KS)
#include "conv_cfgs.def"
#undef OSA_CFG_X
#undef OSA_KS_X
};

static const KernelFns g_ks_fns[] = {
#define OSA_CFG_X(RING, MT, NT, WM, WN, TH, TW)
#define OSA_KS_X(MT, NT, WM, WN, TH, TW, KS)                                                                             \
    { OSA_K(1, 1, MT, NT, WM, WN, TH, TW, 0, 0, KS), OSA_K(1, 3, MT, NT, WM, WN, TH, TW, 0, 0, KS), nullptr, nullptr, nullptr, nullptr },
#include "conv_cfgs.def"
#undef OSA_CFG_X
#undef OSA_KS_X
};

// fused transposed convs: 128 input-resolution positions x 32 channels x 8 parity classes per workgroup (brick 4x4x8); 2-D: 4 classes, 8x16
static const KernelFns g_deconv_fns[] = {
#if OSA_INST_REDIR
    { OSA_K(8, 1, 1, 1, 4, 1, 4, 8, 1, 0, 1), nullptr, OSA_KO(8, 1, 1, 1, 4, 1, 4, 8, 1, 1), nullptr, OSA_KB(8, 1, 1, 4, 1, 4, 8, 1, 0), OSA_KB(8, 1, 1, 4, 1, 4, 8, 1, 1) },
    { OSA_K(8, 1, 1, 1, 4, 1, 4, 8, 2, 0, 1), nullptr, OSA_KO(8, 1, 1, 1, 4, 1, 4, 8, 2, 1), nullptr, OSA_KB(8, 1, 1, 4, 1, 4, 8, 2, 0), OSA_KB(8, 1, 1, 4, 1, 4, 8, 2, 1) },
#else
    { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr }, { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr },
#endif
    { OSA_K(8, 1, 1, 1, 4, 1, 4, 8, 0, 0, 1), nullptr, OSA_KO(8, 1, 1, 1, 4, 1, 4, 8, 0, 1), nullptr, OSA_KB(8, 1, 1, 4, 1, 4, 8, 0, 0), OSA_KB(8, 1, 1, 4, 1, 4, 8, 0, 1) },
    { OSA_K(4, 1, 1, 1, 4, 1, 8, 16, 0, 0, 1), nullptr, nullptr, nullptr, OSA_KB(4, 1, 1, 4, 1, 8, 16, 0, 0), nullptr },
};

const ConvFnTables& OSA_INST_FUNC() {
    static const ConvFnTables t = { g_cfg_fns, (int)(sizeof(g_cfg_fns) / sizeof(g_cfg_fns[0])), g_ks_fns, g_deconv_fns };
    return t;
}

}  // namespace osa
