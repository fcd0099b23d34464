// conv_mfma_kernel instantiations of the f32 arithmetic mode (conv_inst_impl.h)
#define OSA_INST_PREC PREC_F32
#define OSA_INST_FUNC conv_tables_f32
#define OSA_INST_OUTS 0
#define OSA_INST_REDIR 1
#include "conv_inst_impl.h"
