// conv_mfma_kernel instantiations of the f16x3 arithmetic mode (conv_inst_impl.h)
#define OSA_INST_PREC PREC_F16X3
#define OSA_INST_FUNC conv_tables_f16x3
#define OSA_INST_OUTS 1
#define OSA_INST_REDIR 1
#include "conv_inst_impl.h"
