// conv_mfma_kernel instantiations of the f16 arithmetic mode (conv_inst_impl.h)
#define OSA_INST_PREC PREC_F16
#define OSA_INST_FUNC conv_tables_f16
#define OSA_INST_OUTS 1
#define OSA_INST_REDIR 0
#include "conv_inst_impl.h"
