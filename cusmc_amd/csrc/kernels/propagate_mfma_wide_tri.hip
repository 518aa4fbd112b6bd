// The proposal kernels of propagate_mfma_wide.hip instantiated for a LOWER TRIANGULAR Q
// (launch_propagate_mfma_wide_tri): a translation unit of its own so that the two sets of instantiations compile side
// by side.
#define CUSMC_TRIQ 1
#include "propagate_mfma_wide.hip"
