// gg_inst_dma.hip -- instantiations of the gather-GEMM kernel template (gather_gemm_kernel.h):
// fp32, direct-to-LDS staging (buffer_load ... lds), four ring slots; the deep (eight-slot) 64 x 64 instance for launches of about one workgroup per CU.
#include "gather_gemm_kernel.h"

namespace sntc {

#define INST(TM, TN, WM, WN) template __global__ void gg_kernel<TM, TN, WM, WN, true, false, false, true>(const GGArgs);
SNTC_GG_DMA_SHAPES(INST)
template __global__ void gg_kernel<1, 1, 2, 2, true, false, false, true, kDeepRing>(const GGArgs);

}  // namespace sntc
