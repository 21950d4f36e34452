// gg_inst_vec.hip -- instantiations of the gather-GEMM kernel template (gather_gemm_kernel.h):
// fp32, vector loader, no prologue, register-staged: the ten tile variants and the column-major stream-K twin of the 128 x 128 one
// (the instances that carry the decode and most of the encoder).
#include "gather_gemm_kernel.h"

namespace sntc {

#define INST(TM, TN, WM, WN) template __global__ void gg_kernel<TM, TN, WM, WN, true, false>(const GGArgs);
SNTC_GG_SHAPES(INST)
template __global__ void gg_kernel<2, 2, 2, 2, true, false, false, false, 0, false, true>(const GGArgs);

}  // namespace sntc
