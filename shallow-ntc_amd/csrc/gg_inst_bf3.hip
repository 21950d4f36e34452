// gg_inst_bf3.hip -- instantiations of the gather-GEMM kernel template (gather_gemm_kernel.h):
// the round-2 bf16 x 3 experiment (operands split while they are staged): 128 x 64 and 128 x 128 tiles.
#include "gather_gemm_kernel.h"

namespace sntc {

template __global__ void gg_kernel<1, 2, 4, 1, true, false, true>(const GGArgs);
template __global__ void gg_kernel<1, 4, 4, 1, true, false, true>(const GGArgs);

}  // namespace sntc
