// gg_inst_fuse.hip -- instantiations of the gather-GEMM kernel template (gather_gemm_kernel.h):
// the ResidualBlock tail fused behind the 128 x 96 tile (3x3, N = 96 -> 1x1, 96 -> 192).
#include "gather_gemm_kernel.h"

namespace sntc {

template __global__ void gg_kernel<1, 3, 4, 1, true, false, false, false, 0, true>(const GGArgs);

}  // namespace sntc
