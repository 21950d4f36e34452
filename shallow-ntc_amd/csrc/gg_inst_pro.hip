// gg_inst_pro.hip -- instantiations of the gather-GEMM kernel template (gather_gemm_kernel.h):
// fp32 with an input prologue (|x|, x^2: the GDN norm pools) on the vector loader, and the dword gather path (Cin % 16 != 0).
#include "gather_gemm_kernel.h"

namespace sntc {

#define INST(TM, TN, WM, WN)                                                    \
  template __global__ void gg_kernel<TM, TN, WM, WN, true, true>(const GGArgs); \
  template __global__ void gg_kernel<TM, TN, WM, WN, false, true>(const GGArgs);
SNTC_GG_SHAPES(INST)

}  // namespace sntc
