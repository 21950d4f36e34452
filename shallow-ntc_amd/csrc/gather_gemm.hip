// gather_gemm.hip -- host side of the gather GEMM: the split-K finish kernel, the table of instantiations (defined in
// gg_inst_*.hip, declared here), residency tables per device, the launch.
#include "gather_gemm_kernel.h"

namespace sntc {

// the instantiations live in their own translation units (gather_gemm_kernel.h, bottom)
#define SNTC_GG_DECL_VEC(TM, TN, WM, WN) extern template __global__ void gg_kernel<TM, TN, WM, WN, true, false>(const GGArgs);
#define SNTC_GG_DECL_PRO(TM, TN, WM, WN)                                                   \
  extern template __global__ void gg_kernel<TM, TN, WM, WN, true, true>(const GGArgs);    \
  extern template __global__ void gg_kernel<TM, TN, WM, WN, false, true>(const GGArgs);
#define SNTC_GG_DECL_DMA(TM, TN, WM, WN) extern template __global__ void gg_kernel<TM, TN, WM, WN, true, false, false, true>(const GGArgs);
SNTC_GG_SHAPES(SNTC_GG_DECL_VEC)
SNTC_GG_SHAPES(SNTC_GG_DECL_PRO)
SNTC_GG_DMA_SHAPES(SNTC_GG_DECL_DMA)
extern template __global__ void gg_kernel<1, 1, 2, 2, true, false, false, true, kDeepRing>(const GGArgs);
extern template __global__ void gg_kernel<1, 3, 4, 1, true, false, false, false, 0, true>(const GGArgs);
extern template __global__ void gg_kernel<2, 2, 2, 2, true, false, false, false, 0, false, true>(const GGArgs);
extern template __global__ void gg_kernel<1, 2, 4, 1, true, false, true>(const GGArgs);
extern template __global__ void gg_kernel<1, 4, 4, 1, true, false, true>(const GGArgs);

// Split-K finish: y = epilogue(act(sum_{s = 0..S-1} slab[g][s][m][col] + bias)), splits added in index
// order (deterministic; the K ranges depend only on the layer and the image shape, never on the batch).
__global__ void __launch_bounds__(256) gg_reduce_kernel(const GGArgs a) {
  const GGGroup G = a.g[blockIdx.y];
  const size_t total = (size_t)a.M * G.Ncol;
  const int per = a.Qh * a.Qw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / G.Ncol);
    const int col = (int)(i - (size_t)m * G.Ncol);
    float v = 0.0f;
    for (int sp = 0; sp < a.ksplit; ++sp) v += a.slab[G.slab_off + (size_t)sp * total + i];
    const unsigned ce = G.cols[col];
    const int ch = ce & 0xffff;
    const int n = m / per;
    const int rem = m - n * per;
    const int qy = rem / a.Qw + G.q0y, qx = rem - (rem / a.Qw) * a.Qw + G.q0x;
    const int oy = qy * a.sO + (int)((ce >> 24) & 0xff) - 128;
    const int ox = qx * a.sO + (int)((ce >> 16) & 0xff) - 128;
    if ((unsigned)oy >= (unsigned)a.Ho || (unsigned)ox >= (unsigned)a.Wo) continue;
    const size_t idx = (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.Cout + ch;
    v = apply_act(v + (a.bias ? a.bias[ch] : 0.0f), a.act);
    a.y[idx] = apply_epilogue1(v, a.epi, a.res, a.aux, idx);
  }
}

int gg_reduce_launch(const GGArgs& args, hipStream_t stream) {
  size_t most = 0;
  for (int gi = 0; gi < args.ngroups; ++gi) most = std::max(most, (size_t)args.M * args.g[gi].Ncol);
  int blocks = (int)std::min<size_t>((most + 255) / 256, 2048);
  hipLaunchKernelGGL(gg_reduce_kernel, dim3(blocks, args.ngroups), dim3(256), 0, stream, args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "split-K reduce launch");
  return SNTC_OK;
}

// ---------------------------------------------------------------------------------------------
// variants + launch
// ---------------------------------------------------------------------------------------------
// variant -> (TM, TN, WM, WN): 1..7 one 32-row tile x v column tiles per wave, four waves stacked (128 x 32v);
// 8: 64 x 64 (2 x 2 waves of 32 x 32); 9: 128 x 128 as 2 x 2 waves of 64 x 64; 10: 256 x 128 as four waves of 64 x 128
int gg_variant_bm(int v) { return v == 8 ? 64 : v == 10 ? 256 : 128; }
int gg_variant_bn(int v) { return v == 8 ? 64 : v >= 9 ? 128 : 32 * v; }
size_t gg_sk_slab_floats(int v) {
  const int tiles = v == 8 ? 1 : v == 9 ? 4 : v == 10 ? 8 : v;
  return (size_t)tiles * 16 * 256;
}

static size_t lds_bytes(int v) {
  return (size_t)3 * (gg_variant_bm(v) + gg_variant_bn(v)) * kStage * sizeof(float) + 2 * gg_variant_bm(v) * sizeof(int4);
}

template <int TM, int TN, int WM, int WN>
static const void* kernel_ptr(bool vec, bool pro) {
  if (vec && !pro) return reinterpret_cast<const void*>(&gg_kernel<TM, TN, WM, WN, true, false>);
  if (vec) return reinterpret_cast<const void*>(&gg_kernel<TM, TN, WM, WN, true, true>);
  return reinterpret_cast<const void*>(&gg_kernel<TM, TN, WM, WN, false, true>);
}

static const void* variant_kernel_dma(int v) {
  switch (v) {
    case 1: return reinterpret_cast<const void*>(&gg_kernel<1, 1, 4, 1, true, false, false, true>);
    case 2: return reinterpret_cast<const void*>(&gg_kernel<1, 2, 4, 1, true, false, false, true>);
    case 3: return reinterpret_cast<const void*>(&gg_kernel<1, 3, 4, 1, true, false, false, true>);
    case 4: return reinterpret_cast<const void*>(&gg_kernel<1, 4, 4, 1, true, false, false, true>);
    case 5: return reinterpret_cast<const void*>(&gg_kernel<1, 5, 4, 1, true, false, false, true>);
    case 8: return reinterpret_cast<const void*>(&gg_kernel<1, 1, 2, 2, true, false, false, true>);
    case 9: return reinterpret_cast<const void*>(&gg_kernel<2, 2, 2, 2, true, false, false, true>);
    default: return nullptr;
  }
}

static size_t lds_bytes_dma(int v) {
  return (size_t)4 * (gg_variant_bm(v) + gg_variant_bn(v)) * kStage * sizeof(float) + 2 * gg_variant_bm(v) * sizeof(int4);
}

// 8-slot ring (six stages in flight per workgroup) for launches of about one workgroup per CU or fewer, where nothing else
// hides the memory latency: the 64 x 64 tile, which is what such launches are cut into
static const void* variant_kernel_deep(int v) {
  switch (v) {
    case 8: return reinterpret_cast<const void*>(&gg_kernel<1, 1, 2, 2, true, false, false, true, kDeepRing>);
    default: return nullptr;
  }
}

static size_t lds_bytes_deep(int v) {
  return (size_t)kDeepRing * (gg_variant_bm(v) + gg_variant_bn(v)) * kStage * sizeof(float) + 2 * gg_variant_bm(v) * sizeof(int4);
}

// the ResidualBlock tail fused behind the 128 x 96 tile (3x3, N = 96 -> 1x1, 96 -> 192)
static const void* kernel_fused() {
  return reinterpret_cast<const void*>(&gg_kernel<1, 3, 4, 1, true, false, false, false, 0, true>);
}

// the column-tile-outermost twin of the 128 x 128 stream-K instance (GGArgs::order == 0; single-group plans)
static const void* kernel_colm() {
  return reinterpret_cast<const void*>(&gg_kernel<2, 2, 2, 2, true, false, false, false, 0, false, true>);
}
static bool colm_shape(int variant, bool vec, int pro, int dma) { return variant == 9 && vec && pro == SNTC_PRO_NONE && dma == 0; }

static const void* variant_kernel_bf3(int v) {
  switch (v) {
    case 2: return reinterpret_cast<const void*>(&gg_kernel<1, 2, 4, 1, true, false, true>);
    case 4: return reinterpret_cast<const void*>(&gg_kernel<1, 4, 4, 1, true, false, true>);
    default: return nullptr;
  }
}

static size_t lds_bytes_bf3(int v) {
  return (size_t)3 * (gg_variant_bm(v) + gg_variant_bn(v)) * 96 + 2 * gg_variant_bm(v) * sizeof(int4);
}

static const void* variant_kernel(int v, bool vec, bool pro) {
  switch (v) {
    case 1: return kernel_ptr<1, 1, 4, 1>(vec, pro);
    case 2: return kernel_ptr<1, 2, 4, 1>(vec, pro);
    case 3: return kernel_ptr<1, 3, 4, 1>(vec, pro);
    case 4: return kernel_ptr<1, 4, 4, 1>(vec, pro);
    case 5: return kernel_ptr<1, 5, 4, 1>(vec, pro);
    case 6: return kernel_ptr<1, 6, 4, 1>(vec, pro);
    case 7: return kernel_ptr<1, 7, 4, 1>(vec, pro);
    case 8: return kernel_ptr<1, 1, 2, 2>(vec, pro);
    case 9: return kernel_ptr<2, 2, 2, 2>(vec, pro);
    case 10: return kernel_ptr<2, 4, 4, 1>(vec, pro);
    default: return nullptr;
  }
}

// Residency tables, one per device, filled once per process (std::call_once): every thread and every plan sees the same
// schedule whatever thread created the plan, and switching devices costs a table lookup.
constexpr int kMaxDevices = 16;
struct DeviceTables {
  std::once_flag once;
  int rc = SNTC_OK;
  int num_cus = 0;
  int resident[kNumVariants + 1][3] = {};      // per (variant, {vec, vec+pro, gather}) workgroups per device
  int resident_bf3[kNumVariants + 1] = {};
  int resident_dma[kNumVariants + 1] = {};
  int resident_deep[kNumVariants + 1] = {};
  int resident_fused = 0;
  bool colm_ok = false;                        // the column-major twin of variant 9 is as resident as variant 9 itself
  int* status = nullptr;                       // sticky status word (device memory)
};
static DeviceTables g_dev[kMaxDevices];

static int fill_tables(DeviceTables& T, int dev) {
  hipDeviceProp_t prop;
  SNTC_HIP(hipGetDeviceProperties(&prop, dev));
  T.num_cus = prop.multiProcessorCount;
  for (int v = 1; v <= kNumVariants; ++v) {
    for (int k = 0; k < 3; ++k) {
      const void* fn = variant_kernel(v, k < 2, k >= 1);
      SNTC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(v)));
      int per_cu = 0;
      SNTC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_bytes(v)));
      // the API can answer one workgroup per CU high near an SGPR allocation edge (MI355X_MICROARCH.md, Residency):
      // stream-K needs every worker resident, so stay at or below 8 and keep the LDS bound exact
      per_cu = std::max(1, std::min({per_cu, 8, (int)(163840 / lds_bytes(v))}));
      T.resident[v][k] = per_cu * T.num_cus;
    }
  }
  for (int v : {1, 2, 3, 4, 5, 8, 9}) {
    const void* fn = variant_kernel_dma(v);
    SNTC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_dma(v)));
    int per_cu = 0;
    SNTC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_bytes_dma(v)));
    T.resident_dma[v] = std::max(1, std::min({per_cu, 8, (int)(163840 / lds_bytes_dma(v))})) * T.num_cus;
  }
  for (int v : {8}) {
    const void* fn = variant_kernel_deep(v);
    SNTC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_deep(v)));
    int per_cu = 0;
    SNTC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_bytes_deep(v)));
    T.resident_deep[v] = std::max(1, std::min({per_cu, 8, (int)(163840 / lds_bytes_deep(v))})) * T.num_cus;
  }
  {
    const void* fn = kernel_fused();
    SNTC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(3)));
    int per_cu = 0;
    SNTC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_bytes(3)));
    T.resident_fused = std::max(1, std::min({per_cu, 8, (int)(163840 / lds_bytes(3))})) * T.num_cus;
  }
  {
    const void* fn = kernel_colm();
    SNTC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(9)));
    int per_cu = 0;
    SNTC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_bytes(9)));
    // stream-K sizes its worker count from the strip-major instance's residency: the twin must hold as many
    T.colm_ok = std::max(1, std::min({per_cu, 8, (int)(163840 / lds_bytes(9))})) * T.num_cus >= T.resident[9][0];
  }
  for (int v : {2, 4}) {
    const void* fn = variant_kernel_bf3(v);
    SNTC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_bf3(v)));
    int per_cu = 0;
    SNTC_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds_bytes_bf3(v)));
    T.resident_bf3[v] = std::max(1, std::min({per_cu, 8, (int)(163840 / lds_bytes_bf3(v))})) * T.num_cus;
  }
  SNTC_HIP(hipMalloc(&T.status, 2 * sizeof(int)));      // [0] the sticky word, [1] where sntc_conv_status's exchange returns it
  SNTC_HIP(hipMemset(T.status, 0, 2 * sizeof(int)));
  return SNTC_OK;
}

static DeviceTables* current_tables() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  DeviceTables& T = g_dev[dev];
  std::call_once(T.once, [&] { T.rc = fill_tables(T, dev); });
  return T.rc == SNTC_OK ? &T : nullptr;
}

int gg_init() {
  int dev = 0;
  SNTC_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(SNTC_ERR_UNSUPPORTED, "device index beyond the residency tables");
  DeviceTables& T = g_dev[dev];
  std::call_once(T.once, [&] { T.rc = fill_tables(T, dev); });
  return T.rc;
}

int gg_resident_blocks(int variant, bool vec, bool pro) {
  const DeviceTables* T = current_tables();
  if (variant < 1 || variant > kNumVariants || !T) return 0;
  return T->resident[variant][!vec ? 2 : (pro ? 1 : 0)];
}

int gg_resident_blocks_dma(int variant) {
  const DeviceTables* T = current_tables();
  return variant >= 1 && variant <= kNumVariants && variant_kernel_dma(variant) && T ? T->resident_dma[variant] : 0;
}

int gg_resident_blocks_deep(int variant) {
  const DeviceTables* T = current_tables();
  return variant >= 1 && variant <= kNumVariants && variant_kernel_deep(variant) && T ? T->resident_deep[variant] : 0;
}

int gg_num_cus() {
  const DeviceTables* T = current_tables();
  return T ? T->num_cus : 0;
}

int* gg_status_word() {
  const DeviceTables* T = current_tables();
  return T ? T->status : nullptr;
}

int gg_resident_blocks_fused() {
  const DeviceTables* T = current_tables();
  return T ? T->resident_fused : 0;
}

// the column-major stream-K twin exists for this launch shape on the current device
bool gg_colm_available(int variant, bool vec, int pro, int dma) {
  const DeviceTables* T = current_tables();
  return T && T->colm_ok && colm_shape(variant, vec, pro, dma);
}

int gg_resident_blocks_bf3(int variant) {
  const DeviceTables* T = current_tables();
  return (variant == 2 || variant == 4) && T ? T->resident_bf3[variant] : 0;
}

int gg_launch(int variant, bool vec, const GGArgs& args, int nblocks, hipStream_t stream) {
  const bool pro = args.pro != SNTC_PRO_NONE;
  const bool deep = args.dma == 2 && vec && !pro && !args.bf3 && variant_kernel_deep(variant);
  const bool dma = !deep && args.dma && vec && !pro && !args.bf3 && variant_kernel_dma(variant);
  const bool fused = args.w2f != nullptr;
  if (fused && (variant != 3 || !vec || pro || args.bf3 || args.ksplit != 1))
    return fail(SNTC_ERR_UNSUPPORTED, "fused ResidualBlock tail: 128 x 96 vector instance, no prologue, no split-K");
  const bool colm = !fused && !args.bf3 && args.sk && args.order == 0 && args.ngroups == 1 && colm_shape(variant, vec, args.pro, args.dma);
  if (!fused && !args.bf3 && args.sk && args.order == 0 && !colm)
    return fail(SNTC_ERR_UNSUPPORTED, "column-major stream-K order: single-group plans on the 128 x 128 register-staged vector instance only");
  const void* fn = fused ? kernel_fused() : args.bf3 ? variant_kernel_bf3(variant) : deep ? variant_kernel_deep(variant)
                   : dma ? variant_kernel_dma(variant) : colm ? kernel_colm() : variant_kernel(variant, vec, pro || !vec);
  if (!fn) return fail(SNTC_ERR_UNSUPPORTED, "unknown gather-GEMM tile variant");
  if (args.bf3 && (pro || !vec)) return fail(SNTC_ERR_UNSUPPORTED, "bf16 x 3 mode: vector path without prologue only");
  GGArgs a = args;
#ifdef SNTC_DIAG
  if (const char* e = getenv("SNTC_GG_DBG")) a.dbg = atoi(e);   // diagnostic builds only (make DIAG=1): results are WRONG with it
#endif
  void* params[] = {&a};
  hipError_t e = hipLaunchKernel(fn, dim3(nblocks), dim3(256), params,
                                 args.bf3 ? lds_bytes_bf3(variant) : deep ? lds_bytes_deep(variant) : dma ? lds_bytes_dma(variant) : lds_bytes(variant), stream);
  if (e != hipSuccess) return hip_fail(e, "gather-GEMM launch");
  return SNTC_OK;
}

}  // namespace sntc
