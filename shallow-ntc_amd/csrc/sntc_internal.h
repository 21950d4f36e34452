// Internal declarations shared by the HIP translation units of libsntc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include "../../include/sntc.h"

namespace sntc {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);
int zero_async(void* p, size_t bytes, hipStream_t stream);   // zero a 4-byte-aligned range on the stream with a kernel (graph-replay safe)

#define SNTC_HIP(expr)                                         \
  do {                                                         \
    hipError_t _e = (expr);                                    \
    if (_e != hipSuccess) return ::sntc::hip_fail(_e, #expr);  \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Gather-GEMM: out[m, col] = sum_{t < T} sum_{c < Cin} x[src(m, t), c] * Wp[col][t*Cin + c]
//   m    = (n, qy, qx) on the macro-pixel grid [Qh, Qw]
//   src  = (n, qy*sA + offy + ty*tstep, qx*sA + offx + tx*tstep), zero outside the image
//   col  -> (oy, ox, ch) = (qy*sO + oyoff[col], qx*sO + oxoff[col], ch[col]); skipped outside
// A forward conv is one group with T = kh*kw; a stride-s transposed conv is up to four groups of
// output phases that share a tap pattern (DESIGN.md "phase-grouped transposed convolution").
// ---------------------------------------------------------------------------------------------
constexpr int kMaxGroups = 4;

// Diagnostic switches (skip global loads / LDS writes / barriers of a K loop to see what it waits on) exist only in
// builds made with -DSNTC_DIAG (make DIAG=1); in the shipped library the tests fold to `false` at compile time and no
// environment variable is read on the launch path.
#ifdef SNTC_DIAG
#define SNTC_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define SNTC_DBG(a, bit) false
#endif

struct GGGroup {
  const float* wp;    // [NcolPad][K], K contiguous (K padded to a multiple of 16 with zeros)
  const int* taps;    // [T]   (ty << 16) | tx
  const int* cols;    // [NcolPad] ((oyoff+128) << 24) | ((oxoff+128) << 16) | ch ; -1 = padding
  int T;              // taps: a dense th x tw grid, tap t = (t / tw, t % tw)
  int tw;             // taps per row of that grid
  int K;              // padded K
  int Ncol;           // real columns
  int ntn;            // N tiles for the launched variant
  int blk0;           // static mode: first block of this group
  size_t slab_off;    // float offset of this group's split-K slabs [ksplit][M][Ncol]
  int q0y, q0x;       // origin of this group's macro-pixel grid (phase groups whose first valid q is 1)
  int steps;          // K / 16: K stages of one tile
  int tile0;          // stream-K: tiles of the preceding groups inside one row strip
  long long unit0;    // stream-K: work units (one unit = one K stage of one tile) of the preceding groups inside one strip
};

struct GGArgs {
  const float* x;
  float* y;
  const float* bias;   // [Cout] or nullptr
  const float* res;    // epilogue operand (output shape) or nullptr
  const float* aux;    // second epilogue operand or nullptr
  unsigned x_bytes;    // size of x (< 2 GiB: 32-bit buffer offsets)
  int N, H, W, Cin;
  int Qh, Qw, M;
  int Ho, Wo, Cout;
  int sA, tstep, offy, offx, sO;
  int act, epi, pro;
  int ntm;
  int ksplit;          // static mode, >= 1: number of K ranges (blocks per tile), summed by gg_reduce_kernel
  float* slab;         // split-K partial sums (workspace) or nullptr
  // Persistent stream-K mode: `nworkers` resident workgroups share the launch's work units evenly;
  // a tile cut between workers w and w + 1 is CONTINUED, not re-summed: w publishes its accumulators (sk_slab[w],
  // sk_flags[w]), w + 1 starts its fma chains from them, so every output is the same k-ordered chain as in an unsplit tile.
  int sk;              // 0: static (one K range of one tile per block); 1: stream-K
  int nworkers;
  long long units;     // sum over groups of tiles * steps
  int tps, ups;        // tiles / units per row strip (all groups): tile order is strip-major, then group, then column tile
  float* sk_slab;      // [nworkers][256 threads * 16 TN floats], raw accumulators in register layout
  int* sk_flags;       // [nworkers], zeroed on the stream before the launch
  int* status;         // sticky per-device status word (never NULL): bit 0 = a stream-K worker gave up waiting for its neighbour's
                       // hand-off (the launch's results are invalid; sntc_conv_status reports and clears it)
  // Fused ResidualBlock tail (FUSE2 instance): after the 3x3 (this launch's groups, Cout = 96, bias / act above) the 1x1
  // 96 -> Cout2 = 192 with bias2, then `epi` with res / aux, into y [M][Cout2]
  const float* w2f;    // W2 in fragment order [2 halves][3 column tiles][12 k-quads pairs][64 lanes][4], or nullptr
  const float* bias2;
  int Cout2;
  int dma;             // 1: direct-to-LDS staging (buffer_load ... lds, four ring slots) where the instantiation exists
  int bf3;             // 1: bf16 x 3 split-precision experiment (weights packed as three bf16 planes)
  int halo;            // bf3_gemm.hip: 1 = patch staging (one activation patch per channel slab shared by its taps); needs sA == 1
                       // and every group's patch within bf3p_patch_rows_max()
  int order;           // stream-K unit order: 1 strip-major (default), 0 column tile outermost (bf3_gemm.hip: A/B switch;
                       // gather_gemm.hip: the COLM twin of the 128 x 128 instance, single-group plans whose weights exceed an XCD's L2)
  int dbg;             // -DSNTC_DIAG builds only (SNTC_GG_DBG): 1 skip global loads, 2 skip LDS writes, 4 skip barriers,
                       // 8 skip fragment reads, 16 skip the epilogue's stores, 32 its residual loads, 64 the MFMAs, 128 the fused
                       // ResidualBlock tail's second contraction -- to see what a launch waits on; results are meaningless with any bit set
  int ngroups;
  GGGroup g[kMaxGroups];
};

// variant ids (BM x BN):  1..7 -> 128 x 32*v ;  8 -> 64 x 64 ;  9 -> 128 x 128 (64 x 64 per wave) ;  10 -> 256 x 128
constexpr int kNumVariants = 10;
constexpr int kStage = 16;         // K depth of one pipeline stage
int gg_variant_bm(int v);
int gg_variant_bn(int v);
int gg_launch(int variant, bool vec, const GGArgs& args, int nblocks, hipStream_t stream);
int gg_reduce_launch(const GGArgs& args, hipStream_t stream);
int gg_init();   // sets the dynamic-LDS attribute on every instantiation (idempotent), measures occupancy
int gg_resident_blocks(int variant, bool vec, bool pro);   // workgroups of this instantiation the device keeps resident
int gg_resident_blocks_deep(int variant);                   // same for the deep-ring (8-slot) direct-to-LDS instances; 0 if none
int gg_num_cus();
int* gg_status_word();                                      // device pointer of the current device's sticky status word
int gg_resident_blocks_dma(int variant);                    // same for the direct-to-LDS instantiations
int gg_resident_blocks_fused();
bool gg_colm_available(int variant, bool vec, int pro, int dma);                             // the FUSE2 instance (variant 3)
int gg_resident_blocks_bf3(int variant);                    // same for the bf16 x 3 instantiations (variants 2 and 4)
size_t gg_sk_slab_floats(int variant);                     // per-worker accumulator slab of the stream-K hand-off

// ---- pre-split bf16 x 3 gather GEMM (bf3_gemm.hip): variants 11 (256 x 256) and 12 (256 x 128), 512 threads, one workgroup per CU
constexpr int kBf3PatchRounds = 5;     // patch staging: at most 5 rounds of 512 16-B chunks = 426 rows of 96 B
constexpr int kBf3DeepRing = 3;        // weight ring slots of the patch-staging 256 x 128 instance (fragments double-buffered)
int bf3p_patch_rows_max();
int bf3p_variant_bm(int v);
int bf3p_variant_bn(int v);
size_t bf3p_sk_slab_floats(int v);
int bf3p_init();
int bf3p_launch(int variant, const GGArgs& args, int nblocks, hipStream_t stream);

// ---- deep-factorized prior (entropy.hip, sga.hip) ----
constexpr int kMaxW = 4;   // max hidden width
constexpr int kMaxL = 5;   // max affine layers
struct DFDesc {
  int nl;
  int w[kMaxL + 1];
  int off_m[kMaxL], off_b[kMaxL], off_f[kMaxL];
  int stride;   // floats per channel record: softplus(matrix), bias, tanh(factor) per layer
};

}  // namespace sntc

struct sntc_prior {
  int channels = 0;
  sntc::DFDesc d{};
  float* rec = nullptr;
};
