/* sntc.h -- C ABI of the MI355X-native shallow-ntc hot path (libsntc_hip.so).
 *
 * The reference (mandt-lab/shallow-ntc) has no FFI: its arithmetic is TensorFlow /
 * tensorflow-compression ops called from Python.  Each entry point below replaces ONE of those
 * dependency ops at the place the reference calls it; the citations are reference file:line.
 * INTEGRATION.md shows the ctypes binding a maintainer would add to the reference.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++ or torch types cross the boundary.
 *   - every function returns an int status (SNTC_OK == 0); sntc_last_error() gives the
 *     thread-local message of the last failure.  No exceptions cross the boundary.
 *   - all tensor pointers are DEVICE pointers to dense NHWC float32 unless stated; the caller
 *     owns every buffer.  The library owns only what a plan / prior object packs at creation.
 *   - every launch takes a hipStream_t (passed as void*) and is asynchronous w.r.t. the host.
 *   - one process per GPU; objects are bound to the device current at creation.
 */
#ifndef SNTC_H_
#define SNTC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNTC_VERSION 100

/* status codes (mirroring the failure classes of the reference: Python ValueError for shapes /
 * configs, tf InvalidArgumentError from check_numerics at mshyper/models.py:308-309,356) */
enum {
  SNTC_OK = 0,
  SNTC_ERR_BAD_SHAPE = 1,      /* inconsistent sizes / null pointer                           */
  SNTC_ERR_UNSUPPORTED = 2,    /* configuration outside what the kernels implement            */
  SNTC_ERR_HIP = 3,            /* HIP runtime error (message carries hipGetErrorString)       */
  SNTC_ERR_NONFINITE = 4,      /* NaN/Inf where the reference's check_numerics would raise    */
  SNTC_ERR_NO_DEVICE = 5       /* no gfx950 device visible                                    */
};

const char* sntc_last_error(void);
int sntc_version(void);
/* Number of visible HIP devices, 0 if none; never fails. */
int sntc_device_count(void);
/* Writes the gcnArchName of `device` ("gfx950:sramecc+:xnack-") into buf. */
int sntc_device_arch(int device, char* buf, size_t buflen);

/* ------------------------------------------------------------------------------------------
 * Convolution plans
 *   replaces tf.keras.layers.Conv2D / Conv2DTranspose (padding="SAME") and tfc.SignalConv2D
 *   (padding="same_zeros") at common/transforms.py:81-90,101-134,152-155,172-175,186-231,
 *   240-262,284-287,307-313,331-357,371-373 and common/elic.py:61-63,92-93,136-140,253-270;
 *   and the 1x1 norm-pool contraction of GDN / GDN1 at common/transforms.py:8-63,150,170.
 * ------------------------------------------------------------------------------------------ */
enum {                       /* sntc_conv_desc.kind */
  SNTC_CONV2D = 0,           /* Keras Conv2D SAME, kernel [kh,kw,Cin,Cout], cross-correlation */
  SNTC_CONV2D_TRANSPOSE = 1, /* Keras Conv2DTranspose SAME, kernel [kh,kw,Cout,Cin]           */
  SNTC_SIGNAL_DOWN = 2,      /* tfc.SignalConv2D corr=True, strides_down, same_zeros, [kh,kw,Cin,Cout] */
  SNTC_SIGNAL_UP = 3         /* tfc.SignalConv2D corr=False, strides_up, same_zeros, [kh,kw,Cin,Cout]  */
};
enum {                       /* activation applied after bias (Keras `activation=`) */
  SNTC_ACT_NONE = 0, SNTC_ACT_RELU = 1, SNTC_ACT_LEAKY_RELU = 2 /* alpha 0.2 */, SNTC_ACT_SIGMOID = 3
};
enum {                       /* input transform applied to x while it is staged (GDN norm pool) */
  SNTC_PRO_NONE = 0, SNTC_PRO_ABS = 1, SNTC_PRO_SQUARE = 2
};
enum {                       /* epilogue, v = act(conv + bias) */
  SNTC_EPI_STORE = 0,        /* y = v                                                          */
  SNTC_EPI_ADD = 1,          /* y = v + res                (ResidualBlock skip, elic.py:66-68)  */
  SNTC_EPI_GATE = 2,         /* y = res + aux * v          (SimpleAttention, elic.py:97-100)    */
  SNTC_EPI_RES_DIV = 3,      /* y = res / v                (GDN1 forward, transforms.py:63)     */
  SNTC_EPI_RES_MUL = 4,      /* y = res * v                (GDN1 inverse, transforms.py:61)     */
  SNTC_EPI_RES_DIV_SQRT = 5, /* y = res / sqrt(v)          (classic tfc.GDN)                    */
  SNTC_EPI_RES_MUL_SQRT = 6, /* y = res * sqrt(v)          (classic inverse tfc.GDN)            */
  SNTC_EPI_MASK_RELU = 7,    /* y = res > 0 ? v : 0        (relu backward; res = forward output) */
  SNTC_EPI_MASK_LEAKY = 8    /* y = res >= 0 ? v : 0.2 v   (leaky-relu backward)                 */
};

typedef struct sntc_conv_desc {
  int32_t kind;        /* SNTC_CONV2D ...                                  */
  int32_t kh, kw;      /* kernel size                                      */
  int32_t stride;      /* strides (down for conv, up for transpose)        */
  int32_t cin, cout;   /* channels                                         */
  int32_t act;         /* SNTC_ACT_*                                       */
  int32_t prologue;    /* SNTC_PRO_*                                       */
  int32_t epilogue;    /* SNTC_EPI_*                                       */
  int32_t reserved[7]; /* reserved[0] != 0: the kernel array has its two channel axes swapped with respect to the kind's
                        * native layout (so that the input-gradient plan of a layer -- the adjoint kind -- packs straight
                        * from the layer's own kernel array; Keras Conv2D <-> Conv2DTranspose are each other's swap already,
                        * tfc.SignalConv2D down <-> up need the flag);
                        * reserved[1] != 0: bf16 x 3 split-precision contraction (hi + mid + lo bfloat16 terms, six cross
                        * products on the bf16 matrix cores, fp32 accumulation): ~1e-7 relative like fp32 but NOT
                        * bit-identical to it; Cin % 16 == 0, no prologue; opt-in (Model(precision="bf16x3")), never a default.
                        *   1: fp32 NHWC input, split while it is staged (the round-2 experiment);
                        *   2: the INPUT is pre-split too -- format S3, see sntc_split3 -- and the launch runs the 256-wide
                        *      direct-to-LDS kernel of csrc/bf3_gemm.hip; output fp32 NHWC; Cout % 4 == 0, epilogues
                        *      STORE / ADD / GATE / MASK_*; no split-K;
                        * reserved[2] == 1: ROW-PACKED small-Cin convolution (the RGB first layer: Keras Conv2D 5x5 / 2, Cin = 3,
                        *   common/elic.py:147, common/transforms.py:183): kw * Cin <= 16, fp32, no prologue.  The caller applies
                        *   the SAME padding itself (sntc_pad_zero) and the plan convolves VALID on that tensor: every kernel row
                        *   is one 16-deep K stage whose 64-B activation slab is ONE contiguous 16-B-per-lane read starting at
                        *   the window's first pixel -- the vector loader of the wide layers instead of 75 dword gathers per
                        *   output; out size = (padded - k) / stride + 1;
                        * the rest must be 0 */
} sntc_conv_desc;

typedef struct sntc_conv_plan sntc_conv_plan;

/* Packs `weight` (device pointer, layout per desc.kind) and `bias` (device [cout] or NULL) into
 * the gather-GEMM layout on `stream`.  The plan keeps its own packed copy; weight/bias may be
 * freed after the stream work completes. */
int sntc_conv_plan_create(const sntc_conv_desc* desc, const float* weight, const float* bias,
                          void* stream, sntc_conv_plan** plan);
void sntc_conv_plan_destroy(sntc_conv_plan* plan);
/* Re-pack an existing plan from new device weights (and bias: present iff the plan has one); tables and tile
 * choices are kept.  The training step calls this after every optimizer update. */
int sntc_conv_plan_update(sntc_conv_plan* plan, const float* weight, const float* bias, void* stream);
/* Every plan of a model re-packed by ONE launch: the training step (reference common/train_lib.py:185-215, one
 * optimizer.apply_gradients per step over all variables) changes every kernel at once, and ~170 plans x (pack + bias copy)
 * is ~400 tiny launches otherwise.  The group remembers, per plan, the device arrays its weights (and bias; NULL iff the
 * plan has none) are read from -- views of the trainer's flat variable store, which never move -- and
 * sntc_plan_group_update() does what sntc_conv_plan_update() would do for each of them, bit for bit.  The plans must
 * outlive the group. */
typedef struct sntc_plan_group sntc_plan_group;
int sntc_plan_group_create(sntc_conv_plan* const* plans, const float* const* weights, const float* const* biases, int count,
                           void* stream, sntc_plan_group** group);
int sntc_plan_group_update(const sntc_plan_group* group, void* stream);
void sntc_plan_group_destroy(sntc_plan_group* group);
/* Output spatial size for an input of h x w. */
int sntc_conv_out_shape(const sntc_conv_plan* plan, int h, int w, int* ho, int* wo);
/* Algorithmic 2*MAC FLOPs of one forward call (dense count, as tf.profiler counts them). */
int64_t sntc_conv_flops(const sntc_conv_plan* plan, int n, int h, int w);
/* y[n,ho,wo,cout] = epilogue(act(conv(prologue(x[n,h,w,cin])) + bias), res, aux).
 * res / aux are NHWC tensors of the OUTPUT shape (NULL unless the epilogue uses them). */
int sntc_conv_forward(const sntc_conv_plan* plan, const float* x, int n, int h, int w, float* y,
                      const float* res, const float* aux, void* workspace, size_t workspace_bytes, void* stream);
/* Device scratch the call above needs.  Large launches run on persistent stream-K workers that hand a tile's
 * accumulators to the next worker through this scratch (a few tens of MB, contents irrelevant before and after the call).
 * Layers that offer few output tiles per image but a long contraction (hyper transforms, SGA input-gradient
 * convolutions) are split along K into slabs that a second kernel adds in a fixed order; the split depends on the
 * layer and the image shape only.  Either way results are bit-identical for any batch size. */
int64_t sntc_conv_workspace_bytes(const sntc_conv_plan* plan, int n, int h, int w);
/* The tail of a ResidualBlock (reference common/elic.py:57-68: conv3x3(c/2 -> c/2, relu) -> conv1x1(c/2 -> c), + x) in ONE
 * launch for c = 192: `first` is the 3x3 plan (stride-1 forward convolution, 96 output channels, plain store), `second` the
 * 1x1 plan 96 -> 192 (no activation; its epilogue -- SNTC_EPI_ADD for the skip -- and `res` / `aux` apply to the final output
 * y[n,ho,wo,192]).  The 96-channel intermediate stays in registers; the result is bit-identical to sntc_conv_forward(first)
 * followed by sntc_conv_forward(second).  sntc_conv_fusable() says whether a pair qualifies (1) or not (0);
 * sntc_conv_fused_workspace_bytes() is the scratch this call needs (as sntc_conv_workspace_bytes). */
int sntc_conv_fusable(const sntc_conv_plan* first, const sntc_conv_plan* second);
int64_t sntc_conv_fused_workspace_bytes(const sntc_conv_plan* first, int n, int h, int w);
int sntc_conv_forward_fused(const sntc_conv_plan* first, const sntc_conv_plan* second, const float* x, int n, int h, int w, float* y,
                            const float* res, const float* aux, void* workspace, size_t workspace_bytes, void* stream);
/* ------------------------------------------------------------------------------------------
 * The whole ResidualBlock (reference common/elic.py:41-68; inside SimpleAttention :85-100 and ElicAnalysis :147-163) in
 * ONE launch:  y = x + conv1x1_{c/2 -> c}( relu(conv3x3_{c/2 -> c/2}( relu(conv1x1_{c -> c/2}(x)) )) ),  NHWC fp32.
 * A workgroup owns an 8 x 32 pixel tile: the 1x1 head is computed on the tile's 10 x 34 halo patch straight into LDS
 * (out-of-image patch pixels = the 3x3's SAME zero padding), the 3x3's nine taps are nine shifted reads of that patch, and
 * its accumulators feed the 1x1 tail + skip from the registers -- neither c/2-channel intermediate touches HBM.  Every
 * output is the same k-ordered fp32 fma chain as in the three sntc_conv_forward launches: bit-identical to them, for any
 * batch size and any number of workgroups.
 *   w0 [1,1,c,c/2], w1 [3,3,c/2,c/2], w2 [1,1,c/2,c]: the Keras HWIO kernels (device pointers); b0, b1, b2: biases or NULL.
 * sntc_resblock_supported(c): 1 where the kernel exists (c = 192), else 0 -- callers then run the three layers.
 * x and y must not alias (tiles read their neighbours' halo); n*h*w*c*4 < 2 GiB per call.
 * precision 0: exact fp32 (the above).  precision 1: the same kernel structure in bf16 x 3 split precision (three bfloat16
 * terms per fp32 operand, six cross products on v_mfma_f32_32x32x16_bf16, fp32 accumulate; weights pre-split at plan creation,
 * pixels split in registers; x and y stay fp32 NHWC): fp32-level accuracy against float64, NOT bit-identical to precision 0 --
 * what Model(precision="bf16x3") runs, batch-invariant like everything else. */
typedef struct sntc_resblock_plan sntc_resblock_plan;
int sntc_resblock_supported(int c);
int sntc_resblock_plan_create(int c, const float* w0, const float* b0, const float* w1, const float* b1,
                              const float* w2, const float* b2, int precision, void* stream, sntc_resblock_plan** plan);
int sntc_resblock_plan_update(sntc_resblock_plan* plan, const float* w0, const float* b0, const float* w1, const float* b1,
                              const float* w2, const float* b2, void* stream);
void sntc_resblock_plan_destroy(sntc_resblock_plan* plan);
/* Algorithmic 2*MAC FLOPs of one call: 2 n h w (c c/2 + 9 (c/2)^2 + c/2 c), the three layers' sntc_conv_flops. */
int64_t sntc_resblock_flops(const sntc_resblock_plan* plan, int n, int h, int w);
int sntc_resblock_forward(const sntc_resblock_plan* plan, const float* x, int n, int h, int w, float* y, void* stream);
/* Cap the persistent workgroups of THIS plan's launches (0 = one per CU): tests assert that results do not depend on it. */
int sntc_resblock_plan_set_workgroups(sntc_resblock_plan* plan, int max_workgroups);
/* ------------------------------------------------------------------------------------------
 * The RGB first layer of the analysis transforms (reference common/elic.py:147 through build_conv :253-270 -- Keras
 * Conv2D(channels[0], 5, strides=2, padding="SAME") on the 3-channel image; common/transforms.py:183 CNNAnalysis' first layer) as
 * ONE launch of a kernel of its own:  y[n, ceil(h/2), ceil(w/2), cout] = act(conv(x) + bias),  x [n, h, w, 3] NHWC fp32.
 * A persistent workgroup owns 8 x 32 output pixels at a time (wave = tile row, lane = pixel, registers = all cout channels); the
 * packed weights stay in LDS, the tile's 19-row input patch is double-buffered there (SAME padding = zeros in the patch: no
 * padded copy of the image), and a pixel fragment is four consecutive floats of a patch row.  Every output is the same
 * k-ordered fp32 fma chain as the row-packed gather-GEMM plan (sntc_conv_desc.reserved[2] == 1 on a zero-padded image):
 * bit-identical to it, for any batch size and any number of workgroups.
 *   w [k, k, 3, cout]: the Keras HWIO kernel (device pointer); bias [cout] or NULL; act: SNTC_ACT_NONE / RELU / LEAKY_RELU.
 * kind SNTC_SIGNAL_DOWN: the same kernel for tfc.SignalConv2D(corr=True, strides_down=2, padding="same_zeros") on the image -- the
 * first layer of MBT2018Analysis (reference common/transforms.py:152-155; BASELINE configs[1]) -- whose only difference is the
 * padding origin (the kernel is centred: k / 2 in front on both axes); it replaces the generic dword-gather plan
 * there (same products, the row-packed order of summation; tolerance-tested against the oracle, batch-invariant).
 * sntc_rgbconv_supported: 1 where the kernel exists (kind SNTC_CONV2D or SNTC_SIGNAL_DOWN, cin = 3, stride 2, k <= 5, cout in
 * {128, 192, 256}), else 0 -- callers then run the row-packed / generic plan.  x and y < 2 GiB per call. */
typedef struct sntc_rgbconv_plan sntc_rgbconv_plan;
int sntc_rgbconv_supported(int kind, int k, int stride, int cin, int cout, int act);
int sntc_rgbconv_plan_create(int kind, int k, int stride, int cin, int cout, const float* w, const float* bias, int act,
                             void* stream, sntc_rgbconv_plan** plan);
int sntc_rgbconv_plan_update(sntc_rgbconv_plan* plan, const float* w, const float* bias, void* stream);
void sntc_rgbconv_plan_destroy(sntc_rgbconv_plan* plan);
/* Algorithmic 2*MAC FLOPs of one call: 2 n ceil(h/2) ceil(w/2) k k 3 cout (sntc_conv_flops of the layer). */
int64_t sntc_rgbconv_flops(const sntc_rgbconv_plan* plan, int n, int h, int w);
int sntc_rgbconv_forward(const sntc_rgbconv_plan* plan, const float* x, int n, int h, int w, float* y, void* stream);
/* Cap the persistent workgroups of THIS plan's launches (0 = one per CU): tests assert that results do not depend on it. */
int sntc_rgbconv_plan_set_workgroups(sntc_rgbconv_plan* plan, int max_workgroups);
/* ------------------------------------------------------------------------------------------
 * The LAST layer of the multi-layer syntheses -- a 5 x 5 / 2 or 9 x 9 / 4 transposed convolution down to the 3 image channels (reference
 * common/transforms.py:172-175, MBT2018Synthesis: tfc.SignalConv2D(3, (5, 5), corr=False, strides_up=2, padding="same_zeros");
 * :195-206, CNNSynthesis: conv_t_k5s2 = tf.keras.layers.Conv2DTranspose(3, 5, strides=2, padding="SAME"), :85-87) -- as ONE launch of a
 * vector-ALU kernel:
 *   y[n, s h, s w, 3] = conv_transpose(x[n, h, w, cin], w) + bias,   NHWC fp32, no activation (the reference has none there);
 * also :131-134, BLS2017Synthesis: tfc.SignalConv2D(3, (9, 9), corr=False, strides_up=4) (k = 9, stride 4).
 * With three output channels the layer's four output phases are four groups of 3 columns in 32-wide MFMA tiles on the gather
 * GEMM (9.9 TFLOP/s at 8 x 128 x 128 x 192: a quarter of the mbt2018 config's decode); here a block owns 16 x 16 macro pixels, the
 * input channels pass through LDS in 16-channel slabs, a thread runs the 25 taps x 3 outputs of its 2 x 2 output quad with the
 * weights as scalar operands.  fp32 fma chains in another order than sntc_conv_forward's (tolerance-tested against the oracle);
 * an image's results do not depend on the batch.
 *   kind SNTC_CONV2D_TRANSPOSE: w [k, k, 3, cin] (Keras);  SNTC_SIGNAL_UP: w [k, k, cin, 3] (tfc);  bias [3] or NULL.
 * sntc_upsmall_supported: 1 where the kernel exists (those kinds, (k, stride) = (5, 2) or (9, 4), cin % 16 == 0, cout = 3), else 0 -- callers then
 * run sntc_conv_forward. */
typedef struct sntc_upsmall_plan sntc_upsmall_plan;
int sntc_upsmall_supported(int kind, int k, int stride, int cin, int cout);
int sntc_upsmall_plan_create(int kind, int k, int stride, int cin, int cout, const float* w, const float* bias, void* stream,
                             sntc_upsmall_plan** plan);
int sntc_upsmall_plan_update(sntc_upsmall_plan* plan, const float* w, const float* bias, void* stream);
void sntc_upsmall_plan_destroy(sntc_upsmall_plan* plan);
/* Algorithmic 2*MAC FLOPs of one call: 2 n h w k k cin cout (sntc_conv_flops of the layer). */
int64_t sntc_upsmall_flops(const sntc_upsmall_plan* plan, int n, int h, int w);
int sntc_upsmall_forward(const sntc_upsmall_plan* plan, const float* x, int n, int h, int w, float* y, void* stream);
/* ------------------------------------------------------------------------------------------
 * The first layer of the two-layer syntheses (reference common/transforms.py:298-317 TwoLayerSynthesis, :320-361
 * TwoLayerResSynthesis) in ONE launch:  hidden = act(base_conv(y_hat)) [+ res(y_hat)]  -- the tensor the reference hands to
 * its output convolution (:315, :359).  base_conv / res = Conv2DTranspose k x k / stride, SAME (:307-313, :331-338, :351-357);
 * act_kind as in sntc_two_layer_tail (0 none, 1 IGDN1, 2 GDN1, 3 relu, 4 leaky relu) is base_conv's `activation`.
 *   w1 [k, k, (1 + has_res) ch, cin]: the Keras transposed kernel(s), the residual branch's concatenated behind the base
 *   convolution's along the output-channel axis; b1 [(1 + has_res) ch] or NULL; beta [ch], gamma [ch, ch] for act_kind 1 / 2.
 * Indexed by the output-aligned macro pixel a tap is one of nine whole-pixel shifts of the input: a workgroup walks
 * (tile of 256 latent pixels) x (unit of 96 output columns) items -- weights through an LDS-DMA ring, the 16-channel
 * patch slab staged once per slab, every shift a shifted fragment read of it -- dealt dynamically over one workgroup per CU;
 * the activation and the residual add are applied to the accumulators.  Every output is the same k-ordered fp32 fma chain
 * (and the same activation arithmetic) as sntc_conv_forward + stage 1 of sntc_two_layer_tail: bit-identical to them, for any
 * batching.  sntc_syn_supported: 1 where the kernel exists (13 x 13 / 8, 5 x 5 / 2, 3 x 3 / 1 ...: taps within one pixel of
 * the output-aligned source; (1 + has_res) ch in {12, 24, 48}; cin % 16 == 0), else 0 -- callers then run the layers.
 * One call takes up to four batches of DIFFERENT image sizes (sntc_syn_batch: y_hat [n, h, w, cin] -> hidden
 * [n, stride h, stride w, ch]; w <= 127; one image's tensors < 2 GiB): they share the launch's work queue.
 * workspace >= sntc_syn_workspace_bytes(), private to the call until it has run. */
typedef struct sntc_syn_plan sntc_syn_plan;
typedef struct sntc_syn_batch {
  const float* y_hat;
  float* hidden;
  int32_t n, h, w;
  int32_t reserved;
} sntc_syn_batch;
int sntc_syn_supported(int k, int stride, int cin, int ch, int has_res);
int sntc_syn_plan_create(int k, int stride, int cin, int ch, int has_res, int act_kind, const float* w1, const float* b1,
                         const float* beta, const float* gamma, void* stream, sntc_syn_plan** plan);
int sntc_syn_plan_update(sntc_syn_plan* plan, const float* w1, const float* b1, const float* beta, const float* gamma, void* stream);
void sntc_syn_plan_destroy(sntc_syn_plan* plan);
/* Algorithmic 2*MAC FLOPs of `latent_pixels` input pixels (all batches of a call): 2 k^2 cin (1 + has_res) ch each. */
int64_t sntc_syn_flops(const sntc_syn_plan* plan, int64_t latent_pixels);
int64_t sntc_syn_workspace_bytes(const sntc_syn_plan* plan);
int sntc_syn_forward(const sntc_syn_plan* plan, const sntc_syn_batch* batches, int nbatches, void* workspace, size_t workspace_bytes,
                     void* stream);
/* Cap the persistent workgroups of THIS plan's launches (0 = one per CU): tests assert that results do not depend on it. */
int sntc_syn_plan_set_workgroups(sntc_syn_plan* plan, int max_workgroups);
/* The plan's units (tests, tools): up to `capacity` records of four ints (shifts per channel slab; phases held | 32-row tile
 * steps per slab << 8 | mask of the steps that leave the third tile out << 16; the shift list packed four bits per step:
 * low word, high word); returns the number of units. */
int sntc_syn_plan_units(const sntc_syn_plan* plan, int* out, int capacity);
/* The units a layer would get, without a device or a plan (same records as sntc_syn_plan_units; -1 where unsupported). */
int sntc_syn_describe(int k, int stride, int cin, int ch, int has_res, int* out, int capacity);
/* Host-only check of the decomposition (no device): units + packed weights driven through the kernel's loop nest on the CPU
 * against the scatter form of Conv2DTranspose(SAME) on a random h x w input; *max_err = largest absolute difference. */
int sntc_syn_selfcheck(int k, int stride, int cin, int ch, int has_res, int h, int w, unsigned seed, double* max_err);
/* Force the gather-GEMM tile variant of THIS plan (0 = back to the heuristic): profiling and the
 * every-variant parity test only; tile choice never changes results beyond fp32 summation order. */
int sntc_conv_plan_set_tile(sntc_conv_plan* plan, int variant);
/* Schedule switches of THIS plan (tests / profiling).  flags bit 0: 1 (default) lets large launches run on the persistent
 * stream-K workers, 0 forces the static schedule (one workgroup per tile); bit 2 set: bit 1 selects the stage path --
 * 1 direct-to-LDS (buffer_load ... lds), 0 register staging; bit 2 clear: the library default; bit 3: stream-K also for launches of
 * short tiles (< 64 K stages on average), which by default run one workgroup per tile because that is faster; bit 5 set: bit 6
 * selects the stream-K unit order -- 1 column tile outermost (where that twin of the kernel exists), 0 row strip outermost; bit 5
 * clear: column-major where one group's packed weights exceed an XCD's L2.  Every combination
 * produces bit-identical outputs (each element is the same k-ordered fma chain); the switches exist so that a test can
 * assert exactly that. */
int sntc_conv_plan_set_schedule(sntc_conv_plan* plan, int flags);
/* ------------------------------------------------------------------------------------------
 * Split-precision operands (bf16 x 3).  Format S3 of an NHWC tensor with C % 16 == 0: per pixel and 16-channel slab
 * 96 bytes [hi x 16 | mid x 16 | lo x 16] bfloat16 with x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi),
 * lo = bf16(x - hi - mid)): 6 bytes per element, what a conv plan with reserved[1] == 2 reads.  The reference has no
 * counterpart (fp32 tf.nn.conv2d, common/transforms.py:81-90); these feed the same contractions.
 * ------------------------------------------------------------------------------------------ */
/* out (S3, npix * c * 6 bytes) = split(x [npix, c] fp32) */
int sntc_split3(const float* x, int64_t npix, int c, void* out, void* stream);
/* decoder side of sntc_entropy_scale_normal fused with the split: y_hat = symbols + mu (mu = first half of hyper
 * [npix, 2c], mshyper/models.py:278-279) -> out (S3); y_hat (fp32 [npix, c]) is also written when not NULL */
int sntc_dequant_split3(const int32_t* symbols, const float* hyper, int64_t npix, int c, void* out, float* y_hat,
                        void* stream);
/* Stream-K health.  The persistent stream-K schedule needs every worker of a launch resident at once, which HIP does not
 * promise on a device that other streams or processes share.  A worker that waits in vain (bounded spin) for its neighbour's
 * hand-off no longer traps: it sets bit 0 of a sticky per-device status word and the launch completes with INVALID results.
 * sntc_conv_status exchanges that word with 0 in one atomic on `stream`, copies the old value to *flags and synchronises
 * `stream` -- call it where the host synchronises anyway (the Python driver does: at every device -> host copy of metrics, in
 * compress / decompress, and in encode / decode unless the caller defers the check); launches still in flight on OTHER streams
 * are not covered: join side streams first.  Non-zero means: discard the results since the last check, and re-run after
 * sntc_conv_set_stream_k(0), which makes every later call use the static one-workgroup-per-tile / split-K schedules
 * (bit-identical results).  sntc_conv_status_inject ORs `flags` into the word on `stream` (tests: the raise
 * paths of the callers).  The reference has no counterpart (single stream, cuDNN; common/transforms.py:81-90). */
int sntc_conv_status(int* flags, void* stream);
int sntc_conv_status_inject(int flags, void* stream);
int sntc_conv_set_stream_k(int enabled);
/* The switch's current value (1 = stream-K where the schedule picks it): callers that turn it off around launches which run beside
 * long-lived kernels (Python: ops.static_schedules) put back what they found. */
int sntc_conv_get_stream_k(void);
/* Gather-GEMM tile variant (1..7: 128 x 32v, 8: 64 x 64, 9: 128 x 128 as 64 x 64 per wave, 10: 256 x 128) picked for this call shape,
 * and the number of workgroups it launches; for profiling / roofline bookkeeping. */
int sntc_conv_launch_info(const sntc_conv_plan* plan, int n, int h, int w, int* variant, int* nblocks);
/* 1 if that launch walks its stream-K units column tile outermost (the weights of a few column tiles stay in an XCD's L2 while
 * the row strips stream past: single-group layers whose packed weights exceed 4 MB), 0 for the row-strip-major order.  Profiling
 * only: the kernel's name differs, the bits do not. */
int sntc_conv_launch_order(const sntc_conv_plan* plan, int n, int h, int w, int* column_major);
/* Measured schedule.  Every (tile variant, stream-K / one-workgroup-per-tile) candidate of a plan computes the same k-ordered
 * chains, so the choice is a question of speed only; sntc_conv_plan_tune times them on the caller's buffers for one (n, h, w)
 * (`reps` launches each, HIP events on `stream`, synchronises it) and records the winner in the plan: later calls of that shape
 * run it instead of the cost model's pick (a forced tile / schedule still wins).  `workspace` >= sntc_conv_tune_workspace_bytes().
 * y holds the layer's output afterwards.  A recorded choice can change what sntc_conv_workspace_bytes() reports for that shape: tune
 * (or set a choice) before other threads size their workspaces for it, not while they are between the query and the launch.
 * The reference has no counterpart of its own: TensorFlow's convolutions pick their
 * algorithm by cuDNN autotune the same way (tf.nn.conv2d behind common/transforms.py:81-90). */
int64_t sntc_conv_tune_workspace_bytes(const sntc_conv_plan* plan, int n, int h, int w);
int sntc_conv_plan_tune(sntc_conv_plan* plan, const float* x, int n, int h, int w, float* y, const float* res,
                        const float* aux, void* workspace, size_t workspace_bytes, int reps, int* variant, int* stream_k,
                        void* stream);
/* The candidates sntc_conv_plan_tune would time for this call shape (returns how many were written, < 0 on a bad shape), and a
 * way to record a choice made elsewhere -- e.g. by timing a whole step with several streams in flight, where a schedule
 * measured with the device to itself is not the best one (bench.py does this for its two-stream decode).  Same remark as above:
 * cuDNN's algorithm selection behind tf.nn.conv2d (common/transforms.py:81-90) is the reference-side counterpart. */
int sntc_conv_plan_candidates(const sntc_conv_plan* plan, int n, int h, int w, int* variants, int* stream_k, int capacity);
int sntc_conv_plan_set_choice(sntc_conv_plan* plan, int n, int h, int w, int variant, int stream_k);
int sntc_conv_plan_clear_tuning(sntc_conv_plan* plan);

/* ------------------------------------------------------------------------------------------
 * Small-channel GDN1 / IGDN1 (C <= 64): wave-shuffle contraction, no MFMA.
 *   replaces GDN1.call, common/transforms.py:26-63.  gamma is [C(in), C(out)], effective values.
 *   alpha in {1,2}; epsilon_is_half selects ^0.5.
 * ------------------------------------------------------------------------------------------ */
int sntc_gdn_small(const float* x, int64_t npix, int c, const float* beta, const float* gamma,
                   int inverse, int alpha, int epsilon_is_half, float* y, void* stream);

/* ------------------------------------------------------------------------------------------
 * Two-layer synthesis tail: h = act(t[..., :Ch]) (+ t[..., Ch:2Ch] if has_res);
 *   x_hat = Conv2DTranspose_{k2 x k2 / s2, SAME}(h) + bias2        (float, NHWC [n, s2*hh, s2*wh, 3])
 *   replaces the IGDN1 + add + out_conv of TwoLayer[Res]Synthesis.call,
 *   common/transforms.py:315-317,359-361 (out_conv kernel [k2,k2,3,Ch]).
 *   act_kind: 0 none, 1 IGDN1, 2 GDN1, 3 relu, 4 leaky_relu.   Ch <= 48, k2 <= 5, s2 == 2.
 * ------------------------------------------------------------------------------------------ */
int sntc_two_layer_tail(const float* t, int n, int hh, int wh, int ch, int has_res, int act_kind,
                        const float* beta, const float* gamma, const float* w2, const float* b2,
                        int k2, int s2, int cout, float* x_hat, void* stream);
/* The same tail with the decoder's last steps fused into the launch: crop to h x w (unpad_images, common/image_utils.py:69-71),
 * floats_to_pixels + quantize_image (common/data_lib.py:48-52, image_utils.py:22-23: (v + .5) * 255, round half to even,
 * saturate) -> pixels uint8 [n, h, w, 3]; with ref (float [n, h, w, 3], normalised) also sse[n] = integer squared error
 * of the two quantised images (mse_psnr, image_utils.py:26-38).  No float reconstruction is written. */
int sntc_two_layer_tail_pixels(const float* t, int n, int hh, int wh, int ch, int has_res, int act_kind,
                               const float* beta, const float* gamma, const float* w2, const float* b2, int k2, int s2,
                               int cout, int h, int w, const float* ref, uint8_t* pixels, uint64_t* sse, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pixel domain
 *   pad_images / unpad_images       common/image_utils.py:41-71
 *   floats_to_pixels + mse_psnr     common/data_lib.py:48-52, common/image_utils.py:22-38
 * ------------------------------------------------------------------------------------------ */
/* Reflect-pad bottom/right: y[n,hp,wp,c] from x[n,h,w,c]  (hp>=h, wp>=w, pads < size). */
int sntc_pad_reflect(const float* x, int n, int h, int w, int c, int hp, int wp, float* y, void* stream);
/* y[n, hp, wp, c] <- x[n, h, w, c] placed at (top, left), zeros elsewhere: the explicit form of Keras SAME padding for a
 * row-packed plan (sntc_conv_desc.reserved[2]).  hp >= top + h, wp >= left + w. */
int sntc_pad_zero(const float* x, int n, int h, int w, int c, int top, int left, int hp, int wp, float* y, void* stream);
/* Crop top-left: y[n,h,w,c] from x[n,hp,wp,c]. */
int sntc_crop(const float* x, int n, int hp, int wp, int c, int h, int w, float* y, void* stream);
/* tf.nn.depth_to_space(x, block), NHWC: y[n, h*block, w*block, c/block^2] with input channel (dy*block + dx)*(c/block^2) + k
 * going to output pixel (iy*block + dy, ix*block + dx), channel k -- the upsampling steps of
 * TwoLayerResSynthesis(res_type="d2s"), common/transforms.py:341-348. */
int sntc_depth_to_space(const float* x, int n, int h, int w, int c, int block, float* y, void* stream);
/* y[npix, ca + cb] = [a | b] along the channel axis (tf.concat(..., axis=-1)); b == NULL appends a channel of ones and cb - 1 zero
 * channels instead: the constant input of JPEGLikeSynthesis(use_offset=True), common/transforms.py:291-293, padded to a 16-channel slab. */
int sntc_concat_channels(const float* a, int ca, const float* b, int cb, int64_t npix, float* y, void* stream);
/* Quantise both images to uint8 the reference's way ((v+.5)*255, round-half-even, saturate) and
 * accumulate the per-image integer sum of squared differences.  x_hat may be strided (crop fused):
 * element (b,i,j,k) at x_hat[((b*hs + i)*ws + j)*c + k].  pixels_out (uint8 [n,h,w,c]) may be NULL.
 * sse_out: uint64 [n], OVERWRITTEN (zeroed on stream first).  x == NULL: only quantise x_hat into
 * pixels_out (the decoder's last step); sse_out is then ignored. */
int sntc_pixels_sse(const float* x, const float* x_hat, int n, int h, int w, int c, int hs, int ws,
                    uint8_t* pixels_out, unsigned long long* sse_out, void* stream);
/* Training-mode distortion (SGA): sum over (255*(x - x_hat))^2 in float, per image, double out. */
int sntc_float_sse(const float* x, const float* x_hat, int n, int h, int w, int c, int hs, int ws,
                   double* sse_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Entropy models (compression=False paths: rate *estimates*, as the reference evaluates them)
 * ------------------------------------------------------------------------------------------ */
/* tfc.DeepFactorized parameters for C channels, `nlayers` = len(num_filters)+1 affine layers of
 * widths 1 -> f1 -> ... -> 1 (widths[nlayers+1], each <= 4).  Raw TFC variables (softplus / tanh
 * are applied inside): matrices[k] [C,f_{k+1},f_k], biases[k] [C,f_{k+1}], factors[k] [C,f_{k+1}],
 * concatenated per kind in layer order as host float arrays. */
typedef struct sntc_prior sntc_prior;
int sntc_prior_create(int channels, int nlayers, const int* widths, const float* matrices,
                      const float* biases, const float* factors, void* stream, sntc_prior** prior);
void sntc_prior_destroy(sntc_prior* prior);

/* tfc.ContinuousBatchedEntropyModel(NoisyDeepFactorized, coding_rank=3, compression=False)
 *   __call__(z, training=False): mshyper/models.py:249-255, factorized/models.py:101-105.
 * z[n, hw, C] -> z_hat = round(z) (float) and bits[n] (double, -sum log2 p).
 * values_only != 0: skip rounding and evaluate bits at z as given (log_prob of an explicit sample,
 *   mshyper/models.py:262-268). */
int sntc_entropy_factorized(const sntc_prior* prior, const float* z, int n, int64_t hw, float* z_hat,
                            double* bits, int values_only, void* stream);

/* tfc.LocationScaleIndexedEntropyModel(NoisyNormal, 64, SCALE_FN, coding_rank=3,
 *   compression=False)(y, indexes=exp(raw), loc=mu, training=False): mshyper/models.py:273-279.
 * hyper[n, hw, 2C] holds mu = hyper[..., :C] and raw = hyper[..., C:]  (tf.split + tf.exp fused).
 * Outputs: y_hat = round(y - mu) + mu (float), symbols = round(y - mu) as int32 (may be NULL),
 * bits[n] (double).  values_only != 0: y is an explicit sample; bits evaluated at y - mu without
 * rounding, y_hat/symbols untouched (mshyper/models.py:285-291). */
int sntc_entropy_scale_normal(const float* y, const float* hyper, int n, int64_t hw, int c,
                              float* y_hat, int32_t* symbols, double* bits, int values_only,
                              void* stream);
/* Decoder-side dequantisation: y_hat = symbols + mu  (mu = hyper[..., :C]). */
int sntc_dequant_scale_normal(const int32_t* symbols, const float* hyper, int n, int64_t hw, int c,
                              float* y_hat, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Bitstream: table-driven 64-way interleaved rANS, 16-bit quantised CDFs.  The reference has no
 *   bitstream (compression=False at mshyper/models.py:246-251; its bpp is the estimate); these entry points make
 *   `decode` a real codec.  One stream per (image, segment); an image's elements_per_image values (flat, in memory
 *   order) are cut into `segments` runs of sntc_rans_cap_words()/2 - 64 elements (a multiple of 64).
 *   Tables: cdf = uint16, concatenated, table t holds cdf[0..n) (cdf[n] = 65536 implicit; pad the array to an even
 *   count); meta = uint32 [ntables][2] = { offset into cdf, (n << 16) | (vmin & 0xffff) }; the last symbol of every
 *   table is ESCAPE, followed in the stream by value + 32768 as a raw 16-bit word.
 *   Stream = [lanes x (state hi, state lo)] [16-bit words in decode order]  (format: csrc/rans.hip header);
 *   lanes in {8, 16, 32, 64}: element lanes*j + l belongs to lane l at step j; fewer lanes = fewer flushed bytes.
 * ------------------------------------------------------------------------------------------------------- */
int64_t sntc_rans_cap_words(int64_t elems_per_image, int segments);
/* scratch: uint16 [nimages * segments][cap_words]; stream s ends up in the LAST len_words[s] words of its row. */
int sntc_rans_encode(const int32_t* values, const uint16_t* table_ids, int nimages, int64_t elems_per_image, int segments,
                     int lanes, const uint16_t* cdf, const uint32_t* meta, int ntables, int total_entries, int64_t cap_words,
                     uint16_t* scratch, int32_t* len_words, void* stream);
/* gathers the streams into one payload; offsets int64 [nstreams + 1] = exclusive prefix sum of len_words */
int sntc_rans_compact(const uint16_t* scratch, int64_t cap_words, const int32_t* len_words, const int64_t* offsets,
                      int nstreams, uint16_t* payload, void* stream);
/* bad_streams (int32[1], device) counts streams that did not end at their initial state / length: corruption.
 * dec / lut / lut_meta (all three or none; NULL = the plain decoder, a binary search of cdf per symbol) are the decoder's own
 *   view of the same tables, which it keeps in LDS:
 *   dec      uint32, table t's entry s at dec[offset_t + 3 t + s] = (cdf[s] << 16) | (freq[s] - 1), three 0xffffffff after every
 *            table, the array padded to a multiple of 4 entries (total_entries + 3 ntables, rounded up);
 *   lut      per table a START TABLE of 2^bits uint16 entries, lut[lut_off + b] = the largest symbol s with
 *            cdf[s] <= b << (16 - bits), 0 <= bits <= 16; lut_meta[t] = (lut_off << 5) | bits; lut_entries a multiple of 8 and
 *            <= sntc_rans_lut_budget(ntables, total_entries), the room the other tables leave in a CU's LDS.
 *   The decoded values do not depend on them; with about one start entry per symbol a step costs three LDS round trips
 *   instead of log2(n) + 2. */
int64_t sntc_rans_lut_budget(int ntables, int total_entries);
int sntc_rans_decode(const uint16_t* payload, const int64_t* offsets, const uint16_t* table_ids, int nimages,
                     int64_t elems_per_image, int segments, int lanes, const uint16_t* cdf, const uint32_t* meta, int ntables,
                     int total_entries, const uint32_t* dec, const uint16_t* lut, const uint32_t* lut_meta, int lut_entries,
                     int32_t* values, int32_t* bad_streams, void* stream);
/* table id of every y element = round(clamp(exp(raw), 0, 63)), raw = hyper[..., c:]; of every z element = channel */
int sntc_scale_table_ids(const float* hyper, int64_t npix, int c, uint16_t* table_ids, void* stream);
/* table id = channel (deep-factorized prior: one table per channel). */
int sntc_channel_table_ids(int64_t npix, int c, uint16_t* table_ids, void* stream);
int sntc_round_to_int(const float* x, int64_t total, int32_t* out, void* stream);
int sntc_int_to_float(const int32_t* x, int64_t total, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * SSIM / MS-SSIM statistics (eval-only quality metrics, reference mshyper/models.py:321-336 ->
 *   tf.image.ssim / tf.image.ssim_multiscale).  Images are float NHWC holding pixel
 *   values (0..max_val), c in {1, 3}.
 * ------------------------------------------------------------------------------------------ */
/* One scale: 11 x 11 Gaussian (sigma 1.5) VALID moments; ssim_sum[n*c] = sum over filter outputs of
 * luminance*cs, cs_sum[n*c] = sum of cs (both OVERWRITTEN).  Divide by (h-10)(w-10) for the means. */
int sntc_ssim_scale(const float* a, const float* b, int n, int h, int w, int c, float max_val, double* ssim_sum,
                    double* cs_sum, void* stream);
/* 2 x 2 average pooling to ceil(h/2) x ceil(w/2); odd sizes repeat the last row / column (tf.pad SYMMETRIC). */
int sntc_avgpool2_symmetric(const float* x, int n, int h, int w, int c, float* y, void* stream);
/* crop + floats_to_pixels (round-half-even, clamp 0..255) kept as float: the SSIM input. */
int sntc_pixels_float(const float* x_hat, int n, int h, int w, int c, int hs, int ws, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * SGA iterative inference (config 5): element-wise pieces of Model.itinf_train_step,
 *   mshyper/models.py:397-408 with frame_loss_given_latent_rvs(training=True), :260-268,285-291,343.
 *   The contractions of the backward pass are sntc_conv_forward calls: the input gradient of
 *   Conv2DTranspose(k, s, SAME) with kernel W[kh,kw,Cout,Cin] is SNTC_CONV2D with the same array
 *   read as [kh,kw,Cin'=Cout,Cout'=Cin]; relu / leaky-relu masks ride on SNTC_EPI_MASK_*.
 * sga_round (common/latent_rvs_utils.py:8-48): logits = (-atanh(clip(u - floor u))/tau,
 *   -atanh(clip(ceil u - u))/tau), w = softmax((logits + Gumbel)/tau), sample = w0 floor + w1 ceil.
 *   noise: NULL -> counter-based generator keyed by (seed, step, element); else float [.., 2]
 *   Gumbel values (deterministic tests).
 * ------------------------------------------------------------------------------------------ */
/* z~ = sga_round(z_loc, tau); bits[n] = -sum log2 p_DF(z~); sprime = d z~/d z_loc;
 * dbits_dz = d(-log2 p)/d z~ per element. */
int sntc_sga_factorized_fwd(const sntc_prior* prior, const float* z_loc, int n, int64_t hw, float tau,
                            const float* noise, uint64_t seed, uint64_t step, float* z_tilde, float* sprime,
                            float* dbits_dz, double* bits, void* stream);
/* y~ = sga_round(y_loc - mu, tau) + mu with (mu, raw) = hyper halves; bits[n] = -sum log2 p_N(y~ - mu; sigma);
 * sprime = d sga/d u; dbits_dv = d bits/d (y~ - mu); dbits_draw = d bits/d raw (through exp, clamp, SCALE_FN). */
int sntc_sga_normal_fwd(const float* y_loc, const float* hyper, int n, int64_t hw, int c, float tau,
                        const float* noise, uint64_t seed, uint64_t step, float* y_tilde, float* sprime,
                        float* dbits_dv, float* dbits_draw, double* bits, void* stream);
/* g_yloc = (g_ytilde + weight dbits_dv) sprime;  g_hyper = [g_ytilde (1 - sprime) - weight dbits_dv sprime | weight dbits_draw] */
int sntc_sga_normal_bwd(const float* g_ytilde, const float* sprime, const float* dbits_dv, const float* dbits_draw,
                        float weight, int64_t npix, int c, float* g_yloc, float* g_hyper, void* stream);
/* UQLatentRV.sample(training, method, offset, **kwargs) / .quantize(offset) (common/latent_rvs_lib.py:77-116) on [npix, c]
 * values: u = loc - offset (offset NULL = none; else element (p, ch) is offset[p * offset_stride + ch], e.g. the mean half of
 * the hyper-synthesis output with offset_stride = 2 c; offset_stride 0 broadcasts a per-channel [c] offset).  mode 0: round-half-even(u) + offset (training=False, :95-102; also
 * tfc.round_st's forward value, :77-78); 1 'unoise': loc + U(-.5, .5) (:104-107); 2 'sga': sga_round(u, tau = param) + offset
 * (:108-110, common/latent_rvs_utils.py:8-48); 3 'soft_round': tfc.soft_round(u, alpha = param) + offset (:111-114).
 * noise: NULL -> counter-based generator keyed by (seed, step, element); else the uniform values [npix, c] in (-.5, .5) (mode 1)
 * or the Gumbel pairs [npix, c, 2] (mode 2). */
int sntc_uq_sample(const float* loc, const float* offset, int64_t npix, int c, int offset_stride, int mode, float param,
                   const float* noise, uint64_t seed, uint64_t step, float* out, void* stream);
/* out = (g + weight dbits) sprime */
int sntc_sga_chain(const float* g, const float* dbits, const float* sprime, float weight, int64_t total, float* out,
                   void* stream);
/* Training-mode distortion (common/data_lib.py:48-52, no rounding): sse[n] = sum (255 (x - x_hat))^2 over the
 * un-padded h x w region; g_xhat[n,hs,ws,c] = scale (x_hat - x), zero in the padded margin. */
int sntc_distortion_grad(const float* x, const float* x_hat, int n, int h, int w, int c, int hs, int ws, float scale,
                         float* g_xhat, double* sse, void* stream);
/* Backward of the activation + residual split of sntc_two_layer_tail: t is the forward input, g_h the gradient
 * w.r.t. h; g_t[npix, cp] = [d act(base) | g_h (if has_res) | zeros up to cp]. */
/* Input gradient of the tail's output layer (Conv2DTranspose 5x5 / 2, ch -> 3, SAME; w2 [5,5,3,ch]):
 * g_h[n, hh, wh, ch] from g_xhat[n, 2 hh, 2 wh, 3] -- an HBM stream, not worth a GEMM launch. */
int sntc_two_layer_out_adjoint(const float* g_xhat, int n, int hh, int wh, int ch, const float* w2, int k2, int s2, int cout,
                               float* g_h, void* stream);
/* abs_x / g_x (both [npix, ch], may both be NULL): |base| and g_h * base, the operands of the IGDN1 parameter
 * gradients of the training step (d gamma = |x|^T (g x) summed over pixels, d beta = column sums of g x). */
int sntc_two_layer_tail_bwd(const float* t, const float* g_h, int64_t npix, int ch, int has_res, int act_kind,
                            const float* beta, const float* gamma, int cp, float* g_t, float* abs_x, float* g_x,
                            void* stream);
/* Keras Adam (tf.keras.optimizers.Adam, no amsgrad) on one flat tensor; t = 1-based step; the gradient is
 * multiplied by grad_scale first (global-norm clipping, optimizer_config.global_clipnorm). */
int sntc_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                   float eps, int64_t t, float grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training step: Model.train_step, mshyper/models.py:375-383 = tape.gradient of
 *   frame_loss_given_latent_rvs(training=True) w.r.t. every trainable variable + Adam.
 *   Input gradients are sntc_conv_forward on adjoint plans (see the SGA note above); the pieces below are the
 *   weight / bias gradients, the element-wise backward steps and the training-mode entropy terms.
 * ------------------------------------------------------------------------------------------ */
/* Weight gradient of a Keras-SAME Conv2D (kind SNTC_CONV2D, kernel [kh,kw,cin,cout]) or Conv2DTranspose
 * (SNTC_CONV2D_TRANSPOSE, kernel [kh,kw,cout,cin]) layer, in the layer's own kernel layout:
 *   x      [n,h,w,cin]      the layer input of the forward pass
 *   g_out  [n,ho,wo,cout]   gradient w.r.t. the PRE-activation output (ho = ceil(h/s) or h*s)
 *   dw     written, or accumulated into when accumulate != 0.
 * fp32 MFMA contraction over the pixels, split over blocks, summed in a fixed order (deterministic). */
int64_t sntc_conv_wgrad_workspace_bytes(int kind, int kh, int kw, int stride, int cin, int cout, int n, int h, int w);
int sntc_conv_wgrad(int kind, int kh, int kw, int stride, int cin, int cout, const float* x, const float* g_out, int n,
                    int h, int w, float* dw, int accumulate, void* workspace, int64_t workspace_bytes, void* stream);
/* db[c] (= or +=) sum over pixels of g[npix, c] */
int64_t sntc_bias_grad_workspace_bytes(int64_t npix, int c);
int sntc_bias_grad(const float* g, int64_t npix, int c, float* db, int accumulate, void* workspace, int64_t workspace_bytes,
                   void* stream);
/* g_pre = g * act'(pre) written through the layer output y (relu: y > 0; leaky_relu: 1 or 0.2; sigmoid: y (1 - y)).
 * g_pre may alias g. */
int sntc_act_backward(const float* g, const float* y, int64_t total, int act, float* g_pre, void* stream);
/* SimpleAttention gate unfused (common/elic.py:97-100): out = x + t * s;  g_t = g s, g_spre = g t s (1 - s). */
int sntc_gate_forward(const float* x, const float* t, const float* s, int64_t total, float* out, void* stream);
int sntc_gate_backward(const float* g, const float* t, const float* s, int64_t total, float* g_t, float* g_spre, void* stream);
/* a += alpha * b  (gradient accumulation at fan-outs) */
int sntc_axpy(float* a, const float* b, float alpha, int64_t total, void* stream);
/* out = x + U(-.5, .5): tfc's training=True perturbation (uq method "unoise").  noise NULL -> counter-based
 * generator keyed by (seed, step, element); else float [total] (deterministic tests). */
int sntc_noise_add(const float* x, int64_t total, const float* noise, uint64_t seed, uint64_t step, float* out, void* stream);
/* out[0] (double, device) = sum x^2  (global gradient norm) */
int sntc_sumsq(const float* x, int64_t total, double* out, void* stream);
/* Hidden layer of TwoLayer[Res]Synthesis unfused: h[npix, ch] = act(t[..., :ch]) (+ t[..., ch:2ch]);
 * act_kind as in sntc_two_layer_tail. */
int sntc_two_layer_hidden(const float* t, int64_t npix, int ch, int has_res, int act_kind, const float* beta,
                          const float* gamma, float* h, void* stream);
/* tfc.GDNParameter: eff = max(raw, bound)^2 - pedestal, and its gradient (lower_bound "identity_if_towards"). */
int sntc_gdn_reparam_forward(const float* raw, int64_t total, float bound, float pedestal, float* eff, void* stream);
int sntc_gdn_reparam_backward(const float* raw, const float* g_eff, int64_t total, float bound, float* g_raw, void* stream);
/* Rate terms at a given (noisy) sample: bits[n] = -sum log2 p, d bits / d (y~ - mu), d bits / d raw (normal);
 * d bits / d z~ (deep factorized).  Same formulas as the eval entry points with values_only = 1. */
int sntc_noisy_normal(const float* y_tilde, const float* hyper, int n, int64_t hw, int c, float* dbits_dv,
                      float* dbits_draw, double* bits, void* stream);
/* grad_record (float [sntc_prior_record_floats()], may be NULL): sum over all elements of d bits / d (softplus(matrix),
 * bias, tanh(factor)) per channel, OVERWRITTEN; turn into raw-variable gradients with sntc_prior_param_grad. */
int sntc_noisy_factorized(const sntc_prior* prior, const float* z_tilde, int n, int64_t hw, float* dbits_dz,
                          float* grad_record, double* bits, void* stream);
int sntc_prior_record_floats(const sntc_prior* prior);
/* Refresh the prior from DEVICE arrays of the raw variables (layouts of sntc_prior_create). */
int sntc_prior_update(sntc_prior* prior, const float* matrices, const float* biases, const float* factors, void* stream);
int sntc_prior_param_grad(const sntc_prior* prior, const float* matrices, const float* factors, const float* grad_record,
                          float weight, float* g_matrices, float* g_biases, float* g_factors, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training of the GDN / SignalConv2D stacks (MBT2018*, BLS2017*: reference common/transforms.py:93-175 under
 * Model.train_step, mshyper/models.py:375-383).  GDN with alpha = 1, epsilon = 1 (the reference's GDN1, :8-63):
 * norm = beta + |x| gamma comes from a 1x1 convolution plan with the |x| prologue; these are the element-wise parts.
 *   sntc_gdn_apply:            y = x / norm (inverse: x * norm)
 *   sntc_gdn_backward_prep:    q = d loss / d norm = -g x / norm^2 (inverse: g x);  abs_x = |x|
 *   sntc_gdn_backward_finish:  dx = g / norm (inverse: g * norm) + sign(x) * t,  t = q gamma^T from the adjoint 1x1 plan
 * d beta = column sums of q (sntc_bias_grad), d gamma = |x|^T q (sntc_conv_wgrad, 1x1).
 * ------------------------------------------------------------------------------------------ */
int sntc_gdn_apply(const float* x, const float* norm, int64_t total, int inverse, float* y, void* stream);
int sntc_gdn_backward_prep(const float* g, const float* x, const float* norm, int64_t total, int inverse, float* q, float* abs_x,
                           void* stream);
int sntc_gdn_backward_finish(const float* g, const float* x, const float* norm, const float* t, int64_t total, int inverse,
                             float* dx, void* stream);
/* out[rows, cols] = a[rows, k] b[k, cols] (transpose_a: a is [k, rows]) for a small left matrix (<= 48 KB): the
 * tfc.RDFTParameter of SignalConv2D kernels (transforms.py:101-112), kernel = M rdft and d rdft = M^T d kernel. */
int sntc_small_matmul(const float* a, const float* b, int rows, int k, int64_t cols, int transpose_a, float* out, void* stream);
/* dst[t, b, a] = src[t, a, b]: sntc_conv_wgrad of a SNTC_SIGNAL_UP layer returns [kh, kw, Cout, Cin]; its kernel is
 * [kh, kw, Cin, Cout] (tfc.SignalConv2D, transforms.py:123-134,172-175). */
int sntc_transpose_last2(const float* src, int taps, int a, int b, float* dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SNTC_H_ */
