"""Layer nodes the transforms are assembled from; each node owns its packed device plans.

A node has
    shapes(cin) -> (OrderedDict name -> shape, cout)      variable inventory in Keras / TFC layouts
    build(w, cin) -> cout                                 create device plans from device weights w
    __call__(x) -> y                                      launch
Fusions: bias + activation in every conv epilogue; the ResidualBlock skip (elic.py:66-68) and the
SimpleAttention gate (elic.py:97-100) ride on the last 1x1 conv's epilogue; GDN's norm pool is a
1x1 gather-GEMM with |x| (or x^2) applied while staging and x / norm in the epilogue.
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from .. import _capi as capi
from .. import ops


S3_MIN_ROWS = 256      # a layer runs on the pre-split bf16 x 3 kernel when one IMAGE offers at least one 256-row strip of its GEMM ...
S3_MIN_TILES = 12      # ... and at least this many 256 x 128 tiles: the kernel runs one workgroup per CU, and a layer that offers a
                       # batch only a few dozen tiles leaves most CUs idle (5x5/2 320 -> 320 at 1/32 resolution: 6 tiles per image,
                       # 0.52 ms against the fp32 kernel's 0.32 ms on 18 images; the hyper-synthesis layers offer 30)


def s3_eligible(kind, cin, cout, epilogue):
    """Static part of the rule: what csrc/bf3_gemm.hip can run (include/sntc.h, sntc_conv_desc.reserved[1] == 2)."""
    return cin % 16 == 0 and cout % 4 == 0 and epilogue in (capi.EPI_STORE, capi.EPI_ADD, capi.EPI_GATE, capi.EPI_MASK_RELU,
                                                              capi.EPI_MASK_LEAKY)


def s3_rows(kind, stride, h, w):
    """Rows of the layer's GEMM per image (the macro-pixel grid): input pixels of a transposed layer, output pixels of a conv."""
    if kind in ("convT", "sigup"):
        return h * w
    return (-(-h // stride)) * (-(-w // stride))


def s3_geometry_ok(kind, stride, cout, h, w):
    """Geometry part of the rule -- a function of the LAYER and of ONE image's size, never of the batch: at least one 256-row
    strip per image and at least S3_MIN_TILES 256 x 128 tiles per image (GEMM columns: Cout, times stride^2 output phases for the
    transposed kinds)."""
    rows = s3_rows(kind, stride, h, w)
    cols = cout * (stride * stride if kind in ("convT", "sigup") else 1)
    return rows >= S3_MIN_ROWS and (-(-rows // 256)) * (-(-cols // 128)) >= S3_MIN_TILES


class DualPlan:
    """A convolution's fp32 plan and, under ``precision="bf16x3"``, its pre-split split-precision plan.  Which of the two
    runs is a function of the LAYER and of the per-image geometry only -- never of the batch size -- so an encoder and a
    decoder that see the same image size take the same arithmetic whatever their batching (the decoder must reproduce the
    encoder's mu / sigma bit for bit)."""

    def __init__(self, kind, weight, bias, stride, act=None, prologue=capi.PRO_NONE, epilogue=capi.EPI_STORE, precision="fp32"):
        self.kind, self.stride = kind, int(stride)
        self.fp32 = ops.ConvPlan(kind, weight, bias, stride, act, prologue, epilogue)
        self.s3 = None
        if precision == "bf16x3" and prologue == capi.PRO_NONE and s3_eligible(kind, self.fp32.cin, self.fp32.cout, epilogue):
            self.s3 = ops.ConvPlan(kind, weight, bias, stride, act, prologue, epilogue, bf16x3="presplit")
        self.cin, self.cout = self.fp32.cin, self.fp32.cout

    def takes_s3(self, h, w):
        return self.s3 is not None and s3_geometry_ok(self.kind, self.stride, self.cout, h, w)

    def __call__(self, x, res=None, aux=None):
        if self.takes_s3(x.shape[1], x.shape[2]):
            return self.s3(x if x.dtype == torch.bfloat16 else ops.split3(x), res, aux)
        return self.fp32(x, res, aux)

    # the fp32 plan's interface, for the callers that fuse / profile / shape-infer
    def __getattr__(self, name):
        return getattr(self.fp32, name)


class Conv:
    """kind: conv (Keras Conv2D SAME) | convT (Keras Conv2DTranspose SAME) | sigdown / sigup (tfc.SignalConv2D)."""

    precision = "fp32"

    def __init__(self, name, kind, cout, k, s, act=None, bias=True, epilogue=capi.EPI_STORE):
        self.name, self.kind, self.cout, self.k, self.s, self.act, self.bias = name, kind, cout, k, s, act, bias
        self.epilogue = epilogue
        self.plan = None

    def shapes(self, cin):
        d = OrderedDict()
        d[f"{self.name}/kernel"] = (self.k, self.k, self.cout, cin) if self.kind == "convT" else (self.k, self.k, cin, self.cout)
        if self.bias:
            d[f"{self.name}/bias"] = (self.cout,)
        return d, self.cout

    def build(self, w, cin):
        args = (self.kind, w[f"{self.name}/kernel"], w.get(f"{self.name}/bias") if self.bias else None, self.s, self.act,
                capi.PRO_NONE, self.epilogue)
        if (self.kind in ("conv", "sigdown") and self.epilogue == capi.EPI_STORE and ops.RGB_FIRST_LAYER
                and ops.RgbConvPlan.supported(self.k, self.s, cin, self.cout, self.act, self.kind)):
            # the RGB first layer on its own kernel: Keras Conv2D (same bits as the row-packed plan below) or tfc.SignalConv2D
            self.plan = ops.RgbConvPlan(args[1], args[2], self.s, self.act, self.kind)
        elif (self.act is None and self.epilogue == capi.EPI_STORE and ops.SMALL_OUTPUT_LAYER
                and ops.UpSmallPlan.supported(self.kind, self.k, self.s, cin, self.cout)):
            self.plan = ops.UpSmallPlan(self.kind, args[1], args[2], self.s)       # the syntheses' last layer (5x5/2 -> 3 channels) on the vector ALU
        elif (self.kind == "conv" and cin <= 4 and self.k > 1 and self.k * cin <= 16 and self.epilogue == capi.EPI_STORE
                and ops.ROW_PACKED_FIRST_LAYER):
            self.plan = ops.RowPackedConv(args[1], args[2], self.s, self.act)      # ... or as a row-packed gather-GEMM plan
        else:
            self.plan = DualPlan(*args, precision=self.precision) if self.precision != "fp32" else ops.ConvPlan(*args)
        return self.cout

    def __call__(self, x, res=None, aux=None):
        return self.plan(x, res, aux)

    def out_hw(self, h, w):
        if self.kind in ("convT", "sigup"):
            return h * self.s, w * self.s
        return -(-h // self.s), -(-w // self.s)

    def flops(self, n, h, w):
        return self.plan.flops(n, h, w), self.plan.out_hw(h, w)


def set_precision(node, precision):
    """Mark every convolution and ResidualBlock under ``node`` (before it is built) with the arithmetic it should prefer."""
    if isinstance(node, (Conv, ResidualBlock, SimpleAttention)):
        node.precision = precision
    for child in getattr(node, "layers", ()) or ():
        set_precision(child, precision)


class GDN:
    """GDN1 (transforms.py:8-63) or tfc.GDN with fixed alpha in {1,2}, epsilon in {1, .5}."""

    def __init__(self, name, inverse=False, alpha=1, epsilon=1.0):
        if alpha not in (1, 2) or epsilon not in (1, 1.0, 0.5):
            raise NotImplementedError(f"GDN alpha={alpha} epsilon={epsilon}")
        self.name, self.inverse, self.alpha, self.epsilon = name, inverse, alpha, float(epsilon)
        self.plan = None
        self.beta = self.gamma = None

    def shapes(self, cin):
        return OrderedDict([(f"{self.name}/beta", (cin,)), (f"{self.name}/gamma", (cin, cin))]), cin

    def build(self, w, cin):
        self.beta, self.gamma = w[f"{self.name}/beta"], w[f"{self.name}/gamma"]
        self.c = cin
        if cin not in ops.GDN_SMALL_CHANNELS:
            pro = capi.PRO_ABS if self.alpha == 1 else capi.PRO_SQUARE
            if self.epsilon == 0.5:
                epi = capi.EPI_RES_MUL_SQRT if self.inverse else capi.EPI_RES_DIV_SQRT
            else:
                epi = capi.EPI_RES_MUL if self.inverse else capi.EPI_RES_DIV
            # gamma[in, out] is exactly a 1x1 HWIO kernel; beta is its bias
            self.plan = ops.ConvPlan("conv", self.gamma.reshape(1, 1, cin, cin), self.beta, 1, None, pro, epi)
        return cin

    def __call__(self, x):
        if self.plan is not None:
            return self.plan(x, res=x)
        return ops.gdn_small(x, self.beta, self.gamma, self.inverse, self.alpha, self.epsilon)

    def out_hw(self, h, w):
        return h, w


class Seq:
    def __init__(self, layers):
        self.layers = list(layers)

    def shapes(self, cin):
        d = OrderedDict()
        for l in self.layers:
            s, cin = l.shapes(cin)
            for k, v in s.items():
                if k in d and d[k] != v:
                    raise ValueError(f"shared variable {k} used with shapes {d[k]} and {v}")
                d[k] = v
        return d, cin

    def build(self, w, cin):
        for l in self.layers:
            cin = l.build(w, cin)
        return cin

    def __call__(self, x):
        for l in self.layers:
            x = l(x)
        return x

    def out_hw(self, h, w):
        for l in self.layers:
            h, w = l.out_hw(h, w)
        return h, w


class ResidualBlock:
    """elic.py:41-68."""

    precision = "fp32"        # "bf16x3" (set_precision): the whole-block kernel in split precision where it exists (c = 192)

    def __init__(self, name):
        self.name = name
        self._convs = None
        self._block = None

    def _mk(self, c):
        n = self.name
        return [Conv(f"{n}/conv0", "conv", c // 2, 1, 1, "relu"), Conv(f"{n}/conv1", "conv", c // 2, 3, 1, "relu"),
                Conv(f"{n}/conv2", "conv", c, 1, 1, None, epilogue=capi.EPI_ADD)]

    def shapes(self, cin):
        return Seq(self._mk(cin)).shapes(cin)

    def build(self, w, cin):
        self._convs = self._mk(cin)
        c = cin
        for l in self._convs:
            c = l.build(w, c)
        # the whole block as one launch (csrc/rb_fused.hip), bit-identical to the three layers below
        self._block = None
        if (ops.FUSE_RESIDUAL_BLOCK or self.precision != "fp32") and ops.ResBlockPlan.supported(cin):
            n = self.name
            self._block = ops.ResBlockPlan(w[f"{n}/conv0/kernel"], w.get(f"{n}/conv0/bias"), w[f"{n}/conv1/kernel"],
                                           w.get(f"{n}/conv1/bias"), w[f"{n}/conv2/kernel"], w.get(f"{n}/conv2/bias"),
                                           precision=self.precision)
        return cin

    def __call__(self, x):
        a, b, c = self._convs
        if self.precision != "fp32":
            # split-precision model: which arithmetic a layer takes may depend on the layer and on ONE image's geometry, never on
            # the batch (DualPlan).  c = 192: ALWAYS the split-precision block (csrc/rb_fused_bf3.hip), whatever the launch size;
            # other widths: the three fp32 layers (bit-identical for any batch)
            if self._block is not None and x.dtype == torch.float32:
                return self._block(x)
            return c(b(a(x)), res=x)
        if self._block is not None and ops.FUSE_RESIDUAL_BLOCK and ops.ResBlockPlan.tiles(*x.shape[:3]) >= ops.FUSED_BLOCK_MIN_TILES:
            return self._block(x)
        # c = 192: the 3x3 and the 1x1 + skip run as one launch -- where the launch has the rows to fill the device: the fused
        # instance's 128-row workgroups run both contractions back to back, and below ~192 of them the two stand-alone launches
        # (64 x 64 tiles, deep ring) are faster (6144 rows: 0.038 against 0.075 ms; 24576: equal; bit-identical either way)
        if x.shape[0] * x.shape[1] * x.shape[2] >= ops.FUSED_TAIL_MIN_ROWS and b.plan.fusable_with(c.plan):
            return b.plan.fused(c.plan, a(x), res=x)
        return c(b(a(x)), res=x)

    def out_hw(self, h, w):
        return h, w


class SimpleAttention:
    """elic.py:71-100: x + trunk(x) * sigmoid(conv1x1(branch(x)))."""

    def __init__(self, name):
        self.name = name
        self._trunk = self._branch = self._gate = None

    def _mk(self, c):
        n = self.name
        trunk = [ResidualBlock(f"{n}/trunk/rb{i}") for i in range(3)]
        branch = [ResidualBlock(f"{n}/branch/rb{i}") for i in range(3)]
        gate = Conv(f"{n}/branch/conv", "conv", c, 1, 1, "sigmoid", epilogue=capi.EPI_GATE)
        return trunk, branch, gate

    def shapes(self, cin):
        t, b, g = self._mk(cin)
        return Seq(t + b + [g]).shapes(cin)

    precision = "fp32"

    def build(self, w, cin):
        self._trunk, self._branch, self._gate = self._mk(cin)
        for l in self._trunk + self._branch:
            l.precision = self.precision
        for l in self._trunk + self._branch + [self._gate]:
            l.build(w, cin)
        return cin

    def __call__(self, x):
        # trunk and branch are independent until the gate: where one launch leaves most of the device idle (an image alone at
        # 1/4 or 1/16 resolution offers 72 - 384 workgroups to 256 CUs) they run on two streams at once -- same kernels, same bits
        side = None
        if ops.CONCURRENT_BRANCHES and ops.ResBlockPlan.tiles(*x.shape[:3]) < ops.CONCURRENT_BRANCH_MAX_TILES:
            side = ops.companion_stream()
        if side is None:
            t = x
            for l in self._trunk:
                t = l(t)
            b = x
            for l in self._branch:
                b = l(b)
            return self._gate(b, res=x, aux=t)
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            b = x
            for l in self._branch:
                b = l(b)
        t = x
        for l in self._trunk:
            t = l(t)
        cur.wait_stream(side)
        b.record_stream(cur)              # allocated on the side stream, consumed here: the allocator must not hand it out early
        x.record_stream(side)
        return self._gate(b, res=x, aux=t)

    def out_hw(self, h, w):
        return h, w
